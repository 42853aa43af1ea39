import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import torch

    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = torch.load(os.path.join(GOLDEN, name), weights_only=False)
        return cache[name]

    return load


# ---- the row-panel engine (csrc/panel.h) is opt-in (MTLORA_PNL=1) and then only takes launches with >= 32768 rows; every
# GPU test that runs an MTLoRALinear in bf16 is ALSO run with it switched on and the threshold at 1 row ("[panel]"
# variants), so the golden / oracle cases cover it.  (The library reads both variables at every call.)
_PANEL_TESTS = ("linear", "mlp", "swin_block", "backbone", "config_model", "module_golden", "c1_reference", "task_streams",
                "reducer_on_the_real_model")


@pytest.fixture(autouse=True, params=["auto", "panel"])
def _panel_engine(request, monkeypatch):
    if request.param == "panel":
        monkeypatch.setenv("MTLORA_PNL", "1")
        monkeypatch.setenv("MTLORA_PNL_MIN_M", "1")
    else:
        monkeypatch.delenv("MTLORA_PNL", raising=False)
        monkeypatch.delenv("MTLORA_PNL_MIN_M", raising=False)
    yield


def pytest_collection_modifyitems(config, items):
    keep = []
    for it in items:
        if "[panel" in it.name or "-panel]" in it.name:
            is_gpu = it.get_closest_marker("gpu") is not None
            wants = any(k in it.name for k in _PANEL_TESTS)
            import torch
            vals = list(getattr(getattr(it, "callspec", None), "params", {}).values())
            fp32_only = ("fp32" in it.name) or ((torch.float32 in vals or torch.float16 in vals) and torch.bfloat16 not in vals)  # the engine is bf16-only
            if not (is_gpu and wants) or fp32_only:
                continue
        keep.append(it)
    items[:] = keep
