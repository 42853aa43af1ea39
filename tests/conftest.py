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


# ---- two kernel families serve the MTLoRALinear launches: the wave-streaming kernels (csrc/stream.h: fused T = 0 forward / dX,
# P / Q passes -- the default wherever a shape is eligible) and the tiled kernels (k_nt / k_ntl: everything else).  Every GPU
# test that runs an MTLoRALinear in a 16-bit type is ALSO run with the streaming family switched off ("[tiled]" variants:
# MTLORA_SP=0, read by the library at every call), so the golden / oracle cases pin both the default path and the fallback.
_TILED_TESTS = ("linear", "mlp", "swin_block", "backbone", "config_model", "module_golden", "c1_reference", "task_streams",
                "reducer_on_the_real_model")


@pytest.fixture(autouse=True, params=["auto", "tiled"])
def _kernel_family(request, monkeypatch):
    if request.param == "tiled":
        monkeypatch.setenv("MTLORA_SP", "0")
    else:
        monkeypatch.delenv("MTLORA_SP", raising=False)
    yield


def pytest_collection_modifyitems(config, items):
    keep = []
    for it in items:
        if "[tiled" in it.name or "-tiled]" in it.name:
            is_gpu = it.get_closest_marker("gpu") is not None
            wants = any(k in it.name for k in _TILED_TESTS)
            import torch
            vals = list(getattr(getattr(it, "callspec", None), "params", {}).values())
            fp32_only = ("fp32" in it.name) or (torch.float32 in vals and torch.bfloat16 not in vals and torch.float16 not in vals)
            if not (is_gpu and wants) or fp32_only:  # (the streaming kernels are 16-bit only: fp32 cases have one path)
                continue
        keep.append(it)
    items[:] = keep
