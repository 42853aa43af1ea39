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


# "dense": every launch that is ELIGIBLE for one of the shape-selected kernels takes it, whatever the size heuristics say (k_ntd for
# the single-output GEMMs, k_sp_tn for the factor gradients, k_sp_projk for the P / Q passes with large K R): the test shapes are
# far below the sizes at which the library picks them on its own
_DENSE_TESTS = ("linear", "mlp", "swin_block")


@pytest.fixture(autouse=True, params=["auto", "tiled", "dense"])
def _kernel_family(request, monkeypatch):
    for k in ("MTLORA_SP", "MTLORA_NTD", "MTLORA_SP_TN", "MTLORA_SP_PROJK"):
        monkeypatch.delenv(k, raising=False)
    if request.param == "tiled":
        monkeypatch.setenv("MTLORA_SP", "0")
        monkeypatch.setenv("MTLORA_NTD", "0")
    elif request.param == "dense":
        monkeypatch.setenv("MTLORA_NTD", "2")
        monkeypatch.setenv("MTLORA_SP_TN", "2")
        monkeypatch.setenv("MTLORA_SP_PROJK", "2")
    yield


def pytest_collection_modifyitems(config, items):
    keep = []
    for it in items:
        variant = next((v for v in ("tiled", "dense") if f"[{v}" in it.name or f"-{v}]" in it.name), None)
        if variant:
            is_gpu = it.get_closest_marker("gpu") is not None
            wants = any(k in it.name for k in (_TILED_TESTS if variant == "tiled" else _DENSE_TESTS))
            import torch
            vals = list(getattr(getattr(it, "callspec", None), "params", {}).values())
            fp32_only = ("fp32" in it.name) or (torch.float32 in vals and torch.bfloat16 not in vals and torch.float16 not in vals)
            if not (is_gpu and wants) or fp32_only:  # (the selected kernels are 16-bit only: fp32 cases have one path)
                continue
        keep.append(it)
    items[:] = keep
