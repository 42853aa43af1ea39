"""Build libmtlora_hip.so for gfx950 with bare hipcc (no torch / pybind dependency).

    python -m mtlora_amd.csrc.build [--force]

hipcc cross-compiles without a GPU.  The .so stays in-tree (git-ignored, shipped to the GPU box
with the working tree).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["window_process.hip", "linear.hip", "attention.hip", "glue.hip", "loss.hip", "upsample.hip", "reduce.hip", "selftest.hip", "block.hip", "hid.hip"]
LIB = os.path.join(HERE, "libmtlora_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-unused-result"]
if os.environ.get("MTLORA_ABLATE") == "1":  # developer build: MTLORA_NT_DBG ablation toggles in the NT kernels (use --force)
    FLAGS.append("-DMTL_NT_ABLATE=1")


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hdrs = [os.path.join(HERE, f) for f in sorted(os.listdir(HERE)) if f.endswith(".h")]  # common.h, panel.h, ...
    hdrs.append(os.path.join(HERE, "..", "..", "include", "mtlora_hip.h"))
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(HERE, src)
        o = os.path.join(HERE, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append([_hipcc()] + FLAGS + ["-c", s, "-o", o])
    if jobs:
        def run(cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True, cwd=HERE)
        with ThreadPoolExecutor(max_workers=4) as ex:
            list(ex.map(run, jobs))
    if jobs or force or _stale(LIB, objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=HERE)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
