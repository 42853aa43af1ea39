#!/usr/bin/env python3
"""profiles/rNN_pmc_traffic[_cfg].json from the two PMC passes of tools/pmc.sh (FETCH_SIZE and WRITE_SIZE, KB, summed per kernel):
    python tools/pmc_traffic.py gpurun_out/pmc_fetch.csv gpurun_out/pmc_write.csv [steps=2] [fetch.csv write.csv names to cite] > profiles/...json
"k_nt" = every MTLoRALinear GEMM launch: the tiled kernels (k_nt / k_ntl / k_ntd) AND the wave-streaming ones (k_sp_proj / xres / ares /
projsum / projk) and k_rank_out; "k_tn" = the factor gradients (k_tn and k_sp_tn).
FETCH_SIZE is doubled (gfx950 note in MI355X_MICROARCH.md, HBM section: wide coalesced reads are tallied at half their bytes)."""
import csv, json, sys

GROUPS = {"k_nt": "k_nt", "k_tn": "k_tn", "k_sum": "k_sum", "k_attn_fwd": "k_attn_fwd", "k_attn_bwd": "k_attn_bwd",
          "k_ln_fwd": "k_ln_fwd", "k_ln_bwd": "k_ln_bwd", "k_residual": "k_residual", "k_bn": "k_bn_", "k_up_loss": "k_up_loss"}


def load(path, col):
    out = {}
    for r in csv.DictReader(open(path)):
        out[r["name"]] = (float(r["dispatches"]), float(r[col]))
    return out


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    steps = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) over `python bench.py "
                     "--steps 1 --warmup 1 --no-cpu-baseline --no-roofline` (tools/pmc.sh, tools/pmc_traffic.py); FETCH_SIZE (KB) "
                     "doubled per the gfx950 note of MI355X_MICROARCH.md (HBM section); WRITE_SIZE (KB) as reported",
           "files": sys.argv[4:6] if len(sys.argv) > 5 else ["profiles/r03_pmc_fetch.csv", "profiles/r03_pmc_write.csv"]}
    for g, key in GROUPS.items():
        def match(n):
            tn = "k_tn" in n or "k_sp_tn" in n  # factor-gradient kernels (tiled / wave-streaming) and their reduce kernels
            if g == "k_nt":
                return ("k_nt" in n or "k_sp_" in n or "k_rank_out" in n) and not tn
            if g == "k_tn":
                return tn and "reduce" not in n
            return key in n
        d = sum(v[0] for n, v in fetch.items() if match(n))
        if d == 0:
            continue
        fb = 2.0 * 1024.0 * sum(v[1] for n, v in fetch.items() if match(n))
        wb = 1024.0 * sum(v[1] for n, v in write.items() if match(n))
        res[g] = {"launches_per_step": d / steps, "fetch_bytes_per_step": fb / steps, "write_bytes_per_step": wb / steps,
                  "traffic_bytes_per_launch": (fb + wb) / d}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
