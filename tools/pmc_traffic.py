#!/usr/bin/env python3
"""profiles/rNN_pmc_traffic[_cfg].json from the two PMC passes of tools/pmc.sh (FETCH_SIZE and WRITE_SIZE, KB, summed per kernel):
    python tools/pmc_traffic.py gpurun_out/pmc_fetch.csv gpurun_out/pmc_write.csv [steps=2] [fetch.csv write.csv names to cite] > profiles/...json
"k_nt" = every MTLoRALinear GEMM launch: the tiled kernels (k_nt / k_ntl / k_ntd) AND the wave-streaming ones (k_sp_proj / xres / ares /
projsum / projk) and k_rank_out; "k_tn" = the factor gradients (k_tn and k_sp_tn).
FETCH_SIZE is doubled (gfx950 note in MI355X_MICROARCH.md, HBM section: wide coalesced reads are tallied at half their bytes).

With KINDS=1 passes (tools/pmc.sh writes pmc_<tag>_dispatch.csv and pmc_<tag>_kinds.csv) the counters are also attributed PER
LAUNCH KIND of the library's profiler (k_nt:fwd_outputs / fwd_lowrank_P / bwd_lowrank_Q / bwd_dX = the hot path; k_nt:plain_* = the
callers' rank-0 GEMMs that run on the same kernels; k_tn:dA_dB; k_pack; k_tn_reduce; k_sum): the last step of such a run executes
under the profiler, whose record list is in issue order, one record per kernel dispatch -- the GEMM-family dispatches of that last
step are matched with the records one to one ("by_kind", "hot_path", "linear_path")."""
import csv, json, os, sys

GROUPS = {"k_nt": "k_nt", "k_tn": "k_tn", "k_sum": "k_sum", "k_attn_fwd": "k_attn_fwd", "k_attn_bwd": "k_attn_bwd",
          "k_ln_fwd": "k_ln_fwd", "k_ln_bwd": "k_ln_bwd", "k_residual": "k_residual", "k_bn": "k_bn_", "k_up_loss": "k_up_loss"}
HOT = ("k_nt:fwd_outputs", "k_nt:fwd_lowrank_P", "k_nt:bwd_lowrank_Q", "k_nt:bwd_dX")
LINEAR = HOT + ("k_tn:dA_dB", "k_pack", "k_tn_reduce", "k_sum")
# kernels that execute under a record of these kinds (exactly one dispatch per record)
KIND_KERNELS = {"k_nt:": ("k_nt", "k_sp_xres", "k_sp_ares", "k_sp_proj", "k_rank_out", "k_pq", "k_hid_fwd", "k_hid_bwd", "k_hid_proj"), "k_tn:dA_dB": ("k_tnI", "k_tn<", "k_sp_tnI", "k_sp_tn<"),
                "k_tn:plain_dW": ("k_tnI", "k_tn<"), "k_pack": ("k_pack",), "k_tn_reduce": ("k_tn_reduce", "k_sp_tn_reduce", "k_hid_reduce", "k_hid_rows_finish"), "k_sum": ("k_sum",)}


def load(path, col):
    out = {}
    for r in csv.DictReader(open(path)):
        out[r["name"]] = (float(r["dispatches"]), float(r[col]))
    return out


def lib_linear_kernel(n):
    return any(k in n for k in ("k_nt", "k_sp_", "k_rank_out", "k_tn", "k_pack", "k_sum", "k_pq", "k_hid_")) and "k_ln" not in n


def by_kind(dispatch_csv, kinds_csv, col, scale):
    """bytes per launch kind of the LAST profiled step: zip the record list with the trailing linear-family dispatches."""
    recs = [ln.rstrip("\n").split(",") for ln in open(kinds_csv)]
    recs = [r for r in recs if r and (r[0].startswith("k_nt:") or r[0] in ("k_tn:dA_dB", "k_tn:plain_dW", "k_pack", "k_tn_reduce", "k_sum"))]
    rows = [r for r in csv.DictReader(open(dispatch_csv)) if lib_linear_kernel(r["name"])]
    rows = rows[-len(recs):]
    if len(rows) != len(recs):
        raise SystemExit(f"{dispatch_csv}: {len(rows)} linear-family dispatches for {len(recs)} profiler records")
    out = {}
    for rec, row in zip(recs, rows):
        kind = rec[0]
        want = KIND_KERNELS["k_nt:" if kind.startswith("k_nt:") else kind]
        if not any(w in row["name"] for w in want):
            raise SystemExit(f"record {kind} ({rec[-1]}) does not line up with dispatch {row['dispatch']} {row['name']}")
        e = out.setdefault(kind, {"launches": 0, "bytes": 0.0, "s8d_bytes": 0.0, "launched_bytes": 0.0})
        e["launches"] += 1
        e["bytes"] += scale * float(row[col])
        e["launched_bytes"] += float(rec[1])
        e["s8d_bytes"] += float(rec[2])
    return out


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    steps = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) over `python bench.py "
                     "--steps 1 --warmup 1 --no-cpu-baseline [--no-roofline]` (tools/pmc.sh, tools/pmc_traffic.py); FETCH_SIZE (KB) "
                     "doubled per the gfx950 note of MI355X_MICROARCH.md (HBM section); WRITE_SIZE (KB) as reported",
           "files": sys.argv[4:6] if len(sys.argv) > 5 else [os.path.basename(sys.argv[1]), os.path.basename(sys.argv[2])]}
    for g, key in GROUPS.items():
        def match(n):
            tn = "k_tn" in n or "k_sp_tn" in n  # factor-gradient kernels (tiled / wave-streaming) and their reduce kernels
            if g == "k_nt":
                return ("k_nt" in n or "k_sp_" in n or "k_rank_out" in n or "k_pq" in n or "k_hid_fwd" in n or "k_hid_bwd" in n or "k_hid_proj" in n) and not tn
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
    fd, wd = sys.argv[1].replace(".csv", "_dispatch.csv"), sys.argv[2].replace(".csv", "_dispatch.csv")
    fk, wk = sys.argv[1].replace(".csv", "_kinds.csv"), sys.argv[2].replace(".csv", "_kinds.csv")
    if all(os.path.exists(p) for p in (fd, wd, fk, wk)):
        f, w = by_kind(fd, fk, "FETCH_SIZE", 2.0 * 1024.0), by_kind(wd, wk, "WRITE_SIZE", 1024.0)
        kinds = {}
        for k in f:
            kinds[k] = {"launches_per_step": f[k]["launches"], "fetch_bytes": f[k]["bytes"], "write_bytes": w[k]["bytes"],
                        "traffic_bytes": f[k]["bytes"] + w[k]["bytes"], "s8d_bytes": f[k]["s8d_bytes"], "launched_bytes": f[k]["launched_bytes"]}
        res["by_kind"] = kinds

        def tot(names):
            n = sum(kinds[k]["launches_per_step"] for k in names if k in kinds)
            t = sum(kinds[k]["traffic_bytes"] for k in names if k in kinds)
            a = sum(kinds[k]["s8d_bytes"] for k in names if k in kinds)
            return {"launches_per_step": n, "traffic_bytes_per_step": t, "s8d_bytes_per_step": a, "traffic_bytes_per_launch": t / max(n, 1),
                    "traffic_ratio": t / a if a else None}
        res["hot_path"] = tot(HOT)
        res["linear_path"] = tot(LINEAR)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
