#!/usr/bin/env python3
"""VGPRs / scratch of every kernel of a translation unit: hipcc ... -Rpass-analysis=kernel-resource-usage 2> remarks.txt ;
python tools/kernel_regs.py remarks.txt [--scratch-only]"""
import re, sys
cur, rows = None, []
for line in open(sys.argv[1]):
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("sgpr", r"TotalSGPRs: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None:
            cur[key] = int(m.group(1))
only = "--scratch-only" in sys.argv
for r in rows:
    if only and not r.get("scratch"):
        continue
    print(f"{r.get('vgpr', 0):4d} vgpr {r.get('agpr', 0):4d} agpr {r.get('sgpr', 0):4d} sgpr {r.get('scratch', 0):5d} B scratch  occ {r.get('occ', 0)}  {r['name'][:110]}")
