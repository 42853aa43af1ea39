#!/usr/bin/env python3
"""VGPRs / scratch / occupancy of every kernel in a .hip file (cross-compiles for gfx950; no GPU needed):
    python tools/kernel_regs.py mtlora_amd/csrc/linear.hip [name-filter]"""
import re, subprocess, sys, os
src = sys.argv[1]; filt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value",
                      "-Wno-unused-result", "-c", os.path.basename(src), "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"],
                     capture_output=True, text=True, cwd=os.path.dirname(os.path.abspath(src))).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1); rows[cur] = {}
        continue
    for key in ("VGPRs", "ScratchSize \[bytes/lane\]", "Occupancy \[waves/SIMD\]", "SGPRs Spill", "VGPRs Spill"):
        m = re.search(r"remark:\s+" + key + r": (\d+)", line)
        if m and cur:
            rows[cur][key.split(" ")[0] + ("Spill" if "Spill" in key else "")] = int(m.group(1))
for n, r in rows.items():
    if filt in n:
        d = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
        print(f"{r.get('VGPRs', '?'):>4} vgpr  scratch {r.get('ScratchSize', '?'):>3}  occ {r.get('Occupancy', '?')}  sgprspill {r.get('SGPRsSpill', '?'):>3}  {d[:110]}")
