#!/usr/bin/env python
"""Print per-kernel register / scratch / LDS usage of a HIP source (hipcc -Rpass-analysis)."""
import re, subprocess, sys
for src in sys.argv[1:]:
    out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "-Rpass-analysis=kernel-resource-usage",
                          "-o", "/dev/null", src], capture_output=True, text=True).stderr
    cur = {}
    for line in out.splitlines():
        m = re.search(r"remark: +(Function Name|VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (.*?) \[-R", line)
        if not m: continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()[:90]}
        else:
            cur[k.split(" ")[0]] = v
        if k.startswith("LDS"):
            print(f"{cur['name']:<92} vgpr={cur.get('VGPRs')} agpr={cur.get('AGPRs')} sgpr={cur.get('SGPRs')} scratch={cur.get('ScratchSize')} occ={cur.get('Occupancy')} lds={cur.get('LDS')}")
