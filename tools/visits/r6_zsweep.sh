#!/bin/bash
# tools/dev/z_sweep.py under rocprofv3 --kernel-trace: the kernels' own durations per (layer, depth) -- consecutive launches of one conv kernel form a group
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
( cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/zs_prof -o zs -- python $R/tools/dev/z_sweep.py > $R/gpurun_out/r6_z_sweep_events.txt 2>&1 )
t=$(find gpurun_out/zs_prof -name "*kernel_trace.csv" | head -1)
python - "$t" <<'PY' | tee gpurun_out/r6_z_sweep_kernels.txt
import csv, sys, statistics
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
groups = []
for r in rows:
    k = r["Kernel_Name"]
    if "conv3d_z" not in k and "conv3d_x3_kernel" not in k: continue
    name = k.split("(")[0].replace("void rcmvs::", "")
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    key = (name, r.get("Grid_Size_X", r.get("Grid_Size", "?")))
    if groups and groups[-1][0] == key: groups[-1][1].append(d)
    else: groups.append([key, [d]])
for key, ds in groups:             # (z_sweep.py: 5 + 5 x 40 launches per configuration, depths 1 2 4 8 16 32 in that order; depth 1 takes the planar kernel)
    for i in range(0, len(ds), 205):
        c = ds[i:i + 205]
        if len(c) >= 100: print(f"{key[0]:40s} grid {key[1]:>8s} launches {len(c):4d} median {statistics.median(c):7.1f} us  min {min(c):7.1f}")
PY
rm -rf gpurun_out/zs_prof
