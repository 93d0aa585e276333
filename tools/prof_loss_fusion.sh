#!/bin/bash
# rocprofv3 kernel stats of the self-supervised loss and the fusion filter (kernels only, no CPU oracle).  Logs -> gpurun_out/.
set -u
exec < /dev/null
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$R/gpurun_out"
export TMPDIR=/tmp
cd /tmp
timeout 60 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_loss" -o ls -- python "$R/tools/loss_bench.py" > "$R/gpurun_out/prof_loss.log" 2>&1
timeout 80 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_fuse" -o fu -- python "$R/tools/fusion_bench.py" > "$R/gpurun_out/prof_fuse.log" 2>&1
cd "$R"
for d in prof_loss prof_fuse; do
    tail -n 4 "gpurun_out/$d.log" | cut -c1-200
    f=$(find "gpurun_out/$d" -name "*kernel_stats.csv" 2>/dev/null | head -n 1)
    if [ -n "$f" ] && [ -f "$f" ]; then head -n 14 "$f" | cut -c1-220; fi
    find "gpurun_out/$d" -name "*kernel_trace.csv" -delete 2>/dev/null
done
exit 0
