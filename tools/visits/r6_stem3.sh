#!/bin/bash
# kernel duration of conv2d_stem_kernel for each variant library (rocprofv3 --kernel-trace --stats over tools/dev/stem_time.py)
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
: > gpurun_out/r6_stem3.log
for lib in "$@"; do
  ( cd /tmp && export TMPDIR=/tmp && RCMVS_LIB=$([ $lib = product ] && echo product || echo $R/$lib) timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/stem3_prof -o s -- python $R/tools/dev/stem_time.py 2>&1 | grep checksum )
  f=$(find gpurun_out/stem3_prof -name "*kernel_stats.csv" | head -1)
  echo "$lib: $(grep conv2d_stem_kernel $f | awk -F, '{print $(NF-5), "calls", $(NF-4), "total ns; avg", $(NF-3), "min", $(NF-1)}')" | tee -a gpurun_out/r6_stem3.log
  grep conv2d_stem_kernel $f | tail -c 120 >> gpurun_out/r6_stem3.log
  rm -rf gpurun_out/stem3_prof
done
