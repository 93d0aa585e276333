#!/bin/bash
# Build the POSITIVE CONTROL of the two-stream hazard (run in the build container before `gpurun`; the .so travels with the snapshot;
# select it with RCMVS_LIB=tools/dev/_variants/lib_plain.so in tools/dev/two_stream_depth.py): the product library with PLAIN loads of the
# previous stage's depth map in the hypothesis-planes kernel (csrc/geometry.hip: ld_agent) -- the pre-fix behaviour, ~7 % of the scenes wrong
# with two scenes in flight (profiles/r3_two_streams.txt, profiles/r4_two_streams_ab.txt).
set -e
cd "$(dirname "$0")/../.."
python -c "import __graft_entry__ as g; g.build()" > /dev/null
mkdir -p tools/dev/_variants
T=$(mktemp -d)          # (an empty directory: the sources include "common.h" by quote, which looks next to the file first)
OBJS=$(ls rc_mvsnet_amd/_obj/*.o | grep -v "geometry.o")
sed 's/__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)/\*p/' rc_mvsnet_amd/csrc/geometry.hip > $T/geometry_plain.hip
grep -q "return \*p;" $T/geometry_plain.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I rc_mvsnet_amd/csrc -I include -c $T/geometry_plain.hip -o $T/geometry_plain.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/dev/_variants/lib_plain.so $OBJS $T/geometry_plain.o
echo built tools/dev/_variants/lib_plain.so
