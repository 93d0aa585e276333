#!/bin/bash
# Build library variants for the two-stream investigation (run in the build container before `gpurun`; the .so files travel with the
# snapshot; select one with RCMVS_LIB=tools/dev/_variants/lib_<name>.so in tools/dev/two_stream_*.py).  All differ from the product
# library in the hypothesis-planes kernel only (csrc/geometry.hip), whose loads of the previous stage's depth map are
#   plain  plain loads                                  -- the pre-fix behaviour (profiles/r3_two_streams.txt (i)-(o): ~7 % of the scenes wrong)
#   acq    plain loads after a one-lane agent-scope acquire + __syncthreads() at the top of the kernel (L1 invalidated: should be clean)
#   sc0    workgroup-scope loads (sc0: hit the L1 like plain loads by the hardware guide: should fail like `plain`)
# the product library reads them at agent scope (sc1: bypass the L1).
set -e
cd "$(dirname "$0")/../.."
python -c "import __graft_entry__ as g; g.build()" > /dev/null
mkdir -p tools/dev/_variants
OBJS=$(ls rc_mvsnet_amd/_obj/*.o | grep -v "geometry.o")
sed 's/__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)/\*p/' rc_mvsnet_amd/csrc/geometry.hip > /tmp/geometry_plain.hip
grep -q "return \*p;" /tmp/geometry_plain.hip
sed 's/__HIP_MEMORY_SCOPE_AGENT/__HIP_MEMORY_SCOPE_WORKGROUP/' rc_mvsnet_amd/csrc/geometry.hip > /tmp/geometry_sc0.hip
build() {   # name, source, extra flags
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I rc_mvsnet_amd/csrc -I include $3 -c $2 -o /tmp/geometry_$1.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/dev/_variants/lib_$1.so $OBJS /tmp/geometry_$1.o
    echo built tools/dev/_variants/lib_$1.so
}
build plain /tmp/geometry_plain.hip ""
build acq /tmp/geometry_plain.hip "-DRCMVS_EXP_PLANES_ACQUIRE"
build sc0 /tmp/geometry_sc0.hip ""
