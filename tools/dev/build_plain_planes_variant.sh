#!/bin/bash
# Build library variants for the two-stream investigation (run in the build container before `gpurun`; the .so files travel with the
# snapshot; select one with RCMVS_LIB=tools/dev/_variants/lib_<name>.so in tools/dev/two_stream_*.py).  All differ from the product
# library in the hypothesis-planes kernel only (csrc/geometry.hip), whose loads of the previous stage's depth map are
#   plain  plain loads                                  -- the pre-fix behaviour (profiles/r3_two_streams.txt (i)-(o): ~7 % of the scenes wrong)
#   acq    plain loads after a one-lane agent-scope acquire + __syncthreads() at the top of the kernel (L1 invalidated: should be clean)
#   sc0    workgroup-scope loads (sc0: hit the L1 like plain loads by the hardware guide: should fail like `plain`)
# the product library reads them at agent scope (sc1: bypass the L1).  And the candidate GENERAL protection of the two-stream mode:
#   l1acq        every kernel of the inference path starts with a one-lane agent-scope acquire (-DRCMVS_L1_ACQUIRE, csrc/common.h:
#                RCMVS_KERNEL_ENTRY), planes kernel as in the product            -- for its cost on the one-stream throughput
#   l1acq_plain  the same with PLAIN loads in the planes kernel                  -- should be clean if the L1 reading of DESIGN.md section 8 holds
set -e
cd "$(dirname "$0")/../.."
python -c "import __graft_entry__ as g; g.build()" > /dev/null
mkdir -p tools/dev/_variants
T=$(mktemp -d)          # (an empty directory: the sources include "common.h" by quote, which looks next to the file first)
OBJS=$(ls rc_mvsnet_amd/_obj/*.o | grep -v "geometry.o")
sed 's/__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)/\*p/' rc_mvsnet_amd/csrc/geometry.hip > $T/geometry_plain.hip
grep -q "return \*p;" $T/geometry_plain.hip
sed 's/__HIP_MEMORY_SCOPE_AGENT/__HIP_MEMORY_SCOPE_WORKGROUP/' rc_mvsnet_amd/csrc/geometry.hip > $T/geometry_sc0.hip
build() {   # name, source, extra flags
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I rc_mvsnet_amd/csrc -I include $3 -c $2 -o $T/geometry_$1.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/dev/_variants/lib_$1.so $OBJS $T/geometry_$1.o
    echo built tools/dev/_variants/lib_$1.so
}
INFER="absmax conv2d conv3d conv3d_lds conv3d_mfma conv3d_x3 deconv3d_lds depth_head fpn_fused warp_variance"
mkdir -p $T/l1acq_obj
for f in $INFER; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include -DRCMVS_L1_ACQUIRE -c rc_mvsnet_amd/csrc/$f.hip -o $T/l1acq_obj/$f.o 2> /dev/null &
done
wait
REST=$(for o in rc_mvsnet_amd/_obj/*.o; do b=$(basename $o .o); case " $INFER geometry " in *" $b "*) ;; *) echo $o;; esac; done)
for v in "l1acq rc_mvsnet_amd/csrc/geometry.hip" "l1acq_plain $T/geometry_plain.hip"; do
    set -- $v
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I rc_mvsnet_amd/csrc -I include -DRCMVS_L1_ACQUIRE -c $2 -o $T/l1acq_obj/geometry_$1.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/dev/_variants/lib_$1.so $REST $T/l1acq_obj/[a-fh-z]*.o $T/l1acq_obj/geometry_$1.o
    echo built tools/dev/_variants/lib_$1.so
done
build plain $T/geometry_plain.hip ""
build acq $T/geometry_plain.hip "-DRCMVS_EXP_PLANES_ACQUIRE"
build sc0 $T/geometry_sc0.hip ""
