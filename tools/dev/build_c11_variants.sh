#!/bin/bash
# Timing ablations of the fused conv11 + prob kernel (csrc/conv11_prob.hip, CP_ABL): tools/dev/_variants/lib_c11_<mask>.so; select with RCMVS_LIB (tools/dev/c11_time.py).
set -e
cd "$(dirname "$0")/../.."
python -c "import __graft_entry__ as g; g.build()" > /dev/null
mkdir -p tools/dev/_variants
OBJS=$(ls rc_mvsnet_amd/_obj/*.o | grep -v "/conv11_prob.o")
for m in "$@"; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DCP_ABL=$m $CP_EXTRA -c rc_mvsnet_amd/csrc/conv11_prob.hip -o /tmp/c11_$m.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/dev/_variants/lib_c11_$m.so $OBJS /tmp/c11_$m.o
    echo built tools/dev/_variants/lib_c11_$m.so
done
