#!/bin/bash
# Timing variants of conv0's z-streaming kernel (csrc/conv3d_z8.hip, Z8_EXP bit mask): tools/dev/_v/lib_z8_<mask>.so; select with RCMVS_LIB (tools/dev/layer_times.py).
set -e
cd "$(dirname "$0")/../.."
python -c "import __graft_entry__ as g; g.build()" > /dev/null
mkdir -p tools/dev/_v
OBJS=$(ls rc_mvsnet_amd/_obj/*.o | grep -v "/conv3d_z8.o")
for m in "$@"; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DZ8_EXP=$m -c rc_mvsnet_amd/csrc/conv3d_z8.hip -o /tmp/z8_$m.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/dev/_v/lib_z8_$m.so $OBJS /tmp/z8_$m.o
    echo built tools/dev/_v/lib_z8_$m.so
done
