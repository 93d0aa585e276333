#!/bin/bash
# Build rc_mvsnet_amd-compatible library variants for the two-stream investigation (run in the build container before `gpurun`;
# the .so files travel with the snapshot): tools/dev/_variants/lib_plain.so = the product library with PLAIN loads of the previous
# stage's depth map in the hypothesis-planes kernel (the pre-fix behaviour, profiles/r3_two_streams.txt (i)-(o)).
set -e
cd "$(dirname "$0")/../.."
python -c "import __graft_entry__ as g; g.build()" > /dev/null
mkdir -p tools/dev/_variants
OBJS=$(ls rc_mvsnet_amd/_obj/*.o | grep -v "geometry.o")
sed 's/__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)/\*p/' rc_mvsnet_amd/csrc/geometry.hip > /tmp/geometry_plain.hip
grep -q "return \*p;" /tmp/geometry_plain.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I rc_mvsnet_amd/csrc -I include -c /tmp/geometry_plain.hip -o /tmp/geometry_plain.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/dev/_variants/lib_plain.so $OBJS /tmp/geometry_plain.o
echo built tools/dev/_variants/lib_plain.so
