#!/bin/bash
# Round 6 discriminator of the two-stream hazard: the product library with PLAIN stores of the variance volume in the K1 kernels (they are
# non-temporal in the product).  Run in the build container before `gpurun` (the .so travels with the snapshot); select it with
# RCMVS_LIB=tools/dev/_variants/lib_plain_store.so in tools/dev/two_stream_diag.py.
set -e
cd "$(dirname "$0")/../.."
python -c "import __graft_entry__ as g; g.build()" > /dev/null
mkdir -p tools/dev/_variants
T=$(mktemp -d)          # (the sources include their headers by quote, which looks next to the file first: patched copies win)
OBJS=$(ls rc_mvsnet_amd/_obj/*.o | grep -v "/warp_variance.o")
for f in warp_variance.hip k1_win.h k1_pp.h; do
    sed 's/__builtin_nontemporal_store(o, reinterpret_cast<v4f\*>(\([^;]*\)));/*reinterpret_cast<v4f*>(\1) = o;/' rc_mvsnet_amd/csrc/$f > $T/$f
done
! grep -q "__builtin_nontemporal_store" $T/warp_variance.hip $T/k1_win.h $T/k1_pp.h
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I rc_mvsnet_amd/csrc -I include -c $T/warp_variance.hip -o $T/warp_variance_plain.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/dev/_variants/lib_plain_store.so $OBJS $T/warp_variance_plain.o
echo built tools/dev/_variants/lib_plain_store.so
