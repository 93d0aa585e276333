#!/bin/bash
# One-source variant builds of the library: tools/dev/build_variant.sh <source stem> <tag> <extra hipcc flags...> -> tools/dev/_v/lib_<stem>_<tag>.so (select with RCMVS_LIB
# in the tools/dev timing scripts).  Built in the container: tools/dev/_v/ travels to the GPU box (tools/dev/_variants/ does not).
set -e
cd "$(dirname "$0")/../.."
src=$1; tag=$2; shift 2
python -c "import __graft_entry__ as g; g.build()" > /dev/null
mkdir -p tools/dev/_v
OBJS=$(ls rc_mvsnet_amd/_obj/*.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c rc_mvsnet_amd/csrc/$src.hip -o /tmp/${src}_$tag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/dev/_v/lib_${src}_$tag.so $OBJS /tmp/${src}_$tag.o
echo built tools/dev/_v/lib_${src}_$tag.so
