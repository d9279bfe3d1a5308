#!/bin/bash
# An experiment build of the library next to the production one (runs anywhere, hipcc cross-compiles):
#   tools/build_variant.sh <name> <source.hip> [-DMACRO=value ...]   ->  build/variants/lib_<name>.so (git-ignored, outside the package)
# = the production objects with ONE source file recompiled under the given macros.  Select it with GFPP_LIB_PATH (tools/ab_lib.sh <tag>
# libgfpp_radnerf.so lib_<name>.so runs both on the same box; `GFPP_LIB_PATH=... python -m pytest tests -m gpu` runs the suite on it).
set -e
mkdir -p "$(dirname "$0")/../build/variants"
name=$1; src=$2; shift 2
cd "$(dirname "$0")/../genefaceplusplus_amd/csrc"
make -s
tmp=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-function "$@" -c "$src" -o "$tmp/variant.o"
objs=$(ls *.o | grep -v "^${src%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "../../build/variants/lib_${name}.so" $objs "$tmp/variant.o"
rm -rf "$tmp"
echo "built build/variants/lib_${name}.so ($src $*)"
