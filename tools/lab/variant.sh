#!/bin/bash
# Builds an A/B variant of libokvfe.so that differs in ONE kernel source:
#   bash tools/lab/variant.sh <name> <k_file.hip> "<extra hipcc flags / -D...>" [sed-expression on the source]
#   -> okvis2_amd/libokvfe_<name>.so   (use with OKVFE_LIB=$PWD/okvis2_amd/libokvfe_<name>.so)
set -e
name=$1; file=$2; defs=$3; sedx=${4:-}
cd $(dirname $0)/../../okvis2_amd/csrc
make -s
src=$file
if [ -n "$sedx" ]; then sed "$sedx" $file > /tmp/variant_$name.hip; src=/tmp/variant_$name.hip; fi
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -I. -I../../include"
mkdir -p /tmp/variant_obj
/opt/rocm/bin/hipcc $FLAGS $defs -x hip -c $src -o /tmp/variant_obj/$name.o
objs=$(ls build/*.hip.o build/*.cpp.o | grep -v "build/$file.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libokvfe_$name.so /tmp/variant_obj/$name.o $objs
echo built libokvfe_$name.so
