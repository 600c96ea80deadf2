#!/bin/bash
# Builds A/B variants of libokvfe.so that differ only in k_harris.hip macros:
#   bash tools/lab/k1variants.sh name1 "-DFOO=1 -DBAR" name2 "..."   -> okvis2_amd/libokvfe_<name>.so
set -e
cd $(dirname $0)/../../okvis2_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function"
while [ $# -ge 2 ]; do
  name=$1; defs=$2; shift 2
  /opt/rocm/bin/hipcc $FLAGS $defs -x hip -c k_harris.hip -o /tmp/k_harris_$name.o
  objs=$(ls build/*.hip.o build/*.cpp.o | grep -v k_harris)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libokvfe_$name.so /tmp/k_harris_$name.o $objs
  echo built libokvfe_$name.so
done
