#!/bin/bash
# Dev tool (round 6): variant libraries that differ from the in-tree build in hgr_gemm_duo.hip only (its epilogue switches), for
# same-box A/B runs through tools/lib_ab.sh.   tools/build_duo_variants.sh name1 "flags1" name2 "flags2" ...  ->  tools/_ab/<name>.so
set -e
cd "$(dirname "$0")/../hgr-net_amd/csrc"
make -j8 >/dev/null
others=$(ls build/*.o | grep -v hgr_gemm_duo.o)
mkdir -p ../../tools/_ab
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  (
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $flags -c hgr_gemm_duo.hip -o /tmp/duo_$name.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/_ab/$name.so $others /tmp/duo_$name.o
    echo "built $name ($flags)"
  ) &
done
wait
