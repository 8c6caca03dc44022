#!/bin/bash
# Dev tool: the same measurement under several builds of libhgr.so (HGR_LIB), processes interleaved on ONE box.
#   tools/lib_ab.sh <out dir> <rounds> <lib1> <lib2> ... -- <command ...>      (lib = path of a .so, or "tree" for the in-tree build)
out=$1; rounds=$2; shift 2
libs=()
while [ "$1" != "--" ]; do libs+=("$1"); shift; done
shift
mkdir -p "$out"
for r in $(seq 1 "$rounds"); do
  for lib in "${libs[@]}"; do
    tag=$(basename "$lib" .so)
    if [ "$lib" = "tree" ]; then env -u HGR_LIB "$@" > "$out/${tag}_r${r}.log" 2>&1
    else HGR_LIB="$PWD/$lib" "$@" > "$out/${tag}_r${r}.log" 2>&1; fi
    echo "$tag round $r: $(tail -1 "$out/${tag}_r${r}.log" | cut -c1-1500)"
  done
done
