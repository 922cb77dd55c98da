#!/bin/bash
# Experimental builds of libgravomg_hip.so with -D flags, side by side (A/B through GMG_LIB_PATH):
#   bash scripts/build_variants.sh name1 "-DX=1" name2 "-DX=2" ...   ->  gravo_mg_amd/lib/variants/libgmg_<name>.so
here="$(cd "$(dirname "$0")" && pwd)"; src="$here/../gravo_mg_amd/csrc"; out="$here/../gravo_mg_amd/lib/variants"; mkdir -p "$out"
while [ $# -ge 2 ]; do
  n=$1; f=$2; shift; shift
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -w $f -I"$here/../include" "$src/engine.hip" -o "$out/libgmg_$n.so" -lpthread 2>&1 | grep -E "error" ) &
done
wait; ls -la "$out"
