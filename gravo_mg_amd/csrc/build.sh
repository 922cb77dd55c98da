#!/bin/bash
# Builds libgravomg_hip.so (gfx950) in-tree.  hipcc cross-compiles without a GPU.
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
out="${here}/../lib"
mkdir -p "${out}"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
"${HIPCC}" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function \
    -I"${here}/../../include" "${here}/engine.hip" -o "${out}/libgravomg_hip.so" -lpthread -ldl "$@"
echo "built ${out}/libgravomg_hip.so"
