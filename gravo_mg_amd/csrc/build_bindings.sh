#!/bin/bash
# Builds the pybind11 module `gravomg_bindings` (host C++, g++) next to the `gravomg` package in
# gravo_mg_amd/dropin/, linked against the in-tree libgravomg_hip.so (rpath-relative).
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
out="${here}/../dropin"
mkdir -p "${out}"
PY="${PYTHON:-python3}"
inc="$(${PY} -m pybind11 --includes)"
suffix="$(${PY} -c 'import sysconfig; print(sysconfig.get_config_var("EXT_SUFFIX"))')"
g++ -O2 -std=c++17 -fPIC -shared -fvisibility=hidden ${inc} "${here}/bindings.cpp" "${here}/multigrid_solver.cpp" \
    -L"${here}/../lib" -lgravomg_hip -Wl,-rpath,'$ORIGIN/../lib' -o "${out}/gravomg_bindings${suffix}"
echo "built ${out}/gravomg_bindings${suffix}"
# The same module with -DGMG_TESTING for ONE test (tests/test_dropin_api.py: the exact-GS fallback of MGBS::MultigridSolver::solve on a system
# every smoother solves): lives under tests/_native/, is never on the product's import path, links the production libgravomg_hip.so.
tdir="${here}/../../tests/_native"
mkdir -p "${tdir}"
g++ -O2 -std=c++17 -fPIC -shared -fvisibility=hidden -DGMG_TESTING ${inc} "${here}/bindings.cpp" "${here}/multigrid_solver.cpp" \
    -L"${here}/../lib" -lgravomg_hip -Wl,-rpath,'$ORIGIN/../../gravo_mg_amd/lib' -o "${tdir}/gravomg_bindings${suffix}"
echo "built ${tdir}/gravomg_bindings${suffix}"
