"""SELL padding of the blocked levels (stored vs real entries of the in-block / off-block operators)."""
import sys
sys.path.insert(0, '.')
import bench
from gravo_mg_amd import cabi
H, mass, lhs, rhs = bench.build_workload(1732, 1732, "natural")
eng = cabi.Engine(); eng.use_hierarchy(H); eng.set_mass(mass); eng.set_system(lhs)
for k in range(eng.num_levels):
    info = eng.level_info(k)
    line = f"level {k}: n={info['n']} nnz={info['nnz']}"
    for which, name in ((0, "A"), (1, "Ain"), (2, "Aout"), (3, "P"), (4, "R")):
        try:
            s = eng.debug_sell(k, which)
        except Exception:
            continue
        if s is None:
            continue
        real = int((s["val"] != 0).sum())
        line += f" | {name}: stored {len(s['val'])} nonzero {real} ({len(s['val']) / max(real, 1):.2f}x) lpr {s['lpr']}"
    print(line)
