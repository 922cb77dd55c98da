"""Registers / LDS / occupancy of the library's kernels as hipcc reports them (no GPU needed).

    python scripts/kernel_resources.py [substring ...]     # e.g.  gs_block_ep transfer<

Compiles gravo_mg_amd/csrc/engine.hip for the device only with -Rpass-analysis=kernel-resource-usage and prints one line per kernel
whose demangled name contains one of the substrings (all kernels without arguments)."""
import os, re, subprocess, sys, tempfile

here = os.path.dirname(os.path.abspath(__file__))
csrc = os.path.join(here, "..", "gravo_mg_amd", "csrc")
src = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".hip") else os.path.join(csrc, "engine.hip")
pats = [a for a in sys.argv[1:] if not a.endswith(".hip")]
with tempfile.TemporaryDirectory() as tmp:
    p = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(csrc, "..", "..", "include"), "-I" + csrc,
                        "--cuda-device-only", "-c", src, "-o", os.path.join(tmp, "dev.o"), "-Rpass-analysis=kernel-resource-usage"],
                       capture_output=True, text=True)
    txt = p.stderr
    if p.returncode:
        sys.exit(txt[-4000:])
rows = []
for blk in re.split(r"(?=remark: [^\n]*Function Name: )", txt):
    m = re.search(r"Function Name: (\S+)", blk)
    if not m:
        continue
    def g(k):
        mm = re.search(k + r": (\d+)", blk)
        return int(mm.group(1)) if mm else -1
    rows.append((m.group(1), g("VGPRs"), g("AGPRs"), g("TotalSGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n).replace("void ", "")
    if pats and not any(k in n for k in pats):
        continue
    print(f"{n:72s} vgpr {r[1]:4d} agpr {r[2]:3d} sgpr {r[3]:3d} scratch {r[4]:4d} waves/SIMD {r[5]} static-lds {r[6]}")
