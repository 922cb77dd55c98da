"""Static instruction mix of one kernel per basic block (no GPU needed):  python scripts/isa_count.py file.hip 'gs_block_epIdLi1E' [-v]"""
import os, re, subprocess, sys, collections, tempfile
here = os.path.dirname(os.path.abspath(__file__)); csrc = os.path.join(here, "..", "gravo_mg_amd", "csrc")
src, pat = sys.argv[1], sys.argv[2]
with tempfile.TemporaryDirectory() as tmp:
    out = os.path.join(tmp, "k.s")
    p = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(csrc, "..", "..", "include"), "-I" + csrc,
                        "--cuda-device-only", "-S", src, "-o", out], capture_output=True, text=True)
    if p.returncode: sys.exit(p.stderr[-3000:])
    lines = open(out).read().splitlines()
    if len(sys.argv) > 3 and sys.argv[3] == "-s": open("/tmp/isa_last.s", "w").write("\n".join(lines))
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(pat) + r"\w*:", l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
blocks = [("entry", [])]
for l in lines[start + 1:end]:
    m = re.match(r"^(\.LBB[0-9_]+):", l)
    if m: blocks.append((m.group(1), [])); continue
    t = l.strip()
    if t and not t.startswith(";") and not t.startswith("."): blocks[-1][1].append(t.split()[0])
def kind(i): return "valu" if i.startswith("v_") else "salu" if i.startswith("s_") else "lds" if i.startswith("ds_") else "vmem" if i.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other"
tot = collections.Counter()
for name, ins in blocks:
    c = collections.Counter(kind(i) for i in ins); tot += c
    if "-v" in sys.argv or len(ins) >= 20: print(f"{name:14s} n={len(ins):4d} " + " ".join(f"{k}={v}" for k, v in sorted(c.items())))
print("total", sum(tot.values()), dict(tot))
