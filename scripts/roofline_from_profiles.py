"""Recompute roofline.frac of the dominant kernel from the committed evidence of ONE job: algorithmic bytes per launch (the bench line that ran under
rocprofv3, profiles/rNN/bench_prof_<workload>.json) / the kernel's average duration in that run's rocprofv3 statistics (kernel_stats_<workload>.csv),
and the HBM traffic of the PMC passes beside it.   python scripts/roofline_from_profiles.py profiles/r06"""
import csv, glob, json, os, re, sys
root = sys.argv[1] if len(sys.argv) > 1 else "profiles/r06"
PEAK = 8000.0
for jf in sorted(glob.glob(os.path.join(root, "bench_prof_*.json"))):
    name = os.path.basename(jf)[len("bench_prof_"):-len(".json")]
    try:
        d = json.loads(open(jf).read().strip().splitlines()[-1])
    except Exception as e:
        print(name, "no bench line:", e); continue
    roof = d["roofline"]
    m = re.match(r"gmgk::(\w+)<([^>]*)>", roof["kernel"])
    rows = list(csv.DictReader(open(os.path.join(root, f"kernel_stats_{name}.csv"))))
    want = (f"gmgk::{m.group(1)}<" + ", ".join(p.strip() for p in m.group(2).split(",")) + ">") if m else ""
    hit = [r for r in rows if want and want in r["Name"]]
    if not hit:      # (a line whose label is not the listed name -- "..." in a block sweep's template list, an older bench.py: the level-0 sweep kernel with the most launches)
        cand = [r for r in rows if re.search(r"gmgk::(gs_color|gs_block_ep)<double, \d", r["Name"]) and "gs_color_" not in r["Name"]]
        fam = "gs_block_ep" if "gs_block_ep" in roof["kernel"] else "gs_color"
        if not any(f"gmgk::{fam}<" in r["Name"] for r in cand): fam = "gs_block_ep" if fam == "gs_color" else "gs_color"
        cand = [r for r in cand if f"gmgk::{fam}<" in r["Name"]]
        if fam == "gs_block_ep":      # level 0's sweep is the longest-running instance
            cand.sort(key=lambda r: -float(r["AverageNs"]))
        else:
            cand.sort(key=lambda r: -int(r["Calls"]))
        hit = cand[:1]
        if hit: want = re.search(r"gmgk::\w+<[^>]*>", hit[0]["Name"]).group(0)
    avg_ns = calls = None
    if hit: avg_ns, calls = float(hit[0]["AverageNs"]), int(hit[0]["Calls"])
    by = roof["algorithmic_bytes_per_launch"]
    line = f"{name:22s} {want:34s} algorithmic {by / 1e6:8.2f} MB per launch"
    if avg_ns:
        gbs = by / avg_ns
        line += f" | rocprofv3: {calls} launches, {avg_ns / 1e3:6.2f} us -> {gbs:7.1f} GB/s = {gbs / PEAK:.3f} of {PEAK:.0f}"
    line += f" | HIP events in the same run: {1e3 * roof['launch_ms']:6.2f} us -> frac {roof['frac']:.3f}"
    pm = os.path.join(root, f"pmc_fetch_write_{name}.txt")
    if os.path.exists(pm):
        fetch = write = None
        for l in open(pm):
            # (a full launch: the maximum -- a few dispatches of the level-0 colour kernel are heads of a next cycle that found the iteration stopped and
            # returned at once, gmg_config::speculate_head; they pull the mean down.  Where min is close to max, max == mean to four digits.)
            if want in l and "FETCH_SIZE" in l: fetch = float(re.search(r"max=\s*([0-9.]+)", l).group(1))
            if want in l and "WRITE_SIZE" in l: write = float(re.search(r"max=\s*([0-9.]+)", l).group(1))
        if fetch and write:
            hbm = (2 * fetch + write) * 1024
            line += f" | PMC (2 FETCH + WRITE) x 1024, full launch = {hbm / 1e6:7.2f} MB = {hbm / by:.3f} x algorithmic"
    print(line)
