"""Summarise rocprofv3 --pmc counter_collection CSVs: mean counter value per kernel name (per dispatch)."""
import csv, sys, collections
for path in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void gmgk::", "gmgk::")
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("#", path)
    for name, cs in sorted(acc.items()):
        for c, v in cs.items():
            print(f"{name:45s} {c:12s} n={len(v):5d} mean={sum(v)/len(v):14.1f} min={min(v):14.1f} max={max(v):14.1f}")
