"""Timeline of the LAST gmg_set_system in a rocprofv3 run of scripts/setup_trace.py: merges the kernel trace and the memory-copy
trace, prints every activity > 30 us (start offset, duration, idle gap on the device before it) and the busy / idle totals.
usage: python scripts/setup_timeline.py <kernel_trace.csv> <memory_copy_trace.csv>"""
import csv, sys
ev = []
for r in csv.DictReader(open(sys.argv[1])):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].split("(")[0].replace("void ", "")[:60]))
if len(sys.argv) > 2:
    for r in csv.DictReader(open(sys.argv[2])):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C %s %s B" % (r.get("Direction", "?"), r.get("Bytes", r.get("Size", "?")))))
ev.sort()
# the last set_system: from the last 'row_lengths' burst back to the preceding big H2D copy chain; take the window that starts at the last
# gap > 100 ms before the final rap_rows<0> launch
raps = [i for i, e in enumerate(ev) if "rap_rows<" in e[2]]
cold = [i for i in raps]
# first rap_rows<0> of the last cold set-up = the one preceded (within 30 ms) by large H2D copies and whose chain has >= 3 count passes
starts = []
for i in raps:
    if not starts or ev[i][0] - ev[starts[-1]][0] > 50e6:
        starts.append(i)
i0 = starts[-1]
# walk back to the first activity after an idle period of > 20 ms
j = i0
while j > 0 and ev[j][0] - ev[j - 1][1] < 20e6:
    j -= 1
t0 = ev[j][0]
end = j
while end + 1 < len(ev) and ev[end + 1][0] - ev[end][1] < 20e6:
    end += 1
prev_end, busy = t0, 0
print(f"window: {end - j + 1} activities, {(ev[end][1] - t0) / 1e6:.2f} ms")
acc_small, n_small = 0, 0
for s, e, name in ev[j:end + 1]:
    gap = s - prev_end
    if e - s > 30e3 or gap > 100e3:
        if n_small:
            print(f"            ... {n_small} short activities, {acc_small / 1e3:.0f} us busy")
            acc_small, n_small = 0, 0
        print(f"{(s - t0) / 1e6:8.2f} ms  dur {(e - s) / 1e3:8.1f} us  gap {gap / 1e3:8.1f} us  {name}")
    else:
        acc_small += e - s; n_small += 1
    busy += max(0, e - max(s, prev_end)) if e > prev_end else 0
    prev_end = max(prev_end, e)
print(f"busy {busy / 1e6:.2f} ms of {(ev[end][1] - t0) / 1e6:.2f} ms")
