"""Gaps between consecutive dispatches of a kernel in a rocprofv3 --kernel-trace CSV: durations and idle time between the end of one launch and the start of the next.
Usage: kernel_gaps.py <*_kernel_trace.csv> <kernel name substring>"""
import csv, sys, statistics
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
sel = [i for i, r in enumerate(rows) if sys.argv[2] in r[2]]
dur = [rows[i][1] - rows[i][0] for i in sel]
gaps = [rows[i][0] - rows[i - 1][1] for i in sel if i > 0 and sys.argv[2] in rows[i - 1][2]]
q = lambda v, p: sorted(v)[int(p * (len(v) - 1))]
print(f"{len(sel)} dispatches of *{sys.argv[2]}*: duration median {statistics.median(dur) / 1e3:.2f} us (10 % {q(dur, 0.1) / 1e3:.2f}, 90 % {q(dur, 0.9) / 1e3:.2f}, mean {statistics.mean(dur) / 1e3:.2f}); "
      f"gap to the previous dispatch of the same kernel median {statistics.median(gaps) / 1e3:.2f} us (10 % {q(gaps, 0.1) / 1e3:.2f}, 90 % {q(gaps, 0.9) / 1e3:.2f}, mean {statistics.mean(gaps) / 1e3:.2f})")
