"""Idle time between kernels of the LAST whole-path run in a rocprofv3 kernel trace: python tools/trace_gaps.py <kernel_trace.csv>
(runs are delimited by k_apply_filter; per kernel: time spent in it and the idle time in front of it)."""
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "k_apply_filter" in r["Kernel_Name"]]
seg = rows[(ends[-2] + 1 if len(ends) > 1 else 0):ends[-1] + 1]
t0 = int(seg[0]["Start_Timestamp"]); prev_end = None; tot = 0
stat = collections.defaultdict(lambda: [0, 0, 0])
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"]); n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("glf::", "")[:44]
    st = stat[n]; st[0] += e - s; st[1] += 1
    if prev_end is not None: st[2] += max(s - prev_end, 0)
    prev_end = max(e, prev_end or 0); tot += e - s
print("wall %.2f ms, in kernels %.2f ms, idle %.2f ms, launches %d" % ((prev_end - t0) / 1e6, tot / 1e6, (prev_end - t0 - tot) / 1e6, len(seg)))
print("%-46s %5s %10s %10s" % ("kernel", "calls", "busy us", "idle-before us"))
for n, (b, c, g) in sorted(stat.items(), key=lambda x: -(x[1][0] + x[1][2]))[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print("%-46s %5d %10.1f %10.1f" % (n, c, b / 1e3, g / 1e3))
