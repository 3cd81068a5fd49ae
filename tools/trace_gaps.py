"""Busy/idle analysis of a rocprofv3 kernel trace: python tools/trace_gaps.py DIR"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for r in rows]
# keep the last whole-path run only (a run starts with k_sample_tables): the first one pays the pool's hipMallocs
starts = [i for i, k in enumerate(ks) if "k_sample_tables" in k[2]]
if starts: ks = ks[starts[-1]:]
t0 = ks[0][0]
busy = 0; last_end = t0; gaps = collections.Counter(); gap_total = 0
per = collections.defaultdict(float)
for s, e, n in ks:
    if s > last_end:
        gap_total += s - last_end
        gaps[n[:40]] += s - last_end
    busy += e - max(s, last_end) if e > last_end else 0
    last_end = max(last_end, e)
    per[n[:48]] += (e - s)
print("span %.2f ms busy %.2f ms idle %.2f ms" % ((last_end - t0) / 1e6, busy / 1e6, gap_total / 1e6))
print("largest idle-before-kernel:")
for n, g in gaps.most_common(12): print("   %-42s %8.3f ms" % (n, g / 1e6))
print("largest kernels:")
for n, g in sorted(per.items(), key=lambda kv: -kv[1])[:14]: print("   %-50s %8.3f ms" % (n, g / 1e6))
print("kernel launches in the run:", len(ks))
