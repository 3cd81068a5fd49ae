"""Turn one tools/profile_round.sh output directory into the committed summaries under profiles/.

    python tools/collect_profiles.py gpurun_out/prof_TAG TAG

Writes profiles/TAG_kernel_stats.csv (rocprofv3 --kernel-trace --stats of bench.py), TAG_bench.json /
TAG_bench_under_rocprof.json (the JSON lines of the two bench runs), TAG_pmc_summary.json (per-kernel
counter sums of the --pmc passes; one whole-path run each) and updates profiles/pmc_traffic.json, which
bench.py reads for roofline.traffic. FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled as
MI355X_MICROARCH.md (HBM section) prescribes for gfx950.
"""
import csv, glob, json, os, shutil, sys, collections

src, tag = sys.argv[1], sys.argv[2]
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
dst = os.path.join(root, "profiles")

stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(dst, tag + "_kernel_stats.csv"))
for name in ("bench_plain", "bench_under_rocprof"):
    f = os.path.join(src, name + ".json")
    if os.path.exists(f):
        line = [l for l in open(f).read().splitlines() if l.startswith("{")][-1]
        json.dump(json.loads(line), open(os.path.join(dst, "%s_%s.json" % (tag, name.replace("_plain", ""))), "w"), indent=1)

summary = collections.defaultdict(dict)
series = collections.defaultdict(dict)
for d in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    dur = collections.defaultdict(float)
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Kernel_Name"].split("(")[0]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    calls = collections.defaultdict(lambda: collections.defaultdict(int))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            c = r["Counter_Name"]
            summary[k][c] = summary[k].get(c, 0.0) + float(r["Counter_Value"])
            calls[k][c] += 1
            # the same kernel serves launches of very different size (k_grid_rowpass: the Nystroem passes and the much
            # smaller L_A sweeps): keep the per-dispatch series to tell them apart by their WRITE_SIZE
            series[k].setdefault(c, []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for k in calls:
        summary[k]["calls"] = max(calls[k].values())
        summary[k]["total_ms_pass_" + os.path.basename(d)[4:]] = round(dur.get(k, 0.0), 3)
for k in list(series):
    if "k_grid_rowpass" in k and "WRITE_SIZE" in series[k] and "FETCH_SIZE" in series[k]:
        w = [v for _, v in sorted(series[k]["WRITE_SIZE"])]
        f = [v for _, v in sorted(series[k]["FETCH_SIZE"])]
        if len(w) == len(f) and w:
            big = [i for i, v in enumerate(w) if v > 0.5 * max(w)]   # the Nystroem passes (dispatch order is the same in every run)
            summary[k + "@nystroem_passes"] = {"WRITE_SIZE": sum(w[i] for i in big), "FETCH_SIZE": sum(f[i] for i in big),
                                               "calls": len(big)}
json.dump(summary, open(os.path.join(dst, tag + "_pmc_summary.json"), "w"), indent=1, sort_keys=True)

nys = [k for k in summary if "k_nystroem_f16s" in k and "FETCH_SIZE" in summary[k]]
if nys:
    k = max(nys, key=lambda k: summary[k]["FETCH_SIZE"])
    fetch = summary[k]["FETCH_SIZE"] / summary[k]["calls"] * 1024.0 * 2.0
    write = summary[k]["WRITE_SIZE"] / summary[k]["calls"] * 1024.0
    tf = os.path.join(dst, "pmc_traffic.json")
    traffic = json.load(open(tf)) if os.path.exists(tf) else {}
    traffic["4096x4096_m64_f16s_gpus1"] = {
        "nystroem_bytes_per_launch": fetch + write, "fetch_bytes_x2_corrected": fetch, "write_bytes": write,
        "kernel": k,
        "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, profiles/%s_pmc_summary.json); FETCH_SIZE "
                  "doubled per MI355X_MICROARCH.md (gfx950 counts 16-B/lane reads at half)" % tag}
    json.dump(traffic, open(tf, "w"), indent=1)
    print("nystroem traffic per launch: %.2f GB fetch (x2) + %.2f GB write" % (fetch / 1e9, write / 1e9))
tf = os.path.join(dst, "pmc_traffic.json")
traffic = json.load(open(tf)) if os.path.exists(tf) else {}
entry = traffic.setdefault("4096x4096_m64_f16s_gpus1", {})
# (k_rank_colpassILi2ELb0: the Nystroem launch of the rank form -- Lb1 is the L_A sweeps' instantiation of the same kernel;
#  k_grid_rowpass_rt..ELb0ELb1: the rank-form row pass of the Nystroem stage)
for name, key in (("k_block_matvec_f16s", "matvec"), ("k_grid_rowpass", "grid_rowpass"), ("k_grid_colpass", "grid_colpass"),
                  ("k_rank_colpassILi2ELb0", "rank_colpass"), ("k_rank_colpassILi2ELb1", "rank_colpass_op"),
                  ("k_band<2, 2, 8, false>", "band"), ("k_band<2, 1, 8, true>", "band_op")):
    ks = [k for k in summary if name in k and "FETCH_SIZE" in summary[k] and "WRITE_SIZE" in summary[k]]
    if not ks:
        continue
    # per-launch traffic; the largest launches are the ones bench.py's roofline refers to (the Nystroem passes)
    k = max(ks, key=lambda k: (summary[k]["FETCH_SIZE"] + summary[k]["WRITE_SIZE"]) / summary[k]["calls"])
    fetch = summary[k]["FETCH_SIZE"] / summary[k]["calls"] * 1024.0 * 2.0
    write = summary[k]["WRITE_SIZE"] / summary[k]["calls"] * 1024.0
    entry[key + "_bytes_per_launch"] = fetch + write
    entry[key + "_fetch_bytes_x2_corrected"] = fetch
    entry[key + "_write_bytes"] = write
    entry[key + "_source"] = "profiles/%s_pmc_summary.json, %d launches" % (tag, summary[k]["calls"])
    print("%s traffic per launch: %.2f GB fetch (x2) + %.2f GB write" % (name, fetch / 1e9, write / 1e9))
json.dump(traffic, open(tf, "w"), indent=1)
for k in sorted(summary, key=lambda k: -summary[k].get("GRBM_GUI_ACTIVE", 0))[:6]:
    print(k[:70], {c: v for c, v in summary[k].items() if c in ("FETCH_SIZE", "WRITE_SIZE", "calls")})
