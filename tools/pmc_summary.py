"""Summarises gpurun_out/<tag>_<mode>_trace and <tag>_<mode>_pmc* (tools/gpu_prof.sh) into
   profiles/r02_<mode>_kernel_stats.csv  (rocprofv3 --kernel-trace --stats, copied)
   profiles/r02_pmc_summary.txt          (per-kernel means of every counter)
   profiles/pmc_r02.json                 (what bench.py reads: per point VALU lane-instructions, HBM bytes, L2 hit rate)
HBM bytes follow MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE from separate passes, in KB (x1024), FETCH_SIZE doubled
on gfx950 (the doubling is calibrated for wide streaming reads: an upper bound for this gather), WRITE_SIZE as counted."""
import collections, csv, glob, json, os, shutil, sys
tag = sys.argv[1]
def short(n): return n.split("(")[0].replace("void ", "")
out_txt, result = [], {"_how": __doc__}
for mode in ("online", "large", "dense", "build", "buildlarge"):
    metaf = f"gpurun_out/prof_meta_{mode}.json"
    if not os.path.exists(metaf):
        continue
    meta = json.load(open(metaf))
    for f in glob.glob(f"gpurun_out/{tag}_{mode}_trace/*/*kernel_stats.csv"):
        shutil.copy(f, f"profiles/r02_{mode}_kernel_stats.csv")
        out_txt.append(f"## {mode}: kernel trace (rocprofv3 --kernel-trace --stats)")
        for r in list(csv.DictReader(open(f)))[:14]:
            out_txt.append(f"{short(r['Name'])[:40]:42s} calls {r['Calls']:>6s} avg_ns {float(r['AverageNs']):10.0f} min {r['MinNs']} max {r['MaxNs']}")
    stats = {}
    for f in glob.glob(f"gpurun_out/{tag}_{mode}_trace/*/*kernel_stats.csv"):
        for r in csv.DictReader(open(f)):
            stats[short(r["Name"])] = float(r["AverageNs"])
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(f"gpurun_out/{tag}_{mode}_pmc*/*/*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out_txt.append(f"## {mode}: PMC per-launch means")
    for k in sorted(acc):
        if not mode.startswith("build") and not ("k_residual" in k or "k_reduce" in k):
            continue
        out_txt.append("== " + k)
        for c, v in sorted(acc[k].items()):
            out_txt.append(f"   {c:36s} mean {sum(v)/len(v):16.1f}  n {len(v)}")
    mean = lambda k, c: (sum(acc[k][c]) / len(acc[k][c])) if acc.get(k, {}).get(c) else None
    for key, m in meta.items():
        if key.startswith("_"):
            result[f"{mode}{key}"] = m
            continue
        k, n = m["match"], m["points_per_launch"]
        e = {"points_per_launch": n, "rocprof_avg_us": None if k not in stats else stats[k] / 1e3}
        v = mean(k, "SQ_INSTS_VALU")
        if v is not None:
            e["valu_wave_insts_per_launch"] = v
            e["valu_insts_per_point"] = v * 64.0 / n
        av, gui = mean(k, "SQ_ACTIVE_INST_VALU"), mean(k, "GRBM_GUI_ACTIVE")
        if av is not None and gui:
            e["valu_busy"] = av * 4.0 / 1024.0 / (gui / 8.0)       # quad-cycles x 4 / SIMDs / (GUI_ACTIVE summed over 8 XCDs / 8)
        fs, ws = mean(k, "FETCH_SIZE"), mean(k, "WRITE_SIZE")
        if fs is not None and ws is not None:
            e["fetch_size_kb"], e["write_size_kb"] = fs, ws
            e["hbm_bytes_per_launch"] = (2.0 * fs + ws) * 1024.0
            e["hbm_bytes_per_point"] = e["hbm_bytes_per_launch"] / n
        hit, miss = mean(k, "TCC_HIT_sum"), mean(k, "TCC_MISS_sum")
        if hit is not None and miss is not None and hit + miss > 0:
            e["l2_hit_rate"] = hit / (hit + miss)
        result[key] = e
    if mode.startswith("build"):       # whole-build traffic: sum over all kernels of a rebuild / a merge is left to the reader of the txt
        tot = collections.defaultdict(float)
        for k in acc:
            for c in ("FETCH_SIZE", "WRITE_SIZE"):
                tot[c] += sum(acc[k].get(c, []))
        result[f"{mode}_traffic_total_kb"] = dict(tot)
open("profiles/r02_pmc_summary.txt", "w").write("\n".join(out_txt) + "\n")
json.dump(result, open("profiles/pmc_r02.json", "w"), indent=1)
print("\n".join(out_txt[:60]))
