"""Summarises gpurun_out/<tag>_<mode>_trace and <tag>_<mode>_pmc* (tools/gpu_prof.sh) into
   profiles/r03_<mode>_kernel_stats.csv  (rocprofv3 --kernel-trace --stats, copied)
   profiles/r03_pmc_summary.txt          (per-kernel means of every counter)
   profiles/pmc_r03.json                 (what bench.py reads: per point VALU lane-instructions, HBM bytes, L2 hit rate; stamped
                                          with the content hash of lio-slam_amd/csrc it was taken on, bench.csrc_hash())
HBM bytes follow MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE from separate passes, in KB (x1024).  FETCH_SIZE's factor
is the one CALIBRATED for the pattern (profiles/r03_fetch_calibration.txt, tools/ubench/fetch_calibration.hip): the residual
kernels gather 64-byte slots, for which the counter reports the line bytes as they are (factor 1.0); the guide's doubling holds
for 16-byte-per-lane streams (the build's kernels: factor 2.0, an upper bound where they gather).  WRITE_SIZE as counted."""
import collections, csv, glob, json, os, shutil, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
tag = sys.argv[1]
try:
    GATHER_FACTOR = json.load(open("profiles/fetch_calibration_r03.json"))["k_cal_pattern_coh"]["factor_vs_lines"]
except (OSError, KeyError, ValueError):
    GATHER_FACTOR = 1.0
def short(n): return n.split("(")[0].replace("void ", "")
out_txt, result = [], {"_how": __doc__, "_csrc_hash": bench.csrc_hash(), "_fetch_factor_gather": GATHER_FACTOR}
for mode in ("online", "large", "dense", "build", "buildlarge"):
    metaf = f"gpurun_out/prof_meta_{mode}.json"
    if not os.path.exists(metaf):
        continue
    meta = json.load(open(metaf))
    for f in glob.glob(f"gpurun_out/{tag}_{mode}_trace/*/*kernel_stats.csv"):
        shutil.copy(f, f"profiles/r03_{mode}_kernel_stats.csv")
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
            e["fetch_factor"] = GATHER_FACTOR
            e["hbm_bytes_per_launch"] = (GATHER_FACTOR * fs + ws) * 1024.0
            e["hbm_bytes_per_launch_if_streamed_factor_2"] = (2.0 * fs + ws) * 1024.0
            e["hbm_bytes_per_point"] = e["hbm_bytes_per_launch"] / n
        hit, miss = mean(k, "TCC_HIT_sum"), mean(k, "TCC_MISS_sum")
        if hit is not None and miss is not None and hit + miss > 0:
            e["l2_hit_rate"] = hit / (hit + miss)
        result[key] = e
    if mode.startswith("build"):       # whole-build traffic: totals, and per kernel (KB over the whole run: divide by the rebuilds)
        tot = collections.defaultdict(float)
        per = {}
        for k in acc:
            for c in ("FETCH_SIZE", "WRITE_SIZE"):
                tot[c] += sum(acc[k].get(c, []))
            per[k] = {"FETCH_SIZE_kb": sum(acc[k].get("FETCH_SIZE", [])), "WRITE_SIZE_kb": sum(acc[k].get("WRITE_SIZE", [])),
                      "launches": len(acc[k].get("FETCH_SIZE", []))}
        result[f"{mode}_traffic_total_kb"] = dict(tot)
        result[f"{mode}_traffic_per_kernel_kb"] = per
        out_txt.append(f"## {mode}: FETCH_SIZE / WRITE_SIZE per kernel over the run (KB; streams: x2 for FETCH_SIZE bytes)")
        for k, v in sorted(per.items(), key=lambda kv: -(2 * kv[1]["FETCH_SIZE_kb"] + kv[1]["WRITE_SIZE_kb"])):
            out_txt.append(f"   {k[:44]:46s} fetch {v['FETCH_SIZE_kb']:14.0f}  write {v['WRITE_SIZE_kb']:14.0f}  launches {v['launches']}")
open("profiles/r03_pmc_summary.txt", "w").write("\n".join(out_txt) + "\n")
json.dump(result, open("profiles/pmc_r03.json", "w"), indent=1)
print("\n".join(out_txt[:60]))
