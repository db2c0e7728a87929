"""Turns the counter passes of tools/ubench/fetch_calibration (run by tools/gpu_fetch_cal.sh) into profiles/r03_fetch_calibration.txt
and profiles/fetch_calibration_r03.json: for every access pattern, known bytes / (counter x unit) = the factor that turns
FETCH_SIZE (KB) into bytes for that pattern.  pmc_summary.py and bench.py use `pattern_coh` — the engine's own gather with
LiDAR-like lane coherence — for the residual kernels, and `stream16` is the guide's control (factor 2.0 expected)."""
import collections, csv, glob, json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r3cal"
known = json.load(open(f"gpurun_out/{tag}_known.json"))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(f"gpurun_out/{tag}_pmc*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
lines, res = [], {"_how": __doc__}
lines.append("# FETCH_SIZE calibration for the residual kernel's gather (tools/ubench/fetch_calibration.hip; 2 GiB table, every line touched once)")
lines.append("# factor = known bytes / (FETCH_SIZE x 1024); 'lines' = distinct 64-byte lines touched x 64, 'requested' = bytes the lanes load")
for k in ("k_cal_stream16", "k_cal_slot64", "k_cal_slot40", "k_cal_pattern", "k_cal_pattern_coh"):
    if k not in acc or k not in known:
        continue
    kn = known[k]
    c = {n: sum(v) / len(v) for n, v in acc[k].items()}
    e = {"known_requested_bytes": kn["requested"], "known_line_bytes": kn["lines"], "index_bytes": kn["index"], "counters": c}
    fs = c.get("FETCH_SIZE")
    if fs:
        fb = fs * 1024.0
        # the index array is streamed (4 B per lane, coalesced): its bytes are taken off at the streaming factor's worth
        # (FETCH_SIZE counts it at 1/2 as 16-B streams are; it is < 2 % of any gather here)
        fb_data = fb - kn["index"] / 2.0
        e["fetch_size_bytes"] = fb
        e["factor_vs_lines"] = kn["lines"] / fb_data
        e["factor_vs_requested"] = kn["requested"] / fb_data
    rd = c.get("TCC_EA0_RDREQ_sum")
    if rd:
        e["bytes_per_rdreq_if_lines"] = kn["lines"] / rd
    r32 = c.get("TCC_EA0_RDREQ_32B_sum")
    if rd and r32 is not None:
        e["rdreq_32B_share"] = r32 / rd
    res[k] = e
    lines.append(f"{k:20s} requested {kn['requested']/1e6:9.1f} MB  lines {kn['lines']/1e6:9.1f} MB  "
                 + (f"FETCH_SIZE {fs*1024/1e6:9.1f} MB  factor(lines) {e['factor_vs_lines']:.3f}  factor(requested) {e['factor_vs_requested']:.3f}" if fs else "FETCH_SIZE n/a")
                 + (f"  RDREQ {rd:.0f} ({e['bytes_per_rdreq_if_lines']:.1f} B each if every line was fetched once)" if rd else "")
                 + (f"  32B share {e['rdreq_32B_share']:.2f}" if 'rdreq_32B_share' in e else ""))
open("profiles/r03_fetch_calibration.txt", "w").write("\n".join(lines) + "\n")
json.dump(res, open("profiles/fetch_calibration_r03.json", "w"), indent=1)
print("\n".join(lines))
