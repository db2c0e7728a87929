"""Summarises gpurun_out/<tag>_trace and <tag>_pmc* into per-kernel means (prints a table)."""
import csv, glob, sys, collections
tag = sys.argv[1]
def short(n): return n.split("(")[0].replace("void ", "")[:28]
for f in glob.glob(f"gpurun_out/{tag}_trace/*/*kernel_stats.csv"):
    for r in list(csv.DictReader(open(f)))[:6]:
        print(f"{short(r['Name']):30s} calls {r['Calls']:>6s} avg_ns {float(r['AverageNs']):10.0f} min {r['MinNs']} max {r['MaxNs']}")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(f"gpurun_out/{tag}_pmc*/*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in acc:
    if not ("k_residual" in k or "k_reduce" in k): continue
    print("==", k)
    for c, v in acc[k].items():
        print(f"   {c:36s} mean {sum(v)/len(v):14.1f}  n {len(v)}")
