"""Timeline of map builds / merges out of a rocprofv3 kernel trace: python tools/trace_timeline.py <dir with *kernel_trace.csv>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
seqs, cur = [], None
for r in rows:
    n = r['Kernel_Name'].split('(')[0]
    if 'k_target_keys' in n: cur = [r]
    elif cur is not None:
        cur.append(r)
        if 'k_ghost_link' in n: seqs.append(cur); cur = None
for s in seqs[2:3] + seqs[-1:]:
    t0 = int(s[0]['Start_Timestamp'])
    print('--- span us', (int(s[-1]['End_Timestamp']) - t0) / 1e3, 'sum us', sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in s) / 1e3)
    prev = t0
    for r in s:
        st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        print(f"  {r['Kernel_Name'].split('(')[0][:28]:30s} start +{(st - t0) / 1e3:7.1f} gap {(st - prev) / 1e3:6.1f} dur {(en - st) / 1e3:6.1f}")
        prev = en
