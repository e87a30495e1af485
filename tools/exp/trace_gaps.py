#!/usr/bin/env python3
"""Given a rocprofv3 --kernel-trace CSV: per kernel name, the median begin->end duration and the median gap from the previous dispatch's
end to this dispatch's begin, over runs of >= 10 consecutive dispatches of the same kernel (the back-to-back graphs of tools/stage_times.py)."""
import csv, sys, collections, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
runs = collections.defaultdict(lambda: {'dur': [], 'gap': []})
prev = None; streak = 0
for r in rows:
    name = r['Kernel_Name'].split('(')[0]
    b, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if prev and prev[0] == name:
        streak += 1
        if streak >= 10:
            runs[name]['dur'].append(e - b); runs[name]['gap'].append(b - prev[2])
    else:
        streak = 0
    prev = (name, b, e)
for name, d in sorted(runs.items(), key=lambda kv: -len(kv[1]['dur'])):
    if len(d['dur']) < 20: continue
    print(f"{name[:70]:70s} n={len(d['dur']):5d}  begin->end median {statistics.median(d['dur'])/1e3:7.2f} us   gap median {statistics.median(d['gap'])/1e3:6.2f} us")
