"""Per-queue timeline of a pipelined run under `rocprofv3 --kernel-trace`: for each hardware queue, the kernels in start order with
the gap to the previous kernel of that queue, over one steady-state train() period.  Shows where the feature chain waits.
    python tools/exp/trace_gaps.py gpurun_out/trace_pipe"""
import csv, glob, os, sys, re, collections
d = sys.argv[1]
f = max(glob.glob(os.path.join(d, '*', '*_kernel_trace.csv')), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
def short(n): return re.sub(r'\(.*$', '', n).replace('void ', '').strip()[:58]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
byq = collections.defaultdict(list)
for r in rows: byq[r['Queue_Id']].append(r)
print('queues:', {q: len(v) for q, v in byq.items()})
# steady state: take the last third of the trace
t_all0, t_all1 = int(rows[0]['Start_Timestamp']), int(rows[-1]['End_Timestamp'])
lo = t_all0 + (t_all1 - t_all0) * 0.90
hi = lo + 900e3          # 0.9 ms window
for q, v in sorted(byq.items(), key=lambda kv: -len(kv[1]))[:3]:
    print(f'--- queue {q}')
    prev = None; busy = 0; gaps = 0
    for r in v:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if s < lo or s > hi: prev = e; continue
        gap = (s - prev) / 1e3 if prev else 0
        print(f'  +{(s - lo) / 1e3:8.1f} us  gap {gap:6.1f}  dur {(e - s) / 1e3:6.1f}  {short(r["Kernel_Name"])}  grid {r.get("Grid_Size", "")}')
        busy += (e - s) / 1e3; gaps += max(gap, 0); prev = e
    print(f'  window: kernel time {busy:.0f} us, gaps {gaps:.0f} us')
