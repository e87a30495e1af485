"""What does the other chain cost each launch?  From a `rocprofv3 --kernel-trace` of the pipelined run: for every kernel of the feature queue
(steady state), its duration and the gap in front of it, split by what was running on the OTHER queue at that moment (a noise-critic kernel,
another kernel, nothing).  Aggregated per kernel name.
    python tools/exp/trace_overlap.py gpurun_out/trace_pipe"""
import csv, glob, os, sys, re, collections, bisect
d = sys.argv[1]
f = max(glob.glob(os.path.join(d, '*', '*_kernel_trace.csv')), key=os.path.getmtime)
rows = [r for r in csv.DictReader(open(f))]
def short(n): return re.sub(r'\(.*$', '', n).replace('void ', '').strip()[:52]
for r in rows: r['s'], r['e'], r['n'] = int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])
rows.sort(key=lambda r: r['s'])
t0, t1 = rows[0]['s'], rows[-1]['e']
lo = t0 + 0.5 * (t1 - t0)                      # steady state: second half
byq = collections.defaultdict(list)
for r in rows:
    if r['s'] >= lo: byq[r['Queue_Id']].append(r)
qs = sorted(byq, key=lambda q: -len(byq[q]))[:2]
# the feature queue is the one that runs heads_vae_kernel
fq = next(q for q in qs if any(r['n'].startswith('heads_vae') for r in byq[q]))
cq = next(q for q in qs if q != fq)
F, Cq = byq[fq], byq[cq]
print(f'feature queue {fq}: {len(F)} kernels; critic/actor queue {cq}: {len(Cq)} kernels; window {(t1 - lo) / 1e3:.0f} us')
cs = [r['s'] for r in Cq]
def other_at(a, b):
    """share of [a, b) during which the other queue runs a noise-critic kernel / any kernel"""
    nc = anyk = 0
    i = max(0, bisect.bisect_left(cs, a) - 2)
    while i < len(Cq) and Cq[i]['s'] < b:
        o = max(0, min(b, Cq[i]['e']) - max(a, Cq[i]['s']))
        if o > 0:
            anyk += o
            if Cq[i]['n'].startswith('nc_'): nc += o
        i += 1
    L = max(b - a, 1)
    return nc / L, anyk / L
agg = collections.defaultdict(lambda: collections.defaultdict(list))
prev = None
for r in F:
    nc, anyk = other_at(r['s'], r['e'])
    cls = 'beside nc' if nc > 0.5 else ('beside other' if anyk > 0.5 else 'alone')
    agg[r['n']][cls].append((r['e'] - r['s']) / 1e3)
    if prev is not None and r['s'] - prev < 60e3:
        gnc, gany = other_at(prev, r['s'])
        gcls = 'beside nc' if gnc > 0.5 else ('beside other' if gany > 0.5 else 'alone')
        agg['(gap before a launch)'][gcls].append((r['s'] - prev) / 1e3)
    prev = r['e']
print(f'{"kernel":54s} {"alone":>16s} {"beside other":>16s} {"beside nc_*":>16s}')
tot = collections.defaultdict(float)
for n, c in sorted(agg.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values())):
    cells = []
    for cls in ('alone', 'beside other', 'beside nc'):
        v = c.get(cls, [])
        cells.append(f'{sum(v) / len(v):6.2f} us x{len(v):5d}' if v else ' ' * 16)
        tot[cls] += sum(v)
    print(f'{n:54s} ' + ' '.join(cells))
print('total time of the feature queue by class (us):', {k: round(v) for k, v in tot.items()})
# period: distance between consecutive train_prologue kernels
pro = [r['s'] for r in F if r['n'].startswith('train_prologue')]
if len(pro) > 2:
    import statistics
    print('period under the profiler (prologue to prologue): median %.1f us' % (statistics.median(b - a for a, b in zip(pro, pro[1:])) / 1e3))
