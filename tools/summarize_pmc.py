#!/usr/bin/env python3
"""rocprofv3 --pmc passes of one bench.py workload -> profiles/<tag>_pmc_<workload>.json (what bench.py's roofline.traffic reads).

    python tools/summarize_pmc.py r04 vlsac_halfcheetah_f256_b256 <train_calls>

Inputs: gpurun_out/pmc_{sq,fetch,write}_<workload>/*/*_counter_collection.csv (three SEPARATE --pmc passes with --kernel-trace only:
tools/_collect_r04.sh).  Per kernel: the mean over dispatches of the per-dispatch sums (over XCDs / SEs) of every counter, and the
number of dispatches.  FETCH_SIZE / WRITE_SIZE are in KB as rocprofv3 reports them (raw: bench.py applies the gfx950 correction,
2 x FETCH_SIZE, MI355X_MICROARCH.md)."""
import csv, glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, wl, calls = sys.argv[1], sys.argv[2], int(sys.argv[3])
G = os.path.join(ROOT, 'gpurun_out'); P = os.environ.get('RLREP_PROFILES_OUT') or os.path.join(ROOT, 'profiles')
os.makedirs(P, exist_ok=True)


def short(name):
    return re.sub(r'\(.*$', '', name).replace('void ', '').strip()


pmc, passes = {}, {}
for sub in ('pmc_sq', 'pmc_fetch', 'pmc_write'):
    files = glob.glob(os.path.join(G, f'{sub}_{wl}', '*', '*_counter_collection.csv'))
    if not files:
        continue
    acc = {}
    for r in csv.DictReader(open(max(files, key=os.path.getmtime))):
        k = short(r['Kernel_Name'])
        if k.startswith('at::native') or 'rocclr' in k:
            continue
        d = acc.setdefault(k, {}).setdefault(r['Counter_Name'], {})
        d[r['Dispatch_Id']] = d.get(r['Dispatch_Id'], 0.0) + float(r['Counter_Value'])
    passes[sub] = {k: len(next(iter(cs.values()))) for k, cs in acc.items()}
    for k, cs in acc.items():
        for c, per in cs.items():
            pmc.setdefault(k, {})[c] = round(sum(per.values()) / len(per), 1)
            pmc[k]['dispatches'] = len(per)
            pmc[k].setdefault('dispatches_by_pass', {})[sub] = len(per)
# The three passes are three RUNS of the same command: a summary is only meaningful if they launched the same kernels the same number of
# times (round 4's headline summary mixed two routings of the tile engine -- fast front ends in one pass, record front end in the others --
# and every figure derived from "kernels that carry both counters" was an artefact).  Refuse to write such a file.
names = [set(v) for v in passes.values()]
bad = []
if len(passes) > 1:
    allk = set().union(*names)
    for k in sorted(allk):
        cnt = {sub: passes[sub].get(k, 0) for sub in passes}
        if len(set(cnt.values())) != 1:
            bad.append((k, cnt))
if bad and os.environ.get('RLREP_PMC_ALLOW_MISMATCH') != '1':
    print('summarize_pmc: the passes disagree on kernels / dispatches per kernel -- not written:', file=sys.stderr)
    for k, cnt in bad[:20]:
        print('   ', k, cnt, file=sys.stderr)
    sys.exit(3)
pmc['__meta__'] = {'workload': wl, 'train_calls': calls, 'passes': sorted(passes), 'passes_agree': not bad,
                   'form': os.environ.get('RLREP_PMC_FORM', 'eager'),       # 'graph': the passes ran the timed form (hipGraph replay); 'eager': --no-graph
                   'command': f'rocprofv3 --pmc <counters> --kernel-trace --output-format csv -- python3 bench.py --workload {wl} --steps <N> --warmup 5 --no-cpu ' + ('' if os.environ.get('RLREP_PMC_FORM') == 'graph' else '--no-graph ') + '--no-profile --quick  (three passes: SQ_* / FETCH_SIZE / WRITE_SIZE + LDS, instruction counters)',
                   'units': 'FETCH_SIZE / WRITE_SIZE in KB, raw (gfx950: x2 on FETCH_SIZE for wide reads before comparing with bytes)'}
out = os.path.join(P, f'{tag}_pmc_{wl}.json')
json.dump(pmc, open(out, 'w'), indent=1, sort_keys=True)
if wl == 'vlsac_halfcheetah_f256_b256':
    json.dump(pmc, open(os.path.join(P, f'{tag}_pmc_summary.json'), 'w'), indent=1, sort_keys=True)
print(out, len(pmc) - 1, 'kernels')
