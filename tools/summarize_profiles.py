#!/usr/bin/env python3
"""Turn the rocprofv3 outputs under gpurun_out/ into the committed summaries under profiles/.

    python tools/summarize_profiles.py r01          # round tag

Inputs (written on the GPU box by the commands quoted in profiles/<tag>_README.md):
    gpurun_out/prof_<tag>b/*/*_kernel_stats.csv                  rocprofv3 --kernel-trace --stats
    gpurun_out/pmc_{sq,fetch,write}/*/*_counter_collection.csv   three separate --pmc passes (tools/_pmc.sh)
    gpurun_out/bench_<tag>b.json                                 un-profiled `python bench.py`
"""
import csv, glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
G = os.path.join(ROOT, 'gpurun_out'); P = os.environ.get('RLREP_PROFILES_OUT') or os.path.join(ROOT, 'profiles')
os.makedirs(P, exist_ok=True)


def short(name):
    name = re.sub(r'\(.*$', '', name).replace('void ', '').strip()
    return name


ks = max(glob.glob(os.path.join(G, f'prof_{tag}b', '*', '*_kernel_stats.csv')), key=os.path.getmtime)
rows = list(csv.DictReader(open(ks)))
with open(os.path.join(P, f'{tag}_vlsac_b256_kernel_stats.csv'), 'w') as f:
    f.write(open(ks).read())
table = ['| kernel | calls | avg us | % of GPU time |', '|---|---|---|---|']
for r in rows[:22]:
    table.append(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.2f} | {float(r['Percentage']):.1f} |")

pmc = {}
for sub in ('pmc_sq', 'pmc_fetch', 'pmc_write'):
    files = glob.glob(os.path.join(G, sub, '*', '*_counter_collection.csv'))
    if not files:
        continue
    acc = {}
    for r in csv.DictReader(open(max(files, key=os.path.getmtime))):
        k = short(r['Kernel_Name']); c = r['Counter_Name']; v = float(r['Counter_Value'])
        d = acc.setdefault(k, {}).setdefault(c, {})
        d[r['Dispatch_Id']] = d.get(r['Dispatch_Id'], 0.0) + v          # sum over XCDs / SEs of one dispatch
    for k, cs in acc.items():
        for c, per in cs.items():
            pmc.setdefault(k, {})[c] = round(sum(per.values()) / len(per), 1)
            pmc[k]['dispatches'] = len(per)
json.dump(pmc, open(os.path.join(P, f'{tag}_pmc_summary.json'), 'w'), indent=1, sort_keys=True)

bench = open(os.path.join(G, f'bench_{tag}b.json')).read().strip().splitlines()[-1]
open(os.path.join(P, f'{tag}_bench.json'), 'w').write(bench + '\n')
# ---- the large-dimension workloads (BASELINE configs 3-5) and the GEMM engines ----------------------------------
extra = {}
for wl in ('ctrlsac_halfcheetah_f2048_b256', 'spedersac_ant_f512_b1024', 'diffsrsac_humanoid_b2048'):
    fs = glob.glob(os.path.join(G, f'prof_{wl}', '*', '*_kernel_stats.csv'))
    if fs:
        src = max(fs, key=os.path.getmtime)
        rws = [r for r in csv.DictReader(open(src)) if 'at::native' not in r['Name'] and 'rocclr' not in r['Name']]
        with open(os.path.join(P, f'{tag}_{wl}_kernel_stats.csv'), 'w') as f:
            w = csv.DictWriter(f, fieldnames=list(rws[0].keys())); w.writeheader(); w.writerows(rws[:20])
        extra[wl] = [(short(r['Name']), int(r['Calls']), float(r['AverageNs']) / 1e3, float(r['Percentage'])) for r in rws[:8]]
lines = []
for f in sorted(glob.glob(os.path.join(G, 'bench_*.log'))):
    last = [l for l in open(f).read().strip().splitlines() if l.startswith('{')]
    if last:
        lines.append(last[-1])
seen = {}
for l in lines:
    d = json.loads(l); seen[d['config']['workload']] = l          # one line per workload (latest file wins)
open(os.path.join(P, f'{tag}_bench_all.jsonl'), 'w').write('\n'.join(seen[k] for k in sorted(seen)) + '\n')
gp = {}
for sub in ('pmc_gemm', 'pmc_gemm2'):
    files = glob.glob(os.path.join(G, sub, '*', '*_counter_collection.csv'))
    if not files:
        continue
    acc = {}
    for r in csv.DictReader(open(max(files, key=os.path.getmtime))):
        k = short(r['Kernel_Name'])
        if 'gemm' not in k:
            continue
        d = acc.setdefault(k, {}).setdefault(r['Counter_Name'], {})
        d[r['Dispatch_Id']] = d.get(r['Dispatch_Id'], 0.0) + float(r['Counter_Value'])
    for k, cs in acc.items():
        for c, per in cs.items():
            gp.setdefault(k, {})[c] = round(sum(per.values()) / len(per), 1)
if gp:
    json.dump(gp, open(os.path.join(P, f'{tag}_pmc_gemm_4096.json'), 'w'), indent=1, sort_keys=True)
if os.path.exists(os.path.join(G, 'bench_gemm.log')):
    keep = [l for l in open(os.path.join(G, 'bench_gemm.log')).read().splitlines() if ' GF |' in l]
    open(os.path.join(P, f'{tag}_gemm_engines.txt'), 'w').write('\n'.join(keep) + '\n')
if extra:
    json.dump(extra, open(os.path.join(P, f'{tag}_large_workloads_top_kernels.json'), 'w'), indent=1)

print('\n'.join(table))
b = json.loads(bench)
print('\nbench:', b['value'], b['unit'], b['ms_per_step'], 'ms;', 'roofline', json.dumps(b.get('roofline')))
for k in ('nc_fwd_kernel<1>', 'nc_dw_kernel', 'nc_dx_kernel<true>'):
    if k in pmc:
        print(k, {c: pmc[k].get(c) for c in ('SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_WAIT_ANY', 'SQ_INSTS_MFMA', 'SQ_VALU_MFMA_BUSY_CYCLES', 'FETCH_SIZE', 'WRITE_SIZE', 'SQ_LDS_BANK_CONFLICT')})
