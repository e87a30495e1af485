#!/usr/bin/env python3
"""Regenerate profiles/<tag>_README.md from the committed summaries (run after tools/summarize_profiles.py)."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'summarize_profiles.py'), tag], capture_output=True, text=True).stdout
table = '\n'.join(l for l in out.split('\n\nbench:')[0].split('\n')[2:] if 'at::native' not in l)
b = json.loads(open(os.path.join(ROOT, 'profiles', f'{tag}_bench.json')).read())
pmc = json.load(open(os.path.join(ROOT, 'profiles', f'{tag}_pmc_summary.json')))


def row(k):
    d = pmc[k]; wc = max(d.get('SQ_WAVE_CYCLES', 0), 1)
    return (f"| `{k}` | {d.get('SQ_INSTS_MFMA', 0):.0f} | {100 * d.get('SQ_WAIT_ANY', 0) / wc:.0f} % | {100 * d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / wc:.0f} % | "
            f"{d.get('FETCH_SIZE', 0) / 1024:.1f} | {d.get('WRITE_SIZE', 0) / 1024:.1f} | {d.get('SQ_LDS_BANK_CONFLICT', 0):.0f} |")


keys = [k for k in ('nc_fwd_x3q_kernel', 'nc_fwd_x3w_kernel<8>', 'nc_fwd_x3_kernel<1>', 'nc_fwd_kernel<1>', 'nc_dw_x3_kernel', 'nc_dw_fin_kernel', 'nc_dw_kernel<false>', 'nc_dw_kernel<true>', 'nc_dw_kernel', 'nc_dx_x3_kernel', 'nc_dx_kernel<true>',
                    'gemm16_kernel<0, 0, 1, true, true>', 'gemm16_kernel<0, 1, 1, true, false>',
                    'gemm16_kernel<1, 1, 4, false, false>', 'adam_kernel', 'train_prologue_kernel') if k in pmc]
txt = f'''# profiles, round 1

All files here are produced by `python tools/summarize_profiles.py {tag}` (+ `tools/write_profiles_readme.py`) from
rocprofv3 output collected on one MI355X (gpurun box) with these commands (`cd /tmp; export TMPDIR=/tmp` first):

* `rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_{tag}b -- python3 bench.py --steps 300 --warmup 30 --no-cpu`
  -> `{tag}_vlsac_b256_kernel_stats.csv` (workload vlsac_halfcheetah_f256_b256, hipGraph replay; 331 train() calls incl.
  capture/warm-up, plus bench.py's roofline loop: 220 extra critic-stage `nc_fwd_x3q_kernel` launches)
* three separate `--pmc` passes, `--kernel-trace` only (`tools/_pmc.sh`, eager launches, 25 train() calls each)
  -> `{tag}_pmc_summary.json` (per-kernel mean over dispatches of the per-dispatch sums)
* un-profiled `python bench.py` on the same box -> `{tag}_bench.json`

| kernel (gemm16 template args: loader A, loader B, NF, 16-byte A, 16-byte B) | calls | avg us | % of GPU time |
|---|---|---|---|
{table}

Under rocprofv3 every dispatch interval includes the inter-kernel boundary (the trace shows zero gaps), so the small
kernels read ~5.5 us each; un-profiled, a chain of trivial graph-captured kernels costs 1.77 us per launch on this box
and the instrumented timeline (`tools/exp/gemm_timeline.py`) puts a gemm16 launch at 3.3-5.7 us of kernel span plus
1.4-2.2 us to the first wave of the next launch.

bench.py (un-profiled): **{b['value']} {b['unit']}**, {b['ms_per_step']} ms per train() -- graph mode with the critic + actor steps of
train(t) on a second stream beside the feature steps of train(t+1) (`config.deferred_critic_actor_branch`; DESIGN.md 2 and 5.0);
`RLREP_PIPELINE=0` gives the strictly sequential graph: 2 020-2 040 train()/s (0.49 ms).  Box-to-box spread over the gpurun pool
is about 5 %.  The kernel table above was collected in the same (pipelined) mode: under rocprofv3 the two streams' kernels are
serialised, so it shows per-kernel durations, not the overlap.

roofline: `{json.dumps(b['roofline'])}`

cpu_baseline: `{json.dumps(b['cpu_baseline'])}`

The noise-critic first layer runs on the bf16 pipe as an exact three-way split (bf16x3, `csrc/x3.h`): `nc_fwd_x3q_kernel` is the critic
step's four-head launch (2.68 GFLOP; also bench.py's roofline loop, 220 launches), `nc_fwd_x3_kernel<1>` the actor step's two-head launch
(1.34 GFLOP), `nc_dx_x3_kernel` the actor step's dL/d(mean, log_std), `nc_dw_x3_kernel` (+ `nc_dw_fin_kernel`, the fixed-order sum of its
split-K slabs) the critic step's weight gradient (1.34 GFLOP).
`roofline.us_per_launch` times the four-head launch alone with HIP events on the launch stream.

## PMC counters (`{tag}_pmc_summary.json`), per launch

| kernel | MFMA insts | WAIT_ANY / WAVE_CYCLES | VALU_MFMA_BUSY / WAVE_CYCLES | FETCH_SIZE MB (raw) | WRITE_SIZE MB | LDS bank-conflict cycles |
|---|---|---|---|---|---|---|
''' + '\n'.join(row(k) for k in keys) + '''

VALU_MFMA_BUSY / (launch time x 1024 SIMDs x clock) is the chip-wide matrix-pipe utilisation: 15.7 M busy cycles (491 520 bf16 MFMAs
of 32 cycles) over 22.6 us x 1024 SIMDs x 2.0 GHz = 34 % for the four-head `nc_fwd_x3q_kernel` launch -- the same product on fp32 MFMA kept
its pipe 60 % busy for 30.8 us.  The x3 kernels are bound by vector-instruction ISSUE and its scheduling, not by the matrix pipe:
SQ_INSTS_VALU / SQ_INSTS_MFMA = 7.5 (x3q, 32x32x16), 6.1 (two-head forward, 16x16x32), 5.9 (dX), 9.2 (dW); tools/exp/nc_timeline.py
puts a 32-deep step at 3.2k cycles for 1.9k of matrix-pipe time.  `roofline.frac` prices the launch against 2500 / 6 = 417 TF (six bf16
MFMA flops per fp32 product).
WRITE_SIZE of the four-head forward, 11 520 KB, is exactly its stores (U 10 485 760 + Hm 1 048 576 + sigma 262 144 bytes): the counter is
calibrated for this kernel's dword stores.  FETCH_SIZE is raw (KB as reported / 1024); the guide's gfx950 rule (x2 for 16-byte-per-lane reads) gives bench.py's `roofline.traffic` =
2 x FETCH + WRITE = 18.5 MB per four-head launch against 11.0 MB algorithmic (10.5 MB of elu outputs U written + inputs; each XCD's L2
fetches its own copy of the 1 MB of weights).  `nc_dw_x3` / `nc_dx_x3` read U (10.5 MB) once with dword / 8-byte loads (uncalibrated /
half-counted access widths in FETCH_SIZE); the column tiles that share rows of U are placed on one XCD.

## Micro-benchmarks behind the design decisions (`tools/exp/*.hip`, run with gpurun; results quoted in DESIGN.md)

* `gridbar`   in-kernel device-wide barrier: 13-41 us per phase at 256-512 workgroups vs 1.9-2.0 us per graph launch
* `mfma_peak` sustained v_mfma_f32_16x16x4_f32 (5 accumulators): 81 / 122 / 126 / 129 TFLOP/s at 1 / 2 / 4 / 8 waves per SIMD (52 cycles per
  MFMA for a wave alone on its SIMD); v_mfma_f32_32x32x2_f32: 83 / 108 / 130 TF with 1 / 2 / 4 accumulators at one wave per SIMD, 143 TF at two
* `kernarg`   dependent scalar loads from the kernel-argument segment: hidden (1.77 us per launch with or without)
* `icache`, `ijump`  cold instruction fetch: no measurable penalty (straight-line or taken branches)
* `loadlat`, `xcdlat`  first load of the previous kernel's output: 470 cycles same XCD, 810-2200 cycles other XCD;
  workgroup -> XCD placement is round-robin on blockIdx.x
'''
# ---- large-dimension workloads and the LDS-tiled GEMM engines ----------------------------------------------------
big = json.load(open(os.path.join(ROOT, 'profiles', f'{tag}_large_workloads_top_kernels.json')))
allb = {json.loads(l)['config']['workload']: json.loads(l) for l in open(os.path.join(ROOT, 'profiles', f'{tag}_bench_all.jsonl'))}
gp = json.load(open(os.path.join(ROOT, 'profiles', f'{tag}_pmc_gemm_4096.json')))
txt += '''
## The other workloads of bench.py (`--workload ...`; un-profiled lines in `''' + tag + '''_bench_all.jsonl`)

| workload | train()/s | ms per train() | algorithmic GFLOP per train() (SURVEY 8d) | fraction of the 157.3 TF fp32 peak |
|---|---|---|---|---|
''' + '\n'.join(f"| `{k}` | {v['value']} | {v['ms_per_step']} | {v.get('algorithmic_gflop_per_train', '')} | {v.get('train_flop_frac_of_fp32_peak', '')} |" for k, v in sorted(allb.items())) + '''

Top kernels of the three large-dimension workloads (`rocprofv3 --kernel-trace --stats -- python3 bench.py --workload W --steps 60|6 --no-cpu`,
full tables in `''' + tag + '''_<workload>_kernel_stats.csv`; `gemm_lds_kernel<tile, loader A, loader B>`, loader 0 = row-major, 1 = k-major):

'''
for wl, rows in big.items():
    txt += f'`{wl}`\n\n| kernel | calls | avg us | % |\n|---|---|---|---|\n' + '\n'.join(f'| `{n}` | {c} | {a:.1f} | {pc:.1f} |' for n, c, a, pc in rows) + '\n\n'
txt += '''## GEMM engines on the path's large shapes (`python tools/bench_gemm.py`, HIP events, `''' + tag + '''_gemm_engines.txt`)

```
''' + open(os.path.join(ROOT, 'profiles', f'{tag}_gemm_engines.txt')).read() + '''```

PMC at 4096^3 (`tools/_pmc_gemm.sh`, two `--pmc` passes with `--kernel-trace` only; `''' + tag + '''_pmc_gemm_4096.json`, per launch, summed over XCDs):

| kernel | MFMA insts | VALU_MFMA_BUSY cycles | GRBM_GUI_ACTIVE / 8 (cycles) | matrix-pipe utilisation = BUSY / (cycles x 1024 SIMDs) | WAIT_INST_LDS / WAVE_CYCLES | LDS bank-conflict / LDS active cycles |
|---|---|---|---|---|---|---|
''' + '\n'.join(f"| `{k}` | {d['SQ_INSTS_MFMA']:.0f} | {d['SQ_VALU_MFMA_BUSY_CYCLES']:.0f} | {d['GRBM_GUI_ACTIVE'] / 8:.0f} | {100 * d['SQ_VALU_MFMA_BUSY_CYCLES'] / (d['GRBM_GUI_ACTIVE'] / 8 * 1024):.0f} % | {100 * d['SQ_WAIT_INST_LDS'] / d['SQ_WAVE_CYCLES']:.0f} % | {100 * d['SQ_LDS_BANK_CONFLICT'] / d['SQ_LDS_IDX_ACTIVE']:.0f} % |" for k, d in gp.items()) + '''

The fp32-MFMA tile keeps the matrix pipe ~77 % busy (123 TF of 157.3); the bf16x3 tile issues 6 bf16 MFMAs per fp32 product and
holds its pipe ~54 % busy at a lower clock (184 TF fp32-equivalent = 1.10 PF executed bf16): it is bound by LDS issue (three
bf16 images per operand: WAIT_INST_LDS) and by the split's VALU work, not by the matrix pipe.
'''
open(os.path.join(ROOT, 'profiles', f'{tag}_README.md'), 'w').write(txt)
print('wrote', f'profiles/{tag}_README.md')
