#!/usr/bin/env python3
"""bench.py -- gradient steps/sec of the rl-rep update hot path on MI355X.

One "step" = one `agent.train(buffer, 256)` of BASELINE.json config[1]: vlsac on HalfCheetah-v3 dims
(S=17, A=6, hidden=256, feature_dim=256, batch=256): 4 feature (VAE-ELBO) steps + critic step + actor and
temperature step + Polyak updates = 7 optimizer steps, on synthetic replay resident in HBM.

    python bench.py --gpus N --steps K --warmup W
N>1 is launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`:
one process per GPU, replay sharded (each rank owns its own ring and samples its own B=256 minibatch:
weak scaling, global batch 256*N), gradients of every optimizer step all-reduced over RCCL/xGMI.

Prints ONE JSON line (rank 0) with the throughput, a `roofline` object for the dominant kernel and a
`cpu_baseline` object (the CPU oracle timed on this box's host cores, N=1 only).
"""
import argparse
import json
import os
import sys
import time

# HIP maps streams onto a few hardware queues (4 by default) and streams that share one are serialised.  The pipelined train() uses two
# streams (plus, with N > 1 ranks, one RCCL stream per process group): give the runtime enough queues, before it initialises.
os.environ.setdefault('GPU_MAX_HW_QUEUES', '16')

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))

WORKLOADS = {
    # name: (agent, S, A, B, ctor kwargs)
    'vlsac_halfcheetah_f256_b256': ('vlsac', 17, 6, 256, dict(hidden_dim=256, feature_dim=256, extra_feature_steps=3)),
    'sac_halfcheetah_b256': ('sac', 17, 6, 256, dict(hidden_dim=256)),
    'sac_pendulum_b64': ('sac', 3, 1, 64, dict(hidden_dim=256)),
    # main.py:87-104 hard-codes these ctrlsac / spedersac dimensions; BASELINE.json names the 256 / 512 variants
    'ctrlsac_halfcheetah_f2048_b256': ('ctrlsac', 17, 6, 256, dict(hidden_dim=1024, feature_dim=2048, extra_feature_steps=3)),
    'ctrlsac_halfcheetah_f256_b256': ('ctrlsac', 17, 6, 256, dict(hidden_dim=256, feature_dim=256, extra_feature_steps=3)),
    'spedersac_ant_f512_b1024': ('spedersac', 111, 8, 1024, dict(
        phi_and_mu_lr=1e-5, phi_hidden_dim=512, phi_hidden_depth=1, mu_hidden_dim=512, mu_hidden_depth=0,
        critic_and_actor_lr=3e-4, critic_and_actor_hidden_dim=256, feature_dim=512, hidden_dim=256, extra_feature_steps=5)),
    'diffsrsac_halfcheetah_b256': ('diffsrsac', 17, 6, 256, dict(hidden_dim=256, extra_feature_steps=3)),
    'diffsrsac_humanoid_b2048': ('diffsrsac', 376, 17, 2048, dict(hidden_dim=256, extra_feature_steps=3)),
}
OPT_STEPS = {'sac': 3, 'vlsac': 7, 'ctrlsac': 7, 'spedersac': 9, 'diffsrsac': 10}
# algorithmic GFLOP per train() (2*MAC; SURVEY.md 8d, "algorithmic" column: the minimum that produces the reference's outputs)
ALG_GFLOP = {'vlsac_halfcheetah_f256_b256': 10.59, 'sac_halfcheetah_b256': 0.582, 'sac_pendulum_b64': 0.136,
             'ctrlsac_halfcheetah_f2048_b256': 59.44, 'ctrlsac_halfcheetah_f256_b256': 2.82, 'spedersac_ant_f512_b1024': 32.88,
             'diffsrsac_halfcheetah_b256': 15.15, 'diffsrsac_humanoid_b2048': 2450.0}
BF16_MFMA_PEAK_TFLOPS = 2500.0    # dense (MI355X_MICROARCH.md); the bf16x3 tile executes 6 bf16 MFMA flops per algorithmic fp32 flop
FP32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
REPLAY_N = 65536

class Space:
    def __init__(self, A):
        self.low, self.high = -np.ones(A, np.float32), np.ones(A, np.float32)


def make_agent(alg, S, A, B, kw):
    import importlib
    name = {'sac': 'SACAgent', 'vlsac': 'VLSACAgent', 'ctrlsac': 'CTRLSACAgent', 'spedersac': 'SPEDERSACAgent',
            'diffsrsac': 'DIFFSRSACAgent'}[alg]
    cls = getattr(importlib.import_module(f'rlrep_amd.agent.{alg}.{alg}_agent'), name)
    return cls(state_dim=S, action_dim=A, action_space=Space(A), max_batch=B, **kw)


def synth_buffer(S, A, seed):
    import synth
    from rlrep_amd.utils.buffer import ReplayBuffer
    data = synth.replay(S, A, REPLAY_N, seed=seed)
    buf = ReplayBuffer(S, A, max_size=REPLAY_N)
    buf.load(data['state'], data['action'], data['next_state'], data['reward'], data['done'])
    return buf, data


def dominant_kernel_roofline(agent, B, F, H, reps=200):
    """Time the heaviest launch of the path (the noise-critic first layer of the critic step: four
    [B*20 x F] x [F x H] products, target+live heads) standalone with HIP events on the launch stream."""
    core = agent.core
    names = core.stages(2)
    st = [i for i, n in enumerate(names) if n.startswith('noise critic l1/l4')]
    if not st:
        return None
    s = st[0]
    for _ in range(20):
        core.run_stage(2, s)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        core.run_stage(2, s)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    flops = 4 * 2.0 * (B * 20) * F * H          # algorithmic: 4 heads x 2*M*K*N
    achieved = flops / (us * 1e-6) / 1e12
    import ctypes as C
    from rlrep_amd import _lib
    plan = [C.c_int32() for _ in range(3)]
    _lib.lib.rlrep_nc_fwd_plan(4, B, F, H, *[C.byref(o) for o in plan])
    x3 = plan[0].value == 1
    # bf16x3 engine: six bf16 MFMA flops are executed per algorithmic fp32 flop, so the ceiling for ALGORITHMIC flops is the dense
    # bf16 peak / 6 (417 TF); on the fp32-MFMA engine (RLREP_DISABLE=x3) it is the fp32 MFMA peak
    peak = BF16_MFMA_PEAK_TFLOPS / 6.0 if x3 else FP32_MFMA_PEAK_TFLOPS
    # the launcher's choice (noisecritic.hip rl_launch_nc_fwd): the 32x32x16 one-role kernel for a 128-wide tile
    kname = ('nc_fwd_x3q_kernel' if plan[2].value == 128 else 'nc_fwd_x3_kernel<%d>' % (plan[2].value // 64)) if x3 else 'nc_fwd_kernel'
    out = {'bound': 'mfma', 'kernel': kname + (' (critic step, 4 heads, bf16x3)' if x3 else ' (critic step, 4 heads)'),
           'achieved': round(achieved, 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s', 'frac': round(achieved / peak, 4),
           # (traffic is not measured inside a timed run: the PMC passes -- FETCH_SIZE x2 for wide reads on gfx950 + WRITE_SIZE -- are
           #  committed under profiles/ and quoted in DESIGN.md)
           'traffic': None, 'us_per_launch': round(us, 2), 'flop_per_launch': flops}
    if x3:
        out['note'] = 'achieved = algorithmic fp32 flops / time; peak = dense bf16 MFMA peak (2500 TF) / 6 executed flops per product'
        out['executed_bf16_tflops'] = round(6 * achieved, 1)
        out['vs_fp32_mfma_peak'] = round(achieved / FP32_MFMA_PEAK_TFLOPS, 4)
    return out


def big_gemm_roofline(agent, B, S, F, Hn, reps=20):
    """diffsrsac: the nabla-mu head forward, [B, Hn] x [Hn, F*S] (202 GFLOP at Humanoid dims), timed as its stage of the
    feature program with HIP events on the launch stream.  It runs on the bf16 pipe as an exact three-way split (bf16x3):
    `achieved` counts ALGORITHMIC fp32 flops against the fp32-MFMA peak; the executed bf16 rate is 6x that."""
    core = agent.core
    names = core.stages(0)
    st = [i for i, n in enumerate(names) if n.startswith('phi / nabla-mu layer')]
    if not st:
        return None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def time_stage(s, n):
        for _ in range(2):
            core.run_stage(0, s)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            core.run_stage(0, s)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n
    # a builder stage is one launch per engine (bf16x3 tile, fp32 tiles, 16-row engine): the head is the longest of them
    us, s = max((time_stage(s, 3), s) for s in st)
    us = time_stage(s, reps)
    flops = 2.0 * B * Hn * (F * S)
    achieved = flops / (us * 1e-6) / 1e12
    peak = BF16_MFMA_PEAK_TFLOPS / 6.0          # six bf16 MFMA flops per algorithmic fp32 flop
    return {'bound': 'mfma', 'kernel': 'gemm_x3w_kernel<row,row> (nabla-mu head forward [B,512]x[512,F*S], bf16x3, 256 x 128 persistent tile)',
            'achieved': round(achieved, 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s', 'frac': round(achieved / peak, 4), 'traffic': None,
            'us_per_launch': round(us, 1), 'flop_per_launch': flops,
            'note': 'achieved = algorithmic fp32 flops / time; peak = dense bf16 MFMA peak (2500 TF) / 6 executed flops per product (a bare v_mfma_f32_32x32x16_bf16 loop on every SIMD sustains 1830-2030 TF at the clock the chip then holds: tools/exp/mfma_bf16.hip)',
            'executed_bf16_tflops': round(6 * achieved, 1), 'vs_fp32_mfma_peak': round(achieved / FP32_MFMA_PEAK_TFLOPS, 4)}


def _cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_baseline(alg, S, A, B, kw, data, threads, budget_s=6.0):
    """The CPU oracle (a from-scratch port of the reference's PyTorch-CPU path, oracle/) on this box's host cores, same workload, same
    loop: a SWEEP over torch thread counts -- 1, 16, 64 and os.cpu_count() (SURVEY.md 8d names the last and 1), plus `--cpu-threads` --
    each a bounded sample; `value` is the best leg, `cores` its thread count, `sweep` every leg.  NOTE: the oracle deduplicates work the
    reference executes (second encoder pass, [B,B,F] broadcast, critic / feature weight gradients in the actor step: 13.35 vs 10.59 GFLOP
    for vlsac), so the reference itself is slower than this number."""
    from oracle import make_oracle
    from oracle.agents import gather_batch
    from oracle.shapes import param_shapes
    import synth
    init = synth.init_like(param_shapes(alg, S, A, **kw))
    for k in list(init):
        for s_, d_ in (('critic', 'critic_target'), ('f', 'f_target'), ('phi', 'phi_target')):
            if k.startswith(s_ + '.') and (d_ + k[len(s_):]) in init:
                init[d_ + k[len(s_):]] = init[k].copy()
    if alg == 'vlsac':
        init['critic_target.noise'] = init['critic.noise'].copy()
    init['log_alpha'] = np.log(np.float64(0.1))
    rs = np.random.RandomState(5)
    F = kw.get('feature_dim', 256)
    nf = kw.get('extra_feature_steps', 0) + 1 if alg != 'sac' else 0
    tens = {k: torch.from_numpy(v) for k, v in data.items()}

    def leg(nthreads, budget):
        torch.set_num_threads(nthreads)
        o = make_oracle(alg, S, A, init, **kw)
        if alg == 'diffsrsac':
            from rlrep_amd.agent.diffsrsac.diffsrsac_agent import generate_alphabars
            o.P['noise_alphabars'] = torch.from_numpy(generate_alphabars(0.3, 0.1, 1000))

        def one():
            nb = o.n_batches()
            idx = [rs.randint(0, REPLAY_N, size=B) for _ in range(nb)]
            eps = []
            if alg == 'vlsac':
                eps = [torch.from_numpy(rs.standard_normal((B, F)).astype(np.float32)) for _ in range(nf)]
            if alg == 'diffsrsac':
                for _ in range(nf):
                    eps += [torch.from_numpy(rs.randint(0, 1000, size=B)), torch.from_numpy((0.449 * rs.standard_normal((B, S))).astype(np.float32))]
            eps += [torch.from_numpy(rs.standard_normal((B, A)).astype(np.float32)) for _ in range(2)]
            o.train([gather_batch(tens, i) for i in idx], eps)
        tw = time.time()
        one()                                           # warm-up (allocator, thread pool)
        slow = time.time() - tw > 1.0                   # (Humanoid: seconds per call -- one warm-up and at least one timed call per leg)
        if not slow:
            one()
        t0 = time.time()
        n = 0
        while n < 200 and (n < (1 if slow else 2) or time.time() - t0 < budget):
            one()
            n += 1
        return n, time.time() - t0

    ncpu = os.cpu_count() or 1
    counts = sorted({c for c in (1, 16, 64, ncpu, threads) if 1 <= c <= ncpu})
    sweep = []
    for c in counts:
        # (measured on the 2 x 64-core box, profiles/r04_cpu_sweep_all_threads.json: with every logical core as a torch thread ONE vlsac
        # train() takes 130 s -- 0.008 /s against 23 /s on 16 threads -- oversubscribed intra-op pools on matrices this small.  A leg whose
        # predecessor already fell below half of the best rate is recorded as skipped instead of burning minutes of the run.)
        best_so_far = max((r['value'] for r in sweep if 'value' in r), default=0.0)
        if c > 16 and sweep and sweep[-1].get('value', 0.0) < 0.5 * best_so_far and os.environ.get('RLREP_CPU_SWEEP_ALL') != '1':
            sweep.append({'threads': c, 'skipped': f"{sweep[-1]['threads']} threads already ran at {sweep[-1]['value']} /s against {best_so_far} /s best: more threads are slower"})
            continue
        n, dt = leg(c, budget_s)
        sweep.append({'threads': c, 'value': round(n / dt, 3), 'calls': n, 'seconds': round(dt, 1)})
    best = max((r for r in sweep if 'value' in r), key=lambda r: r['value'])
    one_thr = next(r for r in sweep if r['threads'] == 1)
    return {'value': best['value'], 'unit': 'train()/s', 'cores': best['threads'], 'kind': 'port',
            'sample': f"{best['calls']} train() calls of the same workload on the CPU oracle (torch {torch.__version__} CPU, "
                      f"{best['threads']} threads of {ncpu} logical cores: the best leg of the sweep), {best['seconds']} s",
            'sweep': sweep,
            'single_thread': {'value': one_thr['value'], 'cores': 1, 'sample': f"{one_thr['calls']} train() calls, {one_thr['seconds']} s"},
            'cpu_model': _cpu_model(),
            'note': 'oracle = deduplicated restatement of the reference (10.59 vs 13.35 executed GFLOP per vlsac train()): the reference '
                    'itself measured 10.3 train()/s on 8 cores of the build container (SURVEY.md section 6)'}


# Kernel families of the step programs, by the engine id the library reports per stage (include/rlrep.h rlrep_stage_info):
# name -> (engine ids, bound, peak, unit, regex over rocprofv3 kernel names: which rows of profiles/*_kernel_stats.csv / *_pmc_*.json belong to it)
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 measured)
FAMILIES = {
    'gemm16': ((1, 2), 'mfma', FP32_MFMA_PEAK_TFLOPS, 'TFLOP/s', r'^(gemm16_kernel|gemm16_fast\w*_kernel|gemm16_duo_kernel|heads_vae_kernel)',
               'gemm16_kernel + gemm16_fast_kernel / gemm16_fast4_kernel / gemm16_fastpre_kernel + gemm16_duo_kernel + heads_vae_kernel (16-row fp32-MFMA tiles: every 256-wide layer forward / dX / dW; the Gaussian heads with vae_mid)'),
    'gemm_lds64': ((3,), 'mfma', FP32_MFMA_PEAK_TFLOPS, 'TFLOP/s', r'^(gemm_lds_kernel<64|gemm_lds_fin_kernel)',
                   'gemm_lds_kernel<64,...> + gemm_lds_fin_kernel (64-wide LDS tiles on fp32 MFMA, split-K slabs + finishing blocks)'),
    'gemm_lds128': ((4,), 'mfma', FP32_MFMA_PEAK_TFLOPS, 'TFLOP/s', r'^gemm_lds_kernel<128', 'gemm_lds_kernel<128,...> (128-wide LDS tiles on fp32 MFMA)'),
    'gemm_x3': ((5,), 'mfma', BF16_MFMA_PEAK_TFLOPS / 6.0, 'TFLOP/s', r'^gemm_x3(s|t|w)?_kernel',
                'gemm_x3w_kernel (256 x 128, persistent) + gemm_x3_kernel / gemm_x3t_kernel (128-wide) + gemm_x3s_kernel (64-wide): tiles on the bf16 pipe, exact 3-way split: peak = dense bf16 peak / 6 executed flops per product'),
    'noise_critic': ((6,), 'mfma', BF16_MFMA_PEAK_TFLOPS / 6.0, 'TFLOP/s', r'^nc_', 'nc_fwd / nc_dx / nc_dw kernels (vlsac noise critic, bf16x3: peak = dense bf16 peak / 6)'),
    'optimizer': ((7,), 'hbm', HBM_PEAK_GBS, 'GB/s', r'^(void )?adam_(l1_|dp_)?kernel', 'adam_kernel / adam_l1_kernel / adam_dp_kernel (Adam + Polyak + metrics + riders: 28 B per parameter + 12 B per target element)'),
    'score': ((8,), 'hbm', HBM_PEAK_GBS, 'GB/s', r'^diffsr_score', 'diffsr_score kernels (one pass over the [B, F*S] tensor)'),
    'other': ((0,), None, None, None, r'.*', 'losses, gathers, copies (elementwise + wave reductions)'),
}
_ENGINE_FAMILY = {e: k for k, v in FAMILIES.items() for e in v[0]}


def _arm_feature_inputs(agent, buf, B):
    """The stage hooks launch a stage with whatever per-call inputs (noise, noise-level indices) the LAST step left armed.  After a train() that is the
    actor step's [B, A] noise -- but a feature-step stage reads [B, F] (vlsac) or [B, S] (diffsrsac: 3 MB at Humanoid dims) from that pointer: far
    beyond the buffer.  One eager feature step arms inputs of the right shape (it performs one more update: harmless after the timed loops)."""
    if agent._feature_iters() <= 0:
        return
    agent.flush()
    buf.flush()
    agent._pool, agent._next_key, agent._early_key = None, {}, None
    agent._feature_once(buf, B, 0, False)
    torch.cuda.synchronize()


def stage_profile(agent, buf, B, reps=40):
    """Every stage of the sequential step programs timed ALONE (hipGraph of `reps` back-to-back launches, HIP events on the launch stream),
    grouped into kernel families by the engine the library reports for the stage (rlrep_stage_info), with the library's own count of the
    stage's algorithmic flops / bytes: launches per train(), us per train(), share, achieved rate against the family's peak.  Standalone
    times are a LOWER bound of what a launch costs inside the dependent chain (there its operands arrive cold from another XCD's L2)."""
    import ctypes as C
    from rlrep_amd._lib import lib, check
    core = agent.core
    nf = agent._feature_iters()
    mult = {0: nf, 1: nf, 2: 1, 3: 1, 4: 1, 5: 1, 6: 1}
    slow = agent.max_batch >= 2048
    if slow:
        reps = 4
    fam = {}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    _arm_feature_inputs(agent, buf, B)          # (feature noise is at least as large as the policy noise the critic / actor stages read)
    for prog in range(7):
        if mult[prog] == 0:
            continue
        for i, name in enumerate(core.stages(prog)):
            eng, fl, by = C.c_int32(), C.c_double(), C.c_double()
            check(lib.rlrep_stage_info(core.h, prog, i, C.byref(eng), C.byref(fl), C.byref(by)), 'stage_info')
            for _ in range(1 if slow else 3):
                core.run_stage(prog, i)
            torch.cuda.synchronize()
            g, st = torch.cuda.CUDAGraph(), torch.cuda.Stream()
            with torch.cuda.graph(g, stream=st):
                for _ in range(reps):
                    core.run_stage(prog, i)
            g.replay()
            torch.cuda.synchronize()
            nrep = 2 if slow else 3
            e0.record()
            for _ in range(nrep):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / (nrep * reps)
            f = fam.setdefault(_ENGINE_FAMILY.get(eng.value, 'other'), {'launches_per_train': 0, 'us_per_train': 0.0, 'gflop': 0.0, 'mbytes': 0.0, 'stages': []})
            f['launches_per_train'] += mult[prog]
            f['us_per_train'] += us * mult[prog]
            f['gflop'] += fl.value * mult[prog] / 1e9
            f['mbytes'] += by.value * mult[prog] / 1e6
            f['stages'].append([name, mult[prog], round(us, 2)])
    tot = sum(f['us_per_train'] for f in fam.values())
    out = []
    for k, f in sorted(fam.items(), key=lambda kv: -kv[1]['us_per_train']):
        _, bound, peak, unit, rx, desc = FAMILIES[k]
        rec = {'family': k, 'kernels': desc, 'kernel_regex': rx, 'launches_per_train': f['launches_per_train'], 'us_per_train': round(f['us_per_train'], 1),
               'us_per_launch': round(f['us_per_train'] / max(f['launches_per_train'], 1), 2),
               'share_of_stage_time': round(f['us_per_train'] / tot, 3),
               'algorithmic_gflop_per_train': round(f['gflop'], 4), 'algorithmic_mbytes_per_train': round(f['mbytes'], 3)}
        if bound == 'mfma' and f['gflop'] > 0:
            tf = f['gflop'] * 1e9 / (f['us_per_train'] * 1e-6) / 1e12
            rec.update({'bound': 'mfma', 'achieved': round(tf, 2), 'peak': round(peak, 1), 'unit': unit, 'frac': round(tf / peak, 4)})
        elif bound == 'hbm' and f['mbytes'] > 0:
            gbs = f['mbytes'] * 1e6 / (f['us_per_train'] * 1e-6) / 1e9
            rec.update({'bound': 'hbm', 'achieved': round(gbs, 1), 'peak': round(peak, 1), 'unit': unit, 'frac': round(gbs / peak, 4)})
        rec['stages'] = f['stages']
        out.append(rec)
    return out


def pmc_traffic(path, fam):
    """HBM-side traffic of a kernel family from a committed PMC summary (tools/summarize_profiles.py: per kernel the mean over dispatches of
    the per-dispatch sums of FETCH_SIZE / WRITE_SIZE, in KB, collected in SEPARATE --pmc passes).  Corrected as MI355X_MICROARCH.md (HBM /
    rocprofv3) prescribes for gfx950: FETCH_SIZE counts 64 B per 128-B request of a wide read -> x2; WRITE_SIZE is exact.  Infinity-Cache
    hits are counted (these are L2-miss bytes, an upper bound of HBM bytes).  -> bytes per launch (dispatch-weighted mean over the family's
    kernels), per train() and the ratio to the family's algorithmic bytes."""
    import re
    try:
        d = json.load(open(path))
    except Exception:
        return None
    rx = re.compile(fam['kernel_regex'])
    # each counter summed over ALL the dispatches that carry it (the passes are separate runs; a kernel missing from one pass is reported in
    # `coverage`, never silently dropped from the mean), then divided by the train() calls of the profiled run
    calls = int(d.get('__meta__', {}).get('train_calls', 0) or 0)
    tot = {'FETCH_SIZE': 0.0, 'WRITE_SIZE': 0.0}
    nd = {'FETCH_SIZE': 0, 'WRITE_SIZE': 0}
    per_kernel = {}
    for k, v in d.items():
        if k.startswith('__') or not rx.search(k):
            continue
        by_pass = v.get('dispatches_by_pass', {})
        for c, sub in (('FETCH_SIZE', 'pmc_fetch'), ('WRITE_SIZE', 'pmc_write')):
            if c in v:
                n = int(by_pass.get(sub, v.get('dispatches', 1)))
                tot[c] += v[c] * 1024.0 * n
                nd[c] += n
        if 'FETCH_SIZE' in v and 'WRITE_SIZE' in v:
            per_kernel[k] = round((2.0 * v['FETCH_SIZE'] + v['WRITE_SIZE']) * 1024.0)
    if not nd['FETCH_SIZE'] or not nd['WRITE_SIZE'] or not calls:
        return None
    per_train = (2.0 * tot['FETCH_SIZE'] + tot['WRITE_SIZE']) / calls
    launches = nd['FETCH_SIZE'] / calls                      # launches of the family per train() IN THE PROFILED RUN (see profiled_form)
    per_launch = per_train / launches
    alg = fam['algorithmic_mbytes_per_train'] * 1e6
    return {'bytes_per_launch': round(per_launch), 'bytes_per_train': round(per_train), 'algorithmic_bytes_per_train': round(alg),
            'ratio_to_algorithmic': round(per_train / alg, 2) if alg > 0 else None,
            'source': os.path.relpath(path, ROOT), 'correction': '(2 x FETCH_SIZE + WRITE_SIZE) x 1024 B, each counter summed over every dispatch of its pass / train() calls of the run; separate --pmc passes; Infinity-Cache hits counted',
            'dispatches': nd['FETCH_SIZE'], 'train_calls': calls, 'launches_per_train_profiled': round(launches, 2),
            # 'graph': the PMC passes ran the form that is timed (hipGraph replay, two chains); 'eager': --no-graph (same kernels, sequential programs)
            'profiled_form': d.get('__meta__', {}).get('form', 'eager'),
            'coverage': {'fetch_dispatches': nd['FETCH_SIZE'], 'write_dispatches': nd['WRITE_SIZE'], 'passes_agree': d.get('__meta__', {}).get('passes_agree'),
                         'expected_dispatches': fam['launches_per_train'] * calls}}


def find_pmc_json(workload, explicit=None):
    """profiles/rNN_pmc_<workload>.json of the latest round that has one (rNN_pmc_summary.json = the headline workload)."""
    import glob
    if explicit:
        return explicit if os.path.exists(explicit) else None
    c = sorted(glob.glob(os.path.join(ROOT, 'profiles', f'r*_pmc_{workload}.json')))
    if not c and workload == 'vlsac_halfcheetah_f256_b256':
        c = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_summary.json')))
    return c[-1] if c else None


def family_roofline(fam, pmc_path):
    """The `roofline` object of the bench line for one kernel family of stage_profile()."""
    r = {'bound': fam.get('bound'), 'kernel': fam['kernels'], 'family': fam['family'], 'achieved': fam.get('achieved'), 'peak': fam.get('peak'),
         'unit': fam.get('unit'), 'frac': fam.get('frac'), 'traffic': None, 'us_per_launch': fam['us_per_launch'], 'us_per_train': fam['us_per_train'],
         'launches_per_train': fam['launches_per_train'], 'share_of_gpu_time': fam['share_of_stage_time'],
         'algorithmic_gflop_per_train': fam['algorithmic_gflop_per_train'], 'algorithmic_mbytes_per_train': fam['algorithmic_mbytes_per_train'],
         'note': 'achieved = algorithmic flops (bytes) of the family per train(), as the library counts them per stage (rlrep_stage_info), / the time its '
                 'launches take (each stage timed alone as back-to-back launches inside this run, HIP events on the launch stream); traffic = '
                 'bytes per launch from the committed PMC summary (not collectable inside a timed run)'}
    tr = pmc_traffic(pmc_path, fam) if pmc_path else None
    if tr:
        r['traffic'] = tr['bytes_per_launch']
        r['traffic_detail'] = tr
    return r


def chain_times(agent, reps=60):
    """The two launch chains of the pipelined train() replayed ALONE: the feature chain (critical path) and the critic + actor chain."""
    P = getattr(agent, '_pipe', None)
    if not P or 'fs' not in P:
        return None
    agent.flush()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    out = {}
    for key, graphs in (('feature_chain_us', P['fs']), ('critic_actor_chain_us', P['ca'])):
        g = graphs[0]
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        out[key] = round(e0.elapsed_time(e1) * 1e3 / reps, 1)
    if 'launches' in P:
        out['feature_chain_launches'], out['critic_actor_chain_launches'] = int(P['launches'][0]), int(P['launches'][1])
    return out


_PROBE_CODE = r"""
import json, os, subprocess, sys
line = sys.stdin.readline()                       # parked until the benchmark asks (or goes away: EOF)
if line.strip() == 'go':
    exe = sys.argv[1]
    try:
        # (run the script with this interpreter: its `#!/usr/bin/env python3` line would be two more exec hops)
        out = subprocess.run([sys.executable, exe, '--showclocks', '--json'], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=25).stdout
    except Exception as e:
        out = json.dumps({'error': str(e)[:80]})
    sys.stdout.write(out)
"""


def _spawn_clock_probe(under_profiler):
    """A parked helper process that will run rocm-smi when asked.  It is started FIRST THING, before this process has made any HIP call: a
    process that has initialised the GPU must not fork + exec another program on this pool (and under rocprofv3 the profiler's preloaded
    library has initialised it before main() runs: no probe at all then).  The helper itself never touches HIP."""
    import shutil
    import subprocess
    if under_profiler:
        return None
    exe = shutil.which('rocm-smi') or '/opt/rocm/bin/rocm-smi'
    exe = os.path.realpath(exe)
    if not os.path.exists(exe):
        return None
    try:
        with open(exe, 'rb') as f:
            if b'python' not in f.readline():
                return None
        return subprocess.Popen([sys.executable, '-c', _PROBE_CODE, exe], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    except OSError:
        return None


def _start_clock_probe(p):
    """Tell the parked helper to sample the clocks now (while the caller keeps the GPU busy)."""
    if p is None:
        return None
    try:
        p.stdin.write('go\n'); p.stdin.flush()
    except Exception:
        return None
    return p


def _read_clock_probe(p):
    if p is None:
        return None
    try:
        out, _ = p.communicate(timeout=30)
        d = json.loads(out)
        card = d.get('card0') or next(iter(d.values()))
        pick = {k: v for k, v in card.items() if 'sclk' in k.lower() or 'mclk' in k.lower() or 'fclk' in k.lower()}
        return {'source': 'rocm-smi --showclocks, sampled while the median repeats ran', **pick}
    except Exception as e:           # the probe must never fail the benchmark
        return {'source': 'rocm-smi --showclocks', 'error': str(e)[:80]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2000)
    ap.add_argument('--warmup', type=int, default=200)
    ap.add_argument('--workload', default='vlsac_halfcheetah_f256_b256', choices=sorted(WORKLOADS))
    ap.add_argument('--cpu-threads', type=int, default=int(os.environ.get('RLREP_CPU_THREADS', 16)))
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--no-graph', action='store_true')
    ap.add_argument('--no-profile', action='store_true', help='skip the per-stage / per-chain timing loops (rocprofv3 and PMC runs)')
    ap.add_argument('--scaling', choices=['weak', 'strong'], default='weak',
                    help='N > 1 data-parallel: weak = every rank its own batch B (global batch N*B); strong = the global batch stays B, each rank samples B/N '
                         '(SURVEY.md 8e; BASELINE configs 4 and 5 name a total batch on 4 / 8 GPUs)')
    ap.add_argument('--replicas', action='store_true',
                    help='N > 1: N independent agents (own parameters, own replay, NO gradient all-reduce) instead of data-parallel training')
    ap.add_argument('--dp-form', choices=['fused', 'segments', 'captured', 'captured_two_chain'], default=None,
                    help='N > 1: run ONLY this data-parallel form.  fused: every exchange inside the launches (csrc/dp_pull.h); segments: hipGraph segments '
                         'around eager collectives; captured: the RCCL all-reduces captured into the train() graph; captured_two_chain: the same on the two '
                         'chains with one communicator each (vlsac).  Default: --dp-sweep.')
    ap.add_argument('--dp-sweep', choices=['safe', 'all'], default='safe',
                    help='N > 1 without --dp-form: the forms that are timed, each over the whole warm-up + --steps protocol with a replicas_identical '
                         'check; `value` is the fastest one whose replicas stayed identical, all are listed in `dp_forms`.  safe: fused + segments (bounded '
                         'waits / eager collectives); all: + the captured RCCL forms (rehearsed with one rank only: a hang there is RCCL\'s to time out)')
    ap.add_argument('--no-extra-warmup', action='store_true', help='only the --warmup calls before the timed window (tests: a short run is a protocol check, not a measurement)')
    ap.add_argument('--quick', action='store_true', help='only the warm-up and the --steps window (no median repeats, no add / metric-fetch / main-loop legs): profiler runs')
    ap.add_argument('--pmc-json', default=None, help='PMC summary to take roofline.traffic from (default: the latest profiles/r*_pmc_<workload>.json)')
    args = ap.parse_args()
    under_profiler = args.no_profile or any(k.startswith(('ROCPROF', 'ROCP_')) for k in os.environ) or 'rocprof' in os.environ.get('LD_PRELOAD', '')
    clock_helper = _spawn_clock_probe(under_profiler) if (args.gpus == 1 and int(os.environ.get('RANK', 0)) == 0) else None

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: become the launcher.  Nothing in this process has touched the GPU yet (no HIP call, no
        # torch.cuda query), the ranks are CHILD processes of torch.distributed.run and rank 0's JSON line is relayed as is.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(('127.0.0.1', 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
               '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd).returncode)
    rank = int(os.environ.get('RANK', 0))
    local = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if world != args.gpus and not (world == 1 and args.gpus == 1):
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    replicas = args.replicas and world > 1          # N independent B-sized agents, no collective: the literal "steps at batch=B" reading
    torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
    dist = None
    force_dp = world == 1 and os.environ.get('RLREP_FORCE_DP') == '1'    # rehearsal: DP step forms over a one-rank RCCL group
    if force_dp:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29531')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
    if world > 1 or force_dp:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        backend = os.environ.get('RLREP_DIST_BACKEND', 'nccl')      # 'gloo' lets two ranks share one GPU in tests
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local), timeout=datetime.timedelta(seconds=300))
        else:
            dist.init_process_group(backend, timeout=datetime.timedelta(seconds=300))
    if args.no_graph:
        os.environ['RLREP_GRAPH'] = '0'

    alg, S, A, B, kw = WORKLOADS[args.workload]
    strong = args.scaling == 'strong' and world > 1 and not replicas
    B_global = B
    if strong:
        if B % world:
            raise SystemExit(f'--scaling strong: batch {B} is not a multiple of {world} ranks')
        B = B // world              # per-rank minibatch; the loss kernels scale by 1 / (B * world) = 1 / B_global
    torch.manual_seed(0)
    if replicas:
        # independent replicas: the agent must not see the process group (it would broadcast parameters and all-reduce gradients)
        import rlrep_amd.agent.sac.sac_agent as _sa
        _sa._world = lambda: (1, 0)
        torch.manual_seed(rank)
    buf, data = synth_buffer(S, A, seed=rank)
    n_calls = [0]

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def replicas_check(agent):
        """parameter / target arenas and the fp64 temperature state equal on every rank: wrap-around int64 sums of the bit patterns, max == min"""
        agent.flush()
        torch.cuda.synchronize()
        chk = torch.stack([agent.core.params.view(torch.int32).to(torch.int64).sum(), agent.core.targets.view(torch.int32).to(torch.int64).sum(),
                           agent.core.alpha_state.view(torch.int64).sum()]).to(torch.float64)
        hi, lo = chk.clone(), chk.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        return bool((hi == lo).all().item())

    # untimed: the W warm-up steps the caller asked for, and at least ~0.15 s of train() calls on top (graph instantiation, the two-stream
    # pipeline's probe, the clock ramp: with the driver's W = 5 the 20-step window otherwise sits 6 % below the steady state)
    extra = 0 if args.no_extra_warmup else max(0, (300 if alg in ('vlsac', 'sac', 'ctrlsac') and B <= 256 else 10) - args.warmup)

    def build_and_time(form):
        """One agent in data-parallel form `form` (None: whatever the environment says), the driver's protocol: W (+ extra) untimed calls, then
        EXACTLY --steps calls between barriers, the MAX over ranks."""
        if form is not None:
            os.environ['RLREP_DP_FUSED'] = '1' if form == 'fused' else '0'
            os.environ['RLREP_DP_CAPTURE'] = '1' if form in ('captured', 'captured_two_chain') else '0'
            os.environ['RLREP_PIPELINE_DP'] = '1' if form == 'captured_two_chain' else '0'
        torch.manual_seed(rank if replicas else 0)
        agent = make_agent(alg, S, A, B, kw)
        # every train() call of this process is counted: a rocprofv3 kernel-stats / PMC table of the same command divides its `Calls` by it
        _train = agent.train

        def _counted(*a, **k):
            n_calls[0] += 1
            return _train(*a, **k)
        agent.train = _counted
        for _ in range(args.warmup + extra):
            agent.train(buf, B)
        agent.flush()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            agent.train(buf, B)
        agent.flush()           # pipelined graph mode: the last train()'s critic / actor steps belong inside the timed window
        barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([dt], dtype=torch.float64, device='cuda')
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return agent, dt

    def form_of(agent):
        """what the agent actually runs (the fused form falls back to torch.distributed when its probe fails on this set of ranks)"""
        if getattr(agent, '_fused_all', False):
            return 'fused'
        if getattr(agent, '_pipe', None) is not None and agent._pipe.get('mode') == 3:
            return 'captured_two_chain'
        if getattr(agent, '_seg_capture_colls', False):
            return 'captured'
        return 'segments' if isinstance(getattr(agent, '_graph', None), list) else ('fused+' if getattr(agent.core, 'fused_groups', None) else 'eager')

    dp_forms = None
    if dist is not None and world > 1 and not replicas and not force_dp:
        # N > 1: every available data-parallel form over the whole protocol; `value` = the fastest one whose replicas are identical
        nccl = dist.get_backend() == 'nccl'
        if 'dp_timeout_s' not in os.environ.get('RLREP_ENABLE', ''):          # (a form that stalls on this node must fail within the bench's minutes)
            os.environ['RLREP_ENABLE'] = ','.join(t for t in (os.environ.get('RLREP_ENABLE', ''), 'dp_timeout_s=30') if t)
        forms = [args.dp_form] if args.dp_form else (['fused', 'segments'] + (['captured'] + (['captured_two_chain'] if alg == 'vlsac' else []) if (nccl and args.dp_sweep == 'all') else []))
        dp_forms, best = [], None
        for form in forms:
            rec = {'form': form}
            agent_f, dt_f, ok = None, None, True
            try:
                agent_f, dt_f = build_and_time(form)
                rec['ran_as'] = form_of(agent_f)
                rec['replicas_identical'] = replicas_check(agent_f)
            except Exception as e:          # noqa: BLE001  (every rank must drop the form together: agreed below)
                ok, rec['error'] = False, f'{type(e).__name__}: {str(e)[:200]}'
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device='cuda' if nccl else 'cpu')
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = bool(int(flag.item()))
            if ok:
                rec.update(value=round((1 if args.scaling == 'strong' else world) * args.steps / dt_f, 2), ms_per_step=round(dt_f / args.steps * 1e3, 4),
                           fused_groups=sorted(agent_f.core.fused_groups) if getattr(agent_f.core, 'fused_groups', None) else None)
                if rec['replicas_identical'] and (best is None or dt_f < best[1]):
                    best = (agent_f, dt_f, form)
            elif 'error' not in rec:
                rec['error'] = 'failed on another rank'
            dp_forms.append(rec)
        if best is None:
            raise SystemExit('bench: no data-parallel form completed with identical replicas: ' + json.dumps(dp_forms))
        agent, dt = best[0], best[1]
        for d in dp_forms:
            d['chosen'] = d['form'] == best[2]
    else:
        agent, dt = build_and_time(args.dp_form if (dist is not None and args.dp_form) else None)
    # median of 5 repeats.  The --steps window above is what `value` reports (the driver's contract: EXACTLY K steps); a short
    # window (the driver's 20 steps = 7 ms) is noisy by construction, so the same loop is also timed as 5 repeats of `rep_len` calls and the
    # median reported beside it.  The shader clock is sampled by rocm-smi WHILE those repeats run.
    rep_len = 500 if dt / args.steps < 2e-3 else max(20, min(args.steps, 100))
    smi = _start_clock_probe(clock_helper) if rank == 0 else None
    rep_rates = []
    quick = args.quick          # profiler runs: only the warm-up and the --steps window (clean launch counts, short traces)
    for _ in range(0 if quick else 5):
        barrier()
        tr = time.perf_counter()
        for _ in range(rep_len):
            agent.train(buf, B)
        agent.flush()
        barrier()
        dr = time.perf_counter() - tr
        if dist is not None:
            tt = torch.tensor([dr], dtype=torch.float64, device='cuda')
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dr = float(tt.item())
        rep_rates.append((1 if strong else world) * rep_len / dr)
    clocks = _read_clock_probe(smi) if rank == 0 else None
    info = agent.train(buf, B)
    finite = all(np.isfinite(float(v)) for v in info.values())
    # the same loop with the metric dict READ after every train() (a device sync per call, as the reference's .item()s do)
    n_sync = 0 if quick else min(args.steps, 300)
    barrier()
    t1 = time.perf_counter()
    for _ in range(n_sync):
        float(agent.train(buf, B)[next(iter(info))])
    barrier()
    dt_sync = time.perf_counter() - t1
    # ... and with a ReplayBuffer.add() before every train() (main.py:137-144's call pattern without the env step: the staged row and the size
    # scalar travel to the device ring inside the timed loop)
    n_add = 0 if quick else min(args.steps, 500)
    zs, za = np.zeros(S, np.float32), np.zeros(A, np.float32)
    for _ in range(0 if quick else 20):
        buf.add(zs, za, zs, 0.0, 0.0); agent.train(buf, B)
    agent.flush()
    barrier()
    t2 = time.perf_counter()
    for _ in range(n_add):
        buf.add(zs, za, zs, 0.0, 0.0)
        agent.train(buf, B)
    agent.flush()
    barrier()
    dt_add = time.perf_counter() - t2
    # ... and main.py:126-144's whole iteration without the environment step: select_action (needs the finished actor: a device sync), add, train
    n_loop = min(args.steps, 300) if (world == 1 and not quick) else 0
    dt_loop = None
    if n_loop:
        for _ in range(10):
            act = agent.select_action(zs, explore=True); buf.add(zs, act, zs, 0.0, 0.0); agent.train(buf, B)
        agent.flush()
        barrier()
        t3 = time.perf_counter()
        for _ in range(n_loop):
            act = agent.select_action(zs, explore=True)
            buf.add(zs, act, zs, 0.0, 0.0)
            agent.train(buf, B)
        agent.flush()
        barrier()
        dt_loop = time.perf_counter() - t3

    # N > 1 self-checks: are the replicas still bit-identical after everything above, and what do the gradient
    # all-reduces of one train() cost when nothing else runs
    replicas_identical, allreduce_us = None, None
    if dist is not None and not replicas:
        replicas_identical = replicas_check(agent)
        colls = [x for kind, x in (agent._graph or []) if kind == 'coll'] if isinstance(getattr(agent, '_graph', None), list) else []
        if colls:
            for fn in colls:
                fn()
            barrier()
            tc = time.perf_counter()
            nrep = 10
            for _ in range(nrep):
                for fn in colls:
                    fn()
            barrier()
            allreduce_us = (time.perf_counter() - tc) / nrep * 1e6
            agent.core.grads.zero_()            # (the repeated sums are garbage; the next backward rewrites every gradient anyway)

    if rank == 0:
        updates = args.steps / dt                      # synchronized train() calls per second (each rank performs every one of them)
        value = updates if strong else world * updates       # strong scaling: one synchronized update IS one batch-B_global gradient step
        mode = 'single' if world == 1 and not force_dp else ('replicas' if replicas else 'dp')
        out = {
            'metric': f'gradient steps/sec (encoder+critic+actor) at batch={B_global}',
            'value': round(value, 2), 'unit': 'train()/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'warmup_extra_untimed': extra,
            'ms_per_step': round(dt / args.steps * 1e3, 4), 'higher_is_better': True, 'scaling': 'strong' if strong else 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': args.workload, 'agent': alg, 'state_dim': S, 'action_dim': A, 'batch_per_gpu': B,
                       'global_batch': (B_global if strong else B * world) if mode == 'dp' else B, 'scaling': 'strong' if strong else 'weak', 'feature_dim': kw.get('feature_dim'), 'hidden_dim': kw.get('hidden_dim'),
                       'feature_steps_per_train': (kw.get('extra_feature_steps', -1) + 1), 'replay_rows_per_gpu': REPLAY_N,
                       'parallelism': {'single': 'single GPU',
                                       'dp': f'dp{world} (replay sharded, ' + (('gradients summed over the ranks INSIDE the optimizer launches: peer-mapped gradient arenas, rank-ordered sums, no collective launch' + ('' if getattr(agent, '_fused_all', False) else '; larger slices / batch-coupled exchanges over ' + (dist.get_backend() if dist is not None else '-')))
                                                                                if getattr(agent.core, 'fused_groups', None) else
                                                                                (('RCCL' if (dist is not None and dist.get_backend() == 'nccl') else (dist.get_backend() if dist is not None else 'no')) + ' all-reduce of gradients per optimizer step')) + ')',
                                       'replicas': f'{world} independent agents (own parameters and replay, no collective)'}[mode],
                       'hipgraph': (bool(agent.use_graph) if (not agent._dp or getattr(agent, '_fused_all', False)) else
                                    (('collectives captured into the graph(s)' if getattr(agent, '_seg_capture_colls', False) else 'segments between eager collectives')
                                     if agent.use_graph and agent.use_graph_dp else False)),
                       # critic + actor steps of train(t) as a graph branch beside the feature steps of train(t+1) (same updates, same order)
                       'deferred_critic_actor_branch': bool(getattr(agent, '_pipe', None))},
            # what `value` counts: with N > 1 data-parallel ranks every rank performs the same `global_updates_per_sec` train() calls, each
            # on its own batch-B shard of a global batch N*B; value = N * global_updates_per_sec = samples_per_sec / B ("batch-B gradient
            # steps" worth of samples).  --replicas runs N independent batch-B agents instead: value = their train() calls summed.
            'value_definition': {'single': 'train() calls per second at batch B',
                                 'dp': ('synchronized global updates per second; one global update = the batch B split over world_size ranks' if strong else
                                        'world_size x synchronized global updates per second (= samples_per_sec / B); one global update = batch B x world_size'),
                                 'replicas': 'sum over the independent replicas of their train() calls per second at batch B'}[mode],
            'global_updates_per_sec': round(updates if mode != 'replicas' else value, 2),
            'rccl_world_size': (dist.get_world_size() if dist is not None else 1),
            # N > 1: parameter / target / temperature checksums equal on every rank after the run (max == min over ranks); the gradient
            # all-reduces of ONE train() issued back to back with nothing else running (segments form only: in the captured form they are graph nodes)
            'replicas_identical': replicas_identical,
            'allreduce_us_per_train': (round(allreduce_us, 1) if allreduce_us is not None else None),
            'dp_fused_groups': (sorted(agent.core.fused_groups) if getattr(agent.core, 'fused_groups', None) else None),
            # N > 1: every data-parallel form that was timed over the same protocol (form asked for, form actually run, value, replicas_identical);
            # `value` above is the chosen one -- the fastest with identical replicas
            'dp_forms': dp_forms,
            'optimizer_steps_per_sec': round(value * OPT_STEPS[alg], 1),
            'samples_per_sec': round(value * B_global if strong else value * B, 1),
            'metrics_finite': bool(finite),
            'total_train_calls': n_calls[0],
            'value_with_per_step_metric_fetch': (round((1 if strong else world) * n_sync / dt_sync, 2) if n_sync else None),
            'value_with_replay_add_per_call': (round((1 if strong else world) * n_add / dt_add, 2) if n_add else None),
            'main_loop_iterations_per_sec': (round(n_loop / dt_loop, 2) if dt_loop else None),        # select_action + add + train, no environment
            # median of 5 repeats of `rep_len` calls each (same loop, same barriers): the low-noise companion of the --steps window
            'value_median_500' if rep_len == 500 else 'value_median_repeats': (round(float(np.median(rep_rates)), 2) if rep_rates else None),
            'repeats': {'n': len(rep_rates), 'calls_each': rep_len, 'values': [round(v, 1) for v in rep_rates]},
            'clocks': clocks,
        }
        # `roofline` describes the RUN: the kernel family with the largest share of GPU time -- its algorithmic flops (bytes), counted by the
        # library per stage, over the time its launches take inside this run, against the peak that bounds it; `traffic` from the committed
        # PMC summary of this workload.  Every family is listed in `kernel_families`.
        pmc_path = find_pmc_json(args.workload, args.pmc_json)
        if world == 1 and not args.no_profile:
            fams = stage_profile(agent, buf, B)
            out['kernel_families'] = [{k: v for k, v in f.items() if k != 'stages'} for f in fams]
            out['stage_times_us'] = {f['family']: f['stages'] for f in fams}
            priced = [f for f in fams if 'frac' in f]
            if priced:
                out['roofline'] = family_roofline(priced[0], pmc_path)
            out['chains'] = chain_times(agent)
            if out['chains']:
                out['critical_path_us'] = out['chains']['feature_chain_us']
                out['launches_per_train'] = out['chains'].get('feature_chain_launches', 0) + out['chains'].get('critic_actor_chain_launches', 0)
            elif getattr(agent, '_graph_launches', None):
                out['launches_per_train'] = int(agent._graph_launches)
        if 'launches_per_train' not in out:          # (N > 1, or --no-profile: the count needs no timing)
            P = getattr(agent, '_pipe', None)
            if P and 'launches' in P:
                out['launches_per_train'] = int(P['launches'][0]) + int(P['launches'][1])
            elif getattr(agent, '_graph_launches', None):
                out['launches_per_train'] = int(agent._graph_launches)
        # which front end the 16-row tile engine's launches of ONE train() got when the running graphs were captured (rlrep_front_end_counts):
        # fast / fast4 / fastpre issue their operand loads from preloaded scalars, `record` fetches its task record first
        fe = (getattr(agent, '_pipe', None) or {}).get('front_ends') or getattr(agent, '_graph_front_ends', None)
        if fe:
            out['front_end_launches'] = fe
            td = (out.get('roofline') or {}).get('traffic_detail')
            if td and td.get('profiled_form') == 'graph' and (out.get('roofline') or {}).get('family') == 'gemm16':
                # the PMC passes ran the TIMED form (graph replay): its family launches per train() = the 16-row engine's launches in the captured graphs
                # (+ the heads_vae launch of every vlsac feature step); the stage count above it (`launches_per_train` of the family) is that of the
                # sequential step programs, in which the chained feature steps' first layers are launches of their own
                n_timed = sum(int(v) for v in fe.values()) + ((kw.get('extra_feature_steps', -1) + 1) if alg == 'vlsac' else 0)
                td['launches_per_train_timed_form'] = n_timed
                td['coverage']['expected_dispatches'] = n_timed * int(td['train_calls'])
        if alg == 'vlsac' and not args.quick:
            out['roofline_heaviest_kernel'] = dominant_kernel_roofline(agent, B, kw['feature_dim'], kw['hidden_dim'])
            out.setdefault('roofline', out['roofline_heaviest_kernel'])
        elif alg == 'diffsrsac' and not args.quick:
            _arm_feature_inputs(agent, buf, B)
            out['roofline_heaviest_kernel'] = big_gemm_roofline(agent, B, S, 256, 512)
            out.setdefault('roofline', out['roofline_heaviest_kernel'])
        # whole-train() view: algorithmic GFLOP per train() per GPU against the fp32 peak
        gf = ALG_GFLOP.get(args.workload)
        if gf:
            out['algorithmic_gflop_per_train'] = gf
            out['train_flop_frac_of_fp32_peak'] = round(gf * 1e9 * value / world / (FP32_MFMA_PEAK_TFLOPS * 1e12), 4)
        if world == 1 and not args.no_cpu:
            out['cpu_baseline'] = cpu_baseline(alg, S, A, B, kw, data, args.cpu_threads)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
