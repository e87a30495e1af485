"""GPU unit tests of the path's two GEMM engines through the C ABI (rlrep_gemm): the 16-row tile engine (gemm16.hip)
and the LDS-tiled engine (gemm_lds.hip) against a float64 NumPy product, on every operand layout the step programs
use (forward X W^T, dX = G W, dW = G^T X), every generic epilogue, ragged tile edges and every split-K plan.

Tolerance: 1e-5 relative L2 per output (fp32 MFMA is an exact fma chain; the only difference to NumPy is summation
order), well inside the 1e-4 parity bar of BASELINE.json."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ACT = {'none': 0, 'relu': 1, 'elu': 2, 'sin': 3, 'tanh': 4}


def _act(x, a):
    if a == 'relu':
        return np.maximum(x, 0)
    if a == 'elu':
        return np.where(x > 0, x, np.expm1(np.minimum(x, 0)))
    if a == 'sin':
        return np.sin(x)
    if a == 'tanh':
        return np.tanh(x)
    return x


def _dact(aux, a):
    if a == 'relu':
        return (aux > 0).astype(np.float64)
    if a == 'elu':
        return np.where(aux > 0, 1.0, aux + 1.0)
    if a == 'sin':
        return np.cos(aux)
    if a == 'tanh':
        return 1.0 - aux * aux
    return np.ones_like(aux)


def _dev(x):
    return torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32)).cuda()


def _ptr(t):
    return t.data_ptr() if t is not None else None


def run_gemm(engine, mode, R, Cn, K, act='none', accum=False, bias=True, bt=0, splits=0, seed=0, inline_fin=False):
    """mode: 'fwd' (A [R,K], B [Cn,K]), 'dx' (A [R,K], B [K,Cn]), 'dw' (A [K,R], B [K,Cn]).  Returns (got, want, extra)."""
    from rlrep_amd import _lib
    rs = np.random.RandomState(seed)
    la, lb = {'fwd': (0, 0), 'dx': (0, 1), 'dw': (1, 1)}[mode]
    A = rs.standard_normal((K, R) if la else (R, K)).astype(np.float32)
    Bm = rs.standard_normal((K, Cn) if lb else (Cn, K)).astype(np.float32)
    A64 = (A.T if la else A).astype(np.float64)
    B64 = (Bm.T if lb else Bm).astype(np.float64)
    A = (A / np.sqrt(K)).astype(np.float32)           # keep outputs O(1)
    prod = (A.T if la else A).astype(np.float64) @ B64.T
    C0 = rs.standard_normal((R, Cn)).astype(np.float32)
    dA, dB, dC = _dev(A), _dev(Bm), _dev(C0)
    bias_v = aux = out2 = None
    epi = {'fwd': 0, 'dx': 1, 'dw': 3}[mode]
    flags = (1 if accum else 0) | (4 if inline_fin else 0)
    extra_want = None
    if mode == 'fwd':
        bias_v = rs.standard_normal(Cn).astype(np.float32) if bias else None
        pre = prod + (bias_v.astype(np.float64) if bias else 0.0)
        want = _act(pre, act)
        out2 = torch.zeros(R, Cn, device='cuda')
        if act == 'sin':
            extra_want = pre
    elif mode == 'dx':
        aux = rs.standard_normal((R, Cn)).astype(np.float32)
        want = prod * _dact(aux.astype(np.float64), act) + (C0 if accum else 0.0)
    else:
        want = prod + (C0 if accum else 0.0)
        out2 = torch.zeros(R, device='cuda')
        flags |= 2
        extra_want = (A.T if la else A).astype(np.float64).sum(axis=1)
    d_bias = _dev(bias_v) if bias_v is not None else None
    d_aux = _dev(aux) if aux is not None else None
    # split-K slabs: splits * R * (Cn + 1) floats; the automatic plan never exceeds ~256 tiles' worth beyond the output
    ws_floats = (splits * R * (Cn + 5) + 4096) if splits else min(32 * R * (Cn + 5), 10_000_000 + 2 * (R + 128) * (Cn + 133))
    ws = torch.zeros(max(1, ws_floats), device='cuda')
    rc = _lib.lib.rlrep_gemm(engine, la, lb, _ptr(dA), dA.shape[1], _ptr(dB), dB.shape[1], _ptr(dC), Cn, R, Cn, K, epi, ACT[act], flags,
                             _ptr(d_bias), _ptr(d_aux), Cn, _ptr(out2), bt, splits, _ptr(ws), ws.numel(),
                             torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, 'gemm')
    torch.cuda.synchronize()
    got = dC.cpu().numpy().astype(np.float64)
    extra = None
    if extra_want is not None:
        extra = (out2.cpu().numpy().astype(np.float64), extra_want)
    return got, want, extra


def rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


def check(engine, mode, R, Cn, K, **kw):
    got, want, extra = run_gemm(engine, mode, R, Cn, K, **kw)
    assert np.all(np.isfinite(got)), (engine, mode, R, Cn, K, kw)
    assert rel(got, want) < 1e-5, (engine, mode, R, Cn, K, kw, rel(got, want))
    if extra is not None:
        assert rel(extra[0], extra[1]) < 1e-5, ('second output', engine, mode, R, Cn, K, kw, rel(extra[0], extra[1]))


@pytest.mark.parametrize('mode', ['fwd', 'dx'])
def test_bf16x3_32_by_32_tile_with_k_split_over_its_waves_is_fp32_accurate(mode):
    """gemm_x3q_kernel (csrc/gemm_x3q.h): 32 x 32 output tiles, the four waves of a workgroup take a quarter of K each through private LDS
    patches, partial tiles added in quarter order -- ctrlsac's M = 256 layers at main.py's dimensions without slabs or a finishing launch.
    fp32 accuracy against float64 (bf16x3: exact three-way split), ragged rows / columns / quarters, every epilogue of its forms, bit-identical
    reruns, and through the automatic plan."""
    act = 'elu'
    check(2, mode, 256, 1024, 1024, bt=32, act=act, seed=101)
    check(2, mode, 256, 2048, 1024, bt=32, act='relu', seed=102)
    check(2, mode, 256, 1024, 2048, bt=32, seed=103, accum=(mode == 'dx'))
    check(2, mode, 200, 136, 528, bt=32, act='tanh', seed=104)            # ragged rows, columns (multiple of 8) and K quarters (528 = 4 x 132 -> 160-deep quarters)
    check(2, mode, 36, 64, 48, bt=32, seed=105)                           # K shorter than four slices: two waves multiply nothing
    check(2, mode, 64, 40, 32, bt=32, act='sin' if mode == 'fwd' else 'none', seed=106)
    a, _, _ = run_gemm(2, mode, 256, 1024, 1024, bt=32, seed=107)
    b, _, _ = run_gemm(2, mode, 256, 1024, 1024, bt=32, seed=107)
    assert np.array_equal(a, b)
    ref, _, _ = run_gemm(1, mode, 256, 1024, 1024, bt=64, splits=1, seed=107)
    assert rel(a, ref) < 2e-6
    if mode == 'fwd':
        from rlrep_amd import _lib
        eng, tile, sp = C.c_int32(), C.c_int32(), C.c_int32()
        _lib.check(_lib.lib.rlrep_gemm_plan(0, 0, 256, 1024, 1024, 1024, 1024, 1024, C.byref(eng), C.byref(tile), C.byref(sp), None, None), 'plan')
        assert (eng.value, tile.value, sp.value) == (2, 32, 1)
        _lib.check(_lib.lib.rlrep_gemm_plan(0, 0, 2048, 512, 512, 512, 512, 512, C.byref(eng), C.byref(tile), C.byref(sp), None, None), 'plan')
        assert tile.value == 64          # (spedersac's M = 2048 layers: enough 64-wide tiles without a split)


@pytest.mark.parametrize('mode', ['fwd', 'dx', 'dw'])
@pytest.mark.parametrize('splits', [2, 3, 7])
def test_bf16x3_64_tile_finishes_split_k_inside_the_launch_bit_for_bit(mode, splits):
    """The 64-wide bf16x3 tile with split-K: the LAST split workgroup of every output tile sums the slabs in split order and runs the epilogue
    (FLAG_FIN_INLINE: write-through slab stores, a ticket word per tile) == the finishing launch it replaces, bit for bit -- ragged edges, the
    activation epilogues, accumulation and the bias gradient included; run twice over the same workspace (the tickets reset themselves)."""
    for R, Cn, K, kw in ((256, 1024, 1024, dict(act='elu' if mode != 'dw' else 'none')), (200, 132, 708, dict(accum=(mode != 'fwd'))), (256, 2048, 1056, {})):
        for rep in range(2):
            a, want, ea = run_gemm(2, mode, R, Cn, K, bt=64, splits=splits, seed=90 + rep, inline_fin=True, **kw)
            b, _, eb = run_gemm(2, mode, R, Cn, K, bt=64, splits=splits, seed=90 + rep, inline_fin=False, **kw)
            assert np.array_equal(a, b), (mode, splits, R, Cn, K)
            assert rel(a, want) < 1e-5
            if ea is not None:
                assert np.array_equal(ea[0], eb[0]), ('second output', mode, splits, R, Cn, K)


@pytest.mark.parametrize('mode', ['fwd', 'dx', 'dw'])
@pytest.mark.parametrize('bt', [64, 128])
def test_lds_engine_layouts_and_tiles(mode, bt):
    """exact multiples of the tile, one split"""
    check(1, mode, 256, 256, 128, bt=bt, splits=1, seed=1)


@pytest.mark.parametrize('mode', ['fwd', 'dx', 'dw'])
@pytest.mark.parametrize('bt', [64, 128])
def test_lds_engine_ragged_edges(mode, bt):
    """rows / columns / inner length that end inside a tile and inside a 32-deep slice (dW needs R % 4 == 0)"""
    check(1, mode, 148, 92, 100, bt=bt, splits=1, seed=2)
    check(1, mode, 36, 260, 68, bt=bt, splits=1, seed=3)
    if mode != 'dw':
        check(1, mode, 33, 64, 64, bt=bt, splits=1, seed=4)        # odd row count (row-major A only)


@pytest.mark.parametrize('mode', ['fwd', 'dx', 'dw'])
@pytest.mark.parametrize('splits', [2, 3, 7])
def test_lds_engine_split_k(mode, splits):
    """split-K slabs + finishing launch (slabs added in split order), including a last split shorter than the others and K not a multiple of 32"""
    check(1, mode, 128, 192, 708, bt=64, splits=splits, seed=5, accum=(mode != 'fwd'))
    check(1, mode, 200, 128, 1024, bt=128, splits=splits, seed=6)
    a, _, _ = run_gemm(1, mode, 200, 128, 1024, bt=64, splits=splits, seed=6)
    b, _, _ = run_gemm(1, mode, 200, 128, 1024, bt=64, splits=splits, seed=6)
    assert np.array_equal(a, b), 'fixed summation order: reruns are bit-identical'


@pytest.mark.parametrize('act', ['relu', 'elu', 'sin', 'tanh'])
def test_lds_engine_epilogues(act):
    check(1, 'fwd', 192, 128, 96, act=act, bt=64, splits=1, seed=7)
    check(1, 'fwd', 192, 128, 512, act=act, bt=64, splits=2, seed=8)
    check(1, 'dx', 192, 128, 96, act=act, bt=128, splits=1, seed=9, accum=True)
    check(1, 'dx', 192, 128, 512, act=act, bt=64, splits=4, seed=10)
    check(1, 'fwd', 64, 64, 64, act='none', bias=False, bt=64, splits=1, seed=11)


def test_lds_engine_auto_plan_matches_path_shapes():
    """the shapes the step programs actually route here, with the planner's own tile / split choice"""
    check(1, 'fwd', 256, 1024, 1024, act='elu', seed=12)        # ctrlsac phi.l2 (M = 256: 64-wide tiles, split-K 4)
    check(1, 'dx', 256, 1024, 2048, act='elu', seed=13)         # ctrlsac phi.l3 dx
    check(1, 'dw', 1024, 1024, 256, seed=14)                    # ctrlsac phi.l2 dW
    check(1, 'fwd', 2048, 512, 512, act='elu', seed=15)         # spedersac phi layer (both batches)
    check(1, 'fwd', 2048, 2560, 512, seed=16)                   # diffsrsac nabla-mu head (scaled-down width)
    check(1, 'dx', 2048, 512, 2560, act='elu', seed=17)         # ... its dX: small output, long inner dimension
    check(1, 'dw', 2560, 512, 2048, seed=18)


@pytest.mark.parametrize('mode', ['fwd', 'dx', 'dw'])
def test_bf16x3_engine_is_fp32_accurate(mode):
    """engine 2: 128-wide tiles on the bf16 matrix pipe with every operand split exactly into three bf16 pieces and the
    six significant partial products kept -- held to the same 1e-5 as the fp32-MFMA engines (measured ~2e-7), on exact
    tiles, ragged edges, split-K and wide-dynamic-range operands"""
    check(2, mode, 256, 256, 128, splits=1, seed=31)
    check(2, mode, 148, 92, 100, splits=1, seed=32)
    check(2, mode, 200, 128, 1024, splits=3, seed=33, accum=(mode != 'fwd'))
    check(2, mode, 384, 640, 512, act='elu' if mode != 'dw' else 'none', seed=34)
    got, want, _ = run_gemm(2, mode, 256, 128, 256, splits=1, seed=35)
    ref, _, _ = run_gemm(1, mode, 256, 128, 256, bt=128, splits=1, seed=35)
    assert rel(got, want) < 2e-6 and rel(ref, want) < 2e-6, (rel(got, want), rel(ref, want))


@pytest.mark.parametrize('mode', ['fwd', 'dx', 'dw'])
def test_bf16x3_256_by_128_persistent_tile_is_fp32_accurate(mode):
    """engine 2 with bt = 256: gemm_x3w_kernel, the 256 x 128 tile of ONE persistent workgroup per CU (csrc/gemm_x3w.h: the nabla-mu head of
    diffsrsac at Humanoid dims).  One tile, tiles that end inside the 256 rows / 128 columns / the 32-deep block, a one-block K (every tile
    hands its bias sums over at once), more tiles than workgroups (the stream of (tile, block) items crosses tile boundaries; 40 x 33 = 1 320
    tiles of one block each on 256 workgroups), split-K slabs with accumulation, the ELU epilogues, and the bias gradient of the k-major form.
    2e-6 against float64 like the 128-wide tile."""
    check(2, mode, 256, 128, 64, bt=256, splits=1, seed=71)
    check(2, mode, 148, 92, 100, bt=256, splits=1, seed=72)
    check(2, mode, 1032, 644, 196, bt=256, splits=1, seed=73)
    check(2, mode, 300, 132, 1024, bt=256, splits=3, seed=74, accum=(mode != 'fwd'))
    check(2, mode, 768, 640, 512, bt=256, act='elu' if mode != 'dw' else 'none', seed=75)
    check(2, mode, 768, 640, 96, bt=256, splits=1, act='relu' if mode != 'dw' else 'none', seed=76, bias=False)
    check(2, mode, 10240, 4224, 32, bt=256, splits=1, seed=77)
    check(2, mode, 5000, 128, 320, bt=256, splits=2, seed=78)
    got, want, _ = run_gemm(2, mode, 512, 256, 256, bt=256, splits=1, seed=79)
    ref, _, _ = run_gemm(1, mode, 512, 256, 256, bt=128, splits=1, seed=79)
    assert rel(got, want) < 2e-6 and rel(ref, want) < 2e-6, (rel(got, want), rel(ref, want))


def test_bf16x3_256_by_128_tile_refuses_the_epilogues_it_does_not_have():
    from rlrep_amd import _lib
    x = torch.zeros(256 * 256, device='cuda')
    st = torch.cuda.current_stream().cuda_stream
    rc = _lib.lib.rlrep_gemm(2, 0, 0, x.data_ptr(), 256, x.data_ptr(), 256, x.data_ptr(), 128, 256, 128, 256, 0, 3, 0, None, None, 128, x.data_ptr(), 256, 1, None, 0, st)
    assert rc != 0 and b'sin / tanh' in _lib.lib.rlrep_last_error()


@pytest.mark.parametrize('mode', ['fwd', 'dx', 'dw'])
def test_bf16x3_64_wide_tile_is_fp32_accurate(mode):
    """engine 2 with bt = 64: gemm_x3s_kernel, the 64 x 64 tile on the bf16 pipe (row-major operands through [row][80-byte] images, k-major
    ones staged as they lie and read with ds_read_b64_tr_b16) -- the M = 256 / 2048 layers of ctrlsac / spedersac.  Exact tiles, ragged edges
    in every dimension, split-K with a short last split, epilogues, against float64 at the fp32 engines' 1e-5 (measured ~2e-7)."""
    check(2, mode, 256, 256, 128, bt=64, splits=1, seed=41)
    check(2, mode, 148, 92, 100, bt=64, splits=1, seed=42)
    check(2, mode, 36, 260, 68, bt=64, splits=1, seed=43)
    check(2, mode, 200, 128, 1024, bt=64, splits=3, seed=44, accum=(mode != 'fwd'))
    check(2, mode, 128, 192, 708, bt=64, splits=7, seed=45)
    check(2, mode, 256, 1024, 2048, bt=64, act='elu' if mode != 'dw' else 'none', seed=46)        # a ctrlsac layer, the planner's own split
    # 320 workgroups, every split ONE 32-deep slice (the tile prefetches two slices ahead: what it loads past the end must never reach the result --
    # an asm-load form of the prefetch produced random wrong tiles here in round 5), and diffsrsac's dX at HalfCheetah dims whose last split is one
    # slice (the planner's own plan)
    for rep in range(3):
        check(2, mode, 256, 512, 320, bt=64, splits=10, seed=48 + rep)
    check(2, mode, 256, 512, 4352, bt=64, seed=51)
    got, want, _ = run_gemm(2, mode, 256, 128, 256, bt=64, splits=1, seed=47)
    assert rel(got, want) < 2e-6, rel(got, want)


@pytest.mark.parametrize('mode', ['fwd', 'dx', 'dw'])
def test_bf16x3_64_wide_tile_takes_operands_of_any_alignment(mode):
    """x3s_load_*<2>: rows of 119 / 111 / 333 floats and widths that are no multiples of four (spedersac's K = 119 and 111 first layers and their
    [512, 119] weight gradients) staged with 16-byte loads at 4-byte-aligned addresses, the tail of a row fetched as its LAST four elements
    and shifted into place; split-K included.  Same shapes as the fp32 tile's scalar-access test, at the bf16x3 tile's accuracy."""
    check(2, mode, 260, 119, 119, bt=64, splits=1, seed=60, act='elu' if mode != 'dw' else 'none')
    check(2, mode, 128, 119, 333, bt=64, splits=3, seed=61, accum=(mode != 'fwd'))
    check(2, mode, 132, 65, 111, bt=64, splits=1, seed=62)
    check(2, mode, 2048, 512, 119, bt=64, seed=63)                   # spedersac phi.l0 (forward) / its shapes in the other two forms
    check(2, mode, 512, 119, 2048, bt=64, seed=64)                   # ... and the [512, 119] weight gradient's, the planner's own split
    got, want, _ = run_gemm(2, mode, 255, 127, 253, bt=64, splits=1, seed=65)
    assert rel(got, want) < 2e-6, rel(got, want)


def test_bf16x3_engine_wide_dynamic_range():
    """operands spanning 12 decades: the split is exact per element, so the error stays relative to sum |a||b|"""
    from rlrep_amd import _lib
    rs = np.random.RandomState(40)
    R, Cn, K = 128, 128, 256
    A = (rs.standard_normal((R, K)) * 10.0 ** rs.uniform(-6, 6, (R, K))).astype(np.float32)
    B = (rs.standard_normal((Cn, K)) * 10.0 ** rs.uniform(-6, 6, (Cn, K))).astype(np.float32)
    dA, dB, dC = _dev(A), _dev(B), torch.zeros(R, Cn, device='cuda')
    rc = _lib.lib.rlrep_gemm(2, 0, 0, dA.data_ptr(), K, dB.data_ptr(), K, dC.data_ptr(), Cn, R, Cn, K, 0, 0, 0, None, None, Cn, None,
                             0, 1, None, 0, torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, 'gemm')
    got = dC.cpu().numpy().astype(np.float64)
    want = A.astype(np.float64) @ B.astype(np.float64).T
    bound = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64).T
    assert np.all(np.abs(got - want) <= 4e-6 * bound + 1e-30), float(np.max(np.abs(got - want) / bound))


@pytest.mark.parametrize('mode', ['fwd', 'dx', 'dw'])
def test_engines_agree(mode):
    """the two engines compute the same product (different summation order only)"""
    a, want, _ = run_gemm(0, mode, 96, 80, 160, seed=20)
    b, _, _ = run_gemm(1, mode, 96, 80, 160, bt=64, splits=1, seed=20)
    assert rel(a, want) < 1e-5 and rel(b, want) < 1e-5 and rel(a, b) < 1e-5


@pytest.mark.parametrize('mode', ['fwd', 'dx', 'dw'])
def test_lds_engine_unaligned_shapes_use_scalar_accesses(mode):
    """row strides / inner lengths / widths that are not multiples of 4 floats (spedersac's K = 119 and 111 first layers,
    their [512, 119] weight gradients): the affected side falls back to clamped 4-byte accesses, split-K included"""
    check(1, mode, 260, 119, 119, bt=64, splits=1, seed=50, act='elu' if mode != 'dw' else 'none')
    check(1, mode, 128, 119, 333, bt=64, splits=3, seed=51, accum=(mode != 'fwd'))
    check(1, mode, 132, 64, 111, bt=128, splits=1, seed=52)
    check(1, mode, 2048 if mode != 'dw' else 512, 512 if mode != 'dw' else 119, 119 if mode != 'dw' else 2048, seed=53)


def test_bf16x3_engine_rejects_unaligned_shapes():
    from rlrep_amd import _lib
    x = torch.zeros(64 * 64, device='cuda')
    st = torch.cuda.current_stream().cuda_stream
    # inner length not a multiple of 4: fine for engine 1 (scalar staging), refused by the bf16x3 tile
    rc = _lib.lib.rlrep_gemm(2, 0, 0, x.data_ptr(), 30, x.data_ptr(), 30, x.data_ptr(), 64, 16, 64, 30, 0, 0, 0, None, None, 0, None, 0, 0, None, 0, st)
    assert rc < 0 and b'not eligible' in _lib.lib.rlrep_last_error()
