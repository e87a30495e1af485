"""Static guard for the kernels' inline-asm loads (tools/check_async_asm.py): on no path from an `asm volatile("global_load...")` to the
s_waitcnt that claims it may the compiler's code touch the destination registers.  The compiler cannot see that such a load is in flight, so
a change of register allocation can turn a correct kernel into a racy one without a single source change near the load (it did: round 2,
rowprog.hip, run-to-run differences of vae_loss).  Runs on the CPU: hipcc -S only."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import check_async_asm  # noqa: E402

CSRC = os.path.join(ROOT, 'rlrep_amd', 'csrc')


def _source_with_local_includes(path, seen=None):
    """The text of a .hip file plus that of the csrc headers it includes (the tile bodies live in headers shared by two kernels)."""
    import re
    seen = set() if seen is None else seen
    if path in seen or not os.path.exists(path):
        return ''
    seen.add(path)
    src = open(path).read()
    for inc in re.findall(r'#include "([\w./]+\.h)"', src):
        src += _source_with_local_includes(os.path.join(CSRC, inc), seen)
    return src


def _files_with_asm_loads():
    out = []
    for f in sorted(os.listdir(CSRC)):
        if not f.endswith('.hip'):
            continue
        src = _source_with_local_includes(os.path.join(CSRC, f))
        if 'asm volatile("global_load' in src or 'asm volatile("ds_read' in src:
            out.append(f)
    return out


def test_guard_sees_the_files_it_is_meant_for():
    # (gemm16_tile.h, shared by gemm16.hip and xchain.hip, no longer issues asm loads: its epilogue operands are plain loads pinned by a
    # memory-clobbering asm since round 3 -- the guard flagged the asm form as soon as the tile body moved into a header)
    assert {'rowprog.hip', 'noisecritic.hip'} <= set(_files_with_asm_loads())
    assert not {'gemm16.hip', 'xchain.hip'} & set(_files_with_asm_loads())


@pytest.mark.skipif(shutil.which('hipcc') is None, reason='hipcc not on PATH')
@pytest.mark.parametrize('name', _files_with_asm_loads())
def test_no_instruction_touches_a_register_with_an_asm_load_in_flight(name, tmp_path):
    out = tmp_path / 'k.s'
    subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-S', '--cuda-device-only', '-I', CSRC, '-o', str(out), os.path.join(CSRC, name)],
                   check=True, stderr=subprocess.DEVNULL)
    bad = check_async_asm.scan(out.read_text())
    assert not bad, bad[:5]


def test_guard_flags_a_reuse_before_the_wait():
    asm = '\n'.join([
        'k:',
        '\t;;#ASMSTART', '\tglobal_load_dwordx4 v[4:7], v[0:1], off', '\t;;#ASMEND',
        '\tglobal_load_dword v9, v[2:3], off',
        '\tv_mov_b32_e32 v5, 0',                 # v5 still has the asm load in flight
        '\ts_waitcnt vmcnt(1)',
        '\tv_add_f32_e32 v8, v4, v5',            # claimed: fine
        '\ts_endpgm', '.Lfunc_end0:'])
    bad = check_async_asm.scan(asm)
    assert [b[0] for b in bad] == [6], bad


def test_guard_follows_the_back_edge_of_a_loop():
    asm = '\n'.join([
        'k:',
        '.LBB0_1:    ; =>This Inner Loop Header',
        '\tv_add_f32_e32 v8, v4, v4',            # second trip: the load of the first trip is in flight
        '\t;;#ASMSTART', '\tglobal_load_dword v4, v[0:1], off', '\t;;#ASMEND',
        '\ts_cbranch_scc1 .LBB0_1',
        '\ts_waitcnt vmcnt(0)',
        '\ts_endpgm', '.Lfunc_end0:'])
    bad = check_async_asm.scan(asm)
    assert [b[0] for b in bad] == [3], bad


def test_guard_covers_lds_reads_claimed_by_lgkmcnt():
    asm = '\n'.join([
        'k:',
        '\t;;#ASMSTART', '\tds_read_b128 v[4:7], v0 offset:0', '\t;;#ASMEND',
        '\t;;#ASMSTART', '\tds_read_b128 v[8:11], v0 offset:64', '\t;;#ASMEND',
        '\t;;#ASMSTART', '\ts_waitcnt lgkmcnt(1)', '\t;;#ASMEND',
        '\tv_add_f32_e32 v12, v4, v5',           # claimed (the older read)
        '\tv_add_f32_e32 v13, v8, v9',           # NOT claimed: one read may still be in flight
        '\ts_waitcnt lgkmcnt(0)',
        '\tv_add_f32_e32 v13, v8, v9',
        '\ts_endpgm', '.Lfunc_end0:'])
    bad = check_async_asm.scan(asm)
    assert [b[0] for b in bad] == [12], bad


def test_scalar_loads_in_flight_make_partial_lgkm_waits_claim_nothing():
    asm = '\n'.join([
        'k:',
        '\t;;#ASMSTART', '\tds_read_b128 v[4:7], v0 offset:0', '\t;;#ASMEND',
        '\ts_load_dwordx2 s[0:1], s[4:5], 0x0',
        '\t;;#ASMSTART', '\tds_read_b128 v[8:11], v0 offset:64', '\t;;#ASMEND',
        '\ts_waitcnt lgkmcnt(1)',                # the scalar load may be the one that returned
        '\tv_add_f32_e32 v12, v4, v5',
        '\ts_waitcnt lgkmcnt(0)',
        '\ts_endpgm', '.Lfunc_end0:'])
    bad = check_async_asm.scan(asm)
    assert [b[0] for b in bad] == [10], bad
