"""RLREP_DISABLE / RLREP_ENABLE: the two token lists every diagnostic switch lives in, parsed by the package
(rlrep_amd/utils/switches.py) and by the library at its entry points (csrc/engine.hip rl_switches_read) -- never on a launch path.
And the bank arithmetic the bf16x3 tiles' LDS images were chosen by (tools/lds_banks.py; MI355X_MICROARCH.md lane groups)."""
import ctypes as C
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), os.path.join(os.path.dirname(HERE), 'tools')]


def test_python_side_tokens(monkeypatch):
    from rlrep_amd.utils import switches as sw
    monkeypatch.delenv('RLREP_DISABLE', raising=False)
    monkeypatch.delenv('RLREP_ENABLE', raising=False)
    assert not sw.off('prologue') and sw.opt('defer_sets') is None and sw.opt('defer_sets', '3') == '3'
    monkeypatch.setenv('RLREP_DISABLE', 'prologue, info_history;graph_dp')
    monkeypatch.setenv('RLREP_ENABLE', 'stamp,defer_sets=2 , dp_fused_mb=8')
    assert sw.off('prologue') and sw.off('info_history') and sw.off('graph_dp') and not sw.off('chain_next')
    assert sw.opt('stamp') == '1' and sw.opt('defer_sets') == '2' and sw.opt('dp_fused_mb') == '8' and sw.opt('managed_images') is None


def test_library_reads_the_same_tokens_at_its_entry_points(monkeypatch):
    """rlrep_nc_fwd_plan is a host-only entry point that re-reads the switches: `x3` among other tokens turns the bf16x3 engine off."""
    from rlrep_amd import _lib
    out = [C.c_int32() for _ in range(3)]

    def engine():
        assert _lib.lib.rlrep_nc_fwd_plan(4, 256, 256, 256, *[C.byref(o) for o in out]) == 0
        return out[0].value
    monkeypatch.delenv('RLREP_DISABLE', raising=False)
    assert engine() == 1
    monkeypatch.setenv('RLREP_DISABLE', 'chain_next, x3 ,fold_mse')
    assert engine() == 0
    monkeypatch.setenv('RLREP_DISABLE', 'chain_next,x3s')          # (a token that merely starts with x3 is another switch)
    assert engine() == 1


def test_lds_images_of_the_bf16x3_tiles_are_conflict_free_by_the_guides_rules():
    import lds_banks as lb
    old = lambda row, kb: row * 80 + kb
    new = lambda row, kb: row * 64 + (((kb >> 4) ^ ((row >> 2) & 3)) << 4) + (kb & 15)          # = gemm_lds.hip x3r_off
    for off, want_write in ((old, 4), (new, 0)):
        w = [lb.extra_cycles('ds_write_b64', [off((64 * wv + l) >> 3, 8 * (l & 7)) for l in range(64)]) for wv in range(8)]
        assert all(x == want_write for x in w), (w, want_write)
        for c in (0, 1):
            assert lb.extra_cycles('ds_read_b128', [off(l & 31, 32 * c + 16 * (l >> 5)) for l in range(64)]) == 0
    # the 16x16x32 fragment read of the noise-critic kernels on 80-byte rows: 4 extra cycles (docs/history/r05.md)
    assert lb.extra_cycles('ds_read_b128', [(l & 15) * 80 + (l >> 4) * 16 for l in range(64)]) == 4
