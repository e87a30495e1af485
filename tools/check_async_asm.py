#!/usr/bin/env python3
"""Guard for the inline-asm prefetch loads of rowprog.hip.

rp_gemm's dX inner loop issues its weight loads as `asm volatile("global_load_dwordx4 ...")` and claims them with explicit
`s_waitcnt vmcnt(8)` (the compiler's own wait-count pass drains the prefetch at the loop head otherwise).  The noise-critic
kernels read their LDS fragments the same way (`asm volatile("ds_read_b128 ...")` claimed by `s_waitcnt lgkmcnt(n)`).  The compiler does
not know that such a load is still in flight: it is free to copy, reuse or spill the destination registers before the wait.  This script
compiles the file to gfx950 assembly and runs a forward data flow over each kernel's control-flow graph: on no path from an asm load to the
s_waitcnt that claims it may another instruction name one of its destination registers.  (vmcnt returns in order, so `s_waitcnt vmcnt(n)` claims every load but the n youngest.)

  python tools/check_async_asm.py [file.hip]      exit status 0 = clean
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _regs(t):
    regs = set()
    for m in re.finditer(r'v\[(\d+):(\d+)\]', t):
        regs.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r'\bv(\d+)\b', t):
        regs.add(int(m.group(1)))
    return regs


def _blocks(lines):
    """Basic blocks of one function: list of (label, [(line_no, text, in_asm)], successors-by-label, falls_through)."""
    blocks, cur, inasm = [], {'label': None, 'ins': [], 'succ': [], 'fall': True}, False
    for i, line in lines:
        t = line.strip()
        if 'ASMSTART' in t:
            inasm = True
            continue
        if 'ASMEND' in t:
            inasm = False
            continue
        if not t or t.startswith(';'):
            continue
        t = t.split(';')[0].strip()              # trailing comments ("; =>This Inner Loop Header")
        if t.startswith('.') and not t.endswith(':'):
            continue
        m = re.match(r'^([.\w$]+):$', t)
        if m and not inasm:
            blocks.append(cur)
            cur = {'label': m.group(1), 'ins': [], 'succ': [], 'fall': True}
            continue
        cur['ins'].append((i + 1, t, inasm))
        m = re.match(r'^s_c?branch\w*\s+([.\w$]+)', t)
        if m:
            cur['succ'].append(m.group(1))
            fall = not t.startswith('s_branch')
            blocks.append(dict(cur, fall=fall))
            cur = {'label': None, 'ins': [], 'succ': [], 'fall': True}
        elif t.startswith('s_endpgm') or t.startswith('s_setpc'):
            blocks.append(dict(cur, fall=False))
            cur = {'label': None, 'ins': [], 'succ': [], 'fall': True}
    blocks.append(cur)
    return blocks


_VLOAD = re.compile(r'^(global_load|buffer_load|flat_load|scratch_load|global_atomic\w*_rtn)')
_DS = re.compile(r'^ds_')
_SMEM = re.compile(r'^(s_load|s_buffer_load|s_memtime|s_memrealtime|s_dcache)')


def _dest(t):
    m = re.match(r'\S+\s+v(?:\[(\d+):(\d+)\]|(\d+))', t)
    if not m:
        return set()
    lo, hi = (int(m.group(1)), int(m.group(2))) if m.group(1) else (int(m.group(3)), int(m.group(3)))
    return set(range(lo, hi + 1))


def _transfer(block, state, bad):
    """state = (vm, lgkm): the operations in flight on the two counters, oldest first, each a frozenset of the registers an ASM load will
    still write (empty for everything the compiler tracks itself).
      vmcnt  : vector-memory loads return in order, so `s_waitcnt vmcnt(n)` leaves the n youngest in flight.  Stores share the counter
               but may retire out of order with loads: ignoring them is the conservative reading.
      lgkmcnt: LDS operations (reads AND writes) complete in order among themselves; scalar-memory operations share the counter and
               return out of order, so while one is in flight only lgkmcnt(0) is taken to claim anything.
    """
    vm, lg = list(state[0]), list(state[1])
    smem = state[2]
    for ln, t, inasm in block['ins']:
        if t.startswith('s_waitcnt'):
            m = re.search(r'vmcnt\((\d+)\)', t)
            if m:
                n = int(m.group(1))
                vm = vm[len(vm) - n:] if n else []
            m = re.search(r'lgkmcnt\((\d+)\)', t)
            if m:
                n = int(m.group(1))
                if n == 0:
                    lg, smem = [], 0
                elif not smem:
                    lg = lg[len(lg) - n:]
            continue
        inflight = (set().union(*vm) if vm else set()) | (set().union(*lg) if lg else set())
        if _VLOAD.match(t) or _DS.match(t):
            is_ds = bool(_DS.match(t))
            dest = _dest(t) if (inasm and (not is_ds or 'read' in t)) else set()
            hit = (_regs(t) - dest) & inflight
            if hit and bad is not None:
                bad.add((ln, t, tuple(sorted(hit))))
            (lg if is_ds else vm).append(frozenset(dest))
            continue
        if _SMEM.match(t):
            smem = 1
            continue
        if inflight:
            hit = _regs(t) & inflight
            if hit and bad is not None:
                bad.add((ln, t, tuple(sorted(hit))))
    while vm and not vm[0]:
        vm.pop(0)
    while lg and not lg[0]:
        lg.pop(0)
    return (tuple(vm[-64:]), tuple(lg[-64:]), smem if lg else 0)


def scan(asm_text):
    """Every path through every kernel's control-flow graph, memoised on (block, loads in flight)."""
    lines = list(enumerate(asm_text.split('\n')))
    funcs, cur = [], []
    for i, l in lines:
        cur.append((i, l))
        if l.startswith('.Lfunc_end'):
            funcs.append(cur)
            cur = []
    if cur:
        funcs.append(cur)
    bad = set()
    for fl in funcs:
        blocks = _blocks(fl)
        index = {b['label']: k for k, b in enumerate(blocks) if b['label']}
        succ = []
        for k, b in enumerate(blocks):
            s_ = [index[x] for x in b['succ'] if x in index]
            if b['fall'] and k + 1 < len(blocks):
                s_.append(k + 1)
            succ.append(s_)
        seen, work = set(), [(0, ((), (), 0))]
        while work:
            k, st = work.pop()
            if (k, st) in seen:
                continue
            seen.add((k, st))
            if len(seen) > 2000000:
                raise RuntimeError('state space too large')
            out = _transfer(blocks[k], st, bad)
            for j in succ[k]:
                if (j, out) not in seen:
                    work.append((j, out))
    return sorted(bad)


def main(argv):
    src = argv[1] if len(argv) > 1 else os.path.join(ROOT, 'rlrep_amd', 'csrc', 'rowprog.hip')
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, 'k.s')
        subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-S', '--cuda-device-only', '-I', os.path.dirname(src), '-o', out, src],
                       check=True, stderr=subprocess.DEVNULL)
        bad = scan(open(out).read())
    for ln, t, regs in bad[:20]:
        print(f'line {ln}: {t}   <- registers {regs} have an asm load in flight')
    print(f'{len(bad)} suspicious instruction(s)')
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main(sys.argv))
