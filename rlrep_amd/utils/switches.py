"""Diagnostic switches: TWO environment variables, comma-separated tokens (INTEGRATION.md has the table).

RLREP_DISABLE lists default mechanisms to switch off (each has an equivalence test that compares the two forms), RLREP_ENABLE lists opt-in ones,
optionally with a value (`token=value`).  The library parses the same two variables when an agent is created (csrc/engine.hip rl_switches_read);
the Python side reads them where an agent is constructed or a graph is captured -- never per train() call."""
import os


def _parse(name):
    out = {}
    for tok in os.environ.get(name, '').replace(';', ',').replace(' ', ',').split(','):
        if tok:
            k, _, v = tok.partition('=')
            out[k] = v or '1'
    return out


def off(token):
    return token in _parse('RLREP_DISABLE')


def opt(token, default=None):
    return _parse('RLREP_ENABLE').get(token, default)
