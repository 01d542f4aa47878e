"""Stage timer with the reference's scope names (core/utils/timer.py: `with timer.env(name)`,
`print_stats`, `reset`, `total_time`), measuring DEVICE time with HIP events on the current
stream instead of the reference's un-synchronised wall clock.

Disabled by default (a scope then costs one attribute test); `timer.enable_all()` turns it on.
Scopes on the path, as in snake_head.py:1786-1864: contour_sampling, snake_feat_prep,
snake_deformation, snake_postprocess.  Events are resolved lazily in `total_times()` /
`print_stats()`, so an enabled timer adds two event records per scope and no synchronisation
until the statistics are read.
"""
from collections import defaultdict

import torch

_enabled = False
_pending = []                      # (name, start_event, stop_event)
_totals = defaultdict(float)       # ms
_counts = defaultdict(int)


def enable_all():
    global _enabled
    _enabled = True


def disable_all():
    global _enabled
    _enabled = False


def reset():
    _pending.clear()
    _totals.clear()
    _counts.clear()


class env:
    """`with timer.env("snake_deformation"): ...` (reference: timer.py:122-138)."""

    def __init__(self, fn_name, use_stack=True):
        self.name = fn_name
        self.ev = None

    def __enter__(self):
        if _enabled and torch.cuda.is_available():
            self.ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            self.ev[0].record()
        return self

    def __exit__(self, e, ev, t):
        if self.ev is not None:
            self.ev[1].record()
            _pending.append((self.name, self.ev[0], self.ev[1]))
            self.ev = None


def _resolve():
    if _pending:
        torch.cuda.synchronize()
        for name, a, b in _pending:
            _totals[name] += a.elapsed_time(b)
            _counts[name] += 1
        _pending.clear()


def total_times():
    """{scope: (total device ms, calls)}"""
    _resolve()
    return {k: (_totals[k], _counts[k]) for k in _totals}


def total_time():
    _resolve()
    return sum(_totals.values())


def print_stats(divider=None):
    """Per-scope device time; `divider` = number of images, as the reference's ms/img table
    (timer.py:92-114)."""
    stats = total_times()
    if not stats:
        print("dance_amd.timer: nothing recorded (timer.enable_all() first)")
        return
    width = max(len(k) for k in stats)
    print("%-*s | %10s | %6s%s" % (width, "scope", "total ms", "calls", " | ms / img" if divider else ""))
    for k, (ms, n) in sorted(stats.items(), key=lambda kv: -kv[1][0]):
        extra = " | %8.3f" % (ms / divider) if divider else ""
        print("%-*s | %10.3f | %6d%s" % (width, k, ms, n, extra))
