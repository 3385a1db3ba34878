"""Exchange bookkeeping for the multi-GPU path: every ``torch.distributed`` collective of the hot path (RCCL over xGMI on
MI355X -- backend string "nccl"; gloo in the CPU tests) is issued through ``exchange(kind)``, which, when ``EXCHANGE_LOG``
is a dict, brackets the call with events on the current stream (the collective is ordered behind them) and sums seconds and
bytes per kind.  ``bench.py`` reports the totals (``exchanges``) so the factor all-reduce, eigenvector broadcasts,
query-gradient all-gather and score gather of SURVEY.md section 8(e) can be read off a scaling run."""

from __future__ import annotations

import contextlib
import time
from typing import Dict, Optional

import torch

EXCHANGE_LOG: Optional[Dict[str, dict]] = None


@contextlib.contextmanager
def exchange(kind: str, nbytes: int = 0):
    log = EXCHANGE_LOG
    if log is None:
        yield
        return
    entry = log.setdefault(kind, {"calls": 0, "bytes": 0, "events": [], "host_seconds": 0.0})
    entry["calls"] += 1
    entry["bytes"] += int(nbytes)
    if torch.cuda.is_available():
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        yield
        end.record()
        entry["events"].append((start, end))
    else:
        t0 = time.perf_counter()
        yield
        entry["host_seconds"] += time.perf_counter() - t0


def summary(log: Optional[Dict[str, dict]]) -> Dict[str, dict]:
    """``{kind: {calls, bytes, seconds}}``; call after a device synchronisation."""
    out = {}
    for kind, entry in (log or {}).items():
        seconds = entry["host_seconds"] + sum(s.elapsed_time(e) for s, e in entry["events"]) * 1e-3
        out[kind] = {"calls": entry["calls"], "bytes": entry["bytes"], "seconds": seconds}
    return out
