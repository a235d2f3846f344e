"""Multi-GPU evaluation: independent tracks shard over ranks, one all-reduce of the 1+p doubles.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI; "gloo" in the CPU
tests).  This module is the framework-level route for Python hosts that already run under torch.distributed; the
engine's own routes need no Python (ssde_desc.n_devices: one process, several GPUs; ssde_comm_init_rank: one
process per GPU with the all-reduce inside ssde_eval -- what bench.py times).  Tracks (ID segments) are independent -- the filter state is re-initialised at every ID
change and the log-likelihood is a plain sum (/root/reference/src/nllk/nllk_ctcrw.hpp:196-200,
234) -- so every rank owns a contiguous block of whole tracks for good, evaluates its partial
[nllk, gradient] on its own GPU, and a single sum all-reduce of that tiny vector (latency-bound,
a few hundred bytes) yields the batch value on every rank.  The parameter-only smoothing penalty
is added once, after the reduction.  There is no other data-path collective.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import numpy as np


def shard_rows(ID, world_size: int, rank: int) -> Tuple[int, int]:
    """Row range [lo, hi) of this rank: contiguous whole tracks, balanced by row count."""
    ID = np.asarray(ID)
    n = len(ID)
    first = np.ones(n, dtype=bool)
    first[1:] = ID[1:] != ID[:-1]
    starts = np.append(np.flatnonzero(first), n)
    # boundary k is the first segment start at or beyond k * n / world_size
    cuts = [int(starts[np.searchsorted(starts, (k * n) // world_size, side="left")]) for k in range(world_size)] + [n]
    return cuts[rank], cuts[rank + 1]


class ShardedObjective:
    """fn/gr over all ranks' tracks.

    local_eval(par) -> torch tensor [nllk_data, grad (n_par_full), window_check] on the rank's device
                       (Engine.eval_device on a GPU; the tests inject an oracle-backed callable)
    penalty(par)    -> (value, grad) of the smoothing penalty (parameter-only, identical on every rank)
    """

    def __init__(self, local_eval: Callable, n_par_full: int, penalty: Optional[Callable] = None, group=None,
                 on_window_failure: Optional[Callable] = None, window_tol: float = 1e-11,
                 on_window_calm: Optional[Callable] = None, cooldown: int = 32, relax_per_widen: int = 2):
        """on_window_failure(factor=4): widen the warm-up (Engine.widen_windows; factor 0 = one sequential window)
        before the re-evaluation -- without it a failed check raises;
        on_window_calm: narrow it again (Engine.relax_windows) after `cooldown` evaluations accepted at the first
        try -- ssde_eval's own policy, taken identically on every rank because every rank sees the reduced check."""
        self.local_eval, self.n_par_full, self.penalty = local_eval, n_par_full, penalty
        self.group, self.on_window_failure, self.window_tol = group, on_window_failure, window_tol
        self.on_window_calm, self.cooldown = on_window_calm, cooldown
        self.relax_per_widen = relax_per_widen      # widen_windows(4) is undone by two relax_windows() (halvings)
        self._calm, self._widened = 0, 0

    def eval(self, par):
        import torch.distributed as dist
        par = np.ascontiguousarray(par, dtype=np.float64)
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        max_attempts = 6
        for attempt in range(max_attempts + 1):
            out = self.local_eval(par)
            if world > 1:
                # [nllk, gradient] are summed; the hand-over check is a maximum (a sum compared with tol * world would
                # let one rank pass at almost twice the tolerance): its own tiny reduction
                chk = out[-1:].clone()
                dist.all_reduce(out, op=dist.ReduceOp.SUM, group=self.group)
                dist.all_reduce(chk, op=dist.ReduceOp.MAX, group=self.group)
                out[-1] = chk[0]
            res = out.detach().cpu().numpy()
            # every rank sees the same reduced values, so every rank takes the same branch
            if res[-1] <= self.window_tol:
                break
            # a non-finite nllk is rejected by the caller whatever the windows did (ssde_eval's rule): no widening
            # because an optimiser probed an absurd point once
            if not np.isfinite(res[0]):
                break
            if self.on_window_failure is None:
                raise RuntimeError(f"window hand-over check failed ({res[-1]:.3g} > {self.window_tol:g}) and no "
                                   "on_window_failure handler was given: the result is invalid (include/ssde.h)")
            if attempt == max_attempts:
                raise RuntimeError(f"window hand-over check still failing ({res[-1]:.3g}) on one sequential window")
            if attempt == max_attempts - 1:
                self.on_window_failure(0)          # last resort: one sequential window per track (widen_windows(0))
            else:
                self.on_window_failure()
                self._widened += self.relax_per_widen
        if attempt == 0:
            self._calm += 1
            if self._widened > 0 and self.on_window_calm is not None and self._calm >= self.cooldown:
                self.on_window_calm()
                self._calm, self._widened = 0, self._widened - 1
        elif np.isfinite(res[0]):
            self._calm = 0
            self.cooldown = min(2 * self.cooldown, 1 << 14)
        value, grad = float(res[0]), res[1:1 + self.n_par_full].copy()
        if self.penalty is not None:
            pv, pg = self.penalty(par)
            value += pv
            grad += pg
        return value, grad
