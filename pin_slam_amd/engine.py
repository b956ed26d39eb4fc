"""Host-side drivers of the two hot loops, on top of the C ABI (pin_slam_amd.ops):

* :class:`GNTracker`  -- Tracker.tracking / registration_step (utils/tracker.py:43-225, 367-611):
  per Gauss-Newton iteration one kNN launch (pose applied in-kernel), one fused
  SDF+Jacobian+normal-equation launch, one 16 KiB read-back and a 6x6 float64 solve.
* :class:`MapTrainer` -- Mapper.mapping (utils/mapper.py:600-844): per iteration batch gather,
  query generation, kNN, fused forward/loss/backward, (optional RCCL all-reduce), Adam.

The drop-in classes in ``pin_slam_amd.dropin`` wrap these with the reference's signatures.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from . import ops
from ._lib import PIN_GN_NSUMS, PIN_GN_REPLICAS, GnParams


class GNTracker:
    def __init__(self, st: ops.SearchState, fs: ops.FieldState, gp: GnParams, lm_lambda: float, n_max: int):
        self.st, self.fs, self.gp, self.lm_lambda = st, fs, gp, lm_lambda
        dev = fs.feats.device
        self.nbr = torch.empty((n_max, fs.k, 4), dtype=torch.float32, device=dev)
        self.nn = torch.empty((n_max,), dtype=torch.int32, device=dev)
        self.cur = torch.empty((n_max, 3), dtype=torch.float32, device=dev)
        self.sums = torch.empty((PIN_GN_REPLICAS, PIN_GN_NSUMS), dtype=torch.float64, device=dev)
        self.sums_host = torch.empty((PIN_GN_REPLICAS, PIN_GN_NSUMS), dtype=torch.float64).pin_memory()
        self.on_knn = None  # optional hook(start: bool) used by bench.py to bracket the kNN launch

    def step(self, src: torch.Tensor, T: Optional[np.ndarray], time_filtering=True, local=True, labels=None):
        n = src.shape[0]
        out = (self.nbr[:n], self.nn[:n], self.cur[:n])
        if self.on_knn:
            self.on_knn(True)
        ops.knn_query(self.st, src, self.fs.k, time_filtering=time_filtering, local=local, pose=T, out=out)
        if self.on_knn:
            self.on_knn(False)
        cur = out[2] if T is not None else src
        ops.gn_accumulate(self.fs, self.gp, cur, out[0], out[1], sdf_labels=labels, sums=self.sums)
        self.sums_host.copy_(self.sums, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        return ops.solve_gn(self.sums_host.numpy(), self.lm_lambda)

    def track(self, src: torch.Tensor, T_init: np.ndarray, iters: int, term_deg: float = 0.01,
              term_m: float = 0.001, early_exit: bool = True):
        """GN loop with the reference's termination rule (tracker.py:174-184).  Returns
        (T, valid_count, residual_cm, iterations)."""
        T = np.array(T_init, dtype=np.float64)
        converged = False
        cnt, res, it = 0, 0.0, 0
        for it in range(iters):
            dT, cnt, res, _ = self.step(src, T)
            T = dT @ T
            if converged:
                break
            if early_exit:
                ang = np.degrees(np.arccos(np.clip((np.trace(dT[:3, :3]) - 1) / 2, -1, 1)))
                if (abs(ang) < term_deg and np.linalg.norm(dT[:3, 3]) < term_m) or it == iters - 2:
                    converged = True
        return T, cnt, res, it + 1


class MapTrainer:
    def __init__(self, st: ops.SearchState, fs: ops.FieldState, pool_coord, pool_label, pool_weight, pool_ts,
                 ts_update, *, bs: int, decimation: int, sigma: float, weight_e: float, eik_eps: float,
                 lr: float = 0.01, adam_eps: float = 1e-15, loss_weight_on: bool = False, train_decoder: bool = True,
                 eikonal: bool = True, rank: int = 0, world: int = 1):
        self.st, self.fs = st, fs
        self.pool = (pool_coord, pool_label, pool_weight, pool_ts)
        self.ts_update = ts_update
        self.bs, self.dec = int(bs), int(decimation)
        self.sigma, self.weight_e, self.eik_eps = sigma, weight_e, eik_eps
        self.lr, self.adam_eps, self.loss_weight_on = lr, adam_eps, loss_weight_on
        self.train_decoder = train_decoder
        self.rank, self.world = rank, world
        assert self.bs % world == 0, "global batch must divide over the ranks"
        self.bs_local = self.bs // world
        dev = fs.feats.device
        nf, nd = fs.feats.numel(), fs.dec.numel()
        # one flat gradient buffer [decoder | features]: a single all-reduce payload
        self.grad = torch.zeros((nd + nf,), dtype=torch.float32, device=dev)
        self.gdec, self.gfeat = self.grad[:nd], self.grad[nd:]
        self.m = torch.zeros_like(self.grad)
        self.v = torch.zeros_like(self.grad)
        from .sharding import n_eik_global, shard_range
        start, _ = shard_range(self.bs, rank, world)
        self.buf = ops.TrainBuffers(self.bs_local, self.dec, fs.k, fs.hidden, fs.levels, eikonal=eikonal,
                                    shard_start=start)
        self.coord = torch.empty((self.bs_local, 3), dtype=torch.float32, device=dev)
        self.label = torch.empty((self.bs_local,), dtype=torch.float32, device=dev)
        self.weight = torch.empty((self.bs_local,), dtype=torch.float32, device=dev)
        self.ts = torch.empty((self.bs_local,), dtype=torch.int32, device=dev)
        # Eikonal samples are the global batch's coord[::dec]; each rank owns those in its shard
        self.n_eik_global = n_eik_global(self.bs, self.dec) if eikonal else 0
        self.total_iter = 0

    def iteration(self, index_local: torch.Tensor, step: int):
        ops.gather_batch(*self.pool, index_local, (self.coord, self.label, self.weight, self.ts))
        self.step_batch(self.coord, self.label, self.weight, self.ts, step)

    def step_batch(self, coord, label, weight, ts, step: int):
        """One iteration on an explicit (already gathered) batch shard."""
        ops.train_step(self.st, self.fs, self.buf, coord, label, weight, ts,
                       self.fs.certainty, self.ts_update, self.gfeat, self.gdec if self.train_decoder else None,
                       sigma=self.sigma, weight_e=self.weight_e, eik_eps=self.eik_eps,
                       loss_weight_on=self.loss_weight_on, global_n_main=self.bs, global_n_eik=self.n_eik_global)
        if self.world > 1:
            import torch.distributed as dist
            dist.all_reduce(self.grad if self.train_decoder else self.gfeat)
        nd = self.gdec.numel()
        ops.adam_step(self.fs.feats, self.gfeat, self.m[nd:], self.v[nd:], step, self.lr, eps=self.adam_eps)
        if self.train_decoder:
            ops.adam_step(self.fs.dec, self.gdec, self.m[:nd], self.v[:nd], step, self.lr, eps=self.adam_eps)
        self.total_iter += 1

    def reset_optimizer(self):
        """setup_optimizer is called anew by every Mapper.mapping (mapper.py:615)."""
        self.m.zero_()
        self.v.zero_()
        self.grad.zero_()

    def mapping(self, index_batches):
        """One Mapper.mapping call: a fresh Adam state (mapper.py:615) and len(index_batches)
        iterations.  index_batches[i] is this rank's int32 shard of the i-th global batch."""
        self.m.zero_()
        self.v.zero_()
        if self.world > 1:
            cert0 = self.fs.certainty.clone()
        for i, idx in enumerate(index_batches):
            self.iteration(idx, i + 1)
        if self.world > 1:  # certainty / ts side effects of the other ranks' shards
            import torch.distributed as dist
            delta = self.fs.certainty - cert0
            dist.all_reduce(delta)
            self.fs.certainty.copy_(cert0 + delta)
            dist.all_reduce(self.ts_update, op=dist.ReduceOp.MAX)
