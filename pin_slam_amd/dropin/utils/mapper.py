"""Drop-in for the reference's ``utils.mapper.Mapper`` (utils/mapper.py:33).

``mapping`` (the online-training hot loop, mapper.py:600-844) and ``sdf`` run on the fused HIP
kernels.  When the reference tree is on the ``utils`` package path (drop-in mode, see
pin_slam_amd.dropin.install) this class *inherits* the reference's Mapper, so the data-pool
management around the hot loop (process_frame, pool filtering, ... -- SURVEY 8f "next" rows)
keeps running from the reference's own code, unchanged; stand-alone it provides the minimal
pool surface the hot loop needs."""
from __future__ import annotations

import importlib.util
import math
import os
import sys

import torch

from ... import engine, ops


def _reference_mapper_base():
    """The reference's Mapper class if its source is reachable through ``utils.__path__``."""
    utils_pkg = sys.modules.get("utils")
    for p in list(getattr(utils_pkg, "__path__", []))[1:]:
        f = os.path.join(p, "mapper.py")
        if os.path.exists(f):
            spec = importlib.util.spec_from_file_location("pin_reference_utils_mapper", f)
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            return mod.Mapper
    return None


class _StandaloneBase:
    """Pool surface of the reference Mapper used by the hot loop (mapper.py:34-97, 452-503)."""

    def __init__(self, config, dataset, neural_points, decoders: dict):
        self.config = config
        self.silence = config.silence
        self.dataset = dataset
        self.neural_points = neural_points
        self.sdf_mlp = decoders["sdf"]
        self.sem_mlp = decoders.get("semantic")
        self.color_mlp = decoders.get("color")
        self.device = config.device
        self.dtype = config.dtype
        self.used_poses = None
        self.total_iter = 0
        self.sdf_scale = config.logistic_gaussian_ratio * config.sigma_sigmoid_m
        self.new_idx = None
        self.ba_done_flag = False
        self.adaptive_iter_offset = 0
        self.init_pool()

    def init_pool(self):
        d, f = self.device, self.dtype
        self.coord_pool = torch.empty((0, 3), device=d, dtype=f)
        self.global_coord_pool = torch.empty((0, 3), device=d, dtype=f)
        self.sdf_label_pool = torch.empty((0,), device=d, dtype=f)
        self.color_pool = torch.empty((0, self.config.color_channel), device=d, dtype=f) if self.config.color_on else None
        self.weight_pool = torch.empty((0,), device=d, dtype=f)
        self.time_pool = torch.empty((0,), device=d, dtype=torch.int)
        self.pool_sample_count = 0

    def free_pool(self):
        self.coord_pool = self.weight_pool = self.sdf_label_pool = self.time_pool = None

    def get_batch(self, global_coord=False):
        """Uniform pool sampling (mapper.py:477-503; the 'new sample' half needs process_frame)."""
        index = torch.randint(0, self.pool_sample_count, (self.config.bs,), device=self.device)
        coord = (self.global_coord_pool if global_coord else self.coord_pool)[index, :]
        color = self.color_pool[index] if self.color_pool is not None else None
        return coord, self.sdf_label_pool[index], self.time_pool[index], None, None, color, self.weight_pool[index]


_Base = _reference_mapper_base() or _StandaloneBase


class Mapper(_Base):
    def __init__(self, config, dataset, neural_points, decoders: dict):
        super().__init__(config, dataset, neural_points, decoders)
        self._trainer = None

    # ------------------------------------------------------------------ hot loop
    def _check_supported(self):
        c = self.config
        bad = []
        if getattr(c, "semantic_on", False): bad.append("semantic_on")
        if getattr(c, "color_on", False) and c.color_channel != 3: bad.append("color_channel != 3")
        if c.main_loss_type != "bce": bad.append("main_loss_type != bce")
        if c.proj_correction_on or c.consistency_loss_on: bad.append("proj_correction / consistency loss")
        if c.ekional_loss_on and not c.numerical_grad: bad.append("analytic Eikonal (numerical_grad=False)")
        if c.ekional_loss_on and c.ekional_add_to != "all": bad.append("ekional_add_to != all")
        if not c.opt_adam: bad.append("SGD")
        if c.weight_decay != 0.0: bad.append("weight_decay")
        if self.ba_done_flag: bad.append("mapping after bundle adjustment")
        if bad:
            raise NotImplementedError("Mapper.mapping on libpinhip does not cover: " + ", ".join(bad))

    def _get_trainer(self) -> engine.MapTrainer:
        c, npts = self.config, self.neural_points
        st = npts.search_state()
        fs = npts.field_state(self.sdf_mlp, query_locally=True)
        train_dec = bool(self.sdf_mlp.lout.weight.requires_grad)  # freeze_decoders (tools.py:263-292)
        eik = bool(c.ekional_loss_on and c.weight_e > 0)
        t = self._trainer
        if (t is None or t.fs.feats.numel() != fs.feats.numel() or t.bs != c.bs or t.fs.dec.numel() != fs.dec.numel()
                or (t.buf.n_eik > 0) != eik or t.fs.weighted_first != fs.weighted_first):
            t = engine.MapTrainer(st, fs, None, None, None, None, npts.local_point_ts_update, bs=c.bs,
                                  decimation=c.gradient_decimation, sigma=self.sdf_scale,
                                  weight_e=c.weight_e if eik else 0.0,
                                  eik_eps=c.voxel_size_m * c.num_grad_step_ratio, lr=c.lr, adam_eps=c.adam_eps,
                                  loss_weight_on=c.loss_weight_on, eikonal=eik)
            self._trainer = t
        t.st, t.fs, t.ts_update, t.train_decoder = st, fs, npts.local_point_ts_update, train_dec
        if c.color_on and c.weight_i > 0:  # colour branch (mapper.py:668-671, 802-812)
            fc = npts.field_state(self.color_mlp, query_locally=True, color=True)
            t.set_color(fc, surface_range=c.surface_sample_range_m, weight_i=c.weight_i,
                        train_decoder=bool(self.color_mlp.lout.weight.requires_grad))
        else:
            t.set_color(None)
        b = npts._bricks
        tf = bool(npts.temporal_local_map_on and npts.travel_dist is not None)
        t.bricks = b if (b is not None and b.mode[:2] == (tf, True) and npts.neighbor_K == b.cand_dx.shape[0]) else None
        return t

    def mapping(self, iter_count):
        """PIN map online training given fixed poses (mapper.py:600-844)."""
        self._check_supported()
        iter_count = max(1, iter_count + self.adaptive_iter_offset)
        t = self._get_trainer()
        t.reset_optimizer()  # a new Adam per call (mapper.py:615)
        for it in range(iter_count):
            coord, sdf_label, ts, _, _, color_label, weight = self.get_batch(global_coord=not self.ba_done_flag)
            if t.fc is not None and color_label is None:
                raise RuntimeError("color_on but the data pool holds no colour labels")
            t.step_batch(coord.to(torch.float32).contiguous(), sdf_label.to(torch.float32).contiguous(),
                         weight.to(torch.float32).contiguous(), ts.to(torch.int32).contiguous(), it + 1,
                         color_label=None if t.fc is None else color_label[:, :3].to(torch.float32).contiguous())
            self.total_iter += 1
        self.neural_points.assign_local_to_global()

    def sdf(self, x, get_std=False, min_nn_count=1, accumulate_stability=False):
        """mapper.py:940-956 (forward only)."""
        if accumulate_stability:
            raise NotImplementedError("Mapper.sdf(accumulate_stability=True) is unused by the reference")
        npts = self.neural_points
        q = x.detach().to(torch.float32).contiguous()
        nbr, nn, _ = npts.knn(q, True)
        fs = npts.field_state(self.sdf_mlp, query_locally=True)
        sdf, _, std, _ = ops.sdf_query(fs, q, nbr, nn, grad=False, certainty=False)
        return sdf, (std if (get_std and not self.config.weighted_first) else None), nn >= min_nn_count

    def sdf_batch(self, x, bs, get_std=False, min_nn_count=1, accumulate_stability=False):
        outs = [self.sdf(x[i:i + bs], get_std, min_nn_count, accumulate_stability) for i in range(0, x.shape[0], bs)]
        sdf = torch.cat([o[0] for o in outs])
        std = torch.cat([o[1] for o in outs]) if outs and outs[0][1] is not None else None
        return sdf, std, torch.cat([o[2] for o in outs])
