"""Drop-in for the reference's ``utils.tracker.Tracker`` (utils/tracker.py:21): same constructor
and ``tracking`` / ``registration_step`` / ``query_source_points`` signatures, executed by the
fused HIP kernels.  ``tracking`` runs the whole Gauss-Newton loop on the device (per iteration: kNN with the pose
read from the device state, SDF + analytic Jacobian + normal-equation sums, a one-wave 6x6 solve with the
reference's validity / convergence rules) and reads back 512 bytes once per call; ``registration_step`` is the
host-driven single step (one read-back of the sums, float64 solve on the host)."""
from __future__ import annotations

import math

import numpy as np
import torch

from ... import engine, hostcache, ops
from ..._lib import GnParams


class Tracker:
    def __init__(self, config, neural_points, decoders: dict):
        self.config = config
        self.silence = config.silence
        self.neural_points = neural_points
        self.sdf_mlp = decoders["sdf"]
        self.sem_mlp = decoders.get("semantic")
        self.color_mlp = decoders.get("color")
        self.device = config.device
        self.dtype = config.dtype
        self.reg_local_map = True  # False in localisation mode (pin_slam.py:168)
        self.sdf_scale = config.logistic_gaussian_ratio * config.sigma_sigmoid_m
        self._gn = None

    # ------------------------------------------------------------------ helpers
    def _gn_params(self, min_grad_norm, max_grad_norm, GM_dist, GM_grad) -> GnParams:
        c = self.config
        gp = GnParams()
        gp.valid_nn_k = int(c.track_mask_query_nn_k)
        gp.min_grad_norm, gp.max_grad_norm = float(min_grad_norm), float(max_grad_norm)
        gp.max_sdf_std = float(c.surface_sample_range_m * c.max_sdf_std_ratio)
        gp.gm_dist = float(GM_dist) if GM_dist else 0.0
        gp.gm_grad = float(GM_grad) if GM_grad else 0.0
        gp.dist_div_grad_norm = int(bool(getattr(c, "reg_dist_div_grad_norm", False)))  # tracker.py:452-456
        return gp

    def _engine(self, n, gp, lm_lambda) -> engine.GNTracker:
        npts = self.neural_points
        st = npts.search_state()
        fs = npts.field_state(self.sdf_mlp, query_locally=self.reg_local_map)
        if self._gn is None or self._gn.nbr.shape[0] < n or self._gn.nbr.shape[1] != fs.k:
            self._gn = engine.GNTracker(st, fs, gp, lm_lambda, n)
        g = self._gn
        g.st, g.fs, g.gp, g.lm_lambda = st, fs, gp, lm_lambda
        tf = bool(npts.temporal_local_map_on and self.reg_local_map and npts.travel_dist is not None)
        b = npts._use_bricks() if self.reg_local_map else None
        g.bricks = b if (b is not None and b.mode[:2] == (tf, True) and npts.neighbor_K == b.cand_dx.shape[0]) else None
        g.local = self.reg_local_map
        return g

    def _unsupported(self, colors, normals):
        if normals is not None:
            raise NotImplementedError("normal-consistency weight (tracker.py:482-488) is not built")
        if colors is not None and self.config.color_on and self.config.color_channel != 3:
            raise NotImplementedError("libpinhip colour decoders have 3 heads (color_channel = 3)")

    def _color_field(self, query_locally):
        return self.neural_points.field_state(self.color_mlp, query_locally=query_locally, color=True)

    def _color_term(self, colors):
        """Colour term of registration_step (tracker.py:385-386, 492-518): photometric residual
        rows, or exp(-|dI|) consistency weights, on the intensity of the RGB prediction."""
        c = self.config
        if colors is None or not c.color_on:
            return None, None
        photo = bool(c.photometric_loss_on)
        if not photo and not c.consist_wieght_on:
            return None, None
        col = colors.detach()[:, :3].to(torch.float32).contiguous()
        return ops.color_term(self._color_field(self.reg_local_map), col, photometric=photo,
                              photo_weight=c.photometric_loss_weight, consist_weight=bool(c.consist_wieght_on))

    # ------------------------------------------------------------------ API
    def tracking(self, source_points, init_pose=None, source_colors=None, source_normals=None,
                 source_semantics=None, source_sdf=None, cur_ts=None, loop_reg: bool = False,
                 vis_result: bool = False):
        c = self.config
        self._unsupported(source_colors, source_normals)
        T0 = torch.eye(4, dtype=torch.float64, device=self.device) if init_pose is None else init_pose
        T = T0.detach().cpu().numpy().astype(np.float64)
        gp = self._gn_params(c.reg_min_grad_norm, c.reg_max_grad_norm,
                             c.reg_GM_dist_m if c.reg_GM_dist_m > 0 else None,
                             c.reg_GM_grad if c.reg_GM_grad > 0 else None)
        src = source_points.detach().to(torch.float32).contiguous()
        n = src.shape[0]
        gn = self._engine(n, gp, c.reg_lm_lambda)
        labels = None if source_sdf is None else source_sdf.detach().to(torch.float32).contiguous()
        iter_n = c.reg_iter_n
        max_valid_final = c.surface_sample_range_m * c.final_residual_ratio_thre * 100.0
        tf = bool(self.neural_points.temporal_local_map_on and self.reg_local_map
                  and self.neural_points.travel_dist is not None)
        # the whole GN loop runs on the device (tracker.py:114-184); one read-back
        ct, _keep = self._color_term(source_colors)
        T, cnt, res_cm, iters, valid_flag, extra = gn.track(
            src, T, iter_n, term_deg=c.reg_term_thre_deg, term_m=c.reg_term_thre_m, early_exit=True,
            min_valid_ratio=0.15 if loop_reg else 0.2, time_filtering=tf, local=self.reg_local_map, labels=labels,
            color=ct)
        i = iters - 1
        converged = extra["converged"]
        if res_cm > max_valid_final:
            valid_flag = False
        cov_mat = None
        if vis_result and converged and cnt >= 10:  # only the last iteration computes these
            N_raw = extra["N_raw"]
            eig = np.linalg.eigvals(N_raw[3:, 3:]).real
            if c.eigenvalue_check and eig.min() < cnt * c.eigenvalue_ratio_thre:
                valid_flag = False
            cov_mat = np.linalg.inv(N_raw) * extra["mse"]
        T_out = torch.tensor(T, dtype=torch.float64, device=self.device)
        hostcache.remember(T_out, T)  # (Mapper.process_frame reads this pose on the host again a moment later)
        if not valid_flag and i < 10:
            T_out, cov_mat = init_pose, None
        return T_out, cov_mat, None, valid_flag

    def query_source_points(self, coord, bs, query_sdf=True, query_sdf_grad=True, query_color=False,
                            query_color_grad=False, query_sem=False, query_mask=True, query_certainty=True,
                            query_locally=True, mask_min_nn_count: int = 4):
        if query_sem and self.sem_mlp is None:
            raise RuntimeError("query_sem without a semantic decoder (config.semantic_on)")
        if (query_color or query_color_grad) and self.config.color_channel != 3:
            raise NotImplementedError("libpinhip colour decoders have 3 heads (color_channel = 3)")
        npts = self.neural_points
        q = coord.detach().to(torch.float32).contiguous()
        nbr, nn, _ = npts.knn(q, query_locally)
        sdf = grad = std = cert = None
        if query_sdf or query_sdf_grad or query_certainty:
            fs = npts.field_state(self.sdf_mlp, query_locally=query_locally)
            fs.stage_decoder()
            sdf, grad, std, cert = ops.sdf_query(fs, q, nbr, nn, grad=query_sdf_grad)
        color = color_grad = None
        if query_color:
            fc = self._color_field(query_locally)
            if query_color_grad:  # d colour_c / d x for each channel (tracker.py:346-349)
                color_grad = torch.empty((q.shape[0], 3, 3), dtype=torch.float32, device=q.device)
                for ch in range(3):
                    color, _, g = ops.color_query(fc, q, nbr, nn, kappa=[float(ch == j) for j in range(3)])
                    color_grad[:, ch, :] = g
            else:
                color, _, _ = ops.color_query(fc, q, nbr, nn, want_grad=False)
        sem = None
        if query_sem:  # argmax of the (weighted) log-probabilities (tracker.py:336-341); a float tensor as the reference's buffer
            fsem = npts.field_state(self.sem_mlp, query_locally=query_locally)
            lab, _ = ops.sem_query(fsem, q, nbr, nn, int(self.sem_mlp.out_dim))
            sem = lab.to(torch.float32)
        mask = (nn >= mask_min_nn_count) if query_mask else None
        return sdf, grad, color, color_grad, sem, mask, (cert if query_certainty else None), std

    def registration_step(self, points, normals, sdf_labels, colors, min_grad_norm, max_grad_norm, GM_dist=None,
                          GM_grad=None, lm_lambda=0.0, vis_weight_pc=False):
        self._unsupported(colors, normals)
        gp = self._gn_params(min_grad_norm, max_grad_norm, GM_dist, GM_grad)
        pts = points.detach().to(torch.float32).contiguous()
        gn = self._engine(pts.shape[0], gp, lm_lambda)
        tf = bool(self.neural_points.temporal_local_map_on and self.reg_local_map
                  and self.neural_points.travel_dist is not None)
        labels = None if sdf_labels is None else sdf_labels.detach().to(torch.float32).contiguous()
        ct, _keep = self._color_term(colors)
        dT, cnt, res_cm, extra = gn.step(pts, None, time_filtering=tf, local=self.reg_local_map, labels=labels, color=ct)
        T = torch.tensor(dT, dtype=torch.float64, device=self.device)
        if cnt < 10:
            return T, None, None, None, pts[:0], 0.0, 0.0
        photo_res = None
        if ct is not None and ct.mode == 2 and extra is not None:  # mean |I_pred - I_meas| (tracker.py:523-525)
            photo_res = extra.get("photo_residual")
        cov = eig = None
        if vis_weight_pc and extra is not None:
            eig = torch.tensor(np.linalg.eigvals(extra["N_raw"][3:, 3:]).real)
            cov = torch.tensor(np.linalg.inv(extra["N_raw"]) * extra["mse"])
        valid_points = pts[:cnt]  # callers only use the COUNT of valid points (tracker.py:161)
        return T, cov, eig, None, valid_points, res_cm, photo_res
