"""Drop-in for the reference's ``utils.mesher.Mesher`` (utils/mesher.py:21): ``query_points`` -- the
bulk forward-only SDF / colour / marching-cubes-mask query over dense grids (mesher.py:40-164) -- runs on
the fused search + decode kernels instead of query_feature -> masked Decoder.sdf.  In drop-in mode this
class inherits the reference's Mesher, so bounding-box handling, marching cubes (skimage) and the mesh
output keep running from the reference's own code."""
from __future__ import annotations

import importlib.util
import math
import os
import sys

import numpy as np
import torch

from ... import ops


def _reference_base():
    utils_pkg = sys.modules.get("utils")
    for p in list(getattr(utils_pkg, "__path__", []))[1:]:
        f = os.path.join(p, "mesher.py")
        if os.path.exists(f):
            spec = importlib.util.spec_from_file_location("pin_reference_utils_mesher", f)
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            return mod.Mesher
    return None


class _StandaloneBase:
    def __init__(self, config, neural_points, decoders: dict):
        self.config = config
        self.silence = config.silence
        self.neural_points = neural_points
        self.sdf_mlp = decoders["sdf"]
        self.sem_mlp = decoders.get("semantic")
        self.color_mlp = decoders.get("color")
        self.device = config.device
        self.cur_device = self.device
        self.dtype = config.dtype
        self.global_transform = np.eye(4)


_Base = _reference_base() or _StandaloneBase


class Mesher(_Base):
    global_bricks_min_queries = int(os.environ.get("PIN_MESHER_BRICKS_MIN", "3000000"))  # (0: always; a huge number: never)

    def query_points(self, coord, bs, query_sdf=True, query_sem=False, query_color=False, query_mask=True,
                     query_locally=False, mask_min_nn_count: int = 4, out_torch: bool = False):
        if query_sem and self.sem_mlp is None:
            raise RuntimeError("query_sem without a semantic decoder (config.semantic_on)")
        if query_color and self.config.color_channel != 3:
            raise NotImplementedError("libpinhip colour decoders have 3 heads (color_channel = 3)")
        npts = self.neural_points
        n = coord.shape[0]
        dev = torch.device(self.device)
        sdf_d = torch.zeros(n, dtype=torch.float32, device=dev) if query_sdf else None
        col_d = torch.zeros((n, 3), dtype=torch.float32, device=dev) if query_color else None
        mask_d = torch.zeros(n, dtype=torch.bool, device=dev) if query_mask else None
        sem_d = torch.zeros(n, dtype=torch.float32, device=dev) if query_sem else None
        fsem = npts.field_state(self.sem_mlp, query_locally=query_locally) if query_sem else None
        fs = npts.field_state(self.sdf_mlp, query_locally=query_locally) if query_sdf else None
        fc = npts.field_state(self.color_mlp, query_locally=query_locally, color=True) if query_color else None
        for fld in (fs, fc):  # the decoders do not change during the call: one staged image for all its launches
            if fld is not None:
                fld.stage_decoder()
        # Millions of queries against the GLOBAL map (a reconstruction grid): one brick cache over the whole map for the call
        # (ops.BrickCache, ~0.4 ms per 2 M points) instead of the direct probe of the hash table per query -- the same records
        # (tests/test_gpu_bricks.py), a third less search time per query; small calls and local queries go through
        # NeuralPoints.knn as before
        gb = None
        nei = int(getattr(getattr(npts, "config", None), "num_nei_cells", 99))
        if not query_locally and n >= self.global_bricks_min_queries and nei <= 2 and npts.count() > 0:
            gb = getattr(self, "_global_bricks", None)
            if gb is None or gb.cand_dx.shape[0] != npts.neighbor_K:
                gb = self._global_bricks = ops.BrickCache(npts.neighbor_dx.cpu().numpy(), nei, self.device)
            npts._wait_bricks()
            gb.build(npts.search_state(), time_filtering=False, local=False)
        for i in range(math.ceil(n / bs)):
            head, tail = i * bs, min((i + 1) * bs, n)
            q = coord[head:tail].detach().to(device=dev, dtype=torch.float32).contiguous()
            if gb is not None:
                nbr, nn, _ = ops.knn_query(npts.search_state(), q, self.config.query_nn_k, time_filtering=False, local=False, bricks=gb)
            else:
                nbr, nn, _ = npts.knn(q, query_locally)
            if query_sdf:  # 0 where no neural point is near (mesher.py:113-123)
                sdf, _, _, _ = ops.sdf_query(fs, q, nbr, nn, grad=False, std=False, certainty=False)
                sdf_d[head:tail] = sdf.masked_fill_(nn < 1, 0.0)
            if query_color:
                col, _, _ = ops.color_query(fc, q, nbr, nn, want_grad=False)
                col_d[head:tail] = col
            if query_sem:  # argmax of the (weighted) log-probabilities (mesher.py:137-145)
                lab, _ = ops.sem_query(fsem, q, nbr, nn, int(self.sem_mlp.out_dim))
                sem_d[head:tail] = lab
            if query_mask:
                mask_d[head:tail] = nn >= mask_min_nn_count
        def out(t, as_float64=False):
            if t is None:
                return None
            t = t.float().cpu()
            return t if out_torch else (t.numpy().astype(np.float64))
        return out(sdf_d), out(sem_d), out(col_d), out(mask_d)
