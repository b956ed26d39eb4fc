"""Worker of tests/test_gpu_mapping_scale.py: ONE rank of a whole `Mapper.mapping` call at scale, through the product path
(drop-in Mapper -> engine.MapTrainer -> libpinhip) on the bench's own workload (bench.WORKLOADS: the c3 map of BASELINE
configs 3 / 4, or the c5 colour map), with a LARGE global batch (2^17 by default): recompute-dW, the rows form of the lazy
Adam, the pool-record reuse (forced on for the one-GPU run) and -- world > 1 -- the spatial shards (pin_slam_amd.dp) with the
host-staged exchange between ranks that share cuda:0.

    _map_scale_worker.py <rank> <world> <port> <out.npz> <workload> <bs> <iters> [diverge]

Every rank is seeded alike, so all draw the same batches; the drawn indices are recorded (Mapper._draw_all is wrapped).
Rank 0 writes the state before the call (features, decoder, certainty, ts: what the oracle starts from), the drawn batches and
the state after it; the other ranks write SHA-256 digests of their end state (ranks must be bit-identical).
`diverge`: rank 1's replica of the drawn indices is changed in one entry before the partition -- the replica-consistency
check of the spatial mapper (dp.SpatialShards._check_replicas) must stop every rank."""
import hashlib
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pin_slam_amd import collective, synth  # noqa: E402
from pin_slam_amd.config import PinConfig  # noqa: E402
from pin_slam_amd.dropin.model.decoder import Decoder  # noqa: E402
from pin_slam_amd.dropin.model.neural_points import NeuralPoints  # noqa: E402
from pin_slam_amd.dropin.utils.mapper import Mapper  # noqa: E402

POOL = {"c3": 1_000_000, "c5": 600_000}


def build(workload, bs, iters):
    """bench.py's set-up of the workload (same map, same pool generator), with a global batch of `bs`."""
    wl = bench.WORKLOADS[workload]
    H, L = wl["hidden"], wl["levels"]
    colour = bool(wl.get("color", False))
    cfg = PinConfig(buffer_size=int(5e7), feature_std=0.1, bs=bs, iters=iters, local_map_travel_dist_ratio=5.0,
                    pool_capacity=POOL[workload], pool_filter_freq=1, bs_new_sample=0, geo_mlp_level=L, geo_mlp_hidden_dim=H,
                    color_mlp_level=L, color_mlp_hidden_dim=H, **wl["cfg"])
    torch.manual_seed(42)
    m = synth.build_map(layers=wl["layers"], resolution=wl["cfg"]["voxel_size_m"], **wl["map"])
    npts = NeuralPoints(cfg)
    npts.travel_dist = torch.zeros(8, dtype=torch.float32, device="cuda")
    npts.update(torch.from_numpy(m.positions).cuda(), torch.zeros(3), torch.eye(3), 0)
    # (NeuralPoints.update drops the few points whose hash slot is taken, neural_points.py:352-377: the oracle is given THIS
    # map's positions and table, not the generator's)
    assert 0.999 * len(m.positions) < npts.count() <= len(m.positions)
    dec = Decoder(cfg, H, L, 1)
    cdec = Decoder(cfg, H, L, 3) if colour else None
    mp = Mapper(cfg, bench.Dataset(4), npts, {"sdf": dec, "semantic": None, "color": cdec})
    pool_c, pool_l = synth.make_pool(m, n=POOL[workload], sigma=wl.get("pool_sigma", 0.25))
    if colour:
        mp.color_pool = torch.from_numpy(np.random.default_rng(9).random((len(pool_l), 3), dtype=np.float32)).cuda()
    mp.coord_pool = torch.from_numpy(pool_c).cuda()
    mp.global_coord_pool = mp.coord_pool.clone()
    mp.sdf_label_pool = torch.from_numpy(pool_l).cuda()
    mp.weight_pool = torch.ones(len(pool_l), dtype=torch.float32, device="cuda")
    mp.time_pool = torch.zeros(len(pool_l), dtype=torch.int32, device="cuda")
    mp.pool_sample_count = len(pool_l)
    mp._pool()
    mp._publish_pool()
    return cfg, npts, dec, cdec, mp


def digest(t):
    return hashlib.sha256(t.detach().cpu().contiguous().numpy().tobytes()).hexdigest()


def main(rank, world, port, out, workload, bs, iters, diverge=False, mode="spatial"):
    torch.cuda.set_device(0)
    comm = None
    if world > 1:
        import torch.distributed as dist
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        comm = collective.make_comm(rank, world, "host")
    cfg, npts, dec, cdec, mp = build(workload, bs, iters)
    if world > 1:
        mp.dp_rank, mp.dp_world, mp.dp_comm, mp.dp_mode = rank, world, comm, mode
    # the 2^20 path's record reuse -- one search over the (rank's) pool samples per call -- forced on at this batch (it is chosen
    # by batch x iterations / pool)
    mp.reuse_pool_records = True
    before = {}
    if rank == 0:
        before = dict(feat0=npts.local_geo_features.data.cpu().numpy().copy(), dec0=dec.flat_params().cpu().numpy().copy(),
                      cert0=npts.local_point_certainties.cpu().numpy().copy(), tsu0=npts.local_point_ts_update.cpu().numpy().copy())
        assert npts.local_count() == npts.count()  # (the whole synthetic map is inside the local radius: local row = global row)
        table = npts.buffer_pt_index
        slots = torch.nonzero(table >= 0).reshape(-1)
        before.update(pos=npts.neural_points.cpu().numpy().copy(), table_slots=slots.cpu().numpy(), table_vals=table[slots].cpu().numpy(),
                      table_size=np.array(table.shape[0]))
        if cdec is not None:
            before.update(cfeat0=npts.local_color_features.data.cpu().numpy().copy(), cdec0=cdec.flat_params().cpu().numpy().copy())
    drawn = {}
    draw_all = mp._draw_all

    def recording(n_iters):
        d = draw_all(n_iters)
        assert d["new"] is None
        if diverge and rank == 1:
            d["hist"][0, 5] = (d["hist"][0, 5] + 1) % mp.pool_sample_count
        drawn["hist"] = d["hist"].cpu().numpy().copy()
        return d
    mp._draw_all = recording
    err = ""
    try:
        mp.mapping(iters)
        torch.cuda.synchronize()
    except RuntimeError as e:
        if not diverge:
            raise
        err = str(e)
    t = mp._trainer
    info = dict(err=np.array(err), world=np.array(world), lazy_rows_launches=np.array(int(getattr(t.lazy, "rows_launches", -1))),
                records_reused=np.array((getattr(mp, "_pool_rec", None) is not None) if t.dp is None else bool(t.dp.n_own)))
    if t.dp is not None and not err:
        info.update(n_main=np.asarray(t.dp.n_main), n_eik=np.asarray(t.dp.n_eik), n_halo=np.array(t.dp.n_halo),
                    halo_rows=t.dp.halo_rows[:t.dp.n_halo].cpu().numpy(), owner=t.dp.owner[:t.dp.stats["rows"]].cpu().numpy())
    end = dict(feats=npts.local_geo_features.data, dec=dec.flat_params(), cert=npts.local_point_certainties,
               tsu=npts.local_point_ts_update, gfeats=npts.geo_features, gcert=npts.point_certainties)
    if cdec is not None:
        end.update(cfeats=npts.local_color_features.data, cdec=cdec.flat_params())
    sha = {"sha_" + k: np.array(digest(v)) for k, v in end.items()}
    if rank == 0:
        np.savez(out, hist=drawn.get("hist", np.zeros((0, 0), np.int64)), **before, **sha, **info,
                 **{k: v.detach().cpu().numpy() for k, v in end.items()})
    else:
        np.savez(out, **sha, **info)
    if comm is not None:
        comm.close()
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    a = sys.argv
    main(int(a[1]), int(a[2]), int(a[3]), a[4], a[5], int(a[6]), int(a[7]), diverge=len(a) > 8 and a[8] == "diverge",
         mode="dense" if len(a) > 8 and a[8] == "dense" else "spatial")
