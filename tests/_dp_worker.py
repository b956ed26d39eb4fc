"""Worker of tests/test_gpu_dp.py: one rank of engine.MapTrainer on cuda:0, two mapping iterations on a golden
fixture's fixed batches.  transport: none | host (gloo over pinned host buffers, for ranks that share the device) |
rccl (RCCL through the C ABI; one rank per GPU, so world must be 1 on a single-GPU box) | null (collective.NullComm).
mode: dense (contiguous index shards, whole-table all-reduce) | spatial (pin_slam_amd.dp: k-d boxes, halo exchange)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pin_slam_amd import collective, engine, sharding  # noqa: E402
from tests import golden_util as G  # noqa: E402
from tests import gpu_util as U  # noqa: E402


def main(rank, world, port, case, transport, out, mode="dense"):
    torch.cuda.set_device(0)
    comm = None
    if transport == "null":
        comm = collective.NullComm(rank, world)
    elif transport != "none":
        import torch.distributed as dist
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        # rccl (strict: TransportError on every rank if it cannot be brought up) | auto (rccl, else torch.distributed on every
        # rank together) | torch (here: gloo on device tensors) | host
        comm = collective.make_comm(rank, world, "rccl" if transport == "auto" else transport, fallback=transport == "auto")
    d = G.load(case)
    st, fs = U.search_state(d), U.field_state(d)
    tsu = U.dev(d["local_point_ts_update"], torch.int32)
    bs = d["map_coord0"].shape[0]
    t = engine.MapTrainer(st, fs, None, None, None, None, tsu, bs=bs, decimation=int(d["map_dec"]), sigma=d["sdf_scale"],
                          weight_e=d["map_weight_e"], eik_eps=d["map_eps"], lr=d["map_lr"], adam_eps=d["map_adam_eps"],
                          loss_weight_on=bool(d.get("map_loss_weight_on", False)), rank=rank, world=world, comm=comm,
                          dp_mode=mode.split("-")[0])
    if mode == "spatial-skew" and rank != 0:
        # this rank's HOST comes up with other boxes than rank 0's: it must still run with rank 0's (pin_dp_boxes_decode)
        from pin_slam_amd import dp as dpm
        t.dp.fixed_boxes = dpm.kd_boxes(np.floor(d["map_coord1"][::3] / d["resolution"]).astype(np.int32) + 2, world)
    if mode == "spatial-reduce":  # the end-of-call merge as an all-reduce of the table instead of the all-gather of owned rows
        t.dp.merge = "reduce"
    mode = mode.split("-")[0]
    fc = None
    if "cdec_flat" in d:  # a colour map (replica_color): the colour branch rides along (mapper.py:668-671, 802-812)
        import dataclasses
        fc = dataclasses.replace(fs, feats=U.dev(d["local_color_features"]), dec=U.dev(d["cdec_flat"]), hidden=int(d["cdec_hidden"]),
                                 levels=int(d["cdec_levels"]), out_dim=3)
        t.set_color(fc, surface_range=d["surface_sample_range_m"], weight_i=d["weight_i"])
    grads, cgrads = [], []

    def on_grads(g):
        grads.append(g.cpu().numpy().copy())
        if fc is not None:  # this rank's colour-feature gradient after the exchange: private rows (the halo rows were moved out)
            cgrads.append(t.cgrad[fc.dec.numel():].cpu().numpy().copy())
    t.on_grads = on_grads
    cpay = []  # dense shards: the colour payload [colour decoder | colour features] behind its all-reduce
    t.on_color_grads = lambda g: cpay.append(g.cpu().numpy().copy())
    nd = fs.dec.numel()
    extra = {}
    if comm is not None and mode == "spatial":
        t.reset_optimizer(2)
        t.begin_side_effects()
        cat = lambda key, dt=None: U.dev(np.concatenate([d[f"{key}0"], d[f"{key}1"]]), dt)
        pool = dict(coord=cat("map_coord"), sdf_label=cat("map_label"), weight=cat("map_w"), ts=cat("map_ts", torch.int32))
        pool["global_coord"] = pool["coord"]
        if fc is not None:
            pool["color"] = cat("map_color")
        hist = torch.arange(2 * bs, dtype=torch.int64, device="cuda").reshape(2, bs)  # batch it = pool rows [it*bs, (it+1)*bs)
        stats = t.plan_shards(pool["coord"], hist, None, None, num_nei_cells=int(np.abs(d["neighbor_dx"]).max()),
                              pool_label=pool["sdf_label"])
        t.run_shards(pool, False, 2)
        t.finish_optimizer()
        t.merge_side_effects()
        torch.cuda.synchronize()
        extra = dict(n_main=t.dp.n_main, n_eik=t.dp.n_eik, n_halo=np.array(t.dp.n_halo), rows=np.array(stats["rows"]),
                     boxes=t.dp.boxes, halo_rows=t.dp.halo_rows[:t.dp.n_halo].cpu().numpy(),
                     owner=t.dp.owner[:stats["rows"]].cpu().numpy())
        ndx, nh8 = t.dp.nd, 8 * t.dp.n_halo  # exchange buffer: [decoders | geometry halo | colour halo]
        gsave = dict(gdec0=grads[0][:nd], gdec1=grads[1][:nd], ghalo0=grads[0][ndx:ndx + nh8], ghalo1=grads[1][ndx:ndx + nh8])
        if fc is not None:
            gsave.update(cgdec0=grads[0][nd:ndx], chalo0=grads[0][ndx + nh8:ndx + 2 * nh8], cprivate0=cgrads[0])
    else:
        t.reset_optimizer(2 if transport == "none" else None)  # one GPU: the lazy exact Adam, as Mapper.mapping runs it
        t.begin_side_effects()
        a, b = sharding.shard_range(bs, rank, world)
        for it in range(2):
            t.step_batch(U.dev(d[f"map_coord{it}"][a:b]), U.dev(d[f"map_label{it}"][a:b]), U.dev(d[f"map_w{it}"][a:b]),
                         U.dev(d[f"map_ts{it}"][a:b], torch.int32), it + 1,
                         color_label=None if fc is None else U.dev(d[f"map_color{it}"][a:b]))
        t.finish_optimizer()
        t.merge_side_effects()
        torch.cuda.synchronize()
        gsave = dict(gdec0=grads[0][:nd], gfeat0=grads[0][nd:], gdec1=grads[1][:nd], gfeat1=grads[1][nd:])
        if cpay:
            cnd = fc.dec.numel()
            gsave.update(cgdec0=cpay[0][:cnd], cgfeat0=cpay[0][cnd:])
    if fc is not None:
        extra = dict(extra, cfeats=fc.feats.cpu().numpy(), cdec=fc.dec.cpu().numpy())
    np.savez(out, feats=fs.feats.cpu().numpy(), dec=fs.dec.cpu().numpy(), cert=fs.certainty.cpu().numpy(),
             tsu=tsu.cpu().numpy(), kind=np.array(getattr(comm, "kind", "none")), **gsave, **extra)
    if comm is not None:
        comm.close()
        if transport != "null":
            import torch.distributed as dist
            dist.destroy_process_group()
        if getattr(comm, "abandoned_thread", False):
            os._exit(0)  # (collective.make_comm: a bootstrap thread of the unused transport is still blocked)


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5], sys.argv[6],
         *(sys.argv[7:8]))
