"""Build device-side state (pin_slam_amd.ops.SearchState / FieldState) from a golden fixture."""
import numpy as np
import torch

from pin_slam_amd import ops
from pin_slam_amd._lib import PIN_NONLOCAL


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda().contiguous()


def g2l_to_device_format(global2local, local_mask):
    """Reference global2local maps NON-LOCAL points to 1 (neural_points.py:498 quirk); the
    device format marks them PIN_NONLOCAL so the kernels can reproduce-and-flag them."""
    g = np.asarray(global2local).astype(np.int32).copy()
    P = g.shape[0] - 1
    g[:P][~np.asarray(local_mask)[:P]] = PIN_NONLOCAL
    return g


def search_state(d, table=None):
    P = d["neural_points"].shape[0]
    if table is None:
        table = np.full(int(d["buffer_size"]), -1, np.int32)
        table[d["table_slots"]] = d["table_vals"].astype(np.int32)
    pos = dev(d["neural_points"], torch.float32)
    ts = dev(d["point_ts_create"], torch.int32)
    pos4 = torch.empty((P, 4), dtype=torch.float32, device="cuda")
    ops.pack_positions(pos, ts, pos4)
    cand = ops.candidate_offsets(d["neighbor_dx"].astype(np.int32), int(d["buffer_size"]))
    return ops.SearchState(
        table=dev(table, torch.int32), pos4=pos4, cand_off=dev(cand), n_points=P,
        resolution=d["resolution"], max_valid_dist2=d["max_valid_dist2"],
        travel_dist=dev(d["travel_dist"], torch.float32), cur_ts=int(d["cur_ts"]),
        diff_travel_dist_local=d["diff_travel_dist_local"],
        global2local=dev(g2l_to_device_format(d["global2local"], d["local_mask"])))


def field_state(d, local=True, orient=None, weighted_first=None, dec=None):
    pre = "local_" if local else ""
    feats = d["local_geo_features"] if local else d["geo_features"]
    cert = d["local_point_certainties"] if local else d["point_certainties"]
    pos = d["local_neural_points"] if local else d["neural_points"]
    return ops.FieldState(
        feats=dev(feats, torch.float32), dec=dev(d["dec_flat"] if dec is None else dec, torch.float32),
        k=int(d["query_nn_k"]), hidden=int(d["dec_hidden"]), levels=int(d["dec_levels"]),
        weighted_first=bool(d["weighted_first"]) if weighted_first is None else weighted_first,
        sdf_scale=d["sdf_scale"], certainty=dev(cert, torch.float32),
        orient=None if orient is None else dev(orient, torch.float32), pos=dev(pos, torch.float32))


def nbr_split(nbr):
    """[N,k,4] device kNN record -> (vec [N,k,3] f32, idx [N,k] int32 without the flag bit,
    flagged [N,k] bool) on the host."""
    a = nbr.cpu().numpy()
    raw = a[..., 3].view(np.int32).copy()
    flag = (raw >= 0) & ((raw & 0x40000000) != 0)
    idx = np.where(raw >= 0, raw & ~0x40000000, raw)
    return a[..., :3], idx, flag
