"""Transports of the data-parallel mapper's two exchanges:

* ``allreduce_grads(flat)``  -- in-place SUM of the flat fp32 buffer [decoder grads | feature grads], once per
  Mapper.mapping iteration (the parameters of utils/mapper.py:604 are the feature table and the decoder);
* ``sync_side_effects(certainty, certainty0, scratch, ts_update)`` -- once per mapping call: certainty becomes
  certainty0 + SUM_ranks(certainty - certainty0), ts_update the MAX over ranks (query_feature's training-mode
  side effects, model/neural_points.py:660-683).

:class:`RcclComm` is the product transport: RCCL over xGMI through the C ABI (pin_allreduce_grads,
pin_dp_sync_side_effects) on the caller's HIP stream, no torch op involved.  The communicator is created from a
``torch.distributed`` process group only in the sense that the group carries the 128-byte ncclUniqueId from rank 0
to the others (set-up, once).

:class:`HostStagedComm` exists for ranks that SHARE one device (RCCL refuses two ranks on one GPU): tests on the
single-GPU box run two processes on cuda:0 and exchange through pinned host buffers over a gloo group.  It uses
the same element-wise kernels (pin_dp_cert_delta / pin_dp_cert_apply); it is never selected implicitly.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib, ops
from ._lib import check


def _stream():
    return ops._stream()


def rccl_library_path() -> str:
    """The librccl that ships with the torch in this process (one RCCL, one HIP runtime per process)."""
    p = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    return p if os.path.exists(p) else ""


class RcclComm:
    def __init__(self, rank: int, world: int, group=None, uid: bytes | None = None):
        """uid: the 128-byte ncclUniqueId already exchanged by the caller (make_comm does that on its MAIN thread, so that no
        torch.distributed collective is ever issued from the watchdog thread); None = exchange it here."""
        L = _lib.lib()
        if uid is None:
            uid = self.exchange_id(rank, world, group)
        buf = C.create_string_buffer(uid, _lib.PIN_COMM_ID_BYTES)
        handle = C.c_void_p()
        check(L.pin_comm_init_rank(buf, rank, world, C.byref(handle)), "pin_comm_init_rank")
        self._h, self.rank, self.world = handle, rank, world
        self.kind = "rccl"

    @staticmethod
    def exchange_id(rank: int, world: int, group=None) -> bytes:
        """Load librccl and carry rank 0's ncclUniqueId to every rank over the job's process group.  Collective: every rank
        calls it, every rank raises when ANY rank could not load the library or rank 0 could not make the id -- the
        failure travels in the same two object collectives, so the ranks never disagree about what comes next."""
        import torch.distributed as dist
        L = _lib.lib()
        err, uid = "", None
        try:
            check(L.pin_comm_load(rccl_library_path().encode()), "pin_comm_load")
            if rank == 0:
                b = C.create_string_buffer(_lib.PIN_COMM_ID_BYTES)
                check(L.pin_comm_unique_id(b), "pin_comm_unique_id")
                uid = bytes(b.raw)
        except Exception as e:  # noqa: BLE001
            err = f"{type(e).__name__}: {e}"
        if world > 1:
            box = [uid]
            dist.broadcast_object_list(box, src=0, group=group)
            uid = box[0]
            errs = [None] * world
            dist.all_gather_object(errs, err, group=group)
            err = next((f"rank {i}: {e}" for i, e in enumerate(errs) if e), "")
        if err or uid is None:
            raise RuntimeError(err or "rank 0 produced no ncclUniqueId")
        return uid

    def allreduce_grads(self, flat: torch.Tensor):
        check(_lib.lib().pin_allreduce_grads(self._h, flat.data_ptr(), flat.numel(), _stream()), "pin_allreduce_grads")

    def allreduce(self, send: torch.Tensor, recv: torch.Tensor):
        """SUM all-reduce of a flat fp32 buffer, out of place (send may be recv): pin_allreduce_f32."""
        check(_lib.lib().pin_allreduce_f32(self._h, send.data_ptr(), recv.data_ptr(), send.numel(), _stream()), "pin_allreduce_f32")

    def allgather(self, send: torch.Tensor, recv: torch.Tensor):
        """recv [world][send.numel()] <- every rank's send (pin_allgather_f32)."""
        check(_lib.lib().pin_allgather_f32(self._h, send.data_ptr(), recv.data_ptr(), send.numel(), _stream()), "pin_allgather_f32")

    def sync_side_effects(self, certainty, certainty0, scratch, ts_update):
        n = certainty.shape[0]
        check(_lib.lib().pin_dp_sync_side_effects(self._h, certainty.data_ptr(), certainty0.data_ptr(), scratch.data_ptr(),
                                                  ts_update.data_ptr(), n, _stream()), "pin_dp_sync_side_effects")

    def close(self):
        if self._h:
            check(_lib.lib().pin_comm_destroy(self._h), "pin_comm_destroy")
            self._h = C.c_void_p()


class HostStagedComm:
    """gloo over pinned host buffers, for ranks sharing one device (tests)."""

    def __init__(self, rank: int, world: int, group=None):
        self.rank, self.world, self.group = rank, world, group
        self._host = {}
        self.kind = "host-staged gloo"

    def _stage(self, t: torch.Tensor):
        """One growing pinned buffer per dtype, sliced to the message (the spatial exchange changes size every call)."""
        n = t.numel()
        h = self._host.get(t.dtype)
        if h is None or h.numel() < n:
            h = self._host[t.dtype] = torch.empty(int(n * 1.25) + 1024, dtype=t.dtype).pin_memory()
        return h[:n]

    def _allreduce(self, t: torch.Tensor, op):
        import torch.distributed as dist
        h = self._stage(t)
        h.copy_(t.reshape(-1), non_blocking=True)
        torch.cuda.current_stream().synchronize()
        dist.all_reduce(h, op=op, group=self.group)
        t.reshape(-1).copy_(h, non_blocking=True)

    def allreduce_grads(self, flat: torch.Tensor):
        import torch.distributed as dist
        self._allreduce(flat, dist.ReduceOp.SUM)

    def allreduce(self, send: torch.Tensor, recv: torch.Tensor):
        import torch.distributed as dist
        h = self._stage(send)
        h.copy_(send.reshape(-1), non_blocking=True)
        torch.cuda.current_stream().synchronize()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
        recv.reshape(-1).copy_(h, non_blocking=True)

    def allgather(self, send: torch.Tensor, recv: torch.Tensor):
        import torch.distributed as dist
        h = send.reshape(-1).cpu()
        parts = [torch.empty_like(h) for _ in range(self.world)]
        dist.all_gather(parts, h, group=self.group)
        recv.reshape(-1)[:self.world * h.numel()].copy_(torch.cat(parts))

    def sync_side_effects(self, certainty, certainty0, scratch, ts_update):
        import torch.distributed as dist
        L, n = _lib.lib(), certainty.shape[0]
        check(L.pin_dp_cert_delta(certainty.data_ptr(), certainty0.data_ptr(), scratch.data_ptr(), n, _stream()),
              "pin_dp_cert_delta")
        self._allreduce(scratch[:n], dist.ReduceOp.SUM)
        self._allreduce(ts_update[:n], dist.ReduceOp.MAX)
        check(L.pin_dp_cert_apply(certainty.data_ptr(), certainty0.data_ptr(), scratch.data_ptr(), n, _stream()),
              "pin_dp_cert_apply")

    def close(self):
        pass


class TorchComm:
    """The same exchanges through ``torch.distributed``'s own communicator (backend "nccl" = the RCCL inside torch, or gloo
    with device tensors in tests).  Selected explicitly (`bench.py --dp-transport torch`) or by :func:`make_comm` when
    :class:`RcclComm` does not pass its self-test on some rank; the element-wise kernels stay the C ABI's."""

    def __init__(self, rank: int, world: int, group=None):
        import torch.distributed as dist
        self.rank, self.world, self.group = rank, world, group
        self.kind = f"torch.distributed {dist.get_backend(group)}"

    def allreduce_grads(self, flat: torch.Tensor):
        import torch.distributed as dist
        dist.all_reduce(flat, group=self.group)

    def allreduce(self, send: torch.Tensor, recv: torch.Tensor):
        import torch.distributed as dist
        if recv.data_ptr() != send.data_ptr():
            recv.reshape(-1)[:send.numel()].copy_(send.reshape(-1))
        dist.all_reduce(recv.reshape(-1)[:send.numel()], group=self.group)

    def allgather(self, send: torch.Tensor, recv: torch.Tensor):
        import torch.distributed as dist
        n = send.numel()
        parts = list(recv.reshape(-1)[:self.world * n].split(n))
        dist.all_gather(parts, send.reshape(-1).contiguous(), group=self.group)

    def sync_side_effects(self, certainty, certainty0, scratch, ts_update):
        import torch.distributed as dist
        L, n = _lib.lib(), certainty.shape[0]
        check(L.pin_dp_cert_delta(certainty.data_ptr(), certainty0.data_ptr(), scratch.data_ptr(), n, _stream()),
              "pin_dp_cert_delta")
        dist.all_reduce(scratch[:n], group=self.group)
        dist.all_reduce(ts_update[:n], op=dist.ReduceOp.MAX, group=self.group)
        check(L.pin_dp_cert_apply(certainty.data_ptr(), certainty0.data_ptr(), scratch.data_ptr(), n, _stream()),
              "pin_dp_cert_apply")

    def close(self):
        pass


def self_test(comm, device="cuda") -> bool:
    """Both collectives of a transport on known values: SUM of (rank + 1) and the gathered ranks."""
    w, r = comm.world, comm.rank
    send = torch.full((257,), float(r + 1), dtype=torch.float32, device=device)
    recv = torch.zeros_like(send)
    comm.allreduce(send, recv)
    gat = torch.zeros((w, 5), dtype=torch.float32, device=device)
    comm.allgather(torch.full((5,), float(r), dtype=torch.float32, device=device), gat)
    want = torch.arange(w, dtype=torch.float32, device=device)[:, None].expand(w, 5)
    return bool((recv == w * (w + 1) / 2).all().item()) and bool((gat == want).all().item())


class TransportError(RuntimeError):
    """The requested transport could not be brought up (raised on every rank of the job together)."""


def make_comm(rank: int, world: int, transport: str = "rccl", group=None, device="cuda", fallback: bool = False):
    """The mapper's transport for this job.  "rccl": RCCL through the C ABI, checked with :func:`self_test` on every rank.
    When ANY rank fails to create the communicator or fails the test, EVERY rank raises :class:`TransportError` with the
    reason (the ranks agree on it through the job's process group first: nobody is left waiting in a collective) -- a job
    asked to run on RCCL never runs on something else.  `fallback=True` (an explicit opt-in; `bench.py --dp-transport auto`)
    makes all ranks move to :class:`TorchComm` together instead and say so in `kind`; mixed transports never happen."""
    import torch.distributed as dist
    if transport == "host":
        return HostStagedComm(rank, world, group)
    if transport == "torch":
        return TorchComm(rank, world, group)
    comm, why = None, ""
    box = {}
    # The id exchange uses the job's process group, so it runs HERE, on the main thread, and reports its failures through the
    # same collectives on every rank.  Only ncclCommInitRank and the self-test -- which touch no torch.distributed state --
    # run under the watchdog: a bootstrap that never returns (it blocks inside ncclCommInitRank on every rank alike) must end
    # in the other transport, not in a job that hangs.
    uid = None
    try:
        uid = RcclComm.exchange_id(rank, world, group)
    except Exception as e:  # noqa: BLE001 -- raised on every rank together (exchange_id)
        box["err"] = f"{type(e).__name__}: {e}"

    def bring_up():
        if uid is None:
            return
        try:
            c = RcclComm(rank, world, group, uid=uid)
            box["comm"] = c
            box["ok"] = self_test(c, device)
        except Exception as e:  # noqa: BLE001 -- any failure of the optional path selects the other one, on every rank
            box["err"] = f"{type(e).__name__}: {e}"

    if world > 1:
        import threading
        limit = float(os.environ.get("PIN_COMM_INIT_TIMEOUT", "120"))
        dev_index = torch.cuda.current_device() if device != "cpu" else None

        def in_thread():  # (a new thread starts on device 0: the kernels of the self-test must run on this rank's device)
            if dev_index is not None:
                torch.cuda.set_device(dev_index)
            bring_up()

        th = threading.Thread(target=in_thread, daemon=True)
        th.start()
        th.join(limit)
        if th.is_alive():
            box.setdefault("err", f"set-up or self-test did not finish within {limit:.0f} s")
            box["stuck"] = True
    else:
        bring_up()
    comm = box.get("comm")
    if "err" in box:
        why = box["err"]
    elif not box.get("ok", False):
        why = "self-test mismatch"
    if world > 1:
        flags = [None] * world
        dist.all_gather_object(flags, why, group=group)
        why = next((f"rank {i}: {f}" for i, f in enumerate(flags) if f), "")
    if not why:
        return comm
    if comm is not None and not box.get("stuck"):
        try:
            comm.close()
        except Exception:  # noqa: BLE001
            pass
    if not fallback:
        err = TransportError(f"RCCL transport not available ({why[:300]}); no other transport was asked for "
                             f"(make_comm(fallback=True) / bench.py --dp-transport auto | torch | host select one)")
        err.abandoned_thread = bool(box.get("stuck"))  # a bootstrap thread may still sit in the library: leave with os._exit
        raise err
    alt = TorchComm(rank, world, group)
    if not self_test(alt, device):
        raise RuntimeError(f"no working transport: RcclComm failed ({why}) and torch.distributed fails its self-test")
    alt.kind += f" (RcclComm not used: {why[:200]})"
    alt.abandoned_thread = bool(box.get("stuck"))  # the caller should leave with os._exit: a thread still sits in the bootstrap
    return alt


class NullComm:
    """One rank of an N-rank job measured ALONE (bench.py's per-rank emulation on a single-GPU box): every exchange is
    the identity, so the rank does exactly its own share of the work (its samples, its rows, the whole halo) and
    nothing arrives from the others.  Timing only -- the trained map is not the N-rank result."""

    def __init__(self, rank: int, world: int):
        self.rank, self.world, self.kind = rank, world, "none (single rank emulated)"

    def allreduce_grads(self, flat):
        pass

    def allreduce(self, send, recv):
        pass  # (also for the owner merge: the map of the one emulated rank stays whole)

    def allgather(self, send, recv):
        pass

    def sync_side_effects(self, certainty, certainty0, scratch, ts_update):
        pass

    def close(self):
        pass
