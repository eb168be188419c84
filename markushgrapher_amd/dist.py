"""Data-parallel sharding of image batches over the GPUs of one node: images are independent
(the reference processes them one by one, ref: markushgrapher/utils/ocsr/utils_evaluation.py:140), weights are
replicated, so the path shards with NO data-path collective; the only exchange is the all-gather of the decoded token
ids per batch (RCCL over xGMI on the GPU box — backend "nccl"; "gloo" in the CPU tests).

One process per GPU (torch.distributed.run sets RANK / LOCAL_RANK / WORLD_SIZE).

The payload is the one SURVEY.md §8e fixes: per rank a STATIC [rows, 512] int32 block of ids (rows padded with the pad id) and
a [rows] int32 vector of row lengths - 64 KiB + 128 B per rank at 32 rows, latency-bound.  `IdExchange` posts both gathers
asynchronously on double-buffered send/receive blocks, so that the exchange of batch i overlaps the encoder of batch i+1;
`sharded_generate` is the blocking library form.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

ID_COLS = 512          # static width of the exchanged id block (= the reference's max_length, utils_evaluation.py:280)


def shard_bounds(n_items: int, world: int, rank: int):
    """Contiguous shards; the first n_items % world ranks take one extra item."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class IdExchange:
    """All-gather of decoded ids as [rows, 512] int32 + [rows] int32 lengths, asynchronous and double-buffered.

        ex = IdExchange(rows_per_rank, device)
        h = ex.post(ids)            # ids [n <= rows, T <= 512] integer tensor of this rank; returns immediately
        ...                         # next batch's encoder runs here
        all_ids, all_len = ex.wait(h)   # [world * rows, 512] int32, [world * rows] int32 (length 0 = unused row);
                                        # views of the slot's receive blocks: valid until the slot is posted again
    """

    def __init__(self, rows: int, device, pad_token_id: int = 0, group: Optional[dist.ProcessGroup] = None, cols: int = ID_COLS,
                 always_collective: bool = False, native: Optional[bool] = None):
        """always_collective: issue the all-gather even in a group of one rank (exercises the RCCL path on a single GPU).
        native: the gathers go through the library's own RCCL entry points (include/mgrapher.h mg_dist_*: ncclAllGather on a stream of
        this object) instead of torch.distributed's; torch.distributed then only carries the 128-byte unique id at set-up.  Default: the
        environment's MG_DIST_NATIVE=1, else torch.distributed (the path the multi-rank CPU tests cover)."""
        self.rows, self.cols, self.pad, self.group = rows, cols, pad_token_id, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.collective = self.world > 1 or (always_collective and dist.is_initialized())
        if native is None:
            import os
            native = os.environ.get("MG_DIST_NATIVE", "0") == "1"
        self._native = None
        if native and self.collective and torch.device(device).type == "cuda":
            self._native_setup(torch.device(device))
        mk = lambda *shape: torch.empty(shape, dtype=torch.int32, device=device)
        self.send = [mk(rows, cols), mk(rows, cols)]
        self.slen = [mk(rows), mk(rows)]
        self.recv = [mk(self.world * rows, cols), mk(self.world * rows, cols)]
        self.rlen = [mk(self.world * rows), mk(self.world * rows)]
        self.pending: List[Optional[Tuple]] = [None, None]
        self.turn = 0

    def post(self, ids: torch.Tensor, lengths: Optional[torch.Tensor] = None) -> int:
        slot = self.turn
        self.turn ^= 1
        if self.pending[slot] is not None:            # the buffers of this slot are still in flight: finish that exchange
            self.wait(slot)
        n, t = int(ids.shape[0]), int(ids.shape[1])
        if n > self.rows or t > self.cols:
            raise ValueError(f"ids {tuple(ids.shape)} do not fit the exchange block [{self.rows}, {self.cols}]")
        s, sl = self.send[slot], self.slen[slot]
        s.fill_(self.pad)
        s[:n, :t] = ids.to(torch.int32)
        sl.zero_()
        sl[:n] = t if lengths is None else lengths.to(torch.int32)
        if not self.collective:
            self.recv[slot].copy_(s)
            self.rlen[slot].copy_(sl)
            self.pending[slot] = ()
        elif self._native is not None:
            import ctypes as C
            lib, comm, st = self._native
            ready = torch.cuda.Event()
            ready.record()                               # the send blocks are complete on the caller's stream
            st.wait_event(ready)
            for src, dst in ((s, self.recv[slot]), (sl, self.rlen[slot])):
                rc = lib.mg_dist_allgather(comm, C.c_void_p(st.cuda_stream), C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()),
                                           C.c_size_t(src.numel() * src.element_size()))
                if rc != 0:
                    raise RuntimeError(f"mg_dist_allgather failed ({rc}): {lib.mg_last_error().decode()}")
            done = torch.cuda.Event()
            done.record(st)
            self.pending[slot] = (_NativeWork(done),)
        else:
            self.pending[slot] = (dist.all_gather_into_tensor(self.recv[slot], s, group=self.group, async_op=True),
                                  dist.all_gather_into_tensor(self.rlen[slot], sl, group=self.group, async_op=True))
        return slot

    def _native_setup(self, device):
        """RCCL communicator of the library (mg_dist_create): rank 0's unique id travels through torch.distributed once."""
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        lib.mg_last_error.restype = C.c_char_p
        lib.mg_dist_unique_id.argtypes = [C.c_void_p, C.c_int]
        lib.mg_dist_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        lib.mg_dist_allgather.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        lib.mg_dist_destroy.argtypes = [C.c_void_p]
        rank = dist.get_rank(self.group)
        buf = (C.c_char * 128)()
        if rank == 0 and lib.mg_dist_unique_id(buf, 128) != 0:
            raise RuntimeError("mg_dist_unique_id: " + lib.mg_last_error().decode())
        box = [bytes(buf)]
        dist.broadcast_object_list(box, src=dist.get_global_rank(self.group, 0) if self.group is not None else 0, group=self.group)
        comm = C.c_void_p()
        with torch.cuda.device(device):
            rc = lib.mg_dist_create(box[0], 128, rank, self.world, C.byref(comm))
            if rc != 0:
                raise RuntimeError("mg_dist_create: " + lib.mg_last_error().decode())
            self._native = (lib, comm, torch.cuda.Stream(device=device))

    def close(self):
        if self._native is not None:
            lib, comm, st = self._native
            st.synchronize()
            lib.mg_dist_destroy(comm)
            self._native = None

    def wait(self, slot: int):
        works = self.pending[slot]
        if works is None:
            raise RuntimeError("nothing posted on this slot")
        for w in works:
            w.wait()
        self.pending[slot] = None
        return self.recv[slot], self.rlen[slot]


class _NativeWork:
    """wait() of a gather issued through mg_dist_allgather: its event, polled (the runtime's own waits spin)."""

    def __init__(self, event):
        self.event = event

    def wait(self):
        import time
        while not self.event.query():
            time.sleep(0.0001)
        torch.cuda.current_stream().wait_event(self.event)


def sharded_generate(generate_fn: Callable[..., torch.Tensor], batch: Dict[str, torch.Tensor], max_length: int,
                     pad_token_id: int = 0, group: Optional[dist.ProcessGroup] = None, **gen_kw) -> torch.Tensor:
    """Every rank passes the same global batch; rank r decodes images [lo_r, hi_r) with `generate_fn(**shard, max_length=...)`
    (e.g. MarkushgrapherForConditionalGeneration.generate) and the ids are exchanged with IdExchange.  Returns
    [B_global, max_length] int64 on every rank, in input order, rows padded with pad_token_id."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    B = batch["input_ids"].shape[0]
    lo, hi = shard_bounds(B, world, rank)
    per = (B + world - 1) // world
    dev = batch["input_ids"].device
    ex = IdExchange(per, dev, pad_token_id, group, cols=max(ID_COLS, max_length))
    if hi > lo:
        shard = {k: v[lo:hi] for k, v in batch.items() if v is not None}
        ids = generate_fn(**shard, max_length=max_length, **gen_kw)
    else:
        ids = torch.zeros((0, 1), dtype=torch.int64, device=dev)
    got, _ = ex.wait(ex.post(ids))
    rows = []
    for r in range(world):
        a, b = shard_bounds(B, world, r)
        rows.append(got[r * per:r * per + (b - a), :max_length])
    return torch.cat(rows, dim=0).to(torch.int64)
