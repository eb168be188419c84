"""Data-parallel sharding of image batches over the GPUs of one node: images are independent
(the reference processes them one by one, ref: markushgrapher/utils/ocsr/utils_evaluation.py:140), weights are
replicated, so the path shards with NO data-path collective; the only exchange is one all-gather of the decoded
token ids per batch (RCCL over xGMI on the GPU box — backend "nccl"; "gloo" in the CPU tests).

One process per GPU (torch.distributed.run sets RANK / LOCAL_RANK / WORLD_SIZE).
"""
from __future__ import annotations

from typing import Callable, Dict, Optional

import torch
import torch.distributed as dist


def shard_bounds(n_items: int, world: int, rank: int):
    """Contiguous shards; the first n_items % world ranks take one extra item."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def sharded_generate(generate_fn: Callable[..., torch.Tensor], batch: Dict[str, torch.Tensor], max_length: int,
                     pad_token_id: int = 0, group: Optional[dist.ProcessGroup] = None, **gen_kw) -> torch.Tensor:
    """Every rank passes the same global batch; rank r decodes images [lo_r, hi_r) with `generate_fn(**shard, max_length=...)`
    (e.g. MarkushgrapherForConditionalGeneration.generate) and the ids, padded to [*, max_length] with pad_token_id so
    the shape is static, are all-gathered.  Returns [B_global, max_length] int64 on every rank, in input order."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    B = batch["input_ids"].shape[0]
    lo, hi = shard_bounds(B, world, rank)
    per = (B + world - 1) // world
    dev = batch["input_ids"].device
    out = torch.full((per, max_length), pad_token_id, dtype=torch.int64, device=dev)
    if hi > lo:
        shard = {k: v[lo:hi] for k, v in batch.items() if v is not None}
        ids = generate_fn(**shard, max_length=max_length, **gen_kw)
        out[:hi - lo, :ids.shape[1]] = ids
    if world == 1:
        return out[:B]
    gathered = torch.empty((world * per, max_length), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(gathered, out, group=group)
    rows = []
    for r in range(world):
        a, b = shard_bounds(B, world, r)
        rows.append(gathered[r * per:r * per + (b - a)])
    return torch.cat(rows, dim=0)
