"""Multi-GPU: independent e-mails shard across ranks with no data-path collective; the only exchange
is collecting the fixed-size result records.

One process per GPU, ``torch.distributed`` ("nccl" is RCCL on ROCm; "gloo" for the CPU tests).
``shard_range`` balances contiguous ranges by cumulative raw bytes, not by count, so a ragged batch
loads the ranks evenly; ``gather_records`` is one all_gather of ``n_max x 192`` bytes per rank,
``gather_witnesses`` one of ``n_max x 72`` bytes (status + the two output hashes: what ``verify_email`` returns).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

from ._abi import RESULT_DTYPE, WITNESS_DTYPE


def shard_bounds(sizes: Sequence[int], world: int) -> List[int]:
    """bounds[r]..bounds[r+1] is rank r's contiguous range; cut points sit where the cumulative byte count
    crosses r/world of the total."""
    n = len(sizes)
    if world <= 1 or n == 0:
        return [0] + [n] * max(world, 1)
    cum = np.concatenate([[0], np.cumsum(np.asarray(sizes, dtype=np.int64))])
    total = int(cum[-1])
    bounds = [0]
    for r in range(1, world):
        target = total * r / world
        cut = int(np.searchsorted(cum, target, side="left"))
        cut = min(max(cut, bounds[-1]), n)
        bounds.append(cut)
    bounds.append(n)
    return bounds


def shard_range(sizes: Sequence[int], world: int, rank: int) -> Tuple[int, int]:
    b = shard_bounds(sizes, world)
    return b[rank], b[rank + 1]


def gather_records(local: np.ndarray, counts: Sequence[int], device=None) -> np.ndarray:
    """All-gather the per-rank result records (RESULT_DTYPE) into batch order on every rank.
    ``counts[r]`` = number of e-mails rank r holds (known to all ranks from ``shard_bounds``)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    n_max = max(counts) if len(counts) else 0
    buf = np.zeros(n_max, dtype=RESULT_DTYPE)
    buf[:len(local)] = local
    t = torch.from_numpy(buf.view(np.uint8).reshape(-1).copy())
    if device is not None:
        t = t.to(device)
    out = torch.empty(world * t.numel(), dtype=torch.uint8, device=t.device)
    dist.all_gather_into_tensor(out, t)
    rec = out.cpu().numpy().view(RESULT_DTYPE).reshape(world, n_max)
    return np.concatenate([rec[r, :counts[r]] for r in range(world)]) if world else rec.reshape(-1)


def witness_of(records: np.ndarray) -> np.ndarray:
    """The 72-byte witness of each 192-byte record: status, detail, from_domain_hash, public_key_hash."""
    w = np.zeros(len(records), dtype=WITNESS_DTYPE)
    for f in WITNESS_DTYPE.names:
        w[f] = records[f]
    return w


def witness_tensor(records_u8):
    """Same on a torch uint8 tensor of whole records (any device): (n * 192,) -> (n * 72,), one fused copy."""
    import torch
    r = records_u8.view(-1, RESULT_DTYPE.itemsize)
    return torch.cat([r[:, 0:8], r[:, 32:96]], dim=1).reshape(-1)


def gather_witnesses(local: np.ndarray, counts: Sequence[int], device=None) -> np.ndarray:
    """All-gather the witnesses of the per-rank result records into batch order on every rank (WITNESS_DTYPE)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    n_max = max(counts) if len(counts) else 0
    buf = np.zeros(n_max, dtype=RESULT_DTYPE)
    buf[:len(local)] = local
    t = torch.from_numpy(buf.view(np.uint8).reshape(-1).copy())
    if device is not None:
        t = t.to(device)
    w = witness_tensor(t)
    out = torch.empty(world * w.numel(), dtype=torch.uint8, device=w.device)
    dist.all_gather_into_tensor(out, w)
    rec = out.cpu().numpy().view(WITNESS_DTYPE).reshape(world, n_max)
    return np.concatenate([rec[r, :counts[r]] for r in range(world)]) if world else rec.reshape(-1)
