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


class ShardedVerifier:
    """ONE batch of e-mails verified by N ranks (BASELINE.json configs[3]: "65 536 e-mails ... sharded across 8 x MI355X";
    SURVEY §8(e)).  Every e-mail is independent (core/src/circuits.rs:9-29 touches only its own &Email), so the data path has
    no collective: rank r takes the contiguous range ``shard_bounds(sizes, world)[r : r + 1]`` — balanced by cumulative raw
    bytes, not by count —, runs the identical pipeline on it, and ONE all-gather of the 72-byte witnesses (status, detail,
    the two output hashes) puts the whole batch's results on every rank, in batch order.

    ``engine`` is a :class:`zkemail_rs_amd.Engine` (the product: the rank's range lives in HBM, it is cut into chunks that go
    through ``zke_verify_batch_device`` slot by slot, the all-gather is ordered behind them on the device by
    ``zke_engine_join`` — no host wait between the last launch and the collective).  For the CPU tier of the tests any object
    with ``verify_batch(PackedBatch) -> records`` does (the oracle): the same sharding and gathering code runs over gloo."""

    def __init__(self, engine, rank: int = 0, world: int = 1, device=None, chunk: int = 1024, slots: int = 8):
        self.engine, self.rank, self.world, self.device = engine, int(rank), max(1, int(world)), device
        self.chunk, self.slots = int(chunk), int(slots)
        self.on_device = device is not None and getattr(device, "type", str(device)) == "cuda" and hasattr(engine, "verify_batch_device")
        self.bounds: List[int] = []
        self.n_total = 0
        self._local = None          # PackedBatch of this rank's range
        self._dev = None            # device tensors + per-chunk zke_batch descriptors

    # ---- the batch
    def load(self, emails: Sequence) -> Tuple[int, int]:
        """Take this rank's byte-balanced range of ``emails`` (the whole batch, in batch order; every rank is given the same
        list, or at least the same sizes).  Returns (lo, hi)."""
        from ._abi import PackedBatch
        sizes = [len(e.raw_email) for e in emails]
        self.bounds = shard_bounds(sizes, self.world)
        self.n_total = len(emails)
        lo, hi = self.bounds[self.rank], self.bounds[self.rank + 1]
        self._local = PackedBatch(list(emails[lo:hi])) if hi > lo else None
        self._dev = None
        if self.on_device and self._local is not None:
            self._to_device()
        return lo, hi

    def _to_device(self):
        import ctypes as C
        import torch
        from . import _abi as A
        p, dev = self._local, self.device

        def t(arr, pad=0):
            a = np.ascontiguousarray(arr).view(np.uint8)
            if pad:
                a = np.concatenate([a, np.zeros(pad, np.uint8)])
            return torch.from_numpy(a.copy()).to(dev)

        keep = {"raw": t(p.raw_blob, 64), "raw_off": t(p.raw_off), "dom": t(p.domain_blob, 64), "dom_off": t(p.domain_off),
                "key": t(p.key_blob, 64), "key_off": t(p.key_off), "ktype": t(p.key_type), "ext": t(p.ext_null)}
        n = p.n
        keep["records"] = torch.zeros(max(n, 1) * RESULT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        chunks = []
        max_raw = 0
        for a in range(0, n, self.chunk):
            b = min(n, a + self.chunk)
            cb = A.zke_batch()
            cb.n = b - a
            cb.raw_blob, cb.domain_blob, cb.key_blob = keep["raw"].data_ptr(), keep["dom"].data_ptr(), keep["key"].data_ptr()
            cb.raw_off = keep["raw_off"].data_ptr() + 8 * a         # absolute offsets into the same blobs: the kernels
            cb.domain_off = keep["dom_off"].data_ptr() + 8 * a      # subtract off[0] where they need a range-relative one
            cb.key_off = keep["key_off"].data_ptr() + 8 * a
            cb.key_type, cb.ext_null = keep["ktype"].data_ptr() + a, keep["ext"].data_ptr() + a
            cb.with_regex = 0
            tot = (int(p.raw_off[b] - p.raw_off[a]), int(p.domain_off[b] - p.domain_off[a]), int(p.key_off[b] - p.key_off[a]))
            max_raw = max(max_raw, tot[0])
            chunks.append((cb, tot, keep["records"].data_ptr() + RESULT_DTYPE.itemsize * a))
        n_max = max(self.bounds[r + 1] - self.bounds[r] for r in range(self.world))
        keep["wit_local"] = torch.zeros(n_max * WITNESS_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        keep["wit_all"] = torch.zeros(self.world * n_max * WITNESS_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        self.engine.reserve(min(self.chunk, n), max_raw, min(self.slots, max(1, len(chunks))), 0)
        self._dev = (keep, chunks, n_max)

    # ---- one pass over the batch
    def verify(self):
        """Verify this rank's range and exchange the witnesses.  Returns the WHOLE batch's witnesses in batch order:
        a torch uint8 tensor of ``n_total * 72`` bytes on the device (product path) or a numpy WITNESS_DTYPE array (CPU tier)."""
        counts = [self.bounds[r + 1] - self.bounds[r] for r in range(self.world)]
        if self.n_total == 0:                    # every rank knows it: nothing to verify, nothing to exchange
            if not self.on_device:
                return np.zeros(0, WITNESS_DTYPE)
            import torch
            return torch.zeros(0, dtype=torch.uint8, device=self.device)
        if not self.on_device:
            local = self.engine.verify_batch(self._local) if self._local is not None else np.zeros(0, RESULT_DTYPE)
            if self.world == 1:
                return witness_of(local)
            return gather_witnesses(local, counts, self.device)
        import torch
        import torch.distributed as dist
        if self._dev is None:                    # this rank's range is empty
            n_max = max(counts)
            wit_local = torch.zeros(n_max * WITNESS_DTYPE.itemsize, dtype=torch.uint8, device=self.device)
            wit_all = torch.zeros(self.world * n_max * WITNESS_DTYPE.itemsize, dtype=torch.uint8, device=self.device)
        else:
            keep, chunks, n_max = self._dev
            for cb, tot, out_ptr in chunks:
                self.engine.verify_batch_device(cb, tot[0], tot[1], tot[2], out_ptr, 0)
            # the consumer of the records — two strided copies and the collective — is ordered behind every chunk on the device
            self.engine.join(torch.cuda.current_stream(self.device).cuda_stream)
            wit_local, wit_all = keep["wit_local"], keep["wit_all"]
            n = self._local.n
            r = keep["records"][:n * RESULT_DTYPE.itemsize].view(-1, RESULT_DTYPE.itemsize)
            w = wit_local[:n * WITNESS_DTYPE.itemsize].view(-1, WITNESS_DTYPE.itemsize)
            w[:, 0:8].copy_(r[:, 0:8])
            w[:, 8:72].copy_(r[:, 32:96])
        if self.world == 1:
            return wit_local[:counts[0] * WITNESS_DTYPE.itemsize]
        dist.all_gather_into_tensor(wit_all, wit_local)
        per = wit_all.view(self.world, n_max * WITNESS_DTYPE.itemsize)
        return torch.cat([per[r, :counts[r] * WITNESS_DTYPE.itemsize] for r in range(self.world)])

    def local_records(self):
        """This rank's full 192-byte records (numpy RESULT_DTYPE) of the last verify() — the intermediates stay on the rank."""
        if not self.on_device or self._dev is None:
            return None
        import torch
        torch.cuda.synchronize(self.device)
        keep = self._dev[0]
        return keep["records"][:self._local.n * RESULT_DTYPE.itemsize].cpu().numpy().view(RESULT_DTYPE)
