"""Multi-GPU side of the hot path: one process per GPU, crops sharded over ranks, results gathered.

The reference reaches its collectives only through mmengine: ``DefaultSampler(shuffle=False,
round_up=False)`` gives rank r the dataset indices ``r, r + P, r + 2P, ...`` and ``collect_results`` gathers
pickled per-sample dicts on rank 0 and re-interleaves them. Person crops are independent units, so the
MI355X path needs exactly one exchange: an RCCL ``all_gather`` of a fixed-layout result record
(K x [x, y, conf, prob, vis, oks, err] float64 = 952 B per crop at K = 17) -- latency-bound on xGMI, no
bucket or ring tuning applies. Weights are replicated (80 MB in bf16).
"""
from typing import Dict, List, Optional

import torch
import torch.distributed as dist

RECORD_FIELDS = ("x", "y", "conf", "prob", "vis", "oks", "err")


def shard_indices(n: int, rank: int, world: int) -> List[int]:
    """mmengine ``DefaultSampler(shuffle=False, round_up=False)``: strided shards, no padding."""
    return list(range(rank, n, world))


def interleave(gathered: torch.Tensor, n: Optional[int] = None) -> torch.Tensor:
    """(world, per_rank, ...) rank-major -> dataset order, i.e. what ``collect_results`` returns
    (``zip(*part_list)`` then truncate to the dataset size)."""
    world, per = gathered.shape[:2]
    out = gathered.transpose(0, 1).reshape((world * per,) + tuple(gathered.shape[2:]))
    return out if n is None else out[:n]


def pack_records(out: Dict[str, torch.Tensor], into: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Engine outputs -> (B, K, 7) float64 records [x, y, conf, prob, vis, oks, err(raw)]; on a GPU one launch of
    pp_pack_records, straight into ``into`` when given."""
    kp, scores, sc = out["keypoints"], out["scores"], out["scalars"]
    if kp.is_cuda:
        from . import _lib

        kp, scores, sc = kp.contiguous(), scores.contiguous(), sc.contiguous()
        rec = into if into is not None else torch.empty(kp.shape[:2] + (len(RECORD_FIELDS),), dtype=torch.float64, device=kp.device)
        assert rec.is_contiguous() and rec.dtype == torch.float64 and kp.dtype == torch.float64 and scores.dtype == torch.float32 \
            and sc.dtype == torch.float32 and sc.shape[0] == 4
        _lib.call("pp_pack_records", kp.data_ptr(), scores.data_ptr(), sc.data_ptr(), rec.data_ptr(), kp.shape[0] * kp.shape[1],
                  torch.cuda.current_stream(kp.device).cuda_stream)
        return rec
    rec = torch.cat([kp, scores.to(torch.float64)[..., None], sc.to(torch.float64).permute(1, 2, 0)], dim=-1)
    if into is not None:
        into.copy_(rec)
        return into
    return rec


class ResultGather:
    """Per-step: pack the batch's result record, all_gather it over RCCL when world > 1 and start the copy
    into pinned host memory. Buffers are allocated once, for ``batch`` rows per rank.

    A rank may hold FEWER than ``batch`` rows (the tail batch of a ``round_up=False`` shard): its record is padded to
    ``batch`` rows inside, the row counts travel with the records (one extra row), and ``counts`` says how many rows of
    every rank's block are valid. ``__call__`` returns the pinned host tensor (world, batch, K, 7) right after ENQUEUING
    the device-to-host copy: ``wait()`` (or a device synchronize) must come before the host reads it."""

    def __init__(self, batch: int, num_keypoints: int, device, world: int = 1, group=None, force_collective: bool = False):
        self.world, self.group, self.batch = world, group, batch
        self.collective = world > 1 or force_collective  # (force: run the all_gather even on a 1-rank group - tests)
        self.device = torch.device(device)
        K, F = num_keypoints, len(RECORD_FIELDS)
        # one extra row per rank carries the rank's valid-row count, so sizes and records are ONE collective
        self._send = torch.zeros((batch + 1, K, F), dtype=torch.float64, device=self.device)
        self._recv = torch.empty((world, batch + 1, K, F), dtype=torch.float64, device=self.device)
        self.gathered = self._recv[:, :batch]
        pin = self.device.type == "cuda"
        self._host = torch.empty((world, batch + 1, K, F), dtype=torch.float64, pin_memory=pin)
        self.host = self._host[:, :batch]
        self._event = torch.cuda.Event() if pin else None

    def __call__(self, out: Dict[str, torch.Tensor]) -> torch.Tensor:
        n = int(out["keypoints"].shape[0])
        if n > self.batch:
            raise ValueError(f"ResultGather was built for at most {self.batch} rows per rank, got {n}")
        dst = self._send if self.collective else self._recv[0]
        pack_records(out, into=dst[:n])
        if n < self.batch:
            dst[n:self.batch].zero_()  # no stale rows from an earlier, fuller batch
        dst[self.batch].fill_(float(n))
        if self.collective:
            dist.all_gather_into_tensor(self._recv.flatten(0, 1), self._send, group=self.group)
        self._host.copy_(self._recv, non_blocking=True)
        if self._event is not None:
            self._event.record(torch.cuda.current_stream(self.device))
        return self.host

    def wait(self) -> torch.Tensor:
        """Block the host until the last step's device-to-host copy has landed; returns the host tensor."""
        if self._event is not None:
            self._event.synchronize()
        return self.host

    @property
    def counts(self) -> List[int]:
        """Valid rows per rank of the last gathered step (after ``wait()``)."""
        return [int(self._host[r, self.batch, 0, 0].item()) for r in range(self.world)]

    def ordered(self) -> torch.Tensor:
        """Last step's records in dataset order (``collect_results``): rank-major blocks interleaved, padding rows of
        short ranks dropped. Strided shards put the short ranks last, so the valid rows are a prefix."""
        self.wait()
        return interleave(self.host, sum(self.counts))
