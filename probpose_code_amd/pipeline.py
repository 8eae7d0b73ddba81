"""Consecutive batches in flight: the device side of the test loop.

The reference's evaluation loop (tools/test.py -> mmengine ``Runner.test()`` [3P]: ``for data_batch in dataloader:
outputs = model.test_step(data_batch); evaluator.process(...)``) handles one batch at a time, and so does a single
hipGraph replay of the engine. One step of the path ends in kernels that cannot fill the chip (the small tower
convolutions, pooling, the latency-bound decode, the record pack) and starts with HBM-bound ones (im2col, patch embed).
``StepPipeline`` keeps ``depth`` steps in flight: every slot has its own HIP stream, workspace, static input buffer,
captured graph and pinned result buffer, and consecutive batches go to consecutive slots, so the tail of batch n shares
the chip with the head of batch n + 1. Nothing is computed differently - each batch runs the same captured launch
sequence as ``ProbPoseEngine.forward_graph`` and its record is bit-identical (tests/test_pipeline.py) - only the stream
the replay is enqueued on changes. Measured on MI355X at bs 64: 2.32 -> 2.13 ms per batch (bf16), 6.65 -> 6.11 (f16x3).

Ordering rules:
  * a slot's work is ordered by its stream; reusing slot j for batch i + depth first waits (on the host) until batch i's
    record has landed in pinned memory - the natural back-pressure of the loop - so a record is never overwritten before
    ``result()`` could have read it, provided results are collected in submission order at most ``depth`` batches late;
  * the RCCL all_gather of a record (world > 1) is issued from the slot's stream; torch's process group serialises
    collectives on its own stream in issue order, which is the same on every rank.
"""
import contextlib
from typing import Dict, List, Optional

import torch

from .dist import ResultGather


class StepPipeline:
    def __init__(self, engine, batch: int, flip_indices, flip_test: bool = True, depth: int = 2, world: int = 1, group=None,
                 use_graph: bool = True, force_collective: bool = False, shift_heatmap: bool = False):
        if depth < 1:
            raise ValueError("depth must be >= 1")
        self.engine, self.batch, self.depth = engine, batch, depth
        self.flip_test, self.flip_indices = flip_test, flip_indices
        self._kw = dict(shift_heatmap=True) if shift_heatmap else {}  # (stub engines of the CPU tests take no such keyword)
        # True: every batch has exactly ``batch`` rows and replays the slot's captured graph (captured here, up front);
        # "full": batches of exactly ``batch`` rows replay the slot's graph (captured when the first one arrives in the slot),
        # smaller ones are launched kernel by kernel; False: always kernel by kernel
        if use_graph not in (True, False, "full"):
            raise ValueError(f"use_graph must be True, False or 'full', got {use_graph!r}")
        self.use_graph = use_graph
        dev = torch.device(engine.device)
        self.device = dev
        self.cuda = dev.type == "cuda"  # (a CPU device only occurs with the stub engine of the gloo tests: slots without streams)
        # depth 1 runs on the caller's current stream (exactly forward_graph + ResultGather); deeper pipelines own their streams
        self.streams: List[Optional[torch.cuda.Stream]] = \
            [None] * depth if (depth == 1 or not self.cuda) else [torch.cuda.Stream(device=dev) for _ in range(depth)]
        self.gathers = [ResultGather(batch, engine.K, dev, world, group, force_collective) for _ in range(depth)]
        self.outs: List[Optional[Dict[str, torch.Tensor]]] = [None] * depth
        self._ticket_of_slot = [-1] * depth
        self._next = 0
        # device-side input buffer of a slot: the captured graph's static input, or (eager launches) a buffer of its own;
        # batches that arrive in HOST memory are copied there on the slot's stream (see submit)
        self._dev_in: List[Optional[torch.Tensor]] = [None] * depth
        self._staging: List[Optional[torch.Tensor]] = [None] * depth      # host batches land here first (copy stream)
        self._staged_ev: List[Optional[torch.cuda.Event]] = [None] * depth
        self._consumed_ev: List[Optional[torch.cuda.Event]] = [None] * depth
        self._copy_stream = torch.cuda.Stream(device=dev) if (self.cuda and depth > 1) else None
        if use_graph is True and self.cuda:
            for j in range(depth):
                self._dev_in[j] = engine.capture(batch, flip_test, flip_indices, slot=j, **self._kw)
        if self.cuda:
            torch.cuda.synchronize(dev)

    def submit(self, crops_u8: torch.Tensor) -> int:
        """Enqueue one batch (uint8 crops, at most ``batch`` rows ... exactly ``batch`` under graph replay); returns its ticket.
        Crops on the device: the caller's current stream is their producer, the slot's stream waits for it. Crops in HOST
        memory (pinned, as a ``pin_memory`` data loader delivers them - pageable memory works but the copy then blocks):
        the host-to-device copy is enqueued at once on a copy stream, ahead of the wait for the slot, so that it runs on the
        copy engine under the kernels still in flight; the tensor must stay untouched until ``result()`` of this ticket has
        returned."""
        t = self._next
        j = t % self.depth
        s = self.streams[j]
        host_src = self.cuda and not crops_u8.is_cuda
        if host_src and s is not None:
            # Host batch: the host-to-device copy goes FIRST, on the copy stream, into the slot's staging buffer - before
            # this call blocks on the slot's previous batch - so it runs on the copy engine under the kernels of the batches
            # still in flight (enqueued behind the slot's own step it would start only when that slot has drained, and the
            # other slot's step would run alone meanwhile: measured 2.43 against 2.07 ms per batch). The staging buffer is
            # free again as soon as the device-to-device copy of the batch before has run (event below).
            n = crops_u8.shape[0]
            if self._staging[j] is None:
                self._staging[j] = torch.empty((self.batch,) + tuple(crops_u8.shape[1:]), dtype=crops_u8.dtype, device=self.device)
                self._staged_ev[j] = torch.cuda.Event()
                self._consumed_ev[j] = torch.cuda.Event()
            else:
                self._copy_stream.wait_event(self._consumed_ev[j])
            with torch.cuda.stream(self._copy_stream):
                self._staging[j][:n].copy_(crops_u8, non_blocking=True)
                self._staged_ev[j].record(self._copy_stream)
        if self._ticket_of_slot[j] >= 0:
            self.gathers[j].wait()  # batch t - depth has been delivered; its buffers may be reused
        if s is not None:
            s.wait_stream(torch.cuda.current_stream(self.device))
        with (torch.cuda.stream(s) if s is not None else contextlib.nullcontext()):
            eng = self.engine
            if host_src:
                n = crops_u8.shape[0]
                if self._dev_in[j] is None:
                    self._dev_in[j] = torch.empty((self.batch,) + tuple(crops_u8.shape[1:]), dtype=crops_u8.dtype, device=self.device)
                if s is not None:
                    s.wait_event(self._staged_ev[j])
                    self._dev_in[j][:n].copy_(self._staging[j][:n], non_blocking=True)  # 9.4 MB inside HBM
                    self._consumed_ev[j].record(s)
                else:  # depth 1: nothing to hide the copy under
                    self._dev_in[j][:n].copy_(crops_u8, non_blocking=True)
                crops_u8 = self._dev_in[j][:n]
            if self.use_graph is True or (self.use_graph == "full" and self.cuda and crops_u8.shape[0] == self.batch
                                          and crops_u8.dtype == torch.uint8):
                out = eng.forward_graph(crops_u8, self.flip_test, self.flip_indices, slot=j, **self._kw)
            else:
                out = eng.forward(crops_u8, self.flip_test, self.flip_indices, slot=j, **self._kw)
            self.gathers[j](out)
        if s is not None and crops_u8.is_cuda:
            crops_u8.record_stream(s)
        self.outs[j] = out
        self._ticket_of_slot[j] = t
        self._next = t + 1
        return t

    def result(self, ticket: int) -> torch.Tensor:
        """Pinned host records (world, batch, K, 7) of a submitted batch; blocks until they have landed. Valid until the
        slot is reused, i.e. until ``depth`` more batches have been submitted."""
        j = ticket % self.depth
        if self._ticket_of_slot[j] != ticket:
            raise RuntimeError(f"batch {ticket} is no longer (or not yet) in the pipeline: slot {j} holds batch {self._ticket_of_slot[j]}")
        return self.gathers[j].wait()

    def gather_of(self, ticket: int) -> ResultGather:
        j = ticket % self.depth
        if self._ticket_of_slot[j] != ticket:
            raise RuntimeError(f"batch {ticket} is not in the pipeline")
        return self.gathers[j]

    def device_outputs(self, ticket: int) -> Dict[str, torch.Tensor]:
        """The engine's device-side outputs of a batch (views into the slot's workspace; synchronise first)."""
        j = ticket % self.depth
        if self._ticket_of_slot[j] != ticket:
            raise RuntimeError(f"batch {ticket} is not in the pipeline")
        return self.outs[j]

    def drain(self):
        """Host-wait for everything submitted so far."""
        for j in range(self.depth):
            if self._ticket_of_slot[j] >= 0:
                self.gathers[j].wait()
