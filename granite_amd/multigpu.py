"""Row-band tiling of one frame across ranks: the collectives behind gra_set_exchange_callback.

The executor (C++) decides WHAT each rank computes (StripPlan) and WHERE bands must meet.  On the GPU box the bands meet
through the executor's own RCCL all-gather (csrc/host/collective.cpp, gra_comm_init); this module holds the gloo form of
the same in-place all-gather for the multi-process CPU test and the one-GPU emulation used by the GPU parity test.
Nothing here touches pixel values.
"""
from __future__ import annotations

import threading
from typing import Callable, Dict, Tuple

import numpy as np


def all_gather_chunks_inplace(full, rank: int, chunk_elems: int, group=None):
    """In-place all-gather on a flat torch tensor `full` of world * chunk_elems elements: rank r's chunk lives at
    [r * chunk_elems, (r + 1) * chunk_elems)."""
    import torch.distributed as dist
    mine = full[rank * chunk_elems:(rank + 1) * chunk_elems]
    dist.all_gather_into_tensor(full, mine, group=group)


class LocalExchange:
    """The same meeting points for N executor instances living in ONE process on ONE GPU (one Python thread per rank):
    used by the GPU parity test that emulates an N-rank frame on the single device available to it.  copy(dst, src, nbytes,
    stream) enqueues a device-to-device copy, sync(stream) waits for a stream."""

    def __init__(self, world: int, copy: Callable[[int, int, int, int], None], sync: Callable[[int], None]):
        self.world = world
        self.copy, self.sync = copy, sync
        self.barrier = threading.Barrier(world)
        self.posted: Dict[int, Tuple[int, int]] = {}
        self.lock = threading.Lock()

    def for_rank(self, rank: int):
        def exchange(tag: str, ptr: int, chunk_bytes: int, ranks: int, stream: int):
            assert ranks == self.world
            self.sync(stream)  # my chunk is complete
            with self.lock:
                self.posted[rank] = (ptr, chunk_bytes)
            self.barrier.wait()
            for other in range(self.world):
                if other != rank:
                    src_ptr, nbytes = self.posted[other]
                    assert nbytes == chunk_bytes
                    self.copy(ptr + other * chunk_bytes, src_ptr + other * chunk_bytes, chunk_bytes, stream)
            self.sync(stream)
            self.barrier.wait()  # nobody moves on (and rewrites its chunk) while someone still reads it
        return exchange


def weak_scaled_frame(world: int, base=(3840, 2160)) -> Tuple[int, int]:
    """bench.py's frame for `world` ranks: one base frame's worth of pixels per rank, grown alternately in height and
    width (1: 3840x2160, 2: 3840x4320, 4: 7680x4320 = BASELINE config 5's frame, 8: 7680x8640)."""
    w, h = base
    n, grow_h = world, True
    while n > 1:
        if n % 2:
            raise ValueError("world size must be a power of two")
        if grow_h:
            h *= 2
        else:
            w *= 2
        grow_h = not grow_h
        n //= 2
    return w, h


def plan_numpy(index: int, count: int, width: int, height: int, post_aa: int = 0, pre_aa: int = 0, taa_history_reach_rows: int = 0) -> dict:
    """StripPlan::build through the C ABI without a GPU (dry application); post_aa / pre_aa = app.POST_AA_* values."""
    from . import app as gapp
    a = gapp.Application(width, height, device=-1, strip_index=index, strip_count=count, post_aa=post_aa, pre_aa=pre_aa,
                         taa_history_reach_rows=taa_history_reach_rows)
    try:
        return a.strip_plan()
    finally:
        a.close()
