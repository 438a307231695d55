"""Multi-GPU glue: one process per GPU, streams sharded by rank, ONE small collective per tick.

The per-stream stages never exchange data.  The only coupling in the reference is the global track
id counter shared by all streams (tracker.py:47, shared instance pipeline.py:452,502).  To hand out
the same ids when streams live on different GPUs, every rank all-gathers its per-stream new-track
counts (``streams_per_rank`` int32 each -- 128 B for 32 streams) over RCCL/xGMI; each rank then
runs the same exclusive scan in canonical stream order inside ``k4_assign_ids``.  Latency-bound
(a few microseconds of payload), so it rides on the tick's stream right before id assignment.

``backend="nccl"`` is RCCL on ROCm; ``"gloo"`` is used by the CPU tests of this logic.

Failure path.  The reference has one process, so its counter cannot lose a peer; here a dead rank must not leave the others
blocked in the all-gather for ever (first contact with 8 real ranks must not be able to hang a node).  Three layers, all of
them ending in a NON-ZERO EXIT of the surviving process (never a retry, never a re-exec of a process that has touched the GPU
-- a supervisor restarts fresh children):
  * the process group is created with a timeout (``RVA_DIST_TIMEOUT_S``, default 60 s): a host-blocking collective (gloo, the
    rehearsal mode) raises when a peer is gone, and ``IdSync`` turns that into ``fail()``;
  * RCCL collectives are enqueued on the tick's stream and never block the host, so ``PipelinedTicks`` waits for a sharded
    tick with a deadline (``wait_event``: polls the tick's event) and calls ``fail()`` on expiry;
  * torch's own NCCL watchdog (``TORCH_NCCL_ASYNC_ERROR_HANDLING``, on by default) aborts the process when a collective
    exceeds the group timeout -- whichever fires first.
"""
from __future__ import annotations

import datetime
import os
import sys
import time
from typing import List, Optional, Sequence

import torch
import torch.distributed as dist

EXIT_PEER_LOST = 13


def timeout_s() -> float:
    return float(os.environ.get("RVA_DIST_TIMEOUT_S", "60"))


def fail(why: str) -> None:
    """A peer is gone or a collective timed out: say so and leave with a non-zero status, at once.  ``os._exit``: no
    atexit handlers, no destructor that could wait for the dead peer again (``destroy_process_group`` does)."""
    rank = os.environ.get("RANK", "?")
    print(f"[rva dist] rank {rank}: {why}; exiting with status {EXIT_PEER_LOST} (restart the job with fresh processes)",
          file=sys.stderr, flush=True)
    os._exit(EXIT_PEER_LOST)


def wait_event(event, what: str = "tick", deadline_s: Optional[float] = None) -> None:
    """Host wait for a HIP event of a sharded tick, bounded: a collective whose peer died never completes, and
    ``event.synchronize()`` would then block for ever.  Polls; on expiry -> ``fail``."""
    budget = timeout_s() + 5.0 if deadline_s is None else deadline_s
    limit = time.monotonic() + budget
    spins = 0
    while not event.query():
        spins += 1
        if spins > 2000:                      # the common case (a tick takes milliseconds) never sleeps
            time.sleep(2e-4)
        if time.monotonic() > limit:
            fail(f"{what} did not complete within {budget:.0f} s (a peer of the id exchange is gone or stuck)")


def init_from_env(backend: Optional[str] = None) -> tuple[int, int, int]:
    """Read RANK / WORLD_SIZE / LOCAL_RANK (torch.distributed.run contract); returns (rank, world, local).

    ``RVA_SHARE_GPU=1`` is a rehearsal mode for boxes with a single GPU: every rank uses device 0 and the
    process group is gloo (RCCL refuses two ranks on one device); the id exchange is then staged through
    the host.  It exercises the sharded code path, not RCCL."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("RVA_SHARE_GPU") == "1":
        local = 0
        backend = backend or "gloo"
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=timeout_s()))
    return rank, world, local


def shard_streams(n_streams_total: int, rank: int, world: int) -> range:
    """Contiguous slice of the canonical stream order owned by ``rank`` (4 per GPU for 32 streams on 8)."""
    if n_streams_total % world:
        raise ValueError(f"{n_streams_total} streams do not shard evenly over {world} ranks")
    per = n_streams_total // world
    return range(rank * per, (rank + 1) * per)


class IdSync:
    """All-gather of per-stream new-track counts; result is in canonical (rank-major) stream order."""

    def __init__(self, streams_per_rank: int, device: torch.device, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.backend = dist.get_backend(group) if dist.is_initialized() else "none"
        self.per = streams_per_rank
        self.buf = torch.zeros(self.world * streams_per_rank, dtype=torch.int32, device=device)
        self.calls = 0
        self.sample_every = 0          # > 0: bracket every n-th exchange with timing events on the calling stream
        self._events: list = []

    def exchange_us(self) -> List[float]:
        """Device time of the sampled exchanges (call after a synchronize)."""
        return [a.elapsed_time(b) * 1e3 for a, b in self._events]

    def all_gather_counts(self, local_counts: torch.Tensor) -> torch.Tensor:
        assert local_counts.numel() == self.per and local_counts.dtype == torch.int32
        self.calls += 1
        if self.sample_every and self.buf.is_cuda and self.calls % self.sample_every == 0:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            out = self._gather(local_counts)
            b.record()
            self._events.append((a, b))
            return out
        return self._gather(local_counts)

    def _gather(self, local_counts: torch.Tensor) -> torch.Tensor:
        if self.world == 1:
            self.buf.copy_(local_counts)
            return self.buf
        try:
            if self.backend == "gloo" and self.buf.is_cuda:        # rehearsal mode: stage through the host
                host = torch.zeros(self.buf.numel(), dtype=torch.int32)
                dist.all_gather_into_tensor(host, local_counts.cpu().contiguous(), group=self.group)
                self.buf.copy_(host)
            else:
                dist.all_gather_into_tensor(self.buf, local_counts.contiguous(), group=self.group)
        except Exception as exc:  # noqa: BLE001 -- a timed-out or broken collective (peer died, connection reset): no recovery here
            fail(f"id exchange failed ({type(exc).__name__}: {str(exc).splitlines()[0][:200] if str(exc) else ''})")
        return self.buf


def exclusive_id_bases(counts_all: Sequence[int], next_id: int) -> List[int]:
    """Host statement of what k4_assign_ids computes (used by the gloo tests)."""
    out, acc = [], next_id
    for c in counts_all:
        out.append(acc)
        acc += int(c)
    return out
