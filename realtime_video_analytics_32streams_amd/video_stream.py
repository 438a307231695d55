"""Capture side of the hot path: frames that are already in HBM.

Mirrors the reference's capture surface (video_stream.py:26-243): ``FramePacket`` has the same four
fields, a stream object is an async context manager with ``open() / close() / frames()``, frame ids
start at 0 per open (video_stream.py:85,238) and pacing sleeps ``1/target_fps`` AFTER the consumer
returns (video_stream.py:242-243).  What differs is where the pixels live: ``packet.frame`` is an
:class:`~.ops.Nv12Surface` (device planes, as a hardware decoder produces them) instead of a host
BGR ndarray; detectors only ever use ``.frame``, ``.frame_id`` and ``.stream.name``
(detector.py:331-332).

Sources:
  * :class:`SyntheticNv12Stream` -- deterministic NV12 frames generated on the device (SURVEY.md
    8(d) generator); the always-available source for tests and the bench.
  * :class:`RocDecodeStream` -- the decode slot.  librocdecode is absent from this image; the class
    probes for it (``rva_decode_available``) and raises ``RuntimeError`` from ``open()`` when it is
    missing, exactly as the reference raises when a stream cannot be opened (video_stream.py:78-79).
    Decode throughput is therefore reported as "not measured", never substituted.
"""
from __future__ import annotations

import asyncio
import ctypes
import time
from dataclasses import dataclass
from typing import Any, AsyncGenerator, Optional

import numpy as np
import torch

from . import _native as N
from . import synth
from .config import StreamConfig
from .ops import Nv12Surface


@dataclass(slots=True)
class FramePacket:
    """Container for a video frame and associated metadata (video_stream.py:26-33)."""

    stream: StreamConfig
    frame: Any          # Nv12Surface (device) or uint8 BGR ndarray / tensor [h, w, 3]
    frame_id: int
    timestamp: float


class _BaseStream:
    def __init__(self, stream_config: StreamConfig):
        self.config = stream_config
        self._frame_id = 0
        self._opened = False

    async def __aenter__(self):
        await self.open()
        return self

    async def __aexit__(self, exc_type, exc, tb):
        await self.close()

    async def open(self) -> None:
        self.open_sync()
        if self.config.warmup_seconds > 0:
            await asyncio.sleep(self.config.warmup_seconds)

    async def close(self) -> None:
        self._opened = False

    def open_sync(self) -> None:
        self._frame_id = 0
        self._opened = True

    def next_surface(self) -> Optional[Nv12Surface]:
        raise NotImplementedError

    def next_packet(self) -> Optional[FramePacket]:
        """Synchronous pull used by the batched tick loop."""
        if not self._opened:
            self.open_sync()
        surf = self.next_surface()
        if surf is None:
            return None
        pkt = FramePacket(stream=self.config, frame=surf, frame_id=self._frame_id, timestamp=time.time())
        self._frame_id += 1
        return pkt

    async def frames(self) -> AsyncGenerator[FramePacket, None]:
        if not self._opened:
            await self.open()
        while True:
            pkt = self.next_packet()
            if pkt is None:
                break
            yield pkt
            if self.config.target_fps:
                await asyncio.sleep(max(0.0, 1.0 / self.config.target_fps))


class SyntheticNv12Stream(_BaseStream):
    """Deterministic 1080p-style NV12 source held in HBM.

    ``n_unique`` distinct frames are generated once with :func:`synth.make_nv12`
    (seed = SEED_BASE + 1000 * stream_index, moving rectangles advance with the tick) and uploaded;
    the stream then cycles through them, so a steady-state run touches real, changing pixel data
    without paying host generation inside the timed region.
    """

    def __init__(self, stream_config: StreamConfig, index: int = 0, width: int = 1920, height: int = 1080,
                 pitch: Optional[int] = None, n_unique: int = 4, n_frames: Optional[int] = None, device="cuda"):
        super().__init__(stream_config)
        self.index, self.width, self.height = index, width, height
        self.pitch = pitch or ((width + 255) // 256) * 256
        self.n_unique, self.n_frames, self.device = n_unique, n_frames, device
        self._ring = []

    def open_sync(self) -> None:
        super().open_sync()
        if not self._ring:
            seed = synth.SEED_BASE + 1000 * self.index
            for t in range(self.n_unique):
                y, uv = synth.make_nv12(seed, self.width, self.height, self.pitch, tick=t)
                self._ring.append(Nv12Surface.from_numpy(y, uv, self.width, self.height, self.device))

    def next_surface(self) -> Optional[Nv12Surface]:
        if self.n_frames is not None and self._frame_id >= self.n_frames:
            return None
        return self._ring[self._frame_id % self.n_unique]


def rocdecode_status() -> str:
    buf = ctypes.create_string_buffer(256)
    rc = N.lib().rva_decode_available(buf, 256)
    return ("available: " if rc == N.RVA_OK else "unavailable: ") + buf.value.decode()


class RocDecodeStream(_BaseStream):
    """H.264/H.265 -> NV12 in HBM through rocDecode (VCN).  Probe-only in this round: the library is
    not present in the image, so ``open()`` raises; the slot and its contract are fixed here."""

    def open_sync(self) -> None:
        status = rocdecode_status()
        if not status.startswith("available"):
            raise RuntimeError(f"Unable to open stream {self.config.name}: rocDecode {status}")
        raise RuntimeError(f"Unable to open stream {self.config.name}: rocDecode session setup not implemented "
                           f"in this round ({status})")


def open_stream(cfg: StreamConfig, index: int = 0, **kw) -> _BaseStream:
    """``synthetic://WxH`` urls select the generator; anything else needs rocDecode."""
    if cfg.url.startswith("synthetic://"):
        spec = cfg.url[len("synthetic://"):] or "1920x1080"
        w, h = (int(v) for v in spec.lower().split("x"))
        return SyntheticNv12Stream(cfg, index=index, width=w, height=h, **kw)
    return RocDecodeStream(cfg)
