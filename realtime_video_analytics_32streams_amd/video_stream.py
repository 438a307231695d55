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
  * :class:`RocDecodeStream` -- H.264 / H.265 files (MP4 or raw Annex-B) through the rocDecode session of
    ``csrc/rva_decode.hip``; surfaces alias decoder memory.  librocdecode is absent from this project's
    images, so ``open()`` raises ``RuntimeError`` there -- exactly as the reference raises when a stream
    cannot be opened (video_stream.py:78-79) -- and decode throughput is reported as "not measured",
    never substituted.
"""
from __future__ import annotations

import asyncio
import ctypes
import logging
import time
from dataclasses import dataclass
from typing import Any, AsyncGenerator, Optional

import numpy as np
import torch

from . import _native as N
from . import synth
from .config import StreamConfig
from .ops import Nv12Surface


LOGGER = logging.getLogger(__name__)


@dataclass(slots=True)
class FramePacket:
    """Container for a video frame and associated metadata (video_stream.py:26-33)."""

    stream: StreamConfig
    frame: Any          # Nv12Surface (device) or uint8 BGR ndarray / tensor [h, w, 3]
    frame_id: int
    timestamp: float


@dataclass
class _Misses:
    """Failed reads of a capture loop.  ``run`` counts them since the last delivered frame and is what ``max_retries``
    is held against (a reopen does not clear it, video_stream.py:177,187); ``streak`` counts them since the last frame
    OR the last successful reopen and sizes the back-off / triggers the reopen (:178,203-206,213-220)."""

    run: int = 0
    streak: int = 0

    def frame(self) -> None:
        self.run = self.streak = 0

    def miss(self) -> None:
        self.run += 1
        self.streak += 1

    def pause(self, base: float) -> float:
        return min(base * (1 + 0.5 * self.streak), 30.0)


class _BaseStream:
    """Capture plugin surface of the reference's ``VideoStream`` (video_stream.py:47-243) over a device-frame source.

    Subclasses supply three synchronous primitives -- ``open_sync()`` (raise ``RuntimeError`` when the source cannot be
    opened, like video_stream.py:78-79), ``next_surface()`` (one frame, ``None`` = the read failed) and ``close_sync()`` --
    and inherit the reference's policy around them: frame ids restart at 0 on every (re)open (:85), a failed read
    backs off ``min(reconnect_backoff * (1 + 0.5 * streak), 30)`` seconds (:203-206), the third failure in a streak closes
    and reopens the source (:213-224; a failed reopen leaves the stream closed and the loop idling one
    ``reconnect_backoff`` per turn, :169-171), ``max_retries`` failed reads in a row end the generator (:187-197), and
    pacing sleeps ``1 / target_fps`` AFTER the consumer has taken the frame (:242-243).  The observable sequence (opens,
    closes, every sleep, frame ids) is pinned by tests/golden/capture_loop.json, recorded from the reference's own
    ``frames()``."""

    def __init__(self, stream_config: StreamConfig):
        self.config = stream_config
        self._frame_id = 0
        self._opened = False
        self._misses = _Misses()
        self._sleep = asyncio.sleep          # the one suspension point besides the read: injectable for tests

    async def __aenter__(self):
        await self.open()
        return self

    async def __aexit__(self, exc_type, exc, tb):
        await self.close()

    async def open(self) -> None:
        if self._opened:
            return
        self.open_sync()
        self._misses.streak = 0
        if self.config.warmup_seconds > 0:
            await self._sleep(self.config.warmup_seconds)

    async def close(self) -> None:
        if self._opened:
            await asyncio.to_thread(self.close_sync)

    def open_sync(self) -> None:
        self._frame_id = 0
        self._opened = True

    def close_sync(self) -> None:
        self._opened = False

    def next_surface(self) -> Optional[Nv12Surface]:
        raise NotImplementedError

    def _packet(self, surface) -> FramePacket:
        pkt = FramePacket(stream=self.config, frame=surface, frame_id=self._frame_id, timestamp=time.time())
        self._frame_id += 1
        return pkt

    def next_packet(self) -> Optional[FramePacket]:
        """Synchronous pull used by the batched tick loop (no retry policy: ``None`` masks the stream out of the tick)."""
        if not self._opened:
            self.open_sync()
        surf = self.next_surface()
        return None if surf is None else self._packet(surf)

    async def _reopen(self) -> None:
        """Close and open again; a source that does not come back stays closed (the loop then idles until it does)."""
        LOGGER.info("stream '%s': %d reads failed in a row, reopening the source", self.config.name, self._misses.streak)
        await self.close()
        try:
            await self.open()
        except Exception as exc:  # noqa: BLE001
            LOGGER.error("stream '%s': reopen failed (%s)", self.config.name, exc)

    async def frames(self) -> AsyncGenerator[FramePacket, None]:
        cfg = self.config
        if not self._opened:
            await self.open()
        misses = self._misses = _Misses()
        gap = max(0.0, 1.0 / cfg.target_fps) if cfg.target_fps else None
        while True:
            if not self._opened:                      # the last reopen failed
                await self._sleep(cfg.reconnect_backoff)
                continue
            surface = await asyncio.to_thread(self.next_surface)
            if surface is not None:
                misses.frame()
                yield self._packet(surface)
                if gap is not None:
                    await self._sleep(gap)            # pacing comes after the consumer has had the frame
                continue
            misses.miss()
            LOGGER.warning("stream '%s': no frame (%d in a row, %d since the last reopen)", cfg.name, misses.run, misses.streak)
            if cfg.max_retries is not None and misses.run >= cfg.max_retries:
                LOGGER.error("stream '%s': %d reads failed in a row, giving up", cfg.name, misses.run)
                return
            pause = misses.pause(cfg.reconnect_backoff)     # sized before a successful reopen clears the streak
            if misses.streak >= 3:
                await self._reopen()
            await self._sleep(pause)


class SyntheticNv12Stream(_BaseStream):
    """Deterministic 1080p-style NV12 source held in HBM.

    ``n_unique`` distinct frames are generated once with :func:`synth.make_nv12`
    (seed = SEED_BASE + 1000 * stream_index, moving rectangles advance with the tick) and uploaded;
    the stream then cycles through them, so a steady-state run touches real, changing pixel data
    without paying host generation inside the timed region.

    ``ring_frames`` > ``n_unique`` adds frames derived ON THE DEVICE from the generated ones (frame j = frame j mod
    n_unique shifted horizontally by 16 * (j // n_unique) pixels, wrapping; valid NV12, own memory): a decoder hands the
    pipeline new bytes every tick, and a ring whose frames of one tick (streams x 3.3 MB at 1080p) come round again only
    after more than 2 x 256 MiB of other surfaces have been read cannot be served from the Infinity Cache.  ``bench.py``
    sizes its rings that way; tests that compare against the oracle keep ``ring_frames`` unset (host-generated frames only).
    """

    def __init__(self, stream_config: StreamConfig, index: int = 0, width: int = 1920, height: int = 1080,
                 pitch: Optional[int] = None, n_unique: int = 4, n_frames: Optional[int] = None, device="cuda",
                 ring_frames: Optional[int] = None):
        super().__init__(stream_config)
        self.index, self.width, self.height = index, width, height
        self.pitch = pitch or ((width + 255) // 256) * 256
        self.n_unique, self.n_frames, self.device = n_unique, n_frames, device
        self.ring_frames = max(int(ring_frames or n_unique), n_unique)
        self._ring = []

    def open_sync(self) -> None:
        super().open_sync()
        if not self._ring:
            seed = synth.SEED_BASE + 1000 * self.index
            for t in range(self.n_unique):
                y, uv = synth.make_nv12(seed, self.width, self.height, self.pitch, tick=t)
                self._ring.append(Nv12Surface.from_numpy(y, uv, self.width, self.height, self.device))
            for j in range(self.n_unique, self.ring_frames):
                base, sh = self._ring[j % self.n_unique], 16 * (j // self.n_unique)
                y, uv = base.y.clone(), base.uv.clone()
                y[:, :self.width] = torch.roll(base.y[:, :self.width], sh, 1)
                uv[:, :self.width] = torch.roll(base.uv[:, :self.width], sh, 1)      # interleaved UV: an even byte shift keeps U / V in place
                self._ring.append(Nv12Surface(y, uv, self.width, self.height))

    def next_surface(self) -> Optional[Nv12Surface]:
        if self.n_frames is not None and self._frame_id >= self.n_frames:
            return None
        return self._ring[self._frame_id % len(self._ring)]


def rocdecode_status() -> str:
    buf = ctypes.create_string_buffer(256)
    rc = N.lib().rva_decode_available(buf, 256)
    return ("available: " if rc == N.RVA_OK else "unavailable: ") + buf.value.decode()


def _device_u8(ptr: int, rows: int, pitch: int, device: torch.device) -> torch.Tensor:
    """Zero-copy uint8 [rows, pitch] view of decoder-owned device memory."""
    class _Holder:
        pass
    h = _Holder()
    h.__cuda_array_interface__ = {"shape": (rows, pitch), "typestr": "|u1", "data": (int(ptr), False), "version": 3}
    return torch.as_tensor(h, device=device)


class RocDecodeStream(_BaseStream):
    """H.264 / H.265 -> NV12 in HBM through rocDecode (VCN): the capture plugin for real bitstreams.

    Sources: a local ``.mp4`` / ``.mov`` / ``.m4v`` file (demuxed by :mod:`.mp4`) or a raw Annex-B file (``.h264 .264
    .h265 .265 .hevc``, split at access-unit delimiters / first-slice NAL units).  ``open_sync`` raises ``RuntimeError``
    -- as the reference does when a stream cannot be opened, video_stream.py:78-79 -- when librocdecode is absent, the
    file is missing or unreadable, or the URL is a network source (RTSP demuxing is FFmpeg's job in the reference and is
    not rebuilt here).  ``next_surface`` feeds access units until the decoder has a displayable picture; the surfaces
    it returns alias decoder memory and are recycled ``hold`` frames later (enough for two ticks in flight).  At the end
    of the file ``next_surface`` returns ``None``: a failed read, which the inherited capture loop turns into
    back-off and, after three of them, a reopen from frame 0 -- what cv2.VideoCapture does to the reference on a file.
    """

    RAW_SUFFIX = {".h264": N.RVA_CODEC_H264, ".264": N.RVA_CODEC_H264, ".h265": N.RVA_CODEC_HEVC, ".265": N.RVA_CODEC_HEVC,
                  ".hevc": N.RVA_CODEC_HEVC}

    def __init__(self, stream_config: StreamConfig, device: Optional[int] = None, hold: int = 4, num_surfaces: int = 0):
        super().__init__(stream_config)
        self.device_index, self.hold, self.num_surfaces = device, hold, num_surfaces
        self._dec = None
        self._units = None
        self._demux = None
        self._held: list = []
        self._eos = False
        self.info: dict = {}

    # -- source -------------------------------------------------------------------------------------------
    def _open_source(self):
        from pathlib import Path
        from . import mp4
        url = self.config.url
        path = Path(url[7:] if url.startswith("file://") else url)
        if "://" in url and not url.startswith("file://"):
            raise RuntimeError(f"Unable to open stream {self.config.name}: network sources need an RTSP/RTP demuxer, which this "
                               "library does not have (local MP4 and Annex-B files only)")
        if not path.is_file():
            raise RuntimeError(f"Unable to open stream {self.config.name}: {path} does not exist")
        suffix = path.suffix.lower()
        if suffix in self.RAW_SUFFIX:
            data = path.read_bytes()
            self.info = dict(codec="h264" if self.RAW_SUFFIX[suffix] == N.RVA_CODEC_H264 else "hevc", container="annexb")
            return self.RAW_SUFFIX[suffix], _annexb_access_units(data, self.RAW_SUFFIX[suffix])
        try:
            self._demux = mp4.Mp4Demuxer(path)
        except (mp4.Mp4Error, OSError) as exc:
            raise RuntimeError(f"Unable to open stream {self.config.name}: {exc}") from exc
        self.info = self._demux.describe()
        codec = N.RVA_CODEC_H264 if self._demux.track.codec == "h264" else N.RVA_CODEC_HEVC
        return codec, self._demux.access_units()

    def open_sync(self) -> None:
        status = rocdecode_status()
        if not status.startswith("available"):
            raise RuntimeError(f"Unable to open stream {self.config.name}: rocDecode {status}")
        from . import ops
        codec, self._units = self._open_source()
        self._ctx = ops.context(self.device_index)
        h = ctypes.c_void_p()
        rc = N.lib().rva_decoder_create(self._ctx.handle, codec, self.num_surfaces, ctypes.byref(h))
        if rc != N.RVA_OK:
            msg = N.lib().rva_last_error(self._ctx.handle)
            raise RuntimeError(f"Unable to open stream {self.config.name}: {msg.decode() if msg else rc}")
        self._dec = h
        self._eos = False
        super().open_sync()

    def close_sync(self) -> None:
        if self._dec is not None:
            N.lib().rva_decoder_destroy(self._dec)
            self._dec = None
        if self._demux is not None:
            self._demux.close()
            self._demux = None
        self._held.clear()
        self._units = None
        super().close_sync()

    # -- frames -------------------------------------------------------------------------------------------
    def _pop(self) -> Optional[Nv12Surface]:
        L = N.lib()
        y, uv = ctypes.c_void_p(), ctypes.c_void_p()
        pitch, w, h, pic = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        pts = ctypes.c_int64()
        self._ctx.check(L.rva_decoder_next_frame(self._dec, ctypes.byref(y), ctypes.byref(uv), ctypes.byref(pitch), ctypes.byref(w),
                                                 ctypes.byref(h), ctypes.byref(pts), ctypes.byref(pic)), "rva_decoder_next_frame")
        if pic.value < 0:
            return None
        dev = torch.device("cuda", self._ctx.device)
        surf = Nv12Surface(_device_u8(y.value, h.value, pitch.value, dev), _device_u8(uv.value, h.value // 2, pitch.value, dev),
                           w.value, h.value)
        self._held.append(pic.value)
        while len(self._held) > self.hold:                           # oldest mapped picture goes back to the decoder
            self._ctx.check(L.rva_decoder_release(self._dec, self._held.pop(0)), "rva_decoder_release")
        return surf

    def next_surface(self) -> Optional[Nv12Surface]:
        if self._dec is None:
            return None
        L = N.lib()
        while True:
            surf = self._pop()
            if surf is not None or self._eos:
                return surf
            unit = next(self._units, None)
            if unit is None:
                self._eos = True
                self._ctx.check(L.rva_decoder_feed(self._dec, None, 0, 0, 1), "rva_decoder_feed")         # flush
            else:
                au, pts, _ = unit
                self._ctx.check(L.rva_decoder_feed(self._dec, au, len(au), pts, 0), "rva_decoder_feed")


def _annexb_access_units(data: bytes, codec: int):
    """Group the NAL units of a raw Annex-B file into access units: a new one starts at an access-unit delimiter, a
    parameter set, or a slice NAL unit whose first_mb / first_slice_segment flag says "first slice of a picture"."""
    from . import mp4
    cur, has_slice, n = [], False, 0
    for nal in mp4.iter_annexb_nals(data):
        if not nal:
            continue
        if codec == N.RVA_CODEC_H264:
            t = nal[0] & 0x1F
            is_slice = t in (1, 5)
            first = is_slice and len(nal) > 1 and (nal[1] & 0x80) != 0                  # first_mb_in_slice == 0 -> ue(0) = '1'
            boundary = t in (6, 7, 8, 9)
        else:
            t = (nal[0] >> 1) & 0x3F
            is_slice = t <= 21
            first = is_slice and len(nal) > 2 and (nal[2] & 0x80) != 0                  # first_slice_segment_in_pic_flag
            boundary = t in (32, 33, 34, 35, 39)
        if has_slice and (first or boundary):
            yield b"".join(mp4.START_CODE + x for x in cur), n * 333_667, True
            cur, has_slice = [], False
            n += 1
        cur.append(nal)
        has_slice = has_slice or is_slice
    if cur:
        yield b"".join(mp4.START_CODE + x for x in cur), n * 333_667, True


def open_stream(cfg: StreamConfig, index: int = 0, **kw) -> _BaseStream:
    """``synthetic://WxH`` urls select the generator; anything else is a bitstream for rocDecode."""
    if cfg.url.startswith("synthetic://"):
        spec = cfg.url[len("synthetic://"):] or "1920x1080"
        w, h = (int(v) for v in spec.lower().split("x"))
        return SyntheticNv12Stream(cfg, index=index, width=w, height=h, **kw)
    return RocDecodeStream(cfg)
