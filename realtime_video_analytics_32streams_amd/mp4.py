"""MP4 (ISO BMFF) -> Annex-B elementary stream, for the decode slot of the hot path.

The reference opens every source through ``cv2.VideoCapture(url, cv2.CAP_FFMPEG)`` (video_stream.py:76): FFmpeg
demuxes the container and decodes in software.  Here the container is demuxed on the host (a few hundred bytes of
tables per second of video) and the elementary stream goes to the VCN decoder through rocDecode's parser, which wants
Annex-B byte streams (start-code separated NAL units).  MP4 stores H.264 / H.265 samples in the length-prefixed form
(ISO/IEC 14496-15) with the parameter sets held out of band in ``avcC`` / ``hvcC``, so this module

  * walks the box tree (``moov/trak/mdia/minf/stbl``: ``stsd stts ctts stsc stsz stco|co64 stss``) into a sample table,
  * rewrites every sample's length prefixes into start codes and re-inserts SPS/PPS(/VPS) in front of sync samples,
  * parses the H.264 SPS far enough to report profile / level / coded and cropped size (what the reference logs from
    ``CAP_PROP_FRAME_WIDTH/HEIGHT``, video_stream.py:133-142).

Pure Python on ``bytes`` / ``memoryview``: host-side control data, nothing here touches pixels.
"""
from __future__ import annotations

import mmap
import struct
from dataclasses import dataclass, field
from pathlib import Path
from typing import Dict, Iterator, List, Optional, Tuple, Union

START_CODE = b"\x00\x00\x00\x01"
_CONTAINERS = {b"moov", b"trak", b"mdia", b"minf", b"stbl", b"edts", b"dinf", b"mvex", b"moof", b"traf"}
_VIDEO_ENTRIES = {b"avc1": "h264", b"avc3": "h264", b"hvc1": "hevc", b"hev1": "hevc"}


class Mp4Error(ValueError):
    pass


def iter_boxes(buf, start: int = 0, end: Optional[int] = None) -> Iterator[Tuple[bytes, int, int]]:
    """Yield ``(type, payload_start, payload_end)`` for the boxes laid out in ``buf[start:end]``."""
    end = len(buf) if end is None else end
    pos = start
    while pos + 8 <= end:
        size, typ = struct.unpack_from(">I4s", buf, pos)
        hdr = 8
        if size == 1:
            if pos + 16 > end:
                raise Mp4Error("truncated 64-bit box header")
            size = struct.unpack_from(">Q", buf, pos + 8)[0]
            hdr = 16
        elif size == 0:
            size = end - pos
        if size < hdr or pos + size > end:
            raise Mp4Error(f"box '{typ.decode('latin1')}' at {pos} overruns its parent ({size} bytes)")
        yield typ, pos + hdr, pos + size
        pos += size


@dataclass
class VideoTrack:
    codec: str = ""                       # "h264" | "hevc"
    entry: str = ""                       # sample entry fourcc (avc1 / hvc1 / ...)
    width: int = 0                        # VisualSampleEntry width / height (display size)
    height: int = 0
    timescale: int = 0
    duration: int = 0
    nal_length_size: int = 4
    parameter_sets: List[bytes] = field(default_factory=list)   # raw NAL units (VPS,) SPS, PPS in configuration order
    sizes: List[int] = field(default_factory=list)
    offsets: List[int] = field(default_factory=list)
    dts: List[int] = field(default_factory=list)
    cts_offset: List[int] = field(default_factory=list)
    sync: Optional[set] = None            # 0-based indices of sync samples; None = every sample is one

    @property
    def n_samples(self) -> int:
        return len(self.sizes)

    @property
    def fps(self) -> Tuple[int, int]:
        """Average frame rate as (numerator, denominator): samples per duration in track time units."""
        if not self.n_samples or not self.duration:
            return (0, 1)
        from math import gcd
        n, d = self.n_samples * self.timescale, self.duration
        g = gcd(n, d)
        return (n // g, d // g)


def _fullbox(buf, pos):
    ver = buf[pos]
    return ver, pos + 4


def _parse_stsd(buf, a, b, tr: VideoTrack) -> None:
    _, p = _fullbox(buf, a)
    n = struct.unpack_from(">I", buf, p)[0]
    p += 4
    for _ in range(n):
        size, fmt = struct.unpack_from(">I4s", buf, p)
        if fmt in _VIDEO_ENTRIES and not tr.codec:
            tr.codec, tr.entry = _VIDEO_ENTRIES[fmt], fmt.decode()
            tr.width, tr.height = struct.unpack_from(">HH", buf, p + 8 + 24)
            for typ, ca, cb in iter_boxes(buf, p + 8 + 78, p + size):
                if typ == b"avcC":
                    _parse_avcc(buf, ca, cb, tr)
                elif typ == b"hvcC":
                    _parse_hvcc(buf, ca, cb, tr)
        p += size


def _parse_avcc(buf, a, b, tr: VideoTrack) -> None:
    if b - a < 7:
        raise Mp4Error("avcC too short")
    tr.nal_length_size = (buf[a + 4] & 3) + 1
    p = a + 5
    for field_mask in (0x1F, 0xFF):       # SPS list, then PPS list
        n = buf[p] & field_mask
        p += 1
        for _ in range(n):
            ln = struct.unpack_from(">H", buf, p)[0]
            tr.parameter_sets.append(bytes(buf[p + 2:p + 2 + ln]))
            p += 2 + ln


def _parse_hvcc(buf, a, b, tr: VideoTrack) -> None:
    if b - a < 23:
        raise Mp4Error("hvcC too short")
    tr.nal_length_size = (buf[a + 21] & 3) + 1
    n_arrays = buf[a + 22]
    p = a + 23
    for _ in range(n_arrays):
        n = struct.unpack_from(">H", buf, p + 1)[0]
        p += 3
        for _ in range(n):
            ln = struct.unpack_from(">H", buf, p)[0]
            tr.parameter_sets.append(bytes(buf[p + 2:p + 2 + ln]))
            p += 2 + ln


def _table(buf, a, fmt: str, n_fields: int):
    _, p = _fullbox(buf, a)
    n = struct.unpack_from(">I", buf, p)[0]
    return struct.unpack_from(">" + fmt * n, buf, p + 4) if n else (), n


def _parse_trak(buf, a, b) -> Optional[VideoTrack]:
    tr = VideoTrack()
    handler = None
    stsc = stco = stts = ctts = None
    sample_size, sizes = 0, None

    def walk(x, y):
        nonlocal handler, stsc, stco, stts, ctts, sample_size, sizes
        for typ, ca, cb in iter_boxes(buf, x, y):
            if typ in _CONTAINERS:
                walk(ca, cb)
            elif typ == b"hdlr":
                handler = bytes(buf[ca + 8:ca + 12])
            elif typ == b"mdhd":
                ver, p = _fullbox(buf, ca)
                tr.timescale, tr.duration = struct.unpack_from(">IQ" if ver == 1 else ">II", buf, p + (16 if ver == 1 else 8))
            elif typ == b"stsd":
                _parse_stsd(buf, ca, cb, tr)
            elif typ == b"stts":
                stts = _table(buf, ca, "II", 2)[0]
            elif typ == b"ctts":
                ver = buf[ca]
                ctts = _table(buf, ca, "Ii" if ver == 1 else "II", 2)[0]
            elif typ == b"stsc":
                stsc = _table(buf, ca, "III", 3)[0]
            elif typ == b"stsz":
                _, p = _fullbox(buf, ca)
                sample_size, n = struct.unpack_from(">II", buf, p)
                sizes = list(struct.unpack_from(f">{n}I", buf, p + 8)) if sample_size == 0 else [sample_size] * n
            elif typ == b"stco":
                stco = list(_table(buf, ca, "I", 1)[0])
            elif typ == b"co64":
                stco = list(_table(buf, ca, "Q", 1)[0])
            elif typ == b"stss":
                tr.sync = {v - 1 for v in _table(buf, ca, "I", 1)[0]}
    walk(a, b)
    if handler != b"vide" or not tr.codec:
        return None
    if sizes is None or stco is None or stsc is None:
        raise Mp4Error("video track without a complete sample table (fragmented MP4 is not supported)")
    tr.sizes = sizes
    # chunk map: stsc runs (first_chunk, samples_per_chunk, description) -> per-sample file offsets
    runs = [(stsc[i], stsc[i + 1]) for i in range(0, len(stsc), 3)]
    s = 0
    for ri, (first, per) in enumerate(runs):
        last = runs[ri + 1][0] - 1 if ri + 1 < len(runs) else len(stco)
        for chunk in range(first, last + 1):
            off = stco[chunk - 1]
            for _ in range(per):
                if s >= len(sizes):
                    break
                tr.offsets.append(off)
                off += sizes[s]
                s += 1
    if len(tr.offsets) != len(sizes):
        raise Mp4Error(f"sample table inconsistent: {len(sizes)} sizes, {len(tr.offsets)} offsets")
    t = 0
    for i in range(0, len(stts or ()), 2):
        for _ in range(stts[i]):
            tr.dts.append(t)
            t += stts[i + 1]
    tr.dts = (tr.dts + [t] * len(sizes))[:len(sizes)]
    if ctts:
        for i in range(0, len(ctts), 2):
            tr.cts_offset += [ctts[i + 1]] * ctts[i]
    tr.cts_offset = (tr.cts_offset + [0] * len(sizes))[:len(sizes)]
    return tr


def length_prefixed_to_annexb(sample, nal_length_size: int = 4) -> bytes:
    """One MP4 sample (``[length][NAL]...``, ISO/IEC 14496-15) -> Annex-B (``00 00 00 01 [NAL]...``)."""
    out = bytearray()
    p, n = 0, len(sample)
    while p + nal_length_size <= n:
        ln = int.from_bytes(sample[p:p + nal_length_size], "big")
        p += nal_length_size
        if ln == 0:
            continue
        if p + ln > n:
            raise Mp4Error("NAL unit overruns its sample")
        out += START_CODE
        out += sample[p:p + ln]
        p += ln
    if p != n:
        raise Mp4Error("trailing bytes after the last NAL unit of a sample")
    return bytes(out)


def iter_annexb_nals(data) -> Iterator[bytes]:
    """Split an Annex-B byte stream at its 3- / 4-byte start codes."""
    data = bytes(data)
    pos = data.find(b"\x00\x00\x01")
    while pos >= 0:
        nxt = data.find(b"\x00\x00\x01", pos + 3)
        end = len(data) if nxt < 0 else (nxt - 1 if data[nxt - 1] == 0 else nxt)
        yield data[pos + 3:end]
        pos = nxt


class _Bits:
    """MSB-first bit reader over an RBSP (emulation-prevention bytes removed)."""

    def __init__(self, nal_payload: bytes):
        raw = bytearray()
        z = 0
        for b in nal_payload:
            if z >= 2 and b == 3:
                z = 0
                continue
            raw.append(b)
            z = z + 1 if b == 0 else 0
        self.d, self.p = bytes(raw), 0

    def u(self, n: int) -> int:
        v = 0
        for _ in range(n):
            v = (v << 1) | ((self.d[self.p >> 3] >> (7 - (self.p & 7))) & 1)
            self.p += 1
        return v

    def ue(self) -> int:
        z = 0
        while self.u(1) == 0:
            z += 1
            if z > 32:
                raise Mp4Error("bad exp-golomb code")
        return (1 << z) - 1 + (self.u(z) if z else 0)

    def se(self) -> int:
        k = self.ue()
        return (k + 1) // 2 if k & 1 else -(k // 2)


def parse_h264_sps(nal: bytes) -> Dict[str, int]:
    """ITU-T H.264 7.3.2.1.1 up to the cropping rectangle: profile, level, coded and displayed size."""
    if (nal[0] & 0x1F) != 7:
        raise Mp4Error("not an SPS NAL unit")
    b = _Bits(nal[1:])
    profile = b.u(8); b.u(8); level = b.u(8)
    b.ue()                                         # seq_parameter_set_id
    chroma = 1
    if profile in (100, 110, 122, 244, 44, 83, 86, 118, 128, 138, 139, 134, 135):
        chroma = b.ue()
        if chroma == 3:
            b.u(1)
        b.ue(); b.ue(); b.u(1)                     # bit depths, transform bypass
        if b.u(1):                                 # seq_scaling_matrix_present_flag
            for i in range(8 if chroma != 3 else 12):
                if b.u(1):
                    last = nxt = 8
                    for _ in range(16 if i < 6 else 64):
                        if nxt:
                            nxt = (last + b.se() + 256) % 256
                        last = nxt or last
    b.ue()                                         # log2_max_frame_num_minus4
    poc = b.ue()
    if poc == 0:
        b.ue()
    elif poc == 1:
        b.u(1); b.se(); b.se()
        for _ in range(b.ue()):
            b.se()
    refs = b.ue(); b.u(1)
    mbs_w = b.ue() + 1
    map_h = b.ue() + 1
    frame_mbs_only = b.u(1)
    if not frame_mbs_only:
        b.u(1)
    b.u(1)                                         # direct_8x8_inference_flag
    cl = cr = ct = cb = 0
    if b.u(1):
        cl, cr, ct, cb = b.ue(), b.ue(), b.ue(), b.ue()
    coded_w, coded_h = mbs_w * 16, map_h * 16 * (2 - frame_mbs_only)
    sub_w = 1 if chroma in (0, 3) else 2
    sub_h = (1 if chroma in (0, 2, 3) else 2) * (2 - frame_mbs_only)
    return dict(profile_idc=profile, level_idc=level, chroma_format_idc=chroma, max_num_ref_frames=refs,
                coded_width=coded_w, coded_height=coded_h, width=coded_w - sub_w * (cl + cr),
                height=coded_h - sub_h * (ct + cb), frame_mbs_only=frame_mbs_only)


class Mp4Demuxer:
    """First video track of an MP4 file as Annex-B access units.

    ``Mp4Demuxer(path)`` memory-maps the file (large recordings are not read whole); ``Mp4Demuxer(bytes)`` works on a
    buffer.  ``access_units()`` yields ``(annexb_bytes, pts_10mhz, is_sync)`` in decode order; parameter sets are put
    in front of every sync sample so that decoding can start (or restart after a reconnect) at any of them.
    """

    def __init__(self, source: Union[str, Path, bytes, bytearray, memoryview]):
        self._mm = self._fh = None
        if isinstance(source, (str, Path)):
            self._fh = open(source, "rb")
            self._mm = mmap.mmap(self._fh.fileno(), 0, access=mmap.ACCESS_READ)
            self.buf = memoryview(self._mm)
        else:
            self.buf = memoryview(source)
        self.brand = ""
        self.track: Optional[VideoTrack] = None
        for typ, a, b in iter_boxes(self.buf):
            if typ == b"ftyp":
                self.brand = bytes(self.buf[a:a + 4]).decode("latin1")
            elif typ == b"moov":
                for t2, ca, cb in iter_boxes(self.buf, a, b):
                    if t2 == b"trak" and self.track is None:
                        self.track = _parse_trak(self.buf, ca, cb)
        if self.track is None:
            raise Mp4Error("no H.264 / H.265 video track found")

    def close(self) -> None:
        self.buf.release()
        if self._mm is not None:
            self._mm.close()
            self._fh.close()
            self._mm = self._fh = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def parameter_sets_annexb(self) -> bytes:
        return b"".join(START_CODE + ps for ps in self.track.parameter_sets)

    def sample(self, i: int) -> bytes:
        t = self.track
        return bytes(self.buf[t.offsets[i]:t.offsets[i] + t.sizes[i]])

    def access_unit(self, i: int) -> Tuple[bytes, int, bool]:
        t = self.track
        au = length_prefixed_to_annexb(self.sample(i), t.nal_length_size)
        sync = t.sync is None or i in t.sync
        if sync:
            au = self.parameter_sets_annexb() + au
        pts = (t.dts[i] + t.cts_offset[i]) * 10_000_000 // max(t.timescale, 1)     # rocDecode's default 10 MHz clock
        return au, pts, sync

    def access_units(self, start: int = 0) -> Iterator[Tuple[bytes, int, bool]]:
        for i in range(start, self.track.n_samples):
            yield self.access_unit(i)

    def describe(self) -> Dict[str, object]:
        t = self.track
        d: Dict[str, object] = dict(codec=t.codec, entry=t.entry, width=t.width, height=t.height, samples=t.n_samples,
                                    fps=t.fps, timescale=t.timescale, duration=t.duration, nal_length_size=t.nal_length_size)
        if t.codec == "h264":
            sps = next((ps for ps in t.parameter_sets if (ps[0] & 0x1F) == 7), None)
            if sps is not None:
                d["sps"] = parse_h264_sps(sps)
        return d
