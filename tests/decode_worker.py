"""Runs csrc/rva_decode.hip + RocDecodeStream end to end against the rocDecode TEST DOUBLE (tests/mock_rocdecode/): started
by tests/test_gpu_decode.py in a process of its own with RVA_ROCDECODE_LIB pointing at the mock (the library is probed once
per process).  The mock's "decoded" pictures are a closed-form pattern restated here, so every check is against an
independent expectation: display-area crop, display order and the one-picture display delay, hold / release of mapped
pictures, end-of-stream flush, a change of picture size mid-stream, reopen after failed reads, error strings, and K1 on the
decoder's surfaces against the oracle.  It pins nothing about VCN pixels (D1 stays "partial")."""
from __future__ import annotations

import asyncio
import sys
import tempfile
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

SC = b"\x00\x00\x00\x01"


def num3(v: int) -> bytes:
    return bytes([0x40 | (v >> 12) & 0x3F, 0x40 | (v >> 6) & 0x3F, 0x40 | v & 0x3F])


def sps(cw, ch, left, top, right, bottom) -> bytes:
    return SC + b"\x67" + b"".join(num3(v) for v in (cw, ch, left, top, right, bottom))


def slice_nal(frame: int, idr: bool) -> bytes:
    return SC + (b"\x65" if idr else b"\x41") + b"\x80" + num3(frame)


def expected_surface(frame: int, left: int, top: int, w: int, h: int):
    """Display area of picture `frame` as the mock paints it (tests/mock_rocdecode/mock_rocdecode.cpp header)."""
    x = np.arange(left, left + w)[None, :]
    y = np.arange(top, top + h)[:, None]
    Y = ((3 * x + 5 * y + 7 * frame) & 0xFF).astype(np.uint8)
    cx = np.arange(left // 2, left // 2 + w // 2)[None, :]
    cy = np.arange(top // 2, top // 2 + h // 2)[:, None]
    uv = np.empty((h // 2, w), np.uint8)
    uv[:, 0::2] = (cx + 3 * cy + 11 * frame) & 0xFF
    uv[:, 1::2] = (5 * cx + cy + 13 * frame) & 0xFF
    return Y, uv


def main():
    from oracle import oracle as orc
    from realtime_video_analytics_32streams_amd import _native as N
    from realtime_video_analytics_32streams_amd import ops
    from realtime_video_analytics_32streams_amd.config import StreamConfig
    from realtime_video_analytics_32streams_amd.video_stream import RocDecodeStream, rocdecode_status

    st = rocdecode_status()
    assert st.startswith("available") and "libmockrocdecode" in st, st
    tmp = Path(tempfile.mkdtemp())

    # ---- stream A: 1920x1088 coded, display area offset (8, 4) 1904 x 1072; then the size changes to 1280x720 mid-stream
    A = dict(cw=1920, ch=1088, left=8, top=4, w=1904, h=1072)
    B = dict(cw=1280, ch=720, left=0, top=0, w=1280, h=720)
    data = sps(A["cw"], A["ch"], A["left"], A["top"], A["left"] + A["w"], A["top"] + A["h"])
    nA, nB = 9, 5
    for f in range(nA):
        data += slice_nal(f, f == 0)
    data += sps(B["cw"], B["ch"], 0, 0, B["w"], B["h"])
    for f in range(nB):
        data += slice_nal(100 + f, f == 0)
    path = tmp / "toy.h264"
    path.write_bytes(data)

    src = RocDecodeStream(StreamConfig(name="mock", url=str(path), warmup_seconds=0.0, reconnect_backoff=0.01), hold=4)
    src.open_sync()
    got = []
    while True:
        pkt = src.next_packet()
        if pkt is None:
            break
        s = pkt.frame
        geo = A if len(got) < nA else B
        frame = len(got) if len(got) < nA else 100 + len(got) - nA
        assert (s.width, s.height) == (geo["w"], geo["h"]), (len(got), s.width, s.height)
        wantY, wantUV = expected_surface(frame, geo["left"], geo["top"], geo["w"], geo["h"])
        assert np.array_equal(s.y.cpu().numpy()[:geo["h"], :geo["w"]], wantY), ("Y", len(got))
        assert np.array_equal(s.uv.cpu().numpy()[:geo["h"] // 2, :geo["w"]], wantUV), ("UV", len(got))
        assert pkt.frame_id == len(got)
        # K1 straight off the decoder's surface == the oracle's pre-process of the same pixels (bit-exact)
        if len(got) in (0, 3, nA - 1, nA, nA + nB - 1):
            out, meta = ops.preprocess_nv12([s], (640, 640), half=True)
            want, m = orc.preprocess_nv12(np.ascontiguousarray(wantY), np.ascontiguousarray(wantUV), geo["w"], geo["h"], 640, 640, True)
            assert np.array_equal(out.cpu().numpy()[0].view(np.uint16), want.view(np.uint16)) and meta.as_meta() == m, len(got)
        got.append(frame)
        assert len(src._held) <= src.hold                       # older mapped pictures went back to the decoder
    # every picture came out, in display order, the last one of each sequence through a flush (new sequence / end of stream)
    assert got == list(range(nA)) + [100 + f for f in range(nB)], got
    assert src._eos and src.next_packet() is None               # end of file: a failed read from now on
    src.close_sync()

    # ---- the inherited capture loop on top: three failed reads -> reopen from frame 0 (video_stream.py:213-224, :85)
    short = tmp / "short.h264"
    short.write_bytes(sps(640, 368, 0, 0, 640, 360) + b"".join(slice_nal(f, f == 0) for f in range(3)))
    loop = RocDecodeStream(StreamConfig(name="loop", url=str(short), warmup_seconds=0.0, reconnect_backoff=0.001), hold=2)

    async def drive():
        ids, sizes = [], []
        sleeps = []

        async def fake_sleep(t):
            sleeps.append(t)
        loop._sleep = fake_sleep
        async for pkt in loop.frames():
            ids.append(pkt.frame_id)
            sizes.append((pkt.frame.width, pkt.frame.height))
            if len(ids) == 8:
                break
        await loop.close()
        return ids, sizes, sleeps
    ids, sizes, sleeps = asyncio.run(drive())
    assert ids == [0, 1, 2, 0, 1, 2, 0, 1], ids                  # frame ids restart with every reopen
    assert set(sizes) == {(640, 360)}
    assert len(sleeps) >= 6                                     # 3 back-offs per end of file

    # ---- holding every surface starves the decoder: the library's error name reaches the caller
    many = tmp / "many.h264"
    many.write_bytes(sps(640, 368, 0, 0, 640, 360) + b"".join(slice_nal(f, f == 0) for f in range(40)))
    greedy = RocDecodeStream(StreamConfig(name="greedy", url=str(many), warmup_seconds=0.0), hold=64, num_surfaces=8)
    greedy.open_sync()
    try:
        for _ in range(40):
            assert greedy.next_packet() is not None
        raise AssertionError("40 pictures held out of 8 surfaces")
    except RuntimeError as exc:
        assert "ROCDEC_OUTOF_MEMORY" in str(exc), exc
    greedy.close_sync()

    # ---- a stream K1 cannot take (10-bit would be the real case; here: an unsupported size) fails at the sequence header
    huge = tmp / "huge.h264"
    huge.write_bytes(sps(8192, 4320, 0, 0, 8192, 4320) + slice_nal(0, True))
    big = RocDecodeStream(StreamConfig(name="huge", url=str(huge), warmup_seconds=0.0))
    big.open_sync()
    try:
        big.next_packet()
        raise AssertionError("an 8K stream went through a 4096 x 2304 decoder")
    except RuntimeError as exc:
        assert "does not support this codec / picture size" in str(exc), exc
    big.close_sync()

    # ---- a missing file is the reference's "Unable to open stream" (video_stream.py:78-79)
    try:
        RocDecodeStream(StreamConfig(name="nofile", url=str(tmp / "absent.h265"), warmup_seconds=0.0)).open_sync()
        raise AssertionError("opened a file that does not exist")
    except RuntimeError as exc:
        assert "Unable to open stream nofile" in str(exc)
    torch.cuda.synchronize()
    print("DECODE-WORKER-OK", st)


if __name__ == "__main__":
    main()
