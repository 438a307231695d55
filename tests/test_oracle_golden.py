"""The CPU oracle (oracle/rva_oracle.c) against the goldens recorded from the reference itself.

These pin the oracle BEFORE it is trusted as the checker of the HIP path.
"""
import numpy as np
import pytest

from oracle import oracle as orc
from realtime_video_analytics_32streams_amd import synth
from tests.conftest import load_golden
from tests.helpers import head_for_case, script_sha, table_digest, temporal_net


def _case_id(c):
    return f"{c['kind']}-{c.get('seed', c.get('name'))}-{c.get('model_type', 'v8')}"


@pytest.mark.parametrize("case", load_golden("post_cases.json")["cases"], ids=_case_id)
def test_postprocess_matches_reference(case):
    raw = head_for_case(case)
    exp = case["expect"]
    got = orc.postprocess(raw, case["conf"], case["iou"], case["classes"], tuple(case["orig_wh"]))
    assert got["n"] == exp["n"]
    assert got["n_cand"] == exp["n_cand"]
    assert got["anchor"].tolist() == exp["anchor"]          # bit-exact box indices
    assert got["keep"].tolist() == exp["keep"]
    assert got["cls"].tolist() == exp["cls"]
    # float32 values widened to Python floats by the reference: compare exactly
    assert [float(v) for v in got["conf"]] == exp["conf"]
    assert [[float(v) for v in b] for b in got["boxes"]] == exp["boxes"]


def test_tracker_inline_probes(tracker_cases):
    for probe in tracker_cases["inline"]:
        cfg = probe["cfg"]
        names = sorted({s["stream"] for s in probe["steps"]})
        trk = orc.Tracker(len(names), cfg["max_age"], cfg["max_iou_distance"], cfg["min_hits"])
        for step in probe["steps"]:
            res = trk.update(names.index(step["stream"]), np.asarray(step["boxes"], np.float64).reshape(-1, 4),
                             step["conf"], step["cls"])
            assert orc.table_of(res) == step["table"], probe["name"]


@pytest.mark.parametrize("idx", range(5))
def test_tracker_seeded_scripts(tracker_cases, idx):
    case = tracker_cases["seeded"][idx]
    cfg = case["cfg"]
    script = synth.make_tracker_script(case["seed"], case["n_streams"], case["n_ticks"], n_obj=case["n_obj"])
    assert script_sha(script) == case["sha"], "numpy drift in the tracker script generator"
    trk = orc.Tracker(case["n_streams"], cfg["max_age"], cfg["max_iou_distance"], cfg["min_hits"])
    k = 0
    last = {}
    for t in range(case["n_ticks"]):
        for s in range(case["n_streams"]):
            fd = script[t][s]
            m = fd.conf >= case["conf_thr"]          # F1 filter_detections (pipeline.py:182)
            res = trk.update(s, fd.boxes[m], fd.conf[m], fd.cls[m])
            tab = orc.table_of(res)
            assert table_digest(tab) == case["digests"][k], (t, s)
            last[f"s{s}"] = tab
            k += 1
    assert last == case["final"]


def test_clip_schedule_matches_reference():
    for c in load_golden("temporal_buffer.json"):
        fired, ids = orc.clip_schedule(c["L"], c["stride"], c["overlap"], 120)
        ref_a = [f for s, f in c["fired"] if s == "a"]
        assert fired == ref_a
        assert ids == c["clips"][:len(ref_a)]
        # second stream buffers independently and identically
        assert ids == c["clips"][len(ref_a):]


def test_letterbox_meta():
    for c in load_golden("letterbox_meta.json"):
        m = orc.letterbox(c["w"], c["h"], c["tw"], c["th"])
        assert m["scale"] == c["scale"] and list(m["new"]) == c["new"] and list(m["pad"]) == c["pad"]


def test_resize_special_cases():
    """Decimation at 3:1 and 2x2 mean at 6:1 / 2:1 (SURVEY.md P1 probed arithmetic)."""
    rng = np.random.default_rng(0)
    src = rng.integers(0, 256, (54, 96, 3), dtype=np.uint8)
    out = orc.resize_linear(src, 32, 18)                      # 3:1 -> src[3y+1, 3x+1]
    assert np.array_equal(out, src[1::3, 1::3])
    out = orc.resize_linear(src, 48, 27)                      # 2:1 -> INTER_AREA 2x2 mean
    s = src.astype(np.int32)
    area = (s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2
    assert np.array_equal(out, area.astype(np.uint8))
    out = orc.resize_linear(src, 16, 9)                       # 6:1 -> mean of src[6y+2..3, 6x+2..3]
    m = (s[2::6, 2::6] + s[2::6, 3::6] + s[3::6, 2::6] + s[3::6, 3::6] + 2) >> 2
    assert np.array_equal(out, m.astype(np.uint8))
    assert np.array_equal(orc.resize_linear(src, 96, 54), src)  # 1:1 copy


def test_fp16_normalise_is_a_half_multiply():
    """uint8.astype(float16) * (1/255) under NEP-50 is a binary16 multiply."""
    v = np.arange(256, dtype=np.uint8)
    want = v.astype(np.float16) * (1.0 / 255.0)
    bgr = np.repeat(v[None, :, None], 3, axis=2).repeat(2, axis=0)       # 2x256 frame
    out, meta = orc.preprocess_bgr(bgr, tw=256, th=2, half=True)
    assert out.dtype == np.float16 and np.array_equal(out[0, 0].view(np.uint16), want.view(np.uint16))
    out32, _ = orc.preprocess_bgr(bgr, tw=256, th=2, half=False)
    assert np.array_equal(out32[0, 0], v.astype(np.float32) * (1.0 / 255.0))


def test_preprocess_letterbox_layout():
    bgr = synth.make_bgr(3, 96, 54)
    out, meta = orc.preprocess_bgr(bgr, tw=64, th=64, half=False)
    assert meta["pad"] == (0, 14) and abs(meta["scale"] - 2 / 3) < 1e-12
    pad = np.float32(114) * np.float32(1.0 / 255.0)
    assert np.all(out[:, :14] == pad) and np.all(out[:, 50:] == pad)
    rs = orc.resize_linear(bgr, 64, 36)
    assert np.array_equal(out[0, 14:50], rs[..., 2].astype(np.float32) * np.float32(1 / 255.0))  # R plane first


# ---- normalisation arithmetic of the remaining heads --------------------------------------------------
@pytest.mark.parametrize("norm", [0, 1, 2], ids=["imagenet_f32", "video_f32", "imagenet_f64"])
@pytest.mark.parametrize("out_dtype", [0, 1, 2], ids=["f16", "f32", "f64"])
def test_oracle_frame_normalisation_matches_numpy_semantics(norm, out_dtype):
    """The oracle's C normalise step against numpy evaluating the reference's own expressions (same dtypes, same
    operation order) on the oracle-resized uint8 image: temporal_detector.py:350-354 (norm 0, also detector.py:988-993),
    :570-573 (norm 1), :741-743 (norm 2, float64 constant arrays).  Pins the float arithmetic incl. the float64 ->
    float16 single rounding; the resize in front of it stays unpinned (no OpenCV here)."""
    if out_dtype == 2 and norm != 2:
        pytest.skip("the reference produces float64 only in the ConvGRU pre-process")
    rng = np.random.default_rng(11 + norm)
    bgr = rng.integers(0, 256, (45, 70, 3), dtype=np.uint8)
    tw, th = 32, 24
    rs = orc.resize_bgr(bgr, tw, th)
    image = rs[..., ::-1].astype(np.float32) / 255.0                     # BGR2RGB; astype(float32) / 255.0
    if norm == 0:
        mean = np.array([0.485, 0.456, 0.406], dtype=np.float32); std = np.array([0.229, 0.224, 0.225], dtype=np.float32)
    elif norm == 1:
        mean = np.array([0.45, 0.45, 0.45], dtype=np.float32); std = np.array([0.225, 0.225, 0.225], dtype=np.float32)
    else:
        mean = np.array([0.485, 0.456, 0.406]); std = np.array([0.229, 0.224, 0.225])
    image = (image - mean) / std
    assert image.dtype == (np.float64 if norm == 2 else np.float32)
    want = np.ascontiguousarray(np.transpose(image, (2, 0, 1)))
    want = want.astype({0: np.float16, 1: np.float32, 2: np.float64}[out_dtype])
    got = orc.preprocess_norm_frames([bgr], tw, th, norm, out_dtype)[0]
    assert got.dtype == want.dtype and np.array_equal(got.view(np.uint8), want.view(np.uint8))
    # clip layouts: [T,C,H,W] stacks frames on axis 0, [C,T,H,W] is its transpose (temporal_detector.py:583-590)
    frames = [bgr, bgr[::-1].copy(), bgr[:, ::-1].copy()]
    tchw = orc.preprocess_norm_frames(frames, tw, th, norm, out_dtype, layout=0)
    cthw = orc.preprocess_norm_frames(frames, tw, th, norm, out_dtype, layout=1)
    assert np.array_equal(np.transpose(tchw, (1, 0, 2, 3)), cthw) and np.array_equal(tchw[0], got)


def test_temporal_networks_reproduce_reference_logits_on_cpu():
    """S3 / 8f-4: CnnLstmNet and Cnn3dNet ARE the reference's DummyCNNLSTM / Dummy3DCNN -- same parameters under the key map
    cnn->stem, lstm->rnn, fc->head, same logits for the same clip (the reference loops over frames, this package batches
    them through the stem: fp32 reassociation only)."""
    import torch
    for case in load_golden("temporal_nets.json"):
        net, x = temporal_net(case)
        with torch.inference_mode():
            got = net(x).numpy()
        want = np.asarray(case["logits"], np.float32)
        assert got.shape == want.shape
        assert np.allclose(got, want, rtol=1e-4, atol=1e-5), (case["kind"], float(np.abs(got - want).max()))


def test_temporal_key_map_loads_a_reference_state_dict():
    from realtime_video_analytics_32streams_amd import synth
    from realtime_video_analytics_32streams_amd.temporal import CnnLstmNet, load_reference_state_dict
    net = CnnLstmNet(8, 16)
    ref_named = {}
    for k, v in synth.seeded_module(lambda: CnnLstmNet(8, 16), 5).state_dict().items():
        for ours, theirs in (("stem.", "cnn."), ("rnn.", "lstm."), ("head.", "fc.")):
            if k.startswith(ours):
                k = theirs + k[len(ours):]
        ref_named[k] = v
    assert any(k.startswith("cnn.") for k in ref_named) and any(k.startswith("lstm.") for k in ref_named)
    load_reference_state_dict(net, ref_named, "cnn_lstm")
    assert synth.state_sha(net.state_dict()) == synth.state_sha(ref_named, synth.TEMPORAL_KEY_MAP)


def test_goldens_replay_clean_under_asan_and_ubsan():
    """SURVEY.md section 5 (race / memory checking; sanitizers run on the CPU build only): the oracle built with
    -fsanitize=address,undefined (oracle/Makefile target ``asan``) replays every golden case of this file in a child process
    with the sanitizer runtimes preloaded; any report (heap overflow in the NMS scratch, signed overflow, misaligned load ...)
    fails the test."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    libs = []
    for name in ("libasan.so", "libubsan.so"):
        p = subprocess.run(["gcc", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
        if not p or not Path(p).is_file():
            pytest.skip(f"{name} is not installed")
        libs.append(str(Path(p).resolve()))
    subprocess.check_call(["make", "-s", "-C", str(root / "oracle"), "asan"])
    env = dict(os.environ, LD_PRELOAD=" ".join(libs), RVA_ORACLE_LIB="liborc_asan.so",
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-m", "pytest", str(Path(__file__)), "-q", "-x", "-p", "no:cacheprovider",
                        "-k", "not asan"], cwd=str(root), env=env, capture_output=True, text=True, timeout=900)
    text = r.stdout + r.stderr
    assert r.returncode == 0, text[-3000:]
    assert "AddressSanitizer" not in text and "runtime error" not in text, text[-3000:]
    assert " passed" in r.stdout


@pytest.mark.parametrize("w,h,q", [(64, 48, 75), (640, 360, 75), (250, 123, 50), (333, 201, 90), (72, 40, 80), (8, 8, 30), (24, 56, 95),
                                   (100, 100, 10), (17, 9, 75), (1, 1, 75), (33, 47, 100), (640, 362, 1), (1288, 728, 55)])
def test_jpeg_oracle_is_pinned_by_pillows_libjpeg(w, h, q):
    """oracle/jpeg_oracle.py (the CPU restatement the device JPEG encoder K7 is held to) against libjpeg itself: Pillow's
    libjpeg-turbo decodes this encoder's stream to EXACTLY the pixels it decodes from its own encoding of the same BGR image at the
    same quality and 4:2:0 sampling, and reports the same quantisation tables -- i.e. colour conversion, edge rules (last column
    before / last downsampled row after the 2x2 chroma downsampling, dummy luma blocks), islow DCT and quantisation are libjpeg's;
    only the entropy coding differs (baseline + restart markers here, which the decoder checks by decoding).  Geometries: whole
    MCUs, 8- and odd-pixel remainders on both axes, a single pixel, quality 1 .. 100."""
    import io
    from PIL import Image
    from oracle import jpeg_oracle as J
    from realtime_video_analytics_32streams_amd import synth
    bgr = synth.make_bgr(7 + w, w, h)
    data = J.encode(bgr, q)
    mine = Image.open(io.BytesIO(data)); mine.load()
    buf = io.BytesIO()
    Image.fromarray(np.ascontiguousarray(bgr[..., ::-1])).save(buf, format="JPEG", quality=q, subsampling="4:2:0")
    ref = Image.open(io.BytesIO(buf.getvalue())); ref.load()
    assert mine.size == ref.size == (w, h)
    assert {k: list(v) for k, v in mine.quantization.items()} == {k: list(v) for k, v in ref.quantization.items()}
    assert np.array_equal(np.asarray(mine.convert("RGB")), np.asarray(ref.convert("RGB")))
    assert len(data) < len(buf.getvalue()) * 1.05 + 64            # the restart markers cost a few bytes per MCU row, no more


@pytest.mark.parametrize("w,h", [(1280, 720), (1000, 562), (853, 480), (1920, 1080), (719, 1281), (640, 360)])
def test_preprocess_oracle_agrees_with_an_independent_float_bilinear(w, h):
    """P1 is PARITY-UNPINNED against OpenCV itself (cv2 absent, no fixture in the reference).  What CAN be checked without it: the
    oracle's restatement of cv2.resize(INTER_LINEAR) -- tap positions from fx = (dx + 0.5) scale - 0.5, 11-bit weights, the
    fixed-point vertical pass -- against an independent implementation of the same sampling rule in floating point
    (torch.nn.functional.interpolate, bilinear, align_corners=False: half-pixel centres, no antialiasing), plus the letterbox
    geometry and the 114 border.  Agreement within ONE grey level everywhere (the integer output's own rounding + the 11-bit
    weights) says the taps and weights are right; it cannot say the last bit is OpenCV's -- that stays unpinned."""
    import torch
    import torch.nn.functional as F
    from realtime_video_analytics_32streams_amd import synth
    bgr = synth.make_bgr(3 + w, w, h)
    out, meta = orc.preprocess_bgr(bgr, 640, 640, False)              # float32 [3, 640, 640], RGB / 255, letterboxed
    scale = min(640 / w, 640 / h)
    nw, nh = int(w * scale), int(h * scale)
    assert meta["scale"] == scale
    x = torch.from_numpy(bgr[..., ::-1].copy()).permute(2, 0, 1)[None].float()
    ref = F.interpolate(x, size=(nh, nw), mode="bilinear", align_corners=False, antialias=False)[0].numpy()
    left, top = meta["pad"]
    got = out[:, top:top + nh, left:left + nw] * 255.0
    assert np.abs(got - ref).max() < 1.0, float(np.abs(got - ref).max())
    assert np.abs(got - ref).mean() < 0.3
    border = np.ones_like(out, dtype=bool)
    border[:, top:top + nh, left:left + nw] = False
    assert np.allclose(out[border], np.float32(114.0) / np.float32(255.0))


def test_nv12_colour_matrix_agrees_with_the_bt601_definition():
    """The NV12 -> BGR step in front of P1 stands in for what FFmpeg's swscale hands cv2.VideoCapture (unpinned: neither is
    present).  Independent check of the integer matrix the oracle and K1 use (298 / 409 / 208 / 100 / 516, >> 8) against the
    BT.601 limited-range DEFINITION in floating point (R = 1.164 (Y - 16) + 1.596 (Cr - 128), ...), nearest chroma: within one
    grey level for every (Y, Cb, Cr) of a synthetic frame, exact clipping at both ends."""
    from realtime_video_analytics_32streams_amd import synth
    y, uv = synth.make_nv12(11, 640, 360, 768)
    bgr = orc.nv12_to_bgr(y, uv, 640, 360).astype(np.float64)
    Y = y[:360, :640].astype(np.float64) - 16.0
    cb = np.repeat(np.repeat(uv[:180, 0:640:2].astype(np.float64), 2, 0), 2, 1) - 128.0
    cr = np.repeat(np.repeat(uv[:180, 1:640:2].astype(np.float64), 2, 0), 2, 1) - 128.0
    ref = np.stack([1.164383 * Y + 2.017232 * cb, 1.164383 * Y - 0.391762 * cb - 0.812968 * cr, 1.164383 * Y + 1.596027 * cr], -1)
    ref = np.clip(ref, 0.0, 255.0)
    assert np.abs(bgr - ref).max() <= 1.0 and np.abs(bgr - ref).mean() < 0.4
