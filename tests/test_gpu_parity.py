"""Parity tests proper: the HIP path (through the C ABI) against the committed goldens recorded
from the reference, and against the CPU oracle on seeded inputs.  Need a real MI355X.

Bars (BASELINE.json north_star): bit-exact track ids / kept-box indices; coords and scores within
1e-3 -- these kernels are in fact held to exact float equality, the tolerance is never used.
"""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from realtime_video_analytics_32streams_amd import _native as N
from realtime_video_analytics_32streams_amd import ops, synth
from tests.conftest import load_golden
from tests.helpers import head_for_case, script_sha, table_digest

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _post_one(raw2d, conf, iou, classes, wh, dtype=torch.float32):
    t = torch.from_numpy(np.ascontiguousarray(raw2d)).to(DEV).to(dtype)[None].contiguous()
    m = N.letterbox(wh[0], wh[1], 640, 640)
    return ops.postprocess(t, conf, iou, classes, [m]).to_host()[0]


def _case_id(c):
    return f"{c['kind']}-{c.get('seed', c.get('name'))}-{c.get('model_type', 'v8')}"


# ---------------------------------------------------------------------------------------- K2+K3
@pytest.mark.parametrize("case", load_golden("post_cases.json")["cases"], ids=_case_id)
def test_postprocess_matches_reference_goldens(case):
    raw = head_for_case(case)
    exp = case["expect"]
    got = _post_one(raw, case["conf"], case["iou"], case["classes"], tuple(case["orig_wh"]))
    assert got["n"] == exp["n"] and got["n_cand"] == exp["n_cand"]
    assert got["anchor"].tolist() == exp["anchor"]
    assert got["keep"].tolist() == exp["keep"]
    assert got["cls"].tolist() == exp["cls"]
    assert [float(v) for v in got["conf"]] == exp["conf"]
    assert [[float(v) for v in b] for b in got["boxes"]] == exp["boxes"]


def test_postprocess_batch32_matches_oracle_fp32_and_fp16():
    """BASELINE config 3 shape: [32, 84, 8400]; every image checked against the oracle."""
    seeds = list(range(100, 132))
    heads = synth.make_head_batch(seeds, layout="CA", n_obj=30)
    metas = [N.letterbox(1920, 1080, 640, 640)]
    for dtype in (torch.float32, torch.float16):
        t = torch.from_numpy(heads).to(DEV).to(dtype).contiguous()
        res = ops.postprocess(t, 0.25, 0.45, None, metas).to_host()
        ref_in = t.float().cpu().numpy()          # the oracle sees exactly the values the kernel saw
        for b in range(len(seeds)):
            want = orc.postprocess(ref_in[b], 0.25, 0.45, None, (1920, 1080))
            got = res[b]
            assert got["n"] == want["n"] and got["n_cand"] == want["n_cand"], (dtype, b)
            if dtype == torch.float16:
                # fp16 heads make score ties likely; the project tie rule is (score desc, anchor asc)
                assert len(np.unique(want["conf"])) <= want["n"]
            assert np.array_equal(got["anchor"], want["anchor"]), (dtype, b)
            assert np.array_equal(got["keep"], want["keep"])
            assert np.array_equal(got["cls"], want["cls"])
            assert np.array_equal(got["conf"], want["conf"])
            assert np.array_equal(got["boxes"], want["boxes"])
    assert ops.post_status() == 0


def test_postprocess_score_ties_follow_project_rule():
    head = synth.make_head(77, n_obj=10)
    head[:, 4:] = np.round(head[:, 4:] * 16) / 16        # quantise -> many exact score ties
    raw = np.ascontiguousarray(head.T)
    got = _post_one(raw, 0.25, 0.45, None, (1920, 1080))
    want = orc.postprocess(raw, 0.25, 0.45, None, (1920, 1080))
    assert want["n"] > 3 and len(np.unique(want["conf"])) < want["n_cand"]
    assert np.array_equal(got["anchor"], want["anchor"]) and np.array_equal(got["boxes"], want["boxes"])


def test_postprocess_worst_case_every_anchor_is_a_candidate():
    """K = 8400 candidates (maximum size): sort + chunked NMS must still equal the sequential loop."""
    rng = np.random.default_rng(5)
    A = 8400
    p = np.empty((A, 84), np.float32)
    p[:, 0] = rng.uniform(0, 640, A); p[:, 1] = rng.uniform(140, 500, A)
    p[:, 2] = rng.uniform(20, 120, A); p[:, 3] = rng.uniform(20, 120, A)
    p[:, 4] = rng.uniform(0.9, 1.0, A)
    p[:, 5:] = rng.uniform(0.3, 1.0, (A, 79))
    raw = np.ascontiguousarray(p.T)
    got = _post_one(raw, 0.25, 0.5, None, (1920, 1080))
    want = orc.postprocess(raw, 0.25, 0.5, None, (1920, 1080))
    assert want["n_cand"] == A and got["n_cand"] == A
    assert got["n"] == want["n"] and np.array_equal(got["anchor"], want["anchor"])
    assert np.array_equal(got["boxes"], want["boxes"]) and np.array_equal(got["conf"], want["conf"])


def test_postprocess_busy_images_across_sort_sizes_and_nms_rounds():
    """Busy images: K3 sorts 512 ... 8192 keys in registers (1 ... 16 keys per thread) and runs greedy NMS in rounds of 512
    sorted boxes (phase 1 against the boxes kept so far, phase 2 the survivors among themselves on a suppression matrix
    in LDS).  One batch with candidate counts on both sides of the sort sizes, of the 64-survivor chunks and of the
    512-box rounds; every image against the oracle, at a threshold with few and one with many survivors per round."""
    rng = np.random.default_rng(11)
    A = 8400
    counts = [128, 129, 191, 192, 193, 511, 512, 513, 1000, 1025, 2500, 4095, 4096, 4097, 0, 64]
    heads = np.zeros((len(counts), 84, A), np.float32)
    for b, k in enumerate(counts):
        p = np.zeros((A, 84), np.float32)
        idx = rng.permutation(A)[:k]                              # the candidates sit anywhere in anchor order
        # dense scene: boxes cluster around 40 centres, so that most candidates are suppressed by an earlier one
        c = rng.integers(0, 40, k)
        cx0, cy0 = rng.uniform(60, 580, 40), rng.uniform(160, 480, 40)
        p[idx, 0] = cx0[c] + rng.normal(0, 6, k); p[idx, 1] = cy0[c] + rng.normal(0, 6, k)
        p[idx, 2] = rng.uniform(30, 90, k); p[idx, 3] = rng.uniform(30, 90, k)
        p[idx, 4] = rng.uniform(0.9, 1.0, k)
        p[idx, 5:] = rng.uniform(0.3, 1.0, (k, 79)).astype(np.float32)
        heads[b] = p.T
    t = torch.from_numpy(heads).to(DEV)
    for thr in (0.45, 0.9):                                       # few survivors / many survivors per chunk
        res = ops.postprocess(t, 0.25, thr, None, [N.letterbox(1920, 1080, 640, 640)]).to_host()
        for b, k in enumerate(counts):
            want = orc.postprocess(heads[b], 0.25, thr, None, (1920, 1080))
            got = res[b]
            assert want["n_cand"] == k == got["n_cand"], (b, k)
            assert got["n"] == want["n"], (thr, k, got["n"], want["n"])
            assert np.array_equal(got["anchor"], want["anchor"]) and np.array_equal(got["keep"], want["keep"]), (thr, k)
            assert np.array_equal(got["cls"], want["cls"]) and np.array_equal(got["conf"], want["conf"])
            assert np.array_equal(got["boxes"], want["boxes"])
    assert ops.post_status() == 0


@pytest.mark.parametrize("A,n_boxes", [(4096, 3000), (8400, 6000)])
def test_postprocess_more_kept_boxes_than_the_lds_list_holds(A, n_boxes):
    """Thousands of boxes that barely overlap: the kept list outgrows its LDS copy (1024 boxes; 512 in the 16384-key layout that
    6000 candidates select) and phase 1 reads the rest back from the output rows -- with the LDS part sorted into centre bins.  A
    jittered grid of small boxes with a sprinkling of duplicates (suppressed by a box that may sit in either part)."""
    rng = np.random.default_rng(A)
    p = np.zeros((A, 84), np.float32)
    idx = rng.permutation(A)[:n_boxes]
    side = int(np.ceil(np.sqrt(n_boxes * 0.9)))
    cell_w, cell_h = 620.0 / side, 340.0 / side
    g = rng.permutation(side * side)[:n_boxes] if side * side >= n_boxes else rng.integers(0, side * side, n_boxes)
    p[idx, 0] = 10 + (g % side + 0.5) * cell_w + rng.uniform(-0.1, 0.1, n_boxes) * cell_w
    p[idx, 1] = 150 + (g // side + 0.5) * cell_h + rng.uniform(-0.1, 0.1, n_boxes) * cell_h
    p[idx, 2] = cell_w * rng.uniform(0.7, 1.0, n_boxes); p[idx, 3] = cell_h * rng.uniform(0.7, 1.0, n_boxes)
    p[idx, 4] = 1.0
    p[idx, 5] = rng.uniform(0.3, 1.0, n_boxes).astype(np.float32)
    raw = np.ascontiguousarray(p.T)
    ops.post_filter_stats()
    got = _post_one(raw, 0.25, 0.5, None, (1920, 1080))
    want = orc.postprocess(raw, 0.25, 0.5, None, (1920, 1080))
    assert ops.post_filter_stats() == 1 and ops.post_status() == 0
    assert want["n"] > 1024 + 512 and want["n"] < n_boxes                     # far past the LDS list; some were suppressed
    assert got["n"] == want["n"] and np.array_equal(got["anchor"], want["anchor"])
    assert np.array_equal(got["boxes"], want["boxes"]) and np.array_equal(got["conf"], want["conf"])


def test_postprocess_detection_cap_keeps_the_first_rows_and_raises_the_flag():
    """max_det below the number of boxes an image keeps (the reference has no cap): the rows written are the first max_det rows of
    the uncapped result, the count is max_det and bit 0 of rva_post_status() is set -- with the kept list short (unsorted scan) and
    long (centre bins: the cap sits inside the sorted list)."""
    heads = synth.make_head_batch([700, 701, 702, 703], layout="CA", n_obj=220)
    t = torch.from_numpy(heads).to(DEV)
    lb = [N.letterbox(1920, 1080, 640, 640)]
    full = ops.postprocess(t, 0.25, 0.45, None, lb).to_host()
    assert ops.post_status() == 0 and min(r["n"] for r in full) > 150
    for cap in (16, 128):
        res = ops.postprocess(t, 0.25, 0.45, None, lb, max_det=cap).to_host()
        assert ops.post_status() & 1
        for got, ref in zip(res, full):
            assert got["n"] == cap
            assert np.array_equal(got["anchor"], ref["anchor"][:cap]) and np.array_equal(got["boxes"], ref["boxes"][:cap])
            assert np.array_equal(got["conf"], ref["conf"][:cap]) and np.array_equal(got["cls"], ref["cls"][:cap])
    want = orc.postprocess(heads[0], 0.25, 0.45, None, (1920, 1080))
    assert np.array_equal(full[0]["anchor"], want["anchor"])


def test_postprocess_suppression_decisions_at_the_threshold_boundary():
    """K3 decides `iou > thr` without dividing (an exact comparison in float64 against the point where the rounded quotient
    leaves thr).  Pairs of boxes whose float32 IoU lands within a few ulps of the threshold, on both sides of it and on it:
    the lower-scored box must survive exactly when the oracle's divided `iou <= thr` says so.  Identity letterbox (640x640
    source), so the head's cx / cy / w / h are the boxes."""
    rng = np.random.default_rng(3)
    f32 = np.float32
    for thr in (0.45, 0.5, 0.7, 0.3):
        t = f32(thr)
        r = (1.0 - thr) / (1.0 + thr)                                 # shift / width at which two equal boxes have IoU = thr
        a = rng.integers(40, 300, 4000).astype(f32) * f32(0.5)        # widths / heights on a 0.5 grid: halves are exact
        b = rng.integers(40, 300, 4000).astype(f32) * f32(0.5)
        dx = (a * f32(r)).astype(f32)
        for _ in range(3):                                            # walk dx onto the boundary in float32 steps
            inter = (a - dx) * b
            q = inter / np.maximum(a * b + a * b - inter, f32(1e-6))
            dx = np.where(q > t, np.nextafter(dx, f32(1e9)), np.nextafter(dx, f32(-1e9))).astype(f32)
        dx = np.concatenate([np.nextafter(dx, f32(-1e9)), dx, np.nextafter(dx, f32(1e9))]).astype(f32)
        a, b = np.tile(a, 3), np.tile(b, 3)
        inter = (a - dx) * b
        q = inter / np.maximum(a * b + a * b - inter, f32(1e-6))
        near = np.abs(q.astype(np.float64) - float(t)) <= 4 * np.spacing(t)
        pick = np.concatenate([np.flatnonzero(near & (q > t))[:256], np.flatnonzero(near & (q <= t))[:256]])
        a, b, dx, q = a[pick], b[pick], dx[pick], q[pick]
        assert (q > t).sum() >= 64 and (q <= t).sum() >= 64
        heads = np.zeros((len(a), 84, 128), f32)                      # [84, 128]: 128 anchors (rows < cols), two of them candidates
        heads[:, 0, 0] = a / 2 + 100; heads[:, 1, 0] = b / 2 + 100; heads[:, 2, 0] = a; heads[:, 3, 0] = b
        heads[:, 0, 1] = a / 2 + 100 + dx; heads[:, 1, 1] = b / 2 + 100; heads[:, 2, 1] = a; heads[:, 3, 1] = b
        heads[:, 4, :2] = 1.0
        heads[:, 5, 0] = 0.9; heads[:, 5, 1] = 0.8
        n_sup = 0
        for i0 in range(0, len(a), 64):
            h = np.ascontiguousarray(heads[i0:i0 + 64])
            res = ops.postprocess(torch.from_numpy(h).to(DEV), 0.25, thr, None, [N.letterbox(640, 640, 640, 640)]).to_host()
            for k, got in enumerate(res):
                want = orc.postprocess(h[k], 0.25, thr, None, (640, 640))
                assert got["n"] == want["n"] and np.array_equal(got["anchor"], want["anchor"]), (thr, i0 + k)
                n_sup += want["n"] == 1
        assert 0 < n_sup < len(a)                                     # both outcomes occurred


@pytest.mark.parametrize("src", [(640, 640), (1920, 1080)])
def test_postprocess_centre_bin_filter_finds_every_suppressor(src):
    """K3 scans, for a box, only the kept boxes whose centre-x bin lies within the distance at which a suppressor can sit
    (rva_postprocess.hip, "centre-bin filter").  Scenes built against that bound: per threshold, hundreds of (kept, victim)
    pairs at the LARGEST centre distance the geometry allows for an IoU just above the threshold -- victim much narrower than
    its suppressor (flush left / flush right), much wider, equal widths shifted -- next to the same constructions just below
    the threshold (victim survives), spread over the whole width with filler boxes so that the kept list spans every bin and
    the slices matter.  Every image must equal the oracle's sequential NMS, and the diagnostic counter must say the filter ran;
    an image with one improper candidate (negative raw width) steps aside to the unfiltered scan -- same answer -- and the flag
    does not leak into the next launch."""
    rng = np.random.default_rng(17)
    f32 = np.float32
    sw, sh = src
    lb = N.letterbox(sw, sh, 640, 640)
    ops.post_filter_stats()
    for thr in (0.2, 0.3, 0.45, 0.5, 0.7, 0.9):
        heads = np.zeros((8, 84, 2048), f32)
        for img in range(8):
            k = 0
            def put(cx, cy, w, h, score):
                nonlocal k
                heads[img, 0:4, k] = (cx, cy, w, h); heads[img, 4, k] = 1.0; heads[img, 5 + (k % 3), k] = score
                k += 1
            y0, y1 = (150.0, 490.0) if sw == 1920 else (10.0, 630.0)          # inside the letterboxed content
            for j in range(60):
                c = 1.03 if j % 2 == 0 else 0.97                               # IoU = thr c: suppressed / survives
                kind = j % 4 if thr * c < 0.95 else 3
                h = float(rng.uniform(12, 30)); cy = float(rng.uniform(y0 + 20, y1 - 20))
                x0 = float(rng.uniform(5, 400)); score = float(rng.uniform(0.5, 0.9))
                if kind == 0:            # kept = wide, victim = narrow, flush left: IoU = w / W
                    W = float(rng.uniform(60, 200)); w = W * thr * c
                    put(x0 + W / 2, cy, W, h, score); put(x0 + w / 2, cy, w, h, score - 0.2)
                elif kind == 1:          # the same, flush right
                    W = float(rng.uniform(60, 200)); w = W * thr * c
                    put(x0 + W / 2, cy, W, h, score); put(x0 + W - w / 2, cy, w, h, score - 0.2)
                elif kind == 2:          # kept = narrow, victim = wide
                    W = float(rng.uniform(60, 200)); w = W * thr * c
                    put(x0 + w / 2, cy, w, h, score); put(x0 + W / 2, cy, W, h, score - 0.2)
                else:                    # equal widths, shifted: IoU = (w - d) / (w + d)
                    w = float(rng.uniform(30, 150)); d = w * (1 - thr * c) / (1 + thr * c)
                    put(x0 + w / 2, cy, w, h, score); put(x0 + d + w / 2, cy, w, h, score - 0.2)
            for j in range(300):         # filler: small boxes everywhere (most are kept: the list covers every bin)
                put(float(rng.uniform(10, 630)), float(rng.uniform(y0, y1)), float(rng.uniform(4, 12)), float(rng.uniform(4, 12)), float(rng.uniform(0.3, 0.95)))
        t = torch.from_numpy(heads).to(DEV)
        res = ops.postprocess(t, 0.25, thr, None, [lb]).to_host()
        assert ops.post_status() == 0 and ops.post_filter_stats() == 8, thr
        n_kept = []
        for i, got in enumerate(res):
            want = orc.postprocess(heads[i], 0.25, thr, None, (sw, sh))
            assert got["n"] == want["n"] and np.array_equal(got["anchor"], want["anchor"]), (thr, i)
            assert np.array_equal(got["boxes"], want["boxes"]) and np.array_equal(got["conf"], want["conf"])
            n_kept.append(got["n"])
        assert min(n_kept) > 200                                               # a long kept list: slices, not the whole list
        # one improper candidate in image 3: that image alone takes the unfiltered scan
        bad = heads.copy()
        bad[3, 0:4, 2040] = (320.0, 320.0, -20.0, 10.0); bad[3, 4, 2040] = 1.0; bad[3, 5, 2040] = 0.99
        res = ops.postprocess(torch.from_numpy(bad).to(DEV), 0.25, thr, None, [lb]).to_host()
        assert ops.post_filter_stats() == 7
        for i, got in enumerate(res):
            want = orc.postprocess(bad[i], 0.25, thr, None, (sw, sh))
            assert got["n"] == want["n"] and np.array_equal(got["anchor"], want["anchor"]), ("improper", thr, i)
        ops.postprocess(t, 0.25, thr, None, [lb])
        assert ops.post_filter_stats() == 8                                    # the flag was cleared
    res = ops.postprocess(t, 0.25, 0.1, None, [lb]).to_host()                  # below the filter's threshold range: unfiltered
    assert ops.post_filter_stats() == 0
    for i, got in enumerate(res):
        want = orc.postprocess(heads[i], 0.25, 0.1, None, (sw, sh))
        assert got["n"] == want["n"] and np.array_equal(got["anchor"], want["anchor"])


def test_postprocess_properties_at_full_size():
    """Size-independent properties on the [32,84,8400] workload: idempotence of NMS on its own
    output, descending scores, every kept pair has IoU <= thr."""
    heads = synth.make_head_batch(list(range(200, 232)), layout="CA", n_obj=60)
    t = torch.from_numpy(heads).to(DEV)
    res = ops.postprocess(t, 0.2, 0.5, None, [N.letterbox(1920, 1080, 640, 640)]).to_host()
    for r in res:
        assert r["n"] > 0 and np.all(np.diff(r["conf"]) <= 0)
        b = r["boxes"].astype(np.float32)
        x1 = np.maximum(b[:, None, 0], b[None, :, 0]); y1 = np.maximum(b[:, None, 1], b[None, :, 1])
        x2 = np.minimum(b[:, None, 2], b[None, :, 2]); y2 = np.minimum(b[:, None, 3], b[None, :, 3])
        inter = np.maximum(0, x2 - x1) * np.maximum(0, y2 - y1)
        area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
        iou = inter / np.clip(area[:, None] + area[None, :] - inter, 1e-6, None)
        np.fill_diagonal(iou, 0)
        assert np.all(iou <= np.float32(0.5))
        assert np.all(b[:, [0, 2]] >= 0) and np.all(b[:, [0, 2]] <= 1919) and np.all(b[:, [1, 3]] <= 1079)


def test_postprocess_bad_shape_and_empty():
    t = torch.zeros((2, 10, 4), device=DEV)
    res = ops.postprocess(t, 0.25, 0.45, None, [N.letterbox(1920, 1080, 640, 640)]).to_host()
    assert [r["n"] for r in res] == [0, 0]
    t = torch.zeros((3, 84, 8400), device=DEV)
    res = ops.postprocess(t, 0.25, 0.45, None, [N.letterbox(1920, 1080, 640, 640)]).to_host()
    assert [r["n"] for r in res] == [0, 0, 0]


# ---------------------------------------------------------------------------------------- K4
def _gpu_table(res):
    return orc.table_of(res)


def test_tracker_inline_probes_match_reference(tracker_cases):
    for probe in tracker_cases["inline"]:
        cfg = probe["cfg"]
        names = sorted({s["stream"] for s in probe["steps"]})
        trk = ops.DeviceTracker(len(names), cfg["max_age"], cfg["max_iou_distance"], cfg["min_hits"], capacity=64)
        for step in probe["steps"]:
            s = names.index(step["stream"])
            trk.update_from_host({s: (np.asarray(step["boxes"], np.float64).reshape(-1, 4), step["conf"], step["cls"])})
            trk.assign_ids()
            assert _gpu_table(trk.read(s)) == step["table"], probe["name"]
        trk.close()


@pytest.mark.parametrize("idx", range(5))
def test_tracker_seeded_scripts_match_reference(tracker_cases, idx):
    """Streams are updated one at a time in canonical order, exactly as the golden harness drove
    the reference tracker; per-update digests of the full table must match."""
    case = tracker_cases["seeded"][idx]
    cfg = case["cfg"]
    script = synth.make_tracker_script(case["seed"], case["n_streams"], case["n_ticks"], n_obj=case["n_obj"])
    assert script_sha(script) == case["sha"]
    trk = ops.DeviceTracker(case["n_streams"], cfg["max_age"], cfg["max_iou_distance"], cfg["min_hits"], capacity=1024)
    k = 0
    for t in range(case["n_ticks"]):
        for s in range(case["n_streams"]):
            fd = script[t][s]
            m = fd.conf >= case["conf_thr"]
            trk.update_from_host({s: (fd.boxes[m], fd.conf[m], fd.cls[m])})
            trk.assign_ids()
            assert table_digest(_gpu_table(trk.read(s))) == case["digests"][k], (t, s)
            k += 1
    final = {f"s{s}": _gpu_table(trk.read(s)) for s in range(case["n_streams"])}
    assert final == case["final"]
    assert trk.state()[1] == 0
    trk.close()


@pytest.mark.parametrize("idx", [0, 2, 3])
def test_tracker_batched_tick_equals_sequential_reference(tracker_cases, idx):
    """All streams of a tick updated CONCURRENTLY (one wavefront each) + one id-assignment pass
    must reproduce the ids the reference hands out when called stream after stream."""
    case = tracker_cases["seeded"][idx]
    cfg = case["cfg"]
    S = case["n_streams"]
    script = synth.make_tracker_script(case["seed"], S, case["n_ticks"], n_obj=case["n_obj"])
    trk = ops.DeviceTracker(S, cfg["max_age"], cfg["max_iou_distance"], cfg["min_hits"], capacity=1024)
    k = 0
    for t in range(case["n_ticks"]):
        upd = {}
        for s in range(S):
            fd = script[t][s]
            m = fd.conf >= case["conf_thr"]
            upd[s] = (fd.boxes[m], fd.conf[m], fd.cls[m])
        trk.update_from_host(upd)
        trk.assign_ids()
        tabs = trk.read_all()
        for s in range(S):
            assert table_digest(_gpu_table(tabs[s])) == case["digests"][k], (t, s)
            k += 1
    trk.close()


def test_tracker_fused_f32_path_with_filter_and_skips():
    """Device-resident detections (the post-process output layout) + F1 filter inside the kernel,
    idle (-1) and skipped-frame (-2) slots; oracle driven in canonical order."""
    S, T, max_det = 6, 25, 64
    script = synth.make_tracker_script(21, S, T, n_obj=10)
    thr = 0.45
    trk = ops.DeviceTracker(S, 4, 0.4, 2, capacity=256)
    ref = orc.Tracker(S, 4, 0.4, 2)
    rng = np.random.default_rng(3)
    for t in range(T):
        post = ops.PostBuffers.allocate(S, max_det, DEV)
        slots, boxes, scores, cls, counts = [], np.zeros((S, max_det, 4), np.float32), np.zeros((S, max_det), np.float32), \
            np.zeros((S, max_det), np.int32), np.zeros(S, np.int32)
        row = 0
        want = {}
        for s in range(S):
            mode = rng.random()
            fd = script[t][s]
            if mode < 0.1:
                slots.append(-1)                 # no frame this tick: table untouched
                continue
            if mode < 0.2:
                slots.append(-2)                 # skipped frame: update(name, [])
                want[s] = ref.update(s, np.zeros((0, 4)), [], [])
                continue
            d = len(fd.conf)
            boxes[row, :d] = fd.boxes; scores[row, :d] = fd.conf; cls[row, :d] = fd.cls; counts[row] = d
            slots.append(row)
            m = fd.conf >= thr
            want[s] = ref.update(s, fd.boxes[m], fd.conf[m], fd.cls[m])
            row += 1
        post.boxes.copy_(torch.from_numpy(boxes)); post.scores.copy_(torch.from_numpy(scores))
        post.cls.copy_(torch.from_numpy(cls)); post.counts.copy_(torch.from_numpy(counts))
        trk.update_from_post(slots, post, thr)
        trk.assign_ids()
        tabs = trk.read_all()
        for s, w in want.items():
            assert _gpu_table(tabs[s]) == orc.table_of(w), (t, s)
    assert trk.state()[0] == ref.next_id
    trk.close()


@pytest.mark.parametrize("cfg", [
    # S, T, n_obj, capacity, max_det, (max_age, min_iou, min_hits), filter threshold, box scale of stream 1
    dict(S=32, T=50, n_obj=256, cap=1024, max_det=512, trk=(5, 0.3, 2), thr=0.45, scale=1.0),    # the load sweep's busy scene
    dict(S=4, T=30, n_obj=300, cap=1024, max_det=512, trk=(30, 0.5, 1), thr=0.3, scale=2.0),     # _rescale_detections inside
    dict(S=3, T=12, n_obj=580, cap=1024, max_det=1024, trk=(1, 0.5, 1), thr=0.3, scale=1.0),     # frames beyond the matrix's 512 detections
    dict(S=5, T=40, n_obj=40, cap=160, max_det=128, trk=(3, 0.4, 1), thr=0.5, scale=1.0),        # capacity that is no multiple of the 128-column pad
    dict(S=4, T=25, n_obj=220, cap=1024, max_det=512, trk=(3, 0.02, 1), thr=0.3, scale=1.0),     # nearly every detection overlaps an earlier one: all scans
], ids=["32x256x50", "rescaled", "beyond-matrix", "small-capacity", "all-scans"])
def test_tracker_busy_frames_matrix_form_equals_reference_scan(cfg):
    """K4 under load: k4_iou fills the float64 IoU matrix of the tick on all CUs, k4_update only picks maxima -- the tables
    after EVERY tick must equal the oracle's sequential scan (ids, hits, ages, float64 boxes), in dense scenes where a
    detection often has several candidate tracks, matches a track created or moved earlier in the same frame, and where
    three classes share the frame.  Skipped frames and idle streams ride along."""
    S, T = cfg["S"], cfg["T"]
    script = synth.make_tracker_script(97 + cfg["n_obj"], S, T, n_obj=cfg["n_obj"])
    max_age, min_iou, min_hits = cfg["trk"]
    trk = ops.DeviceTracker(S, max_age, min_iou, min_hits, capacity=cfg["cap"])
    ref = orc.Tracker(S, max_age, min_iou, min_hits)
    scales = [1.0] * S
    scales[1] = cfg["scale"]
    if cfg["scale"] != 1.0:
        trk.set_box_scale(scales)
    md = cfg["max_det"]
    post = ops.PostBuffers.allocate(S, md, DEV)
    rng = np.random.default_rng(1)
    seen_multi = 0
    for t in range(T):
        boxes, scores = np.zeros((S, md, 4), np.float32), np.zeros((S, md), np.float32)
        cls, counts = np.zeros((S, md), np.int32), np.zeros(S, np.int32)
        slots, want, row = [], {}, 0
        for s in range(S):
            fd = script[t][s]
            u = rng.random()
            if u < 0.04:
                slots.append(-1)
                continue
            if u < 0.08:
                slots.append(-2)
                want[s] = ref.update(s, np.zeros((0, 4)), [], [])
                continue
            d = min(len(fd.conf), md)
            boxes[row, :d] = fd.boxes[:d]; scores[row, :d] = fd.conf[:d]; cls[row, :d] = fd.cls[:d]; counts[row] = d
            slots.append(row)
            m = fd.conf[:d] >= cfg["thr"]
            want[s] = ref.update(s, fd.boxes[:d][m] * scales[s], fd.conf[:d][m], fd.cls[:d][m])
            seen_multi = max(seen_multi, d)
            row += 1
        post.boxes.copy_(torch.from_numpy(boxes)); post.scores.copy_(torch.from_numpy(scores))
        post.cls.copy_(torch.from_numpy(cls)); post.counts.copy_(torch.from_numpy(counts))
        trk.update_from_post(slots, post, cfg["thr"])
        trk.assign_ids()
        tabs = trk.read_all()
        for s, w in want.items():
            assert _gpu_table(tabs[s]) == orc.table_of(w), (t, s)
    assert trk.state() == (ref.next_id, 0)
    assert seen_multi >= min(cfg["n_obj"] * 0.8, md * 0.8)
    trk.close()


def test_tracker_full_table_drops_the_same_detections_in_both_forms():
    """The table has a fixed capacity (ours, not the reference's: the flag says detections were dropped).  When it fills
    up, the form that applies 64 detections per step must drop exactly the detections the in-loop form (float64 host
    source, one detection at a time) drops: new tracks once the table is full, matches never."""
    S, T, md, cap = 3, 25, 256, 120
    script = synth.make_tracker_script(411, S, T, n_obj=150)
    step = ops.DeviceTracker(S, 8, 0.3, 1, capacity=cap)
    loop = ops.DeviceTracker(S, 8, 0.3, 1, capacity=cap)
    post = ops.PostBuffers.allocate(S, md, DEV)
    flagged = False
    for t in range(T):
        boxes, scores = np.zeros((S, md, 4), np.float32), np.zeros((S, md), np.float32)
        cls, counts = np.zeros((S, md), np.int32), np.zeros(S, np.int32)
        host = {}
        for s in range(S):
            fd = script[t][s]
            d = min(len(fd.conf), md)
            boxes[s, :d] = fd.boxes[:d]; scores[s, :d] = fd.conf[:d]; cls[s, :d] = fd.cls[:d]; counts[s] = d
            host[s] = (boxes[s, :d].astype(np.float64), scores[s, :d].astype(np.float64), cls[s, :d].astype(np.int64))
        post.boxes.copy_(torch.from_numpy(boxes)); post.scores.copy_(torch.from_numpy(scores))
        post.cls.copy_(torch.from_numpy(cls)); post.counts.copy_(torch.from_numpy(counts))
        step.update_from_post(list(range(S)), post, 0.0)
        loop.update_from_host(host)
        step.assign_ids(); loop.assign_ids()
        a, b = step.read_all(), loop.read_all()
        for s in range(S):
            assert _gpu_table(a[s]) == _gpu_table(b[s]), (t, s)
        assert step.state() == loop.state()
        flagged = flagged or step.state()[1] != 0
    assert flagged, "the scenario never filled the table"
    step.close(); loop.close()


def test_tracker_sharded_ids_match_single_process():
    """Two trackers owning interleaved halves of 8 streams + exchanged new-track counts reproduce the
    ids of one tracker owning all 8 (the multi-GPU scheme of SURVEY.md 8e, on one device)."""
    S, T = 8, 20
    script = synth.make_tracker_script(31, S, T, n_obj=6)
    full = ops.DeviceTracker(S, 10, 0.5, 1, capacity=256)
    own = [[0, 2, 4, 6], [1, 3, 5, 7]]
    parts = [ops.DeviceTracker(4, 10, 0.5, 1, capacity=256) for _ in own]
    for t in range(T):
        full.update_from_host({s: (script[t][s].boxes, script[t][s].conf, script[t][s].cls) for s in range(S)})
        full.assign_ids()
        counts_all = torch.zeros(S, dtype=torch.int32, device=DEV)
        for p, streams in zip(parts, own):
            p.update_from_host({i: (script[t][s].boxes, script[t][s].conf, script[t][s].cls) for i, s in enumerate(streams)})
            counts_all[torch.tensor(streams, device=DEV)] = p.new_counts_tensor()
        for p, streams in zip(parts, own):
            p.assign_ids(counts_all, streams)
        ftab = full.read_all()
        for p, streams in zip(parts, own):
            ptab = p.read_all()
            for i, s in enumerate(streams):
                assert _gpu_table(ptab[i]) == _gpu_table(ftab[s]), (t, s)
    assert parts[0].state()[0] == full.state()[0] == parts[1].state()[0]


def test_tracker_capacity_overflow_is_flagged():
    trk = ops.DeviceTracker(1, 30, 0.5, 1, capacity=8)
    boxes = np.array([[i * 50.0, 0, i * 50.0 + 20, 20] for i in range(12)])
    trk.update_from_host({0: (boxes, np.full(12, 0.9), np.zeros(12, np.int64))})
    trk.assign_ids()
    assert trk.read(0)["n"] == 8 and trk.state()[1] & 1


# ---------------------------------------------------------------------------------------- K1
GEOMS = [(1920, 1080), (3840, 2160), (1280, 720), (640, 360), (2560, 1440), (1000, 700), (722, 1282), (96, 54)]


@pytest.mark.parametrize("wh", GEOMS, ids=lambda g: f"{g[0]}x{g[1]}")
@pytest.mark.parametrize("half", [True, False], ids=["fp16", "fp32"])
def test_preprocess_nv12_bit_exact(wh, half):
    w, h = wh
    pitch = ((w + 255) // 256) * 256
    frames = [synth.make_nv12(synth.SEED_BASE + 1000 * s, w, h, pitch, tick=s) for s in range(2)]
    surfs = [ops.Nv12Surface.from_numpy(y, uv, w, h) for y, uv in frames]
    out, meta = ops.preprocess_nv12(surfs, (640, 640), half=half)
    got = out.cpu().numpy()
    for i, (y, uv) in enumerate(frames):
        want, m = orc.preprocess_nv12(y, uv, w, h, 640, 640, half)
        assert meta.as_meta() == m
        view = np.uint16 if half else np.uint32
        assert np.array_equal(got[i].view(view), want.view(view)), (wh, half, i)


@pytest.mark.parametrize("wh", [(1920, 1080), (1001, 701), (333, 777), (640, 640), (64, 48)], ids=lambda g: f"{g[0]}x{g[1]}")
def test_preprocess_bgr_bit_exact(wh):
    w, h = wh
    frames = [synth.make_bgr(9 + i, w, h) for i in range(2)]
    dev = [torch.from_numpy(f).to(DEV) for f in frames]
    for half in (True, False):
        out, meta = ops.preprocess_bgr(dev, (640, 640), half=half)
        got = out.cpu().numpy()
        for i, f in enumerate(frames):
            want, m = orc.preprocess_bgr(f, 640, 640, half)
            assert meta.as_meta() == m
            view = np.uint16 if half else np.uint32
            assert np.array_equal(got[i].view(view), want.view(view))


def test_preprocess_other_input_sizes():
    for dst in [(416, 416), (320, 416), (1280, 1280)]:
        y, uv = synth.make_nv12(5, 1920, 1080)
        out, meta = ops.preprocess_nv12([ops.Nv12Surface.from_numpy(y, uv, 1920, 1080)], dst, half=True)
        want, m = orc.preprocess_nv12(y, uv, 1920, 1080, dst[1], dst[0], True)
        assert meta.as_meta() == m and np.array_equal(out.cpu().numpy()[0].view(np.uint16), want.view(np.uint16))


def test_preprocess_full_tick_32x1080p_properties():
    """BASELINE config 3 size (32 x 1080p): pad rows are 114/255, content equals the decimated
    colour-converted source (size-independent property), identical surfaces give identical planes."""
    y, uv = synth.make_nv12(1, 1920, 1080, 2048)
    s0 = ops.Nv12Surface.from_numpy(y, uv, 1920, 1080)
    surfs = [s0] * 31 + [ops.Nv12Surface.from_numpy(*synth.make_nv12(2, 1920, 1080, 2048), 1920, 1080)]
    out, meta = ops.preprocess_nv12(surfs, (640, 640), half=True)
    pad = (np.float16(114) * np.float16(1 / 255)).view(np.uint16)
    o = out.cpu().numpy().view(np.uint16)
    assert np.all(o[:, :, :140] == pad) and np.all(o[:, :, 500:] == pad)
    assert all(np.array_equal(o[0], o[i]) for i in range(1, 31)) and not np.array_equal(o[0], o[31])
    bgr = orc.nv12_to_bgr(y, uv, 1920, 1080)[1::3, 1::3]
    want = (bgr[..., ::-1].astype(np.float16) * np.float16(1 / 255)).transpose(2, 0, 1)
    assert np.array_equal(o[0][:, 140:500], want.view(np.uint16))


@pytest.mark.parametrize("half", [False, True], ids=["fp32", "fp16"])
def test_preprocess_clip_frames(half):
    y, uv = synth.make_nv12(8, 3840, 2160)
    out, _ = ops.preprocess_nv12([ops.Nv12Surface.from_numpy(y, uv, 3840, 2160)], (224, 224), half=half, clip=True)
    want = orc.preprocess_clip_frame(nv12=(y, uv), wh=(3840, 2160), tw=224, th=224, half=half)
    view = np.uint16 if half else np.uint32
    assert np.array_equal(out.cpu().numpy()[0].view(view), want.view(view))
    f = synth.make_bgr(4, 500, 300)
    out, _ = ops.preprocess_bgr([torch.from_numpy(f).to(DEV)], (112, 112), half=half, clip=True)
    want = orc.preprocess_clip_frame(bgr=f, tw=112, th=112, half=half)
    assert np.array_equal(out.cpu().numpy()[0].view(view), want.view(view))


# ---- frame pre-process of the ResNet / 3D-CNN / ConvGRU heads -----------------------------------------
@pytest.mark.parametrize("norm,dtype", [(0, "f16"), (0, "f32"), (1, "f16"), (1, "f32"), (2, "f16"), (2, "f32"), (2, "f64")])
@pytest.mark.parametrize("layout", [0, 1], ids=["TCHW", "CTHW"])
def test_preprocess_frames_norms_and_layouts(norm, dtype, layout):
    """rva_preprocess_frames_* against the oracle, bit-for-bit (float16 / float32 / float64 words), NV12 4K surfaces
    and odd-sized BGR frames, both clip layouts."""
    tdt = {"f16": torch.float16, "f32": torch.float32, "f64": torch.float64}[dtype]
    odt = {"f16": 0, "f32": 1, "f64": 2}[dtype]
    surf = [synth.make_nv12(20 + i, 3840, 2160, tick=i) for i in range(3)]
    out = ops.preprocess_frames([ops.Nv12Surface.from_numpy(y, uv, 3840, 2160) for y, uv in surf], (112, 112), norm, layout, tdt)
    want = orc.preprocess_norm_frames(surf, 112, 112, norm, odt, layout=layout, nv12_wh=(3840, 2160))
    assert out.shape == want.shape
    assert np.array_equal(out.cpu().numpy().view(np.uint8), want.view(np.uint8))
    frames = [synth.make_bgr(30 + i, 333, 201) for i in range(2)]
    out = ops.preprocess_frames([torch.from_numpy(f).to(DEV) for f in frames], (56, 40), norm, layout, tdt)
    want = orc.preprocess_norm_frames(frames, 40, 56, norm, odt, layout=layout)
    assert np.array_equal(out.cpu().numpy().view(np.uint8), want.view(np.uint8))


def test_preprocess_frames_rejects_f64_outside_convgru():
    f = torch.from_numpy(synth.make_bgr(1, 64, 48)).to(DEV)
    with pytest.raises(RuntimeError):
        ops.preprocess_frames([f], (32, 32), 0, 0, torch.float64)
