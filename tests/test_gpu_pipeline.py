"""Pipeline-level parity on the GPU.

* the bench configuration end to end against the ORACLE: 32 streams, YOLOv8s fused plan, two ticks in flight, captured
  hipGraphs -- each tick's head tensor goes through the oracle's post-process + tracker in canonical order and the track
  tables must be identical (ids, age, hits, float64 boxes);
* device-side gates against the recording of the reference's StreamWorker (tests/golden/adaptive_fps.json);
* a gated, two-resolution, two-detector configuration: PipelinedTicks (gates on the device, graphs) == TickPipeline.tick
  (gates on the host) == the configuration routed per detector_id like pipeline.py:470-489;
* overflow of the bounded device tables is an error, never a silent divergence from the unbounded reference.
"""
import copy

import numpy as np
import pytest
import torch

from oracle import oracle as orc
from realtime_video_analytics_32streams_amd import _native as N
from realtime_video_analytics_32streams_amd import ops, synth
from realtime_video_analytics_32streams_amd.config import DetectorConfig, StreamConfig, TrackerConfig, config_from_dict
from realtime_video_analytics_32streams_amd.detector import HipYoloDetector
from realtime_video_analytics_32streams_amd.gates import AdaptiveFps, MotionGate
from realtime_video_analytics_32streams_amd.pipeline import PipelinedTicks, TickPipeline
from realtime_video_analytics_32streams_amd.tracker import IouTracker
from realtime_video_analytics_32streams_amd.video_stream import SyntheticNv12Stream
from realtime_video_analytics_32streams_amd.yolov8 import build_detector_net, calibrate_detection_density
from tests.conftest import load_golden

pytestmark = pytest.mark.gpu


def _dcfg(**kw):
    base = dict(model_path="yolov8n.pt", backend="hip", model_type="yolov8", warmup=False, half=True, confidence_threshold=0.25)
    base.update(kw)
    return DetectorConfig(**base)


def _calibrated(scale, streams, srcs, target, seed=0, **cfg):
    det = HipYoloDetector(_dcfg(**cfg), net=build_detector_net(scale, seed=seed))
    with torch.inference_mode():
        sample, _ = ops.preprocess_nv12([s._ring[0] for s in srcs[:8]], (640, 640), half=True)
    calibrate_detection_density(det.net, sample.contiguous(memory_format=torch.channels_last), 0.25, target)
    det.invalidate_engine()
    return det


def _tab(tracks):
    return [(t.track_id, t.class_id, t.age, t.hits, t.confidence, t.bbox_xyxy) for t in tracks]


@pytest.mark.parametrize("depth", [2, 4])
def test_bench_configuration_end_to_end_against_the_oracle(depth):
    """BASELINE configs[2] as bench.py runs it (32 x 1080p, YOLOv8s fused plan, tick chains on probed streams -- four by default,
    two as the sharded runs use --, hipGraph tails), 12 ticks."""
    S, T = 32, 12
    streams = [StreamConfig(name=f"cam{i:03d}", url="synthetic://1920x1080", warmup_seconds=0.0) for i in range(S)]
    srcs = [SyntheticNv12Stream(s, index=i, n_unique=3) for i, s in enumerate(streams)]
    for s in srcs:
        s.open_sync()
    det = _calibrated("s", streams, srcs, 120)
    tcfg = TrackerConfig(max_age=30, max_iou_distance=0.5, min_hits=1)
    trk = IouTracker(tcfg, max_streams=S, capacity=1024)
    runner = PipelinedTicks(TickPipeline(streams, det, trk, sources=srcs), depth=depth, use_graph=True)
    assert runner.net_streams == depth
    otr = orc.Tracker(S, tcfg.max_age, tcfg.max_iou_distance, tcfg.min_hits)
    checked = 0

    def check(k):
        nonlocal checked
        _, tables = runner.collect()
        par = k % runner.nslots                                    # every slot runs on its own plan, head tensor 0 of that plan:
        head = det._plans[(S, 640, 640) if par == 0 else (S, 640, 640, par)]._outs[0].float().cpu().numpy()   # intact until tick k + depth's network
        for s in range(S):                                         # canonical order: tick-major, stream-minor
            r = orc.postprocess(head[s], det.config.confidence_threshold, det.config.iou_threshold, None, (1920, 1080))
            m = r["conf"].astype(np.float64) >= det.config.confidence_threshold          # filter_detections
            want = otr.update(s, r["boxes"][m].astype(np.float64), r["conf"][m].astype(np.float64), r["cls"][m].astype(np.int64))
            assert orc.table_of(tables[s]) == orc.table_of(want), (k, s)
            checked += want["n"]
    done = 0
    for k in range(T):
        if k - done == runner.depth:
            check(done); done += 1
        runner.submit()
    while done < T:
        check(done); done += 1
    assert runner._captured and checked > 20 * T                  # the graphs were in use and there was something to track


def test_device_gates_follow_the_reference_worker_recording():
    """K4's on-device adaptive-fps gate against tests/golden/adaptive_fps.json (recorded from StreamWorker._process_packet).
    The recording's stand-in tracker returns its detections as tracks; max_age = 0 makes the real tracker do the same."""
    dev = torch.device("cuda")
    for case in load_golden("adaptive_fps.json"):
        st = StreamConfig(name="s", url="x", adaptive_fps=True, **case["cfg"])
        a = AdaptiveFps(st)
        trk = ops.DeviceTracker(1, max_age=0, max_iou_distance=0.5, min_hits=0, capacity=64)
        trk.set_gates([1], [a.max_process_every], [a.idle_tolerance], [0])
        post = ops.PostBuffers.allocate(1, 16, dev)
        for d in range(16):                                        # well separated boxes: one track per detection
            post.boxes[0, d] = torch.tensor([100.0 * d, 10.0, 100.0 * d + 50.0, 60.0])
        post.scores.fill_(0.9); post.cls.fill_(1)
        processed, updates = [], []
        for f, n_det in enumerate(case["script"]):
            post.counts[0] = n_det
            trk.update_from_post([0], post, 0.25, gated=True)
            trk.assign_ids()
            tabs = trk.read_all()
            em, pr, fl = trk.snapshot_status(0)
            assert fl == 0
            if pr[0] == 1:
                processed.append(f)
                assert em[0] == n_det and tabs[0]["n"] == n_det
            else:
                assert pr[0] == 0 and em[0] == 0 and tabs[0]["n"] == 0
            updates.append(int(em[0]))
        assert processed == case["processed"], case["cfg"]
        assert updates == case["tracker_updates"]
        trk.close()


def test_motion_gate_rejects_geometry_and_kind_changes():
    gate = MotionGate(2, thresholds=[0.02, 0.02])
    y, uv = synth.make_nv12(1, 640, 360, 768)
    big = synth.make_nv12(2, 1280, 720, 1280)
    a = ops.Nv12Surface.from_numpy(y, uv, 640, 360)
    b = ops.Nv12Surface.from_numpy(big[0], big[1], 1280, 720)
    assert gate.step([a, b]) == [True, True]                       # two geometries in one tick: one K5 launch each
    assert gate.geom == [(640, 360, "nv12"), (1280, 720, "nv12")]
    with pytest.raises(ValueError, match="geometry changed"):
        gate.step([b, b])                                          # stream 0 suddenly delivers 720p
    with pytest.raises(ValueError, match="geometry changed"):
        gate.step([torch.zeros((360, 640, 3), dtype=torch.uint8, device="cuda"), None])   # ... or another kind of frame
    with pytest.raises(ValueError):
        gate.step([a])                                             # wrong number of streams


def _mixed_setup():
    """6 streams: two resolutions, two detectors (the second one sees nothing at its 0.99 threshold, so its adaptive-fps
    streams go idle), a motion-gated still camera, a motion-gated moving camera."""
    mk = lambda n, wh, **kw: StreamConfig(name=n, url=f"synthetic://{wh[0]}x{wh[1]}", warmup_seconds=0.0, target_fps=30.0, **kw)  # noqa: E731
    streams = [mk("a0", (1920, 1080)),
               mk("a1-still", (1920, 1080), motion_filter=True, motion_threshold=0.01),
               mk("b0", (1280, 720), detector_id="quiet", adaptive_fps=True, min_target_fps=10.0, idle_frame_tolerance=2),
               mk("a2-moving", (1280, 720), motion_filter=True, motion_threshold=0.001),
               mk("b1", (1920, 1080), detector_id="quiet", adaptive_fps=True, min_target_fps=6.0, idle_frame_tolerance=3),
               mk("a3", (1280, 720), adaptive_fps=True, min_target_fps=10.0, idle_frame_tolerance=2)]
    uniq = [3, 1, 3, 4, 2, 3]

    def sources():
        out = []
        for i, s in enumerate(streams):
            w, h = (int(v) for v in s.url.split("//")[1].split("x"))
            out.append(SyntheticNv12Stream(s, index=i, width=w, height=h, n_unique=uniq[i]))
            out[-1].open_sync()
        return out
    return streams, sources


def test_gated_mixed_two_detector_ticks_agree_across_modes():
    streams, sources = _mixed_setup()
    srcs = sources()
    det_a = _calibrated("n", streams, [srcs[0], srcs[1], srcs[4]], 60)
    det_b = HipYoloDetector(_dcfg(confidence_threshold=0.99), net=build_detector_net("n", seed=3))
    per_stream = [det_b if s.detector_id == "quiet" else det_a for s in streams]
    tcfg = TrackerConfig(max_age=3, max_iou_distance=0.5, min_hits=1)
    T = 12

    def run(mode):
        trk = IouTracker(tcfg, max_streams=8, capacity=512)        # more tracker streams than the pipeline uses (padding path)
        pipe = TickPipeline(streams, per_stream, trk, sources=sources())
        out = []
        if mode.startswith("tick"):
            for _ in range(T):
                r = pipe.tick(device_gates=(mode == "tick-device"))
                out.append(({n: _tab(v) for n, v in r.tracks.items()}, dict(r.detections_emitted)))
            return out, pipe
        depth = 3 if mode.endswith("3") else 2
        runner = PipelinedTicks(pipe, depth=depth, use_graph=mode.startswith("graph"))
        inflight = 0
        for _ in range(T):
            if inflight == depth:
                r = runner.collect_result(); inflight -= 1
                out.append(({n: _tab(v) for n, v in r.tracks.items()}, dict(r.detections_emitted)))
            runner.submit(); inflight += 1
        while inflight:
            r = runner.collect_result(); inflight -= 1
            out.append(({n: _tab(v) for n, v in r.tracks.items()}, dict(r.detections_emitted)))
        if mode.startswith("graph"):
            assert runner._captured
        return out, pipe

    host, pipe = run("tick-host")                                  # gates on the host: gated-out frames skip the detector
    assert len(pipe.detectors) == 2 and pipe.det_of == [0, 0, 1, 0, 1, 0]
    # the gates did something: the still camera is processed once, the quiet detector's streams thin out
    proc = {n: [n in e for _, e in host] for n in pipe.names}
    assert sum(proc["a1-still"]) == 1 and sum(proc["a2-moving"]) >= T // 2 and sum(proc["a0"]) == T
    assert sum(proc["b0"]) < T and sum(proc["b1"]) < T and pipe.adaptive[2].process_every == 3 and pipe.adaptive[4].process_every == 5
    assert sum(len(v) for tr, _ in host for v in tr.values()) > 0
    # gates on the device: the same decisions tick by tick (detector batches differ from the host-gated run, so the
    # boxes agree only to the network's fp16 tolerance; which frames were processed is exact)
    want, _ = run("tick-device")
    assert {n: [n in e for _, e in want] for n in pipe.names} == proc
    # throughput mode == synchronous mode, bit for bit (same batches, same kernels, same order per stream)
    for mode in ("eager", "graph", "graph3"):                   # two ticks in flight, and three (three chains on three streams)
        got, _ = run(mode)
        assert got == want, mode


def test_from_config_routes_streams_by_detector_id():
    cfg = config_from_dict({
        "streams": [{"name": "x", "url": "synthetic://640x360", "warmup_seconds": 0.0},
                    {"name": "off", "url": "synthetic://640x360", "enabled": False},
                    {"name": "y", "url": "synthetic://640x360", "warmup_seconds": 0.0, "detector_id": "strict"},
                    {"name": "z", "url": "synthetic://640x360", "warmup_seconds": 0.0, "detector_id": "strict"}],
        "detector": {"backend": "hip", "model_path": "yolov8n.pt", "half": True, "confidence_threshold": 0.3, "warmup": False},
        "detectors": {"strict": {"backend": "hip", "model_path": "yolov8n.pt", "half": True, "confidence_threshold": 0.9,
                                 "classes": [3], "warmup": False},
                      "unused": {"backend": "hip", "model_path": "yolov8m.pt", "warmup": False}},
        "tracker": {"max_age": 5, "max_iou_distance": 0.5, "min_hits": 1}})
    pipe = TickPipeline.from_config(cfg)
    assert pipe.names == ["x", "y", "z"] and pipe.det_of == [0, 1, 1] and len(pipe.detectors) == 2      # "unused" is never built
    assert pipe.detectors[0].config.confidence_threshold == 0.3 and pipe.detectors[1].config.classes == [3]
    r = pipe.tick()
    assert set(r.tracks) == {"x", "y", "z"}


def test_overflow_of_the_device_tables_is_an_error():
    streams = [StreamConfig(name=f"c{i}", url="synthetic://1920x1080", warmup_seconds=0.0) for i in range(2)]
    srcs = [SyntheticNv12Stream(s, index=i, n_unique=2) for i, s in enumerate(streams)]
    for s in srcs:
        s.open_sync()
    det = _calibrated("n", streams, srcs, 80)
    tcfg = TrackerConfig(max_age=5, max_iou_distance=0.5, min_hits=1)
    pipe = TickPipeline(streams, det, IouTracker(tcfg, max_streams=2, capacity=4), sources=srcs)    # far fewer rows than detections
    with pytest.raises(RuntimeError, match="capacity exceeded"):
        pipe.tick()
    srcs2 = [SyntheticNv12Stream(s, index=i, n_unique=2) for i, s in enumerate(streams)]
    runner = PipelinedTicks(TickPipeline(streams, det, IouTracker(tcfg, max_streams=2, capacity=4), sources=srcs2), depth=1, use_graph=False)
    runner.submit()
    with pytest.raises(RuntimeError, match="capacity exceeded"):
        runner.collect()


def test_host_path_detector_feeds_the_shared_tracker():
    """A temporal head (host detections) and a YOLO group in one pipeline: ids come from the one global counter in
    canonical stream order regardless of which path a stream takes (BASELINE configs[4] in miniature)."""
    from realtime_video_analytics_32streams_amd.temporal import CnnLstmNet, HipCNNLSTMDetector
    streams = [StreamConfig(name="clip", url="synthetic://640x360", warmup_seconds=0.0),
               StreamConfig(name="yolo", url="synthetic://640x360", warmup_seconds=0.0)]
    srcs = [SyntheticNv12Stream(s, index=i, width=640, height=360, n_unique=2) for i, s in enumerate(streams)]
    for s in srcs:
        s.open_sync()
    tcfg_det = DetectorConfig(model_path="x.onnx", backend="hip", model_type="cnn_lstm", sequence_length=2, sequence_stride=1,
                              temporal_overlap=0.5, confidence_threshold=-1e9, num_action_classes=8, input_size=[64, 64], warmup=False)
    torch.manual_seed(1)
    clip_det = HipCNNLSTMDetector(tcfg_det, net=CnnLstmNet(8, hidden=16))
    yolo = _calibrated("n", streams, [srcs[1]], 30)
    trk = IouTracker(TrackerConfig(max_age=5, max_iou_distance=0.5, min_hits=1), max_streams=2, capacity=256)
    pipe = TickPipeline(streams, [clip_det, yolo], trk, sources=srcs)
    r0 = pipe.tick()
    assert r0.tracks["clip"] == [] and len(r0.tracks["yolo"]) > 0
    n0 = len(r0.tracks["yolo"])
    r1 = pipe.tick()                                               # the clip fires: 5 full-frame TemporalDetections, two classes may repeat
    clip_tracks = r1.tracks["clip"]
    assert len(clip_tracks) >= 1 and all(t.bbox_xyxy == (0.0, 0.0, 640.0, 360.0) for t in clip_tracks)
    assert clip_tracks[0].track_id == n0 + 1                       # the global counter continues where tick 0 left off; stream 0 first
    assert clip_tracks[0].sequence_end_frame == 1 and clip_tracks[0].temporal_score is not None
    assert r1.detections_emitted["clip"] == 5
    # the same mixed pipeline in throughput mode (temporal head + YOLO group, two ticks in flight): same tables
    srcs2 = [SyntheticNv12Stream(s, index=i, width=640, height=360, n_unique=2) for i, s in enumerate(streams)]
    for s in srcs2:
        s.open_sync()
    torch.manual_seed(1)
    clip_det2 = HipCNNLSTMDetector(tcfg_det, net=copy.deepcopy(clip_det.net).float().cpu())
    trk2 = IouTracker(TrackerConfig(max_age=5, max_iou_distance=0.5, min_hits=1), max_streams=2, capacity=256)
    runner = PipelinedTicks(TickPipeline(streams, [clip_det2, yolo], trk2, sources=srcs2), depth=2)
    runner.submit(); runner.submit()
    q0, q1 = runner.collect_result(), runner.collect_result()
    assert _tab(q0.tracks["yolo"]) == _tab(r0.tracks["yolo"]) and q0.tracks["clip"] == []
    assert [(t.track_id, t.class_id, t.action_label, t.sequence_end_frame) for t in q1.tracks["clip"]] == \
           [(t.track_id, t.class_id, t.action_label, t.sequence_end_frame) for t in clip_tracks]
    assert q1.detections_emitted == r1.detections_emitted


def test_config4_shape_8x4k_cnn_lstm_clips_through_the_pipeline():
    """BASELINE configs[4] at its real sizes on one GPU: 8 x 3840x2160 NV12 streams, CNN-LSTM with L = 16, stride 2,
    overlap 0.5 at 224x224 (sample-temporal-pipeline.yaml:34-36,48), every stream through the shared tracker.
    Schedule against the reference's recording (first clip at frame 31, then every 8 frames; tests/golden/temporal_buffer.json),
    logits of a first clip against CPU fp32 on oracle-preprocessed frames, ids in canonical stream order."""
    from realtime_video_analytics_32streams_amd.temporal import CnnLstmNet, HipCNNLSTMDetector
    S, T = 8, 40
    rec = next(c for c in load_golden("temporal_buffer.json") if (c["L"], c["stride"], c["overlap"]) == (16, 2, 0.5))
    streams = [StreamConfig(name=f"uhd{i}", url="synthetic://3840x2160", warmup_seconds=0.0) for i in range(S)]
    srcs = [SyntheticNv12Stream(s, index=i, width=3840, height=2160, n_unique=2) for i, s in enumerate(streams)]
    for s in srcs:
        s.open_sync()
    cfg = DetectorConfig(model_path="cnn_lstm.onnx", backend="hip", model_type="cnn_lstm", sequence_length=16, sequence_stride=2,
                         temporal_overlap=0.5, confidence_threshold=-1e9, num_action_classes=400, input_size=[224, 224], warmup=False,
                         action_classes=[f"act{i}" for i in range(400)])
    torch.manual_seed(1)
    net = CnnLstmNet(400).eval()
    det = HipCNNLSTMDetector(cfg, net=copy.deepcopy(net))
    trk = IouTracker(TrackerConfig(max_age=30, max_iou_distance=0.5, min_hits=1), max_streams=S, capacity=64)
    pipe = TickPipeline(streams, det, trk, sources=srcs)
    fired = []
    first = None
    for t in range(T):
        r = pipe.tick()
        if any(r.detections_emitted.get(n, 0) for n in pipe.names):
            fired.append(t)
            assert all(r.detections_emitted[n] == 5 for n in pipe.names)           # top-5 of the raw output, every stream
            if first is None:      # Track objects alias tracker state (as in the reference): copy what tick 31 looked like
                first = {n: [copy.copy(t) for t in v] for n, v in r.tracks.items()}
    assert fired == [f for s, f in rec["fired"] if s == "a" and f < T]             # 31, 39
    # tick 31: 8 streams x 5 new tracks (5 distinct classes each), ids 1..40 in stream order
    ids = [[t.track_id for t in first[n]] for n in pipe.names]
    assert ids == [list(range(5 * i + 1, 5 * i + 6)) for i in range(S)]
    tr = first["uhd3"][0]
    assert tr.bbox_xyxy == (0.0, 0.0, 3840.0, 2160.0) and tr.sequence_start_frame == 0 and tr.sequence_end_frame == 30
    assert tr.action_label == f"act{tr.class_id}"
    # the clip of stream 3 at tick 31: frames 0, 2, ..., 30 (the ring of 2 synthetic frames alternates) -> CPU fp32 reference
    ring = [(s.y.cpu().numpy(), s.uv.cpu().numpy()) for s in srcs[3]._ring]
    x = np.stack([orc.preprocess_clip_frame(nv12=ring[f % 2], wh=(3840, 2160), tw=224, th=224, half=False) for f in rec["clips"][0]])
    with torch.inference_mode():
        want = net(torch.from_numpy(x)[None]).flatten().numpy()
    top = np.argsort(want, kind="stable")[-5:][::-1]
    got = first["uhd3"]
    assert [t.class_id for t in got] == top.tolist()
    assert np.allclose([t.confidence for t in got], want[top], atol=1e-3)
    # throughput mode (two tick chains: the clip pre-process of tick k+1 beside the network of tick k) gives the same tracks
    srcs2 = [SyntheticNv12Stream(s, index=i, width=3840, height=2160, n_unique=2) for i, s in enumerate(streams)]
    for s in srcs2:
        s.open_sync()
    det2 = HipCNNLSTMDetector(cfg, net=copy.deepcopy(net))
    trk2 = IouTracker(TrackerConfig(max_age=30, max_iou_distance=0.5, min_hits=1), max_streams=S, capacity=64)
    runner = PipelinedTicks(TickPipeline(streams, det2, trk2, sources=srcs2), depth=2)
    assert runner.net_streams == 2 and not runner.use_graph
    fired2, last = [], None
    runner.submit()
    for t in range(1, T + 1):
        if t < T:
            runner.submit()
        r = runner.collect_result()
        if any(r.detections_emitted.get(n, 0) for n in pipe.names):
            fired2.append(r.tick)
            if last is None:
                last = {n: [(t_.track_id, t_.class_id, t_.action_label, t_.sequence_start_frame, t_.sequence_end_frame) for t_ in v]
                        for n, v in r.tracks.items()}
                conf2 = {n: [t_.confidence for t_ in v] for n, v in r.tracks.items()}
    assert fired2 == fired
    assert last == {n: [(t_.track_id, t_.class_id, t_.action_label, t_.sequence_start_frame, t_.sequence_end_frame) for t_ in v]
                    for n, v in first.items()}
    assert all(np.allclose(conf2[n], [t_.confidence for t_ in first[n]], atol=1e-4) for n in first)


def test_wire_bytes_from_a_device_snapshot_equal_the_track_object_path():
    """Event wire format on real device output (/root/reference/src/realtime_analytics/sinks/kafka_sink.py:103-132):
    the tables ``PipelinedTicks.collect()`` hands the host (pinned snapshot written by K4) turned into Kafka message bytes by
    ``wire.payload_from_table`` -- no Track objects -- must be byte-identical to ``wire.tracks_payload`` on the Track objects
    the tracker materialises from the same snapshot (what the reference's sink serialises), and parse back on the consumer side."""
    from realtime_video_analytics_32streams_amd import wire
    S, T = 4, 8
    streams = [StreamConfig(name=f"cam{i}", url="synthetic://1920x1080", warmup_seconds=0.0) for i in range(S)]
    srcs = [SyntheticNv12Stream(s, index=i, n_unique=3) for i, s in enumerate(streams)]
    for s in srcs:
        s.open_sync()
    det = _calibrated("n", streams, srcs, 60)
    trk = IouTracker(TrackerConfig(max_age=30, max_iou_distance=0.5, min_hits=1), max_streams=S, capacity=512)
    pipe = TickPipeline(streams, det, trk, sources=srcs)
    runner = PipelinedTicks(pipe, depth=3, use_graph=True)
    msgs = tracks_seen = 0

    def check(k):
        nonlocal msgs, tracks_seen
        _, tables = runner.collect()
        for i, s in enumerate(streams):
            tab = tables[pipe.slots[i]]
            direct = wire.serialize(wire.payload_from_table(s.name, k, tab))
            objs = trk.tracks_from_tables([s.name], [tab])[0]
            via_tracks = wire.serialize(wire.tracks_payload(s.name, k, objs))
            assert direct == via_tracks, (k, s.name)
            ev = wire.parse_event(direct)
            assert ev["stream"] == s.name and ev["frame_id"] == k and len(ev["tracks"]) == int(tab["n"])
            assert [t["track_id"] for t in ev["tracks"]] == [int(v) for v in tab["id"][:int(tab["n"])]]
            msgs += 1
            tracks_seen += int(tab["n"])
    for k in range(T):
        if k >= runner.depth:
            check(k - runner.depth)
        runner.submit()
    for k in range(max(T - runner.depth, 0), T):
        check(k)
    assert msgs == S * T and tracks_seen > S * T


def test_gated_temporal_stream_keeps_its_gates_on_the_host():
    """A gate (adaptive_fps / motion_filter) on a stream with a TEMPORAL head: the reference never shows a skipped frame to the
    detector (pipeline.py:156-181), so its clip buffer holds processed frames only.  The device-gate modes pre-process every
    delivered frame before the gate decides -- fine for the stateless YOLO head, a silent divergence for a clip ring -- and
    therefore refuse the combination; ``tick()`` with host-decided gates runs it, and the clip schedule then sees exactly the
    frames the gate let through."""
    from realtime_video_analytics_32streams_amd.temporal import CnnLstmNet, HipCNNLSTMDetector
    streams = [StreamConfig(name="clip", url="synthetic://640x360", warmup_seconds=0.0, target_fps=30.0, adaptive_fps=True,
                            min_target_fps=10.0, idle_frame_tolerance=2),
               StreamConfig(name="plain", url="synthetic://640x360", warmup_seconds=0.0)]
    srcs = [SyntheticNv12Stream(s, index=i, width=640, height=360, n_unique=2) for i, s in enumerate(streams)]
    for s in srcs:
        s.open_sync()
    dcfg = DetectorConfig(model_path="x.onnx", backend="hip", model_type="cnn_lstm", sequence_length=2, sequence_stride=1,
                          temporal_overlap=0.5, confidence_threshold=1e9, num_action_classes=8, input_size=[64, 64], warmup=False)
    torch.manual_seed(1)
    det = HipCNNLSTMDetector(dcfg, net=CnnLstmNet(8, hidden=16))          # threshold 1e9: never emits -> the stream goes idle
    trk = IouTracker(TrackerConfig(max_age=5, max_iou_distance=0.5, min_hits=1), max_streams=2, capacity=64)
    pipe = TickPipeline(streams, det, trk, sources=srcs)
    assert pipe.gated_stateful == ["clip"]
    with pytest.raises(NotImplementedError, match="temporal"):
        PipelinedTicks(pipe, depth=2)
    with pytest.raises(NotImplementedError, match="temporal"):
        pipe.tick(device_gates=True)
    seen_before = det._arrivals.get("clip", 0)
    processed = 0
    for _ in range(12):
        r = pipe.tick()                                                   # host-decided gates
        processed += 1 if "clip" in r.detections_emitted else 0
    arrivals = det._arrivals["clip"] - seen_before
    assert 0 < arrivals < 12 and det._arrivals["plain"] == 12              # the idle stream was throttled; skipped frames never reached its ring
