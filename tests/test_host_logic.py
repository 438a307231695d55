"""CPU tests of the host-side mirror: config loader, clip buffering rule, plugin API surface,
multi-process id scan over gloo (world_size 2)."""
import os
import socket
import subprocess
import sys
import textwrap
from collections import deque
from pathlib import Path

import pytest
import yaml

from realtime_video_analytics_32streams_amd import config as C
from realtime_video_analytics_32streams_amd.temporal import ClipSchedule
from tests.conftest import load_golden

ROOT = Path(__file__).resolve().parents[1]


def test_clip_schedule_matches_reference_recording():
    for c in load_golden("temporal_buffer.json"):
        sch = ClipSchedule(c["L"], c["stride"], c["overlap"])
        assert sch.step == c["step"]
        buf, fired, clips = deque(), [], []
        for f in range(120):
            clip, _ = sch.push(buf, f)
            if clip is not None:
                fired.append(f); clips.append(clip)
        n = len([1 for s, _ in c["fired"] if s == "a"])
        assert fired == [f for s, f in c["fired"] if s == "a"]
        assert clips == c["clips"][:n]


def _yaml(tmp_path, doc):
    p = tmp_path / "cfg.yaml"
    p.write_text(yaml.safe_dump(doc))
    return p


def test_load_config_reference_style_yaml(tmp_path):
    doc = {
        "max_concurrent_streams": 4, "stats_interval_seconds": 10,
        "streams": [{"name": "sim-1", "url": "synthetic://1920x1080", "target_fps": 12, "batch_size": 1,
                     "warmup_seconds": 0.5, "ffmpeg_simulator": {"enabled": False}, "max_frame_rate_per_stream": 9}],
        "detector": {"model_path": "yolov8n.pt", "device": "cpu", "backend": "hip", "confidence_threshold": 0.35,
                     "iou_threshold": 0.5, "half": False, "warmup": False, "bogus_key": 1},
        "tracker": {"type": "byte_track", "max_age": 30, "max_iou_distance": 0.5, "min_hits": 1},
        "kafka": {"enabled": False}, "prometheus": {"enabled": False},
    }
    cfg = C.load_config(_yaml(tmp_path, doc))
    assert cfg.streams[0].name == "sim-1" and cfg.streams[0].target_fps == 12
    assert cfg.detector.confidence_threshold == 0.35 and cfg.tracker.min_hits == 1
    assert cfg.detector_for(cfg.streams[0]) is cfg.detector
    assert not hasattr(cfg.detector, "bogus_key")          # unknown keys dropped (config.py:304-307)


@pytest.mark.parametrize("mut,msg", [
    (lambda d: d["streams"].clear(), "At least one stream"),
    (lambda d: d["streams"][0].update(url=""), "non-empty url"),
    (lambda d: d["streams"][0].update(batch_size=0), "batch_size"),
    (lambda d: d["streams"][0].update(downsample_ratio=0.01), "downsample_ratio"),
    (lambda d: d["streams"][0].update(detector_id="nope"), "unknown detector_id"),
    (lambda d: d["detector"].update(backend="cuda"), "backend must be one of"),
    (lambda d: d["detector"].update(confidence_threshold=0.0), "confidence_threshold"),
    (lambda d: d["detector"].update(iou_threshold=1.5), "iou_threshold"),
    (lambda d: d["detector"].update(input_size=[640]), "input_size"),
    (lambda d: d["tracker"].update(max_age=0), "max_age"),
    (lambda d: d["tracker"].update(max_iou_distance=0), "max_iou_distance"),
    (lambda d: d.update(max_concurrent_streams=0), "max_concurrent_streams"),
])
def test_config_validation_errors(mut, msg):
    doc = {"streams": [{"name": "a", "url": "synthetic://64x48"}], "detector": {"backend": "hip"}, "tracker": {}}
    mut(doc)
    with pytest.raises(C.ConfigError, match=msg):
        C.config_from_dict(doc)


def test_config_structure_errors(tmp_path):
    with pytest.raises(C.ConfigError, match="not found"):
        C.load_config(tmp_path / "missing.yaml")
    with pytest.raises(C.ConfigError, match="mapping"):
        C.config_from_dict([1, 2])
    with pytest.raises(C.ConfigError, match="'streams' must be a list"):
        C.config_from_dict({"streams": {}})
    with pytest.raises(C.ConfigError, match="'detectors' section"):
        C.config_from_dict({"streams": [{"name": "a", "url": "u"}], "detectors": [1]})


def test_temporal_config_and_reference_backends():
    d = C.DetectorConfig(model_type="cnn_lstm", backend="hip", sequence_length=16, sequence_stride=2)
    d.validate()
    with pytest.raises(C.ConfigError, match="temporal_overlap"):
        C.DetectorConfig(model_type="cnn_lstm", backend="hip", temporal_overlap=1.0).validate()
    from realtime_video_analytics_32streams_amd.detector import create_detector
    with pytest.raises(RuntimeError, match="belongs to the reference"):
        create_detector(C.DetectorConfig(backend="onnxruntime"))
    with pytest.raises(ValueError, match="Unsupported detector backend"):
        create_detector(C.DetectorConfig(backend="nope"))


def test_plugin_api_surface_matches_reference_names():
    from realtime_video_analytics_32streams_amd import detector as D
    from realtime_video_analytics_32streams_amd import tracker as T
    from realtime_video_analytics_32streams_amd import video_stream as V
    import dataclasses, inspect
    assert [f.name for f in dataclasses.fields(D.Detection)] == ["stream_name", "frame_id", "class_id", "confidence", "bbox_xyxy"]
    assert [f.name for f in dataclasses.fields(T.Track)] == [
        "track_id", "class_id", "confidence", "bbox_xyxy", "age", "hits", "action_label", "temporal_score",
        "sequence_start_frame", "sequence_end_frame"]
    assert [f.name for f in dataclasses.fields(V.FramePacket)] == ["stream", "frame", "frame_id", "timestamp"]
    assert inspect.isabstract(D.BaseDetector) and list(inspect.signature(D.BaseDetector.predict).parameters) == ["self", "packet"]
    assert list(inspect.signature(T.IouTracker.update).parameters) == ["self", "stream_name", "detections"]
    dets = [D.Detection("s", 0, 1, c, (0.0, 0.0, 1.0, 1.0)) for c in (0.2, 0.5, 0.7)]
    assert [d.confidence for d in D.filter_detections(dets, 0.5)] == [0.5, 0.7]


def test_hip_detector_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from realtime_video_analytics_32streams_amd.detector import create_detector
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        create_detector(C.DetectorConfig(backend="hip"))
    from realtime_video_analytics_32streams_amd.tracker import IouTracker
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        IouTracker(C.TrackerConfig())


def test_shard_streams():
    from realtime_video_analytics_32streams_amd.dist import exclusive_id_bases, shard_streams
    assert list(shard_streams(32, 3, 8)) == [12, 13, 14, 15]
    with pytest.raises(ValueError):
        shard_streams(30, 0, 8)
    assert exclusive_id_bases([2, 0, 3], 10) == [10, 12, 12]


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def test_id_sync_world2_gloo_matches_single_process(tmp_path):
    """Two gloo ranks each own half of 8 streams and drive the ORACLE tracker (CPU) with the id
    scheme of the multi-GPU path: all-gather of new counts -> exclusive scan -> final ids.  The ids
    must equal those of one process owning all 8 streams (the reference's global counter)."""
    script = textwrap.dedent(f"""
        import os, sys, json
        sys.path.insert(0, {str(ROOT)!r})
        import numpy as np, torch, torch.distributed as dist
        from realtime_video_analytics_32streams_amd import synth
        from realtime_video_analytics_32streams_amd.dist import IdSync, init_from_env, shard_streams, exclusive_id_bases
        from oracle import oracle as orc
        rank, world, _ = init_from_env("gloo")
        S, T = 8, 15
        script = synth.make_tracker_script(31, S, T, n_obj=6)
        mine = list(shard_streams(S, rank, world))
        sync = IdSync(len(mine), torch.device("cpu"))
        # local trackers hand out provisional ids from a private counter; we re-map like k4_assign_ids does
        loc = orc.Tracker(len(mine), 10, 0.5, 1)
        next_id = 1
        final = {{}}   # (stream, provisional id) -> final id
        out = []
        for t in range(T):
            counts, news = [], []
            for i, s in enumerate(mine):
                before = loc.next_id
                r = loc.update(i, script[t][s].boxes, script[t][s].conf, script[t][s].cls)
                counts.append(loc.last_new); news.append((before, loc.last_new, r))
            allc = sync.all_gather_counts(torch.tensor(counts, dtype=torch.int32)).tolist()
            bases = exclusive_id_bases(allc, next_id)
            next_id += sum(allc)
            for i, s in enumerate(mine):
                before, n_new, r = news[i]
                for k in range(n_new):
                    final[(i, before + k)] = bases[rank * len(mine) + i] + k
                out.append([t, s, [final[(i, int(v))] for v in r["id"]]])
        json.dump(out, open(os.path.join({str(tmp_path)!r}, f"ids_{{rank}}.json"), "w"))
        dist.barrier(); dist.destroy_process_group()
    """)
    sp = tmp_path / "worker.py"
    sp.write_text(script)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(sp)], env=env))
    for p in procs:
        assert p.wait(timeout=240) == 0
    import json
    import numpy as np
    from oracle import oracle as orc
    from realtime_video_analytics_32streams_amd import synth
    script_ = synth.make_tracker_script(31, 8, 15, n_obj=6)
    ref = orc.Tracker(8, 10, 0.5, 1)
    want = {}
    for t in range(15):
        for s in range(8):
            r = ref.update(s, script_[t][s].boxes, script_[t][s].conf, script_[t][s].cls)
            want[(t, s)] = [int(v) for v in r["id"]]
    got = {}
    for r in range(2):
        for t, s, ids in json.load(open(tmp_path / f"ids_{r}.json")):
            got[(t, s)] = ids
    assert got == want


def test_bench_refuses_to_run_fewer_ranks_than_asked_for():
    """`python bench.py --gpus N` starts N ranks itself; with fewer visible devices it fails loudly instead of measuring
    one GPU (this container has none)."""
    import os
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs are visible")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "RVA_SHARE_GPU")}
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode != 0 and "refusing to run fewer ranks" in p.stderr and not p.stdout.strip()
