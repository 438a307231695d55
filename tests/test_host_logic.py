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


def test_capture_loop_matches_reference_recording():
    """_BaseStream.frames(): retry / back-off / reconnect / pacing against the event traces recorded from the reference's
    own VideoStream.frames() (tests/golden/capture_loop.json; oracle/gen_golden.py drives it with a scripted capture)."""
    import asyncio
    from realtime_video_analytics_32streams_amd.video_stream import _BaseStream

    class Stop(Exception):
        pass

    for case in load_golden("capture_loop.json"):
        events, opens, reads = [], list(case["opens"]), list(case["reads"])

        def rec(*e):
            events.append(list(e))
            if len(events) >= case["cap"]:
                raise Stop()

        class Scripted(_BaseStream):
            def open_sync(self):
                ok = bool(opens.pop(0)) if opens else False
                rec("open", int(ok))
                if not ok:
                    raise RuntimeError(f"Unable to open stream {self.config.name}")
                super().open_sync()

            def close_sync(self):
                rec("close")
                super().close_sync()

            def next_surface(self):
                if not reads:
                    raise Stop()
                return "frame" if reads.pop(0) == "1" else None

        async def drive():
            st = Scripted(C.StreamConfig(name="cam", url="rtsp://x", **case["cfg"]))

            async def fake_sleep(t):
                rec("sleep", float(t))
            st._sleep = fake_sleep
            try:
                async for pkt in st.frames():
                    rec("frame", pkt.frame_id)
                return "ended"
            except Stop:
                return "stopped"
            except RuntimeError as exc:
                return f"RuntimeError: {exc}"
        result = asyncio.run(drive())
        assert events == case["events"], case["name"]
        assert result == case["result"], case["name"]


def test_mp4_demuxer_on_the_reference_sample():
    """MP4 -> Annex-B on the one real bitstream that exists offline (the reference's data/samples/demo.mp4; read in place,
    never copied; absent on the GPU box -> skipped there).  Facts from SURVEY.md section 2 #23: H.264 High@L3.0, 640x360,
    400 frames at 30000/1001 fps."""
    from realtime_video_analytics_32streams_amd import mp4
    sample = Path("/root/reference/data/samples/demo.mp4")
    if not sample.exists():
        pytest.skip("reference sample not present on this machine")
    with mp4.Mp4Demuxer(sample) as d:
        info = d.describe()
        assert (info["codec"], info["entry"], info["width"], info["height"], info["samples"]) == ("h264", "avc1", 640, 360, 400)
        assert info["fps"] == (30000, 1001) and info["nal_length_size"] == 4
        sps = info["sps"]
        assert (sps["profile_idc"], sps["level_idc"], sps["width"], sps["height"]) == (100, 30, 640, 360)
        assert (sps["coded_width"], sps["coded_height"]) == (640, 368)          # 23 macroblock rows, 8 cropped
        n_idr = n_bytes = 0
        last_pts = -1
        for i, (au, pts, sync) in enumerate(d.access_units()):
            types = [n[0] & 0x1F for n in mp4.iter_annexb_nals(au)]
            assert au.startswith(mp4.START_CODE) and types, i
            if sync:
                assert types[:2] == [7, 8] and 5 in types, (i, types)              # SPS, PPS re-inserted in front of the IDR
                n_idr += 1
            else:
                assert 5 not in types and 7 not in types
            assert all(t in (1, 5, 6, 7, 8, 9) for t in types)
            n_bytes += len(au)
            last_pts = max(last_pts, pts)
        assert i == 399 and n_idr == len(d.track.sync) >= 1
        assert n_bytes == sum(d.track.sizes) + n_idr * len(d.parameter_sets_annexb())  # 4-byte lengths -> 4-byte start codes
        assert abs(last_pts / 1e7 - 399 * 1001 / 30000) < 0.2                       # ~13.3 s of 10 MHz timestamps


def test_mp4_demuxer_synthetic_hevc_and_errors():
    """A hand-built MP4 (hvc1, 2-byte NAL lengths, co64, two chunks) exercises the paths demo.mp4 does not."""
    import struct
    from realtime_video_analytics_32streams_amd import mp4

    def box(t, payload):
        return struct.pack(">I4s", 8 + len(payload), t) + payload

    def full(t, payload, ver=0):
        return box(t, bytes([ver, 0, 0, 0]) + payload)
    vps, sps, pps = b"\x40\x01\xaa", b"\x42\x01\xbb\xcc", b"\x44\x01\xdd"
    arrays = b"".join(bytes([0x80 | typ]) + struct.pack(">HH", 1, len(ps)) + ps for typ, ps in ((32, vps), (33, sps), (34, pps)))
    hvcc = box(b"hvcC", bytes(21) + bytes([0xFC | 1, 3]) + arrays)                  # lengthSizeMinusOne = 1
    entry = struct.pack(">I4s", 8 + 78 + len(hvcc), b"hvc1") + bytes(24) + struct.pack(">HH", 1920, 1080) + bytes(50) + hvcc
    nal = lambda t, n: struct.pack(">H", n) + bytes([t << 1, 1]) + bytes(range(n - 2))   # noqa: E731
    samples = [nal(19, 9) + nal(39, 4), nal(1, 6), nal(1, 5)]
    mdat_payload = b"".join(samples)
    stbl = box(b"stbl", full(b"stsd", struct.pack(">I", 1) + entry) + full(b"stts", struct.pack(">III", 1, 3, 3000)) +
               full(b"stsc", struct.pack(">IIIIIII", 2, 1, 2, 1, 2, 1, 1)) +
               full(b"stsz", struct.pack(">II", 0, 3) + struct.pack(">III", *map(len, samples))) +
               full(b"stss", struct.pack(">II", 1, 1)) + full(b"co64", struct.pack(">IQQ", 2, 0, 0)))
    mdia = box(b"mdia", full(b"mdhd", struct.pack(">IIII", 0, 0, 90000, 9000) + bytes(4)) +
               full(b"hdlr", bytes(4) + b"vide" + bytes(13)) + box(b"minf", stbl))
    moov = box(b"moov", box(b"trak", mdia))
    head = box(b"ftyp", b"isom" + bytes(4)) + moov
    base = len(head) + 8
    offs = struct.pack(">IQQ", 2, base, base + len(samples[0]) + len(samples[1]))
    blob = (head + box(b"mdat", mdat_payload)).replace(struct.pack(">IQQ", 2, 0, 0), offs)
    d = mp4.Mp4Demuxer(blob)
    assert d.track.codec == "hevc" and d.track.nal_length_size == 2 and d.track.n_samples == 3 and d.track.fps == (30, 1)
    aus = list(d.access_units())
    assert [s for _, _, s in aus] == [True, False, False]
    assert aus[0][0] == b"".join(mp4.START_CODE + p for p in (vps, sps, pps)) + mp4.START_CODE + samples[0][2:11] + mp4.START_CODE + samples[0][13:]
    assert aus[2][0] == mp4.START_CODE + samples[2][2:] and aus[2][1] == 6000 * 10_000_000 // 90000
    with pytest.raises(mp4.Mp4Error):
        mp4.Mp4Demuxer(box(b"ftyp", b"isom") + box(b"moov", b""))
    with pytest.raises(mp4.Mp4Error):
        mp4.length_prefixed_to_annexb(b"\x00\x00\x00\x09abc", 4)


def test_rocdecode_stream_fails_like_an_unopenable_capture():
    """Without librocdecode (this image, and the GPU boxes) open() raises the reference's 'Unable to open stream' RuntimeError
    and names the cause; the raw Annex-B splitter is host code and is checked on demo.mp4's own elementary stream."""
    import asyncio
    from realtime_video_analytics_32streams_amd import _native as N
    from realtime_video_analytics_32streams_amd import mp4
    from realtime_video_analytics_32streams_amd.video_stream import RocDecodeStream, _annexb_access_units, open_stream, rocdecode_status
    st = open_stream(C.StreamConfig(name="door", url="/nonexistent/clip.mp4", warmup_seconds=0.0))
    assert isinstance(st, RocDecodeStream)
    if not rocdecode_status().startswith("available"):
        with pytest.raises(RuntimeError, match="Unable to open stream door: rocDecode unavailable"):
            asyncio.run(st.open())
    sample = Path("/root/reference/data/samples/demo.mp4")
    if sample.exists():
        with mp4.Mp4Demuxer(sample) as d:
            es = b"".join(au for au, _, _ in d.access_units())          # the file's elementary stream as one Annex-B blob
            want = [au for au, _, _ in d.access_units()]
        got = [au for au, _, _ in _annexb_access_units(es, N.RVA_CODEC_H264)]
        assert len(got) == 400 and got == want                          # same access-unit boundaries as the container's samples


def test_ultralytics_named_state_dict_loads(tmp_path):
    """model_path may name a state dict in the original YOLOv8 project's key naming (model.N.*, head = model.22.cv2/cv3/dfl):
    a synthetic dict in that naming -- this module's seeded weights renamed backwards -- must load to the same network."""
    import torch
    from realtime_video_analytics_32streams_amd import yolov8 as Y
    src = Y.build_detector_net("n", seed=4)
    inv = {v: k for k, v in Y._ULTRALYTICS_LAYERS.items()}
    theirs = {}
    for k, v in src.state_dict().items():
        head, _, rest = k.partition(".")
        if head == "detect":
            branch, _, tail = rest.partition(".")
            theirs[f"model.22.{'cv2' if branch == 'box' else 'cv3'}.{tail}"] = v
        else:
            theirs[f"model.{inv[head]}.{rest}"] = v
    theirs["model.22.dfl.conv.weight"] = torch.arange(16.0).view(1, 16, 1, 1)
    assert "model.2.m.0.cv1.conv.weight" in theirs and "model.22.cv3.2.2.bias" in theirs and "model.9.cv2.bn.running_var" in theirs
    f = tmp_path / "yolov8n-state.pt"
    torch.save(theirs, f)
    dst = Y.build_detector_net("n", seed=99, weights=str(f))          # goes through torch.load(weights_only=True)
    a, b = src.state_dict(), dst.state_dict()
    assert all(torch.equal(a[k], b[k]) for k in a if not k.endswith("num_batches_tracked"))
    x = torch.rand(1, 3, 64, 64)
    with torch.inference_mode():
        assert torch.equal(src(x), dst(x))
    inner = {k[len("model."):]: v for k, v in theirs.items()}           # the inner nn.Sequential's naming works too
    Y.load_detector_state_dict(Y.build_detector_net("n", seed=5), inner)
    with pytest.raises(ValueError, match="does not fit"):
        Y.load_detector_state_dict(Y.build_detector_net("s", seed=5), theirs)      # an n dict into an s network
    bad = dict(theirs); bad["model.22.dfl.conv.weight"] = torch.ones(1, 16, 1, 1)
    with pytest.raises(ValueError, match="dfl"):
        Y.load_detector_state_dict(Y.build_detector_net("n", seed=5), bad)
    del bad["model.22.dfl.conv.weight"], bad["model.0.conv.weight"]
    with pytest.raises(KeyError, match="lacks"):
        Y.load_detector_state_dict(Y.build_detector_net("n", seed=5), bad)


REF_CONFIGS = Path("/root/reference/config")
REF_TEMPORAL = Path("/root/reference/sample-temporal-pipeline.yaml")


@pytest.mark.skipif(not REF_CONFIGS.is_dir(), reason="the reference tree is not on this box (GPU box)")
def test_reference_yaml_files_load_in_place():
    """Every configuration file the reference ships (config/*.yaml, sample-temporal-pipeline.yaml) through ``load_config``, read
    IN PLACE: each stream / detector / tracker field equals what the YAML says (known keys), defaults fill the rest, unknown
    keys are dropped.  BASELINE configs[0] is config/pipeline-sim.yaml: its values are spelled out below
    (/root/reference/config/pipeline-sim.yaml:6-30)."""
    import dataclasses
    files = sorted(REF_CONFIGS.glob("*.yaml")) + ([REF_TEMPORAL] if REF_TEMPORAL.is_file() else [])
    assert len(files) >= 7
    for f in files:
        raw = yaml.safe_load(f.read_text())
        cfg = C.load_config(f)
        assert [s.name for s in cfg.streams] == [s["name"] for s in raw["streams"]], f.name
        for s, rs in zip(cfg.streams, raw["streams"]):
            for fld in dataclasses.fields(C.StreamConfig):
                if fld.name in rs:
                    assert getattr(s, fld.name) == rs[fld.name], (f.name, s.name, fld.name)
            assert cfg.detector_for(s) is (cfg.detectors[s.detector_id] if s.detector_id else cfg.detector)
        for name, rd in [("", raw.get("detector") or {})] + list((raw.get("detectors") or {}).items()):
            d = cfg.detectors[name] if name else cfg.detector
            for fld in dataclasses.fields(C.DetectorConfig):
                if fld.name in rd:
                    assert getattr(d, fld.name) == rd[fld.name], (f.name, name, fld.name)
        for fld in dataclasses.fields(C.TrackerConfig):
            if fld.name in (raw.get("tracker") or {}):
                assert getattr(cfg.tracker, fld.name) == raw["tracker"][fld.name], (f.name, fld.name)
    sim = C.load_config(REF_CONFIGS / "pipeline-sim.yaml")
    assert len(sim.streams) == 1 and sim.max_concurrent_streams == 4 and sim.stats_interval_seconds == 10
    s = sim.streams[0]
    assert (s.name, s.url, s.enabled, s.target_fps, s.batch_size, s.warmup_seconds, s.reconnect_backoff) == \
        ("sim-1", "/app/data/samples/demo.mp4", True, 12, 1, 0.5, 2.0)
    d = sim.detector
    assert (d.model_path, d.device, d.backend, d.confidence_threshold, d.iou_threshold, d.half, d.warmup) == \
        ("/app/models/yolo/yolov8n.pt", "cpu", "ultralytics", 0.35, 0.5, False, False)
    assert (sim.tracker.type, sim.tracker.max_age, sim.tracker.max_iou_distance, sim.tracker.min_hits) == ("byte_track", 30, 0.5, 1)
    if REF_TEMPORAL.is_file():
        t = C.load_config(REF_TEMPORAL)
        cl = t.detectors["temporal_cnn_lstm"]
        assert (cl.model_type, cl.sequence_length, cl.sequence_stride, cl.temporal_overlap, cl.half) == ("cnn_lstm", 16, 2, 0.5, False)
        assert list(cl.input_size) == [224, 224] and cl.num_action_classes == 400


@pytest.mark.skipif(not REF_CONFIGS.is_dir(), reason="the reference tree is not on this box (GPU box)")
def test_reference_config_routed_to_this_library_fails_with_the_documented_errors():
    """configs[0] as the reference ships it names the reference's own backend and a container path.  Routed to this library
    unchanged: ``backend: ultralytics`` -> the 'belongs to the reference implementation' RuntimeError of create_detector
    (detector.py:496-502: a backend whose runtime is missing raises); the stream ``/app/data/samples/demo.mp4`` -> 'Unable to
    open stream' (video_stream.py:78-79), whether because the decoder library is absent or because the file is."""
    import dataclasses
    from realtime_video_analytics_32streams_amd.detector import create_detector
    from realtime_video_analytics_32streams_amd.video_stream import RocDecodeStream
    sim = C.load_config(REF_CONFIGS / "pipeline-sim.yaml")
    with pytest.raises(RuntimeError, match="belongs to the reference implementation"):
        create_detector(sim.detector)
    with pytest.raises(RuntimeError, match="Unable to open stream sim-1"):
        RocDecodeStream(sim.streams[0]).open_sync()
    # the one edit a maintainer makes: backend -> hip.  Without a GPU that fails loudly too -- never a CPU path.
    hip = dataclasses.replace(sim.detector, backend="hip")
    hip.validate()
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            create_detector(hip)


def test_id_sync_dead_rank_makes_the_survivor_exit_nonzero(tmp_path):
    """Failure path of the id exchange (dist.py): rank 1 dies after three exchanges; rank 0's next all-gather must not block for
    ever -- it exits with dist.EXIT_PEER_LOST within the process-group timeout (RVA_DIST_TIMEOUT_S), and says why on stderr."""
    script = textwrap.dedent(f"""
        import os, sys, time
        sys.path.insert(0, {str(ROOT)!r})
        import torch
        from realtime_video_analytics_32streams_amd.dist import IdSync, init_from_env
        rank, world, _ = init_from_env("gloo")
        sync = IdSync(4, torch.device("cpu"))
        for t in range(1000):
            if rank == 1 and t == 3:
                os._exit(0)                      # the peer vanishes without a goodbye
            sync.all_gather_counts(torch.full((4,), t, dtype=torch.int32))
        print("unreachable: the exchange kept going without its peer")
    """)
    sp = tmp_path / "worker.py"
    sp.write_text(script)
    port = _free_port()
    procs = []
    import time
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   RVA_DIST_TIMEOUT_S="8")
        procs.append(subprocess.Popen([sys.executable, str(sp)], env=env, stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True))
    t0 = time.monotonic()
    try:
        out0, err0 = procs[0].communicate(timeout=90)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    from realtime_video_analytics_32streams_amd import dist as rdist
    assert procs[1].wait(timeout=10) == 0
    assert procs[0].returncode == rdist.EXIT_PEER_LOST, (procs[0].returncode, err0[-400:])
    assert "unreachable" not in out0 and "id exchange failed" in err0 and "exiting with status" in err0
    assert time.monotonic() - t0 < 60
