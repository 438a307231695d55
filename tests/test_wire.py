"""Wire format against messages produced by the reference's own KafkaSink.send_tracks."""
import json

from realtime_video_analytics_32streams_amd import wire
from realtime_video_analytics_32streams_amd.tracker import Track
from tests.conftest import load_golden


def _tracks(case):
    tr = []
    for tid, cls, age, hits, conf, box in case["table"]:
        t = Track(track_id=tid, class_id=cls, confidence=conf, bbox_xyxy=tuple(box), age=age, hits=hits)
        for k, v in case.get("temporal", {}).items():
            setattr(t, k, v)
        tr.append(t)
    return tr


def test_messages_are_byte_identical_to_the_reference():
    for case in load_golden("wire_cases.json"):
        payload = wire.tracks_payload(case["stream"], case["frame_id"], _tracks(case))
        assert wire.serialize(payload) == case["message"].encode("utf-8"), case["stream"]


def test_payload_from_table_equals_track_path():
    import numpy as np
    for case in load_golden("wire_cases.json"):
        if "temporal" in case:
            continue
        tab = dict(n=len(case["table"]), id=np.array([r[0] for r in case["table"]], np.int64),
                   cls=np.array([r[1] for r in case["table"]], np.int32), conf=np.array([r[4] for r in case["table"]], np.float64),
                   boxes=np.array([r[5] for r in case["table"]], np.float64).reshape(-1, 4))
        assert wire.serialize(wire.payload_from_table(case["stream"], case["frame_id"], tab)) == case["message"].encode()


def test_consumer_side_parse():
    case = load_golden("wire_cases.json")[0]
    ev = wire.parse_event(case["message"])
    assert ev["stream"] == case["stream"] and ev["frame_id"] == case["frame_id"]
    assert [t["track_id"] for t in ev["tracks"]] == [r[0] for r in case["table"]]
    bad = json.loads(case["message"]); bad["tracks"] = [dict(track_id=1, class_id=0, confidence=1.5, bbox_xyxy=[0, 0, 1, 1])]
    import pytest
    with pytest.raises(ValueError):
        wire.parse_event(json.dumps(bad))      # the dashboard schema rejects confidence > 1 (raw temporal logits can be)


def test_preview_plan_and_policy_match_reference_recording():
    """8f-3 host logic against tests/golden/preview_plan.json, recorded from the reference's own KafkaSink with a call-level
    cv2 recorder: every resize / rectangle / text / encode operation of _render_frame (target size, integer coordinates,
    colours, thickness, label text and origin, encoder parameters), the adaptive quality table, the per-class colours and
    the 10-per-second rate limit."""
    from realtime_video_analytics_32streams_amd import preview as P
    from tests.conftest import load_golden
    g = load_golden("preview_plan.json")
    scripted = lambda label: ((9 * len(label), 11), 4)            # noqa: E731  the recorder's getTextSize rule
    for r in g["render"]:
        ops = P.plan_render(tuple(r["wh"]), r["tracks"], r["quality"], r["webp"], text_size=scripted)
        assert ops == r["calls"], (r["wh"], len(r["tracks"]))
        mime = "image/webp" if ops[-1][1] == ".webp" else "image/jpeg"
        assert r["url"].startswith(f"data:{mime};base64,")
    for q in g["quality"]:
        pol = P.PreviewPolicy(frame_quality=q["base"])
        assert [pol.adaptive_quality(c) for c in range(16)] == q["by_count"]
    assert [[c, list(P.color_for(c))] for c, _ in g["colors"]] == g["colors"]
    now = [0.0]
    pol = P.PreviewPolicy(clock=lambda: now[0])
    got = []
    for name, t, _ in g["rate"]:
        now[0] = t
        got.append([name, t, pol.should_send_frame(name)])
    assert got == g["rate"]


def test_snapshot_plan_and_interval_match_reference_recording():
    """8f-3, the five-minute snapshot: tests/golden/snapshot_plan.json is StreamWorker._maybe_save_snapshot (pipeline.py:264-290)
    recorded at the cv2 / Path call level under a scripted clock -- when a snapshot is taken (300 s per stream, first frame
    always), the copy, every outline and label with integer coordinates, colour and thickness, the directory and the file name."""
    from types import SimpleNamespace
    from realtime_video_analytics_32streams_amd import preview as P
    from tests.conftest import load_golden
    g = load_golden("snapshot_plan.json")
    assert sum(1 for c in g if c["calls"]) >= 3 and sum(1 for c in g if not c["calls"]) >= 3
    now = [0.0]
    wr = P.SnapshotWriter(clock=lambda: now[0])
    for c in g:
        now[0] = c["now"]
        at = wr.due(c["stream"])
        assert (at is not None) == bool(c["calls"]), c["now"]
        if at is None:
            continue
        as_objects = [SimpleNamespace(**{k: v for k, v in t.items() if not (k == "track_id" and v is None)}) for t in c["tracks"]]
        for tracks in (c["tracks"], as_objects):                   # wire dicts and Track / Detection objects alike
            assert P.plan_snapshot(c["stream"], c["frame_id"], at, tuple(c["wh"]), tracks) == c["calls"], c["now"]
    assert wr.due("another-stream") is not None                    # the interval is per stream


def test_preview_encoder_roundtrip_on_host():
    """The host encoder (Pillow standing in for cv2.imencode): a data URL whose payload decodes back to the image."""
    import base64, io
    import numpy as np
    from PIL import Image
    from realtime_video_analytics_32streams_amd import preview as P
    rng = np.random.default_rng(3)
    img = np.zeros((72, 128, 3), np.uint8)
    img[..., 0] = np.linspace(0, 255, 128)[None, :]; img[..., 1] = np.linspace(0, 255, 72)[:, None]; img[20:40, 30:90, 2] = 200
    for ext, params in ((".jpg", [P.IMWRITE_JPEG_QUALITY, 90, P.IMWRITE_JPEG_PROGRESSIVE, 1, P.IMWRITE_JPEG_OPTIMIZE, 1]),
                        (".webp", [P.IMWRITE_WEBP_QUALITY, 90])):
        data, mime = P.encode_image(img, ext, params)
        back = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))[..., ::-1]
        assert back.shape == img.shape and np.abs(back.astype(int) - img.astype(int)).mean() < 6.0, (ext, mime)
