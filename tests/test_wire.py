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
