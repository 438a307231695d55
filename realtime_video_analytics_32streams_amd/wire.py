"""Event wire format of the hot path's output.

The reference publishes one JSON message per processed frame to Kafka
(sinks/kafka_sink.py:103-132, serialised with ``json.dumps(value).encode("utf-8")`` at :85) and the
dashboard consumer parses it back (api/kafka_consumer.py:112-128, schema api/schemas.py:12-35).
This module produces byte-identical messages from this library's ``Track`` objects -- or straight
from a device-table snapshot, without materialising ``Track`` objects -- so the existing dashboard
can consume the accelerated path unchanged.  Transport (aiokafka) and the optional JPEG preview stay
out of scope.
"""
from __future__ import annotations

import json
from typing import Iterable, List, Optional, Sequence


def tracks_payload(stream_name: str, frame_id: int, tracks: Iterable) -> dict:
    """kafka_sink.py:103-132: temporal keys are added only when not None; ``is_temporal`` is true iff some
    track carries an ``action_label``."""
    track_list: List[dict] = []
    has_temporal = False
    for t in tracks:
        d = {"track_id": t.track_id, "class_id": t.class_id, "confidence": t.confidence, "bbox_xyxy": t.bbox_xyxy}
        if getattr(t, "action_label", None) is not None:
            d["action_label"] = t.action_label
            has_temporal = True
        if getattr(t, "temporal_score", None) is not None:
            d["temporal_score"] = t.temporal_score
        if getattr(t, "sequence_start_frame", None) is not None:
            d["sequence_start_frame"] = t.sequence_start_frame
        if getattr(t, "sequence_end_frame", None) is not None:
            d["sequence_end_frame"] = t.sequence_end_frame
        track_list.append(d)
    return {"stream": stream_name, "frame_id": frame_id, "tracks": track_list, "is_temporal": has_temporal}


def payload_from_table(stream_name: str, frame_id: int, table: dict) -> dict:
    """Same message from one stream's snapshot (``ops.DeviceTracker.read*`` dict): no Track objects."""
    n = int(table["n"])
    ids, cls, conf, box = table["id"], table["cls"], table["conf"], table["boxes"]
    return {"stream": stream_name, "frame_id": frame_id,
            "tracks": [{"track_id": int(ids[i]), "class_id": int(cls[i]), "confidence": float(conf[i]),
                        "bbox_xyxy": [float(v) for v in box[i]]} for i in range(n)],
            "is_temporal": False}


def serialize(payload: dict) -> bytes:
    """The producer's value_serializer (kafka_sink.py:85)."""
    return json.dumps(payload).encode("utf-8")


def parse_event(raw) -> dict:
    """Consumer side (api/kafka_consumer.py:112-128): keeps the four base track fields, defaults for the rest;
    raises ValueError where the pydantic schema would (confidence outside [0, 1], bbox not 4 long)."""
    p = json.loads(raw)
    tracks = []
    for t in p.get("tracks", []):
        conf, box = float(t["confidence"]), [float(v) for v in t["bbox_xyxy"]]
        if not (0.0 <= conf <= 1.0):
            raise ValueError("confidence must be in [0, 1] (api/schemas.py:15)")
        if len(box) != 4:
            raise ValueError("bbox_xyxy must have 4 items (api/schemas.py:16)")
        tracks.append({"track_id": int(t["track_id"]), "class_id": int(t["class_id"]), "confidence": conf, "bbox_xyxy": box})
    return {"stream": p.get("stream", "unknown"), "frame_id": p.get("frame_id", 0), "tracks": tracks,
            "frame_jpeg": p.get("frame_jpeg"), "is_temporal": False}
