"""Shared helpers for the parity tests (rebuild golden inputs, serialise tables)."""
from __future__ import annotations

import hashlib
import json

import numpy as np

from realtime_video_analytics_32streams_amd import synth


def head_for_case(case) -> np.ndarray:
    """Return the raw 2-D head exactly as the golden generator fed it to the reference."""
    if case["kind"] == "seeded":
        gen = dict(case["gen"])
        if "content" in gen:
            gen["content"] = tuple(gen["content"])
        head = synth.make_head(case["seed"], **gen)  # [A, C]
        assert synth.sha256_of(head) == case["sha"], "numpy drift: regenerated input differs from golden input"
        return np.ascontiguousarray(head.T) if case["layout"] == "CA" else head
    return np.asarray(case["data"], np.float32).reshape(case["shape"])


def table_digest(table) -> str:
    return hashlib.sha256(json.dumps(table).encode()).hexdigest()[:16]


def script_sha(script) -> str:
    sha = hashlib.sha256()
    for row in script:
        for fd in row:
            sha.update(fd.boxes.tobytes()); sha.update(fd.conf.tobytes()); sha.update(fd.cls.tobytes())
    return sha.hexdigest()
