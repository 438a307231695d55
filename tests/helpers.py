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


def temporal_net(case):
    """This package's network for a G6 case, rebuilt from the seed; its state dict must hash to what the REFERENCE's
    module (scripts/convert_temporal_model_to_onnx.py:34-121) held when the logits were recorded."""
    from realtime_video_analytics_32streams_amd import synth
    from realtime_video_analytics_32streams_amd.temporal import Cnn3dNet, CnnLstmNet
    kw = case["ctor"]
    if case["kind"] == "cnn_lstm":
        ctor = lambda: CnnLstmNet(kw["num_classes"], kw["hidden_size"])      # noqa: E731
    else:
        ctor = lambda: Cnn3dNet(kw["num_classes"])                           # noqa: E731
    net = synth.seeded_module(ctor, case["seed"])
    assert synth.state_sha(net.state_dict()) == case["state_sha"], "torch drift: rebuilt weights differ from the reference's"
    x = synth.seeded_clip(case["clip_shape"], case["clip_seed"])
    assert synth.sha256_of(x.numpy()) == case["clip_sha"]
    return net, x
