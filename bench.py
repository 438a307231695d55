#!/usr/bin/env python3
"""Headline benchmark: aggregate detected FPS across streams + p99 per-frame latency, 32x1080p30.

One *step* = one tick of the hot path over one batch of synthetic input on every GPU:
32 x 1080p NV12 surfaces resident in HBM -> K1 pre-process -> YOLOv8s fp16 (librva fused MFMA plan) -> K2 decode
+ K3 NMS -> K4 tracker update + id assignment -> tracks read back to the host (BASELINE.json
configs[2], the headline single-GPU configuration).  With --gpus N every rank runs 32 streams of its
own (weak scaling) and the ranks exchange one 128-byte all-gather of new-track counts per tick over
RCCL so that track ids stay globally consistent.

Prints ONE JSON line on rank 0 (contract in the task statement) including
  "roofline"     the steady-state K1 pre-process kernel against the HBM roofline (2,764,800 algorithmic B/frame: the
                 source rows it reads + the 360 content rows it writes; the first launch into a buffer also writes the
                 constant letterbox border: 3,840,000 B/frame, reported beside it as "full_tensor_kernel"),
  "cpu_baseline" the CPU oracle (oracle/, kind "port") + torch-CPU fp32 network on a bounded sample.
Decode is NOT part of the step: librocdecode and an H.265 source are absent (reported as such).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

# The HIP runtime spreads streams over GPU_MAX_HW_QUEUES hardware queues (4 by default) in order of first use; two streams that
# share a queue serialise.  The pipeline needs its two tick chains on different queues: with 8 queues the measured throughput is
# the same as with 4 (19.9 k frames/s) and further streams in the process (RCCL's, under torch.distributed) cannot push the two
# chains onto one queue as easily.  Must be set before the runtime initialises; an explicit setting of the caller wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

K1_BYTES_FULL = 3_840_000               # 360 Y rows + 360 UV rows x 1920 B in, 3x640x640 fp16 out (border included)
K1_BYTES_PER_FRAME = 2_764_800          # SURVEY.md 8(d) alternate, what the steady-state kernel moves: the same input rows
                                        # + the 360 content rows of the tensor (the constant border is written once per buffer)
HBM_PEAK_GBS = 8000.0                   # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_PEAK_TFLOPS = 2500.0               # dense fp16


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--streams", type=int, default=32, help="streams per GPU (weak scaling)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: --streams per GPU on every rank (default; 32 = BASELINE configs[2] per GPU).  strong: "
                         "--total-streams sharded evenly over the ranks (32 over 8 GPUs = 4 per GPU = BASELINE configs[3] "
                         "with --model m)")
    ap.add_argument("--total-streams", type=int, default=32, help="streams of the whole job (strong scaling)")
    ap.add_argument("--workload", default="detect", choices=["detect", "temporal"],
                    help="detect: the headline (BASELINE configs[1..3]).  temporal: BASELINE configs[4] -- 3840x2160 NV12 streams, "
                         "CNN-LSTM over 16-frame clips (stride 2, overlap 0.5, 224x224, fp32 as sample-temporal-pipeline.yaml "
                         "says), every stream through the shared tracker; --streams per GPU (default 8)")
    ap.add_argument("--model", default="s", choices=["n", "s", "m"])
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--conf", type=float, default=0.25)
    ap.add_argument("--iou", type=float, default=0.45)
    ap.add_argument("--target-dets", type=int, default=120, help="calibrated candidates per frame (synthetic weights)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the measurements taken after the timed region (cold K1 launch, device copy bandwidth, "
                         "post-process / tracker load sweep)")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the two child runs (4 x YOLOv8m, 4 x YOLOv8n) that the default headline run appends to its record")
    ap.add_argument("--cpu-frames", type=int, default=0, help="frames in the CPU-baseline sample (0 = auto)")
    ap.add_argument("--miopen-find", action="store_true", help="torch.backends.cudnn.benchmark = True")
    ap.add_argument("--no-graph", action="store_true",
                    help="launch every kernel eagerly instead of replaying the captured hipGraph of the post-process / tracker tail")
    ap.add_argument("--net-graph", nargs="?", const="on", default="auto", choices=["auto", "on", "off"],
                    help="how the detector network of a tick is launched: 'off' = ~75-110 eager launches per tick (~0.5 ms of host "
                         "time), 'on' = one hipGraph replay (~0.15 ms; the graph executor costs the GPU 2-5 %%), 'auto' (default) = "
                         "both are timed during warm-up and the faster one runs the timed region (the graph wins where the host is "
                         "the limit: YOLOv8n x 4 streams)")
    ap.add_argument("--depth", type=int, default=None, choices=[1, 2, 3, 4, 5, 6, 7, 8],
                    help="ticks in flight, each a chain on its own HIP stream.  Default on one GPU: 4 -- the chains run on streams chosen by "
                         "a start-up probe so that no two share a hardware lane (ops.chain_streams); measured on one box: 32 x YOLOv8s 21.35 k "
                         "(3) -> 21.85 k (4) frames/s at p99 4.9 -> 6.5 ms, 4 x YOLOv8m 5 450 -> 6 040, 4 x YOLOv8n 13.6 k -> 15.2 k; five "
                         "chains are slower than three everywhere.  1 = strictly synchronous ticks (lowest latency).  With sharded streams "
                         "(--gpus N > 1) the default is 2: no 8-GPU record shows yet which lane RCCL's collective takes "
                         "(profiles/r03_experiments_not_kept.txt #14) -- pass --depth 3 or 4 to compare")
    return ap.parse_args()


K1_SAMPLE_EVERY = 4      # ticks between dispatch-level timings of K1 inside the timed region


def k1_bytes(w, h, content_only=True, dst=640):
    """Algorithmic bytes of one frame through K1: the source rows the resize taps touch (Y + UV, whole
    rows) + the tensor rows written (content rows only in steady state, the whole 3 x dst x dst tensor otherwise)."""
    scale = min(dst / w, dst / h)
    nw, nh = int(w * scale), int(h * scale)
    if w % nw == 0 and h % nh == 0 and w // nw == h // nh:
        r = w // nw
        ys = {r * y + (r - 1) // 2 for y in range(nh)} if r % 2 else {r * y + r // 2 - 1 + k for y in range(nh) for k in (0, 1)}
    else:
        ys = set(range(h))                     # fractional geometry: two taps per output row, in practice every source row
    rows_in = len(ys) + len({y >> 1 for y in ys})
    out = 3 * (nh * nw if content_only else dst * dst) * 2
    return rows_in * w + out


def input_ring_frames(streams, w, h, cache_bytes=256 << 20):
    """Frames per stream in the synthetic input ring: one tick's surfaces (streams x pitch x 1.5 h bytes) must not survive to
    their reuse in a ``cache_bytes`` last-level cache -> at least 2 x cache_bytes of other surfaces in between."""
    per_tick = streams * ((w + 255) // 256 * 256) * (h + h // 2)
    return int(min(max(2, -(-2 * cache_bytes // per_tick) + 1), 64))


def spawn_ranks(args) -> int:
    """``python bench.py --gpus N`` without a launcher: start N fresh rank processes (one per GPU) and relay rank 0's
    JSON line.  Runs BEFORE anything touches the GPU in this process (``torch.cuda.device_count()`` does not initialise
    it on this stack) and never exec()s: the children are ordinary subprocesses, the parent only waits."""
    import socket
    import subprocess
    n = args.gpus
    shared = os.environ.get("RVA_SHARE_GPU") == "1"          # rehearsal on a one-GPU box: every rank uses device 0 (gloo)
    visible = torch.cuda.device_count()
    if not shared and visible < n:
        raise SystemExit(f"bench.py --gpus {n}: only {visible} HIP device(s) visible; refusing to run fewer ranks than "
                         "asked for (set RVA_SHARE_GPU=1 to rehearse the sharded path on one device)")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve()), *sys.argv[1:]], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    # wait for all ranks; when one dies the others leave through the failure path of the id exchange (dist.fail: non-zero exit
    # within RVA_DIST_TIMEOUT_S) -- a rank still alive well after that is killed by its exact pid, never left hanging
    import threading
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    first_exit = None
    grace = float(os.environ.get("RVA_DIST_TIMEOUT_S", "60")) + 30.0
    while any(p.poll() is None for p in procs):
        if first_exit is None and any(p.poll() not in (None, 0) for p in procs):
            first_exit = time.monotonic()
        if first_exit is not None and time.monotonic() - first_exit > grace:
            for p in procs:
                if p.poll() is None:
                    p.kill()
        time.sleep(0.05)
    reader.join(timeout=10)
    codes = [p.returncode for p in procs]
    sys.stdout.write(out0[0] if out0 else "")
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        print(f"[bench] rank(s) failed: {bad}", file=sys.stderr)
        return 1
    return 0


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args))
    from realtime_video_analytics_32streams_amd import dist as rdist
    rank, world, local = rdist.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"bench.py: WORLD_SIZE={world} from the launcher but --gpus {args.gpus}; they must agree")
    if args.depth is None:
        args.depth = 4 if world == 1 else 2        # sharded default: two chains until a SCALE record shows where RCCL puts its collective (--depth 3 / 4 to compare)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP hot path has no CPU fallback")
    if os.environ.get("RVA_SHARE_GPU") != "1" and torch.cuda.device_count() <= local:
        raise SystemExit(f"bench.py: rank {rank} wants device {local} but only {torch.cuda.device_count()} are visible")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    torch.backends.cudnn.benchmark = bool(args.miopen_find)
    if args.workload == "temporal":
        return temporal_main(args, rank, world, local, dev)

    from realtime_video_analytics_32streams_amd import ops
    from realtime_video_analytics_32streams_amd.config import DetectorConfig, StreamConfig, TrackerConfig
    from realtime_video_analytics_32streams_amd.detector import HipYoloDetector
    from realtime_video_analytics_32streams_amd.pipeline import TickPipeline
    from realtime_video_analytics_32streams_amd.tracker import IouTracker
    from realtime_video_analytics_32streams_amd.video_stream import SyntheticNv12Stream, rocdecode_status
    from realtime_video_analytics_32streams_amd.yolov8 import (build_detector_net, calibrate_detection_density,
                                                               count_macs)

    if args.scaling == "strong":
        if args.total_streams % world:
            raise SystemExit(f"--scaling strong: {args.total_streams} streams do not shard evenly over {world} ranks")
        S = args.total_streams // world
    else:
        S = args.streams
    first = rank * S
    streams = [StreamConfig(name=f"cam{first + i:03d}", url=f"synthetic://{args.width}x{args.height}", target_fps=30.0,
                            warmup_seconds=0.0) for i in range(S)]
    # Input rings sized so that K1 reads HBM, not the 256 MiB Infinity Cache: a frame comes round again only after more than
    # 2 x 256 MiB of OTHER surfaces have been read (a decoder delivers new bytes every tick).  Two host-generated frames per
    # stream, the rest derived on the device (SyntheticNv12Stream.ring_frames).
    ring_frames = input_ring_frames(S, args.width, args.height)
    sources = [SyntheticNv12Stream(s, index=first + i, width=args.width, height=args.height, n_unique=2, device=dev,
                                   ring_frames=ring_frames) for i, s in enumerate(streams)]
    for src in sources:
        src.open_sync()

    dcfg = DetectorConfig(model_path=f"yolov8{args.model}.pt", backend="hip", model_type="yolov8", half=True,
                          confidence_threshold=args.conf, iou_threshold=args.iou, warmup=False)
    net_cpu = build_detector_net(args.model, seed=0)
    macs = count_macs(net_cpu)
    import copy
    det = HipYoloDetector(dcfg, net=copy.deepcopy(net_cpu), device=local)
    # synthetic weights: shift the class biases so a realistic number of anchors clears the threshold
    with torch.inference_mode():
        sample, _ = ops.preprocess_nv12([src._ring[0] for src in sources[:8]], (640, 640), half=True)
    # The class logits come from the framework's own plan (logit of the probabilities it emits for 8 sample frames), so the only
    # convolutions this process runs on the GPU are the framework's kernels and the kernel trace of a bench run shows nothing else.
    from realtime_video_analytics_32streams_amd.engine import FusedYoloV8
    with torch.inference_mode():
        probs = FusedYoloV8(det.net, sample.shape[0], device=dev, autotune=False)(sample.contiguous())[:, 4:, :].float()
        probs = probs.clamp(2.0 ** -20, 1.0 - 2.0 ** -11)
        shifts = calibrate_detection_density(det.net, None, args.conf, args.target_dets, class_logits=torch.log(probs / (1.0 - probs)))
    del probs
    det.invalidate_engine()
    tcfg = TrackerConfig(max_age=30, max_iou_distance=0.5, min_hits=1)
    trk = IouTracker(tcfg, max_streams=S, capacity=1024, device=local)
    id_sync = rdist.IdSync(S, dev) if world > 1 else None
    if world == 1 and os.environ.get("RVA_FIFTH_STREAM"):
        # experiment (profiles/r03_experiments_not_kept.txt #14): the id exchange as ProcessGroupNCCL would issue it -- a tiny kernel on
        # a stream of its own, event-ordered behind the tick's K4 and in front of its id assignment -- on a single GPU
        class _FifthStream(rdist.IdSync):
            def __init__(self, per, device, early):
                super().__init__(per, device)
                self.s = torch.cuda.Stream(device=device)
                if early:                                   # first use before the chains' streams exist
                    with torch.cuda.stream(self.s):
                        self.buf.zero_()
                    torch.cuda.synchronize()

            def _gather(self, local_counts):
                cur = torch.cuda.current_stream()
                self.s.wait_stream(cur)
                with torch.cuda.stream(self.s):
                    self.buf[: self.per].copy_(local_counts)
                cur.wait_stream(self.s)
                return self.buf
        id_sync = _FifthStream(S, dev, os.environ["RVA_FIFTH_STREAM"] == "early")
    pipe = TickPipeline(streams, det, trk, sources=sources, id_sync=id_sync, first_global_index=first,
                        n_global_streams=world * S)
    ops.context(local)
    from realtime_video_analytics_32streams_amd import _native as N
    N.lib().rva_reserve(ops.context(local).handle, S, 8400)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # ---- warm-up (untimed), through the same runner as the timed region: the first tick sizes every buffer and
    # autotunes the plan eagerly, the second one captures the hipGraphs, the rest replay them -----------------
    use_graph = not args.no_graph
    from realtime_video_analytics_32streams_amd.pipeline import PipelinedTicks
    runner = PipelinedTicks(pipe, depth=args.depth, use_graph=use_graph, net_graph=args.net_graph != "off")
    runner.replay_net = args.net_graph == "on" and runner.net_graph
    # at least 20 untimed ticks: the first one builds and tunes the plan, the second builds the odd ticks' plan, the third captures
    # the hipGraphs, and the clocks / caches of a fresh process take a few more to settle (20 timed ticks after 5 warm-up ticks
    # read 3 % lower than after 30); the timed region below is exactly --steps ticks either way
    warm_ticks = max(args.warmup, 20)
    for _ in range(warm_ticks):
        runner.submit()
        runner.collect()
    torch.cuda.synchronize()
    net_launch = {"mode": "hipGraph" if runner.replay_net else "eager", "chosen_by": "--net-graph " + args.net_graph}
    if args.net_graph == "auto" and runner.net_graph and runner._captured:
        def rate(n=40):
            done = 0
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(n):
                if k - done == runner.depth:
                    runner.collect(); done += 1
                runner.submit()
            while done < n:
                runner.collect(); done += 1
            return S * n / (time.perf_counter() - t0)
        trial = {}
        for mode in (False, True, False, True):               # interleaved: eager, graph, eager, graph
            runner.replay_net = mode
            rate(10)
            trial.setdefault(mode, []).append(rate())
        eager_fps, graph_fps = max(trial[False]), max(trial[True])
        runner.replay_net = graph_fps > eager_fps * 1.01
        net_launch = {"mode": "hipGraph" if runner.replay_net else "eager", "chosen_by": "warm-up trial (2 x 40 ticks each, untimed)",
                      "eager_frames_per_s": round(eager_fps, 1), "graph_frames_per_s": round(graph_fps, 1)}
        torch.cuda.synchronize()

    # ---- timed region: exactly K steps, per-stage HIP events on the launch stream ---------------------
    K = args.steps
    lat = np.empty(K)
    t_enq = np.empty(K)
    t_sub = np.empty(K)                                   # host time inside submit(): what enqueueing one tick costs the CPU
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(K)]
    # dispatch-level event pair per tick for K1 (hipExtLaunchKernelGGL start/stop events, set through
    # rva_profile_next_preprocess): brackets exactly the kernel.  torch creates the hipEvent at the first record().
    evk = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(K)]
    for pair in evk:
        for x in pair:
            x.record()
    torch.cuda.synchronize()
    rctx = ops.context(local)
    n_tracks = 0
    dt = trk.device_tracker

    def enqueue(k):
        t_enq[k] = time.perf_counter()
        arm = None
        if k % K1_SAMPLE_EVERY == 0:
            arm = lambda: N.lib().rva_profile_next_preprocess(rctx.handle, evk[k][0].cuda_event, evk[k][1].cuda_event)
        runner.submit(events=ev[k], before_k1=arm)
        t_sub[k] = time.perf_counter() - t_enq[k]

    def finish(k):
        _, tables = runner.collect()
        lat[k] = time.perf_counter() - t_enq[k]
        return sum(t["n"] for t in tables)

    if id_sync is not None:
        id_sync.sample_every = 4
    barrier()
    t_begin = time.perf_counter()
    done = 0
    for k in range(K):                                   # `depth` ticks in flight: collect the oldest before the (depth + 1)-th
        if k - done == runner.depth:
            n_tracks += finish(done); done += 1
        enqueue(k)
    while done < K:
        n_tracks += finish(done); done += 1
    post = runner.last_post
    torch.cuda.synchronize()
    own_elapsed = time.perf_counter() - t_begin          # this rank alone, before the closing barrier
    barrier()
    elapsed = time.perf_counter() - t_begin
    multi = None
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tt.item())
        # a record that shows by itself that N ranks ran and what the exchange costs: every rank's own frames/s, the ranks
        # and the backend the process group reports, the id exchange on the tick's chain (events around the all-gather)
        xs = id_sync.exchange_us()
        mine = torch.tensor([S * K / own_elapsed, float(np.mean(xs)) if xs else 0.0, float(np.max(xs)) if xs else 0.0,
                             float(id_sync.calls)], device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        torch.distributed.all_gather(allr, mine)
        allr = torch.stack(allr).cpu().numpy()
        multi = {"ranks_seen": int(torch.distributed.get_world_size()), "backend": str(torch.distributed.get_backend()),
                 "ticks_in_flight": runner.depth,
                 "devices_visible": int(torch.cuda.device_count()), "shared_gpu_rehearsal": os.environ.get("RVA_SHARE_GPU") == "1",
                 "per_rank_frames_per_s": {"min": round(float(allr[:, 0].min()), 1), "max": round(float(allr[:, 0].max()), 1)},
                 "id_exchange_us_per_tick": {"mean": round(float(allr[:, 1].mean()), 1), "max_over_ranks": round(float(allr[:, 2].max()), 1),
                                             "what": "HIP events around IdSync.all_gather_counts on the tick's own stream (includes the wait for the "
                                                     "slowest rank's tick), every 4th tick"},
                 "id_exchanges_per_rank": int(allr[0, 3])}
    dets_emitted = int(post.counts.sum().item())

    k1_bracket_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in ev]))            # hipEventRecord before / after the launch
    k1_ms = float(np.mean([a.elapsed_time(b) for a, b in evk[::K1_SAMPLE_EVERY]]))   # the dispatch's own start / stop events
    if not (0.0 < k1_ms <= k1_bracket_ms * 1.05):                                    # events not written: fall back
        k1_ms = k1_bracket_ms
    if use_graph:   # per-stage split of the captured part: eager passes AFTER the timed region (informational)
        # one forward pass alone, with the detect branches on side streams and in line: how side streams map onto hardware
        # queues depends on which streams the process created before (a plan taken from the autotune cache never creates
        # the tuner's streams), so both layouts are timed and the faster one is reported with its name
        stage, stage_mode = None, None
        for mode, serial in (("detect branches on side streams", "0"), ("detect branches in line", "1")):
            os.environ["RVA_SERIAL_HEADS"] = serial
            ev2 = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(20)]
            eager = PipelinedTicks(pipe, depth=1, use_graph=False, overlap=False)
            for k in range(20):
                eager.submit(events=ev2[k])
                eager.collect()
            torch.cuda.synchronize()
            st = np.array([[e[i].elapsed_time(e[i + 1]) for i in range(1, 4)] for e in ev2[4:]])
            if stage is None or st.mean(0)[0] < stage.mean(0)[0]:
                stage, stage_mode = st, mode
        os.environ.pop("RVA_SERIAL_HEADS", None)
    else:
        stage = np.array([[e[i].elapsed_time(e[i + 1]) for i in range(1, 4)] for e in ev])
        stage_mode = "timed region (no graph)"
    net_ms, post_ms, trk_ms = stage.mean(0)
    frames = world * S * K
    fps = frames / elapsed
    k1_frame_bytes = k1_bytes(args.width, args.height)
    assert (args.width, args.height) != (1920, 1080) or k1_frame_bytes == K1_BYTES_PER_FRAME
    k1_gbs = k1_frame_bytes * S / (k1_ms * 1e-3) / 1e9
    # HBM traffic of the K1 launch from rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, see the file): a
    # committed measurement of exactly this launch shape, not something bench.py can collect while timing
    k1_traffic = None
    pmc = ROOT / "profiles" / "r04_k1_pmc.json"
    if pmc.exists() and S == 32 and (args.width, args.height) == (1920, 1080):
        k1_traffic = json.loads(pmc.read_text())["content_1080p"]["traffic_bytes_per_launch"]
    net_tflops = 2 * macs * S / (net_ms * 1e-3) / 1e12

    if world > 1:
        workload = (f"{world * S}x{args.width}x{args.height} NV12 streams sharded {S} per GPU over {world} GPUs "
                    f"({args.scaling} scaling), resident in HBM, YOLOv8{args.model} fp16 batch={S} per GPU, IoU tracker, "
                    "global track ids via one RCCL all-gather per tick")
    else:
        workload = (f"{S}x{args.width}x{args.height} NV12 streams resident in HBM, YOLOv8{args.model} fp16 batch={S}, "
                    f"IoU tracker ({'BASELINE configs[2]' if (S, args.model) == (32, 's') else args.scaling + ' scaling leg'})")
    out = {
        "metric": "aggregate detected FPS across streams + p99 per-frame latency, 32x1080p30",
        "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
        "ms_per_step": round(elapsed / K * 1e3, 4), "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": workload,
                   "streams_per_gpu": S, "detector": f"yolov8{args.model}", "detector_engine": det.engine, "input": [640, 640],
                   "weights": "seeded random, class biases calibrated to ~%d candidates/frame" % args.target_dets,
                   "conf": args.conf, "iou": args.iou, "tracker": {"max_age": 30, "max_iou_distance": 0.5, "min_hits": 1},
                   "decode": "not measured: " + rocdecode_status()},
        "p99_latency_ms": round(float(np.percentile(lat, 99)) * 1e3, 3),
        "p50_latency_ms": round(float(np.percentile(lat, 50)) * 1e3, 3),
        "max_latency_ms": round(float(lat.max()) * 1e3, 3), "latency_samples": int(K),
        "host_submit_us_per_tick": {"mean": round(float(t_sub.mean()) * 1e6, 1), "p50": round(float(np.percentile(t_sub, 50)) * 1e6, 1),
                                    "what": "wall time inside PipelinedTicks.submit() on the host thread (every launch of the tick enqueued)"},
        "chain_stream_probe": ops.chain_stream_report(dev),
        "ticks_in_flight": runner.depth, "network_streams": runner.net_streams, "warmup_ticks_run": warm_ticks, "hip_graph": bool(use_graph), "network_launch": net_launch, "hip_graph_scope": ("network + tail" if runner.replay_net else ("post-process / tracker tail (networks launched eagerly; a tick is one chain on its own stream, consecutive ticks rotate over %d streams)" % runner.net_streams if runner.net_streams >= 2 else "post-process / tracker tail (network launched eagerly: concurrent detect branches)")) if use_graph else None, "realtime_32x30fps": bool(fps / world >= 30.0 * S and np.percentile(lat, 99) < 1 / 30),
        "stages_ms": {"k1_preprocess": round(float(k1_ms), 4), "detector": round(float(net_ms), 4),
                      "k2k3_postprocess": round(float(post_ms), 4), "k4_tracker": round(float(trk_ms), 4)},
        "detections_per_frame": round(dets_emitted / S, 2), "tracks_per_stream": round(n_tracks / (K * S), 2),
        "detector_tflops": round(net_tflops, 2), "detector_frac_of_mfma_peak": round(net_tflops / MFMA_PEAK_TFLOPS, 4),
        "detector_alone_layout": stage_mode,
        # one forward pass alone (the eager per-stage pass) above; below: the network FLOPs of a tick over the tick period of the
        # timed region, where the forward passes of consecutive ticks overlap on two streams
        "detector_tflops_in_pipeline": round(2 * macs * S / (elapsed / K) / 1e12, 2),
        "detector_frac_of_mfma_peak_in_pipeline": round(2 * macs * S / (elapsed / K) / 1e12 / MFMA_PEAK_TFLOPS, 4),
        "roofline": {"kernel": "k1_ratio_content<3,half,2> (NV12 1080p -> content rows of fp16 3x640x640, one launch per tick; the "
                               "constant letterbox border was written by the first launch into the buffer)", "bound": "hbm",
                     "achieved": round(k1_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(k1_gbs / HBM_PEAK_GBS, 4), "traffic": k1_traffic,
                     "traffic_source": "profiles/r04_k1_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; not collectable while timing)" if k1_traffic else None,
                     "algorithmic_bytes_per_launch": k1_frame_bytes * S, "avg_launch_us": round(float(k1_ms) * 1e3, 2),
                     "inputs": f"ring of {ring_frames} surfaces per stream ({ring_frames * S * ((args.width + 255) // 256 * 256) * args.height * 3 // 2 >> 20} MiB "
                               "in all): a surface is read again only after > 2 x 256 MiB of other surfaces, so K1 reads HBM, not the Infinity Cache",
                     "avg_launch_us_between_event_records": round(float(k1_bracket_ms) * 1e3, 2),
                     "timing": f"HIP start/stop events of the K1 dispatch itself (hipExtLaunchKernelGGL), every {K1_SAMPLE_EVERY}th timed tick "
                               "(the event packets cost ~10 us of queue time per use); K1 runs beside the forward passes of the other "
                               "ticks in flight -- alone it takes cold_launch_us, and under rocprofv3 --kernel-trace, which stretches the tick "
                               "by 40 % and thins out what co-runs, ~20 us (profiles/r03_tick_breakdown.csv)"},
    }
    if multi is not None:
        out["multi_gpu"] = multi
    out["kernel_selection"] = getattr(det._plans.get((S, 640, 640)), "tuning_source", None)
    refined = getattr(det._plans.get((S, 640, 640)), "refined", None)
    if refined:     # what the second pass of the kernel selection (whole forward passes, three in flight) changed: layer, from, to, us before / after
        out["kernel_selection_refinements"] = [list(r) for r in refined]
    if rank == 0 and not args.no_extras:
        try:
            if world == 1:
                # what the metric names: a real percentile over >= 300 saturated ticks whatever --steps was --
                # taken right behind the timed region, before the other legs allocate and free gigabytes of scratch --, and (below)
                # the deployment itself: one tick per 33.33 ms from a host timer on an otherwise idle GPU
                out["long_run"] = long_run_leg(runner, S, lat if K >= 300 else None, elapsed if K >= 300 else None)
                out["power_clock"] = power_leg(runner, S)
                clk = [x["sclk_mhz"] for x in out["power_clock"]["samples"] if x.get("sclk_mhz")]
                if clk:     # the dense peak is quoted at 2400 MHz; under the package power limit the chip holds less
                    out["power_clock"]["detector_frac_of_mfma_peak_at_sampled_clock"] = round(
                        out["detector_frac_of_mfma_peak_in_pipeline"] * 2400.0 / (sum(clk) / len(clk)), 4)
            extras(args, out, sources[:S], rctx, dev, dcfg, tcfg)
            if world == 1:
                if K < 300:     # the headline percentile is a percentile: taken from the >= 300-tick leg, the short region's kept beside it
                    out["p99_latency_ms_timed_region"], out["latency_samples_timed_region"] = out["p99_latency_ms"], int(K)
                    out["p99_latency_ms"], out["p50_latency_ms"] = out["long_run"]["p99_ms"], out["long_run"]["p50_ms"]
                    out["max_latency_ms"], out["latency_samples"] = out["long_run"]["max_ms"], out["long_run"]["ticks"]
                    out["latency_source"] = "long_run (saturated ticks after the timed region, same runner)"
                out["paced_30fps"] = paced_leg(args, det, sources[:S], streams, tcfg, fps)
                if args.model == "s" and S == 32 and not args.no_other_configs:
                    out["other_configs"] = other_configs_leg(args)
        except Exception as exc:  # noqa: BLE001  -- the measurements after the timed region never cost the headline line
            import traceback
            traceback.print_exc()
            out["extras_error"] = f"{type(exc).__name__}: {exc}"
    if rank == 0 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, net_cpu, shifts, sources, dcfg, tcfg)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def other_configs_leg(args):
    """BASELINE configs[1], one GPU's share of configs[3] and configs[4] in the same record: 4 x YOLOv8n, 4 x YOLOv8m (the 4 streams per GPU
    of 32 x YOLOv8m over 8 GPUs) and the 4K CNN-LSTM clip workload, each measured by a CHILD process running this file (this process has finished its own legs and
    leaves the GPU idle; a child, never an exec).  A failure or timeout of a child is recorded, it never costs the headline."""
    import subprocess
    res = {"what": "python bench.py --model <m> --streams 4 --steps <k> --warmup <w> --no-cpu-baseline --no-extras (and --workload temporal), "
                   "one child process each, after every other leg of this run"}
    for key, model, steps, warm in (("configs3_share_4x_yolov8m", "m", 200, 40), ("configs1_4x_yolov8n", "n", 300, 60),
                                    ("configs4_temporal_8x4k_cnn_lstm", None, 0, 0)):
        if model is None:      # BASELINE configs[4]: `--workload temporal` with its own defaults (8 x 4K streams, CNN-LSTM clips)
            cmd = [sys.executable, os.path.abspath(__file__), "--workload", "temporal", "--no-cpu-baseline", "--no-extras"]
        else:
            cmd = [sys.executable, os.path.abspath(__file__), "--model", model, "--streams", "4", "--steps", str(steps), "--warmup", str(warm),
                   "--no-cpu-baseline", "--no-extras", "--width", str(args.width), "--height", str(args.height)]
        try:
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
            line = next((ln for ln in reversed(p.stdout.strip().splitlines()) if ln.startswith("{")), None)
            if p.returncode != 0 or line is None:
                res[key] = {"error": f"exit {p.returncode}: {(p.stderr or '').strip().splitlines()[-1][:200] if p.stderr else ''}"}
                continue
            d = json.loads(line)
            res[key] = {"frames_per_s": d["value"], "ms_per_tick": d["ms_per_step"], "p99_latency_ms": d.get("p99_latency_ms"),
                        "steps": d["steps"], "ticks_in_flight": d.get("ticks_in_flight"), "detector": d["config"].get("detector"),
                        "network_launch": (d.get("network_launch") or {}).get("mode"),
                        "detector_frac_of_mfma_peak_in_pipeline": d.get("detector_frac_of_mfma_peak_in_pipeline")}
            if model is None:
                res[key]["clips_per_s"] = d.get("clips_per_s")
        except Exception as exc:  # noqa: BLE001
            res[key] = {"error": f"{type(exc).__name__}: {exc}"}
    return res


def _percentiles(lat_s):
    lat_ms = np.asarray(lat_s) * 1e3
    return {"p50_ms": round(float(np.percentile(lat_ms, 50)), 3), "p99_ms": round(float(np.percentile(lat_ms, 99)), 3),
            "p999_ms": round(float(np.percentile(lat_ms, 99.9)), 3), "max_ms": round(float(lat_ms.max()), 3)}


def long_run_leg(runner, S, lat=None, elapsed=None, K=320):
    """p99 over >= 9 600 frame latencies: >= 300 saturated light-load ticks through the headline's own runner
    (same ticks in flight); a frame's latency is its tick's (enqueue -> tracks on the host).  Reuses the timed region when
    --steps already gave that many."""
    reused = lat is not None
    if not reused:
        lat, t_enq, done = np.empty(K), np.empty(K), 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(K):
            if k - done == runner.depth:
                runner.collect(); lat[done] = time.perf_counter() - t_enq[done]; done += 1
            t_enq[k] = time.perf_counter()
            runner.submit()
        while done < K:
            runner.collect(); lat[done] = time.perf_counter() - t_enq[done]; done += 1
        elapsed = time.perf_counter() - t0
    K = len(lat)
    return {"ticks": int(K), "frame_latency_samples": int(K * S), "ticks_in_flight": runner.depth,
            "frames_per_s": round(S * K / elapsed, 1), **_percentiles(lat),
            "source": "the timed region itself" if reused else f"{K} further saturated ticks after the timed region (same runner)"}


def _smi_sample():
    """Package power (W) and shader clock (MHz) from rocm-smi, or None (the tool's JSON keys differ between releases: matched loosely)."""
    import re
    import subprocess
    try:
        p = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=15)
        cards = json.loads(p.stdout)
        card = cards.get("card0") or next(iter(cards.values()))
        power = next((float(v) for k, v in card.items() if "ower" in k and "(W)" in k), None)
        sclk = next((int(re.search(r"(\d+)\s*Mhz", str(v), re.I).group(1)) for k, v in card.items() if k.lower().startswith("sclk") and re.search(r"\d+\s*Mhz", str(v), re.I)), None)
        return None if power is None and sclk is None else {"package_power_w": power, "sclk_mhz": sclk}
    except Exception:  # noqa: BLE001
        return None


def power_leg(runner, S, seconds=3.0):
    """What the GPU draws and clocks at while the pipeline is saturated: ~3 s of further saturated ticks on the same runner, rocm-smi
    sampled from a host thread after 1 s and 2 s (an instantaneous reading each).  DESIGN.md section 9 (1b): four ticks in flight run at
    the package power limit."""
    import threading
    samples = []

    def sampler():
        for _ in range(2):
            time.sleep(1.0)
            smp = _smi_sample()
            if smp:
                samples.append(smp)
    th = threading.Thread(target=sampler, daemon=True)
    torch.cuda.synchronize()
    th.start()
    t0 = time.perf_counter()
    n = inflight = 0
    while time.perf_counter() - t0 < seconds:
        if inflight == runner.depth:
            runner.collect(); inflight -= 1
        runner.submit(); inflight += 1; n += 1
    while inflight:
        runner.collect(); inflight -= 1
    elapsed = time.perf_counter() - t0
    th.join(timeout=20)
    return {"ticks": n, "frames_per_s": round(S * n / elapsed, 1), "ticks_in_flight": runner.depth, "samples": samples,
            "what": "rocm-smi --showpower --showclocks while the saturated pipeline runs (two instantaneous readings)"}


def paced_leg(args, det, sources, streams, tcfg, saturated_fps, ticks=300, period=1.0 / 30.0):
    """The operating point the metric names: S streams at 30 fps -- one tick every 33.33 ms from a host timer, the GPU idle
    (and free to drop its clocks) in between, as a deployment that paces with sleep(1 / target_fps) sees it
    (reference: video_stream.py:241-243, pipeline.py:143-212).  Latency = timer fires (the tick's surfaces are ready) ->
    tracks of that tick on the host.  The tick is collected as soon as it is done; `chains` only says how many tick slots
    (streams, plans, snapshot slots) the runner rotates over."""
    from realtime_video_analytics_32streams_amd.pipeline import PipelinedTicks, TickPipeline
    from realtime_video_analytics_32streams_amd.tracker import IouTracker
    S = len(sources)
    legs = {}
    for chains in (1, 3):
        trk = IouTracker(tcfg, max_streams=S, capacity=1024, device=det.device.index)
        pipe = TickPipeline(streams, det, trk, sources=sources)
        runner = PipelinedTicks(pipe, depth=chains, use_graph=not args.no_graph)
        for _ in range(12):                                   # sizes buffers, captures the graphs
            runner.submit(); runner.collect()
        torch.cuda.synchronize()
        lat = np.empty(ticks)
        late = 0
        t0 = time.perf_counter() + 0.05
        for k in range(ticks):
            due = t0 + k * period
            while True:
                now = time.perf_counter()
                if now >= due:
                    break
                if due - now > 1.5e-3:
                    time.sleep(due - now - 1e-3)              # sleep to within a millisecond, spin the rest
            late += now - due > 1e-3
            runner.submit()
            runner.collect()
            lat[k] = time.perf_counter() - due                # from the moment the surfaces were due, timer jitter included
        legs[f"chains_{chains}"] = {"ticks": ticks, "frame_latency_samples": ticks * S, **_percentiles(lat),
                                    "timer_fired_late_over_1ms": int(late)}
        del runner, pipe, trk
    worst = max(v["p99_ms"] for v in legs.values())
    return {"streams": S, "tick_period_ms": round(period * 1e3, 3), "what": paced_leg.__doc__.split("\n")[0].strip(),
            **legs, "p99_under_33ms": bool(worst < period * 1e3),
            "stream_sets_that_fit": round(saturated_fps / (S * 30.0), 2),
            "stream_sets_note": f"saturated frames/s of the timed region / ({S} streams x 30 fps)"}


def clip_k1_bytes(w, h, dst, out_bytes):
    """Algorithmic bytes of one frame through the clip pre-process ("K1 temporal"): the source rows the two
    vertical taps of every output row touch (Y rows whole + the UV rows under them, whole: at 17x horizontal decimation
    every 64-byte sector of a touched row holds a tap) + the [3, dst, dst] output."""
    ys = set()
    for oy in range(dst):
        fy = (oy + 0.5) * (h / dst) - 0.5
        y0 = int(np.floor(fy))
        ys.update((min(max(y0, 0), h - 1), min(max(y0 + 1, 0), h - 1)))
    return (len(ys) + len({y >> 1 for y in ys})) * w + 3 * dst * dst * out_bytes


def temporal_main(args, rank, world, local, dev):
    """BASELINE configs[4]: S x 3840x2160 NV12 streams per GPU, CNN-LSTM (sample-temporal-pipeline.yaml:24-48: L = 16,
    stride 2, overlap 0.5, 224 x 224, half: false) through the tick pipeline in throughput mode: every frame is pre-processed
    on arrival into an HBM ring (one launch per tick for all streams), a stream's clip fires every 8 frames after the first
    32 (all streams of a tick as ONE network batch), top-5 -> shared IoU tracker -> ids -> snapshot.  Consecutive ticks run as
    two chains on two HIP streams, so the clip pre-process of tick k+1 runs beside the network of tick k."""
    from realtime_video_analytics_32streams_amd import _native as N
    from realtime_video_analytics_32streams_amd import dist as rdist
    from realtime_video_analytics_32streams_amd import ops
    from realtime_video_analytics_32streams_amd.config import DetectorConfig, StreamConfig, TrackerConfig
    from realtime_video_analytics_32streams_amd.pipeline import PipelinedTicks, TickPipeline
    from realtime_video_analytics_32streams_amd.temporal import CnnLstmNet, HipCNNLSTMDetector
    from realtime_video_analytics_32streams_amd.tracker import IouTracker
    from realtime_video_analytics_32streams_amd.video_stream import SyntheticNv12Stream, rocdecode_status
    W, H = (3840, 2160) if (args.width, args.height) == (1920, 1080) else (args.width, args.height)
    S = 8 if args.streams == 32 else args.streams
    first = rank * S
    streams = [StreamConfig(name=f"uhd{first + i:03d}", url=f"synthetic://{W}x{H}", target_fps=30.0, warmup_seconds=0.0) for i in range(S)]
    sources = [SyntheticNv12Stream(s, index=first + i, width=W, height=H, n_unique=2, device=dev,
                                   ring_frames=input_ring_frames(S, W, H)) for i, s in enumerate(streams)]
    for src in sources:
        src.open_sync()
    dcfg = DetectorConfig(model_path="cnn_lstm_kinetics400.onnx", backend="hip", model_type="cnn_lstm", sequence_length=16,
                          sequence_stride=2, temporal_overlap=0.5, confidence_threshold=-1e9, num_action_classes=400,
                          input_size=[224, 224], half=False, warmup=False)
    torch.manual_seed(1)
    det = HipCNNLSTMDetector(dcfg, net=CnnLstmNet(400).eval(), device=local)
    tcfg = TrackerConfig(max_age=30, max_iou_distance=0.5, min_hits=1)
    trk = IouTracker(tcfg, max_streams=S, capacity=256, device=local)
    id_sync = rdist.IdSync(S, dev) if world > 1 else None
    if world == 1 and os.environ.get("RVA_FIFTH_STREAM"):
        # experiment (profiles/r03_experiments_not_kept.txt #14): the id exchange as ProcessGroupNCCL would issue it -- a tiny kernel on
        # a stream of its own, event-ordered behind the tick's K4 and in front of its id assignment -- on a single GPU
        class _FifthStream(rdist.IdSync):
            def __init__(self, per, device, early):
                super().__init__(per, device)
                self.s = torch.cuda.Stream(device=device)
                if early:                                   # first use before the chains' streams exist
                    with torch.cuda.stream(self.s):
                        self.buf.zero_()
                    torch.cuda.synchronize()

            def _gather(self, local_counts):
                cur = torch.cuda.current_stream()
                self.s.wait_stream(cur)
                with torch.cuda.stream(self.s):
                    self.buf[: self.per].copy_(local_counts)
                cur.wait_stream(self.s)
                return self.buf
        id_sync = _FifthStream(S, dev, os.environ["RVA_FIFTH_STREAM"] == "early")
    pipe = TickPipeline(streams, det, trk, sources=sources, id_sync=id_sync, first_global_index=first, n_global_streams=world * S)
    runner = PipelinedTicks(pipe, depth=args.depth)
    rctx = ops.context(local)
    warm = max(args.warmup, 48)                                  # the first clips fire at tick 31, the second ones at 39
    for _ in range(warm):
        runner.submit(); runner.collect()
    torch.cuda.synchronize()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
    K = args.steps
    lat, t_enq = np.empty(K), np.empty(K)
    evk = [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(K)]
    for pair in evk:
        for x in pair:
            x.record()
    torch.cuda.synchronize()
    clips = rows = 0

    def enqueue(k):
        t_enq[k] = time.perf_counter()
        if k % K1_SAMPLE_EVERY == 0:
            N.lib().rva_profile_next_preprocess(rctx.handle, evk[k][0].cuda_event, evk[k][1].cuda_event)
        runner.submit()

    def finish(k):
        nonlocal clips, rows
        r = runner.collect_result()
        lat[k] = time.perf_counter() - t_enq[k]
        clips += sum(1 for v in r.detections_emitted.values() if v)
        rows += sum(len(v) for v in r.tracks.values())
    barrier()
    t0 = time.perf_counter()
    done = 0
    for k in range(K):
        if k - done == runner.depth:
            finish(done); done += 1
        enqueue(k)
    while done < K:
        finish(done); done += 1
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed, float(clips)], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt[:1], op=torch.distributed.ReduceOp.MAX)
        torch.distributed.all_reduce(tt[1:], op=torch.distributed.ReduceOp.SUM)
        elapsed, clips = float(tt[0].item()), int(tt[1].item())
    k1_ms = float(np.mean([a.elapsed_time(b) for a, b in evk[::K1_SAMPLE_EVERY]]))
    per_frame = clip_k1_bytes(W, H, 224, 4)
    gbs = per_frame * S / (k1_ms * 1e-3) / 1e9 if k1_ms > 0 else 0.0
    fps = world * S * K / elapsed
    out = {"metric": "aggregate detected FPS across streams + p99 per-frame latency, 32x1080p30", "value": round(fps, 2),
           "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": round(elapsed / K * 1e3, 4),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"{world * S}x{W}x{H} NV12 streams resident in HBM ({S} per GPU), CNN-LSTM over 16-frame clips "
                                  "(stride 2, overlap 0.5, 224x224, fp32), shared IoU tracker (BASELINE configs[4]; NOT the headline "
                                  "configuration, which `python bench.py` without --workload measures)",
                      "streams_per_gpu": S, "detector": "cnn_lstm (seeded random weights; the network itself runs through PyTorch-ROCm)",
                      "clip": {"sequence_length": 16, "stride": 2, "overlap": 0.5, "input": [224, 224], "half": False},
                      "tracker": {"max_age": 30, "max_iou_distance": 0.5, "min_hits": 1},
                      "decode": "not measured: " + rocdecode_status()},
           "clips_per_s": round(clips / elapsed, 2), "clips_in_timed_region": int(clips),
           "p99_latency_ms": round(float(np.percentile(lat, 99)) * 1e3, 3), "p50_latency_ms": round(float(np.percentile(lat, 50)) * 1e3, 3),
           "max_latency_ms": round(float(lat.max()) * 1e3, 3), "latency_samples": int(K), "ticks_in_flight": runner.depth,
           "network_streams": runner.net_streams, "warmup_ticks_run": warm, "tracks_per_stream": round(rows / (K * S), 2),
           "realtime_30fps": bool(fps / world >= 30.0 * S and np.percentile(lat, 99) < 1 / 30),
           "roofline": {"kernel": "k1_generic<nv12, clip, float> (3840x2160 NV12 -> fp32 [3,224,224] per stream, stretch resize + "
                                  "ImageNet mean/std, one launch per tick straight into the clip ring)", "bound": "hbm",
                        "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                        "traffic": None, "algorithmic_bytes_per_launch": per_frame * S, "algorithmic_bytes_per_frame": per_frame,
                        "avg_launch_us": round(k1_ms * 1e3, 2),
                        "timing": f"HIP start/stop events of the dispatch itself, every {K1_SAMPLE_EVERY}th timed tick"}}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def extras(args, out, sources, rctx, dev, dcfg, tcfg):
    """Measurements taken AFTER the timed region on rank 0 (none of them changes `value`):
      * K1 from cold caches: the same launch right after a 512 MiB fill has pushed the surfaces (and everything else) out
        of the 256 MiB Infinity Cache -- the in-pipeline figure re-reads a 2-frame ring per stream that can sit in it;
      * the device-to-device copy rate of this GPU (what a pure streaming kernel reaches here), K1 as a fraction of it;
      * K2/K3 and K4 at 64 and 256 kept detections per frame x 32 streams (the bench's own scene has ~13)."""
    from realtime_video_analytics_32streams_amd import _native as N
    from realtime_video_analytics_32streams_amd import ops, synth
    S = len(sources)
    surf = [src._ring[0] for src in sources]
    outb = torch.empty((S, 3, 640, 640), dtype=torch.float16, device=dev)
    scratch = torch.empty(512 << 20, dtype=torch.uint8, device=dev).fill_(1)
    ops.preprocess_nv12(surf, (640, 640), half=True, out=outb, ctx=rctx)      # full launch: writes the border of `outb`

    def k1_us(content, sweep, reps=8):
        ts = []
        for _ in range(reps):
            if sweep == "read":
                scratch.view(torch.int64).sum()                   # 512 MiB of reads: evicts the 256 MiB Infinity Cache, lines stay clean
            elif sweep == "write":
                scratch.fill_(1)                                  # ... or leaves it full of dirty lines K1 has to push out
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); b.record()
            torch.cuda.synchronize()
            N.lib().rva_profile_next_preprocess(rctx.handle, a.cuda_event, b.cuda_event)
            ops.preprocess_nv12(surf, (640, 640), half=True, out=outb, ctx=rctx, content_only=content)
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        return float(np.median(ts))
    roof = out["roofline"]
    cold_ms = k1_us(True, "read")
    if cold_ms > 1e-3 and (args.width, args.height) == (1920, 1080):      # events were written by the dispatch (integer-ratio K1 path)
        gbs = roof["algorithmic_bytes_per_launch"] / (cold_ms * 1e-3) / 1e9
        roof["cold_launch_us"] = round(cold_ms * 1e3, 2)
        roof["cold_achieved"] = round(gbs, 1)
        roof["cold_frac"] = round(gbs / HBM_PEAK_GBS, 4)
        roof["cold_state"] = "after a 512 MiB read sweep (Infinity Cache evicted, clean)"
        dirty_ms = k1_us(True, "write")
        roof["cold_after_write_sweep_us"] = round(dirty_ms * 1e3, 2)
        full = {"kernel": "k1_ratio<3,half> (border included: first launch into a buffer)", "algorithmic_bytes_per_launch": K1_BYTES_FULL * S}
        for key, sweep in (("warm", None), ("cold", "read")):
            ms = k1_us(False, sweep)
            full[f"{key}_launch_us"] = round(ms * 1e3, 2)
            full[f"{key}_frac"] = round(K1_BYTES_FULL * S / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        roof["full_tensor_kernel"] = full
    # the other two K1 rooflines SURVEY.md 8(d) lists: 4K -> 640x640 (k1_ratio<6>: 2x2 mean of every 6th pair of rows and
    # columns) and the temporal heads' 4K -> 224x224 stretch (k1_generic, two taps per axis); 8 x 3840x2160 surfaces each
    from realtime_video_analytics_32streams_amd.video_stream import SyntheticNv12Stream
    from realtime_video_analytics_32streams_amd.config import StreamConfig
    uhd = []
    for i in range(8):
        src = SyntheticNv12Stream(StreamConfig(name=f"uhd{i}", url="synthetic://3840x2160", warmup_seconds=0.0), index=i, width=3840,
                                  height=2160, n_unique=1, device=dev)
        src.open_sync()
        uhd.append(src._ring[0])
    out4k = torch.empty((8, 3, 640, 640), dtype=torch.float16, device=dev)
    ring = torch.empty((8, 3, 224, 224), dtype=torch.float32, device=dev)

    def timed(fn, sweep, reps=8):
        ts = []
        for _ in range(reps):
            if sweep:
                scratch.view(torch.int64).sum()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); b.record()
            torch.cuda.synchronize()
            N.lib().rva_profile_next_preprocess(rctx.handle, a.cuda_event, b.cuda_event)
            fn()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        return float(np.median(ts))
    per4k = k1_bytes(3840, 2160, content_only=False)
    assert per4k == 6_604_800
    perclip = clip_k1_bytes(3840, 2160, 224, 4)
    for key, fn, per, kern in (("roofline_4k", lambda: ops.preprocess_nv12(uhd, (640, 640), half=True, out=out4k, ctx=rctx), per4k,
                                "k1_ratio<6,half> (8 x 3840x2160 NV12 -> fp16 3x640x640, border included)"),
                               ("roofline_clip", lambda: ops.preprocess_frames(uhd, (224, 224), N.NORM_IMAGENET_F32, N.LAYOUT_NCHW, torch.float32,
                                                                               out=ring, ctx=rctx), perclip,
                                "k1_generic<nv12, clip, float> (8 x 3840x2160 NV12 -> fp32 3x224x224, stretch + mean/std)")):
        fn()
        rec = {"kernel": kern, "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "algorithmic_bytes_per_frame": per,
               "algorithmic_bytes_per_launch": per * 8, "traffic": None}
        pmc = ROOT / "profiles" / "r04_k1_pmc.json"
        if pmc.exists():                                   # committed PMC passes of exactly this launch shape (tools/r04_n.sh)
            rec["traffic"] = json.loads(pmc.read_text())["k1_ratio6_4k" if key == "roofline_4k" else "k1_generic_clip_4k"]["traffic_bytes_per_launch"]
            rec["traffic_source"] = "profiles/r04_k1_pmc.json"
        for state, sweep in (("warm", False), ("cold", True)):
            ms = timed(fn, sweep)
            if ms > 1e-4:
                rec[f"{state}_launch_us"] = round(ms * 1e3, 2)
                rec[f"{state}_achieved"] = round(per * 8 / (ms * 1e-3) / 1e9, 1)
                rec[f"{state}_frac"] = round(per * 8 / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        rec["achieved"], rec["frac"] = rec.get("cold_achieved"), rec.get("cold_frac")      # cold = how it runs in a pipeline
        out[key] = rec
    del uhd, out4k, ring
    # device copy rate: 1 GiB -> 1 GiB, bytes moved = read + written
    n = 1 << 30
    src_b = torch.empty(n, dtype=torch.uint8, device=dev).fill_(3)
    dst_b = torch.empty(n, dtype=torch.uint8, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    dst_b.copy_(src_b)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        dst_b.copy_(src_b)
    e1.record()
    torch.cuda.synchronize()
    copy_gbs = 2.0 * n * 10 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    roof["device_copy_gbs"] = round(copy_gbs, 1)
    roof["frac_of_device_copy"] = round(roof["achieved"] / copy_gbs, 4)
    if "cold_achieved" in roof:
        roof["cold_frac_of_device_copy"] = round(roof["cold_achieved"] / copy_gbs, 4)
    del src_b, dst_b, scratch
    # post-process / tracker under load
    sweep = {}
    meta = [N.letterbox(args.width, args.height, 640, 640)]
    for D in (64, 256):
        heads = synth.make_head_batch([9000 + D + i for i in range(S)], layout="CA", n_obj=D)
        raw = torch.from_numpy(heads).to(dev).half()
        post = ops.PostBuffers.allocate(S, raw.shape[2], dev)
        trk = ops.DeviceTracker(S, tcfg.max_age, tcfg.max_iou_distance, tcfg.min_hits, capacity=1024, ctx=rctx)
        slots = list(range(S))
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(30)]
        for t in range(32):
            if t >= 2: ev[t - 2][0].record()
            ops.postprocess(raw, dcfg.confidence_threshold, dcfg.iou_threshold, None, meta, out=post, ctx=rctx)
            if t >= 2: ev[t - 2][1].record()
            trk.update_from_post(slots, post, dcfg.confidence_threshold)
            trk.assign_ids()
            if t >= 2: ev[t - 2][2].record()
        torch.cuda.synchronize()
        tabs = trk.read_all()
        sweep[f"D{D}"] = {"kept_per_frame": round(float(post.counts.float().mean().item()), 1),
                          "tracks_per_stream": round(float(np.mean([t["n"] for t in tabs])), 1),
                          "k2k3_us_per_tick": round(float(np.mean([e[0].elapsed_time(e[1]) for e in ev])) * 1e3, 1),
                          "k4_us_per_tick": round(float(np.mean([e[1].elapsed_time(e[2]) for e in ev])) * 1e3, 1)}
        trk.close()
    out["post_tracker_load_sweep"] = {"streams": S, "ticks": 30, **sweep}
    if out["n_gpus"] == 1:
        out["load_sweep_end_to_end"] = end_to_end_load_sweep(args, sources, rctx, dev, dcfg, tcfg, out["value"])
    out["decode"] = decode_stage(dev)


def end_to_end_load_sweep(args, sources, rctx, dev, dcfg, tcfg, light_fps):
    """Frames/s of the WHOLE pipeline (same streams, same runner, the network running in full) with the tail under load: K2
    reads a PLANTED head -- the synthetic heads of `post_tracker_load_sweep`, 64 and 256 objects per frame -- instead of the
    network's own output, whose seeded random weights keep ~13 boxes per frame.  Why planted: the class field of a random
    head is spatially smooth, so what survives NMS is a staircase in the calibration knob (6, 12, 113, 1731 boxes per frame
    on this scene), not a dial.  Everything else is the headline pipeline: K1, every layer of the detector, K2 (same
    tensor shape, from another buffer), K3, K4, ids, snapshot; 150 timed ticks after 30 warm-up ticks per leg."""
    import copy
    from realtime_video_analytics_32streams_amd import ops, synth
    from realtime_video_analytics_32streams_amd.config import StreamConfig
    from realtime_video_analytics_32streams_amd.detector import HipYoloDetector
    from realtime_video_analytics_32streams_amd.pipeline import PipelinedTicks, TickPipeline
    from realtime_video_analytics_32streams_amd.tracker import IouTracker
    from realtime_video_analytics_32streams_amd.yolov8 import build_detector_net
    S = len(sources)
    base_net = build_detector_net(args.model, seed=0)
    streams = [StreamConfig(name=src.config.name, url=src.config.url, target_fps=30.0, warmup_seconds=0.0) for src in sources]
    legs = {}
    for D in (64, 256):
        planted = torch.from_numpy(synth.make_head_batch([9000 + D + i for i in range(S)], layout="CA", n_obj=D)).to(dev).half()
        det = HipYoloDetector(dcfg, net=copy.deepcopy(base_net), device=dev.index)
        own_post = det.stage_post
        det.stage_post = lambda raw, pre, own_post=own_post, planted=planted: own_post(planted, pre)   # the network ran; K2 reads the planted head
        trk = IouTracker(tcfg, max_streams=S, capacity=1024, device=dev.index)
        pipe = TickPipeline(streams, det, trk, sources=sources)
        runner = PipelinedTicks(pipe, depth=args.depth, use_graph=not args.no_graph)
        for _ in range(30):
            runner.submit(); runner.collect()
        torch.cuda.synchronize()
        K = 150
        lat, t_enq, rows = np.empty(K), np.empty(K), 0
        t0 = time.perf_counter()
        done = 0
        for k in range(K + 1):                               # the same number of ticks in flight as the headline run
            while done < k and (k == K or k - done == runner.depth):
                rows += sum(t["n"] for t in runner.collect()[1]); lat[done] = time.perf_counter() - t_enq[done]; done += 1
            if k < K:
                t_enq[k] = time.perf_counter()
                runner.submit()
        el = time.perf_counter() - t0
        fps = S * K / el
        legs[f"D{D}"] = {"planted_objects_per_frame": D, "frames_per_s": round(fps, 1), "ms_per_tick": round(el / K * 1e3, 4),
                         "p99_latency_ms": round(float(np.percentile(lat, 99)) * 1e3, 3),
                         "candidates_per_frame": round(float(runner.last_post.ncand.float().mean().item()), 0),
                         "kept_per_frame": round(float(runner.last_post.counts.float().mean().item()), 1),
                         "tracks_per_stream": round(rows / (K * S), 1), "vs_light_load": round(fps / light_fps, 4)}
        del runner, pipe, trk, det, planted
    return {"streams": S, "ticks": 150, "light_load_frames_per_s": light_fps,
            "note": "the network runs in full; K2 reads a planted head (see the docstring of end_to_end_load_sweep)", **legs}


def decode_stage(dev):
    """D1 as its own stage: frames/s of one rocDecode session on a local bitstream (RVA_DECODE_SAMPLE = an MP4 or Annex-B
    file), surfaces left in HBM.  Needs librocdecode AND a file; this project's boxes have neither, so the stage reports
    why it was not measured instead of substituting anything."""
    from realtime_video_analytics_32streams_amd.config import StreamConfig
    from realtime_video_analytics_32streams_amd.video_stream import RocDecodeStream, rocdecode_status
    status = rocdecode_status()
    sample = os.environ.get("RVA_DECODE_SAMPLE", "")
    if not status.startswith("available"):
        return {"measured": False, "why": "rocDecode " + status}
    if not sample or not Path(sample).is_file():
        return {"measured": False, "why": "librocdecode present but no bitstream: set RVA_DECODE_SAMPLE to a local H.264/H.265 file "
                                           "(no encoder exists offline to synthesise 1080p30 H.265)"}
    st = RocDecodeStream(StreamConfig(name="decode-bench", url=sample, warmup_seconds=0.0), device=dev.index)
    st.open_sync()
    n, t0 = 0, time.perf_counter()
    while n < 2000 and time.perf_counter() - t0 < 20.0:
        if st.next_surface() is None:
            break
        n += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    info = dict(st.info)
    st.close_sync()
    return {"measured": True, "frames": n, "frames_per_s": round(n / dt, 1), "sessions": 1, "source": {k: info.get(k) for k in ("codec", "width", "height")}}


def cpu_baseline(args, net_cpu, shifts, sources, dcfg, tcfg):
    """The same tick on the host cores of the GPU box: the CPU oracle (C restatement of the reference) for pre-process /
    post-process / tracker + the same network in torch CPU fp32, on a bounded sample.  Two legs: everything the reference
    runs on ONE thread per frame (its asyncio loop is single-threaded apart from what ORT parallelises), and the same
    sample with the frames spread over all host cores (what N reference processes could reach)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as orc
    from realtime_video_analytics_32streams_amd.yolov8 import count_macs
    S = len(sources)
    n = args.cpu_frames or min(S, 16)
    cores = os.cpu_count() or 1
    net = net_cpu.fuse().float()
    for seq in net.detect.cls:
        seq[-1].bias.data[0] += shifts[0]
        seq[-1].bias.data[1:] += shifts[1]
    frames = []
    for src in sources[:n]:
        s = src._ring[0]
        frames.append((s.y.cpu().numpy(), s.uv.cpu().numpy()))

    def pre(f):
        return orc.preprocess_nv12(f[0], f[1], args.width, args.height, 640, 640, False)[0]

    def post_one(raw_i):
        r = orc.postprocess(raw_i, dcfg.confidence_threshold, dcfg.iou_threshold, None, (args.width, args.height))
        m = r["conf"].astype(np.float64) >= dcfg.confidence_threshold
        return r["boxes"][m].astype(np.float64), r["conf"][m].astype(np.float64), r["cls"][m].astype(np.int64)

    # the network leg first, on its own: one untimed warm-up call (oneDNN primitive creation, thread pool start), then a small
    # sweep over torch's intra-op thread count -- a 256-core host is not fastest with 128 threads on a 16-image batch
    x_net = torch.from_numpy(np.stack([pre(f) for f in frames]))
    nthr0 = int(torch.get_num_threads())
    sweep = {}
    with torch.inference_mode():
        net(x_net[:2])
        for t in sorted({min(t, cores) for t in (8, 16, 32, 64, nthr0, cores)}):
            torch.set_num_threads(t)
            net(x_net[:2])
            a = time.perf_counter()
            net(x_net)
            sweep[t] = time.perf_counter() - a
            if sweep[t] > 2.0 * min(sweep.values()):         # past the knee (256 threads on a 16-image batch: 50 s): stop
                break
    nthr = min(sweep, key=sweep.get)
    torch.set_num_threads(nthr)
    # leg 1: oracle stages on one thread, network on the best torch thread count
    trk = orc.Tracker(n, tcfg.max_age, tcfg.max_iou_distance, tcfg.min_hits)
    t0 = time.perf_counter()
    tens = np.stack([pre(f) for f in frames])
    t1 = time.perf_counter()
    with torch.inference_mode():
        raw = net(torch.from_numpy(tens)).numpy()
    t2 = time.perf_counter()
    for i in range(n):
        trk.update(i, *post_one(raw[i]))
    t3 = time.perf_counter()
    # leg 2: the frames of the sample in parallel over the host cores (the C oracle releases the GIL inside ctypes)
    workers = max(1, min(cores, n))
    trk2 = orc.Tracker(n, tcfg.max_age, tcfg.max_iou_distance, tcfg.min_hits)
    with ThreadPoolExecutor(workers) as ex:
        u0 = time.perf_counter()
        tens2 = np.stack(list(ex.map(pre, frames)))
        u1 = time.perf_counter()
        with torch.inference_mode():
            raw2 = net(torch.from_numpy(tens2)).numpy()
        u2 = time.perf_counter()
        dets = list(ex.map(post_one, [raw2[i] for i in range(n)]))
        for i in range(n):                                     # the tracker tables are per stream: one update each
            trk2.update(i, *dets[i])
        u3 = time.perf_counter()
    torch.set_num_threads(nthr0)
    gflop = 2.0 * count_macs(net) * n / 1e9
    return {"value": round(n / (u3 - u0), 2), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{n} frames of the same workload through oracle/rva_oracle.c (pre/post/tracker) + the same "
                      f"YOLOv8{args.model} in torch CPU fp32; NOT OpenCV+ONNXRuntime (neither is installed).  `value` is the "
                      f"all-cores leg ({workers} oracle threads, {nthr} torch threads = the fastest of the sweep, after a warm-up call)",
            "torch_thread_sweep_s": {str(k): round(v, 3) for k, v in sweep.items()},
            "detector_cpu_gflops": round(gflop / (u2 - u1), 1),
            "legs": {"one_oracle_thread": {"frames_per_s": round(n / (t3 - t0), 2), "threads": {"oracle": 1, "torch": nthr},
                                           "seconds": {"preprocess": round(t1 - t0, 3), "detector": round(t2 - t1, 3),
                                                       "post_tracker": round(t3 - t2, 3)}},
                     "all_cores": {"frames_per_s": round(n / (u3 - u0), 2), "threads": {"oracle": workers, "torch": nthr},
                                   "seconds": {"preprocess": round(u1 - u0, 3), "detector": round(u2 - u1, 3),
                                               "post_tracker": round(u3 - u2, 3)}}}}


if __name__ == "__main__":
    main()
