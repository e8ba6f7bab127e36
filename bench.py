#!/usr/bin/env python3
"""frames/s of the MI355X hot path: voxelize + mean VFE -> VoxelResBackBone8x (bf16, fp32 accumulate) -> dense BEV
-> batched rotated NMS, on the nuScenes CenterPoint configuration (BASELINE.json configs[2], the configuration the
metric is quoted on), frames sharded over the GPUs of one node with no data-path collective.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one pass of the whole hot path over one batch of `--batch` synthetic frames per GPU (inputs already in HBM).
Rank 0 prints ONE JSON line.  Besides the contract's keys it carries
  roofline      the dominant kernel (the implicit-GEMM sparse conv) priced from its algorithmic flops and its launch
                durations measured with HIP events on the launching stream in an instrumented pass of the same plan
  cpu_baseline  the CPU oracle (oracle/, a port: spconv itself cannot run here) timed on the host cores on a bounded sample
  stages        per-stage device time of one step (ms), for orientation
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from openpcdet_amd import _lib, backbones, engine, synth  # noqa: E402
from openpcdet_amd import dist as pdist  # noqa: E402

WORKLOAD = "nuscenes_centerpoint"
HEADS, BOXES_PER_HEAD, NMS_THRESH, NMS_POST = 6, 500, 0.2, 83   # cbgs_voxel0075_res3d_centerpoint.yaml:89-132
PEAK_BF16_TFLOPS = 2500.0     # dense bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def randomize_bn(model, seed):
    g = torch.Generator().manual_seed(seed)
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm1d):
            m.weight.data = torch.rand(m.num_features, generator=g) * 0.5 + 0.75
            m.bias.data = torch.randn(m.num_features, generator=g) * 0.1
            m.running_mean.data = torch.randn(m.num_features, generator=g) * 0.1
            m.running_var.data = torch.rand(m.num_features, generator=g) * 0.5 + 0.5


def build_model(cfg, dev):
    torch.manual_seed(11)
    grid = [int(round((cfg["range"][3 + j] - cfg["range"][j]) / cfg["voxel"][j])) for j in range(3)]
    net = backbones.VoxelResBackBone8x(input_channels=cfg["C"], grid_size=np.array(grid))
    randomize_bn(net, 12)
    return net.to(dev).eval()


def make_inputs(cfg, rank, pool, dev):
    # one contiguous pool [pool, n_points, C]: a step's B frames are one slice of it (two when the ring wraps)
    frames = torch.from_numpy(np.stack([synth.config_cloud(WORKLOAD, 1000 * rank + 3 + i) for i in range(pool)])).to(dev)
    S = HEADS
    boxes = np.zeros((pool, S, BOXES_PER_HEAD, 9), np.float32)
    scores = np.zeros((pool, S, BOXES_PER_HEAD), np.float32)
    counts = np.zeros((pool, S), np.int32)
    rng = np.random.default_rng(77 + rank)
    for i in range(pool):
        for s in range(S):
            boxes[i, s], scores[i, s] = synth.boxes(BOXES_PER_HEAD, 9, cfg["range"], 5000 * rank + 10 * i + s, clustered=40)
            counts[i, s] = rng.integers(300, BOXES_PER_HEAD + 1)
    return frames, torch.from_numpy(boxes).to(dev), torch.from_numpy(scores).to(dev), torch.from_numpy(counts).to(dev)


def cpu_baseline(cfg, net_cpu, n_frames, boxes, scores, counts, warmups=3, single_thread_frames=1):
    """The oracle chain on the host cores, BASELINE.md section 3: `warmups` untimed frames, then n_frames timed full frames (the GPU run's
    own synthetic frames and seeds), frames/s from the MEDIAN frame time, per-stage medians, on all threads; then the same chain on ONE
    thread for single_thread_frames frames (a full frame takes ~35 s there, so that sample is bounded to what the default run can afford).
    -> dict for the JSON line"""
    from oracle import chain, loader
    threads = int(loader.oracle().orc_num_threads())

    def one(i, stages=None):
        pts = synth.config_cloud(WORKLOAD, 3 + i)
        t0 = time.perf_counter()
        tm = {}
        chain.backbone_chain(net_cpu, [pts], cfg, timings=tm)
        t1 = time.perf_counter()
        chain.heads_nms(boxes[i % len(boxes)], scores[i % len(boxes)], counts[i % len(boxes)], NMS_THRESH, NMS_POST)
        t2 = time.perf_counter()
        if stages is not None:
            tm["nms"] = t2 - t1
            stages.append(tm)
        return t2 - t0

    for i in range(warmups):
        one(i)
    stages = []
    ts = sorted(one(warmups + i, stages) for i in range(n_frames))
    med = ts[len(ts) // 2]
    stage_ms = {k: round(sorted(st[k] for st in stages)[len(stages) // 2] * 1e3, 2) for k in stages[0]} if stages else {}
    out = {"value": round(1.0 / med, 4), "unit": "frames/s", "cores": threads, "kind": "port", "host_cpus": os.cpu_count(),
           "stage_ms_median": stage_ms,
           "sample": "%d warm-ups + %d timed full frames (300k points each), median frame time, fp32, all stages; convs on %d OpenMP "
                     "threads, voxelizer / rulebook / NMS single-threaded (BASELINE.md section 3)" % (warmups, n_frames, threads)}
    if single_thread_frames > 0:
        prev = torch.get_num_threads()
        loader.oracle().orc_set_num_threads(1)
        torch.set_num_threads(1)
        t1 = sorted(one(i) for i in range(single_thread_frames))
        out["value_1_thread"] = round(1.0 / t1[len(t1) // 2], 4)
        out["sample_1_thread"] = "%d timed full frame(s), no warm-up (a frame takes ~35 s on one thread)" % single_thread_frames
        torch.set_num_threads(prev)
        loader.oracle().orc_set_num_threads(threads)
    return out


# which HIP kernel a (cin, cout, kernel volume) layer of this backbone launches (csrc/sparse_conv_slab.h PVX_SLAB_SHAPES,
# csrc/sparse_conv_w32.h); the names are the ones rocprofv3 --kernel-trace --stats prints
def conv_kernel_name(cin, cout, kvol):
    if kvol == 27 and (cin, cout) == (64, 64) and os.environ.get("PVX_CONV_ST", "1") != "0" and os.environ.get("PVX_ENGINE_TILEBOOK", "1") == "1":
        return "sparse_conv_slab_ws_kernel<bf16,%d,%d>" % (cin, cout)
    if kvol == 27 and (cin, cout) in engine.HotPathEngine.TILEBOOK_SHAPES and os.environ.get("PVX_ENGINE_TILEBOOK", "1") == "1":
        if (cin, cout) == (32, 32) and os.environ.get("PVX_CONV_ROLL", "1") != "0":
            return "sparse_conv_slab_roll_kernel<bf16,%d,%d>" % (cin, cout)
        if (cin, cout) in ((16, 16), (16, 32)):
            return "sparse_conv_slab_tb_kernel<bf16,%d,%d>" % (cin, cout)
    if kvol == 27 and (cin, cout) in ((16, 16), (16, 32), (32, 32), (32, 64), (64, 64)):
        return "sparse_conv_slab_kernel<bf16,%d,%d>" % (cin, cout)
    if kvol == 27 and (cin, cout) in ((64, 128), (128, 64), (128, 128)):
        return "sparse_conv_slab_stream_kernel<bf16,%d,%d>" % (cin, cout)
    return "sparse_conv_w32_kernel<bf16,%d,%d>" % (cin, cout)


class Watchdog:
    """Host-side guard of the timed loop: a daemon thread that, when nothing has ticked for `timeout_s`, prints every stream's last
    COMPLETED step (events recorded at the end of each step's enqueues, queried without blocking) and ends the process with a fresh
    exit (os._exit, never a re-exec) -- so a stalled schedule costs seconds and names the stream that stopped, instead of sitting
    until the harness's limit (round 2's high-priority-geometry run did that: DESIGN.md section 7c)."""

    def __init__(self, timeout_s):
        import threading
        self.timeout_s, self.last, self.label = timeout_s, time.monotonic(), "start"
        self.marks = {}                                        # stream name -> [(step, event)]
        self.done = False
        if timeout_s > 0:
            threading.Thread(target=self._run, daemon=True).start()

    def tick(self, label):
        self.last, self.label = time.monotonic(), label

    def mark(self, name, stream, step):
        ev = torch.cuda.Event()
        ev.record(stream)
        lst = self.marks.setdefault(name, [])
        lst.append((step, ev))
        del lst[:-64]

    def _run(self):
        while not self.done:
            time.sleep(1.0)
            if time.monotonic() - self.last > self.timeout_s:
                state = {}
                for name, lst in self.marks.items():
                    finished = [st for st, ev in lst if ev.query()]
                    state[name] = {"last_completed_step": max(finished) if finished else None, "last_enqueued_step": lst[-1][0] if lst else None}
                print(json.dumps({"watchdog": "no progress for %.0f s after '%s'" % (self.timeout_s, self.label), "streams": state}),
                      file=sys.stderr, flush=True)
                os._exit(3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=24, help="frames per GPU per step")
    ap.add_argument("--no-vox-ahead", action="store_true",
                    help="voxelize each batch inside its own step instead of one step ahead, beside the previous batch's rulebooks")
    ap.add_argument("--geo-after", type=int, default=6,
                    help="start the next batch's voxelization after this many conv launches of the current batch (0 = at once; 6..7 measured "
                         "best for this backbone with the in-kernel NMS front end and the rolling 32->32 kernel: profiles/r02_schedule_sweep.txt)")
    ap.add_argument("--cap-factor", type=float, default=2.0, help="row capacity of a strided conv's output relative to its input's")
    ap.add_argument("--pool", type=int, default=0, help="distinct synthetic frames per GPU (0 = 2 x batch: consecutive steps see different frames)")
    ap.add_argument("--geo-priority", default="normal", choices=("normal", "high"),
                    help="priority of the geometry stream (diagnostic: 'high' is the setting that hung in round 2; run it only under the watchdog)")
    ap.add_argument("--watchdog-s", type=float, default=60.0, help="exit non-zero, with every stream's last completed step, after this many "
                    "seconds without progress (0 = off)")
    ap.add_argument("--nms-stream", default="own", choices=("own", "vox"), help="stream of the NMS launches: its own, or the voxelizer's")
    ap.add_argument("--nms-first", action="store_true", help="issue the NMS launches before the plan's (diagnostic)")
    ap.add_argument("--nms-after", type=int, default=14, help="hold the NMS back until this conv layer of the batch has been reached (0 = do not)")
    ap.add_argument("--graph", action="store_true", help="capture one plan's step -- convs, geometry stream and NMS stream forked from and joined "
                    "to the capturing stream -- in a HIP graph and replay it.  The default launches eagerly (the step is device-bound) and "
                    "alternates two plans, whose cross-step event waits a single capture cannot hold")
    ap.add_argument("--flat-priority", action="store_true", help="do not raise the priority of the conv stream (A/B switch)")
    ap.add_argument("--no-pipeline", action="store_true", help="one plan: geometry overlaps only the convs of its own batch")
    ap.add_argument("--no-overlap", action="store_true", help="one stream: no geometry / conv / NMS overlap (A/B switch)")
    ap.add_argument("--cpu-frames", type=int, default=20, help="timed frames of the CPU-oracle baseline after 3 warm-up frames (0 = skip); ~55 s of CPU work")
    ap.add_argument("--cpu-single-frames", type=int, default=1, help="also time this many frames with the oracle on ONE thread (~35 s each)")
    ap.add_argument("--profile-passes", type=int, default=5)
    args = ap.parse_args()

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hot path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rank, world = pdist.init_from_env(dev)   # RCCL ("nccl") when launched by torchrun; used for the timing bracket only
    dist = torch.distributed if world > 1 else None
    assert args.gpus == world, "--gpus %d but WORLD_SIZE=%d" % (args.gpus, world)

    cfg = synth.CONFIGS[WORKLOAD]
    B = args.batch
    P = max(args.pool if args.pool > 0 else 2 * B, B)
    net = build_model(cfg, dev)
    eng = engine.HotPathEngine(net, B, cfg["range"], cfg["voxel"], cfg["T"], cfg["max_voxels"], cfg["C"], cfg["n_points"],
                               dtype=torch.bfloat16, device=dev, level_capacity_factor=args.cap_factor)
    nms = engine.BatchedNMS(B * HEADS, BOXES_PER_HEAD, NMS_THRESH, NMS_POST, device=dev)
    pipelined = not (args.no_overlap or args.graph or args.no_pipeline)
    eng2 = None
    if pipelined:   # two plans alternate so that the geometry of batch i+1 runs beside the convolutions of batch i
        eng2 = engine.HotPathEngine(net, B, cfg["range"], cfg["voxel"], cfg["T"], cfg["max_voxels"], cfg["C"], cfg["n_points"],
                                    dtype=torch.bfloat16, device=dev, level_capacity_factor=args.cap_factor)
    lo, hi = torch.cuda.Stream.priority_range()
    if pipelined:
        # FOUR streams by construction -- convs, geometry (both plans: batch i + 1's rulebooks follow batch i's anyway), voxelize-ahead,
        # NMS -- so that no two of them share one of the process's four hardware queues (GPU_MAX_HW_QUEUES); round 2 ran six
        geo_shared = torch.cuda.Stream(device=dev, priority=hi if args.geo_priority == "high" else 0)
        eng.use_geometry_stream(geo_shared)
        eng2.use_geometry_stream(geo_shared)
    frames, boxes, scores, counts = make_inputs(cfg, rank, P, dev)
    # static NMS inputs of one step (what the dense head would have decoded for the B frames)
    s_boxes = torch.zeros((B * HEADS, BOXES_PER_HEAD, 9), device=dev)
    s_scores = torch.zeros((B * HEADS, BOXES_PER_HEAD), device=dev)
    s_counts = torch.zeros((B * HEADS,), dtype=torch.int32, device=dev)

    def ring(step):
        """the step's frames (step * B + b) % P, b < B, as (destination offset, pool offset, length) runs"""
        start = (step * B) % P
        first = min(B, P - start)
        return [(0, start, first)] + ([(first, 0, B - first)] if first < B else [])

    def load_points(step, plan=None):
        # the new batch arrives in the plan's input buffer: one device copy per run instead of one per frame
        plan = eng if plan is None else plan
        for dst, src, k in ring(step):
            plan.points[dst:dst + k].copy_(frames[src:src + k])

    def load_boxes(step):
        for dst, src, k in ring(step):
            s_boxes[dst * HEADS:(dst + k) * HEADS].copy_(boxes[src:src + k].flatten(0, 1))
            s_scores[dst * HEADS:(dst + k) * HEADS].copy_(scores[src:src + k].flatten(0, 1))
            s_counts[dst * HEADS:(dst + k) * HEADS].copy_(counts[src:src + k].flatten(0, 1))

    def load(step):
        load_points(step)
        load_boxes(step)

    result = {}

    def note(msg):
        if rank == 0 and os.environ.get("BENCH_VERBOSE"):
            print("[bench] " + msg, file=sys.stderr, flush=True)

    nms_stream = torch.cuda.Stream(device=dev)

    def hot_path():
        # the NMS of the decoded boxes shares nothing with the backbone: it runs on its own stream beside it
        if args.no_overlap:
            eng.forward(overlap=False)
            result["sel"], result["num"] = nms(s_boxes, s_scores, s_counts)
            return
        cur = torch.cuda.current_stream(dev)
        nms_stream.wait_stream(cur)
        with torch.cuda.stream(nms_stream):
            result["sel"], result["num"] = nms(s_boxes, s_scores, s_counts)
        eng.forward(overlap=True)
        cur.wait_stream(nms_stream)

    graph = None
    note("inputs ready")
    load(0)
    torch.cuda.synchronize()
    note("loaded")
    eng.voxelize()
    torch.cuda.synchronize()
    note("voxelized, rows %s" % eng.level_rows()[:1])
    eng.backbone(hook=(lambda name, fn: (fn(), torch.cuda.synchronize(), note("  " + name))) if os.environ.get("BENCH_VERBOSE") else None)
    torch.cuda.synchronize()
    note("backbone done, rows %s" % eng.level_rows())
    hot_path()   # whole path once, eagerly, before any capture
    torch.cuda.synchronize()
    note("eager pass done")
    if args.graph:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            hot_path()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            hot_path()
        note("graph captured")

    vox_ahead = pipelined and not args.no_vox_ahead
    vox_stream = torch.cuda.Stream(device=dev) if vox_ahead else None
    wd = Watchdog(args.watchdog_s)

    nms_on = vox_stream if (vox_ahead and args.nms_stream == "vox") else nms_stream

    def step(i):
        if pipelined:
            plan = eng if i % 2 == 0 else eng2
            cur = torch.cuda.current_stream(dev)
            def run_nms():
                with torch.cuda.stream(nms_on):   # its inputs are loaded on its own stream: nothing to wait for
                    if args.nms_after > 0:
                        nms_on.wait_event(plan.mark2)   # start beside a chosen conv layer of this batch
                    load_boxes(i)
                    result["sel"], result["num"] = nms(s_boxes, s_scores, s_counts)
            if args.nms_first:
                run_nms()
            nxt = eng2 if i % 2 == 0 else eng
            if vox_ahead and args.geo_after > 0:
                # batch i was voxelized during step i-1; the next batch's voxelizer is held back until a chosen conv layer of
                # this batch has been reached
                plan.forward(overlap=True, pipelined=True, prevoxelized=True, mark_after=args.geo_after, mark2_after=args.nms_after)
                nxt.voxelize_ahead(vox_stream, load=lambda: load_points(i + 1, nxt), after=plan.mark)
            else:
                if vox_ahead:
                    # batch i+1 is voxelized now, on its own stream, beside batch i's rulebooks
                    nxt.voxelize_ahead(vox_stream, load=lambda: load_points(i + 1, nxt))
                else:
                    geo = plan.geometry_stream()
                    with torch.cuda.stream(geo):
                        geo.wait_event(plan._done)      # the plan's previous batch has left these buffers
                        load_points(i, plan)
                plan.forward(overlap=True, pipelined=True, prevoxelized=vox_ahead, mark2_after=args.nms_after)
            if not args.nms_first:
                run_nms()
            cur.wait_stream(nms_on)
            for nm, st_ in (("conv", cur), ("geometry", plan.geometry_stream()), ("voxelize", vox_stream), ("nms", nms_on)):
                if st_ is not None:
                    wd.mark(nm, st_, i)
            wd.tick("step %d enqueued" % i)
            return
        load(i)
        if graph is not None:
            graph.replay()
        else:
            hot_path()

    # the convolutions are the critical path: they get the high-priority stream, geometry and NMS fill in around them
    conv_stream = torch.cuda.Stream(device=dev, priority=hi) if (pipelined and not args.flat_priority) else torch.cuda.current_stream(dev)
    conv_stream.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(conv_stream):
        if vox_ahead:      # the first batch's voxelization; from then on every step voxelizes the batch after its own
            eng.voxelize_ahead(vox_stream, load=lambda: load_points(0, eng))
        for i in range(args.warmup):
            step(i)
        # barrier + device sync on both sides, MAX over ranks (openpcdet_amd/dist.py; tests/test_dist_gloo.py runs it at world 2)
        host_ms = []

        def timed_step(i):
            h0 = time.perf_counter()
            step(i)
            host_ms.append((time.perf_counter() - h0) * 1e3)

        wd.tick("warm-up enqueued")
        elapsed = pdist.timed_steps(timed_step, args.steps, device=dev, first=args.warmup)
        wd.tick("timed region done")
        if os.environ.get("PVX_FIN_DUMP"):      # diagnostic build (tools/build_fin.sh): how evenly the rolling / streaming conv launches INSIDE the
            import ctypes                       # overlapped schedule start and finish their workgroups (ring of the last 64 launches per family)
            import numpy as np
            # keep the pipeline full behind the launches that are looked at: a few more steps, then everything but the last ones counts
            for i in range(args.warmup + args.steps, args.warmup + args.steps + 4):
                step(i)
            torch.cuda.synchronize()
            from openpcdet_amd import _lib as _l
            buf = (ctypes.c_ulonglong * (2 * 64 * 1024 * 2))()
            ctr = (ctypes.c_uint * 2)()
            assert ctypes.CDLL(_l.LIB_PATH).pvx_debug_fin(buf, ctr) == 0
            a = np.frombuffer(buf, dtype=np.uint64).reshape(2, 64, 1024, 2).astype(np.float64)
            for nm, k in (("streaming 128->128 / 64->128", 0), ("rolling 32->32", 1)):
                rows = []
                for slot in range(64):
                    w = a[k, slot][a[k, slot][:, 1] > 0]
                    if len(w) < 8:
                        continue
                    t0 = w[:, 0].min()
                    st, en = (w[:, 0] - t0) / 100.0, (w[:, 1] - t0) / 100.0
                    rows.append((t0, np.percentile(st, 50), np.percentile(st, 90), st.max(), en.min(), en.mean(), en.max(), (en - st).mean()))
                rows.sort()
                rows = np.array(rows[:-10])                  # the last launches ran with nothing behind them
                print("[fin] %-30s %3d launches | start p50 %.1f p90 %.1f max %.1f us | finish min %.0f mean %.0f max %.0f us | own run time %.0f us | idle at the tail %.1f %%"
                      % ((nm, len(rows)) + tuple(rows[:, 1:].mean(axis=0)) + (100 * (1 - rows[:, 5].mean() / rows[:, 6].mean()),)), file=sys.stderr, flush=True)
        # the conv launches' durations INSIDE this schedule (beside the geometry / NMS streams): six more steps with an event pair
        # around every conv launch; outside the timed region
        ov_events = []
        if pipelined:
            eng.conv_events, eng2.conv_events = ov_events, ov_events
            for i in range(args.warmup + args.steps, args.warmup + args.steps + 6):
                step(i)
            torch.cuda.synchronize()
            eng.conv_events, eng2.conv_events = None, None
            wd.tick("overlapped conv timing done")
        note("host enqueue ms per step inside the timed region: first %s ... last %s" % (
            ["%.2f" % v for v in host_ms[:4]], ["%.2f" % v for v in host_ms[-4:]]))
        if os.environ.get("BENCH_HOSTTIME"):      # how long the HOST spends enqueuing one step (must stay well below the step)
            ht = []
            for i in range(8):
                h0 = time.perf_counter()
                step(args.warmup + args.steps + i)
                ht.append((time.perf_counter() - h0) * 1e3)
            torch.cuda.synchronize()
            note("host enqueue ms per step: %s" % ["%.2f" % v for v in ht])
    torch.cuda.current_stream(dev).wait_stream(conv_stream)
    wd.done = True
    note("stream priority range %s" % ((lo, hi),))
    note("timed region done: %.3f s" % elapsed)
    assert not eng.overflowed() and (eng2 is None or not eng2.overflowed()), "a level overflowed its row capacity"
    rows = eng.level_rows()

    out = None
    if rank == 0:
        # ---- the last timed step's results against a serial, single-stream re-run of the same batch on the same kernels: the
        #      overlapped schedule (4 streams, 2 plans, or the graph replay) must not change a bit ----
        i_last = args.warmup + args.steps - 1 + (6 if pipelined else 0)
        plan_last = (eng if i_last % 2 == 0 else eng2) if pipelined else eng
        bev_last, sel_last, num_last = plan_last.dense.clone(), result["sel"].clone(), result["num"].clone()
        load(i_last)
        eng.forward(overlap=False)
        sel_s, num_s = nms(s_boxes, s_scores, s_counts)
        torch.cuda.synchronize()
        checked = {"bev_bitwise_equal_to_serial_rerun": bool(torch.equal(bev_last, eng.dense)),
                   "nms_equal_to_serial_rerun": bool(torch.equal(num_last, num_s) and torch.equal(sel_last, sel_s)),
                   "bev_nonzero": int((bev_last != 0).sum().item())}
        assert checked["bev_bitwise_equal_to_serial_rerun"] and checked["nms_equal_to_serial_rerun"] and checked["bev_nonzero"] > 0, checked
        # ---- instrumented pass: HIP events around every launch of the plan, on the stream they are launched on ----
        ev = {}

        def hook(name, fn):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            ev.setdefault(name, []).append((a, b))

        stage_ev = {}

        def timed(name, fn):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            stage_ev.setdefault(name, []).append((a, b))

        for p in range(args.profile_passes + 1):
            load(p)
            if p == 1:          # pass 0 is a warm-up of the eager path
                ev.clear()
                stage_ev.clear()
            timed("voxelize", eng.voxelize)
            eng.backbone(hook=hook)
            timed("nms", lambda: nms(s_boxes, s_scores, s_counts))
        torch.cuda.synchronize()
        passes = args.profile_passes
        conv_ms = sum(a.elapsed_time(b) for n, l in ev.items() if n.startswith("conv") and not n.startswith("conv_out") for a, b in l) / passes
        conv_launches = sum(1 for n in ev if n.startswith("conv") and not n.startswith("conv_out"))
        rule_ms = sum(a.elapsed_time(b) for n, l in ev.items() if n in ("site_index", "subm_rulebook", "conv_out_index", "conv_out_rulebook", "tile_schedule", "index_unset", "index_unset_main") for a, b in l) / passes
        dense_ms = sum(a.elapsed_time(b) for a, b in ev.get("dense", [])) / passes
        vox_ms = sum(a.elapsed_time(b) for a, b in stage_ev["voxelize"]) / passes
        nms_ms = sum(a.elapsed_time(b) for a, b in stage_ev["nms"]) / passes
        flops, per_layer = eng.conv_flops()     # of the last loaded batch
        achieved = flops / (conv_ms * 1e-3) / 1e12
        # per kernel: launches of one step grouped by the HIP kernel they run; algorithmic flops 2 P Cin Cout and compulsory bytes
        # (N_in Cin e + N_out Cout e + K N_out 4 [+ residual rows]) per launch over the launch's event-timed duration
        groups = {}
        for (lname, n_rows, pairs, f), rec in zip(per_layer, eng.conv_records):
            kname = conv_kernel_name(rec["cin"], rec["cout"], rec["kvol"])
            n_in = min(int(rec["level_in"].n_dev.item()), rec["level_in"].cap)
            byts = n_in * rec["cin"] * 2 + n_rows * rec["cout"] * 2 * (2 if rec["residual"] else 1) + \
                rec["kvol"] * n_rows * (2 if rec["rcap"] > 0 else 4) + rec["kvol"] * rec["cin"] * rec["cout"] * 2    # (u16 codes where a tile book is read)
            # SURVEY.md 8d's compulsory bytes: features in + features out + weights, no rulebook, no residual
            comp = n_in * rec["cin"] * 2 + n_rows * rec["cout"] * 2 + rec["kvol"] * rec["cin"] * rec["cout"] * 2
            us = sum(a.elapsed_time(b) for a, b in ev[lname]) / passes * 1e3
            g = groups.setdefault(kname, dict(kernel=kname, launches_per_step=0, flops=0, bytes=0, comp=0, us=0.0))
            g["launches_per_step"] += 1
            g["flops"] += f
            g["bytes"] += byts
            g["comp"] += comp
            g["us"] += us
        # HBM bytes per launch from the committed PMC passes of this same command (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
        # runs, FETCH_SIZE doubled as the gfx950 note prescribes: tools/pmc_traffic.py); null when the file is absent
        traffic, traffic_src, pmc_by_kernel = None, None, {}
        for tag in ("r03", "r02", "r01"):
            tpath = os.path.join(ROOT, "profiles", "%s_pmc_traffic_b%d.json" % (tag, B))
            if os.path.exists(tpath):
                rows_t = [r for r in json.load(open(tpath)) if r["kernel"].startswith("sparse_conv")]
                passes_t = max(1, min(r["launches"] for r in rows_t))
                tot_b = sum((r["fetch_bytes_per_launch"] + r["write_bytes_per_launch"]) * r["launches"] / passes_t for r in rows_t)
                n_l = sum(r["launches"] / passes_t for r in rows_t)
                traffic, traffic_src = round(tot_b / n_l), "profiles/" + os.path.basename(tpath)
                for r in rows_t:
                    pmc_by_kernel[r["kernel"]] = round(r["fetch_bytes_per_launch"] + r["write_bytes_per_launch"])
                break

        def pmc_for(kname):     # tools/pmc_traffic.py keys: sparse_conv_slab<cin=64,cout=64> / sparse_conv_w32<cin=..,cout=..>
            ch = kname.split("<")[1].rstrip(">").split(",")[1:]
            kind = "sparse_conv_slab" if "slab" in kname else "sparse_conv_w32"
            return pmc_by_kernel.get("%s<cin=%s,cout=%s>" % (kind, ch[0], ch[1]))

        per_kernel = []
        for g in sorted(groups.values(), key=lambda g: -g["us"]):
            n = g["launches_per_step"]
            per_kernel.append({"kernel": g["kernel"], "launches_per_step": n, "flops_per_launch": int(g["flops"] / n),
                               "avg_us": round(g["us"] / n, 1), "tflops": round(g["flops"] / (g["us"] * 1e-6) / 1e12, 1),
                               "frac": round(g["flops"] / (g["us"] * 1e-6) / 1e12 / PEAK_BF16_TFLOPS, 4),
                               "alg_bytes_per_launch": int(g["bytes"] / n), "compulsory_bytes_per_launch": int(g["comp"] / n),
                               "alg_GBps": round(g["bytes"] / (g["us"] * 1e-6) / 1e9), "pmc_bytes_per_launch": pmc_for(g["kernel"]),
                               "pmc_over_compulsory": (round(pmc_for(g["kernel"]) / (g["comp"] / n), 2) if pmc_for(g["kernel"]) else None)})
        dom = per_kernel[0]
        # the same sum from the event pairs recorded INSIDE the overlapped schedule (six steps behind the timed region)
        frac_ov, conv_ms_ov = None, None
        if ov_events:
            conv_ms_ov = sum(a.elapsed_time(b) for _, a, b in ov_events) / 6.0
            frac_ov = round(flops / (conv_ms_ov * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 5)
        comp_total = sum(g["comp"] for g in groups.values())
        roofline = {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / PEAK_BF16_TFLOPS, 5), "frac_overlapped": frac_ov,
                    "ms_per_step_overlapped": round(conv_ms_ov, 4) if conv_ms_ov else None,
                    "compulsory_bytes_per_step": int(comp_total),
                    "sustained_mfma_peak_note": "with every CU issuing v_mfma_f32_32x32x16_bf16 back to back this part holds 1.6 GHz = 1650 TFLOP/s "
                                                "(tools/probes/mfma_rate.hip); `peak` stays the guide's 2500",
                    "traffic": traffic, "traffic_source": traffic_src,
                    "basis": "serial: one stream, HIP events around each conv launch of the same plan, %d passes after the timed region; "
                             "frac = sum of algorithmic flops / sum of those durations (all %d conv launches of a step)" % (passes, conv_launches),
                    "kernel": "all sparse-conv launches of a step; dominant: %s" % dom["kernel"],
                    "dominant": {"kernel": dom["kernel"], "achieved": dom["tflops"], "frac": dom["frac"], "avg_us": dom["avg_us"],
                                 "launches_per_step": dom["launches_per_step"]},
                    "flops_per_step": flops, "ms_per_step": round(conv_ms, 4), "per_kernel": per_kernel}
        # the HBM-bound stages against the 8 TB/s roof, ALGORITHMIC bytes over the stage's own launch time
        n_pts = sum(int(f.shape[0]) for f in frames[:B])
        m0 = rows[0]
        vox_bytes = n_pts * 4 * cfg["C"] + n_pts * 4 + m0 * (16 + 4 * cfg["C"])
        # rulebooks as the plan stores them: coordinates read (and, for a strided conv, written) + per output site either the int32 table
        # (4 B x K) or, where every reader goes through one tile book, the book's u16 codes (2 B x 27; the far entries are a few per cent)
        rule_bytes, seen = 0, set()
        for rec in eng.conv_records:
            if rec["nbr"].data_ptr() in seen:
                continue
            seen.add(rec["nbr"].data_ptr())
            n_o = min(int(rec["level_out"].n_dev.item()), rec["level_out"].cap)
            n_i = min(int(rec["level_in"].n_dev.item()), rec["level_in"].cap)
            per_site = 27 * 2 if rec["far_only"] else rec["kvol"] * 4 + (27 * 2 if rec["rcap"] > 0 else 0)
            rule_bytes += n_o * 16 + (0 if rec["subm"] else n_i * 16) + per_site * n_o
        hbm = {"peak_GBps": 8000.0,
               "voxelize": {"bytes": vox_bytes, "GBps": round(vox_bytes / (vox_ms * 1e-3) / 1e9, 1)},
               "rulebook": {"bytes": rule_bytes, "GBps": round(rule_bytes / (rule_ms * 1e-3) / 1e9, 1)}}
        cpu = None
        if args.cpu_frames > 0 and world == 1:     # reported at N = 1 only
            cpu = cpu_baseline(cfg, net.cpu().float(), args.cpu_frames, boxes.cpu().numpy(), scores.cpu().numpy(), counts.cpu().numpy(),
                               single_thread_frames=args.cpu_single_frames)
        value = world * B * args.steps / elapsed
        out = {
            "metric": "frames/sec voxelize+sparse backbone+NMS (nuScenes CenterPoint)",
            "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "CenterPoint nuScenes 10-sweep 300k pts/frame, 0.075 m voxels, VoxelResBackBone8x, "
                                   "6 heads x 500 boxes NMS @0.2 (BASELINE.json configs[2])",
                       "frames_per_gpu_per_step": B, "global_batch": world * B, "parallelism": "frame-sharded x%d" % world,
                       "hip_graph": graph is not None, "streams": 1 if args.no_overlap else (4 if vox_ahead else 3), "plans_in_flight": 2 if pipelined else 1,
                       "frame_pool": P,
                       "voxelize_ahead": bool(vox_ahead), "geo_after": args.geo_after, "rows_per_level": rows},
            "roofline": roofline, "hbm_stages": hbm, "cpu_baseline": cpu, "checked": checked,
            "stages": {"voxelize_ms": round(vox_ms, 4), "rulebook_ms": round(rule_ms, 4), "conv_ms": round(conv_ms, 4),
                       "dense_ms": round(dense_ms, 4), "nms_ms": round(nms_ms, 4)},
        }
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
