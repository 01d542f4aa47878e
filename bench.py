"""Benchmark of the contour-deformation hot path (BASELINE.json metric: contours deformed/sec,
128 vertices, all iterations; gather HBM GB/s).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|cfg3|cfg5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`--gpus N` with N > 1 and no torch.distributed environment starts the N ranks itself, one
process per GPU (as the reference's train_net.py:76-83 `launch(main, args.num_gpus, ...)` does):
the parent never touches the GPU, runs `python -m torch.distributed.run` as a child process and
exits with its status; rank 0's JSON line passes through.  Under an existing launcher
(WORLD_SIZE set) `--gpus` must equal WORLD_SIZE.  `--force-dist` takes the same route with ONE
rank: launcher child, `init_process_group("nccl", device_id=...)`, the counts + padded
`all_gather_into_tensor` of the exchange step and the barriers all run at world size 1 -- the
whole multi-rank path on a single GPU (tests/test_gpu_dist1.py).

Workloads
  cfg2 (default; the configuration the BASELINE metric is quoted on), weak scaling: per GPU,
       DANCE R50 snake head, fp32, 16 images x 100 instances x 128 vertices on synthetic
       800x1216 FPN features (16, 1+256, 200, 304), attention + individual scale, 3 iterations.
  cfg3 the same at 32 images per GPU.
  cfg5 strong scaling: ONE fixed sweep of 256 images x 300 instances x 256 vertices; rank r
       holds and deforms the images `distributed.image_range(256, r, world)`, then every rank
       receives the whole deformed set (counts all-gather + padded RCCL all-gather).
One step = the reference's `snake_deformation` scope (snake_head.py:1822-1862) over the rank's
batch, starting from the NCHW (attention + features) tensor resident in HBM: layout pass,
3 x (fused gather + head block, 7 residual blocks, fusion + max, prediction, offset
update) and, for N > 1, the RCCL all-gather of the deformed contours.

Rank 0 prints ONE JSON line; see DESIGN.md "Measurement" for every field.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

IMAGES, PER_IMAGE, VERTS, CH, HF, WF, ITERS = 16, 100, 128, 256, 200, 304, 3
# MACs per vertex per iteration of one _SnakeNet (C = 256)
MACS_RES_BLOCK = 128 * 128 * 9
MACS_PER_VERTEX = (CH + 2) * 128 * 9 + 7 * MACS_RES_BLOCK + 1024 * 256 + 1280 * 256 + 256 * 64 + 64 * 2
GATHER_BYTES_PER_VERTEX = 4 * (CH + 1) * 4            # 4 taps x 257 channels x fp32
PEAK_F32_MFMA_TFLOPS = 157.3                          # MI355X_MICROARCH.md, f32-input MFMA
PEAK_F16_MFMA_TFLOPS = 2500.0                         # MI355X_MICROARCH.md, BF16/FP16 dense
PEAK_HBM_GBPS = 8000.0
# images (per GPU when weak, in total when strong), instances per image, vertices, scaling
WORKLOADS = {"cfg2": (16, 100, 128, "weak"), "cfg3": (32, 100, 128, "weak"), "cfg5": (256, 300, 256, "strong")}
WORKLOAD_NAMES = {"cfg2": "BASELINE config 2", "cfg3": "BASELINE config 3", "cfg5": "BASELINE config 5"}
# contours per kernel sequence above which a rank's batch is deformed in chunks (workspace:
# 0.83 MB per 128-vertex contour, 1.65 MB per 256-vertex one); a multiple of the 256 CUs, so that
# every chunk but the last is whole rounds of workgroups (38 x 256)
CHUNK_CAP = 9728


def measured_traffic(kernel, workgroups):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/hbm_traffic.json,
    written by tools/traffic_json.py from separate rocprofv3 --pmc runs of this same command:
    counters cannot be read inside the timed run).  None when no pass covers this launch shape."""
    try:
        with open(os.path.join(ROOT, "profiles", "hbm_traffic.json")) as f:
            t = json.load(f)
    except OSError:
        return None, None
    for key, val in t["kernels"].items():
        name, wg = key.split("|")
        if (name == kernel[:-1] if kernel.endswith("|") else name.startswith(kernel)) and int(wg) == workgroups:
            return val["bytes"], t["source"]
    return None, None


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="cfg2")
    ap.add_argument("--images", type=int, default=None,
                    help="images per GPU (weak workloads) / in the whole sweep (cfg5)")
    ap.add_argument("--per-image", type=int, default=None)
    ap.add_argument("--vertices", type=int, default=None)
    ap.add_argument("--dry-run-cpu", action="store_true",
                    help="launcher / shard / all-gather rehearsal on the CPU (gloo, a stub in place of the "
                         "HIP deformer, tiny sizes); prints a line marked dry_run, never a measurement")
    ap.add_argument("--layout", choices=("nchw", "nhwc"), default="nchw",
                    help="nchw (default): the layout pass is inside the timed step")
    ap.add_argument("--no-fused-head", action="store_true")
    ap.add_argument("--precision", choices=("fp16x3", "fp32", "fp16"), default="fp16x3",
                    help="fp16x3 (default): split-half operands, 3 f16 MFMAs per product, fp32 "
                         "accumulate (fp32-level accuracy); fp32: exact-fp32 MFMA kernels; fp16: "
                         "APPROXIMATE plain fp16 operands (~1e-2 coordinate error, outside the parity "
                         "bar: the half-precision variant of BASELINE config 2, not a headline number)")
    ap.add_argument("--streams", type=int, default=1,
                    help="deform the contours in this many parts on separate HIP streams (no per-stage "
                         "events then; measured -4 %% wall with 2)")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the multi-rank path whatever --gpus is: ranks started by the launcher child, RCCL "
                         "process group, exchange-step collectives and barriers also at world size 1")
    ap.add_argument("--dump-output", default=None,
                    help="rank 0 saves its gathered output contours to this file (torch.save) -- for comparing runs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-feat-prep", action="store_true", help="skip the untimed bottom_out measurement")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip every untimed extra (two streams, feature prep, attender, RLE, FCOSE decode)")
    ap.add_argument("--cpu-reps", type=int, default=40,
                    help="timed oracle runs of config 1 (about 10 s of CPU work in total)")
    args = ap.parse_args(argv)
    b, per, v, scaling = WORKLOADS[args.workload]
    args.images = b if args.images is None else args.images
    args.per_image = per if args.per_image is None else args.per_image
    args.vertices = v if args.vertices is None else args.vertices
    args.scaling = scaling
    return args


def launch_ranks(args, argv):
    """Start `args.gpus` ranks of this script, one process per GPU, and exit with their status.
    Counterpart of the reference's own launcher (train_net.py:76-83).  The parent touches no GPU
    (torch.cuda.device_count() does not initialise one) and replaces no process: the launcher is
    a child, its stdout (rank 0's JSON line) and stderr are inherited."""
    if not args.dry_run_cpu:
        have = torch.cuda.device_count()
        if have < args.gpus:
            sys.exit("bench.py: --gpus %d but only %d GPU(s) visible" % (args.gpus, have))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, min(16, (os.cpu_count() or 1) // args.gpus))))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    sys.exit(subprocess.run(cmd, env=env).returncode)


def dry_run_cpu(args, rank, world):
    """The multi-rank plumbing of this script -- launcher, by-image shard, counts + contour
    all-gather, max-over-ranks timing, one JSON line -- on the CPU with gloo and a stub in place
    of the HIP deformer (every vertex moves by (image + 1, 2 * image + 1)), at toy sizes."""
    from dance_amd.distributed import ImageShard, all_gather_contours, gather_shards

    grouped = world > 1 or args.force_dist
    if grouped:
        dist.init_process_group("gloo")
    b, v = 8, 8
    per = [1 + (i * 5) % 4 for i in range(b)]                     # ragged instances per image
    if args.scaling == "weak":
        per = [3] * b
    n = sum(per)
    g = torch.Generator().manual_seed(7)
    full = torch.rand((n, v, 2), generator=g) * 100
    image_idx = torch.repeat_interleave(torch.arange(b), torch.tensor(per)).float()

    def stub(c, idx):
        return c + torch.stack([idx + 1, 2 * idx + 1], 1)[:, None, :]

    if args.scaling == "strong":
        shard = ImageShard(image_idx, b, rank, world)
        lo = shard.images[0]
        out = gather_shards(stub(shard.take(full), shard.local_image_idx + lo))
        want = stub(full, image_idx)
    else:
        out = all_gather_contours(stub(full + rank, image_idx))
        want = torch.cat([stub(full + r, image_idx) for r in range(world)])
    ok = torch.equal(out, want)
    t = torch.tensor([float(ok)])
    if grouped:
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if rank == 0:
        print(json.dumps({"dry_run": True, "metric": "DRY RUN on the CPU (gloo, stub deformer): not a measurement",
                          "n_gpus": world, "gpus_flag": args.gpus, "scaling": args.scaling,
                          "contours": int(out.size(0)), "ok": bool(t.item() == 1.0),
                          "process_group": grouped}))
    if grouped:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if t.item() == 1.0 else 1


def cpu_baseline(reps):
    """The oracle (torch-CPU restatement of the reference path, parity-pinned by
    tests/golden) on BASELINE config 1: 1 image, 100 instances x 128 vertices, 3 iterations."""
    from dance_amd import config, modules, synthetic
    from oracle import snake_ref

    torch.set_flush_denormal(True)
    cfg = config.dance_r50(num_sampling=VERTS, num_convs=0)
    sd = {k: v.clone() for k, v in modules.RefineNet(cfg).state_dict().items()}
    synthetic.fill_state_dict_(sd, 1234)
    feats = synthetic.make_features(1, CH, HF, WF, 1234)
    boxes = synthetic.make_boxes(100, HF * 4, WF * 4, 4321)
    init, _ = snake_ref.uniform_upsample(snake_ref.box_polygon(boxes)[None], VERTS)
    whs = torch.stack([boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]], 1)
    times = []
    with torch.no_grad():
        for r in range(reps + 1):
            t0 = time.perf_counter()
            snake_ref.dance_deform(sd, feats, init, None, (HF * 4, WF * 4), whs)
            if r:
                times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {"value": 100.0 / med, "unit": "contours/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "BASELINE config 1: 1 image (1,257,200,304), 100 contours x 128 vertices x 3 "
                      "iterations; oracle/snake_ref.py, median of %d runs after 1 warm-up, "
                      "flush_denormal on" % reps,
            "seconds_per_run": med}


def main():
    argv = sys.argv[1:]
    args = parse(argv)
    if (args.gpus > 1 or args.force_dist) and "WORLD_SIZE" not in os.environ:
        launch_ranks(args, argv)                                 # does not return
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        sys.exit("bench.py: --gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    if args.dry_run_cpu:
        sys.exit(dry_run_cpu(args, rank, world))
    dist_on = world > 1 or args.force_dist
    if dist_on:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        assert dist.get_world_size() == args.gpus
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from dance_amd import config, modules, ops, snake, synthetic
    from dance_amd._lib import EVOLVE_EVENTS, STAGE_NAMES
    from dance_amd.distributed import CALLS, ImageShard, all_gather_contours, gather_shards

    per, v = args.per_image, args.vertices
    h, w = HF * 4, WF * 4
    strong = args.scaling == "strong"
    if strong:
        # ONE sweep of args.images images for the whole job; this rank holds a block of them
        b_total = args.images
        n_total = b_total * per
        boxes_all = synthetic.make_boxes(n_total, h, w, 4321)
        idx_all = torch.arange(b_total).repeat_interleave(per).float()
        shard = ImageShard(idx_all, b_total, rank, world)
        img_lo, img_hi = shard.images
        b = img_hi - img_lo
        boxes = shard.take(boxes_all)
        image_idx = shard.local_image_idx.to(dev)
        del boxes_all, idx_all
    else:
        b = args.images
        n_total = world * b * per
        boxes = synthetic.make_boxes(b * per, h, w, 4321 + rank)
        image_idx = torch.arange(b).repeat_interleave(per).float().to(dev)
    n = boxes.size(0)
    cfg = config.dance_r50(num_sampling=v, num_convs=0)
    sd = {k: t.clone() for k, t in modules.RefineNet(cfg).state_dict().items()}
    synthetic.fill_state_dict_(sd, 1234)
    nets = [snake.SnakeNetWeights(snake.sub_state_dict(sd, "deformer%d." % i), dev) for i in range(ITERS)]
    deformer = snake.ContourDeformer(nets, fused_head=not args.no_fused_head, precision=args.precision)
    # one kernel sequence per iteration up to CHUNK_CAP contours, chunks of CHUNK_CAP beyond
    single_chunk = n <= CHUNK_CAP
    deformer.MAX_CHUNK = n if single_chunk else CHUNK_CAP
    # deform() peels n mod CUs contours off to a side stream so that the main stream's kernels run whole
    # rounds of workgroups (ContourDeformer.peel_count; DANCE_PEEL=0: off).  The timed events sit on the
    # main stream: they time launches of n - side_n contours, and the roofline uses that count.
    side_n = deformer.peel_count(n, dev) if single_chunk and args.streams == 1 else 0
    n_ev = min(n, CHUNK_CAP) - side_n                         # contours of the launches the events time

    # synthetic inputs, resident in HBM before the timed region
    feats = torch.empty((b, CH + 1, HF, WF), dtype=torch.float32, device=dev)
    if strong:
        # drawn on the device from the GLOBAL image number: a rank's images do not depend on the split
        for i in range(b):
            g = torch.Generator(device=dev).manual_seed(1234 + img_lo + i)
            feats[i].normal_(generator=g)
            feats[i, 0].uniform_(generator=g)
    else:
        for i in range(b):                                   # per image: bounds host memory
            feats[i] = synthetic.make_features(1, CH, HF, WF, 1234 + 1000 * rank + i)[0].to(dev)
    whs = torch.stack([boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]], 1).to(dev)
    init, _ = ops.uniform_upsample(boxes.to(dev), v, kind=ops.POLY_BOX)
    sizes = [(h, w)] * b

    # Timed steps carry three events per iteration only (an event record costs ~4 us of stream
    # time): before the head kernel, after it, after the last residual block -- the gather
    # figure and the roofline of the dominant kernel.  The full per-stage breakdown comes from
    # an instrumented pass after the timed region.  A batch deformed in chunks (n > CHUNK_CAP)
    # carries no events in the timed steps; the instrumented pass runs its first chunk.
    TIMED = (0, 1, 8)
    events = [[[torch.cuda.Event(enable_timing=True) if i in TIMED else None for i in range(EVOLVE_EVENTS)]
               for _ in range(ITERS)] for _ in range(args.steps)] if single_chunk else None
    events_full = [[[torch.cuda.Event(enable_timing=True) for _ in range(EVOLVE_EVENTS)]
                    for _ in range(ITERS)] for _ in range(args.steps)]
    for step_ev in (events or []) + events_full:
        for it_ev in step_ev:
            for e in it_ev:
                if e is not None:
                    e.record()                               # creates the handle
    layout_ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                 for _ in range(args.steps)]
    pre = deformer.prepare_features(feats) if args.layout == "nhwc" else None

    def step(ev=None, lev=None):
        if pre is None:
            if lev:
                lev[0].record()
            f = deformer.prepare_features(feats)
            if lev:
                lev[1].record()
        else:
            f = pre
        preds = deformer.deform(f, init, image_idx, sizes, whs, layout="nhwc",
                                stage_events=ev if args.streams == 1 else None, streams=args.streams)
        # the path's one exchange step: every rank receives all deformed contours
        return gather_shards(preds[-1]) if strong else all_gather_contours(preds[-1])

    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    calls0 = CALLS["all_gather_into_tensor"]
    t0 = time.perf_counter()
    for k in range(args.steps):
        out = step(events[k] if events else None, layout_ev[k])
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timed_collectives = CALLS["all_gather_into_tensor"] - calls0
    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert out.shape == (n_total, v, 2) and bool(torch.isfinite(out).all())
    mine = out[shard.index.to(dev)] if strong else out[rank * n:(rank + 1) * n]   # this rank's rows of the gathered set
    if rank == 0 and args.dump_output:
        torch.save(out.cpu(), args.dump_output)
    if rank == 0:
        # instrumented single-stream pass (every stage boundary) over the first chunk; with
        # --streams > 1 or a chunked batch it also supplies the three timed events
        for k in range(args.steps):
            f = deformer.prepare_features(feats) if pre is None else pre
            deformer.deform(f, init[:n_ev], image_idx[:n_ev], sizes, whs[:n_ev], layout="nhwc",
                            stage_events=events_full[k])
        torch.cuda.synchronize()
        if args.streams > 1 or events is None:
            events = events_full

    # the same steps with the contours split over two HIP streams (bit-identical results; one
    # half's kernel tails and launch gaps overlap the other half's kernels).  Reported beside
    # the headline, never as it: per-kernel event timing is only meaningful on one stream.
    two_streams = None
    # the untimed extras (stages either side of the path) run at the one-GPU weak workloads only
    # (and never in a multi-rank run: the other ranks would sit in the final barrier meanwhile)
    extras = rank == 0 and world == 1 and not strong and b <= 32 and not args.no_extras
    if extras and args.streams == 1:
        def step2():
            f = deformer.prepare_features(feats) if pre is None else pre
            return deformer.deform(f, init, image_idx, sizes, whs, layout="nhwc", streams=2)[-1]
        for _ in range(2):
            out2 = step2()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(args.steps):
            out2 = step2()
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t2
        assert torch.equal(out2, mine), "two-stream deformation differs from the single-stream one"
        two_streams = {"value": n * args.steps / dt2, "unit": "contours/s", "ms_per_step": dt2 / args.steps * 1e3,
                       "note": "deform(..., streams=2) / DANCE_STREAMS=2, same steps, results bit-identical"}

    # the gather stage on its own (k_gather: 4 taps x 257 channels per vertex -> (N, 258, V)),
    # outside the timed region: the HBM-bound form of the stage that k_gather_head* fuses
    gather_alone_ms = None
    if rank == 0:
        f_nhwc, att_plane = deformer.prepare_features(feats) if pre is None else pre
        idx32 = image_idx.to(torch.int32)
        for _ in range(2):
            ops.gather_features(f_nhwc, att_plane, init[:n_ev], idx32[:n_ev], ops.PAD_REFLECTION, "derange")
        g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g0.record()
        for _ in range(10):
            ops.gather_features(f_nhwc, att_plane, init[:n_ev], idx32[:n_ev], ops.PAD_REFLECTION, "derange")
        g1.record()
        torch.cuda.synchronize()
        gather_alone_ms = g0.elapsed_time(g1) / 10
        del f_nhwc, att_plane
        # the reference's `contour_sampling` scope (snake_head.py:1786): boxes -> V-point contours
        bdev = boxes.to(dev)
        for _ in range(2):
            ops.uniform_upsample(bdev, v, kind=ops.POLY_BOX)
        g0.record()
        for _ in range(10):
            ops.uniform_upsample(bdev, v, kind=ops.POLY_BOX)
        g1.record()
        torch.cuda.synchronize()
        sampling_ms = g0.elapsed_time(g1) / 10
        # the reference's `snake_feat_prep` scope (snake_head.py:1816-1820): bottom_out =
        # 2 x [conv3x3 256->256 + GroupNorm(32) + ReLU] on the same maps, HIP path vs stock torch.nn
        feat_prep = mask_rle = attender_fig = fcose_fig = None
        if extras and not args.no_feat_prep:
            cfg2 = config.dance_r50(num_sampling=v, num_convs=2)
            net2 = modules.RefineNet(cfg2)
            sd2 = {k: t.clone() for k, t in net2.state_dict().items()}
            synthetic.fill_state_dict_(sd2, 1234)
            net2.load_state_dict(sd2)
            net2 = net2.to(dev).eval()
            layers = net2._bottom_hip_layers(dev)

            def stock():
                modules._ConvNormAct.use_hip = False             # the plain torch.nn ops (MIOpen)
                try:
                    x = feats[:, 1:].contiguous(memory_format=torch.channels_last)
                    for conv in net2.bottom_out:
                        x = conv(x)
                    return x.permute(0, 2, 3, 1).contiguous()
                finally:
                    modules._ConvNormAct.use_hip = True

            def timed(fn, reps):
                for _ in range(2):
                    fn()
                g0.record()
                for _ in range(reps):
                    fn()
                g1.record()
                torch.cuda.synchronize()
                return g0.elapsed_time(g1) / reps

            with torch.no_grad():
                hip_ms = timed(lambda: ops.bottom_out16(feats, 1, layers), 5)
                stock_ms = timed(stock, 3)
            fp_flops = 2 * 2.0 * CH * CH * 9 * HF * WF * b
            feat_prep = {"stage": "bottom_out: 2 x [conv3x3 %d->%d + GroupNorm(32) + ReLU] on (%d,%d,%d,%d), "
                                  "NHWC fp32 out; outside the timed step" % (CH, CH, b, CH, HF, WF),
                         "kernel": "k_conv3x3_16 (split-half f16 MFMA implicit GEMM) + k_gn_relu",
                         "ms": hip_ms, "algorithmic_TFLOPs": fp_flops / hip_ms / 1e9,
                         "executed_f16_TFLOPs": 3 * fp_flops / hip_ms / 1e9,
                         "frac_of_f16_peak": 3 * fp_flops / hip_ms / 1e9 / PEAK_F16_MFMA_TFLOPS,
                         "stock_torch_ms": stock_ms, "speedup_vs_stock": stock_ms / hip_ms}
            del net2, layers
        if extras:
            # the attention-map producer in front of the path: EdgeSnakeFPNHead.attender
            # (edge_det.py:66-100, 200-204) on the batch's (b,1,200,304) edge maps, HIP vs stock torch.nn
            att_mod = modules.Attender()
            sd_a = {k: t.clone() for k, t in att_mod.state_dict().items()}
            synthetic.fill_state_dict_(sd_a, 1234)
            att_mod.load_state_dict(sd_a)
            att_mod = att_mod.to(dev).eval()
            edge_in = (1 - feats[:, :1]).contiguous()

            def stock_att():
                x = edge_in
                for i in (0, 1):
                    x = torch.relu(torch.nn.functional.group_norm(
                        torch.nn.functional.conv2d(x, att_mod[i].weight, None, padding=1), 4,
                        att_mod[i].norm.weight, att_mod[i].norm.bias, 1e-5))
                return torch.sigmoid(torch.nn.functional.conv2d(x, att_mod[2].weight, att_mod[2].bias, padding=1))

            def timed2(fn, reps):
                for _ in range(2):
                    fn()
                g0.record()
                for _ in range(reps):
                    fn()
                g1.record()
                torch.cuda.synchronize()
                return g0.elapsed_time(g1) / reps

            with torch.no_grad():
                att_ms, att_stock_ms = timed2(lambda: att_mod(edge_in), 10), timed2(stock_att, 5)
                att_err = float((att_mod(edge_in) - stock_att()).abs().max())
            attender_fig = {"stage": "attender: 3 x conv3x3 (1->32->32->1) + 2 x GroupNorm(4,32) + ReLU + sigmoid on "
                                     "(%d,1,%d,%d); outside the timed step" % (b, HF, WF),
                            "kernel": "k_att_stats1 + k_att_conv2 + k_att_out (fp32 FMA)", "ms": att_ms,
                            "algorithmic_TFLOPs": 2.0 * (288 + 9216 + 288) * HF * WF * b / att_ms / 1e9,
                            "stock_torch_ms": att_stock_ms, "speedup_vs_stock": att_stock_ms / att_ms,
                            "max_abs_diff_vs_stock": att_err}
            del att_mod

            # the step after the path: deformed contours -> RLE masks, the
            # reference's get_polygon_rles (coco_evaluation.py:311-318; one pycocotools call per
            # instance on the host there).  Contours clipped to the image as postprocessing does.
            from dance_amd import rle

            def time_rle(polys):
                for _ in range(2):
                    r = rle.polygon_rle(polys, (h, w))
                cap = r.counts.size(1)                            # settled capacity: one launch per call
                g0.record()
                for _ in range(10):
                    r = rle.polygon_rle(polys, (h, w), cap=cap)
                g1.record()
                torch.cuda.synchronize()
                ms = g0.elapsed_time(g1) / 10
                t0 = time.perf_counter()
                strings = r.strings()
                return r, strings, {"ms": ms, "polygons_per_s": n / (ms * 1e-3), "mean_runs": float(r.m.float().mean()),
                                    "strings_ms_incl_d2h": (time.perf_counter() - t0) * 1e3}

            ellipses = synthetic.make_circle_contours(boxes, v).to(dev)
            polys = mine.clone()
            polys[..., 0].clamp_(0, w)
            polys[..., 1].clamp_(0, h)
            r, strings, fig = time_rle(ellipses)
            _, _, fig_jagged = time_rle(polys)
            mask_rle = dict({"stage": "polygon -> RLE of %d %d-vertex contours at %dx%d (k_polygon_rle, one workgroup per "
                                      "polygon, parity bitmap in LDS); outside the timed step" % (n, v, h, w),
                             "contours": "ellipses inscribed in the synthetic boxes (radius 20-100 px)"}, **fig)
            mask_rle["deformed_contours"] = dict(fig_jagged, note="the bench's own output: white-noise features make "
                                                 "these contours jagged (perimeters of ~10^4 px), a stress case")
            mask_rle["note"] = "includes the wrapper's capacity check (one device sync per call)"
            polys = ellipses
            if not args.no_cpu_baseline:
                from oracle import rle_ref
                sample = polys[:200].cpu().numpy()
                t0 = time.perf_counter()
                ref_strings = [rle_ref.to_string(rle_ref.fr_poly(p_, h, w)) for p_ in sample]
                dt = time.perf_counter() - t0
                assert ref_strings == strings[:200], "bench: RLE strings differ from the oracle"
                mask_rle["cpu_oracle"] = {"polygons_per_s": 200 / dt, "cores": 1, "kind": "port",
                                          "sample": "first 200 of the same contours, oracle/rle_ref.c through ctypes"}

            # the producer of the boxes: FCOSE decode of the detector head's outputs,
            # forward_for_single_feature_map of every level and image (fcose_outputs.py:444-521), before NMS
            from dance_amd import fcose_outputs
            fc_shapes, fc_strides, fc_c, fc_th = [(100, 152), (50, 76), (25, 38), (13, 19), (7, 10)], [8, 16, 32, 64, 128], 80, 0.05
            gg = torch.Generator(device=dev).manual_seed(77)
            fc_logits = [torch.randn(b, fc_c, hh, ww, generator=gg, device=dev) * 1.5 - 6.5 for hh, ww in fc_shapes]
            fc_ctr = [torch.randn(b, 1, hh, ww, generator=gg, device=dev) for hh, ww in fc_shapes]
            fc_reg = [torch.rand(b, 4, hh, ww, generator=gg, device=dev) * 6 + 0.25 for hh, ww in fc_shapes]
            fc_ext = [torch.randn(b, 4, hh, ww, generator=gg, device=dev) * 0.3 for hh, ww in fc_shapes]
            fc_loc = fcose_outputs.compute_locations(fc_logits, fc_strides)

            def fc_hip():
                return fcose_outputs.decode_candidates(fc_loc, fc_logits, fc_reg, fc_ext, fc_ctr, fc_strides, fc_th, True)

            def fc_stock():
                # the reference's op sequence with stock torch ops (per level: permute, sigmoid, mask;
                # per image: nonzero, gathers, stack, sqrt)
                n_out = 0
                for l, (hh, ww) in enumerate(fc_shapes):
                    cls = fc_logits[l].permute(0, 2, 3, 1).reshape(b, -1, fc_c).sigmoid()
                    regl = (fc_reg[l] * fc_strides[l]).permute(0, 2, 3, 1).reshape(b, -1, 4)
                    extl = fc_ext[l].permute(0, 2, 3, 1).reshape(b, -1, 4)
                    cls = cls * fc_ctr[l].permute(0, 2, 3, 1).reshape(b, -1).sigmoid()[:, :, None]
                    cand = cls > fc_th
                    for i in range(b):
                        nz = cand[i].nonzero()
                        pl, r, e = fc_loc[l][nz[:, 0]], regl[i][nz[:, 0]], extl[i][nz[:, 0]]
                        det = torch.stack([pl[:, 0] - r[:, 0], pl[:, 1] - r[:, 1], pl[:, 0] + r[:, 2], pl[:, 1] + r[:, 3]], 1)
                        bw, bh = det[:, 2] - det[:, 0], det[:, 3] - det[:, 1]
                        torch.stack([pl[:, 0] + bw * e[:, 0], det[:, 1], det[:, 0], pl[:, 1] + bh * e[:, 1],
                                     pl[:, 0] + bw * e[:, 2], det[:, 3], det[:, 2], pl[:, 1] + bh * e[:, 3]], 1)
                        torch.sqrt(cls[i][cand[i]])
                        n_out += nz.size(0)
                return n_out

            def wall_ms(fn, reps):
                fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    r_ = fn()
                torch.cuda.synchronize()
                return (time.perf_counter() - t0) * 1e3 / reps, r_

            fc_ms, (fc_fields, _) = wall_ms(fc_hip, 10)
            fc_stock_ms, fc_n = wall_ms(fc_stock, 3)
            fc_bytes = sum(b * (fc_c + 9) * hh * ww * 4 for hh, ww in fc_shapes)
            fcose_fig = {"stage": "FCOSE decode of %d images x 5 levels (800x1216) x %d classes: sigmoid scores x centerness, "
                                  "threshold %.2f, compaction in `nonzero` order, box + extreme-point decode; outside the "
                                  "timed step" % (b, fc_c, fc_th),
                         "kernel": "k_fcose_count + torch.cumsum + k_fcose_fill (one launch each for all levels)", "candidates": int(
                             fc_fields["scores"].numel()), "ms_wall_incl_one_sync": fc_ms,
                         "algorithmic_GBps": fc_bytes / fc_ms / 1e6, "stock_torch_ms": fc_stock_ms,
                         "speedup_vs_stock": fc_stock_ms / fc_ms, "stock_candidates": fc_n,
                         "note": "host-inclusive wall time (the decode ends in one device -> host copy of the segment "
                                 "bounds; the stock sequence syncs once per image and level in nonzero())"}
            del fc_logits, fc_ctr, fc_reg, fc_ext, fc_fields

    target_fig = None
    if rank == 0 and extras:
        # training-side target matching, NumPy half: the shipped configuration's
        # chain (dense targets -> auxiliary extreme points -> contour roll + multisegment matching,
        # snake_head.py:1134-1162) for 512 ground-truth polygons, one workgroup per instance
        import numpy as np
        from dance_amd import targets
        from dance_amd._lib import check, current_stream, lib, ptr
        tn, tup = 512, 5
        rng = np.random.default_rng(9)
        tpolys = []
        for i in range(tn):
            pp = int(rng.integers(20, 300))
            ang = np.sort(rng.random(pp) * 2 * np.pi)
            rad = float(rng.integers(20, 250)) * (0.55 + 0.45 * rng.random(pp))
            tpolys.append(np.stack([400 + rad * np.cos(ang), 300 + 0.8 * rad * np.sin(ang)], 1).astype(np.float32))
        tb = torch.tensor([[float(c[:, 0].min()), float(c[:, 1].min()), float(c[:, 0].max()), float(c[:, 1].max())]
                           for c in tpolys]).to(dev)
        tcont, _ = ops.uniform_upsample(tb, v, kind=ops.POLY_BOX)
        flat, off, _ = targets._ragged(tpolys, torch.float32, dev)

        def chain():
            dense = torch.empty((tn, v * tup, 2), dtype=torch.float64, device=dev)
            st = torch.zeros(tn, dtype=torch.int32, device=dev)
            check(lib().dance_dense_targets(ptr(flat), ptr(off), ptr(tcont[:, 0].contiguous()), v * tup, ptr(dense),
                                            ptr(st), tn, current_stream()), "dense_targets")
            re, _ext = targets.get_aux_extreme_points(dense)
            return targets.single_uniform_multisegment_matching(re, tcont, re.ext_idx_padded, re.ext_count, tup)

        chain()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            _, tout, tst = chain()
        torch.cuda.synchronize()
        t_ms = (time.perf_counter() - t0) / 5 * 1e3
        target_fig = {"stage": "training-side target matching (NumPy half of compute_targets_for_polys): uniform_sample("
                               "cnt, P*%d*%d) + roll + stride -> get_aux_extreme_points -> contour roll + "
                               "single_uniform_multisegment_matching, %d polygons of 20-300 vertices; outside the timed "
                               "step" % (v, tup, tn),
                      "kernel": "k_dense_targets + k_aux_extreme_points + k_multisegment_matching (one workgroup per "
                                "instance, fp64, bit-identical to the reference's NumPy results)",
                      "ms_wall_incl_two_syncs": t_ms, "instances_per_s": tn / (t_ms * 1e-3),
                      "fallbacks": int((tst == 1).sum())}
        if not args.no_cpu_baseline:
            from oracle import targets_ref
            k = 2
            t0 = time.perf_counter()
            for i in range(k):
                c = tpolys[i]
                big = targets_ref.uniform_sample(c, len(c) * v * tup)
                f0 = tcont[i, 0].cpu().numpy()
                tt = int(np.argmin(np.power(big - f0, 2).sum(axis=1)))
                d_ = np.roll(big, -tt, axis=0)[::len(c)]
                re_, ext_ = targets_ref.get_aux_extreme_points(d_)
                r_ = targets_ref.roll_to_first_target(tcont[i].cpu().numpy(), re_)
                m_, _s = targets_ref.single_uniform_multisegment_matching(re_, r_, ext_, tup, v)
                assert np.array_equal(m_, tout[i].cpu().numpy()), "bench: target matching differs from the oracle"
            target_fig["cpu_oracle"] = {"instances_per_s": k / (time.perf_counter() - t0), "cores": 1, "kind": "port",
                                        "sample": "first %d of the same polygons, oracle/targets_ref.py (explicit Python "
                                                  "loops: a checker, far slower than the reference's vectorised NumPy)" % k}

    if rank == 0:
        # per-stage device time from the HIP events recorded inside the timed steps
        stage_ms = [0.0] * len(STAGE_NAMES)
        for k in range(args.steps):
            for it in range(ITERS):
                ev = events_full[k][it]
                for s in range(len(STAGE_NAMES)):
                    stage_ms[s] += ev[s].elapsed_time(ev[s + 1])
        launches = args.steps * ITERS
        stage_avg = {name: stage_ms[i] / launches for i, name in enumerate(STAGE_NAMES)}
        # from the timed steps: head kernel and the 7 residual blocks
        head_ms = sum(events[k][it][0].elapsed_time(events[k][it][1])
                      for k in range(args.steps) for it in range(ITERS)) / launches
        res7_ms = sum(events[k][it][1].elapsed_time(events[k][it][8])
                      for k in range(args.steps) for it in range(ITERS)) / launches
        stacked = args.precision != "fp32"                      # fast mode: the 7 blocks are one launch
        res_ms = res7_ms / 7.0                                  # per residual block
        res_tflops = 2.0 * MACS_RES_BLOCK * v * n_ev / (res_ms * 1e-3) / 1e12  # algorithmic
        gather_gbps = GATHER_BYTES_PER_VERTEX * v * n_ev / (head_ms * 1e-3) / 1e9
        layout_ms = (sum(a.elapsed_time(b_) for a, b_ in layout_ev) / args.steps) if pre is None else 0.0
        ms_per_step = elapsed / args.steps * 1e3
        total_flops = 2.0 * MACS_PER_VERTEX * v * n_total * ITERS     # whole job
        # dance_res_stack16's dispatch (csrc/snake16.hip): LDS image width in 32-column units, the
        # 8-wave kernel above 128 vertices and for few contours (policy.few_contour_forms), an odd
        # number of 16-wide vertex tiles as k + (k - 1)
        rt = 2 if v <= 64 else 4 if v <= 128 else 7 if v <= 224 else 8
        rt = 3 if 64 < v <= 96 else 5 if 128 < v <= 160 else 6 if 160 < v <= 192 else rt
        stack_wide = v > 128 or (v > 64 and n_ev <= 256)
        stack13 = stack_wide and v > 128 and (-(-v // 16)) % 2 == 1
        if args.precision in ("fp16x3", "fp16"):
            terms = 3 if args.precision == "fp16x3" else 1
            # every algorithmic product is three f16 MFMA products (A_lo*B_hi + A_hi*B_lo +
            # A_hi*B_hi).  `achieved` / `frac` price the ALGORITHMIC flops (SURVEY 8d's figure) against
            # the f16 dense peak; `frac_executed` counts the three MFMAs each product costs (the pipe
            # utilisation); `frac_vs_f32_mfma_peak` is the same useful rate against the fp32-MFMA peak
            # that the reference's own fp32 arithmetic would be limited to
            kname = (("k_res_stack16w13" if stack13 else "k_res_stack16w" if stack_wide else "k_res_stack16s") +
                     "<%d,%d> (the 7 residual SnakeBlocks 128->128, k=9, in one launch; %s f16 MFMA 16x16x32)") % (
                         4 if stack_wide and v <= 128 else rt, terms, "split-half" if terms == 3 else "single-term")
            roof = {"kernel": kname,
                    "bound": "mfma", "achieved": res_tflops, "peak": PEAK_F16_MFMA_TFLOPS,
                    "unit": "TFLOP/s", "frac": res_tflops / PEAK_F16_MFMA_TFLOPS, "traffic": None,
                    "frac_algorithmic": res_tflops / PEAK_F16_MFMA_TFLOPS,
                    "achieved_executed": terms * res_tflops,
                    "frac_executed": terms * res_tflops / PEAK_F16_MFMA_TFLOPS,
                    "frac_vs_f32_mfma_peak": res_tflops / PEAK_F32_MFMA_TFLOPS,
                    "avg_launch_ms": res7_ms,
                    "algorithmic_GFLOP_per_launch": 2.0 * MACS_RES_BLOCK * v * n_ev * 7 / 1e9,
                    "algorithmic_TFLOPs": res_tflops, "mfma_per_product": terms,
                    "contours_per_launch": n_ev,
                    "note": "achieved = algorithmic flops per launch / spacing of the two HIP events around the "
                            "residual blocks (event spacing includes the inter-kernel gap, ~10 us per launch), "
                            + ("recorded inside the timed steps" if single_chunk and args.streams == 1 else
                               "recorded in an instrumented pass over the first %d contours after the timed "
                               "region" % n_ev) + "; executed = %d f16 MFMAs per product" % terms}
            if side_n:
                # the timed launch shares the CUs with the side stream's small kernels (the peeled
                # remainder's chain); the same launch with the GPU to itself, from the instrumented
                # single-stream pass over the same n_ev contours
                alone_ms = sum(stage_avg["res%d" % i] for i in range(7))
                alone_tflops = 2.0 * MACS_RES_BLOCK * v * n_ev * 7 / (alone_ms * 1e-3) / 1e12
                roof["alone"] = {"avg_launch_ms": alone_ms, "achieved": alone_tflops,
                                 "frac": alone_tflops / PEAK_F16_MFMA_TFLOPS,
                                 "frac_executed": terms * alone_tflops / PEAK_F16_MFMA_TFLOPS,
                                 "note": "the same launch without the side stream's kernels beside it "
                                         "(instrumented pass after the timed region)"}
            if terms == 3:
                roof["traffic"], roof["traffic_source"] = measured_traffic(
                    "k_res_stack16w" if stack_wide else "k_res_stack16s", n_ev)
            dtype = "f16x3/f32acc" if terms == 3 else "f16/f32acc (approximate)"
            halves = v > 128 or (v > 64 and n_ev <= 128)           # dance_gather_head16w's dispatch (csrc/head16.hip)
            gkernel = ("k_gather_head16h (fused gather + head SnakeBlock, two half-ring workgroups per contour)" if halves
                       else "k_gather_head16w<%d,%d> (fused gather + head SnakeBlock, 16-wide vertex tiles)" % (
                           2 if v <= 64 else 4, -(-v // 16)))
            # the head kernel's own MFMA work: 256 feature channels x 9 ring taps + the de_location channels as
            # one 32-deep K step, every product as `terms` f16 MFMAs
            head_alg = 2.0 * (CH + 2) * 128 * 9 * v * n_ev / (head_ms * 1e-3) / 1e12
            head_exec = terms * 2.0 * (CH * 9 + 32) * 128 * v * n_ev / (head_ms * 1e-3) / 1e12
        else:
            roof = {"kernel": "k_circconv<%d> (residual SnakeBlock 128->128, k=9, f32 MFMA)" % rt,
                    "bound": "mfma", "achieved": res_tflops, "peak": PEAK_F32_MFMA_TFLOPS,
                    "unit": "TFLOP/s", "frac": res_tflops / PEAK_F32_MFMA_TFLOPS, "traffic": None,
                    "frac_algorithmic": res_tflops / PEAK_F32_MFMA_TFLOPS,
                    "frac_executed": res_tflops / PEAK_F32_MFMA_TFLOPS,
                    "avg_launch_ms": res_ms, "algorithmic_TFLOPs": res_tflops, "mfma_per_product": 1,
                    "contours_per_launch": n_ev}
            dtype = "f32"
            gkernel = "k_gather_head<%d> (fused gather + head SnakeBlock)" % rt
            head_alg = head_exec = 2.0 * (CH + 2) * 128 * 9 * v * n_ev / (head_ms * 1e-3) / 1e12
        line = {
            "metric": "contours deformed/sec (%d vertices, all iters)" % v,
            "value": n_total * args.steps / elapsed,
            "unit": "contours/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": WORKLOAD_NAMES[args.workload] + (
                           ": ONE sweep of %d images x %d instances x %d vertices split by image over the ranks "
                           "(this rank: images [%d, %d)), %d iterations, counts + padded RCCL all-gather of the "
                           "deformed contours; features (%d,%d,%d,%d) %s fp32 in HBM per rank" % (
                               args.images, per, v, img_lo, img_hi, ITERS, b, CH + 1, HF, WF, args.layout.upper())
                           if strong else
                           ": DANCE R50 snake head, %d images x %d instances x %d vertices per GPU, %d iterations, "
                           "features (%d,%d,%d,%d) %s fp32 in HBM" % (b, per, v, ITERS, b, CH + 1, HF, WF,
                                                                     args.layout.upper())),
                       "contours_total": n_total, "contours_per_gpu": n, "vertices": v, "iterations": ITERS,
                       "chunks_per_iteration": -(-n // deformer.MAX_CHUNK),
                       "precision": args.precision + {
                           "fp16x3": " (fp32-level accuracy: parity 1e-4 norm-wise per iteration vs the fp32 reference)",
                           "fp32": " (exact f32 MFMA)",
                           "fp16": " (APPROXIMATE: ~1e-2 coordinate error, outside the 1e-4 parity bar)"}[args.precision],
                       "fused_head": not args.no_fused_head, "parallelism": "dp%d" % world},
            "roofline": roof,
            "gather": {"kernel": gkernel,
                       "algorithmic_GBps": gather_gbps, "frac_of_hbm_peak": gather_gbps / PEAK_HBM_GBPS,
                       "avg_launch_ms": head_ms,
                       "algorithmic_mfma_TFLOPs": head_alg, "executed_mfma_TFLOPs": head_exec,
                       "frac_algorithmic_mfma": head_alg / (PEAK_F32_MFMA_TFLOPS if args.precision == "fp32"
                                                            else PEAK_F16_MFMA_TFLOPS),
                       "frac_executed_mfma": head_exec / (PEAK_F32_MFMA_TFLOPS if args.precision == "fp32"
                                                          else PEAK_F16_MFMA_TFLOPS),
                       "note": "the fused kernel is MFMA-bound (%d MFLOP per 526 KB gathered per "
                               "contour-iteration), not HBM-bound: frac_executed_mfma prices it against the roofline "
                               "that binds it" % (2.0 * (CH + 2) * 128 * 9 * v / 1e6),
                       "traffic": measured_traffic("k_gather_head16" if args.precision != "fp32" else "k_gather_head<", n_ev)[0],
                       "standalone_traffic": measured_traffic("k_gather|", 4 * n_ev)[0] if v == 128 else None,
                       "traffic_note": "HBM bytes per launch from the committed PMC passes (profiles/hbm_traffic.json; "
                                       "neighbouring vertices share taps in L2, so fewer bytes are fetched than the "
                                       "no-reuse algorithmic figure)",
                       "standalone_kernel": "k_gather (gather + de_location + concat only, untimed extra)",
                       "standalone_ms": gather_alone_ms,
                       "standalone_algorithmic_GBps": GATHER_BYTES_PER_VERTEX * v * n_ev / (gather_alone_ms * 1e-3) / 1e9,
                       "standalone_frac_of_hbm_peak": GATHER_BYTES_PER_VERTEX * v * n_ev / (gather_alone_ms * 1e-3) / 1e9
                                                      / PEAK_HBM_GBPS},
            "whole_path_algorithmic_TFLOPs": total_flops / (ms_per_step * 1e-3) / 1e12,
            "stage_ms": dict(stage_avg, layout_nchw_to_nhwc=layout_ms,
                             note="HIP events on the launch stream (14 boundaries per iteration).  Fast mode: "
                                  "`fusion_max` is k_fuse16w (fusion + ring max + prediction[0]'s state part, 16-wide tiles), "
                                  "`offset_update` is k_tail16 (row constant, ReLU, prediction[2], prediction[4], "
                                  "offset update); row_constant / pred0 / pred1 are inside those two launches"),
            "streams": args.streams,
            "peel": {"main_stream_contours": n - side_n, "side_stream_contours": side_n,
                     "note": "deform() runs n mod CUs contours as their own kernel chain on a side HIP stream, so the "
                             "main stream's launches are whole rounds of workgroups (bit-identical results; "
                             "DANCE_PEEL=0 keeps one stream).  All n contours are inside the timed region; the "
                             "stage events and the roofline time the main stream's launches"},
            "two_streams": two_streams,
            "feat_prep": feat_prep,
            "mask_rle": mask_rle,
            "attender": attender_fig,
            "fcose_decode": fcose_fig,
            "target_matching": target_fig,
            "contour_sampling": {"kernel": "k_uniform_upsample (boxes -> %d-point contours, fused resampler)" % v,
                                 "ms": sampling_ms, "contours_per_s": n / (sampling_ms * 1e-3),
                                 "note": "reference scope `contour_sampling`, outside the timed step"},
        }
        line["rccl"] = {"initialised": bool(dist_on and dist.is_initialized()),
                        "backend": dist.get_backend() if dist_on else None, "world_size": world,
                        "all_gather_into_tensor_calls_in_timed_steps": timed_collectives,
                        "note": "exchange step: %s; none without a process group" % (
                            "counts all-gather + padded contour all-gather per step" if strong else
                            "one contour all-gather per step")}
        if not args.no_cpu_baseline:
            # rank 0 only, after the timed region, at every world size (the other ranks wait in the barrier below)
            line["cpu_baseline"] = cpu_baseline(args.cpu_reps)
        print(json.dumps(line))
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
