#!/usr/bin/env python
"""Headline benchmark: frames/sec of the DenseNet-121 224x224 frame feature-extract
(BASELINE.json configs[1]: batch 256 synthetic frames, fp16, one MI355X per rank).

  python bench.py --gpus N --steps K --warmup W

N > 1 runs one process per GPU over RCCL: under ``python -m torch.distributed.run --nproc-per-node N ...`` the ranks
exist already (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* are read from the environment); started as plain
``python bench.py --gpus N`` the script spawns the N ranks itself (tennis_amd.sharding.launch).  Rank 0 prints ONE
JSON line.

A step = one pass of the hot path over one batch of 256 synthetic frames per rank (inputs resident in HBM):
stem -> 58 dense layers -> 3 transitions -> head -> (B,1024) fp32 features, then for N > 1 the RCCL all-gather of
the feature rows that the temporal / caption stage consumes.  Frames shard across ranks with no other
exchange: weak scaling.  The timed region is EXACTLY K steps between two fences (barrier + device synchronise); when
K is small the region is repeated (each repetition again exactly K fenced steps, at least 200 steps in total) and
the MEDIAN repetition is reported, so that a 20-step run is not a single 48-ms sample.
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

BATCH = 256
SIZE = 224
FLOP_PER_FRAME = 5.666e9          # 2 x 2.8331 GMAC over the 120 convolutions
MFMA_PEAK_TFLOPS = 2500.0         # MI355X dense fp16 MFMA (MI355X_MICROARCH.md)
HBM_PEAK_TBS = 8.0                # MI355X HBM3E (MI355X_MICROARCH.md)
MIN_TOTAL_STEPS = 3000        # (~5.5 s of GPU time at the default batch: a median over 15 regions of 200 steps, and long enough for a 1 Hz power / utilisation sampler outside the process to see several busy samples: VERDICT r4 weak 10)


def parity_note():
    """What tests/test_gpu_calibration.py measured on MI355X for the model this line times (the newest committed matrix
    profiles/r*_calibration_matrix.json: max-abs error of features and of the Dense(11) logits vs the fp32 oracle on the UN-rounded
    weights and un-rounded input, per frame family).  One mode: the timed configuration is the one that is graded for parity."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_calibration_matrix.json")))
    if not files:
        return {"source": None, "note": "see tests/test_gpu_calibration.py (no committed matrix in this tree)"}
    try:
        m = json.load(open(files[-1]))
        feats, logits = m["feature_max_abs_error"], m.get("logit_max_abs_error", {})
        key = [k for k in feats if k.startswith("built-in set, ")][-1]        # the default conversion (the largest built-in set)
        d, ker, pl = feats[key], feats["kernels alone (oracle on the converted weights)"], feats["plain rounding"]
        out = {"source": os.path.relpath(files[-1], ROOT), "conversion": key, "bar": 1e-3, "families": len(d),
               "families_within_bar": sum(v < 1e-3 for v in d.values()),
               "feature_err_worst": max(d.values()), "feature_err_worst_family": max(d, key=d.get),
               "feature_err_by_family": {f: round(v, 6) for f, v in d.items()},
               "kernels_alone_worst": max(ker.values()), "plain_rounding_worst": max(pl.values())}
        if key in logits:
            out["logit_err_worst"] = max(logits[key].values())
        wide = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_parity_wide.json")))
        if wide:       # tests/tools/parity_wide.py: more frames per family, two weight seeds, the exact-weights mode beside it
            w = json.load(open(wide[-1]))
            out["wide"] = {"source": os.path.relpath(wide[-1], ROOT), "frames_per_family": w["frames_per_family"],
                           "by_weight_seed": {k: {"worst_feature": v["worst_feature"], "worst_logit": v["worst_logit"],
                                                  "values_over_bar": v["values_over_bar"], "values": sum(r["values"] for r in v["families"].values()),
                                                  "families_over_bar": sorted(f for f, r in v["families"].items() if r["over_bar"]),
                                                  "exact_mode_worst_feature": v.get("exact_mode_worst_feature"),
                                                  "exact_mode_values_over_bar": v.get("exact_mode_values_over_bar")} for k, v in w["weights"].items()}}
        return out
    except Exception as e:
        return {"source": os.path.relpath(files[-1], ROOT), "note": f"{type(e).__name__}: matrix not readable"}

# compulsory HBM bytes per frame if every intermediate stayed on chip: uint8 NHWC frame in, fp32
# features out (+ 13.7 MB of fp16 weights per batch)
COMPULSORY_BYTES_PER_FRAME = 150528 + 4096
WEIGHT_BYTES = 13.7e6
# profile family (tn_densenet121_profile) -> kernel family key of profiles/*_pmc_traffic.json
PMC_KEYS = {"dense_layer_strip_56x56": ("dense_strip_56x56", "dense_strip_kernel<56"),
            "dense_layer_strip_28x28": ("dense_strip_28x28", "dense_strip_kernel<28"),
            "dense_layer_fused_56x56": ("dense_layer_56x56", "dense_layer_kernel<56"),
            "dense_layer_fused_28x28": ("dense_layer_28x28", "dense_layer_kernel<28"),
            "dense_block_chained_14x14": ("dense_block_14x14", "dense_layer_kernel<14"),
            "dense_block_stream_14x14": ("dense_block14_kernel",),
            "dense_block_chained_7x7": ("dense_block_7x7", "dense_layer_kernel<7"),
            "dense_block_lds_7x7": ("dense_block7_kernel",),
            "stem_conv_bn_relu_maxpool": ("stem_pool_kernel",), "transition_conv1x1_avgpool": ("conv1x1_kernel",),
            "head_bnrelu_avgpool7": ("head_kernel",)}


def make_frames(batch, size, seed, device):
    """uint8 uniform frames, NHWC, generated on the device: decoded frames as the loader (JPEG decode + Resize + CenterCrop on the
    device) hands them to the encoder, which applies ToTensor + Normalize itself (reference evaluate.py:93-98)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    return torch.randint(0, 256, (batch, size, size, 3), generator=g, device=device, dtype=torch.uint8).contiguous()


def normalize_nchw_f32(u8):
    """ToTensor + Normalize of the reference's test transform (evaluate.py:96-97) for the CPU baseline's input"""
    mean = torch.tensor([0.485, 0.456, 0.406])
    std = torch.tensor([0.229, 0.224, 0.225])
    return ((u8.cpu().float() / 255.0 - mean) / std).permute(0, 3, 1, 2).contiguous()


def pmc_traffic(kernel_family):
    """(HBM bytes per launch, file) of a kernel family from the newest committed rocprofv3 PMC summary
    (profiles/*_pmc_traffic.json: FETCH_SIZE / WRITE_SIZE passes with the gfx950 x2 FETCH correction).  PMC counters
    cannot be read from inside the timed process: this is a figure from a file, labelled as such in the line."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    if not files:
        return None, None
    fam = json.load(open(files[-1])).get("families", {})
    for key in PMC_KEYS.get(kernel_family, ()) + (kernel_family,):
        if key in fam:
            return fam[key].get("hbm_bytes_per_dispatch_corrected"), os.path.relpath(files[-1], ROOT)
    return None, os.path.relpath(files[-1], ROOT)


def _median_rate(fn, n, runs):
    """frames/s of fn() over n frames: warm-up + median of `runs` timed calls."""
    fn()
    ts = []
    for _ in range(runs):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return n / float(np.median(ts)), ts


def cpu_baseline(params, frames_u8, full=False):
    """The reference CPU path stand-ins of SURVEY §8(d) on this box's host cores, on bounded samples (MXNet's CPU
    path cannot be installed: both are restatements, kind "port"):
      (i)  oracle/torch_ref.py — the same graph through torch-CPU / oneDNN, fp32, all cores: config C1's 32 frames,
           median of 5 runs (`value`); the 256-frame batch once more with --cpu-baseline-full;
      (ii) oracle/densenet_np.py — the numpy fp32 oracle, 2 frames, median of 3 runs (`numpy_oracle`)."""
    from oracle import densenet_np as dn
    from oracle.torch_ref import TorchDenseNet121
    threads = torch.get_num_threads()
    net = TorchDenseNet121(params)
    x32 = normalize_nchw_f32(frames_u8[:32])
    net(x32[:2])
    fps32, ts32 = _median_rate(lambda: net(x32), 32, 5)
    out = {"value": round(fps32, 2), "unit": "frames/sec", "cores": threads, "kind": "port",
           "sample": f"config C1: 32 frames 224x224 fp32 through oracle/torch_ref.py (torch-CPU/oneDNN restatement; MXNet CPU "
                     f"path not installable), median of 5 runs, {threads} threads of {os.cpu_count()} logical cpus",
           "runs_s": [round(t, 3) for t in ts32]}
    x2 = x32[:2].numpy()
    fps_np, ts_np = _median_rate(lambda: dn.densenet121_features(x2, params), 2, 3)
    out["numpy_oracle"] = {"value": round(fps_np, 3), "unit": "frames/sec", "cores": "numpy/BLAS default threads",
                           "sample": "2 frames 224x224 fp32 through oracle/densenet_np.py, median of 3 runs",
                           "runs_s": [round(t, 3) for t in ts_np]}
    if full:
        n = min(256, frames_u8.shape[0])
        x256 = normalize_nchw_f32(frames_u8[:n])
        fps256, ts256 = _median_rate(lambda: net(x256), n, 5)
        out["batch256"] = {"value": round(fps256, 2), "unit": "frames/sec", "cores": threads,
                           "sample": f"{n}-frame batch through oracle/torch_ref.py, median of 5 runs",
                           "runs_s": [round(t, 3) for t in ts256]}
    return out


PARITY_MIXED = 34          # frames [0, 34) of the timed batch: two of each of the 16 frame families + two fine checkerboards
PARITY_FRAMES = 48         # ... checked together with the first 14 noise frames behind them


def mixed_head_of_batch(size):
    """(uint8 NHWC frames, labels): what replaces the first PARITY_MIXED noise frames of the timed batch"""
    from tennis_amd import calib_frames as CF
    npz = os.path.join(ROOT, "tests", "golden", "jpeg_cases.npz")
    return CF.mixed_batch(PARITY_MIXED, size, seed=2025, jpeg_npz=npz if os.path.exists(npz) else None, fine=2)


def parity_live(enc, x, labels, params32):
    """Parity of the TIMED encoder instance on the batch it was timed on: one more 256-frame forward of `enc`
    through the same default kernels, its first PARITY_FRAMES feature rows (mixed families + noise) against oracle/torch_ref.py on
    the UN-rounded fp32 parameters and the un-rounded normalised input; Dense(11) logits of the frame classifier beside them."""
    from oracle.torch_ref import TorchDenseNet121
    from tennis_amd import weights as W
    n = min(PARITY_FRAMES, x.shape[0])
    t0 = time.perf_counter()
    got = enc(x)[:n].cpu().numpy().astype(np.float64)
    ref = TorchDenseNet121(params32)(normalize_nchw_f32(x[:n])).numpy()
    wd = W.make_dense_weights(1, 11, 1024, "framemodel0_dense0_")["framemodel0_dense0_weight"].astype(np.float64)
    e = got - ref
    el = e @ wd.T
    lab = list(labels[:n]) + ["noise (device-generated)"] * (n - len(labels[:n]))
    fams = {}
    for f in dict.fromkeys(lab):
        idx = [i for i, l in enumerate(lab) if l == f]
        fams[f] = round(float(np.abs(e[idx]).max()), 6)
    fine = np.array([l == "finechecker" for l in lab])
    ab = np.abs(e)
    split = {"sixteen_families_and_noise": {"frames": int((~fine).sum()), "feature_err_max": float(ab[~fine].max()), "logit_err_max": float(np.abs(el[~fine]).max()),
                                            "values_over_bar": int((ab[~fine] > 1e-3).sum()), "values": int(ab[~fine].size)},
             "fine_checkerboards": ({"frames": int(fine.sum()), "feature_err_max": float(ab[fine].max()), "logit_err_max": float(np.abs(el[fine]).max()),
                                     "values_over_bar": int((ab[fine] > 1e-3).sum()), "values": int(ab[fine].size),
                                     "note": "cells of 1 - 3 px at full contrast: beyond 1e-3 in every mode (exact weights 1.5e-3; tests/test_gpu_parity_timed.py)"} if fine.any() else None)}
    return {"frames": n, "values": int(e.size), "bar": 1e-3, "oracle": "oracle/torch_ref.py: fp32 graph, un-rounded fp32 weights and input", **split,
            "feature_err_max": float(np.abs(e).max()), "feature_err_rms": float(np.sqrt((e ** 2).mean())), "logit_err_max": float(np.abs(el).max()),
            "values_over_bar": int((np.abs(e) > 1e-3).sum()), "feature_err_by_family": fams,
            "worst_family": max(fams, key=fams.get), "seconds": round(time.perf_counter() - t0, 2),
            "how": f"a {x.shape[0]}-frame forward of the timed encoder instance after the timed regions, rows [0, {n}) checked"}


class StepLoop:
    """The step / join / all-gather ordering of the timed loop, apart from the GPU so that tests/test_cpu_distributed.py can
    drive exactly this code on gloo with a stand-in encoder (the first multi-GPU run must not be the first time it executes).

    ``enc(x, out=f)`` encodes one batch into ``f``; with pipelined forwards ``f`` is only valid after ``enc.join(lag)``
    (``lag`` 0: every call so far, 1: every call but the last).  Feature buffers and gather buffers alternate by step parity;
    a buffer pair is reused two steps later, behind the wait for the collective that last read / wrote it."""

    def __init__(self, enc, x, feats, gathered, comm, world, pipelined):
        self.enc, self.x, self.feats, self.gathered, self.comm = enc, x, feats, gathered, comm
        self.world, self.pipelined = world, pipelined
        self.gather_h = [None, None]

    def allgather(self, j):
        if self.gather_h[j] is not None:
            self.gather_h[j].wait()          # the previous collective into this buffer pair
        self.gather_h[j] = self.comm.allgather_features(self.feats[j], self.gathered[j])

    def step(self, i):
        self.enc(self.x, out=self.feats[i & 1])
        if self.world > 1:
            if not self.pipelined:
                self.allgather(i & 1)
            elif i > 0:
                self.enc.join(1)
                self.allgather((i - 1) & 1)

    def drain(self, k):
        if self.pipelined and k > 0:
            self.enc.join(0)
            if self.world > 1:
                self.allgather((k - 1) & 1)
        for j in (0, 1):
            if self.gather_h[j] is not None:
                self.gather_h[j].wait()
                self.gather_h[j] = None


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-full", action="store_true", help="also time the 256-frame batch on the CPU (minutes)")
    ap.add_argument("--single-region", action="store_true", help="one timed region of K steps only (no repetitions)")
    ap.add_argument("--no-pipeline", action="store_true", help="join the encoder's two half-batch streams inside every forward (A/B against the pipelined default)")
    ap.add_argument("--plain-rounding", action="store_true", help="seeded weights that are fp16-representable (rounds 1-2's model) instead of "
                    "the calibrated conversion of fp32 weights")
    ap.add_argument("--no-parity-live", action="store_true", help="skip config.parity_live (the timed encoder's features against the fp32 oracle on 48 frames of the timed batch)")
    ap.add_argument("--no-exact-line", action="store_true", help="skip the extra fenced region that times the exact-weights mode (config.exact_weights_frames_per_sec)")
    ap.add_argument("--exact-weights", action="store_true",
                    help="NOT the headline configuration: un-rounded fp32 conv weights evaluated as hi + lo fp16 pairs "
                         "(TN_ENC_EXACT_WEIGHTS), to state what the 1e-3-vs-fp32-weights mode costs")
    return ap


def run(argv):
    args = build_parser().parse_args(argv)
    import torch.distributed as dist
    from tennis_amd import sharding
    rank, world, dev = sharding.init_distributed()
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} ranks")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: tennis_amd has no CPU path")

    from tennis_amd import _lib
    from tennis_amd import weights as W
    from tennis_amd.engine import DenseNet121Features

    ctx = _lib.Context(dev.index)
    # The measured model: seeded fp32 conv weights that are NOT fp16-representable (what a trained checkpoint looks like),
    # converted to ONE fp16 number per weight by calibrated rounding against the library's 72 built-in calibration frames
    # (tennis_amd/calibrate.py: vector error feedback, round 4).  Its features are within 1e-3 of the fp32 oracle evaluated on the
    # un-rounded weights on natural-looking content and on most synthetic families; the measured per-family figures and the
    # worst case are in tests/test_gpu_calibration.py / gpurun_out/parity_report.json and quoted in config.weights below.
    # --plain-rounding: seeded weights that are fp16-representable to begin with (same kernels, same rate).
    params32 = W.make_densenet121_weights(0, fp16_model=False)
    if args.exact_weights:
        params = params32
    elif args.plain_rounding:
        params = W.make_densenet121_weights(0)
    else:
        from tennis_amd.calibrate import calibrated_fp16_model
        params = calibrated_fp16_model(params32, None, SIZE, ctx=ctx)     # the built-in calibration frames: the same model on every rank
    enc = DenseNet121Features(params, SIZE, max_batch=args.batch, ctx=ctx, exact_weights=args.exact_weights)
    x = make_frames(args.batch, SIZE, 1234 + rank, dev)
    mixed_labels = []
    if args.batch >= PARITY_FRAMES:      # the head of the timed batch is a mix of frame families, so that parity is measured on what is timed
        mixed, mixed_labels = mixed_head_of_batch(SIZE)
        x[:len(mixed)] = torch.from_numpy(mixed).to(dev)
    feats = [torch.empty((args.batch, enc.feature_dim), dtype=torch.float32, device=dev) for _ in range(2)]
    gathered = [torch.empty((world * args.batch, enc.feature_dim), dtype=torch.float32, device=dev)
                for _ in range(2)] if world > 1 else None
    # the exchange step goes through the library's own RCCL communicator (tn_allgather_features); torch.distributed only
    # carries the barrier and the max over ranks of the timing
    comm = sharding.feature_comm(dev) if world > 1 else None      # (comm.bring_up: agreed fall-back to torch's RCCL communicator)
    # what every rank's communicator says about itself (tn_comm_world = ncclCommCount, tn_comm_device = ncclCommCuDevice): the
    # record of an N-GPU run shows that RCCL saw N ranks on N devices, not only that the launcher asked for them
    comm_record = None
    if world > 1:
        mine = dict(comm.describe(), launcher_rank=rank, local_rank=int(os.environ.get("LOCAL_RANK", dev.index)), cuda_device=dev.index,
                    gpu=torch.cuda.get_device_name(dev), pid=os.getpid())
        allr = [None] * world
        dist.all_gather_object(allr, mine)
        comm_record = {"world_launched": world, "world_reported_by_rccl": sorted({r["world"] for r in allr}),
                       "ranks_reported": sorted(r["rank"] for r in allr), "devices": [r["device"] for r in sorted(allr, key=lambda r: r["launcher_rank"])],
                       "all_through_rccl": all(r["rccl"] for r in allr), "per_rank": sorted(allr, key=lambda r: r["launcher_rank"])}
    # Pipelined forwards (tn_densenet121_set_pipelined): the encoder runs a batch as two half batches on two streams, and
    # the last chained block of the second half occupies half of the CUs; without a join inside forward the next step's
    # first half starts beside it.  Results are ordered by the caller (StepLoop): the all-gather of step i is issued one step
    # behind (join lag 1), and `drain` joins the last step - every one of the K steps is complete before the closing fence.
    pipelined = not args.no_pipeline
    enc.set_pipelined(pipelined)
    loop = StepLoop(enc, x, feats, gathered, comm, world, pipelined)
    step, drain = loop.step, loop.drain

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_region(k):
        """exactly k steps between two fences; max over ranks"""
        fence()
        t0 = time.perf_counter()
        for i in range(k):
            step(i)
        drain(k)
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt

    for i in range(args.warmup):
        step(i)
    drain(args.warmup)
    repeats = 1 if args.single_region else max(1, -(-MIN_TOTAL_STEPS // args.steps))
    times = [timed_region(args.steps) for _ in range(repeats)]
    dt = float(np.median(times))
    # for the record: the same K steps with every forward joined before it returns (one more fenced region, all ranks)
    dt_joined = None
    if pipelined and not args.single_region:
        loop.pipelined = False
        enc.set_pipelined(False)
        dt_joined = timed_region(args.steps)
        enc.set_pipelined(True)
        loop.pipelined = True

    # for the record: the rate of the configuration that meets "1e-3 of the reference" against UN-rounded fp32
    # parameters - the exact-weights mode (hi + lo fp16 weight pairs: twice the MFMA work of the dense layers and transitions)
    # - on the same box: one more fenced region of exactly K steps with a second encoder
    fps_exact = None
    if not args.exact_weights and not args.single_region and not args.no_exact_line:
        params_x = params32
        enc_f16 = enc
        enc = DenseNet121Features(params_x, SIZE, max_batch=args.batch, ctx=ctx, exact_weights=True)
        enc.set_pipelined(pipelined)
        loop.enc = enc
        for i in range(min(args.warmup, 5)):
            step(i)
        drain(min(args.warmup, 5))
        fps_exact = world * args.batch * args.steps / timed_region(args.steps)
        enc.set_pipelined(False)
        enc = enc_f16
        loop.enc = enc
        del params_x

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        fps = world * args.batch * args.steps / dt
        # ---- roofline of the dominant kernel family: HIP events around every launch on the stream the kernels run
        # on (separate instrumented passes, so the timed region above carries no events)
        fams = {}
        NPROF = 3
        for _ in range(2):
            enc.profile(x)        # (settle: the instrumented pass runs the whole batch on one stream, the timed region did not)
        for _ in range(NPROF):
            stats, _ = enc.profile(x)
            if os.environ.get("TN_BENCH_DEBUG"):
                print("profile pass:", {s["name"][:24]: round(s["ms"] / max(1, s["launches"]) * 1e3, 1) for s in stats}, file=sys.stderr)
            for s in stats:
                a = fams.setdefault(s["name"], dict(ms=0.0, flops=0.0, bytes=0.0, launches=0))
                a["ms"] += s["ms"]; a["flops"] += s["flops"]; a["bytes"] += s["bytes"]; a["launches"] += s["launches"]
        dom = max(fams, key=lambda k: fams[k]["ms"])
        d = fams[dom]
        secs = d["ms"] * 1e-3
        tf, tbs = d["flops"] / secs / 1e12, d["bytes"] / secs / 1e12
        traffic, traffic_file = pmc_traffic(dom)
        enc_tf = FLOP_PER_FRAME * fps / world / 1e12
        # the encoder is a dense contraction -> the MFMA roofline is the graded bound (target 40 % of the
        # 2.5 PFLOP/s dense fp16 peak); the HBM figure (layer-wise algorithmic bytes) is reported next to it
        roofline = {"bound": "mfma", "kernel": dom, "achieved": round(tf, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(tf / MFMA_PEAK_TFLOPS, 4),
                    "traffic": traffic,
                    "traffic_source": (f"from {traffic_file} (rocprofv3 PMC passes of an earlier run of this command, "
                                       "not measured in this process)") if traffic_file else None,
                    "algorithmic_flops_per_launch": round(d["flops"] / d["launches"]),
                    "algorithmic_bytes_per_launch": round(d["bytes"] / d["launches"]),
                    "avg_launch_us": round(d["ms"] / d["launches"] * 1e3, 2), "launches_per_step": d["launches"] // NPROF,
                    "other_roofline": {"bound": "hbm", "achieved": round(tbs * 1e3, 2), "peak": HBM_PEAK_TBS * 1e3,
                                       "unit": "GB/s", "frac": round(tbs / HBM_PEAK_TBS, 4),
                                       "bytes": "layer-wise algorithmic (every layer reads its inputs from HBM once)"},
                    "families_ms_per_step": {k: round(v["ms"] / NPROF, 3) for k, v in fams.items()},
                    "families_mfma_frac": {k: round(v["flops"] / (v["ms"] * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4)
                                           for k, v in fams.items() if v["flops"] > 0},
                    "encoder_tflops": round(enc_tf, 2),
                    "encoder_frac_of_mfma_peak": round(enc_tf / MFMA_PEAK_TFLOPS, 4),
                    "compulsory_bytes_per_step": int(COMPULSORY_BYTES_PER_FRAME * args.batch + WEIGHT_BYTES),
                    "layerwise_bytes_per_step": round(sum(v["bytes"] for v in fams.values()) / NPROF)}
        live = None
        if not args.no_parity_live and args.batch >= PARITY_FRAMES:
            enc.set_pipelined(False)
            live = parity_live(enc, x, mixed_labels, params32)
        out = {"metric": "frames/sec DenseNet-121 224x224 feature-extract", "value": round(fps, 1),
               "unit": "frames/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f16", "data": "synthetic",
               "config": {"workload": "DenseNet-121 frame feature-extract, batch 256 synthetic 224x224x3 "
                                      "(BASELINE.json configs[1])",
                          "frames_per_step_per_gpu": args.batch,
                          "input": "NHWC uint8 decoded frames, HBM-resident (ToTensor + Normalize inside the stem); "
                                   + (f"frames [0, {len(mixed_labels)}) are two of each of the 16 frame families + two fine checkerboards (what config.parity_live checks), the rest uniform noise; " if mixed_labels else "uniform noise; ")
                                   + "every step re-reads the SAME "
                                   f"{args.batch * SIZE * SIZE * 3 / 1e6:.1f} MB buffer, which fits the 256 MB Infinity Cache: the stem's HBM read is not exercised (about 1 % of the step)",
                          "output": "fp32 features (B,1024)" + ("; RCCL all-gather of feature rows" if world > 1 else ""),
                          "weights": ("seeded random-init, fp32 conv weights as hi + lo fp16 pairs (exact-weights mode, 2x MFMA work "
                                      "in the dense layers / transitions)") if args.exact_weights
                                     else ("seeded random-init, conv weights fp16-representable" if args.plain_rounding else
                                           "seeded random-init fp32 conv weights (not fp16-representable: what a trained checkpoint looks like), converted to ONE fp16 "
                                           "number per weight by calibrated rounding against the library's built-in calibration frames with bias correction "
                                           "(tennis_amd/calibrate.py, weights.as_fp16_model)"),
                          "parity_live": live,
                          "parity": (parity_note() if not (args.exact_weights or args.plain_rounding) else None),
                          "comm": comm_record,
                          "exchange": (comm.transport + f", all-gather of {args.batch} x {enc.feature_dim} fp32 rows per rank and step") if comm is not None else "none (1 rank)",
                          "timing": f"median of {repeats} fenced regions of exactly {args.steps} steps" + (", forwards pipelined (results joined one step behind, all joined before the closing fence)" if pipelined else ""),
                          "region_ms": [round(t * 1e3, 2) for t in times],
                          "frames_per_sec_forwards_joined": (round(world * args.batch * args.steps / dt_joined, 1) if dt_joined else None),
                          **({"exact_weights_frames_per_sec": round(fps_exact, 1),
                              "exact_weights_note": "the SAME fp32 parameters with every conv weight (stem included) as hi + lo fp16 pairs, no calibration: the mode for inputs "
                                                    "outside anything a calibration set resembles - in the wide evaluation (config.parity.wide) it has no value over the bar where "
                                                    "the timed configuration has a handful on full-contrast checkerboards of 2-3 px period"} if fps_exact else {})},
               "roofline": roofline}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(params32, x, full=args.cpu_baseline_full)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = build_parser().parse_args(argv)
    from tennis_amd import sharding
    if args.gpus > 1 and not sharding.under_launcher():
        n = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if n < args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but {n} GPUs are visible")
        sharding.launch(run, args.gpus, (argv,))
        return
    run(argv)


if __name__ == "__main__":
    main()
