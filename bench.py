#!/usr/bin/env python
"""Headline benchmark: frames/sec of the DenseNet-121 224x224 frame feature-extract
(BASELINE.json configs[1]: batch 256 synthetic frames, fp16, one MI355X per rank).

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A step = one pass of the hot path over one batch of 256 synthetic frames per rank
(inputs resident in HBM): stem -> 58 dense layers -> 3 transitions -> head ->
(B,1024) fp32 features, then for N>1 the RCCL all-gather of the feature rows that
the temporal/caption stage consumes.  Frames shard across ranks with
no other exchange: weak scaling.  Rank 0 prints ONE JSON line.
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


def make_frames(batch, size, seed, device):
    """uint8 uniform frames -> ToTensor+Normalize -> NHWC fp16, generated on the device."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    u8 = torch.randint(0, 256, (batch, size, size, 3), generator=g, device=device, dtype=torch.uint8)
    mean = torch.tensor([0.485, 0.456, 0.406], device=device)
    std = torch.tensor([0.229, 0.224, 0.225], device=device)
    return ((u8.float() / 255.0 - mean) / std).half().contiguous()


def pmc_traffic(kernel_family):
    """HBM bytes per launch of the dominant kernel family from the newest committed rocprofv3 PMC
    summary (profiles/*_pmc_traffic.json: FETCH_SIZE / WRITE_SIZE passes with the gfx950 x2
    FETCH correction).  PMC counters cannot be read from inside the timed process."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
    if not files:
        return None
    fam = json.load(open(files[-1])).get("families", {})
    return fam.get(kernel_family, {}).get("hbm_bytes_per_dispatch_corrected")


def cpu_baseline(params, frames_nhwc_f16, seconds_target=12.0):
    """Reference CPU path stand-in (oracle/torch_ref.py, fp32, oneDNN) on a bounded sample."""
    from oracle.torch_ref import TorchDenseNet121
    net = TorchDenseNet121(params)
    n = 16
    x = frames_nhwc_f16[:n].float().permute(0, 3, 1, 2).contiguous().cpu()
    threads = torch.get_num_threads()
    net(x[:2])  # warm-up
    t0 = time.time()
    reps = 0
    while True:
        net(x)
        reps += 1
        if time.time() - t0 > seconds_target or reps >= 8:
            break
    dt = time.time() - t0
    return {"value": round(n * reps / dt, 2), "unit": "frames/sec", "cores": threads, "kind": "port",
            "sample": f"{reps} x {n} frames 224x224 fp32, torch-CPU(oneDNN) restatement oracle/torch_ref.py "
                      f"(MXNet CPU path not installable), host has {os.cpu_count()} logical cpus"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run",
                  file=sys.stderr)
        args.gpus = world
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    from tennis_amd import _lib
    from tennis_amd import weights as W
    from tennis_amd.engine import DenseNet121Features

    ctx = _lib.Context(local_rank)
    params = W.make_densenet121_weights(0)
    enc = DenseNet121Features(params, SIZE, max_batch=args.batch, ctx=ctx)
    x = make_frames(args.batch, SIZE, 1234 + rank, dev)
    feats = [torch.empty((args.batch, enc.feature_dim), dtype=torch.float32, device=dev) for _ in range(2)]
    gathered = [torch.empty((world * args.batch, enc.feature_dim), dtype=torch.float32, device=dev)
                for _ in range(2)] if world > 1 else None

    def step(i):
        f = feats[i & 1]
        enc(x, out=f)
        if world > 1:
            dist.all_gather_into_tensor(gathered[i & 1], f)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        fps = world * args.batch * args.steps / dt
        # ---- roofline of the dominant kernel family: HIP events around every launch
        # (separate instrumented passes, so the timed region above carries no events)
        fams = {}
        for _ in range(3):
            stats, _ = enc.profile(x)
            for s in stats:
                a = fams.setdefault(s["name"], dict(ms=0.0, flops=0.0, bytes=0.0, launches=0))
                a["ms"] += s["ms"]; a["flops"] += s["flops"]; a["bytes"] += s["bytes"]; a["launches"] += s["launches"]
        dom = max(fams, key=lambda k: fams[k]["ms"])
        d = fams[dom]
        # which roofline binds the dominant kernel: time at HBM peak for its algorithmic bytes vs time at the
        # dense-fp16 MFMA peak for its algorithmic flops (DenseNet's concatenated inputs make the dense layers
        # byte-heavy: 128K+36864 flop per K+32 fp16 values read/written is below the 312 flop/B ridge from K=160 on)
        secs = d["ms"] * 1e-3
        tf, tbs = d["flops"] / secs / 1e12, d["bytes"] / secs / 1e12
        hbm_bound = d["bytes"] / (HBM_PEAK_TBS * 1e12) >= d["flops"] / (MFMA_PEAK_TFLOPS * 1e12)
        roofline = {"bound": "hbm" if hbm_bound else "mfma", "kernel": dom,
                    "achieved": round(tbs * 1e3 if hbm_bound else tf, 2),
                    "peak": HBM_PEAK_TBS * 1e3 if hbm_bound else MFMA_PEAK_TFLOPS,
                    "unit": "GB/s" if hbm_bound else "TFLOP/s",
                    "frac": round(tbs / HBM_PEAK_TBS if hbm_bound else tf / MFMA_PEAK_TFLOPS, 4),
                    "traffic": pmc_traffic(dom),
                    "algorithmic_bytes_per_launch": round(d["bytes"] / d["launches"]),
                    "algorithmic_flops_per_launch": round(d["flops"] / d["launches"]),
                    "avg_launch_us": round(d["ms"] / d["launches"] * 1e3, 2), "launches_per_step": d["launches"] // 3,
                    "other_roofline": {"bound": "mfma" if hbm_bound else "hbm",
                                       "achieved": round(tf if hbm_bound else tbs * 1e3, 2),
                                       "peak": MFMA_PEAK_TFLOPS if hbm_bound else HBM_PEAK_TBS * 1e3,
                                       "unit": "TFLOP/s" if hbm_bound else "GB/s",
                                       "frac": round(tf / MFMA_PEAK_TFLOPS if hbm_bound else tbs / HBM_PEAK_TBS, 4)},
                    "families_ms_per_step": {k: round(v["ms"] / 3, 3) for k, v in fams.items()},
                    "encoder_tflops": round(FLOP_PER_FRAME * fps / world / 1e12, 2),
                    "encoder_frac_of_mfma_peak": round(FLOP_PER_FRAME * fps / world / 1e12 / MFMA_PEAK_TFLOPS, 4)}
        out = {"metric": "frames/sec DenseNet-121 224x224 feature-extract", "value": round(fps, 1),
               "unit": "frames/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f16", "data": "synthetic",
               "config": {"workload": "DenseNet-121 frame feature-extract, batch 256 synthetic 224x224x3 "
                                      "(BASELINE.json configs[1])",
                          "frames_per_step_per_gpu": args.batch, "input": "NHWC fp16 normalised, HBM-resident",
                          "output": "fp32 features (B,1024)" + ("; RCCL all-gather of feature rows" if world > 1 else ""),
                          "weights": "seeded random-init, conv weights fp16"},
               "roofline": roofline}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(params, x)
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
