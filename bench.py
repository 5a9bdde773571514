#!/usr/bin/env python
"""bench.py -- BASELINE.json's headline metric on the fused HIP path.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Metric  : directional samples / second, forward + backward (one sample = one (image, direction)
          pair through encoding -> SIREN -> loss -> gradients).
Workload: BASELINE config 2 -- a 615-image training set, 128x256 equirect (P = 32768), ND = 36,
          5x128 SIREN, SO2, tanh, AutoDecoder, RENITrainLoss, bf16 MFMA; per-GPU batch = 64 images
          (the largest power of two that 8 ranks can each draw from their 76-77 owned images).  Synthetic images / random-init weights (seed 42).
Step    : ONE full training iteration = fused forward+loss+backward (decoder and latent gradients)
          -> [N > 1: one RCCL all-reduce of the flat decoder gradient] -> Adam on decoder + latents.
Scaling : weak (per-GPU batch fixed; images and their latent rows sharded round-robin over ranks).

Rank 0 prints ONE JSON line (contract in the task statement) including `roofline` (dominant kernel,
timed live with HIP events on its own stream) and, at N = 1, `cpu_baseline` (the CPU oracle's
reference-shaped step timed on this box's host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

FLOP_PER_SAMPLE = 522784       # fwd+bwd, factored form (SURVEY.md 8d / BASELINE.md section 4)
PEAK_TFLOPS = {"bf16": 2500.0, "f32": 157.3}   # dense MFMA peaks, MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU per step (<= 615/8 so 8 ranks can own them)")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=8)
    return ap.parse_args()


def cpu_baseline(n_steps):
    """Reference-shaped CPU step (oracle: materialised encoding, linear+sin, autograd, WeightedMSE) on
    the same workload shape, B = 1 image per step; median of n_steps after 2 warm-ups."""
    from oracle import reni_oracle as O
    spec = O.DecoderSpec(36, "SO2", 128, 5, 3, True, "tanh")
    g = torch.Generator().manual_seed(42)
    params = O.init_params(spec, g)
    Z = torch.randn(1, 36, 3, generator=g)
    D = O.get_directions(256); S = O.get_sineweight(256)
    T = O.synthetic_images([0], 128, 256).permute(0, 2, 3, 1).reshape(1, -1, 3)
    times = []
    for i in range(n_steps + 2):
        t0 = time.perf_counter()
        O.fwd_loss_bwd(spec, params, Z, D, T, S)
        if i >= 2:
            times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {"value": D.shape[1] / med, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n_steps} fwd+loss+bwd steps of 1 image x 32768 directions (config-2 shape, fp32, "
                      f"reference-shaped: materialised 1370-column encoding + autograd), median"}


def main():
    args = parse()
    from reni_amd import dist as rdist
    rank, world, local = rdist.init_from_env()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    from reni_amd import ops
    from reni_amd.data import SyntheticEnvMapDataset
    from reni_amd.engine import TrainEngine
    from reni_amd.models import RENIAutoDecoder
    from reni_amd.utils import get_directions, get_sineweight

    N_IMAGES, H_IMG, W_IMG, ND = 615, 128, 256, 36
    B = args.batch
    owned = rdist.owned_indices(N_IMAGES, rank, world)
    torch.manual_seed(42)
    model = RENIAutoDecoder(N_IMAGES, ND, "SO2", 128, 5, 3, True, "tanh", 30.0, 30.0, False)
    model.set_compute_dtype(args.dtype)
    with torch.no_grad():  # this rank keeps only the latent rows of the images it owns
        model.Z = torch.nn.Parameter(model.Z[owned].clone())
    model.to(dev)
    rdist.broadcast_(model._flat_params(), 0)
    eng = TrainEngine(model, lr=1e-5)

    # this rank's shard of the synthetic training set, resident in HBM before the timed region
    ds = SyntheticEnvMapDataset(N_IMAGES, H_IMG, W_IMG)
    n_local = len(owned)
    assert n_local >= B, f"per-GPU batch {B} exceeds the {n_local} images this rank owns"
    imgs = torch.stack([ds.make(i) for i in owned]).to(dev)      # [n_local,3,H,W], ~0.39 MB per image
    directions = get_directions(W_IMG).to(dev)
    sineweight = get_sineweight(W_IMG).to(dev)
    P = directions.shape[1]

    idx_all = torch.arange(n_local, device=dev)  # (the loader's indices: resident like the images)

    def batch(step):
        """B consecutive owned images; targets are the reference's permute+view of [B,3,H,W]
        (RENI_module.py:83-84): a channel-planar strided view, never copied."""
        start = (step * B) % (n_local - B + 1)
        return idx_all[start:start + B], imgs[start:start + B].permute(0, 2, 3, 1).view(B, P, 3)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    if world > 1:  # create the RCCL communicator outside the timed region even with --warmup 0
        torch.distributed.all_reduce(torch.zeros(1, device=dev))
    for s in range(args.warmup):
        idx, tgt = batch(s)
        eng.step(idx, tgt, sineweight, directions)
    barrier()
    ops.profile_enable(True)
    ops.profile_read(reset=True)
    t0 = time.perf_counter()
    last = None
    for s in range(args.steps):
        idx, tgt = batch(args.warmup + s)
        last = eng.step(idx, tgt, sineweight, directions)
    barrier()
    dt = time.perf_counter() - t0
    kern_ms, kern_n = ops.profile_read(reset=True)
    ops.profile_enable(False)
    loss = float(last[0])
    assert loss == loss and abs(loss) < 1e6, f"non-finite loss {loss}"

    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    dt = float(tmax)
    samples_per_step = world * B * P
    value = samples_per_step * args.steps / dt

    if rank == 0:
        kavg_ms = kern_ms / max(kern_n, 1)
        achieved = B * P * FLOP_PER_SAMPLE / (kavg_ms * 1e-3) / 1e12
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            except Exception:  # noqa: BLE001
                traffic = None
        line = {
            "metric": "directional-samples/sec (fwd+bwd), 128x256 equirect, ND=36 latent",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "BASELINE config 2: 615-image set, 128x256 equirect, ND=36, 5x128 SIREN, SO2, "
                                   "tanh, AutoDecoder, RENITrainLoss; full training step (fwd+loss+bwd, grad "
                                   "all-reduce, Adam)", "images_per_gpu_per_step": B,
                       "global_batch_images": world * B, "directions_per_image": P,
                       "parallelism": f"dp{world}", "loss_last_step": loss},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_TFLOPS[args.dtype], "unit": "TFLOP/s",
                         "frac": achieved / PEAK_TFLOPS[args.dtype], "traffic": traffic,
                         "kernel": "k_reni_train_bf16<128,true>" if args.dtype == "bf16" else "k_reni_main<f32,H=128,FWD_BWD>",
                         "kernel_avg_ms": kavg_ms,
                         "kernel_launches": kern_n, "flop_per_sample": FLOP_PER_SAMPLE},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.cpu_steps)
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
