#!/usr/bin/env python3
"""bench.py -- clips/sec (forward + backward + optimizer step) of ORViT-Motionformer 16x224, bf16, on N MI355X.

  python bench.py --gpus N --steps K --warmup W            (N>1: launched by torch.distributed.run)

One step = one pass of the hot path over one synthetic minibatch (batch 8 clips per GPU, 4 boxes/frame):
build_model(cfg) -> model(inputs, meta) -> label-smoothing CE -> backward (DDP/RCCL all-reduce overlapped) ->
clip-norm -> AdamW.  Prints ONE JSON line on rank 0 (contract in the task statement), with
  roofline     : the dominant kernel (bf16 MFMA GEMM), executed 2*M*N*K FLOPs / HIP-event time, vs 2.5 PF/s
  cpu_baseline : the CPU oracle (oracle/focus_oracle.py, fp32) timed on this box's host cores on 1 clip.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0                 # MI355X dense bf16 MFMA (MI355X_MICROARCH.md chip table)
ALG_GF_PER_CLIP = 1894.9                  # fwd+bwd GFLOP per clip, 16x224, O=4


def make_cfg(n_gpus, per_gpu_batch, mixed=True):
    from focus_amd.slowfast.config.defaults import get_cfg
    cfg = get_cfg()
    # configs/ORViT/SSv2_ORViT-MF_224_16x4.yaml (hot-path keys) with the synthetic-run overrides of SURVEY 8(d)
    cfg.merge_from_list([
        "ORVIT.ENABLE", True, "ORVIT.O", 4, "ORVIT.LAYERS", [1, 6, 10], "ORVIT.USE_MOTION_STREAM", True,
        "ORVIT.MOTION_STREAM_ATTN_TYPE", "joint", "TRAIN.DATASET", "Ssv2", "TRAIN.METHOD", "sup",
        "TRAIN.BATCH_SIZE", per_gpu_batch * max(n_gpus, 1), "TRAIN.MIXED_PRECISION", mixed,
        "DATA.NUM_FRAMES", 16, "DATA.TRAIN_CROP_SIZE", 224, "MF.PATCH_SIZE", 16, "MF.PATCH_SIZE_TEMP", 2,
        "MF.EMBED_DIM", 768, "MF.DEPTH", 12, "MF.NUM_HEADS", 12, "MF.MLP_RATIO", 4, "MF.QKV_BIAS", True,
        "MF.TEMPORAL_RESOLUTION", 8, "MF.USE_MLP", True, "MF.DROP_PATH", 0.2, "MF.HEAD_ACT", "tanh",
        "MODEL.NUM_CLASSES", 174, "MODEL.MODEL_NAME", "Motionformer",
        "MODEL.LOSS_FUNC", "label_smoothing_cross_entropy", "SOLVER.BASE_LR", 5e-5, "SOLVER.WEIGHT_DECAY", 5e-2,
        "SOLVER.OPTIMIZING_METHOD", "adamw", "NUM_GPUS", n_gpus, "RNG_SEED", 0,
    ])
    return cfg


def cpu_baseline(model, cfg, seconds_budget=30.0):
    """The CPU oracle on a bounded sample (1 clip, fwd+bwd, fp32), on this host's cores."""
    from focus_amd.train import synthetic_batch
    from oracle import focus_oracle as fo
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    base = model.module if hasattr(model, "module") else model
    params = {k: v.detach().float().cpu().clone().requires_grad_(v.is_floating_point())
              for k, v in base.state_dict().items()}
    inputs, labels, meta = synthetic_batch(cfg, 1, "cpu", seed=123)
    ocfg = dict(depth=cfg.MF.DEPTH, heads=cfg.MF.NUM_HEADS, orvit_layers=list(cfg.ORVIT.LAYERS),
                temporal_resolution=cfg.MF.TEMPORAL_RESOLUTION,
                patch=(cfg.MF.PATCH_SIZE_TEMP, cfg.MF.PATCH_SIZE, cfg.MF.PATCH_SIZE), crop=cfg.DATA.TRAIN_CROP_SIZE)
    times, logits = [], None
    t_all = time.time()
    for it in range(3):
        t0 = time.time()
        logits = fo.motionformer_forward(params, inputs[0], meta["orvit_bboxes"], ocfg, training=True)
        fo.label_smoothing_ce(logits, labels).backward()
        times.append(time.time() - t0)
        if time.time() - t_all > seconds_budget:
            break
    t = min(times[1:]) if len(times) > 1 else times[0]
    # the same clip through the HIP path (fp32 masters, bf16 compute): a full-size parity figure
    was_training = base.training
    base.eval()
    with torch.no_grad():
        gi, _, gm = synthetic_batch(cfg, 1, "cuda", seed=123)
        probs = base(gi, gm).float().cpu()
    base.train(was_training)
    ref = torch.softmax(logits.detach(), dim=-1)
    err = float((probs - ref).abs().max() / ref.abs().max())
    return {"value": round(1.0 / t, 4), "unit": "clips/s", "cores": cores, "kind": "port",
            "sample": "1 clip ORViT-MF 16x224 O=4 fwd+bwd fp32 through oracle/focus_oracle.py, best of %d" % max(1, len(times) - 1),
            "hip_vs_oracle_rel_err_bf16": round(err, 5)}


def pmc_traffic():
    """HBM-side bytes per NT-GEMM launch from the committed PMC passes (bench.py cannot run rocprofv3 on itself):
    last line of profiles/r01_pmc_hbm_traffic.txt, or None when the file is absent."""
    try:
        last = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01_pmc_hbm_traffic.txt")).read().strip().splitlines()[-1]
        return round(float(last.rsplit("=", 1)[1].split("MB")[0]) * 1e6)
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="clips per GPU")
    ap.add_argument("--fp32", action="store_true", help="TRAIN.MIXED_PRECISION False (precision path)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gemm-shapes", action="store_true", help="print per-shape GEMM timings to stderr")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="process-group backend (nccl = RCCL; gloo for a rehearsal)")
    ap.add_argument("--same-device", action="store_true",
                    help="rehearsal only: every rank uses cuda:0 (needs --backend gloo; RCCL wants one GPU per rank)")
    args = ap.parse_args()

    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`"
                         % (args.gpus, args.gpus))
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=args.backend, rank=rank, world_size=world)   # "nccl" == RCCL on ROCm

    from focus_amd import ops
    from focus_amd.slowfast.models import build_model
    from focus_amd.slowfast.models.losses import get_loss_func
    from focus_amd.slowfast.models.optimizer import construct_optimizer
    from focus_amd.train import synthetic_batch, train_step

    cfg = make_cfg(world if world > 1 else 1, args.batch, mixed=not args.fp32)
    cfg.DIST_BACKEND = args.backend
    torch.manual_seed(cfg.RNG_SEED)
    model = build_model(cfg, gpu_id=local_rank)
    model.train()
    optimizer = construct_optimizer(model, cfg)
    loss_fun = get_loss_func(cfg)(reduction="mean")
    inputs, labels, meta = synthetic_batch(cfg, args.batch, dev, seed=1000 + rank)   # resident in HBM

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        train_step(model, optimizer, loss_fun, inputs, labels, meta, cfg)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        _, loss = train_step(model, optimizer, loss_fun, inputs, labels, meta, cfg)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    clips = args.batch * world * args.steps
    value = clips / dt
    out = {
        "metric": "clips/sec (fwd+bwd+opt step) ORViT-MF 16x224", "value": round(value, 3), "unit": "clips/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if args.fp32 else "bf16", "data": "synthetic",
        "config": {"workload": "ORViT-Motionformer 16x224, 4 objects, batch=%d per GPU (BASELINE configs[1])" % args.batch,
                   "global_batch": args.batch * world, "frames": 16, "crop": 224, "objects": 4,
                   "parallelism": "dp%d" % world, "optimizer": "adamw+clipnorm", "drop_path": cfg.MF.DROP_PATH},
        "clips_per_sec_per_gpu": round(value / world, 3),
        "model_mfma_frac_algorithmic": round(value / world * ALG_GF_PER_CLIP * 1e9 / (PEAK_BF16_TFLOPS * 1e12), 5),
        "final_loss": round(float(loss.detach()), 4),
    }

    if rank == 0 and not args.no_roofline and not args.fp32:
        # second pass over the same K steps with HIP events around every launch of the dominant kernel
        ops.GEMM_TIMING = []
        for _ in range(args.steps):
            train_step(model, optimizer, loss_fun, inputs, labels, meta, cfg)
        torch.cuda.synchronize()
        allrecs, ops.GEMM_TIMING = ops.GEMM_TIMING, None
        recs = [r for r in allrecs if r[3] == "nt"]          # the dominant kernel: forward / dX GEMMs
        tn = [r for r in allrecs if r[3] == "tn"]            # weight-gradient kernel, reported beside it
        fl = sum(r[0] for r in recs)
        ms = sum(r[1].elapsed_time(r[2]) for r in recs)
        ach = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        tn_ms = sum(r[1].elapsed_time(r[2]) for r in tn)
        if args.gemm_shapes:
            import collections
            by = collections.defaultdict(lambda: [0, 0.0, 0.0])
            for r in allrecs:
                e = by[(r[3],) + r[4]]
                e[0] += 1; e[1] += r[1].elapsed_time(r[2]); e[2] += r[0]
            print("kind (M,N,K,batch,epi)  calls/step  avg_us  TF/s  ms/step", file=sys.stderr)
            for k, e in sorted(by.items(), key=lambda kv: -kv[1][1]):
                print("%-44s %5.1f %8.1f %7.0f %7.3f" % (k, e[0] / args.steps, 1e3 * e[1] / e[0], e[2] / e[1] / 1e9,
                                                         e[1] / args.steps), file=sys.stderr)
        tn_tf = sum(r[0] for r in tn) / (tn_ms * 1e-3) / 1e12 if tn_ms > 0 else 0.0
        # algorithmic HBM bytes of the same launches: A + B + C (+ aux, residual are not known here: lower bound)
        alg = sum(2.0 * r[4][3] * (r[4][0] * r[4][2] + r[4][1] * r[4][2] + r[4][0] * r[4][1]) for r in recs)
        out["roofline"] = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                           "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": pmc_traffic(),
                           "traffic_unit": "bytes/launch (PMC: 2*FETCH_SIZE + WRITE_SIZE, separate rocprofv3 passes, "
                                           "profiles/r01_pmc_hbm_traffic.txt)",
                           "algorithmic_bytes_per_launch": round(alg / max(len(recs), 1)),
                           "kernel": "gemm_nt_ws_kernel / gemm_nt_kernel, bf16 NT GEMM (focus_amd/csrc/gemm_mfma_ws.hip, gemm_mfma.hip)",
                           "launches_per_step": len(recs) // max(args.steps, 1),
                           "avg_launch_us": round(1e3 * ms / max(len(recs), 1), 2),
                           "kernel_ms_per_step": round(ms / max(args.steps, 1), 3),
                           "weight_grad_kernel": {"kernel": "gemm_tn_ws_kernel / gemm_tn_kernel (gemm_mfma_tn_ws.hip, gemm_mfma_tn.hip)",
                                                  "achieved": round(tn_tf, 2),
                                                  "ms_per_step": round(tn_ms / max(args.steps, 1), 3)}}
    elif world > 1 and not args.no_roofline and not args.fp32:
        for _ in range(args.steps):       # keep ranks in lock-step with rank 0's instrumented pass
            train_step(model, optimizer, loss_fun, inputs, labels, meta, cfg)
        torch.cuda.synchronize()

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(model, cfg)
        except Exception as e:  # the baseline is a reported figure; never lose the bench line to it
            out["cpu_baseline"] = {"value": None, "unit": "clips/s", "cores": os.cpu_count(), "kind": "port",
                                   "sample": "failed: %r" % (e,)}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
