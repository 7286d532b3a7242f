#!/usr/bin/env python3
"""bench.py -- clips/sec (forward + backward + optimizer step) of ORViT-Motionformer 16x224, bf16, on N MI355X.

  python bench.py --gpus N --steps K --warmup W

N > 1 launches itself: the parent (which never touches the GPU) spawns one process per GPU through the mirror of the
reference's launcher (focus_amd/slowfast/utils/misc.py:launch_job -> multiprocessing.run, slowfast/utils/misc.py:285-313)
and exits non-zero if any child fails.  Under `python -m torch.distributed.run ... bench.py --gpus N` (RANK/WORLD_SIZE in
the environment) each process joins the group it was given instead.

One step = one pass of the hot path over one synthetic minibatch (batch 8 clips per GPU, 4 boxes/frame):
build_model(cfg) -> model(inputs, meta) -> label-smoothing CE -> backward (DDP/RCCL all-reduce overlapped) ->
clip-norm -> AdamW.  Prints ONE JSON line on rank 0 (contract in the task statement), with
  roofline     : the dominant kernel (bf16 MFMA GEMM), executed 2*M*N*K FLOPs / HIP-event time, vs 2.5 PF/s
  cpu_baseline : the CPU oracle (oracle/focus_oracle.py, fp32) timed on this box's host cores on a bounded sample
  steve / hr   : sub-records for BASELINE configs[2] (slot-attention update at B=32,T=24,N=4096,K=11) and the
                 configs[4] shape (16x336, 6 objects, EK heads) on one GPU  (N == 1 only; --workload picks one alone).
"""
import argparse
import json
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0                 # MI355X dense bf16 MFMA (MI355X_MICROARCH.md chip table)
PEAK_HBM_GBS = 8000.0                     # MI355X HBM3E spec
ALG_GF_PER_CLIP = 1894.9                  # fwd+bwd GFLOP per clip, 16x224, O=4
ALG_GF_PER_CLIP_HR = 5099.0               # 16x336, O=6


def make_cfg(n_gpus, per_gpu_batch, mixed=True, hr=False):
    from focus_amd.slowfast.config.defaults import get_cfg
    cfg = get_cfg()
    # configs/ORViT/SSv2_ORViT-MF_224_16x4.yaml (hot-path keys) with the synthetic-run overrides of SURVEY 8(d)
    cfg.merge_from_list([
        "ORVIT.ENABLE", True, "ORVIT.O", 4, "ORVIT.LAYERS", [1, 6, 10],
        # (FOCUS_BENCH_NO_MOTION=1: a tuning probe that drops the motion stream to show how much of it the step still waits for)
        "ORVIT.USE_MOTION_STREAM", os.environ.get("FOCUS_BENCH_NO_MOTION", "0") != "1",
        "ORVIT.MOTION_STREAM_ATTN_TYPE", "joint", "TRAIN.DATASET", "Ssv2", "TRAIN.METHOD", "sup",
        "TRAIN.BATCH_SIZE", per_gpu_batch * max(n_gpus, 1), "TRAIN.MIXED_PRECISION", mixed,
        "DATA.NUM_FRAMES", 16, "DATA.TRAIN_CROP_SIZE", 224, "MF.PATCH_SIZE", 16, "MF.PATCH_SIZE_TEMP", 2,
        "MF.EMBED_DIM", 768, "MF.DEPTH", 12, "MF.NUM_HEADS", 12, "MF.MLP_RATIO", 4, "MF.QKV_BIAS", True,
        "MF.TEMPORAL_RESOLUTION", 8, "MF.USE_MLP", True, "MF.DROP_PATH", 0.2, "MF.HEAD_ACT", "tanh",
        "MODEL.NUM_CLASSES", 174, "MODEL.MODEL_NAME", "Motionformer",
        "MODEL.LOSS_FUNC", "label_smoothing_cross_entropy", "SOLVER.BASE_LR", 5e-5, "SOLVER.WEIGHT_DECAY", 5e-2,
        "SOLVER.OPTIMIZING_METHOD", "adamw", "NUM_GPUS", n_gpus, "RNG_SEED", 0,
    ])
    if hr:
        # configs/ORViT/EK_ORVIT_MF_HR.yaml:3,11,15,22 with ORVIT.O 6 (BASELINE configs[4])
        cfg.merge_from_list(["DATA.TRAIN_CROP_SIZE", 336, "DATA.TEST_CROP_SIZE", 336, "ORVIT.O", 6,
                             "TRAIN.DATASET", "epickitchens", "MODEL.NUM_CLASSES", 97])
    return cfg


def host_cores():
    """(threads this process may use, physical cores among them)."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        allowed = list(range(os.cpu_count() or 1))
    phys = set()
    try:
        cur = {}
        for line in open("/proc/cpuinfo"):
            if ":" in line:
                k, v = [s.strip() for s in line.split(":", 1)]
                cur[k] = v
            elif cur:
                if int(cur.get("processor", -1)) in allowed:
                    phys.add((cur.get("physical id", "0"), cur.get("core id", cur.get("processor"))))
                cur = {}
        if cur and int(cur.get("processor", -1)) in allowed:
            phys.add((cur.get("physical id", "0"), cur.get("core id", cur.get("processor"))))
    except Exception:
        pass
    return len(allowed), (len(phys) or len(allowed))


def cpu_baseline(model, cfg, seconds_budget=75.0):
    """the CPU oracle, fp32, B=2 clips fwd+bwd on this host: 1 warm-up + 3 timed iterations with 32 threads
    (the setting that was fastest on the shared GPU-box host) AND, budget permitting, ONE iteration with one thread per
    physical core (consistently the slower setting, so one sample is enough); `value` is the better of the two, both
    are reported.  Plus a 1-thread figure on one
    TrajectoryAttentionBlock of the same sample (the whole model on one thread would take minutes)."""
    import torch
    from focus_amd.train import synthetic_batch
    from oracle import focus_oracle as fo
    nthreads, phys = host_cores()
    base = model.module if hasattr(model, "module") else model
    params = {k: v.detach().float().cpu().clone().requires_grad_(v.is_floating_point())
              for k, v in base.state_dict().items()}
    Bc = 2
    inputs, labels, meta = synthetic_batch(cfg, Bc, "cpu", seed=123)
    ocfg = dict(depth=cfg.MF.DEPTH, heads=cfg.MF.NUM_HEADS, orvit_layers=list(cfg.ORVIT.LAYERS),
                temporal_resolution=cfg.MF.TEMPORAL_RESOLUTION,
                patch=(cfg.MF.PATCH_SIZE_TEMP, cfg.MF.PATCH_SIZE, cfg.MF.PATCH_SIZE), crop=cfg.DATA.TRAIN_CROP_SIZE)
    t_all = time.time()
    state = {"logits": None}

    def run(threads, budget_left, iters=4):
        """-> (mean seconds of the timed iterations, how many were timed) with `threads` threads; 1 warm-up first
        (with iters=1 the single iteration is the sample: an upper bound on the time, allocator warm-up included)."""
        torch.set_num_threads(threads)
        times = []
        for it in range(iters):
            t0 = time.time()
            for p in params.values():
                p.grad = None
            state["logits"] = fo.motionformer_forward(params, inputs[0], meta["orvit_bboxes"], ocfg, training=True)
            fo.label_smoothing_ce(state["logits"], labels).backward()
            times.append(time.time() - t0)
            if it >= 1 and time.time() - t_all + times[-1] > budget_left:
                break
        timed = times[1:] if len(times) > 1 else times
        return sum(timed) / len(timed), len(timed)

    c32 = max(1, min(phys, nthreads, 32))
    t32, n32 = run(c32, seconds_budget * 0.55)
    runs = {"threads_%d" % c32: {"clips_per_s": round(Bc / t32, 4), "timed_iterations": n32}}
    best_t, best_c = t32, c32
    call = max(1, min(phys, nthreads))
    if call != c32 and time.time() - t_all + 2.5 * t32 < seconds_budget:
        # one thread per core has been 3-4x SLOWER than 32 threads on every GPU-box host so far (oversubscribed
        # small-tensor ops): one iteration is enough to show it, a warm-up + 3 more would add a minute to the bench
        tp, npn = run(call, seconds_budget, iters=1)
        runs["threads_%d" % call] = {"clips_per_s": round(Bc / tp, 4), "timed_iterations": npn}
        if tp < best_t:
            best_t, best_c = tp, call
    cores, t, logits = best_c, best_t, state["logits"]
    torch.set_num_threads(cores)
    # one block, same sample size, all cores vs one thread (thread-scaling normalisation)
    D = cfg.MF.EMBED_DIM
    N = 1 + cfg.MF.TEMPORAL_RESOLUTION * (cfg.DATA.TRAIN_CROP_SIZE // cfg.MF.PATCH_SIZE) ** 2
    bp = {k: v.detach().clone().requires_grad_(True) for k, v in params.items() if k.startswith("blocks.0.")}
    xb = torch.randn(1, N, D, generator=torch.Generator().manual_seed(5))
    side = cfg.DATA.TRAIN_CROP_SIZE // cfg.MF.PATCH_SIZE

    def block_once():
        t0 = time.time()
        y = fo.trajectory_block(bp, "blocks.0", xb.clone().requires_grad_(True),
                                [cfg.MF.TEMPORAL_RESOLUTION, side, side], cfg.MF.NUM_HEADS)
        y.sum().backward()
        return time.time() - t0
    one = None
    try:
        block_once()
        t_all_block = block_once()
        torch.set_num_threads(1)
        t_one_block = block_once()
        one = {"block": "TrajectoryAttentionBlock fwd+bwd, 1 clip", "all_core_s": round(t_all_block, 3),
               "one_thread_s": round(t_one_block, 3),
               "clips_per_s_one_thread_extrapolated": round(Bc / t * t_all_block / t_one_block, 5)}
    except Exception as e:       # the normalisation figure must never lose the baseline
        one = {"failed": repr(e)}
    finally:
        torch.set_num_threads(cores)
    # the same clips through the HIP path (fp32 masters, bf16 compute): a full-size parity figure
    was_training = base.training
    base.eval()
    with torch.no_grad():
        gi, _, gm = synthetic_batch(cfg, Bc, "cuda", seed=123)
        probs = base(gi, gm).float().cpu()
    base.train(was_training)
    ref = torch.softmax(logits.detach(), dim=-1)
    err = float((probs - ref).abs().max() / ref.abs().max())
    return {"value": round(Bc / t, 4), "unit": "clips/s", "cores": cores, "cores_physical": phys, "threads_usable": nthreads,
            "kind": "port",
            "sample": "B=2 clips ORViT-MF 16x224 O=4 fwd+bwd fp32 through oracle/focus_oracle.py; 1 warm-up + up to 3 timed "
                      "iterations per thread count, mean; value = the faster thread count",
            "runs": runs, "one_thread": one, "hip_vs_oracle_rel_err_bf16": round(err, 5)}


def pmc_record(workload):
    """The newest committed PMC pass of `workload` (profiles/r*_pmc_hbm_traffic_<workload>.txt, last line `#json {...}`,
    written by tools/pmc_traffic.py) IF it was taken on the kernel sources this process runs (focus_amd.build.source_hash);
    otherwise None: a traffic figure measured on other kernels is not quoted (bench.py cannot run rocprofv3 on itself)."""
    import glob
    try:
        from focus_amd.build import source_hash
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic_%s.txt" % workload)))
        for path in reversed(files):
            last = open(path).read().strip().splitlines()[-1]
            if not last.startswith("#json "):
                continue
            rec = json.loads(last[6:])
            if rec.get("src") == source_hash():
                rec["file"] = os.path.basename(path)
                return rec
    except Exception:
        pass
    return None


def pmc_traffic(family="gemm_nt_ws_kernel", workload="orvit"):
    """(HBM-side bytes per launch of a kernel family, source string) or (None, reason)."""
    rec = pmc_record(workload)
    if rec is None:
        return None, "no PMC pass committed for the running kernel sources"
    fam = rec.get("families", {}).get(family)
    if not fam:
        return None, "%s: family %s not in the pass" % (rec["file"], family)
    return fam["bytes_per_launch"], "%s @ build %s" % (rec["file"], rec.get("head"))


# ----------------------------------------------------------------------------------------------------------------
# the job every rank runs
# ----------------------------------------------------------------------------------------------------------------
def bench_job(cfg):
    import torch
    import torch.distributed as dist
    a = cfg.BENCH
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    dev = torch.device("cuda", torch.cuda.current_device())

    from focus_amd import ops
    from focus_amd.slowfast.models import build_model
    from focus_amd.slowfast.models.losses import get_loss_func
    from focus_amd.slowfast.models.optimizer import construct_optimizer
    from focus_amd.slowfast.utils import distributed as du
    from focus_amd.train import synthetic_batch, train_step

    du.init_distributed_training(cfg)
    torch.manual_seed(cfg.RNG_SEED)
    out = None
    if a.workload in ("orvit", "all"):
        model = build_model(cfg, gpu_id=dev.index)
        model.train()
        optimizer = construct_optimizer(model, cfg)
        loss_fun = get_loss_func(cfg)(reduction="mean")
        inputs, labels, meta = synthetic_batch(cfg, a.batch, dev, seed=1000 + rank)   # resident in HBM

        def sync():
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()

        def timed(n, ctx=None):
            sync()
            t0 = time.perf_counter()
            loss = None
            for _ in range(n):
                if ctx is not None:
                    with ctx():
                        _, loss = train_step(model, optimizer, loss_fun, inputs, labels, meta, cfg)
                else:
                    _, loss = train_step(model, optimizer, loss_fun, inputs, labels, meta, cfg)
            sync()
            dt = time.perf_counter() - t0
            if world > 1:
                tt = torch.tensor([dt], device=dev, dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dt = float(tt.item())
            return dt, loss

        for _ in range(a.warmup):
            train_step(model, optimizer, loss_fun, inputs, labels, meta, cfg)
        dt, loss = timed(a.steps)

        clips = a.batch * world * a.steps
        value = clips / dt
        out = {
            "metric": "clips/sec (fwd+bwd+opt step) ORViT-MF 16x224", "value": round(value, 3), "unit": "clips/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(1e3 * dt / a.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if a.fp32 else "bf16", "data": "synthetic",
            "config": {"workload": "ORViT-Motionformer 16x224, 4 objects, batch=%d per GPU (BASELINE configs[1])" % a.batch,
                       "global_batch": a.batch * world, "frames": 16, "crop": 224, "objects": 4,
                       "parallelism": "dp%d" % world, "optimizer": "adamw+clipnorm", "drop_path": cfg.MF.DROP_PATH,
                       "backend": (cfg.DIST_BACKEND if world > 1 else None),
                       "grad_allreduce": ("bf16" if cfg.DDP_BF16_GRADS else "fp32") if world > 1 else None},
            "clips_per_sec_per_gpu": round(value / world, 3),
            "model_mfma_frac_algorithmic": round(value / world * ALG_GF_PER_CLIP * 1e9 / (PEAK_BF16_TFLOPS * 1e12), 5),
            "final_loss": round(float(loss.detach()), 4),
        }
        del loss          # no autograd graph of this workload may outlive it (bench_steve captures a HIP graph later)
        if world > 1:
            # exposed (non-overlapped) all-reduce: the same K steps with DDP's reducer switched off
            dt_ns, _ = timed(a.steps, ctx=model.no_sync)
            out["ms_per_step_no_allreduce"] = round(1e3 * dt_ns / a.steps, 3)
            out["exposed_allreduce_ms"] = round(max(0.0, 1e3 * (dt - dt_ns) / a.steps), 3)

        if rank == 0 and not a.no_roofline and not a.fp32:
            # second pass over the same K steps with HIP events around every launch of the dominant kernel
            ops.GEMM_TIMING = []
            for _ in range(a.steps):
                train_step(model, optimizer, loss_fun, inputs, labels, meta, cfg)
            torch.cuda.synchronize()
            allrecs, ops.GEMM_TIMING = ops.GEMM_TIMING, None
            recs = [r for r in allrecs if r[3] in ("nt_ws", "nt_ws8")]   # the dominant kernel: gemm_nt_ws_kernel (forward / dX GEMMs)
            other = [r for r in allrecs if r[3] == "nt"]         # uniform 128x128 kernel: skinny (M = 256) products
            other_ms = sum(r[1].elapsed_time(r[2]) for r in other)
            tn = [r for r in allrecs if r[3] == "tn"]            # weight-gradient kernel, reported beside it
            fl = sum(r[0] for r in recs)
            ms = sum(r[1].elapsed_time(r[2]) for r in recs)
            ach = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            tn_ms = sum(r[1].elapsed_time(r[2]) for r in tn)
            if a.gemm_shapes:
                import collections
                by = collections.defaultdict(lambda: [0, 0.0, 0.0])
                for r in allrecs:
                    e = by[(r[3],) + r[4]]
                    e[0] += 1; e[1] += r[1].elapsed_time(r[2]); e[2] += r[0]
                print("kind (M,N,K,batch,epi)  calls/step  avg_us  TF/s  ms/step", file=sys.stderr)
                for k, e in sorted(by.items(), key=lambda kv: -kv[1][1]):
                    print("%-44s %5.1f %8.1f %7.0f %7.3f" % (k, e[0] / a.steps, 1e3 * e[1] / e[0], e[2] / e[1] / 1e9,
                                                             e[1] / a.steps), file=sys.stderr)
            tn_tf = sum(r[0] for r in tn) / (tn_ms * 1e-3) / 1e12 if tn_ms > 0 else 0.0
            # algorithmic HBM bytes of the same launches: A + B + C, + one more C-sized tensor for the GELU forms (the saved
            # pre-activation written / read); residual operands are not known here (lower bound)
            alg = sum(2.0 * r[4][3] * (r[4][0] * r[4][2] + r[4][1] * r[4][2] + r[4][0] * r[4][1] * (2 if r[4][4] in (1, 4) else 1))
                      for r in recs)
            traffic, traffic_src = pmc_traffic()
            out["roofline"] = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                               "traffic_unit": "bytes/launch (PMC: 2*FETCH_SIZE + WRITE_SIZE, separate rocprofv3 passes)",
                               "traffic_source": traffic_src,
                               "algorithmic_bytes_per_launch": round(alg / max(len(recs), 1)),
                               "kernel": "gemm_nt_ws_kernel: wave-specialised bf16 NT GEMM (focus_amd/csrc/gemm_mfma_ws.hip)",
                               "launches_per_step": len(recs) // max(a.steps, 1),
                               "avg_launch_us": round(1e3 * ms / max(len(recs), 1), 2),
                               "kernel_ms_per_step": round(ms / max(a.steps, 1), 3),
                               "uniform_nt_kernel": {"kernel": "gemm_nt_kernel 128x128 (gemm_mfma.hip) and gemm_nt_small_kernel "
                                                               "(gemm_mfma_small.hip): M = 256 motion-stream products and "
                                                               "other small shapes",
                                                     "launches_per_step": len(other) // max(a.steps, 1),
                                                     "ms_per_step": round(other_ms / max(a.steps, 1), 3)},
                               "weight_grad_kernel": {"kernel": "bf16 TN GEMM (gemm_mfma_tn_ws.hip, gemm_mfma_tn.hip)",
                                                      "achieved": round(tn_tf, 2),
                                                      "ms_per_step": round(tn_ms / max(a.steps, 1), 3)}}
        elif world > 1 and not a.no_roofline and not a.fp32:
            for _ in range(a.steps):       # keep ranks in lock-step with rank 0's instrumented pass
                train_step(model, optimizer, loss_fun, inputs, labels, meta, cfg)
            torch.cuda.synchronize()

        if rank == 0 and world == 1 and not a.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(model, cfg)
            except Exception as e:  # the baseline is a reported figure; never lose the bench line to it
                out["cpu_baseline"] = {"value": None, "unit": "clips/s", "cores": host_cores()[1], "kind": "port",
                                       "sample": "failed: %r" % (e,)}
        del model, optimizer, inputs
        ops.drop_caches()
        torch.cuda.empty_cache()

    if world == 1 and a.workload in ("steve", "all") and not a.fp32:
        rec = _guard(lambda: bench_steve(a, dev))
        if a.steve_model_batch > 0 and not a.steve_eager:
            import torch as _t
            _t.cuda.empty_cache()
            rec["model_step"] = _guard(lambda: bench_steve_model(a, dev))
        if out is None:
            out = rec
        else:
            out["steve"] = rec
    if world == 1 and a.workload in ("hr", "all") and not a.fp32:
        rec = _guard(lambda: bench_hr(a, dev))
        if out is None:
            out = rec
        else:
            out["hr"] = rec
    if rank == 0:
        print(json.dumps(out), flush=True)
    return 0


def _guard(fn):
    try:
        return fn()
    except Exception as e:          # a sub-record must never lose the headline line
        import traceback
        traceback.print_exc(file=sys.stderr)
        return {"failed": repr(e)}


def bench_steve_model(a, dev):
    """The whole STEVE training step (tools/steve_train_net.py:57-126 through focus_amd.train.slot_train_step) at the
    BASELINE frame shape, 24 x 128 x 128: dVAE, CNN encoder, the slot update above, and the 8-block transformer decoder over
    the 1024 image tokens of every frame (causal flash attention, csrc/flash_attn.hip) with the vocabulary-4096 cross
    entropy; decoder dropout 0.1 as in configs/movi_e/base.yaml.  Also the forward alone (STEVE.forward, eval-mode)."""
    import torch
    from focus_amd import ops
    from focus_amd.slowfast.config.defaults import get_cfg
    from focus_amd.slowfast.models import MODEL_REGISTRY
    from focus_amd.slowfast.models.optimizer import construct_optimizer_slot
    from focus_amd.train import slot_train_step
    cfg = get_cfg()
    cfg.MODEL.MODEL_NAME = "STEVE"
    cfg.NUM_GPUS = 1
    cfg.TRAIN.MIXED_PRECISION = True
    cfg.SOLVER.OPTIMIZING_METHOD = "adam"
    cfg.SOLVER.CLIP_GRAD_L2NORM = 0.05
    sl = cfg.SLOTS
    sl.NUM_ITERS, sl.NUM_SLOTS, sl.CNN_HID_SIZE, sl.SIZE, sl.DIM, sl.MLP_HID_SIZE, sl.IMG_SIZE, sl.VOCAB_SIZE = 3, 11, 64, 192, 192, 768, 128, 4096
    sl.NUM_PREDICTOR_BLOCKS, sl.NUM_PREDICTOR_HEADS, sl.PREDICTOR_DROPOUT = 1, 4, 0.0
    sl.DECODER.DIM, sl.DECODER.NUM_BLOCKS, sl.DECODER.NUM_HEADS, sl.DECODER.DROPOUT = 192, 8, 4, 0.1
    sl.GRAPH_SLOT_UPDATE = os.environ.get("FOCUS_BENCH_STEVE_GRAPH", "0") != "0"   # (=1: slot update from HIP graphs; no gain at batch 16)
    B, T = a.steve_model_batch, 24
    torch.manual_seed(0)
    m = MODEL_REGISTRY.get("STEVE")(cfg).to(dev)
    opt = construct_optimizer_slot(m, cfg)
    video = torch.rand(B, T, 3, 128, 128, device=dev)
    flash = []
    real = ops.flash_attention
    ops.flash_attention = lambda *x, **k: (flash.append(1), real(*x, **k))[1]
    try:
        m.train()
        step = 0
        for _ in range(2):
            loss = slot_train_step(m, opt, video, step, cfg)[0]
            step += 1
        torch.cuda.synchronize()
        calls = len(flash) // 2
        n = max(2, min(a.steps, 3))
        t0 = time.perf_counter()
        for _ in range(n):
            loss = slot_train_step(m, opt, video, step, cfg)[0]
            step += 1
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / n
        peak = torch.cuda.max_memory_allocated() / 2 ** 30
        fwd_ms = float("nan")
        if getattr(a, "steve_model_eval", True):
            m.eval()
            with torch.no_grad():
                m(video, 1.0, True)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n):
                    m(video, 1.0, True)
                torch.cuda.synchronize()
            fwd_ms = 1e3 * (time.perf_counter() - t0) / n
    finally:
        ops.flash_attention = real
    # the causal self-attention products that reach the matrix pipe: 3 of 4 16-channel steps of q.k, 2 d-blocks of P.v,
    # half of the 1024 x 1024 square; backward 2.5x the forward (dq, dkv recompute the logits)
    attn_flops = 8 * B * T * 4 * (1024 * 1024 / 2) * 2 * (48 + 48) * 3.5
    del m, opt
    ops.drop_caches()
    torch.cuda.empty_cache()
    return {"workload": "STEVE training step, movi_e 24x128x128, 11 slots, 3 iterations, decoder 8 blocks x 4 heads x 48, "
                        "vocabulary 4096, dropout 0.1, batch=%d, bf16 token path (configs/movi_e/base.yaml at IMG_SIZE 128)%s" % (
                            B, ", slot update replayed from HIP graphs (SLOTS.GRAPH_SLOT_UPDATE)" if sl.GRAPH_SLOT_UPDATE else ""),
            "ms_per_step": round(ms, 2), "clips_per_s": round(B / (ms * 1e-3), 2), "forward_ms": round(fwd_ms, 2),
            "steps": n, "final_loss": round(float(loss.detach()), 4), "peak_memory_GiB": round(peak, 1),
            "flash_attention_calls_per_step": calls,
            "decoder_attention": "csrc/flash_attn.hip: no [1024, 1024] probabilities or dropout masks in memory "
                                 "(the materialised form is 6.4 GB per block and direction at batch 32)",
            "decoder_attention_algorithmic_tflop_per_step": round(attn_flops / 1e12, 2)}


def bench_steve(a, dev):
    """BASELINE configs[2]: SlotAttentionVideo update at B=32, T=24, N=4096 (64x64 feature map of 128x128 frames),
    D=192, K=11, 3 iterations, fwd+bwd, bf16.  HBM-bound: algorithmic bytes from SURVEY 8(d) (6.11 GB forward;
    backward re-reads inputs/k/v and writes their gradients: x3 in total)."""
    import torch
    from focus_amd import ops
    from focus_amd.slowfast.models.STEVE.steve import SlotAttentionVideo
    B, T, N, D, K, IT = a.steve_batch, 24, 4096, 192, 11, 3
    torch.manual_seed(0)
    m = SlotAttentionVideo(IT, K, D, D, 4 * D, num_predictor_blocks=1, num_predictor_heads=4, dropout=0.0).to(dev)
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(B, T, N, D, device=dev, dtype=torch.bfloat16, generator=g).requires_grad_(True)
    noise = torch.randn(B, K, D, device=dev, generator=g)

    def reset_grads():
        x.grad = None
        for p in m.parameters():
            p.grad = None

    def fwd_bwd():
        slots, attn = m(x, noise)
        (slots.float().square().mean() + attn.float().mean()).backward()
        return slots, attn

    def step():
        reset_grads()
        return fwd_bwd()
    for _ in range(max(1, min(a.warmup, 2))):
        step()
    torch.cuda.synchronize()
    n = max(2, min(a.steps, 5))
    t0 = time.perf_counter()
    for _ in range(n):
        slots, attn = step()
    torch.cuda.synchronize()
    eager_ms = 1e3 * (time.perf_counter() - t0) / n
    ms, graphed = eager_ms, False
    if not a.steve_eager:
        # the step is launch bound (~2900 kernels): replay it from one HIP graph (focus_amd.train.GraphedStep)
        from focus_amd.train import GraphedStep
        ref_slots, ref_grad = slots.detach().clone(), x.grad.detach().clone()
        del slots, attn             # no output of an eager run (and with it that run's autograd graph) may be alive at capture
        reset_grads()               # captured with no .grad present: every replay assigns fresh gradients (no accumulation)
        gs = GraphedStep(fwd_bwd, reset=reset_grads)
        gs.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            slots, attn = gs.replay()
        torch.cuda.synchronize()
        ms, graphed = 1e3 * (time.perf_counter() - t0) / n, True
        assert torch.equal(slots, ref_slots) and torch.equal(x.grad, ref_grad), "graph replay differs from the eager step"
        # what the PRODUCT loop runs under SLOTS.GRAPH_SLOT_UPDATE (STEVE._slots): the module as a pair of captured graphs
        # (forward / backward) called from an eager autograd step, the loss and everything around it eager
        product_ms = None
        try:
            del slots, attn, gs
            reset_grads()
            gm = torch.cuda.make_graphed_callables(m, (x.detach().clone().requires_grad_(), noise), num_warmup_iters=2)

            def prod():
                reset_grads()
                s_, a_ = gm(x, noise)
                (s_.float().square().mean() + a_.float().mean()).backward()
                return s_, a_
            prod()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                slots, attn = prod()
            torch.cuda.synchronize()
            product_ms = 1e3 * (time.perf_counter() - t0) / n
            assert torch.equal(slots, ref_slots) and torch.equal(x.grad, ref_grad), "graphed callable differs from the eager step"
        except Exception as e:
            import traceback
            traceback.print_exc(file=sys.stderr)
            product_ms = "failed: %r" % (e,)
            slots, attn = step()
    fwd_bytes = (B * T * N * D * 2) * (1 + 2 + 2) + B * T * N * K * 2         # inputs, write k/v, read k/v, attn out
    alg = 3.0 * fwd_bytes
    ach = alg / (ms * 1e-3) / 1e9
    rowsum = float(attn.detach().float().sum(-1).sub(1).abs().max())
    ops.drop_caches()
    prec = pmc_record("steve")          # whole-step HBM-side bytes from the PMC passes of the SAME kernel sources, or None
    traffic = round(prec["trace_total_bytes"] / max(prec.get("executions", 4), 1)) if prec else None
    return {"workload": "STEVE slot-attention update, movi_e 24x128x128 (N=4096 tokens/frame), 11 slots, 3 iters, "
                        "batch=%d, fwd+bwd bf16 (BASELINE configs[2])" % B,
            "ms_per_step": round(ms, 3), "clips_per_s": round(B / (ms * 1e-3), 2),
            "slot_updates_per_s": round(B * T * IT * K / (ms * 1e-3), 1), "steps": n,
            "launch": "one HIP graph replay per step" if graphed else "eager", "eager_ms_per_step": round(eager_ms, 3),
            "product_loop_ms_per_step": (round(product_ms, 3) if isinstance(product_ms, float) else product_ms) if graphed else None,
            "product_loop": "SLOTS.GRAPH_SLOT_UPDATE: forward and backward graphs (torch.cuda.make_graphed_callables) inside the "
                            "eager training step (focus_amd/slowfast/models/STEVE/steve.py:_savi_graphed)",
            "roofline": {"bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": round(ach / PEAK_HBM_GBS, 4), "traffic": traffic,
                         "traffic_unit": "bytes/step (PMC: 2*FETCH_SIZE + WRITE_SIZE over every kernel of the step, separate "
                                         "rocprofv3 passes)" if traffic else None,
                         "traffic_source": ("%s @ build %s" % (prec["file"], prec.get("head"))) if prec else None,
                         "algorithmic_bytes_per_step": int(alg),
                         "note": "6.11 GB forward (SURVEY 8d) x3 for forward+backward"},
            "attn_rowsum_max_err": round(rowsum, 5), "finite": bool(torch.isfinite(slots.float()).all())}


def bench_hr(a, dev):
    """BASELINE configs[4] on one GPU: ORViT-Motionformer-HR 16x336, 6 objects, EK verb+noun heads, fp8 (OCP e4m3) Linear
    weights with bf16 activations (TRAIN.FP8_WEIGHTS; the weights are widened to bf16 between LDS and the MFMA, so the
    bf16 dense MFMA peak is the one that applies), fwd+bwd+clip+AdamW, at batch 4 per GPU and at a
    large batch; the same step with bf16 weights beside it."""
    import torch
    from focus_amd import ops
    from focus_amd.slowfast.models import build_model
    from focus_amd.slowfast.models.losses import get_loss_func
    from focus_amd.slowfast.models.optimizer import construct_optimizer
    from focus_amd.train import synthetic_batch, train_step

    def run(batch, fp8):
        cfg = make_cfg(1, batch, hr=True)
        cfg.merge_from_list(["TRAIN.FP8_WEIGHTS", bool(fp8)])
        torch.manual_seed(0)
        model = build_model(cfg, gpu_id=dev.index)
        model.train()
        opt = construct_optimizer(model, cfg)
        loss_fun = get_loss_func(cfg)(reduction="mean")
        inputs, labels, meta = synthetic_batch(cfg, batch, dev, seed=77)
        for _ in range(max(1, min(a.warmup, 2))):
            train_step(model, opt, loss_fun, inputs, labels, meta, cfg)
        torch.cuda.synchronize()
        n = max(2, min(a.steps, 5))
        t0 = time.perf_counter()
        for _ in range(n):
            _, loss = train_step(model, opt, loss_fun, inputs, labels, meta, cfg)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / n
        rec = {"batch": batch, "ms_per_step": round(ms, 3), "clips_per_s": round(batch / (ms * 1e-3), 3), "steps": n,
               "final_loss": round(float(loss.detach()), 4)}
        gemm = None
        if fp8 and not a.no_roofline:
            ops.GEMM_TIMING = []
            train_step(model, opt, loss_fun, inputs, labels, meta, cfg)
            torch.cuda.synchronize()
            recs, ops.GEMM_TIMING = ops.GEMM_TIMING, None
            r8 = [r for r in recs if r[3] == "nt_ws8"]
            fl, t8 = sum(r[0] for r in r8), sum(r[1].elapsed_time(r[2]) for r in r8)
            if t8 > 0:
                gemm = {"kernel": "gemm_nt_ws_kernel<..., BFP8>: e4m3 weights widened to bf16 on the LDS->register path",
                        "launches_per_step": len(r8), "achieved": round(fl / (t8 * 1e-3) / 1e12, 2), "unit": "TFLOP/s",
                        "ms_per_step": round(t8, 3)}
        del model, opt
        ops.drop_caches()
        torch.cuda.empty_cache()
        return rec, gemm

    Bh = a.hr_batch
    f8, gemm = run(Bh, True)
    out = {"workload": "ORViT-Motionformer-HR 16x336, 6 objects, EK heads (97+300), fp8 (OCP e4m3) Linear weights, bf16 "
                       "activations, batch=%d (BASELINE configs[4] on 1 GPU)" % Bh,
           "dtype": "fp8w", "ms_per_step": f8["ms_per_step"], "clips_per_s": f8["clips_per_s"], "steps": f8["steps"],
           "final_loss": f8["final_loss"]}
    frac = f8["clips_per_s"] * ALG_GF_PER_CLIP_HR * 1e9 / (PEAK_BF16_TFLOPS * 1e12)
    out["roofline"] = {"bound": "mfma", "achieved": round(f8["clips_per_s"] * ALG_GF_PER_CLIP_HR / 1e3, 2), "peak": PEAK_BF16_TFLOPS,
                       "unit": "TFLOP/s", "frac": round(frac, 5), "traffic": None,
                       "note": "model-level: clips/s x 5099 GF (BASELINE.md) vs the bf16 dense MFMA peak -- the e4m3 weights are "
                               "widened to bf16 before the MFMA (non-scaled fp8 MFMA runs at the bf16 rate; the 5 PF/s scaled form "
                               "needs fp8 activations too)", "fp8_gemm": gemm}
    if a.hr_large_batch and a.hr_large_batch != Bh:
        try:
            out["large_batch"] = run(a.hr_large_batch, True)[0]
        except Exception as e:          # e.g. out of memory: the batch-4 record stands
            out["large_batch"] = {"batch": a.hr_large_batch, "failed": repr(e)[:200]}
            ops.drop_caches()
            torch.cuda.empty_cache()
    bf, _ = run(Bh, False)
    out["bf16_weights"] = bf
    out["model_mfma_frac_algorithmic"] = round(frac, 5)
    return out


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def main(argv=None, job=None):
    """`job(cfg)` is what every rank runs (bench_job; tests pass a CPU stand-in to exercise the launch path)."""
    job = job or bench_job
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8, help="clips per GPU")
    ap.add_argument("--workload", default="all", choices=["all", "orvit", "steve", "hr"],
                    help="all = the headline ORViT-MF line with the steve / hr sub-records (N == 1)")
    ap.add_argument("--steve-batch", type=int, default=32)
    ap.add_argument("--hr-batch", type=int, default=4)
    ap.add_argument("--hr-large-batch", type=int, default=16, help="second HR fp8 record at this batch (0: skip)")
    ap.add_argument("--fp32", action="store_true", help="TRAIN.MIXED_PRECISION False (precision path)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gemm-shapes", action="store_true", help="print per-shape GEMM timings to stderr")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--steve-eager", action="store_true", help="time the STEVE sub-record eagerly (no HIP graph)")
    ap.add_argument("--steve-model-batch", type=int, default=16,
                    help="clips of the whole-model STEVE training-step record at 24x128x128 (0: skip)")
    ap.add_argument("--backend", default="nccl", help="process-group backend (nccl = RCCL; gloo for a rehearsal)")
    ap.add_argument("--bf16-grads", action="store_true", help="cfg.DDP_BF16_GRADS: bf16 gradient buckets on the wire")
    ap.add_argument("--same-device", action="store_true",
                    help="rehearsal only: every rank uses cuda:0 (needs --backend gloo; RCCL wants one GPU per rank)")
    args = ap.parse_args(argv)
    if args.same_device and args.backend == "nccl":
        raise SystemExit("--same-device needs --backend gloo")

    env_world = os.environ.get("WORLD_SIZE")
    n = int(env_world) if env_world is not None else max(args.gpus, 1)
    if env_world is not None and args.gpus > 1 and n != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, n))

    from focus_amd.slowfast.config.defaults import CfgNode
    cfg = make_cfg(n, args.batch, mixed=not args.fp32)
    cfg.DIST_BACKEND = args.backend
    cfg.DDP_BF16_GRADS = bool(args.bf16_grads)
    cfg.BENCH = CfgNode(vars(args))
    if args.same_device:
        os.environ["FOCUS_SAME_DEVICE"] = "1"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    if env_world is not None and n > 1:
        # launched by torch.distributed.run: this process is one rank already
        from focus_amd.slowfast.utils import multiprocessing as mpu
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        # one machine: shard 0 of 1, local rank == rank (env:// rendezvous)
        mpu.run(local_rank, n, job, "env://", 0, 1, args.backend, cfg)
        return 0
    if n > 1:
        # self-launch through the reference's launcher surface; the parent has not touched the GPU
        from focus_amd.slowfast.utils.misc import launch_job
        launch_job(cfg, "tcp://127.0.0.1:%d" % _free_port(), job)
        return 0
    if job is bench_job:
        import torch
        torch.cuda.set_device(0)
    return job(cfg)


if __name__ == "__main__":
    sys.exit(main())
