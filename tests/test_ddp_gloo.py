"""The N>1 path on CPU: two gloo ranks.  The hot path needs a GPU, so the model here is a stand-in with ordinary
torch ops; what is exercised is OUR data-parallel plumbing -- wrap_ddp (bucketed all-reduce, gradients as bucket
views), the packed metric all-reduce of slowfast.utils.distributed, per-rank sharding of the synthetic batch, and
the max-over-ranks timing reduction bench.py uses."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from focus_amd.parallel import wrap_ddp
    from focus_amd.slowfast.utils import distributed as du
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.GELU(), torch.nn.Linear(32, 4))
    ddp = wrap_ddp(model, None)
    g = torch.Generator().manual_seed(100 + rank)          # each rank sees its own shard of the global batch
    x, y = torch.randn(8, 16, generator=g), torch.randint(0, 4, (8,), generator=g)
    loss = torch.nn.functional.cross_entropy(ddp(x), y)
    loss.backward()
    grads = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    # reference: average of the per-rank gradients computed without DDP
    ref_model = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.GELU(), torch.nn.Linear(32, 4))
    ref_model.load_state_dict(model.state_dict())
    acc = None
    for r in range(world):
        gr = torch.Generator().manual_seed(100 + r)
        xr, yr = torch.randn(8, 16, generator=gr), torch.randint(0, 4, (8,), generator=gr)
        ref_model.zero_grad()
        torch.nn.functional.cross_entropy(ref_model(xr), yr).backward()
        v = torch.cat([p.grad.reshape(-1) for p in ref_model.parameters()])
        acc = v if acc is None else acc + v
    ok_grad = torch.allclose(grads, acc / world, atol=1e-6)
    l2, = du.all_reduce([loss.detach()])
    losses = [torch.zeros(()) for _ in range(world)]
    dist.all_gather(losses, loss.detach())
    ok_metric = torch.allclose(l2, torch.stack(losses).mean(), atol=1e-6)
    t = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok_time = abs(float(t) - 0.1 * world) < 1e-12
    assert du.get_world_size() == world and du.get_rank() == rank and du.is_master_proc() == (rank == 0)
    if rank == 0:
        with open(out, "w") as f:
            f.write("%d %d %d" % (ok_grad, ok_metric, ok_time))
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_two_ranks_gloo(tmp_path):
    out = str(tmp_path / "res.txt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    assert open(out).read() == "1 1 1"


# ---- the launcher: bench.py --gpus N -> misc.launch_job -> torch.multiprocessing.spawn(multiprocessing.run) ----------
def standin_job(cfg):
    """What every spawned rank runs in the launch-path tests: a CPU stand-in for bench_job (the hot path needs a GPU)."""
    from focus_amd.parallel import wrap_ddp
    from focus_amd.slowfast.utils import distributed as du
    du.init_distributed_training(cfg)
    world, rank = du.get_world_size(), du.get_rank()
    assert world == cfg.NUM_GPUS and du.get_local_size() == world and du.get_local_rank() == rank
    if cfg.BENCH.get("fail_rank", -1) == rank:
        raise RuntimeError("injected failure on rank %d" % rank)
    torch.manual_seed(0)
    model = wrap_ddp(torch.nn.Linear(8, 4), None, cfg)
    g = torch.Generator().manual_seed(rank)
    model(torch.randn(4, 8, generator=g)).sum().backward()
    gsum = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).sum()
    every = du.all_gather([gsum.reshape(1)])[0]
    assert torch.allclose(every, every[0].expand_as(every))         # gradients were averaged: identical on every rank
    if rank == 0:
        with open(cfg.BENCH.out, "w") as f:
            f.write("world=%d backend=%s steps=%d" % (world, cfg.DIST_BACKEND, cfg.BENCH.steps))
    return 0


def _bench_main(argv, extra):
    sys.path.insert(0, ROOT)
    import bench
    from focus_amd.slowfast.config.defaults import CfgNode
    orig = bench.make_cfg

    def make_cfg(*a, **k):
        cfg = orig(*a, **k)
        cfg.EXTRA = CfgNode(extra)
        return cfg
    bench.make_cfg = make_cfg
    try:
        return bench.main(argv, job=_standin_with_extra)
    finally:
        bench.make_cfg = orig


def _standin_with_extra(cfg):
    for k, v in cfg.EXTRA.items():
        cfg.BENCH[k] = v
    return standin_job(cfg)


def test_bench_self_launch_two_ranks_gloo(tmp_path):
    """`python bench.py --gpus 2 --backend gloo` launches its own ranks (no torchrun) through launch_job."""
    out = str(tmp_path / "launch.txt")
    rc = _bench_main(["--gpus", "2", "--backend", "gloo", "--steps", "3"], {"out": out})
    assert rc == 0
    assert open(out).read() == "world=2 backend=gloo steps=3"


def test_bench_self_launch_child_failure_propagates(tmp_path):
    import pytest
    with pytest.raises(Exception, match="injected failure|terminated|exit"):
        _bench_main(["--gpus", "2", "--backend", "gloo"], {"out": str(tmp_path / "x.txt"), "fail_rank": 1})


def test_bench_under_external_launcher_env(tmp_path):
    """The driver's form: RANK / WORLD_SIZE / MASTER_* in the environment (torch.distributed.run), one process = one rank."""
    import subprocess
    port = _free_port()
    out = str(tmp_path / "env.txt")
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r); import test_ddp_gloo as t; "
            "sys.exit(t._bench_main(['--gpus','2','--backend','gloo','--steps','4'], {'out': %r}))"
            % (ROOT, os.path.join(ROOT, "tests"), out))
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", code], env=env))
    assert [p.wait(timeout=300) for p in procs] == [0, 0]
    assert open(out).read() == "world=2 backend=gloo steps=4"
