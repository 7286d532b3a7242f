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
