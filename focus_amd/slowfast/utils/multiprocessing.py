"""One process per GPU (mirror of slowfast/utils/multiprocessing.py:9-67).

`run` is the entry point of every child spawned by misc.launch_job: it joins the process group
(backend "nccl" is RCCL over xGMI on ROCm; "gloo" for CPU-side rehearsals), binds the process to its GPU and calls
`func(cfg)`.  Nothing in the PARENT may have touched the GPU before the spawn (children are fresh interpreters:
torch.multiprocessing.spawn uses the "spawn" start method, never a fork or an exec of a GPU-initialised process)."""
import os

import torch


def run(local_rank, num_proc, func, init_method, shard_id, num_shards, backend, cfg, output_queue=None):
    """multiprocessing.py:9-67: rank = shard_id * num_proc + local_rank, world = num_proc * num_shards."""
    world_size = num_proc * num_shards
    rank = shard_id * num_proc + local_rank
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL's intra-node transport needs it
    torch.distributed.init_process_group(backend=backend, init_method=init_method, world_size=world_size, rank=rank)
    if torch.cuda.is_available():
        # FOCUS_SAME_DEVICE=1 (rehearsal on a 1-GPU box with backend gloo): every rank shares cuda:0.  RCCL itself
        # needs one GPU per rank, so this is refused for the nccl backend.
        if os.environ.get("FOCUS_SAME_DEVICE", "0") == "1":
            if backend == "nccl":
                raise RuntimeError("FOCUS_SAME_DEVICE=1 needs DIST_BACKEND gloo (RCCL wants one GPU per rank)")
            torch.cuda.set_device(0)
        else:
            try:
                torch.cuda.set_device(local_rank)
            except Exception:
                print("LOCAL RANK: %d, HIP_VISIBLE_DEVICES: %s" % (local_rank, os.environ.get("HIP_VISIBLE_DEVICES")))
                raise
    ret = func(cfg)
    if output_queue is not None and local_rank == 0:
        output_queue.put(ret)
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
