"""torch.distributed helpers the train step uses (mirror of the hot subset of slowfast/utils/distributed.py)."""
import torch
import torch.distributed as dist

_LOCAL_PROCESS_GROUP = None


def get_world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def get_rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def is_master_proc(num_gpus=8):
    return get_rank() % num_gpus == 0 if dist.is_initialized() else True


def is_root_proc():
    return get_rank() == 0 if dist.is_initialized() else True


def all_gather(tensors):
    """distributed.py:15-34: gather along dim 0 from every rank."""
    world = get_world_size()
    if world == 1:
        return list(tensors)
    out = []
    for t in tensors:
        parts = [torch.ones_like(t) for _ in range(world)]
        dist.all_gather(parts, t, async_op=False)
        out.append(torch.cat(parts, dim=0))
    return out


def all_reduce(tensors, average=True):
    """distributed.py:37-53 issues one blocking all-reduce per scalar; here the scalars are packed into one
    buffer and reduced with a single collective."""
    if get_world_size() == 1:
        return tensors
    flat = torch.stack([t.detach().float().reshape(()) for t in tensors])
    dist.all_reduce(flat)
    if average:
        flat = flat / get_world_size()
    return [flat[i].to(t.dtype) for i, t in enumerate(tensors)]


def init_distributed_training(cfg):
    """distributed.py:268-285: one process group per machine (the ranks that share xGMI)."""
    global _LOCAL_PROCESS_GROUP
    if cfg.NUM_GPUS <= 1 or not (dist.is_available() and dist.is_initialized()):
        return
    per = cfg.NUM_GPUS
    for i in range(dist.get_world_size() // per):
        pg = dist.new_group(list(range(i * per, (i + 1) * per)))
        if i == cfg.SHARD_ID:
            _LOCAL_PROCESS_GROUP = pg


def get_local_size():
    if not (dist.is_available() and dist.is_initialized()):
        return 1
    return dist.get_world_size(group=_LOCAL_PROCESS_GROUP)


def get_local_rank():
    if not (dist.is_available() and dist.is_initialized()):
        return 0
    assert _LOCAL_PROCESS_GROUP is not None
    return dist.get_rank(group=_LOCAL_PROCESS_GROUP)


def synchronize():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
