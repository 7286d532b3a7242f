"""torch.distributed helpers the train step uses (mirror of the hot subset of slowfast/utils/distributed.py)."""
import torch
import torch.distributed as dist


def get_world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def get_rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def is_master_proc(num_gpus=8):
    return get_rank() % num_gpus == 0 if dist.is_initialized() else True


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
    pass
