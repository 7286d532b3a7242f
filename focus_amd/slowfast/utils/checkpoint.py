"""Checkpoint I/O in the reference's file format (mirror of slowfast/utils/checkpoint.py:21-159, 201-394, 499-597;
SURVEY.md section 8(f) rank 2): a `.pyth` file is torch.save({'epoch', 'model_state', 'optimizer_state', 'cfg'[,
'scaler_state']}); loading applies, in the reference's order, clear-name patterns, the qkv split, replace-name
patterns, the copy of backbone qkv weights into ORViT layers, and then keeps exactly the entries whose name AND shape
match the model.  The hot path has no BatchNorm, so the Sub-BN conversions of the reference are identities here.

Deliberate differences:
  * files are read with torch.load(weights_only=True) (tensors / containers / strings only: nothing in a checkpoint of
    this format needs unpickling arbitrary objects);
  * caffe2 checkpoints and 2D->3D weight inflation (checkpoint.py:165-198, 244-315) belong to the CNN families that are
    out of scope: requesting them raises NotImplementedError;
  * after a load the bf16 weight shadows are invalidated (focus_amd.ops.invalidate_shadows) when the HIP library is there.
"""
import os
from collections import OrderedDict

import torch

from . import distributed as du


def make_checkpoint_dir(path_to_job, ex_name="test"):
    """checkpoint.py:21-36."""
    checkpoint_dir = os.path.join(path_to_job, ex_name)
    if du.is_master_proc() and not os.path.exists(checkpoint_dir):
        try:
            os.makedirs(checkpoint_dir)
        except Exception:
            pass
    return checkpoint_dir


def get_checkpoint_dir(path_to_job):
    """checkpoint.py:39-45."""
    return os.path.join(path_to_job, "checkpoints")


def get_path_to_checkpoint(path_to_job, epoch, name="ckp_ep", fmt=".pyth"):
    """checkpoint.py:48-59 (the reference's current naming ignores the epoch: `{name}{fmt}`)."""
    return os.path.join(get_checkpoint_dir(path_to_job), "%s%s" % (name, fmt))


def get_last_checkpoint(path_to_job):
    """checkpoint.py:61-74."""
    d = get_checkpoint_dir(path_to_job)
    names = os.listdir(d) if os.path.exists(d) else []
    names = [f for f in names if "checkpoint" in f]
    assert len(names), "No checkpoints found in '{}'.".format(d)
    return os.path.join(d, sorted(names)[-1])


def has_checkpoint(path_to_job):
    """checkpoint.py:76-84."""
    d = get_checkpoint_dir(path_to_job)
    files = os.listdir(d) if os.path.exists(d) else []
    return any("checkpoint" in f for f in files)


def is_checkpoint_epoch(cfg, cur_epoch, multigrid_schedule=None):
    """checkpoint.py:87-109."""
    if cfg.TRAIN.VAL_ONLY:
        return False
    if cur_epoch + 1 == cfg.SOLVER.MAX_EPOCH:
        return True
    if multigrid_schedule is not None:
        prev_epoch = 0
        for s in multigrid_schedule:
            if cur_epoch < s[-1]:
                period = max((s[-1] - prev_epoch) // cfg.MULTIGRID.EVAL_FREQ + 1, 1)
                return (s[-1] - 1 - cur_epoch) % period == 0
            prev_epoch = s[-1]
    return (cur_epoch + 1) % cfg.TRAIN.CHECKPOINT_PERIOD == 0


def sub_to_normal_bn(sd):
    """checkpoint.py:397-427: no Sub-BN layers on this path."""
    return sd


def normal_to_sub_bn(checkpoint_sd, model_sd):
    """checkpoint.py:449-496: no Sub-BN layers on this path."""
    return checkpoint_sd


def save_checkpoint(path_to_job, model, optimizer, epoch, cfg, ckp_name="test", name="ckp_ep", fmt=".pyth", scaler=None):
    """checkpoint.py:112-159: master process only; returns the path written."""
    if not du.is_master_proc(cfg.NUM_GPUS * cfg.NUM_SHARDS):
        return None
    os.makedirs(get_checkpoint_dir(path_to_job), exist_ok=True)
    sd = model.module.state_dict() if cfg.NUM_GPUS > 1 else model.state_dict()
    checkpoint = {"epoch": epoch, "model_state": sub_to_normal_bn(sd), "optimizer_state": optimizer.state_dict(),
                  "cfg": cfg.dump()}
    if scaler is not None:
        checkpoint["scaler_state"] = scaler.state_dict()
    path_to_checkpoint = get_path_to_checkpoint(path_to_job, epoch + 1, name, fmt)
    with open(path_to_checkpoint, "wb") as f:
        torch.save(checkpoint, f)
    return path_to_checkpoint


def split_qkv(d):
    """checkpoint.py:586-597: every '...qkv...' entry becomes three entries '...q...', '...k...', '...v...'."""
    out = OrderedDict()
    for k, v in d.items():
        if "qkv" in k:
            for a, new_v in zip(["q", "k", "v"], v.chunk(3, dim=0)):
                out[k.replace("qkv", a)] = new_v
        else:
            out[k] = v
    return out


def load_checkpoint(path_to_checkpoint, model, data_parallel=True, optimizer=None, scaler=None, inflation=False,
                    convert_from_caffe2=False, epoch_reset=False, clear_name_pattern=(), replace_name_pattern=(),
                    load_orvit_attn_from_bb=False, should_split_qkv=False):
    """checkpoint.py:201-394 for the pytorch format.  Returns the checkpoint's epoch (or -1)."""
    assert os.path.exists(path_to_checkpoint), "Checkpoint '{}' not found".format(path_to_checkpoint)
    if convert_from_caffe2:
        raise NotImplementedError("caffe2 checkpoints (checkpoint.py:244-315) belong to the out-of-scope CNN families")
    if inflation:
        raise NotImplementedError("2D -> 3D weight inflation (checkpoint.py:165-198) belongs to the out-of-scope CNN families")
    ms = model.module if data_parallel else model
    with open(path_to_checkpoint, "rb") as f:
        checkpoint = torch.load(f, map_location="cpu", weights_only=True)
    model_sd = ms.state_dict()
    checkpoint["model_state"] = normal_to_sub_bn(checkpoint["model_state"], model_sd)
    if clear_name_pattern:                                              # :339-353
        for item in clear_name_pattern:
            renamed = OrderedDict()
            for k in checkpoint["model_state"]:
                renamed[k.replace(item, "") if item in k else k] = checkpoint["model_state"][k]
            checkpoint["model_state"] = renamed
    pre_train_dict = checkpoint["model_state"]
    load_orvit_attn = epoch_reset and load_orvit_attn_from_bb           # :357-358
    should_split_qkv = epoch_reset and should_split_qkv
    if should_split_qkv:
        pre_train_dict = split_qkv(pre_train_dict)
    if len(replace_name_pattern) > 0:                                   # :361-367
        renamed = {}
        for k, v in pre_train_dict.items():
            for a, b in replace_name_pattern:
                if a in k:
                    k = k.replace(a, b)
            renamed[k] = v
        pre_train_dict = renamed
    if load_orvit_attn:                                                 # :368-375
        qkv_keys = [k for k in checkpoint["model_state"].keys() if k.startswith("blocks") and "qkv" in k]
        names = model_sd.keys()
        for k in qkv_keys:
            for kk in [k, "orvit_%s" % k]:
                if kk in names:
                    pre_train_dict[kk] = checkpoint["model_state"][k]
    # keep the entries whose name and shape match the model (:377-392)
    match = {k: v for k, v in pre_train_dict.items() if k in model_sd and v.size() == model_sd[k].size()}
    ms.load_state_dict(match, strict=False)
    _invalidate_shadows()
    epoch = -1
    if "epoch" in checkpoint.keys() and not epoch_reset:                # :396-403
        epoch = checkpoint["epoch"]
        if optimizer:
            optimizer.load_state_dict(checkpoint["optimizer_state"])
        if scaler:
            scaler.load_state_dict(checkpoint["scaler_state"])
    return epoch


def load_test_checkpoint(cfg, model):
    """checkpoint.py:499-545."""
    if cfg.TEST.TEST_EPOCH_NUM > 0:
        n = cfg.TEST.TEST_EPOCH_NUM
        cfg.TEST.CHECKPOINT_FILE_PATH = os.path.join(cfg.OUTPUT_DIR, "checkpoints", "ckp_ep_%05d.pyth" % n)
    split = getattr(cfg, "SPLIT_QKV_CHECKPOINT", False)
    if cfg.TEST.CHECKPOINT_FILE_PATH != "":
        load_checkpoint(cfg.TEST.CHECKPOINT_FILE_PATH, model, cfg.NUM_GPUS > 1, None, inflation=False,
                        convert_from_caffe2=cfg.TEST.CHECKPOINT_TYPE == "caffe2", should_split_qkv=split)
    elif has_checkpoint(cfg.OUTPUT_DIR):
        load_checkpoint(get_last_checkpoint(cfg.OUTPUT_DIR), model, cfg.NUM_GPUS > 1)
    elif cfg.TRAIN.CHECKPOINT_FILE_PATH != "":
        load_checkpoint(cfg.TRAIN.CHECKPOINT_FILE_PATH, model, cfg.NUM_GPUS > 1, None, inflation=False,
                        convert_from_caffe2=cfg.TRAIN.CHECKPOINT_TYPE == "caffe2",
                        load_orvit_attn_from_bb=cfg.ORVIT.ENABLE and cfg.ORVIT.LOAD_ORVIT_ATTN_LAYERS_FROM_BB,
                        should_split_qkv=split)
    # else: random initialisation, "only for debugging" (:541-544)


def load_train_checkpoint(cfg, model, optimizer, scaler=None):
    """checkpoint.py:548-584: auto-resume from the job directory, else TRAIN.CHECKPOINT_FILE_PATH, else epoch 0."""
    if cfg.TRAIN.AUTO_RESUME and has_checkpoint(cfg.OUTPUT_DIR):
        if cfg.TRAIN.VAL_ONLY and cfg.TEST.TEST_EPOCH_NUM > 0:
            last_checkpoint = os.path.join(cfg.OUTPUT_DIR, "checkpoints", "checkpoint_epoch_%05d.pyth" % cfg.TEST.TEST_EPOCH_NUM)
        else:
            last_checkpoint = get_last_checkpoint(cfg.OUTPUT_DIR)
        return load_checkpoint(last_checkpoint, model, cfg.NUM_GPUS > 1, optimizer, scaler=scaler) + 1
    if cfg.TRAIN.CHECKPOINT_FILE_PATH not in (None, ""):
        # (the reference tests `!= None` only and would try to open ""; an empty path means "no checkpoint" here)
        return load_checkpoint(cfg.TRAIN.CHECKPOINT_FILE_PATH, model, cfg.NUM_GPUS > 1, optimizer, scaler=scaler,
                               inflation=cfg.TRAIN.CHECKPOINT_INFLATE,
                               convert_from_caffe2=cfg.TRAIN.CHECKPOINT_TYPE == "caffe2",
                               epoch_reset=cfg.TRAIN.CHECKPOINT_EPOCH_RESET,
                               clear_name_pattern=cfg.TRAIN.CHECKPOINT_CLEAR_NAME_PATTERN,
                               replace_name_pattern=cfg.TRAIN.CHECKPOINT_REPLACE_NAME_PATTERN,
                               should_split_qkv=getattr(cfg, "SPLIT_QKV_CHECKPOINT", False)) + 1
    return 0


def _invalidate_shadows():
    try:
        from focus_amd import ops
        ops.invalidate_shadows()
    except Exception:          # no HIP library in this process (CPU-side tooling): there are no shadows either
        pass
