"""Checkpoint I/O in the reference's format (focus_amd/slowfast/utils/checkpoint.py mirrors
slowfast/utils/checkpoint.py:112-159, 201-394, 548-597).  CPU only: module construction and state dicts need no GPU."""
import os

import pytest
import torch

from focus_amd.slowfast.config.defaults import get_cfg
from focus_amd.slowfast.utils import checkpoint as cu


def small_cfg(tmp):
    cfg = get_cfg()
    cfg.merge_from_list(["ORVIT.ENABLE", True, "ORVIT.O", 3, "ORVIT.LAYERS", [1], "DATA.TRAIN_CROP_SIZE", 64,
                         "DATA.NUM_FRAMES", 4, "MF.EMBED_DIM", 64, "MF.DEPTH", 3, "MF.NUM_HEADS", 4,
                         "MF.TEMPORAL_RESOLUTION", 2, "MF.USE_MLP", True, "MODEL.NUM_CLASSES", 10,
                         "MODEL.MODEL_NAME", "Motionformer", "TRAIN.DATASET", "Ssv2", "NUM_GPUS", 0])
    cfg.OUTPUT_DIR = str(tmp)
    return cfg


def build(cfg, seed):
    from focus_amd.slowfast.models import MODEL_REGISTRY
    torch.manual_seed(seed)
    m = MODEL_REGISTRY.get(cfg.MODEL.MODEL_NAME)(cfg)
    with torch.no_grad():
        for p in m.parameters():
            p.normal_(0, 0.1)
    return m


def test_save_then_resume_restores_model_optimizer_and_epoch(tmp_path):
    cfg = small_cfg(tmp_path)
    m = build(cfg, 0)
    opt = torch.optim.SGD(m.parameters(), lr=0.1, momentum=0.9)
    for p in m.parameters():
        p.grad = torch.ones_like(p)
    opt.step()
    path = cu.save_checkpoint(cfg.OUTPUT_DIR, m, opt, 4, cfg, name="checkpoint_epoch_00005")
    assert path.endswith(os.path.join("checkpoints", "checkpoint_epoch_00005.pyth")) and cu.has_checkpoint(cfg.OUTPUT_DIR)
    ck = torch.load(path, weights_only=True)
    assert set(ck.keys()) == {"epoch", "model_state", "optimizer_state", "cfg"} and ck["epoch"] == 4
    assert isinstance(ck["cfg"], str) and "MF" in ck["cfg"]
    m2 = build(cfg, 1)
    opt2 = torch.optim.SGD(m2.parameters(), lr=0.1, momentum=0.9)
    start = cu.load_train_checkpoint(cfg, m2, opt2)                 # AUTO_RESUME finds the file in OUTPUT_DIR
    assert start == 5
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    s1, s2 = opt.state_dict()["state"], opt2.state_dict()["state"]
    assert all(torch.equal(s1[i]["momentum_buffer"], s2[i]["momentum_buffer"]) for i in s1)


def test_fine_tune_load_keeps_name_and_shape_matches_only(tmp_path):
    """epoch_reset (fine-tuning): entries whose shape differs (a 174-way head into a 10-way model) or whose name is
    unknown are skipped, everything else is loaded, the epoch restarts at 0 and the optimizer state is not touched."""
    cfg = small_cfg(tmp_path)
    src = build(cfg, 0)
    sd = {("module_prefix." + k): v for k, v in src.state_dict().items()}
    sd["module_prefix.head.weight"] = torch.zeros(174, 64)
    sd["module_prefix.head.bias"] = torch.zeros(174)
    sd["module_prefix.not_in_the_model"] = torch.zeros(3)
    path = os.path.join(str(tmp_path), "pretrained.pyth")
    torch.save({"epoch": 30, "model_state": sd, "optimizer_state": {}, "cfg": ""}, path)
    dst = build(cfg, 1)
    head_before = dst.head.weight.detach().clone()
    ep = cu.load_checkpoint(path, dst, data_parallel=False, epoch_reset=True, clear_name_pattern=("module_prefix.",))
    assert ep == -1
    assert torch.equal(dst.head.weight, head_before)               # shape mismatch: left alone
    for k, v in src.state_dict().items():
        if not k.startswith("head."):
            assert torch.equal(dst.state_dict()[k], v), k


def test_qkv_split_replace_patterns_and_orvit_attention_copy(tmp_path):
    """checkpoint.py:357-375: with epoch_reset, SPLIT_QKV renames '...qkv...' into q / k / v thirds, replace patterns
    rename keys, and LOAD_ORVIT_ATTN_LAYERS_FROM_BB copies the backbone's block qkv weights under their own name and under
    'orvit_<name>' when the model has such entries."""
    d = {"blocks.0.attn.qkv.weight": torch.arange(12.0).reshape(6, 2), "blocks.0.norm1.weight": torch.ones(2)}
    out = cu.split_qkv(d)
    assert list(out.keys()) == ["blocks.0.attn.q.weight", "blocks.0.attn.k.weight", "blocks.0.attn.v.weight",
                                "blocks.0.norm1.weight"]
    assert torch.equal(out["blocks.0.attn.k.weight"], d["blocks.0.attn.qkv.weight"][2:4])

    class Toy(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.blocks = torch.nn.ModuleList([torch.nn.ModuleDict({"attn": torch.nn.ModuleDict({"qkv": torch.nn.Linear(2, 6)})})])
            self.renamed = torch.nn.Linear(2, 2)
    toy = Toy()
    sd = {"blocks.0.attn.qkv.weight": torch.full((6, 2), 7.0), "blocks.0.attn.qkv.bias": torch.full((6,), 3.0),
          "old_name.weight": torch.full((2, 2), 5.0), "old_name.bias": torch.full((2,), 4.0)}
    path = os.path.join(str(tmp_path), "bb.pyth")
    torch.save({"epoch": 1, "model_state": sd, "optimizer_state": {}, "cfg": ""}, path)
    ep = cu.load_checkpoint(path, toy, data_parallel=False, epoch_reset=True, replace_name_pattern=[("old_name", "renamed")],
                            load_orvit_attn_from_bb=True)
    assert ep == -1
    assert float(toy.blocks[0]["attn"]["qkv"].weight.detach().min()) == 7.0 and float(toy.renamed.bias.detach().max()) == 4.0
    # without epoch_reset neither the split nor the ORViT copy happens and the epoch / optimizer state are restored
    toy2 = Toy()
    opt = torch.optim.SGD(toy2.parameters(), lr=0.1)
    torch.save({"epoch": 9, "model_state": sd, "optimizer_state": opt.state_dict(), "cfg": ""}, path)
    assert cu.load_checkpoint(path, toy2, data_parallel=False, optimizer=opt, should_split_qkv=True) == 9
    assert float(toy2.blocks[0]["attn"]["qkv"].weight.detach().min()) == 7.0


def test_reference_formats_that_are_out_of_scope_fail_loudly(tmp_path):
    cfg = small_cfg(tmp_path)
    m = build(cfg, 0)
    path = os.path.join(str(tmp_path), "x.pyth")
    torch.save({"epoch": 0, "model_state": m.state_dict(), "optimizer_state": {}, "cfg": ""}, path)
    with pytest.raises(NotImplementedError):
        cu.load_checkpoint(path, m, data_parallel=False, convert_from_caffe2=True)
    with pytest.raises(NotImplementedError):
        cu.load_checkpoint(path, m, data_parallel=False, inflation=True)
    with pytest.raises(AssertionError):
        cu.load_checkpoint(os.path.join(str(tmp_path), "missing.pyth"), m, data_parallel=False)


def test_checkpoint_epoch_schedule():
    cfg = get_cfg()
    cfg.TRAIN.CHECKPOINT_PERIOD, cfg.SOLVER.MAX_EPOCH = 5, 12
    assert [e for e in range(12) if cu.is_checkpoint_epoch(cfg, e)] == [4, 9, 11]
    cfg.TRAIN.VAL_ONLY = True
    assert not cu.is_checkpoint_epoch(cfg, 11)


# ---- files WRITTEN by the reference's own checkpoint.py (oracle/make_golden.py main_ckpt) ------------------------------
def _golden(name):
    from conftest import GOLDEN
    return os.path.join(GOLDEN, name)


def _expected():
    import numpy as np
    return np.load(_golden("ckpt_small_expected.npz"), allow_pickle=False)


def test_reference_written_checkpoint_resumes_like_the_reference(tmp_path):
    """ckpt_small.pyth was written by the reference's save_checkpoint (checkpoint.py:112-159); loading it here must leave
    the model, the optimizer state and the epoch exactly as the reference's load_checkpoint (:201-394) left its own."""
    import numpy as np
    e = _expected()
    ck = torch.load(_golden("ckpt_small.pyth"), map_location="cpu", weights_only=True)     # nothing to unpickle but data
    assert set(ck.keys()) == {"epoch", "model_state", "optimizer_state", "cfg"} and ck["epoch"] == 3
    cfg = small_cfg(tmp_path)
    m = build(cfg, 5)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
    epoch = cu.load_checkpoint(_golden("ckpt_small.pyth"), m, data_parallel=False, optimizer=opt)
    assert epoch == int(e["resume_epoch"]) == 3
    sd = m.state_dict()
    names = [k[len("resume."):] for k in e.files if k.startswith("resume.")]
    assert sorted(names) == sorted(sd.keys())
    for k in names:
        assert np.array_equal(sd[k].numpy(), e["resume." + k]), k
    st = opt.state_dict()["state"][0]
    assert float(st["step"]) == float(e["resume_opt_step"]) and np.array_equal(st["exp_avg"].numpy(), e["resume_opt_exp_avg0"])
    # and our writer produces a file the same reader accepts, with the same keys
    path = cu.save_checkpoint(str(tmp_path), m, opt, 3, cfg, name="again")
    ck2 = torch.load(path, weights_only=True)
    assert set(ck2.keys()) == set(ck.keys()) and list(ck2["model_state"].keys()) == list(ck["model_state"].keys())


def test_reference_fine_tune_switches_match(tmp_path):
    """epoch_reset + clear_name_pattern + replace_name_pattern + load_orvit_attn_from_bb on a `module.`-prefixed file with a
    174-way head and an unknown entry: the loaded state equals what the reference's loader produced, entry by entry, and the
    same entries stay untouched; split_qkv (:586-597) cuts the fused projection the same way."""
    import numpy as np
    from collections import OrderedDict
    e = _expected()
    cfg = small_cfg(tmp_path)
    torch.manual_seed(2)
    m = build(cfg, 7)
    before = {k: v.clone() for k, v in m.state_dict().items()}
    epoch = cu.load_checkpoint(_golden("ckpt_small_pretrained.pyth"), m, data_parallel=False, epoch_reset=True,
                               clear_name_pattern=("module.",), replace_name_pattern=(("nonexistent_a", "nonexistent_b"),),
                               load_orvit_attn_from_bb=True)
    assert epoch == int(e["finetune_epoch"]) == -1
    untouched_ref = set(e["finetune_untouched"].tolist())
    sd = m.state_dict()
    for k in sd:
        if k in untouched_ref:
            assert torch.equal(sd[k], before[k]), "%s should not have been loaded" % k
        else:
            assert np.array_equal(sd[k].numpy(), e["finetune." + k]), k
    assert {"head.weight", "head.bias"} <= untouched_ref
    sq = cu.split_qkv(OrderedDict([("blocks.0.attn.qkv.weight", torch.arange(24.).reshape(6, 4)), ("x", torch.ones(2))]))
    assert list(sq.keys()) == e["split_keys"].tolist()
    assert np.array_equal(sq["blocks.0.attn.q.weight"].numpy(), e["split_q"])
    assert np.array_equal(sq["blocks.0.attn.v.weight"].numpy(), e["split_v"])
