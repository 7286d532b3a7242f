"""CPU-side checks of the boundary: the C-ABI library loads, exports every symbol include/focus_amd.h
declares, refuses CPU tensors (no fallback), and the host mirror keeps the reference's surface."""
import ctypes
import os

import numpy as np
import pytest
import torch

from conftest import ROOT


@pytest.fixture(scope="module")
def built():
    from focus_amd.build import build
    return build(verbose=False)


def test_library_exports_every_declared_symbol(built):
    from focus_amd import _lib
    decl = _lib.parse_header()
    assert len(decl) >= 30
    L = ctypes.CDLL(built)
    for name in decl:
        assert hasattr(L, name), "missing export: " + name
    lib = _lib.lib()
    assert lib.focus_abi_version() == 2
    assert lib.focus_strerror(-1) == b"bad shape" and lib.focus_strerror(0) == b"ok"
    # pure host-side queries work without a GPU
    assert lib.focus_layernorm_bwd_blocks(10) == 3 and lib.focus_layernorm_bwd_blocks(10 ** 6) == 512
    assert lib.focus_traj_space_workspace_bytes(1, 8, 196, 12, 64, 0, 0) >= 12 * 1568 * 1568 * 4
    assert lib.focus_slot_attn_workspace_bytes(2, 4096, 11, 192) == 2 * 64 * 11 * 193 * 4     # 64-row chunks


def test_gemm_desc_layout_matches_header():
    from focus_amd._lib import GemmDesc
    assert ctypes.sizeof(GemmDesc) == 208       # 5 int32 (+pad) | 3 x (ptr + 4 int64) | 3 ptr | f32 + 4 int32 (+pad) | dtype_b, pad, b_scale ptr
    assert GemmDesc.A.offset == 24 and GemmDesc.bias.offset == 144 and GemmDesc.alpha.offset == 168


def test_hot_path_has_no_cpu_fallback(built):
    from focus_amd import ops
    x = torch.randn(4, 8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.linear(x, torch.randn(8, 8), None)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.layer_norm(x, torch.ones(8), torch.zeros(8), 1e-6)


def test_row_kernels_validate_before_launching(built):
    """gumbel.hip's entry points refuse bad arguments with the ABI's error codes before any launch (no GPU needed), and the
    operators above them refuse CPU tensors."""
    from focus_amd import _lib, ops
    lib = _lib.lib()
    F32, BF16 = _lib.F32, _lib.BF16
    assert lib.focus_rows_ok(196608, 4096, BF16) == 1 and lib.focus_rows_ok(5, 8192, F32) == 1
    assert lib.focus_rows_ok(5, 4100, F32) == 0 and lib.focus_rows_ok(5, 8200, F32) == 0         # V % 8, V <= 8192
    assert lib.focus_rows_ok(0, 4096, F32) == 0 and lib.focus_rows_ok(5, 4096, _lib.FP8_E4M3) == 0
    buf = ctypes.create_string_buffer(1 << 16)
    a = ctypes.addressof(buf)
    a += -a % 16
    ptr = ctypes.c_void_p(a)
    NULL, SHAPE, DTYPE, ALIGN = -5, -1, -2, -3
    assert lib.focus_gumbel_fwd(None, None, None, ptr, ptr, None, ptr, 4, 64, 1.0, 0, F32, F32, None) == NULL
    assert lib.focus_gumbel_fwd(ptr, None, None, None, ptr, None, ptr, 4, 64, 1.0, 0, F32, F32, None) == NULL   # no noise, no seed
    assert lib.focus_gumbel_fwd(ptr, ptr, None, None, ptr, ptr, ptr, 4, 64, 1.0, 0, F32, F32, None) == NULL     # target without e_hard
    assert lib.focus_gumbel_fwd(ptr, None, None, ptr, ptr, None, ptr, 4, 60, 1.0, 0, F32, F32, None) == SHAPE
    assert lib.focus_gumbel_fwd(ptr, None, None, ptr, ptr, None, ptr, 4, 64, 0.0, 0, F32, F32, None) == SHAPE    # tau > 0
    assert lib.focus_gumbel_fwd(ptr, None, None, ptr, ptr, None, ptr, 4, 64, 1.0, 0, BF16, F32, None) == DTYPE   # fp32 sample of bf16 logits
    odd = ctypes.c_void_p(a + 4)
    assert lib.focus_gumbel_fwd(odd, None, None, ptr, ptr, None, ptr, 4, 64, 1.0, 0, F32, F32, None) == ALIGN
    assert lib.focus_gumbel_bwd(ptr, None, None, ptr, ptr, ptr, 4, 64, 1.0, F32, F32, None) == NULL              # neither noise nor seed
    assert lib.focus_xent_rows_fwd(ptr, ptr, ptr, None, 4, 64, 0.0, F32, None) == NULL
    assert lib.focus_xent_rows_bwd(ptr, ptr, ptr, ptr, ptr, 4, 12, 0.0, F32, None) == SHAPE
    assert lib.focus_dropout_add(ptr, None, None, 6554, ptr, 64, F32, None) == NULL
    assert lib.focus_dropout_add(ptr, None, ptr, 6554, ptr, 60, F32, None) == SHAPE
    assert lib.focus_dropout_add(ptr, None, ptr, 65536, ptr, 64, F32, None) == SHAPE
    assert lib.focus_dropout_add(ptr, odd, ptr, 6554, ptr, 64, BF16, None) == ALIGN
    x = torch.randn(4, 64)
    assert not ops.rows_ok(x)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.gumbel_softmax_rows(x, 1.0, False)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.dropout_add(x, x, 0.1, True)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.label_smoothing_ce(x, torch.zeros(4, dtype=torch.long), 0.0)
    assert torch.equal(ops.dropout_add(x, x, 0.1, False), x + x)          # evaluation: the plain sum, any device


def test_config_and_registry_surface():
    from focus_amd.slowfast.config.defaults import assert_and_infer_cfg, get_cfg
    from focus_amd.slowfast.models import MODEL_REGISTRY, build_model
    cfg = get_cfg()
    assert cfg.ORVIT.O == 5 and cfg.MF.TEMPORAL_RESOLUTION == 8 and cfg.SOLVER.CLIP_GRAD_L2NORM == 0.05
    assert cfg.TRAIN.METHOD == "slots" and cfg.DIST_BACKEND == "nccl" and cfg.SLOTS.NUM_ITERS == 3
    cfg.merge_from_list(["ORVIT.LAYERS", "[1,6,10]", "NUM_GPUS", "0", "MODEL.MODEL_NAME", "Motionformer",
                         "ORVIT.ENABLE", "True", "ORVIT.O", 4, "TRAIN.DATASET", "Ssv2", "MF.USE_MLP", True,
                         "MODEL.NUM_CLASSES", 174, "DATA.NUM_FRAMES", 16])
    assert cfg.ORVIT.LAYERS == [1, 6, 10] and cfg.NUM_GPUS == 0 and cfg.ORVIT.ENABLE is True
    assert_and_infer_cfg(cfg)
    assert "Motionformer" in MODEL_REGISTRY
    with pytest.raises(KeyError):
        MODEL_REGISTRY.get("SlowFast")
    model = build_model(cfg)
    sd = model.state_dict()
    z = np.load(os.path.join(ROOT, "tests", "golden", "motionformer_224_keys.npz"))
    names = sorted(sd.keys())
    assert names == list(z["names"])
    assert [",".join(map(str, sd[k].shape)) for k in names] == list(z["shapes"])
    assert sum(p.numel() for p in model.parameters()) == int(z["nparams"]) == 147506862
    assert model.no_weight_decay() == {"pos_embed", "cls_token", "temp_embed"}
    c2 = cfg.clone()
    c2.MF.DEPTH = 1
    assert cfg.MF.DEPTH == 12
    cfg.freeze()
    with pytest.raises(AttributeError):
        cfg.NUM_GPUS = 3


def test_yaml_configs_of_the_reference_shape_load(tmp_path):
    """A YAML with the reference's ORViT keys plus keys for out-of-scope subsystems merges cleanly."""
    from focus_amd.slowfast.config.defaults import get_cfg
    y = tmp_path / "c.yaml"
    y.write_text("ORVIT:\n  ENABLE: True\n  O: 4\n  LAYERS: [1,6,10]\nAUG:\n  ENABLE: True\n"
                 "MF:\n  DROP_PATH: 0.2\n  USE_MLP: True\nSOLVER:\n  BASE_LR: 5e-5\n  OPTIMIZING_METHOD: adamw\n"
                 "NUM_GPUS: 8\n")
    cfg = get_cfg()
    cfg.merge_from_file(str(y))
    assert cfg.ORVIT.LAYERS == [1, 6, 10] and cfg.AUG.ENABLE is True and cfg.MF.DROP_PATH == 0.2
    assert cfg.SOLVER.BASE_LR == 5e-5


def test_slot_module_surface():
    from focus_amd.slowfast.models.STEVE.steve import SlotAttentionVideo
    from conftest import load_golden
    _, p = load_golden("slot_attention")
    m = SlotAttentionVideo(3, 3, 12, 8, 16, num_predictor_blocks=2, num_predictor_heads=2, dropout=0.0)
    assert sorted(m.state_dict().keys()) == sorted(p.keys())
    m.load_state_dict({k: v.float() for k, v in p.items()})


@pytest.mark.parametrize("mixed", [False, True])
def test_steve_precision_plan(mixed):
    """Which parts of STEVE.forward run in which type: under TRAIN.MIXED_PRECISION the token path and the
    dVAE decoder / CNN encoder convolutions in bf16, the dVAE encoder and its Gumbel-softmax arithmetic in fp32; otherwise
    fp32 throughout.  The module tree and parameter types are the reference's in both modes (fp32 masters)."""
    from focus_amd.slowfast.config.defaults import get_cfg
    from focus_amd.slowfast.models import MODEL_REGISTRY
    cfg = get_cfg()
    cfg.MODEL.MODEL_NAME = "STEVE"
    cfg.TRAIN.MIXED_PRECISION = mixed
    s = cfg.SLOTS
    s.NUM_ITERS, s.NUM_SLOTS, s.CNN_HID_SIZE, s.SIZE, s.DIM, s.MLP_HID_SIZE, s.IMG_SIZE, s.VOCAB_SIZE = 2, 3, 8, 16, 16, 32, 16, 32
    s.NUM_PREDICTOR_BLOCKS, s.NUM_PREDICTOR_HEADS = 1, 2
    s.DECODER.DIM, s.DECODER.NUM_BLOCKS, s.DECODER.NUM_HEADS = 16, 1, 2
    m = MODEL_REGISTRY.get("STEVE")(cfg)
    assert m.compute_dtype == (torch.bfloat16 if mixed else torch.float32)
    assert m.conv_dtype == (torch.bfloat16 if mixed else None)
    assert m.fused_rows and m.channels_last
    assert all(p.dtype == torch.float32 for p in m.parameters() if p.is_floating_point())     # (the causal mask is a bool parameter)
    x = torch.rand(2, 3, 16, 16)
    assert m._conv(m.dvae.encoder, x).dtype == torch.float32          # CPU tensors: no autocast, and the encoder never gets one
    with pytest.raises(RuntimeError, match="no CPU fallback"):        # the model itself runs on the GPU only
        m(torch.rand(1, 2, 3, 16, 16), 1.0, True)
