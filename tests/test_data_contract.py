"""Data-side contract: spatial sampling with boxes, pathway packing and the
`orvit_bboxes` wire format against a fixture produced by the reference's own datasets/utils.py, datasets/transform.py and
utils/box_ops.py (oracle/make_golden.py main_data).  Same numpy seed -> the same draws -> bit-identical frames and boxes."""
import numpy as np
import pytest
import torch

from conftest import GOLDEN
from focus_amd.slowfast.config.defaults import get_cfg
from focus_amd.slowfast.datasets import utils as du
from focus_amd.slowfast.utils import box_ops

CASES = [("train", dict(spatial_idx=-1, min_scale=36, max_scale=48, crop_size=32)),
         ("train_inv", dict(spatial_idx=-1, min_scale=36, max_scale=48, crop_size=32, inverse_uniform_sampling=True)),
         ("test0", dict(spatial_idx=0, min_scale=36, max_scale=36, crop_size=32)),
         ("test1", dict(spatial_idx=1, min_scale=36, max_scale=36, crop_size=32)),
         ("test2", dict(spatial_idx=2, min_scale=36, max_scale=36, crop_size=32))]


def fixture():
    import os
    return np.load(os.path.join(GOLDEN, "data_contract.npz"), allow_pickle=False)


def run_case(z, tag, kw, device):
    cfg = get_cfg()
    cfg.DATA.REVERSE_INPUT_CHANNEL = True
    cfg.MODEL.ARCH = "mformer"
    np.random.seed(int(z[tag + ".seed"]))
    f, b = du.spatial_sampling(torch.from_numpy(z["frames"]).to(device), boxes=z["boxes"].copy(), random_horizontal_flip=True, **kw)
    packed = du.pack_pathway_output(cfg, f.permute(1, 0, 2, 3))[0]
    return packed, b, du.boxes_to_orvit_format(b, packed.shape[-2], packed.shape[-1])


@pytest.mark.parametrize("tag,kw", CASES)
def test_spatial_sampling_with_boxes_matches_the_reference(tag, kw):
    z = fixture()
    packed, b, ob = run_case(z, tag, kw, "cpu")
    assert torch.equal(packed, torch.from_numpy(z[tag + ".frames"]))
    assert np.array_equal(b, z[tag + ".boxes_px"])
    assert torch.equal(ob, torch.from_numpy(z[tag + ".orvit_bboxes"]))
    # wire format: [T,O,4] cxcywh in [0,1]; absent / degenerate objects are all-zero rows
    assert ob.shape == (5, 3, 4) and float(ob.min()) >= 0 and float(ob.max()) <= 1
    assert float(ob[1, 2].abs().max()) == 0.0 and float(ob[3, 0].abs().max()) == 0.0


def test_tensor_normalize_and_box_helpers():
    z = fixture()
    u8 = (torch.from_numpy(z["frames"]) * 255).to(torch.uint8).permute(0, 2, 3, 1)
    assert torch.equal(du.tensor_normalize(u8, [0.45, 0.45, 0.45], [0.225, 0.225, 0.225]), torch.from_numpy(z["norm"]))
    b = torch.tensor([[0.2, 0.3, 0.6, 0.5], [0.1, 0.1, 0.12, 0.9]])
    c = box_ops.box_xyxy_to_cxcywh(b)
    assert torch.allclose(box_ops.box_cxcywh_to_xyxy(c), b, atol=1e-7)
    assert torch.equal(box_ops.zero_empty_boxes(c.clone())[1], torch.zeros(4))          # 0.02 wide: "no object"
    assert torch.equal(box_ops.remove_empty_boxes(b), b[:1])
    assert torch.equal(box_ops.box_xywh_to_xyxy(torch.tensor([[1.0, 2.0, 3.0, 4.0]])), torch.tensor([[1.0, 2.0, 4.0, 6.0]]))
    with pytest.raises(NotImplementedError):
        du.spatial_sampling(torch.zeros(2, 3, 8, 8), aspect_ratio=[0.75, 1.33], scale=[0.08, 1.0])


@pytest.mark.gpu
@pytest.mark.parametrize("tag,kw", CASES)
def test_same_augmentation_on_the_gpu(tag, kw):
    """The clip can be augmented where it is decoded to: identical boxes, frames equal to bilinear-resize rounding."""
    z = fixture()
    packed, b, ob = run_case(z, tag, kw, "cuda:0")
    assert packed.is_cuda
    assert float((packed.cpu() - torch.from_numpy(z[tag + ".frames"])).abs().max()) < 1e-5
    assert np.array_equal(b, z[tag + ".boxes_px"]) and torch.equal(ob, torch.from_numpy(z[tag + ".orvit_bboxes"]))
