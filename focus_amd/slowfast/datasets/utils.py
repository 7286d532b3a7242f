"""Clip post-processing between decoding and the model (mirror of slowfast/datasets/utils.py:75-188, 319-336 and of the
box hand-off in slowfast/datasets/ssv2.py:333-346)."""
import numpy as np
import torch

from ..utils import box_ops
from . import transform


def tensor_normalize(tensor, mean, std):
    """utils.py:319-336."""
    if tensor.dtype == torch.uint8:
        tensor = tensor.float() / 255.0
    if type(mean) == list:
        mean = torch.tensor(mean, device=tensor.device)
    if type(std) == list:
        std = torch.tensor(std, device=tensor.device)
    return (tensor - mean) / std


def pack_pathway_output(cfg, frames):
    """utils.py:75-108: optional BGR<->RGB channel reversal, then one tensor per pathway.  The hot-path models are
    single-pathway (Motionformer / STEVE); multi-pathway archs get the reference's temporal sub-sampling too."""
    if cfg.DATA.REVERSE_INPUT_CHANNEL:
        frames = frames[[2, 1, 0], :, :, :]
    single = getattr(cfg.MODEL, "SINGLE_PATHWAY_ARCH", ["c2d", "i3d", "slow", "x3d", "mvit", "mformer", "slowfast_mf"])
    multi = getattr(cfg.MODEL, "MULTI_PATHWAY_ARCH", ["slowfast"])
    if cfg.MODEL.ARCH in multi and cfg.MODEL.ARCH not in single:
        alpha = cfg.SLOWFAST.ALPHA
        idx = torch.linspace(0, frames.shape[1] - 1, frames.shape[1] // alpha).long().to(frames.device)
        return [torch.index_select(frames, 1, idx), frames]
    if cfg.MODEL.ARCH in single:
        return [frames]
    raise NotImplementedError("Model arch {} is not in {}".format(cfg.MODEL.ARCH, single + multi))


def spatial_sampling(frames, spatial_idx=-1, min_scale=256, max_scale=320, crop_size=224, random_horizontal_flip=True,
                     inverse_uniform_sampling=False, aspect_ratio=None, scale=None, motion_shift=False, boxes=None):
    """utils.py:111-188: spatial_idx -1 = random short-side jitter + random crop + random flip; 0/1/2 = the
    deterministic test views.  frames [T,C,H,W]; boxes (optional) xyxy pixels.  Returns frames or (frames, boxes)."""
    assert spatial_idx in [-1, 0, 1, 2]
    if spatial_idx == -1:
        if aspect_ratio is None and scale is None:
            frames, boxes = transform.random_short_side_scale_jitter(images=frames, min_size=min_scale, max_size=max_scale,
                                                                     inverse_uniform_sampling=inverse_uniform_sampling,
                                                                     boxes=boxes)
            frames, boxes = transform.random_crop(frames, crop_size, boxes=boxes)
        else:
            raise NotImplementedError("random_resized_crop (utils.py:151-164, the RandAugment pipeline) is not on the "
                                      "ORViT data path")
        if random_horizontal_flip:
            frames, boxes = transform.horizontal_flip(0.5, frames, boxes=boxes)
    else:
        assert len({min_scale, max_scale}) == 1
        frames, boxes = transform.random_short_side_scale_jitter(frames, min_scale, max_scale, boxes=boxes)
        frames, boxes = transform.uniform_crop(frames, crop_size, spatial_idx, boxes=boxes)
    if boxes is not None:
        return frames, boxes
    return frames


def boxes_to_orvit_format(boxes, height, width):
    """ssv2.py:337-346: pixel xyxy boxes of the augmented clip [T,O,4] -> metadata['orvit_bboxes'] (cxcywh in [0,1],
    float tensor, boxes thinner than 0.05 in either direction zeroed = "no object")."""
    boxes = boxes.copy() if isinstance(boxes, np.ndarray) else boxes.detach().cpu().numpy().copy()
    boxes[..., [0, 2]] = boxes[..., [0, 2]] / width
    boxes[..., [1, 3]] = boxes[..., [1, 3]] / height
    boxes = np.clip(boxes, 0, 1)
    boxes = torch.from_numpy(boxes)
    boxes = box_ops.box_xyxy_to_cxcywh(boxes)
    return box_ops.zero_empty_boxes(boxes, mode="cxcywh")
