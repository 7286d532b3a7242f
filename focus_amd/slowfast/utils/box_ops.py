"""Box helpers on the hot path and on the data hand-off (mirror of slowfast/utils/box_ops.py:10-28, 108-131)."""
import torch


def box_xywh_to_xyxy(x):
    x0, y0, w, h = x.unbind(-1)
    return torch.stack([x0, y0, x0 + w, y0 + h], dim=-1)


def box_cxcywh_to_xyxy(x):
    cx, cy, w, h = x.unbind(-1)
    return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)


def box_xyxy_to_cxcywh(x):
    x0, y0, x1, y1 = x.unbind(-1)
    return torch.stack([(x0 + x1) / 2, (y0 + y1) / 2, (x1 - x0), (y1 - y0)], dim=-1)


def zero_empty_boxes(boxes, mode="cxcywh", eps=0.05):
    """box_ops.py:108-122: boxes whose width or height is <= eps become all-zero rows (the "no object" encoding the
    ORViT block tests for).  Modifies `boxes` in place like the reference when it can be viewed as [N,4]."""
    assert isinstance(boxes, torch.Tensor)
    oshape = boxes.shape
    boxes = boxes.reshape(-1, 4)
    if mode == "xyxy":
        wh = boxes[..., [2, 3]] - boxes[..., [0, 1]]
    elif mode == "cxcywh":
        wh = boxes[..., -2:]
    else:
        raise NotImplementedError(mode)
    assert torch.all(wh >= 0)
    boxes[torch.any(wh <= eps, dim=-1)] = 0
    return boxes.reshape(oshape)


def remove_empty_boxes(box, eps=0.05, mode="xyxy"):
    """box_ops.py:124-131."""
    assert mode in ["xyxy"]
    assert len(box.shape) == 2
    H, W = box[:, 3] - box[:, 1], box[:, 2] - box[:, 0]
    return box[(H > eps) * (W > eps)]
