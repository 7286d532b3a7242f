"""Box layout (mirror of slowfast/models/ORViT/layout.py:28-63, 98-130, 205-237).

The reference samples a constant 8x8 image per object with F.grid_sample and scatter-adds over objects;
that composition has the closed form documented in SURVEY.md A4, which focus_box_layout_* evaluates in
one launch per call (no per-object images, no python loop)."""
from focus_amd import ops


def boxes_to_layout(vecs, boxes_cxcywh, H, W=None, pooling="sum"):
    """vecs [NF,O,C]; boxes [NF,O,4] in **cxcywh** (the xyxy conversion of ORViT/utils.py:20 and the
    width/height reading of layout.py:113-120 happen inside the kernel) -> [NF, H*W, C]."""
    if pooling != "sum":
        raise ValueError('Invalid pooling "%s"' % pooling)
    return ops.box_layout(vecs, boxes_cxcywh, H, W if W is not None else H)
