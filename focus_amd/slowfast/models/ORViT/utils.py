"""ObjectsCrops / box2spatial_layout / Mlp (mirror of slowfast/models/ORViT/utils.py)."""
import torch
from torch import nn

from focus_amd import ops
from focus_amd.slowfast.utils.box_ops import box_cxcywh_to_xyxy

from ..common import Mlp  # noqa: F401  (ORViT/utils.py:79-98 is the same fc1-GELU-fc2 module)
from .layout import boxes_to_layout


def box2spatial_layout(box_tensors, action_map, H, W):
    """box_tensors [B,T,O,4] cxcywh, action_map [B,T,O,C] -> [B,C,T,H,W]  (ORViT/utils.py:8-28)."""
    B, T, O, C = action_map.shape
    lay = boxes_to_layout(action_map.reshape(B * T, O, C), box_tensors.reshape(B * T, O, 4), H, W)
    return lay.view(B, T, H, W, C).permute(0, 4, 1, 2, 3)


class ObjectsCrops(nn.Module):
    """Per-frame RoIAlign of the patch-token feature map (ORViT/utils.py:30-76)."""

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.aligned = True
        self.sampling_ratio = -1
        self.video_hw = (cfg.DATA.TRAIN_CROP_SIZE, cfg.DATA.TRAIN_CROP_SIZE)

    def rois(self, boxes):
        """[B,T,O,4] cxcywh in [0,1] -> (rois [B*T*O,4] xyxy pixels fp32, roi_img [B*T*O] int32)."""
        B, T, O, _ = boxes.shape
        Horig, Worig = self.video_hw
        xy = box_cxcywh_to_xyxy(boxes.reshape(B * T * O, 4).float())
        scale = torch.tensor([Worig, Horig, Worig, Horig], device=boxes.device, dtype=torch.float32)
        img = torch.arange(B * T, device=boxes.device, dtype=torch.int32).repeat_interleave(O)
        return (xy * scale).contiguous(), img

    def crop_tokens(self, patch_tokens, boxes, T, H, W):
        """Channels-last fast path: patch_tokens [B, T*H*W, C] -> crops [B*T*O, H*W, C]."""
        B, _, C = patch_tokens.shape
        rois, img = self.rois(boxes)
        return ops.roi_align_tokens(patch_tokens.reshape(B * T, H * W, C), rois, img, H, W, H, W,
                                    H / self.video_hw[0], self.sampling_ratio, self.aligned)

    def crop_stream(self, x, boxes, T, H, W, relu=False):
        """Same crops from the residual stream x [B, 1+T*H*W, C] itself: the patch tokens are read in place behind the
        cls row (no slice copy), and the gradient comes back in the stream's shape.  relu: max(., 0) of the crops, fused."""
        rois, img = self.rois(boxes)
        return ops.roi_align_stream(x, rois, img, T, H, W, H, W, H / self.video_hw[0], self.sampling_ratio, self.aligned,
                                    relu=relu)

    def forward(self, features, boxes):
        """Reference signature: features [B,d,T,H,W], boxes [B,T,O,4] -> [B,O,T,d,H,W]."""
        B, d, T, H, W = features.shape
        O = boxes.size(2)
        tok = features.permute(0, 2, 3, 4, 1).reshape(B, T * H * W, d)
        crops = self.crop_tokens(tok, boxes, T, H, W)                    # [B*T*O, H*W, d]
        return crops.view(B, T, O, H, W, d).permute(0, 2, 1, 5, 3, 4).contiguous()
