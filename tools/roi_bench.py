"""Times RoIAlign forward/backward at the bench shape (64 frames, 14x14x768 maps, 4 boxes per frame, 7x7 bins)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from focus_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
NI, H, W, C, O = 64, 14, 14, 768, 4
PHW = int(sys.argv[1]) if len(sys.argv) > 1 else 14     # ORViT pools every box to the map size (utils.py:64-71)
feat = torch.randn(NI, H * W, C, device=dev).bfloat16().requires_grad_(True)
xy = torch.rand(NI * O, 2, device=dev) * 0.5
wh = torch.rand(NI * O, 2, device=dev) * 0.45 + 0.05
rois = torch.cat([xy, xy + wh], dim=1) * 224.0
roi_img = torch.arange(NI, device=dev, dtype=torch.int32).repeat_interleave(O)
out = ops.roi_align_tokens(feat, rois, roi_img, H, W, PHW, PHW, 1.0 / 16.0)
g = torch.randn_like(out)
out.backward(g)
ref = feat.grad.clone()
torch.cuda.synchronize()
e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tf = tb = 0.0
n = 10
for _ in range(n):
    feat.grad = None
    e[0].record()
    out = ops.roi_align_tokens(feat, rois, roi_img, H, W, PHW, PHW, 1.0 / 16.0)
    e[1].record()
    out.backward(g)
    e[2].record()
    torch.cuda.synchronize()
    tf += e[0].elapsed_time(e[1])
    tb += e[1].elapsed_time(e[2])
print("ROI fwd %.1f us  bwd %.1f us  |grad| %.4f" % (tf / n * 1e3, tb / n * 1e3, float(ref.float().abs().mean())), flush=True)
