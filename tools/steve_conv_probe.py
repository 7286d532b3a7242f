"""Probe: the convolution stacks of STEVE (dVAE encoder / decoder, CNN encoder; MIOpen) forward + backward in fp32 and
under bf16 autocast (what the reference's fp16 autocast does to them), channels-last, B*T = 192 frames of 128 x 128.
   python3 tools/steve_conv_probe.py [frames=192]"""
import sys
import time
import types

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from focus_amd.slowfast.models.STEVE.dvae import dVAE  # noqa: E402
from focus_amd.slowfast.models.STEVE.steve import BaseCNN  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 192
dev = torch.device("cuda:0")
torch.manual_seed(0)
args = types.SimpleNamespace(SLOTS=types.SimpleNamespace(IMG_CHANNELS=3, CNN_HID_SIZE=64, IMG_SIZE=128,
                                                          DECODER=types.SimpleNamespace(DIM=192)))
dv = dVAE(4096, 3).to(dev).to(memory_format=torch.channels_last)
cnn = BaseCNN(args).to(dev).to(memory_format=torch.channels_last)
img = torch.rand(n, 3, 128, 128, device=dev).contiguous(memory_format=torch.channels_last)
z = torch.rand(n, 4096, 32, 32, device=dev).contiguous(memory_format=torch.channels_last)
cases = [("dvae.encoder", dv.encoder, img), ("dvae.decoder", dv.decoder, z), ("cnn", cnn, img)]
for name, mod, x in cases:
    for label, dt in (("fp32", None), ("bf16 autocast", torch.bfloat16)):
        def once():
            xx = x.detach().requires_grad_(True)
            with torch.autocast("cuda", dtype=dt, enabled=dt is not None):
                y = mod(xx)
            y.float().square().mean().backward()
            return y
        try:
            for _ in range(2):
                y = once()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                once()
            torch.cuda.synchronize()
            print("%-14s %-14s fwd+bwd %8.2f ms   out %s %s" % (name, label, 1e3 * (time.perf_counter() - t0) / 3,
                                                              tuple(y.shape), y.dtype), flush=True)
        except Exception as e:
            print("%-14s %-14s FAILED: %s" % (name, label, str(e)[:200]), flush=True)
