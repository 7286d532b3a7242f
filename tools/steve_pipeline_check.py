"""Diagnostic: is the slot update reproducible run to run (eager), and does a whole-step graph replay equal it, with the
next-frame k/v pipeline on the side stream?   python tools/steve_pipeline_check.py [batch=32] [captures=1]
captures > 1: that many FRESH captures, each replayed three times (the mismatch seen in round 3 was in the first replay of
one capture in nine); for every mismatch the first frame whose slots differ is printed."""
import sys

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from focus_amd import ops  # noqa: E402
from focus_amd.slowfast.models.STEVE.steve import SlotAttentionVideo  # noqa: E402
from focus_amd.train import GraphedStep  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T, N, D, K = 24, 4096, 192, 11
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = SlotAttentionVideo(3, K, D, D, 4 * D, num_predictor_blocks=1, num_predictor_heads=4, dropout=0.0).to(dev)
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(B, T, N, D, device=dev, dtype=torch.bfloat16, generator=g).requires_grad_(True)
noise = torch.randn(B, K, D, device=dev, generator=g)


def reset():
    x.grad = None
    for p in m.parameters():
        p.grad = None


def fwd_bwd():
    s, a = m(x, noise=noise)
    (s.float().square().mean() + a.float().mean()).backward()
    return s, a


def snap(s, a):
    torch.cuda.synchronize()
    return [s.detach().clone(), a.detach().clone(), x.grad.detach().clone()] + [p.grad.detach().clone() for p in m.parameters()]


names = ["slots", "attn", "dx"] + [n for n, _ in m.named_parameters()]
runs = []
for i in range(4):
    reset()
    runs.append(snap(*fwd_bwd()))
for i in range(1, 4):
    bad = [n for n, a_, b_ in zip(names, runs[0], runs[i]) if not torch.equal(a_, b_)]
    print("eager run %d vs 0: %s" % (i, "identical" if not bad else "DIFFERS in " + ", ".join(bad[:8])))
s = a = None
ncap = int(sys.argv[2]) if len(sys.argv) > 2 else 1
for c in range(ncap):
    reset()
    gs = GraphedStep(fwd_bwd, reset=reset)
    for i in range(3):
        out = gs.replay()
        cur = snap(*out)
        bad = [n for n, a_, b_ in zip(names, runs[0], cur) if not torch.equal(a_, b_)]
        where = ""
        if bad:
            ds = (cur[0].float() - runs[0][0].float()).abs().amax(dim=(0, 2, 3))        # slots [B,T,K,D] -> per frame
            da = (cur[1].float() - runs[0][1].float()).abs().amax(dim=(0, 2, 3))
            ft = [int(t) for t in torch.nonzero(ds > 0).flatten()[:4]]
            fa = [int(t) for t in torch.nonzero(da > 0).flatten()[:4]]
            where = "  (slots differ first at frames %s, max %.3g; attn at %s, max %.3g)" % (ft, float(ds.max()), fa, float(da.max()))
        print("capture %d replay %d vs eager 0: %s%s" % (c, i, "identical" if not bad else "DIFFERS in " + ", ".join(bad[:8]), where),
              flush=True)
    del gs, out, cur
    torch.cuda.synchronize()
