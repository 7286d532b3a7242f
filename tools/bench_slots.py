"""STEVE slot-attention update at BASELINE configs[2] shape: emb_set ~ N(0,1) [B,T,N,D] = [32,24,4096,192], 11 slots,
3 iterations, forward + backward through focus_amd's SlotAttentionVideo (bf16).  Prints clips/s and frame-slot
updates/s.  usage: python tools/bench_slots.py [--batch 32] [--frames 24]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--frames", type=int, default=24)
    ap.add_argument("--tokens", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=3)
    args = ap.parse_args()
    from focus_amd.slowfast.models.STEVE.steve import SlotAttentionVideo
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = SlotAttentionVideo(num_iterations=3, num_slots=11, input_size=192, slot_size=192, mlp_hidden_size=1024,
                           num_predictor_blocks=1, num_predictor_heads=4, dropout=0.0).to(dev)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(args.batch, args.frames, args.tokens, 192, generator=g).to(dev, torch.bfloat16).requires_grad_()
    noise = torch.randn(args.batch, 11, 192, generator=g).to(dev)

    def step():
        slots, attns = m(x, noise=noise)
        (slots.float().square().mean() + attns.float().mean()).backward()

    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    print(json.dumps({"workload": "STEVE SlotAttentionVideo fwd+bwd bf16", "batch": args.batch, "frames": args.frames,
                      "tokens": args.tokens, "slots": 11, "iters": 3, "ms_per_step": round(dt * 1e3, 2),
                      "clips_per_s": round(args.batch / dt, 2),
                      "slot_updates_per_s": round(args.batch * args.frames * 3 / dt, 1)}))


if __name__ == "__main__":
    main()
