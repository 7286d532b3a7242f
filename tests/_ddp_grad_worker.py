"""Worker of tests/test_gpu_ddp.py::test_ddp_gradients_equal_single_process: rank r of 2 (gloo, both on cuda:0) runs the
small ORViT-Motionformer on clip r through build_model -> wrap_ddp with the motion side stream and the deferred-gradient
machinery ON; rank 0 then runs the unwrapped module on both clips in one batch and compares the gradients.
usage: python tests/_ddp_grad_worker.py <rank> <port> <mixed 0|1> <out json>"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    rank, port, mixed, out = int(sys.argv[1]), int(sys.argv[2]), bool(int(sys.argv[3])), sys.argv[4]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=2)
    from focus_amd.slowfast.config.defaults import get_cfg
    from focus_amd.slowfast.models import build_model
    from focus_amd.slowfast.models.losses import get_loss_func
    from focus_amd.slowfast.models.ORViT import orvit
    assert orvit._USE_SIDE_STREAM
    z = np.load(os.path.join(ROOT, "tests", "golden", "motionformer_small.npz"))
    params = {k[2:]: torch.from_numpy(z[k]).float() for k in z.files if k.startswith("p.")}
    cfg = get_cfg()
    cfg.merge_from_list(["ORVIT.ENABLE", True, "ORVIT.O", 3, "ORVIT.LAYERS", [1], "DATA.TRAIN_CROP_SIZE", 64,
                         "DATA.NUM_FRAMES", 4, "MF.EMBED_DIM", 64, "MF.DEPTH", 3, "MF.NUM_HEADS", 4,
                         "MF.TEMPORAL_RESOLUTION", 2, "MF.USE_MLP", True, "MF.DROP_PATH", 0.0, "MODEL.NUM_CLASSES", 10,
                         "MODEL.MODEL_NAME", "Motionformer", "TRAIN.DATASET", "Ssv2", "NUM_GPUS", 2,
                         "TRAIN.MIXED_PRECISION", mixed, "MODEL.LOSS_FUNC", "label_smoothing_cross_entropy",
                         "DIST_BACKEND", "gloo"])
    model = build_model(cfg, gpu_id=0)
    base = model.module
    base.load_state_dict(params)
    model.train()
    x, boxes = torch.from_numpy(z["x"]).cuda(), torch.from_numpy(z["boxes"]).cuda()
    labels = torch.from_numpy(z["labels"]).cuda()
    assert x.shape[0] == 2
    loss_fun = get_loss_func(cfg)(reduction="mean")
    for _ in range(2):                                     # twice: the second pass reuses DDP's bucket views
        model.zero_grad(set_to_none=True)
        loss = loss_fun(model([x[rank:rank + 1]], {"orvit_bboxes": boxes[rank:rank + 1]}), labels[rank:rank + 1])
        loss.backward()
    torch.cuda.synchronize()
    ddp = {n: p.grad.detach().clone() for n, p in base.named_parameters()}
    assert orvit._SIDE_STREAMS, "the motion stream did not fork"
    res = {"rank": rank}
    if rank == 0:
        base.zero_grad(set_to_none=True)
        loss = loss_fun(base([x], {"orvit_bboxes": boxes}), labels)       # both clips, one process, no reducer
        loss.backward()
        torch.cuda.synchronize()
        worst, name = 0.0, None
        for n, p in base.named_parameters():
            ref = p.grad
            e = float((ddp[n] - ref).abs().max() / ref.abs().max().clamp_min(1e-12))
            if ref.abs().max() > 1e-9 and e > worst:
                worst, name = e, n
        res.update(worst=worst, name=name, n=len(ddp))
    dist.barrier()
    with open(out, "w") as f:
        json.dump(res, f)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
