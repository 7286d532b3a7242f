"""Which ATen operators (with input shapes and the Python line that issued them) own the device time of one whole STEVE
training step:   python3 tools/steve_op_profile.py [batch=8] > ops.txt
The rocprofv3 view (tools/trace_last_step.py) names kernels; this one names their callers."""
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from focus_amd.slowfast.config.defaults import get_cfg  # noqa: E402
from focus_amd.slowfast.models import MODEL_REGISTRY  # noqa: E402
from focus_amd.slowfast.models.optimizer import construct_optimizer_slot  # noqa: E402
from focus_amd.train import slot_train_step  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
cfg = get_cfg()
cfg.MODEL.MODEL_NAME = "STEVE"
cfg.NUM_GPUS = 1
cfg.TRAIN.MIXED_PRECISION = True
cfg.SOLVER.OPTIMIZING_METHOD = "adam"
cfg.SOLVER.CLIP_GRAD_L2NORM = 0.05
sl = cfg.SLOTS
sl.NUM_ITERS, sl.NUM_SLOTS, sl.CNN_HID_SIZE, sl.SIZE, sl.DIM, sl.MLP_HID_SIZE, sl.IMG_SIZE, sl.VOCAB_SIZE = 3, 11, 64, 192, 192, 768, 128, 4096
sl.NUM_PREDICTOR_BLOCKS, sl.NUM_PREDICTOR_HEADS, sl.PREDICTOR_DROPOUT = 1, 4, 0.0
sl.DECODER.DIM, sl.DECODER.NUM_BLOCKS, sl.DECODER.NUM_HEADS, sl.DECODER.DROPOUT = 192, 8, 4, 0.1
torch.manual_seed(0)
m = MODEL_REGISTRY.get("STEVE")(cfg).to(dev).train()
opt = construct_optimizer_slot(m, cfg)
video = torch.rand(B, 24, 3, 128, 128, device=dev)
for s in range(2):
    slot_train_step(m, opt, video, s, cfg)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    slot_train_step(m, opt, video, 2, cfg)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=6):
    t = getattr(e, "self_device_time_total", None)
    if t is None:
        t = e.self_cuda_time_total
    if t > 0 and str(getattr(e, 'device_type', '')).endswith('CPU'):
        stack = [s for s in e.stack if "focus_amd" in s or "bench.py" in s][:2]
        rows.append((t, e.count, e.key, str(e.input_shapes)[:110], " <- ".join(x.split("focus_amd/")[-1][:70] for x in stack)))
rows.sort(reverse=True)
total = sum(r[0] for r in rows)
print("self device time %.1f ms over %d (operator, shapes, stack) groups" % (total / 1e3, len(rows)))
for t, n, k, shp, st in rows[:110]:
    print("%8.3f ms %5.1f%% %4d x  %-38s %s\n%32s%s" % (t / 1e3, 100.0 * t / total, n, k[:38], shp, "", st))
