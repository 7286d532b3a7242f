"""Runs the whole-model STEVE training step of bench.py (bench_steve_model) on its own, for rocprofv3:
   rocprofv3 --kernel-trace --output-format csv -d /tmp/t -- python3 tools/steve_model_step.py [batch=8] [steps=2]
   python3 tools/trace_last_step.py /tmp/t xent_ls_kernel 50     (training steps only: the last complete one)"""
import sys
import types

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import bench  # noqa: E402

a = types.SimpleNamespace(steve_model_batch=int(sys.argv[1]) if len(sys.argv) > 1 else 8,
                          steps=int(sys.argv[2]) if len(sys.argv) > 2 else 2, steve_model_eval=False)
print(bench.bench_steve_model(a, torch.device("cuda:0")))
