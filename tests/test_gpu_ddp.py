"""N > 1 on the GPU box: bench.py launches its own ranks (no torchrun) and runs the REAL HIP Motionformer through
build_model -> wrap_ddp for a few steps.  The box has one GPU, so the two ranks share cuda:0 over gloo (RCCL wants one
GPU per rank: that leg is the driver's 8-GPU run); what is covered is the launcher, DDP's hooks on our autograd
Functions, the bucket views, the packed timing reduction and the exposed-all-reduce measurement."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_two_ranks_same_device_gloo():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--same-device",
           "--steps", "2", "--warmup", "1", "--batch", "1", "--no-cpu-baseline", "--no-roofline", "--workload", "orvit"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-4000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["config"]["global_batch"] == 2 and rec["config"]["parallelism"] == "dp2"
    assert rec["value"] > 0 and rec["final_loss"] == rec["final_loss"]
    assert "exposed_allreduce_ms" in rec and rec["ms_per_step_no_allreduce"] > 0
