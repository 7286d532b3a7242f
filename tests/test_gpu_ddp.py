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


@pytest.mark.parametrize("mixed", [0, 1])
def test_ddp_gradients_equal_single_process(mixed, tmp_path):
    """with the motion side stream ON and the gradients flowing through DDP's reducer (communication
    hook joining the side stream, focus_amd/parallel.py), the 2-rank gradients (one clip each, averaged by the reducer)
    equal the single-process gradients of the 2-clip batch.  fp32 mode: 1e-5.  bf16 mode: the per-clip activations are
    bit-identical; what differs is where the sum over the two clips happens -- inside the single process some gradients are
    summed over the batch in bf16 (autograd's expand / broadcast adjoints on bf16 tensors, e.g. box_categories: one bf16
    rounding, 2^-8 = 3.9e-3), across ranks every clip's gradient is rounded first and averaged in fp32: 8e-3."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    outs = [str(tmp_path / ("r%d.json" % r)) for r in range(2)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_ddp_grad_worker.py"), str(r), str(port), str(mixed),
                               outs[r]], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    logs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(l[-3000:] for l in logs)
    rec = json.load(open(outs[0]))
    assert rec["n"] > 50
    assert rec["worst"] < (8e-3 if mixed else 1e-5), rec
