"""The N>1 code path on the one-GPU box.  RCCL needs one device per rank, so what CAN run here is
(a) bench.py's multi-rank path end to end with two ranks sharing cuda:0 and gloo collectives
(--rehearse-on-one-gpu: same sharding, same reduce call, same timing protocol), and (b) the "nccl"
(= RCCL) process group itself with a single rank: communicator set-up and an all-reduce of the device
counter tensor.  A multi-rank RCCL reduce has never executed in this repo's tests."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_bench_two_rank_rehearsal_on_one_gpu():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    F = 2000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "2", "--warmup", "1", "--frames-per-step", str(F),
           "--rehearse-on-one-gpu", "--no-cpu-baseline"]
    two = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert two.returncode == 0, two.stderr[-3000:]
    line = [l for l in two.stdout.splitlines() if l.startswith("{")][-1]
    d2 = json.loads(line)
    assert d2["n_gpus"] == 2 and d2["scaling"] == "weak" and d2["value"] > 0
    assert d2["config"]["symbols_per_step_per_gpu"] == F * 16 * 12
    # the same frames on one rank: rank r of 2 simulated frames [(3 + 2 i + r) F', ...) -- the union over
    # both ranks and both timed steps is frames [2F, 6F) of every cell
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0",
                          "--frames-per-step", str(4 * F), "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-3000:]
    d1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    # not the same frame range (warm-up offsets differ), so compare the curves statistically
    b1, b2 = np.array(d1["ber"]), np.array(d2["ber"])
    assert np.all(np.abs(b1[:6] - b2[:6]) < 0.02 * b1[:6] + 1e-4)


def test_rccl_process_group_single_rank_reduces_device_counters():
    code = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %r)
import wofdm_amd as W
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=%r, RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
assert W.distributed._backend() == "nccl"
t = torch.arange(48, dtype=torch.int64, device="cuda").reshape(1, 12, 1, 4)
ref = t.clone()
W.distributed.all_reduce_counts(t)                 # RCCL all-reduce on the device tensor
torch.cuda.synchronize()
assert torch.equal(t, ref)
h = np.arange(48, dtype=np.uint64).reshape(1, 12, 1, 4)
assert np.array_equal(W.distributed.reduce_counts_numpy(h), h)    # host array: through a device copy
dist.barrier()
dist.destroy_process_group()
print("RCCL_OK")
""" % (ROOT, str(_free_port()))
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"),
                         capture_output=True, text=True, timeout=300)
    assert "RCCL_OK" in out.stdout, (out.stdout[-500:], out.stderr[-3000:])
