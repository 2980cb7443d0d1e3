"""The N>1 code path on the one-GPU box.  RCCL needs one device per rank, so what CAN run here is
(a) bench.py's multi-rank path end to end with two ranks sharing cuda:0 and gloo collectives
(--rehearse-on-one-gpu: same sharding, same reduce call, same timing protocol), (b) the sharding contract itself --
two ranks, contiguous frame ranges, Plan.launch + all_reduce_counts, BIT-IDENTICAL to one rank over the whole range --
and (c) the "nccl" (= RCCL) process group itself with a single rank: communicator set-up and an all-reduce of the device
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
                          "--frames-per-step", str(4 * F), "--frame-offset", str(2 * F), "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-3000:]
    d1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    # the same frames, so the same counters, bit for bit (the streams are keyed by the global frame index)
    assert d1["bits"] == d2["bits"] and d1["bit_errors"] == d2["bit_errors"] and d1["ber"] == d2["ber"]
    assert min(d1["bits"]) > 0 and d1["bit_errors"][0] > 0


_TWO_RANK = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
import wofdm_amd as W
rank, world, F = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), %(frames)d
torch.cuda.set_device(0)                                   # both ranks share the box's one GPU: gloo collectives
dist.init_process_group("gloo", rank=rank, world_size=world)
ch = np.load(os.path.join(%(root)r, "tests", "golden", "channels_vehA.npz"))["h"][:1]
st = W.make_structure("wtx", 256, 32)
cfg = W.make_cfg(st, 4, 16, 21, 1, 12, 1, seed=2024)       # C2's twelve cells
snr = np.arange(-5.0, 51.0, 5.0).astype(np.float32)
with W.Plan(cfg, W.tx_rc_window(st), W.rx_rc_window(st), ch.astype(np.complex64), snr) as plan:
    off, cnt = W.distributed.frame_shard(F, rank, world, frame_offset=%(offset)d)
    counts = plan.new_counts()
    plan.launch(off, cnt, counts)                          # this rank's contiguous share of every cell's frames
    W.distributed.all_reduce_counts(counts)                # the sweep's one exchange (device tensor; host copy under gloo)
    torch.cuda.synchronize()
    np.save(os.path.join(%(out)r, "rank%%d.npy" %% rank), counts.cpu().numpy())
dist.barrier()
dist.destroy_process_group()
"""


def test_two_ranks_on_one_gpu_add_up_bit_exactly(tmp_path):
    """The sharding contract on the GPU (the reference's parfor, matlab/main_BER_calculation.m:31, and Pool.map,
    python/wofdm_optimization.py:127-129): rank g of G launches frames [g F / G, (g + 1) F / G) of every cell through
    Plan.launch, one all_reduce_counts follows, and EVERY rank then holds exactly the counters of a one-rank run over [0, F) --
    array_equal, not a tolerance: the Philox streams are keyed by the global frame index."""
    F, offset = 3001, 7                                        # (odd: the two shares differ in length)
    script = tmp_path / "two_rank.py"
    script.write_text(_TWO_RANK % dict(root=ROOT, frames=F, offset=offset, out=str(tmp_path)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(_free_port()), str(script)], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    sys.path.insert(0, ROOT)
    import wofdm_amd as W
    ch = np.load(os.path.join(ROOT, "tests", "golden", "channels_vehA.npz"))["h"][:1]
    st = W.make_structure("wtx", 256, 32)
    cfg = W.make_cfg(st, 4, 16, 21, 1, 12, 1, seed=2024)
    snr = np.arange(-5.0, 51.0, 5.0).astype(np.float32)
    with W.Plan(cfg, W.tx_rc_window(st), W.rx_rc_window(st), ch.astype(np.complex64), snr) as plan:
        whole = plan.run(offset, F)
    assert whole[..., 1].min() > 0 and whole[:, :4, :, 0].min() > 0          # bits were counted, and errors at the low SNR points
    for rank in range(2):
        got = np.load(str(tmp_path / ("rank%d.npy" % rank))).view(np.uint64).reshape(whole.shape)
        assert np.array_equal(got, whole), rank


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
