"""The N>1 path on CPU: two gloo ranks shard the frame range, each computes its counters
(the oracle stands in for the GPU kernel -- same stream definition), one all-reduce, and the
result must equal the single-process sweep bit for bit."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup():
    sys.path.insert(0, ROOT)
    import wofdm_amd as W
    from oracle import oracle as O
    ch = np.load(os.path.join(ROOT, "tests", "golden", "channels_vehA.npz"))["h"][:2]
    st = W.make_structure("wtx", 64, 16)
    osys = O.make_sys(64, 2, 16, st.cp, st.cs, st.tail_tx, st.tail_rx, st.prefix_rm, st.circ_shift, 21, 1)
    args = (osys, W.tx_rc_window(st), W.rx_rc_window(st), ch, [3.0, 18.0], 42)
    return W, O, args


def _worker(rank, world, port, total, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    W, O, args = _setup()
    off, cnt = W.distributed.frame_shard(total, rank, world, frame_offset=5)
    local = O.run(*args, off, cnt, n_threads=1)
    t = torch.from_numpy(local.view(np.int64).copy())
    W.distributed.all_reduce_counts(t)
    red2 = W.distributed.reduce_counts_numpy(local)
    assert np.array_equal(t.numpy().view(np.uint64), red2)
    np.save(os.path.join(out_dir, "r%d.npy" % rank), t.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_reduce_equals_single_process(tmp_path):
    total, world = 21, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, total, str(tmp_path)), nprocs=world, join=True)
    W, O, args = _setup()
    whole = O.run(*args, 5, total, n_threads=2)
    for r in range(world):
        got = np.load(str(tmp_path / ("r%d.npy" % r))).view(np.uint64)
        assert np.array_equal(got, whole)


def test_all_reduce_is_noop_without_process_group():
    sys.path.insert(0, ROOT)
    import wofdm_amd as W
    t = torch.arange(8, dtype=torch.int64)
    assert torch.equal(W.distributed.all_reduce_counts(t.clone()), t)
