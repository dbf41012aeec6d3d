"""World-size-2 run of the multi-GPU path on CPU (gloo): batch sharding + landmark all-gather.

The per-rank compute is injected (the oracle's decode on CPU, standing in for the HIP path,
which needs a GPU); what is under test is `distributed.shard_range`, `all_gather_landmarks`
(equal and ragged shards), its pipelined form `all_gather_landmarks_async`, and `sharded_predict`.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import flm_amd  # noqa: F401
    from flm_amd import distributed
    from oracle import decode_ref
    r, lr, w = distributed.init_process_group("gloo")
    assert (r, w) == (rank, world)
    hm_all = np.random.default_rng(0).random((total, 24, 20, 6), dtype=np.float32)
    lo, hi = distributed.shard_range(total, rank, world)

    def predict_fn(batch):
        with np.errstate(all="ignore"):
            return torch.from_numpy(decode_ref.transfer_target_ref(batch, 0, 4).reshape(len(batch), 6, 2))

    full = distributed.sharded_predict(predict_fn, hm_all[lo:hi], total)
    np.save(os.path.join(out_dir, "rank%d.npy" % rank), full.numpy())
    # the same exchange without waiting for it: the shard may be overwritten while the gather is in flight
    mine = predict_fn(hm_all[lo:hi])
    pend = distributed.all_gather_landmarks_async(mine, total)
    mine.fill_(-7.0)
    assert torch.equal(pend.wait(), full)
    assert torch.equal(pend.wait(), full)     # waiting twice is harmless
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [8, 7])
def test_world2_gather_equals_single_process(tmp_path, total):
    from oracle import decode_ref
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, total, str(tmp_path)), nprocs=world, join=True)
    hm_all = np.random.default_rng(0).random((total, 24, 20, 6), dtype=np.float32)
    with np.errstate(all="ignore"):
        exp = decode_ref.transfer_target_ref(hm_all, 0, 4).reshape(total, 6, 2)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "rank%d.npy" % r))
        assert got.shape == exp.shape
        assert np.array_equal(got, exp)       # every rank holds the whole, in batch order


def test_single_process_gather_is_identity():
    import flm_amd  # noqa: F401
    from flm_amd import distributed
    x = torch.arange(24, dtype=torch.float64).reshape(4, 3, 2)
    assert distributed.all_gather_landmarks(x, 4) is x
    with pytest.raises(ValueError):
        distributed.all_gather_landmarks(x, 5)
    assert distributed.all_gather_landmarks_async(x, 4).wait() is x
    with pytest.raises(ValueError):
        distributed.all_gather_landmarks_async(x, 5)
