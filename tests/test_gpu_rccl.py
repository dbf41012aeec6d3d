"""RCCL on the box: the collectives the multi-GPU harness uses, at world size 1, in a child process.

A one-GPU box cannot run two RCCL ranks (RCCL refuses two ranks on one device), so what this pins is narrower than
the world-2 gloo tests: that `backend="nccl"` initialises on this image with the environment `distributed.py` sets
(dmabuf IPC), and that `all_gather_into_tensor` (blocking, and queued / waited for later as bench.py's pipelined step does) /
`all_reduce(MIN)` / `barrier` -- the calls of bench.py's multi-rank flow -- run on device tensors produced by the HIP path.  The N > 1 numbers are the driver's to measure.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys
sys.path.insert(0, %r)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=%r, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import numpy as np, torch, torch.distributed as dist
import flm_amd
from flm_amd.utils.metrics import decode_device
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
assert dist.get_backend() == "nccl"
hm = torch.from_numpy(np.random.default_rng(3).random((4, 24, 20, 6), dtype=np.float32)).cuda()
lm = decode_device(hm, n_points=4, thresh=0.0)                    # [4, 6, 2] on the device, from the HIP decode
out = torch.empty_like(lm)
dist.all_gather_into_tensor(out, lm.contiguous())
assert torch.equal(out, lm)
# the pipelined step of bench.py (pipeline_gather): the gather of batch i is queued behind it on RCCL's stream and
# waited for after batch i + 1 has been queued; the decode writes the SAME output buffer every batch
from flm_amd import distributed
buf = torch.empty_like(lm)
pending, got, want = None, [], []
for i in range(4):
    hm_i = hm * (1.0 + 0.25 * i) + 0.01 * i
    decode_device(hm_i, n_points=4, thresh=0.0, out=buf)
    want.append(decode_device(hm_i, n_points=4, thresh=0.0).clone())
    nxt = distributed.all_gather_landmarks_async(buf, 4, single_rank_collective=True)
    if pending is not None:
        got.append(pending.wait())
    pending = nxt
got.append(pending.wait())
torch.cuda.synchronize()
assert len(got) == 4 and all(torch.equal(g, w) for g, w in zip(got, want)), "pipelined gathers"
n = torch.tensor([7], device="cuda", dtype=torch.int64)
dist.all_reduce(n, op=dist.ReduceOp.MIN)
assert int(n.item()) == 7
dist.barrier()
torch.cuda.synchronize()
print("RCCL", ".".join(str(v) for v in torch.cuda.nccl.version()), "ok", flush=True)
dist.destroy_process_group()
'''


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.gpu
def test_rccl_world1_collectives_on_device_landmarks():
    r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, str(_free_port()))], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "RCCL" in r.stdout and " ok" in r.stdout, r.stdout
