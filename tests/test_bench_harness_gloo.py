"""bench.py's harness (warm-up, barrier-bracketed timed region, MAX over ranks, the self-describing collective block) on
two and eight CPU ranks over gloo, with an injected step standing in for the HIP path: the same functions `bench.py --gpus N` runs
under RCCL, so a SCALE run's JSON can be trusted to say how many ranks the collective really saw."""
import json
import os
import socket
import sys
import time

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


def _worker(rank, world, port, out_dir, total):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import bench
    import flm_amd  # noqa: F401
    from flm_amd import distributed
    import torch.distributed as dist
    torch.set_num_threads(1)
    distributed.init_process_group("gloo")
    lo, hi = distributed.shard_range(total, rank, world)
    per_rank = hi - lo
    calls = []

    buf = torch.empty((per_rank, 68, 2), dtype=torch.float64)   # ONE output buffer, overwritten by every step

    def produce():   # injected predict: rank r's landmarks are r + face/10000 + step/100
        calls.append(1)
        if rank == world - 1:
            time.sleep(0.02)   # the slower rank sets the job's time
        buf.copy_(torch.arange(per_rank, dtype=torch.float64).reshape(per_rank, 1, 1).expand(per_rank, 68, 2) / 10000
                  + rank + len(calls) / 100)
        return buf, None

    # the bench's own step: the gather of step i is waited for after step i + 1 has been queued, the last one by the
    # closing fence of the timed region
    step = bench.pipeline_gather(produce, world, total)

    dt, (full, _), stats = bench.timed_region(step, steps=4, warmup=2, world=world, device=torch.device("cpu"))
    coll = bench.describe_collective(world, torch.device("cpu"), per_rank, 68)
    gather_ms = bench.time_gather(world, torch.device("cpu"), 512 if total % world == 0 else 8, 68, reps=3)
    starts = [distributed.shard_range(total, r, world)[0] for r in range(world)]
    rec = {"dt": dt, "calls": len(calls), "coll": coll, "rows": int(full.shape[0]), "per_rank": per_rank, "stats": stats,
           "gather_ms": gather_ms, "first_of_each_rank": [round(float(full[st, 0, 0]) - 0.06, 6) for st in starts],
           "last_row": float(full[total - 1, 5, 1]) - 0.06}   # 0.06: the result is the LAST (sixth) call's gather
    with open(os.path.join(out_dir, "rank%d.json" % rank), "w") as f:
        json.dump(rec, f)
    dist.barrier()
    dist.destroy_process_group()


def _run(tmp_path, world, total):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), total), nprocs=world, join=True)
    return [json.load(open(os.path.join(str(tmp_path), "rank%d.json" % r))) for r in range(world)]


def test_harness_world2(tmp_path):
    world = 2
    recs = _run(tmp_path, world, 12)
    for r in recs:
        assert r["calls"] == 6                                  # 2 warm-ups + exactly 4 timed steps
        assert r["rows"] == 12 and r["first_of_each_rank"] == [0.0, 1.0]   # every rank holds the whole, in rank order
        c = r["coll"]
        assert c["ranks_seen"] == 2 and c["backend"] == "gloo" and c["op"] == "all_gather_into_tensor"
        assert c["devices"] == ["cpu", "cpu"] and c["rccl_version"] is None
        assert "6,68,2" in c["payload"] and str(6 * 68 * 2 * 8) in c["payload"]
        assert "278,528" in c["payload_deviation"]              # the fp32 figure north_star names is stated as a deviation
    assert recs[0]["dt"] == recs[1]["dt"] >= 4 * 0.02          # MAX over ranks: the slow rank's time on both


@pytest.mark.parametrize("total", [4096, 4090])
def test_harness_world8_config4_shards(tmp_path, total):
    """The 8-rank flow of `bench.py --gpus 8 --config 4` on CPU: 4096 = 8 x 512 faces (and a ragged 4090), the agreed
    step counts, MAX-over-ranks time, per-rank step times that expose the straggler, the gather timed alone, and a
    collective block that names 8 ranks and 8 devices."""
    world = 8
    recs = _run(tmp_path, world, total)
    counts = [512] * 8 if total == 4096 else [512, 512] + [511] * 6
    for rank, r in enumerate(recs):
        assert r["calls"] == 6 and r["rows"] == total and r["per_rank"] == counts[rank]
        assert r["first_of_each_rank"] == [float(k) for k in range(8)]
        assert abs(r["last_row"] - (7 + (counts[7] - 1) / 10000)) < 1e-12
        c = r["coll"]
        assert c["ranks_seen"] == 8 and len(c["devices"]) == 8 and c["backend"] == "gloo"
        st = r["stats"]
        assert len(st["ranks"]) == 8 and st["min"] <= st["median"] <= st["max"]
        assert abs(st["max"] - 1e3 * r["dt"] / 4) < 1e-6       # the job's time is the slowest rank's
        assert st["max"] >= 20.0                                # the sleeping rank: >= 20 ms per step
        assert r["gather_ms"] is not None and r["gather_ms"] > 0
    assert len({r["dt"] for r in recs}) == 1


def test_work_accounting_is_physical():
    sys.path.insert(0, ROOT)
    import bench
    assert abs(bench.GFLOP_PER_FACE - 17.844) < 2e-3
    dense, useful = bench.WORK["fc6"]
    assert abs(dense - 6.5767) < 1e-3 and abs(useful / dense - 44 * 44 / (56 * 56)) < 1e-12
    assert 15.0 < bench.USEFUL_GFLOP_PER_FACE < 15.4
    for b in (1, 64, 512):                                                   # issued >= useful, <= dense
        assert useful * b - 1e-6 <= bench.fc6_issued_gflop(b) <= dense * max(b, 2) + 1e-6
    r = bench.mfma_roofline("fc6", "k", 2.2026, 64, bench.PEAK_F32_TFLOPS, "traffic_latest.json")
    assert 0.74 < r["frac"] < 0.76 and r["frac_dense"] > 1.2               # round 1's launch: 0.75 useful, 1.21 dense
