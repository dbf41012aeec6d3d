"""bench.py's harness (warm-up, barrier-bracketed timed region, MAX over ranks, the self-describing collective block) on
two CPU ranks over gloo, with an injected step standing in for the HIP path: the same functions `bench.py --gpus N` runs
under RCCL, so a SCALE run's JSON can be trusted to say how many ranks the collective really saw."""
import json
import os
import socket
import sys
import time

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import bench
    import flm_amd  # noqa: F401
    from flm_amd import distributed
    import torch.distributed as dist
    distributed.init_process_group("gloo")
    per_rank, total = 6, 12
    calls = []

    def step():   # injected predict: rank r's landmarks are r + face/100, then the real gather
        calls.append(1)
        if rank == 1:
            time.sleep(0.02)   # the slower rank sets the job's time
        lm = torch.arange(per_rank, dtype=torch.float64).reshape(per_rank, 1, 1).expand(per_rank, 68, 2) / 100 + rank
        return distributed.all_gather_landmarks(lm.contiguous(), total), None

    dt, (full, _) = bench.timed_region(step, steps=4, warmup=2, world=world, device=torch.device("cpu"))
    coll = bench.describe_collective(world, torch.device("cpu"), per_rank, 68)
    rec = {"dt": dt, "calls": len(calls), "coll": coll, "rows": int(full.shape[0]),
           "first_of_rank1": float(full[per_rank, 0, 0])}
    with open(os.path.join(out_dir, "rank%d.json" % rank), "w") as f:
        json.dump(rec, f)
    dist.barrier()
    dist.destroy_process_group()


def test_harness_world2(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    recs = [json.load(open(os.path.join(str(tmp_path), "rank%d.json" % r))) for r in range(world)]
    for r in recs:
        assert r["calls"] == 6                                  # 2 warm-ups + exactly 4 timed steps
        assert r["rows"] == 12 and r["first_of_rank1"] == 1.0   # every rank holds the whole, in rank order
        c = r["coll"]
        assert c["ranks_seen"] == 2 and c["backend"] == "gloo" and c["op"] == "all_gather_into_tensor"
        assert c["devices"] == ["cpu", "cpu"] and c["rccl_version"] is None
        assert "6,68,2" in c["payload"] and str(6 * 68 * 2 * 8) in c["payload"]
    assert recs[0]["dt"] == recs[1]["dt"] >= 4 * 0.02          # MAX over ranks: the slow rank's time on both


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
