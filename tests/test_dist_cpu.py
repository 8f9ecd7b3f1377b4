"""world_size-2 gloo test of the batch sharding + final state gather (CPU)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from arboris_python_amd.dist import shard_bounds, gather_state


def test_shard_bounds_partition():
    for n in (0, 1, 7, 4096, 65536, 65537):
        for ws in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, ws) for r in range(ws)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a, b), (c, d) in zip(spans, spans[1:]):
                assert b == c and a <= b
            # world w -> rank w // ceil(n / ws)
            per = -(-n // ws) if n else 1
            for r, (a, b) in enumerate(spans):
                assert all(w // per == r for w in (a, b - 1) if a < b)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, ws, port, n_worlds, nq, n, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    rng = np.random.default_rng(0)                      # same global batch on every rank
    Q = torch.tensor(rng.normal(size=(n_worlds, nq)))
    DQ = torch.tensor(rng.normal(size=(n_worlds, n)))
    a, b = shard_bounds(n_worlds, rank, ws)
    # stand-in for the device step: any per-world map (worlds are independent)
    q_loc = Q[a:b] * 2.0 + 1.0
    dq_loc = DQ[a:b] - 3.0
    q_all, dq_all = gather_state(q_loc, dq_loc, n_worlds, dist)
    ok = bool(torch.equal(q_all, Q * 2.0 + 1.0) and torch.equal(dq_all, DQ - 3.0))
    open(os.path.join(out_dir, "rank%d.ok" % rank), "w").write("1" if ok else "0")
    dist.destroy_process_group()


@pytest.mark.parametrize("n_worlds", [64, 37])
def test_gather_state_gloo_world_size_2(tmp_path, n_worlds):
    ws = 2
    port = _free_port()
    mp.spawn(_worker, args=(ws, port, n_worlds, 52, 42, str(tmp_path)), nprocs=ws, join=True)
    for r in range(ws):
        assert open(os.path.join(str(tmp_path), "rank%d.ok" % r)).read() == "1"
