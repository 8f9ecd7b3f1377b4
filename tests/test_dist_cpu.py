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


# ---------------------------------------------------------------------------
# round 6: per-world tensors other than the state, 8 ranks, the single-process driver
# ---------------------------------------------------------------------------
def _worker8(rank, ws, port, n_worlds, out_dir):
    from arboris_python_amd.dist import gather_rows
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    a, b = shard_bounds(n_worlds, rank, ws)
    idx = torch.arange(a, b, dtype=torch.float64)
    q_all, dq_all = gather_state(idx[:, None] * torch.ones(1, 5, dtype=torch.float64), -idx[:, None] * torch.ones(1, 3, dtype=torch.float64), n_worlds, dist)
    cost = gather_rows(idx * 0.5, n_worlds, dist)                       # per-rollout costs: (shard,) -> (n_worlds,)
    cf = gather_rows(idx[:, None, None] * torch.ones(1, 4, 4, dtype=torch.float64), n_worlds, dist)
    full = torch.arange(n_worlds, dtype=torch.float64)
    ok = (torch.equal(q_all[:, 0], full) and torch.equal(dq_all[:, 2], -full) and torch.equal(cost, full * 0.5)
          and tuple(cf.shape) == (n_worlds, 4, 4) and torch.equal(cf[:, 3, 1], full))
    open(os.path.join(out_dir, "rank%d.ok" % rank), "w").write("1" if ok else "0")
    dist.destroy_process_group()


@pytest.mark.parametrize("n_worlds", [16384, 65536 - 5, 3])        # config 4's batch; a ragged config-5 batch; fewer worlds than ranks
def test_eight_ranks_gather_state_and_costs(tmp_path, n_worlds):
    """The final gather of BASELINE configs 4 / 5 with EIGHT ranks (gloo): 2048- and 8192-world shards, a ragged last shard,
    empty shards; state, per-rollout costs and contact forces."""
    ws = 8
    mp.spawn(_worker8, args=(ws, _free_port(), n_worlds, str(tmp_path)), nprocs=ws, join=True)
    for r in range(ws):
        assert open(os.path.join(str(tmp_path), "rank%d.ok" % r)).read() == "1"


class _FakeStream(object):
    def __init__(self):
        self.synced = 0

    def synchronize(self):
        self.synced += 1


class _FakeStepper(object):
    """Stand-in for BatchedWorlds on the CPU: `step` is any per-world map (worlds are independent); it records the stream it
    was given and the number of calls."""

    def __init__(self, model, device):
        self.model, self.device, self.calls, self.closed = model, device, [], False

    def new_stream(self):
        return _FakeStream()

    def to_device(self, q, dq, dtype=None):
        dtype = torch.float32 if dtype is None else dtype
        return torch.tensor(np.asarray(q), dtype=dtype), torch.tensor(np.asarray(dq), dtype=dtype)      # (copies, like an upload)

    def new_cforce(self, B, dtype):
        return torch.zeros((B, 2, 4), dtype=dtype)

    def step(self, q, dq, dt, nsteps=1, stream=None, cforce=None, ext_gforce=None, cost=None):
        self.calls.append((q.shape[0], stream))
        for _ in range(nsteps):
            dq += dt * (1. if ext_gforce is None else ext_gforce)
            q += dt * dq[:, :1]
        if cforce is not None:
            cforce += 1.
        if cost is not None:
            cost["out"] += (dq ** 2).sum(dim=1)

    def close(self):
        self.closed = True


@pytest.mark.parametrize("B,devices", [(37, [0, 1, 2]), (4096, list(range(8))), (3, list(range(8))), (10, [0, 0])])
def test_sharded_worlds_single_process_driver(B, devices):
    """dist.ShardedWorlds with stand-in steppers: contiguous shards by shard_bounds, one stepper and one stream per shard,
    per-shard inputs and costs, empty shards skipped, gather in world order == the unsharded result."""
    from arboris_python_amd.dist import ShardedWorlds
    sw = ShardedWorlds("model", devices=devices, factory=_FakeStepper)
    assert sw.nshards == len(devices) and [s.device for s in sw.steppers] == devices
    rng = np.random.default_rng(1)
    q, dq = rng.normal(size=(B, 4)), rng.normal(size=(B, 3))
    tau = torch.as_tensor(rng.normal(size=(B, 3)))
    shards = sw.scatter(q, dq, torch.float64, cforce=True)
    assert [(s["lo"], s["hi"]) for s in shards] == [shard_bounds(B, k, len(devices)) for k in range(len(devices))]
    costs = [dict(out=torch.zeros(s["hi"] - s["lo"], dtype=torch.float64)) for s in shards]
    per = [dict(ext_gforce=tau[s["lo"]:s["hi"]], cost=c) for s, c in zip(shards, costs)]
    sw.step(shards, 0.5, 3, per_shard=per)
    out = sw.gather(shards, keys=("q", "dq", "cforce"), extra=[dict(cost=c["out"]) for c in costs])
    ref = _FakeStepper("model", -1)
    rq, rdq = ref.to_device(q, dq, torch.float64)
    rc = dict(out=torch.zeros(B, dtype=torch.float64))
    rcf = ref.new_cforce(B, torch.float64)
    ref.step(rq, rdq, 0.5, 3, cforce=rcf, ext_gforce=tau, cost=rc)
    assert torch.equal(out["q"], rq) and torch.equal(out["dq"], rdq) and torch.equal(out["cost"], rc["out"]) and torch.equal(out["cforce"], rcf)
    for s, st, sh in zip(sw.steppers, sw.streams, shards):
        n = sh["hi"] - sh["lo"]
        assert s.calls == ([(n, st)] if n else [])           # its own stream; an empty shard is never launched
        assert st.synced >= 1
    sw.close()
    assert sw.steppers == []
