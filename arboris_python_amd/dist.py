"""Sharding of a batch of worlds over the ranks of one node (one process per GPU).

Worlds never interact (each arboris ``World`` owns all of its state), so the only
distributed pieces are (1) the contiguous partition of the world index over ranks
and (2) the final gather of ``(q, dq)``.  ``torch.distributed`` with backend
``"nccl"`` is RCCL over xGMI on ROCm; the same code runs with ``"gloo"`` on CPU
tensors (tests/test_dist_cpu.py).  No collective is issued while stepping.
"""


def shard_bounds(n_worlds, rank, world_size):
    """Contiguous split of [0, n_worlds): world w belongs to rank w // ceil(n/ws)
    (SURVEY 8e); returns (start, stop) of this rank's shard."""
    per = -(-n_worlds // world_size)
    start = min(n_worlds, rank * per)
    return start, min(n_worlds, start + per)


def gather_state(q_local, dq_local, n_worlds, dist=None):
    """All-gather the per-rank ``(q, dq)`` shards into the full ``(n_worlds, .)``
    tensors on every rank.  Shards may be ragged (last rank shorter); they are
    padded to the common shard size for the collective and trimmed afterwards."""
    import torch
    if dist is None:
        import torch.distributed as dist
    ws = dist.get_world_size()
    per = -(-n_worlds // ws)
    state = torch.cat([q_local, dq_local], dim=1)
    if state.shape[0] < per:
        pad = torch.zeros((per - state.shape[0], state.shape[1]), dtype=state.dtype, device=state.device)
        state = torch.cat([state, pad], dim=0)
    state = state.contiguous()
    parts = [torch.empty_like(state) for _ in range(ws)]
    dist.all_gather(parts, state)
    full = torch.cat(parts, dim=0)[:n_worlds]
    nq = q_local.shape[1]
    return full[:, :nq].contiguous(), full[:, nq:].contiguous()
