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


def gather_rows(local, n_worlds, dist=None):
    """All-gather one per-world tensor ``(shard, ...)`` -- per-rollout costs ``(shard,)``, contact forces ``(shard, nc, 4)``
    -- into ``(n_worlds, ...)`` on every rank; ragged shards as in ``gather_state``."""
    import torch
    if dist is None:
        import torch.distributed as dist
    ws = dist.get_world_size()
    per = -(-n_worlds // ws)
    width = 1
    for d in local.shape[1:]:
        width *= int(d)
    flat = local.reshape(local.shape[0], width)      # (an empty shard: -1 would be ambiguous)
    if flat.shape[0] < per:
        flat = torch.cat([flat, torch.zeros((per - flat.shape[0], flat.shape[1]), dtype=flat.dtype, device=flat.device)], dim=0)
    flat = flat.contiguous()
    parts = [torch.empty_like(flat) for _ in range(ws)]
    dist.all_gather(parts, flat)
    return torch.cat(parts, dim=0)[:n_worlds].reshape((n_worlds,) + tuple(local.shape[1:]))


class ShardedWorlds(object):
    """ONE host thread drives every GPU of the node ("one host thread + one stream per GPU"): the library-level
    counterpart of ``bench.py``'s process-per-GPU launch, for callers that want a batch larger than one device -- or
    simply all devices -- without ``torch.distributed``.

    The global batch is cut into contiguous shards with ``shard_bounds`` (world ``w`` -> shard ``w // ceil(B / G)``);
    every shard lives on its own device, is stepped by that device's ``BatchedWorlds`` handle on a stream of its own, and
    no shard ever waits for another: ``step`` returns as soon as every launch is queued (the C ABI is asynchronous on the
    caller's stream, include/arbstep.h), ``synchronize`` waits for all of them.  There is no collective: ``gather``
    copies the shards to the host (or to one device) and concatenates them in world order -- the "final state gather" of
    the north star; per-rollout costs are per-world tensors like the state and travel the same way.

    ``devices``: device indices, one shard each (default: every visible device; an index may repeat -- two shards on one
    device, each on its own stream: how the single-GPU tests exercise the driver).
    ``factory(model, device) -> stepper``: what steps a shard; default ``BatchedWorlds``.  A stepper provides
    ``device``, ``to_device(q, dq, dtype)``, ``new_cforce(B, dtype)``, ``step(q, dq, dt, nsteps, stream=..., **kw)`` and
    ``close()`` (the CPU tests pass a stand-in, tests/test_dist_cpu.py)."""

    def __init__(self, model, devices=None, factory=None):
        if devices is None:
            import torch
            devices = list(range(torch.cuda.device_count()))
        if not devices:
            raise RuntimeError("ShardedWorlds needs at least one device (there is no CPU fallback for the step)")
        if factory is None:
            from .batch import BatchedWorlds
            factory = BatchedWorlds
        self.devices = [int(d) for d in devices]
        self.steppers = [factory(model, d) for d in self.devices]
        self.model = getattr(self.steppers[0], "model", model)
        self.streams = [self._new_stream(s) for s in self.steppers]

    @staticmethod
    def _new_stream(stepper):
        make = getattr(stepper, "new_stream", None)
        if make is not None:                 # (stand-in steppers of the CPU tests)
            return make()
        import torch
        return torch.cuda.Stream(device=stepper.device)

    @property
    def nshards(self):
        return len(self.steppers)

    def bounds(self, n_worlds):
        return [shard_bounds(n_worlds, k, self.nshards) for k in range(self.nshards)]

    def scatter(self, q, dq, dtype=None, cforce=False):
        """Host arrays ``(B, nq)``, ``(B, ndof)`` -> a list of per-shard dicts ``{"q", "dq"[, "cforce"], "lo", "hi"}`` on the
        shards' devices.  A shard may be empty (fewer worlds than devices)."""
        B = len(q)
        out = []
        for s, (lo, hi) in zip(self.steppers, self.bounds(B)):
            tq, tdq = s.to_device(q[lo:hi], dq[lo:hi], dtype)
            sh = dict(q=tq, dq=tdq, lo=lo, hi=hi)
            if cforce:
                sh["cforce"] = s.new_cforce(hi - lo, tq.dtype)
            out.append(sh)
        return out

    def step(self, shards, dt, nsteps=1, per_shard=None, **kw):
        """Queue ``nsteps`` steps of every shard on its own stream and return at once.  ``per_shard``: a list of extra
        keyword dicts, one per shard (its rows of ``ext_gforce``, ``pd_targets``, ``cost`` ...); ``kw`` goes to all."""
        for k, (s, st, sh) in enumerate(zip(self.steppers, self.streams, shards)):
            if sh["hi"] == sh["lo"]:
                continue
            extra = dict(kw)
            if per_shard is not None:
                extra.update(per_shard[k])
            if "cforce" in sh and "cforce" not in extra:
                extra["cforce"] = sh["cforce"]
            s.step(sh["q"], sh["dq"], dt, nsteps, stream=st, **extra)

    def synchronize(self):
        for st in self.streams:
            st.synchronize()

    def gather(self, shards, keys=("q", "dq"), extra=None):
        """Synchronise, then concatenate the shards' tensors in world order on the host: ``{key: (B, ...) tensor}``.
        ``extra``: a list of per-shard dicts of further per-world tensors (per-rollout costs ``(shard,)``)."""
        import torch
        self.synchronize()
        out = {}
        for key in keys:
            out[key] = torch.cat([sh[key].detach().cpu() for sh in shards], dim=0)
        if extra is not None:
            for key in extra[0]:
                out[key] = torch.cat([e[key].detach().cpu() for e in extra], dim=0)
        return out

    def close(self):
        for s in self.steppers:
            s.close()
        self.steppers = []
