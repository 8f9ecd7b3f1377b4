#!/usr/bin/env python3
"""Benchmark of the batched arboris step on MI355X.

Metric (BASELINE.json): world-steps/s, human36 (42 dof) + 4 floor
SoftFingerContacts, batch 4096 worlds per GPU (BASELINE config #3), synthetic
standing/falling states, float32 state and arithmetic.

A "step" is one pass of the hot path (update_dynamic -> update_controllers ->
update_constraints -> integrate, arboris/core.py:1358-1363) over the whole batch:
every world advances by one dt.  The workload is the reference's falling scenario
(tests/test_human36_falling.py): an EPISODE of 40 steps from the standing-drop
states -- free fall, impact, sliding contacts.  The cost of a step depends on where in
the episode it is, so the timed region is always a whole number of episodes: `--steps K`
is rounded up to whole episodes and the episodes are repeated until the region lasts
`--min-seconds` (and at least 50 launches).  One episode = one arb_step launch (the state
stays on chip for its 40 steps); the state is restored from the pristine batch before
every episode by a device-to-device copy inside the timed region.

Launch:  python bench.py [--gpus N --steps K --warmup W --config {2,3,4,5}]
         With N > 1 and no WORLD_SIZE in the environment bench.py starts
         `python -m torch.distributed.run --nproc-per-node N ... bench.py ...` itself
         (a child process, before anything touches the GPU).
Worlds are independent, so ranks shard the batch with no data-path collective
(weak scaling: the per-GPU batch is fixed; dist.shard_bounds gives each rank its
contiguous range of the global batch); RCCL is used only for the barrier/max-time
reduction and for the final state gather, which is timed separately.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP32_PEAK_TFLOPS = 157.3     # vector FP32 spec peak

# BASELINE.json configs that fit the bench (config 1 is the single-world CPU parity case)
CONFIGS = {
    2: dict(model="human36", contacts=0, batch=1024, dtype="f32", dt=5e-3, episode=40, states="random",
            name="human36 (42 dof), no contacts, random states (BASELINE config #2)"),
    3: dict(model="human36", contacts=4, batch=4096, dtype="f32", dt=5e-3, episode=40, states="standing",
            name="human36 (42 dof) + 4 floor SoftFingerContact, standing-drop states (BASELINE config #3)"),
    4: dict(model="snake64", contacts=0, batch=2048, dtype="f64", dt=1e-3, episode=40, states="random",
            name="snake-64 (64 Rz joints), random states, 2048 worlds/GPU = 16384 on 8 GPUs (BASELINE config #4)"),
    5: dict(model="human36", contacts=4, batch=8192, dtype="f32", dt=5e-3, episode=32, states="standing",
            name="human36 + 4 contacts, 32-step horizon, 8192 worlds/GPU = 65536 on 8 GPUs (BASELINE config #5)"),
}
# the literal MPC shape of config 5 (secondary run): 2048 rollouts x 32 in-kernel steps over 8 GPUs = 256
# rollouts per GPU, every rollout with its own user torques (arb_step's ext_gforce, controllers.py:63-158's hook);
# one launch per horizon, only the final state is written.  The latency regime: one world per wave slot at most.
MPC = dict(model="human36", contacts=4, batch=256, dtype="f32", dt=5e-3, episode=32, states="standing", torques="sequence",
           name="human36 + 4 contacts, MPC shape: 32-step horizon resident in one launch, per-rollout torques, "
                "256 rollouts/GPU = 2048 on 8 GPUs (BASELINE config #5, literal shape)")


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=int, default=3, choices=sorted(CONFIGS),
                    help="BASELINE.json config (3 = the headline metric)")
    ap.add_argument("--batch", type=int, default=None, help="worlds per GPU (default: the config's)")
    ap.add_argument("--contacts", type=int, default=None, choices=(0, 4, 8))
    ap.add_argument("--dtype", default=None, choices=("f32", "f64"))
    ap.add_argument("--dt", type=float, default=None)
    ap.add_argument("--min-seconds", type=float, default=3.0,
                    help="lower bound of the timed region (whole episodes are repeated until it is reached)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--split", default=None, choices=("wave",),
                    help="Gauss-Seidel sweeps in their own kernel, one wavefront per world (ARB_STEP_SPLIT_WAVE); "
                         "default: inside the step kernel")
    ap.add_argument("--mpc", action="store_true",
                    help="with --config 5: the literal MPC shape (2048 rollouts x 32 in-kernel steps over 8 GPUs = 256 "
                         "rollouts per GPU, per-rollout torques) instead of 65536 independent worlds")
    ap.add_argument("--no-per-step-leg", action="store_true", help="skip the one-launch-per-step comparison leg")
    ap.add_argument("--extra", action="store_true", help="also time the other BASELINE configs (N=1)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher/sharding rehearsal on CPU: gloo backend, no GPU call, no stepping")
    ap.add_argument("--global-batch", type=int, default=None,
                    help="--dry-run only: a global batch that need not be a multiple of the ranks (ragged last shard, "
                         "empty shards); default ranks x the per-GPU batch")
    return ap.parse_args(argv)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launcher_command(gpus, argv):
    """The command bench.py starts when asked for N > 1 ranks without a launcher."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
            os.path.abspath(__file__)] + list(argv)


_CPU_WORKER = r"""
import sys, time, numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/oracle")
import arb_oracle as O
from arboris_python_amd.flatten import FlatModel
d = np.load(sys.argv[2])
m = FlatModel.from_npz_dict({k: d[k] for k in d.files if k not in ("q", "dq")})
rank, nw, nsteps, dt = int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), float(sys.argv[6])
q, dq = d["q"][rank * nw:(rank + 1) * nw].copy(), d["dq"][rank * nw:(rank + 1) * nw].copy()
O.step(m, q[:2], dq[:2], dt)
cf = None
t0 = time.perf_counter()
for _ in range(nsteps):
    q, dq, cf = O.step(m, q, dq, dt, cf)
print(time.perf_counter() - t0)
"""


def usable_cores():
    """Host cores this process may actually run on: the scheduler affinity mask, capped by the cgroup CPU quota when
    there is one (cpu.max of cgroup v2, cfs_quota_us / cfs_period_us of v1).  os.cpu_count() reports the machine's
    hardware threads (256 on the GPU box), of which a one-GPU lease gets a share."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:                                   # pragma: no cover
        n = os.cpu_count() or 1
    quota = None
    try:
        a, b = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if a != "max":
            quota = float(a) / float(b)
    except Exception:
        try:
            qv = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            pv = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if qv > 0:
                quota = qv / pv
        except Exception:
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n, quota


def cpu_baseline(model, q, dq, dt, budget_s, episode):
    """The NumPy float64 oracle (a port of the reference algorithm) timed on the host:
    one core, then one single-threaded process per core over disjoint world shards
    (SURVEY 8d), both on a bounded sample of the bench workload."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import arb_oracle as O
    import contextlib
    import tempfile
    try:
        from threadpoolctl import threadpool_limits
        ctx = threadpool_limits(limits=1)
    except Exception:                                   # pragma: no cover
        ctx = contextlib.nullcontext()
    nw = min(64, q.shape[0])
    qs, dqs = q[:nw].copy(), dq[:nw].copy()
    with ctx:
        O.step(model, qs[:4], dqs[:4], dt)              # warm caches / imports
        done = 0
        cf = None
        t0 = time.perf_counter()
        while True:
            qs, dqs, cf = O.step(model, qs, dqs, dt, cf)
            done += nw
            if time.perf_counter() - t0 > budget_s or done >= nw * episode:
                break
        el = time.perf_counter() - t0
    out = dict(value=done / el, unit="world-steps/s", cores=1, kind="port",
               sample="%d worlds x %d steps of the bench workload, NumPy float64 oracle "
                      "(oracle/arb_oracle.py), single thread" % (nw, done // nw))
    # all cores: child processes (they never touch the GPU), one BLAS thread each
    try:
        per = 32                                         # worlds per process (the oracle batches its NumPy calls over worlds)
        usable, quota = usable_cores()
        ncores = max(1, min(usable, 128, q.shape[0] // per))
        nsteps = max(2, min(episode, int(budget_s * out["value"] / per)))
        env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
        with tempfile.TemporaryDirectory() as td:
            f = os.path.join(td, "shard.npz")
            np.savez(f, q=q[:per * ncores], dq=dq[:per * ncores], **model.to_npz_dict())
            procs = [subprocess.Popen([sys.executable, "-c", _CPU_WORKER, ROOT, f, str(r), str(per), str(nsteps), repr(dt)],
                                      stdout=subprocess.PIPE, env=env) for r in range(ncores)]
            times = [float(p.communicate(timeout=600)[0].decode().strip().splitlines()[-1]) for p in procs]
        rates = sorted(per * nsteps / t for t in times)
        out["all_cores"] = dict(value=per * ncores * nsteps / max(times), unit="world-steps/s", cores=ncores,
                                usable_cores=usable, cgroup_cpu_quota=quota, hardware_threads=os.cpu_count(),
                                per_process_rate={"min": rates[0], "median": rates[len(rates) // 2], "max": rates[-1]},
                                per_core_vs_single=rates[len(rates) // 2] / out["value"],
                                sample="%d single-threaded processes (one per usable core: scheduler affinity, capped by the "
                                       "cgroup CPU quota) x %d worlds x %d steps (stepping loops only, slowest process); the "
                                       "single-core figure runs 64 worlds per NumPy call, these 32" % (ncores, per, nsteps))
    except Exception as e:                              # pragma: no cover
        out["all_cores"] = dict(value=None, error=repr(e))
    return out


def run_episodes(bw, q0, dq0, dt, episode, n_episodes, torch, dist=None, spl=None, split=False, timed=True, ext=None,
                 general=False, cost=None, body_columns=False, mixed=None, static_worlds=False, waves=None, classic=False):
    """Run `n_episodes` whole episodes: restore the pristine states, advance `episode` steps (one arb_step
    launch per `spl` steps; default the whole episode in one launch).  Returns wall seconds between the
    two barrier + synchronize brackets, the launch durations in ms (HIP events on the launch stream =
    torch's current stream), and the final state."""
    dev = bw.device
    spl = episode if spl is None else spl
    q, dq = q0.clone(), dq0.clone()
    cf = bw.new_cforce(q.shape[0], q.dtype) if bw.model.nc else None
    chunks = []
    k = 0
    while k < episode:
        chunks.append(min(spl, episode - k))
        k += chunks[-1]
    ev = []
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    kw = {}
    if general:
        kw["general_kernels"] = True
    if body_columns:
        kw["body_columns"] = True
    if classic:
        kw["classic_columns"] = True
    if cost is not None:
        kw["cost"] = cost
    if mixed is not None:
        kw["mixed"] = mixed
    if static_worlds:
        kw["static_worlds"] = True
    if waves is not None:
        kw["waves"] = waves
    seq = ext is not None and ext.dim() == 3         # a torque SEQUENCE (one row per step): chunked launches take their rows
    for _ in range(n_episodes):
        q.copy_(q0); dq.copy_(dq0)
        if cf is not None:
            cf.zero_()
        if cost is not None:
            cost["out"].zero_()
        k0 = 0
        for c in chunks:
            if timed:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
            bw.step(q, dq, dt, c, cforce=cf, split=split or False, ext_gforce=(ext[k0:k0 + c] if seq else ext), **kw)
            k0 += c
            if timed:
                b.record()
                ev.append((a, b))
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t0
    ms = [a.elapsed_time(b) for a, b in ev]
    return wall, ms, (q, dq)


def build_model(cfg):
    from arboris_python_amd import scenes
    if cfg["model"] == "snake64":
        return scenes.flat(scenes.snake_world(64))
    if cfg["model"] in ("snake100", "snake128"):
        return scenes.flat(scenes.snake_world(int(cfg["model"][5:])))
    if cfg["model"] == "human36_objects":
        return scenes.flat(scenes.human36_and_objects_world(cfg.get("objects", 4)))
    if cfg["model"] == "human36_balls":
        return scenes.flat(scenes.human36_and_balls_world(cfg.get("objects", 3)))
    return scenes.flat(scenes.human36_world(cfg["contacts"], pd=bool(cfg.get("pd"))))


def make_states(cfg, model, lo, hi, seed):
    """Worlds [lo, hi) of the config's seeded global batch (world w has its own stream, seed = (seed, w))."""
    from arboris_python_amd import synth
    if cfg["model"] in ("human36_objects", "human36_balls"):
        # the scene as built (the human standing, the boxes beside it a centimetre above the floor), every world with its own
        # small velocities (world w draws from its own stream)
        import numpy as np
        from arboris_python_amd import scenes
        from arboris_python_amd.flatten import flatten_world
        _, q0, dq0 = flatten_world(scenes.human36_and_balls_world(cfg.get("objects", 3)) if cfg["model"] == "human36_balls"
                                   else scenes.human36_and_objects_world(cfg.get("objects", 4)))
        q = np.tile(q0, (hi - lo, 1))
        dq = np.stack([dq0 + np.random.default_rng([seed, w]).uniform(-0.1, 0.1, size=len(dq0)) for w in range(lo, hi)])
        return q, dq
    if cfg["states"] == "standing":
        # config 3/5 distribution: standing pose dropped from U(0, 3 cm), small velocities
        return synth.world_states(model, range(lo, hi), "standing", seed, drop=0.03, vel=0.1)
    if cfg["model"] in ("snake64", "snake100", "snake128"):
        return synth.world_states(model, range(lo, hi), "random", seed, angle=0.5, vel=1.0)
    # config 2: random poses, hinge angles U(-0.7, 0.7) rad, velocities U(-1, 1).  (More energetic draws -- angle 1,
    # velocities 3, the generator's defaults -- send the reference's own time stepping beyond 100 rad/s within 40 steps
    # for 80 % of the worlds, in the float64 oracle as on the device, tools/config2_finite.py: the step's cost does not
    # depend on the data, but a benchmark should not integrate garbage.)
    return synth.world_states(model, range(lo, hi), "random", seed, angle=0.7, vel=1.0)


def make_torques(model, lo, hi, seed, steps=None):
    """Per-rollout user torques of the MPC shape for rollouts [lo, hi): U(-0.05, 0.05) N m on every joint dof, none on
    the floating base (the distal bodies of human36 are light); rollout w draws from its own stream (seed, w).
    `steps`: a torque SEQUENCE (steps, rollouts, ndof) -- a control input per step of the horizon, as an MPC rollout has
    (the reference polls its controllers every step, core.py:811-817): amplitude, frequency and phase per rollout and dof."""
    import numpy as np
    if steps is None:
        tau = np.stack([np.random.default_rng([seed, w]).uniform(-0.05, 0.05, size=model.ndof) for w in range(lo, hi)])
        tau[:, :6] = 0.
        return tau
    seqs = []
    for w in range(lo, hi):
        rng = np.random.default_rng([seed, w])
        a, om, ph = rng.uniform(-0.05, 0.05, model.ndof), rng.uniform(0.2, 1.0, model.ndof), rng.uniform(0., 6.283, model.ndof)
        seqs.append(a * np.sin(om * np.arange(steps)[:, None] + ph))
    tau = np.stack(seqs, axis=1)
    tau[:, :, :6] = 0.
    return tau


def resolve_config(args):
    cfg = dict(MPC if (args.mpc and args.config == 5) else CONFIGS[args.config])
    if args.batch is not None:
        cfg["batch"] = args.batch
    if args.contacts is not None and cfg["model"] == "human36":
        if args.contacts != cfg["contacts"]:
            cfg["name"] += " -- overridden: %d floor contacts" % args.contacts
        cfg["contacts"] = args.contacts
        cfg["states"] = "standing" if args.contacts else "random"
    if args.dtype is not None:
        cfg["dtype"] = args.dtype
    if args.dt is not None:
        cfg["dt"] = args.dt
    return cfg


def dry_run(args, cfg):
    """Launcher / sharding rehearsal without a GPU: every rank joins a gloo group, takes its range of
    the global batch from dist.shard_bounds, and the final gather runs on CPU stand-in tensors."""
    import torch
    import torch.distributed as dist
    from arboris_python_amd.dist import shard_bounds, gather_state
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
    ws = dist.get_world_size() if world > 1 else 1
    B = cfg["batch"]
    G = args.global_batch if args.global_batch is not None else ws * B
    a, b = shard_bounds(G, rank, ws)
    shards = [None] * ws
    if world > 1:
        from arboris_python_amd.dist import gather_rows
        dist.all_gather_object(shards, (a, b))
        q_loc = torch.arange(a, b, dtype=torch.float64).reshape(-1, 1).repeat(1, 3)
        dq_loc = -torch.arange(a, b, dtype=torch.float64).reshape(-1, 1).repeat(1, 2)
        q_all, dq_all = gather_state(q_loc, dq_loc, G, dist)
        cost = gather_rows(0.5 * torch.arange(a, b, dtype=torch.float64), G, dist)      # per-rollout costs: (shard,) -> (G,)
        ok = bool(torch.equal(q_all[:, 0], torch.arange(G, dtype=torch.float64))
                  and torch.equal(dq_all[:, 0], -torch.arange(G, dtype=torch.float64))
                  and torch.equal(cost, 0.5 * torch.arange(G, dtype=torch.float64)))
    else:
        shards, ok = [(a, b)], True
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": ws, "requested_gpus": args.gpus, "worlds_per_gpu": B,
                          "global_batch": G, "shards": [list(s) for s in shards], "gather_ok": ok,
                          "config": args.config}))
    if world > 1:
        dist.destroy_process_group()


def chain_roof(bw, model, cfg, q0, dq0, torch, np, plan, value):
    """The LATENCY roof of the dominant kernel, measured live (round 6; tools/chain_probe.py is the long form).  One world is
    one wavefront and a world's step is a chain of dependent instructions: alone on its SIMD (4 worlds per CU, the LDS padded so
    that no second wavefront fits a SIMD) a wavefront of the headline's build advances one step in T1; with `wave_slots`
    wavefronts resident the chip cannot pass wave_slots / T1 world-steps/s however well they interleave."""
    cus = torch.cuda.get_device_properties(bw.device).multi_processor_count
    waves = plan["waves_per_simd"] if plan["waves_per_simd"] in (2, 3) else None
    lds0 = bw.plan(4 * cus, cfg["episode"], dtype=q0.dtype, waves=waves, static_worlds=True)["lds_bytes"]
    pad = max(0, 32 * 1280 - lds0 - 8)            # 32 of a CU's 128 LDS granules per wavefront: four wavefronts per CU
    n1 = 4 * cus
    reps = -(-n1 // q0.shape[0])
    qa, da = q0.repeat(reps, 1)[:n1].contiguous(), dq0.repeat(reps, 1)[:n1].contiguous()
    bw.set_knob("lds_pad", pad)
    try:
        run_episodes(bw, qa, da, cfg["dt"], cfg["episode"], 2, torch, timed=False, static_worlds=True, waves=waves)
        _, ms, _ = run_episodes(bw, qa, da, cfg["dt"], cfg["episode"], 8, torch, static_worlds=True, waves=waves)
    finally:
        bw.set_knob("lds_pad", 0)
    t1 = float(np.min(ms)) * 1e-3 / cfg["episode"]
    slots = plan["wave_slots"]
    return {"lone_wave_us_per_step": t1 * 1e6, "lone_wave_worlds": n1, "wave_slots": slots, "waves_per_simd": plan["waves_per_simd"],
            "latency_roof": slots / t1, "unit": "world-steps/s", "achieved": value, "frac": value / (slots / t1),
            "note": "T1 = one step of a wavefront that has its SIMD to itself (%d worlds = 4 per CU, LDS padded by %d B, the "
                    "headline's kernel build, %d-step episodes); latency_roof = wave_slots / T1: what the chip reaches when "
                    "its resident wavefronts -- dependent chains -- interleave perfectly.  The per-phase lone-wave cycles "
                    "and the intermediate occupancies are in profiles/r06_chain.json (tools/chain_probe.py, "
                    "tools/subphase_probe.py); the pipe roof (instruction mix x pipe rates) is `valu`." % (n1, pad, cfg["episode"])}


def timed_leg(BatchedWorlds, torch, np, local_rank, cfg, min_seconds, seed=1000, general=False, min_launches=10, body_columns=False,
              classic=False):
    """One more workload timed like the headline (whole episodes, one launch per episode, states resident in HBM, at least
    `min_seconds` and `min_launches` launches): world-steps/s, the launch durations from HIP events on the launch stream,
    the build.  cfg["torques"]: True = one torque row per rollout, "sequence" = a torque row per step and rollout
    (arb_step_args.ext_gforce_steps) plus the per-rollout running cost (arb_step_cost), as an MPC horizon has."""
    mdl = build_model(cfg)
    b2 = BatchedWorlds(mdl, local_rank)
    dt2 = torch.float32 if cfg["dtype"] == "f32" else torch.float64
    qa, da = make_states(cfg, mdl, 0, cfg["batch"], seed=seed)
    ta, tb = b2.to_device(qa, da, dt2)
    ex2, cost = None, None
    if cfg.get("torques") == "sequence":
        ex2 = torch.as_tensor(make_torques(mdl, 0, cfg["batch"], seed=2000, steps=cfg["episode"]), dtype=dt2, device=b2.device).contiguous()
        ones = torch.ones(mdl.ndof, dtype=dt2, device=b2.device)
        cost = dict(out=torch.zeros(cfg["batch"], dtype=dt2, device=b2.device), w_q=ones, w_dq=0.01 * ones, w_tau=ones.clone())
    elif cfg.get("torques"):
        ex2 = torch.as_tensor(make_torques(mdl, 0, cfg["batch"], seed=2000), dtype=dt2, device=b2.device).contiguous()
    kw = dict(ext=ex2, general=general, cost=cost, body_columns=body_columns, mixed=cfg.get("mixed"), classic=classic)
    run_episodes(b2, ta, tb, cfg["dt"], cfg["episode"], 2, torch, timed=False, **kw)
    cal, _, _ = run_episodes(b2, ta, tb, cfg["dt"], cfg["episode"], 2, torch, timed=False, **kw)
    n_ep = max(min_launches, int(np.ceil(min_seconds / max(cal / 2, 1e-6))))
    wl, me, (qe, dqe) = run_episodes(b2, ta, tb, cfg["dt"], cfg["episode"], n_ep, torch, **kw)
    out = {"workload": cfg["name"] + ", batch %d, %s" % (cfg["batch"], cfg["dtype"]),
           "value": cfg["batch"] * n_ep * cfg["episode"] / wl, "unit": "world-steps/s",
           "kernel_ms": float(np.mean(me)), "episodes": n_ep, "steps_per_launch": cfg["episode"], "timed_region_s": wl,
           "finite": bool(torch.isfinite(qe).all() and torch.isfinite(dqe).all()),
           "kernel_build": b2.plan(cfg["batch"], cfg["episode"], dtype=dt2, ext_gforce=bool(cfg.get("torques")), general_kernels=general,
                                   body_columns=body_columns, cost=cost is not None, mixed=cfg.get("mixed"), classic_columns=classic),
           "model_info": {k: b2.info[k] for k in ("ndof", "nc", "wide", "mixed_default")}}
    if cost is not None:
        out["cost_finite"] = bool(torch.isfinite(cost["out"]).all())
        out["mean_cost_per_rollout"] = float(cost["out"].double().mean())
    b2.close()
    return out


def main():
    args = parse()
    cfg = resolve_config(args)
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and env_world is None:
        # No launcher around us: start one rank per GPU as child processes.  This process has not
        # imported torch nor made any HIP call, and it never does; it only waits for the children.
        cmd = launcher_command(args.gpus, sys.argv[1:])
        sys.stderr.write("[bench] starting %d ranks: %s\n" % (args.gpus, " ".join(cmd)))
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.exit(subprocess.run(cmd, env=env).returncode)
    world = int(env_world or "1")
    if world != args.gpus:
        sys.stderr.write("[bench] --gpus %d but WORLD_SIZE=%d: refusing to report a mislabelled line\n"
                         % (args.gpus, world))
        sys.exit(2)
    if args.dry_run:
        return dry_run(args, cfg)

    # stdout carries the ONE JSON line and nothing else: whatever libraries print there while the job runs (RCCL
    # prints its version banner on stdout at communicator creation) goes to stderr instead
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import numpy as np
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1 or "RANK" in os.environ:          # under a launcher (torchrun sets RANK), also with a single rank
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist_mod.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        dist = dist_mod
    n_gpus = dist.get_world_size() if dist is not None else 1        # the ranks RCCL saw
    from arboris_python_amd.batch import BatchedWorlds
    from arboris_python_amd.dist import shard_bounds
    dtype = torch.float32 if cfg["dtype"] == "f32" else torch.float64

    model = build_model(cfg)
    bw = BatchedWorlds(model, local_rank)
    B = cfg["batch"]
    lo, hi = shard_bounds(n_gpus * B, rank, n_gpus)                  # this rank's worlds of the global batch
    # ONE seeded global batch in which world w has its own random stream: rank r of N steps worlds [lo, hi) of it,
    # and world w is the same world in an N = 1 and in an N = 8 run
    q, dq = make_states(cfg, model, lo, hi, seed=1000)
    q0, dq0 = bw.to_device(q, dq, dtype)
    dt, EP = cfg["dt"], cfg["episode"]
    ext = None
    if cfg.get("torques"):
        ext = torch.as_tensor(make_torques(model, lo, hi, seed=2000, steps=EP if cfg["torques"] == "sequence" else None),
                              dtype=dtype, device=bw.device).contiguous()

    # ---- warmup: W steps rounded up to whole episodes (untimed), then one calibration episode -------------
    warm_eps = max(1, -(-args.warmup // EP))
    run_episodes(bw, q0, dq0, dt, EP, warm_eps, torch, dist, split=args.split, timed=False, ext=ext)
    cal, _, _ = run_episodes(bw, q0, dq0, dt, EP, 2, torch, dist, split=args.split, timed=False, ext=ext)
    n_ep = max(-(-args.steps // EP), 50, int(np.ceil(args.min_seconds / max(cal / 2, 1e-6))))
    t = torch.tensor([n_ep], dtype=torch.int64, device=bw.device)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)                      # every rank times the same number of episodes
    n_ep = int(t.item())

    # ---- the timed region: n_ep whole episodes between barrier + synchronize brackets ----------------------
    wall, ms, (qf, dqf) = run_episodes(bw, q0, dq0, dt, EP, n_ep, torch, dist, split=args.split, ext=ext)
    t = torch.tensor([wall], dtype=torch.float64, device=bw.device)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall = float(t.item())
    finite = bool(torch.isfinite(qf).all() and torch.isfinite(dqf).all())
    steps_timed = n_ep * EP
    launches_per_episode = len(ms) // n_ep
    # which build of the step kernel the library picked for this launch shape (arb_step_plan)
    kernel_build = bw.plan(B, EP // launches_per_episode, dtype=dtype, ext_gforce=bool(cfg.get("torques")), split=args.split or False)
    ep_ms = np.asarray(ms).reshape(n_ep, launches_per_episode).sum(axis=1)       # kernel time per episode

    # final state gather over RCCL/xGMI (outside the timed region, reported separately)
    gather_ms = None
    if dist is not None:
        from arboris_python_amd.dist import gather_state
        torch.cuda.synchronize()
        g0 = time.perf_counter()
        q_all, dq_all = gather_state(qf, dqf, n_gpus * B, dist)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - g0) * 1e3
        assert q_all.shape[0] == n_gpus * B

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    elem = 4 if cfg["dtype"] == "f32" else 8
    bytes_per_world_step = 2 * (model.nq + model.ndof) * elem           # state in + state out
    if model.nc:
        bytes_per_world_step += 2 * model.nc * 4 * elem                  # cforce in + out
    if ext is not None:
        bytes_per_world_step += model.ndof * elem * (EP if ext.dim() == 3 else 1)   # the rollout's torques: one row per launch, or per step
    value = n_gpus * B * steps_timed / wall
    kern_ms = float(np.mean(ep_ms))
    # one launch reads and writes the state once, whatever the number of steps it advances on chip
    achieved_gbs = bytes_per_world_step * B / (kern_ms * 1e-3) / 1e9
    flop_dense = {0: 1.66e6, 4: 1.9e6, 8: 2.13e6}.get(cfg["contacts"], 1.66e6) if cfg["model"] == "human36" else 1.23e7
    res = {
        "metric": "world-steps/sec at batch, human36 ~40-DOF + 4 contacts, 1/2/4/8 MI355X",
        "value": value, "unit": "world-steps/s", "n_gpus": n_gpus, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": wall / steps_timed * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": cfg["dtype"], "data": "synthetic",
        "steps_timed": steps_timed, "episodes": n_ep, "episode_steps": EP, "timed_region_s": wall,
        "warmup_steps_run": (warm_eps + 2) * EP,
        "steps_note": "a step's cost depends on its place in the %d-step episode (free fall, impact, sliding), so "
                      "--steps is rounded up to whole episodes and episodes are repeated for >= %.1f s and >= 50 "
                      "launches; value = worlds x steps_timed / timed_region_s" % (EP, args.min_seconds),
        "episode_kernel_ms": {"median": float(np.median(ep_ms)), "mean": kern_ms, "p10": float(np.percentile(ep_ms, 10)),
                              "p90": float(np.percentile(ep_ms, 90)), "min": float(ep_ms.min()), "max": float(ep_ms.max())},
        "config": {"workload": "%s, batch %d worlds/GPU, dt=%g, %s, whole %d-step episodes, one arb_step launch per episode%s"
                               % (cfg["name"], B, dt, cfg["dtype"], EP, (", Gauss-Seidel in a %s-per-world kernel" % args.split) if args.split else ""),
                   "baseline_config": args.config, "worlds_per_gpu": B, "global_batch": n_gpus * B,
                   "parallelism": "dp%d" % n_gpus, "steps_per_launch": EP / launches_per_episode,
                   "kernel_build": kernel_build,
                   "launch": "arb_step default: with more worlds than resident wavefronts the kernel draws (4-step chunk, "
                             "world) work items from a device-side queue (include/arbstep.h, ARB_STEP_STATIC_WORLDS turns it off)"},
        "roofline": {"bound": "valu-issue",
                     "bound_note": "north_star asks for the HBM fraction, which achieved/peak/frac report; compulsory "
                                   "traffic is the state in+out once per episode launch, so the path is bound by the "
                                   "per-wave VALU issue/latency chain (see `valu`), not by HBM",
                     "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": None,
                     "kernel": "arb_step_kernel", "kernel_ms": kern_ms,
                     "algorithmic_bytes_per_launch": bytes_per_world_step * B,
                     "algorithmic_bytes_per_world_step": bytes_per_world_step / float(EP),
                     "fp32_vector_context": {"dense_equiv_flop_per_world_step": flop_dense,
                                             "achieved_tflops": flop_dense * B * EP / (kern_ms * 1e-3) / 1e12,
                                             "peak_tflops": FP32_PEAK_TFLOPS,
                                             "note": "dense-as-written FLOP count of the reference (SURVEY 6), not the "
                                                     "instructions the kernel executes: context only"}},
        "state_finite": finite,
    }
    # HBM traffic + issue counters of the dominant kernel from the committed rocprofv3 PMC passes (tools/profile_gpu.sh,
    # profiles/): attached only when the profiled launch shape is the one timed here
    import glob
    profs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_summary.json")))   # newest round last by name
    profs = [f for f in profs if "baseline" not in f]
    prof = profs[-1] if profs else ""
    shape = {"config": args.config, "batch": B, "dtype": cfg["dtype"], "contacts": cfg["contacts"],
             "steps_per_launch": EP, "split": args.split or False}
    if cfg.get("torques"):
        shape["mpc"] = True
    if os.path.exists(prof) and launches_per_episode == 1:
        try:
            pj = json.load(open(prof))
            pshape = pj.get("launch_shape", {"config": 3, "batch": 4096, "dtype": "f32", "contacts": 4,
                                             "steps_per_launch": 40, "split": False})
            if pshape == shape:
                pm = pj["pmc_per_launch"]
                res["roofline"]["traffic"] = (pm["FETCH_SIZE"]["mean_per_launch"] + pm["WRITE_SIZE"]["mean_per_launch"]) * 1024.
                # hand-overs of a world's state between wavefronts in the work queue (csrc launch_one: ARB_QUEUE_CHUNK /
                # ARB_QUEUE_TAIL defaults 4 / 4): chunks of 4 steps, then the last 4 steps one by one
                chunk_, tail_ = 4, min(6, EP - 1)          # (the library's work-item sizes: csrc Knobs)
                items_ = (-(-(EP - tail_) // chunk_) + tail_) if chunk_ > 0 else 1
                handover = items_ * bytes_per_world_step * B
                res["roofline"]["traffic_note"] = ("bytes per launch, FETCH_SIZE+WRITE_SIZE from separate rocprofv3 --pmc passes "
                                                   "(profiles/%s); 4 B/lane accesses, reported uncorrected.  With the work queue the "
                                                   "state of every world is handed from wavefront to wavefront %d times per "
                                                   "%d-step episode (%d x %d B x worlds = %.1f MB of the figure)."
                                                   % (os.path.basename(prof), items_, EP, items_, bytes_per_world_step, handover / 1e6))
                if res["roofline"]["traffic"] > 20 * handover:
                    # where the bytes go (round 5): the scratch instructions of the profiled launch, 256 B per wave-instruction
                    note = ("  The rest is SCRATCH traffic of the kernel build compiled for three waves per SIMD (168 VGPRs; the "
                            "register table of DESIGN.md 3 is generated from the shipped library)")
                    if "SQ_INSTS_VMEM_WR" in pm and "SQ_INSTS_VMEM_RD" in pm:
                        wr, rd = pm["SQ_INSTS_VMEM_WR"]["mean_per_launch"], pm["SQ_INSTS_VMEM_RD"]["mean_per_launch"]
                        note += (": %.0f spill stores and %.0f reloads per world-step (vector-memory wave-instructions of 256 B; the "
                                 "state's own loads and stores are 10 of them) = %.1f GB stored, %.1f GB loaded per launch at the "
                                 "L1; WRITE_SIZE counts %.1f GB (stores are written through), FETCH_SIZE %.1f GB (%.0f %% of the "
                                 "reloads are served by the L2)"
                                 % (wr / (B * EP), rd / (B * EP), wr * 256 / 1e9, rd * 256 / 1e9, pm["WRITE_SIZE"]["mean_per_launch"] * 1024 / 1e9,
                                    pm["FETCH_SIZE"]["mean_per_launch"] * 1024 / 1e9, 100. * (1. - pm["FETCH_SIZE"]["mean_per_launch"] * 1024 / max(rd * 256., 1.))))
                    note += (".  FETCH_SIZE / WRITE_SIZE are the L2's fabric-side request counters: they include Infinity-Cache hits "
                             "(MI355X_MICROARCH.md), and the spill footprint of the whole chip -- 3072 wavefronts x 64 lanes x 512 B = "
                             "100 MB -- fits the 256 MiB Infinity Cache, so how much of the %.0f GB/s reaches HBM is NOT known from these "
                             "counters (no HBM-side counter is exposed to rocprofv3 on this pool); at most %.0f %% of the 8 TB/s roof "
                             "either way.  The two-wave build (ARB_STEP_WAVES2, no spills) moves %.1f MB per launch."
                             % (res["roofline"]["traffic"] / (kern_ms * 1e-3) / 1e9, 100. * res["roofline"]["traffic"] / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                (handover + bytes_per_world_step * B) / 1e6))
                    res["roofline"]["traffic_note"] += note
                ws = float(B * EP)
                simds, clk = 1024, 2.4e9                     # 256 CUs x 4 SIMD-32; peak shader clock
                valu = {
                    "valu_insts_per_world_step": pm["SQ_INSTS_VALU"]["mean_per_launch"] / ws,
                    "salu_insts_per_world_step": pm["SQ_INSTS_SALU"]["mean_per_launch"] / ws,
                    "lds_insts_per_world_step": pm["SQ_INSTS_LDS"]["mean_per_launch"] / ws,
                    "wave_cycles_per_world_step": 4. * pm["SQ_WAVE_CYCLES"]["mean_per_launch"] / ws,
                    "wave_frac_issuing_valu": pm["SQ_ACTIVE_INST_VALU"]["mean_per_launch"] / pm["SQ_WAVE_CYCLES"]["mean_per_launch"],
                    "wave_frac_waiting": (pm["SQ_WAIT_ANY"]["mean_per_launch"] + pm["SQ_WAIT_INST_ANY"]["mean_per_launch"])
                                         / pm["SQ_WAVE_CYCLES"]["mean_per_launch"],
                    # The fraction that says something about this kernel.  CDNA4 SIMDs are SIMD-32: a wave64 VALU
                    # instruction occupies the pipe for 2 cycles (MI355X_MICROARCH.md; 4 is what ONE wave alone
                    # sustains), so the issue roof is 1024 SIMDs x 2.4 GHz / 2 wave-instructions per second.
                    "valu_issue_frac": 2. * pm["SQ_INSTS_VALU"]["mean_per_launch"] / (kern_ms * 1e-3 * simds * clk),
                    "valu_issue_frac_note": "SQ_INSTS_VALU x 2 cycles (wave64 on SIMD-32) / (kernel time x 1024 SIMDs x 2.4 GHz): "
                                            "the profiled launch's wave-instructions against THIS run's kernel time -- every "
                                            "instruction priced at the float32 datasheet rate; see valu_pipe_frac for the mix",
                    "note": "SQ counters of profiles/%s" % os.path.basename(prof)}
                mix = pj.get("valu_mix_per_launch")
                if mix:
                    # Pipe-weighted (round 4): the dynamic instruction mix of the profiled launch (SQ_INSTS_VALU_{ADD,MUL,FMA,
                    # TRANS}_F{32,64}) priced with the rates this machine sustains (tools/exec_mask_probe.hip: 2.4 cycles per
                    # wave64 float32 instruction, 4.3 per float64 one, two or more independent waves per SIMD)
                    valu["valu_mix_per_world_step"] = {k: mix[k] / ws for k in ("float32", "float64", "other", "total")}
                    valu["valu_pipe_frac"] = mix["pipe_cycles"] / (kern_ms * 1e-3 * simds * clk)
                    valu["valu_pipe_frac_note"] = ("(2.4 x (float32 + other) + 4.3 x float64 instructions) / (kernel time x 1024 SIMDs x "
                                                   "2.4 GHz): the share of the vector pipes' time the launch's instructions occupy. "
                                                   "'other' = moves, selects, compares, integer, lane exchanges, priced like float32 "
                                                   "(DPP operands cost 4.2: a lower bound); the shader clock under this load is below "
                                                   "2.4 GHz (2.0-2.2 measured), so the pipes are busier than the figure says.  The "
                                                   "rest of the time every resident wave of a SIMD waits: one wave alone issues an "
                                                   "instruction every 4.7 cycles at best, a dependent one every 8")
                if "valu_lane_utilisation" in pj:
                    # active lanes per executed VALU instruction: thread-cycles over 64 x instruction-cycles, one pass
                    valu["valu_lane_utilisation"] = pj["valu_lane_utilisation"]
                    valu["valu_lane_utilisation_note"] = ("SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU): the share of the 64 "
                                                          "lanes enabled (EXEC) when a VALU instruction executes; lanes that run "
                                                          "along on don't-care data (the quads beside a constraint's own quad in "
                                                          "the Gauss-Seidel sweeps) count as enabled")
                res["roofline"]["valu"] = valu
                if "SQ_INSTS_MFMA" in pm:
                    res["roofline"]["valu"]["mfma_insts_per_world_step"] = pm["SQ_INSTS_MFMA"]["mean_per_launch"] / ws
                    res["roofline"]["valu"]["mfma_note"] = ("v_mfma_f32_4x4x1_16b_f32 in the constraint-space products of "
                                                            "phase D; the elimination of phase C runs on the vector ALU "
                                                            "(the matrix-core variant measured slower, DESIGN.md 3)")
        except Exception:
            pass
    # (not under the profiling runs of tools/profile_gpu.sh: the lone-wave launches would mix into the per-kernel statistics)
    if (n_gpus == 1 and launches_per_episode == 1 and not args.split and cfg["dtype"] == "f32" and not args.no_per_step_leg
            and os.environ.get("ARB_BENCH_LEGS") != "perstep"):
        try:
            res["roofline"]["chain"] = chain_roof(bw, model, cfg, q0, dq0, torch, np, kernel_build, value)
        except Exception as e:                        # (never lose the line to a diagnostic leg)
            res["roofline"]["chain"] = {"error": repr(e)}
    if gather_ms is not None:
        res["final_state_allgather_ms"] = gather_ms
    if n_gpus == 1 and not args.no_per_step_leg:
        # the same workload with one launch per step (a non-uniform timeline, or observers between steps)
        w1, m1, _ = run_episodes(bw, q0, dq0, dt, EP, 5, torch, None, spl=1, split=args.split, ext=ext)
        res["per_step_launch"] = {"value": B * 5 * EP / w1, "unit": "world-steps/s", "kernel_ms": float(np.mean(m1)),
                                  "steps": 5 * EP}
    if n_gpus == 1 and not args.no_per_step_leg and args.config == 3 and not args.split and os.environ.get("ARB_BENCH_LEGS") != "perstep":
        # (same switch as the one-launch-per-step leg: the profiling runs skip all of these)
        # The path that meets 1e-5 on EVERY world-step: the float64 kernels on the headline workload (the float32 kernels
        # meet it on 99.9 % of the world-steps, the others are decisions that are marginal for the reference itself or
        # that the float32 elimination moves: profiles/r04_replay_stats.txt)
        c64 = dict(cfg, dtype="f64")
        res["strict_f64"] = timed_leg(BatchedWorlds, torch, np, local_rank, c64, 1.0)
        res["strict_f64"]["tolerance"] = "1e-7 against the float64 reference on every world-step (tests: 1e-9 without contacts)"
        # the reference's own falling scenario has 8 contact points (tests/test_human36_falling.py:32)
        c8 = dict(cfg, contacts=8, name=cfg["name"].replace("4 floor", "8 floor") + " -- the reference's own 8 contact points")
        res["contacts8"] = timed_leg(BatchedWorlds, torch, np, local_rank, c8, 1.0)
        # The literal MPC shape of BASELINE config 5 -- 2048 rollouts x 32 in-kernel steps, per-rollout torques -- on ONE
        # GPU: 2048 rollouts are fewer than the 3072 wave slots of one MI355X, every rollout has a wavefront to itself
        # from the first to the last step, and the horizon costs the latency of one world's 32 steps.  Splitting the
        # 2048 rollouts over 8 GPUs (256 each, `--config 5 --mpc`) takes just as long per horizon: the shape is latency
        # bound, not throughput bound, and one GPU is the right place for it.
        # Round 5: every rollout applies a torque SEQUENCE -- one row per step, read inside the launch
        # (arb_step_args.ext_gforce_steps; the reference polls its controllers every step, core.py:811-817) -- and returns
        # its running cost (arb_step_cost): what an MPC horizon is.  (Until round 4 the leg held one torque per rollout.)
        cm = dict(MPC, batch=2048, torques="sequence",
                  name=MPC["name"].replace("256 rollouts/GPU = 2048 on 8 GPUs", "all 2048 rollouts on ONE GPU")
                                  .replace("per-rollout torques", "a torque sequence per rollout (one row per step) + per-rollout cost"))
        res["mpc_2048_rollouts_one_gpu"] = timed_leg(BatchedWorlds, torch, np, local_rank, cm, 0.5)
        mp_ = res["mpc_2048_rollouts_one_gpu"]
        mp_["ms_per_horizon"] = mp_["kernel_ms"]
        mp_["algorithmic_bytes_per_world_step"] = (2 * (model.nq + model.ndof) * 4 + 2 * model.nc * 4 * 4 + 4) / 32. + model.ndof * 4
        mp_["algorithmic_bytes_note"] = ("state + contact forces in and out and the cost once per horizon, plus this step's "
                                         "torque row: %d B per world-step" % (model.ndof * 4))
        # the general kernels on the headline workload (classical constraint columns; a human36 outside the model classes gets these)
        res["general_kernel"] = timed_leg(BatchedWorlds, torch, np, local_rank, cfg, 0.7, general=True)
        # Round 6: the headline runs BODY-SPACE constraint columns (the default for every model of the class since this round,
        # decided on 119 808 replayed world-steps per path, profiles/r06_replay_stats.txt).  The classical columns on request
        # (ARB_STEP_CLASSIC_COLUMNS: the kernels specialised for four contacts, rounds 4-5's headline path) are 1.7 % faster
        # and have twice the float32 world-steps beyond 1e-5, among them the only ones the device's own arithmetic causes
        res["classic_columns_f32"] = timed_leg(BatchedWorlds, torch, np, local_rank, cfg, 0.5, classic=True)
        res["classic_columns_f32"]["accuracy"] = ("world-steps beyond 1e-5 of the float64 reference per 39 936 replayed: 43-57 (criteria d + e, "
                                                  "caused by the device's float32 system: 4-11); the default (body-space columns): 21-28 (d + e: 0); "
                                                  "strict_f64: none")
        # the other BASELINE configs and the throughput regime, under the same driver clock (short legs)
        cfgs = {}
        for key, c_ in (("config2", CONFIGS[2]), ("config4", CONFIGS[4]), ("config5", CONFIGS[5]),
                        ("batch65536", dict(cfg, batch=65536, name=cfg["name"] + " -- 65 536 worlds on ONE GPU (the throughput regime)"))):
            cfgs[key] = timed_leg(BatchedWorlds, torch, np, local_rank, dict(c_), 0.5, seed=1000 if key == "batch65536" else 0,
                                  min_launches=5 if key == "batch65536" else 10)
        # BASELINE config 4's model through float32 buffers (round 6): by default such a launch is promoted to the float64
        # kernels (config4 above is that path's speed); the mixed build on request -- float32 state and LDS, float64
        # elimination and right-hand side: 7e-5 against the reference instead of 1e-5 -- has more wave slots
        c4m = dict(CONFIGS[4], dtype="f32", mixed=True, name=CONFIGS[4]["name"] + " -- float32 buffers, the mixed build (ARB_STEP_MIXED)")
        cfgs["config4_mixed_f32"] = timed_leg(BatchedWorlds, torch, np, local_rank, c4m, 0.5, seed=0)
        cfgs["config4_mixed_f32"]["accuracy"] = "dq+ within 7e-5 of the float64 reference (median 8e-6); float64 / promoted: 1e-5 (8e-6 against the oracle's explicit inverse)"
        res["configs"] = cfgs
        # worlds PAST one wavefront (round 6): the wide kernels, one workgroup per world, float64 arithmetic
        wide = {}
        for key, c_ in (("snake100", dict(model="snake100", contacts=0, batch=1024, dtype="f64", dt=1e-3, episode=16, states="random",
                                          name="snake-100 (100 Rz joints): past the 64 lanes of a wavefront, the wide kernels")),
                        ("snake128", dict(model="snake128", contacts=0, batch=1024, dtype="f64", dt=1e-3, episode=16, states="random",
                                          name="snake-128: 129 columns, the compact wide build with four columns per lane")),
                        ("human36_and_4_objects", dict(model="human36_objects", contacts=8, batch=1024, dtype="f32", dt=5e-3, episode=40,
                                                       states="standing", name="human36 on four floor contacts beside four free boxes "
                                                                               "with a ball each on the floor: 66 dofs, 8 contacts, the wide kernels")),
                        ("human36_and_12_objects", dict(model="human36_objects", objects=12, contacts=16, batch=1024, dtype="f32", dt=5e-3,
                                                        episode=40, states="standing", name="human36 beside twelve free boxes: 114 dofs, "
                                                                                            "16 contacts (179 columns), the wide kernels")),
                        ("human36_and_3_balls_all_pairs", dict(model="human36_balls", objects=3, contacts=38, batch=512, dtype="f64", dt=5e-3,
                                                               episode=40, states="standing", name="human36 beside three free balls with EVERY "
                                                               "pair of get_all_contacts registered: 60 dofs, 38 contacts (10-15 active per step), "
                                                               "the wide kernels' active-constraint slots"))):
            wide[key] = timed_leg(BatchedWorlds, torch, np, local_rank, c_, 0.4, seed=0, min_launches=3)
        res["wide_worlds"] = wide
        # human36 OUTSIDE the model class of the specialised headline kernels (round 4 review: what does a caller get whose
        # model differs slightly?): six floor contacts (three per foot: body-space columns, one column set), and the headline
        # model with a PD controller on its 36 hinge dofs (the general kernels)
        mc = {}
        for key, c_ in (("contacts6", dict(cfg, contacts=6, name=cfg["name"].replace("4 floor", "6 floor"))),
                        ("pd_controlled", dict(cfg, pd=True, name=cfg["name"] + " + a PD controller on the 36 hinge dofs"))):
            mc[key] = timed_leg(BatchedWorlds, torch, np, local_rank, c_, 0.4)
        res["model_classes"] = mc
    if n_gpus == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(model, q, dq, dt, args.cpu_seconds, EP)
        res["cpu_baseline"]["host_cores_available"] = os.cpu_count()
    if args.extra and n_gpus == 1:
        extra = {}
        for cid in (2, 4, 5, "5mpc"):
            c2 = dict(MPC if cid == "5mpc" else CONFIGS[cid])
            mdl = build_model(c2)
            b2 = BatchedWorlds(mdl, local_rank)
            qa, da = make_states(c2, mdl, 0, c2["batch"], seed=0)
            dt2 = torch.float32 if c2["dtype"] == "f32" else torch.float64
            ta, tb = b2.to_device(qa, da, dt2)
            ex2 = None
            if c2.get("torques"):
                ex2 = torch.as_tensor(make_torques(mdl, 0, c2["batch"], seed=2000, steps=c2["episode"] if c2["torques"] == "sequence" else None),
                                      dtype=dt2, device=b2.device).contiguous()
            run_episodes(b2, ta, tb, c2["dt"], c2["episode"], 2, torch, timed=False, ext=ex2)
            wl, me, (qe, _) = run_episodes(b2, ta, tb, c2["dt"], c2["episode"], 20, torch, ext=ex2)
            extra["config%s" % cid] = {"workload": c2["name"], "world_steps_per_s": c2["batch"] * 20 * c2["episode"] / wl,
                                       "episode_kernel_ms_median": float(np.median(me)),
                                       "finite": bool(torch.isfinite(qe).all())}
            b2.close()
        # BASELINE config #1's robot as a batch: simplearm (3 dofs), 65 536 worlds x 64 steps, float32 -- small worlds
        # share wavefronts (the library's forest of 8 copies, include/arbstep.h ARB_STEP_ONE_WORLD), against one world
        # per wavefront
        from arboris_python_amd import scenes
        mdl = scenes.flat(scenes.simplearm_world())
        b2 = BatchedWorlds(mdl, local_rank)
        rng = np.random.default_rng(7)
        nb1, ns1 = 65536, 64
        ta, tb = b2.to_device(rng.uniform(-1., 1., (nb1, mdl.nq)), rng.uniform(-1., 1., (nb1, mdl.ndof)), torch.float32)
        arm = {"workload": "simplearm (3 dof), %d worlds x %d steps in one launch, float32 (BASELINE config #1's robot, batched)"
                           % (nb1, ns1), "forest_copies": b2.info["forest_copies"]}
        for key, one in (("world_steps_per_s", False), ("world_steps_per_s_one_world_per_wavefront", True)):
            best = None
            for _ in range(6):
                qa, da = ta.clone(), tb.clone()
                torch.cuda.synchronize(b2.device)
                t0 = time.perf_counter()
                b2.step(qa, da, 1e-3, ns1, one_world=one)
                torch.cuda.synchronize(b2.device)
                el = time.perf_counter() - t0
                best = el if best is None else min(best, el)
            arm[key] = nb1 * ns1 / best
            arm["finite"] = bool(torch.isfinite(qa).all()) and arm.get("finite", True)
        extra["config1_batched"] = arm
        b2.close()
        res["extra"] = extra
    sys.stdout.flush()
    os.write(json_fd, (json.dumps(res) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
