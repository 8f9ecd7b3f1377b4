#!/usr/bin/env python3
"""Benchmark of the batched arboris step on MI355X.

Metric (BASELINE.json): world-steps/s, human36 (42 dof) + 4 floor
SoftFingerContacts, batch 4096 worlds per GPU (BASELINE config #3), synthetic
standing/falling states, float32 state and arithmetic.

A "step" is one pass of the hot path (update_dynamic -> update_controllers ->
update_constraints -> integrate, arboris/core.py:1358-1363) over the whole batch
= one arb_step launch advancing every world by one dt.  The state is restored
from the pristine synthetic batch every 40 steps (the length of the reference's
falling-human scenario, tests/test_human36_falling.py) so the workload stays the
free-fall-then-impact regime; the restore is a device-to-device copy inside the
timed region.

Launch:  python bench.py [--gpus N --steps K --warmup W]
         (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)
Worlds are independent, so ranks shard the batch with no data-path collective
(weak scaling: 4096 worlds per GPU); RCCL is used only for the barrier/max-time
reduction and for the final state gather, which is timed separately.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP32_PEAK_TFLOPS = 157.3     # vector FP32 spec peak
RESET_EVERY = 40


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=4096, help="worlds per GPU")
    ap.add_argument("--contacts", type=int, default=4, choices=(0, 4, 8))
    ap.add_argument("--dtype", default="f32", choices=("f32", "f64"))
    ap.add_argument("--dt", type=float, default=5e-3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--steps-per-launch", type=int, default=RESET_EVERY,
                    help="steps advanced by one arb_step call (state stays on chip in between); "
                         "%d = one falling episode per launch, 1 = one launch per step" % RESET_EVERY)
    ap.add_argument("--extra", action="store_true", help="also time the other BASELINE configs")
    return ap.parse_args()


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


def cpu_baseline(model, q, dq, dt, budget_s):
    """The NumPy float64 oracle (a port of the reference algorithm) timed on the host:
    one core, then one single-threaded process per core over disjoint world shards
    (SURVEY 8d), both on a bounded sample of the bench workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import arb_oracle as O
    import contextlib
    import subprocess
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
            if time.perf_counter() - t0 > budget_s or done >= nw * RESET_EVERY:
                break
        el = time.perf_counter() - t0
    out = dict(value=done / el, unit="world-steps/s", cores=1, kind="port",
               sample="%d worlds x %d steps of the bench workload, NumPy float64 oracle "
                      "(oracle/arb_oracle.py), single thread" % (nw, done // nw))
    # all cores: child processes (they never touch the GPU), one BLAS thread each
    try:
        ncores = min(os.cpu_count() or 1, 128, q.shape[0] // 16)
        nsteps = max(2, min(RESET_EVERY, int(budget_s * out["value"] / 16)))
        env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
        with tempfile.TemporaryDirectory() as td:
            f = os.path.join(td, "shard.npz")
            np.savez(f, q=q[:16 * ncores], dq=dq[:16 * ncores], **model.to_npz_dict())
            t0 = time.perf_counter()
            procs = [subprocess.Popen([sys.executable, "-c", _CPU_WORKER, ROOT, f, str(r), "16", str(nsteps), repr(dt)],
                                      stdout=subprocess.PIPE, env=env) for r in range(ncores)]
            times = [float(p.communicate(timeout=600)[0].decode().strip().splitlines()[-1]) for p in procs]
        out["all_cores"] = dict(value=16 * ncores * nsteps / max(times), unit="world-steps/s", cores=ncores,
                                sample="%d single-threaded processes x 16 worlds x %d steps (stepping loops only, "
                                       "slowest process)" % (ncores, nsteps))
    except Exception as e:                              # pragma: no cover
        out["all_cores"] = dict(value=None, error=repr(e))
    return out


def time_config(bw, q0, dq0, dt, steps, warmup, torch, dist=None, use_cf=True, spl=1):
    """Time exactly `steps` steps (after `warmup` untimed ones).  The workload is the reference's
    falling scenario: every RESET_EVERY steps the worlds restart from the initial states.  One
    arb_step call advances min(spl, steps left in the episode) steps.  Returns wall seconds, the
    mean duration of a launch (ms, events on the launch stream), the mean steps per launch,
    finiteness of the final state and the final state."""
    dev = bw.device
    q, dq = q0.clone(), dq0.clone()
    cf = bw.new_cforce(q.shape[0], q.dtype) if (bw.model.nc and use_cf) else None

    def chunks(total):
        k = 0
        while k < total:
            c = min(spl, RESET_EVERY - k % RESET_EVERY, total - k)
            yield k, c
            k += c
    for k, c in chunks(warmup):
        if k % RESET_EVERY == 0:
            q.copy_(q0); dq.copy_(dq0)
        bw.step(q, dq, dt, c, cforce=cf)
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    plan = list(chunks(steps))
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in plan]
    t0 = time.perf_counter()
    for i, (k, c) in enumerate(plan):
        if k % RESET_EVERY == 0:
            q.copy_(q0); dq.copy_(dq0)
        ev[i][0].record()               # on torch's current stream = the stream arb_step launches on
        bw.step(q, dq, dt, c, cforce=cf)
        ev[i][1].record()
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    wall = time.perf_counter() - t0
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    finite = bool(torch.isfinite(q).all() and torch.isfinite(dq).all())
    return wall, kern_ms, steps / float(len(plan)), finite, (q, dq)


def main():
    args = parse()
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist_mod.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        dist = dist_mod
    n_gpus = world
    from arboris_python_amd import scenes, synth
    from arboris_python_amd.batch import BatchedWorlds
    dtype = torch.float32 if args.dtype == "f32" else torch.float64

    model = scenes.flat(scenes.human36_world(args.contacts))
    bw = BatchedWorlds(model, local_rank)
    B = args.batch
    if args.contacts:
        # config 3/5 distribution: standing pose dropped from U(0, 3 cm), small velocities
        q, dq = synth.standing_states(model, B, seed=1000 + rank, drop=0.03, vel=0.1)
    else:
        q, dq = synth.random_states(model, B, seed=1000 + rank)
    q0, dq0 = bw.to_device(q, dq, dtype)

    spl = max(1, args.steps_per_launch)
    wall, kern_ms, spl_avg, finite, (qf, dqf) = time_config(bw, q0, dq0, args.dt, args.steps, args.warmup, torch, dist, spl=spl)
    t = torch.tensor([wall], dtype=torch.float64, device=bw.device)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall = float(t.item())

    # final state gather over RCCL/xGMI (outside the timed region, reported separately)
    gather_ms = None
    if dist is not None:
        from arboris_python_amd.dist import gather_state
        torch.cuda.synchronize()
        g0 = time.perf_counter()
        q_all, dq_all = gather_state(qf, dqf, world * B, dist)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - g0) * 1e3
        assert q_all.shape[0] == world * B

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    elem = 4 if args.dtype == "f32" else 8
    bytes_per_world_step = 2 * (model.nq + model.ndof) * elem           # state in + state out
    value = n_gpus * B * args.steps / wall
    # one launch reads and writes the state once, whatever the number of steps it advances on chip
    achieved_gbs = bytes_per_world_step * B / (kern_ms * 1e-3) / 1e9
    flop_dense = {0: 1.66e6, 4: 1.9e6, 8: 2.13e6}[args.contacts]          # dense-as-written count, SURVEY 6
    res = {
        "metric": "world-steps/sec at batch, human36 ~40-DOF + 4 contacts, 1/2/4/8 MI355X",
        "value": value, "unit": "world-steps/s", "n_gpus": n_gpus, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "human36 (42 dof) + %d floor SoftFingerContact, batch %d worlds/GPU, dt=%g, "
                               "standing-drop states reset every %d steps (BASELINE config #3), "
                               "%g steps per arb_step launch"
                               % (args.contacts, B, args.dt, RESET_EVERY, spl_avg),
                   "worlds_per_gpu": B, "global_batch": n_gpus * B, "parallelism": "dp%d" % n_gpus,
                   "steps_per_launch": spl_avg},
        "roofline": {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved_gbs / HBM_PEAK_GBS, "traffic": None,
                     "kernel": "arb_step_kernel", "kernel_ms": kern_ms,
                     "algorithmic_bytes_per_launch": bytes_per_world_step * B,
                     "algorithmic_bytes_per_world_step": bytes_per_world_step / spl_avg,
                     "note": "compulsory traffic is state in+out only; the path is VALU/latency bound, "
                             "see fp32_vector",
                     "fp32_vector": {"dense_equiv_flop_per_world_step": flop_dense,
                                     "achieved_tflops": flop_dense * B * spl_avg / (kern_ms * 1e-3) / 1e12,
                                     "peak_tflops": FP32_PEAK_TFLOPS,
                                     "frac": flop_dense * B * spl_avg / (kern_ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS}},
        "state_finite": finite,
    }
    # HBM traffic of the dominant kernel from the committed rocprofv3 PMC passes (tools/profile_gpu.sh,
    # profiles/): only valid for the workload those passes were taken on
    import glob
    profs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_summary.json")))   # newest round last by name
    profs = [f for f in profs if "baseline" not in f]
    prof = profs[-1] if profs else ""
    if os.path.exists(prof) and args.contacts == 4 and B == 4096 and args.dtype == "f32" and spl == RESET_EVERY:
        try:
            pm = json.load(open(prof))["pmc_per_launch"]
            res["roofline"]["traffic"] = (pm["FETCH_SIZE"]["mean_per_launch"] + pm["WRITE_SIZE"]["mean_per_launch"]) * 1024.
            res["roofline"]["traffic_note"] = ("bytes per launch, FETCH_SIZE+WRITE_SIZE from separate rocprofv3 --pmc passes "
                                               "(profiles/%s); 4 B/lane accesses, reported uncorrected" % os.path.basename(prof))
            # what actually bounds the kernel: the serial instruction stream of two waves per SIMD
            ws = float(B * RESET_EVERY)
            res["roofline"]["issue"] = {
                "valu_insts_per_world_step": pm["SQ_INSTS_VALU"]["mean_per_launch"] / ws,
                "salu_insts_per_world_step": pm["SQ_INSTS_SALU"]["mean_per_launch"] / ws,
                "lds_insts_per_world_step": pm["SQ_INSTS_LDS"]["mean_per_launch"] / ws,
                "wave_cycles_per_world_step": 4. * pm["SQ_WAVE_CYCLES"]["mean_per_launch"] / ws,
                "wave_frac_issuing_valu": pm["SQ_ACTIVE_INST_VALU"]["mean_per_launch"] / pm["SQ_WAVE_CYCLES"]["mean_per_launch"],
                "wave_frac_waiting": (pm["SQ_WAIT_ANY"]["mean_per_launch"] + pm["SQ_WAIT_INST_ANY"]["mean_per_launch"])
                                     / pm["SQ_WAVE_CYCLES"]["mean_per_launch"],
                "simd_valu_busy_frac": 2. * pm["SQ_ACTIVE_INST_VALU"]["mean_per_launch"] / pm["SQ_WAVE_CYCLES"]["mean_per_launch"],
                "note": "SQ counters of profiles/%s (two waves per SIMD)" % os.path.basename(prof)}
        except Exception:
            pass
    if gather_ms is not None:
        res["final_state_allgather_ms"] = gather_ms
    if n_gpus == 1 and spl != 1:
        # the same workload with one launch per step (how rounds up to r01b were measured)
        w1, k1, _, _, _ = time_config(bw, q0, dq0, args.dt, 2 * RESET_EVERY, RESET_EVERY // 2, torch, None, spl=1)
        res["per_step_launch"] = {"value": B * 2 * RESET_EVERY / w1, "unit": "world-steps/s", "kernel_ms": k1,
                                  "steps": 2 * RESET_EVERY}
    if n_gpus == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(model, q, dq, args.dt, args.cpu_seconds)
        res["cpu_baseline"]["host_cores_available"] = os.cpu_count()
    if args.extra and n_gpus == 1:
        extra = {}
        for name, mdl, gen, kw, bsz, dtt, dty in (
                ("config2_human36_nocontact_b1024_f32", scenes.flat(scenes.human36_world(0)), synth.random_states,
                 dict(seed=0), 1024, 5e-3, torch.float32),
                ("config4_snake64_b2048_f64", scenes.flat(scenes.snake_world(64)), synth.random_states,
                 dict(seed=0, angle=0.5, vel=1.0), 2048, 1e-3, torch.float64),
                ("human36_8contacts_b4096_f32", scenes.flat(scenes.human36_world(8)), synth.standing_states,
                 dict(seed=0, drop=0.03, vel=0.1), 4096, 5e-3, torch.float32)):
            b2 = BatchedWorlds(mdl, local_rank)
            qa, da = gen(mdl, bsz, **kw)
            ta, tb = b2.to_device(qa, da, dty)
            wl, km, _, fin, _ = time_config(b2, ta, tb, dtt, 80, 10, torch, spl=1)
            wr, kr, _, _, _ = time_config(b2, ta, tb, dtt, 80, 10, torch, spl=RESET_EVERY)
            # (from the launch durations: these short runs are dominated by host-side setup otherwise)
            extra[name] = {"world_steps_per_s": bsz * RESET_EVERY / (kr * 1e-3), "launch_ms_%d_steps" % RESET_EVERY: kr,
                           "per_step_launch_world_steps_per_s": bsz / (km * 1e-3), "per_step_launch_ms": km, "finite": fin}
            b2.close()
        res["extra"] = extra
    print(json.dumps(res))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
