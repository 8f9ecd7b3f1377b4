"""Round-3 GPU tests (all through the C ABI):

  * SURVEY 8(f4) ON THE DEVICE: a `BatchedWorlds.rollout` log of the falling human36 goes through
    `observers.batched_trajectory` -> `visu_collada.write_collada_animation`, and the matrices in the written
    COLLADA file equal the oracle's body poses of the logged states (reference writer: visu_collada.py:326-388,
    traversal core.py:562-606);
  * the work queue's failure mode is no longer silent: a wavefront whose wait for a chunk expires raises a status
    word, `arb_model_status` / the next call report ARB_ERR_STALLED;
  * a per-step `dt` tensor on a side stream: uploaded on the launch stream and kept alive;
  * unknown flag bits (the removed ARB_STEP_SPLIT = 4) are refused;
  * two PROCESSES on the one visible GPU, a gloo group, each stepping its `dist.shard_bounds`
    shard of one global batch for a 40-step episode; gathered and compared bit for bit with the unsharded launch;
  * `bench.py` under a 1-rank `torch.distributed.run` (RCCL initialised, barrier / all-reduce / all-gather in place)
    and the literal MPC shape of config 5.
"""
import json
import os
import socket
import subprocess
import sys
import xml.etree.ElementTree as ET

import numpy as np
import pytest

import arb_oracle as O
from conftest import load_model, ROOT
from arboris_python_amd import _capi

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

NS = "http://www.collada.org/2005/11/COLLADASchema"
Q = lambda tag: "{%s}%s" % (NS, tag)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


# ---------------------------------------------------------------------------
# (f4) exporters fed by a DEVICE rollout
# ---------------------------------------------------------------------------
def test_device_rollout_to_collada_animation(tmp_path):
    from arboris_python_amd import scenes, synth, observers
    from arboris_python_amd.batch import BatchedWorlds
    from arboris_python_amd.flatten import flatten_world
    from arboris_python_amd.visu_collada import write_collada_scene, write_collada_animation
    w = scenes.human36_world(4)
    m, _, _ = flatten_world(w)
    bw = BatchedWorlds(m)
    B, T, dt, pick = 16, 24, 5e-3, 11
    q, dq = synth.standing_states(m, B, seed=21, drop=0.03, vel=0.2)
    q[:, 7] -= 0.02                                     # the feet reach the floor within the rollout
    tq, tdq = bw.to_device(q, dq, torch.float64)
    cf = bw.new_cforce(B, torch.float64)
    log = bw.rollout(tq, tdq, dt, T, cforce=cf)
    torch.cuda.synchronize()
    assert float(cf[:, :, 3].max()) > 10.               # contacts engaged
    data = observers.batched_trajectory(bw, w, log, dt, t0=0., world_index=pick, flat=True)
    # the scene in the configuration of the first logged state, the animation from the device log
    w.update_geometric()
    scene, anim = str(tmp_path / "scene.dae"), str(tmp_path / "anim.dae")
    write_collada_scene(w, scene, flat=True)
    names = [b.name for b in w.ground.iter_descendant_bodies() if b.name is not None]
    count = write_collada_animation(anim, scene, data)
    assert count == len(names) + 1                       # every named body + the ground
    # what the oracle says the bodies' poses are at the logged states (Body.update_geometric, core.py:1135-1156)
    ref = O.body_poses(m, log["q"][:, pick].cpu().numpy())            # (T, nb, 4, 4)
    root = ET.parse(anim).getroot()
    anims = {a.get("id"): a for a in root.find(Q("library_animations")).findall(Q("animation"))}
    bodies = [b for b in w.ground.iter_descendant_bodies()]
    checked = 0
    for b, body in enumerate(bodies):
        if body.name is None:
            continue
        from arboris_python_amd.visu_collada import _safe_id
        a = anims[_safe_id(body.name) + ".anim"]
        arrays = {s.get("id"): s for s in a.findall(Q("source"))}
        aid = _safe_id(body.name) + ".anim"
        t = np.array([float(x) for x in arrays[aid + ".input"].find(Q("float_array")).text.split()])
        H = np.array([float(x) for x in arrays[aid + ".output"].find(Q("float_array")).text.split()]).reshape(-1, 4, 4)
        assert np.allclose(t, dt * np.arange(T), atol=1e-15)
        assert H.shape == (T, 4, 4)
        assert np.abs(H - ref[:, b]).max() < 1e-9, body.name
        checked += 1
    assert checked == len(names) >= 17
    # the rollout moved: the animation is not a still image
    assert np.abs(ref[-1] - ref[0]).max() > 1e-3
    bw.close()


# ---------------------------------------------------------------------------
# work queue: a wait that expires is reported
# ---------------------------------------------------------------------------
def test_queue_stall_is_reported(monkeypatch):
    from arboris_python_amd import synth, _capi
    from arboris_python_amd.batch import BatchedWorlds
    m, _, _ = load_model("human36_c4")
    bw = BatchedWorlds(m)
    B, T, dt = 6000, 12, 5e-3                            # more worlds than wave slots: the queue is in use
    q, dq = synth.standing_states(m, B, seed=2, drop=0.03, vel=0.1)
    tq, tdq = bw.to_device(q, dq, torch.float32)
    cf = bw.new_cforce(B, torch.float32)
    bw.step(tq, tdq, dt, T, cforce=cf)
    torch.cuda.synchronize()
    bw.status()                                          # healthy launch: nothing raised
    good_q = tq.clone()
    # fault injection: every wait for an earlier chunk "expires" (a negative cap)
    bw.set_knob("queue_spin_cap", -1)                    # (arbstep_hooks.h: the library reads no environment variable)
    sq, sdq = bw.to_device(q, dq, torch.float32)
    scf = bw.new_cforce(B, torch.float32)
    bw.step(sq, sdq, dt, T, cforce=scf)
    torch.cuda.synchronize()
    bw.set_knob("queue_spin_cap", 1 << 24)
    with pytest.raises(_capi.ArbError) as ei:
        bw.status()
    assert "status %d" % _capi.ARB_ERR_STALLED in str(ei.value)
    bw.status()                                          # reading the word cleared it
    # a stall is also reported by the next step call on the handle, which then does nothing
    bw.set_knob("queue_spin_cap", -1)
    sq, sdq = bw.to_device(q, dq, torch.float32)
    bw.step(sq, sdq, dt, T, cforce=bw.new_cforce(B, torch.float32))
    torch.cuda.synchronize()
    bw.set_knob("queue_spin_cap", 1 << 24)
    before = sq.clone()
    for _ in range(2):                                   # the word is sticky: EVERY call fails until it is acknowledged
        with pytest.raises(_capi.ArbError):
            bw.step(sq, sdq, dt, 1, cforce=scf)
        torch.cuda.synchronize()
        assert torch.equal(sq, before)
    with pytest.raises(_capi.ArbError):
        bw.inspect(sq[:4], sdq[:4], dt, ["q_next"], cforce=scf[:4])
    with pytest.raises(_capi.ArbError):
        bw.status()                                      # the acknowledgement (arb_model_status reads and clears)
    bw.status()
    # and the handle works again afterwards, bit for bit
    rq, rdq = bw.to_device(q, dq, torch.float32)
    bw.step(rq, rdq, dt, T, cforce=bw.new_cforce(B, torch.float32))
    torch.cuda.synchronize()
    assert torch.equal(rq, good_q)
    bw.close()


def test_unknown_flags_are_refused():
    import ctypes as C
    from arboris_python_amd import synth, _capi
    from arboris_python_amd.batch import BatchedWorlds
    m, _, _ = load_model("human36_c4")
    bw = BatchedWorlds(m)
    q, dq = synth.standing_states(m, 4, seed=2)
    tq, tdq = bw.to_device(q, dq, torch.float32)
    before = tq.clone()
    for bad in (4, 16384, 1 << 20):                        # 4 was ARB_STEP_SPLIT (the removed lane-per-world sweep kernel); 2048 .. 8192: ABI 8
        rc = bw._lib.arb_step(bw._handle, _capi.ARB_F32, tq.data_ptr(), tdq.data_ptr(), None, None, 4, 5e-3, 1, bad, None)
        assert rc == 1
    with pytest.raises(ValueError):
        bw.step(tq, tdq, 5e-3, 1, split=True)
    torch.cuda.synchronize()
    assert torch.equal(tq, before)
    bw.close()


# ---------------------------------------------------------------------------
# per-step dt on a side stream
# ---------------------------------------------------------------------------
def test_per_step_dt_on_a_side_stream():
    from arboris_python_amd import synth
    from arboris_python_amd.batch import BatchedWorlds
    m, _, _ = load_model("human36_c4")
    bw = BatchedWorlds(m)
    B, T = 512, 8
    dts = np.array([5e-3, 2e-3, 4e-3, 1e-3, 5e-3, 3e-3, 2.5e-3, 5e-3])
    q, dq = synth.standing_states(m, B, seed=4, drop=0.02, vel=0.3)
    q[:, 7] -= 0.01
    # reference: one launch per step on the default stream
    rq, rdq = bw.to_device(q, dq, torch.float32)
    rcf = bw.new_cforce(B, torch.float32)
    for k in range(T):
        bw.step(rq, rdq, float(dts[k]), 1, cforce=rcf)
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=bw.device)
    sq, sdq = bw.to_device(q, dq, torch.float32)
    scf = bw.new_cforce(B, torch.float32)
    torch.cuda.synchronize()
    # keep the default stream busy so that an upload left on it would still be in flight when the side stream starts
    junk = torch.empty((64, 1024, 1024), device=bw.device)
    for _ in range(4):
        junk.normal_()
    bw.step(sq, sdq, dts, T, cforce=scf, stream=side)
    # a second timeline right behind it: the first call's dt tensor must outlive its kernel
    tq2, tdq2 = bw.to_device(q, dq, torch.float32)
    torch.cuda.current_stream(bw.device).synchronize()
    bw.step(tq2, tdq2, dts[::-1].copy(), T, cforce=bw.new_cforce(B, torch.float32), stream=side)
    side.synchronize()
    torch.cuda.synchronize()
    assert torch.equal(sq, rq) and torch.equal(sdq, rdq) and torch.equal(scf, rcf)
    # scalars of every kind are scalars (np.float32, 0-d array, 0-d tensor)
    for s in (np.float32(0.005), np.asarray(0.005), torch.tensor(0.005)):
        aq, adq = bw.to_device(q[:8], dq[:8], torch.float32)
        bw.step(aq, adq, s, 1, cforce=bw.new_cforce(8, torch.float32))
    torch.cuda.synchronize()
    bw.close()


# ---------------------------------------------------------------------------
# multi-GPU readiness on ONE GPU: two processes, one global batch
# ---------------------------------------------------------------------------
_SHARD_WORKER = r"""
import os, sys
import numpy as np
root, out, rank, ws, port, B, T = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], int(sys.argv[6]), int(sys.argv[7])
sys.path.insert(0, root); sys.path.insert(0, root + "/tests")
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
import torch
import torch.distributed as dist
dist.init_process_group("gloo", rank=rank, world_size=ws)
from conftest import load_model
from arboris_python_amd import synth
from arboris_python_amd.batch import BatchedWorlds
from arboris_python_amd.dist import shard_bounds, gather_state
m, _, _ = load_model("human36_c4")
bw = BatchedWorlds(m, 0)                                  # both ranks on the one visible GPU
lo, hi = shard_bounds(B, rank, ws)
q, dq = synth.world_states(m, range(lo, hi), "standing", 77, drop=0.03, vel=0.1)
tq, tdq = bw.to_device(q, dq, torch.float32)
cf = bw.new_cforce(hi - lo, torch.float32)
dist.barrier()
bw.step(tq, tdq, 5e-3, T, cforce=cf, waves=3)      # the kernel build the unsharded batch runs (picked by batch size)
torch.cuda.synchronize()
bw.status()
q_all, dq_all = gather_state(tq.cpu(), tdq.cpu(), B, dist)     # gloo: the collective runs on host copies
if rank == 0:
    np.savez(out, q=q_all.numpy(), dq=dq_all.numpy(), shard=np.array([lo, hi]))
dist.barrier()
dist.destroy_process_group()
bw.close()
"""


def test_two_process_shards_equal_unsharded_bitwise(tmp_path):
    from arboris_python_amd import synth
    from arboris_python_amd.batch import BatchedWorlds
    B, T, ws = 5000, 40, 2                               # a ragged split is exercised by the odd shard of 3 ranks below
    out = str(tmp_path / "gathered.npz")
    port = str(_free_port())
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, "-c", _SHARD_WORKER, ROOT, out, str(r), str(ws), port, str(B), str(T)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(ws)]
    logs = [p.communicate(timeout=900)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    g = np.load(out)
    m, _, _ = load_model("human36_c4")
    bw = BatchedWorlds(m)
    q, dq = synth.world_states(m, range(B), "standing", 77, drop=0.03, vel=0.1)
    tq, tdq = bw.to_device(q, dq, torch.float32)
    bw.step(tq, tdq, 5e-3, T, cforce=bw.new_cforce(B, torch.float32), waves=3)
    torch.cuda.synchronize()
    assert g["q"].shape == (B, m.nq) and g["dq"].shape == (B, m.ndof)
    assert np.array_equal(g["q"], tq.cpu().numpy()) and np.array_equal(g["dq"], tdq.cpu().numpy())
    assert np.isfinite(g["q"]).all()
    # k shards on one GPU from one process as well: three ragged shards on three streams
    parts_q = []
    for r in range(3):
        lo, hi = __import__("arboris_python_amd.dist", fromlist=["shard_bounds"]).shard_bounds(B, r, 3)
        sq, sdq = bw.to_device(q[lo:hi], dq[lo:hi], torch.float32)
        scf = bw.new_cforce(hi - lo, torch.float32)
        st = torch.cuda.Stream(device=bw.device)
        st.wait_stream(torch.cuda.current_stream(bw.device))      # uploads and the zero fill are done before the shard runs
        bw.step(sq, sdq, 5e-3, T, cforce=scf, stream=st, waves=3)
        parts_q.append((sq, st, sdq, scf))
    for p in parts_q:
        p[1].synchronize()
    assert torch.equal(torch.cat([p[0] for p in parts_q]), tq)
    bw.close()


# ---------------------------------------------------------------------------
# bench.py: the RCCL leg with one rank, and the MPC shape
# ---------------------------------------------------------------------------
def _bench(args, launcher=False, timeout=900):
    cmd = [sys.executable]
    if launcher:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port())]
    cmd += [os.path.join(ROOT, "bench.py")] + args
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()[-2000:]
    return json.loads(lines[0])


def test_bench_under_one_rank_launcher_uses_rccl():
    r = _bench(["--gpus", "1", "--steps", "40", "--warmup", "40", "--min-seconds", "0.3", "--no-cpu-baseline",
                "--no-per-step-leg"], launcher=True)
    assert r["n_gpus"] == 1 and r["config"]["global_batch"] == 4096 and r["config"]["parallelism"] == "dp1"
    assert r["final_state_allgather_ms"] is not None and r["final_state_allgather_ms"] > 0.     # RCCL all-gather ran
    assert r["state_finite"] and r["value"] > 1e6
    assert r["unit"] == "world-steps/s" and r["scaling"] == "weak" and r["vs_baseline"] is None
    assert r["roofline"]["unit"] == "GB/s" and 0 < r["roofline"]["frac"] < 1


def test_bench_mpc_shape():
    r = _bench(["--config", "5", "--mpc", "--steps", "32", "--warmup", "32", "--min-seconds", "0.3", "--no-cpu-baseline",
                "--no-per-step-leg"])
    assert r["config"]["worlds_per_gpu"] == 256 and r["episode_steps"] == 32 and r["config"]["steps_per_launch"] == 32
    assert "MPC" in r["config"]["workload"]
    assert r["state_finite"] and r["value"] > 1e5


# ---------------------------------------------------------------------------
# the sweep kernel of the split execution == the fused kernel, bit for bit
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("name,B", [("human36_c4", 1001), ("human36_c8", 301)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_split_sweeps_equal_fused_bitwise(monkeypatch, name, B, dtype):
    """The one-wavefront-per-world sweep kernel of the split execution (arb_gsw_kernel, three or four waves per SIMD) against
    the fused kernel: same forces, velocities and positions bit for bit over a whole falling episode.  (Rounds 3 and 4 also
    carried sweep kernels with two and four worlds per wavefront and the packed / rendezvous builds of the step kernel:
    bit-identical, measured slower at every batch size -- DESIGN.md section 6 --, removed in round 5.)"""
    from arboris_python_amd import synth
    from arboris_python_amd.batch import BatchedWorlds
    m, _, _ = load_model(name)
    bw = BatchedWorlds(m)
    q, dq = synth.world_states(m, range(B), "standing", 11, drop=0.03, vel=0.2)
    q[:, 7] -= 0.01
    res = {}
    for mode in ("fused", "wave3", "wave4"):
        bw.set_knob("gsw_waves", 4 if mode == "wave4" else 3)
        tq, tdq = bw.to_device(q, dq, dtype)
        cf = bw.new_cforce(B, dtype)
        # (general_kernels: the split execution runs the general kernels; the fused reference must too -- the eight-contact
        # model's default are the body-space-column kernels of round 5, equal to rounding only)
        bw.step(tq, tdq, 5e-3, 40, cforce=cf, split=("wave" if mode != "fused" else False), general_kernels=(mode == "fused"))
        torch.cuda.synchronize()
        res[mode] = (tq, tdq, cf)
    assert float(res["fused"][2][:, :, 3].max()) > 100.                      # contacts engaged, sliding included
    for mode in ("wave3", "wave4"):
        assert all(torch.equal(a, b) for a, b in zip(res["fused"], res[mode])), mode
    bw.close()


def test_fast_sweeps_hand_over_to_the_complete_variant_bitwise(monkeypatch):
    """Round 4: worlds whose active constraints are all SoftFingerContacts with eps = (1,1,1) run their sweeps through a
    variant of the local solve without the rare routes (the 6x6 eigenvalue routine, row exchanges in the 4x4 solve); a
    solve that needs one hands the step over to the complete variant (gs_stage, arb_kernels.hip).  libarbstep_variants.so
    is compiled without the fast variant: whole falling episodes of 4096 worlds -- late steps have 2-6 worlds with such
    solves, counted here with the inspect kernel -- must agree bit for bit, in every build (two waves, three waves, one
    launch per step)."""
    from arboris_python_amd import synth
    from arboris_python_amd.batch import BatchedWorlds
    m, _, _ = load_model("human36_c4")
    variants = _capi.load_variants()
    assert variants.arb_build_variants() == 12 and _capi.load().arb_build_variants() == 0
    bws = {"fast": BatchedWorlds(m), "complete": BatchedWorlds(m, lib=variants)}
    B, T = 4096, 40
    q, dq = synth.standing_states(m, B, seed=1000, drop=0.03, vel=0.1)
    handed_over = 0
    for mode, kw in (("episode", {}), ("two_waves", dict(waves=2)), ("per_step", {})):
        res = {}
        for key, bw in bws.items():
            tq, tdq = bw.to_device(q, dq, torch.float32)
            cf = bw.new_cforce(B, torch.float32)
            if mode == "per_step":
                for k in range(T):
                    if key == "fast" and k >= 30:
                        st = bw.inspect(tq, tdq, 5e-3, ["gs_stats"], cforce=cf.clone())["gs_stats"]
                        handed_over += int((st[:, 3] > 0).sum())
                    bw.step(tq, tdq, 5e-3, 1, cforce=cf, classic_columns=True)
            else:
                # (classical columns in both libraries: the test library has no body-space-column kernels, the shipped library's
                # default for this model since round 6)
                bw.step(tq, tdq, 5e-3, T, cforce=cf, classic_columns=True, **kw)
            torch.cuda.synchronize()
            bw.status()
            res[key] = (tq, tdq, cf)
        assert all(torch.equal(a, b) for a, b in zip(res["fast"], res["complete"])), mode
    assert handed_over >= 5, handed_over            # (worlds whose sweeps met the eigenvalue route in the last ten steps)
    for bw in bws.values():
        bw.close()


# ---------------------------------------------------------------------------
# arb_step_plan: which build of the step kernel a launch gets
# ---------------------------------------------------------------------------
def test_step_plan_reports_the_batch_size_rules(monkeypatch):
    """The float32 step kernel of a human36-sized model exists as a two-wave and a three-wave build; arb_step_plan reports
    which one a launch shape gets.  On an MI355X (256 CUs): two waves for small batches and one-step launches, three from
    ~3400 worlds of a multi-step launch; float64 and models with two column sets have the two-wave build only."""
    from arboris_python_amd.batch import BatchedWorlds
    m, _, _ = load_model("human36_c4")
    bw = BatchedWorlds(m)
    cus = torch.cuda.get_device_properties(bw.device).multi_processor_count
    build = lambda p: (p["waves_per_simd"], p["worlds_per_wavefront"])
    p = bw.plan(4 * cus, 40)
    # (feat 52 / 53 = 16 + 32 + 4 [+ 1]: body-space constraint columns compiled for four contacts, this model's default since
    # round 6; feat 4 / 5 with classic_columns: the classical columns specialised for four plane / sphere SoftFingerContacts,
    # round 4 -- tests/test_gpu_round4.py)
    assert build(p) == (2, 1) and p["work_queue"] == 0 and p["feat"] == 52 and p["wave_slots"] == 8 * cus
    assert bw.plan(4 * cus, 40, classic_columns=True)["feat"] == 4
    p = bw.plan(16 * cus, 40)
    assert build(p) == (3, 1) and p["work_queue"] == 1 and p["wave_slots"] == 12 * cus
    assert p["lds_bytes"] * 12 <= 160 * 1024
    assert build(bw.plan(16 * cus, 1)) == (2, 1) and bw.plan(16 * cus, 1)["work_queue"] == 0
    assert build(bw.plan(16 * cus, 40, waves=2)) == (2, 1)
    assert build(bw.plan(4 * cus, 40, waves=3)) == (3, 1)
    p = bw.plan(64 * cus, 40, ext_gforce=True)
    assert build(p) == (3, 1) and p["feat"] == 53 and p["work_queue"] == 1
    assert bw.plan(64 * cus, 40, ext_gforce=True, classic_columns=True)["feat"] == 5
    assert p["lds_bytes"] <= 10 * 1280                       # twelve wavefronts per CU at the 1280-byte LDS granule
    assert build(bw.plan(64 * cus, 40, other_inputs=True)) == (3, 1) and bw.plan(64 * cus, 40, other_inputs=True)["feat"] == 19
    assert bw.plan(64 * cus, 40, other_inputs=True, classic_columns=True)["feat"] == 3
    assert build(bw.plan(64 * cus, 40, waves=3)) == (3, 1)
    assert build(bw.plan(64 * cus, 40, dtype=torch.float64)) == (2, 1)
    assert bw.plan(64 * cus, 40, static_worlds=True)["work_queue"] == 0
    bw.close()
    # a small model: one world per wavefront on request ...
    m, _, _ = load_model("ballsocket")
    bw = BatchedWorlds(m)
    assert bw.plan(64 * cus, 40, one_world=True)["worlds_per_wavefront"] == 1
    # (by default its worlds share wavefronts another way: a forest of 5 copies, tests/test_gpu_forest.py)
    assert bw.plan(64 * cus, 40)["worlds_per_wavefront"] == bw.info["forest_copies"] == 5
    bw.close()


# ---------------------------------------------------------------------------
# deep trees in the float64 kernels: log-depth pose / twist / acceleration chains (ARB_JUMP_DEPTH)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("nlinks,free_root", [(20, False), (24, True)])
def test_deep_chain_with_contacts_float64(nlinks, free_root):
    """A 20-link hinge chain bolted to the ground and a free-floating 24-link one (30 dofs), each with spheres touching a
    wall (and one another) and a joint limit -- trees deeper than ARB_JUMP_DEPTH, so the float64 kernels chain poses, twists and bias
    accelerations by pointer jumping / world-frame prefix sums instead of level by level -- against the oracle: single
    steps from random states and a 12-step launch (work queue), at the float64 gates."""
    from arboris_python_amd.core import World, SubFrame
    from arboris_python_amd.robots.snake import add_snake
    from arboris_python_amd.shapes import Sphere, Plane
    from arboris_python_amd.controllers import WeightController
    from arboris_python_amd.constraints import get_all_contacts, JointLimits
    from arboris_python_amd.flatten import flatten_world
    from arboris_python_amd.batch import BatchedWorlds
    from arboris_python_amd import synth
    w = World()
    w.register(Plane(w.ground, (1., 0., 0., -0.06), "wall"))             # (the chains stand along y: a wall beside them)
    add_snake(w, nlinks, is_fixed=not free_root, lengths=[0.07] * nlinks, masses=[0.3] * nlinks)
    bodies = [b for b in w.getbodies() if b is not w.ground]
    for k, b in enumerate((bodies[-1], bodies[len(bodies) // 2], bodies[3])):
        w.register(Sphere(SubFrame(b, np.eye(4), name="sf%d" % k), 0.04, name="ball%d" % k))
    w.register(WeightController())
    for c in get_all_contacts(w, friction_coeff=0.7):
        w.register(c)
    hinge = [j for j in w.getjoints() if j.ndof == 1][5]
    w.register(JointLimits(hinge, -0.3, 0.3))
    w.init()
    m, q0, dq0 = flatten_world(w)
    assert int(max(m.parent)) >= 12 and m.nc >= 3
    bw = BatchedWorlds(m)
    B = 64
    q, dq = synth.random_states(m, B, seed=3, angle=0.25, vel=0.8, root_box=((-.05, .15), (-.1, .1), (-.1, .1)), root_rot=False)
    q[0], dq[0] = q0, dq0
    cf0 = np.zeros((B, m.nc, 4))
    oq, odq, ocf = O.step(m, q, dq, 2e-3, cforce=cf0)
    ok = np.isfinite(odq).all(axis=1) & (np.abs(odq).max(axis=1) < 1e3)
    assert ok.sum() >= B // 2
    assert (np.abs(ocf[ok]).max(axis=(1, 2)) > 0).sum() >= 4              # some worlds touch the floor
    tq, tdq = bw.to_device(q, dq, torch.float64)
    tcf = bw.new_cforce(B, torch.float64)
    bw.step(tq, tdq, 2e-3, 1, cforce=tcf)
    torch.cuda.synchronize()
    eq = np.abs(tq.cpu().numpy() - oq).max(axis=1) / np.maximum(1., np.abs(oq).max(axis=1))
    edq = np.abs(tdq.cpu().numpy() - odq).max(axis=1) / np.maximum(1., np.abs(odq).max(axis=1))
    assert eq[ok].max() < 1e-8 and edq[ok].max() < 1e-7, (eq[ok].max(), edq[ok].max())
    # 12 steps in one launch, more worlds than wave slots: the queue, against one workgroup per world bit for bit, and
    # a sample against the oracle's rollout
    reps = -(-3000 // B)
    qb, dqb = np.tile(q, (reps, 1)), np.tile(dq, (reps, 1))
    res = []
    for static in (True, False):
        aq, adq = bw.to_device(qb, dqb, torch.float64)
        acf = bw.new_cforce(len(qb), torch.float64)
        bw.step(aq, adq, 2e-3, 12, cforce=acf, static_worlds=static)
        res.append((aq, adq, acf))
    torch.cuda.synchronize()
    bits = lambda t: t.contiguous().view(torch.int64)
    assert all(torch.equal(bits(a), bits(b)) for a, b in zip(*res))
    sel = np.flatnonzero(ok)[:6]
    rq, rdq, _ = O.rollout(m, q[sel], dq[sel], [2e-3] * 12)
    good = np.isfinite(rdq).all(axis=1) & (np.abs(rdq).max(axis=1) < 1e3)
    got_q, got_dq = res[0][0].cpu().numpy()[sel], res[0][1].cpu().numpy()[sel]
    assert good.sum() >= 3
    assert (np.abs(got_q - rq)[good].max(axis=1) / np.maximum(1., np.abs(rq)[good].max(axis=1))).max() < 1e-6
    bw.close()
