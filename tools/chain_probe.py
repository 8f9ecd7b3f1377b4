#!/usr/bin/env python3
"""The LATENCY roof of the step kernel (round 6): what bounds a kernel whose waves are dependent chains.

One world = one wavefront, and a world's step is a chain of dependent instructions.  Alone on its SIMD a wavefront advances
one step in T1 seconds (nothing to wait for but itself); with W wavefronts per SIMD the chip cannot advance more than
`wave_slots / T1` world-steps per second however well the waves interleave -- the LATENCY roof -- nor more than the vector
pipes issue, `1024 SIMDs x clock / pipe cycles per world-step` -- the PIPE roof (from the PMC instruction mix).  The headline
is the smaller of the two times an interleaving efficiency; this script measures T1 (worlds = 4 per CU, the LDS padded so
that no second wavefront fits a SIMD), the throughput at every occupancy in between, and the per-phase cycles of a lone
wavefront (inspect kernel stamps) along the episode.

usage (GPU box): python tools/chain_probe.py [out.json]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from arboris_python_amd import scenes, synth
from arboris_python_amd.batch import BatchedWorlds

out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r6_chain.json")
nc = 4
m = scenes.flat(scenes.human36_world(nc))
bw = BatchedWorlds(m)
cus = torch.cuda.get_device_properties(0).multi_processor_count
EP, dt = 40, 5e-3
res = {"workload": "human36 + 4 contacts, float32, %d-step episodes (BASELINE config 3)" % EP, "cus": cus, "legs": []}


def timed(B, waves, pad, static=True, reps=6):
    bw.set_knob("lds_pad", pad)
    q, dq = synth.standing_states(m, B, seed=1000, drop=0.03, vel=0.1)
    best = None
    for _ in range(reps):
        tq, tdq = bw.to_device(q, dq, torch.float32)
        cf = bw.new_cforce(B, torch.float32)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        bw.step(tq, tdq, dt, EP, cforce=cf, waves=waves, static_worlds=static)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        best = ms if best is None else min(best, ms)
    plan = bw.plan(B, EP, waves=waves, static_worlds=static)
    bw.set_knob("lds_pad", 0)
    return best, plan


base_lds = bw.plan(4096, EP, waves=3)["lds_bytes"]
for waves in (2, 3):
    lds0 = bw.plan(4 * cus, EP, waves=waves, static_worlds=True)["lds_bytes"]
    for per_cu in (4, 8, 12) if waves == 3 else (4, 8):
        # pad the LDS so that exactly `per_cu` wavefronts fit a CU (128 granules of 1280 B)
        gran = 128 // per_cu
        pad = max(0, gran * 1280 - lds0 - 8)
        B = per_cu * cus
        ms, plan = timed(B, waves, pad)
        leg = {"build_waves_per_simd": waves, "wavefronts_per_cu": per_cu, "worlds": B, "lds_pad": pad, "ms_per_episode": ms,
               "us_per_step": ms * 1e3 / EP, "world_steps_per_s": B * EP / (ms * 1e-3), "plan": plan}
        res["legs"].append(leg)
        print("build %d waves/SIMD, %2d wavefronts per CU (%5d worlds, one each): %.3f ms per episode = %.1f us per step -> %.2f M world-steps/s"
              % (waves, per_cu, B, ms, leg["us_per_step"], leg["world_steps_per_s"] / 1e6))
# the headline launch itself
ms, plan = timed(4096, None, 0, static=False)
res["headline"] = {"worlds": 4096, "ms_per_episode": ms, "world_steps_per_s": 4096 * EP / (ms * 1e-3), "plan": plan}
print("headline (4096 worlds, default launch): %.3f ms per episode -> %.2f M world-steps/s, plan %s" % (ms, res["headline"]["world_steps_per_s"] / 1e6, plan))
for waves in (2, 3):
    t1 = [l for l in res["legs"] if l["build_waves_per_simd"] == waves and l["wavefronts_per_cu"] == 4][0]["us_per_step"] * 1e-6
    slots = 4 * waves * cus
    res["latency_roof_build%d" % waves] = {"lone_wave_us_per_step": t1 * 1e6, "wave_slots": slots, "roof_world_steps_per_s": slots / t1}
    print("build %d: T1 = %.1f us per step alone on a SIMD; %d wave slots -> latency roof %.2f M world-steps/s"
          % (waves, t1 * 1e6, slots, slots / t1 / 1e6))
w = res["headline"]["plan"]["waves_per_simd"]
roof = res["latency_roof_build%d" % w]["roof_world_steps_per_s"]
res["headline"]["latency_roof"] = roof
res["headline"]["frac_of_latency_roof"] = res["headline"]["world_steps_per_s"] / roof
print("headline / latency roof = %.3f" % res["headline"]["frac_of_latency_roof"])
# per-phase cycles of a lone wavefront along the episode (inspect kernel: the two-wave build's arithmetic + stores)
lds_i = bw.info["lds_bytes_f32"]
bw.set_knob("lds_pad", max(0, 32 * 1280 - lds_i - 8))
B = 4 * cus
q, dq = synth.standing_states(m, B, seed=1000, drop=0.03, vel=0.1)
tq, tdq = bw.to_device(q, dq, torch.float32)
cf = bw.new_cforce(B, torch.float32)
phases = []
for k in range(EP):
    r = bw.inspect(tq, tdq, dt, ["stamps"], cforce=cf)
    st = r["stamps"].double()
    phases.append((st[:, 1:] - st[:, :-1]).mean(0).tolist())
    bw.step(tq, tdq, dt, 1, cforce=cf)
bw.set_knob("lds_pad", 0)
ph = np.array(phases)
names = ["A", "A'", "B", "C", "D", "GS", "E"]
res["phase_cycles_lone_wave"] = {"names": names, "mean_over_episode": ph.mean(0).tolist(), "free_fall_step0": ph[0].tolist(),
                                 "late_episode_step35": ph[35].tolist(), "total_mean": float(ph.sum(1).mean())}
print("lone-wave cycles per phase, mean over the episode: " + "  ".join("%s %.0f" % (n_, v) for n_, v in zip(names, ph.mean(0))) + "   total %.0f" % ph.sum(1).mean())
print("   step 0 (free fall): " + "  ".join("%s %.0f" % (n_, v) for n_, v in zip(names, ph[0])))
print("   step 35 (sliding):  " + "  ".join("%s %.0f" % (n_, v) for n_, v in zip(names, ph[35])))
json.dump(res, open(out_path, "w"), indent=1)
print("wrote", out_path)
