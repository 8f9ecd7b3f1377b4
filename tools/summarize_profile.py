#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/profile_gpu.sh (gpurun_out/<dir>) into
profiles/<tag>_summary.json + profiles/<tag>_kernel_stats.csv.

usage: summarize_profile.py gpurun_out/<dir> <tag>
"""
import csv, glob, json, os, shutil, sys

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {"launch_shape": {"config": 3, "batch": 4096, "dtype": "f32", "contacts": 4, "steps_per_launch": 40, "split": False}}
stats = sorted(glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime)[-1:]
if stats:
    shutil.copy(stats[0], os.path.join(ROOT, "profiles", tag + "_kernel_stats.csv"))
    rows = list(csv.DictReader(open(stats[0])))
    out["kernel_stats"] = [{k: r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")}
                           for r in rows[:4]]
trace = sorted(glob.glob(os.path.join(src, "trace", "*", "*_kernel_trace.csv")), key=os.path.getmtime)[-1:]
if trace:
    d = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(trace[0]))
         if "arb_step_kernel" in r["Kernel_Name"]]
    d.sort()
    ms = [(b - a) / 1e6 for a, b in d]
    # bench.py launches whole 40-step episodes (warmup, calibration, timed), then one launch per step (per_step_launch leg)
    epi = [x for x in ms if x > 4.0]
    one = [x for x in ms if x <= 4.0]
    epi_s = sorted(epi)
    out["arb_step_kernel"] = dict(dispatches=len(ms), episode_launches=len(epi),
                                  mean_ms_episode_launch=(sum(epi) / len(epi)) if epi else None,
                                  median_ms_episode_launch=epi_s[len(epi_s) // 2] if epi else None,
                                  min_ms_episode_launch=epi_s[0] if epi else None, max_ms_episode_launch=epi_s[-1] if epi else None,
                                  single_step_launches=len(one), mean_ms_single_step_launch=(sum(one) / len(one)) if one else None,
                                  single_step_ms_first_episode=[round(x, 3) for x in one[:40]])
pmc = {}
lanes = {}
def newest_per_pass(pattern):
    """gpurun merges every call's files into the same local directory: keep the newest file of each pass"""
    best = {}
    for f in glob.glob(pattern):
        d = os.path.dirname(f)
        if d not in best or os.path.getmtime(f) > os.path.getmtime(best[d]):
            best[d] = f
    return sorted(best.values())


for f in newest_per_pass(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv")):
    acc = {}
    rows = [r for r in csv.DictReader(open(f)) if "arb_step_kernel" in r["Kernel_Name"]]
    for r in rows:                      # every dispatch of these passes is a whole-episode launch (--no-per-step-leg)
        a = acc.setdefault(r["Counter_Name"], [0.0, set()])
        a[0] += float(r["Counter_Value"]); a[1].add(r["Dispatch_Id"])
    for name, (tot, ids) in acc.items():
        if os.sep + "pmc_lanes" + os.sep in f:
            lanes[name] = tot / len(ids)
            if name == "SQ_ACTIVE_INST_VALU":
                continue                  # (the issue fractions use the value of the pmc_sq2 pass)
        pmc[name] = dict(mean_per_launch=tot / len(ids), launches=len(ids))
if "SQ_THREAD_CYCLES_VALU" in lanes and lanes.get("SQ_ACTIVE_INST_VALU"):
    out["valu_lane_utilisation"] = lanes["SQ_THREAD_CYCLES_VALU"] / (64. * lanes["SQ_ACTIVE_INST_VALU"])
out["pmc_per_launch"] = pmc
# Dynamic instruction mix (round 4) and the pipe-weighted issue fraction: a wave64 float32 instruction holds a SIMD-32's
# vector pipe for 2.4 cycles, a float64 one for 4.3 (tools/exec_mask_probe.hip / valu_issue_probe.hip, two to four
# independent waves per SIMD, shader clock from s_memtime against s_memrealtime); everything the float classes leave --
# moves, selects, compares, integer, lane exchanges -- is priced like float32 (DPP operands cost 4.2: a lower bound).
f32 = ["SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32", "SQ_INSTS_VALU_TRANS_F32"]
f64 = ["SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"]
if all(k in pmc for k in f32 + f64 + ["SQ_INSTS_VALU"]):
    n32 = sum(pmc[k]["mean_per_launch"] for k in f32); n64 = sum(pmc[k]["mean_per_launch"] for k in f64)
    tot = pmc["SQ_INSTS_VALU"]["mean_per_launch"]
    out["valu_mix_per_launch"] = {"float32": n32, "float64": n64, "other": tot - n32 - n64, "total": tot,
                                  "pipe_cycles": 2.4 * (tot - n64) + 4.3 * n64,
                                  "pipe_cycles_note": "2.4 cycles per float32 / other instruction, 4.3 per float64 instruction"}
out["notes"] = ("bench.py config 3: human36 + 4 contacts, 4096 worlds, f32, one 40-step episode per launch; kernel-trace pass: "
                "--steps 40 --warmup 40 --min-seconds 1 (>= 50 timed episode launches + the one-launch-per-step leg); each --pmc "
                "group in its own pass (--min-seconds 0.2 --no-per-step-leg), counters averaged over all episode launches. "
                "FETCH_SIZE / WRITE_SIZE are in KiB (rocprofv3 units), summed over the XCDs per dispatch; the state is "
                "moved with 4 B/lane accesses, for which the guide gives no calibration: reported uncorrected.")
json.dump(out, open(os.path.join(ROOT, "profiles", tag + "_summary.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "kernel_stats"}, indent=1)[:2500])
