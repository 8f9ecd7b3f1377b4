#!/bin/bash
# Run on the GPU box (through gpurun): kernel-trace stats and PMC passes of bench.py (config 3, the headline workload).
# PMC counters go in their own passes (never combined with trace domains other than kernel-trace).
# usage: tools/profile_gpu.sh <tag>      -> gpurun_out/<tag>/{trace,pmc_*}; then tools/summarize_profile.py gpurun_out/<tag> <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-prof}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# >= 50 whole-episode launches (one 40-step episode each) + the one-launch-per-step leg
BENCH="python3 $R/bench.py --steps 40 --warmup 40 --min-seconds 1 --no-cpu-baseline"
# PMC passes: episode launches only
BENCH_S="python3 $R/bench.py --steps 40 --warmup 40 --min-seconds 0.2 --no-cpu-baseline --no-per-step-leg"
# (ARB_BENCH_LEGS=perstep: the headline region + the one-launch-per-step leg, without the float64 / 8-contact / MPC legs the
# default bench line also carries -- their kernels would mix into the per-kernel statistics)
ARB_BENCH_LEGS=perstep rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH_S > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH_S > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq1 -- $BENCH_S > $OUT/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_sq2 -- $BENCH_S > $OUT/pmc_sq2.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $OUT/pmc_mfma -- $BENCH_S > $OUT/pmc_mfma.log 2>&1
# dynamic instruction mix of the vector ALU (round 4): float32 / float64 add, multiply, fused multiply-add, transcendental;
# integer, conversion.  What these classes leave of SQ_INSTS_VALU are moves, selects, compares, lane exchanges.
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d $OUT/pmc_mix1 -- $BENCH_S > $OUT/pmc_mix1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH --output-format csv -d $OUT/pmc_mix2 -- $BENCH_S > $OUT/pmc_mix2.log 2>&1
# lane utilisation: enabled lanes per executed VALU instruction (its own pass: the ratio comes from one run)
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/pmc_lanes -- $BENCH_S > $OUT/pmc_lanes.log 2>&1
find $OUT -name "*.csv" | head -40
for f in $OUT/*.log; do echo "== $f"; tail -2 $f | cut -c1-300; done
