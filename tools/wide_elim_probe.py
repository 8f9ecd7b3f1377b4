#!/usr/bin/env python3
"""Where the cycles of one pivot of the compact wide build's elimination go (csrc/arb_wide_kernel.h: wide_eliminate).
Generates a stand-alone HIP program around the function's text (taken from the header, with shader-clock stamps between
its parts), compiles it with hipcc and runs it: one workgroup per CU, a diagonally dominant system of n dofs.
usage (GPU box): python tools/wide_elim_probe.py [n] [KMAX]      output: cycles per pivot and part, mean over the pivots"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
kmax = int(sys.argv[2]) if len(sys.argv) > 2 else (20 if n <= 80 else 28 if n <= 112 else 32)
src = open(os.path.join(ROOT, "arboris_python_amd", "csrc", "arb_wide_kernel.h")).read()
body = src[src.index("typedef double wide_d4"):src.index("#define WIDE_XK")]
prog = r'''
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "arb_math.h"
#define WIDE_THREADS 256
extern __shared__ __attribute__((aligned(16))) unsigned char arb_lds_raw[];
__device__ long long g_acc[8];
#define WIDE_STAMP(i) { const long long c_ = clock64(); if ((i) > 0) acc_[(i)] += c_ - last_; last_ = clock64(); }
#define WIDE_STAMP_DECL long long acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = 0;
#define WIDE_STAMP_END if (blockIdx.x == 0 && threadIdx.x == STAMP_LANE) for (int i_ = 0; i_ < 8; ++i_) g_acc[i_] = acc_[i_];
''' + body.replace("    const int tid = threadIdx.x;\n", "    const int tid = threadIdx.x;\n    WIDE_STAMP_DECL\n", 1) \
          .replace("    __syncthreads();                // (the chain arrays", "    WIDE_STAMP_END\n    __syncthreads();                // (the chain arrays", 1) + r'''
__global__ __launch_bounds__(256) void k(const double *Z, int ld, int n, int nact, double *out, int sld, const double *DQS) {
    wide_eliminate<KMAXV, 2>(Z + (size_t)blockIdx.x * n * ld, ld, n, nact, 0, 1024, sld, DQS, nullptr);
    if (threadIdx.x < n) out[blockIdx.x * n + threadIdx.x] = reinterpret_cast<double *>(arb_lds_raw)[1024 + threadIdx.x * sld];
}
int main() {
    const int n = NV, ld = n + 3, nact = n + 1, sld = 1, nwg = 256;
    std::vector<double> Z((size_t)nwg * n * ld, 0.0), dq(n, 0.0);
    for (int w = 0; w < nwg; ++w) for (int i = 0; i < n; ++i) for (int c = 0; c <= n; ++c)
        Z[((size_t)w * n + i) * ld + c] = c == n ? 1.0 + i : (i == c ? 4.0 + 0.01 * i : 1.0 / (1 + (i > c ? i - c : c - i)));
    double *dZ, *dO, *dD;
    hipMalloc(&dZ, Z.size() * 8); hipMalloc(&dO, (size_t)nwg * n * 8); hipMalloc(&dD, n * 8);
    hipMemcpy(dZ, Z.data(), Z.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dD, dq.data(), n * 8, hipMemcpyHostToDevice);
    const size_t lds = (1024 + (size_t)n * 2) * 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(nwg), dim3(256), lds, 0, dZ, ld, n, nact, dO, sld, dD);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long acc[8];
        hipMemcpyFromSymbol(acc, HIP_SYMBOL(g_acc), sizeof(acc));
        std::vector<double> o((size_t)nwg * n);
        hipMemcpy(o.data(), dO, o.size() * 8, hipMemcpyDeviceToHost);
        printf("n %d KMAX %d lane %d: %.1f us/launch; per pivot: row hand-over %lld  multipliers %lld  barrier %lld  reciprocal + update %lld  set %lld   (x[0] %.6f)\n",
               n, KMAXV, STAMP_LANE, ms * 1e3, acc[1] / n, acc[2] / n, acc[3] / n, acc[4] / n, acc[5] / n, o[0]);
    }
    return 0;
}
'''
out = os.path.join(ROOT, "gpurun_out", "wide_elim_probe")
os.makedirs(out, exist_ok=True)
for lane in (0, 64, 128, 255):
    path = os.path.join(out, "probe_%d.hip" % lane)
    open(path, "w").write(prog.replace("KMAXV", str(kmax)).replace("NV", str(n)).replace("STAMP_LANE", str(lane)))
    exe = path[:-4]
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=on", "-std=c++17", "-mllvm", "-simplifycfg-sink-common=false", "-Wno-unused-value", "-I", os.path.join(ROOT, "arboris_python_amd", "csrc"), "-I", os.path.join(ROOT, "include"), path, "-o", exe])
    sys.stdout.write(subprocess.run([exe], stdout=subprocess.PIPE, universal_newlines=True).stdout)
    sys.stdout.flush()
