// valu_issue_probe.hip -- what does one SIMD of an MI355X sustain, in wave64 VALU instructions per cycle, with 1, 2, 3 or
// 4 resident waves?  (round 3, VERDICT item 2: the denominator of the step kernel's VALU roofline.)
//
// (256 instructions per loop iteration: the first version looped over 16 and measured its branch.)
// Every workgroup is one wavefront; the grid is waves_per_simd x 1024 (256 CUs x 4 SIMDs) and a dynamic-LDS pad pins the
// number of resident workgroups per CU, so each SIMD holds exactly `waves_per_simd` waves.  Each wave issues N blocks of
// 16 instructions: either 16 INDEPENDENT chains (issue-bound) or ONE dependent chain (latency-bound), float32 or float64,
// and stamps s_memtime around the loop.  Reported: wave-instructions per cycle per SIMD from the in-kernel stamps of the
// median wave and from the hipEvent kernel time at 2.4 GHz.
//
// build: hipcc --offload-arch=gfx950 -O3 -o build/valu_issue_probe tools/valu_issue_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

extern __shared__ unsigned char pad_lds[];

template <int MODE>   // 0 f32 independent, 1 f32 dependent, 2 f64 independent, 3 f64 dependent, 4 f32 DPP quad_perm dependent
__global__ __launch_bounds__(64) void probe(long long *out, float *sink, int iters, float seed) {
    float a[16];
    double d[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = seed + i + threadIdx.x; d[i] = seed + i + threadIdx.x; }
    const float x = 0.999f + seed * 1e-9f, y = 1e-3f;
    const double xd = 0.999 + seed * 1e-9, yd = 1e-3;
    if (pad_lds[threadIdx.x] == 77 && seed == 123.f) a[0] += 1.f;     // keep the LDS allocation
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int rep = 0; rep < 16; ++rep) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(x), "v"(y));
        } else if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(xd), "v"(yd));
        } else if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[0]) : "v"(xd), "v"(yd));
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_add_f32_dpp %0, %0, %1 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf" : "+v"(a[0]) : "v"(y));
        }
      }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i] + (float)d[i];
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (s == 1234.5678f) sink[0] = s;
}

template <int MODE>
static void run(const char *name, int wps, int iters) {
    const int cus = 256;
    // resident workgroups per CU = 4 * wps: pad the LDS so that no more fit (160 KB per CU)
    const size_t lds = (size_t)(160 * 1024) / (4 * wps) - 512;
    hipFuncSetAttribute(reinterpret_cast<const void *>(probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int grid = cus * 4 * wps;
    long long *out; float *sink;
    hipMalloc(&out, sizeof(long long) * grid); hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(64), lds, 0, out, sink, iters / 8, 1.f);      // warm up
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(64), lds, 0, out, sink, iters, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(grid);
    hipMemcpy(h.data(), out, sizeof(long long) * grid, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double insts = 256.0 * iters;
    const double cyc_med = (double)h[grid / 2];
    // s_memtime ticks at a constant 100 MHz on this part when read through readcyclecounter? report both views
    printf("%-28s waves/SIMD %d: median wave %.0f ticks for %.0f insts; kernel %.3f ms -> %.3f wave-insts/cycle/SIMD at 2.4 GHz "
           "(%.2f cycles per inst per wave)\n", name, wps, cyc_med, insts, ms, insts * wps / (ms * 1e-3 * 2.4e9),
           ms * 1e-3 * 2.4e9 / insts);
    hipFree(out); hipFree(sink);
}

int main() {
    const int iters = 20000;
    for (int wps = 1; wps <= 4; ++wps) {
        run<0>("v_fma_f32 independent", wps, iters);
        run<1>("v_fma_f32 dependent chain", wps, iters);
        run<2>("v_fma_f64 independent", wps, iters);
        run<3>("v_fma_f64 dependent chain", wps, iters);
        run<4>("v_add_f32 dpp dependent", wps, iters);
    }
    return 0;
}
