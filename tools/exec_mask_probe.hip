// exec_mask_probe.hip -- does a wave64 VALU instruction cost less pipe time when only part of the wavefront is enabled?
// (round 4, VERDICT item 1: the Gauss-Seidel sweeps run float64 / DPP instructions on 64 lanes of which one quad is useful.)
//
// Every workgroup is one wavefront; a dynamic-LDS pad (a multiple of the 1280-byte allocation granule) pins `wps` waves per
// SIMD.  Each wave issues blocks of 16 INDEPENDENT instructions (issue-bound) of one kind with lanes [0, active) enabled:
// v_fma_f32, v_fma_f64, v_add_f32 with a DPP quad_perm operand, v_mul_f64.  Reported: wave-instructions per cycle per SIMD
// from the hipEvent kernel time at 2.4 GHz.  If the rate of a 16-lane or 4-lane instruction is higher than that of the
// full wavefront, masking the lanes that carry don't-care data shortens the kernel.
//
// build: hipcc --offload-arch=gfx950 -O3 -o build/exec_mask_probe tools/exec_mask_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

extern __shared__ unsigned char pad_lds[];

template <int KIND>   // 0 v_fma_f32, 1 v_fma_f64, 2 v_add_f32 dpp quad_perm, 3 v_fma_f32 dependent chain, 4 v_fma_f64 dependent chain
__global__ __launch_bounds__(64) void probe(float *sink, int iters, float seed, unsigned long long mask) {
    float a[16];
    double d[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = seed + i + threadIdx.x; d[i] = seed + i + threadIdx.x; }
    const float x = 0.999f + seed * 1e-9f, y = 1e-3f;
    const double xd = 0.999 + seed * 1e-9, yd = 1e-3;
    if (pad_lds[threadIdx.x] == 77 && seed == 123.f) a[0] += 1.f;     // keep the LDS allocation
    if ((mask >> threadIdx.x) & 1ull) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 16; ++rep) {
                if (KIND == 0) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
                } else if (KIND == 1) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(xd), "v"(yd));
                } else if (KIND == 2) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) asm volatile("v_add_f32_dpp %0, %0, %1 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(y));
                } else if (KIND == 3) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(x), "v"(y));
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[0]) : "v"(xd), "v"(yd));
                }
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i] + (float)d[i];
    if (s == 1234.5678f) sink[0] = s;
}

template <int KIND>
static void run(const char *name, int wps, int iters, unsigned long long mask) {
    const int cus = 256;
    const size_t lds = ((size_t)(160 * 1024) / (4 * wps)) / 1280 * 1280;     // exactly 4 * wps workgroups per CU
    hipFuncSetAttribute(reinterpret_cast<const void *>(probe<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int grid = cus * 4 * wps;
    float *sink;
    hipMalloc(&sink, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<KIND>, dim3(grid), dim3(64), lds, 0, sink, iters / 8, 1.f, mask);      // warm up
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<KIND>, dim3(grid), dim3(64), lds, 0, sink, iters, 1.f, mask);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double insts = 256.0 * iters;
    printf("%-26s mask %016llx (%2d lanes) waves/SIMD %d: kernel %7.3f ms -> %.3f wave-insts/cycle/SIMD (%.2f pipe cycles per inst)\n",
           name, mask, __builtin_popcountll(mask), wps, ms, insts * wps / (ms * 1e-3 * 2.4e9), ms * 1e-3 * 2.4e9 / (insts * wps));
    fflush(stdout);
    hipFree(sink);
}


// Mixed residency: even workgroups run with mask A, odd ones with mask B, side by side on the same SIMDs; the loop of every
// wave is timed with s_memtime (shader cycles).  Tells a property of the vector pipe (the sparse waves alone take more
// cycles) from a property of the chip's clock management (everything slows down together, cycles unchanged).
template <int KIND>
__global__ __launch_bounds__(64) void probe_mixed(long long *ticks, long long *real, float *sink, int iters, float seed, unsigned long long maskA,
                                                  unsigned long long maskB, int sel_shift) {
    float a[16];
    double d[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = seed + i + threadIdx.x; d[i] = seed + i + threadIdx.x; }
    const float x = 0.999f + seed * 1e-9f, y = 1e-3f;
    const double xd = 0.999 + seed * 1e-9, yd = 1e-3;
    if (pad_lds[threadIdx.x] == 77 && seed == 123.f) a[0] += 1.f;
    const unsigned long long mask = ((blockIdx.x >> sel_shift) & 1) ? maskB : maskA;
    const long long r0 = wall_clock64();
    const long long t0 = __builtin_readcyclecounter();
    if ((mask >> threadIdx.x) & 1ull) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 16; ++rep) {
                if (KIND == 0) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
                } else if (KIND == 2) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(x), "v"(y));
                } else {
#pragma unroll
                    for (int i = 0; i < 16; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(xd), "v"(yd));
                }
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    const long long r1 = wall_clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i] + (float)d[i];
    if (s == 1234.5678f) sink[0] = s;
    if (threadIdx.x == 0) { ticks[blockIdx.x] = t1 - t0; real[blockIdx.x] = r1 - r0; }
}

template <int KIND>
static void run_mixed(const char *name, int wps, int iters, unsigned long long mA, unsigned long long mB, int sel_shift = 0) {
    const int cus = 256;
    const size_t lds = ((size_t)(160 * 1024) / (4 * wps)) / 1280 * 1280;
    hipFuncSetAttribute(reinterpret_cast<const void *>(probe_mixed<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int grid = cus * 4 * wps;
    float *sink; long long *ticks, *real;
    hipMalloc(&sink, 4); hipMalloc(&ticks, sizeof(long long) * grid); hipMalloc(&real, sizeof(long long) * grid);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe_mixed<KIND>, dim3(grid), dim3(64), lds, 0, ticks, real, sink, iters / 8, 1.f, mA, mB, sel_shift);
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe_mixed<KIND>, dim3(grid), dim3(64), lds, 0, ticks, real, sink, iters, 1.f, mA, mB, sel_shift);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(grid), hr(grid), ea, eb;
    hipMemcpy(h.data(), ticks, sizeof(long long) * grid, hipMemcpyDeviceToHost);
    hipMemcpy(hr.data(), real, sizeof(long long) * grid, hipMemcpyDeviceToHost);
    double mhz = 0.; for (int i = 0; i < grid; ++i) mhz += (double)h[i] / (double)hr[i] * 100.; mhz /= grid;
    for (int i = 0; i < grid; ++i) (((i >> sel_shift) & 1) ? eb : ea).push_back(h[i]);
    std::sort(ea.begin(), ea.end()); std::sort(eb.begin(), eb.end());
    const double insts = 256.0 * iters;
    printf("mixed(sel bit %d) %-22s waves/SIMD %d: A %016llx median %.2f ticks/inst, B %016llx median %.2f ticks/inst; kernel %.3f ms; s_memtime runs at %.0f MHz (s_memrealtime = 100 MHz)\n", sel_shift, name, wps,
           mA, ea[ea.size() / 2] / insts, mB, eb[eb.size() / 2] / insts, ms, mhz);
    fflush(stdout);
    hipFree(sink); hipFree(ticks);
}

int main() {
    const int iters = 4000;
    // which workgroups are sparse: bit `sel` of the workgroup index (bit 0: alternate XCDs; bits 3..6: the same XCD, then
    // the same CU / SIMD sooner or later)
    for (int wps = 2; wps <= 2; ++wps)
        for (int sel : {0, 3, 4, 5, 6, 8, 10}) {
            run_mixed<0>("v_fma_f32 independent", wps, iters, ~0ull, 0xfull, sel);
            run_mixed<2>("v_fma_f32 dependent", wps, iters, ~0ull, 0xfull, sel);
            run_mixed<1>("v_fma_f64 independent", wps, iters, ~0ull, 0xfull, sel);
        }
    return 0;
}
