// spill_residency_probe.hip -- where does scratch (register-spill) traffic of the step kernel's size live?  (round 6)
//
// The three-wave build of the headline kernel spills ~350 VGPRs: 508 B of scratch per lane, 3072 resident wavefronts x 64 lanes
// = 100 MB touched over and over (92 spill stores and 215 reloads per world-step).  FETCH_SIZE / WRITE_SIZE count that traffic at
// the L2's fabric side -- Infinity-Cache hits included --, so they cannot say how much of it reaches HBM.  This probe answers with
// a knee: 3072 wavefronts (one workgroup each, like the step kernel) sweep a PRIVATE (scratch) array of S dwords per lane --
// whole-array store pass, whole-array load pass, repeated -- for footprints from 6 MB to 1.6 GB.  Below the L2's 32 MB the rate
// is the L2's, up to the Infinity Cache's 256 MiB it is the fabric's, beyond it HBM's: the step kernel's 100 MB sit on the
// middle plateau.
// build: hipcc --offload-arch=gfx950 -O3 -o build/spill_residency_probe tools/spill_residency_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int S>
__global__ __launch_bounds__(64) void sweep(unsigned *out, int reps, int seed) {
    volatile unsigned buf[S];                    // private: lives in scratch (dynamic indexing below keeps it there)
    unsigned acc = 0;
    for (int r = 0; r < reps; ++r) {
        for (int i = 0; i < S; ++i) buf[(i + seed) % S] = (unsigned)(i * 2654435761u) + r;     // one 256-byte store per wavefront
        for (int i = 0; i < S; ++i) acc += buf[(i * 7 + seed) % S];                            // one 256-byte load per wavefront
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int S>
static void run(unsigned *dout, int waves) {
    const int reps = std::max(2, (int)(2048 / S));
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(sweep<S>, dim3(waves), dim3(64), 0, 0, dout, 1, 0);      // warm (scratch is allocated on first use)
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int k = 0; k < 5; ++k) {
        hipEventRecord(a);
        hipLaunchKernelGGL(sweep<S>, dim3(waves), dim3(64), 0, 0, dout, reps, k);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best;
    }
    const double foot = (double)waves * 64 * 4 * S, bytes = foot * 2 * reps;
    printf("scratch %5d B/lane  footprint %8.1f MB  %6.2f ms  %8.1f GB/s (store + load passes)\n", 4 * S, foot / 1e6, best, bytes / (best * 1e-3) / 1e9);
}

int main() {
    unsigned *dout; hipMalloc(&dout, 64);
    int cus = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int waves = 12 * cus;                // the step kernel's residency: twelve wavefronts per CU
    printf("%d CUs, %d wavefronts of 64 lanes, one workgroup each\n", cus, waves);
    run<8>(dout, waves); run<16>(dout, waves); run<32>(dout, waves); run<64>(dout, waves); run<96>(dout, waves); run<127>(dout, waves);
    run<192>(dout, waves); run<256>(dout, waves); run<320>(dout, waves); run<400>(dout, waves); run<512>(dout, waves);
    run<768>(dout, waves); run<1024>(dout, waves); run<2048>(dout, waves);
    return 0;
}
