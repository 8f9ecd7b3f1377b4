// lds_granule_probe.hip -- how many one-wave workgroups with X bytes of dynamic LDS does a CU of the MI355X really hold?
// (a) what hipOccupancyMaxActiveBlocksPerMultiprocessor says, (b) what a timed launch of k x 256 spinning workgroups
// shows: one round if they are all resident, two if not.
// build: hipcc --offload-arch=gfx950 -O2 -o build/lds_probe tools/lds_granule_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ unsigned char lds[];
__global__ __launch_bounds__(64) void spin(int *out, long long cycles) {
    const long long t0 = clock64();
    while (clock64() - t0 < cycles) { }
    if (lds[threadIdx.x] == 77) out[0] = 1;
}
static float timed(int grid, int bytes) {
    int *out; (void)hipMalloc(&out, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(spin, dim3(grid), dim3(64), bytes, 0, out, 1000ll);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(spin, dim3(grid), dim3(64), bytes, 0, out, 10000ll);       // ~100 us at 100 MHz
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipFree(out);
    return ms;
}
int main() {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(spin), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    int cus = 0; (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    for (int k : {4, 5, 8, 12}) {
        const int nominal = 160 * 1024 / k;
        printf("-- %d workgroups per CU (nominal limit %d B)\n", k, nominal);
        for (int bytes = nominal - 2048; bytes <= nominal + 256; bytes += 128) {
            int n = 0; (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, spin, 64, bytes);
            printf("   %6d B: API %2d per CU, %d workgroups take %.3f ms\n", bytes, n, k * cus, timed(k * cus, bytes));
        }
    }
    return 0;
}
