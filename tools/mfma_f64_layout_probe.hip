// mfma_f64_layout_probe.hip -- operand / result layout and timing of v_mfma_f64_16x16x4_f64 on gfx950 (round 4: the rows
// of Z as float64 matrix products).  A (16 x 4) and B (4 x 16) hold small integers, D is compared with a host product for the
// assumed layout:  A: lane l -> A[l % 16][l / 16],  B: lane l -> B[l / 16][l % 16],  D: lane l, register v -> D[4 (l / 16) + v][l % 16].
// build: hipcc --offload-arch=gfx950 -O3 -w -o build/mfma_f64_layout_probe tools/mfma_f64_layout_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void k(const double *A, const double *B, double *D, long long *cyc) {
    const int l = threadIdx.x;
    const double a = A[(l % 16) * 4 + l / 16], b = B[(l / 16) * 16 + l % 16];
    d4 acc = {0., 0., 0., 0.};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    for (int v = 0; v < 4; ++v) D[(4 * (l / 16) + v) * 16 + l % 16] = acc[v];
    // timing: 64 dependent, then 64 independent (4 accumulators) MFMAs
    d4 c0 = acc, c1 = acc, c2 = acc, c3 = acc;
    long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < 64; ++i) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    asm volatile("" :: "v"(c0));
    long long t1 = __builtin_readcyclecounter();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
    }
    asm volatile("" :: "v"(c0), "v"(c1), "v"(c2), "v"(c3));
    long long t2 = __builtin_readcyclecounter();
    if (l == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; }
    if (c0[0] + c1[1] + c2[2] + c3[3] == 12345.678) D[0] = 0.;
}
int main() {
    double hA[64], hB[64], hD[256], ref[256];
    for (int i = 0; i < 16; ++i) for (int kk = 0; kk < 4; ++kk) hA[i * 4 + kk] = 1 + i + 17 * kk;
    for (int kk = 0; kk < 4; ++kk) for (int j = 0; j < 16; ++j) hB[kk * 16 + j] = 2 + 3 * j - 5 * kk;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int kk = 0; kk < 4; ++kk) s += hA[i * 4 + kk] * hB[kk * 16 + j]; ref[i * 16 + j] = s; }
    double *A, *B, *D; long long *c, hc[2];
    hipMalloc(&A, sizeof(hA)); hipMalloc(&B, sizeof(hB)); hipMalloc(&D, sizeof(hD)); hipMalloc(&c, 16);
    hipMemcpy(A, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(B, hB, sizeof(hB), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, A, B, D, c);
    hipMemcpy(hD, D, sizeof(hD), hipMemcpyDeviceToHost); hipMemcpy(hc, c, 16, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; ++i) bad += hD[i] != ref[i];
    // which D element does (lane, register) hold?  search the reference for the value stored under the assumed layout
    for (int l : {0, 1, 15, 16, 17, 32, 48, 63}) {
        printf("lane %2d:", l);
        for (int v = 0; v < 4; ++v) {
            const double val = hD[(4 * (l / 16) + v) * 16 + l % 16];
            int fi = -1, fj = -1, cnt = 0;
            for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) if (ref[i * 16 + j] == val) { fi = i; fj = j; ++cnt; }
            printf("  v%d -> D[%d][%d]%s", v, fi, fj, cnt == 1 ? "" : "?");
        }
        printf("\n");
    }
    printf("layout A[l%%16][l/16], B[l/16][l%%16], D[4(l/16)+v][l%%16]: %d of 256 entries differ\n", bad);
    printf("64 dependent MFMAs: %lld cycles (%.1f each); 64 MFMAs on four accumulators: %lld cycles (%.1f each)\n", hc[0], hc[0] / 64., hc[1], hc[1] / 64.);
    return bad != 0;
}
