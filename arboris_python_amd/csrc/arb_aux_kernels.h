// arb_aux_kernels.h -- the sweep kernel of the split execution and the device unit-test kernels of the local solve
// (included by arb_kernels.hip only).
#ifndef ARB_AUX_KERNELS_H
#define ARB_AUX_KERNELS_H
// ===========================================================================
// Gauss-Seidel with one WAVEFRONT per world as its own kernel (split execution, ARB_STEP_SPLIT_WAVE): the
// quad-local sweeps of gs_stage, fed from the SplitIO buffers.  The sweeps are one dependent chain per world
// and need few registers, so this kernel is compiled for several waves per SIMD (the step kernel is pinned
// at two by its 256 VGPRs): other worlds' chains fill the issue slots one chain leaves empty.
// LDS per world: Y' (ndol^2) + the constraint blocks + 64 elements of scratch.
// ===========================================================================
template <typename T, int WV>
__global__ __launch_bounds__(WAVE, WV) void arb_gsw_kernel(
    const DevModel<T> *__restrict__ mp, const T *__restrict__ wsA, const T *__restrict__ wsv,
    T *__restrict__ wsf, const T *__restrict__ wsc, long nworlds, T dt_in, const double *__restrict__ dts)
{
    const int lane = threadIdx.x;
    const long w = blockIdx.x;
    if (w >= nworlds) return;
    const int nc = mp->nc, ndol = mp->ndol;
    const T dt = dts != nullptr ? (T)dts[0] : dt_in;
    const T inv_dt = T(1) / dt;
    T *lds = reinterpret_cast<T *>(arb_lds_raw);
    auto al = [](int x) { return (x + 3) & ~3; };
    T *AM = lds, *CD = AM + al(ndol * ndol), *VV = CD + al(nc * CD_STRIDE), *FF = VV + al(ndol), *WORK = FF + al(ndol);
    const int nA = ndol * ndol;
    for (int i = lane; i < nA; i += WAVE) AM[i] = wsA[w * nA + i];
    if (lane < ndol) { VV[lane] = wsv[w * ndol + lane]; FF[lane] = wsf[w * ndol + lane]; }
    if (lane < nc) {
        const T *cs = wsc + (w * nc + lane) * 8;
        T *cd = CD + lane * CD_STRIDE;
        cd[CD_ACTIVE] = cs[0]; cd[CD_SDIST] = cs[1]; cd[CD_POS0] = cs[2]; cd[CD_POS0 + 1] = cs[3]; cd[CD_POS0 + 2] = cs[4];
    }
    WAVE_SYNC();
    DebugOut<T> nodbg;
    nodbg.gs_stats = nullptr; nodbg.gs_trace = nullptr; nodbg.ablate = 0;
    gs_stage<T, 0>(mp, lane, nc, ndol, ndol, dt, inv_dt, AM, CD, VV, FF, WORK, nodbg, w);
    if (lane < ndol) wsf[w * ndol + lane] = FF[lane];
}

// ===========================================================================
// Element-wise conversion between float32 and float64 buffers (the promotion of float32 launches of models only float64
// can carry, step_promoted in arb_kernels.hip): grid-stride, coalesced.
// ===========================================================================
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void arb_cvt_kernel(const TI *__restrict__ in, TO *__restrict__ out, size_t count)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) out[i] = (TO)in[i];
}

// ===========================================================================
// Device unit test of the local solve (test hook arb_dev_softfinger_solve): one LANE per input tuple, the
// same arb_math.h code the kernels run -- inverse of the 4x4 block, SoftFingerContact.solve with the fast
// sliding shift or the eig6 fallback on a lane-private LDS work array.
// in: [n][27] = vel 4 | adm 16 | force 4 | sdist, dt, mu ;  out: [n][9] = force 4 | dforce 4 | branch
// ===========================================================================
template <typename T>
__global__ __launch_bounds__(WAVE) void arb_softfinger_test_kernel(const double *__restrict__ in, double *__restrict__ out,
                                                                   int n, int use_fast)
{
    T *lds = reinterpret_cast<T *>(arb_lds_raw);
    const int lane = threadIdx.x;
    const int i = blockIdx.x * WAVE + lane;
    T *work = lds + lane * 41;
    if (i >= n) return;
    const double *t = in + (size_t)i * 27;
    T v[4], Y[16], P[16], f[4], df[4], eps[3] = {T(1), T(1), T(1)};
    for (int k = 0; k < 4; ++k) { v[k] = (T)t[k]; f[k] = (T)t[20 + k]; }
    for (int k = 0; k < 16; ++k) Y[k] = (T)t[4 + k];
    if (!inv_block<T>(Y, 4, 4, P)) pinv_block<T>(Y, 4, 4, P);
    const int br = softfinger_solve<T>(v, Y, P, f, df, (T)t[24], (T)t[25], (T)t[26], eps, work, use_fast != 0);
    double *o = out + (size_t)i * 9;
    for (int k = 0; k < 4; ++k) { o[k] = (double)f[k]; o[4 + k] = (double)df[k]; }
    o[8] = (double)br;
}

// Device unit test of the wavefront's eig6 (test hook arb_dev_eig6_pair): one wavefront per 6x6 matrix, the one-lane
// routine and the wavefront routine side by side.  out: [n][28] = shift, nfound, wr 6, wi 6 of eig6 | the same of eig6_wave
template <typename T>
__global__ __launch_bounds__(WAVE) void arb_eig6_test_kernel(const double *__restrict__ in, double *__restrict__ out, int n)
{
    T *lds = reinterpret_cast<T *>(arb_lds_raw);
    const int lane = threadIdx.x;
    const int i = blockIdx.x;
    if (i >= n) return;
    T *w0 = lds, *w1 = lds + 48;
    if (lane < 36) { w0[lane] = (T)in[(size_t)i * 36 + lane]; w1[lane] = w0[lane]; }
    WAVE_SYNC();
    double *o = out + (size_t)i * 28;
    const T sw = slide_shift_from_eig_wave<T>(w1, lane);
    {
        T wr[6] = {T(0), T(0), T(0), T(0), T(0), T(0)}, wi[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
        const int nf = eig6_wave<T>(lane < 36 ? w1[lane] : T(0), lane, wr, wi);
        if (lane == 5) {
            o[14] = (double)sw; o[15] = (double)nf;
            for (int k = 0; k < 6; ++k) { o[16 + k] = (double)wr[k]; o[22 + k] = (double)wi[k]; }
        }
    }
    WAVE_SYNC();
    if (lane == 0) {
        T wr[6] = {T(0), T(0), T(0), T(0), T(0), T(0)}, wi[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
        const int nf = eig6<T>(w0, wr, wi);
        for (int k = 0; k < 36; ++k) w0[k] = (T)in[(size_t)i * 36 + k];
        o[0] = (double)slide_shift_from_eig<T>(w0); o[1] = (double)nf;
        for (int k = 0; k < 6; ++k) { o[2 + k] = (double)wr[k]; o[8 + k] = (double)wi[k]; }
    }
}
#endif  // ARB_AUX_KERNELS_H
