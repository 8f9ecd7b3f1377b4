// arb_device.h -- lane exchange (readlane / DPP), lane-dense execution helpers, small LDS load / store helpers and the
// wavefront barrier the kernels hand LDS over with (included by arb_kernels.hip only).
#ifndef ARB_DEVICE_H
#define ARB_DEVICE_H
// ---------------------------------------------------------------------------
__device__ __forceinline__ float bcast(float x, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), lane));
}
__device__ __forceinline__ double bcast(double x, int lane) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}

// x moved across lanes by a DPP control (row_shr:n = 0x110 + n, row_bcast:15 = 0x142, row_bcast:31 = 0x143);
// lanes without a source, or in rows outside ROW_MASK, get 0.0
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double x) {
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, ROW_MASK, 0xF, ROW_MASK == 0xF);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xF, ROW_MASK == 0xF);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// x of lane I of the caller's quad (lanes 4k .. 4k+3), a DPP quad_perm operand: no SGPR round trip
template <int I>
__device__ __forceinline__ float quad_bcast(float x) {
    const int b = __float_as_int(x);
    return __int_as_float(__builtin_amdgcn_update_dpp(b, b, I * 0x55, 0xF, 0xF, true));
}
template <int I>
__device__ __forceinline__ double quad_bcast(double x) {
    const long long b = __double_as_longlong(x);
    const int lo = __builtin_amdgcn_update_dpp((int)b, (int)b, I * 0x55, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp((int)(b >> 32), (int)(b >> 32), I * 0x55, 0xF, 0xF, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// Lane-dense execution (round 4).  tools/exec_mask_probe.hip: a wavefront whose EXEC mask has 8 or fewer lanes set issues
// a vector instruction every 15 cycles (independent) / 17-26 cycles (dependent chain) instead of every 4.7 / 8.1 -- float32,
// float64 and DPP alike, whatever the position of the lanes, 12 lanes or more run at full speed, and a co-resident dense
// wave is not slowed down.  The step kernel had such sparse regions all over phase A (one tree level = 1-4 bodies, one
// joint type = 1-7 bodies, 4 contacts, the one FreeJoint of phase E).  They now run on ALL lanes -- lanes without work
// compute on clamped indices, their results are never stored -- and only the stores stay predicated.  `keep` marks a
// value as used by every enabled lane at that point, so that the compiler cannot sink its computation into the
// predicated store block that follows.  Same arithmetic on the lanes that count: bit-identical results.
#ifndef ARB_DENSE
#define ARB_DENSE 0x7f      // bit mask (development): 1 level loops, 2 phase A', 4 FreeJoint integration, 8 gvel add, 16 block inverses, 32 own columns, 64 sin/cos
#endif
#define ARB_DENSE_LVL (ARB_DENSE & 1)
#define ARB_DENSE_AP (ARB_DENSE & 2)
#define ARB_DENSE_FJ (ARB_DENSE & 4)
#define ARB_DENSE_GV (ARB_DENSE & 8)
#define ARB_DENSE_INV (ARB_DENSE & 16)
#define ARB_DENSE_COL (ARB_DENSE & 32)
#define ARB_DENSE_SC (ARB_DENSE & 64)
__device__ __forceinline__ void keep(float x) { asm volatile("" :: "v"(x)); }
__device__ __forceinline__ void keep(double x) { asm volatile("" :: "v"(x)); }
__device__ __forceinline__ void keep(int x) { asm volatile("" :: "v"(x)); }
template <typename T> __device__ __forceinline__ void keep(V3<T> v) { keep(v.x); keep(v.y); keep(v.z); }
template <typename T> __device__ __forceinline__ void keep(const M3<T> &m) {
#pragma unroll
    for (int i = 0; i < 9; ++i) keep(m.a[i]);
}

template <typename T> __device__ __forceinline__ M3<T> ld_m3(const T *p) {
    M3<T> r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.a[i] = p[i];
    return r;
}
template <typename T> __device__ __forceinline__ V3<T> ld_v3(const T *p) { return v3<T>(p[0], p[1], p[2]); }
template <typename T> __device__ __forceinline__ void st_m3(T *p, const M3<T> &m) {
#pragma unroll
    for (int i = 0; i < 9; ++i) p[i] = m.a[i];
}
template <typename T> __device__ __forceinline__ void st_v3(T *p, V3<T> v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; }

template <typename TO, typename TI> __device__ __forceinline__ M3<TO> cvt_m3(const M3<TI> &m) {
    M3<TO> r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.a[i] = (TO)m.a[i];
    return r;
}
template <typename TO, typename TI> __device__ __forceinline__ V3<TO> cvt_v3(V3<TI> v) {
    return v3<TO>((TO)v.x, (TO)v.y, (TO)v.z);
}
template <typename TO, typename TI> __device__ __forceinline__ M3<TO> ld_m3_as(const TI *p) {
    M3<TO> r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.a[i] = (TO)p[i];
    return r;
}
template <typename TO, typename TI> __device__ __forceinline__ V3<TO> ld_v3_as(const TI *p) {
    return v3<TO>((TO)p[0], (TO)p[1], (TO)p[2]);
}

// y = M x for a row-major 6x6 M (wave-uniform address -> scalar loads)
template <typename T>
__device__ __forceinline__ void mat6_vec(const T *__restrict__ M, const T x[6], T y[6]) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        T s = T(0);
#pragma unroll
        for (int j = 0; j < 6; ++j) s += M[6 * i + j] * x[j];
        y[i] = s;
    }
}

// f(integral_constant<int, N-1>), ..., f(integral_constant<int, 0>): a loop whose index is a constant expression
template <int... I, typename F>
__device__ __forceinline__ void static_for_asc(std::integer_sequence<int, I...>, F &&f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int... I, typename F>
__device__ __forceinline__ void static_for_desc(std::integer_sequence<int, I...>, F &&f) {
    (f(std::integral_constant<int, (int)sizeof...(I) - 1 - I>{}), ...);
}

extern __shared__ __attribute__((aligned(16))) unsigned char arb_lds_raw[];

// One workgroup = one wavefront.  The DS (LDS) instructions of a wave are executed in issue
// order, so a value written by one lane is seen by any lane's later ds_read without a
// hardware barrier and without waiting for the write to retire.  The only thing to prevent is
// the COMPILER moving LDS accesses across the hand-off points: an empty asm with a memory
// clobber plus the wave_barrier scheduling fence does that and emits no instruction (a
// workgroup-scope fence would add s_waitcnt lgkmcnt(0) = a full drain of LDS and scalar loads
// at every hand-off; __syncthreads() additionally drains global loads and executes s_barrier).
#define WAVE_SYNC() do { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); \
                         asm volatile("" ::: "memory"); } while (0)
#endif  // ARB_DEVICE_H
