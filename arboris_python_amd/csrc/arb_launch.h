// arb_launch.h -- host side of ONE kernel launch: knobs, wave slots, the work queue's set-up, launch_one and the explicit
// instantiations of the translation units of the split build (included by arb_kernels.hip only).
#ifndef ARB_LAUNCH_H
#define ARB_LAUNCH_H
// ===========================================================================
// Host side: model upload, launch dispatch, C ABI
// ===========================================================================
// arb_step_plan asks the launch path itself: with a probe installed (per thread) launch_one describes the launch it would make
// -- instantiation, LDS, wave slots from the kernel's own register count, queue or not -- and returns before it allocates
// or launches anything.
struct LaunchProbe { int nmax, nsets, feat, cm, waves_compiled, waves_by_regs, wave_slots, work_queue; long lds_bytes; };
#ifdef ARB_PART
extern thread_local std::string g_hip_err;
extern thread_local LaunchProbe *g_probe;
#else
thread_local std::string g_hip_err;
thread_local LaunchProbe *g_probe = nullptr;
#endif

#define HIP_TRY(expr)                                                          \
    do {                                                                       \
        hipError_t e_ = (expr);                                                \
        if (e_ != hipSuccess) {                                                \
            g_hip_err = std::string(#expr) + ": " + hipGetErrorString(e_);     \
            return ARB_ERR_HIP;                                                \
        }                                                                      \
    } while (0)

// ---------------------------------------------------------------------------
// One launcher per (T, NMAX, NSETS, MODE).  The library is built from several translation units
// of this same file (csrc/Makefile): -DARB_PART_NMAX=<tile> -DARB_PART_T=<float|double> compiles the
// kernels of one register tile and precision only (explicit instantiations below) and none of the
// host code; the main unit declares them extern and holds the C ABI.
// ---------------------------------------------------------------------------
// Stream-ordered scratch (the work queue's flags, the split execution's hand-over buffers): a pool of this library's
// own per device that keeps what it is given back (release threshold = max), so a launch costs no driver allocation
// after the first; the default pool of the device -- whose settings belong to the application -- is the fallback.
// One definition, in the host unit: the kernel units of the split build call it.
hipError_t arb_scratch_alloc(void **p, size_t bytes, hipStream_t st);
#ifndef ARB_PART
hipError_t arb_scratch_alloc(void **p, size_t bytes, hipStream_t st) {
    static hipMemPool_t pools[64] = {};
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
        hipMemPool_t pool = nullptr;
        {
            std::lock_guard<std::mutex> lk(mu);
            if (pools[dev] == nullptr) {
                hipMemPoolProps props;
                memset(&props, 0, sizeof(props));
                props.allocType = hipMemAllocationTypePinned;
                props.location.type = hipMemLocationTypeDevice;
                props.location.id = dev;
                hipMemPool_t np = nullptr;
                if (hipMemPoolCreate(&np, &props) == hipSuccess) {
                    uint64_t keep = ~(uint64_t)0;
                    (void)hipMemPoolSetAttribute(np, hipMemPoolAttrReleaseThreshold, &keep);
                    pools[dev] = np;
                } else {
                    (void)hipGetLastError();
                }
            }
            pool = pools[dev];
        }
        if (pool != nullptr) {
            if (hipMallocFromPoolAsync(p, bytes, pool, st) == hipSuccess) return hipSuccess;
            (void)hipGetLastError();
        }
    }
    return hipMallocAsync(p, bytes, st);
}
#endif

// Development / test knobs of a handle (include/arbstep_hooks.h: arb_hook_set_knob).  The library reads NO environment
// variable (ABI 7); a build with -DARB_DEVELOPMENT (tools/quick_build.sh) fills them from ARB_<NAME> once, at
// arb_model_create.  queue_spin_cap: a TEST knob -- a small positive cap makes healthy launches report stalls (and skip
// worlds) whenever a producer is merely slow.
struct Knobs {
    // (queue_tail 6 since round 6: swept on the body-space-column kernels, tools/queue_sweep.py -- 4096 worlds +1.4 %, eight
    // contacts +1 %, 8192 / 65 536 worlds unchanged against 4; the schedule does not change a bit of the results)
    int lds_pad = 0, queue_chunk = 4, queue_tail = 6, queue_spin_cap = 1 << 24;
    int force_waves = 0, gsw_waves = 3, ablate = 0, wide_compact = 1, wide_gs_groups = 1;
};

// Wave slots of a device for one-wavefront workgroups of a kernel that runs `waves_per_simd` wavefronts per SIMD by its
// registers and asks for `lds_bytes` of LDS: ONE model behind the launch (queue grid, "more units than slots?"), the choice
// of the build (choose_build) and arb_step_plan.  The 160 KB of a CU are handed out in 128 granules of 1280 B
// (tools/lds_granule_probe.hip): hipOccupancyMaxActiveBlocksPerMultiprocessor divides 160 KB by the request instead and
// overestimates between the granule boundaries (twelve wavefronts per CU up to 12 800 B, not 13 653 B), which is why it is
// not asked.
static long slots_per_cu(int waves_per_simd, long lds_bytes) {
    const long by_lds = 128l / std::max(1l, (lds_bytes + 1279) / 1280);
    return std::max(1l, std::min((long)(4 * waves_per_simd), by_lds));
}
// wavefronts per SIMD a compiled kernel runs by its register allocation (512 registers per lane and SIMD, handed out in
// blocks of eight; at most eight wavefronts)
template <typename K>
static int kernel_waves_per_simd(K kern) {
    hipFuncAttributes fa;
    if (hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(kern)) != hipSuccess || fa.numRegs <= 0) { (void)hipGetLastError(); return 0; }
    return std::min(8, 512 / (((int)fa.numRegs + 7) / 8 * 8));
}
template <typename K>
static int wave_slots(K kern, size_t lds) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    const int wps = kernel_waves_per_simd(kern);
    if (wps <= 0) return 0;
    return (int)(cus * slots_per_cu(wps, (long)lds));
}

template <typename T, int NMAX, int NSETS, int MODE, int FEAT, int CM>
int launch_one(const DevModel<T> *dm, const Layout &L, T *q, T *dq, T *cf, const T *ext, const PerWorldPD<T> &pwd, long nw, double dt,
                      int nsteps, unsigned flags, const DebugOut<T> &dbg, int zmode, const LogOut<T> &logo,
                      const SplitIO<T> &sio, const double *dts, hipStream_t st, const Knobs &kn, long ext_stride, long pd_stride,
                      const CostIO<T> &cost) {
    auto kern = arb_step_kernel<T, NMAX, NSETS, MODE, FEAT, CM>;
    const size_t lds = (size_t)(MODE == 1 ? L.total_inspect : L.total) * sizeof(T) + (size_t)std::max(0, kn.lds_pad);
    if (g_probe != nullptr) {
        LaunchProbe &pr = *g_probe;
        pr.nmax = NMAX; pr.nsets = NSETS; pr.feat = FEAT; pr.cm = CM; pr.lds_bytes = (long)lds;
        pr.waves_compiled = ARB_KERNEL_WAVES(T, NMAX, NSETS, MODE, FEAT, CM);
        pr.waves_by_regs = kernel_waves_per_simd(kern);
        pr.wave_slots = wave_slots(kern, lds);
        pr.work_queue = (MODE == 0 && kn.queue_chunk > 0 && sio.mode == 0 && !(flags & ARB_STEP_STATIC_WORLDS) && nsteps >= 2 &&
                         (cf != nullptr || L.ndol == 0) && nw * (long)nsteps < (1l << 30) && pr.wave_slots > 0 && nw > pr.wave_slots) ? 1 : 0;
        return ARB_OK;
    }
    if (lds > 64 * 1024) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    // Work queue (see the kernel): multi-step launches of more worlds than the chip has wave slots.  Constraint forces
    // that persist from step to step travel between chunks through `cf`, so it must be there when the model has
    // constraints.  ARB_STEP_STATIC_WORLDS (or ARB_QUEUE_CHUNK=0 in the environment) keeps one workgroup per world.
    const int chunk = kn.queue_chunk;
    const int tail = std::max(0, std::min(kn.queue_tail, nsteps - 1));
    const int spin_cap = kn.queue_spin_cap;
    int *queue = nullptr;
    const long units = nw;                                   // work units: worlds
    unsigned grid = (unsigned)units;
    constexpr bool QUEUE_LOOP = ARB_QUEUE_LOOP && (ARB_QUEUE_LOOP_ALL || !((sizeof(T) == 8 || CM == 3) && NMAX == 64));     // (see the kernel)
    if (MODE == 0 && chunk > 0 && sio.mode == 0 && !(flags & ARB_STEP_STATIC_WORLDS) && nsteps >= 2 &&
        (cf != nullptr || L.ndol == 0) && nw * (long)nsteps < (1l << 30)) {
        // wave slots of this kernel on the current device, cached per thread for the last (device, LDS size) asked
        static thread_local size_t slots_lds = ~(size_t)0;
        static thread_local int slots_dev = -1, slots = 0;
        int dev = -1;
        (void)hipGetDevice(&dev);
        if (slots_lds != lds || slots_dev != dev) { slots = wave_slots(kern, lds); slots_lds = lds; slots_dev = dev; }
        if (slots > 0 && units > slots) {
            const size_t bytes = (size_t)(1 + units) * sizeof(int);
            if (arb_scratch_alloc(reinterpret_cast<void **>(&queue), bytes, st) == hipSuccess) {
                if (hipMemsetAsync(queue, 0, bytes, st) != hipSuccess) {
                    g_hip_err = "hipMemsetAsync(queue)";
                    (void)hipFreeAsync(queue, st);
                    return ARB_ERR_HIP;
                }
                // resident wavefronts that loop over items -- or one workgroup per item, looped by the dispatcher
                const int nbig = (nsteps - tail + chunk - 1) / chunk;
                grid = QUEUE_LOOP ? (unsigned)slots : (unsigned)(units * (long)(nbig + tail));
            } else {
                (void)hipGetLastError();
                queue = nullptr;
            }
        }
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVE), lds, st, dm, L, q, dq, cf, ext, pwd, nw, (T)dt, nsteps, flags, dbg, zmode, logo, sio, dts,
                       queue, chunk > 0 ? chunk : 1, tail, spin_cap, ext_stride, pd_stride, cost);
    const hipError_t le = hipGetLastError();
    if (queue != nullptr) {
        const hipError_t fe = hipFreeAsync(queue, st);         // (also after a failed launch: nothing leaks)
        if (le == hipSuccess && fe != hipSuccess) { g_hip_err = std::string("hipFreeAsync(queue): ") + hipGetErrorString(fe); return ARB_ERR_HIP; }
    }
    if (le != hipSuccess) { g_hip_err = std::string("kernel launch: ") + hipGetErrorString(le); return ARB_ERR_HIP; }
    return ARB_OK;
}

#if defined(ARB_PART_NMAX) && !defined(ARB_PART)
#error "define ARB_PART together with ARB_PART_NMAX / ARB_PART_T"
#endif
#define ARB_LAUNCH_ONE_ARGS(T)                                                                                             \
    const DevModel<T> *, const Layout &, T *, T *, T *, const T *, const PerWorldPD<T> &, long, double, int, unsigned,   \
    const DebugOut<T> &, int, const LogOut<T> &, const SplitIO<T> &, const double *, hipStream_t, const Knobs &, long, long, \
    const CostIO<T> &
#if defined(ARB_PART) && defined(ARB_PART_WIDE)      /* (the workgroup-per-world kernels: instantiated in arb_wide_kernel.h) */
#elif defined(ARB_PART) && defined(ARB_PART_SPEC)    /* (translation units of their own: the specialised kernels, tiles 44 / 48) */
#if ARB_PART_SPEC == 1         /* float32, one column set: two and three waves */
template int launch_one<float, ARB_PART_NMAX, 1, 0, 4, 0>(ARB_LAUNCH_ONE_ARGS(float));
template int launch_one<float, ARB_PART_NMAX, 1, 0, 5, 0>(ARB_LAUNCH_ONE_ARGS(float));
template int launch_one<float, ARB_PART_NMAX, 1, 0, 4, 2>(ARB_LAUNCH_ONE_ARGS(float));
template int launch_one<float, ARB_PART_NMAX, 1, 0, 5, 2>(ARB_LAUNCH_ONE_ARGS(float));
#elif ARB_PART_SPEC == 5       /* float32, body-space constraint columns (FEAT bit 16): plain / torques / every input, two and three waves; inspect */
template int launch_one<float, ARB_PART_NMAX, 1, 0, 20, 0>(ARB_LAUNCH_ONE_ARGS(float));
template int launch_one<float, ARB_PART_NMAX, 1, 0, 21, 0>(ARB_LAUNCH_ONE_ARGS(float));
template int launch_one<float, ARB_PART_NMAX, 1, 0, 20, 2>(ARB_LAUNCH_ONE_ARGS(float));
template int launch_one<float, ARB_PART_NMAX, 1, 0, 21, 2>(ARB_LAUNCH_ONE_ARGS(float));
#elif ARB_PART_SPEC == 8       /* float32, body-space columns compiled for four contacts (FEAT bit 32): the headline */
template int launch_one<float, ARB_PART_NMAX, 1, 0, 52, 0>(ARB_LAUNCH_ONE_ARGS(float));
template int launch_one<float, ARB_PART_NMAX, 1, 0, 53, 0>(ARB_LAUNCH_ONE_ARGS(float));
template int launch_one<float, ARB_PART_NMAX, 1, 0, 52, 2>(ARB_LAUNCH_ONE_ARGS(float));
template int launch_one<float, ARB_PART_NMAX, 1, 0, 53, 2>(ARB_LAUNCH_ONE_ARGS(float));
#elif ARB_PART_SPEC == 6
template int launch_one<float, ARB_PART_NMAX, 1, 0, 19, 0>(ARB_LAUNCH_ONE_ARGS(float));
template int launch_one<float, ARB_PART_NMAX, 1, 0, 19, 2>(ARB_LAUNCH_ONE_ARGS(float));
template int launch_one<float, ARB_PART_NMAX, 1, 1, 19, 0>(ARB_LAUNCH_ONE_ARGS(float));
#elif ARB_PART_SPEC == 7       /* float64, body-space constraint columns */
template int launch_one<double, ARB_PART_NMAX, 1, 0, 20, 0>(ARB_LAUNCH_ONE_ARGS(double));
template int launch_one<double, ARB_PART_NMAX, 1, 0, 21, 0>(ARB_LAUNCH_ONE_ARGS(double));
template int launch_one<double, ARB_PART_NMAX, 1, 0, 19, 0>(ARB_LAUNCH_ONE_ARGS(double));
template int launch_one<double, ARB_PART_NMAX, 1, 1, 19, 0>(ARB_LAUNCH_ONE_ARGS(double));
#elif ARB_PART_SPEC == 3       /* float64, one column set */
template int launch_one<double, ARB_PART_NMAX, 1, 0, 4, 0>(ARB_LAUNCH_ONE_ARGS(double));
template int launch_one<double, ARB_PART_NMAX, 1, 0, 5, 0>(ARB_LAUNCH_ONE_ARGS(double));
#else                          /* float32, no constraints (FEAT bit 8): two and three waves */
template int launch_one<float, ARB_PART_NMAX, 1, 0, 8, 0>(ARB_LAUNCH_ONE_ARGS(float));
template int launch_one<float, ARB_PART_NMAX, 1, 0, 9, 0>(ARB_LAUNCH_ONE_ARGS(float));
template int launch_one<float, ARB_PART_NMAX, 1, 0, 8, 2>(ARB_LAUNCH_ONE_ARGS(float));
template int launch_one<float, ARB_PART_NMAX, 1, 0, 9, 2>(ARB_LAUNCH_ONE_ARGS(float));
#endif
#elif defined(ARB_PART) && defined(ARB_PART_MIXED)      /* (translation units of their own: the mixed-precision build, CM 3, every tile) */
template int launch_one<float, ARB_PART_NMAX, 1, 0, 0, 3>(ARB_LAUNCH_ONE_ARGS(float));
template int launch_one<float, ARB_PART_NMAX, 2, 0, 0, 3>(ARB_LAUNCH_ONE_ARGS(float));
template int launch_one<float, ARB_PART_NMAX, 1, 0, 1, 3>(ARB_LAUNCH_ONE_ARGS(float));
template int launch_one<float, ARB_PART_NMAX, 2, 0, 1, 3>(ARB_LAUNCH_ONE_ARGS(float));
template int launch_one<float, ARB_PART_NMAX, 1, 0, 3, 3>(ARB_LAUNCH_ONE_ARGS(float));
template int launch_one<float, ARB_PART_NMAX, 2, 0, 3, 3>(ARB_LAUNCH_ONE_ARGS(float));
#elif defined(ARB_PART)
template int launch_one<ARB_PART_T, ARB_PART_NMAX, 1, 0, 0, 0>(ARB_LAUNCH_ONE_ARGS(ARB_PART_T));
template int launch_one<ARB_PART_T, ARB_PART_NMAX, 2, 0, 0, 0>(ARB_LAUNCH_ONE_ARGS(ARB_PART_T));
template int launch_one<ARB_PART_T, ARB_PART_NMAX, 1, 0, 1, 0>(ARB_LAUNCH_ONE_ARGS(ARB_PART_T));
template int launch_one<ARB_PART_T, ARB_PART_NMAX, 2, 0, 1, 0>(ARB_LAUNCH_ONE_ARGS(ARB_PART_T));
template int launch_one<ARB_PART_T, ARB_PART_NMAX, 1, 0, 3, 0>(ARB_LAUNCH_ONE_ARGS(ARB_PART_T));
template int launch_one<ARB_PART_T, ARB_PART_NMAX, 2, 0, 3, 0>(ARB_LAUNCH_ONE_ARGS(ARB_PART_T));
template int launch_one<ARB_PART_T, ARB_PART_NMAX, 1, 1, 3, 0>(ARB_LAUNCH_ONE_ARGS(ARB_PART_T));
template int launch_one<ARB_PART_T, ARB_PART_NMAX, 2, 1, 3, 0>(ARB_LAUNCH_ONE_ARGS(ARB_PART_T));
#if ARB_PART_IS_FLOAT
template int launch_one<ARB_PART_T, ARB_PART_NMAX, 1, 0, 3, 1>(ARB_LAUNCH_ONE_ARGS(ARB_PART_T));
template int launch_one<ARB_PART_T, ARB_PART_NMAX, 2, 0, 3, 1>(ARB_LAUNCH_ONE_ARGS(ARB_PART_T));
#if ARB_PART_NMAX >= 44 && ARB_PART_NMAX <= 48      /* (the 16- and 32-row tiles: two waves are faster at every batch size, see choose_build) */
template int launch_one<ARB_PART_T, ARB_PART_NMAX, 1, 0, 0, 2>(ARB_LAUNCH_ONE_ARGS(ARB_PART_T));
template int launch_one<ARB_PART_T, ARB_PART_NMAX, 1, 0, 1, 2>(ARB_LAUNCH_ONE_ARGS(ARB_PART_T));
template int launch_one<ARB_PART_T, ARB_PART_NMAX, 1, 0, 3, 2>(ARB_LAUNCH_ONE_ARGS(ARB_PART_T));
#endif
#endif
#endif
#ifndef ARB_PART
#if defined(ARB_SPLIT_BUILD)
#define ARB_EXTERN_TILE(T, NM)                                                        \
    extern template int launch_one<T, NM, 1, 0, 0, 0>(ARB_LAUNCH_ONE_ARGS(T));        \
    extern template int launch_one<T, NM, 2, 0, 0, 0>(ARB_LAUNCH_ONE_ARGS(T));        \
    extern template int launch_one<T, NM, 1, 0, 1, 0>(ARB_LAUNCH_ONE_ARGS(T));        \
    extern template int launch_one<T, NM, 2, 0, 1, 0>(ARB_LAUNCH_ONE_ARGS(T));        \
    extern template int launch_one<T, NM, 1, 0, 3, 0>(ARB_LAUNCH_ONE_ARGS(T));        \
    extern template int launch_one<T, NM, 2, 0, 3, 0>(ARB_LAUNCH_ONE_ARGS(T));        \
    extern template int launch_one<T, NM, 1, 1, 3, 0>(ARB_LAUNCH_ONE_ARGS(T));        \
    extern template int launch_one<T, NM, 2, 1, 3, 0>(ARB_LAUNCH_ONE_ARGS(T));
#define ARB_EXTERN_TILE_CM(NM)                                                          \
    extern template int launch_one<float, NM, 1, 0, 3, 1>(ARB_LAUNCH_ONE_ARGS(float));  \
    extern template int launch_one<float, NM, 2, 0, 3, 1>(ARB_LAUNCH_ONE_ARGS(float));
ARB_EXTERN_TILE_CM(16) ARB_EXTERN_TILE_CM(32) ARB_EXTERN_TILE_CM(44) ARB_EXTERN_TILE_CM(48) ARB_EXTERN_TILE_CM(64)
#undef ARB_EXTERN_TILE_CM
#define ARB_EXTERN_TILE_W3(NM)                                                          \
    extern template int launch_one<float, NM, 1, 0, 0, 2>(ARB_LAUNCH_ONE_ARGS(float));  \
    extern template int launch_one<float, NM, 1, 0, 1, 2>(ARB_LAUNCH_ONE_ARGS(float));  \
    extern template int launch_one<float, NM, 1, 0, 3, 2>(ARB_LAUNCH_ONE_ARGS(float));
ARB_EXTERN_TILE_W3(44) ARB_EXTERN_TILE_W3(48)
#undef ARB_EXTERN_TILE_W3
#define ARB_EXTERN_TILE_MIXED(NM)                                                       \
    extern template int launch_one<float, NM, 1, 0, 0, 3>(ARB_LAUNCH_ONE_ARGS(float));  \
    extern template int launch_one<float, NM, 2, 0, 0, 3>(ARB_LAUNCH_ONE_ARGS(float));  \
    extern template int launch_one<float, NM, 1, 0, 1, 3>(ARB_LAUNCH_ONE_ARGS(float));  \
    extern template int launch_one<float, NM, 2, 0, 1, 3>(ARB_LAUNCH_ONE_ARGS(float));  \
    extern template int launch_one<float, NM, 1, 0, 3, 3>(ARB_LAUNCH_ONE_ARGS(float));  \
    extern template int launch_one<float, NM, 2, 0, 3, 3>(ARB_LAUNCH_ONE_ARGS(float));
ARB_EXTERN_TILE_MIXED(16) ARB_EXTERN_TILE_MIXED(32) ARB_EXTERN_TILE_MIXED(44) ARB_EXTERN_TILE_MIXED(48) ARB_EXTERN_TILE_MIXED(64)
#undef ARB_EXTERN_TILE_MIXED
#if ARB_WITH_SPEC
#define ARB_EXTERN_TILE_SPEC(NM)                                                        \
    extern template int launch_one<float, NM, 1, 0, 4, 0>(ARB_LAUNCH_ONE_ARGS(float));  \
    extern template int launch_one<float, NM, 1, 0, 5, 0>(ARB_LAUNCH_ONE_ARGS(float));  \
    extern template int launch_one<float, NM, 1, 0, 4, 2>(ARB_LAUNCH_ONE_ARGS(float));  \
    extern template int launch_one<float, NM, 1, 0, 5, 2>(ARB_LAUNCH_ONE_ARGS(float));  \
    extern template int launch_one<float, NM, 1, 0, 20, 0>(ARB_LAUNCH_ONE_ARGS(float)); \
    extern template int launch_one<float, NM, 1, 0, 21, 0>(ARB_LAUNCH_ONE_ARGS(float)); \
    extern template int launch_one<float, NM, 1, 0, 20, 2>(ARB_LAUNCH_ONE_ARGS(float)); \
    extern template int launch_one<float, NM, 1, 0, 21, 2>(ARB_LAUNCH_ONE_ARGS(float)); \
    extern template int launch_one<float, NM, 1, 0, 52, 0>(ARB_LAUNCH_ONE_ARGS(float)); \
    extern template int launch_one<float, NM, 1, 0, 53, 0>(ARB_LAUNCH_ONE_ARGS(float)); \
    extern template int launch_one<float, NM, 1, 0, 52, 2>(ARB_LAUNCH_ONE_ARGS(float)); \
    extern template int launch_one<float, NM, 1, 0, 53, 2>(ARB_LAUNCH_ONE_ARGS(float)); \
    extern template int launch_one<float, NM, 1, 0, 19, 0>(ARB_LAUNCH_ONE_ARGS(float)); \
    extern template int launch_one<float, NM, 1, 0, 19, 2>(ARB_LAUNCH_ONE_ARGS(float)); \
    extern template int launch_one<float, NM, 1, 1, 19, 0>(ARB_LAUNCH_ONE_ARGS(float)); \
    extern template int launch_one<double, NM, 1, 0, 20, 0>(ARB_LAUNCH_ONE_ARGS(double)); \
    extern template int launch_one<double, NM, 1, 0, 21, 0>(ARB_LAUNCH_ONE_ARGS(double)); \
    extern template int launch_one<double, NM, 1, 0, 19, 0>(ARB_LAUNCH_ONE_ARGS(double)); \
    extern template int launch_one<double, NM, 1, 1, 19, 0>(ARB_LAUNCH_ONE_ARGS(double)); \
    extern template int launch_one<double, NM, 1, 0, 4, 0>(ARB_LAUNCH_ONE_ARGS(double)); \
    extern template int launch_one<double, NM, 1, 0, 5, 0>(ARB_LAUNCH_ONE_ARGS(double)); \
    extern template int launch_one<float, NM, 1, 0, 8, 0>(ARB_LAUNCH_ONE_ARGS(float));  \
    extern template int launch_one<float, NM, 1, 0, 9, 0>(ARB_LAUNCH_ONE_ARGS(float));  \
    extern template int launch_one<float, NM, 1, 0, 8, 2>(ARB_LAUNCH_ONE_ARGS(float));  \
    extern template int launch_one<float, NM, 1, 0, 9, 2>(ARB_LAUNCH_ONE_ARGS(float));
ARB_EXTERN_TILE_SPEC(44) ARB_EXTERN_TILE_SPEC(48)
#undef ARB_EXTERN_TILE_SPEC
#endif
ARB_EXTERN_TILE(float, 16) ARB_EXTERN_TILE(float, 32) ARB_EXTERN_TILE(float, 44) ARB_EXTERN_TILE(float, 48) ARB_EXTERN_TILE(float, 64)
ARB_EXTERN_TILE(double, 16) ARB_EXTERN_TILE(double, 32) ARB_EXTERN_TILE(double, 44) ARB_EXTERN_TILE(double, 48) ARB_EXTERN_TILE(double, 64)
#undef ARB_EXTERN_TILE
#endif  // ARB_SPLIT_BUILD
#endif  // !ARB_PART
#endif  // ARB_LAUNCH_H
