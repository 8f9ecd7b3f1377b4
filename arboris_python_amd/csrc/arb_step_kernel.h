// arb_step_kernel.h -- arb_step_kernel: one wavefront advances one world (work queue, state load, the step loop, the
// sweeps' call, phase E = integrate, state store).  The phases of the step loop are the fragments arb_phase_*.h
// (included by arb_kernels.hip only).
#ifndef ARB_STEP_KERNEL_H
#define ARB_STEP_KERNEL_H
// ===========================================================================
// The step kernel.  MODE 0 = production, 1 = inspect (debug stores, no state
// write-back).  zmode (inspect only): 0 full Z, 1 M only, 2 B only, 3 N only.
// ===========================================================================
// FEAT (bit mask) 0 = the plain step (arb_step without user torques: no per-world PD inputs, no per-step logs, no split
// execution) -- those arguments are compiled out, which keeps their kernargs and the predicates derived from them
// out of the SGPR file; bit 0 = user torques (ext_gforce: the MPC rollouts' input, one extra load per item);
// bit 1 = every other optional input (per-world PD, logs, split execution, per-step dt, flags); 3 = all of them.
// CM 1 = phase C eliminates on the matrix cores (float32 only; ARB_STEP_MFMA_ELIM), 0 = on the vector ALU;
// CM 2 = as 0, compiled for three waves per SIMD (float32, one column set, tiles up to 48 rows; see ARB_WAVES).
// (float64 worlds on the 64-row tile -- snake-64 -- need 36 KB of LDS per wave: four waves per CU, one per SIMD, so
// their kernels may take the whole 512-entry register file of a SIMD instead of spilling at 256)
template <typename T, int NMAX, int NSETS, int MODE, int FEAT, int CM>
// (float64 / 64 rows: two column sets need the whole register file of a SIMD; one column set fits 256 registers and must
// stay there -- two wavefronts per SIMD, five per CU with snake-64's LDS -- whatever else is compiled into the kernel: at
// 258 registers config 4 ran at 12.2 instead of 13.1 M world-steps/s)
__global__ __launch_bounds__(WAVE, ARB_KERNEL_WAVES(T, NMAX, NSETS, MODE, FEAT, CM)) void arb_step_kernel(
    const DevModel<T> *__restrict__ mp_in, const Layout L, T *__restrict__ gq_in, T *__restrict__ gdq_in,
    T *__restrict__ gcforce_in, const T *__restrict__ gext_in, const PerWorldPD<T> pwd_in, long nworlds, T dt_in, int nsteps,
    unsigned flags_in, const DebugOut<T> dbg, int zmode, const LogOut<T> logo_in, const SplitIO<T> sio_in,
    const double *__restrict__ dts_in, int *__restrict__ queue_in, int queue_chunk, int queue_tail, int queue_spin_cap,
    const long ext_stride_in, const long pd_stride_in, const CostIO<T> cost_in)
{
    static_assert(MODE == 0 || FEAT == 3 || FEAT == 19, "the inspect kernels take every input");
    static_assert(CM != 1 || (FEAT == 3 && MODE == 0 && std::is_same<T, float>::value), "matrix-core elimination: float32 step kernels");
    static_assert(CM != 2 || (MODE == 0 && NSETS == 1 && NMAX <= 48 && std::is_same<T, float>::value), "three-wave build: float32, one column set");
    // CM 3 (round 6, ARB_STEP_MIXED): float32 state buffers and LDS, the register tile [Z | rhs | J'^T] of phases C / D in float64
    // (the assembly of Z is float64 in every kernel; here it is no longer rounded to float32 before it is eliminated).  The
    // reference inverts Z in float64 (core.py:818); pivot-free float32 elimination loses log2(Z_jj / pivot) bits per pivot --
    // 17-20 of 24 on a 64-link chain.  Deep trees chain poses, twists and accelerations in log2(depth) rounds like the
    // float64 kernels (the level loop of a 64-link chain is a third of the step).
    static_assert(CM != 3 || (MODE == 0 && std::is_same<T, float>::value && !(FEAT & 28)), "mixed build: float32 step kernels, general model");
    static_assert(CM >= 0 && CM <= 3, "builds: 0 two waves, 1 matrix-core elimination, 2 three waves, 3 mixed precision");
    constexpr bool WIDE_REGS = (sizeof(T) == 8 || CM == 3);     // the register tile is float64
    constexpr bool FEAT_EXT = (FEAT & 1) != 0, FEAT_ALL = (FEAT & 2) != 0;
    // FEAT bit 4 (round 4): the kernel specialised for the model class of the headline workload -- exactly four constraints (eight
    // for a model with two column sets: human36 with the reference's eight contact points), every one an enabled
    // SoftFingerContact of a plane / sphere (or point) pair, and neither a PD controller nor joint viscosity in the model
    // (arb_model::spec_ok, checked by the host): the constraint type, the shape pair, nc and ndol are compile-time constants
    // and the code of the absent features is not compiled in.  Same expressions: bit-identical results.  (Measured, float32,
    // 4096 worlds: constants +3 %, without the viscosity / PD / warm-start code +8 %.  One by one in the general kernel:
    // viscosity +3 % -- its block was the FIRST term of phase B's accumulators, see there --, PD -1 %, warm start 0 %.)
    // FEAT bit 8: the same for models WITHOUT constraints (BASELINE config 2: human36 in free motion) -- nc = 0 is a constant,
    // phases A', D, the sweeps and the constraint columns of phase C are not compiled in.
    // FEAT bit 16 (round 5): BODY-SPACE constraint columns -- the same model class as bit 4 with ANY number of contacts on up
    // to ARB_MAXPAIR pairs of bodies (human36 with the reference's eight contact points, tests/test_human36_falling.py:32: two
    // feet): the augmented system carries the six columns Y J_p^T of every pair instead of the 4 nc columns Y J'^T (see
    // DevModel::nbp), so the model fits ONE column set; Y' = T (J_p Y J_p^T) T^T and v' = T J_p Y rhs are formed from the
    // 6 nbp x 6 nbp body-space admittance after phase D, phase E applies Y J_p^T (sum of T_c^T f_c).  nc is a run-time value
    // here.  Inspect kernels of such a model (MODE 1, FEAT 19) run the same arithmetic.
    constexpr bool BODYCOL = (FEAT & 16) != 0;
    constexpr bool SPEC = (FEAT & 12) != 0 || BODYCOL;
#ifndef ARB_BC_NC
#define ARB_BC_NC 0           // development: the BODYCOL kernels compiled for this number of contacts (0: a run-time value)
#endif
    // FEAT bit 5 (32, round 6): the BODYCOL kernels compiled for exactly FOUR contacts (plain inputs 52, user torques 53): since
    // round 6 body-space columns are the default of every qualifying model, so the headline model (four contacts) runs them,
    // and a compile-time nc is worth 1.2 % there (measured in round 5, ARB_BC_NC; not worth the builds while it was opt-in)
    constexpr int BC_NC = (FEAT & 32) ? 4 : ARB_BC_NC;
    constexpr bool NC_CONST = ((FEAT & 12) != 0 && !BODYCOL) || (BODYCOL && BC_NC > 0 && MODE == 0);       // nc, ndol compile-time constants
    constexpr int SPEC_NC = BODYCOL ? BC_NC : (FEAT & 8) ? 0 : 4 * NSETS;
    static_assert(!NC_CONST || (!FEAT_ALL && MODE == 0 && (CM == 0 || CM == 2)), "specialised kernels: plain inputs / user torques");
    static_assert(!BODYCOL || (NSETS == 1 && (CM == 0 || CM == 2) && (FEAT == 20 || FEAT == 21 || FEAT == 19 || FEAT == 52 || FEAT == 53)),
                  "body-space columns: one column set; plain inputs (20), user torques (21), every optional input / inspect (19); + 32: four contacts");
    static_assert(!(FEAT & 32) || BODYCOL, "FEAT bit 32 qualifies the body-space-column kernels");
    static_assert((FEAT & 12) != 12 && (!(FEAT & 8) || NSETS == 1), "specialised kernels: one model class at a time");
    const T *__restrict__ gext = FEAT_EXT ? gext_in : nullptr;
    // ABI 7: control inputs that change along the horizon -- step t reads row t of [nsteps][nworlds][ndof] arrays (stride 0:
    // one row for the whole launch) -- and the running cost of the rollout; both travel with the user torques (FEAT bit 0),
    // the per-step PD targets with the other optional inputs (bit 1)
    const long ext_stride = FEAT_EXT ? ext_stride_in : 0l, pd_stride = FEAT_ALL ? pd_stride_in : 0l;
    const CostIO<T> cost = FEAT_EXT ? cost_in : CostIO<T>{nullptr, nullptr, nullptr, nullptr, nullptr};
    const PerWorldPD<T> pwd = FEAT_ALL ? pwd_in : PerWorldPD<T>{nullptr, nullptr, nullptr, nullptr, nullptr};
    const LogOut<T> logo = FEAT_ALL ? logo_in : LogOut<T>{nullptr, nullptr, nullptr};
    const SplitIO<T> sio = FEAT_ALL ? sio_in : SplitIO<T>{0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    const unsigned flags = FEAT_ALL ? flags_in : 0u;
    // per-step dt (core.py:1357: dt = next_time - current_time), or null = dt_in for every step
    const double *__restrict__ dts = FEAT_ALL ? dts_in : nullptr;
    const DevModel<T> *mp = mp_in;     // device-resident model, fields fetched with scalar loads
    const int lane0 = threadIdx.x;
    int lane = lane0;
    // Work queue (multi-step launches of more worlds than the chip holds wavefronts): a world's episode is cut into
    // chunks of `queue_chunk` steps and the resident wavefronts draw (chunk, world) items from one atomic counter
    // instead of owning one world each.  With one workgroup per world a launch lasts as long as its unluckiest
    // slot -- the sum of two or three whole episodes whose lengths differ by tens of per cent (the sweeps) -- and
    // 17 % of the wave slots sat idle at 4096 worlds; drawn chunk by chunk the slots stay full until the last
    // chunk.  queue[0] = next item, queue[1 + w] = chunks of world w that are finished (the state travels through
    // global memory between wavefronts on different XCDs: coherent accesses, see `ldg` / `stg`).  Items are numbered
    // chunk-major, so the chunk an item waits for was drawn nworlds items earlier: it is finished, or it is running
    // on a resident wavefront that waits for nothing drawn later -- no circular wait.  The spin is capped all the same
    // (a producer stalled by a debugger or by serialised workgroups must not hang the device): a wavefront whose wait
    // expires raises the handle's host-visible status word, poisons the world's flag -- for good: flags only grow -- so
    // that its later chunks neither wait nor run, and goes on to the next item WITHOUT touching the world; the host
    // reports ARB_ERR_STALLED on every call until arb_model_status has been read.
    // (the float64 64-row kernels, compiled for one wave per SIMD with part of the register tile in AGPRs, faulted on
    // their first launch -- queue or not -- with the item loop around the body, ROCm 7.2: there every workgroup draws
    // ONE item and the grid is the number of items; the hardware dispatcher does the looping)
    // (reproducer: tools/f64_64_item_loop_repro.sh builds with -DARB_QUEUE_LOOP_ALL=1, which puts the loop back)
#ifndef ARB_QUEUE_LOOP
#define ARB_QUEUE_LOOP 1
#endif
#ifndef ARB_QUEUE_LOOP_ALL
#define ARB_QUEUE_LOOP_ALL 0
#endif
    constexpr bool QUEUE_LOOP = ARB_QUEUE_LOOP && (ARB_QUEUE_LOOP_ALL || !(WIDE_REGS && NMAX == 64));
    int *const queue = (MODE == 0) ? queue_in : nullptr;
    T *gq = gq_in, *gdq = gdq_in, *gcforce = gcforce_in;
    long w = blockIdx.x;
    int step_lo = 0, step_hi = nsteps, qitem_chunk = 0;
    for (;;) {     // one pass per work item; a single pass without the queue
    if (queue != nullptr) {
        int item = 0;
        if (lane0 == 0) item = __hip_atomic_fetch_add(queue, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        item = __builtin_amdgcn_readfirstlane(item);
        // chunks of queue_chunk steps, then the last queue_tail steps one by one: what is left of the idle time is the
        // length of the last items
        const int nhead = nsteps - queue_tail, nbig = (nhead + queue_chunk - 1) / queue_chunk;
        const int nchunks = nbig + queue_tail;
        if ((long)item >= nworlds * (long)nchunks) return;
        w = item % (int)nworlds;
        qitem_chunk = item / (int)nworlds;
        if (qitem_chunk < nbig) {
            step_lo = qitem_chunk * queue_chunk;
            step_hi = step_lo + queue_chunk < nhead ? step_lo + queue_chunk : nhead;
        } else {
            step_lo = nhead + (qitem_chunk - nbig);
            step_hi = step_lo + 1;
        }
        if (qitem_chunk > 0) {
            int spins = 0;      // (the cap, ~7 s of polling by default, guarantees that every wavefront leaves the kernel)
            int flag = 0;
            bool ready = false;
            while (queue_spin_cap >= 0) {       // (a negative cap is the tests' fault injection: every wait "expires")
                flag = __hip_atomic_load(queue + 1 + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ready = flag >= qitem_chunk;
                if (ready || spins >= queue_spin_cap) break;
                __builtin_amdgcn_s_sleep(16);
                ++spins;
            }
            // A wait that expires POISONS the world's flag, and the poison sticks (both writers of the flag use an atomic
            // max): no later chunk of the world waits again, none of them touches the world -- whose late producer may
            // still be writing its state --, and the host reports ARB_ERR_STALLED until the caller acknowledges it.
            constexpr int POISON = 0x7fffffff;
            if (!ready || flag == POISON) {
                if (!ready && lane0 == 0) {
                    __hip_atomic_store(mp->status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    (void)__hip_atomic_fetch_max(queue + 1 + w, POISON, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (!QUEUE_LOOP) return;
                continue;
            }
            asm volatile("" ::: "memory");      // (order only: the coherent loads of the state are issued after the flag was seen)
        }
        // (the state pointers are `restrict` kernel arguments: hand the compiler pointers it knows nothing about, so
        // that no load of the state is scheduled above the acquire)
        asm volatile("" : "+s"(gq), "+s"(gdq), "+s"(gcforce));
    } else if (w >= nworlds) {
        return;
    }
    T *lds = reinterpret_cast<T *>(arb_lds_raw);
    T *qs, *dqs, *qd, *BD, *SC, *CD, *RT, *AM, *VV, *FF, *FF0, *WORK;
    double *PD;
    int *CI;
// (after the first global store the compiler no longer proves the model unclobbered and fetches it with vector
// loads: readfirstlane puts the wave-uniform values back into SGPRs)
#define ARB_UNI(x) __builtin_amdgcn_readfirstlane(x)
#define ARB_LAY() ((CM == 2) ? (BODYCOL ? mp->layb3 : mp->lay3) : (BODYCOL ? mp->layb : mp->lay))
#define ARB_LDS_POINTERS() do { const Layout &lay_ = ARB_LAY();                                                              \
        qs = lds + ARB_UNI(lay_.q); dqs = lds + ARB_UNI(lay_.dq); qd = lds + ARB_UNI(lay_.qd); BD = lds + ARB_UNI(lay_.bd); SC = lds + ARB_UNI(lay_.sc);               \
        PD = reinterpret_cast<double *>(lds + ARB_UNI(lay_.pd)); CD = lds + ARB_UNI(lay_.cd); RT = lds + ARB_UNI(lay_.rt); AM = lds + ARB_UNI(lay_.am);       \
        VV = lds + ARB_UNI(lay_.vv); FF = lds + ARB_UNI(lay_.ff); FF0 = lds + ARB_UNI(lay_.ff0); WORK = lds + ARB_UNI(lay_.work);                             \
        CI = reinterpret_cast<int *>(lds + ARB_UNI(lay_.ci)); } while (0)
    ARB_LDS_POINTERS();
    // (the sizes are re-laundered at every phase boundary, ARB_OPAQUE_LANE: left to itself the compiler hoists
    // the ~90 wave-uniform predicates `i < n` of the unrolled row loops out of the step loop as 64-bit lane masks
    // and then spills them -- 284 SGPR spills in round 1)
    int n = mp->n, nb = mp->nb, nc = NC_CONST ? SPEC_NC : mp->nc, ndol = NC_CONST ? SPEC_NC * ARB_MAXDOL : mp->ndol;
    const int nq = mp->nq;
    // the host picks the smallest register tile that holds ndof (kNmaxChoices): rows below the previous tile
    // size always exist, which folds their `i < n` predicates away
    constexpr int NLOW = NMAX == 16 ? 0 : NMAX == 32 ? 16 : NMAX == 44 ? 32 : NMAX == 48 ? 44 : 48;
    constexpr bool LSCAN_OK = NMAX <= 48;      // (the 64-row tiles are register-bound: only the DPP scan is compiled in)
    constexpr int RS = NMAX;          // row stride of the per-dof LDS arrays (columns >= ndof stay zero)
    constexpr int BDS = (MODE == 1) ? BD_STRIDE_INSPECT : BD_STRIDE;      // per-body block (the gravity wrench slot: inspect only)
    T dt = dt_in, inv_dt = T(1) / dt_in;
    // row stride of Y' in LDS: four elements of padding (bank conflicts of the sweeps' column reads, see gs_stage)
#define lda (ndol + 4)
    // (evaluated where it is used, from the laundered nc: as one hoisted flag it lives in spilled lane masks)
#define do_constraints ((nc > 0) && !(flags & ARB_STEP_SKIP_CONSTRAINTS))

    // ---- load state (coalesced, world-major) -----------------------------
    // (queue mode: the state of a world passes from one wavefront to another, possibly on another XCD with its own
    // L2: its loads and stores are agent-scope relaxed atomics -- coherent by themselves, sc1 -- ordered against the
    // flag by s_waitcnt alone, instead of writing back and invalidating the whole L2 around every item)
    auto ldg = [&](const T *p) -> T {
        return queue != nullptr ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
    };
    auto stg = [&](T *p, T v) {
        if (queue != nullptr) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *p = v;
    };
    for (int i = lane; i < nq; i += WAVE) qs[i] = ldg(gq + w * nq + i);
    if (lane < RS) dqs[lane] = (lane < n) ? ldg(gdq + w * n + lane) : T(0);      // (the velocity array has one element per tile row)
    for (int i = lane; i < ndol; i += WAVE) {
        T f = T(0);
        if (gcforce != nullptr) f = ldg(gcforce + w * ndol + i);
        FF[i] = f;
    }
    // Forest worlds (several copies of a small model in this wavefront, arb_model::forest): bit j = copy j is RETIRED.
    // The copies share the elimination and the sweeps, where a product "exact zero x NaN" would carry one copy's NaN
    // into all the others; so a copy whose state is not finite -- or beyond +-1e8 (float32) / 1e100, i.e. diverged -- at
    // the beginning of a step computes on a state of rest from then on and has NaN written to its state, forces and
    // logs: what the one-world kernels leave behind for a world that overflowed, without touching its neighbours.
    unsigned dead = 0u;
    const T ext_k0 = (gext != nullptr && lane < n) ? gext[w * n + lane] : T(0);
    // running cost of the rollout (arb_step_cost): read here, one addition per step in step order, written back with the
    // state -- a horizon cut into work items or launches adds up bit for bit like one launch
    T cost_acc = T(0);
    if constexpr (FEAT_EXT) { if (cost.out != nullptr) cost_acc = ldg(cost.out + w); }
    bool warn_illcond = false;     // (float32: some pivot of this item's eliminations cancelled more digits than float32 can spare)
    if (!BODYCOL && lane < nc) {     // (body-space columns: the pairs' masks come from the model, the class has one tree)
        const int b1 = mp->cbody[lane], b0 = mp->cbody0[lane];
        const unsigned long long a1 = b1 >= 0 ? mp->anc[b1] : 0ull, a0 = b0 >= 0 ? mp->anc[b0] : 0ull;
        int *ci = CI + CI_STRIDE * lane;
        ci[0] = mp->ctype[lane];
        ci[1] = (int)(unsigned)a1; ci[2] = (int)(unsigned)(a1 >> 32);
        ci[3] = (int)(unsigned)a0; ci[4] = (int)(unsigned)(a0 >> 32);
        ci[5] = mp->cdof[lane];
        ci[6] = b1 >= 0 ? mp->root[b1] : (b0 >= 0 ? mp->root[b0] : 0);      // the tree whose origin the constraint's frame refers to
    }
    WAVE_SYNC();

#ifdef ARB_MARKS
#define ARB_STAMP(k) asm volatile("; ARB_MARK P" #k)
#define ARB_BSTAMP(k) asm volatile("; ARB_MARK B" #k)
#define ARB_CSTAMP(k) asm volatile("; ARB_MARK C" #k)
#define ARB_ASTAMP(k) asm volatile("; ARB_MARK A" #k)
#elif defined(ARB_GSSTAMPS)
#define ARB_STAMP(k) do { } while (0)
#define ARB_BSTAMP(k) do { } while (0)
#elif defined(ARB_ASTAMPS)   /* development: slots 1..6 = inside phase A (joint kinematics + H_pc, block algebra, own columns, level loop, body wrenches, end) */
#define ARB_STAMP(k) do { if (MODE == 1 && dbg.stamps != nullptr && lane0 == 0 && (k) == 0) dbg.stamps[w * 8 + (k)] = (long long)clock64(); } while (0)
#define ARB_BSTAMP(k) do { } while (0)
#define ARB_ASTAMP(k) do { if (MODE == 1 && dbg.stamps != nullptr && lane0 == 0) dbg.stamps[w * 8 + (k)] = (long long)clock64(); } while (0)
#elif defined(ARB_CSTAMPS)   /* development: slots 4..7 = inside phase C (columns loaded, pivot loop done, gvel added = start of D, end of D) */
#define ARB_STAMP(k) do { if (MODE == 1 && dbg.stamps != nullptr && lane0 == 0 && (k) <= 3) dbg.stamps[w * 8 + (k)] = (long long)clock64(); } while (0)
#define ARB_BSTAMP(k) do { } while (0)
#define ARB_CSTAMP(k) do { if (MODE == 1 && dbg.stamps != nullptr && lane0 == 0) dbg.stamps[w * 8 + (k)] = (long long)clock64(); } while (0)
#elif defined(ARB_BSTAMPS)   /* development: slots 3..7 = sub-phases of phase B (levels, dof products, rows of Z, constraint rows, end) */
#define ARB_STAMP(k) do { if (MODE == 1 && dbg.stamps != nullptr && lane0 == 0 && (k) <= 2) dbg.stamps[w * 8 + (k)] = (long long)clock64(); } while (0)
#define ARB_BSTAMP(k) do { if (MODE == 1 && dbg.stamps != nullptr && lane0 == 0) dbg.stamps[w * 8 + (k)] = (long long)clock64(); } while (0)
#else
#define ARB_STAMP(k) do { if (MODE == 1 && dbg.stamps != nullptr && lane0 == 0) dbg.stamps[w * 8 + (k)] = (long long)clock64(); } while (0)
#define ARB_BSTAMP(k) do { } while (0)
#endif
#ifndef ARB_CSTAMP
#define ARB_CSTAMP(k) do { } while (0)
#endif
#ifndef ARB_ASTAMP
#define ARB_ASTAMP(k) do { } while (0)
#endif
#define ARB_OPAQUE_LANE() do { lane = lane0; asm volatile("" : "+v"(lane)); mp = mp_in; asm volatile("" : "+s"(mp)); \
                              n = ARB_UNI(mp->n); nb = ARB_UNI(mp->nb); asm volatile("" : "+s"(n), "+s"(nb));                \
                              if (!NC_CONST) { nc = ARB_UNI(mp->nc); ndol = ARB_UNI(mp->ndol); asm volatile("" : "+s"(nc), "+s"(ndol)); } \
                              ARB_LDS_POINTERS(); } while (0)

    // World.integrate, core.py:974-980: gvel <- Y rhs + Y J'^T (f - f0) from the solution
    // columns in RT, then every joint integrates its position.
    auto integrate_on = [&](const T *RT, const T *FF, const T *FF0, T *qs, T *dqs, bool with_forces) {
        T vnew = T(0);
        if constexpr (BODYCOL) {
            // body-space columns: gvel+ = Y rhs + (Y J_p^T) g with the body-space force g = sum over the pair's contacts of
            // T_c^T (f_c - f0_c): lane j < 6 nbp forms g_j, then every dof lane takes its 6 nbp terms
            const int nb6 = 6 * ARB_UNI(mp->nbp);
            T *const GB = lds + ARB_UNI(ARB_LAY().vb);
            if (with_forces && lane < nb6) {
                const int p = lane / 6, j = lane - 6 * p;
                double g = 0.;
                for (int c = 0; c < nc; ++c) {
                    if (mp->cpair[c] != p) continue;
                    const T *tc = CD + c * CD_STRIDE + j;
#pragma unroll
                    for (int r = 0; r < 4; ++r) g += (double)tc[6 * r] * (double)(FF[4 * c + r] - FF0[4 * c + r]);
                }
                GB[lane] = (T)g;
            }
            WAVE_SYNC();
            if (lane < n) {
                vnew = RT[lane];
                if (with_forces)
                    for (int i = 0; i < nb6; ++i) vnew += RT[(1 + i) * RS + lane] * GB[i];
            }
        } else
        if (lane < n) {
            vnew = RT[lane];
            if (with_forces)
                for (int i = 0; i < ndol; ++i) vnew += RT[(1 + i) * RS + lane] * (FF[i] - FF0[i]);
        }
        WAVE_SYNC();
        if (lane < n) {
            dqs[lane] = vnew;
            const int qi = mp->dof2q[lane];
            if (qi >= 0) qs[qi] += dt * vnew;                               // core.py:238-240
        }
        WAVE_SYNC();
        // (lane-dense, see ARB_DENSE: a world has one FreeJoint or a few; every lane runs the exponential -- lanes of other
        // joints on the first FreeJoint's state --, the FreeJoints' lanes store)
        const bool isfree = lane < nb && mp->jtype[lane] == JT_FREE;
        const unsigned long long freemask = __ballot(isfree);
        if (freemask != 0ull && (ARB_DENSE_FJ || isfree)) {                      // joints.py:54-57
            const int fb = isfree ? lane : __builtin_ctzll(freemask);
            T *qp = qs + mp->q_off[fb];
            const T *vp = dqs + mp->dof_off[fb];
            M3<T> R, Re; V3<T> p, pe;
            R.a[0] = qp[0]; R.a[1] = qp[1]; R.a[2] = qp[2]; p.x = qp[3];
            R.a[3] = qp[4]; R.a[4] = qp[5]; R.a[5] = qp[6]; p.y = qp[7];
            R.a[6] = qp[8]; R.a[7] = qp[9]; R.a[8] = qp[10]; p.z = qp[11];
            exp_twist<T>(dt * v3<T>(vp[0], vp[1], vp[2]), dt * v3<T>(vp[3], vp[4], vp[5]), Re, pe);
            const M3<T> Rn = mul(R, Re);
            const V3<T> pn = mv(R, pe) + p;
            if (ARB_DENSE_FJ) { keep(Rn); keep(pn); }
            if (isfree) {
                qp[0] = Rn.a[0]; qp[1] = Rn.a[1]; qp[2] = Rn.a[2]; qp[3] = pn.x;
                qp[4] = Rn.a[3]; qp[5] = Rn.a[4]; qp[6] = Rn.a[5]; qp[7] = pn.y;
                qp[8] = Rn.a[6]; qp[9] = Rn.a[7]; qp[10] = Rn.a[8]; qp[11] = pn.z;
                qp[12] = T(0); qp[13] = T(0); qp[14] = T(0); qp[15] = T(1);
            }
        }
        WAVE_SYNC();
    };
    auto integrate_from_rt = [&](bool with_forces) { integrate_on(RT, FF, FF0, qs, dqs, with_forces); };

    if (MODE == 0 && (sio.mode & 1)) {
        // split execution: finish the previous step with the forces arb_gsw_kernel left in sio.f
        const int ncol_s = 1 + ndol;
        for (int i = lane; i < ncol_s * n; i += WAVE) RT[(i / n) * RS + (i % n)] = sio.sol[(long)w * ncol_s * n + i];
        for (int i = lane; i < ndol; i += WAVE) { FF[i] = sio.f[w * ndol + i]; FF0[i] = sio.f0[w * ndol + i]; }
        WAVE_SYNC();
        if (FEAT_ALL && dts != nullptr) { dt = (T)dts[-1]; inv_dt = T(1) / dt; }      // the step being finished
        integrate_from_rt(true);
    }
    if (MODE == 0 && sio.mode != 0 && !(sio.mode & 2)) step_hi = step_lo;      // apply only

    for (int step = step_lo; step < step_hi; ++step) {
        T gf0 = T(0);          // controllers' generalized force (inspect output)
        T ext_cost = T(0);     // this step's user torque of the lane's dof (the running cost's tau)
        T ext_k = ext_k0;
        if constexpr (FEAT_EXT) {
            // a torque SEQUENCE (arb_step_args.ext_gforce_steps): this step's row
            if (ext_stride != 0l && gext != nullptr)
                ext_k = (lane0 < ARB_UNI(mp->n)) ? gext[(long)step * ext_stride + w * ARB_UNI(mp->n) + lane0] : T(0);
            ext_cost = ext_k;
        }
#include "arb_phase_a.h"
#include "arb_phase_a2.h"
#include "arb_phase_b.h"
#include "arb_phase_c.h"
#include "arb_phase_d.h"
        if (MODE == 0 && (sio.mode & 2)) {
            // split execution: hand the constraint-space system to arb_gsw_kernel and stop here;
            // the next launch applies the forces (integrate_from_rt above)
            const int ncol_s = 1 + ndol;
            for (int i = lane; i < ncol_s * n; i += WAVE) sio.sol[(long)w * ncol_s * n + i] = RT[(i / n) * RS + (i % n)];
            for (int i = lane; i < ndol * ndol; i += WAVE) sio.A[(long)w * ndol * ndol + i] = AM[(i / ndol) * lda + i % ndol];
            for (int i = lane; i < ndol; i += WAVE) {
                sio.v[w * ndol + i] = VV[i]; sio.f[w * ndol + i] = FF[i]; sio.f0[w * ndol + i] = FF0[i];
            }
            if (lane < nc) {
                const T *cd = CD + lane * CD_STRIDE;
                T *o = sio.c + ((long)w * nc + lane) * 8;
                o[0] = cd[CD_ACTIVE]; o[1] = cd[CD_SDIST]; o[2] = cd[CD_POS0]; o[3] = cd[CD_POS0 + 1]; o[4] = cd[CD_POS0 + 2];
            }
            break;
        }

        if (do_constraints) {
            ARB_STAMP(5);
            ARB_CSTAMP(7);
            using GSG = std::conditional_t<(ARB_GS_F64 != 0) && std::is_same<T, float>::value, double, T>;
            gs_stage<T, MODE, GSG, !(WIDE_REGS && NMAX == 64), SPEC>(mp, lane, nc, ndol, lda, dt, inv_dt, AM, CD, VV, FF, WORK, dbg, w);
        }

        // ================= phase E: new velocity, integrate ==================
        ARB_OPAQUE_LANE();
        ARB_STAMP(6);
#if ARB_ANY_PRIO
        __builtin_amdgcn_s_setprio(ARB_E_PRIO);
#endif
        if (MODE == 1) {
            if (dbg.gforce != nullptr && lane < n) {
                // World._gforce after update_constraints: controllers + sum J_c^T f_c  (core.py:936-937);
                // J'^T was overwritten by the solution columns, so recompute from dbg.c_jac if present
                T g = gf0;
                if (do_constraints && dbg.c_jac != nullptr)
                    for (int i = 0; i < ndol; ++i) g += dbg.c_jac[(w * ndol + i) * n + lane] * FF[i];
                dbg.gforce[w * n + lane] = g;
            }
            if (dbg.c_force != nullptr)
                for (int i = lane; i < ndol; i += WAVE) dbg.c_force[w * ndol + i] = FF[i];
        }
        integrate_from_rt(do_constraints);
        if constexpr (FEAT_EXT) {
            // running cost (arb_step_cost): the state after this step, this step's torques; lane = dof, wave sum, one addition
            if (cost.out != nullptr) {
                T c = T(0);
                if (lane < n) {
                    const int qi = mp->dof2q[lane];
                    const T dd = (qi >= 0 ? qs[qi] : T(0)) - (cost.qref != nullptr ? cost.qref[lane] : T(0));
                    const T vv = dqs[lane];
                    const T cq = (cost.wq != nullptr ? cost.wq[lane] : T(0)) * dd * dd;
                    const T cv = (cost.wdq != nullptr ? cost.wdq[lane] : T(0)) * vv * vv;
                    const T cu = (cost.wtau != nullptr ? cost.wtau[lane] : T(0)) * ext_cost * ext_cost;
                    c = (cq + cv) + cu;
                }
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off);
                cost_acc += c;
            }
        }
    }

    // ---- store state -------------------------------------------------------
    ARB_OPAQUE_LANE();
    ARB_STAMP(7);
    if (dead != 0u) {               // retired copies of a forest world leave NaN behind
        const int fn = ARB_UNI(mp->fn), fnq = ARB_UNI(mp->fnq), fnd = ARB_MAXDOL * ARB_UNI(mp->fnc);
        for (int i = lane; i < nq; i += WAVE) if ((dead >> (i / fnq)) & 1u) qs[i] = (T)NAN;
        if (lane < n && ((dead >> (lane / fn)) & 1u)) dqs[lane] = (T)NAN;
        for (int i = lane; i < ndol; i += WAVE) if ((dead >> (i / fnd)) & 1u) FF[i] = (T)NAN;
        WAVE_SYNC();
    }
    if (MODE == 0) {
        for (int i = lane; i < nq; i += WAVE) stg(gq + w * nq + i, qs[i]);
        if (lane < n) stg(gdq + w * n + lane, dqs[lane]);
        if (gcforce != nullptr && !(sio.mode & 2))
            for (int i = lane; i < ndol; i += WAVE) stg(gcforce + w * ndol + i, FF[i]);
        if constexpr (FEAT_EXT) { if (cost.out != nullptr && lane0 == 0) stg(cost.out + w, cost_acc); }
        if (sizeof(T) == 4 && warn_illcond && lane0 == 0)
            (void)__hip_atomic_fetch_or(mp->warn, (int)ARB_WARN_ILLCOND, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    } else {
        if (dbg.q_next != nullptr) for (int i = lane; i < nq; i += WAVE) dbg.q_next[w * nq + i] = qs[i];
        if (dbg.dq_next != nullptr && lane < n) dbg.dq_next[w * n + lane] = dqs[lane];
    }
    if (queue == nullptr) break;
    // publish the chunk: every lane's stores of the state, then the flag (release, agent scope), then the next item.
    // (LDS is reused by the next item: all lanes are past their last LDS access -- one wavefront, program order)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // the coherent stores above are performed
    WAVE_SYNC();
    // (an atomic max: a flag poisoned by a consumer that gave up waiting for THIS chunk stays poisoned)
    if (lane0 == 0) (void)__hip_atomic_fetch_max(queue + 1 + w, qitem_chunk + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!QUEUE_LOOP) break;
    }     // work items
}
#undef do_constraints
#undef lda
#undef ARB_LDS_POINTERS
#undef ARB_UNI
#endif  // ARB_STEP_KERNEL_H
