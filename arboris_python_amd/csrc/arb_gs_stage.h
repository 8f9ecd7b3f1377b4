// arb_gs_stage.h -- the 20 Gauss-Seidel sweeps of World.update_constraints (core.py:929-935) for one world held by one
// wavefront: quad-local SoftFingerContact solve, the other constraint types, fast / complete variants (included by
// arb_kernels.hip only; the local solve itself is arb_math.h).
#ifndef ARB_GS_STAGE_H
#define ARB_GS_STAGE_H
// ===========================================================================
// The Gauss-Seidel stage of World.update_constraints (core.py:929-935) for ONE world held by ONE wavefront:
// shared by the fused step kernel and by the wave-per-world sweep kernel of the split execution.
// In LDS: AM = Y' (ndol x ndol), CD = per-constraint block (active, sdist, pos0 in; inverse block out),
// VV = v' (in/out), FF = constraint forces (in: warm start, out), WORK = 64 elements of scratch.
// ===========================================================================
// G = the arithmetic type of the sweeps (round 3 experiment, -DARB_GS_F64: float32 worlds whose sweeps -- velocities, forces,
// the two decision inequalities, the (pseudo-)inverse blocks -- run in float64 on the float32 system Y', v'; G = T otherwise)
// SPECK: every constraint is an enabled SoftFingerContact (the specialised step kernels, FEAT bit 4): its type is a constant
template <typename T, int MODE, typename G = T, bool ALLOW_FAST = true, bool SPECK = false>
// lda: row stride of Y' in LDS.  The step kernels pad the rows by four elements (round 5): with the dense stride 4 nc -- 16
// floats for four contacts -- the 16 row lanes' reads of their column block Y'[:, 4c..4c+3], four per local solve, fell on TWO of
// the 32 banks: eight-way conflicts, 28 extra LDS cycles per solve, 80 solves per step -- 370 M of the 438 M conflict cycles per
// launch that SQ_LDS_BANK_CONFLICT had counted since round 3 (28 % of the LDS-active cycles).  With rows of 4 nc + 4 floats
// read as ONE 16-byte vector the 16 / 32 rows lie in distinct 16-byte slots of the 256-byte bank row: conflict-free.
#ifndef ARB_GS_NOINLINE
#define ARB_GS_NOINLINE 0       // 1 (development): the stage as a function of its own -- measured round 6, same bits: classical columns +2.8 %, the headline (body-space columns) -1.1 %, the 64-row mixed build loses a wave per SIMD to the callee's registers: not adopted
#endif
#if ARB_GS_NOINLINE
__device__ __attribute__((noinline)) void gs_stage(const DevModel<T> *mp, const int lane, const int nc_, const int ndol_, const int lda_, const T dt_t,
                                         const T inv_dt_t, const T *AM_, T *CD_, T *VV_, T *FF_, T *WORK_,
                                         const DebugOut<T> &dbg, const long w) {
    // (arguments arrive in vector registers; LDS pointers as generic pointers: back to scalars and to LDS offsets)
    const int nc = __builtin_amdgcn_readfirstlane(nc_), ndol = __builtin_amdgcn_readfirstlane(ndol_), lda = __builtin_amdgcn_readfirstlane(lda_);
    T *const lds0_ = reinterpret_cast<T *>(arb_lds_raw);
    const T *AM = lds0_ + __builtin_amdgcn_readfirstlane((int)(AM_ - lds0_));
    T *CD = lds0_ + __builtin_amdgcn_readfirstlane((int)(CD_ - lds0_)), *VV = lds0_ + __builtin_amdgcn_readfirstlane((int)(VV_ - lds0_)),
      *FF = lds0_ + __builtin_amdgcn_readfirstlane((int)(FF_ - lds0_)), *WORK = lds0_ + __builtin_amdgcn_readfirstlane((int)(WORK_ - lds0_));
#else
__device__ __forceinline__ void gs_stage(const DevModel<T> *mp, const int lane, const int nc, const int ndol, const int lda, const T dt_t,
                                         const T inv_dt_t, const T *AM, T *CD, T *VV, T *FF, T *WORK,
                                         const DebugOut<T> &dbg, const long w) {
#endif
    constexpr bool SAME = std::is_same<T, G>::value;
    const G dt = (G)dt_t, inv_dt = SAME ? (G)inv_dt_t : G(1) / (G)dt_t;
    // (pseudo-)inverse of every active constraint's own admittance block (once per step): pinv(Y_cc) of
    // constraints.py:79, 83, 235, 795.  Pivoted elimination for the regular blocks, all constraints side by side;
    // the blocks it reports as rank deficient are redone one after the other with the SVD-based pinv_block.
    if constexpr (SAME) {
        bool deficient = false;
        {
            // (lane-dense, see ARB_DENSE: every lane inverts a block -- lanes without a constraint that of constraint 0 --,
            // the active constraints' lanes store)
            const bool mine = lane < nc && CD[(lane < nc ? lane : 0) * CD_STRIDE + CD_ACTIVE] != T(0);
            if (ARB_DENSE_INV ? (nc > 0) : mine) {
                const int c = lane < nc ? lane : 0, ct = SPECK ? (int)ARB_CT_SOFTFINGER : mp->ctype[c];
                const int nd = (ct == ARB_CT_SOFTFINGER) ? 4 : (ct == ARB_CT_BALLSOCKET ? 3 : 1);
                T P[16];
                const bool ok = inv_block<T>(AM + (4 * c) * lda + 4 * c, lda, nd, P);
                if (ARB_DENSE_INV) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) keep(P[i]);
                }
                if (mine) {
                    deficient = !ok;
#pragma unroll
                    for (int i = 0; i < 16; ++i) CD[c * CD_STRIDE + CD_PINV + i] = P[i];
                }
            }
        }
        unsigned long long todo = __ballot(deficient);
        while (todo != 0ull) {                           // wave-uniform, rare
            const int c = __builtin_ctzll(todo);
            todo &= todo - 1ull;
            if (lane == c) {
                const int ct = SPECK ? (int)ARB_CT_SOFTFINGER : mp->ctype[c];
                const int nd = (ct == ARB_CT_SOFTFINGER) ? 4 : (ct == ARB_CT_BALLSOCKET ? 3 : 1);
                T P[16];
                pinv_block<T>(AM + (4 * c) * lda + 4 * c, lda, nd, P);
                for (int i = 0; i < 16; ++i) CD[c * CD_STRIDE + CD_PINV + i] = P[i];
            }
        }
    }
    WAVE_SYNC();
    // ---- Gauss-Seidel, core.py:929-935, register resident ----------------------
    // lane = row of the stacked constraint system: it keeps its velocity, its force,
    // its row of the constraint's own admittance block Y_cc and of inv(Y_cc), and the
    // per-step constants of its constraint.  The four rows of a constraint are one QUAD
    // of lanes: the local solve of a SoftFingerContact runs inside that quad on DPP
    // quad_perm operands (vector registers only, branches follow the quad through
    // ballots); v_readlane broadcasts through SGPRs are left for what every row needs,
    // the four force increments.  Lane c also keeps the flags of constraint c.  The 20 x nc
    // sequential solves touch LDS only to read their column block of Y' (read-only).
    G vr = G(0), fr = G(0), Yrow[4], Prow[4];
    G k_sd = G(0), k_mu = G(0), k_e0 = G(1), k_e1 = G(1), k_e2 = G(1), k_p0 = G(0), k_p1 = G(0), k_p2 = G(0);
    bool k_eps1 = false;
    int k_ct = 0;
    bool k_act = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) { Yrow[i] = G(0); Prow[i] = G(0); }
    // the constants of a row's own constraint, replicated on the four lanes of its quad
    G q_sd = G(0), q_sdt = G(0), q_mu = G(0);
    G q_iyn = G(0), q_muyn = G(0), q_yc0 = G(0), q_yc1 = G(0), q_yc2 = G(0), q_bsq = G(0);
    SlidePre q_sp = {0., 0., 0., 0., 0., 0.};
    // ARB_POLY_LANES: lane r of a quad keeps the per-step constants of the sextic's coefficients r and r + 4 (arb_math.h:
    // SlideCoef) and evaluates those two in every sliding solve; the quad exchanges the six values by DPP -- 12 fused
    // multiply-adds and 12 DPP moves instead of the 36 of slide_poly (47 operations before round 4's expansion)
    SlideCoef q_ka = {0., 0., 0., 0., 0., 0., 0.}, q_kb = {0., 0., 0., 0., 0., 0., 0.};
    double q_nq = 0.;
    double q_warm = NAN;                    // root found for this constraint in the previous sweep
    double q_wmove = NAN;                   // how far that root had moved from the sweep before
    if (lane < ndol) {
        const int cc = lane >> 2, rr = lane & 3;
        vr = VV[lane]; fr = FF[lane];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            Yrow[i] = AM[lane * lda + 4 * cc + i];
            if constexpr (SAME) Prow[i] = CD[cc * CD_STRIDE + CD_PINV + 4 * rr + i];
        }
        if constexpr (!SAME) {
            // the (pseudo-)inverse of the row's own constraint block in the arithmetic of the sweeps, by every row lane
            // for itself (the four lanes of a quad do the same work side by side: no hand-over through LDS)
            if (CD[cc * CD_STRIDE + CD_ACTIVE] != T(0)) {
                const int ct = mp->ctype[cc];
                const int nd = (ct == ARB_CT_SOFTFINGER) ? 4 : (ct == ARB_CT_BALLSOCKET ? 3 : 1);
                G Yb[16], P[16];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) Yb[4 * i + j] = (G)AM[(4 * cc + i) * lda + 4 * cc + j];
                if (!inv_block<G, T>(Yb, 4, nd, P)) pinv_block<G, T>(Yb, 4, nd, P);
#pragma unroll
                for (int i = 0; i < 4; ++i) Prow[i] = (rr == 0) ? P[i] : (rr == 1) ? P[4 + i] : (rr == 2) ? P[8 + i] : P[12 + i];
            }
        }
        q_sd = CD[cc * CD_STRIDE + CD_SDIST]; q_sdt = q_sd / dt; q_mu = mp->cmu[cc];
        if (CD[cc * CD_STRIDE + CD_ACTIVE] != G(0) && (SPECK || mp->ctype[cc] == ARB_CT_SOFTFINGER)) {
            // admittance-only part of the sliding-branch polynomial and the other per-step constants of
            // SoftFingerContact.solve (constraints.py:795, 808-812), once per step
            G Yc4[16];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) Yc4[4 * i + j] = AM[(4 * cc + i) * lda + 4 * cc + j];
            q_sp = slide_precompute<G>(Yc4);
            if (ARB_POLY_LANES) {
                SlideCoef all[6];
                slide_coefs_all(q_sp, all);
                const auto pick = [&](double x0, double x1, double x2, double x3) { return rr == 0 ? x0 : rr == 1 ? x1 : rr == 2 ? x2 : x3; };
                q_ka.a0 = pick(all[0].a0, all[1].a0, all[2].a0, all[3].a0); q_ka.a1 = pick(all[0].a1, all[1].a1, all[2].a1, all[3].a1);
                q_ka.a2 = pick(all[0].a2, all[1].a2, all[2].a2, all[3].a2); q_ka.a3 = pick(all[0].a3, all[1].a3, all[2].a3, all[3].a3);
                q_ka.b0 = pick(all[0].b0, all[1].b0, all[2].b0, all[3].b0); q_ka.b1 = pick(all[0].b1, all[1].b1, all[2].b1, all[3].b1);
                q_ka.b2 = pick(all[0].b2, all[1].b2, all[2].b2, all[3].b2);
                q_kb.a0 = pick(all[4].a0, all[5].a0, 0., 0.); q_kb.a1 = pick(all[4].a1, all[5].a1, 0., 0.);
                q_kb.a2 = pick(all[4].a2, all[5].a2, 0., 0.); q_kb.a3 = pick(all[4].a3, all[5].a3, 0., 0.);
                q_kb.b0 = pick(all[4].b0, all[5].b0, 0., 0.); q_kb.b1 = pick(all[4].b1, all[5].b1, 0., 0.);
                q_kb.b2 = pick(all[4].b2, all[5].b2, 0., 0.);
                q_nq = q_sp.nq;
            }
            q_iyn = G(1) / Yc4[15]; q_muyn = q_mu / Yc4[15];
            q_yc0 = Yc4[3]; q_yc1 = Yc4[7]; q_yc2 = Yc4[11];
            const G bq0 = q_muyn * q_yc0, bq1 = q_muyn * q_yc1, bq2 = q_muyn * q_yc2;
            q_bsq = bq0 * bq0 + bq1 * bq1 + bq2 * bq2;
        }
    }
    if (lane < nc) {
        const T *cd = CD + lane * CD_STRIDE;
        k_act = cd[CD_ACTIVE] != G(0);
        k_sd = cd[CD_SDIST]; k_p0 = cd[CD_POS0]; k_p1 = cd[CD_POS0 + 1]; k_p2 = cd[CD_POS0 + 2];
        k_ct = SPECK ? (int)ARB_CT_SOFTFINGER : mp->ctype[lane]; k_mu = mp->cmu[lane];
        k_e0 = mp->ceps[3 * lane]; k_e1 = mp->ceps[3 * lane + 1]; k_e2 = mp->ceps[3 * lane + 2];
        k_eps1 = k_act && k_ct == ARB_CT_SOFTFINGER && (k_e0 == G(1)) && (k_e1 == G(1)) && (k_e2 == G(1));
    }
    unsigned long long actmask = __ballot(k_act);
    const unsigned long long eps1mask = __ballot(k_eps1);
    // forest worlds (several copies of a small model in this wavefront): a copy whose own rows a sweep left bit for bit
    // unchanged is at ITS fixed point and takes no further part -- one world per wavefront stops sweeping there, and a
    // float32 solve repeated beyond it is not exactly idempotent (the root finder's start depends on how far the root
    // moved in the previous sweep), which used to leave the copies a few ulps from the one-world launch (round 4)
    const int g_fk = SPECK ? 1 : mp->fk, g_fnc = mp->fnc;
    int st_rel = 0, st_sta = 0, st_fast = 0, st_slow = 0, st_sweeps = 0;
    int tr_rel = 0, tr_sta = 0, tr_slow = 0;
    G vr_prev = vr, fr_prev = fr;
#ifdef ARB_MARKS       /* development (tools/isa_phase_mix.py): comment markers in the compiler's assembly output at the segment boundaries */
#define ARB_GST(v) asm volatile("; ARB_MARK GS_" #v)
#elif defined(ARB_GSSTAMPS)   /* development: cycles of the segments of a sliding solve, summed over the step's sliding solves */
    long long gst[6] = {0, 0, 0, 0, 0, 0}, gt0 = 0, gt1 = 0, gt2 = 0, gt3 = 0, gt4 = 0;
    int gprobe[2] = {0, 0};
    bool gslid = false;
#define ARB_GST(v) do { if (MODE == 1) v = (long long)clock64(); } while (0)
#else
#define ARB_GST(v) do { } while (0)
#endif
#if ARB_GS_PRIO
    // the sweeps are one long dependent chain: let this wave issue ahead of the SIMD's other wave,
    // whose bulk phases have independent instructions to fill the gaps
    __builtin_amdgcn_s_setprio(ARB_GS_PRIO);
#endif
    // One local solve (constraint c of the current sweep).  FAST: the variant for worlds whose active constraints are all
    // SoftFingerContacts with eps = (1,1,1) -- no other constraint type, no division by eps, and NONE of the rare routes (the
    // 6x6 eigenvalue routine, row exchanges in the 4x4 solve): when a solve needs one, it returns false with the state as it
    // found it and the complete variant below redoes that solve and finishes the step.  Same expressions, same operations:
    // bit-identical results.  (Round 4: the rare routes' registers were paid for by every solve -- the eigenvalue routine
    // alone 95 spilled SGPRs; without them the launch is 3 % faster.)
    const T *const a4row = AM + (lane < ndol ? lane : 0) * lda;
    const unsigned long long rowmask = ndol >= 64 ? ~0ull : ((1ull << ndol) - 1ull);      // the lanes that hold a constraint row
    const auto solve_one = [&](auto fast_tag, const int sweep, const int c) -> bool {
        constexpr bool FAST = decltype(fast_tag)::value;
        (void)sweep;
        {
            const int base = 4 * c;
            ARB_GST(gt0);
            // column block Y'[:, 4c..4c+3] of this lane's row (issued early, used last)
            // (one 16-byte read: rows are 16-byte aligned, see lda.  EVERY lane reads -- the lanes beyond the constraint rows
            // row 0: a predicated read is an exec-mask region of eight instructions per solve; their velocities are never
            // stored and do not take part in the fixed-point test, see `rowmask`)
            G a4[4];
            {
                typedef T A4 __attribute__((ext_vector_type(4)));
                const A4 av = *reinterpret_cast<const A4 *>(a4row + base);
                a4[0] = av.x; a4[1] = av.y; a4[2] = av.z; a4[3] = av.w;
            }
            // (the fast variant runs only when every active constraint is a SoftFingerContact with eps = (1,1,1))
            const int ct = (FAST || SPECK) ? (int)ARB_CT_SOFTFINGER : __builtin_amdgcn_readlane(k_ct, c);
            G vc[4], fc[4], df[4], fnew[4];
            // A constraint's four rows are one quad of lanes: what its local solve needs from its own
            // rows comes as DPP quad_perm operands (every quad evaluates ITS constraint; only the quad of
            // c is used).  Values go through SGPRs (v_readlane) only where the whole wave needs them.
            const G fq0 = quad_bcast<0>(fr), fq1 = quad_bcast<1>(fr), fq2 = quad_bcast<2>(fr), fq3 = quad_bcast<3>(fr);
            // own-row products (meaningful on lanes base..base+3)
            // (measured round 5: the three four-term sums of a solve as two two-term chains joined by an addition -- two dependent
            // operations less each --: -0.5 %.  With three waves per SIMD the sweeps are bound by the NUMBER of instructions a
            // wave issues, not by the depth of its chain)
            const G v0r = vr - (Yrow[0] * fq0 + Yrow[1] * fq1 + Yrow[2] * fq2 + Yrow[3] * fq3);
            bool quad_done = false;      // softfinger release / static: per-lane results, see below
            G dfl = G(0), fnl = G(0);
            if (ct != ARB_CT_SOFTFINGER) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { vc[i] = bcast(vr, base + i); fc[i] = bcast(fr, base + i); }
            }
            if (ct == ARB_CT_SOFTFINGER) {                   // constraints.py:780-836
                // The release test and the static-friction candidate are evaluated side by side
                // (two independent dependent chains that overlap in the pipeline), inside the quad.
                const bool eps1 = FAST ? true : (bool)((eps1mask >> c) & 1ull);          // eps = (1,1,1): x/eps = x exactly
                const G vq0 = quad_bcast<0>(vr), vq1 = quad_bcast<1>(vr), vq2 = quad_bcast<2>(vr), vq3 = quad_bcast<3>(vr);
                const G dfr = -(Prow[0] * vq0 + Prow[1] * vq1 + Prow[2] * vq2 + Prow[3] * (vq3 + q_sdt));
                const G fnr = fr + dfr;
                const G v0n = quad_bcast<3>(v0r);
                const G fn0 = quad_bcast<0>(fnr), fn1 = quad_bcast<1>(fnr), fn2 = quad_bcast<2>(fnr), fn3 = quad_bcast<3>(fnr);
                G eps[3] = {G(1), G(1), G(1)};
                G lhs;
                if (eps1) {
                    lhs = fn0 * fn0 + fn1 * fn1 + fn2 * fn2;
                } else {
                    eps[0] = bcast(k_e0, c); eps[1] = bcast(k_e1, c); eps[2] = bcast(k_e2, c);
                    lhs = (fn0 / eps[0]) * (fn0 / eps[0]) + (fn1 / eps[1]) * (fn1 / eps[1])
                        + (fn2 / eps[2]) * (fn2 / eps[2]);
                }
                const G rhs = (fn3 * q_mu) * (fn3 * q_mu);
                // the quad of c decides for the wave
                const bool release = (__ballot(q_sd + dt * v0n > G(0)) >> base) & 1ull;
                const bool stat = (__ballot(lhs <= rhs) >> base) & 1ull;
                if (release || stat) {
                    // release (zero force) or static friction (df exactly -pinv(Y)(...) as in the
                    // reference, row by row): one branch, the two outcomes by selection
                    // (measured round 5: one ballot of the disjunction + lane-wise selection of the outcome: -0.3 %)
                    if (MODE == 1) { if (release) ++st_rel; else ++st_sta; }
                    dfl = release ? -fr : dfr; fnl = release ? G(0) : fnr; quad_done = true;
                } else {
                    {                                              // sliding friction
                        // Also inside the quad: the four lanes of constraint c carry the live problem in
                        // vector registers (the other quads run along on their own, unused data) and every
                        // branch follows the quad of c (`uni`), so nothing travels through SGPRs but the
                        // final force increments.
                        ARB_GST(gt1);
                        const int rq = lane - base;
                        const bool inquad = rq >= 0 && rq < 4;
                        const auto uni = [&](bool b) { return (bool)((__ballot(b) >> base) & 1ull); };
                        G alpha[4], shift = G(0);
                        alpha[0] = quad_bcast<0>(v0r); alpha[1] = quad_bcast<1>(v0r); alpha[2] = quad_bcast<2>(v0r);
                        alpha[3] = v0n + q_sdt;
                        // the constraint's own 4x4 admittance block (wave-uniform LDS reads)
                        G Y[16];
                        {
                            typedef T Y4 __attribute__((ext_vector_type(4)));      // (storage type)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const Y4 y4 = *reinterpret_cast<const Y4 *>(AM + (base + r) * lda + base);
                                Y[4 * r] = y4.x; Y[4 * r + 1] = y4.y; Y[4 * r + 2] = y4.z; Y[4 * r + 3] = y4.w;
                            }
                        }
                        if (MODE == 1) ++st_fast;
                        double warm = q_warm;
                        const double q_wmove_old = q_wmove;
                        bool have = false;
                        if (eps1) {
                            const G yc[3] = {q_yc0, q_yc1, q_yc2};
                            const G bq[3] = {q_muyn * yc[0], q_muyn * yc[1], q_muyn * yc[2]};
                            double c1, kappa, root;
                            slide_c1_kappa<G>(alpha, yc, q_iyn, q_muyn, bq, q_bsq, &c1, &kappa);
                            ARB_GST(gt2);
                            // The sweeps converge linearly: the root moves less and less from one sweep to the next.
                            // Float32 worlds restart the iteration twice the last move to the left of the previous
                            // root (instead of a fixed 1e-3 |root|): close enough that ONE Laguerre step lands within
                            // the tolerance, far enough that the no-root-to-the-left certificate still holds.  (Float64
                            // worlds keep the fixed offset: their bit-exact fixed point needs a start that depends on
                            // nothing but the previous root.)
                            double woff = -1.;
                            if (sizeof(T) == 4 && q_wmove == q_wmove)
                                woff = fmin(fmax(2. * q_wmove, 1e-9 * fabs(warm)), 0.1 * fabs(warm));
#ifdef ARB_GSSTAMPS
                            int *const probe = (MODE == 1 && lane == base) ? gprobe : nullptr;
                            if (slide_leftmost_root_uni(q_sp, c1, kappa, warm, &root, slide_step_tol<T>(), uni, probe, woff)) {
#elif ARB_ROOT_QM && ARB_POLY_LANES
                            const double pxa = slide_coef_eval(q_ka, -kappa, c1), pxb = slide_coef_eval(q_kb, -kappa, c1);
                            const double pcq[7] = {quad_bcast<0>(pxa), quad_bcast<1>(pxa), quad_bcast<2>(pxa), quad_bcast<3>(pxa),
                                                   quad_bcast<0>(pxb), quad_bcast<1>(pxb), 1.};
                            if (slide_leftmost_root_qm_pc(pcq, q_nq, c1, kappa, warm, &root, slide_step_tol<T>(), base, woff)) {
#elif ARB_ROOT_QM
                            if (slide_leftmost_root_qm(q_sp, c1, kappa, warm, &root, slide_step_tol<T>(), base, woff)) {
#else
                            if (slide_leftmost_root_uni(q_sp, c1, kappa, warm, &root, slide_step_tol<T>(), uni, nullptr, woff)) {
#endif
                                if (inquad) q_wmove = fabs(root - warm);      // (NaN after a cold start)
                                warm = root;
                                // leftmost real eigenvalue; admissible when <= 0 (constraints.py:826-830)
                                shift = (root <= 0.) ? (G)(root > -1e10 ? root : -1e10) : G(-1e10);
                                have = true;
                            }
#if ARB_ROOT_CASCADE && ARB_ROOT_QM && ARB_POLY_LANES && !defined(ARB_GSSTAMPS)
                            if constexpr (!FAST) {
                                if (!have) {
                                    // (rare) the iteration declined -- complex roots in its way --: the derivative cascade decides
                                    // in float64 (arb_math.h: slide_real_root_cascade), every lane on the sextic of c's quad
                                    double pb[7];
#pragma unroll
                                    for (int i = 0; i < 6; ++i) pb[i] = bcast(pcq[i], base);
                                    pb[6] = 1.;
                                    const double rbq = bcast(q_nq + 3. * fabs(c1) + arb_fast_sqrt(fabs(kappa)), base);
                                    int rc = -1;
                                    if (rbq > 0. && rbq < 1e300) rc = slide_real_root_cascade(pb, -1.0001 * rbq - 1e-300, &root, reinterpret_cast<double *>(WORK));
                                    if (rc >= 0) {
                                        if (MODE == 1) { ++st_slow; --st_fast; }
                                        shift = (rc == 1) ? (G)(root > -1e10 ? root : -1e10) : G(-1e10);
                                        have = true;
                                        warm = NAN;
                                        if (inquad) q_wmove = NAN;
                                    }
                                }
                            }
#endif
                        }
                        if constexpr (FAST) {
                            if (!have) return false;        // (rare: the complete variant takes over at this solve)
                        }
                        if (!have) {
                            if (MODE == 1) { ++st_slow; --st_fast; }
                            // rare: generic 6x6 eigenvalues (QR) of the matrix in the LDS work array, by the whole wavefront
                            if (inquad) softfinger_sliding_shift<G>(Y, alpha, q_mu, eps, WORK, &shift, false);
                            WAVE_SYNC();
                            if constexpr (ARB_EIG_WAVE_FOR(T)) {
                                shift = (G)slide_shift_from_eig_wave<T>(WORK, lane);
                            } else {
                                if (lane == 0) WORK[40] = slide_shift_from_eig<T>(WORK);
                                WAVE_SYNC();
                                shift = WORK[40];
                            }
                            WAVE_SYNC();
                            warm = NAN;
                            if (inquad) q_wmove = NAN;
                        }
                        const double q_warm_old = q_warm;
                        if (inquad) q_warm = warm;          // next sweep restarts next to this root
                        ARB_GST(gt3);
                        fnew[0] = fq0; fnew[1] = fq1; fnew[2] = fq2; fnew[3] = fq3;
                        G sie2[3] = {shift, shift, shift};
                        if (!eps1) {
#pragma unroll
                            for (int i = 0; i < 3; ++i) sie2[i] = shift / (eps[i] * eps[i]);
                        }
                        if constexpr (FAST) {
                            if (!softfinger_slide_finish_noex<G>(Y, alpha, sie2, fnew, df, uni)) {      // (rare: row exchanges)
                                q_warm = q_warm_old; q_wmove = q_wmove_old;                            // (the solve is redone)
                                return false;
                            }
                        } else {
                            softfinger_slide_finish_scaled<G>(Y, alpha, sie2, fnew, df, uni);
                        }
                        ARB_GST(gt4);
#ifdef ARB_GSSTAMPS
                        gslid = true;
#endif
                        dfl = (rq == 0) ? df[0] : (rq == 1) ? df[1] : (rq == 2) ? df[2] : df[3];
                        fnl = (rq == 0) ? fnew[0] : (rq == 1) ? fnew[1] : (rq == 2) ? fnew[2] : fnew[3];
                        quad_done = true;
                    }
                }
            } else if (ct == ARB_CT_BALLSOCKET) {                  // constraints.py:235-237
                const G p0 = bcast(k_p0, c), p1 = bcast(k_p1, c), p2 = bcast(k_p2, c);
                const G dfr = -(Prow[0] * (vc[0] + p0 * inv_dt) + Prow[1] * (vc[1] + p1 * inv_dt)
                                + Prow[2] * (vc[2] + p2 * inv_dt));
#pragma unroll
                for (int i = 0; i < 3; ++i) { df[i] = bcast(dfr, base + i); fnew[i] = fc[i] + df[i]; }
                df[3] = G(0); fnew[3] = fc[3];
            } else {                                               // JointLimits.solve constraints.py:73-90
                // pred = pos0 + dt v0 <= min  <=>  v0 <= (min - pos0)/dt =: glo, and (min - pred)/dt = glo - v0
                const G glo = bcast(k_p1, c), ghi = bcast(k_p2, c);
                const G p00 = bcast(Prow[0], base);
                const G v00 = bcast(v0r, base);
                G nf = G(0);
                if (v00 <= glo) nf = p00 * (glo - v00);
                else if (ghi <= v00) nf = p00 * (ghi - v00);
                df[0] = nf - fc[0]; fnew[0] = nf;
#pragma unroll
                for (int i = 1; i < 4; ++i) { df[i] = G(0); fnew[i] = fc[i]; }
            }
            if (MODE == 1 && dbg.gs_trace != nullptr && lane == 0) {
                int code = 4;
                if (ct == ARB_CT_SOFTFINGER) code = (st_rel != tr_rel) ? 0 : (st_sta != tr_sta) ? 1 : (st_slow != tr_slow) ? 3 : 2;
                dbg.gs_trace[((long)w * GS_SWEEPS + sweep) * nc + c] = code;
                tr_rel = st_rel; tr_sta = st_sta; tr_slow = st_slow;
            }
            const int rr = lane - base;
            if (quad_done) {
                // release / static: the quad holds the new forces and the force increments row by row
#pragma unroll
                for (int i = 0; i < 4; ++i) df[i] = bcast(dfl, base + i);
                fr = (rr >= 0 && rr < 4) ? fnl : fr;
            } else {
                fr = (rr == 0) ? fnew[0] : (rr == 1) ? fnew[1] : (rr == 2) ? fnew[2] : (rr == 3) ? fnew[3] : fr;
            }
            // vel += Y'[:, c] dforce                               core.py:935
            vr += a4[0] * df[0] + a4[1] * df[1] + a4[2] * df[2] + a4[3] * df[3];
#ifdef ARB_GSSTAMPS
            if (MODE == 1 && gslid) {
                asm volatile("" :: "v"(vr), "v"(fr));
                const long long gt5 = (long long)clock64();
                gst[0] += gt1 - gt0; gst[1] += gt2 - gt1; gst[2] += gt3 - gt2; gst[3] += gt4 - gt3; gst[4] += gt5 - gt4; gst[5] += 1;
                gslid = false;
            }
#endif
        }
        return true;
    };
    const auto end_of_sweep = [&]() -> bool {          // true: the sweeps are over
        // A sweep that leaves every velocity and force bit-for-bit unchanged is a fixed point
        // of the iteration: the remaining sweeps of core.py:929-935 would repeat it exactly.
        const unsigned long long sameb = __ballot(same_bits(vr, vr_prev) && same_bits(fr, fr_prev)) | ~rowmask;
        if (sameb == ~0ull && !(MODE == 1 && (dbg.ablate & 8))) return true;
        if (g_fk > 1) {
            const int rows = ARB_MAXDOL * g_fnc;              // constraint rows of one copy (g_fk * rows <= 64)
            const unsigned long long rm = (rows >= 64) ? ~0ull : ((1ull << rows) - 1ull), cm = (1ull << g_fnc) - 1ull;
            for (int j = 0; j < g_fk; ++j)
                if (((sameb >> (j * rows)) & rm) == rm) actmask &= ~(cm << (j * g_fnc));
            if (actmask == 0ull) return true;
        }
        vr_prev = vr; fr_prev = fr;
        return false;
    };
    int sweep = 0, c0 = 0;
    bool over = false;
    // (inspect kernels, forests and the float64-sweeps experiment take the complete variant throughout)
    // (ALLOW_FAST: not in the float64 64-row kernels, which are register-bound: a second copy of the solve is nine spilled VGPRs)
    if constexpr (ARB_GS_FAST && ALLOW_FAST && MODE == 0 && SAME) {
        bool fast = g_fk == 1 && (actmask & ~eps1mask) == 0ull;
        for (; sweep < GS_SWEEPS && fast; ++sweep) {
            for (int c = 0; c < nc; ++c) {
                if (!((actmask >> c) & 1ull)) continue;
                if (!solve_one(std::true_type{}, sweep, c)) { fast = false; c0 = c; break; }
            }
            if (!fast) break;
            if (end_of_sweep()) { over = true; break; }
        }
    }
    if (!over) {
        for (; sweep < GS_SWEEPS; ++sweep) {
            if (MODE == 1) ++st_sweeps;
            for (int c = c0; c < nc; ++c) {
                if (!((actmask >> c) & 1ull)) continue;
                (void)solve_one(std::false_type{}, sweep, c);
            }
            c0 = 0;
            if (end_of_sweep()) break;
        }
    }
#if ARB_GS_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
#ifdef ARB_GSSTAMPS
    if (MODE == 1 && dbg.stamps != nullptr && lane == 0)
        for (int i = 0; i < 6; ++i) dbg.stamps[w * 8 + i] = gst[i];
    if (MODE == 1 && dbg.stamps != nullptr) {      // (each quad's base lane counted its own constraint's solves)
        int p0 = gprobe[0], p1 = gprobe[1];
        for (int o = 4; o < 64; o <<= 1) { p0 += __shfl_xor(p0, o); p1 += __shfl_xor(p1, o); }
        if (lane == 0) { dbg.stamps[w * 8 + 6] = p0; dbg.stamps[w * 8 + 7] = p1; }
    }
#endif
    if (MODE == 1 && dbg.gs_stats != nullptr && lane == 0) {
        int *o = dbg.gs_stats + w * 5;
        o[0] = st_rel; o[1] = st_sta; o[2] = st_fast; o[3] = st_slow; o[4] = st_sweeps;
    }
    WAVE_SYNC();
    if (lane < ndol) { FF[lane] = fr; VV[lane] = vr; }
    WAVE_SYNC();
}
#endif  // ARB_GS_STAGE_H
