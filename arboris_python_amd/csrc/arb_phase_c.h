// arb_phase_c.h -- FRAGMENT of arb_step_kernel (arb_step_kernel.h), included inside its step loop: phase C -- Gauss-Jordan elimination of the augmented system in registers.
// Not a header of its own: it reads and writes the kernel's locals (LDS pointers, the register tile, the laundered sizes).
        // ================= phase C: augmented Gauss-Jordan ===================
        ARB_OPAQUE_LANE();
        ARB_BSTAMP(7);
        ARB_STAMP(3);
        const int ncols = do_constraints ? (BODYCOL ? mp->ncols_b : mp->ncols) : n + 1;
        // (the mixed build: the float64 right-hand side travels to its column as a float32 pair hi + lo -- lo in the scratch
        // array, or in row 1 of RT for the late-rhs case, which has no constraint rows and needs the scratch array itself)
        T *const RLO = (NSETS == 1 && n == WAVE && NMAX == WAVE) ? RT + RS : WORK;
        if (lane < RS) {
            const T hi = (lane < n) ? (T)rhs : T(0);
            RT[lane] = hi;
            if constexpr (ELIM64) RLO[lane] = (lane < n) ? (T)((double)rhs - (double)hi) : T(0);
        }
        WAVE_SYNC();
        // Late rhs: 64 dofs, no constraints, one register set (the host's choice for that case): every lane holds a
        // column of Z, the rhs column waits in LDS (row 0 of RT) until the first pivot (dof n-1) has been taken;
        // lane n-1 -- whose own column is finished by that pivot -- applies the pivot to the rhs instead and carries
        // the rhs column from then on.  (A second register set of 64 float64 rows for ONE column is 128 VGPRs.)
        const bool late_rhs = NSETS == 1 && n == WAVE && NMAX == WAVE;
        const int rhs_lane = late_rhs ? n - 1 : n;
        {
            // column r of [rhs | J'^T] = row r of RT, fetched as 16/32-byte vectors (lanes without a column
            // read row 0 and discard it: unconditional loads, no per-element branches)
            typedef T V4 __attribute__((ext_vector_type(4)));
            if (lane >= n) {
                const bool have = lane < ncols;
                const V4 *src = reinterpret_cast<const V4 *>(RT + (have ? lane - n : 0) * RS);
#pragma unroll
                for (int i4 = 0; i4 < NMAX / 4; ++i4) {
                    const V4 v = src[i4];               // (entries >= ndof of a row of RT are zero: zeroed in A', never written)
                    Z[4 * i4] = have ? v.x : T(0);
                    Z[4 * i4 + 1] = have ? v.y : T(0);
                    Z[4 * i4 + 2] = have ? v.z : T(0);
                    Z[4 * i4 + 3] = have ? v.w : T(0);
                }
                if constexpr (ELIM64) {
                    if (lane == n) {                    // the rhs column: + lo
                        const V4 *lo4 = reinterpret_cast<const V4 *>(RLO);
#pragma unroll
                        for (int i4 = 0; i4 < NMAX / 4; ++i4) {
                            const V4 v = lo4[i4];
                            Z[4 * i4] += (ZT)v.x; Z[4 * i4 + 1] += (ZT)v.y; Z[4 * i4 + 2] += (ZT)v.z; Z[4 * i4 + 3] += (ZT)v.w;
                        }
                    }
                }
            }
            if (NSETS == 2) {
                const bool have = (WAVE + lane) < ncols;
                const V4 *src = reinterpret_cast<const V4 *>(RT + (have ? WAVE + lane - n : 0) * RS);
#pragma unroll
                for (int i4 = 0; i4 < NMAX / 4; ++i4) {
                    const V4 v = src[i4];
                    Z2[4 * i4] = have ? v.x : T(0);
                    Z2[4 * i4 + 1] = have ? v.y : T(0);
                    Z2[4 * i4 + 2] = have ? v.z : T(0);
                    Z2[4 * i4 + 3] = have ? v.w : T(0);
                }
            }
        }
#if ARB_ANY_PRIO
        __builtin_amdgcn_s_setprio(ARB_C_PRIO);
#endif
        // Pivots are taken from the last dof to the first (extremities before the
        // root): on these graded, nearly-SPD matrices that order halves the float32
        // error of pivot-free elimination (measured).
        ARB_CSTAMP(4);
        // Growth check (ABI 7, ARB_WARN_ILLCOND).  Pivot-free elimination leaves, for dof j, the pivot Z_jj - (what the dofs
        // eliminated before j take away); when that difference is 2^11 times smaller than Z_jj itself, eleven of float32's 24
        // bits are cancelled in that subtraction alone and the step's velocities cannot hold 1e-5 (a 64-link chain: 2^17; human36:
        // 2^6).  Both magnitudes are wave-uniform (v_readlane): the comparison runs on the SCALAR unit, as a difference of
        // the floats' bit patterns (2^23 log2 of the ratio to 6 %), one extra v_readlane per pivot.
        int growth_bits = -(1 << 30);
        auto track_growth = [&](auto pivv, int j) {
            if constexpr (TRACK_GROWTH) {
                // (Z_jj and a healthy pivot are positive floats: their bit patterns are positive integers and the difference cannot
                // overflow.  A pivot <= 0 -- sign bit set, -0.0 included: a NEGATIVE integer -- or a NaN pivot / diagonal means the
                // float32 elimination has gone indefinite: the growth saturates, the warning comes whatever Z_jj is.  Round 5
                // subtracted the raw patterns, which wraps for pb < 0: Z_jj = 512, pivot = -1e-3 gave -1988301423 and no warning.)
                const int zb = __builtin_amdgcn_readlane(__float_as_int(zdiag), j);
                const int pb = __builtin_amdgcn_readfirstlane(__float_as_int((float)pivv));
                const int gb = arb_growth_bits(zb, pb);
                growth_bits = (gb > growth_bits) ? gb : growth_bits;
            }
        };
        if constexpr (CM == 1 && std::is_same<T, float>::value) {
            // ---- matrix-core elimination (float32): one pivot = one rank-1 update of the whole register tile,
            // issued as NMAX/4 v_mfma_f32_4x4x1_16b_f32: the 16 4x4 blocks of one instruction are the 64 columns
            // (lane = column, B operand = this lane's entry of the scaled pivot row) times four rows (the four
            // accumulator registers of a slab), and the A operand carries the four multipliers of the slab's rows,
            // f[4g + lane % 4] -- the pivot column, which lane j hands over through LDS (NMAX/4 vector writes by one
            // lane, NMAX/4 reads by all).  Exact float32 FMAs (one per element: K = 1), so the reversed pivot order
            // and the error analysis of the VALU elimination carry over; what goes away are the NMAX v_readlane
            // broadcasts + wait states per pivot.  Fully unrolled: the pivot row index is static, no register rotation.
            // MEASURED (MI355X, human36, in-kernel stamps under load, profiles/r02_phaseC_mfma.txt): 46-49 k cycles for
            // the 42 pivots against 27-33 k of the VALU loop below (a variant rolled over slabs with a rotating register
            // tile: 60 k): every pivot waits for an LDS write -> read round trip on its critical path and the one-lane
            // column write costs 11 LDS issues.  Opt-in (ARB_STEP_MFMA_ELIM), parity-tested, not the default.
            typedef float F4 __attribute__((ext_vector_type(4)));
            float *COL = reinterpret_cast<float *>(WORK);           // the pivot column, NMAX <= 64 elements
            F4 *COL4 = reinterpret_cast<F4 *>(WORK);
            const int lq = lane & 3;
            // (pivot steps expanded at template level, NMAX-1 down to 0: with a `#pragma unroll` loop the index
            // only becomes constant late in the pipeline and the register tile ends up in scratch memory)
            static_for_desc(std::make_integer_sequence<int, NMAX>{}, [&](auto jc) {
                constexpr int j = decltype(jc)::value;
                if (j < n) {
                    const float piv = bcast(Z[j], j);
                    track_growth(piv, j);
                    const float nip = -arb_rcp(piv);
                    const float tn = Z[j] * nip;                  // minus this lane's entry of the scaled pivot row
                    float tn2 = 0.f;
                    if (NSETS == 2) tn2 = Z2[j] * nip;
                    if (lane == j) {
#pragma unroll
                        for (int g = 0; g < NMAX / 4; ++g) {
                            F4 v;
                            v.x = (4 * g == j) ? 0.f : Z[4 * g]; v.y = (4 * g + 1 == j) ? 0.f : Z[4 * g + 1];
                            v.z = (4 * g + 2 == j) ? 0.f : Z[4 * g + 2]; v.w = (4 * g + 3 == j) ? 0.f : Z[4 * g + 3];
                            COL4[g] = v;                          // (row j itself: multiplier 0, the row is replaced below)
                        }
                    }
                    WAVE_SYNC();
                    float a[NMAX / 4];
#pragma unroll
                    for (int g = 0; g < NMAX / 4; ++g) a[g] = COL[4 * g + lq];
                    WAVE_SYNC();
#pragma unroll
                    for (int g = 0; g < NMAX / 4; ++g) {
                        F4 acc;
                        acc.x = Z[4 * g]; acc.y = Z[4 * g + 1]; acc.z = Z[4 * g + 2]; acc.w = Z[4 * g + 3];
                        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[g], tn, acc, 0, 0, 0);
                        Z[4 * g] = acc.x; Z[4 * g + 1] = acc.y; Z[4 * g + 2] = acc.z; Z[4 * g + 3] = acc.w;
                        if (NSETS == 2) {
                            F4 ac2;
                            ac2.x = Z2[4 * g]; ac2.y = Z2[4 * g + 1]; ac2.z = Z2[4 * g + 2]; ac2.w = Z2[4 * g + 3];
                            ac2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[g], tn2, ac2, 0, 0, 0);
                            Z2[4 * g] = ac2.x; Z2[4 * g + 1] = ac2.y; Z2[4 * g + 2] = ac2.z; Z2[4 * g + 3] = ac2.w;
                        }
                    }
                    Z[j] = -tn;
                    if (NSETS == 2) Z2[j] = -tn2;
                }
            });
        } else if constexpr (ARB_ELIM_UNROLL && NMAX <= 48 && MODE == 0) {
            // ---- vector-ALU elimination expanded at template level (round 4): every pivot's code exists once, its row
            // indices are constants (no rotation of the register tile), and the groups of rows that dof j is NOT related to
            // -- structural zeros of Z: other branches of the tree, other copies of a forest, the padding rows -- are
            // skipped outright: for human36 (two legs, two arms, trunk and head) 60 % of the row updates.  Same operations on
            // the same values as the rolled loop below (a skipped update is `Z[r] - 0 t`): bit-identical.  Dense impedances
            // (PD controllers: Z_a couples any pair of dofs) switch the skipping off.  (Skipping inside the ROLLED loop was
            // measured too: the rotation of the register tile turns a skipped update into a move, the per-group branches
            // break the interleaving of the broadcasts: -4.5 %.)
            const bool z_dense = (!SPEC && mp->has_pd) || (FEAT_ALL && (pwd.kp != nullptr || pwd.zimp != nullptr));
            // (measured round 5: the groups of rows a pivot touches as a model constant -- one scalar load per pivot and a bit test
            // per group: -3 % (the load's latency is on the pivot's path); the whole pattern in twelve scalar registers: -1 %
            // (spilled scalar registers).  The two v_readlane per pivot of the lanes' own masks stay.)
            const unsigned long long relv = (lane < n && !z_dense) ? (mp->upmask[lane] | mp->descmask[lane]) : ~0ull;
            const unsigned rel_lo = (unsigned)relv, rel_hi = (unsigned)(relv >> 32);
            static_for_desc(std::make_integer_sequence<int, NMAX>{}, [&](auto jc) {
                constexpr int j = decltype(jc)::value;
                if (j < n) {
                    const ZT piv = bcast(Z[j], j);
                    track_growth(piv, j);
                    const ZT ip = arb_rcp(piv);
                    const ZT t = Z[j] * ip;
                    ZT t2 = ZT(0);
                    if (NSETS == 2) t2 = Z2[j] * ip;
                    const unsigned long long rel = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)rel_hi, j) << 32)
                                                 | (unsigned)__builtin_amdgcn_readlane((int)rel_lo, j);
                    constexpr int GB = NMAX >= 44 ? ARB_ELIM_GB_BIG : ARB_ELIM_GB;
#pragma unroll
                    for (int g = 0; g < (NMAX + GB - 1) / GB; ++g) {
                        if (((rel >> (GB * g)) & ((1ull << GB) - 1ull)) == 0ull) continue;
                        ZT f[GB];
#pragma unroll
                        for (int k = 0; k < GB; ++k) if (GB * g + k < NMAX && GB * g + k != j) f[k] = bcast(Z[GB * g + k], j);
                        asm volatile("" ::: "memory");
#pragma unroll
                        for (int k = 0; k < GB; ++k) if (GB * g + k < NMAX && GB * g + k != j) {
                            const int r = GB * g + k;
                            Z[r] = Z[r] - f[k] * t;
                            if (NSETS == 2) Z2[r] = Z2[r] - f[k] * t2;
                        }
                    }
                    Z[j] = t;
                    if (NSETS == 2) Z2[j] = t2;
                }
            });
        } else
        {
        // VALU elimination (float64, and float32 when the matrix-core path is compiled out): the register file
        // is rotated one row per step so that the pivot row always sits in
        // Z[NMAX-1] and every index below is a compile-time constant.
        for (int j = n; j < NMAX; ++j) {           // bring row n-1 into Z[NMAX-1]
            const ZT t = Z[NMAX - 1];
            ZT t2 = ZT(0);
            if (NSETS == 2) t2 = Z2[NMAX - 1];
#pragma unroll
            for (int r = NMAX - 1; r >= 1; --r) { Z[r] = Z[r - 1]; if (NSETS == 2) Z2[r] = Z2[r - 1]; }
            Z[0] = t;
            if (NSETS == 2) Z2[0] = t2;
        }
        // "Late rhs" worlds (64 dofs, no constraints: 65 columns on 64 lanes): lane 63 holds the column of dof 63 until the
        // first pivot has used it, then the rhs column, whose entries wait in LDS.  Its own first update,
        // Z[r] = rhs[r-1] - Z_old[r-1] t, runs AFTER the generic step of the first pivot from a copy of its old column in
        // WORK.  (Round 2 had `if (take_rhs) prev = RT[r-1]` inside the row update: the compiler kept that conditional LDS
        // read -- an exec-mask branch per row -- in EVERY pivot of the rolled loop, ~10 instructions per row instead of 3,
        // and snake-64 spent 57 % of its step in this loop.)
        if constexpr (NMAX == WAVE && NSETS == 1) {
            if (late_rhs && lane == n - 1) {
#pragma unroll
                for (int r = 0; r < NMAX; ++r) WORK[r] = Z[r];
            }
        }
        for (int j = n - 1; j >= 0; --j) {
            const ZT piv = bcast(Z[NMAX - 1], j);
            track_growth(piv, j);
            const ZT ip = arb_rcp(piv);
            const ZT t = Z[NMAX - 1] * ip;
            ZT t2 = ZT(0);
            if (NSETS == 2) t2 = Z2[NMAX - 1] * ip;
            // multipliers in groups of 8 broadcasts: the v_readlane -> SGPR -> v_fma wait states of one row are
            // filled by the broadcasts of the next rows instead of s_nop
#ifndef ARB_PIVOT_GB
#define ARB_PIVOT_GB 8
#endif
            constexpr int GB = ARB_PIVOT_GB;
#pragma unroll
            for (int r0 = NMAX - 1; r0 >= 1; r0 -= GB) {
                ZT f[GB];
#pragma unroll
                for (int k = 0; k < GB; ++k) if (r0 - k >= 1) f[k] = bcast(Z[r0 - k - 1], j);
                asm volatile("" ::: "memory");
#pragma unroll
                for (int k = 0; k < GB; ++k) if (r0 - k >= 1) {
                    const int r = r0 - k;
                    Z[r] = Z[r - 1] - f[k] * t;
                    if (NSETS == 2) Z2[r] = Z2[r - 1] - f[k] * t2;
                }
            }
            Z[0] = t;
            if (NSETS == 2) Z2[0] = t2;
            if constexpr (NMAX == WAVE && NSETS == 1) {
                if (late_rhs && j == n - 1) {                  // (wave-uniform: once per step)
                    asm volatile("");
                    if (lane == j) {
                        if constexpr (ELIM64) {          // (the mixed build: rhs = hi + lo, products in float64)
                            const double tr = ((double)RT[NMAX - 1] + (double)RLO[NMAX - 1]) * arb_rcp((double)WORK[NMAX - 1]);
#pragma unroll
                            for (int r = NMAX - 1; r >= 1; --r) Z[r] = ((double)RT[r - 1] + (double)RLO[r - 1]) - (double)WORK[r - 1] * tr;
                            Z[0] = tr;
                        } else {
                        const T tr = RT[NMAX - 1] * arb_rcp(WORK[NMAX - 1]);
#pragma unroll
                        for (int r = NMAX - 1; r >= 1; --r) Z[r] = RT[r - 1] - WORK[r - 1] * tr;
                        Z[0] = tr;
                        }
                    }
                }
            }
        }
        }
        ARB_CSTAMP(5);
#if ARB_ANY_PRIO
        __builtin_amdgcn_s_setprio(ARB_D_PRIO);
#endif
        if constexpr (TRACK_GROWTH) {
            if (sizeof(T) == 4 && MODE == 0) warn_illcond = warn_illcond || (growth_bits > (11 << 23));      // ARB_ILLCOND_GROWTH = 2^11
            if (MODE == 1 && dbg.pivot_growth != nullptr && lane == 0)
                dbg.pivot_growth[w] = growth_bits >= 0x40000000 ? (T)INFINITY      // (saturated: a pivot <= 0 or not finite)
                                                                : (T)__int_as_float((growth_bits > 0 ? growth_bits : 0) + 0x3f800000);
        }
        // the rhs column holds gvel+ - gvel: add gvel back so that it is Y (M gvel/dt + gforce)
        {
            // (dqs is zero beyond ndof; rows >= ndof of the columns are never used)
            typedef T V4 __attribute__((ext_vector_type(4)));
            const V4 *d4 = reinterpret_cast<const V4 *>(dqs);
            // (lane-dense, see ARB_DENSE: one lane holds the column; every lane adds, the column's lane keeps the sum)
            if (ARB_DENSE_GV) {
                const bool mine = lane == rhs_lane;
#pragma unroll
                for (int i4 = 0; i4 < NMAX / 4; ++i4) {
                    const V4 v = d4[i4];
                    const ZT s0 = Z[4 * i4] + v.x, s1 = Z[4 * i4 + 1] + v.y, s2 = Z[4 * i4 + 2] + v.z, s3 = Z[4 * i4 + 3] + v.w;
                    Z[4 * i4] = mine ? s0 : Z[4 * i4]; Z[4 * i4 + 1] = mine ? s1 : Z[4 * i4 + 1];
                    Z[4 * i4 + 2] = mine ? s2 : Z[4 * i4 + 2]; Z[4 * i4 + 3] = mine ? s3 : Z[4 * i4 + 3];
                }
                if (NSETS == 2) {
                    const bool mine2 = WAVE + lane == n;
#pragma unroll
                    for (int i4 = 0; i4 < NMAX / 4; ++i4) {
                        const V4 v = d4[i4];
                        const ZT s0 = Z2[4 * i4] + v.x, s1 = Z2[4 * i4 + 1] + v.y, s2 = Z2[4 * i4 + 2] + v.z, s3 = Z2[4 * i4 + 3] + v.w;
                        Z2[4 * i4] = mine2 ? s0 : Z2[4 * i4]; Z2[4 * i4 + 1] = mine2 ? s1 : Z2[4 * i4 + 1];
                        Z2[4 * i4 + 2] = mine2 ? s2 : Z2[4 * i4 + 2]; Z2[4 * i4 + 3] = mine2 ? s3 : Z2[4 * i4 + 3];
                    }
                }
            } else {
            if (lane == rhs_lane) {
#pragma unroll
                for (int i4 = 0; i4 < NMAX / 4; ++i4) {
                    const V4 v = d4[i4];
                    Z[4 * i4] += v.x; Z[4 * i4 + 1] += v.y; Z[4 * i4 + 2] += v.z; Z[4 * i4 + 3] += v.w;
                }
            }
            if (NSETS == 2 && WAVE + lane == n) {
#pragma unroll
                for (int i4 = 0; i4 < NMAX / 4; ++i4) {
                    const V4 v = d4[i4];
                    Z2[4 * i4] += v.x; Z2[4 * i4 + 1] += v.y; Z2[4 * i4 + 2] += v.z; Z2[4 * i4 + 3] += v.w;
                }
            }
            }
        }
        // lanes >= n (and the second set) now hold Y rhs and Y J'^T columns
