// arb_phase_d.h -- FRAGMENT of arb_step_kernel (arb_step_kernel.h), included inside its step loop: phase D -- the constraint-space system [v' | Y'] (matrix cores), block inverses; the sweeps follow in the kernel.
// Not a header of its own: it reads and writes the kernel's locals (LDS pointers, the register tile, the laundered sizes).
        // ================= phase D: constraint space + Gauss-Seidel ==========
        ARB_OPAQUE_LANE();
        ARB_STAMP(4);
        ARB_CSTAMP(6);
        if (do_constraints) {
            // [v | Y'] = J' [Y rhs | Y J'^T]                                core.py:925-927
            // (body-space columns: [v_b | Y_b] = J_p [Y rhs | Y J_p^T], 6 nbp rows in slabs of four; Y' and v' follow below)
            typedef T V4 __attribute__((ext_vector_type(4)));
            const int nb6 = BODYCOL ? 6 * ARB_UNI(mp->nbp) : 0;
            const int nslab = BODYCOL ? (nb6 + 3) / 4 : nc;
            T *const OV = BODYCOL ? lds + ARB_UNI(ARB_LAY().vb) : VV, *const OA = BODYCOL ? lds + ARB_UNI(ARB_LAY().yb) : AM;
            const int ost = BODYCOL ? nb6 : lda, orows = BODYCOL ? nb6 : ndol;
            bool anyact = false;
            if constexpr (BODYCOL) anyact = __ballot(lane < nc && CD[(lane < nc ? lane : 0) * CD_STRIDE + CD_ACTIVE] != T(0)) != 0ull;
#if ARB_PHASE_D_MFMA
            if constexpr (std::is_same<T, float>::value && !ELIM64) {
                // On the matrix cores (float32): the four rows of one constraint are the four accumulator registers of
                // a v_mfma_f32_4x4x1_16b_f32 slab, the 64 lanes its 64 columns (B operand = this lane's entry Z[r] of
                // the solution column), and the A operand of step r carries J'[4c + lane%4][r] -- read straight from
                // the rows of J' in LDS, four r per 16-byte read.  Unlike the elimination of phase C nothing here waits
                // for a lane exchange: all reads are independent of the accumulation, the ndof MFMAs of a slab issue
                // back to back (NMAX x 8 cycles per active constraint instead of 4 x (NMAX/2 v_pk_fma + NMAX/4 reads)).
                typedef float F4 __attribute__((ext_vector_type(4)));
                const int lq = lane & 3;
                for (int c = 0; c < nslab; ++c) {
                    F4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
                    // the rows of a constraint outside the active set are zero (phase B): nothing to multiply
                    if (BODYCOL ? anyact : (CD[c * CD_STRIDE + CD_ACTIVE] != T(0))) {
                        const F4 *jr4 = reinterpret_cast<const F4 *>(RT + (1 + 4 * c + lq) * RS);
                        // four partial sums (r mod 4), added pairwise at the end: four independent accumulator chains in
                        // the matrix pipe, and the rounding of a 44-term float32 dot product stays where the vector-ALU
                        // version's grouped sums had it (one sequential chain measured 1.08e-5 on the hardest golden
                        // states, against the 1e-5 gate)
                        F4 pa = acc, pb = acc, pc = acc, pd = acc, qa = acc, qb = acc, qc = acc, qd = acc;
#pragma unroll
                        for (int i4 = 0; i4 < NMAX / 4; ++i4) {
                            const F4 jv = jr4[i4];
                            pa = __builtin_amdgcn_mfma_f32_4x4x1f32(jv.x, Z[4 * i4], pa, 0, 0, 0);
                            pb = __builtin_amdgcn_mfma_f32_4x4x1f32(jv.y, Z[4 * i4 + 1], pb, 0, 0, 0);
                            pc = __builtin_amdgcn_mfma_f32_4x4x1f32(jv.z, Z[4 * i4 + 2], pc, 0, 0, 0);
                            pd = __builtin_amdgcn_mfma_f32_4x4x1f32(jv.w, Z[4 * i4 + 3], pd, 0, 0, 0);
                            if (NSETS == 2) {
                                qa = __builtin_amdgcn_mfma_f32_4x4x1f32(jv.x, Z2[4 * i4], qa, 0, 0, 0);
                                qb = __builtin_amdgcn_mfma_f32_4x4x1f32(jv.y, Z2[4 * i4 + 1], qb, 0, 0, 0);
                                qc = __builtin_amdgcn_mfma_f32_4x4x1f32(jv.z, Z2[4 * i4 + 2], qc, 0, 0, 0);
                                qd = __builtin_amdgcn_mfma_f32_4x4x1f32(jv.w, Z2[4 * i4 + 3], qd, 0, 0, 0);
                            }
                        }
                        // (forest worlds: the dofs of copy j start at j * fn, so the partial sum that holds "r mod 4 = 0" of the
                        // copy's own dofs is accumulator (j * fn) mod 4: added in the order of the copy alone, the sums are
                        // bit for bit those of one world per wavefront for any fn; round 4)
                        const int rot = (NMAX <= 32) ? (ARB_UNI(mp->fk) > 1 ? ((c / ARB_UNI(mp->fnc)) * ARB_UNI(mp->fn)) & 3 : 0) : 0;
                        if (rot == 0) acc = (pa + pb) + (pc + pd);
                        else if (rot == 1) acc = (pb + pc) + (pd + pa);
                        else if (rot == 2) acc = (pc + pd) + (pa + pb);
                        else acc = (pd + pa) + (pb + pc);
                        if (NSETS == 2) {
                            if (rot == 0) acc2 = (qa + qb) + (qc + qd);
                            else if (rot == 1) acc2 = (qb + qc) + (qd + qa);
                            else if (rot == 2) acc2 = (qc + qd) + (qa + qb);
                            else acc2 = (qd + qa) + (qb + qc);
                        }
                    }
                    const float out[4] = {acc.x, acc.y, acc.z, acc.w}, out2[4] = {acc2.x, acc2.y, acc2.z, acc2.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int idx = 4 * c + i;
                        if (BODYCOL && idx >= orows) continue;
                        if (lane == n) OV[idx] = out[i];
                        else if (lane > n && lane < ncols) OA[idx * ost + (lane - n - 1)] = out[i];
                        if (NSETS == 2) {
                            if (WAVE + lane == n) VV[idx] = out2[i];
                            else if ((WAVE + lane) < ncols) AM[idx * lda + (WAVE + lane - n - 1)] = out2[i];
                        }
                    }
                }
            } else
#endif
            for (int idx = 0; idx < orows; ++idx) {
                // row idx of J' (zero beyond ndof), read as 16/32-byte LDS vectors (wave-uniform address)
                const V4 *jr4 = reinterpret_cast<const V4 *>(RT + (1 + idx) * RS);
                ZT acc = ZT(0), acc2 = ZT(0);
                // the rows of a constraint outside the active set are zero (phase B): nothing to multiply
                // (free fall: the whole loop collapses to the stores)
                if (BODYCOL ? anyact : (CD[(idx >> 2) * CD_STRIDE + CD_ACTIVE] != T(0))) {
                    // (forest worlds: the groups of four start at the first dof of the constraint's copy, (j * fn) mod 4 = rot
                    // elements into the row, as they do for the copy alone; what lies before belongs to other copies and is
                    // exactly zero in this row.  Bit for bit the sums of one world per wavefront; round 4)
                    const int rot = (NMAX <= 32) ? (ARB_UNI(mp->fk) > 1 ? (((idx >> 2) / ARB_UNI(mp->fnc)) * ARB_UNI(mp->fn)) & 3 : 0) : 0;
                    if (rot == 0) {
#pragma unroll
                        for (int i4 = 0; i4 < NMAX / 4; ++i4) {
                            const V4 jv = jr4[i4];
                            acc += jv.x * Z[4 * i4] + jv.y * Z[4 * i4 + 1] + jv.z * Z[4 * i4 + 2] + jv.w * Z[4 * i4 + 3];
                            if (NSETS == 2)
                                acc2 += jv.x * Z2[4 * i4] + jv.y * Z2[4 * i4 + 1] + jv.z * Z2[4 * i4 + 2] + jv.w * Z2[4 * i4 + 3];
                        }
                    } else {
                        const T *jr = RT + (1 + idx) * RS;
                        static_for_asc(std::make_integer_sequence<int, 3>{}, [&](auto rc) {
                            constexpr int R0 = decltype(rc)::value + 1;
                            if (rot == R0) {
#pragma unroll
                                for (int i4 = 0; i4 < NMAX / 4; ++i4) {
                                    constexpr int NM = NMAX;
                                    const int r0 = 4 * i4 + R0;
                                    // (elements past the tile are zero: a copy's dofs end before it)
                                    const T jx = jr[r0], jy = (r0 + 1 < NM) ? jr[r0 + 1] : T(0), jz = (r0 + 2 < NM) ? jr[r0 + 2] : T(0),
                                            jw = (r0 + 3 < NM) ? jr[r0 + 3] : T(0);
                                    const int a = r0, b = (r0 + 1 < NM) ? r0 + 1 : r0, c2 = (r0 + 2 < NM) ? r0 + 2 : r0, d = (r0 + 3 < NM) ? r0 + 3 : r0;
                                    acc += jx * Z[a] + jy * Z[b] + jz * Z[c2] + jw * Z[d];
                                    if (NSETS == 2) acc2 += jx * Z2[a] + jy * Z2[b] + jz * Z2[c2] + jw * Z2[d];
                                }
                            }
                        });
                    }
                }
                if (lane == n) OV[idx] = (T)acc;
                else if (lane > n && lane < ncols) OA[idx * ost + (lane - n - 1)] = (T)acc;
                if (NSETS == 2) {
                    if (WAVE + lane == n) VV[idx] = (T)acc2;
                    else if ((WAVE + lane) < ncols) AM[idx * lda + (WAVE + lane - n - 1)] = (T)acc2;
                }
            }
            WAVE_SYNC();
            if constexpr (BODYCOL) {
                // ---- constraint space from body space: Y' = T Y_b T^T, v' = T v_b with T = blockdiag-by-pair of the contacts'
                // 4 x 6 transforms (zero for contacts outside the active set).  Two passes through LDS:
                //   W[i][col] = sum_j Y_b[i][6 p(col) + j] T_c(col)[col % 4][j]           6 nbp x ndol entries, 6 terms each
                //   Y'[row][col] = sum_i T_c(row)[row % 4][i] W[6 p(row) + i][col]       ndol x ndol entries, 6 terms each
                // (human36 with eight contacts: 6 + 16 entries per lane instead of 20 more matrix-core slabs and a second
                // column set in phase C)
                T *const WS = lds + ARB_UNI(ARB_LAY().wst);
                for (int e = lane; e < nb6 * ndol; e += WAVE) {
                    const int i = e / ndol, col = e - i * ndol, c2 = col >> 2, p2 = mp->cpair[c2];
                    const T *tr = CD + c2 * CD_STRIDE + 6 * (col & 3), *yb = OA + i * nb6 + 6 * p2;
                    // (float64 sums: the products of two float32 numbers are exact there, each entry is rounded once)
                    double acc = 0.;
#pragma unroll
                    for (int j = 0; j < 6; ++j) acc += (double)yb[j] * (double)tr[j];
                    WS[e] = (T)acc;
                }
                if (lane < ndol) {
                    const int c2 = lane >> 2, p2 = mp->cpair[c2];
                    const T *tr = CD + c2 * CD_STRIDE + 6 * (lane & 3);
                    double acc = 0.;
#pragma unroll
                    for (int j = 0; j < 6; ++j) acc += (double)tr[j] * (double)OV[6 * p2 + j];
                    VV[lane] = (T)acc;
                }
                WAVE_SYNC();
                const int nd4 = ndol >> 2;          // (ndol = 4 nc)
                for (int e = lane; e < ndol * nd4; e += WAVE) {
                    const int row = e / nd4, c4 = e - row * nd4, c2 = row >> 2, p2 = mp->cpair[c2];
                    const T *tr = CD + c2 * CD_STRIDE + 6 * (row & 3);
                    double a0 = 0., a1 = 0., a2 = 0., a3 = 0.;
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
                        const V4 w4 = *reinterpret_cast<const V4 *>(WS + (6 * p2 + i) * ndol + 4 * c4);
                        const double t = (double)tr[i];
                        a0 += t * (double)w4.x; a1 += t * (double)w4.y; a2 += t * (double)w4.z; a3 += t * (double)w4.w;
                    }
                    const V4 acc = {(T)a0, (T)a1, (T)a2, (T)a3};
                    *reinterpret_cast<V4 *>(AM + row * lda + 4 * c4) = acc;
                }
                WAVE_SYNC();
            }
        }
        // solution columns -> LDS (row r of RT := column r of [Y rhs | Y J'^T])
        WAVE_SYNC();
        {
            // (whole rows of RS elements, vector stores; rows >= ndof of a column register tile are zero -- zero on
            // entry, and an elimination step maps a zero row to 0 - 0 * t -- so "columns >= ndof stay zero" holds)
            typedef T V4 __attribute__((ext_vector_type(4)));
            if ((lane >= n && lane < ncols) || (late_rhs && lane == rhs_lane)) {
                V4 *dst = reinterpret_cast<V4 *>(RT + (late_rhs ? 0 : lane - n) * RS);
#pragma unroll
                for (int i4 = 0; i4 < NMAX / 4; ++i4) {
                    V4 v;
                    v.x = (T)Z[4 * i4]; v.y = (T)Z[4 * i4 + 1]; v.z = (T)Z[4 * i4 + 2]; v.w = (T)Z[4 * i4 + 3];
                    dst[i4] = v;
                }
            }
            if (NSETS == 2 && (WAVE + lane) < ncols) {
                V4 *dst = reinterpret_cast<V4 *>(RT + (WAVE + lane - n) * RS);
#pragma unroll
                for (int i4 = 0; i4 < NMAX / 4; ++i4) {
                    V4 v;
                    v.x = (T)Z2[4 * i4]; v.y = (T)Z2[4 * i4 + 1]; v.z = (T)Z2[4 * i4 + 2]; v.w = (T)Z2[4 * i4 + 3];
                    dst[i4] = v;
                }
            }
        }
        WAVE_SYNC();
        if (MODE == 1 && dbg.vel_free != nullptr && lane < n) dbg.vel_free[w * n + lane] = RT[lane];
        if (MODE == 1 && do_constraints) {
            if (dbg.c_adm != nullptr) for (int i = lane; i < ndol * ndol; i += WAVE) dbg.c_adm[(long)w * ndol * ndol + i] = AM[(i / ndol) * lda + i % ndol];
            if (dbg.c_vel != nullptr) for (int i = lane; i < ndol; i += WAVE) dbg.c_vel[(long)w * ndol + i] = VV[i];
        }
