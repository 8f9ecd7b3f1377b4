// arb_phase_b.h -- FRAGMENT of arb_step_kernel (arb_step_kernel.h), included inside its step loop: phase B -- composite assembly of Z = M/dt + B + N in float64, right-hand side, constraint rows, controllers.
// Not a header of its own: it reads and writes the kernel's locals (LDS pointers, the register tile, the laundered sizes).
        // ================= phase B: lane = dof column =======================
        ARB_OPAQUE_LANE();
        ARB_STAMP(2);
#if ARB_ANY_PRIO
        __builtin_amdgcn_s_setprio(ARB_B_PRIO);
#endif
        // (ZT: the arithmetic type of the register tile -- T, or float64 for float32 worlds in the ARB_ELIM_F64 experiment)
        // (CM 3: the mixed build of round 6 -- the same mechanism as a production kernel for every tile)
        constexpr bool ELIM64 = CM == 3 || ((ARB_ELIM_F64 != 0) && std::is_same<T, float>::value && NMAX <= 48 && CM != 1);
        using ZT = std::conditional_t<ELIM64, double, T>;
        ZT Z[NMAX];
        ZT Z2[NSETS == 2 ? NMAX : 1];
#pragma unroll
        for (int i = 0; i < NMAX; ++i) Z[i] = ZT(0);
        T rhsM = T(0), rhsG = T(0);
        double rhsM_d = 0.;       // (the mixed build carries the right-hand side in float64 up to the register tile)
        // |Z_kk| as assembled (float32 worlds; inspect): the elimination of phase C compares every pivot with it, see there
        // (the mixed build eliminates in float64: nothing to warn about)
        constexpr bool TRACK_GROWTH = (sizeof(T) == 4 && MODE == 0 && CM != 3) || MODE == 1;
        float zdiag = 0.f;
        // ---- composite assembly ---------------------------------------------------------------------
        // With X_k = Ad(g<-body(k)) S_k the column of dof k in WORLD axes (about the root body's
        // origin; the same vector for every body below the joint), the reference's sums over bodies
        // (core.py:722-734) become sums over subtrees of per-body 6x6 matrices:
        //     Z[i][k] = X_i . (Ac_a X_k + Mc_a dX'_k),   a = the deeper of body(i), body(k)
        //     A_b  = Mg/dt - ad(T*_b)^T Mg + Mg ad(Om_b) + Bg,  Mg = Ad^T M_b Ad,  T*_b = [w; c x w]
        //     dX'_k = Ad(g<-b)(dS_k - ad(Om_b) S_k),  Ac_a = sum of A_b over the subtree of a (Mc_a likewise)
        // where Om_b is the accumulated pseudo twist of phase A (the reference's dAd_cp is ad(W_c) Ad_cp
        // with W_c != minus the relative twist for multi-dof joints, so Om_b != -V_b; tools/composite_proto.py
        // checks these identities against the oracle).  All of it in float64: the world-frame matrices of
        // distal bodies are small differences of large numbers.
        {
            constexpr int NACC = (MODE == 1) ? 69 : 63;
            double Acc[NACC];
            T om_b[6];
            double *STG = reinterpret_cast<double *>(BD);
            const bool useM = (MODE == 0) || zmode == 0 || zmode == 1;     // mass term of Z
            const bool useN = (MODE == 0) || zmode == 0 || zmode == 3;     // N (incl. the M dJ part)
            const bool useB = (MODE == 0) || zmode == 0 || zmode == 2;     // viscosity
            const double cM = (MODE == 1 && zmode == 1) ? 1. : (double)inv_dt;
#pragma unroll
            for (int i = 0; i < NACC; ++i) Acc[i] = 0.;
            // (twist, rhs wrench and gravity wrench of the body first: the log-depth sum below borrows the rhs slot)
            T twb[6], ptb[6], pgb[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) { twb[i] = T(0); ptb[i] = T(0); pgb[i] = T(0); }
            if (lane < nb) {
                const T *bd = BD + lane * BDS;
#pragma unroll
                for (int i = 0; i < 6; ++i) { twb[i] = bd[BD_TW + i]; ptb[i] = bd[BD_PT + i]; pgb[i] = (MODE == 1) ? bd[BD_PG + i] : T(0); }
            }
            WAVE_SYNC();
            // accumulated pseudo twist down the tree: Om_c = Ad_cp Om_p + W_c (phase A left W_c in BD_OM; done here,
            // one depth level per iteration, because phase A is the register-pressure peak of the kernel)
            {
                const int mydep = (lane < nb) ? mp->depth[lane] : -1;
                const int par = (lane < nb) ? mp->parent[lane] : -1;
                if (lane < nb) {
                    // W_c = Ad_cn Ad_nr T_rn = Ad_cp Ad_pr T_rn   (H_cn H_nr = H_cp H_pr)
                    T *bd = BD + lane * BDS;
                    const M3<T> R_pr = ld_m3(mp->Hpr + 12 * lane), R_cp = ld_m3(bd + BD_RCP);
                    const V3<T> p_pr = ld_v3(mp->Hpr + 12 * lane + 9), p_cp = ld_v3(bd + BD_PCP);
                    const V3<T> uw = mv(R_pr, ld_v3(bd + BD_OM));
                    const V3<T> uv = cross(p_pr, uw) + mv(R_pr, ld_v3(bd + BD_OM + 3));
                    const V3<T> ww = mv(R_cp, uw);
                    st_v3(bd + BD_OM, ww);
                    st_v3(bd + BD_OM + 3, cross(p_cp, ww) + mv(R_cp, uv));
                }
                WAVE_SYNC();
                // (float64 kernels, deep trees: the same sum in log2(depth) rounds, as phase A does for twists -- in world axes
                // Ad(H_gc) Om_c = Ad(H_gp) Om_p + Ad(H_gc) W_c is a prefix sum over the ancestors; the ancestor pointers
                // travel in the rhs-wrench slot, whose value every lane has taken into registers above)
                bool jumped = false;
                if constexpr (sizeof(T) == 8 || CM == 3) {
                    const int maxdep = ARB_UNI(mp->maxdepth);
                    if (!SPEC && maxdep >= ARB_JUMP_DEPTH) {
                        jumped = true;
                        int rounds = 0;
                        while ((1 << rounds) < maxdep + 1) ++rounds;
                        const bool onb = lane < nb;
                        T *const bdl = BD + (onb ? lane : 0) * BDS;
                        M3<double> Rgb = m3_identity<double>(); V3<double> pgb2 = v3<double>(0., 0., 0.);
                        if (onb) {
                            Rgb = ld_m3(PD + PDS * lane); pgb2 = ld_v3(PD + PDS * lane + 9);
                            const V3<double> ww = mv(Rgb, ld_v3_as<double>(bdl + BD_OM));
                            const V3<double> wv = cross(pgb2, ww) + mv(Rgb, ld_v3_as<double>(bdl + BD_OM + 3));
                            st_v3(bdl + BD_OM, cvt_v3<T>(ww)); st_v3(bdl + BD_OM + 3, cvt_v3<T>(wv));
                            bdl[BD_AB] = (T)par;
                        }
                        WAVE_SYNC();
                        for (int r = 0; r < rounds; ++r) {
                            const int a = onb ? (int)bdl[BD_AB] : -1;
                            T add6[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
                            T na = T(-1);
                            if (a >= 0) {
                                const T *ab = BD + a * BDS;
#pragma unroll
                                for (int i = 0; i < 6; ++i) add6[i] = ab[BD_OM + i];
                                na = ab[BD_AB];
                            }
                            WAVE_SYNC();
                            if (a >= 0) {
#pragma unroll
                                for (int i = 0; i < 6; ++i) bdl[BD_OM + i] += add6[i];
                                bdl[BD_AB] = na;
                            }
                            WAVE_SYNC();
                        }
                        if (onb) {
                            const V3<double> ww = ld_v3_as<double>(bdl + BD_OM), wv = ld_v3_as<double>(bdl + BD_OM + 3);
                            st_v3(bdl + BD_OM, cvt_v3<T>(mtv(Rgb, ww))); st_v3(bdl + BD_OM + 3, cvt_v3<T>(mtv(Rgb, wv - cross(pgb2, ww))));
                        }
                        WAVE_SYNC();
                    }
                }
                if (!jumped)
                for (int lvl = 1; lvl <= mp->maxdepth; ++lvl) {
                    const bool mine = mydep == lvl;          // (lane-dense, see ARB_DENSE)
                    if (ARB_DENSE_LVL || mine) {
                        T *bd = BD + (lane < nb ? lane : 0) * BDS;
                        const T *pb = BD + (par >= 0 ? par : 0) * BDS;
                        const M3<T> R_cp = ld_m3(bd + BD_RCP);
                        const V3<T> p_cp = ld_v3(bd + BD_PCP);
                        const V3<T> rw = mv(R_cp, ld_v3(pb + BD_OM));
                        const V3<T> nv = cross(p_cp, rw) + mv(R_cp, ld_v3(pb + BD_OM + 3)) + ld_v3(bd + BD_OM + 3);
                        const V3<T> nw = rw + ld_v3(bd + BD_OM);
                        if (ARB_DENSE_LVL) { keep(nv); keep(nw); }
                        if (mine) { st_v3(bd + BD_OM + 3, nv); st_v3(bd + BD_OM, nw); }
                    }
                    WAVE_SYNC();
                }
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) om_b[i] = T(0);
            if (lane < nb) {
                const T *bd = BD + lane * BDS;
#pragma unroll
                for (int i = 0; i < 6; ++i) om_b[i] = bd[BD_OM + i];
            }
            // Small trees: the body lanes write the M | rhs part of their accumulators straight into the prefix table,
            // which takes the place of the per-body blocks: every lane has its own block in registers by now.
            // (the three-wave kernels pass the accumulators through a half-size table in two passes -- less LDS, more
            // registers held across the first pass --, the two-wave kernels through a full table in one)
            // (measured on the 16- and 32-row tiles, whose register peak is the same phase A / B as the 44-row tile's: two passes
            // in their two-wave kernels cost 65 spilled VGPRs and 3-6 %, tools/forest_rate.py)
            constexpr bool TWO_PASS = (CM == 2 || CM == 3 || CM == 4 || MODE == 1);
            constexpr int TBS = TWO_PASS ? TB_STRIDE : TB_STRIDE1;
            const bool lscan = LSCAN_OK && mp->lay.lscan;
            const bool use_table = lscan && TWO_PASS;
            WAVE_SYNC();
            // ---- lane = body: world-frame matrices of the body -----------------------------------
            if (lane < nb) {
                const int b = lane;
                const M3<double> R = ld_m3(PD + PDS * b);
                const V3<double> p = ld_v3(PD + PDS * b + 9) - ld_v3(PD + PDS * mp->root[b] + 9);      // about its tree's root
                const T *Mb = mp->mass + 36 * b;
                auto blk = [](const T *m6, int r0, int c0) {
                    M3<double> o;
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int j = 0; j < 3; ++j) o.a[3 * i + j] = (double)m6[6 * (r0 + i) + c0 + j];
                    return o;
                };
                auto rot = [&](const M3<double> &Xm) { return mul(R, mulBT(Xm, R)); };       // R X R^T
                auto rowcross = [](const M3<double> &Xm, V3<double> v) {                      // X v^
                    M3<double> o;
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const V3<double> c = cross(v3<double>(Xm.a[3 * i], Xm.a[3 * i + 1], Xm.a[3 * i + 2]), v);
                        o.a[3 * i] = c.x; o.a[3 * i + 1] = c.y; o.a[3 * i + 2] = c.z;
                    }
                    return o;
                };
                // (before the 3x3 blocks, so that R and p die with them) wrenches to world axes: Ad(b<-g)^T f = (R tau + p x R f, R f)
                double wr[NACC - 57];              // world wrench of the increment rhs (6) [| gravity wrench (6), inspect]
                {
                    const V3<double> f = mv(R, v3<double>((double)ptb[3], (double)ptb[4], (double)ptb[5]));
                    const V3<double> tq = mv(R, v3<double>((double)ptb[0], (double)ptb[1], (double)ptb[2])) + cross(p, f);
                    wr[0] = tq.x; wr[1] = tq.y; wr[2] = tq.z; wr[3] = f.x; wr[4] = f.y; wr[5] = f.z;
                }
                if (MODE == 1) {
                    const V3<double> f = mv(R, v3<double>((double)pgb[3], (double)pgb[4], (double)pgb[5]));
                    const V3<double> tq = mv(R, v3<double>((double)pgb[0], (double)pgb[1], (double)pgb[2])) + cross(p, f);
                    wr[NACC - 63] = tq.x; wr[NACC - 62] = tq.y; wr[NACC - 61] = tq.z; wr[NACC - 60] = f.x; wr[NACC - 59] = f.y; wr[NACC - 58] = f.z;
                }
                double G[36];                  // Mg = Ad(b<-g)^T M_b Ad(b<-g), symmetric
                {
                    const M3<double> M11 = rot(blk(Mb, 0, 0)), M12 = rot(blk(Mb, 0, 3)), M22 = rot(blk(Mb, 3, 3));
                    const M3<double> G12 = add(M12, hatmul(p, M22));
                    const M3<double> G21 = transpose(G12);
                    const M3<double> G11 = add(sub(M11, rowcross(M12, p)), hatmul(p, G21));
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int j = 0; j < 3; ++j) {
                            G[6 * i + j] = G11.a[3 * i + j]; G[6 * i + 3 + j] = G12.a[3 * i + j];
                            G[6 * (3 + i) + j] = G21.a[3 * i + j]; G[6 * (3 + i) + 3 + j] = M22.a[3 * i + j];
                        }
                }
                // M (upper triangle, 21) | rhs wrench (6) [| gravity wrench (6)]: final as soon as Mg is -- into the prefix
                // table at once (small trees: these 27 values never occupy registers beside the 36 of A), or into Acc
                // (large trees -- the DPP scan -- keep them in Acc, assigned at the end of this block as before)
                auto mr_at = [&](int i) -> double {           // entry i of [M upper triangle | rhs wrench | gravity wrench]
                    constexpr int RW[21] = {0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 5};
                    constexpr int CL[21] = {0, 1, 2, 3, 4, 5, 1, 2, 3, 4, 5, 2, 3, 4, 5, 3, 4, 5, 4, 5, 5};
                    if (i < 21) return useN ? G[6 * RW[i] + CL[i]] : 0.;
                    return i < NACC - TB_PASS1 ? wr[i - 21] : 0.;
                };
                if (use_table) {
                    typedef double D2 __attribute__((ext_vector_type(2)));
                    D2 *row = reinterpret_cast<D2 *>(STG + TBS * b);
#pragma unroll
                    for (int i2 = 0; i2 < (NACC - TB_PASS1 + 1) / 2; ++i2) { D2 v; v.x = mr_at(2 * i2); v.y = mr_at(2 * i2 + 1); row[i2] = v; }
                }
                // T* = [w; c x w] (c = centre of mass, core.py:1276-1288) and Om, both in world axes
                const V3<double> wb = v3<double>((double)twb[0], (double)twb[1], (double)twb[2]);
                const double mm = (double)Mb[21];
                V3<double> cm = v3<double>(0., 0., 0.);
                if (!(mm <= 1e-10)) cm = (1. / mm) * v3<double>((double)Mb[6 * 2 + 4], (double)Mb[6 * 0 + 5], (double)Mb[6 * 1 + 3]);
                const V3<double> Tw = mv(R, wb);
                const V3<double> Tv = mv(R, cross(cm, wb)) + cross(p, Tw);
                const V3<double> ow = mv(R, v3<double>((double)om_b[0], (double)om_b[1], (double)om_b[2]));
                const V3<double> ov = mv(R, v3<double>((double)om_b[3], (double)om_b[4], (double)om_b[5])) + cross(p, ow);
#pragma unroll
                for (int i = 0; i < 36; ++i) Acc[i] += useM ? cM * G[i] : 0.;
                if (useN) {
#pragma unroll
                    for (int j = 0; j < 6; ++j) {                        // -ad(T*)^T Mg, column by column
                        const V3<double> gt = v3<double>(G[j], G[6 + j], G[12 + j]), gb = v3<double>(G[18 + j], G[24 + j], G[30 + j]);
                        const V3<double> t = cross(Tw, gt) + cross(Tv, gb), u = cross(Tw, gb);
                        Acc[j] += t.x; Acc[6 + j] += t.y; Acc[12 + j] += t.z;
                        Acc[18 + j] += u.x; Acc[24 + j] += u.y; Acc[30 + j] += u.z;
                    }
#pragma unroll
                    for (int r = 0; r < 6; ++r) {                        // Mg ad(Om), row by row
                        const V3<double> gl = v3<double>(G[6 * r], G[6 * r + 1], G[6 * r + 2]), gr = v3<double>(G[6 * r + 3], G[6 * r + 4], G[6 * r + 5]);
                        const V3<double> t = cross(gl, ow) + cross(gr, ov), u = cross(gr, ow);
                        Acc[6 * r] += t.x; Acc[6 * r + 1] += t.y; Acc[6 * r + 2] += t.z;
                        Acc[6 * r + 3] += u.x; Acc[6 * r + 4] += u.y; Acc[6 * r + 5] += u.z;
                    }
                }
                // Viscosity (rare): Bg = Ad^T B_b Ad, a general 6x6, added LAST, from the pose read again -- as the first term of the
                // sums (until round 4) it made every accumulator a value that is live from its zero on, through this never
                // taken branch, to its first real term: 3 % of the launch for every model without viscosity.
                if (!SPEC && mp->has_visc && useB) {
                    const M3<double> Rv = ld_m3(PD + PDS * b);
                    const V3<double> pv = ld_v3(PD + PDS * b + 9) - ld_v3(PD + PDS * mp->root[b] + 9);
                    auto rotv = [&](const M3<double> &Xm) { return mul(Rv, mulBT(Xm, Rv)); };
                    const T *Vb = mp->visc + 36 * b;
                    const M3<double> B11 = rotv(blk(Vb, 0, 0)), B12 = rotv(blk(Vb, 0, 3)), B21 = rotv(blk(Vb, 3, 0)), B22 = rotv(blk(Vb, 3, 3));
                    const M3<double> H12 = add(B12, hatmul(pv, B22));
                    const M3<double> H21 = sub(B21, rowcross(B22, pv));
                    const M3<double> H11 = add(sub(B11, rowcross(B12, pv)), hatmul(pv, H21));
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int j = 0; j < 3; ++j) {
                            Acc[6 * i + j] += H11.a[3 * i + j]; Acc[6 * i + 3 + j] += H12.a[3 * i + j];
                            Acc[6 * (3 + i) + j] += H21.a[3 * i + j]; Acc[6 * (3 + i) + 3 + j] += B22.a[3 * i + j];
                        }
                }
                if (!use_table) {
#pragma unroll
                    for (int i = 0; i < NACC - TB_PASS1; ++i) Acc[TB_PASS1 + i] = mr_at(i);
                }
            }
            WAVE_SYNC();                       // (table rows / accumulators of every body are complete)
            // One pass of the LDS prefix table over the accumulators OFF .. OFF + CNT - 1: (a) lane = body stores them
            // as a table row, (b) lane = accumulator runs the inclusive prefix down its column.
            // (the rows of the M | rhs pass were written by the body block above: `written`)
            auto tb_pass = [&](auto offc, auto cntc, bool written) {
                constexpr int OFF = decltype(offc)::value, CNT = decltype(cntc)::value;
                typedef double D2 __attribute__((ext_vector_type(2)));
                double *TB = STG;
                if (!written) {
                    if (lane < nb) {
                        D2 *row = reinterpret_cast<D2 *>(TB + TBS * lane);
#pragma unroll
                        for (int i2 = 0; i2 < (CNT + 1) / 2; ++i2) {
                            D2 v; v.x = Acc[OFF + 2 * i2]; v.y = (2 * i2 + 1 < CNT) ? Acc[OFF + 2 * i2 + 1] : 0.;
                            row[i2] = v;
                        }
                    }
                    WAVE_SYNC();
                }
                const unsigned long long roots = mp->rootmask;       // (the sums restart at the root of every tree)
                for (int i = lane; i < CNT; i += WAVE) {
                    double run = 0.;
                    double *col = TB + i;
                    for (int b0 = 0; b0 < nb; b0 += 4) {              // four bodies per round trip
                        const double v0 = col[TBS * b0];
                        const double v1 = (b0 + 1 < nb) ? col[TBS * (b0 + 1)] : 0.;
                        const double v2 = (b0 + 2 < nb) ? col[TBS * (b0 + 2)] : 0.;
                        const double v3 = (b0 + 3 < nb) ? col[TBS * (b0 + 3)] : 0.;
                        const unsigned r4 = (unsigned)(roots >> b0) & 15u;
                        run = (r4 & 1u) ? v0 : run + v0; col[TBS * b0] = run;
                        run = (r4 & 2u) ? v1 : run + v1; if (b0 + 1 < nb) col[TBS * (b0 + 1)] = run;
                        run = (r4 & 4u) ? v2 : run + v2; if (b0 + 2 < nb) col[TBS * (b0 + 2)] = run;
                        run = (r4 & 8u) ? v3 : run + v3; if (b0 + 3 < nb) col[TBS * (b0 + 3)] = run;
                    }
                }
                WAVE_SYNC();
            };
            // ---- subtree sums, deepest level first; children hand their sums over through STG ------
            ARB_BSTAMP(3);
            const int bsrc = (lane < n) ? mp->dofbody[lane] : 0;
            {
                // Bodies in DFS preorder: subtree(a) = lanes a .. a + subsize[a] - 1, so a subtree sum is a
                // difference of inclusive prefix sums over the lanes, P[a + subsize[a] - 1] - P[a - 1].  The scan
                // runs on DPP row shifts in the vector ALU (log2 steps, no LDS traffic); float64 keeps the
                // difference exact to ~1e-13 of the whole-tree sum.  Element by element, and straight on to the
                // dof lanes (lane k takes the composite of body(k)), so that only one element is in flight.
                const bool two_rows = nb > 16, four_rows = nb > 32;
                const int hi = (lane < nb) ? lane + mp->subsize[lane] - 1 : lane;
                if (lscan) {
                    // Small trees (the table fits the staging area): the same inclusive prefix sums, formed in LDS with
                    // the roles transposed -- lane = accumulator, a serial pass over the bodies: nb additions in all
                    // instead of 4-6 DPP steps + two lane exchanges per accumulator (~210 instead of ~1700 wave
                    // instructions for human36; round 2).  (a) lane = body stores its accumulators as a table row;
                    // (b) lane = accumulator i runs the prefix down its column; (c) lane = dof k reads the two rows that
                    // bound the subtree of body(k) and subtracts, element by element as its products consume them.
                    // first pass: M | rhs, whose rows the body lanes have written already (the pass over the 36 entries of
                    // A, still in registers, runs inside the consumer below once the first has been consumed: the table is
                    // half as large that way, and the body block never holds more than A and Mg in registers)
                    if constexpr (TWO_PASS) tb_pass(std::integral_constant<int, TB_PASS1>{}, std::integral_constant<int, NACC - TB_PASS1>{}, true);
                    else tb_pass(std::integral_constant<int, 0>{}, std::integral_constant<int, NACC>{}, false);
                    // (c) happens in the consumer below, which streams the two table rows of body(k) straight
                    // into its products: the 63 composites never sit in registers all at once
                } else {
#pragma unroll
                for (int i = 0; i < NACC; ++i) {
                    double x = Acc[i];
                    x += dpp_f64<0x111, 0xF>(x);            // row_shr:1
                    x += dpp_f64<0x112, 0xF>(x);            // row_shr:2
                    x += dpp_f64<0x114, 0xF>(x);            // row_shr:4
                    x += dpp_f64<0x118, 0xF>(x);            // row_shr:8
                    if (two_rows) x += dpp_f64<0x142, 0xA>(x);      // row_bcast:15 into rows 1 and 3
                    if (four_rows) x += dpp_f64<0x143, 0xC>(x);     // row_bcast:31 into rows 2 and 3
                    const double sub = __shfl(x, hi) - dpp_f64<0x138, 0xF>(x);      // wave_shr:1 (0.0 into lane 0)
                    Acc[i] = __shfl(sub, bsrc);
                    if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);    // four chains in flight (eight: slower, measured)
                }
                }
            }
            // ---- lane = dof k: fetch the composites of body(k), own column, the three products -----
            ARB_BSTAMP(4);
            double Xk[6], dXk[6], Gk[6];
            V3<double> p0k;                    // origin of the root body of dof k's tree
            {
                // (DPP scan: from here on Acc holds the composites of body(k), not of body(lane))
                T omk[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) omk[i] = __shfl(om_b[i], bsrc);
                const M3<double> R = ld_m3(PD + PDS * bsrc);
                p0k = ld_v3(PD + PDS * mp->root[bsrc] + 9);
                const V3<double> p = ld_v3(PD + PDS * bsrc + 9) - p0k;
                const int kc = lane < RS ? lane : 0;
                const V3<double> sw = v3<double>((double)SC[0 * RS + kc], (double)SC[1 * RS + kc], (double)SC[2 * RS + kc]);
                const V3<double> sv = v3<double>((double)SC[3 * RS + kc], (double)SC[4 * RS + kc], (double)SC[5 * RS + kc]);
                const V3<double> dsw = v3<double>((double)SC[6 * RS + kc], (double)SC[7 * RS + kc], (double)SC[8 * RS + kc]);
                const V3<double> dsv = v3<double>((double)SC[9 * RS + kc], (double)SC[10 * RS + kc], (double)SC[11 * RS + kc]);
                const V3<double> okw = v3<double>((double)omk[0], (double)omk[1], (double)omk[2]);
                const V3<double> okv = v3<double>((double)omk[3], (double)omk[4], (double)omk[5]);
                const V3<double> xw = mv(R, sw);
                const V3<double> xv = mv(R, sv) + cross(p, xw);
                const V3<double> aw2 = dsw - cross(okw, sw);                          // dS - ad(Om) S
                const V3<double> av2 = dsv - cross(okv, sw) - cross(okw, sv);
                const V3<double> dw = mv(R, aw2);
                const V3<double> dv = mv(R, av2) + cross(p, dw);
                Xk[0] = xw.x; Xk[1] = xw.y; Xk[2] = xw.z; Xk[3] = xv.x; Xk[4] = xv.y; Xk[5] = xv.z;
                dXk[0] = dw.x; dXk[1] = dw.y; dXk[2] = dw.z; dXk[3] = dv.x; dXk[4] = dv.y; dXk[5] = dv.z;
                if (lane >= n) {
#pragma unroll
                    for (int i = 0; i < 6; ++i) { Xk[i] = 0.; dXk[i] = 0.; }
                }
                // One pass over the composites of body(k) -- A (36, row-major) | M (upper triangle, 21) | rhs wrench (6)
                // [| gravity wrench (6), inspect] -- accumulating G = A X + M dX', P = A^T X, R = M X and the rhs
                // entries as each value arrives: from the registers (DPP scan) or from the prefix table in LDS.
                double Pk[6], Rk[6], Mdk[6];
#pragma unroll
                for (int r = 0; r < 6; ++r) { Gk[r] = 0.; Pk[r] = 0.; Rk[r] = 0.; Mdk[r] = 0.; }
                double rm = 0., rg = 0.;
                auto visit = [&](auto ic, const double e) {
                    constexpr int i = decltype(ic)::value;
                    if constexpr (i < 36) {
                        constexpr int r = i / 6, c2 = i % 6;
                        Gk[r] += e * Xk[c2];                    // A X
                        Pk[c2] += e * Xk[r];                    // A^T X
                    } else if constexpr (i < 57) {
                        // packed upper triangle: i - 36 counts (r, c2 >= r) row by row
                        constexpr int t = i - 36;
                        constexpr int r = t < 6 ? 0 : t < 11 ? 1 : t < 15 ? 2 : t < 18 ? 3 : t < 20 ? 4 : 5;
                        constexpr int c2 = r + (t - (r == 0 ? 0 : r == 1 ? 6 : r == 2 ? 11 : r == 3 ? 15 : r == 4 ? 18 : 20));
                        Rk[r] += e * Xk[c2]; Mdk[r] += e * dXk[c2];                     // M X, M dX'
                        if constexpr (r != c2) { Rk[c2] += e * Xk[r]; Mdk[c2] += e * dXk[r]; }
                    } else if constexpr (i < 63) {
                        rm += Xk[i - 57] * e;
                    } else if constexpr (MODE == 1 && i >= NACC - 6 && i < NACC) {
                        rg += Xk[i - (NACC - 6)] * e;
                    }
                };
                if (lscan && !TWO_PASS) {
                    typedef double D2 __attribute__((ext_vector_type(2)));
                    const int a = bsrc, top = a + mp->subsize[a] - 1;
                    const D2 *ph = reinterpret_cast<const D2 *>(STG + TBS * top);
                    const D2 *pl = reinterpret_cast<const D2 *>(STG + TBS * (a > 0 ? a - 1 : 0));
                    const bool keep = !((mp->rootmask >> a) & 1ull);       // (a root's sums start with itself: nothing to subtract,
                                                                            //  and the row before it belongs to another tree)
                    static_for_asc(std::make_integer_sequence<int, (NACC + 1) / 2>{}, [&](auto i2c) {
                        constexpr int i2 = decltype(i2c)::value;
                        const D2 h = ph[i2], l = pl[i2];
                        visit(std::integral_constant<int, 2 * i2>{}, (keep ? h.x - l.x : h.x));
                        if constexpr (2 * i2 + 1 < NACC) visit(std::integral_constant<int, 2 * i2 + 1>{}, (keep ? h.y - l.y : h.y));
                        if constexpr ((i2 & 3) == 3) asm volatile("" ::: "memory");   // four row pairs in flight
                    });
                } else if (lscan) {
                    typedef double D2 __attribute__((ext_vector_type(2)));
                    const int a = bsrc, top = a + mp->subsize[a] - 1;
                    const D2 *ph = reinterpret_cast<const D2 *>(STG + TBS * top);
                    const D2 *pl = reinterpret_cast<const D2 *>(STG + TBS * (a > 0 ? a - 1 : 0));
                    const bool keep = !((mp->rootmask >> a) & 1ull);       // (a root's sums start with itself: nothing to subtract,
                                                                            //  and the row before it belongs to another tree)
                    static_assert(TB_PASS1 % 2 == 0 && NACC - TB_PASS1 <= TB_STRIDE && TB_STRIDE <= TB_STRIDE1, "prefix table passes");
                    // (the accumulators of the two passes are disjoint -- R, M dX', rhs from M | rhs; G, P from A -- so the
                    // order of the passes does not change a bit of the results)
                    static_for_asc(std::make_integer_sequence<int, (NACC - TB_PASS1 + 1) / 2>{}, [&](auto i2c) {
                        constexpr int i2 = decltype(i2c)::value;
                        const D2 h = ph[i2], l = pl[i2];
                        visit(std::integral_constant<int, TB_PASS1 + 2 * i2>{}, (keep ? h.x - l.x : h.x));
                        if constexpr (TB_PASS1 + 2 * i2 + 1 < NACC) visit(std::integral_constant<int, TB_PASS1 + 2 * i2 + 1>{}, (keep ? h.y - l.y : h.y));
                        if constexpr ((i2 & 3) == 3) asm volatile("" ::: "memory");   // four row pairs in flight
                    });
                    WAVE_SYNC();                   // every lane has consumed the first pass: the table is rewritten
                    tb_pass(std::integral_constant<int, 0>{}, std::integral_constant<int, TB_PASS1>{}, false);
                    static_for_asc(std::make_integer_sequence<int, TB_PASS1 / 2>{}, [&](auto i2c) {
                        constexpr int i2 = decltype(i2c)::value;
                        const D2 h = ph[i2], l = pl[i2];
                        visit(std::integral_constant<int, 2 * i2>{}, (keep ? h.x - l.x : h.x));
                        visit(std::integral_constant<int, 2 * i2 + 1>{}, (keep ? h.y - l.y : h.y));
                        if constexpr ((i2 & 3) == 3) asm volatile("" ::: "memory");   // four row pairs in flight
                    });
                } else {
                    static_for_asc(std::make_integer_sequence<int, NACC>{}, [&](auto ic) { visit(ic, Acc[decltype(ic)::value]); });
                }
#pragma unroll
                for (int r = 0; r < 6; ++r) Gk[r] += Mdk[r];
                if constexpr (TRACK_GROWTH) {
                    double zd = 0.;
#pragma unroll
                    for (int r = 0; r < 6; ++r) zd += Xk[r] * Gk[r];          // Z[k][k] = X_k . G_k
                    zdiag = (float)zd;
                }
                rhsM = (lane < n) ? (T)rm : T(0);
                if constexpr (ELIM64) rhsM_d = (lane < n) ? rm : 0.;
                rhsG = (MODE == 1 && lane < n) ? (T)rg : T(0);
                WAVE_SYNC();                   // every lane is done with the staging area: it becomes XPR
                // ... and with the joints' own columns SC: their space becomes RT = [rhs | rows of J'], zero before
                // the constraint rows and the joint-limit selectors are written (entries >= ndof of a row stay zero)
                // (body-space columns: six rows per pair of bodies, padded to whole slabs of four for phase D)
                const int rt_rows = BODYCOL ? 4 * ((6 * ARB_UNI(mp->nbp) + 3) / 4) : ndol;
                for (int i = lane; i < (1 + rt_rows) * RS; i += WAVE) RT[i] = T(0);
                if (lane < n) {
                    double *o = STG + XPR_STRIDE * lane;
#pragma unroll
                    for (int i = 0; i < 6; ++i) { o[i] = Xk[i]; o[6 + i] = Pk[i]; o[12 + i] = Rk[i]; }
                }
                WAVE_SYNC();
            }
            // ---- lane = column k: rows of Z ----------------------------------------------------------
            ARB_BSTAMP(5);
            {
                typedef double D2 __attribute__((ext_vector_type(2)));
                // DFS numbering: rows related to column k are ancestors' (or own) dofs up to the last own dof
                // e_k, descendants' dofs after it
                const unsigned long long rel = (lane < n) ? (mp->upmask[lane] | mp->descmask[lane]) : 0ull;
                const unsigned rel_lo = (unsigned)rel, rel_hi = (unsigned)(rel >> 32);
                const int e_k = (lane < n) ? (mp->dof_off[bsrc] + mp->jnd[bsrc] - 1) : -1;
#if ARB_ROWS_SPLIT
                // Live ranges split by hand: the 18 float64 operands of the rows below become new values here, defined
                // right in front of their 44 x 18 uses.  (Compiled for three waves per SIMD the register allocator had
                // spilled six of them at their definition, far above, and reloaded them in every row: 265 scratch loads
                // per step, each waited for.)
#pragma unroll
                for (int i = 0; i < 6; ++i) asm volatile("" : "+v"(Gk[i]), "+v"(Xk[i]), "+v"(dXk[i]));
#endif
#ifndef ARB_ROWS_PAIR
#define ARB_ROWS_PAIR 1          // rows of Z per scheduling fence (round 6, measured on the three-wave build: 2 rows -2.3 % -- 54 more
                                 // spilled registers --, 4 rows +-0 %; 1 = row by row stays)
#endif
                constexpr int RP = (sizeof(T) == 4) ? ARB_ROWS_PAIR : 1;
#pragma unroll
                for (int i0 = 0; i0 < NMAX; i0 += RP) {
                    // (the wave-uniform branch per row also keeps the rows apart for the scheduler: as one
                    // branch-free block the compiler hoists the LDS reads of all NMAX rows and spills ~1500 VGPRs)
                    if (i0 < n) {
                        asm volatile("");          // not speculatable: a real scalar branch per row (pair), no if-conversion into lane masks
#pragma unroll
                      for (int i = i0; i < i0 + RP; ++i) {
                        if (i >= NMAX) continue;
                        if (i > i0 && !(i < n)) { Z[i] = ZT(0); continue; }
                        const D2 *xi = reinterpret_cast<const D2 *>(STG + XPR_STRIDE * i);   // wave-uniform: broadcast reads
                        double tu = 0., td = 0.;
                        // all nine reads of the row are issued before the first multiply-add (the asm defines the nine values
                        // at one point): one LDS round trip per row -- left to itself the compiler interleaves reads and
                        // multiply-adds in three round trips (+1.2 % end to end on two waves, +0.4 % on three; same arithmetic)
                        if constexpr (sizeof(T) == 4) {
                            D2 x9[9];
#pragma unroll
                            for (int j = 0; j < 9; ++j) x9[j] = xi[j];
                            asm volatile("" : "+v"(x9[0]), "+v"(x9[1]), "+v"(x9[2]), "+v"(x9[3]), "+v"(x9[4]), "+v"(x9[5]), "+v"(x9[6]), "+v"(x9[7]), "+v"(x9[8]));
                            // (round 5: three chains of six fused multiply-adds and one addition per row -- 19 float64
                            // instructions; written as sums of products, `tu += a.x * G0 + a.y * G1`, the front end's contraction
                            // rule made 24 of them: a multiply, a fused multiply-add and an addition per pair)
                            double tp = 0.;
#pragma unroll
                            for (int j = 0; j < 3; ++j) {
                                const D2 a = x9[j], pq = x9[3 + j], rq = x9[6 + j];
                                tu = fma(a.x, Gk[2 * j], tu); tu = fma(a.y, Gk[2 * j + 1], tu);
                                tp = fma(pq.x, Xk[2 * j], tp); tp = fma(pq.y, Xk[2 * j + 1], tp);
                                td = fma(rq.x, dXk[2 * j], td); td = fma(rq.y, dXk[2 * j + 1], td);
                            }
                            td += tp;
                        } else {          // (float64 kernels: their tile takes two registers per row, no room for nine reads in flight)
                            double tp = 0.;
#pragma unroll
                            for (int j = 0; j < 3; ++j) {
                                const D2 a = xi[j], pq = xi[3 + j], rq = xi[6 + j];
                                tu = fma(a.x, Gk[2 * j], tu); tu = fma(a.y, Gk[2 * j + 1], tu);
                                tp = fma(pq.x, Xk[2 * j], tp); tp = fma(pq.y, Xk[2 * j + 1], tp);
                                td = fma(rq.x, dXk[2 * j], td); td = fma(rq.y, dXk[2 * j + 1], td);
                            }
                            td += tp;
                        }
                        const ZT val = (ZT)((i <= e_k) ? tu : td);
                        Z[i] = (((i < 32 ? rel_lo : rel_hi) >> (i & 31)) & 1u) ? val : ZT(0);
                      }
                    } else {
#pragma unroll
                        for (int i = i0; i < i0 + RP; ++i) if (i < NMAX) Z[i] = ZT(0);
                    }
                }
            }
            // ---- constraint rows: s_k [Ad(c0<-g) X_k] with s_k = [k above body 1] - [k above body 0] --
            ARB_BSTAMP(6);
            if constexpr (BODYCOL) { if (do_constraints) {
                // ---- the six rows of every pair's relative Jacobian J_p = s_k [X_k moved to the pair's reference point]:
                // world axes about the origin o of the pair's reference body (the class has one tree: p0k is its root)
                const unsigned long long actm = __ballot(lane < nc && CD[(lane < nc ? lane : 0) * CD_STRIDE + CD_ACTIVE] != T(0));
                const int nbp = ARB_UNI(mp->nbp);
                for (int p = 0; p < nbp; ++p) {
                    if ((actm & mp->pair_cmask[p]) == 0ull) continue;      // no contact of the pair is active: the rows stay zero
                    const double sgn = (double)((mp->pair_a1[p] >> lane) & 1ull) - (double)((mp->pair_a0[p] >> lane) & 1ull);
                    const V3<double> o = ld_v3(PD + PDS * mp->pair_ref[p] + 9) - p0k;
                    const V3<double> xw = v3<double>(Xk[0], Xk[1], Xk[2]);
                    const V3<double> jv = v3<double>(Xk[3], Xk[4], Xk[5]) + cross(xw, o);       // velocity of the point o
                    if (lane < n) {
                        T *row = RT + (1 + 6 * p) * RS + lane;
                        row[0] = (T)(sgn * xw.x); row[RS] = (T)(sgn * xw.y); row[2 * RS] = (T)(sgn * xw.z);
                        row[3 * RS] = (T)(sgn * jv.x); row[4 * RS] = (T)(sgn * jv.y); row[5 * RS] = (T)(sgn * jv.z);
                    }
                }
            } } else
            if (do_constraints) {
                for (int c = 0; c < nc; ++c) {
                    const int *ci = CI + CI_STRIDE * c;
                    const int ct = SPEC ? (int)ARB_CT_SOFTFINGER : ci[0];
                    if (ct == ARB_CT_JOINTLIMITS) continue;
                    const T *cd = CD + c * CD_STRIDE;
                    if (cd[CD_ACTIVE] == T(0)) {            // not in the active set: zero rows (core.py:913-918)
                        if (lane < n) {
                            T *row = RT + (1 + 4 * c) * RS + lane;
                            row[0] = T(0); row[RS] = T(0); row[2 * RS] = T(0);
                            if (ct == ARB_CT_SOFTFINGER) row[3 * RS] = T(0);
                        }
                        continue;
                    }
                    const unsigned long long a1 = ((unsigned long long)(unsigned)ci[2] << 32) | (unsigned)ci[1];
                    const unsigned long long a0 = ((unsigned long long)(unsigned)ci[4] << 32) | (unsigned)ci[3];
                    const double s = (double)cd[CD_ACTIVE] * ((double)((a1 >> lane) & 1ull) - (double)((a0 >> lane) & 1ull));
                    const M3<double> Rx = ld_m3_as<double>(cd + CD_R1);
                    // (the frame was stored about the root of body 1's tree; the columns of a dof of another tree are about
                    // that tree's root: shift by the difference -- exactly zero inside the frame's own tree)
                    const V3<double> px = cvt_v3<double>(ld_v3(cd + CD_P1)) + mv(Rx, p0k - ld_v3(PD + PDS * ci[6] + 9));
                    const V3<double> cw = mv(Rx, v3<double>(Xk[0], Xk[1], Xk[2]));
                    const V3<double> cv = mv(Rx, v3<double>(Xk[3], Xk[4], Xk[5])) + cross(px, cw);
                    if (lane < n) {
                        T *row = RT + (1 + 4 * c) * RS + lane;
                        if (ct == ARB_CT_SOFTFINGER) {          // rows (w_z, v_x, v_y, v_z)        constraints.py:429-433
                            row[0] = (T)(s * cw.z); row[RS] = (T)(s * cv.x); row[2 * RS] = (T)(s * cv.y); row[3 * RS] = (T)(s * cv.z);
                        } else {                                // BallAndSocket linear rows         constraints.py:203-207
                            row[0] = (T)(s * cv.x); row[RS] = (T)(s * cv.y); row[2 * RS] = (T)(s * cv.z);
                        }
                    }
                }
            }
            // ---- inspect: body Jacobians J_b = Ad(b<-g) X, dJ_b = Ad(b<-g) dX' + ad(Om_b) J_b -----------
            if (MODE == 1 && step == 0 && (dbg.jac != nullptr || dbg.djac != nullptr)) {
                for (int b = 0; b < nb; ++b) {
                    const M3<double> R = ld_m3(PD + PDS * b);
                    const V3<double> p = ld_v3(PD + PDS * b + 9) - ld_v3(PD + PDS * mp->root[b] + 9);
                    const V3<double> obw = v3<double>((double)bcast(om_b[0], b), (double)bcast(om_b[1], b), (double)bcast(om_b[2], b));
                    const V3<double> obv = v3<double>((double)bcast(om_b[3], b), (double)bcast(om_b[4], b), (double)bcast(om_b[5], b));
                    const bool mine = (lane < n) && ((mp->anc[b] >> lane) & 1ull);
                    const V3<double> xw = v3<double>(Xk[0], Xk[1], Xk[2]), xv = v3<double>(Xk[3], Xk[4], Xk[5]);
                    const V3<double> dw = v3<double>(dXk[0], dXk[1], dXk[2]), dv = v3<double>(dXk[3], dXk[4], dXk[5]);
                    const V3<double> jw = mtv(R, xw), jv = mtv(R, xv - cross(p, xw));
                    const V3<double> ew = mtv(R, dw) + cross(obw, jw);
                    const V3<double> ev = mtv(R, dv - cross(p, dw)) + cross(obv, jw) + cross(obw, jv);
                    if (lane < n) {
                        const double j6[6] = {jw.x, jw.y, jw.z, jv.x, jv.y, jv.z}, e6[6] = {ew.x, ew.y, ew.z, ev.x, ev.y, ev.z};
                        for (int i = 0; i < 6; ++i) {
                            if (dbg.jac != nullptr) dbg.jac[((w * nb + b) * 6 + i) * n + lane] = mine ? (T)j6[i] : T(0);
                            if (dbg.djac != nullptr) dbg.djac[((w * nb + b) * 6 + i) * n + lane] = mine ? (T)e6[i] : T(0);
                        }
                    }
                }
            }
        }
        // joint-limit rows are dof selectors                              constraints.py:46-48
        if (!SPEC && do_constraints) {
            for (int c = 0; c < nc; ++c)
                if (CI[CI_STRIDE * c] == ARB_CT_JOINTLIMITS && lane == CI[CI_STRIDE * c + 5])
                    RT[(1 + 4 * c) * RS + lane] = CD[c * CD_STRIDE + CD_ACTIVE];
        }
        // controllers: gravity is in rhsG; PD adds to both sides         controllers.py:141-158
        gf0 = rhsG + ext_k;
        // (RH: float64 in the mixed build -- a serial chain answers a generalized-force error of 1e-7 of the gravity torques with a
        // velocity error of 1e-3: the smallest eigenvalue of its mass matrix is 1e-8 of the largest)
        using RH = std::conditional_t<ELIM64, double, T>;
        RH rhs = (ELIM64 ? (RH)rhsM_d : (RH)rhsM) + ext_k;          // gforce - (N + B + Z_pd) gvel
        if (pwd.kp != nullptr) {
            // per-world diagonal gains and targets (arb_step_ex): tau0 = kp (qdes - q) + kd dqdes, Z += dt kp + kd
            if (lane < n && !lane_dead) {
                const T kp = pwd.kp[w * n + lane], kd = pwd.kd[w * n + lane];
                const long pdo = (long)step * pd_stride + w * n;      // (this step's targets: arb_step_args.pd_qdes_steps)
                const T acc = kp * (pwd.qdes[pdo + lane] - qd[lane]) + kd * pwd.dqdes[pdo + lane];
                const T zd = dt * kp + kd;
                gf0 += acc;
                rhs += acc - zd * dqs[lane];
                if (MODE == 0 || zmode == 0) {
#pragma unroll
                    for (int i = 0; i < NMAX; ++i) Z[i] += (i == lane) ? zd : T(0);
                }
            }
        } else if (!SPEC && mp->has_pd && lane < n && !lane_dead) {
            // model gains (controllers.py:141-158); per-world targets replace the model's tau0 when given
            T acc = (pwd.qdes != nullptr) ? T(0) : mp->pd_tau0[lane], accv = T(0);
            for (int i = 0; i < n; ++i) {
                const T kp = mp->pd_kp[lane * n + i], kd = mp->pd_kd[lane * n + i];
                if (pwd.qdes != nullptr) {
                    // (block-diagonal gains: the targets of another copy meet exact zeros -- which a NaN target of a
                    // retired copy would turn into NaN: kp = kd = 0 means no term)
                    if (kp != T(0) || kd != T(0)) acc += kp * (pwd.qdes[(long)step * pd_stride + w * n + i] - qd[i]) + kd * pwd.dqdes[(long)step * pd_stride + w * n + i];
                } else acc -= kp * qd[i];
                accv += (dt * kp + kd) * dqs[i];
            }
            gf0 += acc;
            rhs += acc - accv;
            if (MODE == 0 || zmode == 0) {
                // (a size of its own: sharing `i < n` with the rows of phase B keeps 44 lane masks alive, spilled)
                int npd = ARB_UNI(mp->n);
                asm volatile("" : "+s"(npd));
                const T *kpp = mp->pd_kp, *kdp = mp->pd_kd;
#pragma unroll
                for (int i = 0; i < NMAX; ++i)
                    if (i < npd) Z[i] += dt * kpp[i * npd + lane] + kdp[i * npd + lane];
            }
        }
        // (ABI 8) the impedance of user-defined Controllers, a dense matrix per world: Z -= Z_a (core.py:815-817); in the
        // increment form Z (gvel+ - gvel) = gforce - (N + B - Z_a) gvel the right-hand side gains Z_a gvel.  Lane = column k
        // reads row i of Z_a coalesced; lane = dof i sums its own row for the right-hand side.
        if (FEAT_ALL && pwd.zimp != nullptr) {
            int nz = ARB_UNI(mp->n);
            asm volatile("" : "+s"(nz));          // (a size of its own, as for the PD gains above)
            const T *za = pwd.zimp + (long)w * nz * nz;
            if (lane < nz && !lane_dead) {
                if (MODE == 0 || zmode == 0) {
#pragma unroll
                    for (int i = 0; i < NMAX; ++i)
                        if (i < nz) Z[i] -= (ZT)za[i * nz + lane];
                }
                T accz = T(0);
                for (int i = 0; i < nz; ++i) accz += za[lane * nz + i] * dqs[i];
                rhs += accz;
            }
        }
        WAVE_SYNC();
        if (MODE == 1) {
            if (dbg.Zout != nullptr && lane < n) {
#pragma unroll
                for (int i = 0; i < NMAX; ++i) if (i < n) dbg.Zout[(w * n + i) * n + lane] = (T)Z[i];
            }
            if (zmode != 0) return;
            if (dbg.gforce0 != nullptr && lane < n) dbg.gforce0[w * n + lane] = gf0;
            if (BODYCOL && dbg.c_jac != nullptr && lane < n) {
                // (body-space columns: J'_c = T_c J_p, formed here for the output only)
                for (int i = 0; i < ndol; ++i) {
                    const int c = i >> 2, pp = mp->cpair[c];
                    T acc = T(0);
                    for (int j = 0; j < 6; ++j) acc += CD[c * CD_STRIDE + 6 * (i & 3) + j] * RT[(1 + 6 * pp + j) * RS + lane];
                    dbg.c_jac[(w * ndol + i) * n + lane] = do_constraints ? acc : T(0);
                }
            } else
            if (dbg.c_jac != nullptr && lane < n)
                for (int i = 0; i < ndol; ++i) dbg.c_jac[(w * ndol + i) * n + lane] = do_constraints ? RT[(1 + i) * RS + lane] : T(0);
            if (lane < nc) {
                const T *cd = CD + lane * CD_STRIDE;
                if (dbg.c_sdist != nullptr) dbg.c_sdist[w * nc + lane] = do_constraints ? cd[CD_SDIST] : T(0);
                if (dbg.c_active != nullptr) dbg.c_active[w * nc + lane] = (do_constraints && cd[CD_ACTIVE] != T(0)) ? 1 : 0;
            }
        }
        // warm-started constraint forces enter the right-hand side          core.py:921-924
        if (!SPEC && do_constraints && mp->has_warm && lane < n) {
            for (int i = 0; i < ndol; ++i) rhs += RT[(1 + i) * RS + lane] * FF[i];
        }
