// arb_phase_a.h -- FRAGMENT of arb_step_kernel (arb_step_kernel.h), included inside its step loop: phase A, lane = body -- joint-local kinematics, poses / twists / accelerations down the tree, per-body blocks.
// Not a header of its own: it reads and writes the kernel's locals (LDS pointers, the register tile, the laundered sizes).
        // ================= phase A: lane = body ===========================
        ARB_OPAQUE_LANE();
        ARB_STAMP(0);
#if ARB_ANY_PRIO
        __builtin_amdgcn_s_setprio(ARB_A_PRIO);
#endif
        if (FEAT_ALL && dts != nullptr) { dt = (T)dts[step]; inv_dt = T(1) / dt; }
        // (forests are built on the 16- and 32-row tiles only: the larger kernels carry none of this)
        const int fk = (NMAX > 32) ? 1 : ARB_UNI(mp->fk);
        if (fk > 1) {               // forest world: retire the copies that have left the finite range (see `dead`)
            const int fn = ARB_UNI(mp->fn), fnq = ARB_UNI(mp->fnq), fnd = ARB_MAXDOL * ARB_UNI(mp->fnc);
            const T lim = (T)(sizeof(T) == 4 ? 1e8 : 1e100);
            unsigned bad = 0u;
            for (int i = lane; i < nq; i += WAVE) if (!(fabs(qs[i]) <= lim)) bad |= 1u << (i / fnq);
            if (lane < n && !(fabs(dqs[lane]) <= lim)) bad |= 1u << (lane / fn);
            for (int i = lane; i < ndol; i += WAVE) if (!(fabs(FF[i]) <= lim)) bad |= 1u << (i / fnd);
            // ... and the copy's per-world INPUTS (round 4): a NaN or Inf in one world's user torques or PD targets / gains
            // would go through the shared elimination like a NaN in its state
            if (lane < n) {
                bool in_bad = !(fabs(ext_k) <= lim);
                if (FEAT_ALL && pwd.qdes != nullptr)
                    in_bad = in_bad || !(fabs(pwd.qdes[(long)step * pd_stride + w * n + lane]) <= lim) || !(fabs(pwd.dqdes[(long)step * pd_stride + w * n + lane]) <= lim);
                if (FEAT_ALL && pwd.kp != nullptr)
                    in_bad = in_bad || !(fabs(pwd.kp[w * n + lane]) <= lim) || !(fabs(pwd.kd[w * n + lane]) <= lim);
                if (in_bad) bad |= 1u << (lane / fn);
            }
            unsigned long long some = __ballot(bad != 0u);
            while (some != 0ull) {                           // wave-uniform, rare
                dead |= (unsigned)__builtin_amdgcn_readlane((int)bad, __builtin_ctzll(some));
                some &= some - 1ull;
            }
        }
        if (MODE == 0) {            // trajectory log: what an Observer sees at time t (core.py:1361-1362)
            if (dead == 0u) {
                if (logo.q != nullptr) for (int i = lane; i < nq; i += WAVE) logo.q[((long)step * nworlds + w) * nq + i] = qs[i];
                if (logo.dq != nullptr && lane < n) logo.dq[((long)step * nworlds + w) * n + lane] = dqs[lane];
            } else {
                const int fn = ARB_UNI(mp->fn), fnq = ARB_UNI(mp->fnq);
                if (logo.q != nullptr)
                    for (int i = lane; i < nq; i += WAVE)
                        logo.q[((long)step * nworlds + w) * nq + i] = ((dead >> (i / fnq)) & 1u) ? (T)NAN : qs[i];
                if (logo.dq != nullptr && lane < n)
                    logo.dq[((long)step * nworlds + w) * n + lane] = ((dead >> (lane / fn)) & 1u) ? (T)NAN : dqs[lane];
            }
        }
        bool lane_dead = false;     // this lane's dof belongs to a retired copy: its inputs are ignored from now on
        if (dead != 0u) {           // retired copies compute on a state of rest
            const int fn = ARB_UNI(mp->fn), fnq = ARB_UNI(mp->fnq), fnd = ARB_MAXDOL * ARB_UNI(mp->fnc);
            for (int i = lane; i < nq; i += WAVE) if ((dead >> (i / fnq)) & 1u) qs[i] = mp->qdef[i];
            lane_dead = lane < n && ((dead >> (lane / fn)) & 1u);
            if (lane_dead) ext_k = T(0);
            if (lane < n && ((dead >> (lane / fn)) & 1u)) dqs[lane] = T(0);
            for (int i = lane; i < ndol; i += WAVE) if ((dead >> (i / fnd)) & 1u) FF[i] = T(0);
            WAVE_SYNC();
        }
        {
            const int b = lane;
            const bool on = b < nb;
            int jt = 0, par = -1, doff = 0, dep = -1, k = 0;
            // Positions are chained in float64 whatever the state type: the contact gap
            // (sdist) is a difference of O(1 m) positions that is then divided by dt, so
            // float32 rounding of the pose chain alone would cost ~1e-7/dt = 2e-5 m/s.
            M3<T> R_pc, R_cp, R_cn, dA_cp, dB_cp; V3<T> p_pc, p_cp, p_cn, Tnw, Tnv, Bnw, Bnv;
            R_pc = R_cp = R_cn = m3_identity<T>();
            dA_cp = dB_cp = m3_zero<T>();
            p_pc = p_cp = p_cn = Tnw = Tnv = Bnw = Bnv = v3<T>(T(0), T(0), T(0));
            if (on) {
                jt = mp->jtype[b]; par = mp->parent[b]; doff = mp->dof_off[b]; dep = mp->depth[b];
                k = mp->jnd[b];
                JointLocal<double> jld;
                if (ARB_DENSE_SC) {
                    // (lane-dense, see ARB_DENSE: the sin / cos of the joint's angles on every body lane, before the
                    // joint-type switch, whose cases run with the 1-7 lanes of one joint type enabled; a FreeJoint's
                    // "angles" are entries of its pose matrix, unused)
                    const T *qj = qs + mp->q_off[b];
                    double ps[3], pc[3];
#pragma unroll
                    for (int i = 0; i < 3; ++i) arb_sincos((double)qj[i], &ps[i], &pc[i]);
                    joint_local<double>(jt, qj, (const T *)(dqs + doff), jld, ps, pc);
                } else {
                    joint_local<double>(jt, qs + mp->q_off[b], dqs + doff, jld);
                }
                JointLocal<T> jl;
                jl.R = cvt_m3<T>(jld.R); jl.p = cvt_v3<T>(jld.p);
#pragma unroll
                for (int i = 0; i < 3; ++i) { jl.jw[i] = cvt_v3<T>(jld.jw[i]); jl.djw[i] = cvt_v3<T>(jld.djw[i]); }
                jl.Tw = cvt_v3<T>(jld.Tw); jl.Tv = cvt_v3<T>(jld.Tv);
                const M3<T> R_pr = ld_m3(mp->Hpr + 12 * b);
                const V3<T> p_pr = ld_v3(mp->Hpr + 12 * b + 9);
                R_cn = ld_m3(mp->Hcn + 12 * b);
                p_cn = ld_v3(mp->Hcn + 12 * b + 9);
                // H_pc = H_pr H_rn inv(H_cn)                       core.py:1298
                {
                    const M3<double> Rpr = ld_m3(mp->Hpr_d + 12 * b), Rcn = ld_m3(mp->Hcn_d + 12 * b);
                    const V3<double> ppr = ld_v3(mp->Hpr_d + 12 * b + 9), pcn = ld_v3(mp->Hcn_d + 12 * b + 9);
                    const M3<double> R_rc = mulBT(jld.R, Rcn);
                    const V3<double> p_rc = mv(jld.R, -mtv(Rcn, pcn)) + jld.p;
                    const M3<double> R_pc_d = mul(Rpr, R_rc);
                    const V3<double> p_pc_d = mv(Rpr, p_rc) + ppr;
                    // parked in the body's own pose slot until its depth level comes (24 registers less across
                    // the level loop: phase A is the register-pressure peak of the kernel)
                    st_m3(PD + PDS * b, R_pc_d); st_v3(PD + PDS * b + 9, p_pc_d);
                    R_pc = cvt_m3<T>(R_pc_d);
                    p_pc = cvt_v3<T>(p_pc_d);
                }
                ARB_ASTAMP(1);
                R_cp = transpose(R_pc);                          // Ad_cp = Ad(inv(H_pc)) :1300
                p_cp = -mtv(R_pc, p_pc);
                // Ad_nr, T_rn = -Ad_nr T_nr, dAd_nr = Ad_nr ad(T_rn)   rigidmotion.py:47-73
                const M3<T> R_nr = transpose(jl.R);
                const V3<T> p_nr = -mtv(jl.R, jl.p);
                const V3<T> aw = mv(R_nr, jl.Tw);
                const V3<T> av = cross(p_nr, aw) + mv(R_nr, jl.Tv);
                const Blk<T> Ad_nr = blk_adjoint(R_nr, p_nr);
                const Blk<T> dAd_nr = blk_mul(Ad_nr, blk_adjacency(-aw, -av));
                // dAd_cp = Ad_cn dAd_nr Ad_rp                        core.py:1304
                const Blk<T> Ad_cn = blk_adjoint(R_cn, p_cn);
                const Blk<T> Ad_rp = blk_adjoint(transpose(R_pr), -mtv(R_pr, p_pr));
                const Blk<T> dAd_cp = blk_mul(Ad_cn, blk_mul(dAd_nr, Ad_rp));
                T *bd = BD + b * BDS;
                st_m3(bd + BD_RCP, R_cp); st_v3(bd + BD_PCP, p_cp);
                // T_rn = -(aw, av) is all phase B needs from here (dAd_cp = ad(W_c) Ad_cp, W_c = Ad_cp Ad_pr T_rn)
                st_v3(bd + BD_OM, -aw); st_v3(bd + BD_OM + 3, -av);
                dA_cp = dAd_cp.A; dB_cp = dAd_cp.B;
                ARB_ASTAMP(2);
                // Ad_cn (dJ_nr gvel_j): the joint's own contribution to dJ_c gvel
                {
                    V3<T> bw = v3<T>(T(0), T(0), T(0));
                    if (jt != JT_FREE && jt != JT_TXTYTZ) {
#pragma unroll
                        for (int i = 0; i < 3; ++i) if (i < k) bw = bw + dqs[doff + i] * jl.djw[i];
                    }
                    Bnw = mv(R_cn, bw);
                    Bnv = cross(p_cn, Bnw);
                }
                // own columns Ad_cn J_nr, Ad_cn dJ_nr               core.py:1310, 1313
                Tnw = mv(R_cn, jl.Tw);
                Tnv = cross(p_cn, Tnw) + mv(R_cn, jl.Tv);
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    // (lane-dense, see ARB_DENSE: column i exists on the lanes of joints with more than i dofs -- the one
                    // FreeJoint for i >= 3 --; every body lane computes it when any has it, those lanes store)
                    if (ARB_DENSE_COL ? (__ballot(i < k) != 0ull) : (i < k)) {
                        V3<T> cw = v3<T>(T(0), T(0), T(0)), cv = cw, dw = cw;
                        if (jt == JT_FREE) {
                            if (i < 3) cw = v3<T>(i == 0 ? T(1) : T(0), i == 1 ? T(1) : T(0), i == 2 ? T(1) : T(0));
                            else cv = v3<T>(i == 3 ? T(1) : T(0), i == 4 ? T(1) : T(0), i == 5 ? T(1) : T(0));
                        } else if (jt == JT_TXTYTZ) {
                            cv = v3<T>(i == 0 ? T(1) : T(0), i == 1 ? T(1) : T(0), i == 2 ? T(1) : T(0));
                        } else if (i < 3) {
                            cw = jl.jw[i]; dw = jl.djw[i];
                        }
                        const V3<T> ow = mv(R_cn, cw);
                        const V3<T> ov = cross(p_cn, ow) + mv(R_cn, cv);
                        const V3<T> dow = mv(R_cn, dw);
                        const V3<T> dov = cross(p_cn, dow);
                        const int col = doff + i;
                        if (ARB_DENSE_COL) { keep(ow); keep(ov); keep(dow); keep(dov); }
                        if (i < k) {
                            SC[0 * RS + col] = ow.x; SC[1 * RS + col] = ow.y; SC[2 * RS + col] = ow.z;
                            SC[3 * RS + col] = ov.x; SC[4 * RS + col] = ov.y; SC[5 * RS + col] = ov.z;
                            SC[6 * RS + col] = dow.x; SC[7 * RS + col] = dow.y; SC[8 * RS + col] = dow.z;
                            SC[9 * RS + col] = dov.x; SC[10 * RS + col] = dov.y; SC[11 * RS + col] = dov.z;
                        }
                    }
                }
            }
            ARB_ASTAMP(3);
#ifndef ARB_JUMP_DEPTH
#define ARB_JUMP_DEPTH 12      // float64 kernels: trees deeper than this chain pose, twist and bias acceleration in log2(depth) rounds
#endif
            // Deep trees (the 64-link snake: 65 levels of ~2 k cycles each, one lane working -- 133 k of the step's 346 k
            // cycles), float64 kernels: log-depth instead.  (1) Poses by pointer jumping: every body composes its pose
            // with its current ancestor's and takes over that ancestor's ancestor, ceil(log2(depth + 1)) rounds.
            // (2) Twists: in WORLD axes about the world origin a body's twist is its parent's plus its own joint's,
            // Ad(H_gc) T_c = Ad(H_gp) T_p + Ad(H_gc) Tn_c, a prefix sum over the ancestors -- pointer jumping again -- and
            // back to body axes.  (3) Bias accelerations likewise: Ad(H_gc) a_c = Ad(H_gp) a_p + Ad(H_gc)(dAd_cp T_p + Bn_c)
            // with the parent's twist from (2).  World-frame sums carry lever arms of the size of the robot: float64 only
            // (the float32 kernels and shallow trees keep the level loop below, whose operation order they are tested with).
            bool jumped = false;
            if constexpr (sizeof(T) == 8 || CM == 3) {      // (CM 3: the mixed build, float32 sums of float64 terms)
                const int maxdep = ARB_UNI(mp->maxdepth);
                if (!SPEC && maxdep >= ARB_JUMP_DEPTH) {          // (the specialised kernels' class: shallow trees)
                    jumped = true;
                    int rounds = 0;
                    while ((1 << rounds) < maxdep + 1) ++rounds;
                    T *const bdl = BD + (on ? b : 0) * BDS;
                    // ancestor pointers travel in the scratch array (unused until the end of phase A)
                    auto jump_sum = [&](int slot) {          // inclusive sum over the ancestors of the 6-vectors in `slot`
                        if (on) WORK[b] = (T)par;
                        WAVE_SYNC();
                        for (int r = 0; r < rounds; ++r) {
                            const int a = on ? (int)WORK[b] : -1;
                            T add6[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
                            T na = T(-1);
                            if (a >= 0) {
                                const T *ab = BD + a * BDS;
#pragma unroll
                                for (int i = 0; i < 6; ++i) add6[i] = ab[slot + i];
                                na = WORK[a];
                            }
                            WAVE_SYNC();
                            if (a >= 0) {
#pragma unroll
                                for (int i = 0; i < 6; ++i) bdl[slot + i] += add6[i];
                                WORK[b] = na;
                            }
                            WAVE_SYNC();
                        }
                    };
                    // (1) poses: PD[b] holds H_pc; after the rounds H_gb
                    if (on) WORK[b] = (T)par;
                    WAVE_SYNC();
                    for (int r = 0; r < rounds; ++r) {
                        const int a = on ? (int)WORK[b] : -1;
                        M3<double> Ra = m3_identity<double>(); V3<double> pa = v3<double>(0., 0., 0.);
                        T na = T(-1);
                        if (a >= 0) { Ra = ld_m3(PD + PDS * a); pa = ld_v3(PD + PDS * a + 9); na = WORK[a]; }
                        WAVE_SYNC();
                        if (a >= 0) {
                            const M3<double> Rb = ld_m3(PD + PDS * b);
                            const V3<double> pb2 = ld_v3(PD + PDS * b + 9);
                            st_m3(PD + PDS * b, mul(Ra, Rb)); st_v3(PD + PDS * b + 9, mv(Ra, pb2) + pa);
                            WORK[b] = na;
                        }
                        WAVE_SYNC();
                    }
                    // (2) twists
                    M3<double> Rgb = m3_identity<double>(); V3<double> pgb = v3<double>(0., 0., 0.);
                    if (on) {
                        Rgb = ld_m3(PD + PDS * b); pgb = ld_v3(PD + PDS * b + 9);
                        const V3<double> ww = mv(Rgb, cvt_v3<double>(Tnw));
                        st_v3(bdl + BD_TW, cvt_v3<T>(ww)); st_v3(bdl + BD_TW + 3, cvt_v3<T>(cross(pgb, ww) + mv(Rgb, cvt_v3<double>(Tnv))));
                    }
                    jump_sum(BD_TW);
                    if (on) {
                        const V3<double> ww = ld_v3_as<double>(bdl + BD_TW), wv = ld_v3_as<double>(bdl + BD_TW + 3);
                        st_v3(bdl + BD_TW, cvt_v3<T>(mtv(Rgb, ww))); st_v3(bdl + BD_TW + 3, cvt_v3<T>(mtv(Rgb, wv - cross(pgb, ww))));
                    }
                    WAVE_SYNC();
                    // (3) bias accelerations: dAd_cp T_p + Bn_c in body axes, to world axes, summed, back
                    if (on) {
                        V3<double> tw = v3<double>(0., 0., 0.), tv = tw;
                        if (par >= 0) { const T *pb = BD + par * BDS; tw = ld_v3_as<double>(pb + BD_TW); tv = ld_v3_as<double>(pb + BD_TW + 3); }
                        const M3<double> dAd = cvt_m3<double>(dA_cp), dBd = cvt_m3<double>(dB_cp);
                        const V3<double> lw = mv(dAd, tw) + cvt_v3<double>(Bnw);
                        const V3<double> lv = mv(dBd, tw) + mv(dAd, tv) + cvt_v3<double>(Bnv);
                        const V3<double> ww = mv(Rgb, lw);
                        st_v3(bdl + BD_AB, cvt_v3<T>(ww)); st_v3(bdl + BD_AB + 3, cvt_v3<T>(cross(pgb, ww) + mv(Rgb, lv)));
                    }
                    jump_sum(BD_AB);
                    if (on) {
                        const V3<double> ww = ld_v3_as<double>(bdl + BD_AB), wv = ld_v3_as<double>(bdl + BD_AB + 3);
                        st_v3(bdl + BD_AB, cvt_v3<T>(mtv(Rgb, ww))); st_v3(bdl + BD_AB + 3, cvt_v3<T>(mtv(Rgb, wv - cross(pgb, ww))));
                    }
                    WAVE_SYNC();
                }
            }
            // pose and twist down the tree, one depth level at a time
#if ARB_ALVL_PRIO
            __builtin_amdgcn_s_setprio(ARB_ALVL_PRIO);
#endif
            if (!jumped)
            for (int lvl = 0; lvl <= mp->maxdepth; ++lvl) {
                // (lane-dense, see ARB_DENSE: every lane goes through the level's arithmetic -- a lane of another level on
                // whatever its parent's block holds at the moment, a lane without a body on body 0 --, the bodies of this
                // level store)
                const bool mine = on && dep == lvl;
                if (ARB_DENSE_LVL || mine) {
                    const int bb = on ? b : 0;
                    M3<double> Rg = m3_identity<double>(); V3<double> pg = v3<double>(0., 0., 0.);
                    V3<T> tw = v3<T>(T(0), T(0), T(0)), tv = tw, aw = tw, av = tw;
                    if (par >= 0) {
                        const T *pb = BD + par * BDS;
                        Rg = ld_m3(PD + PDS * par); pg = ld_v3(PD + PDS * par + 9);
                        tw = ld_v3(pb + BD_TW); tv = ld_v3(pb + BD_TW + 3);
                        aw = ld_v3(pb + BD_AB); av = ld_v3(pb + BD_AB + 3);
                    }
                    const M3<double> R_pc_d = ld_m3(PD + PDS * bb);
                    const V3<double> p_pc_d = ld_v3(PD + PDS * bb + 9);
                    const M3<double> Rc_d = mul(Rg, R_pc_d);         // child_pose  core.py:1299
                    const V3<double> pc_d = mv(Rg, p_pc_d) + pg;
                    const V3<T> cw = mv(R_cp, tw) + Tnw;             // child_twist core.py:1308
                    const V3<T> cv = cross(p_cp, mv(R_cp, tw)) + mv(R_cp, tv) + Tnv;
                    // dJ_c gvel = dAd_cp T_p + Ad_cp (dJ_p gvel) + Ad_cn dJ_nr gvel_j   (core.py:1312-1313 times gvel)
                    const V3<T> raw = mv(R_cp, aw);
                    const V3<T> nbw = mv(dA_cp, tw) + raw + Bnw;
                    const V3<T> nbv = mv(dB_cp, tw) + mv(dA_cp, tv) + cross(p_cp, raw) + mv(R_cp, av) + Bnv;
                    if (ARB_DENSE_LVL) { keep(Rc_d); keep(pc_d); keep(cw); keep(cv); keep(nbw); keep(nbv); }
                    if (mine) {
                        T *bd = BD + b * BDS;
                        st_m3(PD + PDS * b, Rc_d); st_v3(PD + PDS * b + 9, pc_d);
                        st_v3(bd + BD_TW, cw); st_v3(bd + BD_TW + 3, cv);
                        st_v3(bd + BD_AB, nbw); st_v3(bd + BD_AB + 3, nbv);
                    }
                }
                WAVE_SYNC();
            }
            ARB_ASTAMP(4);
#if ARB_ALVL_PRIO
            __builtin_amdgcn_s_setprio(ARB_A_PRIO);
#endif
            if (on) {
                T *bd = BD + b * BDS;
                const T *Mb = mp->mass + 36 * b;
                T tw[6], ab[6], mt[6], ma[6], mg[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) { tw[i] = bd[BD_TW + i]; ab[i] = bd[BD_AB + i]; }
                mat6_vec<T>(Mb, tw, mt);
                mat6_vec<T>(Mb, ab, ma);
                // gravity in the body frame: Ad(inv(H_gb)) [0; g up]   controllers.py:56-58
                T g6[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
                if (mp->has_grav && mp->weighted[b]) {
                    const M3<T> Rg = cvt_m3<T>(ld_m3(PD + PDS * b));
                    const V3<T> gl = mtv(Rg, v3<T>(mp->grav[0], mp->grav[1], mp->grav[2]));
                    g6[3] = gl.x; g6[4] = gl.y; g6[5] = gl.z;
                }
                mat6_vec<T>(Mb, g6, mg);
#pragma unroll
                for (int i = 0; i < 6; ++i) if (MODE == 1) bd[BD_PG + i] = mg[i];
                // N_b = [[wx, rx wx - wx rx],[0, wx]] M_b              core.py:1276-1288
                const V3<T> wv = v3<T>(tw[0], tw[1], tw[2]);
                const M3<T> wx = hat(wv);
                M3<T> rx = m3_zero<T>();
                const T mm = Mb[21];
                if (!(mm <= T(1e-10))) {
                    const T im = T(1) / mm;
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int j = 0; j < 3; ++j) rx.a[3 * i + j] = Mb[6 * i + 3 + j] * im;
                }
                const M3<T> Cm = sub(mul(rx, wx), mul(wx, rx));
                // increment form of core.py:975-976: Z (gvel+ - gvel) = gforce - (N + B) gvel, and
                // (N gvel)|_b = M_b (dJ_b gvel) + N_b T_b ;  (B gvel)|_b = B_b T_b
                const V3<T> mtt = v3<T>(mt[0], mt[1], mt[2]), mtb = v3<T>(mt[3], mt[4], mt[5]);
                const V3<T> ntop = cross(wv, mtt) + mv(Cm, mtb);
                const V3<T> nbot = cross(wv, mtb);
                T pt[6] = {mg[0] - ma[0] - ntop.x, mg[1] - ma[1] - ntop.y, mg[2] - ma[2] - ntop.z,
                           mg[3] - ma[3] - nbot.x, mg[4] - ma[4] - nbot.y, mg[5] - ma[5] - nbot.z};
                if (!SPEC && mp->has_visc) {
                    T vt[6];
                    mat6_vec<T>(mp->visc + 36 * b, tw, vt);
#pragma unroll
                    for (int i = 0; i < 6; ++i) pt[i] -= vt[i];
                }
#pragma unroll
                for (int i = 0; i < 6; ++i) bd[BD_PT + i] = pt[i];
            }
            ARB_ASTAMP(5);
            // dof-indexed copy of the linear joint positions (PD controller, joint limits)
            if (lane < n) { const int qi = mp->dof2q[lane]; qd[lane] = qi >= 0 ? qs[qi] : T(0); }
            WAVE_SYNC();
        }
        if (MODE == 1 && step == 0) {
            if (dbg.pose != nullptr && lane < nb) {
                const double *pw = PD + PDS * lane;
                T *o = dbg.pose + (w * nb + lane) * 16;
                for (int i = 0; i < 3; ++i) {
                    for (int j = 0; j < 3; ++j) o[4 * i + j] = (T)pw[3 * i + j];
                    o[4 * i + 3] = (T)pw[9 + i];
                }
                o[12] = o[13] = o[14] = T(0); o[15] = T(1);
            }
            if (dbg.twist != nullptr && lane < nb)
                for (int i = 0; i < 6; ++i) dbg.twist[(w * nb + lane) * 6 + i] = BD[lane * BDS + BD_TW + i];
        }

        // ---- energies (EnergyMonitor.update, observers.py:40-51): KE = 1/2 sum_b T_b . M_b T_b
        //      (= 1/2 gvel^T M gvel), PE = 9.81 sum_b m_b up . (H_gb c_b); lane = body, wave reduction
        if ((MODE == 0 && logo.energy != nullptr) || (MODE == 1 && dbg.energy != nullptr && step == 0)) {
            double ke = 0., pe = 0.;
            if (lane < nb) {
                const T *bd = BD + lane * BDS;
                const T *Mb = mp->mass + 36 * lane;
                T tw[6], mt[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) tw[i] = bd[BD_TW + i];
                mat6_vec<T>(Mb, tw, mt);
#pragma unroll
                for (int i = 0; i < 6; ++i) ke += 0.5 * (double)tw[i] * (double)mt[i];
                const double *cm = mp->com_d + 4 * lane;
                const M3<double> Rg = ld_m3(PD + PDS * lane); const V3<double> pg = ld_v3(PD + PDS * lane + 9);
                const V3<double> cg = mv(Rg, v3<double>(cm[0], cm[1], cm[2])) + pg;
                pe = 9.81 * cm[3] * (mp->up[0] * cg.x + mp->up[1] * cg.y + mp->up[2] * cg.z);
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) { ke += __shfl_xor(ke, off); pe += __shfl_xor(pe, off); }
            if (lane == 0) {
                T *o = (MODE == 0) ? logo.energy + ((long)step * nworlds + w) * 2 : dbg.energy + w * 2;
                o[0] = (T)ke; o[1] = (T)pe;
            }
        }
