// arb_phase_a2.h -- FRAGMENT of arb_step_kernel (arb_step_kernel.h), included inside its step loop: phase A', lane = constraint -- narrow phase, contact frames, activity.
// Not a header of its own: it reads and writes the kernel's locals (LDS pointers, the register tile, the laundered sizes).
        // ================= phase A': lane = constraint =====================
        ARB_OPAQUE_LANE();
        ARB_ASTAMP(6);
        ARB_STAMP(1);
        // (RT -- the rhs and the rows of J' -- is zeroed in phase B, once the joints' own columns SC, which share its
        // space since round 3, have been consumed)
        // (lane-dense, see ARB_DENSE: with four contacts four lanes would work; every lane runs the arithmetic -- lanes
        // without a constraint on constraint 0 --, the constraints' own lanes store)
        if (do_constraints && (ARB_DENSE_AP || lane < nc)) {
            const bool mine = lane < nc;
            const int c = mine ? lane : 0;
            T *cd = CD + c * CD_STRIDE;
            const int ct = SPEC ? (int)ARB_CT_SOFTFINGER : mp->ctype[c];
            bool active = false;
            T sd = T(0);
            if (SPEC || mp->cen[c]) {
                if (ct == ARB_CT_SOFTFINGER) {
                    // Narrow phase in float64: the gap is a difference of O(1) positions.
                    const int b0 = mp->cbody0[c], b1 = mp->cbody[c];
                    M3<double> Rg0 = m3_identity<double>(), Rg1 = Rg0;
                    V3<double> pg0 = v3<double>(0., 0., 0.), pg1 = pg0;
                    V3<T> bw0 = v3<T>(T(0), T(0), T(0)), bv0 = bw0, bw1 = bw0, bv1 = bw0;
                    if (b0 >= 0) {
                        Rg0 = ld_m3(PD + PDS * b0); pg0 = ld_v3(PD + PDS * b0 + 9);
                        bw0 = ld_v3(BD + b0 * BDS + BD_TW); bv0 = ld_v3(BD + b0 * BDS + BD_TW + 3);
                    }
                    if (b1 >= 0) {
                        Rg1 = ld_m3(PD + PDS * b1); pg1 = ld_v3(PD + PDS * b1 + 9);
                        bw1 = ld_v3(BD + b1 * BDS + BD_TW); bv1 = ld_v3(BD + b1 * BDS + BD_TW + 3);
                    }
                    // pose of shape 0's frame and centre of shape 1 (a Sphere or a Point)
                    const M3<double> Rs0 = mul(Rg0, ld_m3(mp->cb0_d + 12 * c));
                    const V3<double> ps0 = mv(Rg0, ld_v3(mp->cb0_d + 12 * c + 9)) + pg0;
                    const V3<double> p_g1 = mv(Rg1, ld_v3(mp->clocal_d + 3 * c)) + pg1;
                    const double rad = mp->cradius_d[c];
                    const int geom = SPEC ? (int)ARB_CG_PLANE_SPHERE : mp->cgeom[c];
                    V3<double> gc0, gc1;
                    M3<double> Rc;
                    const double sd_d = narrow_phase(geom, Rs0, ps0, p_g1, rad, mp->cradius0_d[c], ld_v3(mp->chalf_d + 3 * c),
                                                     ld_v3(mp->cplane_d + 4 * c), mp->cplane_d[4 * c + 3],
                                                     ld_m3(mp->cRz_d + 9 * c), gc0, gc1, Rc);
                    sd = (T)sd_d;
                    // body k -> contact frame 0: Ad(inv(H_gc0) H_gbk).  With pose0 = H_gc0 this is both
                    // Ad(H_01) Ad(inv(bpose1)) and Ad(inv(bpose0)) of constraints.py:429-433.
                    const M3<T> R1 = cvt_m3<T>(mulTA(Rc, Rg1)), R0 = cvt_m3<T>(mulTA(Rc, Rg0));
                    const V3<T> P1 = cvt_v3<T>(mtv(Rc, pg1 - gc0)), P0 = cvt_v3<T>(mtv(Rc, pg0 - gc0));
                    // gap rate: z velocity of body 1 minus that of body 0 at frame 0   constraints.py:289-291
                    const T vz1 = (mv(R1, bv1) + cross(P1, mv(R1, bw1))).z;
                    const T vz0 = (mv(R0, bv0) + cross(P0, mv(R0, bw0))).z;
                    const T dsd = vz1 - vz0;
                    active = ((double)sd_d + (double)dsd * (double)dt < mp->cprox_d[c]);
                    // phase B works on world-axes columns about the origin of a tree's root body: store world -> contact
                    // frame 0 about the root of body 1's tree (the rows of another tree's dofs shift it, see there)
                    if constexpr (BODYCOL) {
                        // T_c, the contact's rows (w_z, v_x, v_y, v_z) of Ad(c0 <- world axes at o) -- o the origin of the pair's
                        // reference body, a few centimetres from the contact point: [[Rx, 0], [px^ Rx, Rx]] with Rx = Rc^T,
                        // px = -Rx (gc0 - o) -- as 4 x 6 in the block's first 24 slots; ZERO for a contact outside the active set,
                        // whose rows and columns of Y' and entry of v' then come out zero by themselves (core.py:913-918)
                        const V3<double> o = ld_v3(PD + PDS * mp->pair_ref[mp->cpair[c]] + 9);
                        const M3<double> Rx = transpose(Rc);
                        const M3<double> PR = hatmul(-mtv(Rc, gc0 - o), Rx);
                        const double am = active ? 1. : 0.;
                        T t24[24];
#pragma unroll
                        for (int j = 0; j < 3; ++j) {
                            t24[j] = (T)(am * Rx.a[6 + j]); t24[3 + j] = T(0);
#pragma unroll
                            for (int a = 0; a < 3; ++a) { t24[6 * (1 + a) + j] = (T)(am * PR.a[3 * a + j]); t24[6 * (1 + a) + 3 + j] = (T)(am * Rx.a[3 * a + j]); }
                        }
                        if (ARB_DENSE_AP) { for (int i = 0; i < 24; ++i) keep(t24[i]); keep(sd); keep((int)active); }
                        if (mine) {
#pragma unroll
                            for (int i = 0; i < 24; ++i) cd[i] = t24[i];
#pragma unroll
                            for (int i = 0; i < 4; ++i) FF[4 * c + i] = T(0);   // constraints.py:294
                        }
                    } else {
                    const V3<double> p0w = ld_v3(PD + PDS * CI[CI_STRIDE * c + 6] + 9);
                    const M3<T> o_r1 = cvt_m3<T>(transpose(Rc));
                    const V3<T> o_p1 = cvt_v3<T>(-mtv(Rc, gc0 - p0w));
                    if (ARB_DENSE_AP) { keep(o_r1); keep(o_p1); keep(sd); keep((int)active); }
                    if (mine) {
                        st_m3(cd + CD_R1, o_r1); st_v3(cd + CD_P1, o_p1);
#pragma unroll
                        for (int i = 0; i < 4; ++i) FF[4 * c + i] = T(0);   // constraints.py:294
                    }
                    }
                    // inspect: the poses of the two contact frames H_gc0, H_gc1 (constraints.py:284-288), straight from here
                    if (MODE == 1 && mine && step == 0 && dbg.c_frame != nullptr) {
                        for (int f = 0; f < 2; ++f) {
                            T *of = dbg.c_frame + ((w * nc + c) * 2 + f) * 16;
                            const V3<double> gf = f ? gc1 : gc0;
                            for (int i = 0; i < 3; ++i) {
                                for (int j = 0; j < 3; ++j) of[4 * i + j] = (T)Rc.a[3 * i + j];
                                of[4 * i + 3] = (T)(i == 0 ? gf.x : i == 1 ? gf.y : gf.z);
                            }
                            of[12] = of[13] = of[14] = T(0); of[15] = T(1);
                        }
                    }
                } else if (ct == ARB_CT_JOINTLIMITS) {
                    const T p0 = qd[mp->cdof[c]];
                    const double lo_d = mp->cmin_d[c], hi_d = mp->cmax_d[c], px_d = mp->cprox_d[c];
                    active = ((double)p0 - lo_d < px_d) || (hi_d - (double)p0 < px_d);
                    // per-step constants of the solve, formed in float64: (min - pos0)/dt, (max - pos0)/dt
                    const T glo = (T)((lo_d - (double)p0) / (double)dt), ghi = (T)((hi_d - (double)p0) / (double)dt);
                    if (ARB_DENSE_AP) { keep(glo); keep(ghi); keep((int)active); }
                    if (mine) {
                        cd[CD_POS0] = p0;
                        cd[CD_POS0 + 1] = glo;
                        cd[CD_POS0 + 2] = ghi;
#pragma unroll
                        for (int i = 0; i < 4; ++i) FF[4 * c + i] = T(0);   // constraints.py:58-60
                    }
                    sd = p0;
                } else {                                                // BallAndSocket
                    const int b0 = mp->cbody0[c], b1 = mp->cbody[c];
                    M3<double> Rg0 = m3_identity<double>(), Rg1 = Rg0;
                    V3<double> pg0 = v3<double>(0., 0., 0.), pg1 = pg0;
                    if (b0 >= 0) { Rg0 = ld_m3(PD + PDS * b0); pg0 = ld_v3(PD + PDS * b0 + 9); }
                    if (b1 >= 0) { Rg1 = ld_m3(PD + PDS * b1); pg1 = ld_v3(PD + PDS * b1 + 9); }
                    const M3<double> Rf0 = ld_m3(mp->cb0_d + 12 * c);
                    const V3<double> pf0 = ld_v3(mp->cb0_d + 12 * c + 9), pf1 = ld_v3(mp->cb1_d + 12 * c + 9);
                    const M3<double> RP0 = mul(Rg0, Rf0); const V3<double> pP0 = mv(Rg0, pf0) + pg0;
                    const V3<double> pP1 = mv(Rg1, pf1) + pg1;
                    // body1 -> frame 0: Ad(inv(P0) H_gb1);  body0 -> frame 0: Ad(inv(bpose0))
                    const V3<double> p0w = ld_v3(PD + PDS * CI[CI_STRIDE * c + 6] + 9);
                    const V3<T> o_pos = cvt_v3<T>(mtv(RP0, pP1 - pP0));      // p_01  constraints.py:196-197
                    const M3<T> o_r1 = cvt_m3<T>(transpose(RP0));
                    const V3<T> o_p1 = cvt_v3<T>(-mtv(RP0, pP0 - p0w));
                    if (ARB_DENSE_AP) { keep(o_pos); keep(o_r1); keep(o_p1); }
                    if (mine) { st_v3(cd + CD_POS0, o_pos); st_m3(cd + CD_R1, o_r1); st_v3(cd + CD_P1, o_p1); }
                    active = true;
                }
            }
            if (mine) {
                cd[CD_SDIST] = sd;
                cd[CD_ACTIVE] = active ? T(1) : T(0);
            }
        }
        WAVE_SYNC();
        if (lane < ndol) FF0[lane] = FF[lane];
