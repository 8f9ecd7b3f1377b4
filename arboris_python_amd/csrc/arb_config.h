// arb_config.h -- compile-time switches of the kernels, the LDS layout of a world, the device-resident model and the
// argument structs of the step kernel (part of libarbstep: included by arb_kernels.hip only).
#ifndef ARB_CONFIG_H
#define ARB_CONFIG_H
#define WAVE 64
// 2nd __launch_bounds__ argument of the step kernels: minimum waves per SIMD = the VGPR budget (512 / waves).
// Round 3: the float32 production kernels with one register column set are compiled for THREE waves per SIMD (168
// VGPRs) now that a human36 world needs 13.1 KB of LDS instead of 19.4 (twelve wavefronts per CU): the kernel is
// latency-bound -- measured with ARB_LDS_PAD: 4 / 6 / 7 / 8 waves per CU give 10.3 / 13.8 / 15.5 / 17.1 M
// world-steps/s -- and a third wave per SIMD pays for the ~500 register spills it costs (none of them in a loop):
// +5.5 % at 4096 worlds, +7.5 % at 65536 (same box, twice).  Two column sets (8 contacts: 26 KB of LDS, six waves
// per CU whatever the register budget) and float64 stay at two; the float64 64-row tile at one (see the kernel).
// Both builds of those kernels are in the library (template parameter CM = 2: three waves) and the host picks per
// launch: a wave of the three-wave build is ~10 % slower (spills), so it only pays when the batch fills the extra
// wave slots -- 1024 worlds without contacts (BASELINE config 2, one wave per world): 25.9 M at two waves, 18.7 M at three.
#ifdef ARB_WAVES_PER_EU
#define ARB_WAVES(CM) ARB_WAVES_PER_EU
#else
#define ARB_WAVES(CM) ((CM) == 2 ? 3 : 2)
#endif
// (the float64 64-row kernels: see arb_step_kernel.h -- one wave per SIMD unless one column set and plain inputs)
// (CM 3, round 6: the MIXED build -- float32 state and LDS, the register tile of phases C / D in float64: the registers of a
//  float64 kernel, hence its rules)
#define ARB_KERNEL_WAVES(T, NMAX, NSETS, MODE, FEAT, CM) (((sizeof(T) == 8 || (CM) == 3) && (NMAX) == 64) ? (((NSETS) == 1 && (MODE) == 0 && (FEAT) <= 1) ? 2 : 1) : ARB_WAVES(CM))
#ifndef GS_SWEEPS
#define GS_SWEEPS 20            // core.py:929-931 (overridable only for timing experiments: the reference's count is 20)
#endif
#ifndef ARB_PHASE_D_MFMA
#define ARB_PHASE_D_MFMA 1      // float32: the constraint-space products J' [Y rhs | Y J'^T] on the matrix cores (0: vector ALU)
#endif
#ifndef ARB_GS_F64
#define ARB_GS_F64 0            // 1: the Gauss-Seidel sweeps of float32 worlds in float64 arithmetic (measured, not the default: DESIGN.md 2)
#endif
#ifndef ARB_ELIM_F64
#define ARB_ELIM_F64 0          // 1 (experiment, round 4): phases C and D of float32 worlds in float64 -- the register tile [Z | rhs | J'^T],
                                // the elimination and the constraint-space products; the sweeps stay float32.  Settles where the float32
                                // outliers are decided (profiles/r04_replay_stats.txt); not a production build (88 more registers)
#endif
#ifndef ARB_ROOT_QM
#define ARB_ROOT_QM 1           // the sliding root finder decides from lane masks (arb_math.h: slide_leftmost_root_qm)
#endif
#ifndef ARB_ELIM_GB
#define ARB_ELIM_GB 8
#endif
#ifndef ARB_ELIM_UNROLL
#define ARB_ELIM_UNROLL 1       // phase C expanded per pivot with structural-zero skipping (tiles <= 48 rows, step kernels); 0: the rolled loop everywhere
#endif
#ifndef ARB_POLY_LANES
#define ARB_POLY_LANES 1        // sliding solve: the quad's lanes evaluate different coefficients of the sextic (see gs_stage); 0: every lane all of them
#endif
#ifndef ARB_GS_FAST
#define ARB_GS_FAST 1           // the sweeps of SoftFingerContact-only worlds run a variant without the rare routes (see gs_stage); 0: one variant
#endif
#ifndef ARB_EIG_WAVE
#define ARB_EIG_WAVE 1          // the generic 6x6 eigenvalue route of the sliding solve runs on the whole wavefront (eig6_wave); 0: one lane on LDS
#endif
// (float32 worlds only: in float64 every scalar of the QR iteration is two SGPRs, and inlined in the sweeps they took the
// float64 kernels from ~30 to ~550 spilled SGPRs -- nine VGPRs of spill lanes, the snake-64 kernel over 256 registers)
#define ARB_EIG_WAVE_FOR(T) (ARB_EIG_WAVE != 0 && sizeof(T) == 4)
#ifndef ARB_WITH_SPEC
#define ARB_WITH_SPEC 1      // the kernels specialised for four plane / sphere SoftFingerContacts (FEAT bit 4): in the shipped library, not in
#endif                       // libarbstep_variants.so (-DARB_WITH_SPEC=0 -DARB_GS_FAST=0), whose general kernels hold them bit-identical
#ifndef ARB_ROWS_SPLIT
#define ARB_ROWS_SPLIT 1
#endif
// s_setprio level of a wave per phase (round 6).  The sweeps have run at 2 since round 3 (ARB_GS_PRIO: +5 %): a wave whose next
// instructions are one dependent chain should issue ahead of a wave with independent work to fill the gaps.  Phase C -- 42 pivots
// in sequence -- at 1: +0.4 % alone, +0.9 % together with 22-row groups (same-process A/B, bit-identical).
#ifndef ARB_C_PRIO
#define ARB_C_PRIO 1
#endif
#ifndef ARB_A_PRIO
#define ARB_A_PRIO 0
#endif
#ifndef ARB_B_PRIO
#define ARB_B_PRIO 0
#endif
#ifndef ARB_D_PRIO
#define ARB_D_PRIO 1            // (phase D, the constraint-space products and the T Y_b T^T passes: +0.25 % on top; A or B at 1: -0.2 %)
#endif
#ifndef ARB_ALVL_PRIO
#define ARB_ALVL_PRIO 0         // (the level loop of phase A alone)
#endif
#ifndef ARB_E_PRIO
#define ARB_E_PRIO 0
#endif
#define ARB_ANY_PRIO (ARB_A_PRIO | ARB_B_PRIO | ARB_C_PRIO | ARB_D_PRIO | ARB_ALVL_PRIO | ARB_E_PRIO)
#ifndef ARB_ELIM_GB_BIG
#define ARB_ELIM_GB_BIG 22      // rows per skippable group of the unrolled elimination on the 44- and 48-row tiles (8 until round 5:
#endif                          // with 22 a pivot tests two or three groups instead of six; the small tiles -- forests -- keep 8)
#ifndef ARB_GS_PRIO
#define ARB_GS_PRIO 2           // s_setprio level of a wave during its Gauss-Seidel sweeps (0: unchanged; 2 measured +5 %, 3 the same)
#endif

// per-body block in LDS (elements).  The fields phase B reads are first, 16-byte aligned, so
// that it can fetch them with 13 vector LDS loads.
#define BD_RCP 0     // R of Ad_cp (9)
#define BD_PCP 9     // p of Ad_cp (3)
#define BD_OM 12     // W_c, then the accumulated pseudo twist Om_b (6), see phase B
#define BD_TW 18     // body twist (6)
#define BD_AB 24     // bias acceleration dJ_b * gvel (6): phase A, until the rhs wrench is formed from it ...
#define BD_PT 24     // ... M_b g_b - M_b (dJ_b gvel) - N_b T_b - B_b T_b (6): rhs of the increment form, in the same slot
// (round 5, LDS bank conflicts: ODD strides -- with 30 elements per body neighbouring bodies' 16-byte accesses overlapped by two
// banks and bodies b, b + 16 met on one; measured with SQ_LDS_BANK_CONFLICT, tools/pmc_lds.sh: 30 -> 31 takes 27 M of the
// 166 M conflict cycles per launch that were left once the rows of Y' were padded, see gs_stage; 34: none; 36: +47 M)
#ifndef BD_STRIDE
#define BD_STRIDE 31         // the step kernels
#endif
#define BD_PG 30     // M_b g_b (6): the inspect kernels only (World._gforce of the controllers alone), hence last
#define BD_STRIDE_INSPECT 37
// (the world pose H_gb of a body lives in PD, in float64, only: a copy in T here cost 12 elements per body -- 6 KB of the
// 43 KB of a float64 snake-64 wavefront, which kept its kernels at three wavefronts per CU instead of four)
// Composite assembly of Z (phase B): per-body accumulators travelling up the tree, in float64:
// A (36) | M upper triangle (21) | wrench of the increment rhs (6) | gravity wrench (6, inspect only)
#define XPR_STRIDE 18     // float64 per dof: X (6) | P = A^T X (6) | R = M X (6)
// float64 per body in the pose table PD: R (9) | p (3) + one of padding (round 5): rows of 12 doubles = 24 banks put bodies b and
// b + 8 on the same banks -- every level of the pose chain reads a parent's pose and its own with 16-byte accesses --: 99 M of
// the remaining 139 M conflict cycles per launch; rows of 13 doubles are read with 8-byte accesses, all 17 bodies of human36
// on disjoint banks (14 would keep the 16-byte accesses and cost the twelfth wavefront per CU: 12 864 B)
#ifndef PDS
#define PDS 13
#endif

// per-constraint block in LDS (elements)
#define CD_R1 0      // transform body1 -> constraint frame: R (9), p (3)
#define CD_P1 9
#define CD_R0 12     // BallAndSocket: transform body0 -> frame0 (9), (3)
#define CD_P0 21
#define CD_SDIST 24
#define CD_ACTIVE 25
#define CD_POS0 26   // (3) BallAndSocket p_01 / JointLimits pos0
#define CD_PINV 32   // (16) inverse of the constraint's admittance block
#define CD_STRIDE 48 // (round 5: the origins of the two contact frames, six more elements, were kept for the inspect kernels'
                     //  c_frame output only: written from phase A' now -- 32 B x nc of every wavefront's LDS)

struct Layout {      // offsets in elements of T inside the wave's LDS block
    int q, dq, qd, bd, pd, sc, cd, rt, am, vv, ff, ff0, work, ci, total;
    int total_inspect;   // ... of the inspect kernels, whose per-body blocks carry six more elements (BD_STRIDE_INSPECT)
    int lscan;       // phase B forms the subtree sums from a prefix table in LDS (small trees) instead of a DPP scan
    int ndol;        // rows of the stacked constraint system (host side: does the model carry constraint forces?)
    // body-space constraint columns (BODYCOL kernels, round 5): behind Y' in the per-body region -- the body-space admittance
    // YB ((6 nbp)^2), the body-space free velocity VB (6 nbp; phase E: the body-space force), the half product W (6 nbp x ndol)
    int yb, vb, wst;
};
// per-constraint integer constants staged in LDS once per launch (int32 words): type, dof masks of the ancestors of
// body 1 and of body 0 (lo, hi each), constrained dof -- the constraint-row loops of phase B read them with
// wave-uniform LDS reads instead of chains of dependent scalar loads from the model
#define CI_STRIDE 7
// float64 per body in the prefix table of phase B.  The 63 (69 inspect) accumulators go through the table in TWO passes
// (A: 36 values, then M | rhs: 27 (33)) so that the table is no larger than the X | P | R vectors that take its place
// afterwards (round 3: 2380 -> 1292 float32 words for human36, one of the three changes that bring the wave's LDS
// from 19.4 KB to 13.1 KB = twelve waves per CU).  304 B rows.
#define TB_STRIDE 38      // two-pass table (the three-wave kernels)
#define TB_STRIDE1 66     // single-pass table (the two-wave step kernels, 63 accumulators: 528 B rows, consecutive bodies 16 B
                          // apart in the banks; 70 until the inspect kernels, which have 69, went over to two passes -- the 68
                          // float64 words less per human36 world are its eighth wavefront per CU in float64)
#define TB_PASS1 36

// exact (bit pattern) equality, also true for identical NaNs
__device__ __forceinline__ bool same_bits(float a, float b) { return __float_as_int(a) == __float_as_int(b); }
__device__ __forceinline__ bool same_bits(double a, double b) { return __double_as_longlong(a) == __double_as_longlong(b); }

// The batch-shared model as ONE device-resident struct with fixed-capacity tables (a world has at most
// 64 bodies / dofs / constraints: one wavefront): every table is reached from the single base pointer with a
// compile-time offset, so the kernels hold one pointer pair in SGPRs instead of ~45 (round 1 spilled 284
// SGPRs to VGPR lanes, most of them table pointers).
#define ARB_CAP 64
#define ARB_MAXPAIR 4     // body-space constraint columns: at most this many (body 0, body 1) pairs, six columns each
template <typename T>
struct DevModel {
    int nb, n, nq, nc, ndol, ncols, maxdepth;
    // Body-space constraint columns (round 5, the BODYCOL kernels): the 4 nc rows of J' of a model whose constraints are all
    // SoftFingerContacts are T_c J_p -- J_p the six rows of the relative Jacobian of the contact's pair of bodies p (world
    // axes, about the origin of `pair_ref`), T_c the contact's 4 x 6 frame transform (constraints.py:429-433: Ad(H_01) of
    // one body Jacobian for every contact of the body) -- so the augmented system carries 6 nbp columns Y J_p^T instead of
    // 4 nc, and human36 with the reference's eight contact points (two feet: 12 instead of 32 columns) fits ONE column set.
    int nbp, ncols_b;
    int pair_ref[ARB_MAXPAIR], cpair[ARB_CAP];
    unsigned long long pair_a1[ARB_MAXPAIR], pair_a0[ARB_MAXPAIR], pair_cmask[ARB_MAXPAIR];
    Layout layb, layb3;      // LDS layouts of the BODYCOL kernels (two-wave / three-wave)
    int has_visc, has_pd, has_warm, has_grav;
    int *status;     // host-visible word (mapped pinned memory) that a launch raises when it gives up waiting in the work queue
    int *warn;       // host-visible warning bits (ARB_WARN_*), raised by the float32 kernels: see the growth check of phase C
    Layout lay;      // LDS offsets of this precision's kernels: re-read per phase instead of held in SGPRs for the whole launch
    Layout lay3;     // ... of the three-wave kernels (two-pass prefix table: a smaller bd region)
    double up[3];
    T grav[3];
    const T *pd_kp, *pd_kd, *pd_tau0;         // [n][n], [n][n], [n] (merged PD controllers; rarely present)
    // forest worlds (arb_model::forest): fk copies of a model with fn dofs, fnq position scalars, fnc constraints;
    // qdef = a valid state of rest (identity poses, zero angles) for retired copies, see the step kernel
    int fk, fn, fnq, fnc;
    const T *qdef;                            // [nq]
    int parent[ARB_CAP], jtype[ARB_CAP], dof_off[ARB_CAP], jnd[ARB_CAP], q_off[ARB_CAP], depth[ARB_CAP], weighted[ARB_CAP];
    int dof2q[ARB_CAP];
    // composite phase B: body of every dof, bodies in the subtree of a body (DFS preorder: the subtree of b
    // is b .. b + subsize[b] - 1), and per dof the dofs of ancestor-or-own / strictly descendant bodies
    int dofbody[ARB_CAP], subsize[ARB_CAP];
    // Several trees below the ground (a ball beside a robot; the copies of a forest): every tree is assembled about the
    // origin of ITS OWN root body and the prefix sums of phase B restart at every root, so that a tree's numbers never
    // see another tree's positions or magnitudes.  root[b] = root body of b's tree, rootmask = bit b: b is a root.
    int root[ARB_CAP];
    unsigned long long rootmask;
    unsigned long long upmask[ARB_CAP], descmask[ARB_CAP];
    unsigned long long anc[ARB_CAP];          // [nb] dofs of the body's joint and of its ancestors'
    int ctype[ARB_CAP], cen[ARB_CAP], cbody[ARB_CAP], cbody0[ARB_CAP], cdof[ARB_CAP], cgeom[ARB_CAP];
    T Hpr[ARB_CAP * 12], Hcn[ARB_CAP * 12], mass[ARB_CAP * 36], visc[ARB_CAP * 36];
    double Hpr_d[ARB_CAP * 12], Hcn_d[ARB_CAP * 12];          // float64 copies for the pose chain
    double clocal_d[ARB_CAP * 3], cradius_d[ARB_CAP], cradius0_d[ARB_CAP], chalf_d[ARB_CAP * 3], cplane_d[ARB_CAP * 4],
           cRz_d[ARB_CAP * 9], cb0_d[ARB_CAP * 12], cb1_d[ARB_CAP * 12];
    double com_d[ARB_CAP * 4];                // [nb][4] centre of mass in the body frame, mass (EnergyMonitor)
    T cmu[ARB_CAP], ceps[ARB_CAP * 3];
    // thresholds compared against positions stay in float64: a float32-rounded joint limit moves by ~1e-7 rad,
    // which the limit solve divides by dt (JointLimits.solve, constraints.py:73-90)
    double cprox_d[ARB_CAP], cmin_d[ARB_CAP], cmax_d[ARB_CAP];
};

// Split execution (opt-in, ARB_STEP_SPLIT_WAVE): the step kernel stops after the
// constraint-space system is built and writes it here; arb_gsw_kernel then runs the
// Gauss-Seidel sweeps (one wavefront per world), and the next step kernel launch starts
// by applying the resulting forces (core.py:975-979).  World-major blocks.
template <typename T>
struct SplitIO {
    int mode;          // 0 fused; bit 0: apply the pending update first; bit 1: produce a system and stop
    T *A;              // [nw][ndol][ndol]   Y' = J' Y J'^T
    T *v;              // [nw][ndol]         J' Y (M gvel/dt + gforce)
    T *f;              // [nw][ndol]         constraint forces (in: warm start, out: after the sweeps)
    T *f0;             // [nw][ndol]         forces already contained in v (warm start)
    T *c;              // [nw][nc][8]        active, sdist, pos0[3]
    T *sol;            // [nw][1+ndol][ndof] columns of [Y rhs | Y J'^T]
};

// Optional per-world PD inputs of arb_step_ex (all [nworlds][ndof], null = absent): desired
// positions/velocities, and diagonal gains that replace the model's gain matrices.
// (ABI 8) zimp [nworlds][ndof][ndof], null = absent: the summed impedance Z_a of the world's user-defined Controllers this step
// (core.py:815-817: `self._impedance -= impedance`), their generalized force travels in ext_gforce
template <typename T>
struct PerWorldPD { const T *qdes, *dqdes, *kp, *kd, *zimp; };

// Running cost of a rollout (arb_step_cost, ABI 7): a diagonal quadratic form of (q, dq, tau) per step, summed on chip
template <typename T>
struct CostIO { T *out; const T *wq, *wdq, *wtau, *qref; };

// Optional per-step logs of arb_rollout (state and energies as observers see them: before the step)
template <typename T>
struct LogOut {
    T *q, *dq, *energy;   // [nsteps][nw][nq], [nsteps][nw][ndof], [nsteps][nw][2]
};

template <typename T>
struct DebugOut {
    T *pose, *twist, *jac, *djac, *Zout, *gforce0, *vel_free, *c_sdist;
    int *c_active;
    T *c_jac, *c_force, *c_frame, *gforce, *q_next, *dq_next;
    T *energy;          // [nw][2] kinetic, potential energy (EnergyMonitor, observers.py:40-51)
    long long *stamps;  // [nw][8] s_memtime at the phase boundaries (diagnostic)
    int ablate;         // diagnostic (env ARB_ABLATE, inspect only): bit 3 (8) = run all 20 Gauss-Seidel sweeps, no fixed-point exit
    int *gs_stats;      // [nw][5]: release, static, sliding (fast shift), sliding (eig6 fallback) solve counts, sweeps
    T *c_adm, *c_vel;   // [nw][ndol][ndol], [nw][ndol]: the constraint-space system Y' = J' Y J'^T, v' the sweeps start from
    T *pivot_growth;    // [nw]: max_j |Z_jj| / |pivot_j| of the elimination (see ARB_WARN_ILLCOND)
    int *gs_trace;      // [nw][GS_SWEEPS][nc]: decision of every solve (0 release, 1 static, 2 sliding fast shift,
                        // 3 sliding eig6, 4 other constraint types); entries of solves not executed are left alone
};
#endif  // ARB_CONFIG_H
