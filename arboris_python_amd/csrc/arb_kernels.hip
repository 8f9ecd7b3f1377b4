// arb_kernels.hip -- gfx950 (MI355X) kernels of the batched arboris step + C ABI.
//
// One WAVEFRONT (64 lanes, one 64-thread workgroup) advances one world through
//   update_dynamic -> update_controllers -> update_constraints -> integrate
// (arboris/core.py:1356-1363) for `nsteps` steps, keeping the whole state in
// LDS/registers; HBM is touched once to load (q, dq) and once to store them.
//
// Lane roles change from phase to phase:
//   A   lane = body        joint-local kinematics, pose/twist down the tree
//                          (Body.update_dynamic core.py:1158-1315, uniform part)
//   A'  lane = constraint  collision + contact frames (constraints.py:277-294,
//                          collisions.py:161-205), activity test
//   B   lane = body, then  composite assembly of Z = M/dt + B + N in float64: per-body world-frame
//       lane = dof column  blocks, subtree sums by a DPP prefix scan over the body lanes (bodies come
//                          in DFS preorder), one column of Z per lane in registers
//                          (core.py:722-734, 813), rhs of the increment form, constraint rows
//   C   lane = column of the augmented system [Z | rhs | J'^T]: Gauss-Jordan in
//                          registers with v_readlane broadcasts -> Y rhs, Y J'^T
//                          (replaces numpy.linalg.inv, core.py:818, 925-927)
//   D   lane = constraint row: [v | Y'] = J' [Y rhs | Y J'^T], then the 20 Gauss-Seidel
//                          sweeps (core.py:929-935), register resident
//   E   lane = dof         new gvel, joint integration (core.py:974-980)
//
// The library is linked from several translation units of this file (csrc/Makefile): one per
// (register tile NMAX, precision) with the kernels only, and the host unit with the C ABI.
// No CUDA/CPU fallback exists: the library needs a gfx950 device.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <string>
#include <algorithm>
#include <cmath>
#include <type_traits>
#include <utility>
#include <mutex>
#include <cctype>

#include "arbstep.h"
#include "arbstep_hooks.h"
#include "arb_math.h"

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
#define ARB_WAVES(CM) (((CM) == 2 || (CM) == 4) ? 3 : 2)      // (CM = 3, the packed build: two; CM = 4, the rendezvous build: three)
#endif
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
// The rendezvous build (CM = 4, round 4): single steps as work items, the wavefronts of four worlds meet at the
// Gauss-Seidel point -- three park their constraint-space system and what phase E needs in global memory and draw the
// next item, the last to arrive sweeps all four systems (gs_stage_n<T, 4>) and integrates the four worlds.  Bit-identical
// to the other builds; measured -1.2 % at 65 536 worlds, +0.5 % at 16 384, -9 % on config 5, -44 % at 4096 worlds (the
// sliding root finder of four worlds costs what the slowest needs, single-step items cost 2.7 %, 28 KB of parked
// state per world-step): compiled only with -DARB_WITH_RDV=1, selected only with ARB_FORCE_RDV=1 in the environment.
// ARB_ALL_VARIANTS (make variants -> libarbstep_variants.so): the two build variants no launch of the shipped library
// selects -- packed pairs (CM = 3, ARB_FORCE_PACK=1) and the rendezvous build -- compiled for the tests that hold them
// bit-identical to the shipped builds (tests/test_gpu_round3.py); the default library carries neither.
#ifndef ARB_ALL_VARIANTS
#define ARB_ALL_VARIANTS 0
#endif
#ifndef ARB_WITH_RDV
#define ARB_WITH_RDV ARB_ALL_VARIANTS
#endif
#ifndef ARB_WITH_SPEC
#define ARB_WITH_SPEC (!ARB_ALL_VARIANTS)      // the kernels specialised for four plane / sphere SoftFingerContacts (FEAT bit 4): in the shipped library, not in
#endif                                         // libarbstep_variants.so, whose general kernels hold them bit-identical (ARB_FORCE_SPEC=0 in the environment: development)
#ifndef ARB_RDV_DEFAULT
#define ARB_RDV_DEFAULT 0
#endif
#ifndef ARB_PACK_MIN_ROUNDS
#define ARB_PACK_MIN_ROUNDS 4   // the packed build is picked from this many pairs of worlds per wave slot on (16384 worlds on an MI355X: measured +2 %; +0..2 % at 8192, -8 % at 4096, where the three-wave build wins)
#endif
#ifndef ARB_ROWS_SPLIT
#define ARB_ROWS_SPLIT 1
#endif
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
    // packed build (two worlds per wavefront, CM = 3): world A's state and the results of its phases A-D wait here while
    // world B goes through the same phases in the arrays above; `sb*`: world B's state while world A is in those arrays
    int sa_q, sa_dq, sa_am, sa_cd, sa_vv, sa_ff, sa_ff0, sa_rt, sb_q, sb_dq, sb_ff;
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
    Layout layp;     // ... of the packed kernels (two worlds per wavefront: the two-pass layout + the stash)
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
template <typename T>
struct PerWorldPD { const T *qdes, *dqdes, *kp, *kd; };

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
__device__ __forceinline__ void gs_stage(const DevModel<T> *mp, const int lane, const int nc, const int ndol, const int lda, const T dt_t,
                                         const T inv_dt_t, const T *AM, T *CD, T *VV, T *FF, T *WORK,
                                         const DebugOut<T> &dbg, const long w) {
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

// ===========================================================================
// The Gauss-Seidel stage for NG = 2 or 4 worlds held by one wavefront (round 3: two; round 4: four): world g on lanes
// GL g .. GL g + ndol - 1 with GL = 64 / NG (ndol <= GL: up to eight SoftFingerContacts for two worlds, four for four, all
// with eps = (1,1,1), which the host checks).  The sweeps are one dependent instruction chain in which one quad of lanes
// does useful work; here the quad of constraint c of EVERY world works at once.  A stage of the local solve is executed
// when some world needs it and its results are taken lane by lane under the conditions of gs_stage, whose arithmetic every
// lane repeats operation for operation: a world's forces and velocities are bit-identical to gs_stage's.  A world that has
// reached its bit-exact fixed point (or a group without a world: g >= nvalid) takes no further part -- its lanes keep
// their values, as gs_stage's break would.
// LDS per world g: AMp[g] = Y', CDp[g], VVp[g], FFp[g] as for gs_stage; WORK is shared scratch.
// ===========================================================================
template <typename T, int NG>
__device__ __forceinline__ void gs_stage_n(const DevModel<T> *mp, const int lane, const int nc, const int ndol, const T dt,
                                           const T *const (&AMp)[NG], T *const (&CDp)[NG], T *const (&VVp)[NG], T *const (&FFp)[NG],
                                           T *WORK, const int nvalid) {
    static_assert(NG == 2 || NG == 4, "two or four worlds per wavefront");
    constexpr int GL = WAVE / NG;
    const int grp = lane / GL, hl = lane % GL;
    const T *AM = AMp[0];
    T *CD = CDp[0], *VV = VVp[0], *FF = FFp[0];
#pragma unroll
    for (int g = 1; g < NG; ++g) {
        AM = (grp == g) ? AMp[g] : AM; CD = (grp == g) ? CDp[g] : CD; VV = (grp == g) ? VVp[g] : VV; FF = (grp == g) ? FFp[g] : FF;
    }
    const bool mine = grp < nvalid;                       // this lane's world exists
    {
        bool deficient = false;
        if (mine && hl < nc && CD[hl * CD_STRIDE + CD_ACTIVE] != T(0)) {
            const int c = hl;
            T P[16];
            deficient = !inv_block<T>(AM + (4 * c) * ndol + 4 * c, ndol, 4, P);
#pragma unroll
            for (int i = 0; i < 16; ++i) CD[c * CD_STRIDE + CD_PINV + i] = P[i];
        }
        unsigned long long todo = __ballot(deficient);
        while (todo != 0ull) {                           // wave-uniform, rare
            const int L = __builtin_ctzll(todo);
            todo &= todo - 1ull;
            if (lane == L) {
                const int c = hl;
                T P[16];
                pinv_block<T>(AM + (4 * c) * ndol + 4 * c, ndol, 4, P);
                for (int i = 0; i < 16; ++i) CD[c * CD_STRIDE + CD_PINV + i] = P[i];
            }
        }
    }
    WAVE_SYNC();
    T vr = T(0), fr = T(0), Yrow[4], Prow[4];
    bool k_act = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) { Yrow[i] = T(0); Prow[i] = T(0); }
    T q_sd = T(0), q_sdt = T(0), q_mu = T(0);
    T q_iyn = T(0), q_muyn = T(0), q_yc0 = T(0), q_yc1 = T(0), q_yc2 = T(0), q_bsq = T(0);
    SlidePre q_sp = {0., 0., 0., 0., 0., 0.};
    double q_warm = NAN, q_wmove = NAN;
    if (mine && hl < ndol) {
        const int cc = hl >> 2, rr = hl & 3;
        vr = VV[hl]; fr = FF[hl];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            Yrow[i] = AM[hl * ndol + 4 * cc + i];
            Prow[i] = CD[cc * CD_STRIDE + CD_PINV + 4 * rr + i];
        }
        q_sd = CD[cc * CD_STRIDE + CD_SDIST]; q_sdt = q_sd / dt; q_mu = mp->cmu[cc];
        if (CD[cc * CD_STRIDE + CD_ACTIVE] != T(0)) {
            T Yc4[16];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) Yc4[4 * i + j] = AM[(4 * cc + i) * ndol + 4 * cc + j];
            q_sp = slide_precompute<T>(Yc4);
            q_iyn = T(1) / Yc4[15]; q_muyn = q_mu / Yc4[15];
            q_yc0 = Yc4[3]; q_yc1 = Yc4[7]; q_yc2 = Yc4[11];
            const T bq0 = q_muyn * q_yc0, bq1 = q_muyn * q_yc1, bq2 = q_muyn * q_yc2;
            q_bsq = bq0 * bq0 + bq1 * bq1 + bq2 * bq2;
        }
    }
    if (mine && hl < nc) k_act = CD[hl * CD_STRIDE + CD_ACTIVE] != T(0);
    const unsigned long long actmask = __ballot(k_act);          // bit GL g + c: constraint c of world g
    constexpr unsigned ALLG = (1u << NG) - 1u;
    constexpr unsigned long long GMASK = (GL >= 64) ? ~0ull : ((1ull << GL) - 1ull);
    unsigned done = (ALLG << nvalid) & ALLG;                      // groups without a world take no part
    T vr_prev = vr, fr_prev = fr;
#if ARB_GS_PRIO
    __builtin_amdgcn_s_setprio(ARB_GS_PRIO);
#endif
    for (int sweep = 0; sweep < GS_SWEEPS; ++sweep) {
        if (done == ALLG) break;
        for (int c = 0; c < nc; ++c) {
            // the worlds that take part in this solve: bit GL g + 4 c (the leading lane of the quad of c in group g)
            const int base = 4 * c;
            unsigned long long onl = 0ull;
#pragma unroll
            for (int g = 0; g < NG; ++g)
                if (!((done >> g) & 1u) && ((actmask >> (GL * g + c)) & 1ull)) onl |= 1ull << (GL * g + base);
            if (onl == 0ull) continue;
            const int myq = GL * grp + base;                     // the leading lane of this lane's group's quad of c
            const bool my_on = (onl >> myq) & 1ull;              // this lane's world takes part in this solve
            const int rq = hl - base;
            const bool inquad = rq >= 0 && rq < 4;
            T a4[4] = {T(0), T(0), T(0), T(0)};
            if (mine && hl < ndol) {
#pragma unroll
                for (int i = 0; i < 4; ++i) a4[i] = AM[hl * ndol + base + i];
            }
            T df[4], fnew[4];
            const T fq0 = quad_bcast<0>(fr), fq1 = quad_bcast<1>(fr), fq2 = quad_bcast<2>(fr), fq3 = quad_bcast<3>(fr);
            const T v0r = vr - (Yrow[0] * fq0 + Yrow[1] * fq1 + Yrow[2] * fq2 + Yrow[3] * fq3);
            const T vq0 = quad_bcast<0>(vr), vq1 = quad_bcast<1>(vr), vq2 = quad_bcast<2>(vr), vq3 = quad_bcast<3>(vr);
            const T dfr = -(Prow[0] * vq0 + Prow[1] * vq1 + Prow[2] * vq2 + Prow[3] * (vq3 + q_sdt));
            const T fnr = fr + dfr;
            const T v0n = quad_bcast<3>(v0r);
            const T fn0 = quad_bcast<0>(fnr), fn1 = quad_bcast<1>(fnr), fn2 = quad_bcast<2>(fnr), fn3 = quad_bcast<3>(fnr);
            const T lhs = fn0 * fn0 + fn1 * fn1 + fn2 * fn2;
            const T rhs = (fn3 * q_mu) * (fn3 * q_mu);
            // the verdicts of the quads of constraint c
            const unsigned long long relb = __ballot(q_sd + dt * v0n > T(0)), statb = __ballot(lhs <= rhs);
            const bool release = (relb >> myq) & 1ull, stat = (statb >> myq) & 1ull;
            const unsigned long long slm = onl & ~relb & ~statb;                 // leading lanes of the quads that slide
            const bool my_slide = (slm >> myq) & 1ull;
            (void)stat;
            T dfl = release ? -fr : dfr, fnl = release ? T(0) : fnr;             // release / static, row by row
            if (slm != 0ull) {                                                   // sliding friction: some world
                const auto anyq = [&](bool b) { return (bool)((__ballot(b) & slm) != 0ull); };
                const bool want = my_slide && inquad;
                T alpha[4], shift = T(0);
                alpha[0] = quad_bcast<0>(v0r); alpha[1] = quad_bcast<1>(v0r); alpha[2] = quad_bcast<2>(v0r);
                alpha[3] = v0n + q_sdt;
                T Y[16];
                {
                    typedef T Y4 __attribute__((ext_vector_type(4)));
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const Y4 y4 = *reinterpret_cast<const Y4 *>(AM + (base + r) * ndol + base);
                        Y[4 * r] = y4.x; Y[4 * r + 1] = y4.y; Y[4 * r + 2] = y4.z; Y[4 * r + 3] = y4.w;
                    }
                }
                double warm = q_warm;
                bool have = false;
                double c1 = 0., kappa = 0.;
                {
                    const T yc[3] = {q_yc0, q_yc1, q_yc2};
                    const T bq[3] = {q_muyn * yc[0], q_muyn * yc[1], q_muyn * yc[2]};
                    double root;
                    slide_c1_kappa<T>(alpha, yc, q_iyn, q_muyn, bq, q_bsq, &c1, &kappa);
                    double woff = -1.;
                    if (sizeof(T) == 4 && q_wmove == q_wmove)
                        woff = fmin(fmax(2. * q_wmove, 1e-9 * fabs(warm)), 0.1 * fabs(warm));
                    bool ok;
                    slide_leftmost_root_pk(q_sp, c1, kappa, warm, slide_step_tol<T>(), woff, want, anyq, &root, &ok);
                    if (ok) {
                        if (want) q_wmove = fabs(root - warm);
                        warm = root;
                        shift = (root <= 0.) ? (T)(root > -1e10 ? root : -1e10) : T(-1e10);
                        have = true;
                    }
                }
                // rare: generic 6x6 eigenvalues (QR) on the LDS work array, one world after the other
                const unsigned long long needfb = __ballot(want && !have) & slm;
                if (needfb != 0ull) {
                    const T eps[3] = {T(1), T(1), T(1)};
                    for (int h = 0; h < NG; ++h) {
                        if (!((needfb >> (GL * h + base)) & 1ull)) continue;
                        const bool hq = want && grp == h;
#if ARB_ROOT_CASCADE
                        {   // (as gs_stage: the derivative cascade first, every lane on the sextic of this world's quad)
                            double pq[7], pb[7], croot = 0.;
                            slide_poly(q_sp, c1, kappa, pq);
                            const int src = GL * h + base;
#pragma unroll
                            for (int i = 0; i < 6; ++i) pb[i] = bcast(pq[i], src);
                            pb[6] = 1.;
                            const double rbq = bcast(q_sp.nq + 3. * fabs(c1) + arb_fast_sqrt(fabs(kappa)), src);
                            int rc = -1;
                            if (rbq > 0. && rbq < 1e300) rc = slide_real_root_cascade(pb, -1.0001 * rbq - 1e-300, &croot, reinterpret_cast<double *>(WORK));
                            if (rc >= 0) {
                                if (hq) { shift = (rc == 1) ? (T)(croot > -1e10 ? croot : -1e10) : T(-1e10); warm = NAN; q_wmove = NAN; }
                                continue;
                            }
                        }
#endif
                        if (hq) softfinger_sliding_shift<T>(Y, alpha, q_mu, eps, WORK, &shift, false);
                        WAVE_SYNC();
                        T sh_fb;
                        if constexpr (ARB_EIG_WAVE_FOR(T)) {
                            sh_fb = slide_shift_from_eig_wave<T>(WORK, lane);
                        } else {
                            if (lane == 0) WORK[40] = slide_shift_from_eig<T>(WORK);
                            WAVE_SYNC();
                            sh_fb = WORK[40];
                        }
                        if (hq) { shift = sh_fb; warm = NAN; q_wmove = NAN; }
                        WAVE_SYNC();
                    }
                }
                if (want) q_warm = warm;                     // next sweep restarts next to this root
                fnew[0] = fq0; fnew[1] = fq1; fnew[2] = fq2; fnew[3] = fq3;
                const T sie2[3] = {shift, shift, shift};
                softfinger_slide_finish_pk<T>(Y, alpha, sie2, fnew, df, want, anyq);
                const T dfs = (rq == 0) ? df[0] : (rq == 1) ? df[1] : (rq == 2) ? df[2] : df[3];
                const T fns = (rq == 0) ? fnew[0] : (rq == 1) ? fnew[1] : (rq == 2) ? fnew[2] : fnew[3];
                dfl = my_slide ? dfs : dfl; fnl = my_slide ? fns : fnl;
            }
            // the force increments of this lane's world: from the quad of c in its own group
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                T d = bcast(dfl, base + i);
#pragma unroll
                for (int g = 1; g < NG; ++g) { const T dg = bcast(dfl, GL * g + base + i); d = (grp == g) ? dg : d; }
                df[i] = d;
            }
            if (my_on) {
                fr = inquad ? fnl : fr;
                vr += a4[0] * df[0] + a4[1] * df[1] + a4[2] * df[2] + a4[3] * df[3];      // core.py:935
            }
        }
        // a sweep that leaves every velocity and force of a world bit for bit unchanged is its fixed point
        const unsigned long long sameb = __ballot(same_bits(vr, vr_prev) && same_bits(fr, fr_prev));
#pragma unroll
        for (int g = 0; g < NG; ++g)
            if (((sameb >> (GL * g)) & GMASK) == GMASK) done |= 1u << g;
        vr_prev = vr; fr_prev = fr;
    }
#if ARB_GS_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    WAVE_SYNC();
    if (mine && hl < ndol) { FF[hl] = fr; VV[hl] = vr; }
    WAVE_SYNC();
}

// two worlds per wavefront (the packed build of round 3 and its sweep kernel)
template <typename T>
__device__ __forceinline__ void gs_stage2(const DevModel<T> *mp, const int lane, const int nc, const int ndol, const T dt,
                                          const T *AM0, T *CD0, T *VV0, T *FF0w, const T *AM1, T *CD1, T *VV1, T *FF1w,
                                          T *WORK, const bool two) {
    const T *const AMp[2] = {AM0, AM1};
    T *const CDp[2] = {CD0, CD1}, *const VVp[2] = {VV0, VV1}, *const FFp[2] = {FF0w, FF1w};
    gs_stage_n<T, 2>(mp, lane, nc, ndol, dt, AMp, CDp, VVp, FFp, WORK, two ? 2 : 1);
}

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
// CM 3 = the PACKED build (round 3): a wavefront owns TWO worlds (2 p, 2 p + 1).  Phases A-D run for world A, whose
// constraint-space system, solution columns and state then wait in a stash, then for world B; the Gauss-Seidel sweeps
// -- one dependent chain in which one quad of lanes works -- run for both worlds at once (gs_stage2); phase E follows
// for each.  Same arithmetic world by world: bit-identical to the one-world kernels.  Float32, one column set, models
// whose constraints are all SoftFingerContacts with eps = (1,1,1) and fit half a wavefront (nc <= 8), FEAT <= 1.
// (float64 worlds on the 64-row tile -- snake-64 -- need 36 KB of LDS per wave: four waves per CU, one per SIMD, so
// their kernels may take the whole 512-entry register file of a SIMD instead of spilling at 256)
template <typename T, int NMAX, int NSETS, int MODE, int FEAT, int CM>
// (float64 / 64 rows: two column sets need the whole register file of a SIMD; one column set fits 256 registers and must
// stay there -- two wavefronts per SIMD, five per CU with snake-64's LDS -- whatever else is compiled into the kernel: at
// 258 registers config 4 ran at 12.2 instead of 13.1 M world-steps/s)
__global__ __launch_bounds__(WAVE, (sizeof(T) == 8 && NMAX == 64) ? ((NSETS == 1 && MODE == 0 && FEAT <= 1) ? 2 : 1) : ARB_WAVES(CM)) void arb_step_kernel(
    const DevModel<T> *__restrict__ mp_in, const Layout L, T *__restrict__ gq_in, T *__restrict__ gdq_in,
    T *__restrict__ gcforce_in, const T *__restrict__ gext_in, const PerWorldPD<T> pwd_in, long nworlds, T dt_in, int nsteps,
    unsigned flags_in, const DebugOut<T> dbg, int zmode, const LogOut<T> logo_in, const SplitIO<T> sio_in,
    const double *__restrict__ dts_in, int *__restrict__ queue_in, int queue_chunk, int queue_tail, int queue_spin_cap,
    T *__restrict__ park_in, const long ext_stride_in, const long pd_stride_in, const CostIO<T> cost_in)
{
    static_assert(MODE == 0 || FEAT == 3 || FEAT == 19, "the inspect kernels take every input");
    static_assert(CM != 1 || (FEAT == 3 && MODE == 0 && std::is_same<T, float>::value), "matrix-core elimination: float32 step kernels");
    static_assert(CM != 2 || (MODE == 0 && NSETS == 1 && NMAX <= 48 && std::is_same<T, float>::value), "three-wave build: float32, one column set");
    static_assert(CM != 3 || (MODE == 0 && NSETS == 1 && NMAX <= 48 && FEAT <= 1 && std::is_same<T, float>::value), "packed build: float32, one column set, plain inputs");
    static_assert(CM != 4 || (MODE == 0 && NSETS == 1 && NMAX >= 44 && NMAX <= 48 && FEAT <= 1 && std::is_same<T, float>::value), "rendezvous build: float32, one column set, plain inputs");
    constexpr bool PACK = (CM == 3);
    // CM 4 = the RENDEZVOUS build (round 4): one world per wavefront through phases A-D, FOUR worlds per wavefront in the
    // Gauss-Seidel sweeps.  Work items are (step, world); a wavefront that has built its world's constraint-space system
    // parks it (Y', v', forces, the solution columns: ~4 KB) in global memory and counts itself in at its group of four
    // worlds; the wavefront that arrives LAST fetches the three parked systems, runs the sweeps of all four worlds at once
    // (gs_stage_n<T, 4>: bit-identical to gs_stage), finishes the step of each (phase E from the parked solution columns),
    // and publishes the four worlds.  Nobody waits: the other three wavefronts have drawn their next items long before.
    constexpr bool RDV = (CM == 4);
    T *const park = RDV ? park_in : nullptr;
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
    constexpr bool NC_CONST = ((FEAT & 12) != 0 && !BODYCOL) || (BODYCOL && ARB_BC_NC > 0 && MODE == 0);       // nc, ndol compile-time constants
    constexpr int SPEC_NC = BODYCOL ? ARB_BC_NC : (FEAT & 8) ? 0 : 4 * NSETS;
    static_assert(!NC_CONST || (!FEAT_ALL && MODE == 0 && (CM == 0 || CM == 2)), "specialised kernels: plain inputs / user torques");
    static_assert(!BODYCOL || (NSETS == 1 && (CM == 0 || CM == 2) && (FEAT == 20 || FEAT == 21 || FEAT == 19)),
                  "body-space columns: one column set; plain inputs (20), user torques (21), every optional input / inspect (19)");
    static_assert((FEAT & 12) != 12 && (!(FEAT & 8) || NSETS == 1), "specialised kernels: one model class at a time");
    const T *__restrict__ gext = FEAT_EXT ? gext_in : nullptr;
    // ABI 7: control inputs that change along the horizon -- step t reads row t of [nsteps][nworlds][ndof] arrays (stride 0:
    // one row for the whole launch) -- and the running cost of the rollout; both travel with the user torques (FEAT bit 0),
    // the per-step PD targets with the other optional inputs (bit 1)
    const long ext_stride = FEAT_EXT ? ext_stride_in : 0l, pd_stride = FEAT_ALL ? pd_stride_in : 0l;
    const CostIO<T> cost = (FEAT_EXT && CM != 3 && CM != 4) ? cost_in : CostIO<T>{nullptr, nullptr, nullptr, nullptr, nullptr};
    const PerWorldPD<T> pwd = FEAT_ALL ? pwd_in : PerWorldPD<T>{nullptr, nullptr, nullptr, nullptr};
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
    // (reproducer: tools/experiments/f64_64_item_loop_repro.sh builds with -DARB_QUEUE_LOOP_ALL=1, which puts the loop back)
#ifndef ARB_QUEUE_LOOP
#define ARB_QUEUE_LOOP 1
#endif
#ifndef ARB_QUEUE_LOOP_ALL
#define ARB_QUEUE_LOOP_ALL 0
#endif
    constexpr bool QUEUE_LOOP = ARB_QUEUE_LOOP && (ARB_QUEUE_LOOP_ALL || !(sizeof(T) == 8 && NMAX == 64));
    int *const queue = (MODE == 0) ? queue_in : nullptr;
    T *gq = gq_in, *gdq = gdq_in, *gcforce = gcforce_in;
    long w = blockIdx.x;
    const long nunits = (CM == 3) ? (nworlds + 1) / 2 : nworlds;      // work units: worlds, or pairs of worlds
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
        if ((long)item >= nunits * (long)nchunks) return;
        w = item % (int)nunits;
        qitem_chunk = item / (int)nunits;
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
                    if constexpr (RDV) {        // (its group can never complete this step's rendezvous: nobody of it goes on)
                        const long g0 = (w / 4) * 4;
                        for (long wh = g0; wh < g0 + 4 && wh < nunits; ++wh)
                            (void)__hip_atomic_fetch_max(queue + 1 + wh, POISON, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                if (!QUEUE_LOOP) return;
                continue;
            }
            asm volatile("" ::: "memory");      // (order only: the coherent loads of the state are issued after the flag was seen)
        }
        // (the state pointers are `restrict` kernel arguments: hand the compiler pointers it knows nothing about, so
        // that no load of the state is scheduled above the acquire)
        asm volatile("" : "+s"(gq), "+s"(gdq), "+s"(gcforce));
    } else if (w >= nunits) {
        return;
    }
    // (packed build: `w` is the pair, its worlds are w0 = 2 w and w0 + 1; otherwise w0 = w)
    const long w0 = PACK ? 2 * w : w;
    const bool two = PACK && (w0 + 1 < nworlds);
    T *lds = reinterpret_cast<T *>(arb_lds_raw);
    T *qs, *dqs, *qd, *BD, *SC, *CD, *RT, *AM, *VV, *FF, *FF0, *WORK;
    double *PD;
    int *CI;
// (after the first global store the compiler no longer proves the model unclobbered and fetches it with vector
// loads: readfirstlane puts the wave-uniform values back into SGPRs)
#define ARB_UNI(x) __builtin_amdgcn_readfirstlane(x)
#define ARB_LAY() ((CM == 3) ? mp->layp : (CM == 2 || CM == 4) ? (BODYCOL ? mp->layb3 : mp->lay3) : (BODYCOL ? mp->layb : mp->lay))
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
    // row stride of Y' in LDS: four elements of padding (bank conflicts of the sweeps' column reads, see gs_stage); the packed and
    // rendezvous builds (libarbstep_variants.so) keep the dense rows their stash copies assume
#define lda ((PACK || RDV) ? ndol : ndol + 4)
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
    // (packed build: world A = w0 into the stash, world B = w0 + 1 into the working arrays)
    T *SAq = nullptr, *SAdq = nullptr, *SAff = nullptr;
    if constexpr (PACK) { const Layout &lp = mp->layp; SAq = lds + ARB_UNI(lp.sa_q); SAdq = lds + ARB_UNI(lp.sa_dq); SAff = lds + ARB_UNI(lp.sa_ff); }
    {
        T *q_to = PACK ? SAq : qs, *dq_to = PACK ? SAdq : dqs, *ff_to = PACK ? SAff : FF;
        for (int i = lane; i < nq; i += WAVE) q_to[i] = ldg(gq + w0 * nq + i);
        if (lane < RS) dq_to[lane] = (lane < n) ? ldg(gdq + w0 * n + lane) : T(0);      // (the velocity array has one element per tile row)
        for (int i = lane; i < ndol; i += WAVE) {
            T f = T(0);
            if (gcforce != nullptr) f = ldg(gcforce + w0 * ndol + i);
            ff_to[i] = f;
        }
    }
    if (PACK && two) {
        for (int i = lane; i < nq; i += WAVE) qs[i] = ldg(gq + (w0 + 1) * nq + i);
        if (lane < RS) dqs[lane] = (lane < n) ? ldg(gdq + (w0 + 1) * n + lane) : T(0);
        for (int i = lane; i < ndol; i += WAVE) {
            T f = T(0);
            if (gcforce != nullptr) f = ldg(gcforce + (w0 + 1) * ndol + i);
            FF[i] = f;
        }
    }
    // Forest worlds (several copies of a small model in this wavefront, arb_model::forest): bit j = copy j is RETIRED.
    // The copies share the elimination and the sweeps, where a product "exact zero x NaN" would carry one copy's NaN
    // into all the others; so a copy whose state is not finite -- or beyond +-1e8 (float32) / 1e100, i.e. diverged -- at
    // the beginning of a step computes on a state of rest from then on and has NaN written to its state, forces and
    // logs: what the one-world kernels leave behind for a world that overflowed, without touching its neighbours.
    unsigned dead = 0u;
    const T ext_kA = (gext != nullptr && lane < n) ? gext[w0 * n + lane] : T(0);
    const T ext_kB = (PACK && two && gext != nullptr && lane < n) ? gext[(w0 + 1) * n + lane] : T(0);
    // running cost of the rollout (arb_step_cost): read here, one addition per step in step order, written back with the
    // state -- a horizon cut into work items or launches adds up bit for bit like one launch
    T cost_acc = T(0);
    if constexpr (FEAT_EXT) { if (cost.out != nullptr) cost_acc = ldg(cost.out + w0); }
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
    auto integrate_on = [&](const T *RT, const T *FF, const T *FF0, T *qs, T *dqs, bool with_forces, bool parked = false) {
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
            // (parked: the solution columns wait in global memory, written by another wavefront -- coherent loads)
            vnew = parked ? ldg(RT + lane) : RT[lane];
            if (with_forces)
                for (int i = 0; i < ndol; ++i) vnew += (parked ? ldg(RT + (1 + i) * RS + lane) : RT[(1 + i) * RS + lane]) * (FF[i] - FF0[i]);
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

    bool rdv_done = false;     // (rendezvous build: this item's step has been handed over / finished with its group)
    for (int step = step_lo; step < step_hi; ++step) {
        T gf0 = T(0);          // controllers' generalized force (inspect output)
        T ext_cost = T(0);     // this step's user torque of the lane's dof (the running cost's tau)
        // Packed build: phases A-D for world A (isub 0), then for world B (isub 1), in the same working arrays; each
        // world's state is copied in before (world B's is parked meanwhile), world A's results are stashed after.
        const int nsub = (PACK && two) ? 2 : 1;
        for (int isub = 0; isub < nsub; ++isub) {
        if constexpr (PACK) {
            ARB_OPAQUE_LANE();
            const Layout &lp = mp->layp;
            T *SBq = lds + ARB_UNI(lp.sb_q), *SBdq = lds + ARB_UNI(lp.sb_dq), *SBff = lds + ARB_UNI(lp.sb_ff);
            T *Sq = lds + ARB_UNI(lp.sa_q), *Sdq = lds + ARB_UNI(lp.sa_dq), *Sff = lds + ARB_UNI(lp.sa_ff);
            if (isub == 0 && two) {        // park world B's state
                for (int i = lane; i < nq; i += WAVE) SBq[i] = qs[i];
                if (lane < RS) SBdq[lane] = dqs[lane];
                for (int i = lane; i < ndol; i += WAVE) SBff[i] = FF[i];
            }
            WAVE_SYNC();
            const T *fq = isub == 0 ? Sq : SBq, *fdq = isub == 0 ? Sdq : SBdq, *fff = isub == 0 ? Sff : SBff;
            for (int i = lane; i < nq; i += WAVE) qs[i] = fq[i];
            if (lane < RS) dqs[lane] = fdq[lane];
            for (int i = lane; i < ndol; i += WAVE) FF[i] = fff[i];
            WAVE_SYNC();
        }
        T ext_k = (PACK && isub == 1) ? ext_kB : ext_kA;
        if constexpr (FEAT_EXT) {
            // a torque SEQUENCE (arb_step_args.ext_gforce_steps): this step's row
            if (ext_stride != 0l && gext != nullptr)
                ext_k = (lane0 < ARB_UNI(mp->n)) ? gext[(long)step * ext_stride + (w0 + ((PACK && isub == 1) ? 1 : 0)) * ARB_UNI(mp->n) + lane0] : T(0);
            ext_cost = ext_k;
        }
        // ================= phase A: lane = body ===========================
        ARB_OPAQUE_LANE();
        ARB_STAMP(0);
        if (FEAT_ALL && dts != nullptr) { dt = (T)dts[step]; inv_dt = T(1) / dt; }
        // (forests are built on the 16- and 32-row tiles only: the larger kernels carry none of this)
        const int fk = (PACK || NMAX > 32) ? 1 : ARB_UNI(mp->fk);
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
            if constexpr (sizeof(T) == 8) {
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
                        const V3<double> ww = mv(Rgb, Tnw);
                        st_v3(bdl + BD_TW, ww); st_v3(bdl + BD_TW + 3, cross(pgb, ww) + mv(Rgb, Tnv));
                    }
                    jump_sum(BD_TW);
                    if (on) {
                        const V3<double> ww = ld_v3(bdl + BD_TW), wv = ld_v3(bdl + BD_TW + 3);
                        st_v3(bdl + BD_TW, mtv(Rgb, ww)); st_v3(bdl + BD_TW + 3, mtv(Rgb, wv - cross(pgb, ww)));
                    }
                    WAVE_SYNC();
                    // (3) bias accelerations: dAd_cp T_p + Bn_c in body axes, to world axes, summed, back
                    if (on) {
                        V3<double> tw = v3<double>(0., 0., 0.), tv = tw;
                        if (par >= 0) { const T *pb = BD + par * BDS; tw = ld_v3(pb + BD_TW); tv = ld_v3(pb + BD_TW + 3); }
                        const V3<double> lw = mv(dA_cp, tw) + Bnw;
                        const V3<double> lv = mv(dB_cp, tw) + mv(dA_cp, tv) + Bnv;
                        const V3<double> ww = mv(Rgb, lw);
                        st_v3(bdl + BD_AB, ww); st_v3(bdl + BD_AB + 3, cross(pgb, ww) + mv(Rgb, lv));
                    }
                    jump_sum(BD_AB);
                    if (on) {
                        const V3<double> ww = ld_v3(bdl + BD_AB), wv = ld_v3(bdl + BD_AB + 3);
                        st_v3(bdl + BD_AB, mtv(Rgb, ww)); st_v3(bdl + BD_AB + 3, mtv(Rgb, wv - cross(pgb, ww)));
                    }
                    WAVE_SYNC();
                }
            }
            // pose and twist down the tree, one depth level at a time
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

        // ================= phase B: lane = dof column =======================
        ARB_OPAQUE_LANE();
        ARB_STAMP(2);
        // (ZT: the arithmetic type of the register tile -- T, or float64 for float32 worlds in the ARB_ELIM_F64 experiment)
        constexpr bool ELIM64 = (ARB_ELIM_F64 != 0) && std::is_same<T, float>::value && NMAX <= 48 && CM != 1;
        using ZT = std::conditional_t<ELIM64, double, T>;
        ZT Z[NMAX];
        ZT Z2[NSETS == 2 ? NMAX : 1];
#pragma unroll
        for (int i = 0; i < NMAX; ++i) Z[i] = ZT(0);
        T rhsM = T(0), rhsG = T(0);
        // |Z_kk| as assembled (float32 worlds; inspect): the elimination of phase C compares every pivot with it, see there
        constexpr bool TRACK_GROWTH = (sizeof(T) == 4 && MODE == 0) || MODE == 1;
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
                if constexpr (sizeof(T) == 8) {
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
                            const V3<double> ww = mv(Rgb, ld_v3(bdl + BD_OM));
                            const V3<double> wv = cross(pgb2, ww) + mv(Rgb, ld_v3(bdl + BD_OM + 3));
                            st_v3(bdl + BD_OM, ww); st_v3(bdl + BD_OM + 3, wv);
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
                            const V3<double> ww = ld_v3(bdl + BD_OM), wv = ld_v3(bdl + BD_OM + 3);
                            st_v3(bdl + BD_OM, mtv(Rgb, ww)); st_v3(bdl + BD_OM + 3, mtv(Rgb, wv - cross(pgb2, ww)));
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
            // in their two-wave kernels cost 65 spilled VGPRs and 3-6 %, tools/experiments/forest_rate.py)
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
#pragma unroll
                for (int i = 0; i < NMAX; ++i) {
                    // (the wave-uniform branch per row also keeps the rows apart for the scheduler: as one
                    // branch-free block the compiler hoists the LDS reads of all NMAX rows and spills ~1500 VGPRs)
                    if (i < n) {
                        asm volatile("");          // not speculatable: a real scalar branch per row, no if-conversion into lane masks
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
                    } else {
                        Z[i] = ZT(0);
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
        T rhs = rhsM + ext_k;          // gforce - (N + B + Z_pd) gvel
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
        // ================= phase C: augmented Gauss-Jordan ===================
        ARB_OPAQUE_LANE();
        ARB_BSTAMP(7);
        ARB_STAMP(3);
        if (lane < RS) RT[lane] = (lane < n) ? rhs : T(0);
        WAVE_SYNC();
        const int ncols = do_constraints ? (BODYCOL ? mp->ncols_b : mp->ncols) : n + 1;
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
                // (no masking of the sign bits: Z_jj and a healthy pivot are positive; a pivot <= 0 -- a negative integer -- makes the
                // difference huge, and the warning is right to come)
                const int zb = __builtin_amdgcn_readlane(__float_as_int(zdiag), j);
                const int pb = __builtin_amdgcn_readfirstlane(__float_as_int((float)pivv));
                growth_bits = (zb - pb > growth_bits) ? zb - pb : growth_bits;
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
            const bool z_dense = (!SPEC && mp->has_pd) || (FEAT_ALL && pwd.kp != nullptr);
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
                    constexpr int GB = ARB_ELIM_GB;
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
                        const T tr = RT[NMAX - 1] * arb_rcp(WORK[NMAX - 1]);
#pragma unroll
                        for (int r = NMAX - 1; r >= 1; --r) Z[r] = RT[r - 1] - WORK[r - 1] * tr;
                        Z[0] = tr;
                    }
                }
            }
        }
        }
        ARB_CSTAMP(5);
        if constexpr (TRACK_GROWTH) {
            if (sizeof(T) == 4 && MODE == 0) warn_illcond = warn_illcond || (growth_bits > (11 << 23));      // ARB_ILLCOND_GROWTH = 2^11
            if (MODE == 1 && dbg.pivot_growth != nullptr && lane == 0)
                dbg.pivot_growth[w] = (T)__int_as_float((growth_bits > 0 ? growth_bits : 0) + 0x3f800000);
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
                        const int rot = (NMAX <= 32 && !PACK) ? (ARB_UNI(mp->fk) > 1 ? ((c / ARB_UNI(mp->fnc)) * ARB_UNI(mp->fn)) & 3 : 0) : 0;
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
                    const int rot = (NMAX <= 32 && !PACK) ? (ARB_UNI(mp->fk) > 1 ? (((idx >> 2) / ARB_UNI(mp->fnc)) * ARB_UNI(mp->fn)) & 3 : 0) : 0;
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

        if constexpr (PACK) {
            if (isub == 0) {               // stash world A: its system, its solution columns, the forces
                ARB_OPAQUE_LANE();
                const Layout &lp = mp->layp;
                T *Sam = lds + ARB_UNI(lp.sa_am), *Scd = lds + ARB_UNI(lp.sa_cd), *Svv = lds + ARB_UNI(lp.sa_vv);
                T *Sff = lds + ARB_UNI(lp.sa_ff), *Sff0 = lds + ARB_UNI(lp.sa_ff0), *Srt = lds + ARB_UNI(lp.sa_rt);
                for (int i = lane; i < ndol * ndol; i += WAVE) Sam[i] = AM[i];
                for (int i = lane; i < nc * CD_STRIDE; i += WAVE) Scd[i] = CD[i];
                for (int i = lane; i < ndol; i += WAVE) { Svv[i] = VV[i]; Sff[i] = FF[i]; Sff0[i] = FF0[i]; }
                for (int i = lane; i < (1 + ndol) * RS; i += WAVE) Srt[i] = RT[i];
                WAVE_SYNC();
            }
        }
        }   // isub (one pass unless packed)
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

        if constexpr (RDV) {
            ARB_OPAQUE_LANE();
            auto al4 = [](int x) { return (x + 3) & ~3; };
            const int nA = al4(ndol * ndol), nD = al4(ndol), nCDp = al4(2 * nc), nRT = (1 + ndol) * RS;
            const int pVV = nA, pFF = pVV + nD, pFF0 = pFF + nD, pCD = pFF0 + nD, pRT = pCD + nCDp;
            const long PST = pRT + nRT;                                   // parked floats per world
            const int t = step;                                           // (one step per item)
            // ---- park this world's system
            {
                T *pw = park + w * PST;
                for (int i = lane; i < ndol * ndol; i += WAVE) stg(pw + i, AM[i]);
                if (lane < ndol) { stg(pw + pVV + lane, VV[lane]); stg(pw + pFF + lane, FF[lane]); stg(pw + pFF0 + lane, FF0[lane]); }
                if (lane < nc) { stg(pw + pCD + 2 * lane, CD[lane * CD_STRIDE + CD_ACTIVE]); stg(pw + pCD + 2 * lane + 1, CD[lane * CD_STRIDE + CD_SDIST]); }
                for (int i = lane; i < nRT; i += WAVE) stg(pw + pRT + i, RT[i]);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the coherent stores above are performed
            WAVE_SYNC();
            const long g = w / 4, g0 = 4 * g;
            const int nv = (int)((nunits - g0 < 4) ? nunits - g0 : 4);    // worlds of this group
            int old = 0;
            if (lane0 == 0) old = __hip_atomic_fetch_add(queue + 1 + nunits + g, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            old = __builtin_amdgcn_readfirstlane(old);
            rdv_done = true;
            if (old + 1 != nv * (t + 1)) break;                           // not the last of the group: on to the next item
            asm volatile("" ::: "memory");
            // ---- the last of its group: the other worlds' systems into LDS slots (two behind Y' in the per-body region,
            // the third in the space of RT, whose columns are parked), the sweeps of all of them at once
            const int SS = nA + al4(nc * CD_STRIDE) + 3 * nD;
            const int myh = (int)(w - g0);
            const T *AMp[4];
            T *CDp[4], *VVp[4], *FFp[4], *F0p[4];
            {
                int slot = 0;
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    if (h == myh || h >= nv) { AMp[h] = AM; CDp[h] = CD; VVp[h] = VV; FFp[h] = FF; F0p[h] = FF0; continue; }
                    T *b = (slot < 2) ? (AM + nA + slot * SS) : RT;
                    ++slot;
                    AMp[h] = b; CDp[h] = b + nA; VVp[h] = CDp[h] + al4(nc * CD_STRIDE); FFp[h] = VVp[h] + nD; F0p[h] = FFp[h] + nD;
                    const T *ph = park + (g0 + h) * PST;
                    for (int i = lane; i < ndol * ndol; i += WAVE) b[i] = ldg(ph + i);
                    if (lane < ndol) { VVp[h][lane] = ldg(ph + pVV + lane); FFp[h][lane] = ldg(ph + pFF + lane); F0p[h][lane] = ldg(ph + pFF0 + lane); }
                    if (lane < nc) { CDp[h][lane * CD_STRIDE + CD_ACTIVE] = ldg(ph + pCD + 2 * lane); CDp[h][lane * CD_STRIDE + CD_SDIST] = ldg(ph + pCD + 2 * lane + 1); }
                }
            }
            WAVE_SYNC();
            {
                const T *const AMc[4] = {AMp[0], AMp[1], AMp[2], AMp[3]};
                T *const CDc[4] = {CDp[0], CDp[1], CDp[2], CDp[3]}, *const VVc[4] = {VVp[0], VVp[1], VVp[2], VVp[3]},
                  *const FFc[4] = {FFp[0], FFp[1], FFp[2], FFp[3]};
                gs_stage_n<T, 4>(mp, lane, nc, ndol, dt, AMc, CDc, VVc, FFc, WORK, nv);
            }
            // ---- phase E of every world of the group from its parked solution columns (this wavefront's own world first:
            // its state is in LDS), the new states and forces to global memory, the four worlds published
            ARB_OPAQUE_LANE();
            for (int k = 0; k < nv; ++k) {
                const int h = (k == 0) ? myh : (k <= myh ? k - 1 : k);
                const long wh = g0 + h;
                if (h != myh) {
                    for (int i = lane; i < nq; i += WAVE) qs[i] = ldg(gq + wh * nq + i);
                    if (lane < RS) dqs[lane] = (lane < n) ? ldg(gdq + wh * n + lane) : T(0);
                    WAVE_SYNC();
                }
                const T *fh = (h == 0) ? FFp[0] : (h == 1) ? FFp[1] : (h == 2) ? FFp[2] : FFp[3];
                const T *f0h = (h == 0) ? F0p[0] : (h == 1) ? F0p[1] : (h == 2) ? F0p[2] : F0p[3];
                integrate_on(park + wh * PST + pRT, fh, f0h, qs, dqs, true, true);
                for (int i = lane; i < nq; i += WAVE) stg(gq + wh * nq + i, qs[i]);
                if (lane < n) stg(gdq + wh * n + lane, dqs[lane]);
                if (gcforce != nullptr) for (int i = lane; i < ndol; i += WAVE) stg(gcforce + wh * ndol + i, fh[i]);
                WAVE_SYNC();
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            WAVE_SYNC();
            if (lane0 == 0)
                for (int h = 0; h < nv; ++h)
                    (void)__hip_atomic_fetch_max(queue + 1 + g0 + h, t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }

        if (do_constraints) {
            ARB_STAMP(5);
            ARB_CSTAMP(7);
            if constexpr (PACK) {
                const Layout &lp = mp->layp;
                gs_stage2<T>(mp, lane, nc, ndol, dt, lds + ARB_UNI(lp.sa_am), lds + ARB_UNI(lp.sa_cd), lds + ARB_UNI(lp.sa_vv),
                             lds + ARB_UNI(lp.sa_ff), AM, CD, VV, FF, WORK, two);
            } else {
                using GSG = std::conditional_t<(ARB_GS_F64 != 0) && std::is_same<T, float>::value, double, T>;
                gs_stage<T, MODE, GSG, !(sizeof(T) == 8 && NMAX == 64), SPEC>(mp, lane, nc, ndol, lda, dt, inv_dt, AM, CD, VV, FF, WORK, dbg, w);
            }
        }

        // ================= phase E: new velocity, integrate ==================
        ARB_OPAQUE_LANE();
        ARB_STAMP(6);
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
        if constexpr (PACK) {
            const Layout &lp = mp->layp;
            integrate_on(lds + ARB_UNI(lp.sa_rt), lds + ARB_UNI(lp.sa_ff), lds + ARB_UNI(lp.sa_ff0), lds + ARB_UNI(lp.sa_q),
                         lds + ARB_UNI(lp.sa_dq), do_constraints);
            if (two) integrate_from_rt(do_constraints);
        } else {
            integrate_from_rt(do_constraints);
        }
        if constexpr (FEAT_EXT && !PACK && !RDV) {
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

    if (RDV && rdv_done) {         // (states stored and worlds published by the last wavefront of the group)
        if (!QUEUE_LOOP) break;
        continue;
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
        if constexpr (PACK) {
            const Layout &lp = mp->layp;
            const T *Sq = lds + ARB_UNI(lp.sa_q), *Sdq = lds + ARB_UNI(lp.sa_dq), *Sff = lds + ARB_UNI(lp.sa_ff);
            for (int i = lane; i < nq; i += WAVE) stg(gq + w0 * nq + i, Sq[i]);
            if (lane < n) stg(gdq + w0 * n + lane, Sdq[lane]);
            if (gcforce != nullptr) for (int i = lane; i < ndol; i += WAVE) stg(gcforce + w0 * ndol + i, Sff[i]);
        }
        const long wm = PACK ? w0 + 1 : w;         // the world in the working arrays
        if (!PACK || two) {
            for (int i = lane; i < nq; i += WAVE) stg(gq + wm * nq + i, qs[i]);
            if (lane < n) stg(gdq + wm * n + lane, dqs[lane]);
            if (gcforce != nullptr && !(sio.mode & 2))
                for (int i = lane; i < ndol; i += WAVE) stg(gcforce + wm * ndol + i, FF[i]);
        }
        if constexpr (FEAT_EXT) { if (cost.out != nullptr && lane0 == 0) stg(cost.out + w0, cost_acc); }
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

// ===========================================================================
// Gauss-Seidel with one WAVEFRONT per world as its own kernel (split execution, ARB_STEP_SPLIT_WAVE): the
// quad-local sweeps of gs_stage, fed from the SplitIO buffers.  The sweeps are one dependent chain per world
// and need few registers, so this kernel is compiled for several waves per SIMD (the step kernel is pinned
// at two by its 256 VGPRs): other worlds' chains fill the issue slots one chain leaves empty.
// LDS per world: Y' (ndol^2) + the constraint blocks + 64 elements of scratch.
// ===========================================================================
template <typename T, int WV>
__global__ __launch_bounds__(WAVE, WV) void arb_gsw_kernel(
    const DevModel<T> *__restrict__ mp, const T *__restrict__ wsA, const T *__restrict__ wsv,
    T *__restrict__ wsf, const T *__restrict__ wsc, long nworlds, T dt_in, const double *__restrict__ dts)
{
    const int lane = threadIdx.x;
    const long w = blockIdx.x;
    if (w >= nworlds) return;
    const int nc = mp->nc, ndol = mp->ndol;
    const T dt = dts != nullptr ? (T)dts[0] : dt_in;
    const T inv_dt = T(1) / dt;
    T *lds = reinterpret_cast<T *>(arb_lds_raw);
    auto al = [](int x) { return (x + 3) & ~3; };
    T *AM = lds, *CD = AM + al(ndol * ndol), *VV = CD + al(nc * CD_STRIDE), *FF = VV + al(ndol), *WORK = FF + al(ndol);
    const int nA = ndol * ndol;
    for (int i = lane; i < nA; i += WAVE) AM[i] = wsA[w * nA + i];
    if (lane < ndol) { VV[lane] = wsv[w * ndol + lane]; FF[lane] = wsf[w * ndol + lane]; }
    if (lane < nc) {
        const T *cs = wsc + (w * nc + lane) * 8;
        T *cd = CD + lane * CD_STRIDE;
        cd[CD_ACTIVE] = cs[0]; cd[CD_SDIST] = cs[1]; cd[CD_POS0] = cs[2]; cd[CD_POS0 + 1] = cs[3]; cd[CD_POS0 + 2] = cs[4];
    }
    WAVE_SYNC();
    DebugOut<T> nodbg;
    nodbg.gs_stats = nullptr; nodbg.gs_trace = nullptr; nodbg.ablate = 0;
    gs_stage<T, 0>(mp, lane, nc, ndol, ndol, dt, inv_dt, AM, CD, VV, FF, WORK, nodbg, w);
    if (lane < ndol) wsf[w * ndol + lane] = FF[lane];
}

// The same with NG = 2 or 4 worlds per wavefront (gs_stage_n): worlds NG p .. NG p + NG - 1 in workgroup p.  Development /
// test vehicle of the packed sweeps (ARB_GSW_PACK=2|4 in the environment selects it for ARB_STEP_SPLIT_WAVE; 1 means 2):
// bit-identical to arb_gsw_kernel by construction, checked in tests/test_gpu_round3.py.
template <typename T, int WV, int NG>
__global__ __launch_bounds__(WAVE, WV) void arb_gswn_kernel(
    const DevModel<T> *__restrict__ mp, const T *__restrict__ wsA, const T *__restrict__ wsv,
    T *__restrict__ wsf, const T *__restrict__ wsc, long nworlds, T dt_in, const double *__restrict__ dts)
{
    const int lane = threadIdx.x;
    const long w0 = (long)NG * blockIdx.x;
    if (w0 >= nworlds) return;
    const int nvalid = (int)((nworlds - w0 < NG) ? nworlds - w0 : NG);
    const int nc = mp->nc, ndol = mp->ndol;
    const T dt = dts != nullptr ? (T)dts[0] : dt_in;
    T *lds = reinterpret_cast<T *>(arb_lds_raw);
    auto al = [](int x) { return (x + 3) & ~3; };
    const int per = al(ndol * ndol) + al(nc * CD_STRIDE) + 2 * al(ndol);
    const T *AMw[NG];
    T *CDw[NG], *VVw[NG], *FFw[NG];
#pragma unroll
    for (int h = 0; h < NG; ++h) {
        T *b = lds + h * per;
        AMw[h] = b; CDw[h] = b + al(ndol * ndol); VVw[h] = CDw[h] + al(nc * CD_STRIDE); FFw[h] = VVw[h] + al(ndol);
    }
    T *WORK = lds + NG * per;
    const int nA = ndol * ndol;
#pragma unroll
    for (int h = 0; h < NG; ++h) {
        if (h >= nvalid) continue;
        const long w = w0 + h;
        T *am = lds + h * per;
        for (int i = lane; i < nA; i += WAVE) am[i] = wsA[w * nA + i];
        if (lane < ndol) { VVw[h][lane] = wsv[w * ndol + lane]; FFw[h][lane] = wsf[w * ndol + lane]; }
        if (lane < nc) {
            const T *cs = wsc + (w * nc + lane) * 8;
            T *cd = CDw[h] + lane * CD_STRIDE;
            cd[CD_ACTIVE] = cs[0]; cd[CD_SDIST] = cs[1]; cd[CD_POS0] = cs[2]; cd[CD_POS0 + 1] = cs[3]; cd[CD_POS0 + 2] = cs[4];
        }
    }
    WAVE_SYNC();
    gs_stage_n<T, NG>(mp, lane, nc, ndol, dt, AMw, CDw, VVw, FFw, WORK, nvalid);
#pragma unroll
    for (int h = 0; h < NG; ++h)
        if (h < nvalid && lane < ndol) wsf[(w0 + h) * ndol + lane] = FFw[h][lane];
}

// ===========================================================================
// Device unit test of the local solve (test hook arb_dev_softfinger_solve): one LANE per input tuple, the
// same arb_math.h code the kernels run -- inverse of the 4x4 block, SoftFingerContact.solve with the fast
// sliding shift or the eig6 fallback on a lane-private LDS work array.
// in: [n][27] = vel 4 | adm 16 | force 4 | sdist, dt, mu ;  out: [n][9] = force 4 | dforce 4 | branch
// ===========================================================================
template <typename T>
__global__ __launch_bounds__(WAVE) void arb_softfinger_test_kernel(const double *__restrict__ in, double *__restrict__ out,
                                                                   int n, int use_fast)
{
    T *lds = reinterpret_cast<T *>(arb_lds_raw);
    const int lane = threadIdx.x;
    const int i = blockIdx.x * WAVE + lane;
    T *work = lds + lane * 41;
    if (i >= n) return;
    const double *t = in + (size_t)i * 27;
    T v[4], Y[16], P[16], f[4], df[4], eps[3] = {T(1), T(1), T(1)};
    for (int k = 0; k < 4; ++k) { v[k] = (T)t[k]; f[k] = (T)t[20 + k]; }
    for (int k = 0; k < 16; ++k) Y[k] = (T)t[4 + k];
    if (!inv_block<T>(Y, 4, 4, P)) pinv_block<T>(Y, 4, 4, P);
    const int br = softfinger_solve<T>(v, Y, P, f, df, (T)t[24], (T)t[25], (T)t[26], eps, work, use_fast != 0);
    double *o = out + (size_t)i * 9;
    for (int k = 0; k < 4; ++k) { o[k] = (double)f[k]; o[4 + k] = (double)df[k]; }
    o[8] = (double)br;
}

// Device unit test of the wavefront's eig6 (test hook arb_dev_eig6_pair): one wavefront per 6x6 matrix, the one-lane
// routine and the wavefront routine side by side.  out: [n][28] = shift, nfound, wr 6, wi 6 of eig6 | the same of eig6_wave
template <typename T>
__global__ __launch_bounds__(WAVE) void arb_eig6_test_kernel(const double *__restrict__ in, double *__restrict__ out, int n)
{
    T *lds = reinterpret_cast<T *>(arb_lds_raw);
    const int lane = threadIdx.x;
    const int i = blockIdx.x;
    if (i >= n) return;
    T *w0 = lds, *w1 = lds + 48;
    if (lane < 36) { w0[lane] = (T)in[(size_t)i * 36 + lane]; w1[lane] = w0[lane]; }
    WAVE_SYNC();
    double *o = out + (size_t)i * 28;
    const T sw = slide_shift_from_eig_wave<T>(w1, lane);
    {
        T wr[6] = {T(0), T(0), T(0), T(0), T(0), T(0)}, wi[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
        const int nf = eig6_wave<T>(lane < 36 ? w1[lane] : T(0), lane, wr, wi);
        if (lane == 5) {
            o[14] = (double)sw; o[15] = (double)nf;
            for (int k = 0; k < 6; ++k) { o[16 + k] = (double)wr[k]; o[22 + k] = (double)wi[k]; }
        }
    }
    WAVE_SYNC();
    if (lane == 0) {
        T wr[6] = {T(0), T(0), T(0), T(0), T(0), T(0)}, wi[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
        const int nf = eig6<T>(w0, wr, wi);
        for (int k = 0; k < 36; ++k) w0[k] = (T)in[(size_t)i * 36 + k];
        o[0] = (double)slide_shift_from_eig<T>(w0); o[1] = (double)nf;
        for (int k = 0; k < 6; ++k) { o[2 + k] = (double)wr[k]; o[8 + k] = (double)wi[k]; }
    }
}

// ===========================================================================
// Host side: model upload, launch dispatch, C ABI
// ===========================================================================
#ifdef ARB_PART
extern thread_local std::string g_hip_err;
#else
thread_local std::string g_hip_err;
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
    int lds_pad = 0, queue_chunk = 4, queue_tail = 4, queue_spin_cap = 1 << 24;
    int force_waves = 0, force_pack = -1, force_rdv = -1, gsw_waves = 3, gsw_pack = 0, ablate = 0;
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
    if (lds > 64 * 1024) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    // Work queue (see the kernel): multi-step launches of more worlds than the chip has wave slots.  Constraint forces
    // that persist from step to step travel between chunks through `cf`, so it must be there when the model has
    // constraints.  ARB_STEP_STATIC_WORLDS (or ARB_QUEUE_CHUNK=0 in the environment) keeps one workgroup per world.
    // (the rendezvous build, CM = 4: items are single steps, always through the queue)
    const int chunk = (CM == 4) ? 1 : kn.queue_chunk;
    const int tail = (CM == 4) ? 0 : std::max(0, std::min(kn.queue_tail, nsteps - 1));
    const int spin_cap = kn.queue_spin_cap;
    int *queue = nullptr;
    T *park = nullptr;
    const long units = (CM == 3) ? (nw + 1) / 2 : nw;        // work units: worlds, or pairs of worlds (packed build)
    unsigned grid = (unsigned)units;
    constexpr bool QUEUE_LOOP = ARB_QUEUE_LOOP && (ARB_QUEUE_LOOP_ALL || !(sizeof(T) == 8 && NMAX == 64));     // (see the kernel)
    if constexpr (CM == 4) {
        if (cf == nullptr || sio.mode != 0 || nw * (long)nsteps >= (1l << 30)) return ARB_ERR_INVALID;     // (launch() checks before it picks this build)
        auto al4 = [](long x) { return (x + 3) & ~3l; };
        const long ndol = L.ndol, pst = al4(ndol * ndol) + 3 * al4(ndol) + al4(2 * (ndol / ARB_MAXDOL)) + (1 + ndol) * NMAX;
        const size_t qbytes = (size_t)(1 + units + (units + 3) / 4) * sizeof(int), pbytes = (size_t)units * pst * sizeof(T);
        void *blob = nullptr;
        HIP_TRY(arb_scratch_alloc(&blob, qbytes + 256 + pbytes, st));
        queue = static_cast<int *>(blob);
        park = reinterpret_cast<T *>(static_cast<char *>(blob) + ((qbytes + 255) / 256) * 256);
        if (hipMemsetAsync(queue, 0, qbytes, st) != hipSuccess) { g_hip_err = "hipMemsetAsync(queue)"; (void)hipFreeAsync(blob, st); return ARB_ERR_HIP; }
        static thread_local size_t slots_lds = ~(size_t)0;
        static thread_local int slots_dev = -1, slots = 0;
        int dev = -1;
        (void)hipGetDevice(&dev);
        if (slots_lds != lds || slots_dev != dev) { slots = wave_slots(kern, lds); slots_lds = lds; slots_dev = dev; }
        grid = (unsigned)std::max(1l, std::min((long)(slots > 0 ? slots : 1024), units * (long)nsteps));
    } else
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
                       queue, chunk > 0 ? chunk : 1, tail, spin_cap, park, ext_stride, pd_stride, cost);
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
#if defined(ARB_PART) && defined(ARB_PART_SPEC)      /* (translation units of their own: the specialised kernels, tiles 44 / 48) */
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
#if ARB_ALL_VARIANTS
template int launch_one<ARB_PART_T, ARB_PART_NMAX, 1, 0, 0, 3>(ARB_LAUNCH_ONE_ARGS(ARB_PART_T));
template int launch_one<ARB_PART_T, ARB_PART_NMAX, 1, 0, 1, 3>(ARB_LAUNCH_ONE_ARGS(ARB_PART_T));
#endif
#if ARB_WITH_RDV
template int launch_one<ARB_PART_T, ARB_PART_NMAX, 1, 0, 0, 4>(ARB_LAUNCH_ONE_ARGS(ARB_PART_T));
template int launch_one<ARB_PART_T, ARB_PART_NMAX, 1, 0, 1, 4>(ARB_LAUNCH_ONE_ARGS(ARB_PART_T));
#endif
#endif
#endif
#else
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
#if ARB_ALL_VARIANTS
#define ARB_EXTERN_TILE_PK(NM)                                                          \
    extern template int launch_one<float, NM, 1, 0, 0, 3>(ARB_LAUNCH_ONE_ARGS(float));  \
    extern template int launch_one<float, NM, 1, 0, 1, 3>(ARB_LAUNCH_ONE_ARGS(float));
ARB_EXTERN_TILE_PK(44) ARB_EXTERN_TILE_PK(48)
#undef ARB_EXTERN_TILE_PK
#endif
#if ARB_WITH_RDV
#define ARB_EXTERN_TILE_RDV(NM)                                                         \
    extern template int launch_one<float, NM, 1, 0, 0, 4>(ARB_LAUNCH_ONE_ARGS(float));  \
    extern template int launch_one<float, NM, 1, 0, 1, 4>(ARB_LAUNCH_ONE_ARGS(float));
ARB_EXTERN_TILE_RDV(44) ARB_EXTERN_TILE_RDV(48)
#undef ARB_EXTERN_TILE_RDV
#endif
ARB_EXTERN_TILE(float, 16) ARB_EXTERN_TILE(float, 32) ARB_EXTERN_TILE(float, 44) ARB_EXTERN_TILE(float, 48) ARB_EXTERN_TILE(float, 64)
ARB_EXTERN_TILE(double, 16) ARB_EXTERN_TILE(double, 32) ARB_EXTERN_TILE(double, 44) ARB_EXTERN_TILE(double, 48) ARB_EXTERN_TILE(double, 64)
#undef ARB_EXTERN_TILE
#endif

// Makes `device` current for the scope of a C-ABI call and restores the caller's device afterwards (torch reads
// its current device from the HIP runtime: leaving another device current would silently redirect the caller's
// later allocations).
struct DeviceGuard {
    int prev = -1;
    hipError_t err;
    explicit DeviceGuard(int device) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != device) err = hipSetDevice(device);
        else if (err != hipSuccess) prev = -1;
        if (err == hipSuccess && prev == device) prev = -1;      // nothing to restore
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
#define ARB_GUARD_DEVICE(dev)                                                          \
    DeviceGuard guard_(dev);                                                           \
    if (guard_.err != hipSuccess) {                                                    \
        g_hip_err = std::string("hipSetDevice: ") + hipGetErrorString(guard_.err);     \
        return ARB_ERR_HIP;                                                            \
    }

struct arb_model {
    int device;
    int nb, n, nq, nc, ndol, ncols, nsets, nmax;
    std::vector<void *> allocs;
    DevModel<float> df;
    DevModel<double> dd;
    DevModel<float> *df_dev;
    DevModel<double> *dd_dev;
    bool packable = false;         // every constraint a SoftFingerContact with eps = (1,1,1), at most eight: two worlds per wavefront in the sweeps
    bool spec_ok = false;          // four constraints per column set, all enabled SoftFingerContacts of plane / sphere pairs, no PD controller, no viscosity,
                                   // one world per wavefront, tiles 44 / 48: the specialised kernels (FEAT bit 4)
    bool spec0_ok = false;         // no constraints, otherwise the same class: the specialised kernels of FEAT bit 8
    bool bodycols = false;         // the same class with more contacts than one column set holds, on few pairs of bodies: body-space
                                   // constraint columns (FEAT bit 16), ONE column set -- human36 with the reference's eight contact points
    bool bodycols_default = false; // ... chosen without being asked (ARB_STEP_BODY_COLUMNS): when they save the second column set
    Layout lfb, lfb3, ldb;         // ... and their LDS layouts: float32 two-wave / three-wave, float64
    bool rdv_ok = false;           // ... at most FOUR, and the three-wave layout holds three more systems: the rendezvous build (CM = 4)
    int *status_host = nullptr;    // mapped pinned words the kernels raise: [0] a work-queue wait expired (ARB_ERR_STALLED), [1] ARB_WARN_* bits
    Knobs kn;                      // development / test knobs (arb_hook_set_knob)
    Layout lf, lf3, lfp, ld;       // LDS layouts: float32 two-wave kernels, three-wave kernels, packed kernels; float64
    // Small worlds: `forest_k` independent copies of the model as ONE model (copy k owns bodies k nb.., dofs k n.., position
    // scalars k nq.., constraints k nc..), so that a batch of states [nw][nq] of this model IS a batch [nw / k][k nq] of
    // the forest: k worlds share a wavefront's lanes.  Built by arb_model_create for models of at most 16 dofs.
    arb_model *forest = nullptr;
    int forest_k = 1;
};

// ARB_ERR_STALLED when an earlier launch of the handle raised the status word (host memory: no synchronisation).
// The word is STICKY: the stepping / inspecting entry points only look at it (`clear` false) and refuse to launch while
// it is raised -- with asynchronous callers the first call to see it is not necessarily one whose status is checked --;
// arb_model_status alone reads and clears it, which is the caller's acknowledgement.
static int take_status(arb_model *M, bool clear) {
    if (M->status_host == nullptr) return ARB_OK;
    int v = clear ? __atomic_exchange_n(M->status_host, 0, __ATOMIC_RELAXED) : __atomic_load_n(M->status_host, __ATOMIC_RELAXED);
    if (M->forest && M->forest->status_host)
        v |= clear ? __atomic_exchange_n(M->forest->status_host, 0, __ATOMIC_RELAXED)
                   : __atomic_load_n(M->forest->status_host, __ATOMIC_RELAXED);
    return v != 0 ? ARB_ERR_STALLED : ARB_OK;
}

template <typename T>
static int upload(arb_model *M, const std::vector<T> &h, const T **out) {
    void *p = nullptr;
    size_t bytes = std::max<size_t>(h.size(), 1) * sizeof(T);
    HIP_TRY(hipMalloc(&p, bytes));
    M->allocs.push_back(p);
    if (!h.empty()) HIP_TRY(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = reinterpret_cast<const T *>(p);
    return ARB_OK;
}

template <typename T>
static std::vector<T> conv(const double *src, size_t count) {
    std::vector<T> v(count);
    for (size_t i = 0; i < count; ++i) v[i] = static_cast<T>(src ? src[i] : 0.0);
    return v;
}

// 4x4 row-major -> 12 scalars (R row-major, p)
static std::vector<double> h12(const double *H16, int count) {
    std::vector<double> v(static_cast<size_t>(count) * 12);
    for (int b = 0; b < count; ++b) {
        const double *H = H16 + 16 * b;
        double *o = v.data() + 12 * b;
        for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) o[3 * i + j] = H[4 * i + j]; o[9 + i] = H[4 * i + 3]; }
    }
    return v;
}

// LDS layout of one wavefront (elements of T).  Two regions are shared by arrays that are never live together:
//   bd : per-body blocks BD (phases A, A') -> prefix table of the subtree sums (phase B, small trees: nb rows of
//        TB_STRIDE float64) -> per-dof X | P | R vectors (phase B) -> AM = Y' (phase D .. Gauss-Seidel)
//   rt : the joints' own columns SC (phase A .. dof products of phase B) -> RT = [rhs | rows of J'] (end of phase B,
//        phase C) -> solution columns [Y rhs | Y J'^T] (phase D .. E)
// (round 3: SC and AM had regions of their own and the prefix table was 70 float64 wide: 19.4 KB per human36 world;
// 13.1 KB now, which lets twelve wavefronts share a CU's LDS instead of eight)
// (the table restarts at every root, which keeps the trees of a wavefront -- the copies of a forest -- apart; the DPP
// scan of the larger trees runs across all bodies of the wavefront)
static bool lds_scan(int nb, int rs) { return nb <= 24 && rs <= 48; }
// (nbp > 0: the layout of the BODYCOL kernels -- behind Y' the body-space admittance, velocity and the half product W)
static int rt_rows_of(int nbp, int ndol) { return std::max(1 + (nbp > 0 ? 4 * ((6 * nbp + 3) / 4) : ndol), 12); }
// W (6 nbp x ndol) is live between the matrix-core products of phase D, which are the last to read the rows of J in RT, and the
// solution columns' write-back into RT: it borrows that space when it fits
static bool bc_w_in_rt(int nbp, int ndol, int rs) { return 6 * nbp * ndol <= rt_rows_of(nbp, ndol) * rs; }
static int bd_region_elems(int nb, int rs, int ndol, int elems_per_double, bool two_pass, int bd_stride = BD_STRIDE, int nbp = 0, int ndof = 0) {
    auto al = [](int x) { return (x + 3) & ~3; };
    const int tb = lds_scan(nb, rs) ? al(nb * (two_pass ? TB_STRIDE : TB_STRIDE1) * elems_per_double) : 0;
    // (Y' with rows of ndol + 4 elements: see gs_stage; the half product W of the BODYCOL kernels lives in RT's space when it fits)
    const int am = al(std::max(ndol * (ndol + 4), 4)) + (nbp > 0 ? al(36 * nbp * nbp) + al(6 * nbp) + (bc_w_in_rt(nbp, ndol, rs) ? 0 : al(6 * nbp * ndol)) : 0);
    // (the X | P | R vectors: one row per DOF -- rows n .. rs - 1 of the tile are never read; 0: callers that only know the tile)
    return std::max(std::max(std::max(al(nb * bd_stride), al(XPR_STRIDE * (ndof > 0 ? ndof : rs) * elems_per_double)), tb), am);
}

static Layout make_layout(int nb, int nq, int nc, int ndol, int rs, int elems_per_double, int *total_elems, bool two_pass = false,
                          bool pack = false, int nbp = 0, int ndof = 0) {
    auto al = [](int x) { return (x + 3) & ~3; };
    Layout L;
    int o = 0;
    L.q = o; o += al(nq);
    L.dq = o; o += al(rs);                           // (one element per tile row; 64 until round 5)
    L.pd = o; o += al(nb * PDS * elems_per_double);  // body poses kept in float64 (see phase A)
    L.rt = o; L.sc = o; o += rt_rows_of(nbp, ndol) * rs;
    L.cd = o; o += al(nc * CD_STRIDE);               // (nothing without constraints: every access is inside a loop over them)
    L.vv = o; o += al(std::max(ndol, 4));
    L.ff = o; o += al(std::max(ndol, 4));
    L.ff0 = o; o += al(std::max(ndol, 4));
    // the dof-indexed copy of the joint positions (phase A .. the controllers at the end of phase B) shares the scratch
    // array of the later phases (the late-rhs column copy of phase C, the eigenvalue fallback of the sweeps)
    // (44 elements for the eigenvalue fallback; one per dof / per row for the other two.  Sizes matter by the LDS allocation
    // granule, which tools/lds_granule_probe.hip measures at 1280 B -- twelve wavefronts per CU need <= 12 800 B each, not
    // the 13 653 B that 160 KB / 12 and hipOccupancyMaxActiveBlocksPerMultiprocessor suggest: the three-wave human36 + 4
    // contacts layout is 12 800 B with this line, 12 880 B with 64 elements here)
    L.work = o; L.qd = o; o += std::max(44, al(rs));
    {
        const int words = CI_STRIDE * (nbp > 0 ? 1 : std::max(nc, 1));      // int32 words, see CI_STRIDE (BODYCOL kernels: unused)
        L.ci = o; o += al(elems_per_double == 2 ? words : (words + 1) / 2);     // (float: one word per element; double: two)
    }
    // the per-body blocks (and what takes their place) come last: the inspect kernels' larger blocks (BD_STRIDE_INSPECT:
    // the gravity wrench, 6 elements per body more -- 3 KB for a float64 snake-64, the difference between four and five
    // wavefronts per CU for the step kernels) then only lengthen the allocation, every other offset is shared
    L.bd = o; L.am = o;
    const int bd_step = bd_region_elems(nb, rs, ndol, elems_per_double, two_pass, BD_STRIDE, nbp, ndof);
    const int bd_insp = bd_region_elems(nb, rs, ndol, elems_per_double, two_pass, BD_STRIDE_INSPECT, nbp, ndof);
    L.yb = L.am + al(std::max(ndol * (ndol + 4), 4)); L.vb = L.yb + al(36 * nbp * nbp);
    L.wst = (nbp > 0 && bc_w_in_rt(nbp, ndol, rs)) ? L.rt : L.vb + al(6 * nbp);
    L.total_inspect = o + bd_insp;
    o += bd_step;
    L.sa_q = L.sa_dq = L.sa_am = L.sa_cd = L.sa_vv = L.sa_ff = L.sa_ff0 = L.sa_rt = L.sb_q = L.sb_dq = L.sb_ff = 0;
    if (pack) {
        L.sa_rt = o; o += (1 + ndol) * rs;
        L.sa_am = o; o += al(std::max(ndol * ndol, 4));
        L.sa_cd = o; o += al(std::max(nc, 1) * CD_STRIDE);
        L.sa_vv = o; o += al(std::max(ndol, 4));
        L.sa_ff = o; o += al(std::max(ndol, 4));
        L.sa_ff0 = o; o += al(std::max(ndol, 4));
        L.sa_q = o; o += al(nq);
        L.sa_dq = o; o += al(rs);
        L.sb_q = o; o += al(nq);
        L.sb_dq = o; o += al(rs);
        L.sb_ff = o; o += al(std::max(ndol, 4));
    }
    L.total = o;
    if (pack) L.total_inspect = std::max(L.total_inspect, o);      // (no inspect kernel uses the packed layout)
    L.ndol = ndol;
    L.lscan = lds_scan(nb, rs) ? 1 : 0;
    *total_elems = o;
    return L;
}

struct TreeTables {
    std::vector<int> dofbody, subsize;
    std::vector<unsigned long long> upmask, descmask;
};

// zaligned(normal), arboris/homogeneousmatrix.py:201-232 (constant for a contact plane)
static void zaligned_host(const double z[3], double R[9]) {
    int idx[3] = {0, 1, 2};
    double a[3] = {std::fabs(z[0]), std::fabs(z[1]), std::fabs(z[2])};
    std::stable_sort(idx, idx + 3, [&](int i, int j) { return a[i] < a[j]; });
    double x[3] = {0, 0, 0};
    x[idx[0]] = 0; x[idx[1]] = z[idx[2]]; x[idx[2]] = -z[idx[1]];
    double nx = std::sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
    for (int i = 0; i < 3; ++i) x[i] /= nx;
    double y[3] = {z[1] * x[2] - z[2] * x[1], z[2] * x[0] - z[0] * x[2], z[0] * x[1] - z[1] * x[0]};
    for (int i = 0; i < 3; ++i) { R[3 * i] = x[i]; R[3 * i + 1] = y[i]; R[3 * i + 2] = z[i]; }
}

template <typename T, typename S, size_t N>
static void fill(T (&dst)[N], const S *src, size_t count) {
    for (size_t i = 0; i < N; ++i) dst[i] = (i < count && src) ? static_cast<T>(src[i]) : T(0);
}

template <typename T>
static int build_dev(arb_model *M, const arb_model_desc *d, const std::vector<int> &jnd,
                     const std::vector<int> &depth, const std::vector<unsigned long long> &anc,
                     const std::vector<int> &dof2q, int maxdepth, const TreeTables &tt, int fk, DevModel<T> *out) {
    DevModel<T> &m = *out;
    memset(&m, 0, sizeof(m));
    const int nb = d->nb, n = d->ndof, nc = d->nc;
    m.fk = fk; m.fn = n / fk; m.fnq = d->nq / fk; m.fnc = nc / fk;
    {
        std::vector<T> qdef((size_t)d->nq, T(0));
        for (int b = 0; b < nb; ++b)
            if (d->jtype[b] == ARB_JT_FREE)
                for (int i = 0; i < 4; ++i) qdef[(size_t)d->q_off[b] + 5 * i] = T(1);
        int rcq;
        if ((rcq = upload<T>(M, qdef, &m.qdef)) != ARB_OK) return rcq;
    }
    m.nb = nb; m.n = n; m.nq = d->nq; m.nc = nc; m.ndol = ARB_MAXDOL * nc; m.ncols = n + 1 + m.ndol;
    m.maxdepth = maxdepth;
    int rc;
    fill(m.parent, d->parent, nb); fill(m.jtype, d->jtype, nb); fill(m.dof_off, d->dof_off, nb);
    fill(m.jnd, jnd.data(), nb); fill(m.q_off, d->q_off, nb); fill(m.depth, depth.data(), nb);
    fill(m.weighted, d->weighted, nb); fill(m.dof2q, dof2q.data(), n);
    fill(m.dofbody, tt.dofbody.data(), n); fill(m.subsize, tt.subsize.data(), nb);
    m.rootmask = 0ull;
    for (int b = 0; b < nb; ++b) {
        m.root[b] = d->parent[b] < 0 ? b : m.root[d->parent[b]];
        if (d->parent[b] < 0) m.rootmask |= 1ull << b;
    }
    fill(m.upmask, tt.upmask.data(), n); fill(m.descmask, tt.descmask.data(), n); fill(m.anc, anc.data(), nb);

    const std::vector<double> hpr = h12(d->H_pr, nb), hcn = h12(d->H_cn, nb);
    fill(m.Hpr, hpr.data(), 12 * nb); fill(m.Hcn, hcn.data(), 12 * nb);
    fill(m.Hpr_d, hpr.data(), 12 * nb); fill(m.Hcn_d, hcn.data(), 12 * nb);
    fill(m.clocal_d, d->c_local, 3 * nc); fill(m.cradius_d, d->c_radius, nc); fill(m.cradius0_d, d->c_radius0, nc);
    fill(m.chalf_d, d->c_half, 3 * nc); fill(m.cplane_d, d->c_plane, 4 * nc);
    const std::vector<double> cb0 = h12(d->c_bpose0, nc), cb1 = h12(d->c_bpose1, nc);
    fill(m.cb0_d, cb0.data(), 12 * nc); fill(m.cb1_d, cb1.data(), 12 * nc);
    {
        // centre of mass of every body as massmatrix.principalframe places it (massmatrix.py:96-99),
        // and its mass; massless bodies contribute nothing
        std::vector<double> com(4 * (size_t)nb, 0.0);
        for (int b = 0; b < nb; ++b) {
            const double *Mb = d->mass + 36 * b;
            const double mass_b = Mb[35];
            if (mass_b > 0.0) {
                com[4 * b + 0] = Mb[6 * 2 + 4] / mass_b; com[4 * b + 1] = Mb[6 * 0 + 5] / mass_b;
                com[4 * b + 2] = Mb[6 * 1 + 3] / mass_b; com[4 * b + 3] = Mb[21];
            }
        }
        fill(m.com_d, com.data(), com.size());
        for (int i = 0; i < 3; ++i) m.up[i] = d->up[i];
    }
    fill(m.mass, d->mass, 36 * nb); fill(m.visc, d->visc, 36 * nb);
    bool hv = false;
    for (int i = 0; i < 36 * nb; ++i) hv = hv || (d->visc[i] != 0.0);
    m.has_visc = hv;
    m.has_grav = 0;
    for (int i = 0; i < 3; ++i) { m.grav[i] = (T)d->gravity[i]; if (d->gravity[i] != 0.0) m.has_grav = 1; }
    m.has_pd = (d->pd_kp != nullptr);
    if ((rc = upload<T>(M, conv<T>(d->pd_kp, m.has_pd ? n * n : 0), &m.pd_kp)) != ARB_OK) return rc;
    if ((rc = upload<T>(M, conv<T>(d->pd_kd, m.has_pd ? n * n : 0), &m.pd_kd)) != ARB_OK) return rc;
    if ((rc = upload<T>(M, conv<T>(d->pd_tau0, m.has_pd ? n : 0), &m.pd_tau0)) != ARB_OK) return rc;
    // constraints
    fill(m.ctype, d->ctype, nc); fill(m.cen, d->c_enabled, nc); fill(m.cbody, d->c_body, nc);
    fill(m.cbody0, d->c_body0, nc); fill(m.cdof, d->c_dof, nc); fill(m.cgeom, d->c_geom, nc);
    std::vector<double> rz(9 * (size_t)std::max(nc, 1), 0.0);
    m.has_warm = 0;
    for (int c = 0; c < nc; ++c) {
        if (d->ctype[c] == ARB_CT_SOFTFINGER && d->c_geom[c] == ARB_CG_PLANE_SPHERE)
            zaligned_host(d->c_plane + 4 * c, rz.data() + 9 * c);
        if (d->ctype[c] == ARB_CT_BALLSOCKET) m.has_warm = 1;
    }
    fill(m.cRz_d, rz.data(), 9 * nc);
    fill(m.cmu, d->c_mu, nc); fill(m.cprox_d, d->c_prox, nc); fill(m.ceps, d->c_eps, 3 * nc);
    fill(m.cmin_d, d->c_min, nc); fill(m.cmax_d, d->c_max, nc);
    return ARB_OK;
}

// test hooks: host builds of the device narrow phase (same source as the kernel)
extern "C" void arb_host_zaligned(const double z[3], double R[9]) {
    const M3<double> m = zaligned_rot(v3<double>(z[0], z[1], z[2]));
    for (int i = 0; i < 9; ++i) R[i] = m.a[i];
}

extern "C" double arb_host_narrow_phase(int geom, const double H_s0[16], const double p_g1[3], double rad,
                                        double r0, const double half[3], const double plane[4],
                                        double gc0[3], double gc1[3], double Rc[9]) {
    M3<double> Rs0, Rz, R;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rs0.a[3 * i + j] = H_s0[4 * i + j];
    double rz[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (geom == ARB_CG_PLANE_SPHERE) zaligned_host(plane, rz);
    for (int i = 0; i < 9; ++i) Rz.a[i] = rz[i];
    V3<double> a, b;
    const double sd = narrow_phase(geom, Rs0, v3<double>(H_s0[3], H_s0[7], H_s0[11]), v3<double>(p_g1[0], p_g1[1], p_g1[2]),
                                   rad, r0, v3<double>(half[0], half[1], half[2]),
                                   v3<double>(plane[0], plane[1], plane[2]), plane[3], Rz, a, b, R);
    gc0[0] = a.x; gc0[1] = a.y; gc0[2] = a.z; gc1[0] = b.x; gc1[1] = b.y; gc1[2] = b.z;
    for (int i = 0; i < 9; ++i) Rc[i] = R.a[i];
    return sd;
}

extern "C" int arb_abi_version(void) { return ARB_ABI_VERSION; }

extern "C" const char *arb_strerror(int status) {
    switch (status) {
        case ARB_OK: return "ok";
        case ARB_ERR_INVALID: return "invalid argument";
        case ARB_ERR_UNSUPPORTED: return "model not supported by the device step";
        case ARB_ERR_HIP: return "HIP runtime error";
        case ARB_ERR_NOMEM: return "out of memory";
        case ARB_ERR_STALLED: return "an earlier launch on this handle gave up waiting in its work queue: its results are invalid";
        default: return "unknown status";
    }
}

extern "C" const char *arb_last_hip_error(void) { return g_hip_err.c_str(); }

static const int kNmaxChoices[] = {16, 32, 44, 48, 64};    // 44: human36 (42 dofs) wastes 2 rows instead of 6

static int forest_create(const arb_model_desc *d, int K, int device, arb_model **out);

// How many copies of a small model share a wavefront (1: none).  Measured on an MI355X (tools/experiments/forest_probe.py,
// tools/experiments/forest_rate.py, M world-steps/s at 65 536 worlds x 64 steps): simplearm (3 dofs) float32 106
// alone, 274 with 5 copies (still the 16-row tile), 526 with 8, 241 with 21 (64 rows); float64 86 / 345 / ~400 / 104;
// the 15-dof free snake 62 alone, 95 as a pair; ball and socket (6 dofs, 1 constraint) 83 alone, 198 with 5 copies,
// 193 with 7 (two column sets).  Hence: as many copies as fit the 32-row tile with ONE set of columns, and the 24-body
// prefix table whose restart at every root keeps the copies apart.  ARB_FOREST=0 in the environment turns the forest
// off, ARB_FOREST=k asks for k copies (development).
static int forest_copies(int nb, int n, int nc) {
#ifdef ARB_DEVELOPMENT
    const int want = getenv("ARB_FOREST") ? atoi(getenv("ARB_FOREST")) : -1;
#else
    const int want = -1;
#endif
    if (want == 0 || want == 1) return 1;
    int K = 1;
    for (int k = 2; k <= WAVE; ++k) {
        const bool fits = k * nb <= 24 && k * nc * ARB_MAXDOL <= WAVE &&      // (24 bodies: the prefix table of phase B)
                          k * n <= 32 && k * n + 1 + ARB_MAXDOL * k * nc <= (want > 1 ? 2 * WAVE : WAVE);
        if (!fits) break;
        K = k;
        if (want > 1 && k == want) break;
    }
    return K;
}

// fk = 1: the described world, plus its forest when it is small; fk > 1: `d` describes a forest of fk copies
static int model_create(const arb_model_desc *d, int device, arb_model **out, int fk);

extern "C" int arb_model_create(const arb_model_desc *d, int device, arb_model **out) {
    return model_create(d, device, out, 1);
}

static int model_create(const arb_model_desc *d, int device, arb_model **out, int fk) {
    const bool with_forest = fk == 1;
    if (d == nullptr || out == nullptr) return ARB_ERR_INVALID;
    *out = nullptr;
    if (d->abi_version != ARB_ABI_VERSION) return ARB_ERR_INVALID;
    const int nb = d->nb, n = d->ndof, nc = d->nc;
    if (nb <= 0 || n <= 0 || d->nq <= 0 || nc < 0) return ARB_ERR_INVALID;
    if (!d->parent || !d->jtype || !d->dof_off || !d->q_off || !d->H_pr || !d->H_cn || !d->mass ||
        !d->visc || !d->weighted)
        return ARB_ERR_INVALID;
    if (nc > 0 && (!d->ctype || !d->c_enabled || !d->c_body || !d->c_body0 || !d->c_dof || !d->c_local ||
                   !d->c_radius || !d->c_geom || !d->c_radius0 || !d->c_half || !d->c_plane || !d->c_mu || !d->c_prox || !d->c_eps ||
                   !d->c_min || !d->c_max || !d->c_bpose0 || !d->c_bpose1))
        return ARB_ERR_INVALID;
    if (n > WAVE || nb > WAVE || nc > WAVE) return ARB_ERR_UNSUPPORTED;   // one world per wavefront
    const int ndol = ARB_MAXDOL * nc;
    const int ncols = n + 1 + ndol;
    if (ncols > 2 * WAVE || ndol > WAVE) return ARB_ERR_UNSUPPORTED;
    // ---- tree bookkeeping (counterpart of World.init, core.py:608-635) ----
    std::vector<int> jnd(nb), depth(nb);
    std::vector<unsigned long long> anc(nb);
    std::vector<int> dof2q(n, -1);
    int maxdepth = 0, ndof_chk = 0, nq_chk = 0;
    for (int b = 0; b < nb; ++b) {
        const int p = d->parent[b], jt = d->jtype[b];
        if (p >= b || p < -1 || jt < 0 || jt > ARB_JT_TXTYTZ) return ARB_ERR_INVALID;
        jnd[b] = joint_ndof(jt);
        if (d->dof_off[b] != ndof_chk || d->q_off[b] != nq_chk) return ARB_ERR_INVALID;
        ndof_chk += jnd[b]; nq_chk += joint_nq(jt);
        depth[b] = (p < 0) ? 0 : depth[p] + 1;
        maxdepth = std::max(maxdepth, depth[b]);
        unsigned long long own = 0;
        for (int i = 0; i < jnd[b]; ++i) own |= 1ull << (d->dof_off[b] + i);
        anc[b] = own | (p < 0 ? 0ull : anc[p]);
        if (jt != ARB_JT_FREE)
            for (int i = 0; i < jnd[b]; ++i) dof2q[d->dof_off[b] + i] = d->q_off[b] + i;
    }
    if (ndof_chk != n || nq_chk != d->nq) return ARB_ERR_INVALID;
    // composite phase B tables
    TreeTables tt;
    tt.dofbody.assign(n, 0); tt.upmask.assign(n, 0ull); tt.descmask.assign(n, 0ull);
    for (int b = 0; b < nb; ++b)
        for (int i = 0; i < jnd[b]; ++i) { tt.dofbody[d->dof_off[b] + i] = b; tt.upmask[d->dof_off[b] + i] = anc[b]; }
    for (int k = 0; k < n; ++k)
        for (int i = 0; i < n; ++i)
            if (((anc[tt.dofbody[i]] >> k) & 1ull) && tt.dofbody[i] != tt.dofbody[k]) tt.descmask[k] |= 1ull << i;
    // subtree sizes; DFS preorder numbering (what World.init produces, core.py:611-615) makes every subtree a
    // contiguous range of bodies, which lets phase B form the subtree sums from a prefix scan over the lanes
    tt.subsize.assign(nb, 1);
    for (int b = nb - 1; b > 0; --b) if (d->parent[b] >= 0) tt.subsize[d->parent[b]] += tt.subsize[b];   // -1: child of the ground
    for (int b = 0; b < nb; ++b)
        for (int c2 = b + 1; c2 < b + tt.subsize[b]; ++c2) {
            int a = c2;
            while (a > b) a = d->parent[a];                 // (a root's parent is -1)
            if (a != b) return ARB_ERR_UNSUPPORTED;         // bodies must come in DFS preorder (include/arbstep.h)
        }
    // The composite assembly writes N_b as -ad([w; c x w])^T M_b, which needs rigid-body mass matrices
    // [[I, m c^],[m c^T, m 1]] (everything arboris/massmatrix.py builds); anything else is refused.
    for (int b = 0; b < nb; ++b) {
        const double *Mb = d->mass + 36 * b;
        double sc = 0.;
        for (int i = 0; i < 36; ++i) sc = std::max(sc, std::fabs(Mb[i]));
        const double tol = 1e-9 * std::max(sc, 1e-300);
        bool ok = true;
        for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) ok = ok && std::fabs(Mb[6 * i + j] - Mb[6 * j + i]) <= tol;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
            ok = ok && std::fabs(Mb[6 * (3 + i) + 3 + j] - (i == j ? Mb[21] : 0.)) <= tol;       // m 1
            ok = ok && std::fabs(Mb[6 * i + 3 + j] + Mb[6 * j + 3 + i]) <= tol;                  // m c^ skew
        }
        if (!ok) return ARB_ERR_UNSUPPORTED;
    }
    // constraint table checks
    for (int c = 0; c < nc; ++c) {
        const int ct = d->ctype[c];
        if (ct == ARB_CT_SOFTFINGER) {
            if (d->c_body[c] >= nb || d->c_body0[c] >= nb) return ARB_ERR_INVALID;
            if (d->c_geom[c] < ARB_CG_PLANE_SPHERE || d->c_geom[c] > ARB_CG_BOX_SPHERE) return ARB_ERR_INVALID;
        } else if (ct == ARB_CT_BALLSOCKET) {
            if (d->c_body[c] >= nb || d->c_body0[c] >= nb) return ARB_ERR_INVALID;
        } else if (ct == ARB_CT_JOINTLIMITS) {
            if (d->c_dof[c] < 0 || d->c_dof[c] >= n || dof2q[d->c_dof[c]] < 0) return ARB_ERR_INVALID;
        } else {
            return ARB_ERR_INVALID;
        }
    }

    arb_model *M = new (std::nothrow) arb_model();
    if (!M) return ARB_ERR_NOMEM;
    M->device = device;
    M->nb = nb; M->n = n; M->nq = d->nq; M->nc = nc; M->ndol = ndol; M->ncols = ncols;
    // A world with exactly 64 dofs and no constraints has 65 columns: instead of a second register set for the one
    // column that does not fit, the rhs column joins late (phase C: "late rhs") and one set is enough.
    M->nsets = (ncols > WAVE && !(nc == 0 && n == WAVE)) ? 2 : 1;
    M->nmax = 64;
    for (int c : kNmaxChoices) if (c >= n) { M->nmax = c; break; }
    M->packable = nc >= 1 && nc <= 8 && fk == 1;      // (a forest's copies are retired one by one, which the packed build does not do)
    for (int c = 0; c < nc; ++c)
        M->packable = M->packable && d->ctype[c] == ARB_CT_SOFTFINGER && d->c_eps[3 * c] == 1. && d->c_eps[3 * c + 1] == 1. && d->c_eps[3 * c + 2] == 1.;
    M->spec_ok = ARB_WITH_SPEC && nc == 4 && M->nsets == 1 && fk == 1 && M->nmax >= 44 && M->nmax <= 48;
    for (int c = 0; c < nc; ++c)
        M->spec_ok = M->spec_ok && d->ctype[c] == ARB_CT_SOFTFINGER && d->c_enabled[c] != 0 && d->c_geom[c] == ARB_CG_PLANE_SPHERE;
    M->spec0_ok = ARB_WITH_SPEC && nc == 0 && fk == 1 && M->nsets == 1 && M->nmax >= 44 && M->nmax <= 48 &&
                  maxdepth < ARB_JUMP_DEPTH && lds_scan(nb, M->nmax) && d->pd_kp == nullptr;
    for (int i = 0; i < 36 * nb && M->spec0_ok; ++i) M->spec0_ok = d->visc[i] == 0.0;
    M->spec_ok = M->spec_ok && maxdepth < ARB_JUMP_DEPTH;                                     // (a shallow tree: no log-depth chains in float64)
    M->spec_ok = M->spec_ok && lds_scan(nb, M->nmax);                                         // (a small tree: phase B on the prefix table)
    M->spec_ok = M->spec_ok && d->pd_kp == nullptr;                                           // (no PD controller in the model,
    for (int i = 0; i < 36 * nb && M->spec_ok; ++i) M->spec_ok = d->visc[i] == 0.0;           //  no joint viscosity)
    // Body-space constraint columns (FEAT bit 16): the same class -- enabled plane / sphere SoftFingerContacts only, no PD
    // controller, no viscosity, one small shallow tree -- with more contacts than one column set holds (ndof + 1 + 4 nc > 64) on
    // so few pairs of bodies that six columns per pair fit (ndof + 1 + 6 nbp <= 64)
    int bc_nbp = 0, bc_b1[ARB_MAXPAIR], bc_b0[ARB_MAXPAIR];
    std::vector<int> bc_cpair(std::max(nc, 1), 0);
    {
        bool ok = ARB_WITH_SPEC && fk == 1 && nc > 0 && M->nmax >= 44 && M->nmax <= 48 && maxdepth < ARB_JUMP_DEPTH &&
                  lds_scan(nb, M->nmax) && d->pd_kp == nullptr;
        for (int i = 0; i < 36 * nb && ok; ++i) ok = d->visc[i] == 0.0;
        int nroots = 0;
        for (int b = 0; b < nb; ++b) nroots += d->parent[b] < 0;
        ok = ok && nroots == 1;
        for (int c = 0; c < nc && ok; ++c) {
            ok = d->ctype[c] == ARB_CT_SOFTFINGER && d->c_enabled[c] != 0 && d->c_geom[c] == ARB_CG_PLANE_SPHERE &&
                 (d->c_body[c] >= 0 || d->c_body0[c] >= 0);
            int p = 0;
            while (p < bc_nbp && !(bc_b1[p] == d->c_body[c] && bc_b0[p] == d->c_body0[c])) ++p;
            if (p == bc_nbp) {
                if (bc_nbp == ARB_MAXPAIR) { ok = false; break; }
                bc_b1[p] = d->c_body[c]; bc_b0[p] = d->c_body0[c]; ++bc_nbp;
            }
            bc_cpair[c] = p;
        }
        M->bodycols = ok && n + 1 + 6 * bc_nbp <= WAVE;
        // by default where they save the second column set; ARB_STEP_BODY_COLUMNS asks for them wherever the model qualifies
        M->bodycols_default = M->bodycols && M->nsets == 2;
#ifdef ARB_DEVELOPMENT
        if (getenv("ARB_BODYCOL_ALL")) M->bodycols_default = M->bodycols;
#endif
    }
    DeviceGuard guard_(device);
    if (guard_.err != hipSuccess) {
        g_hip_err = std::string("hipSetDevice: ") + hipGetErrorString(guard_.err);
        delete M;
        return ARB_ERR_HIP;
    }
    int rc = build_dev<float>(M, d, jnd, depth, anc, dof2q, maxdepth, tt, fk, &M->df);
    if (rc == ARB_OK)
        rc = build_dev<double>(M, d, jnd, depth, anc, dof2q, maxdepth, tt, fk, &M->dd);
    if (rc != ARB_OK) { arb_model_destroy(M); return rc; }
    {
        void *hp = nullptr, *dp = nullptr;
        hipError_t e = hipHostMalloc(&hp, 2 * sizeof(int), hipHostMallocMapped);
        if (e == hipSuccess) { M->status_host = static_cast<int *>(hp); M->status_host[0] = M->status_host[1] = 0; e = hipHostGetDevicePointer(&dp, hp, 0); }
        if (e != hipSuccess) {
            g_hip_err = std::string("status word: ") + hipGetErrorString(e);
            arb_model_destroy(M);
            return ARB_ERR_HIP;
        }
        M->df.status = M->dd.status = static_cast<int *>(dp);
        M->df.warn = M->dd.warn = static_cast<int *>(dp) + 1;
    }
    int tot;
    M->lf = M->df.lay = make_layout(nb, d->nq, nc, ndol, M->nmax, 2, &tot, false, false, 0, n);
    M->lf3 = M->df.lay3 = make_layout(nb, d->nq, nc, ndol, M->nmax, 2, &tot, true, false, 0, n);
    M->lfp = M->df.layp = make_layout(nb, d->nq, nc, ndol, M->nmax, 2, &tot, true, true, 0, n);
    {
        // the rendezvous build keeps three more constraint-space systems in the three-wave layout: two behind Y' in the
        // per-body region, one in the space of RT
        auto al4 = [](int x) { return (x + 3) & ~3; };
        const int nA = al4(ndol * ndol), SS = nA + al4(nc * CD_STRIDE) + 3 * al4(ndol);
        const int bdr = bd_region_elems(nb, M->nmax, ndol, 2, true, BD_STRIDE, 0, n), rtr = std::max(1 + ndol, 12) * M->nmax;
        M->rdv_ok = M->packable && nc <= 4 && M->nsets == 1 && M->nmax >= 44 && M->nmax <= 48 && nA + 2 * SS <= bdr && SS <= rtr;
    }
    M->ld = M->dd.lay = M->dd.lay3 = M->dd.layp = make_layout(nb, d->nq, nc, ndol, M->nmax, 1, &tot, false, false, 0, n);
    if ((size_t)tot * sizeof(double) > 160 * 1024) { arb_model_destroy(M); return ARB_ERR_UNSUPPORTED; }
    if (M->bodycols) {
        int tb;
        M->lfb = M->df.layb = make_layout(nb, d->nq, nc, ndol, M->nmax, 2, &tb, false, false, bc_nbp, n);
        M->lfb3 = M->df.layb3 = make_layout(nb, d->nq, nc, ndol, M->nmax, 2, &tb, true, false, bc_nbp, n);
        M->ldb = M->dd.layb = M->dd.layb3 = make_layout(nb, d->nq, nc, ndol, M->nmax, 1, &tb, false, false, bc_nbp, n);
        auto fillp = [&](auto &dm) {
            dm.nbp = bc_nbp; dm.ncols_b = n + 1 + 6 * bc_nbp;
            for (int p = 0; p < bc_nbp; ++p) {
                dm.pair_ref[p] = bc_b1[p] >= 0 ? bc_b1[p] : bc_b0[p];
                dm.pair_a1[p] = bc_b1[p] >= 0 ? anc[bc_b1[p]] : 0ull;
                dm.pair_a0[p] = bc_b0[p] >= 0 ? anc[bc_b0[p]] : 0ull;
                dm.pair_cmask[p] = 0ull;
            }
            for (int c = 0; c < nc; ++c) { dm.cpair[c] = bc_cpair[c]; dm.pair_cmask[bc_cpair[c]] |= 1ull << c; }
        };
        fillp(M->df); fillp(M->dd);
    }
    {
        // one blob per precision
        void *pf = nullptr, *pd = nullptr;
        hipError_t e1 = hipMalloc(&pf, sizeof(DevModel<float>));
        if (e1 == hipSuccess) M->allocs.push_back(pf);
        hipError_t e2 = (e1 == hipSuccess) ? hipMalloc(&pd, sizeof(DevModel<double>)) : e1;
        if (e2 == hipSuccess) M->allocs.push_back(pd);
        if (e2 == hipSuccess) e2 = hipMemcpy(pf, &M->df, sizeof(DevModel<float>), hipMemcpyHostToDevice);
        if (e2 == hipSuccess) e2 = hipMemcpy(pd, &M->dd, sizeof(DevModel<double>), hipMemcpyHostToDevice);
        if (e2 != hipSuccess) {
            g_hip_err = std::string("model upload: ") + hipGetErrorString(e2);
            arb_model_destroy(M);
            return ARB_ERR_HIP;
        }
        M->df_dev = static_cast<DevModel<float> *>(pf); M->dd_dev = static_cast<DevModel<double> *>(pd);
    }
#ifdef ARB_DEVELOPMENT
    {   // development builds: the knobs from the environment, once
        const char *names[] = {"lds_pad", "queue_chunk", "queue_tail", "queue_spin_cap", "force_waves", "force_pack", "force_rdv", "gsw_waves", "gsw_pack", "ablate"};
        for (const char *nm : names) {
            std::string e = std::string("ARB_") + nm;
            for (auto &ch : e) ch = (char)toupper((unsigned char)ch);
            if (const char *v = getenv(e.c_str())) (void)arb_hook_set_knob(M, nm, atoi(v));
        }
    }
#endif
    if (with_forest) {
        const int K = forest_copies(nb, n, nc);
        if (K > 1) {
            // (a forest the device step cannot take is not an error of the model: it then runs one world per wavefront)
            arb_model *F = nullptr;
            const int frc = forest_create(d, K, device, &F);
            if (frc == ARB_OK) { M->forest = F; M->forest_k = K; }
            else if (frc == ARB_ERR_HIP || frc == ARB_ERR_NOMEM) { arb_model_destroy(M); return frc; }
        }
    }
    *out = M;
    return ARB_OK;
}

// K copies of the described world as one description (see arb_model::forest), handed to model_create.
static int forest_create(const arb_model_desc *d, int K, int device, arb_model **out) {
    const int nb = d->nb, n = d->ndof, nq = d->nq, nc = d->nc;
    arb_model_desc f = *d;
    f.nb = K * nb; f.ndof = K * n; f.nq = K * nq; f.nc = K * nc;
    // index arrays: entries >= 0 shift by k * step; everything else is repeated
    auto reps_i = [&](const int32_t *src, int count, int step) {
        std::vector<int32_t> v;
        if (!src) return v;
        v.resize((size_t)K * count);
        for (int k = 0; k < K; ++k)
            for (int i = 0; i < count; ++i) v[(size_t)k * count + i] = (step > 0 && src[i] >= 0) ? src[i] + k * step : src[i];
        return v;
    };
    auto reps_d = [&](const double *src, int count) {
        std::vector<double> v;
        if (!src) return v;
        v.resize((size_t)K * count);
        for (int k = 0; k < K; ++k) std::copy(src, src + count, v.begin() + (size_t)k * count);
        return v;
    };
    auto ptr_i = [](const std::vector<int32_t> &v) { return v.empty() ? nullptr : v.data(); };
    auto ptr_d = [](const std::vector<double> &v) { return v.empty() ? nullptr : v.data(); };
    const auto parent = reps_i(d->parent, nb, nb), jtype = reps_i(d->jtype, nb, 0), weighted = reps_i(d->weighted, nb, 0);
    std::vector<int32_t> dof_off2((size_t)K * nb), q_off2((size_t)K * nb);
    for (int k = 0; k < K; ++k)
        for (int i = 0; i < nb; ++i) { dof_off2[(size_t)k * nb + i] = d->dof_off[i] + k * n; q_off2[(size_t)k * nb + i] = d->q_off[i] + k * nq; }
    const auto H_pr = reps_d(d->H_pr, 16 * nb), H_cn = reps_d(d->H_cn, 16 * nb), mass = reps_d(d->mass, 36 * nb),
               visc = reps_d(d->visc, 36 * nb);
    f.parent = ptr_i(parent); f.jtype = ptr_i(jtype); f.dof_off = dof_off2.data(); f.q_off = q_off2.data();
    f.weighted = ptr_i(weighted);
    f.H_pr = ptr_d(H_pr); f.H_cn = ptr_d(H_cn); f.mass = ptr_d(mass); f.visc = ptr_d(visc);
    // merged PD controllers: block-diagonal gain matrices
    std::vector<double> kp, kd, tau0;
    if (d->pd_kp) {
        const size_t N = (size_t)K * n;
        kp.assign(N * N, 0.); kd.assign(N * N, 0.); tau0.assign(N, 0.);
        for (int k = 0; k < K; ++k)
            for (int i = 0; i < n; ++i) {
                for (int j = 0; j < n; ++j) {
                    kp[((size_t)k * n + i) * N + (size_t)k * n + j] = d->pd_kp[(size_t)i * n + j];
                    kd[((size_t)k * n + i) * N + (size_t)k * n + j] = d->pd_kd ? d->pd_kd[(size_t)i * n + j] : 0.;
                }
                tau0[(size_t)k * n + i] = d->pd_tau0 ? d->pd_tau0[i] : 0.;
            }
        f.pd_kp = kp.data(); f.pd_kd = kd.data(); f.pd_tau0 = tau0.data();
    }
    const auto ctype = reps_i(d->ctype, nc, 0), c_enabled = reps_i(d->c_enabled, nc, 0), c_geom = reps_i(d->c_geom, nc, 0),
               c_body = reps_i(d->c_body, nc, nb), c_body0 = reps_i(d->c_body0, nc, nb);
    std::vector<int32_t> c_dof = reps_i(d->c_dof, nc, 0);
    if (d->c_dof)
        for (int k = 0; k < K; ++k)
            for (int c = 0; c < nc; ++c) c_dof[(size_t)k * nc + c] = d->c_dof[c] >= 0 ? d->c_dof[c] + k * n : d->c_dof[c];
    const auto c_local = reps_d(d->c_local, 3 * nc), c_radius = reps_d(d->c_radius, nc), c_radius0 = reps_d(d->c_radius0, nc),
               c_half = reps_d(d->c_half, 3 * nc), c_plane = reps_d(d->c_plane, 4 * nc), c_mu = reps_d(d->c_mu, nc),
               c_prox = reps_d(d->c_prox, nc), c_eps = reps_d(d->c_eps, 3 * nc), c_min = reps_d(d->c_min, nc),
               c_max = reps_d(d->c_max, nc), c_bpose0 = reps_d(d->c_bpose0, 16 * nc), c_bpose1 = reps_d(d->c_bpose1, 16 * nc);
    f.ctype = ptr_i(ctype); f.c_enabled = ptr_i(c_enabled); f.c_geom = ptr_i(c_geom); f.c_body = ptr_i(c_body);
    f.c_body0 = ptr_i(c_body0); f.c_dof = ptr_i(c_dof);
    f.c_local = ptr_d(c_local); f.c_radius = ptr_d(c_radius); f.c_radius0 = ptr_d(c_radius0); f.c_half = ptr_d(c_half);
    f.c_plane = ptr_d(c_plane); f.c_mu = ptr_d(c_mu); f.c_prox = ptr_d(c_prox); f.c_eps = ptr_d(c_eps);
    f.c_min = ptr_d(c_min); f.c_max = ptr_d(c_max); f.c_bpose0 = ptr_d(c_bpose0); f.c_bpose1 = ptr_d(c_bpose1);
    return model_create(&f, device, out, K);
}

extern "C" int arb_model_destroy(arb_model *M) {
    if (!M) return ARB_ERR_INVALID;
    DeviceGuard guard_(M->device);
    for (void *p : M->allocs) (void)hipFree(p);       // (hipFree waits for the work that uses it)
    if (M->status_host) (void)hipHostFree(M->status_host);
    if (M->forest) (void)arb_model_destroy(M->forest);
    delete M;
    return ARB_OK;
}

extern "C" int arb_model_status(arb_model *M) {
    if (!M) return ARB_ERR_INVALID;
    return take_status(M, true);
}

extern "C" int arb_model_warnings(arb_model *M, uint32_t *warnings) {
    if (!M || !warnings) return ARB_ERR_INVALID;
    unsigned v = M->status_host ? (unsigned)__atomic_exchange_n(M->status_host + 1, 0, __ATOMIC_RELAXED) : 0u;
    if (M->forest && M->forest->status_host) v |= (unsigned)__atomic_exchange_n(M->forest->status_host + 1, 0, __ATOMIC_RELAXED);
    *warnings = v;
    return ARB_OK;
}

extern "C" int arb_hook_set_knob(arb_model *M, const char *name, int value) {
    if (!M || !name) return ARB_ERR_INVALID;
    struct { const char *n; int Knobs::*f; } tab[] = {
        {"lds_pad", &Knobs::lds_pad}, {"queue_chunk", &Knobs::queue_chunk}, {"queue_tail", &Knobs::queue_tail},
        {"queue_spin_cap", &Knobs::queue_spin_cap}, {"force_waves", &Knobs::force_waves}, {"force_pack", &Knobs::force_pack},
        {"force_rdv", &Knobs::force_rdv}, {"gsw_waves", &Knobs::gsw_waves}, {"gsw_pack", &Knobs::gsw_pack}, {"ablate", &Knobs::ablate}};
    for (auto &t : tab)
        if (strcmp(t.n, name) == 0) {
            M->kn.*(t.f) = value;
            if (M->forest) M->forest->kn.*(t.f) = value;
            return ARB_OK;
        }
    return ARB_ERR_INVALID;
}

extern "C" int arb_model_get_info(const arb_model *M, arb_model_info *info) {
    if (!M || !info) return ARB_ERR_INVALID;
    info->nb = M->nb; info->ndof = M->n; info->nq = M->nq; info->nc = M->nc;
    info->nmax = M->nmax; info->ncols = M->ncols; info->nsets = M->nsets;
    info->lds_bytes_f32 = M->lf.total * (int)sizeof(float);
    info->lds_bytes_f64 = M->ld.total * (int)sizeof(double);
    info->device = M->device;
    info->forest_copies = M->forest ? M->forest_k : 1;
    return ARB_OK;
}

// Which build of the float32 production kernel runs a launch (models with one column set and a tile of up to 48 rows;
// every other model has the two-wave build only)?  The builds are bit-identical (-ffp-contract=on): a pure performance
// decision, also reported by arb_step_plan.
struct BuildChoice { bool w3 = false, pack = false, rdv = false; long slots2 = 0, slots3 = 0, slotsp = 0; };
static BuildChoice choose_build(const arb_model *M, bool noopt, long nw, int nsteps, unsigned flags, bool bodyc = false) {
    BuildChoice bc;
    // (tiles of 44 and 48 rows.  The 16- and 32-row kernels use ~100 VGPRs less: their two-wave build has no spills and
    // measured faster than a three-wave build at every batch size -- simplearm, one world per wavefront: 101 against
    // 60 M world-steps/s; its forest of 10: 494 against 389 M at 65 536 worlds --, so they have no other)
    if (!((M->nsets == 1 || bodyc) && M->nmax >= 44 && M->nmax <= 48)) return bc;
    static thread_local int cus_dev = -1, cus = 0;
    if (cus_dev != M->device) { (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, M->device); cus_dev = M->device; }
    const long pad = std::max(0, M->kn.lds_pad);
    const long lds2 = (long)(bodyc ? M->lfb : M->lf).total * 4 + pad, lds3 = (long)(bodyc ? M->lfb3 : M->lf3).total * 4 + pad;
    const long s2 = (long)cus * slots_per_cu(2, lds2), s3 = (long)cus * slots_per_cu(3, lds3);
    bc.slots2 = s2; bc.slots3 = s3;
    // Two or three waves per SIMD?  Three when the batch fills the extra wave slots.  ARB_STEP_WAVES2 / ARB_STEP_WAVES3
    // pin the build; the knob "force_waves" = 2|3 overrides both (development).
    if (s3 > s2 && s2 > 0) {
        // measured (human36 + 4 contacts, M world-steps/s, two / three waves; end of round 3, twelve wavefronts per CU
        // for real): 2048 worlds 14.6 / 13.5, 2560: 16.7 / 13.9, 3072: 17.1 / 16.2, 3584: 16.8 / 18.0, 4096: 16.6 / 18.7,
        // 6144: 17.8 / 19.8, 8192: 17.8 / 20.1
        // one launch per step (no queue): three waves when they save a round of workgroups (4096 worlds: 10.4 / 9.7,
        // two rounds either way; 6144: 12.2 / 12.1)
        if (nsteps >= 2) bc.w3 = 10 * nw >= 11 * s3;
        else bc.w3 = 112 * ((nw + s3 - 1) / s3) < 100 * ((nw + s2 - 1) / s2);
        if (flags & ARB_STEP_WAVES2) bc.w3 = false;
        if (flags & ARB_STEP_WAVES3) bc.w3 = true;
    }
    const int force = M->kn.force_waves;
    if (force == 2) bc.w3 = false;
    if (force == 3) bc.w3 = true;
    // Rendezvous build (four worlds per wavefront in the sweeps, CM = 4): multi-step launches of models that qualify.
    // The knob "force_rdv" = 0|1 overrides (development).
    if (ARB_WITH_RDV && M->rdv_ok && !bodyc && noopt && nsteps >= 2 && !(flags & (ARB_STEP_WAVES2 | ARB_STEP_WAVES3 | ARB_STEP_STATIC_WORLDS))) {
        bc.rdv = ARB_RDV_DEFAULT != 0 && 10 * nw >= 11 * s3;
        const int fr = M->kn.force_rdv;
        if (fr == 0) bc.rdv = false;
        if (fr == 1) bc.rdv = true;
    }
    // Two worlds per wavefront (the packed build: the sweeps of both worlds in one instruction stream): models whose
    // constraints are all SoftFingerContacts with eps = (1,1,1), plain inputs or user torques, a stash that still leaves
    // eight wavefronts per CU, and a batch large enough that pairs of worlds fill and balance the wave slots (measured,
    // three-wave / packed: 4096 worlds 18.2 / 16.8, 8192: 19.3 / 19.4, 16384: 19.6 / 20.0, 65536: 20.0 / 20.4).
    // The knob "force_pack" = 0|1 overrides the batch-size rule (development).
    if (ARB_ALL_VARIANTS && M->packable && !bodyc && noopt && M->lfp.lscan) {
        const long ldsp = (long)M->lfp.total * 4 + pad;
        const long sp = (long)cus * slots_per_cu(2, ldsp);
        bc.slotsp = sp;
        // (end of round 3: with its LDS trimmed to the 1280-byte allocation granule the three-wave build really has twelve
        // wavefronts per CU -- eleven until then, whatever the occupancy API said -- and beats the packed build at every
        // batch size: 20.5 against 19.3 M world-steps/s at 16 384 worlds, 20.9 against 19.7 M at 65 536.  The packed
        // build stays in the library, bit-identical and tested, behind ARB_FORCE_PACK=1.)
        bc.pack = false;
        (void)ARB_PACK_MIN_ROUNDS;
        const int fp = M->kn.force_pack;
        if (fp == 0) bc.pack = false;
        if (fp == 1) bc.pack = true;
        if (flags & (ARB_STEP_WAVES2 | ARB_STEP_WAVES3)) bc.pack = false;      // (a pinned build is a pinned build)
    }
    return bc;
}

template <typename T, int MODE>
static int launch(arb_model *M, const DevModel<T> *dm, const Layout &L, T *q, T *dq, T *cf, const T *ext, const PerWorldPD<T> &pwd, long nw,
                  double dt, int nsteps, unsigned flags, const DebugOut<T> &dbg, int zmode, const LogOut<T> &logo,
                  const SplitIO<T> &sio, const double *dts, hipStream_t st, long ext_stride = 0, long pd_stride = 0,
                  const CostIO<T> &cost = CostIO<T>{nullptr, nullptr, nullptr, nullptr, nullptr}) {
    // the plain step (FEAT 0): nothing but the state and the constraint forces; FEAT 1: + user torques (MPC rollouts)
    const bool noopt = MODE == 0 && pwd.qdes == nullptr && pwd.kp == nullptr && logo.q == nullptr &&
                       logo.dq == nullptr && logo.energy == nullptr && sio.mode == 0 && !(flags & ARB_STEP_SKIP_CONSTRAINTS) && dts == nullptr;
    // (the kernels only look at ARB_STEP_SKIP_CONSTRAINTS; the other flags are for the host)
    const bool mfma = MODE == 0 && std::is_same<T, float>::value && (flags & ARB_STEP_MFMA_ELIM) &&
                      !(M->n == WAVE && M->nc == 0);          // (the late-rhs case is handled by the vector-ALU elimination)
    // body-space constraint columns (FEAT bit 16): every launch of a model of that class -- step, rollout, inspect -- except the split
    // execution, the matrix-core elimination and ARB_STEP_GENERAL_KERNELS, which run the general kernels on two column sets
    // (equal to rounding, not bit for bit: Y' is formed as T (J_p Y J_p^T) T^T instead of J' Y J'^T)
    const bool bodyc = ARB_WITH_SPEC && M->bodycols && (M->bodycols_default || (flags & ARB_STEP_BODY_COLUMNS)) && !mfma && sio.mode == 0 &&
                       !(flags & (ARB_STEP_GENERAL_KERNELS | ARB_STEP_SKIP_CONSTRAINTS));
    const BuildChoice bc = (MODE == 0 && std::is_same<T, float>::value && !mfma) ? choose_build(M, noopt, nw, nsteps, flags, bodyc) : BuildChoice();
    // (the packed and rendezvous builds -- libarbstep_variants.so -- know neither torque sequences nor the running cost)
    const bool seq = ext_stride != 0 || cost.out != nullptr;
    const bool w3 = bc.w3, pack = bc.pack && !seq, rdv = bc.rdv && !seq && cf != nullptr && nw * (long)nsteps < (1l << 30);
    // the kernels specialised for the model class "four plane / sphere SoftFingerContacts" (FEAT bit 4): bit-identical to the
    // general ones, which ARB_STEP_GENERAL_KERNELS selects
    // (float32 with one or two column sets, float64 with one)
    const bool spec = ARB_WITH_SPEC && M->spec_ok && noopt && MODE == 0 && !mfma && !pack && !rdv &&
                      (std::is_same<T, float>::value || M->nsets == 1) && !(flags & ARB_STEP_GENERAL_KERNELS);
    const bool spec0 = ARB_WITH_SPEC && M->spec0_ok && noopt && MODE == 0 && std::is_same<T, float>::value && !mfma && !pack && !rdv &&
                       !(flags & ARB_STEP_GENERAL_KERNELS);
    // (the running cost travels with the user torques: FEAT bit 0)
    const bool plain = noopt && ext == nullptr && cost.out == nullptr;
#if ARB_WITH_SPEC
#define ARB_SPEC0_CASE(NM) if (spec0) return w3 ? (plain ? ONE_(NM, 1, 8, 2) : ONE_(NM, 1, 9, 2)) : (plain ? ONE_(NM, 1, 8, 0) : ONE_(NM, 1, 9, 0));
#define ARB_SPEC_CASE(NM) ARB_SPEC0_CASE(NM) if (spec && M->nsets == 1) return w3 ? (plain ? ONE_(NM, 1, 4, 2) : ONE_(NM, 1, 5, 2)) : (plain ? ONE_(NM, 1, 4, 0) : ONE_(NM, 1, 5, 0));
#define ARB_SPEC_CASE2(NM)                                                                                             \
        if constexpr (MODE == 0 && NM >= 44 && NM <= 48) {                                                             \
            if (spec && M->nsets == 1 && !std::is_same<T, float>::value) return plain ? ONE(NM, 1, 4) : ONE(NM, 1, 5); \
        }
#define ARB_BODYC_CASE(NM)                                                                                             \
        if constexpr (NM >= 44 && NM <= 48) {                                                                          \
            if (bodyc) {                                                                                               \
                if constexpr (MODE == 1) return ONE(NM, 1, 19);                                                        \
                else if constexpr (std::is_same<T, float>::value) {                                                    \
                    if (w3) return plain ? ONE_(NM, 1, 20, 2) : noopt ? ONE_(NM, 1, 21, 2) : ONE_(NM, 1, 19, 2);       \
                    return plain ? ONE(NM, 1, 20) : noopt ? ONE(NM, 1, 21) : ONE(NM, 1, 19);                           \
                } else return plain ? ONE(NM, 1, 20) : noopt ? ONE(NM, 1, 21) : ONE(NM, 1, 19);                        \
            }                                                                                                          \
        }
#else
#define ARB_SPEC_CASE(NM) (void)spec; (void)spec0;
#define ARB_SPEC_CASE2(NM)
#define ARB_BODYC_CASE(NM) (void)bodyc;
#endif
#if ARB_WITH_RDV
#define ARB_RDV_CASE(NM) if (rdv) return plain ? ONE_(NM, 1, 0, 4) : ONE_(NM, 1, 1, 4);
#else
#define ARB_RDV_CASE(NM) (void)rdv;
#endif
#if ARB_ALL_VARIANTS
#define ARB_PACK_CASE(NM) if (pack) return plain ? ONE_(NM, 1, 0, 3) : ONE_(NM, 1, 1, 3);
#else
#define ARB_PACK_CASE(NM) (void)pack;
#endif
// (the instantiation must fit the model: a kernel with the wrong tile, column sets or model class computes on, silently wrong --
// round 4's first launch table sent an 8-contact model to the one-set specialised kernel; checked at every launch since)
#define ONE_(NM, NS, FT, CMV) (!(M->nmax == (NM) && (((FT) & 16) ? (M->bodycols && (NS) == 1) : (M->nsets == (NS) && (!((FT) & 4) || (M->spec_ok && M->nc == 4 * (NS))))) && (!((FT) & 8) || (M->spec0_ok && M->nc == 0))) ? (g_hip_err = "internal: kernel instantiation does not fit the model", (int)ARB_ERR_HIP) : launch_one<T, NM, NS, MODE, FT, CMV>(dm, ((FT) & 16) ? ((CMV) == 2 ? M->lfb3 : (std::is_same<T, float>::value ? M->lfb : M->ldb)) : (CMV) == 3 ? M->lfp : ((CMV) == 2 || (CMV) == 4) ? M->lf3 : L, q, dq, cf, ext, pwd, nw, dt, nsteps, flags, dbg, zmode, logo, sio, dts, st, M->kn, ext_stride, pd_stride, cost))
#define ONE(NM, NS, FT) ONE_(NM, NS, FT, 0)
#ifdef ARB_QUICK
    // development build: a single register tile (float, NMAX=44), the production kernels only (-DARB_QUICK=2: also
    // two column sets, the inspect kernel and the optional inputs)
#if ARB_QUICK == 4      /* development: the body-space-column kernels of the 44-row tile (float32 two / three waves + inspect, float64) */
    if (M->nmax == 44 && bodyc) {
        if constexpr (MODE == 1) { if constexpr (std::is_same<T, float>::value) return ONE(44, 1, 19); else return ARB_ERR_UNSUPPORTED; }
        else if constexpr (std::is_same<T, float>::value) { if (plain) return w3 ? ONE_(44, 1, 20, 2) : ONE(44, 1, 20); }
        else { if (plain) return ONE(44, 1, 20); }
    }
    (void)spec; (void)spec0;
    return ARB_ERR_UNSUPPORTED;
#elif ARB_QUICK == 3      /* the headline kernels only: the specialised float32 kernels, plain inputs, two and three waves (~1 min) */
    if constexpr (std::is_same<T, float>::value && MODE == 0) {
        if (M->nmax == 44 && M->nsets == 1 && spec && plain) return w3 ? ONE_(44, 1, 4, 2) : ONE_(44, 1, 4, 0);
    }
    return ARB_ERR_UNSUPPORTED;
#else
    if constexpr (std::is_same<T, float>::value) {
        if (M->nmax == 44 && M->nsets == 1) {
            if constexpr (MODE == 0) {
                ARB_RDV_CASE(44)
                ARB_PACK_CASE(44)
                ARB_SPEC_CASE(44)
                if (w3 && plain) return ONE_(44, 1, 0, 2);
                if (w3 && noopt) return ONE_(44, 1, 1, 2);
                if (plain) return ONE(44, 1, 0);
                if (noopt) return ONE(44, 1, 1);
#if ARB_QUICK >= 2
                return mfma ? ONE_(44, 1, 3, 1) : ONE(44, 1, 3);
#endif
            }
#if ARB_QUICK >= 2
            else return ONE(44, 1, 3);
#endif
        }
#if ARB_QUICK >= 2
        if (M->nmax == 44 && M->nsets == 2) {
#if ARB_WITH_SPEC
            if constexpr (MODE == 0) { if (spec) return plain ? ONE(44, 2, 4) : ONE(44, 2, 5); }
#endif
            if constexpr (MODE == 0) return plain ? ONE(44, 2, 0) : noopt ? ONE(44, 2, 1) : ONE(44, 2, 3);
            else return ONE(44, 2, 3);
        }
#endif
    }
    return ARB_ERR_UNSUPPORTED;
#endif
#else
#define CASE(NM)                                                                                       \
    case NM:                                                                                           \
        if constexpr (MODE == 0 && std::is_same<T, float>::value) {                                    \
            if (mfma) return (M->nsets == 2) ? ONE_(NM, 2, 3, 1) : ONE_(NM, 1, 3, 1);                  \
        }                                                                                              \
        ARB_BODYC_CASE(NM)                                                                             \
        if constexpr (MODE == 0 && std::is_same<T, float>::value && NM >= 44 && NM <= 48) {            \
            ARB_RDV_CASE(NM)                                                                           \
            ARB_PACK_CASE(NM)                                                                          \
            ARB_SPEC_CASE(NM)                                                                          \
            if (w3) return plain ? ONE_(NM, 1, 0, 2) : noopt ? ONE_(NM, 1, 1, 2) : ONE_(NM, 1, 3, 2);  \
        }                                                                                              \
        ARB_SPEC_CASE2(NM)                                                                             \
        if constexpr (MODE == 0) {                                                                     \
            if (plain) return (M->nsets == 2) ? ONE(NM, 2, 0) : ONE(NM, 1, 0);                         \
            if (noopt) return (M->nsets == 2) ? ONE(NM, 2, 1) : ONE(NM, 1, 1);                         \
        }                                                                                              \
        return (M->nsets == 2) ? ONE(NM, 2, 3) : ONE(NM, 1, 3);
    switch (M->nmax) {
        CASE(16) CASE(32) CASE(44) CASE(48) CASE(64)
        default: return ARB_ERR_UNSUPPORTED;
    }
#undef CASE
#endif
#undef ONE
#undef ONE_
}

template <typename T>
static int launch_gsw(const DevModel<T> *dm, int nc, const SplitIO<T> &sio, long nw, double dt, const double *dts, hipStream_t st, int wv, bool pack = false, bool pack4 = false) {
    auto al = [](int x) { return (x + 3) & ~3; };
    const int ndol = 4 * nc;
    const size_t lds = (size_t)(al(ndol * ndol) + al(nc * CD_STRIDE) + 2 * al(ndol) + 64) * sizeof(T);
    // wv: waves per SIMD the sweep kernel is compiled for (knob "gsw_waves": 3 = no spills, 4 = 128 VGPRs)
    if (lds > 64 * 1024) return ARB_ERR_UNSUPPORTED;
    if (pack) {
        // (development: two or four worlds per wavefront; the caller has checked that the model qualifies -- four need
        // their 4 nc rows to fit a quarter of the wavefront)
        const int ng = (pack4 && ndol <= 16) ? 4 : 2;
        const size_t ldsn = (size_t)(ng * (al(ndol * ndol) + al(nc * CD_STRIDE) + 2 * al(ndol)) + 64) * sizeof(T);
        const unsigned grid = (unsigned)((nw + ng - 1) / ng);
        // (the packed sweeps keep every stage's results lane by lane: compiled for three waves per SIMD they spill ~50
        // registers inside the solve -- gsw_waves = 2 selects the 256-register build)
        if (ng == 4 && wv == 2)
            hipLaunchKernelGGL((arb_gswn_kernel<T, 2, 4>), dim3(grid), dim3(WAVE), ldsn, st, dm, sio.A, sio.v, sio.f, sio.c, nw, (T)dt, dts);
        else if (ng == 4)
            hipLaunchKernelGGL((arb_gswn_kernel<T, 3, 4>), dim3(grid), dim3(WAVE), ldsn, st, dm, sio.A, sio.v, sio.f, sio.c, nw, (T)dt, dts);
        else if (wv == 2)
            hipLaunchKernelGGL((arb_gswn_kernel<T, 2, 2>), dim3(grid), dim3(WAVE), ldsn, st, dm, sio.A, sio.v, sio.f, sio.c, nw, (T)dt, dts);
        else
            hipLaunchKernelGGL((arb_gswn_kernel<T, 3, 2>), dim3(grid), dim3(WAVE), ldsn, st, dm, sio.A, sio.v, sio.f, sio.c, nw, (T)dt, dts);
        HIP_TRY(hipGetLastError());
        return ARB_OK;
    }
    if (wv == 4)
        hipLaunchKernelGGL((arb_gsw_kernel<T, 4>), dim3((unsigned)nw), dim3(WAVE), lds, st, dm, sio.A, sio.v, sio.f, sio.c, nw, (T)dt, dts);
    else
        hipLaunchKernelGGL((arb_gsw_kernel<T, 3>), dim3((unsigned)nw), dim3(WAVE), lds, st, dm, sio.A, sio.v, sio.f, sio.c, nw, (T)dt, dts);
    HIP_TRY(hipGetLastError());
    return ARB_OK;
}

template <typename T>
static int step_typed(arb_model *M, const DevModel<T> *dm, const Layout &L, T *q, T *dq, T *cf, const T *ext,
                      const PerWorldPD<T> &pwd, long nw, double dt, const double *dts, int nsteps, unsigned flags,
                      const arb_rollout_log *log, hipStream_t st, long ext_stride, long pd_stride, const CostIO<T> &cost) {
    DebugOut<T> dbg; memset(&dbg, 0, sizeof(dbg));
    LogOut<T> lo; memset(&lo, 0, sizeof(lo));
    if (log) { lo.q = (T *)log->q_log; lo.dq = (T *)log->dq_log; lo.energy = (T *)log->energy_log; }
    SplitIO<T> sio; memset(&sio, 0, sizeof(sio));
    const int nc = M->nc, ndol = M->ndol, n = M->n;
    const bool split = nc > 0 && (flags & ARB_STEP_SPLIT_WAVE) && !(flags & (ARB_STEP_SKIP_CONSTRAINTS | ARB_STEP_FUSED));
    if (!split)
        return launch<T, 0>(M, dm, L, q, dq, cf, ext, pwd, nw, dt, nsteps, flags, dbg, 0, lo, sio, dts, st, ext_stride, pd_stride, cost);
    if (cost.out != nullptr) return ARB_ERR_INVALID;        // (the split execution integrates a step in the NEXT launch: no running cost)
    // ---- split execution (opt-in): step kernel (dynamics + system) / Gauss-Seidel kernel (one wavefront per world) ----
    // The hand-over buffers are allocated per call in stream order and freed in stream order after the last launch:
    // no per-handle state, so a handle may run split steps on several streams at once.
    const size_t per_world = (size_t)ndol * ndol + 3 * (size_t)ndol + 8 * (size_t)nc + (size_t)(1 + ndol) * n;
    const size_t need = per_world * (size_t)nw * sizeof(T);
    void *ws = nullptr;
    HIP_TRY(arb_scratch_alloc(&ws, need, st));
    T *p = (T *)ws;
    sio.A = p; p += (size_t)nw * ndol * ndol;
    sio.v = p; p += (size_t)nw * ndol;
    sio.f = p; p += (size_t)nw * ndol;
    sio.f0 = p; p += (size_t)nw * ndol;
    sio.c = p; p += (size_t)nw * nc * 8;
    sio.sol = p;
    int rc = ARB_OK;
    for (int k = 0; k < nsteps && rc == ARB_OK; ++k) {
        LogOut<T> lk = lo;
        if (lk.q) lk.q += (size_t)k * nw * M->nq;
        if (lk.dq) lk.dq += (size_t)k * nw * n;
        if (lk.energy) lk.energy += (size_t)k * nw * 2;
        sio.mode = 2 | (k > 0 ? 1 : 0);
        // (kernel k finishes step k-1 with dts[k-1], then builds step k with dts[k]; a control sequence: row k)
        PerWorldPD<T> pk = pwd;
        if (pk.qdes != nullptr) { pk.qdes += (size_t)k * pd_stride; pk.dqdes += (size_t)k * pd_stride; }
        rc = launch<T, 0>(M, dm, L, q, dq, cf, ext ? ext + (size_t)k * ext_stride : nullptr, pk, nw, dt, 1, flags, dbg, 0, lk, sio, dts ? dts + k : nullptr, st);
        if (rc == ARB_OK) rc = launch_gsw<T>(dm, nc, sio, nw, dt, dts ? dts + k : nullptr, st, M->kn.gsw_waves, M->packable && M->kn.gsw_pack != 0, M->kn.gsw_pack == 4);
    }
    if (rc == ARB_OK) {
        sio.mode = 1;                                      // apply the last step's forces, write cforce
        LogOut<T> nolog; memset(&nolog, 0, sizeof(nolog));
        rc = launch<T, 0>(M, dm, L, q, dq, cf, ext, pwd, nw, dt, 1, flags, dbg, 0, nolog, sio, dts ? dts + nsteps : nullptr, st);
    }
    (void)hipFreeAsync(ws, st);
    return rc;
}

// Does a launch run on the forest of a small model (arb_model::forest, ARB_STEP_ONE_WORLD)?  When the batch is larger than
// twice the wave slots of the device -- below that every world has a wavefront to itself anyway and the forest's larger tile
// only lengthens the step --, and when its logs keep their layout: state logs [step][world][..] of a batch that is a
// multiple of k are the forest's logs; energies are per world, which a forest world does not have.
static int device_cus(int device) {
    static thread_local int cus_dev = -1, cus = 0;
    if (cus_dev != device) { (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device); cus_dev = device; }
    return cus;
}
static bool use_forest(const arb_model *M, int64_t nworlds, uint32_t flags, const arb_rollout_log *log, bool per_world_out = false) {
    if (!M->forest || (flags & (ARB_STEP_ONE_WORLD | ARB_STEP_SPLIT_WAVE | ARB_STEP_MFMA_ELIM))) return false;
    if (per_world_out) return false;                   // (a running cost is per world, like the energies)
    if (log && (log->energy_log || ((log->q_log || log->dq_log) && nworlds % M->forest_k != 0))) return false;
    return nworlds > 16l * device_cus(M->device);      // (measured, simplearm: 4096 worlds 98 alone / 91 M as a forest, 8192: 100 / 181)
}

// ext_stride / pd_stride: elements between the rows of consecutive steps of a control sequence (0: one row for the launch)
static int step_impl(arb_model *M, int dtype, void *q, void *dq, void *cforce, const void *ext_gforce,
                     const void *pd_qdes, const void *pd_dqdes, const void *pd_kp, const void *pd_kd,
                     int64_t nworlds, double dt, const double *dt_steps, int32_t nsteps, uint32_t flags,
                     const arb_rollout_log *log, void *stream, long ext_stride = 0, long pd_stride = 0,
                     const arb_step_cost *cost = nullptr) {
    if (!M || nworlds < 0 || nsteps < 0) return ARB_ERR_INVALID;
    if (dt_steps == nullptr && !(dt > 0.0)) return ARB_ERR_INVALID;
    if (dt_steps != nullptr) dt = 1.0;                  // unused: every step reads its own dt
    if (dtype != ARB_F32 && dtype != ARB_F64) return ARB_ERR_INVALID;
    if (flags & ~ARB_STEP_KNOWN_FLAGS) return ARB_ERR_INVALID;     // (4u was ARB_STEP_SPLIT up to ABI 4)
    // per-world PD inputs: targets come in pairs; diagonal gains come in pairs and need targets;
    // targets without gains use the model's gain matrices, so the model must hold a PD controller
    if ((pd_qdes == nullptr) != (pd_dqdes == nullptr) || (pd_kp == nullptr) != (pd_kd == nullptr)) return ARB_ERR_INVALID;
    if (pd_kp != nullptr && pd_qdes == nullptr) return ARB_ERR_INVALID;
    if (pd_qdes != nullptr && pd_kp == nullptr && !M->df.has_pd) return ARB_ERR_INVALID;
    if (cost != nullptr && cost->cost_out == nullptr) return ARB_ERR_INVALID;
    if (nworlds == 0 || nsteps == 0) return ARB_OK;     // empty batch: nothing to do (pointers may be null)
    if (!q || !dq) return ARB_ERR_INVALID;
    if (nworlds > 0x7fffffffLL) return ARB_ERR_INVALID;
    if (int stalled = take_status(M, false)) return stalled;
    ARB_GUARD_DEVICE(M->device);
    if (use_forest(M, nworlds, flags, log, cost != nullptr)) {
        // small worlds share wavefronts: nworlds / k worlds of the forest on the same buffers, the rest one per wavefront
        // (a control sequence keeps its row stride: the rows of a step are the whole batch's)
        const int K = M->forest_k;
        const int64_t nf = nworlds / K, done = nf * K;
        const size_t es = dtype == ARB_F32 ? sizeof(float) : sizeof(double);
        int rc = step_impl(M->forest, dtype, q, dq, cforce, ext_gforce, pd_qdes, pd_dqdes, pd_kp, pd_kd, nf, dt, dt_steps, nsteps,
                           flags | ARB_STEP_ONE_WORLD, log, stream, ext_stride, pd_stride);
        if (rc != ARB_OK || done == nworlds) return rc;
        auto at = [&](const void *p, size_t per_world) -> void * {
            return p ? (void *)((const char *)p + (size_t)done * per_world * es) : nullptr;
        };
        return step_impl(M, dtype, at(q, M->nq), at(dq, M->n), at(cforce, (size_t)M->nc * ARB_MAXDOL), at(ext_gforce, M->n),
                         at(pd_qdes, M->n), at(pd_dqdes, M->n), at(pd_kp, M->n), at(pd_kd, M->n), nworlds - done, dt, dt_steps,
                         nsteps, flags | ARB_STEP_ONE_WORLD, nullptr, stream, ext_stride, pd_stride);
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == ARB_F32) {
        const PerWorldPD<float> pwd = {(const float *)pd_qdes, (const float *)pd_dqdes, (const float *)pd_kp, (const float *)pd_kd};
        CostIO<float> ci = {nullptr, nullptr, nullptr, nullptr, nullptr};
        if (cost) ci = CostIO<float>{(float *)cost->cost_out, (const float *)cost->w_q, (const float *)cost->w_dq, (const float *)cost->w_tau, (const float *)cost->q_ref};
        return step_typed<float>(M, M->df_dev, M->lf, (float *)q, (float *)dq, (float *)cforce,
                                 (const float *)ext_gforce, pwd, (long)nworlds, dt, dt_steps, nsteps, flags, log, st, ext_stride, pd_stride, ci);
    }
    const PerWorldPD<double> pwd = {(const double *)pd_qdes, (const double *)pd_dqdes, (const double *)pd_kp, (const double *)pd_kd};
    CostIO<double> ci = {nullptr, nullptr, nullptr, nullptr, nullptr};
    if (cost) ci = CostIO<double>{(double *)cost->cost_out, (const double *)cost->w_q, (const double *)cost->w_dq, (const double *)cost->w_tau, (const double *)cost->q_ref};
    return step_typed<double>(M, M->dd_dev, M->ld, (double *)q, (double *)dq, (double *)cforce,
                              (const double *)ext_gforce, pwd, (long)nworlds, dt, dt_steps, nsteps, flags, log, st, ext_stride, pd_stride, ci);
}

extern "C" int arb_step(arb_model *M, int dtype, void *q, void *dq, void *cforce, const void *ext_gforce,
                        int64_t nworlds, double dt, int32_t nsteps, uint32_t flags, void *stream) {
    return step_impl(M, dtype, q, dq, cforce, ext_gforce, nullptr, nullptr, nullptr, nullptr, nworlds, dt, nullptr, nsteps,
                     flags, nullptr, stream);
}

extern "C" int arb_step_plan(arb_model *M, int dtype, int64_t nworlds, int32_t nsteps, uint32_t flags, int32_t optional_inputs,
                             arb_step_plan_info *out) {
    if (!M || !out || nworlds < 0 || nsteps < 0 || (dtype != ARB_F32 && dtype != ARB_F64)) return ARB_ERR_INVALID;
    if (flags & ~ARB_STEP_KNOWN_FLAGS) return ARB_ERR_INVALID;
    ARB_GUARD_DEVICE(M->device);
    if (optional_inputs < 0 || optional_inputs > 7 || (optional_inputs & 3) == 2) return ARB_ERR_INVALID;
    const bool world_logs = (optional_inputs & 4) != 0;      // per-world energies / costs, or state logs of a ragged batch: no forest
    optional_inputs &= 3;
    if (!world_logs && use_forest(M, nworlds, flags, nullptr)) {
        const int rc = arb_step_plan(M->forest, dtype, nworlds / M->forest_k, nsteps, flags | ARB_STEP_ONE_WORLD, optional_inputs, out);
        if (rc == ARB_OK) out->worlds_per_wavefront = M->forest_k;
        return rc;
    }
    memset(out, 0, sizeof(*out));
    const bool split = M->nc > 0 && (flags & ARB_STEP_SPLIT_WAVE) && !(flags & (ARB_STEP_SKIP_CONSTRAINTS | ARB_STEP_FUSED));
    const bool noopt = optional_inputs <= 1 && !split && !(flags & ARB_STEP_SKIP_CONSTRAINTS);
    const bool mfma = dtype == ARB_F32 && (flags & ARB_STEP_MFMA_ELIM) && !(M->n == WAVE && M->nc == 0);
    BuildChoice bc;
    const bool bodyc = ARB_WITH_SPEC && M->bodycols && (M->bodycols_default || (flags & ARB_STEP_BODY_COLUMNS)) && !mfma && !split &&
                       !(flags & (ARB_STEP_GENERAL_KERNELS | ARB_STEP_SKIP_CONSTRAINTS));
    if (dtype == ARB_F32 && !mfma) bc = choose_build(M, noopt, (long)nworlds, split ? 1 : nsteps, flags, bodyc);
    out->worlds_per_wavefront = bc.pack ? 2 : 1;
    // (the float64 64-row kernels may use the whole register file of a SIMD: those with two column sets do -- one
    // wavefront per SIMD --, those with one fit 256 registers)
    out->waves_per_simd = (dtype == ARB_F64 && M->nmax == 64 && M->nsets == 2 && !bodyc) ? 1 : (bc.w3 && !bc.pack) ? 3 : 2;
    const Layout &L = bodyc ? (dtype == ARB_F64 ? M->ldb : bc.w3 ? M->lfb3 : M->lfb) : dtype == ARB_F64 ? M->ld : bc.pack ? M->lfp : bc.w3 ? M->lf3 : M->lf;
    out->lds_bytes = L.total * (dtype == ARB_F64 ? 8 : 4);
    int cus = 0;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, M->device);
    out->wave_slots = (int32_t)(cus * slots_per_cu(out->waves_per_simd, (long)out->lds_bytes + std::max(0, M->kn.lds_pad)));   // (the launch's own model: slots_per_cu)
    const long units = bc.pack ? (nworlds + 1) / 2 : nworlds;
    out->work_queue = (!split && nsteps >= 2 && !(flags & ARB_STEP_STATIC_WORLDS) && units > out->wave_slots &&
                       M->kn.queue_chunk > 0) ? 1 : 0;
    out->feat = optional_inputs <= 0 ? 0 : optional_inputs == 1 ? 1 : 3;
    if (!noopt) out->feat = 3;
    if (bodyc) out->feat |= noopt ? 20 : 16;       // (4 | 16: the specialised kernels with body-space columns; 19: every optional input)
    // (the specialised kernels, see launch(): plain inputs or user torques of a model of their class, float32)
    if (ARB_WITH_SPEC && M->spec_ok && noopt && (dtype == ARB_F32 || M->nsets == 1) && !mfma && !bc.pack && !bc.rdv && !(flags & ARB_STEP_GENERAL_KERNELS)) out->feat |= 4;
    if (ARB_WITH_SPEC && M->spec0_ok && noopt && dtype == ARB_F32 && !mfma && !bc.pack && !bc.rdv && !(flags & ARB_STEP_GENERAL_KERNELS)) out->feat |= 8;
    return ARB_OK;
}

extern "C" int arb_step_ex(arb_model *M, int dtype, const arb_step_args *a, void *stream) {
    if (!a || !M) return ARB_ERR_INVALID;
    // control sequences (ABI 7): [nsteps][nworlds][ndof] arrays, one row per step
    if (a->ext_gforce_steps != nullptr && a->ext_gforce != nullptr) return ARB_ERR_INVALID;
    if ((a->pd_qdes_steps == nullptr) != (a->pd_dqdes_steps == nullptr)) return ARB_ERR_INVALID;
    if (a->pd_qdes_steps != nullptr && a->pd_qdes != nullptr) return ARB_ERR_INVALID;
    if (a->nworlds < 0) return ARB_ERR_INVALID;
    const long row = (long)a->nworlds * M->n;
    const void *ext = a->ext_gforce_steps ? a->ext_gforce_steps : a->ext_gforce;
    const void *qdes = a->pd_qdes_steps ? a->pd_qdes_steps : a->pd_qdes, *dqdes = a->pd_dqdes_steps ? a->pd_dqdes_steps : a->pd_dqdes;
    return step_impl(M, dtype, a->q, a->dq, a->cforce, ext, qdes, dqdes, a->pd_kp, a->pd_kd,
                     a->nworlds, a->dt, a->dt_steps, a->nsteps, a->flags, a->log, stream,
                     a->ext_gforce_steps ? row : 0l, a->pd_qdes_steps ? row : 0l, a->cost);
}

extern "C" int arb_rollout(arb_model *M, int dtype, void *q, void *dq, void *cforce, const void *ext_gforce,
                           int64_t nworlds, double dt, int32_t nsteps, uint32_t flags,
                           const arb_rollout_log *log, void *stream) {
    if (!log) return ARB_ERR_INVALID;
    return step_impl(M, dtype, q, dq, cforce, ext_gforce, nullptr, nullptr, nullptr, nullptr, nworlds, dt, nullptr, nsteps,
                     flags, log, stream);
}

template <typename T>
static int inspect_t(arb_model *M, const DevModel<T> *dm, const Layout &L, const void *q, const void *dq,
                     const void *cforce, const void *ext, long nw, double dt, unsigned flags,
                     const arb_inspect_out *o, hipStream_t st) {
    PerWorldPD<T> pwd; memset(&pwd, 0, sizeof(pwd));
    DebugOut<T> dbg; memset(&dbg, 0, sizeof(dbg));
    int rc;
    // the three world matrices need one pass each (they share the accumulator registers)
    struct { void *ptr; int zmode; } passes[3] = {{o->M, 1}, {o->B, 2}, {o->N, 3}};
    for (auto &ps : passes) {
        if (!ps.ptr) continue;
        DebugOut<T> d1; memset(&d1, 0, sizeof(d1));
        d1.Zout = (T *)ps.ptr;
        LogOut<T> nolog; memset(&nolog, 0, sizeof(nolog));
        SplitIO<T> nosplit; memset(&nosplit, 0, sizeof(nosplit));
        rc = launch<T, 1>(M, dm, L, (T *)q, (T *)dq, (T *)cforce, (const T *)ext, pwd, nw, dt, 1, flags, d1, ps.zmode, nolog, nosplit, nullptr, st);
        if (rc != ARB_OK) return rc;
    }
    dbg.pose = (T *)o->pose; dbg.twist = (T *)o->twist; dbg.jac = (T *)o->jac; dbg.djac = (T *)o->djac;
    dbg.Zout = (T *)o->Z; dbg.gforce0 = (T *)o->gforce0; dbg.vel_free = (T *)o->vel_free;
    dbg.c_sdist = (T *)o->c_sdist; dbg.c_active = (int *)o->c_active; dbg.c_jac = (T *)o->c_jac;
    dbg.c_force = (T *)o->c_force; dbg.c_frame = (T *)o->c_frame; dbg.gforce = (T *)o->gforce;
    dbg.q_next = (T *)o->q_next; dbg.dq_next = (T *)o->dq_next; dbg.gs_stats = (int *)o->gs_stats; dbg.gs_trace = (int *)o->gs_trace; dbg.stamps = (long long *)o->stamps; dbg.energy = (T *)o->energy;
    dbg.c_adm = (T *)o->c_adm; dbg.c_vel = (T *)o->c_vel;
    dbg.ablate = M->kn.ablate; dbg.pivot_growth = (T *)o->pivot_growth;
    LogOut<T> nolog; memset(&nolog, 0, sizeof(nolog));
    SplitIO<T> nosplit; memset(&nosplit, 0, sizeof(nosplit));
    return launch<T, 1>(M, dm, L, (T *)q, (T *)dq, (T *)cforce, (const T *)ext, pwd, nw, dt, 1, flags, dbg, 0, nolog, nosplit, nullptr, st);
}

extern "C" int arb_inspect(arb_model *M, int dtype, const void *q, const void *dq, const void *cforce,
                           const void *ext_gforce, int64_t nworlds, double dt, uint32_t flags,
                           const arb_inspect_out *out, void *stream) {
    if (!M || !out || nworlds < 0 || !(dt > 0.0)) return ARB_ERR_INVALID;
    if (dtype != ARB_F32 && dtype != ARB_F64) return ARB_ERR_INVALID;
    if (out->gforce != nullptr && M->nc > 0 && out->c_jac == nullptr) return ARB_ERR_INVALID;
    if (nworlds == 0) return ARB_OK;
    if (!q || !dq) return ARB_ERR_INVALID;
    if (nworlds > 0x7fffffffLL) return ARB_ERR_INVALID;
    if (int stalled = take_status(M, false)) return stalled;
    ARB_GUARD_DEVICE(M->device);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == ARB_F32)
        return inspect_t<float>(M, M->df_dev, M->lf, q, dq, cforce, ext_gforce, (long)nworlds, dt, flags, out, st);
    return inspect_t<double>(M, M->dd_dev, M->ld, q, dq, cforce, ext_gforce, (long)nworlds, dt, flags, out, st);
}

// ---------------------------------------------------------------------------
// Host-side self-test hooks: the device math of arb_math.h compiled for the host
// (no GPU needed).  Used by the CPU test-suite to check the hand-written small
// solvers (4x4 GEPP, 6x6 eigenvalues, SoftFingerContact.solve) against captured
// reference tuples.
// ---------------------------------------------------------------------------
// dtype: ARB_F32 / ARB_F64, optionally | 0x100 to force the generic eig6 route of the sliding branch
extern "C" int arb_host_softfinger_solve(int dtype, const double *vel, const double *adm, double *force,
                                         double sdist, double dt, double mu, const double *eps, double *dforce) {
    if (!vel || !adm || !force || !eps || !dforce) return -1;
    const bool use_fast = !(dtype & 0x100);
    dtype &= 0xff;
    if (dtype == ARB_F64) {
        double P[16], work[36], f[4], df[4];
        if (!inv_block<double>(adm, 4, 4, P)) pinv_block<double>(adm, 4, 4, P);
        for (int i = 0; i < 4; ++i) f[i] = force[i];
        int br = softfinger_solve<double>(vel, adm, P, f, df, sdist, dt, mu, eps, work, use_fast);
        for (int i = 0; i < 4; ++i) { force[i] = f[i]; dforce[i] = df[i]; }
        return br;
    }
    float v[4], Y[16], P[16], work[36], f[4], df[4], e[3];
    for (int i = 0; i < 4; ++i) { v[i] = (float)vel[i]; f[i] = (float)force[i]; }
    for (int i = 0; i < 16; ++i) Y[i] = (float)adm[i];
    for (int i = 0; i < 3; ++i) e[i] = (float)eps[i];
    if (!inv_block<float>(Y, 4, 4, P)) pinv_block<float>(Y, 4, 4, P);
    int br = softfinger_solve<float>(v, Y, P, f, df, (float)sdist, (float)dt, (float)mu, e, work, use_fast);
    for (int i = 0; i < 4; ++i) { force[i] = f[i]; dforce[i] = df[i]; }
    return br;
}

// the same solve executed on the device, one lane per tuple (see arb_softfinger_test_kernel); `dtype` as above
extern "C" int arb_dev_softfinger_solve(int dtype, int device, int n, const double *in /*[n][27]*/, double *out /*[n][9]*/) {
    if (!in || !out || n <= 0) return ARB_ERR_INVALID;
    const int use_fast = !(dtype & 0x100);
    dtype &= 0xff;
    ARB_GUARD_DEVICE(device);
    double *din = nullptr, *dout = nullptr;
    HIP_TRY(hipMalloc(&din, sizeof(double) * 27 * (size_t)n));
    HIP_TRY(hipMalloc(&dout, sizeof(double) * 9 * (size_t)n));
    HIP_TRY(hipMemcpy(din, in, sizeof(double) * 27 * (size_t)n, hipMemcpyHostToDevice));
    const unsigned grid = (unsigned)((n + WAVE - 1) / WAVE);
    if (dtype == ARB_F64)
        hipLaunchKernelGGL(arb_softfinger_test_kernel<double>, dim3(grid), dim3(WAVE), WAVE * 41 * sizeof(double), 0, din, dout, n, use_fast);
    else
        hipLaunchKernelGGL(arb_softfinger_test_kernel<float>, dim3(grid), dim3(WAVE), WAVE * 41 * sizeof(float), 0, din, dout, n, use_fast);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(out, dout, sizeof(double) * 9 * (size_t)n, hipMemcpyDeviceToHost));
    (void)hipFree(din); (void)hipFree(dout);
    return ARB_OK;
}

// build variants compiled into this library: bit 0 packed pairs (ARB_FORCE_PACK=1), bit 1 the rendezvous build (ARB_FORCE_RDV=1),
// bit 2 sweeps without the fast variant of the local solve (-DARB_GS_FAST=0)
extern "C" int arb_build_variants(void) { return (ARB_ALL_VARIANTS ? 1 : 0) | (ARB_WITH_RDV ? 2 : 0) | (ARB_GS_FAST ? 0 : 4) | (ARB_WITH_SPEC ? 0 : 8); }

// eig6 (one lane, matrix in LDS) and eig6_wave (the whole wavefront) on the same matrices, see arb_eig6_test_kernel
extern "C" int arb_dev_eig6_pair(int dtype, int device, int n, const double *A /*[n][36]*/, double *out /*[n][28]*/) {
    if (!A || !out || n <= 0 || (dtype != ARB_F32 && dtype != ARB_F64)) return ARB_ERR_INVALID;
    ARB_GUARD_DEVICE(device);
    if (n > (1 << 20)) return ARB_ERR_INVALID;          // (a test hook: one workgroup per matrix)
    double *din = nullptr, *dout = nullptr;
    auto body = [&]() -> int {
        HIP_TRY(hipMalloc(&din, sizeof(double) * 36 * (size_t)n));
        HIP_TRY(hipMalloc(&dout, sizeof(double) * 28 * (size_t)n));
        HIP_TRY(hipMemcpy(din, A, sizeof(double) * 36 * (size_t)n, hipMemcpyHostToDevice));
        HIP_TRY(hipMemset(dout, 0, sizeof(double) * 28 * (size_t)n));
        if (dtype == ARB_F64)
            hipLaunchKernelGGL(arb_eig6_test_kernel<double>, dim3((unsigned)n), dim3(WAVE), 96 * sizeof(double), 0, din, dout, n);
        else
            hipLaunchKernelGGL(arb_eig6_test_kernel<float>, dim3((unsigned)n), dim3(WAVE), 96 * sizeof(float), 0, din, dout, n);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipDeviceSynchronize());                 // (the kernel's own execution status, before the buffers go)
        HIP_TRY(hipMemcpy(out, dout, sizeof(double) * 28 * (size_t)n, hipMemcpyDeviceToHost));
        return ARB_OK;
    };
    const int rc = body();
    if (din) (void)hipFree(din);
    if (dout) (void)hipFree(dout);
    return rc;
}

// raw branch code of softfinger_try (3 = the fast sliding-shift path declined and eig6 is needed)
extern "C" int arb_host_softfinger_try(int dtype, const double *vel, const double *adm, const double *force,
                                       double sdist, double dt, double mu, const double *eps) {
    if (dtype == ARB_F64) {
        double P[16], work[36], f[4], df[4], alpha[4], s = 0;
        inv_block<double>(adm, 4, 4, P);
        for (int i = 0; i < 4; ++i) f[i] = force[i];
        return softfinger_try<double>(vel, adm, P, f, df, sdist, dt, mu, eps, work, alpha, &s);
    }
    float v[4], Y[16], P[16], work[36], f[4], df[4], e[3], alpha[4], s = 0;
    for (int i = 0; i < 4; ++i) { v[i] = (float)vel[i]; f[i] = (float)force[i]; }
    for (int i = 0; i < 16; ++i) Y[i] = (float)adm[i];
    for (int i = 0; i < 3; ++i) e[i] = (float)eps[i];
    inv_block<float>(Y, 4, 4, P);
    return softfinger_try<float>(v, Y, P, f, df, (float)sdist, (float)dt, (float)mu, e, work, alpha, &s);
}

// leftmost real root of the sliding-branch sextic for the admittance block Y (4x4, row-major);
// returns 1 on success.  `warm` may be NaN.
extern "C" int arb_host_slide_root(const double *Y, double c1, double kappa, double warm, double *root) {
    const SlidePre k = slide_precompute<double>(Y);
    return slide_leftmost_root(k, c1, kappa, warm, root) ? 1 : 0;
}

// the derivative cascade on a sextic given by its seven coefficients (constant term first): the leftmost real root in
// [lo, 0]; returns 1 with *root, 0 when there is none, -1 for non-finite input
extern "C" int arb_host_real_root_cascade(const double *pc, double lo, double *root) {
    if (!pc || !root) return -1;
    return slide_real_root_cascade(pc, lo, root);
}

// (pseudo-)inverse of an nd x nd block as the kernels form it: returns 1 when the pivoted elimination was kept,
// 0 when the block was found rank deficient and the SVD route (numpy.linalg.pinv semantics) was taken
extern "C" int arb_host_block_pinv(int dtype, int nd, const double *Y /*[nd][nd]*/, double *P /*[nd][nd]*/) {
    if (!Y || !P || nd < 1 || nd > 4) return -1;
    int regular;
    double out[16];
    if (dtype == ARB_F64) {
        double p[16];
        regular = inv_block<double>(Y, nd, nd, p);
        if (!regular) pinv_block<double>(Y, nd, nd, p);
        for (int i = 0; i < 16; ++i) out[i] = p[i];
    } else {
        float y[16], p[16];
        for (int i = 0; i < nd * nd; ++i) y[i] = (float)Y[i];
        regular = inv_block<float>(y, nd, nd, p);
        if (!regular) pinv_block<float>(y, nd, nd, p);
        for (int i = 0; i < 16; ++i) out[i] = p[i];
    }
    for (int i = 0; i < nd; ++i) for (int j = 0; j < nd; ++j) P[i * nd + j] = out[4 * i + j];
    return regular ? 1 : 0;
}

extern "C" int arb_host_eig6(const double *A, double *wr, double *wi) {
    double a[36];
    for (int i = 0; i < 36; ++i) a[i] = A[i];
    return eig6<double>(a, wr, wi);
}

extern "C" int arb_host_joint_local(int jt, const double *q, const double *dq, double *out /*[9+3+9+9+6]*/) {
    JointLocal<double> jl;
    joint_local<double>(jt, q, dq, jl);
    for (int i = 0; i < 9; ++i) out[i] = jl.R.a[i];
    out[9] = jl.p.x; out[10] = jl.p.y; out[11] = jl.p.z;
    for (int i = 0; i < 3; ++i) { out[12 + 3 * i] = jl.jw[i].x; out[13 + 3 * i] = jl.jw[i].y; out[14 + 3 * i] = jl.jw[i].z; }
    for (int i = 0; i < 3; ++i) { out[21 + 3 * i] = jl.djw[i].x; out[22 + 3 * i] = jl.djw[i].y; out[23 + 3 * i] = jl.djw[i].z; }
    out[30] = jl.Tw.x; out[31] = jl.Tw.y; out[32] = jl.Tw.z; out[33] = jl.Tv.x; out[34] = jl.Tv.y; out[35] = jl.Tv.z;
    return 0;
}

extern "C" int arb_host_exp_twist(const double *tw, double *H /*16*/) {
    M3<double> R; V3<double> p;
    exp_twist<double>(v3<double>(tw[0], tw[1], tw[2]), v3<double>(tw[3], tw[4], tw[5]), R, p);
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) H[4 * i + j] = R.a[3 * i + j]; }
    H[3] = p.x; H[7] = p.y; H[11] = p.z; H[12] = H[13] = H[14] = 0; H[15] = 1;
    return 0;
}

#endif  // !ARB_PART
