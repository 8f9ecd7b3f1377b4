// arb_wide_kernel.h -- the WIDE step kernel (round 6): worlds of more than 64 dofs or bodies, one WORKGROUP of 256 lanes per
// world (included by arb_kernels.hip only).
//
// The reference allocates any number of dofs (core.py:608-635; robots/snake.py:17-60 takes any n; human36 beside a few
// free objects is past 64).  The step kernel of arb_step_kernel.h holds a world in ONE wavefront -- column per lane, the
// augmented system in registers -- which ends at 64 dofs.  This kernel takes over above that: the same phases and the same
// formulas (world-frame composite assembly of Z, increment form, pivot-free Gauss-Jordan from the last dof to the first,
// 20 Gauss-Seidel sweeps with the local solves of arb_math.h):
//   * float64 arithmetic whatever the buffers' type (T is only the type of the caller's state buffers);
//   * lane = body / dof / constraint / column as the phase needs, __syncthreads() between phases; workgroups loop over the worlds
//     of the batch, steps loop inside;
//   * two builds of the same source, bit-identical by test: the COMPACT build below (augmented system [Z | rhs | J'^T] in registers,
//     everything else in LDS as far as it fits) and the LDS / scratch build (the system in LDS when it fits 120 KB, else in a
//     per-workgroup block of global scratch, L2-resident; pivot row and column of the elimination in LDS);
//   * the solve works on the step's ACTIVE constraints ("slots", at most 64): a world may register every pair of get_all_contacts;
//   * independent groups of constraints are swept side by side (same bits as the serial sweep).
// Limits: ndof, nb <= ARB_WIDE_MAX (1024), nc <= ARB_WIDE_MAX_CONSTRAINTS (256) of which at most 64 ACTIVE in one step.  Inputs:
// everything arb_step_ex takes -- state, constraint forces, user torques (constant or a sequence), the dense impedance of
// user-defined controllers, per-world PD targets / gains (and target sequences), per-step dt, state and energy logs, the running
// cost; the model's merged PD controllers.
// Not supported (ARB_ERR_UNSUPPORTED): the execution variants of the wavefront kernels (split sweeps, matrix-core elimination).
//
// The COMPACT build (KMAX > 0; worlds of at most 192 dofs and 256 columns: snake-100, human36 beside a few objects) keeps the
// augmented system in REGISTERS: wavefront g of the four holds rows g, g + 4, ... (KMAX of them), lane l the columns l, l + 64, ...
// (CP of them), see wide_eliminate below.  A pivot is then one hand-over of the pivot row through LDS (double-buffered, one
// barrier), the multipliers by v_readlane from the wavefront's own lane, and KMAX x CP fused multiply-adds per lane on registers --
// against a round trip through LDS per entry of the LDS-resident elimination (690 k of snake-100's 970 k cycles per step
// before).  LDS then has room for everything else a step touches: the chain arrays, the composites of phase B, the per-dof
// vectors X_k .. G_k, the rows of J', the solution columns, the admittance of the sweeps, the state (each group falls back to the
// scratch block when a world's size asks for it: the layout levels of wide_create, arb_kernels.hip).
#ifndef ARB_WIDE_KERNEL_H
#define ARB_WIDE_KERNEL_H
#define WIDE_THREADS 256

typedef double wide_d4 __attribute__((ext_vector_type(4)));

// Element k of a lane's column, kept as NCH vectors of four doubles (k is the same for every lane of the wavefront: a branch
// over the vectors, then an indexed register move -- s_set_gpr_idx -- inside one; not a chain of selects.  The unit is
// compiled with -simplifycfg-sink-common=false, csrc/Makefile: sinking the leaves' accesses into one access through a
// computed address would turn the registers into scratch memory)
template <int LO, int HI, int NCH>
__device__ __forceinline__ double wide_reg_get(const wide_d4 (&z)[NCH], int ch, int e) {
    if constexpr (HI - LO == 1) {
        return z[LO][e];
    } else {
        constexpr int MID = (LO + HI) / 2;
        if (ch < MID) return wide_reg_get<LO, MID, NCH>(z, ch, e);
        return wide_reg_get<MID, HI, NCH>(z, ch, e);
    }
}
template <int LO, int HI, int NCH>
__device__ __forceinline__ void wide_reg_set(wide_d4 (&z)[NCH], int ch, int e, double t) {
    if constexpr (HI - LO == 1) {
        z[LO][e] = t;
    } else {
        constexpr int MID = (LO + HI) / 2;
        if (ch < MID) wide_reg_set<LO, MID, NCH>(z, ch, e, t);
        else wide_reg_set<MID, HI, NCH>(z, ch, e, t);
    }
}

#ifndef WIDE_STAMP
#define WIDE_STAMP(i)       // (tools/wide_elim_probe.py: shader-clock stamps between the parts of one pivot)
#endif
// The elimination of the compact build.  Wavefront g of the four holds rows g, g + 4, g + 8 ... (KQ of them) of ALL 64 CP columns
// (CP 2: 128, CP 4: 256, CP 6: 384 -- every world of at most 128 dofs and 64 constraints): lane l the columns l, l + 64, ...  Pivot j: wavefront j & 3 hands the pivot row over (element j >> 2 of its lanes), lane
// j & 63 of every wavefront its KQ entries of the pivot column (the multipliers of that wavefront's rows); double-buffered --
// pivot j - 2 writes a buffer again only behind the barrier of pivot j - 1, which every lane passes after its reads of pivot
// j --: one barrier per pivot, then 2 KQ fused multiply-adds per lane on registers.  Columns j + 1 .. n - 1 are dead (what the
// elimination left of them stays in the registers, nothing reads it).  Per pivot and wavefront LDS returns KQ multipliers to
// 64 lanes: this layout halves that traffic against two row groups of 128 lanes -- it is what bounds the loop.
// A function of its own, not inlined: the kernel around it is one function of ~30 000 instructions whose register
// allocation left this loop's registers -- the system -- in scratch memory; a call boundary gives the loop an allocation of
// its own.  Arguments of a device function arrive in vector registers: readfirstlane makes the pivot counter a scalar again
// (the branches over it scalar branches); LDS goes by offsets, not pointers (a pointer argument is a generic one, every access
// through it a flat instruction).
template <int KQ, int CP>
__device__ __attribute__((noinline)) void wide_eliminate(const double *__restrict__ Z, int ld_, int n_, int nact_, int rb_off_,
                                                         int zl_off_, int sld_, const double *__restrict__ DQS, double *__restrict__ solg)
{
    constexpr int NCH = KQ / 4;
    const int tid = threadIdx.x;
    const int ld = __builtin_amdgcn_readfirstlane(ld_), n = __builtin_amdgcn_readfirstlane(n_), nact = __builtin_amdgcn_readfirstlane(nact_),
              rb_off = __builtin_amdgcn_readfirstlane(rb_off_),
              zl_off = __builtin_amdgcn_readfirstlane(zl_off_), sld = __builtin_amdgcn_readfirstlane(sld_);
    typedef __attribute__((address_space(3))) double lds_double;
    lds_double *const lds0 = (lds_double *)arb_lds_raw;
    lds_double *const RB = lds0 + rb_off, *const ZL = lds0 + zl_off;
    const int l = tid & 63, zg = __builtin_amdgcn_readfirstlane(tid >> 6);
    // out of the scratch block into the registers (coalesced; the only pass over the system that leaves the CU)
    wide_d4 z[CP][NCH];
#pragma unroll
    for (int p = 0; p < CP; ++p)
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * (4 * ch + e) + zg;
                z[p][ch][e] = (r < n && l + 64 * p < nact) ? Z[r * ld + l + 64 * p] : 0.;
            }
    // (column groups past the live columns -- a world that registers many constraints and has few of them active -- are skipped)
    const int pl = (nact + 63) >> 6;
    for (int j = n - 1; j >= 0; --j) {
        lds_double *rb = RB + (j & 1) * (64 * CP);
        const int gj = j & 3, cj = j >> 4;
        int ej = (j >> 2) & 3;
        // (vector cj, element ej; the element index opaque: seen next to cj the compiler folds the two back into ONE index into
        // the whole array -- an address, and the registers become scratch memory)
        asm volatile("" : "+s"(ej));
        WIDE_STAMP(0);
        // (the owner scales the row: its reciprocal chain hides under the readlanes below, the other wavefronts read t itself)
        double t[CP];
        if (zg == gj) {
#pragma unroll
            for (int p = 0; p < CP; ++p) t[p] = (p < 2 || p < pl) ? wide_reg_get<0, NCH, NCH>(z[p], cj, ej) : 0.;
            double own = j < 64 ? t[0] : t[1];                       // (pivots are columns j < n <= 192: the lane's first three)
            if constexpr (CP > 2) own = j < 128 ? own : t[2];
            const double piv = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(own), j & 63), __builtin_amdgcn_readlane(__double2loint(own), j & 63));
            const double ip = arb_rcp(piv);
#pragma unroll
            for (int p = 0; p < CP; ++p) if (p < 2 || p < pl) { t[p] = t[p] * ip; rb[l + 64 * p] = t[p]; }
        }
        WIDE_STAMP(1);
        // the multipliers of this wavefront's rows are the registers of ITS lane j & 63 (column j): broadcast through scalar
        // registers (v_readlane), not through LDS -- nothing of the pivot column leaves the wavefront, and the fused multiply-adds
        // take them as scalar operands.  (Before the barrier: independent of it.)
        const int lj = j & 63;
        wide_d4 f[NCH];
#define WIDE_MULT(P)                                                                                                                 \
        _Pragma("unroll") for (int ch = 0; ch < NCH; ++ch) _Pragma("unroll") for (int e = 0; e < 4; ++e)                               \
            f[ch][e] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(z[P][ch][e]), lj), __builtin_amdgcn_readlane(__double2loint(z[P][ch][e]), lj));
        if (j < 64) { WIDE_MULT(0) }
        else if (CP == 2 || j < 128) { WIDE_MULT(1) }
        else { WIDE_MULT(CP > 2 ? 2 : 1) }
#undef WIDE_MULT
        WIDE_STAMP(2);
        __syncthreads();
        WIDE_STAMP(3);
        if (zg != gj) {
#pragma unroll
            for (int p = 0; p < CP; ++p) t[p] = (p < 2 || p < pl) ? rb[l + 64 * p] : 0.;
        }
#pragma unroll
        for (int p = 0; p < CP; ++p) {
            if (p >= 2 && p >= pl) continue;
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) z[p][ch] = z[p][ch] - f[ch] * t[p];
        }
        WIDE_STAMP(4);
        // (a branch per vector around the selects, inside the loop above, measured slower: 868 against 674 cycles)
        if (zg == gj) {
#pragma unroll
            for (int p = 0; p < CP; ++p) if (p < 2 || p < pl) wide_reg_set<0, NCH, NCH>(z[p], cj, ej, t[p]);
        }
        WIDE_STAMP(5);
    }
    __syncthreads();                // (the chain arrays and the per-dof vectors are dead: the solution columns take their LDS)
    // the rhs column holds gvel+ - gvel: add gvel back so that it is Y (M gvel/dt + gforce)
#pragma unroll
    for (int p = 0; p < CP; ++p)
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * (4 * ch + e) + zg, c = l + 64 * p;
                if (r < n && c >= n && c < nact) {
                    const double v = c == n ? z[p][ch][e] + DQS[r] : z[p][ch][e];
                    if (solg != nullptr) solg[r * sld + (c - n)] = v;          // (many constraints: the solution columns in scratch)
                    else ZL[r * sld + (c - n)] = v;
                }
            }
    __syncthreads();
}

#define WIDE_XK 32          // per dof: X (6) | dX' (6) | P (6) | R (6) | G (6) | the dof's body and the end of that body's subtree (2)
#define WIDE_CD 40          // per constraint: R (9) | p (3) | pos0 (3) | sdist | active | glo | ghi | pad ... | pinv (16) at 24

struct WideModel {
    int nb, n, nq, nc, ndol, ncap, nds, ncols, ld, maxdepth, has_visc, has_pd, has_warm, has_grav, z_in_lds, chain_in_lds;
    int kmax, cp, ac_in_lds, am_in_lds, sld;                  // compact build: rows per lane (0: the LDS / scratch build); what else is in LDS
    long l_ac, l_xk, l_am, l_reg;                                // (offsets, doubles, inside the LDS region behind the sweeps' blocks)
    int am_cap;                                             // compact build: doubles of LDS kept for the admittance of the sweeps (0: none)
    int vec_in_lds;                                         // compact build: 1 = the state and the small per-world vectors in LDS, 2 = per-body wrenches and joint columns too
    long l_vec;
    int jr_in_lds;                                          // compact build: the rows of J' in LDS (under the dead composites)
    int sol_in_lds; long o_sol;                             // compact build: the solution columns in LDS (else in the scratch block at o_sol)
    long l_jr, l_sol;                                       // ... and where; where the solution columns go after the elimination
    int jrounds;                                            // > 0: a deep tree -- the chain runs in this many pointer-jumping rounds
    const int *janc;                                        // [jrounds][nb] the ancestor 2^r levels up (-1: none)
    double grav[3], up[3];
    const int *parent, *jtype, *dof_off, *jnd, *q_off, *depth, *weighted, *dof2q, *dofbody, *subsize;
    const double *Hpr, *Hcn, *mass, *visc;                 // [nb][12], [nb][12], [nb][36], [nb][36]
    const double *pd_kp, *pd_kd, *pd_tau0;                  // [n][n], [n][n], [n]
    const int *ctype, *cen, *cbody, *cbody0, *cdof, *cgeom;
    const double *clocal, *cradius, *cradius0, *chalf, *cplane, *cRz, *cb0, *cb1, *cmu, *ceps, *cprox, *cmin, *cmax;
    // offsets (doubles) of the arrays of ONE world inside a workgroup's scratch block
    long o_q, o_dq, o_qd, o_ff, o_ff0, o_pose, o_pc, o_rcp, o_tw, o_ab, o_om, o_da, o_tn, o_bn, o_pt, o_sc, o_ac, o_mc, o_wc, o_xk,
         o_rh, o_z, o_jr, o_am, o_vv, o_cd, total;
    int *status, *warn;
};

template <typename T>
struct WideIO {
    T *q, *dq, *cf;
    const T *ext, *zimp;
    long ext_stride;
    T *log_q, *log_dq;                                      // [nsteps][nw][nq], [nsteps][nw][n] or null
    T *log_energy;                                          // [nsteps][nw][2] kinetic, potential (observers.py:40-51) or null
    // one ProportionalDerivativeController per world (controllers.py:63-158): targets [nw][n] (pd_stride: a sequence, one row
    // per step) with the model's gain matrices, or with per-world DIAGONAL gains kp / kd [nw][n] that replace them
    const T *pd_qdes, *pd_dqdes, *pd_kp, *pd_kd;
    long pd_stride;
    // running cost of the rollout (arb_step_cost): out[w] += sum_i wq (qj - qref)^2 + wdq dq^2 + wtau tau^2 after every step
    T *cost_out;
    const T *cost_wq, *cost_wdq, *cost_wtau, *cost_qref;
    // inspect outputs (any may be null; one step, the state buffers are not written when `inspect` is set)
    int inspect;
    int zmode;                                              // inspect: what Zout receives -- 0 Z, 1 M, 2 B, 3 N (core.py:722-734)
    T *jac, *djac;                                          // [nw][nb][6][n] Body.jacobian / djacobian (core.py:1273-1274)
    T *pose, *twist, *Zout, *gforce0, *vel_free, *c_sdist, *c_jac, *c_force, *c_frame, *gforce, *q_next, *dq_next, *c_adm, *c_vel;
    int *c_active;
    int gs_serial;                                          // the sweeps over ALL constraints in one sequence (no independent groups)
    long long *stamps;                                      // [nw][8] shader clock at the phase boundaries (diagnostic, inspect)
};

template <typename T, int KMAX, int CP>
__global__ __launch_bounds__(WIDE_THREADS) void arb_wide_kernel(const WideModel *__restrict__ mp_in, const WideIO<T> io, long nworlds,
                                                                double dt_in, const double *__restrict__ dts, int nsteps,
                                                                unsigned flags, double *__restrict__ scratch_all)
{
    const WideModel &M = *mp_in;
    const int tid = threadIdx.x;
    // nc / ndol: the world's constraints and their 4 nc force components; ncap / nds: at most so many of them ACTIVE in one step (and
    // four times that): the slots of everything the solve touches; ncols = n + 1 + nds columns at most
    const int n = M.n, nb = M.nb, nq = M.nq, nc = M.nc, ndol = M.ndol, ncap = M.ncap, nds = M.nds, ncols = M.ncols, ld = M.ld;
    double *lds = reinterpret_cast<double *>(arb_lds_raw);
    double *TROW = lds;                      // [ncols]   the scaled pivot row
    double *FCOL = TROW + ((ncols + 3) & ~3);   // [n]    the pivot column (multipliers)
    double *DF = FCOL + ((n + 3) & ~3);      // [8]       force increment of one local solve
    double *SWORK = DF + 8;                  // [48]      (spare: the sliding solve's eig6 fallback keeps its scratch per lane since the group sweeps)
    double *GVV = SWORK + 48;                // [nds]     v' during the sweeps
    double *GFF = GVV + ((M.nds + 3) & ~3);  // [nds]     constraint forces during the sweeps
    double *GSC = GFF + ((M.nds + 3) & ~3);  // [ncap][52] per-slot blocks and constants of the sweeps
    unsigned long long *GGM = reinterpret_cast<unsigned long long *>(GSC + 52 * M.ncap);   // [ncap]  the group of a slot, a bit per member
    double *GDF = GSC + 53 * M.ncap;         // [ncap][6] per group: the force increment of this round's solve (4), its slot
    int *ORDL = reinterpret_cast<int *>(GSC + 59 * M.ncap);      // [ncap + 2]  slot -> constraint (this step's active ones, in order), their number
    double *ZL = GSC + 59 * M.ncap + (M.ncap & 1) + 2 * ((M.ncap + 5) >> 2);     // (16-byte steps: what follows is read two doubles at a time)
    // ZL: [n][ld]   the augmented system, when it fits (compact build: the region below)
    constexpr bool REGZ = KMAX > 0;
    // compact build: [2][128] pivot rows | the admittance of the sweeps | chain arrays, composites,
    // per-dof vectors (the solution columns take their place after the elimination)
    double *GAM = ZL + M.l_am;
    if constexpr (REGZ) ZL += M.l_reg;
    double *S = scratch_all + (size_t)blockIdx.x * (size_t)M.total;
    // The per-body arrays of the pose / twist chain (84 doubles per body, contiguous in the scratch block from o_pose on) live in
    // LDS while the chain runs -- in the space the augmented system takes afterwards (chain_in_lds: they fit it) --: one
    // depth level of a serial chain is a dependent round trip, 100 of them for snake-100.  Everything that reads them comes
    // before the assembly of Z.
    double *CHB = M.chain_in_lds ? ZL : S + M.o_pose;
    // (the scratch block begins with the state and the small vectors, then the per-body wrenches and the joints' columns: the
    //  compact build keeps them in LDS where that costs no workgroup per CU -- every one of them is an L2 round trip otherwise)
    double *V1 = M.vec_in_lds >= 1 ? ZL + M.l_vec : S, *V2 = M.vec_in_lds >= 2 ? ZL + M.l_vec : S;
    double *QS = V1 + M.o_q, *DQS = V1 + M.o_dq, *QD = V1 + M.o_qd, *FF = V1 + M.o_ff, *FF0 = V1 + M.o_ff0, *POSE = CHB,
           *PC = CHB + (M.o_pc - M.o_pose), *RCP = CHB + (M.o_rcp - M.o_pose), *TW = CHB + (M.o_tw - M.o_pose), *AB = CHB + (M.o_ab - M.o_pose),
           *OM = CHB + (M.o_om - M.o_pose), *DA = CHB + (M.o_da - M.o_pose), *TN = CHB + (M.o_tn - M.o_pose), *BN = CHB + (M.o_bn - M.o_pose),
           *PT = V2 + M.o_pt, *SC = V2 + M.o_sc, *AC = M.ac_in_lds ? ZL + M.l_ac : S + M.o_ac, *MC = M.ac_in_lds ? AC + 36 * nb : S + M.o_mc,
           *WC = M.ac_in_lds ? AC + 72 * nb : S + M.o_wc, *XK = REGZ ? ZL + M.l_xk : S + M.o_xk, *RH = V1 + M.o_rh, *JR = M.jr_in_lds ? ZL + M.l_jr : S + M.o_jr,
           *AMG = S + M.o_am, *VV = V1 + M.o_vv, *CD = V1 + M.o_cd;
    double *Z = M.z_in_lds ? ZL : S + M.o_z;
    // the solution columns [Y rhs | Y J'^T] after the elimination: inside Z, or (compact build) written out of the registers
    const double *SL = REGZ ? (M.sol_in_lds ? ZL + M.l_sol : S + M.o_sol) : Z + n;
    const int sld = REGZ ? M.sld : ld;
    const bool do_con = nc > 0 && !(flags & ARB_STEP_SKIP_CONSTRAINTS);
    auto ld3 = [](const double *p) { return v3<double>(p[0], p[1], p[2]); };
    auto ldm = [](const double *p) { M3<double> r; for (int i = 0; i < 9; ++i) r.a[i] = p[i]; return r; };
    auto st3 = [](double *p, V3<double> v) { p[0] = v.x; p[1] = v.y; p[2] = v.z; };
    auto stm = [](double *p, const M3<double> &m) { for (int i = 0; i < 9; ++i) p[i] = m.a[i]; };
    // body a is an ancestor of (or is) body b: DFS preorder makes every subtree a contiguous range
    auto anc_eq = [&](int a, int b) { return a >= 0 && b >= 0 && a <= b && b < a + M.subsize[a]; };
    // two-dimensional lane mapping of the matrix loops (no integer division per entry): a power-of-two number of column lanes
    // covering n (at most 256), the other lanes stride over rows
    int cwn = 1;
    while (cwn < n && cwn < WIDE_THREADS) cwn <<= 1;
    const int cl_n = tid & (cwn - 1), r0_n = tid / cwn, rs_n = WIDE_THREADS / cwn;

    for (long w = blockIdx.x; w < nworlds; w += gridDim.x) {
    __syncthreads();
    for (int i = tid; i < nq; i += WIDE_THREADS) QS[i] = (double)io.q[w * nq + i];
    for (int i = tid; i < n; i += WIDE_THREADS) DQS[i] = (double)io.dq[w * n + i];
    for (int i = tid; i < ndol; i += WIDE_THREADS) FF[i] = io.cf != nullptr ? (double)io.cf[w * ndol + i] : 0.;
    __syncthreads();

    for (int step = 0; step < nsteps; ++step) {
        const double dt = dts != nullptr ? dts[step] : dt_in, inv_dt = 1. / dt;
        if (io.log_q != nullptr) for (int i = tid; i < nq; i += WIDE_THREADS) io.log_q[((long)step * nworlds + w) * nq + i] = (T)QS[i];
        if (io.log_dq != nullptr) for (int i = tid; i < n; i += WIDE_THREADS) io.log_dq[((long)step * nworlds + w) * n + i] = (T)DQS[i];
        if (io.inspect && io.stamps != nullptr && tid == 0) io.stamps[w * 8 + 0] = (long long)clock64();
        // ================= phase A: lane = body -- joint-local kinematics (core.py:1294-1315, rigidmotion.py:47-73) ==========
        for (int b = tid; b < nb; b += WIDE_THREADS) {
            const int jt = M.jtype[b], doff = M.dof_off[b], k = M.jnd[b];
            JointLocal<double> jl;
            joint_local<double>(jt, (const double *)(QS + M.q_off[b]), (const double *)(DQS + doff), jl);
            const M3<double> R_pr = ldm(M.Hpr + 12 * b), R_cn = ldm(M.Hcn + 12 * b);
            const V3<double> p_pr = ld3(M.Hpr + 12 * b + 9), p_cn = ld3(M.Hcn + 12 * b + 9);
            const M3<double> R_rc = mulBT(jl.R, R_cn);                           // H_pc = H_pr H_rn inv(H_cn)   core.py:1298
            const V3<double> p_rc = mv(jl.R, -mtv(R_cn, p_cn)) + jl.p;
            const M3<double> R_pc = mul(R_pr, R_rc);
            const V3<double> p_pc = mv(R_pr, p_rc) + p_pr;
            stm(PC + 12 * b, R_pc); st3(PC + 12 * b + 9, p_pc);
            const M3<double> R_cp = transpose(R_pc);                             // Ad_cp = Ad(inv(H_pc))        :1300
            const V3<double> p_cp = -mtv(R_pc, p_pc);
            stm(RCP + 12 * b, R_cp); st3(RCP + 12 * b + 9, p_cp);
            const M3<double> R_nr = transpose(jl.R);
            const V3<double> p_nr = -mtv(jl.R, jl.p);
            const V3<double> aw = mv(R_nr, jl.Tw);                               // -T_rn
            const V3<double> av = cross(p_nr, aw) + mv(R_nr, jl.Tv);
            const Blk<double> Ad_nr = blk_adjoint(R_nr, p_nr);
            const Blk<double> dAd_nr = blk_mul(Ad_nr, blk_adjacency(-aw, -av));
            const Blk<double> Ad_cn = blk_adjoint(R_cn, p_cn);
            const Blk<double> Ad_rp = blk_adjoint(transpose(R_pr), -mtv(R_pr, p_pr));
            const Blk<double> dAd_cp = blk_mul(Ad_cn, blk_mul(dAd_nr, Ad_rp));   // core.py:1304
            stm(DA + 18 * b, dAd_cp.A); stm(DA + 18 * b + 9, dAd_cp.B);
            // W_c = Ad_cp Ad_pr T_rn: the pseudo twist's own term (see arb_phase_b.h)
            {
                const V3<double> uw = mv(R_pr, -aw);
                const V3<double> uv = cross(p_pr, uw) + mv(R_pr, -av);
                const V3<double> ww = mv(R_cp, uw);
                st3(OM + 6 * b, ww); st3(OM + 6 * b + 3, cross(p_cp, ww) + mv(R_cp, uv));
            }
            V3<double> bw = v3<double>(0., 0., 0.);
            if (jt != JT_FREE && jt != JT_TXTYTZ)
                for (int i = 0; i < 3; ++i) if (i < k) bw = bw + DQS[doff + i] * jl.djw[i];
            const V3<double> Bnw = mv(R_cn, bw);
            st3(BN + 6 * b, Bnw); st3(BN + 6 * b + 3, cross(p_cn, Bnw));
            const V3<double> Tnw = mv(R_cn, jl.Tw);
            st3(TN + 6 * b, Tnw); st3(TN + 6 * b + 3, cross(p_cn, Tnw) + mv(R_cn, jl.Tv));
            for (int i = 0; i < k; ++i) {                                        // own columns Ad_cn J_nr, Ad_cn dJ_nr  :1310, 1313
                V3<double> cw = v3<double>(0., 0., 0.), cv = cw, dw = cw;
                if (jt == JT_FREE) {
                    if (i < 3) cw = v3<double>(i == 0 ? 1. : 0., i == 1 ? 1. : 0., i == 2 ? 1. : 0.);
                    else cv = v3<double>(i == 3 ? 1. : 0., i == 4 ? 1. : 0., i == 5 ? 1. : 0.);
                } else if (jt == JT_TXTYTZ) {
                    cv = v3<double>(i == 0 ? 1. : 0., i == 1 ? 1. : 0., i == 2 ? 1. : 0.);
                } else if (i < 3) {
                    cw = jl.jw[i]; dw = jl.djw[i];
                }
                const V3<double> ow = mv(R_cn, cw), ov = cross(p_cn, ow) + mv(R_cn, cv);
                const V3<double> dow = mv(R_cn, dw), dov = cross(p_cn, dow);
                double *sc = SC + 12 * (doff + i);
                st3(sc, ow); st3(sc + 3, ov); st3(sc + 6, dow); st3(sc + 9, dov);
            }
        }
        for (int i = tid; i < n; i += WIDE_THREADS) { const int qi = M.dof2q[i]; QD[i] = qi >= 0 ? QS[qi] : 0.; }
        __syncthreads();
        // Deep trees (snake-100: 101 levels of one working lane, 286 k of the step's 1.2 M cycles): log-depth chains, as in the
        // float64 wavefront kernels (arb_phase_a.h).  Poses by pointer jumping -- every body composes its pose with that of its
        // ancestor 2^r levels up, round after round --; twists, bias accelerations and pseudo twists are sums over the
        // ancestors once they are written in WORLD axes about the world origin: Ad(H_gc) T_c = Ad(H_gp) T_p + Ad(H_gc) Tn_c.
        // The ancestor tables are the model's (janc); lane = body (nb <= 256).
        if (M.jrounds > 0) {
            const int b = tid;
            const bool on = b < nb;
            const int par = on ? M.parent[b] : -1;
            if (on) for (int i = 0; i < 12; ++i) POSE[12 * b + i] = PC[12 * b + i];
            __syncthreads();
            for (int r = 0; r < M.jrounds; ++r) {
                const int a = on ? M.janc[r * nb + b] : -1;
                M3<double> Ra = m3_identity<double>(); V3<double> pa = v3<double>(0., 0., 0.);
                if (a >= 0) { Ra = ldm(POSE + 12 * a); pa = ld3(POSE + 12 * a + 9); }
                __syncthreads();
                if (a >= 0) {
                    const M3<double> Rb = ldm(POSE + 12 * b);
                    const V3<double> pb = ld3(POSE + 12 * b + 9);
                    stm(POSE + 12 * b, mul(Ra, Rb)); st3(POSE + 12 * b + 9, mv(Ra, pb) + pa);
                }
                __syncthreads();
            }
            auto jump_sum = [&](double *arr) {          // inclusive sum over the ancestors of the 6-vectors of `arr`
                __syncthreads();
                for (int r = 0; r < M.jrounds; ++r) {
                    const int a = on ? M.janc[r * nb + b] : -1;
                    double add6[6] = {0., 0., 0., 0., 0., 0.};
                    if (a >= 0) for (int i = 0; i < 6; ++i) add6[i] = arr[6 * a + i];
                    __syncthreads();
                    if (a >= 0) for (int i = 0; i < 6; ++i) arr[6 * b + i] += add6[i];
                    __syncthreads();
                }
            };
            M3<double> Rgb = m3_identity<double>(); V3<double> pgb = v3<double>(0., 0., 0.);
            auto to_world = [&](double *arr, V3<double> lw, V3<double> lv) {
                const V3<double> ww = mv(Rgb, lw);
                st3(arr + 6 * b, ww); st3(arr + 6 * b + 3, cross(pgb, ww) + mv(Rgb, lv));
            };
            auto to_body = [&](double *arr) {
                const V3<double> ww = ld3(arr + 6 * b), wv = ld3(arr + 6 * b + 3);
                st3(arr + 6 * b, mtv(Rgb, ww)); st3(arr + 6 * b + 3, mtv(Rgb, wv - cross(pgb, ww)));
            };
            if (on) {
                Rgb = ldm(POSE + 12 * b); pgb = ld3(POSE + 12 * b + 9);
                to_world(TW, ld3(TN + 6 * b), ld3(TN + 6 * b + 3));                                        // core.py:1308
                to_world(OM, ld3(OM + 6 * b), ld3(OM + 6 * b + 3));
            }
            jump_sum(TW);
            if (on) to_body(TW);
            __syncthreads();
            if (on) {                                   // dAd_cp T_p + Bn_c in body axes (core.py:1312-1313 times gvel), to world axes
                V3<double> tw = v3<double>(0., 0., 0.), tv = tw;
                if (par >= 0) { tw = ld3(TW + 6 * par); tv = ld3(TW + 6 * par + 3); }
                const M3<double> dA = ldm(DA + 18 * b), dB = ldm(DA + 18 * b + 9);
                to_world(AB, mv(dA, tw) + ld3(BN + 6 * b), mv(dB, tw) + mv(dA, tv) + ld3(BN + 6 * b + 3));
            }
            jump_sum(AB);
            if (on) to_body(AB);
            jump_sum(OM);
            if (on) to_body(OM);
            __syncthreads();
        } else
        // pose, twist, bias acceleration and pseudo twist down the tree, one depth level per pass
        for (int lvl = 0; lvl <= M.maxdepth; ++lvl) {
            for (int b = tid; b < nb; b += WIDE_THREADS) {
                if (M.depth[b] != lvl) continue;
                const int par = M.parent[b];
                M3<double> Rg = m3_identity<double>();
                V3<double> pg = v3<double>(0., 0., 0.), tw = pg, tv = pg, aw = pg, av = pg, omw = pg, omv = pg;
                if (par >= 0) {
                    Rg = ldm(POSE + 12 * par); pg = ld3(POSE + 12 * par + 9);
                    tw = ld3(TW + 6 * par); tv = ld3(TW + 6 * par + 3);
                    aw = ld3(AB + 6 * par); av = ld3(AB + 6 * par + 3);
                    omw = ld3(OM + 6 * par); omv = ld3(OM + 6 * par + 3);
                }
                const M3<double> R_pc = ldm(PC + 12 * b), R_cp = ldm(RCP + 12 * b), dA = ldm(DA + 18 * b), dB = ldm(DA + 18 * b + 9);
                const V3<double> p_pc = ld3(PC + 12 * b + 9), p_cp = ld3(RCP + 12 * b + 9);
                stm(POSE + 12 * b, mul(Rg, R_pc)); st3(POSE + 12 * b + 9, mv(Rg, p_pc) + pg);          // core.py:1299
                const V3<double> rtw = mv(R_cp, tw);
                st3(TW + 6 * b, rtw + ld3(TN + 6 * b));                                                // core.py:1308
                st3(TW + 6 * b + 3, cross(p_cp, rtw) + mv(R_cp, tv) + ld3(TN + 6 * b + 3));
                const V3<double> raw = mv(R_cp, aw);                                                   // core.py:1312-1313 times gvel
                st3(AB + 6 * b, mv(dA, tw) + raw + ld3(BN + 6 * b));
                st3(AB + 6 * b + 3, mv(dB, tw) + mv(dA, tv) + cross(p_cp, raw) + mv(R_cp, av) + ld3(BN + 6 * b + 3));
                if (par >= 0) {
                    const V3<double> rw = mv(R_cp, omw);
                    const V3<double> nv = cross(p_cp, rw) + mv(R_cp, omv) + ld3(OM + 6 * b + 3);
                    st3(OM + 6 * b, rw + ld3(OM + 6 * b)); st3(OM + 6 * b + 3, nv);
                }
            }
            __syncthreads();
        }
        if (io.inspect && step == 0) {      // (poses and twists leave here: their arrays are gone once Z is assembled)
            if (io.pose != nullptr) for (int b = tid; b < nb; b += WIDE_THREADS) {
                T *o = io.pose + (w * nb + b) * 16;
                for (int i = 0; i < 3; ++i) {
                    for (int j = 0; j < 3; ++j) o[4 * i + j] = (T)POSE[12 * b + 3 * i + j];
                    o[4 * i + 3] = (T)POSE[12 * b + 9 + i];
                }
                o[12] = o[13] = o[14] = T(0); o[15] = T(1);
            }
            if (io.twist != nullptr) for (int i = tid; i < 6 * nb; i += WIDE_THREADS) io.twist[w * nb * 6 + i] = (T)TW[i];
        }
        // energies as an observer sees them at the beginning of the step (EnergyMonitor.update, observers.py:40-51)
        if (io.log_energy != nullptr) {
            double ke = 0., pe = 0.;
            for (int b = tid; b < nb; b += WIDE_THREADS) {
                const double *Mb = M.mass + 36 * b;
                double tw[6], mt[6];
                for (int i = 0; i < 6; ++i) tw[i] = TW[6 * b + i];
                mat6_vec<double>(Mb, tw, mt);
                for (int i = 0; i < 6; ++i) ke += 0.5 * tw[i] * mt[i];
                const double mass_b = Mb[35];
                if (mass_b > 0.) {
                    const V3<double> cm = v3<double>(Mb[6 * 2 + 4] / mass_b, Mb[6 * 0 + 5] / mass_b, Mb[6 * 1 + 3] / mass_b);
                    const V3<double> cg = mv(ldm(POSE + 12 * b), cm) + ld3(POSE + 12 * b + 9);
                    pe += 9.81 * Mb[21] * (M.up[0] * cg.x + M.up[1] * cg.y + M.up[2] * cg.z);
                }
            }
            for (int off = 32; off >= 1; off >>= 1) { ke += __shfl_xor(ke, off); pe += __shfl_xor(pe, off); }
            __syncthreads();
            if ((tid & 63) == 0) { TROW[2 * (tid >> 6)] = ke; TROW[2 * (tid >> 6) + 1] = pe; }
            __syncthreads();
            if (tid == 0) {
                T *o = io.log_energy + ((long)step * nworlds + w) * 2;
                o[0] = (T)(TROW[0] + TROW[2] + TROW[4] + TROW[6]); o[1] = (T)(TROW[1] + TROW[3] + TROW[5] + TROW[7]);
            }
            __syncthreads();
        }
        // body wrench of the increment form: M_b g_b - M_b (dJ_b gvel) - N_b T_b - B_b T_b   (core.py:975-976, 1276-1288)
        for (int b = tid; b < nb; b += WIDE_THREADS) {
            const double *Mb = M.mass + 36 * b;
            double tw[6], ab[6], mt[6], ma[6], mg[6], g6[6] = {0., 0., 0., 0., 0., 0.};
            for (int i = 0; i < 6; ++i) { tw[i] = TW[6 * b + i]; ab[i] = AB[6 * b + i]; }
            mat6_vec<double>(Mb, tw, mt);
            mat6_vec<double>(Mb, ab, ma);
            if (M.has_grav && M.weighted[b]) {                                                          // controllers.py:56-58
                const V3<double> gl = mtv(ldm(POSE + 12 * b), v3<double>(M.grav[0], M.grav[1], M.grav[2]));
                g6[3] = gl.x; g6[4] = gl.y; g6[5] = gl.z;
            }
            mat6_vec<double>(Mb, g6, mg);
            const V3<double> wv = v3<double>(tw[0], tw[1], tw[2]);
            const M3<double> wx = hat(wv);
            M3<double> rx = m3_zero<double>();
            const double mm = Mb[21];
            if (!(mm <= 1e-10)) {
                const double im = 1. / mm;
                for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) rx.a[3 * i + j] = Mb[6 * i + 3 + j] * im;
            }
            const M3<double> Cm = sub(mul(rx, wx), mul(wx, rx));
            const V3<double> mtt = v3<double>(mt[0], mt[1], mt[2]), mtb = v3<double>(mt[3], mt[4], mt[5]);
            const V3<double> ntop = cross(wv, mtt) + mv(Cm, mtb), nbot = cross(wv, mtb);
            double pt[6] = {mg[0] - ma[0] - ntop.x, mg[1] - ma[1] - ntop.y, mg[2] - ma[2] - ntop.z,
                            mg[3] - ma[3] - nbot.x, mg[4] - ma[4] - nbot.y, mg[5] - ma[5] - nbot.z};
            double pg6[6] = {mg[0], mg[1], mg[2], mg[3], mg[4], mg[5]};
            if (M.has_visc) {
                double vt[6];
                mat6_vec<double>(M.visc + 36 * b, tw, vt);
                for (int i = 0; i < 6; ++i) pt[i] -= vt[i];
            }
            for (int i = 0; i < 6; ++i) { PT[12 * b + i] = pt[i]; PT[12 * b + 6 + i] = pg6[i]; }
        }
        if (io.inspect && io.stamps != nullptr && tid == 0) io.stamps[w * 8 + 1] = (long long)clock64();
        // ================= phase A': lane = constraint (constraints.py:277-294, 35-90, collisions.py) =====================
        if (do_con) for (int c = tid; c < nc; c += WIDE_THREADS) {
            double *cd = CD + WIDE_CD * c;
            const int ct = M.ctype[c];
            bool active = false;
            double sd = 0.;
            for (int i = 0; i < 24; ++i) cd[i] = 0.;
            if (M.cen[c]) {
                const int b0 = M.cbody0[c], b1 = M.cbody[c];
                M3<double> Rg0 = m3_identity<double>(), Rg1 = Rg0;
                V3<double> pg0 = v3<double>(0., 0., 0.), pg1 = pg0, bw0 = pg0, bv0 = pg0, bw1 = pg0, bv1 = pg0;
                if (ct != ARB_CT_JOINTLIMITS) {
                    if (b0 >= 0) { Rg0 = ldm(POSE + 12 * b0); pg0 = ld3(POSE + 12 * b0 + 9); bw0 = ld3(TW + 6 * b0); bv0 = ld3(TW + 6 * b0 + 3); }
                    if (b1 >= 0) { Rg1 = ldm(POSE + 12 * b1); pg1 = ld3(POSE + 12 * b1 + 9); bw1 = ld3(TW + 6 * b1); bv1 = ld3(TW + 6 * b1 + 3); }
                }
                if (ct == ARB_CT_SOFTFINGER) {
                    const M3<double> Rs0 = mul(Rg0, ldm(M.cb0 + 12 * c));
                    const V3<double> ps0 = mv(Rg0, ld3(M.cb0 + 12 * c + 9)) + pg0;
                    const V3<double> p_g1 = mv(Rg1, ld3(M.clocal + 3 * c)) + pg1;
                    V3<double> gc0, gc1;
                    M3<double> Rc;
                    sd = narrow_phase(M.cgeom[c], Rs0, ps0, p_g1, M.cradius[c], M.cradius0[c], ld3(M.chalf + 3 * c),
                                      ld3(M.cplane + 4 * c), M.cplane[4 * c + 3], ldm(M.cRz + 9 * c), gc0, gc1, Rc);
                    const M3<double> R1 = mulTA(Rc, Rg1), R0 = mulTA(Rc, Rg0);
                    const V3<double> P1 = mtv(Rc, pg1 - gc0), P0 = mtv(Rc, pg0 - gc0);
                    const double vz1 = (mv(R1, bv1) + cross(P1, mv(R1, bw1))).z;                        // constraints.py:289-291
                    const double vz0 = (mv(R0, bv0) + cross(P0, mv(R0, bw0))).z;
                    active = (sd + (vz1 - vz0) * dt < M.cprox[c]);
                    stm(cd, transpose(Rc)); st3(cd + 9, -mtv(Rc, gc0));      // world axes about the WORLD origin -> contact frame 0
                    for (int i = 0; i < 4; ++i) FF[4 * c + i] = 0.;           // constraints.py:294
                    if (io.inspect && io.c_frame != nullptr && step == 0) {
                        for (int f = 0; f < 2; ++f) {
                            T *of = io.c_frame + ((w * nc + c) * 2 + f) * 16;
                            const V3<double> gf = f ? gc1 : gc0;
                            for (int i = 0; i < 3; ++i) {
                                for (int j = 0; j < 3; ++j) of[4 * i + j] = (T)Rc.a[3 * i + j];
                                of[4 * i + 3] = (T)(i == 0 ? gf.x : i == 1 ? gf.y : gf.z);
                            }
                            of[12] = of[13] = of[14] = T(0); of[15] = T(1);
                        }
                    }
                } else if (ct == ARB_CT_JOINTLIMITS) {
                    const double p0 = QD[M.cdof[c]];
                    active = (p0 - M.cmin[c] < M.cprox[c]) || (M.cmax[c] - p0 < M.cprox[c]);
                    cd[12] = p0; cd[17] = (M.cmin[c] - p0) / dt; cd[18] = (M.cmax[c] - p0) / dt;
                    for (int i = 0; i < 4; ++i) FF[4 * c + i] = 0.;           // constraints.py:58-60
                    sd = p0;
                } else {                                                      // BallAndSocket  constraints.py:196-207
                    const M3<double> RP0 = mul(Rg0, ldm(M.cb0 + 12 * c));
                    const V3<double> pP0 = mv(Rg0, ld3(M.cb0 + 12 * c + 9)) + pg0, pP1 = mv(Rg1, ld3(M.cb1 + 12 * c + 9)) + pg1;
                    st3(cd + 12, mtv(RP0, pP1 - pP0));
                    stm(cd, transpose(RP0)); st3(cd + 9, -mtv(RP0, pP0));
                    active = true;
                }
            }
            cd[15] = sd; cd[16] = active ? 1. : 0.;
        }
        __syncthreads();
        for (int i = tid; i < ndol; i += WIDE_THREADS) FF0[i] = FF[i];
        // ---- this step's ACTIVE constraints, in registration order: slot s holds constraint ORDL[s].  Everything the solve touches --
        // rows of J', columns of the system, admittance, sweeps -- is indexed by slot: a world may REGISTER many more constraints
        // (every pair of get_all_contacts) than are active at a time; at most ncap = min(nc, 64) active ones are kept (more: the
        // later ones are left out of the step and ARB_WARN_ACTIVE_CONSTRAINTS is raised)
        if (tid < WAVE) {
            int base = 0;
            for (int c0 = 0; c0 < nc; c0 += WAVE) {
                const int c = c0 + tid;
                const bool act = do_con && c < nc && CD[WIDE_CD * c + 16] != 0.;
                const unsigned long long bal = __ballot(act);
                const int s = base + __popcll(bal & ((1ull << tid) - 1ull));
                if (act) { if (s < ncap) ORDL[s] = c; else CD[WIDE_CD * c + 16] = 0.; }
                base += __popcll(bal);
            }
            if (tid == 0) {
                ORDL[ncap] = base < ncap ? base : ncap;
                if (base > ncap) (void)__hip_atomic_fetch_or(M.warn, (int)ARB_WARN_ACTIVE_CONSTRAINTS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        __syncthreads();
        const int nca = ORDL[ncap], nda = 4 * nca;
        // the admittance Y' of the step's active constraints, packed (row stride = their 4 nca rows): in LDS when it fits what the
        // layout keeps for it -- a world that registers 108 constraints and has 15 active reads 28 KB, not 0.5 MB of scratch
        const bool am_l = nda * nda <= M.am_cap;
        double *const AM = am_l ? GAM : AMG;
        const int ams = am_l ? nda : nds;
        if (io.inspect && io.stamps != nullptr && tid == 0) io.stamps[w * 8 + 2] = (long long)clock64();
        // ================= phase B: composite assembly of Z = M/dt + B + N (core.py:722-734, 813), see arb_phase_b.h =========
        // ---- lane = body: world-frame matrices about the WORLD origin -------------------------------------------------
        for (int b = tid; b < nb; b += WIDE_THREADS) {
            const M3<double> R = ldm(POSE + 12 * b);
            const V3<double> p = ld3(POSE + 12 * b + 9);
            const double *Mb = M.mass + 36 * b;
            auto blk = [](const double *m6, int r0, int c0) {
                M3<double> o;
                for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) o.a[3 * i + j] = m6[6 * (r0 + i) + c0 + j];
                return o;
            };
            auto rot = [&](const M3<double> &Xm) { return mul(R, mulBT(Xm, R)); };
            auto rowcross = [](const M3<double> &Xm, V3<double> v) {
                M3<double> o;
                for (int i = 0; i < 3; ++i) {
                    const V3<double> c = cross(v3<double>(Xm.a[3 * i], Xm.a[3 * i + 1], Xm.a[3 * i + 2]), v);
                    o.a[3 * i] = c.x; o.a[3 * i + 1] = c.y; o.a[3 * i + 2] = c.z;
                }
                return o;
            };
            double G[36], A[36];
            {
                const M3<double> M11 = rot(blk(Mb, 0, 0)), M12 = rot(blk(Mb, 0, 3)), M22 = rot(blk(Mb, 3, 3));
                const M3<double> G12 = add(M12, hatmul(p, M22));
                const M3<double> G21 = transpose(G12);
                const M3<double> G11 = add(sub(M11, rowcross(M12, p)), hatmul(p, G21));
                for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
                    G[6 * i + j] = G11.a[3 * i + j]; G[6 * i + 3 + j] = G12.a[3 * i + j];
                    G[6 * (3 + i) + j] = G21.a[3 * i + j]; G[6 * (3 + i) + 3 + j] = M22.a[3 * i + j];
                }
            }
            const V3<double> wb = ld3(TW + 6 * b);
            const double mm = Mb[21];
            V3<double> cm = v3<double>(0., 0., 0.);
            if (!(mm <= 1e-10)) cm = (1. / mm) * v3<double>(Mb[6 * 2 + 4], Mb[6 * 0 + 5], Mb[6 * 1 + 3]);
            const V3<double> Tw = mv(R, wb), Tv = mv(R, cross(cm, wb)) + cross(p, Tw);
            const V3<double> ow = mv(R, ld3(OM + 6 * b)), ov = mv(R, ld3(OM + 6 * b + 3)) + cross(p, ow);
            const bool useM = io.zmode == 0 || io.zmode == 1, useN = io.zmode == 0 || io.zmode == 3, useB = io.zmode == 0 || io.zmode == 2;
            const double cM = io.zmode == 1 ? 1. : inv_dt;
            for (int i = 0; i < 36; ++i) A[i] = useM ? cM * G[i] : 0.;
            if (useN) for (int j = 0; j < 6; ++j) {              // -ad(T*)^T Mg
                const V3<double> gt = v3<double>(G[j], G[6 + j], G[12 + j]), gb = v3<double>(G[18 + j], G[24 + j], G[30 + j]);
                const V3<double> t = cross(Tw, gt) + cross(Tv, gb), u = cross(Tw, gb);
                A[j] += t.x; A[6 + j] += t.y; A[12 + j] += t.z; A[18 + j] += u.x; A[24 + j] += u.y; A[30 + j] += u.z;
            }
            if (useN) for (int r = 0; r < 6; ++r) {              // Mg ad(Om)
                const V3<double> gl = v3<double>(G[6 * r], G[6 * r + 1], G[6 * r + 2]), gr = v3<double>(G[6 * r + 3], G[6 * r + 4], G[6 * r + 5]);
                const V3<double> t = cross(gl, ow) + cross(gr, ov), u = cross(gr, ow);
                A[6 * r] += t.x; A[6 * r + 1] += t.y; A[6 * r + 2] += t.z; A[6 * r + 3] += u.x; A[6 * r + 4] += u.y; A[6 * r + 5] += u.z;
            }
            if (M.has_visc && useB) {
                const double *Vb = M.visc + 36 * b;
                const M3<double> B11 = rot(blk(Vb, 0, 0)), B12 = rot(blk(Vb, 0, 3)), B21 = rot(blk(Vb, 3, 0)), B22 = rot(blk(Vb, 3, 3));
                const M3<double> H12 = add(B12, hatmul(p, B22));
                const M3<double> H21 = sub(B21, rowcross(B22, p));
                const M3<double> H11 = add(sub(B11, rowcross(B12, p)), hatmul(p, H21));
                for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
                    A[6 * i + j] += H11.a[3 * i + j]; A[6 * i + 3 + j] += H12.a[3 * i + j];
                    A[6 * (3 + i) + j] += H21.a[3 * i + j]; A[6 * (3 + i) + 3 + j] += B22.a[3 * i + j];
                }
            }
            for (int i = 0; i < 36; ++i) { AC[36 * b + i] = A[i]; MC[36 * b + i] = useN ? G[i] : 0.; }     // (M dX' is a term of N)
            for (int h = 0; h < 2; ++h) {                        // wrenches to world axes: increment rhs, gravity alone
                const double *pt = PT + 12 * b + 6 * h;
                const V3<double> f = mv(R, v3<double>(pt[3], pt[4], pt[5]));
                const V3<double> tq = mv(R, v3<double>(pt[0], pt[1], pt[2])) + cross(p, f);
                double *wc = WC + 12 * b + 6 * h;
                st3(wc, tq); st3(wc + 3, f);
            }
        }
        __syncthreads();
        // ---- subtree sums: lane = entry, bodies from the leaves to the roots (parent[b] < b) ------------------------
        // (the parents from LDS -- the pivot-row buffer is free here --, the next body's entry and parent read before this body's sum
        // is stored, and a body whose parent is the body before it -- every link of a chain, the first child of any body -- hands
        // its sum over in a register: the same additions in the same order, without a dependent LDS write -> read per body)
        {
            int *PARL = reinterpret_cast<int *>(TROW);
            for (int b = tid; b < nb; b += WIDE_THREADS) PARL[b] = M.parent[b];
            __syncthreads();
            for (int e = tid; e < 84; e += WIDE_THREADS) {
                double *arr = e < 36 ? AC : e < 72 ? MC : WC;
                const int st = e < 72 ? 36 : 12, off = e < 36 ? e : e < 72 ? e - 36 : e - 72;
                double carry = 0., cur = arr[st * (nb - 1) + off];
                int carry_to = -1, par = PARL[nb - 1];
                for (int b = nb - 1; b >= 1; --b) {
                    const double nxt = arr[st * (b - 1) + off];          // (nothing below writes entry b - 1 in this iteration)
                    const int par_nxt = PARL[b - 1];
                    const double v = carry_to == b ? cur + carry : cur;
                    if (carry_to == b) arr[st * b + off] = v;
                    if (par == b - 1) { carry = v; carry_to = par; }
                    else { if (par >= 0) arr[st * par + off] += v; carry_to = -1; }
                    cur = nxt; par = par_nxt;
                }
                if (carry_to == 0) arr[off] = cur + carry;
            }
        }
        __syncthreads();
        // ---- lane = dof k: own column in world axes, the products with the composites of body(k) -----------------------
        for (int k = tid; k < n; k += WIDE_THREADS) {
            const int b = M.dofbody[k];
            const M3<double> R = ldm(POSE + 12 * b);
            const V3<double> p = ld3(POSE + 12 * b + 9);
            const double *sc = SC + 12 * k;
            const V3<double> sw = ld3(sc), sv = ld3(sc + 3), dsw = ld3(sc + 6), dsv = ld3(sc + 9);
            const V3<double> okw = ld3(OM + 6 * b), okv = ld3(OM + 6 * b + 3);
            const V3<double> xw = mv(R, sw), xv = mv(R, sv) + cross(p, xw);
            const V3<double> aw2 = dsw - cross(okw, sw), av2 = dsv - cross(okv, sw) - cross(okw, sv);
            const V3<double> dw = mv(R, aw2), dv = mv(R, av2) + cross(p, dw);
            const double X[6] = {xw.x, xw.y, xw.z, xv.x, xv.y, xv.z}, dX[6] = {dw.x, dw.y, dw.z, dv.x, dv.y, dv.z};
            const double *Ac = AC + 36 * b, *Mc = MC + 36 * b, *Wc = WC + 12 * b;
            double *xk = XK + WIDE_XK * k;
            double rm = 0., rg = 0.;
            for (int r = 0; r < 6; ++r) {
                double g = 0., pp = 0., rr = 0.;
                for (int c2 = 0; c2 < 6; ++c2) {
                    g += Ac[6 * r + c2] * X[c2] + Mc[6 * r + c2] * dX[c2];
                    pp += Ac[6 * c2 + r] * X[c2];
                    rr += Mc[6 * r + c2] * X[c2];
                }
                xk[r] = X[r]; xk[6 + r] = dX[r]; xk[12 + r] = pp; xk[18 + r] = rr; xk[24 + r] = g;
                rm += X[r] * Wc[r]; rg += X[r] * Wc[6 + r];
            }
            xk[30] = (double)b; xk[31] = (double)(b + M.subsize[b]);
            RH[k] = rm; RH[n + k] = rg;
        }
        __syncthreads();
        // inspect: body Jacobians J_b = Ad(b<-g) X, dJ_b = Ad(b<-g) dX' + ad(Om_b) J_b over the dofs of b's ancestors (see arb_phase_b.h)
        if (io.inspect && step == 0 && (io.jac != nullptr || io.djac != nullptr)) {
            for (int e = tid; e < nb * n; e += WIDE_THREADS) {
                const int b = e / n, k = e - b * n;
                double j6[6] = {0., 0., 0., 0., 0., 0.}, e6[6] = {0., 0., 0., 0., 0., 0.};
                if (anc_eq(M.dofbody[k], b)) {
                    const M3<double> R = ldm(POSE + 12 * b);
                    const V3<double> p = ld3(POSE + 12 * b + 9), obw = ld3(OM + 6 * b), obv = ld3(OM + 6 * b + 3);
                    const double *xk = XK + WIDE_XK * k;
                    const V3<double> xw = v3<double>(xk[0], xk[1], xk[2]), xv = v3<double>(xk[3], xk[4], xk[5]);
                    const V3<double> dw = v3<double>(xk[6], xk[7], xk[8]), dv = v3<double>(xk[9], xk[10], xk[11]);
                    const V3<double> jw = mtv(R, xw), jv = mtv(R, xv - cross(p, xw));
                    const V3<double> ew = mtv(R, dw) + cross(obw, jw);
                    const V3<double> ev = mtv(R, dv - cross(p, dw)) + cross(obv, jw) + cross(obw, jv);
                    j6[0] = jw.x; j6[1] = jw.y; j6[2] = jw.z; j6[3] = jv.x; j6[4] = jv.y; j6[5] = jv.z;
                    e6[0] = ew.x; e6[1] = ew.y; e6[2] = ew.z; e6[3] = ev.x; e6[4] = ev.y; e6[5] = ev.z;
                }
                for (int i = 0; i < 6; ++i) {
                    if (io.jac != nullptr) io.jac[(((long)w * nb + b) * 6 + i) * n + k] = (T)j6[i];
                    if (io.djac != nullptr) io.djac[(((long)w * nb + b) * 6 + i) * n + k] = (T)e6[i];
                }
            }
        }
        __syncthreads();            // (last reads of the chain arrays: Z takes their LDS space now)
        if (io.inspect && io.stamps != nullptr && tid == 0) io.stamps[w * 8 + 3] = (long long)clock64();
        // ---- the augmented system [Z | rhs | J'^T]: lane = entry (compact build: assembled in scratch, then into the registers) ----
        {
        // lane = column (its per-dof vectors in registers), rows strided; both products of a row are formed, whichever applies
        // is kept: the row's vectors are loaded in one batch (the same for every lane of a wavefront: LDS broadcasts)
        for (int c = cl_n; c < n; c += cwn) {
            const double *xc = XK + WIDE_XK * c;
            double x0[12], xg[6];
            for (int r = 0; r < 12; ++r) x0[r] = xc[r];
            for (int r = 0; r < 6; ++r) xg[r] = xc[24 + r];
            const double bk = xc[30], bke = xc[31];              // (the body of the column's dof, the end of its subtree)
            const auto entry = [&](const int i) {
                const double *xi = XK + WIDE_XK * i;
                double a[6], pp[6], rr[6];
                for (int r = 0; r < 6; ++r) { a[r] = xi[r]; pp[r] = xi[12 + r]; rr[r] = xi[18 + r]; }
                const double bi = xi[30], bie = xi[31];
                double v1 = 0., v2 = 0.;
                for (int r = 0; r < 6; ++r) v1 += a[r] * xg[r];                              // own or an ancestor's dof: X_i . G_k
                for (int r = 0; r < 6; ++r) v2 += pp[r] * x0[r] + rr[r] * x0[6 + r];         // a descendant's: P_i . X_k + R_i . dX'_k
                double v = (bi <= bk && bk < bie) ? v1 : (bk <= bi && bi < bke) ? v2 : 0.;
                if (io.zmode == 0) {
                    if (io.pd_kp != nullptr) { if (i == c) v += dt * (double)io.pd_kp[w * n + i] + (double)io.pd_kd[w * n + i]; }
                    else if (M.has_pd) v += dt * M.pd_kp[i * n + c] + M.pd_kd[i * n + c];     // controllers.py:141-158
                    if (io.zimp != nullptr) v -= (double)io.zimp[((long)w * n + i) * n + c];  // core.py:815-817
                }
                return v;
            };
            // (four rows side by side: a row's sums are dependent chains of float64 operations, ~900 cycles a row one at a time)
            int i = r0_n;
            for (; i + 3 * rs_n < n; i += 4 * rs_n) {
                const double v0 = entry(i), v1 = entry(i + rs_n), v2 = entry(i + 2 * rs_n), v3 = entry(i + 3 * rs_n);
                Z[i * ld + c] = v0; Z[(i + rs_n) * ld + c] = v1; Z[(i + 2 * rs_n) * ld + c] = v2; Z[(i + 3 * rs_n) * ld + c] = v3;
            }
            for (; i < n; i += rs_n) Z[i * ld + c] = entry(i);
        }
        if (io.inspect && io.zmode != 0) {           // the world matrices M, B, N one by one (the object API): Zout and on to the next world
            __syncthreads();
            if (io.Zout != nullptr) for (int e = tid; e < n * n; e += WIDE_THREADS) io.Zout[(long)w * n * n + e] = (T)Z[(e / n) * ld + (e % n)];
            break;
        }
        // constraint rows s_k [Ad(c0<-g) X_k] (constraints.py:429-433, 203-207, 46-48): rows of J' in JR, columns of J'^T in Z
        if (do_con) for (int e = tid; e < nda * n; e += WIDE_THREADS) {
            const int idx = e / n, k = e - idx * n, c = ORDL[idx >> 2], r = idx & 3;
            const double *cd = CD + WIDE_CD * c;
            const int ct = M.ctype[c];
            double v = 0.;
            {
                if (ct == ARB_CT_JOINTLIMITS) {
                    v = (r == 0 && k == M.cdof[c]) ? 1. : 0.;
                } else if (r < (ct == ARB_CT_SOFTFINGER ? 4 : 3)) {
                    const int bk = M.dofbody[k];
                    const double s = (anc_eq(bk, M.cbody[c]) ? 1. : 0.) - (anc_eq(bk, M.cbody0[c]) ? 1. : 0.);
                    if (s != 0.) {
                        const double *xk = XK + WIDE_XK * k;
                        const M3<double> Rx = ldm(cd);
                        const V3<double> cw = mv(Rx, v3<double>(xk[0], xk[1], xk[2]));
                        const V3<double> cv = mv(Rx, v3<double>(xk[3], xk[4], xk[5])) + cross(ld3(cd + 9), cw);
                        if (ct == ARB_CT_SOFTFINGER) v = s * (r == 0 ? cw.z : r == 1 ? cv.x : r == 2 ? cv.y : cv.z);
                        else v = s * (r == 0 ? cv.x : r == 1 ? cv.y : cv.z);
                    }
                }
            }
            JR[idx * n + k] = v;
            Z[k * ld + n + 1 + idx] = v;
        }
        }
        __syncthreads();
        // right-hand side of the increment form: gforce - (N + B - Z_a) gvel (+ J'^T f0, core.py:921-924)
        for (int i = tid; i < n; i += WIDE_THREADS) {
            double ext = 0.;
            if (io.ext != nullptr) ext = (double)io.ext[(long)step * io.ext_stride + w * n + i];
            double rhs = RH[i] + ext, gf0 = RH[n + i] + ext;
            const long pdo = (long)step * io.pd_stride + w * n;       // (this step's PD targets)
            if (io.pd_kp != nullptr) {             // per-world diagonal gains and targets: tau0 = kp (qdes - q) + kd dqdes, Z += dt kp + kd
                const double kp = (double)io.pd_kp[w * n + i], kd = (double)io.pd_kd[w * n + i];
                const double acc = kp * ((double)io.pd_qdes[pdo + i] - QD[i]) + kd * (double)io.pd_dqdes[pdo + i];
                rhs += acc - (dt * kp + kd) * DQS[i]; gf0 += acc;
            } else if (M.has_pd) {
                double acc = io.pd_qdes != nullptr ? 0. : M.pd_tau0[i], accv = 0.;
                for (int j = 0; j < n; ++j) {
                    const double kp = M.pd_kp[i * n + j], kd = M.pd_kd[i * n + j];
                    if (io.pd_qdes != nullptr) {
                        if (kp != 0. || kd != 0.) acc += kp * ((double)io.pd_qdes[pdo + j] - QD[j]) + kd * (double)io.pd_dqdes[pdo + j];
                    } else acc -= kp * QD[j];
                    accv += (dt * kp + kd) * DQS[j];
                }
                rhs += acc - accv; gf0 += acc;
            }
            if (io.zimp != nullptr) {
                double accz = 0.;
                for (int j = 0; j < n; ++j) accz += (double)io.zimp[((long)w * n + i) * n + j] * DQS[j];
                rhs += accz;
            }
            if (do_con && M.has_warm) for (int idx = 0; idx < nda; ++idx) rhs += JR[idx * n + i] * FF[4 * ORDL[idx >> 2] + (idx & 3)];
            if (io.inspect) {
                if (io.gforce0 != nullptr) io.gforce0[w * n + i] = (T)gf0;
                if (io.Zout != nullptr) for (int c = 0; c < n; ++c) io.Zout[((long)w * n + i) * n + c] = (T)Z[i * ld + c];
                RH[n + i] = gf0;
            }
            Z[i * ld + n] = rhs;
        }
        __syncthreads();
        if (io.inspect && io.stamps != nullptr && tid == 0) io.stamps[w * 8 + 4] = (long long)clock64();
        // ================= phase C: pivot-free Gauss-Jordan, pivots from the last dof to the first (core.py:818) ==========
        const int nact = n + 1 + nda;                 // (the live columns: dofs, right-hand side, the active constraints' four each)
        if constexpr (REGZ) {
            // (LDS offsets as integers: the difference of two generic pointers into LDS is an expression the back end mishandles)
            const int zl0 = ((ncols + 3) & ~3) + ((n + 3) & ~3) + 8 + 48 + 2 * ((nds + 3) & ~3) + 59 * ncap + (ncap & 1) + 2 * ((ncap + 5) >> 2);
            wide_eliminate<KMAX, CP>(Z, ld, n, nact, zl0, zl0 + (int)M.l_reg + (int)M.l_sol, sld, DQS, M.sol_in_lds ? nullptr : S + M.o_sol);
        } else {
        for (int j = n - 1; j >= 0; --j) {
            const double ip = arb_rcp(Z[j * ld + j]);
            // (columns j+1 .. n-1 of row j are zero by now: their pivots are taken)
            for (int c = tid; c < nact; c += WIDE_THREADS) TROW[c] = (c <= j || c >= n) ? Z[j * ld + c] * ip : 0.;
            for (int r = tid; r < n; r += WIDE_THREADS) FCOL[r] = (r == j) ? 0. : Z[r * ld + j];
            __syncthreads();
            // live columns: 0 .. j and n .. nact - 1 (TROW is zero in between), compacted; lane = live column x row stride: the
            // column lanes are the next power of two of the live count (a shift and a mask per lane and pivot, no division per entry)
            const int ncj = (j + 1) + (nact - n);
            int sh = 0;
            while ((1 << sh) < ncj && sh < 8) ++sh;
            const int cwj = 1 << sh, clj = tid & (cwj - 1), r0j = tid >> sh, rsj = WIDE_THREADS >> sh;
            // (rows four at a time: the four multipliers and the four entries are loaded before the first is stored -- the loop is
            // a chain of LDS round trips otherwise, one per entry: 10.9 k cycles per pivot of snake-100, 64 % of its step)
            for (int cc = clj; cc < ncj; cc += cwj) {
                const int c = cc <= j ? cc : n + (cc - j - 1);
                const double t = TROW[c];
                double *zc = Z + c;
                int r = r0j;
                for (; r + 3 * rsj < n; r += 4 * rsj) {
                    const int r1 = r + rsj, r2 = r + 2 * rsj, r3 = r + 3 * rsj;
                    const double f0 = FCOL[r], f1 = FCOL[r1], f2 = FCOL[r2], f3 = FCOL[r3];      // (FCOL[j] is 0)
                    const double z0 = zc[r * ld], z1 = zc[r1 * ld], z2 = zc[r2 * ld], z3 = zc[r3 * ld];
                    zc[r * ld] = (r == j) ? t : z0 - f0 * t;
                    zc[r1 * ld] = (r1 == j) ? t : z1 - f1 * t;
                    zc[r2 * ld] = (r2 == j) ? t : z2 - f2 * t;
                    zc[r3 * ld] = (r3 == j) ? t : z3 - f3 * t;
                }
                for (; r < n; r += rsj) {
                    const double f = FCOL[r], z = zc[r * ld];
                    zc[r * ld] = (r == j) ? t : z - f * t;
                }
            }
            __syncthreads();
        }
        // the rhs column holds gvel+ - gvel: add gvel back so that it is Y (M gvel/dt + gforce)
        for (int i = tid; i < n; i += WIDE_THREADS) Z[i * ld + n] += DQS[i];
        __syncthreads();
        }
        if (io.inspect && io.vel_free != nullptr) for (int i = tid; i < n; i += WIDE_THREADS) io.vel_free[w * n + i] = (T)SL[i * sld];
        if (io.inspect && io.stamps != nullptr && tid == 0) io.stamps[w * 8 + 5] = (long long)clock64();
        // ================= phase D: [v' | Y'] = J' [Y rhs | Y J'^T] (core.py:925-927), block inverses =========================
        if (do_con) {
            for (int e = tid; e < nda * (nda + 1); e += WIDE_THREADS) {
                const int idx = e / (nda + 1), c = e - idx * (nda + 1);
                double acc = 0.;
                {
                    // (eight terms' operands in flight at a time -- the rows of J' and the solution columns of a world with many
                    // constraints live in scratch, a dependent load per term was 400 cycles a term --, added in the same order)
                    const double *jr = JR + idx * n, *sl = SL + c;
                    int k = 0;
                    for (; k + 8 <= n; k += 8) {
                        double a[8], b[8];
                        for (int u = 0; u < 8; ++u) { a[u] = jr[k + u]; b[u] = sl[(k + u) * sld]; }
                        for (int u = 0; u < 8; ++u) acc += a[u] * b[u];
                    }
                    for (; k < n; ++k) acc += jr[k] * sl[k * sld];
                }
                if (c == 0) VV[idx] = acc; else AM[idx * ams + (c - 1)] = acc;
            }
            __syncthreads();
            if (io.inspect) {
                // (by constraint, as the caller registered them: zero where a constraint is not active)
                if (io.c_adm != nullptr) {
                    for (int i = tid; i < ndol * ndol; i += WIDE_THREADS) io.c_adm[(long)w * ndol * ndol + i] = T(0);
                    __syncthreads();
                    for (int e = tid; e < nda * nda; e += WIDE_THREADS) {
                        const int i = e / nda, j2 = e - i * nda;
                        io.c_adm[(long)w * ndol * ndol + (long)(4 * ORDL[i >> 2] + (i & 3)) * ndol + 4 * ORDL[j2 >> 2] + (j2 & 3)] = (T)AM[i * ams + j2];
                    }
                }
                if (io.c_vel != nullptr) {
                    for (int i = tid; i < ndol; i += WIDE_THREADS) io.c_vel[(long)w * ndol + i] = T(0);
                    __syncthreads();
                    for (int i = tid; i < nda; i += WIDE_THREADS) io.c_vel[(long)w * ndol + 4 * ORDL[i >> 2] + (i & 3)] = (T)VV[i];
                }
                if (io.c_jac != nullptr) {
                    for (int i = tid; i < ndol * n; i += WIDE_THREADS) io.c_jac[(long)w * ndol * n + i] = T(0);
                    __syncthreads();
                    for (int e = tid; e < nda * n; e += WIDE_THREADS) {
                        const int i = e / n, k = e - i * n;
                        io.c_jac[(long)w * ndol * n + (long)(4 * ORDL[i >> 2] + (i & 3)) * n + k] = (T)JR[e];
                    }
                }
            }
            for (int sl = tid; sl < nca; sl += WIDE_THREADS) {
                const int c = ORDL[sl];
                double *cd = CD + WIDE_CD * c;
                const int ct = M.ctype[c], nd = (ct == ARB_CT_SOFTFINGER) ? 4 : (ct == ARB_CT_BALLSOCKET ? 3 : 1);
                double P[16];
                if (!inv_block<double>(AM + (4 * sl) * ams + 4 * sl, ams, nd, P)) pinv_block<double>(AM + (4 * sl) * ams + 4 * sl, ams, nd, P);
                for (int i = 0; i < 16; ++i) cd[24 + i] = P[i];
            }
            __syncthreads();
            if (io.inspect && io.stamps != nullptr && tid == 0) io.stamps[w * 8 + 6] = (long long)clock64();
            // ---- 20 Gauss-Seidel sweeps, constraints in registration order (core.py:929-935).  The sweeps are a dependent chain:
            // the first wavefront runs them alone, everything it touches per solve in LDS and hand-overs by wave-level ordering --
            // no workgroup barrier inside the 20 x nc solves; the other three wavefronts wait at the barrier below.  Per
            // constraint in LDS (GSC): its admittance block (16) | the block's inverse (16) | sdist, mu, eps (3), pos0 / dt (3),
            // glo, ghi, type, active | constants of the sliding solve (6), its last root.
            //
            // INDEPENDENT GROUPS of constraints are swept side by side.  Two constraints interact through the 4 x 4 blocks of the
            // admittance Y' that couple them; where both blocks are exactly zero -- constraints on different kinematic trees that
            // touch nothing but the ground: the feet of a robot and the objects lying around it -- neither solve reads anything
            // the other writes (the update vel += Y'[:, c] dforce adds exact zeros to the other's rows), so the order between them
            // is immaterial.  The groups are the connected components of "some coupling block is non-zero", found per step; lane
            // g (the lowest constraint of its group) sweeps the group's constraints in registration order, all groups in lockstep:
            // a round = one solve per group, then every row takes the update of its own group's constraint.  A row receives the
            // same additions in the same order as in the serial sweep minus the exact zeros: the same bits (io.gs_serial, the
            // knob "wide_gs_groups" 0: the serial sweep, which the tests compare with).  human36 beside four objects: rounds of
            // four solves instead of eight.
            if (tid < WAVE) {
                // (from here on c, c2 are SLOTS: the step's active constraints in registration order)
                for (int c = tid; c < nca; c += WAVE) {
                    const int cc = ORDL[c];                                      // (the constraint in this slot)
                    const double *cd = CD + WIDE_CD * cc;
                    double *g = GSC + 52 * c;
                    for (int i = 0; i < 4; ++i) for (int j2 = 0; j2 < 4; ++j2) g[4 * i + j2] = AM[(4 * c + i) * ams + 4 * c + j2];
                    for (int i = 0; i < 16; ++i) g[16 + i] = cd[24 + i];
                    g[32] = cd[15]; g[33] = M.cmu[cc]; g[34] = M.ceps[3 * cc]; g[35] = M.ceps[3 * cc + 1]; g[36] = M.ceps[3 * cc + 2];
                    g[37] = cd[12] * inv_dt; g[38] = cd[13] * inv_dt; g[39] = cd[14] * inv_dt; g[40] = cd[17]; g[41] = cd[18];
                    g[42] = (double)M.ctype[cc]; g[43] = cd[16];
                    // per-step constants of the sliding solve (admittance-only part of the sextic) and the warm start of its root
                    // finder, as the wavefront kernels keep them per contact (arb_gs_stage.h): [44..49] SlidePre, [50] last root
                    if (M.ctype[cc] == ARB_CT_SOFTFINGER) {
                        const SlidePre sp = slide_precompute<double>(g);
                        g[44] = sp.tr; g[45] = sp.m2; g[46] = sp.det; g[47] = sp.sQ; g[48] = sp.sA; g[49] = sp.nq;
                    }
                    g[50] = NAN;
                    // the constraints this one is coupled with (itself included), a bit each
                    unsigned long long m = 1ull << c;
                    if (!io.gs_serial) {
                        for (int c2 = 0; c2 < nca; ++c2) {
                            if (c2 == c) continue;
                            bool nz = false;
                            for (int i = 0; i < 4; ++i) for (int j2 = 0; j2 < 4; ++j2)
                                nz = nz || AM[(4 * c + i) * ams + 4 * c2 + j2] != 0. || AM[(4 * c2 + i) * ams + 4 * c + j2] != 0.;
                            if (nz) m |= 1ull << c2;
                        }
                    } else {
                        m = nca >= 64 ? ~0ull : (1ull << nca) - 1ull;            // (one group)
                    }
                    GGM[c] = m;
                }
                for (int r = tid; r < nda; r += WAVE) { GVV[r] = VV[r]; GFF[r] = FF[4 * ORDL[r >> 2] + (r & 3)]; }
                WAVE_SYNC();
                // connected components: every mask takes over the masks of its members, six times (2^6 >= 64 slots)
                for (int it = 0; it < 6; ++it) {
                    unsigned long long m = tid < nca ? GGM[tid] : 0ull, m2 = m;
                    for (unsigned long long rest = m; rest; rest &= rest - 1) m2 |= GGM[__builtin_ctzll(rest)];
                    WAVE_SYNC();
                    if (tid < nca) GGM[tid] = m2;
                    WAVE_SYNC();
                }
            }
            __syncthreads();
            // Independent groups share nothing -- not the rows of v', not the forces, not the fixed-point test (a group whose sweep
            // changed nothing would repeat itself bit for bit whatever the others do): the FOUR wavefronts take the groups in turn
            // (group k of the step on wavefront k mod 4) and sweep them without a word to one another; one workgroup barrier at the
            // end.  human36 beside four objects: the feet on one wavefront, the objects' contacts on the others.
            {
                const int wv = tid >> 6, ln = tid & (WAVE - 1);
                // (lane c < nca: leads its group when it is the group's lowest slot; the groups are numbered by their leaders)
                const unsigned long long grp = ln < nca ? GGM[ln] : 0ull;
                const bool lead = grp != 0ull && __builtin_ctzll(grp) == ln;
                const unsigned long long leaders = __ballot(lead);
                const unsigned long long mine = (lead && (__popcll(leaders & ((1ull << ln) - 1ull)) & 3) == wv) ? grp : 0ull;
                for (int sweep = 0; sweep < GS_SWEEPS; ++sweep) {
                    if (ln == 0) DF[wv] = 0.;                                    // "something changed in this sweep (of this wavefront's groups)"
                    WAVE_SYNC();
                    unsigned long long rem = mine;
                    while (__any(rem != 0ull)) {
                        if (mine != 0ull) GDF[6 * ln + 4] = -1.;                 // (no update from this group in this round)
                        if (rem != 0ull) {
                            const int c = __builtin_ctzll(rem);
                            rem &= rem - 1;
                            const double *g = GSC + 52 * c;
                            // (the constraint's block out of LDS in ONE batch: read where the solve uses them, every value is a
                            // round trip of its own -- one wavefront, nothing to hide it behind)
                            double gl[52];
                            for (int i = 0; i < 52; ++i) gl[i] = g[i];
                            const int ct = (int)gl[42];
                            double v[4], f[4], f_old[4], df[4] = {0., 0., 0., 0.};
                            for (int i = 0; i < 4; ++i) { v[i] = GVV[4 * c + i]; f[i] = f_old[i] = GFF[4 * c + i]; }
                            __builtin_amdgcn_sched_barrier(0);
                            if (ct == ARB_CT_SOFTFINGER) {                       // constraints.py:780-836
                                const double eps[3] = {gl[34], gl[35], gl[36]};
                                double *gm = GSC + 52 * c;
                                const SlidePre sp = {gl[44], gl[45], gl[46], gl[47], gl[48], gl[49]};
                                double alpha[4], shift = 0., warm = gl[50], swork[48];
                                int br = softfinger_try<double>(v, gl, gl + 16, f, df, gl[32], dt, gl[33], eps, swork, alpha, &shift, true, &sp, &warm);
                                if (br == 3) { shift = slide_shift_from_eig<double>(swork); br = 2; warm = NAN; }
                                if (br == 2) softfinger_slide_finish<double>(gl, alpha, eps, shift, f, df);
                                if (br == 2) gm[50] = warm;          // (the next sweep restarts next to this root)
                            } else if (ct == ARB_CT_BALLSOCKET) {                // constraints.py:235-237
                                const double *P = gl + 16;
                                for (int i = 0; i < 3; ++i) {
                                    df[i] = -(P[4 * i] * (v[0] + gl[37]) + P[4 * i + 1] * (v[1] + gl[38]) + P[4 * i + 2] * (v[2] + gl[39]));
                                    f[i] += df[i];
                                }
                            } else {                                             // JointLimits.solve constraints.py:73-90
                                const double v0 = v[0] - gl[0] * f[0], p00 = gl[16];
                                double nf = 0.;
                                if (v0 <= gl[40]) nf = p00 * (gl[40] - v0);
                                else if (gl[41] <= v0) nf = p00 * (gl[41] - v0);
                                df[0] = nf - f[0]; f[0] = nf;
                            }
                            bool ch = false;
                            for (int i = 0; i < 4; ++i) { GFF[4 * c + i] = f[i]; GDF[6 * ln + i] = df[i]; ch = ch || df[i] != 0. || !same_bits(f[i], f_old[i]); }
                            GDF[6 * ln + 4] = (double)c;
                            if (ch) DF[wv] = 1.;
                        }
                        WAVE_SYNC();
                        for (int r = ln; r < nda; r += WAVE) {                    // vel += Y'[:, c] dforce   core.py:935
                            const int ld_ = __builtin_ctzll(GGM[r >> 2]);        // (the row's group; this wavefront's?)
                            if ((__popcll(leaders & ((1ull << ld_) - 1ull)) & 3) != wv) continue;
                            const double *gd = GDF + 6 * ld_;
                            const int c = (int)gd[4];
                            if (c < 0) continue;
                            const double *a = AM + r * ams + 4 * c;
                            GVV[r] += a[0] * gd[0] + a[1] * gd[1] + a[2] * gd[2] + a[3] * gd[3];
                        }
                        WAVE_SYNC();
                    }
                    // a sweep that changes no force and adds nothing to any velocity is a fixed point of the iteration: the
                    // remaining sweeps of core.py:929-935 would repeat it bit for bit
                    if (DF[wv] == 0.) break;
                    WAVE_SYNC();
                }
            }
            __syncthreads();
            for (int r = tid; r < nda; r += WIDE_THREADS) { VV[r] = GVV[r]; FF[4 * ORDL[r >> 2] + (r & 3)] = GFF[r]; }
            __syncthreads();
        }
        if (io.inspect && io.stamps != nullptr && tid == 0) io.stamps[w * 8 + 7] = (long long)clock64();
        // ================= phase E: new velocity, integrate (core.py:974-980, joints.py:54-57) ================================
        if (io.inspect) {
            if (io.gforce != nullptr) for (int i = tid; i < n; i += WIDE_THREADS) {
                double g = RH[n + i];
                if (do_con) for (int idx = 0; idx < nda; ++idx) g += JR[idx * n + i] * FF[4 * ORDL[idx >> 2] + (idx & 3)];
                io.gforce[w * n + i] = (T)g;
            }
            if (io.c_force != nullptr) for (int i = tid; i < ndol; i += WIDE_THREADS) io.c_force[w * ndol + i] = (T)FF[i];
            for (int c = tid; c < nc; c += WIDE_THREADS) {
                if (io.c_sdist != nullptr) io.c_sdist[w * nc + c] = do_con ? (T)CD[WIDE_CD * c + 15] : T(0);
                if (io.c_active != nullptr) io.c_active[w * nc + c] = (do_con && CD[WIDE_CD * c + 16] != 0.) ? 1 : 0;
            }
        }
        for (int i = tid; i < n; i += WIDE_THREADS) {
            double vnew = SL[i * sld];
            if (do_con) for (int idx = 0; idx < nda; ++idx) { const int fi = 4 * ORDL[idx >> 2] + (idx & 3); vnew += SL[i * sld + 1 + idx] * (FF[fi] - FF0[fi]); }
            RH[i] = vnew;
        }
        __syncthreads();
        for (int i = tid; i < n; i += WIDE_THREADS) {
            const double vnew = RH[i];
            DQS[i] = vnew;
            const int qi = M.dof2q[i];
            if (qi >= 0) QS[qi] += dt * vnew;                                   // core.py:238-240
        }
        __syncthreads();
        if (io.cost_out != nullptr) {              // running cost on the state after this step, this step's torques
            double c = 0.;
            for (int i = tid; i < n; i += WIDE_THREADS) {
                const int qi = M.dof2q[i];
                const double dd = (qi >= 0 ? QS[qi] : 0.) - (io.cost_qref != nullptr ? (double)io.cost_qref[i] : 0.);
                const double vv = DQS[i];
                const double tau = io.ext != nullptr ? (double)io.ext[(long)step * io.ext_stride + w * n + i] : 0.;
                c += (io.cost_wq != nullptr ? (double)io.cost_wq[i] : 0.) * dd * dd + (io.cost_wdq != nullptr ? (double)io.cost_wdq[i] : 0.) * vv * vv
                   + (io.cost_wtau != nullptr ? (double)io.cost_wtau[i] : 0.) * tau * tau;
            }
            for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off);
            if ((tid & 63) == 0) FCOL[tid >> 6] = c;
            __syncthreads();
            if (tid == 0) io.cost_out[w] = (T)((double)io.cost_out[w] + ((FCOL[0] + FCOL[1]) + (FCOL[2] + FCOL[3])));
            __syncthreads();
        }
        for (int b = tid; b < nb; b += WIDE_THREADS) {
            if (M.jtype[b] != JT_FREE) continue;
            double *qp = QS + M.q_off[b];
            const double *vp = DQS + M.dof_off[b];
            M3<double> R, Re; V3<double> p, pe;
            R.a[0] = qp[0]; R.a[1] = qp[1]; R.a[2] = qp[2]; p.x = qp[3];
            R.a[3] = qp[4]; R.a[4] = qp[5]; R.a[5] = qp[6]; p.y = qp[7];
            R.a[6] = qp[8]; R.a[7] = qp[9]; R.a[8] = qp[10]; p.z = qp[11];
            exp_twist<double>(dt * v3<double>(vp[0], vp[1], vp[2]), dt * v3<double>(vp[3], vp[4], vp[5]), Re, pe);
            const M3<double> Rn = mul(R, Re);
            const V3<double> pn = mv(R, pe) + p;
            qp[0] = Rn.a[0]; qp[1] = Rn.a[1]; qp[2] = Rn.a[2]; qp[3] = pn.x;
            qp[4] = Rn.a[3]; qp[5] = Rn.a[4]; qp[6] = Rn.a[5]; qp[7] = pn.y;
            qp[8] = Rn.a[6]; qp[9] = Rn.a[7]; qp[10] = Rn.a[8]; qp[11] = pn.z;
            qp[12] = 0.; qp[13] = 0.; qp[14] = 0.; qp[15] = 1.;
        }
        __syncthreads();
    }   // steps
    if (io.inspect) {
        if (io.q_next != nullptr) for (int i = tid; i < nq; i += WIDE_THREADS) io.q_next[w * nq + i] = (T)QS[i];
        if (io.dq_next != nullptr) for (int i = tid; i < n; i += WIDE_THREADS) io.dq_next[w * n + i] = (T)DQS[i];
    } else {
        for (int i = tid; i < nq; i += WIDE_THREADS) io.q[w * nq + i] = (T)QS[i];
        for (int i = tid; i < n; i += WIDE_THREADS) io.dq[w * n + i] = (T)DQS[i];
        if (io.cf != nullptr) for (int i = tid; i < ndol; i += WIDE_THREADS) io.cf[w * ndol + i] = (T)FF[i];
    }
    }   // worlds
}

// One launch of one instantiation.  In the split build (csrc/Makefile) every KMAX is a translation unit of its own
// (-DARB_PART -DARB_PART_WIDE=KMAX), the host unit sees `extern template` declarations.
template <typename T, int KMAX, int CP>
hipError_t wide_launch_one(const WideModel *dev, const WideIO<T> &io, long nw, double dt, const double *dts, int nsteps, unsigned flags,
                           double *ws, unsigned grid, size_t lds, hipStream_t st) {
    auto kern = arb_wide_kernel<T, KMAX, CP>;
    if (lds > 64 * 1024) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WIDE_THREADS), lds, st, dev, io, nw, dt, dts, nsteps, flags, ws);
    return hipGetLastError();
}
#define ARB_WIDE_ONE_ARGS(T) const WideModel *, const WideIO<T> &, long, double, const double *, int, unsigned, double *, unsigned, size_t, hipStream_t
#if defined(ARB_PART_WIDE)
template hipError_t wide_launch_one<float, ARB_PART_WIDE, ARB_PART_WIDE_CP>(ARB_WIDE_ONE_ARGS(float));
template hipError_t wide_launch_one<double, ARB_PART_WIDE, ARB_PART_WIDE_CP>(ARB_WIDE_ONE_ARGS(double));
#elif defined(ARB_SPLIT_BUILD)
#define ARB_EXTERN_WIDE(K, P) \
    extern template hipError_t wide_launch_one<float, K, P>(ARB_WIDE_ONE_ARGS(float)); \
    extern template hipError_t wide_launch_one<double, K, P>(ARB_WIDE_ONE_ARGS(double));
ARB_EXTERN_WIDE(0, 2) ARB_EXTERN_WIDE(20, 2) ARB_EXTERN_WIDE(28, 2) ARB_EXTERN_WIDE(32, 2) ARB_EXTERN_WIDE(20, 4) ARB_EXTERN_WIDE(28, 4) ARB_EXTERN_WIDE(32, 4)
ARB_EXTERN_WIDE(40, 4) ARB_EXTERN_WIDE(48, 4) ARB_EXTERN_WIDE(20, 6) ARB_EXTERN_WIDE(28, 6) ARB_EXTERN_WIDE(32, 6)
#undef ARB_EXTERN_WIDE
#endif
#endif  // ARB_WIDE_KERNEL_H
