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
#include "arb_config.h"
#include "arb_device.h"
#include "arb_gs_stage.h"
#include "arb_step_kernel.h"
#include "arb_aux_kernels.h"
#if (!defined(ARB_PART) || defined(ARB_PART_WIDE)) && !defined(ARB_QUICK)
#include "arb_wide_kernel.h"
#define ARB_WITH_WIDE 1
#else
#define ARB_WITH_WIDE 0
#endif
#include "arb_launch.h"

#ifndef ARB_PART
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
    bool spec_ok = false;          // four constraints per column set, all enabled SoftFingerContacts of plane / sphere pairs, no PD controller, no viscosity,
                                   // one world per wavefront, tiles 44 / 48: the specialised kernels (FEAT bit 4)
    bool spec0_ok = false;         // no constraints, otherwise the same class: the specialised kernels of FEAT bit 8
    bool bodycols = false;         // the same class with more contacts than one column set holds, on few pairs of bodies: body-space
                                   // constraint columns (FEAT bit 16), ONE column set -- human36 with the reference's eight contact points
    bool bodycols_default = false; // ... chosen without being asked (ARB_STEP_BODY_COLUMNS): when they save the second column set
    // What float32 launches of this model run by default (arb_model_info.mixed_default), from the pivot growth of the float32
    // elimination at the model's rest states (probed once at arb_model_create): 0 the float32 kernels; 1 (growth >
    // ARB_ILLCOND_GROWTH / 8) the mixed build, CM 3: float64 elimination and right-hand side; 2 (growth > ARB_ILLCOND_GROWTH:
    // ARB_WARN_ILLCOND territory) PROMOTION: the float64 kernels on converted copies of the buffers (step_promoted)
    int f32_policy = 0;
    bool mixed_default = false;    // (f32_policy >= 1: the dispatch of launch())
    float rest_growth = 0.f;
    Layout lfb, lfb3, ldb;         // ... and their LDS layouts: float32 two-wave / three-wave, float64
    int *status_host = nullptr;    // mapped pinned words the kernels raise: [0] a work-queue wait expired (ARB_ERR_STALLED), [1] ARB_WARN_* bits
    Knobs kn;                      // development / test knobs (arb_hook_set_knob)
    Layout lf, lf3, ld;            // LDS layouts: float32 two-wave kernels, three-wave kernels; float64
    // Small worlds: `forest_k` independent copies of the model as ONE model (copy k owns bodies k nb.., dofs k n.., position
    // scalars k nq.., constraints k nc..), so that a batch of states [nw][nq] of this model IS a batch [nw / k][k nq] of
    // the forest: k worlds share a wavefront's lanes.  Built by arb_model_create for models of at most 16 dofs.
    arb_model *forest = nullptr;
    int forest_k = 1;
#if ARB_WITH_WIDE
    // Worlds past one wavefront (more than 64 dofs / bodies, 16 constraints or 128 augmented columns; up to ARB_WIDE_MAX): the
    // workgroup-per-world kernel of arb_wide_kernel.h, float64 arithmetic, its own device-resident model
    bool is_wide = false;
    WideModel wide;
    WideModel *wide_dev = nullptr;
    size_t wide_lds = 0;
    WideModel wide_c;                  // the compact layout of the same world (system in registers), where the model qualifies
    WideModel *wide_c_dev = nullptr;
    size_t wide_c_lds = 0;
#endif
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
                          int nbp = 0, int ndof = 0) {
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
    L.total = o;
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

// How many copies of a small model share a wavefront (1: none).  Measured on an MI355X (tools/forest_probe.py,
// tools/forest_rate.py, M world-steps/s at 65 536 worlds x 64 steps): simplearm (3 dofs) float32 106
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
static int probe_rest_growth(arb_model *M, const arb_model_desc *d, float *growth);
#if ARB_WITH_WIDE
static int wide_create(const arb_model_desc *d, int device, arb_model **out);
#endif
static int device_cus(int device);
template <typename T>
static int inspect_t(arb_model *M, const DevModel<T> *dm, const Layout &L, const void *q, const void *dq,
                     const void *cforce, const void *ext, long nw, double dt, unsigned flags,
                     const arb_inspect_out *o, hipStream_t st, const arb_step_args *a = nullptr);

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
    const int ndol = ARB_MAXDOL * nc;
    const int ncols = n + 1 + ndol;
    if (n > WAVE || nb > WAVE || nc > WAVE || ncols > 2 * WAVE || ndol > WAVE) {      // past one world per wavefront
#if ARB_WITH_WIDE
        if (with_forest && n <= ARB_WIDE_MAX && nb <= ARB_WIDE_MAX && nc <= ARB_WIDE_MAX_CONSTRAINTS) return wide_create(d, device, out);
#endif
        return ARB_ERR_UNSUPPORTED;
    }
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
        // Round 6: the DEFAULT wherever the model qualifies (until round 5 only where they save the second column set).  Decided on
        // data, 119 808 replayed world-steps of the 4-contact headline model per path (profiles/r06_replay_stats.txt): world-steps
        // beyond 1e-5 of the float64 reference 43-57 -> 21-28 per 39 936, and the class the DEVICE's float32 system causes
        // (criteria d / e of tests/parity_tools.py) 4-11 -> 0 -- every remaining outlier is a decision the reference itself takes
        // marginally -- for 1.7 % of the throughput (same-process A/B, 4096 and 65 536 worlds).  ARB_STEP_CLASSIC_COLUMNS
        // selects the classical columns (the specialised four-contact kernels).
        M->bodycols_default = M->bodycols;
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
    M->lf = M->df.lay = make_layout(nb, d->nq, nc, ndol, M->nmax, 2, &tot, false, 0, n);
    M->lf3 = M->df.lay3 = make_layout(nb, d->nq, nc, ndol, M->nmax, 2, &tot, true, 0, n);
    M->ld = M->dd.lay = M->dd.lay3 = make_layout(nb, d->nq, nc, ndol, M->nmax, 1, &tot, false, 0, n);
    if ((size_t)tot * sizeof(double) > 160 * 1024) { arb_model_destroy(M); return ARB_ERR_UNSUPPORTED; }
    if (M->bodycols) {
        int tb;
        M->lfb = M->df.layb = make_layout(nb, d->nq, nc, ndol, M->nmax, 2, &tb, false, bc_nbp, n);
        M->lfb3 = M->df.layb3 = make_layout(nb, d->nq, nc, ndol, M->nmax, 2, &tb, true, bc_nbp, n);
        M->ldb = M->dd.layb = M->dd.layb3 = make_layout(nb, d->nq, nc, ndol, M->nmax, 1, &tb, false, bc_nbp, n);
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
#ifdef ARB_DEVELOPMENT
    {   // development builds: the knobs from the environment, once -- AFTER the forest exists (arb_hook_set_knob copies the
        // knobs into the forest's handle: before round 6 the forest kept the defaults)
        const char *names[] = {"lds_pad", "queue_chunk", "queue_tail", "queue_spin_cap", "force_waves", "gsw_waves", "ablate", "wide_compact", "wide_gs_groups"};
        for (const char *nm : names) {
            std::string e = std::string("ARB_") + nm;
            for (auto &ch : e) ch = (char)toupper((unsigned char)ch);
            if (const char *v = getenv(e.c_str())) (void)arb_hook_set_knob(M, nm, atoi(v));
        }
    }
#endif
#ifndef ARB_QUICK
    if (with_forest) {
        // Can float32 eliminate this model's impedance matrix?  Asked once, of the float32 inspect kernel, at two states of
        // rest (every angle 0 / every hinge angle 0.4 rad; FreeJoints at the identity), free motion, dt = 1 ms: the pivot growth
        // max_j Z_jj / pivot_j is a property of the tree and the masses far more than of the state (snake-64: 6e4 .. 9e5
        // whatever the pose; human36: 3 .. 81).  Above ARB_ILLCOND_GROWTH / 8 float32 launches run the mixed build by default.
        const int prc = probe_rest_growth(M, d, &M->rest_growth);
        if (prc != ARB_OK) { arb_model_destroy(M); return prc; }
        M->f32_policy = M->rest_growth > (float)ARB_ILLCOND_GROWTH ? 2 : M->rest_growth > (float)(ARB_ILLCOND_GROWTH / 8.0) ? 1 : 0;
        M->mixed_default = M->f32_policy >= 1;
    }
#endif
    *out = M;
    return ARB_OK;
}

#if ARB_WITH_WIDE
// ---------------------------------------------------------------------------
// The wide path (arb_wide_kernel.h): worlds past one wavefront.  Its own validation of the description (the masks of the
// wavefront path are 64 bits wide), its own device-resident model (arrays sized by the model), one launch function.
// ---------------------------------------------------------------------------
static int wide_create(const arb_model_desc *d, int device, arb_model **out) {
    const int nb = d->nb, n = d->ndof, nc = d->nc, ndol = ARB_MAXDOL * nc;
    std::vector<int> jnd(nb), depth(nb), dof2q(n, -1), dofbody(n, 0), subsize(nb, 1);
    int maxdepth = 0, ndof_chk = 0, nq_chk = 0;
    for (int b = 0; b < nb; ++b) {
        const int p = d->parent[b], jt = d->jtype[b];
        if (p >= b || p < -1 || jt < 0 || jt > ARB_JT_TXTYTZ) return ARB_ERR_INVALID;
        jnd[b] = joint_ndof(jt);
        if (d->dof_off[b] != ndof_chk || d->q_off[b] != nq_chk) return ARB_ERR_INVALID;
        for (int i = 0; i < jnd[b]; ++i) {
            if (ndof_chk + i >= n) return ARB_ERR_INVALID;
            dofbody[ndof_chk + i] = b;
            if (jt != ARB_JT_FREE) dof2q[ndof_chk + i] = nq_chk + i;
        }
        ndof_chk += jnd[b]; nq_chk += joint_nq(jt);
        depth[b] = (p < 0) ? 0 : depth[p] + 1;
        maxdepth = std::max(maxdepth, depth[b]);
    }
    if (ndof_chk != n || nq_chk != d->nq) return ARB_ERR_INVALID;
    for (int b = nb - 1; b > 0; --b) if (d->parent[b] >= 0) subsize[d->parent[b]] += subsize[b];
    for (int b = 0; b < nb; ++b)                              // bodies in DFS preorder: every subtree a contiguous range
        for (int c2 = b + 1; c2 < b + subsize[b]; ++c2) {
            int a = c2;
            while (a > b) a = d->parent[a];
            if (a != b) return ARB_ERR_UNSUPPORTED;
        }
    for (int b = 0; b < nb; ++b) {                            // rigid-body mass matrices (see model_create)
        const double *Mb = d->mass + 36 * b;
        double sc = 0.;
        for (int i = 0; i < 36; ++i) sc = std::max(sc, std::fabs(Mb[i]));
        const double tol = 1e-9 * std::max(sc, 1e-300);
        bool ok = true;
        for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) ok = ok && std::fabs(Mb[6 * i + j] - Mb[6 * j + i]) <= tol;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
            ok = ok && std::fabs(Mb[6 * (3 + i) + 3 + j] - (i == j ? Mb[21] : 0.)) <= tol;
            ok = ok && std::fabs(Mb[6 * i + 3 + j] + Mb[6 * j + 3 + i]) <= tol;
        }
        if (!ok) return ARB_ERR_UNSUPPORTED;
    }
    bool has_warm = false;
    for (int c = 0; c < nc; ++c) {
        const int ct = d->ctype[c];
        if (ct == ARB_CT_SOFTFINGER) {
            if (d->c_body[c] >= nb || d->c_body0[c] >= nb) return ARB_ERR_INVALID;
            if (d->c_geom[c] < ARB_CG_PLANE_SPHERE || d->c_geom[c] > ARB_CG_BOX_SPHERE) return ARB_ERR_INVALID;
        } else if (ct == ARB_CT_BALLSOCKET) {
            if (d->c_body[c] >= nb || d->c_body0[c] >= nb) return ARB_ERR_INVALID;
            has_warm = true;
        } else if (ct == ARB_CT_JOINTLIMITS) {
            if (d->c_dof[c] < 0 || d->c_dof[c] >= n || dof2q[d->c_dof[c]] < 0) return ARB_ERR_INVALID;
        } else return ARB_ERR_INVALID;
    }
    arb_model *M = new (std::nothrow) arb_model();
    if (!M) return ARB_ERR_NOMEM;
    M->device = device; M->is_wide = true;
    // (at most 64 of a world's constraints ACTIVE in one step: the solve works on slots, see arb_wide_kernel.h)
    const int ncap = std::min(nc, 64), nds = 4 * ncap;
    M->nb = nb; M->n = n; M->nq = d->nq; M->nc = nc; M->ndol = ndol; M->ncols = n + 1 + nds; M->nsets = 1; M->nmax = n;
    DeviceGuard guard_(device);
    if (guard_.err != hipSuccess) { g_hip_err = std::string("hipSetDevice: ") + hipGetErrorString(guard_.err); delete M; return ARB_ERR_HIP; }
    WideModel &W = M->wide;
    memset(&W, 0, sizeof(W));
    W.nb = nb; W.n = n; W.nq = d->nq; W.nc = nc; W.ndol = ndol; W.ncap = ncap; W.nds = nds; W.ncols = n + 1 + nds; W.ld = (W.ncols + 1) | 1;   // (odd row stride: LDS banks)
    W.maxdepth = maxdepth; W.has_warm = has_warm; W.has_pd = d->pd_kp != nullptr;
    for (int i = 0; i < 3; ++i) { W.grav[i] = d->gravity[i]; W.up[i] = d->up[i]; if (d->gravity[i] != 0.) W.has_grav = 1; }
    for (int i = 0; i < 36 * nb; ++i) if (d->visc[i] != 0.) W.has_visc = 1;
    int rc = ARB_OK;
    auto upi = [&](const int *src, size_t cnt, const int **dst) {
        if (rc != ARB_OK) return;
        std::vector<int> v(src ? src : nullptr, src ? src + cnt : nullptr);
        if (!src) v.assign(cnt, 0);
        rc = upload<int>(M, v, dst);
    };
    auto upd = [&](const std::vector<double> &v, const double **dst) { if (rc == ARB_OK) rc = upload<double>(M, v, dst); };
    auto vec = [](const double *src, size_t cnt) { return src ? std::vector<double>(src, src + cnt) : std::vector<double>(cnt, 0.); };
    upi(d->parent, nb, &W.parent); upi(d->jtype, nb, &W.jtype); upi(d->dof_off, nb, &W.dof_off); upi(jnd.data(), nb, &W.jnd);
    upi(d->q_off, nb, &W.q_off); upi(depth.data(), nb, &W.depth); upi(d->weighted, nb, &W.weighted); upi(dof2q.data(), n, &W.dof2q);
    upi(dofbody.data(), n, &W.dofbody); upi(subsize.data(), nb, &W.subsize);
    // (the wide kernels' own threshold: a level of the serial loop costs a workgroup ~2.8 k cycles, a round of the four jumping
    //  passes ~2 k -- human36's nine levels: 25 k against 10 k)
    if (maxdepth >= 4 && maxdepth < 256 && nb <= 256) {       // the ancestor 2^r levels up, for the log-depth chains (lane = body, <= 8 rounds)
        int rounds = 0;
        while ((1 << rounds) < maxdepth + 1) ++rounds;
        std::vector<int> janc((size_t)rounds * nb, -1);
        for (int b = 0; b < nb; ++b) janc[b] = d->parent[b];
        for (int r = 1; r < rounds; ++r)
            for (int b = 0; b < nb; ++b) {
                const int a = janc[(size_t)(r - 1) * nb + b];
                janc[(size_t)r * nb + b] = a >= 0 ? janc[(size_t)(r - 1) * nb + a] : -1;
            }
        W.jrounds = rounds;
        upi(janc.data(), janc.size(), &W.janc);
    }
    upd(h12(d->H_pr, nb), &W.Hpr); upd(h12(d->H_cn, nb), &W.Hcn);
    upd(vec(d->mass, 36 * (size_t)nb), &W.mass); upd(vec(d->visc, 36 * (size_t)nb), &W.visc);
    upd(vec(d->pd_kp, W.has_pd ? (size_t)n * n : 0), &W.pd_kp); upd(vec(d->pd_kd, W.has_pd ? (size_t)n * n : 0), &W.pd_kd);
    upd(vec(d->pd_tau0, W.has_pd ? (size_t)n : 0), &W.pd_tau0);
    upi(d->ctype, nc, &W.ctype); upi(d->c_enabled, nc, &W.cen); upi(d->c_body, nc, &W.cbody); upi(d->c_body0, nc, &W.cbody0);
    upi(d->c_dof, nc, &W.cdof); upi(d->c_geom, nc, &W.cgeom);
    std::vector<double> rz(9 * (size_t)std::max(nc, 1), 0.0);
    for (int c = 0; c < nc; ++c)
        if (d->ctype[c] == ARB_CT_SOFTFINGER && d->c_geom[c] == ARB_CG_PLANE_SPHERE) zaligned_host(d->c_plane + 4 * c, rz.data() + 9 * c);
    upd(vec(d->c_local, 3 * (size_t)nc), &W.clocal); upd(vec(d->c_radius, nc), &W.cradius); upd(vec(d->c_radius0, nc), &W.cradius0);
    upd(vec(d->c_half, 3 * (size_t)nc), &W.chalf); upd(vec(d->c_plane, 4 * (size_t)nc), &W.cplane); upd(rz, &W.cRz);
    upd(nc ? h12(d->c_bpose0, nc) : std::vector<double>(), &W.cb0); upd(nc ? h12(d->c_bpose1, nc) : std::vector<double>(), &W.cb1);
    upd(vec(d->c_mu, nc), &W.cmu); upd(vec(d->c_eps, 3 * (size_t)nc), &W.ceps); upd(vec(d->c_prox, nc), &W.cprox);
    upd(vec(d->c_min, nc), &W.cmin); upd(vec(d->c_max, nc), &W.cmax);
    if (rc != ARB_OK) { arb_model_destroy(M); return rc; }
    {
        void *hp = nullptr, *dp = nullptr;
        hipError_t e = hipHostMalloc(&hp, 2 * sizeof(int), hipHostMallocMapped);
        if (e == hipSuccess) { M->status_host = static_cast<int *>(hp); M->status_host[0] = M->status_host[1] = 0; e = hipHostGetDevicePointer(&dp, hp, 0); }
        if (e != hipSuccess) { g_hip_err = std::string("status word: ") + hipGetErrorString(e); arb_model_destroy(M); return ARB_ERR_HIP; }
        W.status = static_cast<int *>(dp); W.warn = static_cast<int *>(dp) + 1;
    }
    // Two layouts of one world's data, the same tables behind both.  The COMPACT build (arb_wide_kernel.h; at most 192 dofs and
    // 256 columns): the system in registers, LDS for the rest -- pivot hand-over buffers, the admittance of the sweeps, then one
    // region for chain arrays (24 nb doubles live to the end of phase B + 60 nb dead by then, under the 84 nb of phase B's
    // composites) and per-dof vectors, which the solution columns take over.  Otherwise: the system in LDS when it fits beside
    // the pivot row / column (120 KB: one workgroup per CU), else in scratch.  ("wide_compact" 0, arb_hook_set_knob, selects
    // the second where the first is the default: the tests hold the two bit-identical.)
    const size_t small = (size_t)(((W.ncols + 3) & ~3) + ((n + 3) & ~3) + 8 + 48 + 2 * ((nds + 3) & ~3) + 59 * ncap + (ncap & 1) + 2 * ((ncap + 5) >> 2)) * sizeof(double);
    W.sld = (1 + nds) | 1;
    // (compact: 1 = everything below in LDS, 2 = without the rows of J', 3 = without the admittance of the sweeps as well --
    //  what many constraints ask for: 124 rows of J' for 128 dofs are 127 KB --, 4 .. 6 = the same with the chain arrays and
    //  phase B's composites in scratch: what many BODIES ask for, 108 doubles each)
    auto layout = [&](WideModel &L, int compact, size_t *lds_out) {
        L.kmax = 0; L.cp = 2; L.ac_in_lds = L.am_in_lds = L.jr_in_lds = L.vec_in_lds = L.sol_in_lds = L.am_cap = 0; L.l_am = L.l_ac = L.l_xk = L.l_jr = L.l_sol = L.l_reg = L.l_vec = 0;
        // the scratch block: state and small vectors first, then per-body wrenches and joint columns (the compact build may keep
        // these two groups in LDS), then everything else
        long o = 0;
        auto take = [&](long cnt) { const long at = o; o += (cnt + 1) & ~1l; return at; };
        L.o_q = take(d->nq); L.o_dq = take(n); L.o_qd = take(n); L.o_ff = take(std::max(ndol, 1)); L.o_ff0 = take(std::max(ndol, 1));
        L.o_rh = take(2l * n); L.o_vv = take(std::max(nds, 1)); L.o_cd = take((long)WIDE_CD * std::max(nc, 1));
        const long tier1 = o;
        L.o_pt = take(12l * nb); L.o_sc = take(12l * n);
        const long tier2 = o;
        if (compact) {
            const int kmax = n <= 80 ? 20 : n <= 112 ? 28 : n <= 128 ? 32 : n <= 160 ? 40 : 48;     // rows per wavefront (four wavefronts)
            L.cp = L.ncols <= 128 ? 2 : L.ncols <= 256 ? 4 : 6;            // columns per lane
            // (the rows of J' under the composites, dead by the time they are written, when they fit; the solution columns
            //  behind them -- the per-dof vectors, read while J' is written, lie past 108 nb)
            const long jrsz = ((long)nds * n + 1) & ~1l;
            const bool capoff = compact > 6;                     // (7 .. 12: levels 1 .. 6 without the 32 KB kept for a packed admittance)
            const int lvl = capoff ? compact - 6 : compact;
            const bool bodies = lvl <= 3;                        // (chain arrays and composites in LDS)
            const int sub = bodies ? lvl : lvl - 3;
            const bool under = bodies && ndol > 0 && jrsz <= 84l * nb;
            L.jr_in_lds = (ndol > 0 && sub == 1) ? 1 : 0;
            L.am_in_lds = sub <= 2 ? 1 : 0;
            L.l_sol = (L.jr_in_lds && under) ? 24l * nb + jrsz : 0;
            L.sol_in_lds = (size_t)n * L.sld * sizeof(double) <= 64 * 1024 ? 1 : 0;      // (many constraints: 83 dofs x 249 columns are 165 KB)
            long region = (std::max((bodies ? 108l * nb : 0l) + (long)WIDE_XK * n, L.l_sol + (L.sol_in_lds ? (long)n * L.sld : 0l)) + 1) & ~1l;   // (16-byte steps)
            L.l_jr = under ? 24l * nb : region;                  // (... or a place of their own)
            if (L.jr_in_lds && !under) region += jrsz;
            const long head = 2l * 64 * L.cp;                               // (the pivot rows, double-buffered)
            L.kmax = kmax; L.l_am = head; L.l_ac = 24l * nb; L.l_xk = bodies ? 108l * nb : 0;
            // (the admittance of the sweeps: all of it, or room for what <= 16 active constraints need -- the kernel packs the step's)
            L.am_cap = L.am_in_lds ? nds * nds : ((nds > 0 && !capoff) ? std::min(nds * nds, 64 * 64) : 0);
            L.l_reg = head + ((L.am_cap + 1) & ~1);
            L.ac_in_lds = bodies ? 1 : 0; L.chain_in_lds = bodies ? 1 : 0; L.z_in_lds = 0;
            L.l_vec = region;
            size_t lds = small + (size_t)(L.l_reg + region) * sizeof(double);
            // (the vectors in LDS only where they cost no workgroup per CU: 160 KB / the request, at most four)
            const long per_cu = std::max(1l, std::min(4l, (long)(160 * 1024 / lds)));
            const size_t cap = std::min<size_t>(150 * 1024, 160 * 1024 / per_cu);
            if (lds + tier2 * sizeof(double) <= cap) { L.vec_in_lds = 2; lds += tier2 * sizeof(double); }
            else if (lds + tier1 * sizeof(double) <= cap) { L.vec_in_lds = 1; lds += tier1 * sizeof(double); }
            *lds_out = lds;
        } else {
            L.z_in_lds = (size_t)n * L.ld * sizeof(double) + small <= 120 * 1024 ? 1 : 0;
            *lds_out = small + (L.z_in_lds ? (size_t)n * L.ld * sizeof(double) : 0);
            L.chain_in_lds = (L.z_in_lds && 84l * nb <= (long)n * L.ld) ? 1 : 0;    // (the pose / twist chain borrows the system's LDS space)
            L.ac_in_lds = (L.chain_in_lds && 108l * nb <= (long)n * L.ld) ? 1 : 0;  // (phase B's composites behind the live chain arrays)
            L.l_ac = 24l * nb;
        }
        // (the chain arrays, contiguous from o_pose on: pose, twist and pseudo twist -- read until the per-dof vectors are formed -- first)
        L.o_pose = take(12l * nb); L.o_tw = take(6l * nb); L.o_om = take(6l * nb); L.o_pc = take(12l * nb); L.o_rcp = take(12l * nb);
        L.o_ab = take(6l * nb); L.o_da = take(18l * nb); L.o_tn = take(6l * nb); L.o_bn = take(6l * nb);
        L.o_ac = take(36l * nb); L.o_mc = take(36l * nb); L.o_wc = take(12l * nb); L.o_xk = take((long)WIDE_XK * n);
        L.o_z = take(L.z_in_lds ? 0 : (long)n * L.ld); L.o_jr = take((long)std::max(nds, 1) * n);
        L.o_am = take((long)std::max(nds * nds, 1)); L.o_sol = take((L.kmax && !L.sol_in_lds) ? (long)n * L.sld : 0);
        L.total = o;
    };
    auto to_device = [&](const WideModel &L, WideModel **dst) {
        void *pw = nullptr;
        hipError_t e = hipMalloc(&pw, sizeof(WideModel));
        if (e == hipSuccess) { M->allocs.push_back(pw); e = hipMemcpy(pw, &L, sizeof(WideModel), hipMemcpyHostToDevice); }
        if (e != hipSuccess) { g_hip_err = std::string("model upload: ") + hipGetErrorString(e); return ARB_ERR_HIP; }
        *dst = static_cast<WideModel *>(pw);
        return ARB_OK;
    };
    layout(W, 0, &M->wide_lds);
    rc = to_device(W, &M->wide_dev);
    if (rc == ARB_OK && ((n <= 192 && W.ncols <= 256) || (n <= 128 && W.ncols <= 384))) {      // (rows per wavefront x columns per lane: registers)
        M->wide_c = W;
        for (int level = 1; level <= 12; ++level) {
            layout(M->wide_c, level, &M->wide_c_lds);
            if (M->wide_c_lds <= 150 * 1024) { rc = to_device(M->wide_c, &M->wide_c_dev); break; }
        }
    }
    if (rc != ARB_OK) { arb_model_destroy(M); return rc; }
    *out = M;
    return ARB_OK;
}

// One launch of the wide kernel: grid = min(worlds, two workgroups per CU) workgroups that loop over the worlds, each with its
// own block of stream-ordered scratch.
template <typename T>
static int wide_launch(arb_model *M, const WideIO<T> &io_in, long nw, double dt, const double *dts, int nsteps, unsigned flags, hipStream_t st) {
    const bool compact = M->wide_c_dev != nullptr && M->kn.wide_compact != 0;
    const WideModel &L = compact ? M->wide_c : M->wide;
    WideIO<T> io = io_in;
    io.gs_serial = M->kn.wide_gs_groups == 0;
    const size_t lds = compact ? M->wide_c_lds : M->wide_lds;
    const long per_cu = std::max(1l, std::min(4l, (long)(160 * 1024 / std::max<size_t>(lds, 1))));
    // (at most 2 GB of scratch per launch: a 1024-dof world's block is 8.5 MB)
    const long by_scratch = std::max(1l, (long)((2048l << 20) / ((size_t)L.total * sizeof(double))));
    const unsigned grid = (unsigned)std::min<long>(std::min<long>(nw, by_scratch), per_cu * std::max(1, device_cus(M->device)));
    void *ws = nullptr;
    HIP_TRY(arb_scratch_alloc(&ws, (size_t)grid * (size_t)L.total * sizeof(double), st));
    const WideModel *dev = compact ? M->wide_c_dev : M->wide_dev;
#define ARB_WIDE_GO(K, P) wide_launch_one<T, K, P>(dev, io, nw, dt, dts, nsteps, flags, (double *)ws, grid, lds, st)
    const hipError_t le = L.kmax == 0 ? ARB_WIDE_GO(0, 2)
                        : L.cp == 2 ? (L.kmax == 20 ? ARB_WIDE_GO(20, 2) : L.kmax == 28 ? ARB_WIDE_GO(28, 2) : ARB_WIDE_GO(32, 2))
                        : L.cp == 6 ? (L.kmax == 20 ? ARB_WIDE_GO(20, 6) : L.kmax == 28 ? ARB_WIDE_GO(28, 6) : ARB_WIDE_GO(32, 6))
                                    : (L.kmax == 20 ? ARB_WIDE_GO(20, 4) : L.kmax == 28 ? ARB_WIDE_GO(28, 4) : L.kmax == 32 ? ARB_WIDE_GO(32, 4)
                                       : L.kmax == 40 ? ARB_WIDE_GO(40, 4) : ARB_WIDE_GO(48, 4));
#undef ARB_WIDE_GO
    (void)hipFreeAsync(ws, st);
    if (le != hipSuccess) { g_hip_err = std::string("kernel launch: ") + hipGetErrorString(le); return ARB_ERR_HIP; }
    return ARB_OK;
}

template <typename T>
static int wide_step(arb_model *M, void *q, void *dq, void *cf, const void *ext, const void *zimp, long nw, double dt, const double *dts,
                     int nsteps, unsigned flags, const arb_rollout_log *log, hipStream_t st, long ext_stride,
                     const void *pd_qdes, const void *pd_dqdes, const void *pd_kp, const void *pd_kd, long pd_stride, const arb_step_cost *cost) {
    WideIO<T> io;
    memset(&io, 0, sizeof(io));
    io.q = (T *)q; io.dq = (T *)dq; io.cf = (T *)cf; io.ext = (const T *)ext; io.zimp = (const T *)zimp; io.ext_stride = ext_stride;
    if (log) { io.log_q = (T *)log->q_log; io.log_dq = (T *)log->dq_log; io.log_energy = (T *)log->energy_log; }
    io.pd_qdes = (const T *)pd_qdes; io.pd_dqdes = (const T *)pd_dqdes; io.pd_kp = (const T *)pd_kp; io.pd_kd = (const T *)pd_kd; io.pd_stride = pd_stride;
    if (cost) { io.cost_out = (T *)cost->cost_out; io.cost_wq = (const T *)cost->w_q; io.cost_wdq = (const T *)cost->w_dq;
                io.cost_wtau = (const T *)cost->w_tau; io.cost_qref = (const T *)cost->q_ref; }
    return wide_launch<T>(M, io, nw, dt, dts, nsteps, flags, st);
}

template <typename T>
static int wide_inspect(arb_model *M, const void *q, const void *dq, const void *cf, const void *ext, const void *zimp, long nw, double dt,
                        unsigned flags, const arb_inspect_out *o, hipStream_t st, const arb_step_args *a) {
    // what the wide kernel does not form: the per-solve diagnostics of the wavefront kernels, the energies
    if (o->gs_stats || o->gs_trace || o->energy || o->pivot_growth) return ARB_ERR_UNSUPPORTED;
    WideIO<T> io;
    memset(&io, 0, sizeof(io));
    io.q = (T *)q; io.dq = (T *)dq; io.cf = (T *)cf; io.ext = (const T *)ext; io.zimp = (const T *)zimp;
    io.inspect = 1;
    if (a != nullptr) { io.pd_qdes = (const T *)a->pd_qdes; io.pd_dqdes = (const T *)a->pd_dqdes; io.pd_kp = (const T *)a->pd_kp; io.pd_kd = (const T *)a->pd_kd; }
    // the three world matrices: one pass each, like the wavefront kernels (core.py:722-734)
    struct { void *ptr; int zmode; } passes[3] = {{o->M, 1}, {o->B, 2}, {o->N, 3}};
    for (auto &ps : passes) {
        if (!ps.ptr) continue;
        WideIO<T> i1 = io;
        i1.zmode = ps.zmode; i1.Zout = (T *)ps.ptr;
        const int rc = wide_launch<T>(M, i1, nw, dt, nullptr, 1, flags | ARB_STEP_SKIP_CONSTRAINTS, st);
        if (rc != ARB_OK) return rc;
    }
    io.jac = (T *)o->jac; io.djac = (T *)o->djac; io.stamps = (long long *)o->stamps;
    io.pose = (T *)o->pose; io.twist = (T *)o->twist; io.Zout = (T *)o->Z; io.gforce0 = (T *)o->gforce0; io.vel_free = (T *)o->vel_free;
    io.c_sdist = (T *)o->c_sdist; io.c_active = (int *)o->c_active; io.c_jac = (T *)o->c_jac; io.c_force = (T *)o->c_force;
    io.c_frame = (T *)o->c_frame; io.gforce = (T *)o->gforce; io.q_next = (T *)o->q_next; io.dq_next = (T *)o->dq_next;
    io.c_adm = (T *)o->c_adm; io.c_vel = (T *)o->c_vel;
    return wide_launch<T>(M, io, nw, dt, nullptr, 1, flags, st);
}
#endif  // ARB_WITH_WIDE

// Pivot growth of the float32 elimination at the model's states of rest (see model_create): two worlds through the
// float32 inspect kernel, free motion.  Synchronous (part of arb_model_create).
static int probe_rest_growth(arb_model *M, const arb_model_desc *d, float *growth) {
    const int nq = d->nq, n = d->ndof, nb = d->nb;
    std::vector<float> hq(2 * (size_t)nq, 0.f), hdq(2 * (size_t)n, 0.f);
    for (int w = 0; w < 2; ++w)
        for (int b = 0; b < nb; ++b) {
            float *qb = hq.data() + (size_t)w * nq + d->q_off[b];
            const int jt = d->jtype[b];
            if (jt == ARB_JT_FREE) { qb[0] = qb[5] = qb[10] = qb[15] = 1.f; }
            else if (jt != ARB_JT_TXTYTZ && w == 1) for (int i = 0; i < joint_ndof(jt); ++i) qb[i] = 0.4f;
        }
    float *dq_ = nullptr, *dv_ = nullptr, *dg_ = nullptr;
    float hg[2] = {0.f, 0.f};
    auto cleanup = [&]() { if (dq_) (void)hipFree(dq_); if (dv_) (void)hipFree(dv_); if (dg_) (void)hipFree(dg_); };
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&dq_), hq.size() * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&dv_), hdq.size() * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&dg_), 2 * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(dq_, hq.data(), hq.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(dv_, hdq.data(), hdq.size() * sizeof(float), hipMemcpyHostToDevice);
    if (e != hipSuccess) { g_hip_err = std::string("rest-state probe: ") + hipGetErrorString(e); cleanup(); return ARB_ERR_HIP; }
    arb_inspect_out o;
    memset(&o, 0, sizeof(o));
    o.pivot_growth = dg_;
    const int rc = inspect_t<float>(M, M->df_dev, M->lf, dq_, dv_, nullptr, nullptr, 2l, 1e-3, ARB_STEP_SKIP_CONSTRAINTS, &o, nullptr);
    if (rc == ARB_OK) {
        e = hipDeviceSynchronize();
        if (e == hipSuccess) e = hipMemcpy(hg, dg_, 2 * sizeof(float), hipMemcpyDeviceToHost);
        if (e != hipSuccess) { g_hip_err = std::string("rest-state probe: ") + hipGetErrorString(e); cleanup(); return ARB_ERR_HIP; }
    }
    cleanup();
    if (rc != ARB_OK) return rc;
    // (a NaN growth -- a non-finite pivot -- counts as unbounded)
    *growth = (hg[0] == hg[0] && hg[1] == hg[1]) ? std::max(hg[0], hg[1]) : INFINITY;
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
        {"queue_spin_cap", &Knobs::queue_spin_cap}, {"force_waves", &Knobs::force_waves}, {"gsw_waves", &Knobs::gsw_waves},
        {"ablate", &Knobs::ablate}, {"wide_compact", &Knobs::wide_compact}, {"wide_gs_groups", &Knobs::wide_gs_groups}};
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
#if ARB_WITH_WIDE
    if (M->is_wide) {
        memset(info, 0, sizeof(*info));
        info->nb = M->nb; info->ndof = M->n; info->nq = M->nq; info->nc = M->nc;
        info->nmax = M->n; info->ncols = M->ncols; info->nsets = 1;
        info->lds_bytes_f32 = info->lds_bytes_f64 = (int32_t)((M->wide_c_dev && M->kn.wide_compact) ? M->wide_c_lds : M->wide_lds);
        info->device = M->device; info->forest_copies = 1; info->wide = 1;
        return ARB_OK;
    }
#endif
    info->nb = M->nb; info->ndof = M->n; info->nq = M->nq; info->nc = M->nc;
    info->nmax = M->nmax; info->ncols = M->ncols; info->nsets = M->nsets;
    info->lds_bytes_f32 = M->lf.total * (int)sizeof(float);
    info->lds_bytes_f64 = M->ld.total * (int)sizeof(double);
    info->device = M->device;
    info->forest_copies = M->forest ? M->forest_k : 1;
    info->mixed_default = M->f32_policy;
    info->wide = 0;
    info->rest_pivot_growth = M->rest_growth;
    return ARB_OK;
}

// Which build of the float32 production kernel runs a launch (models with one column set and a tile of up to 48 rows;
// every other model has the two-wave build only)?  The builds are bit-identical (-ffp-contract=on): a pure performance
// decision, also reported by arb_step_plan.
struct BuildChoice { bool w3 = false; long slots2 = 0, slots3 = 0; };
static BuildChoice choose_build(const arb_model *M, long nw, int nsteps, unsigned flags, bool bodyc = false) {
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
    return bc;
}

template <typename T, int MODE>
static int launch(arb_model *M, const DevModel<T> *dm, const Layout &L, T *q, T *dq, T *cf, const T *ext, const PerWorldPD<T> &pwd, long nw,
                  double dt, int nsteps, unsigned flags, const DebugOut<T> &dbg, int zmode, const LogOut<T> &logo,
                  const SplitIO<T> &sio, const double *dts, hipStream_t st, long ext_stride = 0, long pd_stride = 0,
                  const CostIO<T> &cost = CostIO<T>{nullptr, nullptr, nullptr, nullptr, nullptr}) {
    // the plain step (FEAT 0): nothing but the state and the constraint forces; FEAT 1: + user torques (MPC rollouts)
    const bool noopt = MODE == 0 && pwd.qdes == nullptr && pwd.kp == nullptr && pwd.zimp == nullptr && logo.q == nullptr &&
                       logo.dq == nullptr && logo.energy == nullptr && sio.mode == 0 && !(flags & ARB_STEP_SKIP_CONSTRAINTS) && dts == nullptr;
    // (the kernels only look at ARB_STEP_SKIP_CONSTRAINTS; the other flags are for the host)
    const bool mfma = MODE == 0 && std::is_same<T, float>::value && (flags & ARB_STEP_MFMA_ELIM) &&
                      !(M->n == WAVE && M->nc == 0);          // (the late-rhs case is handled by the vector-ALU elimination)
    // body-space constraint columns (FEAT bit 16): every launch of a model of that class -- step, rollout, inspect -- except the split
    // execution, the matrix-core elimination and ARB_STEP_GENERAL_KERNELS, which run the general kernels on two column sets
    // (equal to rounding, not bit for bit: Y' is formed as T (J_p Y J_p^T) T^T instead of J' Y J'^T)
    // the mixed build (CM 3, ARB_STEP_MIXED): float32 buffers, float64 elimination -- the model's default when float32 cannot
    // eliminate its impedance matrix (arb_model::mixed_default), or on request; general kernels, two waves
#ifdef ARB_QUICK
    const bool mixed = false;
#else
    const bool mixed = MODE == 0 && std::is_same<T, float>::value && !mfma && sio.mode == 0 && !(flags & ARB_STEP_NO_MIXED) &&
                       ((flags & ARB_STEP_MIXED) || M->mixed_default);
#endif
    const bool bodyc = ARB_WITH_SPEC && M->bodycols && (M->bodycols_default || (flags & ARB_STEP_BODY_COLUMNS)) && !mfma && sio.mode == 0 &&
                       !(flags & (ARB_STEP_GENERAL_KERNELS | ARB_STEP_SKIP_CONSTRAINTS | ARB_STEP_CLASSIC_COLUMNS)) && !mixed;
    const BuildChoice bc = (MODE == 0 && std::is_same<T, float>::value && !mfma && !mixed) ? choose_build(M, nw, nsteps, flags, bodyc) : BuildChoice();
    const bool w3 = bc.w3;
    // the kernels specialised for the model class "four plane / sphere SoftFingerContacts" (FEAT bit 4): bit-identical to the
    // general ones, which ARB_STEP_GENERAL_KERNELS selects
    // (float32 with one or two column sets, float64 with one)
    const bool spec = ARB_WITH_SPEC && M->spec_ok && noopt && MODE == 0 && !mfma && !mixed &&
                      (std::is_same<T, float>::value || M->nsets == 1) && !(flags & ARB_STEP_GENERAL_KERNELS);
    const bool spec0 = ARB_WITH_SPEC && M->spec0_ok && noopt && MODE == 0 && std::is_same<T, float>::value && !mfma && !mixed &&
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
                    if (M->nc == 4 && noopt) {      /* (compiled for four contacts: FEAT bit 32) */                    \
                        if (w3) return plain ? ONE_(NM, 1, 52, 2) : ONE_(NM, 1, 53, 2);                                \
                        return plain ? ONE(NM, 1, 52) : ONE(NM, 1, 53);                                                \
                    }                                                                                                  \
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
// (the instantiation must fit the model: a kernel with the wrong tile, column sets or model class computes on, silently wrong --
// round 4's first launch table sent an 8-contact model to the one-set specialised kernel; checked at every launch since)
#define ONE_(NM, NS, FT, CMV) (!(M->nmax == (NM) && (!((FT) & 32) || M->nc == 4) && (((FT) & 16) ? (M->bodycols && (NS) == 1) : (M->nsets == (NS) && (!((FT) & 4) || (M->spec_ok && M->nc == 4 * (NS))))) && (!((FT) & 8) || (M->spec0_ok && M->nc == 0))) ? (g_hip_err = "internal: kernel instantiation does not fit the model", (int)ARB_ERR_HIP) : launch_one<T, NM, NS, MODE, FT, CMV>(dm, ((FT) & 16) ? ((CMV) == 2 ? M->lfb3 : (std::is_same<T, float>::value ? M->lfb : M->ldb)) : (CMV) == 2 ? M->lf3 : L, q, dq, cf, ext, pwd, nw, dt, nsteps, flags, dbg, zmode, logo, sio, dts, st, M->kn, ext_stride, pd_stride, cost))
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
            if (mixed) {                                                                               \
                if (M->nsets == 2) return plain ? ONE_(NM, 2, 0, 3) : noopt ? ONE_(NM, 2, 1, 3) : ONE_(NM, 2, 3, 3); \
                return plain ? ONE_(NM, 1, 0, 3) : noopt ? ONE_(NM, 1, 1, 3) : ONE_(NM, 1, 3, 3);      \
            }                                                                                          \
        }                                                                                              \
        ARB_BODYC_CASE(NM)                                                                             \
        if constexpr (MODE == 0 && std::is_same<T, float>::value && NM >= 44 && NM <= 48) {            \
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
static int launch_gsw(const DevModel<T> *dm, int nc, const SplitIO<T> &sio, long nw, double dt, const double *dts, hipStream_t st, int wv) {
    auto al = [](int x) { return (x + 3) & ~3; };
    const int ndol = 4 * nc;
    const size_t lds = (size_t)(al(ndol * ndol) + al(nc * CD_STRIDE) + 2 * al(ndol) + 64) * sizeof(T);
    // wv: waves per SIMD the sweep kernel is compiled for (knob "gsw_waves": 3 = no spills, 4 = 128 VGPRs)
    if (lds > 64 * 1024) return ARB_ERR_UNSUPPORTED;
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
        if (rc == ARB_OK) rc = launch_gsw<T>(dm, nc, sio, nw, dt, dts ? dts + k : nullptr, st, M->kn.gsw_waves);
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

static int step_impl(arb_model *M, int dtype, void *q, void *dq, void *cforce, const void *ext_gforce,
                     const void *pd_qdes, const void *pd_dqdes, const void *pd_kp, const void *pd_kd,
                     int64_t nworlds, double dt, const double *dt_steps, int32_t nsteps, uint32_t flags,
                     const arb_rollout_log *log, void *stream, long ext_stride, long pd_stride,
                     const arb_step_cost *cost, const void *ext_imp);

// PROMOTION (round 6): float32 buffers of a model that only float64 carries to 1e-5 (arb_model::f32_policy 2: a long serial
// chain -- the smallest eigenvalue of snake-64's mass matrix is 3e-9 of the largest, so a float32 rounding ANYWHERE between the
// state and the generalized forces, 6e-8 relative, comes back as a velocity error of 1e-4 .. 1; the mixed build, which keeps
// twists and body wrenches in float32, measures 7e-5) are stepped by the FLOAT64 kernels on converted copies: state, constraint
// forces, user torques and impedance go up, state and forces come down after the last step -- the float32 buffers hold
// exactly what a float64 caller would have rounded.  Stream-ordered scratch, no synchronisation.
static int step_promoted(arb_model *M, float *q, float *dq, float *cf, const float *ext, const float *zimp, int64_t nw, double dt,
                         const double *dts, int nsteps, uint32_t flags, hipStream_t st, long ext_stride) {
    const size_t nq = (size_t)M->nq * nw, nd = (size_t)M->n * nw, nf = cf ? (size_t)M->ndol * nw : 0;
    const size_t ne = ext ? (ext_stride ? (size_t)ext_stride * nsteps : nd) : 0, nz = zimp ? (size_t)M->n * M->n * nw : 0;
    void *ws = nullptr;
    HIP_TRY(arb_scratch_alloc(&ws, (nq + nd + nf + ne + nz) * sizeof(double), st));
    double *pq = (double *)ws, *pdq = pq + nq, *pf = pdq + nd, *pe = pf + nf, *pz = pe + ne;
    auto up = [&](const float *src, double *dst, size_t cnt) {
        if (cnt) hipLaunchKernelGGL((arb_cvt_kernel<float, double>), dim3((unsigned)std::min<size_t>((cnt + 255) / 256, 4096)), dim3(256), 0, st, src, dst, cnt);
    };
    auto down = [&](const double *src, float *dst, size_t cnt) {
        if (cnt) hipLaunchKernelGGL((arb_cvt_kernel<double, float>), dim3((unsigned)std::min<size_t>((cnt + 255) / 256, 4096)), dim3(256), 0, st, src, dst, cnt);
    };
    up(q, pq, nq); up(dq, pdq, nd); up(cf, pf, nf); up(ext, pe, ne); up(zimp, pz, nz);
    int rc = (hipGetLastError() == hipSuccess) ? ARB_OK : ARB_ERR_HIP;
    if (rc == ARB_OK)
        rc = step_impl(M, ARB_F64, pq, pdq, nf ? pf : nullptr, ne ? pe : nullptr, nullptr, nullptr, nullptr, nullptr, nw, dt, dts, nsteps,
                       flags, nullptr, st, ext_stride, 0, nullptr, nz ? pz : nullptr);
    if (rc == ARB_OK) {
        down(pq, q, nq); down(pdq, dq, nd); down(pf, cf, nf);
        if (hipGetLastError() != hipSuccess) rc = ARB_ERR_HIP;
    }
    (void)hipFreeAsync(ws, st);
    return rc;
}

// ext_stride / pd_stride: elements between the rows of consecutive steps of a control sequence (0: one row for the launch)
static int step_impl(arb_model *M, int dtype, void *q, void *dq, void *cforce, const void *ext_gforce,
                     const void *pd_qdes, const void *pd_dqdes, const void *pd_kp, const void *pd_kd,
                     int64_t nworlds, double dt, const double *dt_steps, int32_t nsteps, uint32_t flags,
                     const arb_rollout_log *log, void *stream, long ext_stride = 0, long pd_stride = 0,
                     const arb_step_cost *cost = nullptr, const void *ext_imp = nullptr) {
    if (!M || nworlds < 0 || nsteps < 0) return ARB_ERR_INVALID;
    if (dt_steps == nullptr && !(dt > 0.0)) return ARB_ERR_INVALID;
    if (dt_steps != nullptr) dt = 1.0;                  // unused: every step reads its own dt
    if (dtype != ARB_F32 && dtype != ARB_F64) return ARB_ERR_INVALID;
    if (flags & ~ARB_STEP_KNOWN_FLAGS) return ARB_ERR_INVALID;     // (4u was ARB_STEP_SPLIT up to ABI 4)
    // per-world PD inputs: targets come in pairs; diagonal gains come in pairs and need targets;
    // targets without gains use the model's gain matrices, so the model must hold a PD controller
    if ((pd_qdes == nullptr) != (pd_dqdes == nullptr) || (pd_kp == nullptr) != (pd_kd == nullptr)) return ARB_ERR_INVALID;
    if (pd_kp != nullptr && pd_qdes == nullptr) return ARB_ERR_INVALID;
#if ARB_WITH_WIDE
    const bool model_has_pd = M->is_wide ? (M->wide.has_pd != 0) : (M->df.has_pd != 0);
#else
    const bool model_has_pd = M->df.has_pd != 0;
#endif
    if (pd_qdes != nullptr && pd_kp == nullptr && !model_has_pd) return ARB_ERR_INVALID;
    if (cost != nullptr && cost->cost_out == nullptr) return ARB_ERR_INVALID;
    if (nworlds == 0 || nsteps == 0) return ARB_OK;     // empty batch: nothing to do (pointers may be null)
    if (!q || !dq) return ARB_ERR_INVALID;
    if (nworlds > 0x7fffffffLL) return ARB_ERR_INVALID;
    if (int stalled = take_status(M, false)) return stalled;
    ARB_GUARD_DEVICE(M->device);
#if ARB_WITH_WIDE
    if (M->is_wide) {
        // (the wide kernel: float64 arithmetic for either buffer type; see arb_wide_kernel.h for what it takes)
        if ((flags & (ARB_STEP_SPLIT_WAVE | ARB_STEP_MFMA_ELIM)) && !(flags & ARB_STEP_FUSED)) return ARB_ERR_UNSUPPORTED;
        hipStream_t sw = reinterpret_cast<hipStream_t>(stream);
        return dtype == ARB_F32 ? wide_step<float>(M, q, dq, cforce, ext_gforce, ext_imp, (long)nworlds, dt, dt_steps, nsteps, flags, log, sw, ext_stride,
                                                   pd_qdes, pd_dqdes, pd_kp, pd_kd, pd_stride, cost)
                                : wide_step<double>(M, q, dq, cforce, ext_gforce, ext_imp, (long)nworlds, dt, dt_steps, nsteps, flags, log, sw, ext_stride,
                                                    pd_qdes, pd_dqdes, pd_kp, pd_kd, pd_stride, cost);
    }
#endif
    // (a dense per-world impedance couples any pair of dofs: the copies of a forest would no longer be independent blocks)
    if (use_forest(M, nworlds, flags, log, cost != nullptr || ext_imp != nullptr)) {
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
    // float32 buffers of a model float32 cannot carry (see step_promoted): the float64 kernels, unless the caller pinned a
    // build or uses inputs the promotion does not convert (per-world PD inputs, logs, costs: those run the mixed build)
    if (dtype == ARB_F32 && M->f32_policy == 2 && pd_qdes == nullptr && pd_kp == nullptr && log == nullptr && cost == nullptr &&
        !(flags & (ARB_STEP_MIXED | ARB_STEP_NO_MIXED | ARB_STEP_MFMA_ELIM | ARB_STEP_SPLIT_WAVE)))
        return step_promoted(M, (float *)q, (float *)dq, (float *)cforce, (const float *)ext_gforce, (const float *)ext_imp, nworlds, dt,
                             dt_steps, nsteps, flags, st, ext_stride);
    if (dtype == ARB_F32) {
        const PerWorldPD<float> pwd = {(const float *)pd_qdes, (const float *)pd_dqdes, (const float *)pd_kp, (const float *)pd_kd, (const float *)ext_imp};
        CostIO<float> ci = {nullptr, nullptr, nullptr, nullptr, nullptr};
        if (cost) ci = CostIO<float>{(float *)cost->cost_out, (const float *)cost->w_q, (const float *)cost->w_dq, (const float *)cost->w_tau, (const float *)cost->q_ref};
        return step_typed<float>(M, M->df_dev, M->lf, (float *)q, (float *)dq, (float *)cforce,
                                 (const float *)ext_gforce, pwd, (long)nworlds, dt, dt_steps, nsteps, flags, log, st, ext_stride, pd_stride, ci);
    }
    const PerWorldPD<double> pwd = {(const double *)pd_qdes, (const double *)pd_dqdes, (const double *)pd_kp, (const double *)pd_kd, (const double *)ext_imp};
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

// What would arb_step / arb_step_ex launch for this batch?  The same dispatch as step_typed with placeholder buffers (never
// dereferenced: launch_one returns at its probe): optional_inputs 1 = user torques, 3 = every optional input.
template <typename T>
static int plan_typed(arb_model *M, const DevModel<T> *dm, const Layout &L, long nw, int nsteps, unsigned flags, int optional_inputs,
                      LaunchProbe *pr) {
    DebugOut<T> dbg; memset(&dbg, 0, sizeof(dbg));
    LogOut<T> lo; memset(&lo, 0, sizeof(lo));
    SplitIO<T> sio; memset(&sio, 0, sizeof(sio));
    PerWorldPD<T> pwd = {nullptr, nullptr, nullptr, nullptr, nullptr};
    T *const some = reinterpret_cast<T *>(static_cast<uintptr_t>(256));
    if (optional_inputs == 3) lo.q = some;
    const bool split = M->nc > 0 && (flags & ARB_STEP_SPLIT_WAVE) && !(flags & (ARB_STEP_SKIP_CONSTRAINTS | ARB_STEP_FUSED));
    if (split) sio.mode = 2;          // (one launch per step: dynamics + system, then the sweep kernel)
    g_probe = pr;
    const int rc = launch<T, 0>(M, dm, L, some, some, M->nc > 0 ? some : nullptr, optional_inputs >= 1 ? some : nullptr, pwd, nw, 1e-3,
                                split ? 1 : nsteps, flags, dbg, 0, lo, sio, nullptr, nullptr);
    g_probe = nullptr;
    return rc;
}

extern "C" int arb_step_plan(arb_model *M, int dtype, int64_t nworlds, int32_t nsteps, uint32_t flags, int32_t optional_inputs,
                             arb_step_plan_info *out) {
    if (!M || !out || nworlds < 0 || nsteps < 0 || (dtype != ARB_F32 && dtype != ARB_F64)) return ARB_ERR_INVALID;
    if (flags & ~ARB_STEP_KNOWN_FLAGS) return ARB_ERR_INVALID;
    ARB_GUARD_DEVICE(M->device);
    if (optional_inputs < 0 || optional_inputs > 7 || (optional_inputs & 3) == 2) return ARB_ERR_INVALID;
#if ARB_WITH_WIDE
    if (M->is_wide) {          // one workgroup of four wavefronts per world, workgroups loop over the batch
        memset(out, 0, sizeof(*out));
        const size_t wlds = (M->wide_c_dev && M->kn.wide_compact) ? M->wide_c_lds : M->wide_lds;
        const long per_cu = std::max(1l, std::min(4l, (long)(160 * 1024 / std::max<size_t>(wlds, 1))));
        out->waves_per_simd = 1; out->worlds_per_wavefront = 1; out->feat = 3; out->lds_bytes = (int32_t)wlds;
        out->wave_slots = (int32_t)(per_cu * device_cus(M->device)); out->work_queue = 0;
        return ARB_OK;
    }
#endif
    const bool world_logs = (optional_inputs & 4) != 0;      // per-world energies / costs, or state logs of a ragged batch: no forest
    optional_inputs &= 3;
    // (float32 buffers of a model only float64 carries: the launch runs the float64 kernels, see step_promoted)
    if (dtype == ARB_F32 && M->f32_policy == 2 && optional_inputs <= 1 && !world_logs &&
        !(flags & (ARB_STEP_MIXED | ARB_STEP_NO_MIXED | ARB_STEP_MFMA_ELIM | ARB_STEP_SPLIT_WAVE)))
        dtype = ARB_F64;
    if (!world_logs && use_forest(M, nworlds, flags, nullptr)) {
        const int rc = arb_step_plan(M->forest, dtype, nworlds / M->forest_k, nsteps, flags | ARB_STEP_ONE_WORLD, optional_inputs, out);
        if (rc == ARB_OK) out->worlds_per_wavefront = M->forest_k;
        return rc;
    }
    memset(out, 0, sizeof(*out));
    // the launch path itself answers (launch_one with a probe installed: nothing is allocated or launched), so the plan cannot
    // drift from what a launch does -- the instantiation, its LDS, the wave slots from the kernel's own register count
    LaunchProbe pr;
    memset(&pr, 0, sizeof(pr));
    const int rc = dtype == ARB_F32 ? plan_typed<float>(M, M->df_dev, M->lf, (long)nworlds, nsteps, flags, optional_inputs, &pr)
                                    : plan_typed<double>(M, M->dd_dev, M->ld, (long)nworlds, nsteps, flags, optional_inputs, &pr);
    if (rc != ARB_OK) return rc;
    out->worlds_per_wavefront = 1;
    out->waves_per_simd = pr.waves_compiled;
    out->lds_bytes = (int32_t)(pr.lds_bytes - std::max(0, M->kn.lds_pad));
    out->wave_slots = pr.wave_slots;
    out->work_queue = pr.work_queue;
    out->feat = pr.feat;
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
                     a->ext_gforce_steps ? row : 0l, a->pd_qdes_steps ? row : 0l, a->cost, a->ext_impedance);
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
                     const arb_inspect_out *o, hipStream_t st, const arb_step_args *a) {
    PerWorldPD<T> pwd; memset(&pwd, 0, sizeof(pwd));
    if (a != nullptr) {        // arb_inspect_ex: the per-world controller inputs of arb_step_ex
        pwd.qdes = (const T *)a->pd_qdes; pwd.dqdes = (const T *)a->pd_dqdes; pwd.kp = (const T *)a->pd_kp; pwd.kd = (const T *)a->pd_kd;
        pwd.zimp = (const T *)a->ext_impedance;
    }
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

static int inspect_impl(arb_model *M, int dtype, const void *q, const void *dq, const void *cforce,
                        const void *ext_gforce, int64_t nworlds, double dt, uint32_t flags,
                        const arb_inspect_out *out, void *stream, const arb_step_args *a) {
    if (!M || !out || nworlds < 0 || !(dt > 0.0)) return ARB_ERR_INVALID;
    if (dtype != ARB_F32 && dtype != ARB_F64) return ARB_ERR_INVALID;
    if (out->gforce != nullptr && M->nc > 0 && out->c_jac == nullptr) return ARB_ERR_INVALID;
    if (nworlds == 0) return ARB_OK;
    if (!q || !dq) return ARB_ERR_INVALID;
    if (nworlds > 0x7fffffffLL) return ARB_ERR_INVALID;
    if (int stalled = take_status(M, false)) return stalled;
    ARB_GUARD_DEVICE(M->device);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
#if ARB_WITH_WIDE
    if (M->is_wide) {
        const void *zi = a ? a->ext_impedance : nullptr;
        return dtype == ARB_F32 ? wide_inspect<float>(M, q, dq, cforce, ext_gforce, zi, (long)nworlds, dt, flags, out, st, a)
                                : wide_inspect<double>(M, q, dq, cforce, ext_gforce, zi, (long)nworlds, dt, flags, out, st, a);
    }
#endif
    if (dtype == ARB_F32)
        return inspect_t<float>(M, M->df_dev, M->lf, q, dq, cforce, ext_gforce, (long)nworlds, dt, flags, out, st, a);
    return inspect_t<double>(M, M->dd_dev, M->ld, q, dq, cforce, ext_gforce, (long)nworlds, dt, flags, out, st, a);
}

extern "C" int arb_inspect(arb_model *M, int dtype, const void *q, const void *dq, const void *cforce,
                           const void *ext_gforce, int64_t nworlds, double dt, uint32_t flags,
                           const arb_inspect_out *out, void *stream) {
    return inspect_impl(M, dtype, q, dq, cforce, ext_gforce, nworlds, dt, flags, out, stream, nullptr);
}

// (ABI 8) arb_inspect with the inputs of arb_step_ex: one step of `args` evaluated without touching the state.  Control
// sequences, per-step dt, logs and costs belong to multi-step launches: refused here.
extern "C" int arb_inspect_ex(arb_model *M, int dtype, const arb_step_args *a, const arb_inspect_out *out, void *stream) {
    if (!M || !a) return ARB_ERR_INVALID;
    if (a->ext_gforce_steps || a->pd_qdes_steps || a->pd_dqdes_steps || a->cost || a->log || a->dt_steps) return ARB_ERR_INVALID;
    if ((a->pd_qdes == nullptr) != (a->pd_dqdes == nullptr) || (a->pd_kp == nullptr) != (a->pd_kd == nullptr)) return ARB_ERR_INVALID;
    if (a->pd_kp != nullptr && a->pd_qdes == nullptr) return ARB_ERR_INVALID;
#if ARB_WITH_WIDE
    const bool model_has_pd = M->is_wide ? (M->wide.has_pd != 0) : (M->df.has_pd != 0);
#else
    const bool model_has_pd = M->df.has_pd != 0;
#endif
    if (a->pd_qdes != nullptr && a->pd_kp == nullptr && !model_has_pd) return ARB_ERR_INVALID;
    return inspect_impl(M, dtype, a->q, a->dq, a->cforce, a->ext_gforce, a->nworlds, a->dt, a->flags, out, stream, a);
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

// build variants compiled into this library: bit 2 sweeps without the fast variant of the local solve (-DARB_GS_FAST=0), bit 3 no
// specialised kernels (-DARB_WITH_SPEC=0); bits 0 and 1 were the packed and rendezvous builds, measured slower and removed in round 5
extern "C" int arb_build_variants(void) { return (ARB_GS_FAST ? 0 : 4) | (ARB_WITH_SPEC ? 0 : 8); }

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

extern "C" int arb_host_growth_bits(float zjj, float pivot) {
    int zb, pb;
    memcpy(&zb, &zjj, 4); memcpy(&pb, &pivot, 4);
    return arb_growth_bits(zb, pb);
}

extern "C" int arb_host_exp_twist(const double *tw, double *H /*16*/) {
    M3<double> R; V3<double> p;
    exp_twist<double>(v3<double>(tw[0], tw[1], tw[2]), v3<double>(tw[3], tw[4], tw[5]), R, p);
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) H[4 * i + j] = R.a[3 * i + j]; }
    H[3] = p.x; H[7] = p.y; H[11] = p.z; H[12] = H[13] = H[14] = 0; H[15] = 1;
    return 0;
}

#endif  // !ARB_PART
