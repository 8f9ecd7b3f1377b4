/*
 * arbstep.h -- C ABI of libarbstep.so, the MI355X (gfx950) batched rigid-body step.
 *
 * The reference (sbarthelemy/arboris-python) is pure Python: it has no FFI, no
 * operator registry and no native boundary.  Its "plugin API" is the set of
 * World/Body/Joint/Constraint/Controller classes of arboris/core.py, and its hot
 * path is the loop body of core.simulate (arboris/core.py:1356-1363):
 *
 *     world.update_dynamic()        core.py:682-734  (+ Body.update_dynamic :1158-1315)
 *     world.update_controllers(dt)  core.py:811-818  (+ controllers.py:43-60, 141-158)
 *     world.update_constraints(dt)  core.py:910-937  (+ constraints.py, collisions.py)
 *     world.integrate(dt)           core.py:974-980  (+ joints.py:54-57, core.py:238-240)
 *
 * This header is therefore the boundary a maintainer would bind from a re-authored
 * `World` (see INTEGRATION.md for the ctypes stub): the world tree is flattened
 * once (arb_model_desc, the counterpart of World.init core.py:608-635) and the
 * four calls above become one arb_step() over a batch of independent worlds.
 *
 * Conventions
 *   - plain C, every entry point returns an int status (ARB_OK == 0) and never
 *     throws; arb_strerror() describes a status.
 *   - the caller owns all state buffers and passes DEVICE pointers (e.g. from
 *     torch.Tensor.data_ptr()); the library keeps only the immutable model.
 *   - calls are asynchronous with respect to `stream` (a hipStream_t passed as
 *     void*; NULL = the default stream); there is no hidden synchronisation.
 *   - one model handle per device; calls on different handles are re-entrant, and a handle may be
 *     used from several streams at once (all scratch memory is allocated per call, in stream order).
 *     Every call makes the handle's device current and restores the caller's device on return.
 *   - twists/wrenches are ordered [angular; linear]; matrices are row-major.
 *
 * State layout (world-major): q[nworlds][nq], dq[nworlds][ndof].  q is the
 * concatenation, in depth-first joint order, of each joint's gpos; a FreeJoint
 * stores its 4x4 pose (16 scalars, joints.py:26-31).  dq is World._gvel.
 */
#ifndef ARBSTEP_H
#define ARBSTEP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 6 (round 4): the stall status word is sticky (arb_model_status alone clears it), arb_step_plan's optional_inputs has a
 * "per-world logs" bit, forest copies are bit-identical with constraints and their per-world inputs are screened.  No struct
 * layout changed since 5. */
/* 7 (round 5): arb_step_args carries per-step control schedules (ext_gforce_steps, pd_qdes_steps / pd_dqdes_steps: the
 * reference polls its controllers EVERY step, core.py:811-817) and a per-world running cost (arb_step_cost); arb_inspect_out
 * ends with pivot_growth; arb_model_warnings / ARB_WARN_ILLCOND; ARB_STEP_GENERAL_KERNELS.  The library reads NO environment
 * variable any more (development builds, -DARB_DEVELOPMENT, still do: once, at arb_model_create). */
/* 8 (round 6): arb_step_args ends with ext_impedance -- the dense impedance Z_a of a world's user-defined Controllers
 * (core.py:327-339, 815-817), with their generalized force in ext_gforce: the generic plugin path --; arb_inspect_ex = arb_inspect
 * with the inputs of arb_step_ex; ARB_STEP_MIXED / ARB_STEP_NO_MIXED and arb_model_info.mixed_default: float32 state with the
 * impedance eliminated in float64, chosen by the library for models float32 cannot eliminate (long serial chains) instead of
 * a warning and wrong velocities; arb_model_info.wide: worlds of more than 64 dofs or bodies (up to ARB_WIDE_MAX) run on the
 * workgroup-per-world kernels (float64 arithmetic).  Earlier struct fields keep their offsets. */
#define ARB_ABI_VERSION 8

/* status codes */
enum {
    ARB_OK = 0,
    ARB_ERR_INVALID = 1,      /* bad argument (null pointer, negative size, bad enum) */
    ARB_ERR_UNSUPPORTED = 2,  /* model outside what the kernels handle (ndof > ARB_WIDE_MAX ...) */
    ARB_ERR_HIP = 3,          /* a HIP runtime call failed (see arb_last_hip_error) */
    ARB_ERR_NOMEM = 4,
    ARB_ERR_STALLED = 5       /* an EARLIER launch on this handle gave up waiting inside its device-side work queue (see
                                 arb_model_status): the states that launch wrote are not valid */
};

/* scalar type of the state buffers and of the arithmetic */
enum { ARB_F32 = 0, ARB_F64 = 1 };

/* joint types, arboris/joints.py (ids shared with arboris_python_amd/flatten.py) */
enum {
    ARB_JT_FREE = 0,     /* joints.py:10-57   */
    ARB_JT_RZRYRX = 1,   /* joints.py:59-104  */
    ARB_JT_RZRY = 2,     /* joints.py:107-146 */
    ARB_JT_RZRX = 3,     /* joints.py:149-185 */
    ARB_JT_RYRX = 4,     /* joints.py:188-224 */
    ARB_JT_RZ = 5,       /* joints.py:227-303 */
    ARB_JT_RY = 6,       /* joints.py:305-326 */
    ARB_JT_RX = 7,       /* joints.py:328-349 */
    ARB_JT_TXTYTZ = 8    /* joints.py:352-384 */
};

/* constraint types, arboris/constraints.py */
enum {
    ARB_CT_SOFTFINGER = 0,       /* SoftFingerContact :300-836; the shape pair is c_geom */
    ARB_CT_JOINTLIMITS = 1,      /* JointLimits :15-90 */
    ARB_CT_BALLSOCKET = 2        /* BallAndSocketConstraint :92-237 */
};

/* narrow-phase pair of a SoftFingerContact: (shape 0, shape 1), arboris/collisions.py:27-64.
   Shape 1 is always a Sphere or a Point (radius 0). */
enum {
    ARB_CG_PLANE_SPHERE = 0,     /* plane_sphere_collision / plane_point_collision :161-205 */
    ARB_CG_SPHERE_SPHERE = 1,    /* sphere_sphere_collision / sphere_point_collision :67-159 */
    ARB_CG_BOX_SPHERE = 2        /* box_sphere_collision :207-299 (box / point uses radius 0) */
};

#define ARB_MAXDOL 4   /* rows reserved per constraint in cforce / contact outputs */

/* arb_step flags */
#define ARB_STEP_SKIP_CONSTRAINTS 1u  /* integrate with controller forces only */
#define ARB_STEP_FUSED 2u             /* keep the Gauss-Seidel sweeps inside the step kernel (the default) */
                                      /* (4u was ARB_STEP_SPLIT, the lane-per-world sweep kernel of ABI <= 4: removed, the
                                         bit is refused with ARB_ERR_INVALID) */
#define ARB_STEP_MFMA_ELIM 16u        /* float32 only: eliminate the augmented system [Z | rhs | J'^T] on the matrix cores
                                         (v_mfma_f32_4x4x1_16b_f32 rank-1 updates) instead of the vector ALU; same results to
                                         rounding, measured SLOWER on MI355X: opt-in */
#define ARB_STEP_STATIC_WORLDS 32u     /* one workgroup per world for the whole launch.  Default for multi-step launches of more
                                         worlds than the chip holds wavefronts: the resident wavefronts draw (chunk of 4 steps,
                                         world) items from a device-side queue, which keeps the wave slots full until the last
                                         chunk (same results bit for bit; the queue is a stream-ordered allocation, so a launch
                                         has no hidden synchronisation) */
#define ARB_STEP_SPLIT_WAVE 8u        /* run the sweeps in a second kernel with one WAVEFRONT per world (the fused kernel's
                                         quad-local sweeps, compiled for more waves per SIMD): bit-identical to the
                                         GENERAL kernels (ARB_STEP_GENERAL_KERNELS).  For a model whose default is
                                         body-space constraint columns (arb_step_plan_info.feat bit 16: human36 with the
                                         reference's eight contact points) this flag, ARB_STEP_MFMA_ELIM and
                                         ARB_STEP_GENERAL_KERNELS switch back to the classical columns on two column
                                         sets: equal to the default to ROUNDING there, not bit for bit -- compare such
                                         runs with each other, not with the default.
                                         Measured slower than the default at every batch size: opt-in */
#define ARB_STEP_WAVES2 64u            /* pin the float32 step kernel build: compiled for two waves per SIMD (no register spills: */
#define ARB_STEP_WAVES3 128u           /* the faster wave) or for three (more waves in flight: the faster chip once the batch fills
                                         them).  Default: the library picks by batch size and launch shape (three waves from
                                         ~3400 worlds on an MI355X), for models that have both builds: float32, 33 .. 48 dofs (the 44- and 48-row
                                         register tiles; smaller models run faster on two waves at every batch size), ndof + 1 +
                                         4 nc <= 64.  All builds execute the same float operations in the same order per world --
                                         the library is compiled with -ffp-contract=on, so no fused multiply-add depends on how
                                         the compiler inlined a function -- and give bit-identical results (tested across
                                         builds, batch positions, launch shapes; tests/test_gpu_round3.py).  The pins exist
                                         for performance experiments and as a belt for callers that compare runs bit for bit;
                                         a pin the model has no build for is ignored. */
#define ARB_STEP_ONE_WORLD 256u        /* one world per wavefront even for a small model.  Default: the worlds of a model of at most
                                         16 dofs share wavefronts -- arb_model_create also builds a FOREST of k independent
                                         copies of the model (as many as fit a 32-row tile, one set of columns and 24 bodies:
                                         8 simplearms), and a batch of nw worlds runs as nw / k forest worlds on the same
                                         buffers (world w is copy w % k of forest world w / k; the last nw % k worlds run one per
                                         wavefront) once nw exceeds twice the wave slots of the device (4096 on an MI355X).  The copies share nothing but
                                         ground, gravity and dt: the augmented system is block diagonal and products with the
                                         exact zeros between the blocks change nothing, every tree is assembled about its own
                                         root, and the constraint-space products add a copy's dofs in the groups of four they
                                         form in the copy alone (round 4) -- results are bit-identical to one world per
                                         wavefront, with or without constraints, whatever the batch size and a world's place
                                         in the batch (tests/test_gpu_forest.py, test_gpu_random_models.py).
                                         The one difference: a copy whose state or per-world inputs (user torques, PD targets
                                         and gains) are not finite, or beyond +-1e8 (float32) / 1e100, at the beginning of a
                                         step is retired: NaN in its state, forces and logs from then on, its inputs ignored,
                                         its neighbours untouched; one world per wavefront keeps stepping such a world.  Launches
                                         that log energies (per world) or that log states for a batch that is not a multiple of
                                         k run one world per wavefront. */
#define ARB_STEP_GENERAL_KERNELS 512u   /* run the general kernels also for a model of one of the specialised classes (arb_step_plan_info.feat
                                         bits 4 / 8): bit-identical results, ~1 % slower -- for callers (and tests) that want to
                                         see the difference.  (feat bit 16, body-space columns by default: the general
                                         kernels run the classical columns -- equal to rounding only, see ARB_STEP_SPLIT_WAVE.) */
#define ARB_STEP_BODY_COLUMNS 1024u     /* constraint columns in BODY space wherever the model qualifies.  For a model whose constraints are all
                                         enabled plane / sphere SoftFingerContacts (no PD controller, no joint viscosity, one small tree) the
                                         4 nc rows of the constraint Jacobian are T_c J_p: J_p the six rows of the relative Jacobian of the
                                         contact's pair of bodies, T_c a 4 x 6 frame transform (constraints.py:429-433).  The augmented
                                         system then carries six columns Y J_p^T per PAIR instead of four per contact, and Y' = T (J_p Y
                                         J_p^T) T^T, v' = T J_p Y rhs are formed afterwards (float64 sums).  Since round 6 the library does
                                         this BY DEFAULT for every model of the class (the flag is kept for callers of ABI 7, where it
                                         was the default only where it saves the second column set -- human36 with the reference's
                                         eight contact points: 55 columns instead of 75, +16 % world-steps/s).  With four contacts it
                                         costs 1.7 % and halves the float32 world-steps beyond 1e-5 of the float64 reference; none of
                                         those that remain is caused by the device's float32 system (119 808 replayed
                                         world-steps per path).  Equal to the classical columns to rounding, not bit for bit; not for
                                         models outside the class, nor with ARB_STEP_SPLIT_WAVE, ARB_STEP_MFMA_ELIM,
                                         ARB_STEP_GENERAL_KERNELS or ARB_STEP_CLASSIC_COLUMNS.  arb_step_plan_info.feat reports bit 16. */
#define ARB_STEP_MIXED 2048u            /* float32 state buffers, float64 ELIMINATION: the register tile [Z | rhs | J'^T], the
                                         right-hand side, the pivot-free elimination and the constraint-space products of
                                         phases C / D run in float64 (the assembly of Z is float64 in every kernel); twists,
                                         body wrenches and the LDS stay float32.  The reference inverts Z in float64
                                         (core.py:818); a float32 elimination loses log2(Z_jj / pivot_j) bits per pivot, which
                                         on long serial chains (snake-64: 17-20 of 24) leaves nothing: plain float32 is wrong
                                         by 10 % .. 200 % there, this build measures 7e-5 (median 8e-6) at 1.28 x the
                                         throughput of the float64 kernels -- NOT 1e-5: the smallest eigenvalue of that mass
                                         matrix is 3e-9 of the largest, every float32 rounding between the state and the
                                         generalized forces comes back amplified.
                                         What float32 launches run BY DEFAULT is decided per model from the pivot growth g
                                         at its rest states (arb_model_info.rest_pivot_growth, probed once by
                                         arb_model_create with the float32 inspect kernel), arb_model_info.mixed_default:
                                           0  g <= ARB_ILLCOND_GROWTH / 8: the float32 kernels (human36: g < 100);
                                           1  g <= ARB_ILLCOND_GROWTH: this build;
                                           2  above: PROMOTION -- the launch converts state, constraint forces, user torques
                                              and impedance to float64 scratch copies (stream-ordered), runs the FLOAT64
                                              kernels for all nsteps and converts state and forces back: 1e-5 parity through
                                              float32 buffers at float64 throughput.  Launches with per-world PD inputs,
                                              logs or a running cost are not promoted: they run this build.
                                         The flag asks for this build for any model (and instead of the promotion).
                                         Ignored for float64 buffers, with ARB_STEP_MFMA_ELIM and ARB_STEP_SPLIT_WAVE. */
#define ARB_STEP_NO_MIXED 4096u         /* plain float32 kernels whatever the model (ARB_WARN_ILLCOND reports what that costs) */
#define ARB_STEP_CLASSIC_COLUMNS 8192u   /* (ABI 8) never body-space constraint columns: the classical columns Y J'^T, four per contact --
                                         for a model with exactly four contacts the kernels specialised for that class, bit-identical
                                         to the general kernels (ARB_STEP_GENERAL_KERNELS) and to the split execution.  Since round
                                         6 body-space columns are the DEFAULT for every model that qualifies (see
                                         ARB_STEP_BODY_COLUMNS): this flag is the 1.7 % faster, statistically noisier float32 path */
#define ARB_STEP_KNOWN_FLAGS (1u | 2u | 8u | 16u | 32u | 64u | 128u | 256u | 512u | 1024u | 2048u | 4096u | 8192u)

/*
 * Flattened world (host pointers, copied by arb_model_create).  Bodies are the
 * moving bodies in depth-first PREORDER (the DOF numbering of core.py:611-615):
 * parent[b] < b, and the subtree of b is the contiguous range b .. b + size - 1
 * (the kernel forms subtree sums from a prefix scan over that order; any other
 * numbering is refused with ARB_ERR_UNSUPPORTED).  Body b is attached to
 * parent[b] (-1 = ground; several roots are allowed) through one joint.
 */
typedef struct arb_model_desc {
    int32_t abi_version;      /* ARB_ABI_VERSION */
    int32_t nb, ndof, nq, nc;
    const int32_t *parent;    /* [nb] */
    const int32_t *jtype;     /* [nb] ARB_JT_* */
    const int32_t *dof_off;   /* [nb] first dof of the joint */
    const int32_t *q_off;     /* [nb] first position scalar of the joint */
    const double *H_pr;       /* [nb][16] joint.frames[0].bpose (core.py:1295-1296) */
    const double *H_cn;       /* [nb][16] joint.frames[1].bpose */
    const double *mass;       /* [nb][36] Body.mass */
    const double *visc;       /* [nb][36] Body.viscosity */
    const int32_t *weighted;  /* [nb] body is acted on by the WeightController (controllers.py:37) */
    double gravity[3];        /* sum over WeightControllers of gravity*up (controllers.py:40-41) */
    double up[3];             /* World.up (core.py:351), used by the energy monitor */
    /* merged ProportionalDerivativeControllers (controllers.py:141-158), or NULL:
       gforce += pd_tau0 - pd_kp q ;  Z += dt*pd_kp + pd_kd   (dof-indexed, row-major) */
    const double *pd_kp;      /* [ndof][ndof] */
    const double *pd_kd;      /* [ndof][ndof] */
    const double *pd_tau0;    /* [ndof] */
    /* constraints in registration order (core.py:913, 933) */
    const int32_t *ctype;     /* [nc] ARB_CT_* */
    const int32_t *c_enabled; /* [nc] Constraint.is_enabled() */
    const int32_t *c_body;    /* [nc] body of frame 1 (contact point / socket ball), -1 = ground */
    const int32_t *c_body0;   /* [nc] body of frame 0 (contact shape 0 / socket), -1 = ground */
    const int32_t *c_geom;    /* [nc] SoftFingerContact: ARB_CG_* */
    const int32_t *c_dof;     /* [nc] JointLimits: constrained dof */
    const double *c_local;    /* [nc][3] contact point in its body frame */
    const double *c_radius;   /* [nc] sphere radius (0 for a Point) */
    const double *c_radius0;  /* [nc] ARB_CG_SPHERE_SPHERE: radius of shape 0 */
    const double *c_half;     /* [nc][3] ARB_CG_BOX_SPHERE: half extents of shape 0 */
    const double *c_plane;    /* [nc][4] ARB_CG_PLANE_SPHERE: plane coefficients (unit normal, d) */
    const double *c_mu;       /* [nc] friction coefficient */
    const double *c_prox;     /* [nc] proximity (contacts, joint limits) */
    const double *c_eps;      /* [nc][3] SoftFingerContact._eps */
    const double *c_min;      /* [nc] JointLimits */
    const double *c_max;      /* [nc] JointLimits */
    const double *c_bpose0;   /* [nc][16] bpose of frame 0: socket frame / frame of contact shape 0 in c_body0 */
    const double *c_bpose1;   /* [nc][16] BallAndSocket frame 1 bpose */
} arb_model_desc;

typedef struct arb_model arb_model;   /* opaque, device-resident immutable model */

typedef struct arb_model_info {
    int32_t nb, ndof, nq, nc;
    int32_t nmax;             /* register-tile height the kernels were instantiated for */
    int32_t ncols;            /* ndof + 1 + ARB_MAXDOL*nc columns of the augmented system */
    int32_t nsets;            /* 1 or 2 register column sets */
    int32_t lds_bytes_f32;    /* dynamic LDS per world (= per wavefront) */
    int32_t lds_bytes_f64;
    int32_t device;
    int32_t forest_copies;    /* small models: worlds per wavefront of the forest build (ARB_STEP_ONE_WORLD), 1 = none */
    int32_t mixed_default;    /* (ABI 8) what float32 launches of this model run by default: 0 the float32 kernels, 1 the mixed
                                 build, 2 promotion to the float64 kernels (see ARB_STEP_MIXED) */
    int32_t wide;             /* (ABI 8) 1: more than 64 dofs / bodies / 16 constraints: one WORKGROUP per world (the wide kernels,
                                 float64 arithmetic whatever the buffers' type); nmax, nsets and the LDS sizes then describe them.
                                 They take every input of arb_step_ex and arb_inspect; ARB_STEP_SPLIT_WAVE / ARB_STEP_MFMA_ELIM and
                                 the per-solve diagnostics of arb_inspect (gs_stats, gs_trace, stamps, pivot_growth, energy) are
                                 ARB_ERR_UNSUPPORTED there */
    float rest_pivot_growth;  /* (ABI 8) pivot growth of the float32 elimination at the model's rest state (see ARB_WARN_ILLCOND) */
} arb_model_info;

/*
 * Optional per-stage outputs of arb_inspect (device pointers, any may be NULL).
 * They expose what the reference leaves on its objects after each of the four
 * calls, for parity tests and for the single-world object API.
 */
typedef struct arb_inspect_out {
    void *pose;      /* [nw][nb][16]       Body.pose                         core.py:1272 */
    void *twist;     /* [nw][nb][6]        Body.twist                        core.py:1275 */
    void *jac;       /* [nw][nb][6][ndof]  Body.jacobian                     core.py:1273 */
    void *djac;      /* [nw][nb][6][ndof]  Body.djacobian                    core.py:1274 */
    void *M;         /* [nw][ndof][ndof]   World.mass                        core.py:726-728 */
    void *B;         /* [nw][ndof][ndof]   World.viscosity                   core.py:729-731 */
    void *N;         /* [nw][ndof][ndof]   World.nleffects                   core.py:732-734 */
    void *Z;         /* [nw][ndof][ndof]   World._impedance                  core.py:813-817 */
    void *gforce0;   /* [nw][ndof]         controllers' gforce               core.py:812-816 */
    void *vel_free;  /* [nw][ndof]         Y (M gvel/dt + gforce0): new gvel without constraints */
    void *c_sdist;   /* [nw][nc]           PointContact._sdist / |p_01| ...  constraints.py:293 */
    void *c_active;  /* [nw][nc] int32     Constraint.is_active()            core.py:916 */
    void *c_jac;     /* [nw][nc][4][ndof]  Constraint.jacobian (0 if inactive) core.py:923 */
    void *c_force;   /* [nw][nc][4]        Constraint._force after the 20 sweeps  core.py:929-935 */
    void *c_frame;   /* [nw][nc][2][16]    contact frame poses H_gc0, H_gc1  constraints.py:284-288 */
    void *gforce;    /* [nw][ndof]         World._gforce incl. constraints   core.py:936-937 */
    void *q_next;    /* [nw][nq]           state after integrate             core.py:974-980 */
    void *dq_next;   /* [nw][ndof] */
    void *gs_stats;  /* [nw][5] int32  Gauss-Seidel solve counts of the step: SoftFingerContact release,
                        static, sliding via the fast shift, sliding via the eig6 fallback, and the number
                        of sweeps executed before the iteration reached a bit-exact fixed point (diagnostic) */
    void *energy;    /* [nw][2]            kinetic and potential energy, EnergyMonitor.update observers.py:40-51 */
    void *stamps;    /* [nw][8] int64  shader clock at the phase boundaries A, A', B, C, D, GS, E, end (diagnostic) */
    void *gs_trace;  /* [nw][20][nc] int32  decision of every local solve of the 20 sweeps (core.py:929-935) in execution
                        order: 0 release, 1 static friction, 2 sliding (sextic shift), 3 sliding (eig6 fallback)
                        (SoftFingerContact.solve, constraints.py:781-836), 4 = another constraint type.  Entries of solves
                        that were not executed (inactive constraint, sweeps after the fixed point) are left untouched:
                        pre-fill with -1. */
    void *c_adm;     /* [nw][4 nc][4 nc]  the constraint-space admittance Y' = J' Y J'^T the sweeps run on (core.py:927);
                        rows / columns of inactive constraints are zero */
    void *c_vel;     /* [nw][4 nc]        v' = J' Y (M gvel/dt + gforce') before the sweeps (core.py:925-926) */
    void *pivot_growth; /* [nw]  (ABI 7) max over the dofs j of |Z_jj| / |pivot_j| in the pivot-free elimination of the
                        impedance (core.py:818): how many digits the subtraction that leaves pivot j cancels.  The float32
                        kernels raise ARB_WARN_ILLCOND from it (see arb_model_warnings). */
} arb_inspect_out;

int arb_abi_version(void);
const char *arb_strerror(int status);
const char *arb_last_hip_error(void);

/* Build the device-resident model on HIP device `device`. Replaces World.init
 * (core.py:608-635) + the per-step reads of the object graph. */
int arb_model_create(const arb_model_desc *desc, int device, arb_model **out);
int arb_model_destroy(arb_model *m);
int arb_model_get_info(const arb_model *m, arb_model_info *info);

/*
 * Which kernel build and launch shape arb_step / arb_step_ex would use for a batch (diagnostics: the benchmark records it,
 * the tests check the batch-size rules).  The float32 step kernel of a model with 33 <= ndof <= 48 and ndof + 1 + 4 nc
 * <= 64 exists in two bit-identical builds -- two waves per SIMD (no register spills) and three waves per SIMD (more waves
 * in flight) -- picked by batch size and launch shape; ARB_STEP_WAVES2 / ARB_STEP_WAVES3 pin one.  Models of at most 16 dofs: see ARB_STEP_ONE_WORLD.
 *   wave_slots is an estimate from the registers of the build and the 1280-byte granule in which a CU's 160 KB of LDS are
 *   handed out (hipOccupancyMaxActiveBlocksPerMultiprocessor divides 160 KB by the request and overestimates: twelve
 *   wavefronts per CU fit up to 12 800 B each, not 13 653 B).
 *   optional_inputs: 0 = none, 1 = ext_gforce only, 3 = per-world PD inputs / logs / dt_steps; + 4 = the launch logs
 *   per-world energies, or states for a batch that is not a multiple of the forest's copies (such launches of a small
 *   model run one world per wavefront: without the bit the plan describes arb_step / arb_step_ex without those logs)
 *   wave_slots comes from the model the launch itself uses (wavefronts per SIMD by the build's registers, LDS by the
 *   1280-byte granule), so `work_queue` is what the launch does.
 */
typedef struct arb_step_plan_info {
    int32_t waves_per_simd;        /* register budget the chosen build was compiled for: 1, 2 or 3 */
    int32_t worlds_per_wavefront;  /* 1, or the copies of a small model's forest (the other fields
                                      then describe the launch of the forest) */
    int32_t feat;                  /* optional-input set of the kernel instantiation: 0, 1 or 3; + 4: the kernel specialised for
                                      models with exactly four (eight: two column sets) enabled plane / sphere
                                      SoftFingerContacts, + 8: for models without constraints -- in both classes no PD
                                      controller, no joint viscosity, a small shallow tree; plain inputs or user torques;
                                      results bit-identical to the general kernels' */
    int32_t lds_bytes;             /* dynamic LDS per wavefront */
    int32_t wave_slots;            /* resident wavefronts of that build on the device */
    int32_t work_queue;            /* 1: the resident wavefronts draw (chunk of steps, world or pair) items from a device-side queue */
} arb_step_plan_info;
int arb_step_plan(arb_model *m, int dtype, int64_t nworlds, int32_t nsteps, uint32_t flags, int32_t optional_inputs,
                  arb_step_plan_info *out);

/*
 * Health of the handle's launches.  The device-side work queue of multi-step launches orders the chunks of a world
 * through a flag per world; a wavefront that waits for a flag longer than ~10 s (a stalled producer: a debugger, a
 * preempted queue, counter collection that serialises workgroups) gives up, does NOT advance or publish its item and
 * raises a status word in host-visible memory; the world's flag stays poisoned for the rest of that launch, so none of
 * its later chunks runs (or waits) either.  arb_model_status returns ARB_OK or ARB_ERR_STALLED and CLEARS the word: it
 * is the caller's acknowledgement, and the only call that clears.  It reads host memory only (no synchronisation: call
 * it after synchronising the stream to learn about the launches queued so far).  Every arb_step / arb_step_ex /
 * arb_rollout / arb_inspect call looks at the word on entry and returns ARB_ERR_STALLED instead of launching for as
 * long as it is raised (sticky since round 4: with asynchronous callers the first call to notice a stall is not
 * necessarily one whose status is checked; up to ABI 5.0 that call cleared the word and the next one ran on the invalid
 * state with ARB_OK).  The states of the stalled launch are invalid: reload them before stepping on.
 * (The knob "queue_spin_cap" of arbstep_hooks.h, the number of polls of that wait, is the tests' fault injection only.)
 */
int arb_model_status(arb_model *m);

/*
 * Warnings of the handle's launches so far (ABI 7): a bit mask, returned in *warnings and CLEARED.  Like arb_model_status
 * it reads host memory only.
 *   ARB_WARN_ILLCOND   a float32 launch met a world whose impedance matrix Z = M/dt + B + N (core.py:813-818) loses more
 *                      digits in the elimination than float32 can spare: for some dof j the pivot that is left,
 *                      Z_jj - (what the dofs eliminated before j take away), is smaller than Z_jj / ARB_ILLCOND_GROWTH --
 *                      the subtraction cancels log10 of that ratio digits of float32's seven, and the velocities
 *                      the step returns are accurate to what is left at best.  Long serial chains do this (snake-64:
 *                      growth ~1e5, velocity error 0.25 in float32), branched bodies of the size of human36 do not
 *                      (growth < 100).  The reference computes in float64 throughout (core.py:818): step such a
 *                      model with ARB_F64.  The results are still written; the warning does not stop anything.
 *                      (Since ABI 8 only launches pinned with ARB_STEP_NO_MIXED can raise it for such a model: by default its
 *                      float32 launches are promoted to the float64 kernels, see ARB_STEP_MIXED.)
 *   ARB_WARN_ACTIVE_CONSTRAINTS   (ABI 8, worlds on the wide kernels: arb_model_info.wide) more than 64 of a world's constraints
 *                      were ACTIVE in one step.  Such a world may register up to ARB_WIDE_MAX_CONSTRAINTS -- every pair of
 *                      get_all_contacts, constraints.py:840-875 -- and its steps solve on the active ones; beyond 64 the later
 *                      ones in registration order are left out of that step's solve (the reference would stack them all,
 *                      core.py:898-935).
 */
#define ARB_WARN_ILLCOND 1u
#define ARB_WARN_ACTIVE_CONSTRAINTS 2u
#define ARB_ILLCOND_GROWTH 2048.0   /* 2^11: the float32 kernels compare exponents */
int arb_model_warnings(arb_model *m, uint32_t *warnings);

/*
 * Advance `nworlds` independent worlds by `nsteps` steps of `dt`, in place.
 * Replaces nsteps iterations of core.py:1358-1363 (without observers).
 *   q, dq    device, dtype-typed, [nworlds][nq] / [nworlds][ndof], updated in place
 *   cforce   device [nworlds][nc][ARB_MAXDOL] or NULL: constraint forces; read as the
 *            warm start of BallAndSocket constraints and written back after the
 *            last step (contact / joint-limit forces of that step)
 *   ext_gforce  device [nworlds][ndof] or NULL: extra generalized force with zero
 *            impedance, constant over the call (a user torque Controller)
 * Multi-step calls on more worlds than the GPU holds wavefronts run through a device-side work queue (see
 * ARB_STEP_STATIC_WORLDS): identical results; while the call runs, q / dq / cforce hold intermediate states of
 * the worlds (they are the hand-over buffers between work items), as they would between single-step calls.
 */
int arb_step(arb_model *m, int dtype, void *q, void *dq, void *cforce,
             const void *ext_gforce, int64_t nworlds, double dt, int32_t nsteps,
             uint32_t flags, void *stream);

/*
 * Per-step logs of a rollout (device pointers, any may be NULL): the state and the
 * energies an Observer would see at each step (core.py:1361-1362: after
 * update_constraints, before integrate, i.e. the state at time t).
 * Counterpart of observers.Hdf5Logger(save_state=True) (observers.py:222-229, 264-267)
 * and observers.EnergyMonitor (observers.py:40-51), batched.
 */
typedef struct arb_rollout_log {
    void *q_log;       /* [nsteps][nworlds][nq]    */
    void *dq_log;      /* [nsteps][nworlds][ndof]  */
    void *energy_log;  /* [nsteps][nworlds][2]     kinetic, potential */
} arb_rollout_log;

/* arb_step + per-step logs. */
int arb_rollout(arb_model *m, int dtype, void *q, void *dq, void *cforce, const void *ext_gforce,
                int64_t nworlds, double dt, int32_t nsteps, uint32_t flags,
                const arb_rollout_log *log, void *stream);

/*
 * The general form of arb_step / arb_rollout, with the per-world controller inputs that
 * stand for one ProportionalDerivativeController PER WORLD (controllers.py:63-158):
 *     gforce += Kp (qdes_w - q) + Kd dqdes_w ,   Z += dt Kp + Kd        (controllers.py:141-158)
 *   pd_qdes, pd_dqdes  device [nworlds][ndof] or both NULL: desired positions and velocities,
 *            dof-indexed (entries of dofs no gain touches are ignored).  Without pd_kp/pd_kd the
 *            gains are the model's merged Kp, Kd (arb_model_desc.pd_kp/pd_kd, required then) and
 *            these targets replace the model's pd_tau0.
 *   pd_kp, pd_kd  device [nworlds][ndof] or both NULL: per-world DIAGONAL gains; they replace the
 *            model's gain matrices altogether and need pd_qdes/pd_dqdes.
 *   log      NULL (arb_step) or the per-step logs (arb_rollout)
 *   dt_steps one dt per step (device, float64) instead of the uniform `dt`
 * Other fields as in arb_step.  ext_gforce is the hook for user torques (MPC inputs), constant over the call;
 * ext_gforce_steps is a torque SEQUENCE (one row per step and world), read step by step inside the launch.
 * Zero-initialise the struct (memset): fields added by later ABI versions then mean "absent".
 */
/*
 * Running cost of a rollout (ABI 7; SURVEY 8d config 5: an MPC horizon returns a cost per rollout, 8e: the costs are what
 * the ranks exchange).  With u_t the user torques of step t and x_{t+1} = (q, dq) the state AFTER step t,
 *     cost_out[w] += sum over the launch's steps t of
 *                    sum_i  w_q[i] (qj_i - q_ref[i])^2 + w_dq[i] dq_i^2 + w_tau[i] u_t,i^2
 * where qj is the dof-indexed vector of the linear joint positions (0 for the dofs of a FreeJoint, whose position is a
 * pose) -- a diagonal quadratic form of (q, dq, tau), evaluated on chip at the end of every step; only the sum leaves
 * the chip.  All pointers are DEVICE pointers of the state's dtype; a NULL weight or reference is zero.  cost_out is
 * read and written (initialise it; a horizon cut into several launches accumulates: the additions are made one step
 * at a time in step order, so the sum is bit for bit the same however the horizon is cut).
 */
typedef struct arb_step_cost {
    void *cost_out;            /* [nworlds] */
    const void *w_q;           /* [ndof] or NULL */
    const void *w_dq;          /* [ndof] or NULL */
    const void *w_tau;         /* [ndof] or NULL */
    const void *q_ref;         /* [ndof] or NULL */
} arb_step_cost;

typedef struct arb_step_args {
    void *q, *dq, *cforce;
    const void *ext_gforce;
    const void *pd_qdes, *pd_dqdes;
    const void *pd_kp, *pd_kd;
    int64_t nworlds;
    double dt;
    int32_t nsteps;
    uint32_t flags;
    const arb_rollout_log *log;
    const double *dt_steps;   /* DEVICE pointer [nsteps] (always float64) or NULL: the dt of every step, for a
                                 non-uniform timeline inside one launch (core.py:1357: dt = next_time - current_time);
                                 when given, `dt` is ignored */
    /* ---- ABI 7: control inputs that change along the horizon (the reference polls every controller every step,
       core.py:811-817, controllers.py:141-158) -- one launch simulates a control SEQUENCE ---- */
    const void *ext_gforce_steps;   /* DEVICE [nsteps][nworlds][ndof] or NULL: the user torques of every step; replaces
                                       ext_gforce (giving both is ARB_ERR_INVALID) */
    const void *pd_qdes_steps;      /* DEVICE [nsteps][nworlds][ndof] or both NULL: the PD targets of every step; replace */
    const void *pd_dqdes_steps;     /* pd_qdes / pd_dqdes (same rules for the gains) */
    const arb_step_cost *cost;      /* NULL or the running cost (not with ARB_STEP_SPLIT_WAVE: ARB_ERR_INVALID) */
    /* ---- ABI 8: the generic Controller plugin path.  A reference Controller returns (gforce_a, Z_a) from update(dt)
       (core.py:327-339) and World.update_controllers sums `gforce += gforce_a; impedance -= Z_a` (core.py:814-817).  The
       built-in controllers are lowered to the kernels (gravity, PD); ANY other controller is expressed by the caller as the
       sum of the gforce_a in ext_gforce and the sum of the Z_a here.  The object API (core.World.update_controllers) does
       exactly that for user-defined Controller subclasses, once per step on the host. ---- */
    const void *ext_impedance;      /* DEVICE [nworlds][ndof][ndof] row-major or NULL: Z -= ext_impedance, constant over the
                                       call.  Worlds of a small model run one per wavefront when it is given. */
} arb_step_args;

int arb_step_ex(arb_model *m, int dtype, const arb_step_args *args, void *stream);

/* Evaluate one step WITHOUT modifying q/dq and write the requested intermediate
 * results.  Same arithmetic as arb_step (same kernels, debug stores enabled). */
int arb_inspect(arb_model *m, int dtype, const void *q, const void *dq,
                const void *cforce, const void *ext_gforce, int64_t nworlds, double dt,
                uint32_t flags, const arb_inspect_out *out, void *stream);

/* (ABI 8) arb_inspect with the inputs of arb_step_ex (per-world PD inputs, ext_gforce, ext_impedance; q / dq are not
 * modified; nsteps is ignored: one step).  Control sequences, dt_steps, log and cost are refused (ARB_ERR_INVALID). */
int arb_inspect_ex(arb_model *m, int dtype, const arb_step_args *args, const arb_inspect_out *out, void *stream);

#define ARB_WIDE_MAX_CONSTRAINTS 256   /* (ABI 8) constraints a wide world may REGISTER (every pair of get_all_contacts); at most 64 active per step */
#define ARB_WIDE_MAX 1024  /* (ABI 8) largest ndof / nb of a world (the wide kernels: one workgroup of 256 lanes per world; past 192 dofs the
                              * augmented system lives in scratch memory: a capability, ~n^3 slower) */

#ifdef __cplusplus
}
#endif
#endif /* ARBSTEP_H */
