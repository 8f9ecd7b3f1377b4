/*
 * arbstep_hooks.h -- unit-test and single-constraint entry points of libarbstep.so.
 *
 * The kernels' small hand-written solvers (arboris_python_amd/csrc/arb_math.h) are written once and compiled twice:
 * for gfx950 inside the step kernels, and for the host behind the `arb_host_*` functions below, so that the CPU test
 * suite can check them against captured reference tuples without a GPU, and so that the object API's
 * `Constraint.solve` (arboris/constraints.py:73-90, 235-237, 780-836 called on its own, outside a step) runs the
 * SAME code as the kernels instead of a second implementation.  `arb_dev_*` runs the device build on explicit inputs.
 *
 * They are not part of the batched step boundary (include/arbstep.h): plain C, host pointers, float64 in and out
 * whatever `dtype` (ARB_F32 rounds the inputs and computes in float32 like the float32 kernels do).
 */
#ifndef ARBSTEP_HOOKS_H
#define ARBSTEP_HOOKS_H

#include "arbstep.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Hg.zaligned(z) -> rotation matrix R[9] (row-major), arboris/homogeneousmatrix.py:201-232. */
void arb_host_zaligned(const double z[3], double R[9]);

/* Narrow phase of one shape pair (ARB_CG_*), arboris/collisions.py:67-299: shape 0's frame pose H_s0 (4x4), centre of
 * shape 1 p_g1, its radius `rad`, shape 0's radius / half extents / plane coefficients; returns the signed distance
 * and writes the contact-frame origins gc0, gc1 and their common rotation Rc[9]. */
double arb_host_narrow_phase(int geom, const double H_s0[16], const double p_g1[3], double rad,
                             double r0, const double half[3], const double plane[4],
                             double gc0[3], double gc1[3], double Rc[9]);

/* SoftFingerContact.solve, arboris/constraints.py:780-836: vel[4], adm[16], force[4] (in: current, out: new),
 * eps[3], dforce[4] (out).  dtype: ARB_F32 / ARB_F64, optionally | 0x100 to force the generic eig6 route of the
 * sliding branch.  Returns the branch taken: 0 release, 1 static, 2 sliding; -1 on a null argument. */
int arb_host_softfinger_solve(int dtype, const double *vel, const double *adm, double *force,
                              double sdist, double dt, double mu, const double *eps, double *dforce);

/* The same solve executed ON THE DEVICE, one lane per tuple (needs a GPU): in [n][27] = vel 4 | adm 16 | force 4 |
 * sdist, dt, mu;  out [n][9] = force 4 | dforce 4 | branch.  Returns an ARB_* status. */
int arb_dev_softfinger_solve(int dtype, int device, int n, const double *in, double *out);
/* The generic 6x6 eigenvalue route of the sliding solve on the device, n matrices [n][36] row-major: the one-lane routine
 * and the wavefront routine (arb_math.h: eig6, eig6_wave) side by side; out [n][28] = shift, number found, wr[6], wi[6]
 * of the first, then of the second.  The kernels run the second; the two must agree bit for bit. */
int arb_dev_eig6_pair(int dtype, int device, int n, const double *A, double *out);
/* Build variants compiled into the loaded library: bit 2 Gauss-Seidel sweeps that run the complete variant of the local solve
 * throughout (no fast variant), bit 3 no specialised kernels (the general kernels also for models with four plane / sphere
 * SoftFingerContacts).  0 for libarbstep.so; 12 for libarbstep_variants.so (make variants), which the bit-identity tests of the
 * fast sweeps and of the specialised kernels load.  (Bits 0 and 1 were the packed and rendezvous builds of rounds 3 and 4:
 * bit-identical, measured slower at every batch size, removed in round 5.) */
int arb_build_variants(void);

/* Development / test knobs of ONE handle (ABI 7: the library itself reads no environment variable; a build with
 * -DARB_DEVELOPMENT reads ARB_<NAME IN CAPITALS> once at arb_model_create and calls this).  Names and defaults:
 *   "queue_spin_cap" (1 << 24)  polls after which a wavefront gives up waiting for a chunk of the work queue; negative:
 *                               every wait of a later chunk expires at once -- the fault injection of the ARB_ERR_STALLED tests
 *   "queue_chunk" (4), "queue_tail" (6)  steps per work item, single-step items at the end of an episode (0 chunk: no queue)
 *   "lds_pad" (0)               bytes of dynamic LDS added to every step-kernel workgroup (occupancy experiments)
 *   "force_waves" (0)           2 | 3: pin the float32 build whatever the flags say
 *   "gsw_waves" (3)             the split execution's sweep kernel: 3 | 4 waves per SIMD
 *   "ablate" (0)                inspect kernels: bit 3 = run all 20 Gauss-Seidel sweeps (no fixed-point exit)
 *   "wide_compact" (1)          wide kernels (worlds past 64 dofs): 0 = the build that keeps the augmented system in LDS / scratch
 *                               where the compact build (system in registers; <= 192 dofs, <= 256 columns) is the default
 *   "wide_gs_groups" (1)        wide kernels: 0 = one serial sequence of Gauss-Seidel solves over all constraints instead of the
 *                               independent groups side by side (both knobs: bit-identical results, tests/test_gpu_wide.py)
 * Returns ARB_OK, or ARB_ERR_INVALID for an unknown name.  Applies to the handle's forest as well. */
int arb_hook_set_knob(arb_model *m, const char *name, int value);

/* Raw branch code of the first stage of the solve: 0, 1, 2 as above, 3 = sliding but the register-only shift
 * declined and the 6x6 eigenvalue fallback is needed. */
int arb_host_softfinger_try(int dtype, const double *vel, const double *adm, const double *force,
                            double sdist, double dt, double mu, const double *eps);

/* Leftmost real root of the sliding branch's sextic det(B - sI) (constraints.py:805-830 with eps = (1,1,1)) for the
 * 4x4 admittance block Y and the two sweep-dependent scalars c1, kappa; `warm` = previous root or NaN.
 * Returns 1 and *root on success, 0 when the caller must fall back to eig6. */
int arb_host_slide_root(const double *Y, double c1, double kappa, double warm, double *root);

/* What decides when that iteration declines (a complex pair of roots leftmost, or nearly so), before the 6x6 eigenvalue
 * routine is asked: the derivative cascade (arb_math.h: slide_real_root_cascade) on a sextic given by its seven
 * coefficients pc (constant term first) -- the leftmost real root in [lo, 0], which is the smallest real eigenvalue <= 0 of
 * constraints.py:826-830 when lo is left of the spectrum.  Returns 1 with *root, 0 when there is no real root in [lo, 0],
 * -1 for non-finite input. */
int arb_host_real_root_cascade(const double *pc, double lo, double *root);

/* (Pseudo-)inverse of an nd x nd constraint block as the kernels form it (numpy.linalg.pinv at
 * constraints.py:79, 83, 235, 795): returns 1 when pivoted elimination was kept, 0 when the block was found rank
 * deficient and the Jacobi-SVD route was taken, -1 on a bad argument. */
int arb_host_block_pinv(int dtype, int nd, const double *Y, double *P);

/* Eigenvalues of a real 6x6 matrix (numpy.linalg.eigvals at constraints.py:825): wr/wi[6]; returns how many of the
 * trailing entries converged. */
int arb_host_eig6(const double *A, double *wr, double *wi);

/* Joint-local kinematics of one joint (ARB_JT_*), arboris/joints.py: out = R 9 | p 3 | Jacobian angular columns 9 |
 * their derivatives 9 | relative twist 6. */
int arb_host_joint_local(int jt, const double *q, const double *dq, double *out);

/* The pivot-growth measure of the float32 elimination (phase C; ARB_WARN_ILLCOND of arbstep.h) for one assembled diagonal
 * entry and the pivot left of it, as the kernels compute it on the scalar unit from the floats' bit patterns: ~2^23 log2(zjj /
 * pivot), INT32_MAX for a pivot <= 0 (or -0.0, NaN, inf): the warning fires above 11 << 23. */
int arb_host_growth_bits(float zjj, float pivot);

/* twistvector.exp, arboris/twistvector.py:35-70: tw[6] -> H[16]. */
int arb_host_exp_twist(const double *tw, double *H);

#ifdef __cplusplus
}
#endif
#endif /* ARBSTEP_HOOKS_H */
