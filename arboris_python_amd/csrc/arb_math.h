// arb_math.h -- small fixed-size SE(3) / linear-algebra helpers shared by the
// gfx950 kernels (arb_kernels.hip) and by the host-side self-test hooks.
//
// Everything here is scalar code on registers: in the kernels it runs either
// once per lane with lane = body / lane = constraint (phase A), or wave-uniform
// (Gauss-Seidel).  Conventions follow the reference: twists are [w; v],
// Ad(R,p) = [[R,0],[p^R,R]] (arboris/homogeneousmatrix.py:277-319).
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define ARB_HD __host__ __device__ __forceinline__
#else
#define ARB_HD inline
#endif

template <typename T> struct V3 { T x, y, z; };
template <typename T> struct M3 { T a[9]; };      // row-major

template <typename T> ARB_HD V3<T> v3(T x, T y, T z) { V3<T> r; r.x = x; r.y = y; r.z = z; return r; }
template <typename T> ARB_HD V3<T> operator+(V3<T> a, V3<T> b) { return v3<T>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <typename T> ARB_HD V3<T> operator-(V3<T> a, V3<T> b) { return v3<T>(a.x - b.x, a.y - b.y, a.z - b.z); }
template <typename T> ARB_HD V3<T> operator-(V3<T> a) { return v3<T>(-a.x, -a.y, -a.z); }
template <typename T> ARB_HD V3<T> operator*(T s, V3<T> a) { return v3<T>(s * a.x, s * a.y, s * a.z); }
template <typename T> ARB_HD T dot(V3<T> a, V3<T> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <typename T> ARB_HD V3<T> cross(V3<T> a, V3<T> b) {
    return v3<T>(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
template <typename T> ARB_HD M3<T> m3_identity() {
    M3<T> r; for (int i = 0; i < 9; ++i) r.a[i] = T(0); r.a[0] = r.a[4] = r.a[8] = T(1); return r;
}
template <typename T> ARB_HD M3<T> m3_zero() { M3<T> r; for (int i = 0; i < 9; ++i) r.a[i] = T(0); return r; }
template <typename T> ARB_HD M3<T> mul(const M3<T> &A, const M3<T> &B) {          // A B
    M3<T> r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            r.a[3 * i + j] = A.a[3 * i] * B.a[j] + A.a[3 * i + 1] * B.a[3 + j] + A.a[3 * i + 2] * B.a[6 + j];
    return r;
}
template <typename T> ARB_HD M3<T> mulTA(const M3<T> &A, const M3<T> &B) {        // A^T B
    M3<T> r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            r.a[3 * i + j] = A.a[i] * B.a[j] + A.a[3 + i] * B.a[3 + j] + A.a[6 + i] * B.a[6 + j];
    return r;
}
template <typename T> ARB_HD M3<T> mulBT(const M3<T> &A, const M3<T> &B) {        // A B^T
    M3<T> r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            r.a[3 * i + j] = A.a[3 * i] * B.a[3 * j] + A.a[3 * i + 1] * B.a[3 * j + 1] + A.a[3 * i + 2] * B.a[3 * j + 2];
    return r;
}
template <typename T> ARB_HD M3<T> transpose(const M3<T> &A) {
    M3<T> r;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r.a[3 * i + j] = A.a[3 * j + i];
    return r;
}
template <typename T> ARB_HD M3<T> add(const M3<T> &A, const M3<T> &B) {
    M3<T> r; for (int i = 0; i < 9; ++i) r.a[i] = A.a[i] + B.a[i]; return r;
}
template <typename T> ARB_HD M3<T> sub(const M3<T> &A, const M3<T> &B) {
    M3<T> r; for (int i = 0; i < 9; ++i) r.a[i] = A.a[i] - B.a[i]; return r;
}
template <typename T> ARB_HD V3<T> mv(const M3<T> &A, V3<T> v) {                  // A v
    return v3<T>(A.a[0] * v.x + A.a[1] * v.y + A.a[2] * v.z,
                 A.a[3] * v.x + A.a[4] * v.y + A.a[5] * v.z,
                 A.a[6] * v.x + A.a[7] * v.y + A.a[8] * v.z);
}
template <typename T> ARB_HD V3<T> mtv(const M3<T> &A, V3<T> v) {                 // A^T v
    return v3<T>(A.a[0] * v.x + A.a[3] * v.y + A.a[6] * v.z,
                 A.a[1] * v.x + A.a[4] * v.y + A.a[7] * v.z,
                 A.a[2] * v.x + A.a[5] * v.y + A.a[8] * v.z);
}
// zaligned(vec), arboris/homogeneousmatrix.py:201-232: rotation whose third column is z and whose
// first column zeroes the smallest |z_i| (stable argsort, as numpy's for three elements).
ARB_HD M3<double> zaligned_rot(V3<double> z) {
    const double a0 = fabs(z.x), a1 = fabs(z.y), a2 = fabs(z.z);
    int i1, i2;                                   // middle and largest component
    if (a0 <= a1) {
        if (a1 <= a2) { i1 = 1; i2 = 2; } else if (a0 <= a2) { i1 = 2; i2 = 1; } else { i1 = 0; i2 = 1; }
    } else {
        if (a0 <= a2) { i1 = 0; i2 = 2; } else if (a1 <= a2) { i1 = 2; i2 = 0; } else { i1 = 1; i2 = 0; }
    }
    const double z1 = i1 == 0 ? z.x : (i1 == 1 ? z.y : z.z);
    const double z2 = i2 == 0 ? z.x : (i2 == 1 ? z.y : z.z);
    V3<double> x;                                 // x[i0] = 0, x[i1] = z[i2], x[i2] = -z[i1]
    x.x = i1 == 0 ? z2 : (i2 == 0 ? -z1 : 0.);
    x.y = i1 == 1 ? z2 : (i2 == 1 ? -z1 : 0.);
    x.z = i1 == 2 ? z2 : (i2 == 2 ? -z1 : 0.);
    x = (1. / sqrt(dot(x, x))) * x;
    const V3<double> y = cross(z, x);
    M3<double> R;
    R.a[0] = x.x; R.a[1] = y.x; R.a[2] = z.x;
    R.a[3] = x.y; R.a[4] = y.y; R.a[5] = z.y;
    R.a[6] = x.z; R.a[7] = y.z; R.a[8] = z.z;
    return R;
}

// Narrow phase of a SoftFingerContact, arboris/collisions.py:67-299, in float64.  (Rs0, ps0): pose of
// shape 0's frame; p_g1: centre of shape 1 (Sphere/Point of radius `rad`).  Outputs the signed
// distance, the origins of the two contact frames and their common rotation.
ARB_HD double narrow_phase(int geom, const M3<double> &Rs0, V3<double> ps0, V3<double> p_g1, double rad,
                           double r0, V3<double> he, V3<double> pn, double pd, const M3<double> &Rz,
                           V3<double> &gc0, V3<double> &gc1, M3<double> &Rc) {
    double sd;
    if (geom == 0) {                                        // plane / sphere      collisions.py:194-205
        // (the reference leaves both contact frames in the PLANE's coordinates)
        const V3<double> p01 = mtv(Rs0, p_g1 - ps0);
        const double csd = dot(pn, p01) - pd;
        sd = csd - rad;
        const double sg = sd > 0. ? 1. : (sd < 0. ? -1. : 0.);
        gc0 = p01 - csd * pn;
        gc1 = p01 - (sg * rad) * pn;
        Rc = Rz;
    } else if (geom == 1) {                                 // sphere / sphere     collisions.py:149-159
        const V3<double> vec = p_g1 - ps0;
        const double len = sqrt(dot(vec, vec));
        sd = len - r0 - rad;
        const V3<double> z = v3<double>(vec.x / len, vec.y / len, vec.z / len);
        Rc = zaligned_rot(z);
        gc0 = ps0 + r0 * z;
        gc1 = gc0 + sd * z;
    } else {                                                // box / sphere        collisions.py:268-299
        const V3<double> p01 = mtv(Rs0, p_g1 - ps0);
        V3<double> f0, nrm;
        if (fabs(p01.x) <= he.x && fabs(p01.y) <= he.y && fabs(p01.z) <= he.z) {
            // centre inside the box: nearest face = first minimum of [he - p, he + p]; the normal
            // stays in box coordinates, as in the reference
            const double g[6] = {he.x - p01.x, he.y - p01.y, he.z - p01.z, he.x + p01.x, he.y + p01.y, he.z + p01.z};
            int im = 0;
            double gm = g[0];
            for (int i = 1; i < 6; ++i) if (g[i] < gm) { gm = g[i]; im = i; }
            f0 = p01;
            nrm = v3<double>(0., 0., 0.);
            if (im == 0) { f0.x = he.x; nrm.x = 1.; } else if (im == 1) { f0.y = he.y; nrm.y = 1.; }
            else if (im == 2) { f0.z = he.z; nrm.z = 1.; } else if (im == 3) { f0.x = -he.x; nrm.x = -1.; }
            else if (im == 4) { f0.y = -he.y; nrm.y = -1.; } else { f0.z = -he.z; nrm.z = -1.; }
            gc0 = mv(Rs0, f0) + ps0;
            const V3<double> dv = gc0 - p_g1;
            sd = -sqrt(dot(dv, dv)) - rad;
        } else {
            f0 = v3<double>(fmax(fmin(he.x, p01.x), -he.x), fmax(fmin(he.y, p01.y), -he.y), fmax(fmin(he.z, p01.z), -he.z));
            gc0 = mv(Rs0, f0) + ps0;
            const V3<double> vec = p_g1 - gc0;
            const double len = sqrt(dot(vec, vec));
            nrm = v3<double>(vec.x / len, vec.y / len, vec.z / len);
            sd = len - rad;
        }
        Rc = zaligned_rot(nrm);
        gc1 = p_g1 - rad * nrm;
    }
    return sd;
}

template <typename T> ARB_HD M3<T> hat(V3<T> p) {                                  // p^
    M3<T> r;
    r.a[0] = T(0); r.a[1] = -p.z; r.a[2] = p.y;
    r.a[3] = p.z; r.a[4] = T(0); r.a[5] = -p.x;
    r.a[6] = -p.y; r.a[7] = p.x; r.a[8] = T(0);
    return r;
}
template <typename T> ARB_HD M3<T> hatmul(V3<T> p, const M3<T> &R) {              // p^ R
    M3<T> r;
    for (int j = 0; j < 3; ++j) {
        V3<T> c = cross(p, v3<T>(R.a[j], R.a[3 + j], R.a[6 + j]));
        r.a[j] = c.x; r.a[3 + j] = c.y; r.a[6 + j] = c.z;
    }
    return r;
}

// A 6x6 matrix of the closed family [[A,0],[B,A]] (adjoints, ad-matrices and
// their products all have this shape).
template <typename T> struct Blk { M3<T> A, B; };
template <typename T> ARB_HD Blk<T> blk_adjoint(const M3<T> &R, V3<T> p) {         // Ad(R,p)
    Blk<T> r; r.A = R; r.B = hatmul(p, R); return r;
}
template <typename T> ARB_HD Blk<T> blk_adjacency(V3<T> w, V3<T> v) {              // ad([w;v]) twistvector.py:27-33
    Blk<T> r; r.A = hat(w); r.B = hat(v); return r;
}
template <typename T> ARB_HD Blk<T> blk_mul(const Blk<T> &X, const Blk<T> &Y) {
    Blk<T> r; r.A = mul(X.A, Y.A); r.B = add(mul(X.B, Y.A), mul(X.A, Y.B)); return r;
}

// ---------------------------------------------------------------------------
// Joint-local kinematics, arboris/joints.py + homogeneousmatrix.py:11-199.
// Rotational joints only have angular Jacobian columns (jw, djw); TxTyTz has
// unit linear columns; the FreeJoint Jacobian is the identity.
// ---------------------------------------------------------------------------
enum { JT_FREE = 0, JT_RZRYRX, JT_RZRY, JT_RZRX, JT_RYRX, JT_RZ, JT_RY, JT_RX, JT_TXTYTZ };

ARB_HD int joint_ndof(int jt) {
    switch (jt) {
        case JT_FREE: return 6;
        case JT_RZRYRX: case JT_TXTYTZ: return 3;
        case JT_RZRY: case JT_RZRX: case JT_RYRX: return 2;
        default: return 1;
    }
}
ARB_HD int joint_nq(int jt) { return jt == JT_FREE ? 16 : joint_ndof(jt); }

ARB_HD float arb_abs(float x) { return x < 0.f ? -x : x; }
ARB_HD double arb_abs(double x) { return x < 0. ? -x : x; }
ARB_HD float arb_sqrt(float x) { return sqrtf(x); }
ARB_HD double arb_sqrt(double x) { return sqrt(x); }
ARB_HD void arb_sincos(float a, float *s, float *c) { *s = sinf(a); *c = cosf(a); }
// A float64 constant materialised by two v_mov_b32 exactly where it is used (see arb_sincos): the
// volatile asm cannot be hoisted out of the step loop.
template <unsigned LO, unsigned HI>
ARB_HD double arb_pinned_bits() {
#if defined(__HIP_DEVICE_COMPILE__)
    int lo, hi;
#ifndef ARB_PINNED_VGPR      /* round 5: in SCALAR registers (s_mov_b32: the scalar unit has room, the vector pipes do not -- 144 v_mov per step
                                in the sin / cos of phase A); -DARB_PINNED_VGPR=1: the round-2 form */
    asm volatile("s_mov_b32 %0, %1" : "=s"(lo) : "n"(LO));
    asm volatile("s_mov_b32 %0, %1" : "=s"(hi) : "n"(HI));
#else
    asm volatile("v_mov_b32 %0, %1" : "=v"(lo) : "n"(LO));
    asm volatile("v_mov_b32 %0, %1" : "=v"(hi) : "n"(HI));
#endif
    return __hiloint2double(hi, lo);
#else
    const unsigned long long b = ((unsigned long long)HI << 32) | LO;
    double x;
    memcpy(&x, &b, sizeof(x));
    return x;
#endif
}
#define arb_pinned_const(x)                                                                           \
    arb_pinned_bits<(unsigned)(__builtin_bit_cast(unsigned long long, (double)(x)) & 0xffffffffull),    \
                    (unsigned)(__builtin_bit_cast(unsigned long long, (double)(x)) >> 32)>()
// float64 sin/cos for joint angles.  On the device this is a branch-free Cody-Waite
// reduction by pi/2 (two constants, exact to ~1e-17 * |a| for |a| < 1e5 rad) followed by
// the classic minimax kernels on [-pi/4, pi/4] (fdlibm __kernel_sin/__kernel_cos
// coefficients): ~45 flops instead of the library routine's large-argument machinery,
// which cost a third of phase A and most of the kernel's SGPR spills.
ARB_HD void arb_sincos(double a, double *s, double *c) {
#if defined(__HIP_DEVICE_COMPILE__)
    const double k = rint(a * 6.36619772367581382433e-01);
    double r = fma(-k, 1.57079632673412561417e+00, a);
    r = fma(-k, 6.07710050650619224932e-11, r);
    r = fma(-k, 2.02226624879595063154e-21, r);
    const double z = r * r;
    // The coefficients are materialised where they are used (K): left to itself the compiler hoists the
    // 64-bit literals out of the step loop into registers and then spills them to scratch memory.
#define K(x) arb_pinned_const(x)
    const double ps = K(-1.66666666666666324348e-01) + z * (K(8.33333333332248946124e-03) + z * (K(-1.98412698298579493134e-04)
                    + z * (K(2.75573137070700676789e-06) + z * (K(-2.50507602534068634195e-08) + z * K(1.58969099521155010221e-10)))));
    const double pc = K(4.16666666666666019037e-02) + z * (K(-1.38888888888741095749e-03) + z * (K(2.48015872894767294178e-05)
                    + z * (K(-2.75573143513906633035e-07) + z * (K(2.08757232129817482790e-09) + z * K(-1.13596475577881948265e-11)))));
#undef K
    const double sr = fma(r * z, ps, r);
    const double cr = fma(z * z, pc, fma(-0.5, z, 1.0));
    const int q = (int)k & 3;
    const double ss = (q & 1) ? cr : sr;
    const double cc = (q & 1) ? sr : cr;
    *s = (q & 2) ? -ss : ss;
    *c = ((q + 1) & 2) ? -cc : cc;
#else
    *s = sin(a); *c = cos(a);
#endif
}

template <typename T> struct JointLocal {
    M3<T> R;          // rotation of H_rn
    V3<T> p;          // translation of H_rn
    V3<T> jw[3];      // angular part of Jacobian columns (rotational joints)
    V3<T> djw[3];     // angular part of dJacobian columns
    V3<T> Tw, Tv;     // T_nr = J_nr * gvel   (core.py:197-201; gvel itself for FreeJoint)
};

// q, dq point at the joint's own slice of the state.
// pre_s / pre_c (optional): sin and cos of q[0], q[1], q[2], formed by the caller OUTSIDE the joint-type switch -- on the
// device every case of the switch runs with the few lanes of that joint type enabled, and a wavefront with 8 or fewer
// lanes enabled issues its vector instructions 3-4 times slower (tools/exec_mask_probe.hip); the angles' sin/cos are most
// of the switch.  Same arb_sincos on the same arguments: the results do not change.
template <typename T, typename QP>
ARB_HD void joint_local(int jt, QP q, QP dq, JointLocal<T> &o, const T *pre_s = nullptr, const T *pre_c = nullptr) {
    const T Z = T(0), O = T(1);
    auto arb_sincos = [&](T a, T *sn, T *cs, int i) {
        if (pre_s != nullptr) { *sn = pre_s[i]; *cs = pre_c[i]; }
        else ::arb_sincos(a, sn, cs);
    };
    o.R = m3_identity<T>();
    o.p = v3<T>(Z, Z, Z);
    for (int i = 0; i < 3; ++i) { o.jw[i] = v3<T>(Z, Z, Z); o.djw[i] = v3<T>(Z, Z, Z); }
    o.Tw = v3<T>(Z, Z, Z); o.Tv = v3<T>(Z, Z, Z);
    switch (jt) {
    case JT_FREE: {                                   // joints.py:10-57
        o.R.a[0] = q[0]; o.R.a[1] = q[1]; o.R.a[2] = q[2]; o.p.x = q[3];
        o.R.a[3] = q[4]; o.R.a[4] = q[5]; o.R.a[5] = q[6]; o.p.y = q[7];
        o.R.a[6] = q[8]; o.R.a[7] = q[9]; o.R.a[8] = q[10]; o.p.z = q[11];
        o.Tw = v3<T>(T(dq[0]), T(dq[1]), T(dq[2]));
        o.Tv = v3<T>(T(dq[3]), T(dq[4]), T(dq[5]));
    } break;
    case JT_RZRYRX: {                                 // joints.py:59-104, rotzyx :34-59
        T sz, cz, sy, cy, sx, cx;
        arb_sincos(T(q[0]), &sz, &cz, 0); arb_sincos(T(q[1]), &sy, &cy, 1); arb_sincos(T(q[2]), &sx, &cx, 2);
        o.R.a[0] = cz * cy; o.R.a[1] = cz * sy * sx - sz * cx; o.R.a[2] = cz * sy * cx + sz * sx;
        o.R.a[3] = sz * cy; o.R.a[4] = sz * sy * sx + cz * cx; o.R.a[5] = sz * sy * cx - cz * sx;
        o.R.a[6] = -sy;     o.R.a[7] = cy * sx;                o.R.a[8] = cy * cx;
        o.jw[0] = v3<T>(-sy, sx * cy, cx * cy);
        o.jw[1] = v3<T>(Z, cx, -sx);
        o.jw[2] = v3<T>(O, Z, Z);
        T dy = dq[1], dx = dq[2];
        o.djw[0] = v3<T>(-dy * cy, dx * cx * cy - dy * sx * sy, -dx * sx * cy - dy * cx * sy);
        o.djw[1] = v3<T>(Z, -dx * sx, -dx * cx);
        o.Tw = T(dq[0]) * o.jw[0] + T(dq[1]) * o.jw[1] + T(dq[2]) * o.jw[2];
    } break;
    case JT_RZRY: {                                   // joints.py:107-146, rotzy :60-80
        T sz, cz, sy, cy;
        arb_sincos(T(q[0]), &sz, &cz, 0); arb_sincos(T(q[1]), &sy, &cy, 1);
        o.R.a[0] = cz * cy; o.R.a[1] = -sz; o.R.a[2] = cz * sy;
        o.R.a[3] = sz * cy; o.R.a[4] = cz;  o.R.a[5] = sz * sy;
        o.R.a[6] = -sy;     o.R.a[7] = Z;   o.R.a[8] = cy;
        o.jw[0] = v3<T>(-sy, Z, cy);
        o.jw[1] = v3<T>(Z, O, Z);
        T dy = dq[1];
        o.djw[0] = v3<T>(-dy * cy, Z, -dy * sy);
        o.Tw = T(dq[0]) * o.jw[0] + T(dq[1]) * o.jw[1];
    } break;
    case JT_RZRX: {                                   // joints.py:149-185, rotzx :82-102
        T sz, cz, sx, cx;
        arb_sincos(T(q[0]), &sz, &cz, 0); arb_sincos(T(q[1]), &sx, &cx, 1);
        o.R.a[0] = cz; o.R.a[1] = -sz * cx; o.R.a[2] = sz * sx;
        o.R.a[3] = sz; o.R.a[4] = cz * cx;  o.R.a[5] = -cz * sx;
        o.R.a[6] = Z;  o.R.a[7] = sx;       o.R.a[8] = cx;
        o.jw[0] = v3<T>(Z, sx, cx);
        o.jw[1] = v3<T>(O, Z, Z);
        T dx = dq[1];
        o.djw[0] = v3<T>(Z, dx * cx, -dx * sx);
        o.Tw = T(dq[0]) * o.jw[0] + T(dq[1]) * o.jw[1];
    } break;
    case JT_RYRX: {                                   // joints.py:188-224, rotyx :104-124
        T sy, cy, sx, cx;
        arb_sincos(T(q[0]), &sy, &cy, 0); arb_sincos(T(q[1]), &sx, &cx, 1);
        o.R.a[0] = cy;  o.R.a[1] = sy * sx; o.R.a[2] = sy * cx;
        o.R.a[3] = Z;   o.R.a[4] = cx;      o.R.a[5] = -sx;
        o.R.a[6] = -sy; o.R.a[7] = cy * sx; o.R.a[8] = cy * cx;
        o.jw[0] = v3<T>(Z, cx, -sx);
        o.jw[1] = v3<T>(O, Z, Z);
        T dx = dq[1];
        o.djw[0] = v3<T>(Z, -dx * sx, -dx * cx);
        o.Tw = T(dq[0]) * o.jw[0] + T(dq[1]) * o.jw[1];
    } break;
    case JT_RZ: {                                     // joints.py:227-303
        T s, c; arb_sincos(T(q[0]), &s, &c, 0);
        o.R.a[0] = c; o.R.a[1] = -s; o.R.a[3] = s; o.R.a[4] = c;
        o.jw[0] = v3<T>(Z, Z, O);
        o.Tw = T(dq[0]) * o.jw[0];
    } break;
    case JT_RY: {                                     // joints.py:305-326
        T s, c; arb_sincos(T(q[0]), &s, &c, 0);
        o.R.a[0] = c; o.R.a[2] = s; o.R.a[6] = -s; o.R.a[8] = c;
        o.jw[0] = v3<T>(Z, O, Z);
        o.Tw = T(dq[0]) * o.jw[0];
    } break;
    case JT_RX: {                                     // joints.py:328-349
        T s, c; arb_sincos(T(q[0]), &s, &c, 0);
        o.R.a[4] = c; o.R.a[5] = -s; o.R.a[7] = s; o.R.a[8] = c;
        o.jw[0] = v3<T>(O, Z, Z);
        o.Tw = T(dq[0]) * o.jw[0];
    } break;
    case JT_TXTYTZ: {                                 // joints.py:352-384
        o.p = v3<T>(T(q[0]), T(q[1]), T(q[2]));
        o.Tv = v3<T>(T(dq[0]), T(dq[1]), T(dq[2]));
    } break;
    default: break;
    }
}

// SE(3) exponential, arboris/twistvector.py:35-70 (series below |w| = 1e-3).
template <typename T>
ARB_HD void exp_twist(V3<T> w, V3<T> v, M3<T> &R, V3<T> &p) {
    T t = arb_sqrt(dot(w, w));
    T cc, sc, dsc;
    if (t >= T(0.001)) {
        T s, c; arb_sincos(t, &s, &c);
        cc = (T(1) - c) / (t * t);
        sc = s / t;
        dsc = (t - s) / (t * t * t);
    } else {
        cc = T(0.5);
        sc = T(1) - t * t / T(6);
        dsc = T(1) / T(6);
    }
    M3<T> wx = hat(w);
    M3<T> wx2 = mul(wx, wx);
    R = m3_identity<T>();
    for (int i = 0; i < 9; ++i) R.a[i] += sc * wx.a[i] + cc * wx2.a[i];
    // p = (sc I + cc w^ + dsc w w^T) v
    V3<T> wv = cross(w, v);
    T wdv = dot(w, v);
    p = sc * v + cc * wv + (dsc * wdv) * w;
}

// Reciprocal to ~1 ulp without the IEEE division sequence (v_rcp + Newton steps on the
// device; a pivot of 0 still gives inf/NaN like a division).
ARB_HD float arb_rcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float r = __builtin_amdgcn_rcpf(x);
    return fmaf(r, fmaf(-x, r, 1.f), r);
#else
    return 1.f / x;
#endif
}
ARB_HD double arb_rcp(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(x);
    r = fma(r, fma(-x, r, 1.), r);
    return fma(r, fma(-x, r, 1.), r);
#else
    return 1. / x;
#endif
}

// Growth of one pivot of the pivot-free elimination (phase C, ARB_WARN_ILLCOND) from the BIT PATTERNS of the float32
// diagonal entry Z_jj as assembled (zb) and of the pivot that is left (pb): the difference of the patterns is
// 2^23 log2(Z_jj / pivot) to 6 %.  Both are positive integers for a healthy system; a pivot <= 0 (sign bit set -- -0.0
// included -- or +0.0), a NaN on either side or an infinite pivot saturates: the float32 elimination has gone
// indefinite and the warning must come whatever Z_jj is.  A negative diagonal counts by its magnitude.
// Scalar-unit arithmetic on the device; compiled for the host behind arb_host_growth_bits (tests/test_capi_cpu.py).
ARB_HD int arb_growth_bits(int zb, int pb) {
    if (pb <= 0 || pb >= 0x7f800000) return 0x7fffffff;
    zb &= 0x7fffffff;
    if (zb > 0x7f800000) return 0x7fffffff;
    return zb - pb;
}

// ---------------------------------------------------------------------------
// Small dense solvers (wave-uniform use in the Gauss-Seidel stage)
// ---------------------------------------------------------------------------
// Gaussian elimination with partial pivoting on a 4x4 system with NR right-hand
// sides, fully unrolled (static indices only, so everything stays in registers).
// Stands in for numpy.linalg.solve (constraints.py:834) and, for the
// non-singular blocks met in practice, numpy.linalg.pinv (constraints.py:795).
template <typename T, int NR>
ARB_HD void gepp4(T A[4][4], T B[4][NR]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        // pick the pivot row among c..3 and bubble it to row c
#pragma unroll
        for (int r = c + 1; r < 4; ++r) {
            bool sw = arb_abs(A[r][c]) > arb_abs(A[c][c]);
#pragma unroll
            for (int j = 0; j < 4; ++j) { T a = A[c][j], b = A[r][j]; A[c][j] = sw ? b : a; A[r][j] = sw ? a : b; }
#pragma unroll
            for (int j = 0; j < NR; ++j) { T a = B[c][j], b = B[r][j]; B[c][j] = sw ? b : a; B[r][j] = sw ? a : b; }
        }
        const T ip = arb_rcp(A[c][c]);
        A[c][c] = ip;                                  // the diagonal now holds the pivots' reciprocals
#pragma unroll
        for (int r = c + 1; r < 4; ++r) {
            T f = A[r][c] * ip;
#pragma unroll
            for (int j = c + 1; j < 4; ++j) A[r][j] -= f * A[c][j];
#pragma unroll
            for (int j = 0; j < NR; ++j) B[r][j] -= f * B[c][j];
        }
    }
#pragma unroll
    for (int c = 3; c >= 0; --c) {
        const T ip = A[c][c];
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            T s = B[c][j];
#pragma unroll
            for (int k = c + 1; k < 4; ++k) s -= A[c][k] * B[k][j];
            B[c][j] = s * ip;
        }
    }
}

// Conditioning thresholds of the constraint blocks, per arithmetic type.  numpy.linalg.pinv (constraints.py:79, 83,
// 235, 795) zeroes the singular values below rcond * s_max with rcond = 1e-15: a float64 block gets exactly that
// rule.  A float32 block of a rank-deficient system carries ~1e-7 of rounding noise where the float64 reference
// has an exact zero, so its cut is placed above that noise.
template <typename T> ARB_HD double pinv_rcond() { return sizeof(T) == 4 ? 2e-5 : 1e-15; }
template <typename T> ARB_HD double pinv_guard() { return sizeof(T) == 4 ? 1e-4 : 1e-11; }

// Inverse of a ND x ND block (ND <= 4) embedded in a 4x4 identity, by pivoted elimination.  Returns false when the
// pivots say the block is (numerically) rank deficient -- smallest / largest pivot magnitude below pinv_guard --
// in which case the caller must use pinv_block: the elimination's result is then meaningless.
// (TH: the type whose rounding noise the block carries -- the thresholds' type; differs from T when float32 data is
// processed in float64 arithmetic)
template <typename T, typename TH = T>
ARB_HD bool inv_block(const T *Y, int ld, int nd, T P[16]) {
    T A[4][4], B[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            A[i][j] = (i < nd && j < nd) ? Y[i * ld + j] : (i == j ? T(1) : T(0));
            B[i][j] = (i == j) ? T(1) : T(0);
        }
    gepp4<T, 4>(A, B);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) P[4 * i + j] = B[i][j];
    // the diagonal of A holds the reciprocals of the pivots (rows >= nd: the padding's 1)
    T rmin = T(0), rmax = T(0);
    bool first = true, finite = true;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (i < nd) {
            const T r = arb_abs(A[i][i]);
            finite = finite && (r - r == T(0));            // a zero pivot leaves inf (host) or NaN (device rcp + Newton)
            rmin = first ? r : (r < rmin ? r : rmin);
            rmax = first ? r : (r > rmax ? r : rmax);
            first = false;
        }
    // 1/|pivot|: min over max of the pivots = rmin / rmax
    return finite && (double)rmin > pinv_guard<TH>() * (double)rmax;
}

// Moore-Penrose pseudo-inverse of a ND x ND block (ND <= 4), numpy.linalg.pinv semantics (singular values below
// rcond * s_max are dropped), by one-sided Jacobi rotations in float64: the columns of U = A are rotated until
// mutually orthogonal (A V = U), then A^+ = sum_j v_j u_j^T / |u_j|^2 over the kept columns.  The result is embedded
// in a 4x4 identity like inv_block's.  Rare path (rank-deficient admittance blocks: a planar arm's contact or
// closed loop): clarity over speed.
template <typename T, typename TH = T>
ARB_HD void pinv_block(const T *Y, int ld, int nd, T P[16]) {
    double U[4][4], V[4][4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            U[i][j] = (i < nd && j < nd) ? (double)Y[i * ld + j] : 0.;
            V[i][j] = (i == j) ? 1. : 0.;
        }
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.;
        for (int p = 0; p < 3; ++p)
            for (int q = p + 1; q < 4; ++q) {
                if (q >= nd) continue;
                double al = 0., be = 0., ga = 0.;
                for (int i = 0; i < 4; ++i) { al += U[i][p] * U[i][p]; be += U[i][q] * U[i][q]; ga += U[i][p] * U[i][q]; }
                const double lim = 1e-15 * sqrt(al * be);
                if (!(fabs(ga) > lim) || ga == 0.) continue;
                off = fmax(off, fabs(ga) / fmax(sqrt(al * be), 1e-300));
                const double zeta = (be - al) / (2. * ga);
                const double t = (zeta >= 0. ? 1. : -1.) / (fabs(zeta) + sqrt(1. + zeta * zeta));
                const double c = 1. / sqrt(1. + t * t), sn = c * t;
                for (int i = 0; i < 4; ++i) {
                    const double up = U[i][p], uq = U[i][q];
                    U[i][p] = c * up - sn * uq; U[i][q] = sn * up + c * uq;
                    const double vp = V[i][p], vq = V[i][q];
                    V[i][p] = c * vp - sn * vq; V[i][q] = sn * vp + c * vq;
                }
            }
        if (off < 1e-14) break;
    }
    double s2[4], s2max = 0.;
    for (int j = 0; j < 4; ++j) {
        s2[j] = 0.;
        for (int i = 0; i < 4; ++i) s2[j] += U[i][j] * U[i][j];
        if (j < nd) s2max = fmax(s2max, s2[j]);
    }
    const double cut = pinv_rcond<TH>() * pinv_rcond<TH>() * s2max;       // compare squared singular values
    for (int i = 0; i < 4; ++i)
        for (int k = 0; k < 4; ++k) {
            double acc = 0.;
            for (int j = 0; j < 4; ++j)
                if (j < nd && s2[j] > cut && s2[j] > 0.) acc += V[i][j] * U[k][j] / s2[j];
            P[4 * i + k] = (i < nd && k < nd) ? (T)acc : (i == k ? T(1) : T(0));
        }
}

// Eigenvalues of a real 6x6 matrix: scaling balance + elimination to Hessenberg
// form + Francis double-shift QR (the classic EISPACK balanc/elmhes/hqr
// sequence; LAPACK's dgeev, which numpy.linalg.eigvals calls at
// constraints.py:825, is the same family of algorithm).  `a` is a row-major 6x6
// work array addressed dynamically (it lives in LDS in the kernel).  Returns the
// number of eigenvalues found (6 unless the iteration failed to converge).
template <typename T, typename AP>
ARB_HD int eig6(AP a, T wr[6], T wi[6]) {
    const int n = 6;
#define E_(i, j) a[(i) * 6 + (j)]
    // Non-finite input (e.g. the reference's 2/a with a == 0, where numpy.linalg.eigvals
    // raises LinAlgError) has no eigenvalues to offer: report none instead of iterating on
    // infinities.  Every data-dependent loop below is also capped.
    for (int i = 0; i < 36; ++i) {
        const T x = a[i];
        if (!(x - x == T(0))) return 0;
    }
    // --- balance (powers of 2 only, exact) ---
    {
        const T RADIX = T(2), sqrdx = T(4);
        bool last = false;
        int guard = 0;
        while (!last && guard++ < 64) {
            last = true;
            for (int i = 0; i < n; ++i) {
                T r = T(0), c = T(0);
                for (int j = 0; j < n; ++j)
                    if (j != i) { c += arb_abs(E_(j, i)); r += arb_abs(E_(i, j)); }
                if (c != T(0) && r != T(0)) {
                    T g = r / RADIX, f = T(1), s = c + r;
                    for (int cap = 0; c < g && cap < 1100; ++cap) { f *= RADIX; c *= sqrdx; }
                    g = r * RADIX;
                    for (int cap = 0; c > g && cap < 1100; ++cap) { f /= RADIX; c /= sqrdx; }
                    if ((c + r) / f < T(0.95) * s) {
                        last = false;
                        g = T(1) / f;
                        for (int j = 0; j < n; ++j) E_(i, j) *= g;
                        for (int j = 0; j < n; ++j) E_(j, i) *= f;
                    }
                }
            }
        }
    }
    // --- reduction to upper Hessenberg form by stabilised elimination ---
    for (int m = 1; m < n - 1; ++m) {
        T x = T(0);
        int i = m;
        for (int j = m; j < n; ++j)
            if (arb_abs(E_(j, m - 1)) > arb_abs(x)) { x = E_(j, m - 1); i = j; }
        if constexpr (sizeof(T) == 4) {
        // (loops over a data-dependent range run over all six indices with a predicate, operands read up front: in the
        // kernel the matrix is in LDS and ONE lane works on it -- every read the next operation waits for costs an LDS
        // round trip, so reads that do not depend on one another must be issued together.  Same arithmetic, element by
        // element: bit-identical results (5000 random matrices on the host); a fallback solve costs ~180 k instead of
        // ~200 k cycles -- the rest is the QR iteration's own serial chain of divisions and deflation tests)
        if (i != m) {
            T ti[6], tm[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) { ti[j] = E_(i, j); tm[j] = E_(m, j); }
#pragma unroll
            for (int j = 0; j < 6; ++j) if (j >= m - 1) { E_(i, j) = tm[j]; E_(m, j) = ti[j]; }
#pragma unroll
            for (int j = 0; j < 6; ++j) { ti[j] = E_(j, i); tm[j] = E_(j, m); }
#pragma unroll
            for (int j = 0; j < 6; ++j) { E_(j, i) = tm[j]; E_(j, m) = ti[j]; }
        }
        if (x != T(0)) {
            for (i = m + 1; i < n; ++i) {
                T y = E_(i, m - 1);
                if (y != T(0)) {
                    y /= x;
                    E_(i, m - 1) = y;
                    T ri[6], rm[6];
#pragma unroll
                    for (int j = 0; j < 6; ++j) { ri[j] = E_(i, j); rm[j] = E_(m, j); }
#pragma unroll
                    for (int j = 0; j < 6; ++j) if (j >= m) E_(i, j) = ri[j] - y * rm[j];
#pragma unroll
                    for (int j = 0; j < 6; ++j) { ri[j] = E_(j, m); rm[j] = E_(j, i); }
#pragma unroll
                    for (int j = 0; j < 6; ++j) E_(j, m) = ri[j] + y * rm[j];
                }
            }
        }
            } else {      // (float64: 36 more live registers than the kernels have)
        if (i != m) {
            for (int j = m - 1; j < n; ++j) { T t = E_(i, j); E_(i, j) = E_(m, j); E_(m, j) = t; }
            for (int j = 0; j < n; ++j) { T t = E_(j, i); E_(j, i) = E_(j, m); E_(j, m) = t; }
        }
        if (x != T(0)) {
            for (i = m + 1; i < n; ++i) {
                T y = E_(i, m - 1);
                if (y != T(0)) {
                    y /= x;
                    E_(i, m - 1) = y;
                    for (int j = m; j < n; ++j) E_(i, j) -= y * E_(m, j);
                    for (int j = 0; j < n; ++j) E_(j, m) += y * E_(j, i);
                }
            }
        }
            }
}
    for (int i = 2; i < n; ++i)
        for (int j = 0; j < i - 1; ++j) E_(i, j) = T(0);
    // --- shifted QR on the Hessenberg matrix ---
    T anorm = T(0);
    for (int i = 0; i < n; ++i)
        for (int j = (i > 0 ? i - 1 : 0); j < n; ++j) anorm += arb_abs(E_(i, j));
    int nn = n - 1, found = 0;
    T t = T(0);
    T p = T(0), q = T(0), r = T(0);
    int total_its = 0;
    while (nn >= 0) {
        int its = 0, l;
        do {
            if (++total_its > 600) return found;        // hard cap on the whole iteration
            for (l = nn; l >= 1; --l) {
                T s = arb_abs(E_(l - 1, l - 1)) + arb_abs(E_(l, l));
                if (s == T(0)) s = anorm;
                if ((T)(arb_abs(E_(l, l - 1)) + s) == s) { E_(l, l - 1) = T(0); break; }
            }
            T x = E_(nn, nn);
            if (l == nn) {                              // one real root
                wr[nn] = x + t; wi[nn] = T(0); --nn; ++found;
            } else {
                T y = E_(nn - 1, nn - 1);
                T w = E_(nn, nn - 1) * E_(nn - 1, nn);
                if (l == nn - 1) {                      // a 2x2 block: two roots
                    p = T(0.5) * (y - x);
                    q = p * p + w;
                    T z = arb_sqrt(arb_abs(q));
                    x += t;
                    if (q >= T(0)) {
                        z = p + (p >= T(0) ? arb_abs(z) : -arb_abs(z));
                        wr[nn - 1] = wr[nn] = x + z;
                        if (z != T(0)) wr[nn] = x - w / z;
                        wi[nn - 1] = wi[nn] = T(0);
                    } else {
                        wr[nn - 1] = wr[nn] = x + p;
                        wi[nn] = z; wi[nn - 1] = -z;
                    }
                    nn -= 2; found += 2;
                } else {
                    if (its >= 60) return found;        // no convergence
                    if (its == 10 || its == 20) {       // exceptional shift
                        t += x;
                        for (int i = 0; i <= nn; ++i) E_(i, i) -= x;
                        T s = arb_abs(E_(nn, nn - 1)) + arb_abs(E_(nn - 1, nn - 2));
                        y = x = T(0.75) * s;
                        w = T(-0.4375) * s * s;
                    }
                    ++its;
                    int m;
                    T z;
                    for (m = nn - 2; m >= l; --m) {
                        z = E_(m, m);
                        r = x - z;
                        T s = y - z;
                        p = (r * s - w) / E_(m + 1, m) + E_(m, m + 1);
                        q = E_(m + 1, m + 1) - z - r - s;
                        r = E_(m + 2, m + 1);
                        s = arb_abs(p) + arb_abs(q) + arb_abs(r);
                        p /= s; q /= s; r /= s;
                        if (m == l) break;
                        T u = arb_abs(E_(m, m - 1)) * (arb_abs(q) + arb_abs(r));
                        T v = arb_abs(p) * (arb_abs(E_(m - 1, m - 1)) + arb_abs(z) + arb_abs(E_(m + 1, m + 1)));
                        if ((T)(u + v) == v) break;
                    }
                    for (int i = m + 2; i <= nn; ++i) {
                        E_(i, i - 2) = T(0);
                        if (i != m + 2) E_(i, i - 3) = T(0);
                    }
                    for (int k = m; k <= nn - 1; ++k) {
                        if (k != m) {
                            p = E_(k, k - 1);
                            q = E_(k + 1, k - 1);
                            r = T(0);
                            if (k != nn - 1) r = E_(k + 2, k - 1);
                            x = arb_abs(p) + arb_abs(q) + arb_abs(r);
                            if (x != T(0)) { p /= x; q /= x; r /= x; }
                        }
                        T s = arb_sqrt(p * p + q * q + r * r);
                        if (p < T(0)) s = -s;
                        if (s != T(0)) {
                            if (k == m) {
                                if (l != m) E_(k, k - 1) = -E_(k, k - 1);
                            } else {
                                E_(k, k - 1) = -s * x;
                            }
                            p += s;
                            x = p / s; y = q / s; z = r / s;
                            q /= p; r /= p;
                            if constexpr (sizeof(T) == 4) {
                            const bool three = k != nn - 1;
                            const int k2 = three ? k + 2 : k + 1;            // (row / column k + 2 only exists then)
                            {
                                T e0[6], e1[6], e2[6];
#pragma unroll
                                for (int j = 0; j < 6; ++j) { e0[j] = E_(k, j); e1[j] = E_(k + 1, j); e2[j] = E_(k2, j); }
#pragma unroll
                                for (int j = 0; j < 6; ++j) {
                                    if (j >= k && j <= nn) {
                                        T pp = e0[j] + q * e1[j];
                                        if (three) { pp += r * e2[j]; E_(k2, j) = e2[j] - pp * z; }
                                        E_(k + 1, j) = e1[j] - pp * y;
                                        E_(k, j) = e0[j] - pp * x;
                                    }
                                }
                            }
                            const int mmin = nn < k + 3 ? nn : k + 3;
                            {
                                T c0[6], c1[6], c2[6];
#pragma unroll
                                for (int i = 0; i < 6; ++i) { c0[i] = E_(i, k); c1[i] = E_(i, k + 1); c2[i] = E_(i, k2); }
#pragma unroll
                                for (int i = 0; i < 6; ++i) {
                                    if (i >= l && i <= mmin) {
                                        T pp = x * c0[i] + y * c1[i];
                                        if (three) { pp += z * c2[i]; E_(i, k2) = c2[i] - pp * r; }
                                        E_(i, k + 1) = c1[i] - pp * q;
                                        E_(i, k) = c0[i] - pp;
                                    }
                                }
                            }
                            } else {
                            for (int j = k; j <= nn; ++j) {
                                p = E_(k, j) + q * E_(k + 1, j);
                                if (k != nn - 1) { p += r * E_(k + 2, j); E_(k + 2, j) -= p * z; }
                                E_(k + 1, j) -= p * y;
                                E_(k, j) -= p * x;
                            }
                            int mmin = nn < k + 3 ? nn : k + 3;
                            for (int i = l; i <= mmin; ++i) {
                                p = x * E_(i, k) + y * E_(i, k + 1);
                                if (k != nn - 1) { p += z * E_(i, k + 2); E_(i, k + 2) -= p * r; }
                                E_(i, k + 1) -= p * q;
                                E_(i, k) -= p;
                            }
                            }
                        }
                    }
                }
            }
        } while (l < nn - 1);
    }
#undef E_
    return found;
}

#if defined(__HIP_DEVICE_COMPILE__)
// ---------------------------------------------------------------------------
// eig6 by a whole WAVEFRONT (round 4).  The generic route is rare (9-20 of 270 k sliding solves of a step) but one lane
// working on a matrix in LDS pays an LDS round trip for every operand the next operation waits for: ~180 k cycles per
// matrix, and the world that needs it holds up a one-step launch (and is the critical path of a short episode).  Here
// lane 6 i + j holds E(i, j) in a register; the scalars of the iteration (shifts, the reflector) are formed by every
// lane alike from v_readlane broadcasts, its decisions go through the scalar unit (v_readfirstlane), and the row /
// column operations run on the lanes that own the elements, their operands gathered with ds_bpermute (the LDS
// crossbar without a memory access).  Element by element and scalar by scalar the operations -- and the expressions
// they are written as -- are those of eig6 above: bit-identical eigenvalues (tests/test_gpu_device_solve.py compares
// the two on the device).  All 64 lanes must be active.
// ---------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T wv_lane(T v, int src /* wave-uniform */) {
    if constexpr (sizeof(T) == 4) {
        return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src));
    } else {
        const long long b = __double_as_longlong(v);
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(b & 0xffffffffll), src);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), src);
        return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    }
}
template <typename T>
__device__ __forceinline__ T wv_gather(T v, int src /* per lane */) {
    if constexpr (sizeof(T) == 4) {
        return __int_as_float(__builtin_amdgcn_ds_bpermute(src << 2, __float_as_int(v)));
    } else {
        const long long b = __double_as_longlong(v);
        const unsigned lo = (unsigned)__builtin_amdgcn_ds_bpermute(src << 2, (int)(b & 0xffffffffll));
        const unsigned hi = (unsigned)__builtin_amdgcn_ds_bpermute(src << 2, (int)(b >> 32));
        return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    }
}

// `a`: this lane's element (lanes 36..63: anything).  Eigenvalues in wr / wi (every lane the same values).
template <typename T>
__device__ __forceinline__ int eig6_wave(T a, int lane, T wr[6], T wi[6]) {
#define EL(i, j) wv_lane<T>(a, 6 * (i) + (j))
#define UNI(c) (__builtin_amdgcn_readfirstlane((int)(c)) != 0)
#define WSTORE(idx, re, im)                                                                         \
    do {                                                                                            \
        _Pragma("unroll") for (int e_ = 0; e_ < 6; ++e_)                                            \
            if (e_ == (idx)) { wr[e_] = (re); wi[e_] = (im); }                                      \
    } while (0)
    const bool in = lane < 36;
    const int ri = in ? lane / 6 : 8, cj = in ? lane - 6 * (lane / 6) : 8;      // (8: matches no row / column)
    if (__builtin_amdgcn_ballot_w64(in && !(a - a == T(0))) != 0ull) return 0;
    // --- balance ---
    {
        const T RADIX = T(2), sqrdx = T(4);
        bool last = false;
        int guard = 0;
        while (!last && guard++ < 64) {
            last = true;
            for (int i = 0; i < 6; ++i) {
                T r = T(0), c = T(0);
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const T ec = arb_abs(EL(j, i)), er = arb_abs(EL(i, j));
                    if (j != i) { c += ec; r += er; }
                }
                if (UNI(c != T(0) && r != T(0))) {
                    T g = r / RADIX, f = T(1), s = c + r;
                    for (int cap = 0; UNI(c < g) && cap < 1100; ++cap) { f *= RADIX; c *= sqrdx; }
                    g = r * RADIX;
                    for (int cap = 0; UNI(c > g) && cap < 1100; ++cap) { f /= RADIX; c /= sqrdx; }
                    if (UNI((c + r) / f < T(0.95) * s)) {
                        last = false;
                        g = T(1) / f;
                        if (ri == i) a *= g;
                        if (cj == i) a *= f;
                    }
                }
            }
        }
    }
    // --- Hessenberg form ---
    for (int m = 1; m < 5; ++m) {
        T x = T(0);
        int iv = m;
#pragma unroll
        for (int j = 1; j < 6; ++j) {
            const T e = EL(j < m ? m : j, m - 1);
            if (j >= m && arb_abs(e) > arb_abs(x)) { x = e; iv = j; }
        }
        const int i = __builtin_amdgcn_readfirstlane(iv);
        if (i != m) {
            {
                const int sr = (ri == i) ? m : (ri == m) ? i : ri;
                const T g = wv_gather<T>(a, sr * 6 + cj);
                if ((ri == i || ri == m) && cj >= m - 1 && cj < 6) a = g;
            }
            {
                const int sc = (cj == i) ? m : (cj == m) ? i : cj;
                const T g = wv_gather<T>(a, ri * 6 + sc);
                if (in && (cj == i || cj == m)) a = g;
            }
        }
        if (UNI(x != T(0))) {
            for (int i2 = m + 1; i2 < 6; ++i2) {
                T y = EL(i2, m - 1);
                if (UNI(y != T(0))) {
                    y /= x;
                    if (lane == 6 * i2 + m - 1) a = y;
                    const T em = wv_gather<T>(a, 6 * m + cj);
                    if (ri == i2 && cj >= m && cj < 6) a = a - y * em;
                    const T ei = wv_gather<T>(a, 6 * ri + i2);
                    if (cj == m && ri < 6) a = a + y * ei;
                }
            }
        }
    }
    if (in && ri >= 2 && cj < ri - 1) a = T(0);
    // --- shifted QR ---
    T anorm = T(0);
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = (i > 0 ? i - 1 : 0); j < 6; ++j) anorm += arb_abs(EL(i, j));
    int nn = 5, found = 0;
    T t = T(0);
    T p = T(0), q = T(0), r = T(0);
    int total_its = 0;
    while (nn >= 0) {
        int its = 0, l;
        do {
            if (++total_its > 600) return found;
            for (l = nn; l >= 1; --l) {
                T s = arb_abs(EL(l - 1, l - 1)) + arb_abs(EL(l, l));
                if (s == T(0)) s = anorm;
                if (UNI((T)(arb_abs(EL(l, l - 1)) + s) == s)) { if (lane == 6 * l + l - 1) a = T(0); break; }
            }
            T x = EL(nn, nn);
            if (l == nn) {
                WSTORE(nn, x + t, T(0)); --nn; ++found;
            } else {
                T y = EL(nn - 1, nn - 1);
                T w = EL(nn, nn - 1) * EL(nn - 1, nn);
                if (l == nn - 1) {
                    p = T(0.5) * (y - x);
                    q = p * p + w;
                    T z = arb_sqrt(arb_abs(q));
                    x += t;
                    if (UNI(q >= T(0))) {
                        z = p + (p >= T(0) ? arb_abs(z) : -arb_abs(z));
                        T w0 = x + z, w1 = x + z;
                        if (UNI(z != T(0))) w1 = x - w / z;
                        WSTORE(nn - 1, w0, T(0)); WSTORE(nn, w1, T(0));
                    } else {
                        WSTORE(nn - 1, x + p, -z); WSTORE(nn, x + p, z);
                    }
                    nn -= 2; found += 2;
                } else {
                    if (its >= 60) return found;
                    if (its == 10 || its == 20) {
                        t += x;
                        if (in && ri == cj && ri <= nn) a -= x;
                        T s = arb_abs(EL(nn, nn - 1)) + arb_abs(EL(nn - 1, nn - 2));
                        y = x = T(0.75) * s;
                        w = T(-0.4375) * s * s;
                    }
                    ++its;
                    int m;
                    T z;
                    for (m = nn - 2; m >= l; --m) {
                        z = EL(m, m);
                        r = x - z;
                        T s = y - z;
                        p = (r * s - w) / EL(m + 1, m) + EL(m, m + 1);
                        q = EL(m + 1, m + 1) - z - r - s;
                        r = EL(m + 2, m + 1);
                        s = arb_abs(p) + arb_abs(q) + arb_abs(r);
                        p /= s; q /= s; r /= s;
                        if (m == l) break;
                        T u = arb_abs(EL(m, m - 1)) * (arb_abs(q) + arb_abs(r));
                        T v = arb_abs(p) * (arb_abs(EL(m - 1, m - 1)) + arb_abs(z) + arb_abs(EL(m + 1, m + 1)));
                        if (UNI((T)(u + v) == v)) break;
                    }
                    if (in && ri >= m + 2 && ri <= nn && (cj == ri - 2 || (ri != m + 2 && cj == ri - 3))) a = T(0);
                    for (int k = m; k <= nn - 1; ++k) {
                        if (k != m) {
                            p = EL(k, k - 1);
                            q = EL(k + 1, k - 1);
                            r = T(0);
                            if (k != nn - 1) r = EL(k + 2, k - 1);
                            x = arb_abs(p) + arb_abs(q) + arb_abs(r);
                            if (UNI(x != T(0))) { p /= x; q /= x; r /= x; }
                        }
                        T s = arb_sqrt(p * p + q * q + r * r);
                        if (p < T(0)) s = -s;
                        if (UNI(s != T(0))) {
                            if (k == m) {
                                if (l != m && lane == 6 * k + k - 1) a = -a;
                            } else {
                                if (lane == 6 * k + k - 1) a = -s * x;
                            }
                            p += s;
                            x = p / s; y = q / s; z = r / s;
                            q /= p; r /= p;
                            const bool three = k != nn - 1;
                            const int k2 = three ? k + 2 : k + 1;
                            {
                                const T e0 = wv_gather<T>(a, 6 * k + cj), e1 = wv_gather<T>(a, 6 * (k + 1) + cj),
                                        e2 = wv_gather<T>(a, 6 * k2 + cj);
                                if (cj >= k && cj <= nn) {
                                    T pp = e0 + q * e1;
                                    if (three) pp += r * e2;
                                    if (three && ri == k2) a = e2 - pp * z;
                                    if (ri == k + 1) a = e1 - pp * y;
                                    if (ri == k) a = e0 - pp * x;
                                }
                            }
                            const int mmin = nn < k + 3 ? nn : k + 3;
                            {
                                const T c0 = wv_gather<T>(a, 6 * ri + k), c1 = wv_gather<T>(a, 6 * ri + k + 1),
                                        c2 = wv_gather<T>(a, 6 * ri + k2);
                                if (ri >= l && ri <= mmin) {
                                    T pp = x * c0 + y * c1;
                                    if (three) pp += z * c2;
                                    if (three && cj == k2) a = c2 - pp * r;
                                    if (cj == k + 1) a = c1 - pp * q;
                                    if (cj == k) a = c0 - pp;
                                }
                            }
                        }
                    }
                }
            }
        } while (l < nn - 1);
    }
#undef EL
#undef UNI
#undef WSTORE
    return found;
}

// slide_shift_from_eig (below) with the wavefront's eig6: `work` holds the 6x6 matrix (LDS, written and synchronised
// by the caller); every lane returns the shift.
template <typename T, typename AP>
__device__ __forceinline__ T slide_shift_from_eig_wave(AP work, int lane) {
    T wr[6] = {T(0), T(0), T(0), T(0), T(0), T(0)}, wi[6] = {T(0), T(0), T(0), T(0), T(0), T(0)};
    const T a = lane < 36 ? work[lane] : T(0);
    const int nf = eig6_wave<T>(a, lane, wr, wi);
    bool any = false;
    T smin = T(0);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        bool ok = (i >= 6 - nf) && (wi[i] == T(0)) && (wr[i] <= T(0));
        if (ok) { smin = any ? (wr[i] < smin ? wr[i] : smin) : wr[i]; any = true; }
    }
    return any ? (smin > T(-1e10) ? smin : T(-1e10)) : T(-1e10);   // constraints.py:827-830
}
#elif defined(__HIPCC__)
// (host pass of the kernel sources: declared, never called)
template <typename T> __device__ int eig6_wave(T a, int lane, T wr[6], T wi[6]);
template <typename T, typename AP> __device__ T slide_shift_from_eig_wave(AP work, int lane);
#endif

// ---------------------------------------------------------------------------
// Sliding-friction shift `s` of SoftFingerContact.solve (constraints.py:803-830):
// the smallest real non-positive eigenvalue of the 6x6 matrix
//     B = [[E (Yh + c1 1 1^T), -c2 E], [c3 E - I, E Yh]],   E = diag(eps^2)
// (the reference's 1-D dot() products make c0..c3 scalars).  For eps = (1,1,1)
// the lower-left block is a multiple of I and commutes with everything, so
//     det(B - s I) = det((P - s I)(Q - s I) + c2 (c3 - 1) I),  P = Q + c1 1 1^T,
// a sextic in s whose coefficients come from a 3x3 determinant of quadratics.
// Its leftmost real root is found in float64 registers by Laguerre iterations
// started left of the spectrum (monotone and cubically convergent when all
// roots are real, which is the case met in practice); any anomaly makes the
// caller fall back to the generic QR eigenvalue routine eig6().
// ---------------------------------------------------------------------------

ARB_HD double arb_fast_rcp(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcp(x);
#else
    return 1. / x;
#endif
}
ARB_HD double arb_fast_sqrt(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return x > 0. ? x * __builtin_amdgcn_rsq(x) : 0.;
#else
    return sqrt(x);
#endif
}

// Pieces of det(B - sI) that depend only on the constraint's admittance block (constant over
// the 20 sweeps of a step): with Q = Y_t - c0 11^T,
//   chi(z) = det(zI - Q) = z^3 - tr z^2 + m2 z - det,   rho(z) = 1^T adj(Q - zI) 1 = 3 z^2 - (3 tr - sQ) z + sA
// and, for kappa = c2 (c3 - 1) = -d2,
//   det(B - sI) = E_chi^2 - d2 O_chi^2 - c1 (E_rho E_chi - d2 O_rho O_chi)
// where E/O are the parts of chi(s +- d), rho(s +- d) even/odd in d (all real whatever the
// sign of d2):  E_chi = chi(s) + d2 (3s - tr),  O_chi = chi'(s) + d2,  E_rho = rho(s) + 3 d2,
// O_rho = rho'(s).  (Matrix determinant lemma on (Q - s + c1 11^T)(Q - s) + kappa I.)
#ifndef ARB_ROOT_CASCADE
#define ARB_ROOT_CASCADE 1      // the sliding solve's fallback decides by the derivative cascade (slide_real_root_cascade) before the 6x6 eigenvalue routine is asked
#endif
struct SlidePre { double tr, m2, det, sQ, sA, nq; };

template <typename T>
ARB_HD SlidePre slide_precompute(const T Y[16]) {
    const T ycyc = Y[3] * Y[3] + Y[7] * Y[7] + Y[11] * Y[11];        // dot(Y_c, Y_c.T): a scalar (constraints.py:813)
    const double c0 = (double)ycyc / (double)Y[15];
    double Q[3][3];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Q[i][j] = (double)Y[4 * i + j] - c0;
    SlidePre k;
    k.tr = Q[0][0] + Q[1][1] + Q[2][2];
    // adjugate (transposed cofactors); only its trace and the sum of its entries are needed
    const double A00 = Q[1][1] * Q[2][2] - Q[1][2] * Q[2][1], A01 = Q[0][2] * Q[2][1] - Q[0][1] * Q[2][2],
                 A02 = Q[0][1] * Q[1][2] - Q[0][2] * Q[1][1], A10 = Q[1][2] * Q[2][0] - Q[1][0] * Q[2][2],
                 A11 = Q[0][0] * Q[2][2] - Q[0][2] * Q[2][0], A12 = Q[0][2] * Q[1][0] - Q[0][0] * Q[1][2],
                 A20 = Q[1][0] * Q[2][1] - Q[1][1] * Q[2][0], A21 = Q[0][1] * Q[2][0] - Q[0][0] * Q[2][1],
                 A22 = Q[0][0] * Q[1][1] - Q[0][1] * Q[1][0];
    k.m2 = A00 + A11 + A22;
    k.det = Q[0][0] * A00 + Q[0][1] * A10 + Q[0][2] * A20;
    k.sQ = 0.; k.nq = 0.;
    for (int i = 0; i < 3; ++i) {
        double r = 0.;
        for (int j = 0; j < 3; ++j) { k.sQ += Q[i][j]; r += fabs(Q[i][j]); }
        k.nq = fmax(k.nq, r);
    }
    k.sA = A00 + A01 + A02 + A10 + A11 + A12 + A20 + A21 + A22;
    return k;
}

// The sextic's coefficients as polynomials in the two sweep-dependent scalars (round 4): with d = -kappa, expanding
//   det(B - sI) = E_chi^2 - d O_chi^2 - c1 (E_rho E_chi - d O_rho O_chi)      (see above)
// in powers of s gives  pc_i = A_i(d) - c1 B_i(d),  A_i of degree <= 3 and B_i of degree <= 2 in d with coefficients that
// depend on the admittance block only -- 21 numbers per constraint and step instead of 47 float64 operations per solve:
//   i   A_i                                                   B_i
//   0   det^2 + (2 det tr - m2^2) d + (tr^2 - 2 m2) d^2 - d^3  -sA det - (sA tr + 3 det + l1 m2) d - (3 tr + l1) d^2
//   1   -2 det m2 + (2 tr m2 - 6 det) d - 2 tr d^2             sA m2 - l1 det + (3 sA - 3 m2 + l1 tr) d + 3 d^2
//   2   2 det tr + m2^2 - 2 tr^2 d + 3 d^2                     -sA tr + l1 m2 - 3 det + 6 tr d
//   3   -2 (det + tr m2) + 4 tr d                              sA - l1 tr + 3 m2 - 6 d
//   4   2 m2 + tr^2 - 3 d                                      l1 - 3 tr
//   5   -2 tr                                                  3                       (l1 = sQ - 3 tr; pc_6 = 1)
// Every coefficient is evaluated by the SAME seven-constant Horner form (zeros on top), so that the fused kernel -- where
// the four lanes of a constraint's quad evaluate different coefficients at once and exchange them by DPP -- and the scalar
// callers below execute the same operations: bit-identical coefficients.
struct SlideCoef { double a0, a1, a2, a3, b0, b1, b2; };

ARB_HD double slide_coef_eval(const SlideCoef &c, double d, double c1) {
    const double A = ((c.a3 * d + c.a2) * d + c.a1) * d + c.a0;
    const double B = (c.b2 * d + c.b1) * d + c.b0;
    return A - c1 * B;
}

ARB_HD void slide_coefs_all(const SlidePre &k, SlideCoef c[6]) {
    const double tr = k.tr, m2 = k.m2, det = k.det, sA = k.sA;
    const double l1 = -(3. * tr - k.sQ);
    c[0].a0 = det * det;               c[0].a1 = 2. * det * tr - m2 * m2;   c[0].a2 = tr * tr - 2. * m2;  c[0].a3 = -1.;
    c[0].b0 = -(sA * det);             c[0].b1 = -(sA * tr + 3. * det + l1 * m2);   c[0].b2 = -(3. * tr + l1);
    c[1].a0 = -2. * det * m2;          c[1].a1 = 2. * tr * m2 - 6. * det;   c[1].a2 = -2. * tr;           c[1].a3 = 0.;
    c[1].b0 = sA * m2 - l1 * det;      c[1].b1 = 3. * sA - 3. * m2 + l1 * tr;       c[1].b2 = 3.;
    c[2].a0 = 2. * det * tr + m2 * m2; c[2].a1 = -2. * tr * tr;             c[2].a2 = 3.;                 c[2].a3 = 0.;
    c[2].b0 = -(sA * tr) + l1 * m2 - 3. * det;   c[2].b1 = 6. * tr;         c[2].b2 = 0.;
    c[3].a0 = -2. * (det + tr * m2);   c[3].a1 = 4. * tr;                   c[3].a2 = 0.;                 c[3].a3 = 0.;
    c[3].b0 = sA - l1 * tr + 3. * m2;  c[3].b1 = -6.;                       c[3].b2 = 0.;
    c[4].a0 = 2. * m2 + tr * tr;       c[4].a1 = -3.;                       c[4].a2 = 0.;                 c[4].a3 = 0.;
    c[4].b0 = l1 - 3. * tr;            c[4].b1 = 0.;                        c[4].b2 = 0.;
    c[5].a0 = -2. * tr;                c[5].a1 = 0.;                        c[5].a2 = 0.;                 c[5].a3 = 0.;
    c[5].b0 = 3.;                      c[5].b1 = 0.;                        c[5].b2 = 0.;
}

// the seven coefficients, constant term first
ARB_HD void slide_poly(const SlidePre &k, double c1, double kappa, double pc[7]) {
    SlideCoef c[6];
    slide_coefs_all(k, c);
    const double d2 = -kappa;
#pragma unroll
    for (int i = 0; i < 6; ++i) pc[i] = slide_coef_eval(c[i], d2, c1);
    pc[6] = 1.;
}

// Leftmost real root of det(B - sI).  `warm` (NaN = none) is the root found for the same
// constraint in the previous sweep: the iteration restarts just left of it when a
// Budan-Fourier certificate (all Taylor coefficients at the start point alternate in sign,
// hence no real root to its left) holds, otherwise from the spectrum bound.
// Returns true and the root when the register-only path succeeded.
// `step_tol`: a Laguerre step shorter than step_tol |x| ends the iteration (the steps shrink at
// least geometrically from the left, so the remaining distance is below the last step).
ARB_HD bool slide_leftmost_root(const SlidePre &k, double c1, double kappa, double warm, double *root,
                                double step_tol = 4e-16) {
    double pc[7];
    slide_poly(k, c1, kappa, pc);
    double x = NAN;
    if (warm == warm) {
        const double x0 = warm - 1e-3 * fabs(warm) - 1e-300;
        double t[7];
        for (int i = 0; i < 7; ++i) t[i] = pc[i];
        for (int j = 0; j < 6; ++j)                 // Taylor shift: t[i] = p^(i)(x0) / i!
            for (int i = 5; i >= j; --i) t[i] += x0 * t[i + 1];
        if (t[0] > 0. && t[1] < 0. && t[2] > 0. && t[3] < 0. && t[4] > 0. && t[5] < 0.) x = x0;
    }
    if (!(x == x)) {
        // |P| <= |Q| + 3 |c1| (infinity norms); every eigenvalue has |s| <= max(|P|,|Q|) + sqrt|kappa|
        const double rb = k.nq + 3. * fabs(c1) + arb_fast_sqrt(fabs(kappa));
        if (!(rb > 0.) || !(rb < 1e300)) return false;
        x = -1.0001 * rb - 1e-300;
    }
    const double n = 6.;
    for (int it = 0; it < 40; ++it) {
        double p0 = pc[6], p1 = 0., p2 = 0., ee = fabs(pc[6]);
        const double ax = fabs(x);
        for (int i = 5; i >= 0; --i) {
            p2 = p2 * x + p1; p1 = p1 * x + p0; p0 = p0 * x + pc[i];
            ee = ee * ax + fabs(p0);                    // running Horner error bound
        }
        p2 *= 2.;
        if (fabs(p0) <= 8.9e-16 * (2. * ee - fabs(p0))) { *root = x; return true; }   // p(x) = 0 to rounding
        if (!(p0 > 0.) || !(p1 < 0.)) return false;   // not left of all roots any more: anomaly
        // Laguerre step  dx = n p / (p' - sqrt((n-1)((n-1) p'^2 - n p p'')))  (p' < 0 here).
        // Only the step uses approximate sqrt / reciprocal (hardware v_rsq_f64 / v_rcp_f64,
        // ~1e-8 relative): the accuracy of the root is set by the float64 Horner values and
        // the stopping test above, not by the step.
        const double rad = (n - 1.) * ((n - 1.) * p1 * p1 - n * p0 * p2);
        if (!(rad >= 0.)) return false;               // complex roots nearby
        const double den = p1 - arb_fast_sqrt(rad);   // both terms negative: no cancellation
        // shortened by 2^-20 so that the ~1e-8 error of the approximate sqrt/rcp can never
        // carry the iterate past the root (the exact Laguerre step from the left never does)
        const double dx = (n * (1. - 9.5367431640625e-07)) * p0 * arb_fast_rcp(den); // negative
        const double xn = x - dx;
        if (!(xn > x)) { *root = x; return true; }    // no representable progress: converged
        if (fabs(dx) <= step_tol * fabs(xn)) { *root = xn; return true; }
        if (fabs(dx) <= 1e-2 * fabs(xn)) {
            // Short step: accept xn when the Newton estimate of what is left, p(xn) / |p'(x)|, is
            // below the tolerance (|p'| decreases towards the root, hence the factor 1/4).
            double q0 = pc[6];
            for (int i = 5; i >= 0; --i) q0 = q0 * xn + pc[i];
            if (fabs(q0) <= 0.25 * step_tol * fabs(xn) * (-p1)) { *root = xn; return true; }
        }
        x = xn;
    }
    return false;
}

// ---------------------------------------------------------------------------------------------------------------
// The leftmost real root of the sextic in [lo, 0] when the iteration above declines (a complex pair leftmost, or nearly
// so): the DERIVATIVE CASCADE.  The real roots of p^(j+1) cut [lo, 0] into pieces on which p^(j) is monotone; a piece
// holds a real root of p^(j) exactly when p^(j) changes sign over it, and a bracketed Newton iteration finds it.  From
// the linear p^(5) down to p itself, whose first root is the answer (constraints.py:826-830: the smallest real
// eigenvalue <= 0).  Float64 throughout: the decision "real or complex" is taken by the sign of p at a stationary
// point, i.e. to the rounding of a float64 Horner value -- the reference's float64 eigvals decides the same cases,
// the float32 QR sequence this route runs before (eig6) decided them in float32.  Round 4: ~100 k cycles per
// fallback for the QR sequence on the whole wavefront; this route is plain scalar arithmetic, the same for host and
// device.  Returns 1 with *root, 0 when p has no real root in [lo, 0], -1 when the input is not finite.
// ---------------------------------------------------------------------------------------------------------------
// (one instance of the level code and of the bracketed iteration serves the six levels -- the polynomial of a level is the
// degree-6 form with zeros on top --, and the coefficients and the two root lists live in `buf`, 19 doubles: in the kernels
// the LDS scratch array, every lane writing the same values; the route costs the sweeps' loop neither code nor registers)
ARB_HD void cascade_horner(const double c[7], double x, double &f, double &df) {
    f = c[6]; df = 0.;
#pragma unroll
    for (int i = 5; i >= 0; --i) { df = df * x + f; f = f * x + c[i]; }
}

// root of the polynomial c between a and b, where it is monotone and f(a) = fa, f(b) differ in sign
ARB_HD double cascade_solve(const double c[7], double a, double fa, double b) {
    double xl = (fa < 0.) ? a : b, xh = (fa < 0.) ? b : a;          // f(xl) < 0 < f(xh)
    double x = 0.5 * (a + b), dxold = fabs(b - a), dx = dxold, f, df;
    cascade_horner(c, x, f, df);
    for (int it = 0; it < 200; ++it) {
        if (f == 0.) return x;
        if (f < 0.) xl = x; else xh = x;
        const bool outside = ((x - xh) * df - f) * ((x - xl) * df - f) > 0.;
        const bool slow = fabs(2. * f) > fabs(dxold * df);
        dxold = dx;
        double xn;
        if (outside || slow || !(df != 0.)) { dx = 0.5 * (xh - xl); xn = xl + dx; }      // bisection
        else { dx = f / df; xn = x - dx; }                                                 // Newton
        if (xn == x) return x;
        x = xn;
        if (fabs(dx) <= 2.3e-16 * fabs(x)) return x;
        cascade_horner(c, x, f, df);
    }
    return x;
}

ARB_HD int slide_real_root_cascade(const double pc[7], double lo, double *root, double *buf /* [19] */) {
#pragma unroll
    for (int i = 0; i < 7; ++i) if (!(fabs(pc[i]) < 1e300)) return -1;
    if (!(lo < 0.) || !(lo > -1e300)) return -1;
    // C(i + J, J): p^(J)(x) / J! = sum_i C(i + J, J) pc[i + J] x^i
    const double BINOM[6][7] = {{1., 1., 1., 1., 1., 1., 1.}, {1., 2., 3., 4., 5., 6., 0.}, {1., 3., 6., 10., 15., 0., 0.},
                                {1., 4., 10., 20., 0., 0., 0.}, {1., 5., 15., 0., 0., 0., 0.}, {1., 6., 0., 0., 0., 0., 0.}};
    double *pcs = buf + 12;
#pragma unroll
    for (int i = 0; i < 7; ++i) pcs[i] = pc[i];
    int nprev = 0, off_prev = 0;                        // roots of the level above: buf[off_prev .. off_prev + nprev)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma nounroll
#endif
    for (int J = 5; J >= 0; --J) {
        double c[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) c[i] = (i + J <= 6) ? BINOM[J][i] * pcs[(i + J <= 6) ? i + J : 6] : 0.;
        const double *prev = buf + off_prev;
        double *cur = buf + (6 - off_prev);
        int n = 0;
        double last = lo - 1.;
        double a = lo, fa, fb, d_;
        cascade_horner(c, a, fa, d_);
        bool done = false;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma nounroll
#endif
        for (int k = 0; k <= nprev && !done; ++k) {
            const double b = (k < nprev) ? prev[k] : 0.;
            cascade_horner(c, b, fb, d_);
            double r = 0.;
            bool got = false;
            if (fa == 0.) { r = a; got = true; }
            else if (fb != 0. && ((fa < 0.) != (fb < 0.))) { r = cascade_solve(c, a, fa, b); got = true; }
            if (got && n < 6 && (n == 0 || r > last)) {
                cur[n] = r; last = r; ++n;
                done = (J == 0);                         // of p itself only the first root is asked for
            }
            a = b; fa = fb;
        }
        if (!done && fa == 0. && n < 6 && (n == 0 || 0. > last)) { cur[n] = 0.; ++n; }      // a root at 0 itself
        nprev = n; off_prev = 6 - off_prev;
    }
    if (nprev > 0) { *root = buf[off_prev]; return 1; }
    return 0;
}
ARB_HD int slide_real_root_cascade(const double pc[7], double lo, double *root) {
    double buf[19];
    return slide_real_root_cascade(pc, lo, root, buf);
}
// the cascade from the per-step constants of a contact (what the register-only iteration starts from)
ARB_HD int slide_real_root_cascade_from(const SlidePre &k, double c1, double kappa, double *root) {
    double pc[7];
    slide_poly(k, c1, kappa, pc);
    const double rb = k.nq + 3. * fabs(c1) + arb_fast_sqrt(fabs(kappa));     // (the spectrum bound of slide_leftmost_root)
    if (!(rb > 0.) || !(rb < 1e300)) return -1;
    return slide_real_root_cascade(pc, -1.0001 * rb - 1e-300, root);
}

// The same iteration for the fused kernel, where the four lanes of a constraint's quad carry the live
// problem in vector registers and the rest of the wavefront runs along on don't-care data: every branch
// condition goes through `uni`, which returns the quad's verdict for the whole wave.  (A separate copy
// rather than a template of the function above: the scalar callers keep their text unchanged.)
template <typename UNI>
ARB_HD bool slide_leftmost_root_uni(const SlidePre &k, double c1, double kappa, double warm, double *root,
                                    double step_tol, UNI uni, int *probe = nullptr, double woff = -1.) {
#define U(x) uni(x)
    double pc[7];
    slide_poly(k, c1, kappa, pc);
    // Every data-dependent branch of this variant costs a compare, a ballot and a scalar test on the critical
    // path of the sweeps: the exits of an iteration are evaluated arithmetically and tested once (measured:
    // +2.5 %; going further -- start point by selection, the short-step residual on every pass -- costs more
    // arithmetic than the branches it saves).
    double x = NAN;
    // value, slope and curvature at a certified warm start come with the certificate (the Taylor shift): the first
    // Laguerre step uses them instead of a Horner pass of its own
    double w0 = 0., w1 = 0., w2 = 0.;
    bool from_shift = false;
    if (U(warm == warm)) {
        // restart just left of the previous root: by `woff` when the caller knows how far the root moved last time
        const double x0 = warm - (woff >= 0. ? woff : 1e-3 * fabs(warm)) - 1e-300;
        double t[7];
        for (int i = 0; i < 7; ++i) t[i] = pc[i];
        for (int j = 0; j < 6; ++j)                 // Taylor shift: t[i] = p^(i)(x0) / i!
            for (int i = 5; i >= j; --i) t[i] += x0 * t[i + 1];
        if (U(t[0] > 0. && t[1] < 0. && t[2] > 0. && t[3] < 0. && t[4] > 0. && t[5] < 0.)) {
            x = x0; w0 = t[0]; w1 = t[1]; w2 = 2. * t[2]; from_shift = true;
        }
    }
    if (probe) probe[0] += from_shift ? 1 : 0;   // development: warm start certified
    // (`from_shift` is only ever assigned under wave-uniform branches: a scalar flag, tested without a ballot)
    if (!from_shift) {
        // |P| <= |Q| + 3 |c1| (infinity norms); every eigenvalue has |s| <= max(|P|,|Q|) + sqrt|kappa|
        const double rb = k.nq + 3. * fabs(c1) + arb_fast_sqrt(fabs(kappa));
        if (U(!(rb > 0.) || !(rb < 1e300))) return false;
        x = -1.0001 * rb - 1e-300;
    }
    const double n = 6.;
    for (int it = 0; it < 40; ++it) {
        if (probe) probe[1] += 1;                 // development: Laguerre iterations
        double p0, p1, p2;
        bool zero = false;
        if (from_shift) {
            p0 = w0; p1 = w1; p2 = w2;           // (p > 0 > p' certified: the rounding-level zero test is not needed here)
            from_shift = false;
        } else {
            double ee = fabs(pc[6]);
            p0 = pc[6]; p1 = 0.; p2 = 0.;
            const double ax = fabs(x);
            for (int i = 5; i >= 0; --i) {
                p2 = p2 * x + p1; p1 = p1 * x + p0; p0 = p0 * x + pc[i];
                ee = ee * ax + fabs(p0);                    // running Horner error bound
            }
            p2 *= 2.;
            zero = fabs(p0) <= 8.9e-16 * (2. * ee - fabs(p0));      // p(x) = 0 to rounding
        }
        // Laguerre step  dx = n p / (p' - sqrt((n-1)((n-1) p'^2 - n p p'')))  (p' < 0 here).
        // Only the step uses approximate sqrt / reciprocal (hardware v_rsq_f64 / v_rcp_f64,
        // ~1e-8 relative): the accuracy of the root is set by the float64 Horner values and
        // the stopping tests, not by the step.
        const double rad = (n - 1.) * ((n - 1.) * p1 * p1 - n * p0 * p2);
        // anomalies: not left of all roots any more, or complex roots nearby
        const bool bad = !zero && (!(p0 > 0.) || !(p1 < 0.) || !(rad >= 0.));
        const double den = p1 - arb_fast_sqrt(rad);   // both terms negative: no cancellation
        // shortened by 2^-20 so that the ~1e-8 error of the approximate sqrt/rcp can never
        // carry the iterate past the root (the exact Laguerre step from the left never does)
        const double dx = (n * (1. - 9.5367431640625e-07)) * p0 * arb_fast_rcp(den); // negative
        const double xn = x - dx;
        const bool stall = !(xn > x);                 // no representable progress: converged
        if (U(zero || bad || stall || fabs(dx) <= step_tol * fabs(xn))) {
            if (U(bad)) return false;
            *root = (zero || stall) ? x : xn;
            return true;
        }
        if (U(fabs(dx) <= 1e-2 * fabs(xn))) {
            // Short step: accept xn when the Newton estimate of what is left, p(xn) / |p'(x)|, is
            // below the tolerance (|p'| decreases towards the root, hence the factor 1/4).
            double q0 = pc[6];
            for (int i = 5; i >= 0; --i) q0 = q0 * xn + pc[i];
            if (U(fabs(q0) <= 0.25 * step_tol * fabs(xn) * (-p1))) { *root = xn; return true; }
        }
        x = xn;
    }
    return false;
#undef U
}

#if defined(__HIP_DEVICE_COMPILE__)
// The same again with the quad's verdicts formed from LANE MASKS (round 4).  A wave-uniform test of a compound condition,
// `uni(a || b || c)`, cost a v_cndmask + v_cmp round trip through a vector register on the critical chain of the sweeps:
// the compiler materialises the disjunction as a 0/1 value per lane before it can ballot it.  Here every single comparison
// is balloted as it stands (the v_cmp result IS the mask), the masks are combined by scalar logic, and bit `qbase` -- the
// leading lane of the quad that carries the live problem -- decides for the wave.  The lanes of that quad hold identical
// data, so a selection under the quad's verdict gives them what the lane-wise selection of slide_leftmost_root_uni gives
// them; operation for operation the same arithmetic: bit-identical roots.
// (`pc`: the sextic's coefficients, formed by the caller -- the fused kernel spreads them over the quad's lanes, see
// gs_stage --; `nq`: SlidePre::nq, for the spectrum bound of a cold start)
__device__ __forceinline__ bool slide_leftmost_root_qm_pc(const double pc[7], double nq, double c1, double kappa, double warm,
                                                          double *root, double step_tol, int qbase, double woff) {
#define BM(c) __builtin_amdgcn_ballot_w64(c)
#define QM(m) ((bool)(((m) >> qbase) & 1ull))
    double x = NAN;
    double w0 = 0., w1 = 0., w2 = 0.;
    bool from_shift = false;
    if (QM(BM(warm == warm))) {
        const double x0 = warm - (woff >= 0. ? woff : 1e-3 * fabs(warm)) - 1e-300;
        double t[7];
        for (int i = 0; i < 7; ++i) t[i] = pc[i];
        for (int j = 0; j < 6; ++j)                 // Taylor shift: t[i] = p^(i)(x0) / i!
            for (int i = 5; i >= j; --i) t[i] += x0 * t[i + 1];
        const unsigned long long mc = BM(t[0] > 0.) & BM(t[1] < 0.) & BM(t[2] > 0.) & BM(t[3] < 0.) & BM(t[4] > 0.) & BM(t[5] < 0.);
        if (QM(mc)) {
            x = x0; w0 = t[0]; w1 = t[1]; w2 = 2. * t[2]; from_shift = true;
        }
    }
    if (!from_shift) {
        const double rb = nq + 3. * fabs(c1) + arb_fast_sqrt(fabs(kappa));
        if (QM(BM(!(rb > 0.)) | BM(!(rb < 1e300)))) return false;
        x = -1.0001 * rb - 1e-300;
    }
    const double n = 6.;
    for (int it = 0; it < 40; ++it) {
        double p0, p1, p2;
        unsigned long long m_zero = 0ull;
        if (from_shift) {
            p0 = w0; p1 = w1; p2 = w2;           // (p > 0 > p' certified: the rounding-level zero test is not needed here)
            from_shift = false;
        } else {
            double ee = fabs(pc[6]);
            p0 = pc[6]; p1 = 0.; p2 = 0.;
            const double ax = fabs(x);
            for (int i = 5; i >= 0; --i) {
                p2 = p2 * x + p1; p1 = p1 * x + p0; p0 = p0 * x + pc[i];
                ee = ee * ax + fabs(p0);                    // running Horner error bound
            }
            p2 *= 2.;
            m_zero = BM(fabs(p0) <= 8.9e-16 * (2. * ee - fabs(p0)));      // p(x) = 0 to rounding
        }
        const double rad = (n - 1.) * ((n - 1.) * p1 * p1 - n * p0 * p2);
        // anomalies: not left of all roots any more, or complex roots nearby
        const unsigned long long m_bad = ~m_zero & (BM(!(p0 > 0.)) | BM(!(p1 < 0.)) | BM(!(rad >= 0.)));
        const double den = p1 - arb_fast_sqrt(rad);   // both terms negative: no cancellation
        const double dx = (n * (1. - 9.5367431640625e-07)) * p0 * arb_fast_rcp(den); // negative
        const double xn = x - dx;
        const unsigned long long m_stall = BM(!(xn > x));                 // no representable progress: converged
        if (QM(m_zero | m_bad | m_stall | BM(fabs(dx) <= step_tol * fabs(xn)))) {
            if (QM(m_bad)) return false;
            *root = QM(m_zero | m_stall) ? x : xn;
            return true;
        }
        if (QM(BM(fabs(dx) <= 1e-2 * fabs(xn)))) {
            double q0 = pc[6];
            for (int i = 5; i >= 0; --i) q0 = q0 * xn + pc[i];
            if (QM(BM(fabs(q0) <= 0.25 * step_tol * fabs(xn) * (-p1)))) { *root = xn; return true; }
        }
        x = xn;
    }
    return false;
#undef BM
#undef QM
}
__device__ __forceinline__ bool slide_leftmost_root_qm(const SlidePre &k, double c1, double kappa, double warm, double *root,
                                                       double step_tol, int qbase, double woff) {
    double pc[7];
    slide_poly(k, c1, kappa, pc);
    return slide_leftmost_root_qm_pc(pc, k.nq, c1, kappa, warm, root, step_tol, qbase, woff);
}
#elif defined(__HIPCC__)
// (host pass of the kernel sources: declared, never called)
__device__ bool slide_leftmost_root_qm(const SlidePre &k, double c1, double kappa, double warm, double *root,
                                       double step_tol, int qbase, double woff);
__device__ bool slide_leftmost_root_qm_pc(const double pc[7], double nq, double c1, double kappa, double warm,
                                          double *root, double step_tol, int qbase, double woff);
#endif

// The two sweep-dependent scalars of det(B - sI) from the constants of the constraint's own
// admittance block: yc = Y_c, iyn = 1/y_n, muyn = mu/y_n, b = muyn Y_c, bsq = b.b   (constraints.py:808-812)
template <typename T>
ARB_HD void slide_c1_kappa(const T alpha[4], const T yc[3], T iyn, T muyn, const T b[3], T bsq,
                           double *c1, double *kappa) {
    const T t = alpha[3] * iyn;
    const T beta[3] = {alpha[0] - t * yc[0], alpha[1] - t * yc[1], alpha[2] - t * yc[2]};
    const T a = muyn * alpha[3];
    const T bb = beta[0] * b[0] + beta[1] * b[1] + beta[2] * b[2];
    const T b2 = beta[0] * beta[0] + beta[1] * beta[1] + beta[2] * beta[2];
    const double ia = arb_rcp((double)a);
    *c1 = 2. * ia * (double)bb;
    *kappa = ((double)b2 * ia * ia) * ((double)bsq - 1.);
}

// Laguerre step tolerance per state precision (see slide_leftmost_root)
template <typename T> ARB_HD double slide_step_tol() { return sizeof(T) == 4 ? 1e-7 : 1e-12; }

// Sliding branch, constraints.py:803-830: coefficients of B from (Y, alpha) and the
// shift s.  Returns true with *shift set when the register-only fast path
// succeeded; otherwise writes the 6x6 matrix B to `work` and returns false (the
// caller then runs eig6 on it).
template <typename T, typename AP>
ARB_HD bool softfinger_sliding_shift(const T Y[16], const T alpha[4], T mu, const T eps[3], AP work,
                                     T *shift, bool use_fast = true, const SlidePre *pre = nullptr,
                                     double *warm = nullptr) {
    T Yc[3] = {Y[3], Y[7], Y[11]};
    T yn = Y[15];
    T beta[3], b[3];
    T a = mu / yn * alpha[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { beta[i] = alpha[i] - alpha[3] / yn * Yc[i]; b[i] = mu / yn * Yc[i]; }
    T e2[3] = {eps[0] * eps[0], eps[1] * eps[1], eps[2] * eps[2]};
    T ycyc = Yc[0] * Yc[0] + Yc[1] * Yc[1] + Yc[2] * Yc[2];     // dot(Y_c, Y_c.T): a scalar
    T bb = beta[0] * b[0] + beta[1] * b[1] + beta[2] * b[2];     // dot(beta, b.T)
    T b2 = beta[0] * beta[0] + beta[1] * beta[1] + beta[2] * beta[2];
    T bsq = b[0] * b[0] + b[1] * b[1] + b[2] * b[2];
    if (use_fast && eps[0] == T(1) && eps[1] == T(1) && eps[2] == T(1)) {
        const SlidePre kk = pre ? *pre : slide_precompute<T>(Y);
        const double ia = 1. / (double)a;
        const double c1 = 2. * ia * (double)bb;
        const double kappa = ((double)b2 * ia * ia) * ((double)bsq - 1.);
        double root;
        if (slide_leftmost_root(kk, c1, kappa, warm ? *warm : NAN, &root, slide_step_tol<T>())) {
            if (warm) *warm = root;
            // leftmost real eigenvalue; admissible when <= 0, else no admissible one (constraints.py:826-830)
            *shift = (root <= 0.) ? (T)(root > -1e10 ? root : -1e10) : T(-1e10);
            return true;
        }
#if ARB_ROOT_CASCADE
        // the iteration declined (complex roots in its way): the derivative cascade decides in float64
        const int rc = slide_real_root_cascade_from(kk, c1, kappa, &root);
        if (rc >= 0) {
            if (warm) *warm = NAN;
            *shift = (rc == 1) ? (T)(root > -1e10 ? root : -1e10) : T(-1e10);
            return true;
        }
#endif
    }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            T yhat = Y[4 * i + j] - ycyc / yn;                   // scalar subtracted from every entry
            work[(3 + i) * 6 + (3 + j)] = e2[i] * yhat;
            work[i * 6 + j] = e2[i] * (yhat + T(2) / a * bb);
            work[i * 6 + (3 + j)] = (i == j) ? -(e2[i] * (b2 / (a * a))) : T(0);
            work[(3 + i) * 6 + j] = (i == j) ? (e2[i] * bsq - T(1)) : T(0);
        }
    return false;
}

// ---------------------------------------------------------------------------
// SoftFingerContact.solve, arboris/constraints.py:780-836, including the
// reference's scalar arithmetic in the sliding branch (its dot() of 1-D arrays).
// Split in three so that the kernel can run the rare generic eigenvalue
// fallback on a single lane:
//   softfinger_try   release / static / sliding with the fast shift; returns
//                    0, 1, 2, or 3 = sliding but the 6x6 matrix was written to
//                    `work` and eig6 must provide the shift
//   slide_shift_from_eig   shift from eig6's output
//   softfinger_slide_finish  A = Y - s diag(eps^-2); f = solve(A, -alpha)
// ---------------------------------------------------------------------------
template <typename T, typename AP>
ARB_HD int softfinger_try(const T v[4], const T Y[16], const T P[16], T f[4], T df[4],
                          T sdist, T dt, T mu, const T eps[3], AP work, T alpha[4], T *shift,
                          bool use_fast = true, const SlidePre *pre = nullptr, double *warm = nullptr) {
    T v0[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        v0[i] = v[i] - (Y[4 * i] * f[0] + Y[4 * i + 1] * f[1] + Y[4 * i + 2] * f[2] + Y[4 * i + 3] * f[3]);
    if (sdist + dt * v0[3] > T(0)) {                    // constraints.py:781-785
#pragma unroll
        for (int i = 0; i < 4; ++i) { df[i] = -f[i]; f[i] = T(0); }
        return 0;
    }
    T tgt[4] = {v[0], v[1], v[2], v[3] + sdist / dt};   // constraints.py:795-797
    T fn[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        df[i] = -(P[4 * i] * tgt[0] + P[4 * i + 1] * tgt[1] + P[4 * i + 2] * tgt[2] + P[4 * i + 3] * tgt[3]);
        fn[i] = f[i] + df[i];
    }
    T lhs = (fn[0] / eps[0]) * (fn[0] / eps[0]) + (fn[1] / eps[1]) * (fn[1] / eps[1])
          + (fn[2] / eps[2]) * (fn[2] / eps[2]);
    T rhs = (fn[3] * mu) * (fn[3] * mu);
    if (lhs <= rhs) {                                   // constraints.py:799-802
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = fn[i];
        return 1;
    }
    // sliding, constraints.py:803-836
    alpha[0] = v0[0]; alpha[1] = v0[1]; alpha[2] = v0[2]; alpha[3] = v0[3] + sdist / dt;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ARB_NO_SOFTFINGER_BARRIER)      /* (reproducer: tools/softfinger_barrier_repro.sh) */
    // Compiler workaround (hipcc 7.2, gfx950, -O2 and up): without this barrier the force that comes in, which
    // the finish needs again AFTER the root finder (df = f_new - f), reached softfinger_slide_finish with wrong
    // values in float32 lane-per-world code -- f_new right, df = 1e17 -- for one input in two million solves
    // (-O1 and the host build are right; tests/test_gpu_device_solve.py holds the input).  Pinning f in
    // vector registers here, before the long inlined float64 code, avoids it.
    asm volatile("" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]));
#endif
    return softfinger_sliding_shift<T>(Y, alpha, mu, eps, work, shift, use_fast, pre, warm) ? 2 : 3;
}

template <typename T, typename AP>
ARB_HD T slide_shift_from_eig(AP work) {
    T wr[6], wi[6];
    int nf = eig6<T>(work, wr, wi);
    bool any = false;
    T smin = T(0);
    for (int i = 0; i < 6; ++i) {
        bool ok = (i >= 6 - nf) && (wi[i] == T(0)) && (wr[i] <= T(0));
        if (ok) { smin = any ? (wr[i] < smin ? wr[i] : smin) : wr[i]; any = true; }
    }
    return any ? (smin > T(-1e10) ? smin : T(-1e10)) : T(-1e10);   // constraints.py:827-830
}

// The elimination of gepp4 WITHOUT the row exchanges, for one right-hand side; returns true when partial pivoting
// would not have exchanged anything (|A[r][c]| <= |A[c][c]| below every pivot), in which case x is bit for bit what
// gepp4 returns: the same operations in the same order.  The sliding branch's matrix Y - s diag(eps^-2), s <= 0, is
// an admittance block pushed further towards diagonal dominance: exchanges are the exception, and the 60 selects
// that carry them out were half of the instructions of the solve.
template <typename T>
ARB_HD bool solve4_no_exchange(const T Ain[4][4], const T bin[4], T x[4]) {
    T A[4][4], B[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) A[i][j] = Ain[i][j];
        B[i] = bin[i];
    }
    bool ok = true;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int r = c + 1; r < 4; ++r) ok = ok && !(arb_abs(A[r][c]) > arb_abs(A[c][c]));
        const T ip = arb_rcp(A[c][c]);
        A[c][c] = ip;
#pragma unroll
        for (int r = c + 1; r < 4; ++r) {
            T f = A[r][c] * ip;
#pragma unroll
            for (int j = c + 1; j < 4; ++j) A[r][j] -= f * A[c][j];
            B[r] -= f * B[c];
        }
    }
#pragma unroll
    for (int c = 3; c >= 0; --c) {
        T s = B[c];
#pragma unroll
        for (int k = c + 1; k < 4; ++k) s -= A[c][k] * B[k];
        B[c] = s * A[c][c];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = B[i];
    return ok;
}

// `sie2` = s * eps**-2 (three values).  `uni` (fused kernel: the verdict of the constraint's quad for the whole wave;
// scalar callers: identity) decides between the exchange-free elimination and gepp4.
template <typename T, typename UNI>
ARB_HD void softfinger_slide_finish_scaled(const T Y[16], const T alpha[4], const T sie2[3], T f[4], T df[4], UNI uni) {
    T A[4][4], b[4], x[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) A[i][j] = Y[4 * i + j];
        b[i] = -alpha[i];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) A[i][i] -= sie2[i];                // s * diag(eps**-2)
    if (!uni(solve4_no_exchange<T>(A, b, x))) {
        T Bv[4][1];
#pragma unroll
        for (int i = 0; i < 4; ++i) Bv[i][0] = b[i];
        gepp4<T, 1>(A, Bv);
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = Bv[i][0];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { df[i] = x[i] - f[i]; f[i] = x[i]; }
}

// the fast variant of the sweeps: the exchange-free elimination or nothing -- false (f, df untouched) when the quad's
// verdict is that partial pivoting would exchange rows
template <typename T, typename UNI>
ARB_HD bool softfinger_slide_finish_noex(const T Y[16], const T alpha[4], const T sie2[3], T f[4], T df[4], UNI uni) {
    T A[4][4], b[4], x[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) A[i][j] = Y[4 * i + j];
        b[i] = -alpha[i];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) A[i][i] -= sie2[i];                // s * diag(eps**-2)
    if (!uni(solve4_no_exchange<T>(A, b, x))) return false;
#pragma unroll
    for (int i = 0; i < 4; ++i) { df[i] = x[i] - f[i]; f[i] = x[i]; }
    return true;
}

template <typename T>
ARB_HD void softfinger_slide_finish_scaled(const T Y[16], const T alpha[4], const T sie2[3], T f[4], T df[4]) {
    T A[4][4], Bv[4][1];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) A[i][j] = Y[4 * i + j];
        Bv[i][0] = -alpha[i];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) A[i][i] -= sie2[i];                // s * diag(eps**-2)
    gepp4<T, 1>(A, Bv);
#pragma unroll
    for (int i = 0; i < 4; ++i) { df[i] = Bv[i][0] - f[i]; f[i] = Bv[i][0]; }
}

template <typename T>
ARB_HD void softfinger_slide_finish(const T Y[16], const T alpha[4], const T eps[3], T s, T f[4], T df[4]) {
    const T sie2[3] = {s / (eps[0] * eps[0]), s / (eps[1] * eps[1]), s / (eps[2] * eps[2])};
    softfinger_slide_finish_scaled<T>(Y, alpha, sie2, f, df);
}

// Sequential composition (host tests; `use_fast` = false forces the eig6 route).
template <typename T, typename AP>
ARB_HD int softfinger_solve(const T v[4], const T Y[16], const T P[16], T f[4], T df[4],
                            T sdist, T dt, T mu, const T eps[3], AP work, bool use_fast = true) {
    T alpha[4], s = T(0);
    int br = softfinger_try<T>(v, Y, P, f, df, sdist, dt, mu, eps, work, alpha, &s, use_fast);
    if (br == 3) { s = slide_shift_from_eig<T>(work); br = 2; }
    if (br == 2) softfinger_slide_finish<T>(Y, alpha, eps, s, f, df);
    return br;
}
