/*
 * pm_oracle_impl.h -- TEST INFRASTRUCTURE ONLY (never shipped, never on the product path).
 *
 * Scalar CPU restatement of the reference's hot-path arithmetic, included twice by
 * pm_oracle.c with REAL = float / double.  Every function cites the reference
 * file:line (relative to /root/reference/pymotion) whose arithmetic it follows.
 *
 * Layout conventions (same as the reference): C-contiguous AoS, quaternions
 * [w,x,y,z], matrices row-major [row][col], dual quats [qr(4), qd(4)],
 * ortho6d [3][2] (row-major: r0x r0y r1x r1y r2x r2y).
 */

#ifndef REAL
#error "include from pm_oracle.c"
#endif

#define FN2(name, sfx) name##_##sfx
#define FN1(name, sfx) FN2(name, sfx)
#define FN(name) FN1(name, SFX)

/* ---- small static helpers ------------------------------------------------------- */

/* rotations/quat.py:337-361 (Hamilton product, term order kept) */
static inline void FN(h_qmul)(const REAL *a, const REAL *b, REAL *o) {
    REAL w0 = a[0], x0 = a[1], y0 = a[2], z0 = a[3];
    REAL w1 = b[0], x1 = b[1], y1 = b[2], z1 = b[3];
    o[0] = w0 * w1 - x0 * x1 - y0 * y1 - z0 * z1;
    o[1] = w0 * x1 + w1 * x0 + y0 * z1 - z0 * y1;
    o[2] = w0 * y1 + w1 * y0 + z0 * x1 - x0 * z1;
    o[3] = w0 * z1 + w1 * z0 + x0 * y1 - y0 * x1;
}

/* rotations/quat.py:653-674 */
static inline void FN(h_cross)(const REAL *a, const REAL *b, REAL *o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

/* rotations/quat.py:320-334 : t = 2 (qv x v); v' = v + w t + qv x t */
static inline void FN(h_qmulvec)(const REAL *q, const REAL *v, REAL *o) {
    REAL t[3], u[3];
    FN(h_cross)(q + 1, v, t);
    t[0] *= (REAL)2; t[1] *= (REAL)2; t[2] *= (REAL)2;
    FN(h_cross)(q + 1, t, u);
    o[0] = v[0] + q[0] * t[0] + u[0];
    o[1] = v[1] + q[0] * t[1] + u[1];
    o[2] = v[2] + q[0] * t[2] + u[2];
}

/* rotations/quat.py:364-376, 411-423 : q / (||q|| + eps) */
static inline void FN(h_qnormalize)(const REAL *q, REAL eps, REAL *o) {
    REAL n = SQRT(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    REAL d = n + eps;
    o[0] = q[0] / d; o[1] = q[1] / d; o[2] = q[2] / d; o[3] = q[3] / d;
}

/* rotations/quat.py:276-317 */
static inline void FN(h_q2m)(const REAL *q, REAL *m) {
    REAL qw = q[0], qx = q[1], qy = q[2], qz = q[3];
    REAL x2 = qx + qx, y2 = qy + qy, z2 = qz + qz;
    REAL xx = qx * x2, yy = qy * y2, wx = qw * x2;
    REAL xy = qx * y2, yz = qy * z2, wy = qw * y2;
    REAL xz = qx * z2, zz = qz * z2, wz = qw * z2;
    m[0] = (REAL)1 - (yy + zz); m[1] = xy - wz;               m[2] = xz + wy;
    m[3] = xy + wz;             m[4] = (REAL)1 - (xx + zz);   m[5] = yz - wx;
    m[6] = xz - wy;             m[7] = yz + wx;               m[8] = (REAL)1 - (xx + yy);
}

/* rotations/quat.py:85-156 : 4-branch select, then normalize(eps=1e-8) */
static inline void FN(h_m2q)(const REAL *m, REAL *o) {
    REAL r00 = m[0], r01 = m[1], r02 = m[2];
    REAL r10 = m[3], r11 = m[4], r12 = m[5];
    REAL r20 = m[6], r21 = m[7], r22 = m[8];
    REAL c[4];
    if (r22 < (REAL)0) {
        if (r00 > r11) {
            c[0] = r21 - r12; c[1] = (REAL)1 + r00 - r11 - r22; c[2] = r10 + r01; c[3] = r02 + r20;
        } else {
            c[0] = r02 - r20; c[1] = r10 + r01; c[2] = (REAL)1 - r00 + r11 - r22; c[3] = r21 + r12;
        }
    } else {
        if (r00 < -r11) {
            c[0] = r10 - r01; c[1] = r02 + r20; c[2] = r21 + r12; c[3] = (REAL)1 - r00 - r11 + r22;
        } else {
            c[0] = (REAL)1 + r00 + r11 + r22; c[1] = r21 - r12; c[2] = r02 - r20; c[3] = r10 - r01;
        }
    }
    FN(h_qnormalize)(c, (REAL)1e-8, o);
}

/* rotations/ortho6d.py:67-90 (Gram-Schmidt on columns).  eps: denominators are
 * max(norm, eps); eps = 0 reproduces the NumPy path (NaN on a zero column),
 * eps = 1e-12 the torch twin (ortho6d_torch.py:84-89, F.normalize). */
static inline void FN(h_o6d2m)(const REAL *x, REAL eps, REAL *m) {
    REAL a[3] = {x[0], x[2], x[4]}, b[3] = {x[1], x[3], x[5]};
    REAL na = SQRT(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
    if (na < eps) na = eps;
    REAL c1[3] = {a[0] / na, a[1] / na, a[2] / na};
    REAL d = c1[0] * b[0] + c1[1] * b[1] + c1[2] * b[2];
    REAL c2[3] = {b[0] - d * c1[0], b[1] - d * c1[1], b[2] - d * c1[2]};
    REAL nb = SQRT(c2[0] * c2[0] + c2[1] * c2[1] + c2[2] * c2[2]);
    if (nb < eps) nb = eps;
    c2[0] /= nb; c2[1] /= nb; c2[2] /= nb;
    REAL c3[3];
    FN(h_cross)(c1, c2, c3);
    m[0] = c1[0]; m[1] = c2[0]; m[2] = c3[0];
    m[3] = c1[1]; m[4] = c2[1]; m[5] = c3[1];
    m[6] = c1[2]; m[7] = c2[2]; m[8] = c3[2];
}

/* ---- element-wise conversions (exported) ----------------------------------------- */

void FN(oracle_quat_normalize)(const REAL *q, int64_t n, REAL eps, REAL *out) {
    for (int64_t i = 0; i < n; ++i) FN(h_qnormalize)(q + 4 * i, eps, out + 4 * i);
}
void FN(oracle_quat_length)(const REAL *q, int64_t n, REAL *out) {
    for (int64_t i = 0; i < n; ++i) {
        const REAL *p = q + 4 * i;
        out[i] = SQRT(p[0] * p[0] + p[1] * p[1] + p[2] * p[2] + p[3] * p[3]);
    }
}
void FN(oracle_quat_to_matrix)(const REAL *q, int64_t n, REAL *out) {
    for (int64_t i = 0; i < n; ++i) FN(h_q2m)(q + 4 * i, out + 9 * i);
}
void FN(oracle_quat_from_matrix)(const REAL *m, int64_t n, REAL *out) {
    for (int64_t i = 0; i < n; ++i) FN(h_m2q)(m + 9 * i, out + 4 * i);
}
void FN(oracle_quat_mul)(const REAL *a, const REAL *b, int64_t n, REAL *out) {
    for (int64_t i = 0; i < n; ++i) FN(h_qmul)(a + 4 * i, b + 4 * i, out + 4 * i);
}
void FN(oracle_quat_mul_vec)(const REAL *q, const REAL *v, int64_t n, REAL *out) {
    for (int64_t i = 0; i < n; ++i) FN(h_qmulvec)(q + 4 * i, v + 3 * i, out + 3 * i);
}
/* rotations/quat.py:396-408 (inverse == conjugate, :379-393) */
void FN(oracle_quat_conjugate)(const REAL *q, int64_t n, REAL *out) {
    for (int64_t i = 0; i < n; ++i) {
        out[4 * i] = q[4 * i]; out[4 * i + 1] = -q[4 * i + 1];
        out[4 * i + 2] = -q[4 * i + 2]; out[4 * i + 3] = -q[4 * i + 3];
    }
}

/* rotations/dual_quat.py:12-36 : dq = [qr, 0.5 * (0,t) (x) qr] */
static inline void FN(h_rt2dq)(const REAL *q, const REAL *t, REAL *dq) {
    REAL tq[4] = {(REAL)0, t[0], t[1], t[2]}, d[4];
    FN(h_qmul)(tq, q, d);
    dq[0] = q[0]; dq[1] = q[1]; dq[2] = q[2]; dq[3] = q[3];
    dq[4] = (REAL)0.5 * d[0]; dq[5] = (REAL)0.5 * d[1];
    dq[6] = (REAL)0.5 * d[2]; dq[7] = (REAL)0.5 * d[3];
}
/* rotations/dual_quat.py:62-83 : t = (2 * qd (x) conj(qr))[1:] */
static inline void FN(h_dq2rt)(const REAL *dq, REAL *q, REAL *t) {
    REAL cj[4] = {dq[0], -dq[1], -dq[2], -dq[3]}, d[4];
    FN(h_qmul)(dq + 4, cj, d);
    q[0] = dq[0]; q[1] = dq[1]; q[2] = dq[2]; q[3] = dq[3];
    t[0] = (REAL)2 * d[1]; t[1] = (REAL)2 * d[2]; t[2] = (REAL)2 * d[3];
}
void FN(oracle_dq_from_rt)(const REAL *q, const REAL *t, int64_t n, REAL *out) {
    for (int64_t i = 0; i < n; ++i) FN(h_rt2dq)(q + 4 * i, t + 3 * i, out + 8 * i);
}
void FN(oracle_dq_to_rt)(const REAL *dq, int64_t n, REAL *q, REAL *t) {
    for (int64_t i = 0; i < n; ++i) FN(h_dq2rt)(dq + 8 * i, q + 4 * i, t + 3 * i);
}
/* rotations/dual_quat.py:39-59 */
void FN(oracle_dq_from_t)(const REAL *t, int64_t n, REAL *out) {
    for (int64_t i = 0; i < n; ++i) {
        REAL *o = out + 8 * i;
        o[0] = (REAL)1; o[1] = o[2] = o[3] = o[4] = (REAL)0;
        o[5] = t[3 * i] * (REAL)0.5; o[6] = t[3 * i + 1] * (REAL)0.5; o[7] = t[3 * i + 2] * (REAL)0.5;
    }
}

void FN(oracle_o6d_to_matrix)(const REAL *x, int64_t n, REAL eps, REAL *out) {
    for (int64_t i = 0; i < n; ++i) FN(h_o6d2m)(x + 6 * i, eps, out + 9 * i);
}
/* rotations/ortho6d.py:50-64 */
void FN(oracle_o6d_to_quat)(const REAL *x, int64_t n, REAL eps, REAL *out) {
    for (int64_t i = 0; i < n; ++i) {
        REAL m[9];
        FN(h_o6d2m)(x + 6 * i, eps, m);
        FN(h_m2q)(m, out + 4 * i);
    }
}
/* rotations/ortho6d.py:14-47 : to_matrix then [..., :2] */
void FN(oracle_o6d_from_quat)(const REAL *q, int64_t n, REAL *out) {
    for (int64_t i = 0; i < n; ++i) {
        REAL m[9];
        FN(h_q2m)(q + 4 * i, m);
        REAL *o = out + 6 * i;
        o[0] = m[0]; o[1] = m[1]; o[2] = m[3]; o[3] = m[4]; o[4] = m[6]; o[5] = m[7];
    }
}
void FN(oracle_o6d_from_matrix)(const REAL *m, int64_t n, REAL *out) {
    for (int64_t i = 0; i < n; ++i) {
        const REAL *p = m + 9 * i; REAL *o = out + 6 * i;
        o[0] = p[0]; o[1] = p[1]; o[2] = p[3]; o[3] = p[4]; o[4] = p[6]; o[5] = p[7];
    }
}

/* ---- second wave: trig conversions -------------------------------------------------- */

/* rotations/quat.py:24-40 */
void FN(oracle_quat_from_angle_axis)(const REAL *angle, const REAL *axis, int64_t n, REAL *out) {
    for (int64_t i = 0; i < n; ++i) {
        REAL h = angle[i] / (REAL)2, c = COS(h), s = SIN(h);
        out[4 * i] = c; out[4 * i + 1] = s * axis[3 * i];
        out[4 * i + 2] = s * axis[3 * i + 1]; out[4 * i + 3] = s * axis[3 * i + 2];
    }
}
/* rotations/quat.py:6-21 (zero vector -> 0/0 NaN, kept) */
void FN(oracle_quat_from_scaled_angle_axis)(const REAL *v, int64_t n, REAL *out) {
    for (int64_t i = 0; i < n; ++i) {
        const REAL *p = v + 3 * i;
        REAL a = SQRT(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
        REAL ax[3] = {p[0] / a, p[1] / a, p[2] / a};
        FN(oracle_quat_from_angle_axis)(&a, ax, 1, out + 4 * i);
    }
}
/* rotations/quat.py:247-273 */
void FN(oracle_quat_to_angle_axis)(const REAL *q, int64_t n, REAL *angle, REAL *axis) {
    for (int64_t i = 0; i < n; ++i) {
        REAL w = q[4 * i];
        REAL wc = w < (REAL)-1 ? (REAL)-1 : (w > (REAL)1 ? (REAL)1 : w);
        angle[i] = (REAL)2 * ACOS(wc);
        REAL s2 = (REAL)1 - w * w;
        s2 = s2 < (REAL)0 ? (REAL)0 : (s2 > (REAL)1 ? (REAL)1 : s2);
        REAL s = SQRT(s2);
        for (int k = 0; k < 3; ++k) axis[3 * i + k] = (s > (REAL)1e-8) ? q[4 * i + 1 + k] / s : (REAL)0;
    }
}
/* rotations/quat.py:230-244 */
void FN(oracle_quat_to_scaled_angle_axis)(const REAL *q, int64_t n, REAL *out) {
    for (int64_t i = 0; i < n; ++i) {
        REAL a, ax[3];
        FN(oracle_quat_to_angle_axis)(q + 4 * i, 1, &a, ax);
        out[3 * i] = a * ax[0]; out[3 * i + 1] = a * ax[1]; out[3 * i + 2] = a * ax[2];
    }
}
/* rotations/quat.py:43-82 ; order = uint8 codes 0/1/2 for 'x'/'y'/'z', one triple per element */
void FN(oracle_quat_from_euler)(const REAL *e, const uint8_t *order, int64_t n, REAL *out) {
    for (int64_t i = 0; i < n; ++i) {
        REAL q[3][4];
        for (int k = 0; k < 3; ++k) {
            REAL ax[3] = {0, 0, 0};
            ax[order[3 * i + k]] = (REAL)1;
            FN(oracle_quat_from_angle_axis)(e + 3 * i + k, ax, 1, q[k]);
        }
        REAL t[4];
        FN(h_qmul)(q[1], q[2], t);
        FN(h_qmul)(q[0], t, out + 4 * i);
    }
}
/* rotations/quat.py:159-227 */
void FN(oracle_quat_to_euler)(const REAL *q, const uint8_t *order, int64_t n, REAL *out) {
    const REAL two_pi = (REAL)(2.0 * M_PI);
    for (int64_t i = 0; i < n; ++i) {
        int ii = order[3 * i + 2], jj = order[3 * i + 1], kk = order[3 * i];
        int prod = (ii - jj) * (jj - kk) * (kk - ii);
        /* python floor division by 2 */
        int sgn = (prod >= 0) ? prod / 2 : -((-prod + 1) / 2);
        REAL s = (REAL)sgn;
        const REAL *p = q + 4 * i;
        REAL a = p[0] - p[jj + 1];
        REAL b = p[ii + 1] + p[kk + 1] * s;
        REAL c = p[jj + 1] + p[0];
        REAL d = p[kk + 1] * s - p[ii + 1];
        REAL e1 = (REAL)2 * ATAN2(HYPOT(c, d), HYPOT(a, b)) - (REAL)(M_PI / 2.0);
        REAL hs = ATAN2(b, a), hd = ATAN2(d, c);
        REAL e2 = hs - hd, e0 = (hs + hd) * s;
        REAL ev[3] = {e0, e1, e2};
        for (int k = 0; k < 3; ++k) {
            REAL r = FMOD(ev[k], two_pi);
            if (r != (REAL)0 && r < (REAL)0) r += two_pi; /* np.mod: sign of divisor */
            out[3 * i + k] = r;
        }
    }
}
/* rotations/quat.py:465-501 ; t has one value per element */
void FN(oracle_quat_slerp)(const REAL *q0, const REAL *q1, const REAL *t, int64_t n, int shortest, REAL *out) {
    for (int64_t i = 0; i < n; ++i) {
        const REAL *a = q0 + 4 * i;
        REAL b[4] = {q1[4 * i], q1[4 * i + 1], q1[4 * i + 2], q1[4 * i + 3]};
        REAL dot = a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
        if (shortest && dot < (REAL)0) { b[0] = -b[0]; b[1] = -b[1]; b[2] = -b[2]; b[3] = -b[3]; dot = -dot; }
        dot = dot < (REAL)-1 ? (REAL)-1 : (dot > (REAL)1 ? (REAL)1 : dot);
        REAL th = ACOS(dot) * t[i];
        REAL q2[4], nn = (REAL)0;
        for (int k = 0; k < 4; ++k) { q2[k] = b[k] - a[k] * dot; REAL u = q2[k] + (REAL)0.000001; nn += u * u; }
        nn = SQRT(nn);
        REAL c = COS(th), s = SIN(th);
        for (int k = 0; k < 4; ++k) out[4 * i + k] = c * a[k] + s * (q2[k] / nn);
    }
}

/* rotations/quat.py:426-462 with the unroll axis first: q [T,S,4]; frame i is flipped when its dot with
 * the (already corrected) frame i-1 is negative (the reference's d0 < d1). */
void FN(oracle_quat_unroll)(const REAL *q, int64_t T, int32_t S, REAL *out) {
    for (int64_t i = 0; i < T * S * 4; ++i) out[i] = q[i];
    for (int64_t t = 1; t < T; ++t)
        for (int32_t s = 0; s < S; ++s) {
            REAL *c = out + (t * S + s) * 4;
            const REAL *p = out + ((t - 1) * S + s) * 4;
            REAL d0 = c[0] * p[0] + c[1] * p[1] + c[2] * p[2] + c[3] * p[3];
            if (d0 < -d0) { c[0] = -c[0]; c[1] = -c[1]; c[2] = -c[2]; c[3] = -c[3]; }
        }
}

/* ---- skeleton ops -------------------------------------------------------------------- */

/* ops/skeleton.py:16-61.  G_0 = [R(qhat_0) | root_pos]; G_i = G_parent(i) . [R(qhat_i) | off_i],
 * qhat = q/(||q||+1e-8).  offsets_per_frame != 0 -> offsets is [F,J,3], else [J,3].
 * offsets[0] and parents[0] are ignored exactly like the reference (:49, :53). */
void FN(oracle_fk)(const REAL *rot, const REAL *root_pos, const REAL *offsets, int offsets_per_frame,
                   const int32_t *parents, int64_t F, int32_t J, REAL *pos, REAL *rotmats) {
    for (int64_t f = 0; f < F; ++f) {
        const REAL *q = rot + f * J * 4;
        const REAL *off = offsets_per_frame ? offsets + f * J * 3 : offsets;
        REAL *P = pos + f * J * 3, *R = rotmats + f * J * 9;
        for (int32_t j = 0; j < J; ++j) {
            REAL qn[4], L[9];
            FN(h_qnormalize)(q + 4 * j, (REAL)1e-8, qn);
            FN(h_q2m)(qn, L);
            if (j == 0) {
                for (int k = 0; k < 9; ++k) R[k] = L[k];
                P[0] = root_pos[3 * f]; P[1] = root_pos[3 * f + 1]; P[2] = root_pos[3 * f + 2];
                continue;
            }
            const REAL *Rp = R + 9 * parents[j], *Pp = P + 3 * parents[j], *t = off + 3 * j;
            REAL *Rj = R + 9 * j, *Pj = P + 3 * j;
            /* the reference multiplies homogeneous 4 x 4 matrices (:54-57): row r of the rotation block carries the fourth term
             * p_parent[r] * 0 -- NaN as soon as the parent's position is NaN / Inf -- and the position p_parent[r] * 1 */
            for (int r = 0; r < 3; ++r) {
                for (int c = 0; c < 3; ++c)
                    Rj[3 * r + c] = Rp[3 * r] * L[c] + Rp[3 * r + 1] * L[3 + c] + Rp[3 * r + 2] * L[6 + c] + Pp[r] * (REAL)0;
                Pj[r] = Rp[3 * r] * t[0] + Rp[3 * r + 1] * t[1] + Rp[3 * r + 2] * t[2] + Pp[r] * (REAL)1;
            }
        }
    }
}

/* ops/skeleton.py:207-244.  Joint 0 carries (q_0, root_pos); joints whose parent is 0 stay
 * local (:236-237); deeper joints are composed with their already-root-space parent.
 * Input quats are NOT normalised (as the reference).  Joint axis is -2. */
void FN(oracle_to_root_dq)(const REAL *rot, const REAL *root_pos, const int32_t *parents,
                           const REAL *offsets, int64_t F, int32_t J, REAL *dq) {
    REAL *q = (REAL *)malloc(sizeof(REAL) * 4 * (size_t)J);
    REAL *t = (REAL *)malloc(sizeof(REAL) * 3 * (size_t)J);
    for (int64_t f = 0; f < F; ++f) {
        for (int32_t j = 0; j < J; ++j) {
            for (int k = 0; k < 4; ++k) q[4 * j + k] = rot[(f * J + j) * 4 + k];
            for (int k = 0; k < 3; ++k) t[3 * j + k] = offsets[3 * j + k];
        }
        t[0] = root_pos[3 * f]; t[1] = root_pos[3 * f + 1]; t[2] = root_pos[3 * f + 2];
        for (int32_t j = 1; j < J; ++j) {
            int32_t p = parents[j];
            if (p == 0) continue;
            REAL tv[3], qq[4];
            FN(h_qmulvec)(q + 4 * p, t + 3 * j, tv);
            t[3 * j] = tv[0] + t[3 * p]; t[3 * j + 1] = tv[1] + t[3 * p + 1]; t[3 * j + 2] = tv[2] + t[3 * p + 2];
            FN(h_qmul)(q + 4 * p, q + 4 * j, qq);
            q[4 * j] = qq[0]; q[4 * j + 1] = qq[1]; q[4 * j + 2] = qq[2]; q[4 * j + 3] = qq[3];
        }
        for (int32_t j = 0; j < J; ++j) FN(h_rt2dq)(q + 4 * j, t + 3 * j, dq + (f * J + j) * 8);
    }
    free(q); free(t);
}

/* ops/skeleton.py:173-204.  Reverse joint order; parent still in root space when used.
 * Returns (translations, rotations) like the reference (:204). */
void FN(oracle_from_root_dq)(const REAL *dq, const int32_t *parents, int64_t F, int32_t J,
                             REAL *trans, REAL *rot) {
    for (int64_t f = 0; f < F; ++f) {
        REAL *q = rot + f * J * 4, *t = trans + f * J * 3;
        for (int32_t j = 0; j < J; ++j) FN(h_dq2rt)(dq + (f * J + j) * 8, q + 4 * j, t + 3 * j);
        for (int32_t j = J - 1; j >= 1; --j) {
            int32_t p = parents[j];
            if (p == 0) continue;
            REAL inv[4] = {q[4 * p], -q[4 * p + 1], -q[4 * p + 2], -q[4 * p + 3]};
            REAL d[3] = {t[3 * j] - t[3 * p], t[3 * j + 1] - t[3 * p + 1], t[3 * j + 2] - t[3 * p + 2]};
            REAL tv[3], qq[4];
            FN(h_qmulvec)(inv, d, tv);
            t[3 * j] = tv[0]; t[3 * j + 1] = tv[1]; t[3 * j + 2] = tv[2];
            FN(h_qmul)(inv, q + 4 * j, qq);
            q[4 * j] = qq[0]; q[4 * j + 1] = qq[1]; q[4 * j + 2] = qq[2]; q[4 * j + 3] = qq[3];
        }
    }
}

/* Composite of config 4: rotations/ortho6d.py:50-64 (to_quat) feeding ops/skeleton.py:16-61 (fk).
 * quat_out may be NULL. */
void FN(oracle_fk_from_ortho6d)(const REAL *o6d, const REAL *root_pos, const REAL *offsets,
                                int offsets_per_frame, const int32_t *parents, int64_t F, int32_t J,
                                REAL eps, REAL *pos, REAL *rotmats, REAL *quat_out) {
    REAL *q = (REAL *)malloc(sizeof(REAL) * 4 * (size_t)J);
    for (int64_t f = 0; f < F; ++f) {
        FN(oracle_o6d_to_quat)(o6d + f * J * 6, J, eps, q);
        if (quat_out) memcpy(quat_out + f * J * 4, q, sizeof(REAL) * 4 * (size_t)J);
        FN(oracle_fk)(q, root_pos + 3 * f, offsets_per_frame ? offsets + f * J * 3 : offsets, 0, parents,
                      1, J, pos + f * J * 3, rotmats + f * J * 9);
    }
    free(q);
}

/* ops/skeleton.py:64-93 (from_global_rotations): local_j = conj(global_parent(j)) (x) global_j */
void FN(oracle_from_global_rotations)(const REAL *gq, const int32_t *parents, int64_t F, int32_t J, REAL *out) {
    for (int64_t f = 0; f < F; ++f) {
        const REAL *g = gq + f * J * 4; REAL *o = out + f * J * 4;
        o[0] = g[0]; o[1] = g[1]; o[2] = g[2]; o[3] = g[3];
        for (int32_t j = 1; j < J; ++j) {
            const REAL *gp = g + 4 * parents[j];
            REAL inv[4] = {gp[0], -gp[1], -gp[2], -gp[3]};
            FN(h_qmul)(inv, g + 4 * j, o + 4 * j);
        }
    }
}

/* np.isclose(x, target): |x - target| <= 1e-8 + 1e-5 |target| */
static inline int FN(h_close)(REAL x, REAL target) { return FABS(x - target) <= (REAL)1e-8 + (REAL)1e-5 * FABS(target); }
/* ops/vector.py:4-19 */
static inline void FN(h_vnorm)(const REAL *v, REAL *o) {
    REAL n = SQRT(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]) + (REAL)1e-8;
    o[0] = v[0] / n; o[1] = v[1] / n; o[2] = v[2] / n;
}
/* rotations/quat.py:504-576 */
static inline void FN(h_from_to)(const REAL *v1, const REAL *v2, int normalize_input, REAL *o) {
    REAL a[3] = {v1[0], v1[1], v1[2]}, b[3] = {v2[0], v2[1], v2[2]}, cr[3], ax[3];
    if (normalize_input) { FN(h_vnorm)(v1, a); FN(h_vnorm)(v2, b); }
    FN(h_cross)(a, b, cr);
    REAL dot = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
    FN(h_vnorm)(cr, ax);
    REAL w = SQRT(((REAL)1 + dot) * (REAL)0.5), s = SQRT(((REAL)1 - dot) * (REAL)0.5);
    o[0] = w; o[1] = ax[0] * s; o[2] = ax[1] * s; o[3] = ax[2] * s;
    if (FN(h_close)(dot, (REAL)1)) { o[0] = 1; o[1] = o[2] = o[3] = 0; }
    if (FN(h_close)(dot, (REAL)-1)) {
        int xl = FN(h_close)(FABS(a[0]), (REAL)1);
        REAL og[3] = {xl ? (REAL)0 : (REAL)1, xl ? (REAL)1 : (REAL)0, (REAL)0}, c2[3], a2[3];
        FN(h_cross)(a, og, c2);
        FN(h_vnorm)(c2, a2);
        o[0] = 0; o[1] = a2[0]; o[2] = a2[1]; o[3] = a2[2];
    }
}
/* rotations/quat.py:579-650 */
static inline void FN(h_from_to_axis)(const REAL *v1, const REAL *v2, const REAL *axis, int normalize_input, REAL *o) {
    REAL a[3] = {v1[0], v1[1], v1[2]}, b[3] = {v2[0], v2[1], v2[2]}, cr[3];
    if (normalize_input) { FN(h_vnorm)(v1, a); FN(h_vnorm)(v2, b); }
    FN(h_cross)(a, b, cr);
    REAL dot = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
    REAL w = SQRT(((REAL)1 + dot) * (REAL)0.5), s = SQRT(((REAL)1 - dot) * (REAL)0.5);
    REAL cda = cr[0] * axis[0] + cr[1] * axis[1] + cr[2] * axis[2];
    s *= (cda > 0) ? (REAL)1 : ((cda < 0) ? (REAL)-1 : cda);
    o[0] = w; o[1] = axis[0] * s; o[2] = axis[1] * s; o[3] = axis[2] * s;
    if (FN(h_close)(dot, (REAL)1)) { o[0] = 1; o[1] = o[2] = o[3] = 0; }
    if (FN(h_close)(dot, (REAL)-1)) { o[0] = 0; o[1] = axis[0]; o[2] = axis[1]; o[3] = axis[2]; }
}
void FN(oracle_quat_from_to)(const REAL *v1, const REAL *v2, int64_t n, int normalize_input, REAL *out) {
    for (int64_t i = 0; i < n; ++i) FN(h_from_to)(v1 + 3 * i, v2 + 3 * i, normalize_input, out + 4 * i);
}
void FN(oracle_quat_from_to_axis)(const REAL *v1, const REAL *v2, const REAL *ax, int64_t n, int normalize_input, REAL *out) {
    for (int64_t i = 0; i < n; ++i) FN(h_from_to_axis)(v1 + 3 * i, v2 + 3 * i, ax + 3 * i, normalize_input, out + 4 * i);
}

/* ops/skeleton.py:96-170, followed LITERALLY (an fk + from_matrix per aligned joint and per extra child), one
 * frame at a time.  The product's O(J) kernel must reproduce this. */
void FN(oracle_from_root_positions)(const REAL *positions, const int32_t *parents, const REAL *offsets, int64_t F,
                                    int32_t J, REAL *rotations) {
    REAL *pos = (REAL *)malloc(sizeof(REAL) * 3 * (size_t)J), *rm = (REAL *)malloc(sizeof(REAL) * 9 * (size_t)J);
    REAL *gq = (REAL *)malloc(sizeof(REAL) * 4 * (size_t)J);
    const REAL zero[3] = {0, 0, 0};
    for (int64_t f = 0; f < F; ++f) {
        const REAL *P = positions + f * J * 3;
        REAL *rot = rotations + f * J * 4;
        for (int32_t j = 0; j < J; ++j) { rot[4 * j] = 1; rot[4 * j + 1] = rot[4 * j + 2] = rot[4 * j + 3] = 0; }
        for (int32_t j = 0; j < J; ++j) {
            int first = -1;
            for (int32_t c = 1; c < J; ++c) {
                if (parents[c] != j) continue;
                FN(oracle_fk)(rot, zero, offsets, 0, parents, 1, J, pos, rm);
                FN(oracle_quat_from_matrix)(rm, J, gq);
                REAL inv[4] = {gq[4 * j], -gq[4 * j + 1], -gq[4 * j + 2], -gq[4 * j + 3]};
                if (first < 0) {
                    first = c;
                    REAL rd[3] = {pos[3 * c] - pos[3 * j], pos[3 * c + 1] - pos[3 * j + 1], pos[3 * c + 2] - pos[3 * j + 2]};
                    REAL pd[3] = {P[3 * c] - P[3 * j], P[3 * c + 1] - P[3 * j + 1], P[3 * c + 2] - P[3 * j + 2]};
                    REAL rdl[3], pdl[3];
                    FN(h_qmulvec)(inv, rd, rdl);
                    FN(h_qmulvec)(inv, pd, pdl);
                    FN(h_from_to)(rdl, pdl, 1, rot + 4 * j);
                } else {
                    REAL rd[3] = {pos[3 * c] - pos[3 * j], pos[3 * c + 1] - pos[3 * j + 1], pos[3 * c + 2] - pos[3 * j + 2]};
                    REAL pd[3] = {P[3 * c] - P[3 * j], P[3 * c + 1] - P[3 * j + 1], P[3 * c + 2] - P[3 * j + 2]};
                    REAL d0[3] = {P[3 * first] - P[3 * j], P[3 * first + 1] - P[3 * j + 1], P[3 * first + 2] - P[3 * j + 2]};
                    REAL rdl[3], pdl[3], dn[3], ax[3], roll[4], r2[4];
                    FN(h_qmulvec)(inv, rd, rdl);
                    FN(h_qmulvec)(inv, pd, pdl);
                    FN(h_vnorm)(d0, dn);
                    FN(h_qmulvec)(inv, dn, ax);
                    FN(h_from_to_axis)(rdl, pdl, ax, 1, roll);
                    FN(h_qmul)(rot + 4 * j, roll, r2);
                    rot[4 * j] = r2[0]; rot[4 * j + 1] = r2[1]; rot[4 * j + 2] = r2[2]; rot[4 * j + 3] = r2[3];
                }
            }
        }
    }
    free(pos); free(rm); free(gq);
}

/* ---- time axis ------------------------------------------------------------------------ */

/* ops/time.py:4-66 interpolate_positions (linear), positions viewed as [A,T,B] with the time axis in the
 * middle, out [A,S,B].  idx = clamp(searchsorted(original, sample, side='left') - 1, 0, T-2) (:49-52),
 * w = (sample - original[idx]) / (original[idx+1] - original[idx]) (:53-54),
 * out = (1 - w) * p[idx] + w * p[idx+1] (:61-64): samples outside the original range extrapolate. */
void FN(oracle_interpolate_positions)(const REAL *sample_times, const REAL *original_times, const REAL *positions,
                                      int64_t A, int64_t T, int64_t S, int64_t B, REAL *out) {
    for (int64_t s = 0; s < S; ++s) {
        const REAL v = sample_times[s];
        int64_t lo = 0, hi = T;  /* first i with original[i] >= v */
        while (lo < hi) { const int64_t mid = (lo + hi) / 2; if (original_times[mid] < v) lo = mid + 1; else hi = mid; }
        int64_t i = lo - 1;
        if (i < 0) i = 0;
        if (i > T - 2) i = T - 2;
        const REAL w = (v - original_times[i]) / (original_times[i + 1] - original_times[i]);
        for (int64_t a = 0; a < A; ++a) {
            const REAL *p0 = positions + (a * T + i) * B, *p1 = p0 + B;
            REAL *o = out + (a * S + s) * B;
            for (int64_t b = 0; b < B; ++b) o[b] = ((REAL)1 - w) * p0[b] + w * p1[b];
        }
    }
}

#undef FN
#undef FN1
#undef FN2
