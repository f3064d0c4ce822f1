/*
 * ingvio_oracle.c — CPU restatement of the InGVIO covariance hot path (see ingvio_oracle.h).
 * TEST INFRASTRUCTURE ONLY: the checker and the timed CPU baseline, never the product.
 * Plain C99, FP64 everywhere (the reference is all-double: State.h:133, VecState.h).
 * Operation order follows the cited reference lines; Eigen / SPQR / Boost calls are replaced by
 * the textbook algorithms named at each site.
 */
#include "ingvio_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <malloc.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define CM(A, ld, i, j) ((A)[(size_t)(j) * (size_t)(ld) + (size_t)(i)])

/* ------------------------------------------------------------------------------------------ */
/* 3x3 helpers, row-major                                                                      */
/* ------------------------------------------------------------------------------------------ */
static void m3_mul(const double A[9], const double B[9], double C[9])
{
    double T[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            T[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
    memcpy(C, T, sizeof T);
}
static void m3_mulv(const double A[9], const double x[3], double y[3])
{
    double t[3];
    for (int i = 0; i < 3; ++i) t[i] = A[3 * i] * x[0] + A[3 * i + 1] * x[1] + A[3 * i + 2] * x[2];
    y[0] = t[0]; y[1] = t[1]; y[2] = t[2];
}
static void m3_tmulv(const double A[9], const double x[3], double y[3]) /* y = A^T x */
{
    double t[3];
    for (int i = 0; i < 3; ++i) t[i] = A[i] * x[0] + A[3 + i] * x[1] + A[6 + i] * x[2];
    y[0] = t[0]; y[1] = t[1]; y[2] = t[2];
}
static void m3_T(const double A[9], double B[9])
{
    double T[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) T[3 * i + j] = A[3 * j + i];
    memcpy(B, T, sizeof T);
}
static void m3_scale(double A[9], double s) { for (int i = 0; i < 9; ++i) A[i] *= s; }
static void m3_eye(double A[9]) { memset(A, 0, 9 * sizeof(double)); A[0] = A[4] = A[8] = 1.0; }
static double v3_norm(const double v[3]) { return sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }

/* AuxGammaFunc.cpp:28-35 */
void orc_skew(const double v[3], double M[9])
{
    M[0] = 0.0;   M[1] = -v[2]; M[2] = v[1];
    M[3] = v[2];  M[4] = 0.0;   M[5] = -v[0];
    M[6] = -v[1]; M[7] = v[0];  M[8] = 0.0;
}

/* AuxGammaFunc.cpp:46-113 */
void orc_gamma(const double v[3], int m, double out[9])
{
    double theta = v3_norm(v);
    if (fabs(theta) < 1e-06) {
        double factor = 1.0;
        if (m == 3) factor = 1.0 / 6.0;
        else if (m == 2) factor = 0.5;
        m3_eye(out);
        m3_scale(out, factor);
        return;
    }
    double n[3] = { v[0] / theta, v[1] / theta, v[2] / theta };
    double nx[9], nx2[9];
    orc_skew(n, nx);
    m3_mul(nx, nx, nx2);
    double f0, f1, f2;
    double s = sin(theta), c = cos(theta);
    switch (m) {
    case 1:
        f0 = 1.0; f1 = (1.0 - c) / theta; f2 = (theta - s) / theta; break;
    case 2:
        f0 = 0.5; f1 = (theta - s) / pow(theta, 2);
        f2 = (pow(theta, 2) + 2.0 * c - 2.0) / (2.0 * pow(theta, 2)); break;
    case 3: {
        double t3 = pow(theta, 3);
        f0 = 1.0 / 6.0; f1 = (pow(theta, 2) + 2.0 * c - 2.0) / (2.0 * t3);
        f2 = (t3 - 6.0 * theta + 6.0 * s) / (6.0 * t3); break;
    }
    default:
        f0 = 1.0; f1 = s; f2 = 1.0 - c; break;
    }
    for (int i = 0; i < 9; ++i) out[i] = f1 * nx[i] + f2 * nx2[i];
    out[0] += f0; out[4] += f0; out[8] += f0;
}

/* shared body of Psi1Func / Psi2Func: the six skew products (AuxGammaFunc.cpp:123-133, 177-187) */
static void psi_products(const double w[3], const double a[3], double WA[9], double WAW[9],
                         double WAW2[9], double W2A[9], double W2AW[9], double W2AW2[9])
{
    double W[9], A[9];
    orc_skew(w, W); orc_skew(a, A);
    m3_mul(W, A, WA);
    m3_mul(WA, W, WAW);
    m3_mul(WAW, W, WAW2);
    m3_mul(W, WA, W2A);
    m3_mul(W2A, W, W2AW);
    m3_mul(W2AW, W, W2AW2);
}

/* AuxGammaFunc.cpp:115-166.  NB the reference multiplies M1 by the bracket (:163); kept as written. */
void orc_psi1(const double w[3], const double a[3], double dt, double out[9])
{
    double wdt[3] = { w[0] * dt, w[1] * dt, w[2] * dt };
    if (v3_norm(wdt) < 1e-08) { memset(out, 0, 9 * sizeof(double)); return; }
    double A[9], G2[9], M1[9], mw[3] = { -wdt[0], -wdt[1], -wdt[2] };
    orc_skew(a, A);
    orc_gamma(mw, 2, G2);
    m3_mul(A, G2, M1);
    m3_scale(M1, pow(dt, 2.0));
    double WA[9], WAW[9], WAW2[9], W2A[9], W2AW[9], W2AW2[9];
    psi_products(w, a, WA, WAW, WAW2, W2A, W2AW, W2AW2);
    double eta = v3_norm(w), xi = eta * dt, xi2 = pow(xi, 2.0);
    double sx = sin(xi), cx = cos(xi), s2 = sin(2 * xi), c2x = cos(2 * xi);
    double eta3 = pow(eta, 3), eta4 = eta * eta3, eta5 = eta * eta4, eta6 = eta * eta5;
    double c1 = (sx - xi * cx) / eta3;
    double c2 = (c2x - 4 * cx + 3) / (4 * eta4);
    double c3 = (4 * sx + s2 - 4 * xi * cx - 2 * xi) / (4 * eta5);
    double c4 = (xi2 - 2 * xi * sx - 2 * cx + 2) / (2 * eta4);
    double c5 = (6 * xi - 8 * sx + s2) / (4 * eta5);
    double c6 = (2 * xi2 - 4 * xi * sx - c2x + 1) / (4 * eta6);
    double S[9];
    for (int i = 0; i < 9; ++i)
        S[i] = c1 * WA[i] + c2 * WAW[i] + c3 * WAW2[i] + c4 * W2A[i] + c5 * W2AW[i] + c6 * W2AW2[i];
    m3_mul(M1, S, out);
}

/* AuxGammaFunc.cpp:168-225 */
void orc_psi2(const double w[3], const double a[3], double dt, double out[9])
{
    double wdt[3] = { w[0] * dt, w[1] * dt, w[2] * dt };
    if (v3_norm(wdt) < 1e-07) { memset(out, 0, 9 * sizeof(double)); return; }
    double A[9], G3[9], M1[9], mw[3] = { -wdt[0], -wdt[1], -wdt[2] };
    orc_skew(a, A);
    orc_gamma(mw, 3, G3);
    m3_mul(A, G3, M1);
    m3_scale(M1, pow(dt, 3));
    double WA[9], WAW[9], WAW2[9], W2A[9], W2AW[9], W2AW2[9];
    psi_products(w, a, WA, WAW, WAW2, W2A, W2AW, W2AW2);
    double eta = v3_norm(w), xi = eta * dt, xi2 = pow(xi, 2.0), xi3 = xi * xi2;
    double sx = sin(xi), cx = cos(xi), s2 = sin(2 * xi), c2x = cos(2 * xi);
    double eta3 = pow(eta, 3), eta4 = eta * eta3, eta5 = eta * eta4, eta6 = eta * eta5, eta7 = eta * eta6;
    double c1 = (xi * sx + 2 * cx - 2) / eta4;
    double c2 = (6 * xi - 8 * sx + s2) / (8 * eta5);
    double c3 = (2 * xi2 + 8 * xi * sx + 16 * cx + c2x - 17) / (8 * eta6);
    double c4 = (xi3 + 6 * xi - 12 * sx + 6 * xi * cx) / (6 * eta5);
    double c5 = (6 * xi2 + 16 * cx - c2x - 15) / (8 * eta6);
    double c6 = (4 * xi3 + 6 * xi - 24 * sx - 3 * s2 + 24 * xi * cx) / (24 * eta7);
    double S[9];
    for (int i = 0; i < 9; ++i)
        S[i] = c1 * WA[i] + c2 * WAW[i] + c3 * WAW2[i] + c4 * W2A[i] + c5 * W2AW[i] + c6 * W2AW2[i];
    m3_mul(M1, S, out);
}

/* PoseState.cpp:79-88 */
void orc_se3_update(double R[9], double p[3], const double dx[6])
{
    double G0[9], G1[9], t1[3], t2[3];
    orc_gamma(dx, 0, G0);
    orc_gamma(dx, 1, G1);
    m3_mul(G0, R, R);
    m3_mulv(G0, p, t1);
    m3_mulv(G1, dx + 3, t2);
    for (int i = 0; i < 3; ++i) p[i] = t1[i] + t2[i];
}

/* PoseState.cpp:174-186 */
void orc_se23_update(double R[9], double p[3], double v[3], const double dx[9])
{
    double G0[9], G1[9], t1[3], t2[3];
    orc_gamma(dx, 0, G0);
    orc_gamma(dx, 1, G1);
    m3_mul(G0, R, R);
    m3_mulv(G0, p, t1); m3_mulv(G1, dx + 3, t2);
    for (int i = 0; i < 3; ++i) p[i] = t1[i] + t2[i];
    m3_mulv(G0, v, t1); m3_mulv(G1, dx + 6, t2);
    for (int i = 0; i < 3; ++i) v[i] = t1[i] + t2[i];
}

/* write a row-major 3x3 block B (scaled by s) into column-major M (ld) at (r0,c0) */
static void put33(double* M, int ld, int r0, int c0, const double B[9], double s)
{
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) CM(M, ld, r0 + i, c0 + j) = s * B[3 * i + j];
}

/* ImuPropagator.cpp:98-162, analytic branch */
void orc_imu_transition(double R[9], double p[3], double v[3],
                        const double bg[3], const double ba[3],
                        const double gyro[3], const double acc[3],
                        const double gravity[3], double dt,
                        double Phi[225], double G[180])
{
    memset(Phi, 0, 225 * sizeof(double));
    memset(G, 0, 180 * sizeof(double));
    for (int i = 0; i < 15; ++i) CM(Phi, 15, i, i) = 1.0;

    double Rh[9], ph[3], vh[3];
    memcpy(Rh, R, sizeof Rh); memcpy(ph, p, sizeof ph); memcpy(vh, v, sizeof vh);

    /* :112-117 */
    double S[9], T[9], I3[9];
    m3_eye(I3);
    put33(G, 15, 0, 0, Rh, 1.0);
    orc_skew(ph, S); m3_mul(S, Rh, T); put33(G, 15, 3, 0, T, 1.0);
    orc_skew(vh, S); m3_mul(S, Rh, T); put33(G, 15, 6, 0, T, 1.0);
    put33(G, 15, 6, 3, Rh, 1.0);
    put33(G, 15, 9, 6, I3, 1.0);
    put33(G, 15, 12, 9, I3, 1.0);

    double w[3] = { gyro[0] - bg[0], gyro[1] - bg[1], gyro[2] - bg[2] };
    double a[3] = { acc[0] - ba[0], acc[1] - ba[1], acc[2] - ba[2] };
    double wdt[3] = { dt * w[0], dt * w[1], dt * w[2] };
    double G0[9], G1[9], G2[9];
    orc_gamma(wdt, 0, G0); orc_gamma(wdt, 1, G1); orc_gamma(wdt, 2, G2);

    double Rn[9];
    m3_mul(Rh, G0, Rn);                                     /* :130 */
    double RG1[9], RG2[9], t[3];
    m3_mul(Rh, G1, RG1); m3_mul(Rh, G2, RG2);
    double vn[3], pn[3];
    m3_mulv(RG1, a, t);
    for (int i = 0; i < 3; ++i) vn[i] = vh[i] + gravity[i] * dt + t[i] * dt;          /* :133 */
    m3_mulv(RG2, a, t);
    for (int i = 0; i < 3; ++i)
        pn[i] = ph[i] + vh[i] * dt + 0.5 * gravity[i] * pow(dt, 2) + t[i] * pow(dt, 2); /* :136 */

    double Sg[9];
    orc_skew(gravity, Sg);
    put33(Phi, 15, 3, 0, Sg, 0.5 * pow(dt, 2));             /* :150 */
    put33(Phi, 15, 3, 6, I3, dt);                           /* :151 */
    put33(Phi, 15, 6, 0, Sg, dt);                           /* :152 */
    put33(Phi, 15, 0, 9, RG1, -dt);                         /* :154 */
    put33(Phi, 15, 6, 12, RG1, -dt);                        /* :155 */
    put33(Phi, 15, 3, 12, RG2, -pow(dt, 2));                /* :157 */

    double P1[9], P2[9], B[9], Sv[9];
    orc_psi1(w, a, dt, P1); orc_psi2(w, a, dt, P2);
    orc_skew(vn, Sv); m3_mul(Sv, RG1, B);                   /* :159 */
    m3_mul(Rh, P1, T);
    for (int i = 0; i < 9; ++i) B[i] = -B[i] * dt + T[i];
    put33(Phi, 15, 6, 9, B, 1.0);
    orc_skew(pn, Sv); m3_mul(Sv, RG1, B);                   /* :161 */
    m3_mul(Rh, P2, T);
    for (int i = 0; i < 9; ++i) B[i] = -B[i] * dt + T[i];
    put33(Phi, 15, 3, 9, B, 1.0);

    memcpy(R, Rn, sizeof Rn); memcpy(p, pn, sizeof pn); memcpy(v, vn, sizeof vn);
}

/* ------------------------------------------------------------------------------------------ */
/* dense column-major helpers                                                                  */
/* ------------------------------------------------------------------------------------------ */
static double* dalloc(size_t n) { double* p = (double*)calloc(n ? n : 1, sizeof(double)); return p; }

/* C(m x n) = A(m x k) * B(k x n) [+ C if acc] ; transB: use B^T (B is n x k) */
static void gemm(int m, int n, int k, const double* A, int lda, const double* B, int ldb, int transB,
                 double* C, int ldc, int acc)
{
    for (int j = 0; j < n; ++j) {
        double* cj = C + (size_t)j * ldc;
        if (!acc) for (int i = 0; i < m; ++i) cj[i] = 0.0;
        for (int l = 0; l < k; ++l) {
            double b = transB ? CM(B, ldb, j, l) : CM(B, ldb, l, j);
            if (b == 0.0) continue;
            const double* al = A + (size_t)l * lda;
            for (int i = 0; i < m; ++i) cj[i] += al[i] * b;
        }
    }
}

static void symmetrize(double* P, int n, int ld)
{
    for (int j = 0; j < n; ++j)
        for (int i = j + 1; i < n; ++i) {
            double s = 0.5 * (CM(P, ld, i, j) + CM(P, ld, j, i));
            CM(P, ld, i, j) = s; CM(P, ld, j, i) = s;
        }
}

/* StateManager.cpp:42-119 — same three O(N^2) passes as the reference (copy, assemble, symmetrise) */
void orc_propagate_cov(double* P, int n, int ld, const double* Phi, const double* G, double dt,
                       const double sigma[4], int enable_gnss, const int gnss_idx[5],
                       double sigma_cb, double sigma_rw)
{
    int r = n - 15;
    double* tmp = dalloc((size_t)n * n);
    double A[225], T[225];
    /* :51  Phi * P11 * Phi^T */
    gemm(15, 15, 15, Phi, 15, P, ld, 0, T, 15, 0);
    gemm(15, 15, 15, T, 15, Phi, 15, 1, A, 15, 0);
    /* :53-54 */
    double* c21 = dalloc((size_t)r * 15);
    double* c22 = dalloc((size_t)r * r);
    if (r > 0) {
        gemm(r, 15, 15, P + 15, ld, Phi, 15, 1, c21, r, 0);
        for (int j = 0; j < r; ++j) for (int i = 0; i < r; ++i) CM(c22, r, i, j) = CM(P, ld, 15 + i, 15 + j);
    }
    /* :56-86 clock-bias <- clock-drift coupling */
    if (enable_gnss && gnss_idx[4] >= 0 && r > 0) {
        double* c21t = dalloc((size_t)r * 15);
        double* c22t = dalloc((size_t)r * r);
        memcpy(c21t, c21, sizeof(double) * r * 15);
        memcpy(c22t, c22, sizeof(double) * r * r);
        int lc = gnss_idx[4] - 15;
        for (int g = 0; g < 4; ++g) if (gnss_idx[g] >= 0) {
            int lr = gnss_idx[g] - 15;
            for (int j = 0; j < 15; ++j) CM(c21t, r, lr, j) += dt * CM(c21, r, lc, j);
            for (int j = 0; j < r; ++j) CM(c22t, r, lr, j) += dt * CM(c22, r, lc, j);
        }
        memcpy(c21, c21t, sizeof(double) * r * 15);
        memcpy(c22, c22t, sizeof(double) * r * r);
        for (int g = 0; g < 4; ++g) if (gnss_idx[g] >= 0) {
            int lr = gnss_idx[g] - 15;
            for (int i = 0; i < r; ++i) CM(c22t, r, i, lr) += dt * CM(c22, r, i, lc);
        }
        memcpy(c22, c22t, sizeof(double) * r * r);
        free(c21t); free(c22t);
    }
    /* :88-90 */
    for (int j = 0; j < 15; ++j) for (int i = 0; i < 15; ++i) CM(tmp, n, i, j) = CM(A, 15, i, j);
    for (int j = 0; j < 15; ++j) for (int i = 0; i < r; ++i) {
        CM(tmp, n, 15 + i, j) = CM(c21, r, i, j);
        CM(tmp, n, j, 15 + i) = CM(c21, r, i, j);
    }
    for (int j = 0; j < r; ++j) for (int i = 0; i < r; ++i) CM(tmp, n, 15 + i, 15 + j) = CM(c22, r, i, j);
    /* :92-97 */
    double Gt[180], PG[180], Q[225];
    memcpy(Gt, G, sizeof Gt);
    for (int b = 0; b < 4; ++b)
        for (int j = 3 * b; j < 3 * b + 3; ++j) for (int i = 0; i < 15; ++i) CM(Gt, 15, i, j) *= sigma[b];
    gemm(15, 12, 15, Phi, 15, Gt, 15, 0, PG, 15, 0);
    gemm(15, 15, 12, PG, 15, PG, 15, 1, Q, 15, 0);
    for (int j = 0; j < 15; ++j) for (int i = 0; i < 15; ++i) CM(tmp, n, i, j) += dt * CM(Q, 15, i, j);
    /* :99-116 clock process noise */
    if (enable_gnss)
        for (int i = 0; i < 5; ++i) {
            if (gnss_idx[i] < 0) continue;
            for (int j = 0; j < 5; ++j) {
                if (gnss_idx[j] < 0) continue;
                double add;
                if (i != 4 && j != 4) add = dt * pow(sigma_cb, 2) + pow(dt, 3) * pow(sigma_rw, 2);
                else if (i == 4 && j == 4) add = dt * pow(sigma_rw, 2);
                else add = pow(dt, 2) * pow(sigma_rw, 2);
                CM(tmp, n, gnss_idx[i], gnss_idx[j]) += add;
            }
        }
    /* :118 */
    for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i)
        CM(P, ld, i, j) = 0.5 * (CM(tmp, n, i, j) + CM(tmp, n, j, i));
    free(tmp); free(c21); free(c22);
}

/* StateManager.cpp:279-293 */
void orc_augment_clone(double* P, int n, int ld, const double R_i2w[9])
{
    int nn = n + 6;
    double J[6 * 21];
    memset(J, 0, sizeof J);
    for (int i = 0; i < 6; ++i) CM(J, 6, i, i) = 1.0;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        CM(J, 6, i, 15 + j) = R_i2w[3 * i + j];
        CM(J, 6, 3 + i, 18 + j) = R_i2w[3 * i + j];
    }
    double* JP = dalloc((size_t)6 * n);          /* J * P[0:21, 0:n] */
    gemm(6, n, 21, J, 6, P, ld, 0, JP, 6, 0);
    double JPJ[36];
    gemm(6, 6, 21, JP, 6, J, 6, 1, JPJ, 6, 0);   /* (J P[0:21,0:21]) J^T */
    double* cn = dalloc((size_t)nn * nn);
    for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) CM(cn, nn, i, j) = CM(P, ld, i, j);
    for (int j = 0; j < n; ++j) for (int i = 0; i < 6; ++i) {
        CM(cn, nn, n + i, j) = CM(JP, 6, i, j);
        CM(cn, nn, j, n + i) = CM(JP, 6, i, j);
    }
    for (int j = 0; j < 6; ++j) for (int i = 0; i < 6; ++i) CM(cn, nn, n + i, n + j) = CM(JPJ, 6, i, j);
    for (int j = 0; j < nn; ++j) for (int i = 0; i < nn; ++i)
        CM(P, ld, i, j) = 0.5 * (CM(cn, nn, i, j) + CM(cn, nn, j, i));
    free(JP); free(cn);
}

/* StateManager.cpp:163-177 */
void orc_marginalize(double* P, int n, int ld, int idx, int size)
{
    int nn = n - size;
    double* t = dalloc((size_t)nn * nn);
    for (int j = 0; j < nn; ++j) {
        int sj = j < idx ? j : j + size;
        for (int i = 0; i < nn; ++i) {
            int si = i < idx ? i : i + size;
            CM(t, nn, i, j) = CM(P, ld, si, sj);
        }
    }
    for (int j = 0; j < nn; ++j) for (int i = 0; i < nn; ++i) CM(P, ld, i, j) = CM(t, nn, i, j);
    free(t);
}

/* StateManager.cpp:194-214 */
void orc_append_independent(double* P, int n, int ld, int size, const double* blk)
{
    for (int j = 0; j < size; ++j)
        for (int i = 0; i < n + size; ++i) { CM(P, ld, i, n + j) = 0.0; CM(P, ld, n + j, i) = 0.0; }
    for (int j = 0; j < size; ++j) for (int i = 0; i < size; ++i) CM(P, ld, n + i, n + j) = CM(blk, size, i, j);
}

/* StateManager.cpp:128-153 */
void orc_marginal_cov(const double* P, int ld, const int* vidx, const int* vsize, int k, double* out)
{
    int ns = 0;
    for (int i = 0; i < k; ++i) ns += vsize[i];
    int r0 = 0;
    for (int a = 0; a < k; ++a) {
        int c0 = 0;
        for (int b = 0; b < k; ++b) {
            for (int j = 0; j < vsize[b]; ++j) for (int i = 0; i < vsize[a]; ++i)
                CM(out, ns, r0 + i, c0 + j) = CM(P, ld, vidx[a] + i, vidx[b] + j);
            c0 += vsize[b];
        }
        r0 += vsize[a];
    }
}

static void add_R(double* S, int m, const double* R, int r_kind)
{
    if (r_kind == 0) for (int i = 0; i < m; ++i) CM(S, m, i, i) += R[0];
    else if (r_kind == 1) for (int i = 0; i < m; ++i) CM(S, m, i, i) += R[i];
    else for (int j = 0; j < m; ++j) for (int i = 0; i < m; ++i) CM(S, m, i, j) += CM(R, m, i, j);
}

/* S = H Pcc H^T + R (m x m) ; Pcc gathered as getMarginalCov does */
static double* form_S(const double* P, int ld, const int* vidx, const int* vsize, int k,
                      const double* H, int ldh, int m, const double* R, int r_kind, int* ns_out)
{
    int ns = 0;
    for (int i = 0; i < k; ++i) ns += vsize[i];
    double* Pcc = dalloc((size_t)ns * ns);
    orc_marginal_cov(P, ld, vidx, vsize, k, Pcc);
    double* T = dalloc((size_t)m * ns);
    gemm(m, ns, ns, H, ldh, Pcc, ns, 0, T, m, 0);
    double* S = dalloc((size_t)m * m);
    gemm(m, m, ns, T, m, H, ldh, 1, S, m, 0);
    add_R(S, m, R, r_kind);
    free(Pcc); free(T);
    if (ns_out) *ns_out = ns;
    return S;
}

/* Update.cpp:36-79 — S.ldlt().solve(res): LDL^T without pivoting (S is SPD), then the quadratic form */
double orc_whiten_residual(const double* P, int ld, const int* vidx, const int* vsize, int k,
                           const double* H, int ldh, int m, const double* res,
                           const double* R, int r_kind)
{
    double* S = form_S(P, ld, vidx, vsize, k, H, ldh, m, R, r_kind, NULL);
    double* d = dalloc(m);
    double* y = dalloc(m);
    /* in-place LDL^T: L strictly lower in S, D in d */
    for (int j = 0; j < m; ++j) {
        double dj = CM(S, m, j, j);
        for (int l = 0; l < j; ++l) dj -= CM(S, m, j, l) * CM(S, m, j, l) * d[l];
        d[j] = dj;
        for (int i = j + 1; i < m; ++i) {
            double s = CM(S, m, i, j);
            for (int l = 0; l < j; ++l) s -= CM(S, m, i, l) * CM(S, m, j, l) * d[l];
            CM(S, m, i, j) = s / dj;
        }
    }
    for (int i = 0; i < m; ++i) {
        double s = res[i];
        for (int l = 0; l < i; ++l) s -= CM(S, m, i, l) * y[l];
        y[i] = s;
    }
    double g = 0.0;
    for (int i = 0; i < m; ++i) g += y[i] * y[i] / d[i];
    free(S); free(d); free(y);
    return g;
}

/* partial-pivot LU inverse (what Eigen's MatrixXd::inverse() does for dynamic sizes) */
static int lu_inverse(double* A, int m, double* Ainv)
{
    int* piv = (int*)malloc(sizeof(int) * (m ? m : 1));
    for (int j = 0; j < m; ++j) {
        int p = j; double mx = fabs(CM(A, m, j, j));
        for (int i = j + 1; i < m; ++i) if (fabs(CM(A, m, i, j)) > mx) { mx = fabs(CM(A, m, i, j)); p = i; }
        piv[j] = p;
        if (p != j) for (int c = 0; c < m; ++c) { double t = CM(A, m, j, c); CM(A, m, j, c) = CM(A, m, p, c); CM(A, m, p, c) = t; }
        double d = CM(A, m, j, j);
        if (d == 0.0) { free(piv); return -1; }
        for (int i = j + 1; i < m; ++i) CM(A, m, i, j) /= d;
        for (int c = j + 1; c < m; ++c) {
            double a = CM(A, m, j, c);
            if (a == 0.0) continue;
            for (int i = j + 1; i < m; ++i) CM(A, m, i, c) -= CM(A, m, i, j) * a;
        }
    }
    /* solve A X = I column by column */
    for (int c = 0; c < m; ++c) {
        double* x = Ainv + (size_t)c * m;
        for (int i = 0; i < m; ++i) x[i] = (i == c) ? 1.0 : 0.0;
        for (int j = 0; j < m; ++j) if (piv[j] != j) { double t = x[j]; x[j] = x[piv[j]]; x[piv[j]] = t; }
        for (int j = 0; j < m; ++j) { double xj = x[j]; if (xj != 0.0) for (int i = j + 1; i < m; ++i) x[i] -= CM(A, m, i, j) * xj; }
        for (int j = m - 1; j >= 0; --j) { x[j] /= CM(A, m, j, j); double xj = x[j]; for (int i = 0; i < j; ++i) x[i] -= CM(A, m, i, j) * xj; }
    }
    free(piv);
    return 0;
}

/* StateManager.cpp:359-423 */
int orc_ekf_update(double* P, int n, int ld, const int* vidx, const int* vsize, int k,
                   const double* H, int ldh, int m, const double* res,
                   const double* R, int r_kind, double* dx)
{
    /* :381-397  PH^T = sum over measured variables of P[:, var] * H[:, var]^T */
    double* PHT = dalloc((size_t)n * m);
    int hc = 0;
    for (int a = 0; a < k; ++a) {
        gemm(n, m, vsize[a], P + (size_t)vidx[a] * ld, ld, H + (size_t)hc * ldh, ldh, 1, PHT, n, 1);
        hc += vsize[a];
    }
    /* :399-403 */
    double* S = form_S(P, ld, vidx, vsize, k, H, ldh, m, R, r_kind, NULL);
    /* :405  K = PH^T * S.inverse() */
    double* Sinv = dalloc((size_t)m * m);
    int rc = lu_inverse(S, m, Sinv);
    double* K = dalloc((size_t)n * m);
    gemm(n, m, m, PHT, n, Sinv, m, 0, K, n, 0);
    /* :407-411 */
    double* tmp = dalloc((size_t)n * n);
    gemm(n, n, m, K, n, PHT, n, 1, tmp, n, 0);
    for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) CM(tmp, n, i, j) = CM(P, ld, i, j) - CM(tmp, n, i, j);
    for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i)
        CM(P, ld, i, j) = 0.5 * (CM(tmp, n, i, j) + CM(tmp, n, j, i));
    int neg = 0;
    for (int i = 0; i < n; ++i) if (CM(P, ld, i, i) < 0.0) neg = 1;  /* :413-421 */
    /* :423 */
    for (int i = 0; i < n; ++i) { double s = 0.0; for (int l = 0; l < m; ++l) s += CM(K, n, i, l) * res[l]; dx[i] = s; }
    free(PHT); free(S); free(Sinv); free(K); free(tmp);
    return rc < 0 ? -1 : neg;
}

/* ------------------------------------------------------------------------------------------ */
/* Householder QR (stands in for JacobiSVD's full-U nullspace and for SPQR)                    */
/* ------------------------------------------------------------------------------------------ */
/* A (m x n) <- Q^T A, extra columns E (m x ne) <- Q^T E ; reflectors for min(m-1,n) columns */
static void house_qr_apply(double* A, int m, int n, int lda, double* E, int ne, int lde)
{
    double* v = dalloc(m);
    int steps = n < m - 1 ? n : m - 1;
    for (int kcol = 0; kcol < steps; ++kcol) {
        double nrm = 0.0;
        for (int i = kcol; i < m; ++i) nrm += CM(A, lda, i, kcol) * CM(A, lda, i, kcol);
        nrm = sqrt(nrm);
        if (nrm == 0.0) continue;
        double x0 = CM(A, lda, kcol, kcol);
        double alpha = x0 >= 0.0 ? -nrm : nrm;
        double v0 = x0 - alpha;
        v[kcol] = 1.0;
        for (int i = kcol + 1; i < m; ++i) v[i] = CM(A, lda, i, kcol) / v0;
        double tau = -v0 / alpha;           /* = 2 / (v^T v) with v0 normalised to 1 */
        CM(A, lda, kcol, kcol) = alpha;
        for (int i = kcol + 1; i < m; ++i) CM(A, lda, i, kcol) = 0.0;
        for (int c = kcol + 1; c < n; ++c) {
            double w = 0.0;
            for (int i = kcol; i < m; ++i) w += v[i] * CM(A, lda, i, c);
            w *= tau;
            for (int i = kcol; i < m; ++i) CM(A, lda, i, c) -= w * v[i];
        }
        for (int c = 0; c < ne; ++c) {
            double w = 0.0;
            for (int i = kcol; i < m; ++i) w += v[i] * CM(E, lde, i, c);
            w *= tau;
            for (int i = kcol; i < m; ++i) CM(E, lde, i, c) -= w * v[i];
        }
    }
    free(v);
}

void orc_qr_compress(double* A, int m, int n, int lda, double* b)
{
    house_qr_apply(A, m, n, lda, b, 1, m);
}

/* ------------------------------------------------------------------------------------------ */
/* MSCKF per-feature block                                                                     */
/* ------------------------------------------------------------------------------------------ */
static int has_nan(const double* x, int n) { for (int i = 0; i < n; ++i) if (isnan(x[i])) return 1; return 0; }

static void proj_jac(const double q[3], double Hp[6]) /* 2x3 row-major, RemoveLostUpdate.cpp:452-456 */
{
    Hp[0] = 1.0 / q[2]; Hp[1] = 0.0;        Hp[2] = -q[0] / pow(q[2], 2);
    Hp[3] = 0.0;        Hp[4] = 1.0 / q[2]; Hp[5] = -q[1] / pow(q[2], 2);
}

/* Builds the projected block of feature j in GLOBAL window-slot columns (6*C wide).
 * Returns rho (rows after nullspace) or 0 if unusable.  Hj: rho x 6C col-major (ldh), rj: rho. */
int orc_msckf_feature_block(const orc_msckf_in* in, int j, double* Hj, int ldh, double* rj)
{
    const int C = in->n_clones, nc = 6 * C;
    const int rpo = in->stereo ? 4 : 2;               /* rows per observation */
    const double* pf = in->pf + 3 * j;
    const int a = in->anchor[j];
    const unsigned long long mask = in->obs_mask[j];
    int nobs = 0;
    for (int s = 0; s < C; ++s) if (mask >> s & 1ULL) ++nobs;
    int maxr = rpo * nobs;
    if (maxr == 0) return 0;
    double* Hx = dalloc((size_t)maxr * nc);           /* H_block_tmp            */
    double* Ha = dalloc((size_t)maxr * 6);            /* H_anchor_block_tmp (selected variant) */
    double* Hf = dalloc((size_t)maxr * 3);            /* Hf_block_tmp           */
    double* r = dalloc(maxr);
    double Xf[9];
    orc_skew(pf, Xf);
    int row = 0;
    for (int s = 0; s < C; ++s) {
        if (!(mask >> s & 1ULL)) continue;
        const double* Rc = in->clone_R + 9 * s;
        const double* pc = in->clone_p + 3 * s;
        const double* z = in->uv + ((size_t)j * C + s) * 4;
        double d[3] = { pf[0] - pc[0], pf[1] - pc[1], pf[2] - pc[2] };
        double q[3], qr[3];
        m3_tmulv(Rc, d, q);                                        /* :448 */
        m3_mulv(in->R_cl2cr, q, qr);
        for (int i = 0; i < 3; ++i) qr[i] += in->t_cl2cr[i];       /* :450 */
        double Hp[6], Hpr[6];
        proj_jac(q, Hp); proj_jac(qr, Hpr);
        /* H_pf2x pieces (3x3 each, row-major): theta_obs, p_obs, theta_anchor */
        double Rt[9], E_th[9], E_p[9], E_an[9];
        m3_T(Rc, Rt);
        memset(E_th, 0, sizeof E_th); memset(E_an, 0, sizeof E_an);
        if (s != a) {
            m3_mul(Rt, Xf, E_th);                                  /* :478 */
            for (int i = 0; i < 9; ++i) E_an[i] = -E_th[i];        /* :479 */
        }
        for (int i = 0; i < 9; ++i) E_p[i] = -Rt[i];               /* :482 */
        if (has_nan(Hp, 6) || has_nan(E_th, 9) || has_nan(E_p, 9) || has_nan(E_an, 9))
            continue;                                              /* :486-495 */
        /* left rows */
        double RlR[9];
        m3_mul(in->R_cl2cr, Rt, RlR);                              /* R_cl2cr * R^T  (H_pf2pf_r) */
        for (int half = 0; half < (in->stereo ? 2 : 1); ++half) {
            const double* HP = half ? Hpr : Hp;
            double Mth[9], Mp[9], Man[9], Mf[9];
            if (half) { m3_mul(in->R_cl2cr, E_th, Mth); m3_mul(in->R_cl2cr, E_p, Mp); m3_mul(in->R_cl2cr, E_an, Man); memcpy(Mf, RlR, sizeof Mf); }
            else { memcpy(Mth, E_th, sizeof Mth); memcpy(Mp, E_p, sizeof Mp); memcpy(Man, E_an, sizeof Man); memcpy(Mf, Rt, sizeof Mf); }
            for (int rr = 0; rr < 2; ++rr) {
                int ri = row + 2 * half + rr;
                for (int c = 0; c < 3; ++c) {
                    double vth = HP[3 * rr] * Mth[c] + HP[3 * rr + 1] * Mth[3 + c] + HP[3 * rr + 2] * Mth[6 + c];
                    double vp  = HP[3 * rr] * Mp[c]  + HP[3 * rr + 1] * Mp[3 + c]  + HP[3 * rr + 2] * Mp[6 + c];
                    double van = HP[3 * rr] * Man[c] + HP[3 * rr + 1] * Man[3 + c] + HP[3 * rr + 2] * Man[6 + c];
                    double vf  = HP[3 * rr] * Mf[c]  + HP[3 * rr + 1] * Mf[3 + c]  + HP[3 * rr + 2] * Mf[6 + c];
                    CM(Hx, maxr, ri, 6 * s + c) = vth;
                    CM(Hx, maxr, ri, 6 * s + 3 + c) = vp;
                    if (in->selected_variant) CM(Ha, maxr, ri, c) = van;         /* SwMargUpdate.cpp:651,666 */
                    else if (s != a) CM(Hx, maxr, ri, 6 * a + c) = van;          /* RemoveLostUpdate.cpp:479 */
                    CM(Hf, maxr, ri, c) = vf;
                }
            }
        }
        r[row + 0] = z[0] - q[0] / q[2];
        r[row + 1] = z[1] - q[1] / q[2];
        if (in->stereo) { r[row + 2] = z[2] - qr[0] / qr[2]; r[row + 3] = z[3] - qr[1] / qr[2]; }   /* :503 */
        row += rpo;
    }
    int rho = row - 3;
    if (rho <= 0) { free(Hx); free(Ha); free(Hf); free(r); return 0; }
    /* compact to `row` rows: matrices were allocated with ld = maxr, only the first `row` rows used */
    /* nullspace: V = last row-3 columns of the full Q of Hf  (:518-522, JacobiSVD full U there) */
    int ne = nc + 6 + 1;
    double* E = dalloc((size_t)row * ne);
    for (int c = 0; c < nc; ++c) for (int i = 0; i < row; ++i) CM(E, row, i, c) = CM(Hx, maxr, i, c);
    for (int c = 0; c < 6; ++c) for (int i = 0; i < row; ++i) CM(E, row, i, nc + c) = CM(Ha, maxr, i, c);
    for (int i = 0; i < row; ++i) CM(E, row, i, nc + 6) = r[i];
    double* Hf2 = dalloc((size_t)row * 3);
    for (int c = 0; c < 3; ++c) for (int i = 0; i < row; ++i) CM(Hf2, row, i, c) = CM(Hf, maxr, i, c);
    house_qr_apply(Hf2, row, 3, row, E, ne, row);
    for (int c = 0; c < nc; ++c) for (int i = 0; i < rho; ++i) CM(Hj, ldh, i, c) = CM(E, row, 3 + i, c);
    if (in->selected_variant)                     /* SwMargUpdate.cpp:302: assignment, not += (Q10) */
        for (int c = 0; c < 6; ++c) for (int i = 0; i < rho; ++i) CM(Hj, ldh, i, 6 * a + c) = CM(E, row, 3 + i, nc + c);
    for (int i = 0; i < rho; ++i) rj[i] = CM(E, row, 3 + i, nc + 6);
    free(Hx); free(Ha); free(Hf); free(r); free(E); free(Hf2);
    return rho;
}

/* RemoveLostUpdate.cpp:276-405 / SwMargUpdate.cpp:216-365 / KeyframeUpdate.cpp:587-735 on flat inputs */
int orc_msckf_update(double* P, int n, int ld, const orc_msckf_in* in,
                     double* dx, int* accepted, double* gamma)
{
    const int C = in->n_clones, F = in->n_feat, nc = 6 * C;
    const int rpo = in->stereo ? 4 : 2;
    for (int i = 0; i < n; ++i) dx[i] = 0.0;
    int maxrows = 0;
    for (int j = 0; j < F; ++j) { accepted[j] = 0; if (gamma) gamma[j] = NAN; maxrows += rpo * C; }
    if (F == 0) return 0;
    int* vidx = (int*)malloc(sizeof(int) * C);
    int* vsize = (int*)malloc(sizeof(int) * C);
    for (int s = 0; s < C; ++s) { vidx[s] = in->clone_idx[s]; vsize[s] = 6; }
    double* HL = dalloc((size_t)maxrows * nc);
    double* rL = dalloc(maxrows);
    int ldb = rpo * C;
    double* Hj = dalloc((size_t)ldb * nc);
    double* rj = dalloc(ldb);
    int rows = 0, nacc = 0;
    double var = in->noise * in->noise;
    /* column use: which window slots appear in an accepted block (sw_index_map, :338-348) */
    int* used = (int*)calloc(C, sizeof(int));
    for (int j = 0; j < F; ++j) {
        memset(Hj, 0, sizeof(double) * ldb * nc);
        int rho = orc_msckf_feature_block(in, j, Hj, ldb, rj);
        if (rho <= 0) continue;
        /* Update.cpp:104-124 with the block's own variables; zero columns change nothing, so the
         * gather is done over all window clones (global slot order) */
        double g = orc_whiten_residual(P, ld, vidx, vsize, C, Hj, ldb, rho, rj, &var, 0);
        if (gamma) gamma[j] = g;
        int dof = in->dof[j];
        if (dof < 1 || dof >= in->chi2_len || !(g < in->chi2_table[dof])) continue;
        for (int c = 0; c < nc; ++c) for (int i = 0; i < rho; ++i) CM(HL, maxrows, rows + i, c) = CM(Hj, ldb, i, c);
        for (int i = 0; i < rho; ++i) rL[rows + i] = rj[i];
        for (int s = 0; s < C; ++s) {
            if (in->obs_mask[j] >> s & 1ULL) used[s] = 1;
        }
        used[in->anchor[j]] = 1;
        rows += rho;
        accepted[j] = 1;
        ++nacc;
        if (in->max_accept > 0 && nacc >= in->max_accept) break;   /* RemoveLostUpdate.cpp:359 */
    }
    int mout = 0;
    if (rows > 0) {
        /* drop never-used clone columns (conservativeResize to col_cnt, :367-371) */
        int kk = 0, ncol = 0;
        int* vi = (int*)malloc(sizeof(int) * C); int* vs = (int*)malloc(sizeof(int) * C);
        double* Hc = dalloc((size_t)rows * nc);
        for (int s = 0; s < C; ++s) if (used[s]) {
            vi[kk] = in->clone_idx[s]; vs[kk] = 6; ++kk;
            for (int c = 0; c < 6; ++c) for (int i = 0; i < rows; ++i) CM(Hc, rows, i, ncol + c) = CM(HL, maxrows, i, 6 * s + c);
            ncol += 6;
        }
        double* rc = dalloc(rows);
        memcpy(rc, rL, sizeof(double) * rows);
        int m = rows;
        if (rows > ncol) {                                         /* :376-392 */
            house_qr_apply(Hc, rows, ncol, rows, rc, 1, rows);
            m = in->compress_rule == 1 ? ncol : rows;              /* Q2: RemoveLost keeps all rows */
        }
        orc_ekf_update(P, n, ld, vi, vs, kk, Hc, rows, m, rc, &var, 0, dx);
        mout = m;
        free(vi); free(vs); free(Hc); free(rc);
    }
    free(vidx); free(vsize); free(HL); free(rL); free(Hj); free(rj); free(used);
    return mout;
}

/* ------------------------------------------------------------------------------------------ */
/* GNSS rows: GnssUpdate.cpp:148-272                                                           */
/* ------------------------------------------------------------------------------------------ */
int orc_gnss_rows(const double* P, int ld, const orc_gnss_in* in,
                  double* H, int ldh, double* res, double* Rdiag,
                  int* vidx, int* vsize, int* nvar)
{
    int rows = 0, col_cnt = 10, nv = 0;
    int cb_col[4] = { -1, -1, -1, -1 };
    vidx[nv] = in->idx_se23; vsize[nv++] = 9;
    vidx[nv] = in->idx_yof;  vsize[nv++] = 1;
    int maxc = 9 + 6;
    for (int c = 0; c < maxc; ++c) for (int i = 0; i < 2 * in->nsat; ++i) CM(H, ldh, i, c) = 0.0;
    double Sp[9], Sv[9], M[9];
    orc_skew(in->p_w, Sp); orc_skew(in->v_w, Sv);
    for (int i = 0; i < in->nsat; ++i) {
        int g = in->sys[i];
        if (g < 0 || g > 3 || in->idx_cb[g] < 0) continue;                       /* :153 */
        const double* u = in->los + 3 * i;
        double uR[3], h9[9] = { 0 };
        m3_tmulv(in->R_w2ecef, u, uR);                                           /* u^T R_w2ecef */
        m3_mul(in->R_w2ecef, Sp, M);
        for (int c = 0; c < 3; ++c) { h9[c] = u[0] * M[c] + u[1] * M[3 + c] + u[2] * M[6 + c]; h9[3 + c] = -uR[c]; }  /* :161-162 */
        double sin_el = in->sin_el[i];
        if (fabs(sin_el) < 1e-6) sin_el = 1e-6;
        double noise = in->psr_amp * pow(in->ura[i] * in->psr_std[i] / (sin_el * sin_el), 0.5);  /* :187 */
        double r_i = -in->res_pos[i];
        if (in->chi2_test) {                                                     /* :190 */
            double Hi[11]; int svi[3] = { in->idx_se23, in->idx_yof, in->idx_cb[g] }, svs[3] = { 9, 1, 1 };
            for (int c = 0; c < 9; ++c) Hi[c] = h9[c];
            Hi[9] = 0.0; Hi[10] = 1.0;
            double var = noise * noise;
            double gm = orc_whiten_residual(P, ld, svi, svs, 3, Hi, 1, 1, &r_i, &var, 0);
            if (!(gm < in->chi2_table[1])) continue;
        }
        res[rows] = r_i; Rdiag[rows] = noise * noise;
        for (int c = 0; c < 9; ++c) CM(H, ldh, rows, c) = h9[c];
        if (cb_col[g] < 0) { cb_col[g] = col_cnt; col_cnt += 1; vidx[nv] = in->idx_cb[g]; vsize[nv++] = 1; }   /* :200-206 */
        CM(H, ldh, rows, cb_col[g]) = 1.0;
        ++rows;
    }
    int fs_col = col_cnt; col_cnt += 1;                                          /* :213-218 */
    vidx[nv] = in->idx_fs; vsize[nv++] = 1;
    for (int i = 0; i < in->nsat; ++i) {
        int g = in->sys[i];
        if (g < 0 || g > 3 || in->idx_cb[g] < 0) continue;                       /* :225 */
        const double* u = in->los + 3 * i;
        double uR[3], h9[9] = { 0 };
        m3_tmulv(in->R_w2ecef, u, uR);
        m3_mul(in->R_w2ecef, Sv, M);
        for (int c = 0; c < 3; ++c) { h9[c] = u[0] * M[c] + u[1] * M[3 + c] + u[2] * M[6 + c]; h9[6 + c] = -uR[c]; }  /* :236-237 */
        double sin_el = in->sin_el[i];
        if (fabs(sin_el) < 1e-6) sin_el = 1e-6;
        double noise = in->dopp_amp * pow(in->ura[i] * in->dopp_std_mps[i] / (sin_el * sin_el), 0.5);  /* :256 */
        double r_i = -in->res_vel[i];
        if (in->chi2_test) {                                                     /* :259 */
            double Hi[11]; int svi[3] = { in->idx_se23, in->idx_yof, in->idx_fs }, svs[3] = { 9, 1, 1 };
            for (int c = 0; c < 9; ++c) Hi[c] = h9[c];
            Hi[9] = 0.0; Hi[10] = 1.0;
            double var = noise * noise;
            double gm = orc_whiten_residual(P, ld, svi, svs, 3, Hi, 1, 1, &r_i, &var, 0);
            if (!(gm < in->chi2_table[1])) continue;
        }
        res[rows] = r_i; Rdiag[rows] = noise * noise;
        for (int c = 0; c < 9; ++c) CM(H, ldh, rows, c) = h9[c];
        CM(H, ldh, rows, fs_col) = 1.0;
        ++rows;
    }
    *nvar = nv;
    return rows;
}

/* ------------------------------------------------------------------------------------------ */
/* benchmark unit: SURVEY.md 8(d) "one update"                                                 */
/* ------------------------------------------------------------------------------------------ */
int orc_frame_update(double* P, int* n, int ld, const orc_frame_in* fr, const orc_msckf_in* ms,
                     double* dx, int* accepted, double* gamma)
{
    for (int s = 0; s < fr->k; ++s)
        orc_propagate_cov(P, *n, ld, fr->Phi + 225 * s, fr->G + 180 * s, fr->dt[s], fr->sigma,
                          fr->enable_gnss, fr->gnss_idx, fr->sigma_cb, fr->sigma_rw);
    orc_augment_clone(P, *n, ld, fr->R_i2w);
    *n += 6;
    int m = orc_msckf_update(P, *n, ld, ms, dx, accepted, gamma);
    if (fr->marg_idx >= 0) { orc_marginalize(P, *n, ld, fr->marg_idx, 6); *n -= 6; }
    return m;
}

void orc_frame_update_batch(int B, int threads, double* P, int* n_io, int ld,
                            const orc_frame_in* fr, const orc_msckf_in* ms,
                            double* dx, int* accepted, int fmax)
{
    /* the restatement allocates its N x N temporaries per call like the reference's Eigen code does;
     * keep them in the per-thread malloc arenas instead of mmap/munmap so that many threads do not
     * serialise in the kernel (a fair multi-core baseline) */
    mallopt(M_MMAP_THRESHOLD, 1 << 30);
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(dynamic, 1)
#endif
    for (int b = 0; b < B; ++b)
        orc_frame_update(P + (size_t)b * ld * ld, n_io + b, ld, fr + b, ms + b,
                         dx + (size_t)b * ld, accepted + (size_t)b * fmax, NULL);
}

/* ------------------------------------------------------------------------------------------ */
/* f-1: triangulation (Triangulator.cpp)                                                        */
/* ------------------------------------------------------------------------------------------ */
typedef struct { double R[9], p[3], m[2]; } tri_obs;      /* camera pose in the world + its normalised measurement */

/* rel = T_i^-1 * T_last (calcRelaSwPose, :70-88): maps last-camera coordinates into camera i */
static void tri_rel(const tri_obs* oi, const tri_obs* ol, double Rr[9], double tr[3])
{
    double Rt[9];
    m3_T(oi->R, Rt);
    m3_mul(Rt, ol->R, Rr);
    const double d[3] = { ol->p[0] - oi->p[0], ol->p[1] - oi->p[1], ol->p[2] - oi->p[2] };
    m3_mulv(Rt, d, tr);
}

/* calcUnitCost (:107-125) */
static double tri_unit_cost(const double m[2], const double Rr[9], const double tr[3], const double sol[3])
{
    double pf0[3], pf[3];
    pf0[2] = 1.0 / sol[2];
    pf0[0] = sol[0] * pf0[2];
    pf0[1] = sol[1] * pf0[2];
    m3_mulv(Rr, pf0, pf);
    pf[0] += tr[0]; pf[1] += tr[1]; pf[2] += tr[2];
    const double ex = m[0] - pf[0] / pf[2], ey = m[1] - pf[1] / pf[2];
    return ex * ex + ey * ey;
}

static double tri_total_cost(const tri_obs* ob, int n, const double sol[3])      /* :127-137 */
{
    double c = 0.0;
    for (int i = 0; i < n; ++i) {
        double Rr[9], tr[3];
        if (i == n - 1) { m3_eye(Rr); tr[0] = tr[1] = tr[2] = 0.0; } else tri_rel(&ob[i], &ob[n - 1], Rr, tr);
        c += tri_unit_cost(ob[i].m, Rr, tr, sol);
    }
    return c;
}

/* solve (A + lambda I) x = b for symmetric 3x3 (the reference uses Eigen's ldlt, :232) */
static void tri_solve3(const double A[9], double lambda, const double b[3], double x[3])
{
    const double a00 = A[0] + lambda, a10 = A[3], a20 = A[6], a11 = A[4] + lambda, a21 = A[7], a22 = A[8] + lambda;
    const double l10 = a10 / a00, l20 = a20 / a00;
    const double d1 = a11 - l10 * a10;
    const double l21 = (a21 - l20 * a10) / d1;
    const double d2 = a22 - l20 * a20 - l21 * (a21 - l20 * a10);
    const double y0 = b[0], y1 = b[1] - l10 * y0, y2 = b[2] - l20 * y0 - l21 * y1;
    x[2] = y2 / d2;
    x[1] = y1 / d1 - l21 * x[2];
    x[0] = y0 / a00 - l10 * x[1] - l20 * x[2];
}

int orc_triangulate(const orc_tri_in* in, double pf[3])
{
    tri_obs ob[128];
    int n = 0;
    pf[0] = pf[1] = pf[2] = 0.0;
    /* mono-equivalent observations in time order; a stereo pair contributes left, then right with the right camera's
     * pose T_left * T_cl2cr^-1 (:331-354) */
    double Rlr_t[9], tinv[3];
    m3_T(in->R_lr, Rlr_t);
    m3_mulv(Rlr_t, in->t_lr, tinv);
    tinv[0] = -tinv[0]; tinv[1] = -tinv[1]; tinv[2] = -tinv[2];
    for (int s = 0; s < in->C && n + 2 <= 128; ++s) {
        if (!((in->mask >> s) & 1ULL)) continue;
        const double* R = in->clone_R + 9 * s; const double* p = in->clone_p + 3 * s; const double* z = in->uv + 4 * s;
        memcpy(ob[n].R, R, sizeof ob[n].R); memcpy(ob[n].p, p, sizeof ob[n].p);
        ob[n].m[0] = z[0]; ob[n].m[1] = z[1];
        ++n;
        if (in->stereo) {
            double t[3];
            m3_mul(R, Rlr_t, ob[n].R);
            m3_mulv(R, tinv, t);
            ob[n].p[0] = p[0] + t[0]; ob[n].p[1] = p[1] + t[1]; ob[n].p[2] = p[2] + t[2];
            ob[n].m[0] = z[2]; ob[n].m[1] = z[3];
            ++n;
        }
    }
    if (n <= 4) return 0;                                                         /* :183-187 */
    const tri_obs* last = &ob[n - 1];
    /* findLongestTrans (:31-68) */
    double fl[3] = { last->m[0], last->m[1], 1.0 };
    const double fn = v3_norm(fl);
    fl[0] /= fn; fl[1] /= fn; fl[2] /= fn;
    double fw[3];
    m3_mulv(last->R, fl, fw);
    double max_trans = -INFINITY;
    int imax = n - 1;
    for (int i = 0; i < n - 1; ++i) {
        const double d[3] = { ob[i].p[0] - last->p[0], ob[i].p[1] - last->p[1], ob[i].p[2] - last->p[2] };
        const double dot = fw[0] * d[0] + fw[1] * d[1] + fw[2] * d[2];
        const double q[3] = { d[0] - fw[0] * dot, d[1] - fw[1] * dot, d[2] - fw[2] * dot };
        const double tr = fabs(v3_norm(q));
        if (tr > max_trans) { max_trans = tr; imax = i; }
    }
    if (max_trans < in->trans_thres) return 0;                                    /* :192 */
    /* initial guess (:201-203, initDepth :90-105) */
    double sol[3];
    {
        double Rr[9], tr[3], m1[3] = { last->m[0], last->m[1], 1.0 }, tm[3];
        tri_rel(&ob[imax], last, Rr, tr);
        m3_mulv(Rr, m1, tm);
        const double* m2 = ob[imax].m;
        const double A0 = tm[0] - m2[0] * tm[2], A1 = tm[1] - m2[1] * tm[2];
        const double b0 = m2[0] * tr[2] - tr[0], b1 = m2[1] * tr[2] - tr[1];
        const double depth = (A0 * b0 + A1 * b1) / (A0 * A0 + A1 * A1);
        sol[0] = last->m[0]; sol[1] = last->m[1]; sol[2] = 1.0 / depth;
    }
    double total_cost = tri_total_cost(ob, n, sol);
    double lambda = in->init_damping, delta_norm = INFINITY;
    int inner = 0, outer = 0, reduced = 0;
    do {                                                                           /* :215-262 */
        double A[9] = { 0 }, b[3] = { 0 };
        for (int i = 0; i < n; ++i) {                                              /* calcResJacobian :139-171 */
            double Rr[9], tr[3];
            if (i == n - 1) { m3_eye(Rr); tr[0] = tr[1] = tr[2] = 0.0; } else tri_rel(&ob[i], last, Rr, tr);
            const double a[3] = { sol[0], sol[1], 1.0 };
            double h[3];
            m3_mulv(Rr, a, h);
            h[0] += tr[0] * sol[2]; h[1] += tr[1] * sol[2]; h[2] += tr[2] * sol[2];
            const double res[2] = { h[0] / h[2] - ob[i].m[0], h[1] / h[2] - ob[i].m[1] };
            const double W00 = 1.0 / h[2], W02 = -h[0] / (h[2] * h[2]), W12 = -h[1] / (h[2] * h[2]);
            /* U = [Rr(:,0:2) | tr],  J = W U (2x3) */
            double J[6];
            for (int c = 0; c < 3; ++c) {
                const double u0 = c < 2 ? Rr[c] : tr[0], u1 = c < 2 ? Rr[3 + c] : tr[1], u2 = c < 2 ? Rr[6 + c] : tr[2];
                J[c] = W00 * u0 + W02 * u2;
                J[3 + c] = W00 * u1 + W12 * u2;
            }
            const double e = sqrt(res[0] * res[0] + res[1] * res[1]);
            const double w = e <= in->huber_epsilon ? 1.0 : sqrt(2.0 * in->huber_epsilon / e);
            const double w2 = w == 1.0 ? 1.0 : w * w;
            for (int r = 0; r < 3; ++r) {
                for (int c = 0; c < 3; ++c) A[3 * r + c] += w2 * (J[r] * J[c] + J[3 + r] * J[3 + c]);
                b[r] -= w2 * (J[r] * res[0] + J[3 + r] * res[1]);
            }
        }
        do {
            double delta[3], ns[3];
            tri_solve3(A, lambda, b, delta);
            ns[0] = sol[0] + delta[0]; ns[1] = sol[1] + delta[1]; ns[2] = sol[2] + delta[2];
            delta_norm = v3_norm(delta);
            const double nc = tri_total_cost(ob, n, ns);
            if (nc < total_cost) {
                total_cost = nc; sol[0] = ns[0]; sol[1] = ns[1]; sol[2] = ns[2]; reduced = 1;
                lambda = lambda / 10.0 > 1e-10 ? lambda / 10.0 : 1e-10;
            } else {
                reduced = 0;
                lambda = lambda * 10 < 1e12 ? lambda * 10 : 1e12;
            }
        } while (inner++ < in->inner_loop_max_iter && !reduced);
        inner = 0;                                                                 /* :259: the counter is reset, so the
                                                                                      "both loops exhausted" test below can
                                                                                      never fire; kept as written */
    } while (outer++ < in->outer_loop_max_iter && delta_norm > in->conv_precision);
    double pl[3];
    pl[2] = 1.0 / sol[2]; pl[0] = sol[0] * pl[2]; pl[1] = sol[1] * pl[2];
    if ((outer >= in->outer_loop_max_iter && inner >= in->inner_loop_max_iter) || delta_norm > in->conv_precision) return 0;
    for (int i = 0; i < n; ++i) {                                                  /* :273-278 */
        double Rr[9], tr[3], t[3];
        if (i == n - 1) { m3_eye(Rr); tr[0] = tr[1] = tr[2] = 0.0; } else tri_rel(&ob[i], last, Rr, tr);
        m3_mulv(Rr, pl, t);
        if (t[2] + tr[2] <= in->min_depth) return 0;
    }
    if (pl[2] < in->min_depth || pl[2] > in->max_depth) return 0;                  /* :296-297 */
    double w[3];
    m3_mulv(last->R, pl, w);
    w[0] += last->p[0]; w[1] += last->p[1]; w[2] += last->p[2];
    if (w[0] != w[0] || w[1] != w[1] || w[2] != w[2]) return 0;
    pf[0] = w[0]; pf[1] = w[1]; pf[2] = w[2];
    return 1;
}

/* ========================================================================================== */
/* SLAM-landmark path                                                  */
/*   StateManager::addVariableDelayedInvertible (StateManager.cpp:461-543)                     */
/*   StateManager::addVariableDelayed           (:549-637)                                     */
/*   StateManager::replaceVarLinear             (:639-693)                                     */
/*   LandmarkUpdate::calcResJacobianSingleLandmark{Mono,Stereo} (LandmarkUpdate.cpp:521-626,   */
/*   :628-686 and the sliding-window-pose twins)                                               */
/* ========================================================================================== */
void orc_add_variable_delayed_invertible(double* P, int n, int ld, const int* vidx, const int* vsize, int k,
                                         const double* H_old, int ldh, const double* H_new, int ldn, int s, double noise)
{
    /* :490-505  PH^T over the measured variables */
    double* PHT = dalloc((size_t)n * s);
    int hc = 0;
    for (int a = 0; a < k; ++a) {
        gemm(n, s, vsize[a], P + (size_t)vidx[a] * ld, ld, H_old + (size_t)hc * ldh, ldh, 1, PHT, n, 1);
        hc += vsize[a];
    }
    /* :507-514  S = H small_cov H^T + noise^2 I */
    double var = noise * noise;
    double* S = form_S(P, ld, vidx, vsize, k, H_old, ldh, s, &var, 0, NULL);
    /* :516  H_new^-1 */
    double* Hn = dalloc((size_t)s * s);
    double* Hinv = dalloc((size_t)s * s);
    for (int j = 0; j < s; ++j) for (int i = 0; i < s; ++i) CM(Hn, s, i, j) = CM(H_new, ldn, i, j);
    lu_inverse(Hn, s, Hinv);
    /* :518  cov_newnew = Hinv S Hinv^T */
    double* T = dalloc((size_t)s * s);
    double* Cnn = dalloc((size_t)s * s);
    gemm(s, s, s, Hinv, s, S, s, 0, T, s, 0);
    gemm(s, s, s, T, s, Hinv, s, 1, Cnn, s, 0);
    /* :526  cross = -PH^T Hinv^T */
    double* X = dalloc((size_t)n * s);
    gemm(n, s, s, PHT, n, Hinv, s, 1, X, n, 0);
    for (int j = 0; j < s; ++j) {
        for (int i = 0; i < n; ++i) { CM(P, ld, i, n + j) = -CM(X, n, i, j); CM(P, ld, n + j, i) = -CM(X, n, i, j); }
        for (int i = 0; i < s; ++i) CM(P, ld, n + i, n + j) = CM(Cnn, s, i, j);
    }
    symmetrize(P, n + s, ld);                                           /* :534 */
    free(PHT); free(S); free(Hn); free(Hinv); free(T); free(Cnn); free(X);
}

/* Eigen::JacobiRotation::makeGivens for reals (Jacobi.h) */
static void make_givens(double p, double q, double* c, double* s)
{
    if (q == 0.0) { *c = p < 0.0 ? -1.0 : 1.0; *s = 0.0; }
    else if (p == 0.0) { *c = 0.0; *s = q < 0.0 ? 1.0 : -1.0; }
    else if (fabs(p) > fabs(q)) {
        double t = q / p, u = sqrt(1.0 + t * t);
        if (p < 0.0) u = -u;
        *c = 1.0 / u; *s = -t * *c;
    } else {
        double t = p / q, u = sqrt(1.0 + t * t);
        if (q < 0.0) u = -u;
        *s = -1.0 / u; *c = -t * *s;
    }
}

/* rows (r-1, r) of A(:, c0..c1) <- G^* applied on the left: x' = c x - s y, y' = s x + c y */
static void rot_rows(double* A, int lda, int r, int c0, int c1, double c, double s)
{
    for (int j = c0; j < c1; ++j) {
        double x = CM(A, lda, r - 1, j), y = CM(A, lda, r, j);
        CM(A, lda, r - 1, j) = c * x - s * y;
        CM(A, lda, r, j) = s * x + c * y;
    }
}

int orc_add_variable_delayed(double* P, int* n_io, int ld, const int* vidx, const int* vsize, int k,
                             double* H_old, int ldh, double* H_new, int ldn, int m, int s, double* res,
                             double noise, double chi2_mult, int do_chi2, double chi2_check, double* dx, double* chi2_out)
{
    int n = *n_io, nc = 0;
    for (int a = 0; a < k; ++a) nc += vsize[a];
    if (m <= s) return 0;                                               /* :571-575 */
    /* :577-589 Givens: H_new -> upper triangular, same rotations on res and H_old */
    for (int col = 0; col < s; ++col)
        for (int r = m - 1; r > col; --r) {
            double c, sn;
            make_givens(CM(H_new, ldn, r - 1, col), CM(H_new, ldn, r, col), &c, &sn);
            rot_rows(H_new, ldn, r, col, s, c, sn);
            rot_rows(res, m > 0 ? m : 1, r, 0, 1, c, sn);
            rot_rows(H_old, ldh, r, 0, nc, c, sn);
        }
    /* :591-599 split; :601-608 chi2 on the lower part */
    const int mu = m - s;
    double var = noise * noise;
    double chi2 = orc_whiten_residual(P, ld, vidx, vsize, k, H_old + s, ldh, mu, res + s, &var, 0);
    if (chi2_out) *chi2_out = chi2;
    if (chi2 > chi2_mult * chi2_check && do_chi2) return 0;             /* :614-618 */
    orc_add_variable_delayed_invertible(P, n, ld, vidx, vsize, k, H_old, ldh, H_new, ldn, s, noise);
    *n_io = n + s;
    for (int i = 0; i < n + s; ++i) dx[i] = 0.0;
    if (mu > 0) orc_ekf_update(P, n + s, ld, vidx, vsize, k, H_old + s, ldh, mu, res + s, &var, 0, dx);   /* :623-624 */
    return 1;
}

void orc_replace_var_linear(double* P, int n, int ld, int tidx, int tsize, const int* vidx, const int* vsize, int k,
                            const double* H, int ldh)
{
    double* PHT = dalloc((size_t)n * tsize);
    int hc = 0;
    for (int a = 0; a < k; ++a) {
        gemm(n, tsize, vsize[a], P + (size_t)vidx[a] * ld, ld, H + (size_t)hc * ldh, ldh, 1, PHT, n, 1);
        hc += vsize[a];
    }
    double zero = 0.0;
    double* HPH = form_S(P, ld, vidx, vsize, k, H, ldh, tsize, &zero, 0, NULL);          /* :683-685 */
    for (int j = 0; j < tsize; ++j) for (int i = 0; i < n; ++i) CM(P, ld, i, tidx + j) = CM(PHT, n, i, j);      /* :687 */
    for (int j = 0; j < tsize; ++j) for (int i = 0; i < n; ++i) CM(P, ld, tidx + j, i) = CM(PHT, n, i, j);      /* :689 */
    for (int j = 0; j < tsize; ++j) for (int i = 0; i < tsize; ++i) CM(P, ld, tidx + i, tidx + j) = CM(HPH, tsize, i, j);   /* :691 */
    free(PHT); free(HPH);
}

/* H (rows x 24 col-major, ld = 4): [epose 9 | ext 6 | anchor 6 | pf 3]; rows = 2 (mono) / 4 (stereo).
 * LandmarkUpdate.cpp:521-572 (mono), :628-686 (stereo).  Quirk Q12 (as written, :682): the right-camera rows of the
 * anchor block are -H_proj_r * R_cl2cr * skew(pf_w), without the R_w2cl factor the left rows carry. */
int orc_landmark_rows_epose(const double R_i2w[9], const double p_i2w[3], const double R_cl2i[9], const double p_c2i[3],
                            const double pf[3], const double* uv, int stereo, const double R_lr[9], const double t_lr[3],
                            double* H, double* res)
{
    const int rows = stereo ? 4 : 2;
    double d[3], pf_i[3], d2[3], pf_cl[3], pf_cr[3];
    for (int i = 0; i < 3; ++i) d[i] = pf[i] - p_i2w[i];
    m3_tmulv(R_i2w, d, pf_i);
    for (int i = 0; i < 3; ++i) d2[i] = pf_i[i] - p_c2i[i];
    m3_tmulv(R_cl2i, d2, pf_cl);
    double RiT[9], RcT[9], Rw2cl[9], Sk[9], Ski[9];
    m3_T(R_i2w, RiT); m3_T(R_cl2i, RcT);
    m3_mul(RcT, RiT, Rw2cl);
    orc_skew(pf, Sk); orc_skew(pf_i, Ski);
    memset(H, 0, sizeof(double) * 4 * 24);
    for (int eye = 0; eye < (stereo ? 2 : 1); ++eye) {
        double q[3], Hp[6], L[9];
        if (eye == 0) { memcpy(q, pf_cl, sizeof q); m3_eye(L); }
        else {
            m3_mulv(R_lr, pf_cl, pf_cr);
            for (int i = 0; i < 3; ++i) pf_cr[i] += t_lr[i];
            memcpy(q, pf_cr, sizeof q); memcpy(L, R_lr, sizeof L);
        }
        proj_jac(q, Hp);
        res[2 * eye] = uv[2 * eye] - q[0] / q[2];
        res[2 * eye + 1] = uv[2 * eye + 1] - q[1] / q[2];
        double HL[6];                                   /* H_proj * L (2x3 row-major) */
        for (int r = 0; r < 2; ++r) for (int c = 0; c < 3; ++c)
            HL[3 * r + c] = Hp[3 * r] * L[c] + Hp[3 * r + 1] * L[3 + c] + Hp[3 * r + 2] * L[6 + c];
        double A[6], B[6], Cc[6], D[6], E[6];
        for (int r = 0; r < 2; ++r) for (int c = 0; c < 3; ++c) {
            double a = 0, b = 0, cc = 0, dd = 0, e = 0;
            for (int l = 0; l < 3; ++l) {
                a += HL[3 * r + l] * Rw2cl[3 * l + c];          /* HL R_w2cl            */
                cc += HL[3 * r + l] * RcT[3 * l + c];           /* HL R_cl2i^T          */
                e += HL[3 * r + l] * Sk[3 * l + c];             /* HL skew(pf_w)  (Q12) */
            }
            A[3 * r + c] = a; Cc[3 * r + c] = cc; E[3 * r + c] = e; (void)b; (void)dd;
        }
        for (int r = 0; r < 2; ++r) for (int c = 0; c < 3; ++c) {
            double b = 0, dd = 0;
            for (int l = 0; l < 3; ++l) { b += A[3 * r + l] * Sk[3 * l + c]; dd += Cc[3 * r + l] * Ski[3 * l + c]; }
            B[3 * r + c] = b; D[3 * r + c] = dd;
        }
        for (int r = 0; r < 2; ++r) for (int c = 0; c < 3; ++c) {
            const int row = 2 * eye + r;
            CM(H, 4, row, c) = B[3 * r + c];                    /* epose theta:  H_proj (L) R_w2cl skew(pf_w) */
            CM(H, 4, row, 3 + c) = -A[3 * r + c];               /* epose p                                   */
            CM(H, 4, row, 9 + c) = D[3 * r + c];                /* ext theta:    H_proj (L) R_cl2i^T skew(pf_i) */
            CM(H, 4, row, 12 + c) = -Cc[3 * r + c];             /* ext p                                     */
            CM(H, 4, row, 15 + c) = eye == 0 ? -B[3 * r + c] : -E[3 * r + c];      /* anchor theta (right rows: Q12) */
            CM(H, 4, row, 21 + c) = A[3 * r + c];               /* pf                                        */
        }
    }
    return rows;
}

/* H (rows x 15 col-major, ld = 4): [curr pose 6 | anchor 6 | pf 3]; LandmarkUpdate.cpp:574-626 (mono) and its stereo twin. */
int orc_landmark_rows_sw(const double R_cm[9], const double p_cm[3], const double pf[3], const double* uv, int stereo,
                         const double R_lr[9], const double t_lr[3], int curr_is_anchor, double* H, double* res)
{
    const int rows = stereo ? 4 : 2;
    double d[3], pf_cl[3], pf_cr[3], RT[9], Sk[9];
    for (int i = 0; i < 3; ++i) d[i] = pf[i] - p_cm[i];
    m3_tmulv(R_cm, d, pf_cl);
    m3_T(R_cm, RT);
    orc_skew(pf, Sk);
    memset(H, 0, sizeof(double) * 4 * 15);
    for (int eye = 0; eye < (stereo ? 2 : 1); ++eye) {
        double q[3], Hp[6], L[9];
        if (eye == 0) { memcpy(q, pf_cl, sizeof q); m3_eye(L); }
        else {
            m3_mulv(R_lr, pf_cl, pf_cr);
            for (int i = 0; i < 3; ++i) pf_cr[i] += t_lr[i];
            memcpy(q, pf_cr, sizeof q); memcpy(L, R_lr, sizeof L);
        }
        proj_jac(q, Hp);
        res[2 * eye] = uv[2 * eye] - q[0] / q[2];
        res[2 * eye + 1] = uv[2 * eye + 1] - q[1] / q[2];
        double HL[6], A[6], B[6];
        for (int r = 0; r < 2; ++r) for (int c = 0; c < 3; ++c)
            HL[3 * r + c] = Hp[3 * r] * L[c] + Hp[3 * r + 1] * L[3 + c] + Hp[3 * r + 2] * L[6 + c];
        for (int r = 0; r < 2; ++r) for (int c = 0; c < 3; ++c) {
            double a = 0;
            for (int l = 0; l < 3; ++l) a += HL[3 * r + l] * RT[3 * l + c];
            A[3 * r + c] = a;
        }
        for (int r = 0; r < 2; ++r) for (int c = 0; c < 3; ++c) {
            double b = 0;
            for (int l = 0; l < 3; ++l) b += A[3 * r + l] * Sk[3 * l + c];
            B[3 * r + c] = b;
        }
        for (int r = 0; r < 2; ++r) for (int c = 0; c < 3; ++c) {
            const int row = 2 * eye + r;
            if (!curr_is_anchor) { CM(H, 4, row, c) = B[3 * r + c]; CM(H, 4, row, 6 + c) = -B[3 * r + c]; }
            CM(H, 4, row, 3 + c) = -A[3 * r + c];
            CM(H, 4, row, 12 + c) = A[3 * r + c];
        }
    }
    return rows;
}
