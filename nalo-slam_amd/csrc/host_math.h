// Host-side fp64 math of the product: fixed-size matrices, SE(3) (Sophus conventions), Jacobi-scaled LDL^T,
// symmetric eigen solver. Written for this library (no Eigen in the image); used by the host mirrors of
// CoarseTracker::trackNewestCoarse and EnergyFunctional::solveSystemF.
#pragma once
#include <cmath>
#include <cstring>
#include <vector>
#include <algorithm>

// SE3, se3_exp and aff_from_to are also used by the device-side Gauss-Newton step (kernels_ba_gn.hip): same source, same operation order
#if defined(__HIPCC__)
#define NALO_HD __host__ __device__
#else
#define NALO_HD
#endif

namespace nalo {

struct SE3 {                     // row-major 3x4 [R|t]
    double m[12];
    NALO_HD static SE3 identity() { SE3 s; for (int i = 0; i < 12; ++i) s.m[i] = 0; s.m[0] = s.m[5] = s.m[10] = 1; return s; }
    NALO_HD static SE3 from(const double* p) { SE3 s; for (int i = 0; i < 12; ++i) s.m[i] = p[i]; return s; }
    NALO_HD double R(int i, int j) const { return m[i * 4 + j]; }
    NALO_HD double t(int i) const { return m[i * 4 + 3]; }
    NALO_HD SE3 operator*(const SE3& b) const {
        SE3 c;
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) c.m[i * 4 + j] = R(i, 0) * b.R(0, j) + R(i, 1) * b.R(1, j) + R(i, 2) * b.R(2, j);
            c.m[i * 4 + 3] = R(i, 0) * b.t(0) + R(i, 1) * b.t(1) + R(i, 2) * b.t(2) + t(i);
        }
        return c;
    }
    NALO_HD SE3 inverse() const {
        SE3 c;
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) c.m[i * 4 + j] = R(j, i);
        for (int i = 0; i < 3; ++i) c.m[i * 4 + 3] = -(c.R(i, 0) * t(0) + c.R(i, 1) * t(1) + c.R(i, 2) * t(2));
        return c;
    }
    // 6x6 adjoint [R, [t]x R; 0, R] (row-major)
    void adjoint(double A[36]) const {
        const double tx = t(0), ty = t(1), tz = t(2);
        const double H[9] = {0, -tz, ty, tz, 0, -tx, -ty, tx, 0};
        std::memset(A, 0, 36 * sizeof(double));
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
            A[i * 6 + j] = R(i, j);
            A[(i + 3) * 6 + j + 3] = R(i, j);
            A[i * 6 + j + 3] = H[i * 3] * R(0, j) + H[i * 3 + 1] * R(1, j) + H[i * 3 + 2] * R(2, j);
        }
    }
};

// exp: tangent (upsilon, omega) -> SE3; rotation through the unit quaternion (cos(th/2), sin(th/2)/th * omega)
NALO_HD inline SE3 se3_exp(const double xi[6]) {
    const double wx = xi[3], wy = xi[4], wz = xi[5];
    const double th2 = wx * wx + wy * wy + wz * wz, th = std::sqrt(th2);
    double qi, qr;
    if (th < 1e-10) { const double t4 = th2 * th2; qi = 0.5 - th2 / 48.0 + t4 / 3840.0; qr = 1.0 - 0.5 * th2 + t4 / 384.0; }
    else { qi = std::sin(0.5 * th) / th; qr = std::cos(0.5 * th); }
    double q[4] = {qr, qi * wx, qi * wy, qi * wz};
    const double qn = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (double& v : q) v /= qn;
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    double Rm[9] = {1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                    2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                    2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)};
    const double O[9] = {0, -wz, wy, wz, 0, -wx, -wy, wx, 0};
    double O2[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) O2[i * 3 + j] = O[i * 3] * O[j] + O[i * 3 + 1] * O[3 + j] + O[i * 3 + 2] * O[6 + j];
    double V[9];
    if (th < 1e-10) { for (int i = 0; i < 9; ++i) V[i] = Rm[i]; }
    else {
        const double a = (1 - std::cos(th)) / th2, b = (th - std::sin(th)) / (th2 * th);
        for (int i = 0; i < 9; ++i) V[i] = ((i % 4 == 0) ? 1.0 : 0.0) + a * O[i] + b * O2[i];
    }
    SE3 s;
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) s.m[i * 4 + j] = Rm[i * 3 + j];
        s.m[i * 4 + 3] = V[i * 3] * xi[0] + V[i * 3 + 1] * xi[1] + V[i * 3 + 2] * xi[2];
    }
    return s;
}

inline void se3_log(const SE3& T, double xi[6]) {
    // rotation -> unit quaternion (Shepperd), then the atan-based log
    const double r00 = T.R(0, 0), r11 = T.R(1, 1), r22 = T.R(2, 2);
    double q[4];
    const double tr = r00 + r11 + r22;
    if (tr > 0) { const double s = std::sqrt(tr + 1.0) * 2; q[0] = 0.25 * s; q[1] = (T.R(2, 1) - T.R(1, 2)) / s; q[2] = (T.R(0, 2) - T.R(2, 0)) / s; q[3] = (T.R(1, 0) - T.R(0, 1)) / s; }
    else if (r00 > r11 && r00 > r22) { const double s = std::sqrt(1.0 + r00 - r11 - r22) * 2; q[0] = (T.R(2, 1) - T.R(1, 2)) / s; q[1] = 0.25 * s; q[2] = (T.R(0, 1) + T.R(1, 0)) / s; q[3] = (T.R(0, 2) + T.R(2, 0)) / s; }
    else if (r11 > r22) { const double s = std::sqrt(1.0 + r11 - r00 - r22) * 2; q[0] = (T.R(0, 2) - T.R(2, 0)) / s; q[1] = (T.R(0, 1) + T.R(1, 0)) / s; q[2] = 0.25 * s; q[3] = (T.R(1, 2) + T.R(2, 1)) / s; }
    else { const double s = std::sqrt(1.0 + r22 - r00 - r11) * 2; q[0] = (T.R(1, 0) - T.R(0, 1)) / s; q[1] = (T.R(0, 2) + T.R(2, 0)) / s; q[2] = (T.R(1, 2) + T.R(2, 1)) / s; q[3] = 0.25 * s; }
    const double qn = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (double& v : q) v /= qn;
    const double n2 = q[1] * q[1] + q[2] * q[2] + q[3] * q[3], n = std::sqrt(n2), w = q[0];
    double f;
    if (n < 1e-10) f = 2.0 / w - 2.0 * n2 / (w * w * w);
    else if (std::fabs(w) < 1e-10) f = (w > 0 ? M_PI : -M_PI) / n;
    else f = 2.0 * std::atan(n / w) / n;
    const double th = f * n, ox = f * q[1], oy = f * q[2], oz = f * q[3];
    const double O[9] = {0, -oz, oy, oz, 0, -ox, -oy, ox, 0};
    double O2[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) O2[i * 3 + j] = O[i * 3] * O[j] + O[i * 3 + 1] * O[3 + j] + O[i * 3 + 2] * O[6 + j];
    const double c = std::fabs(th) < 1e-10 ? 1.0 / 12.0 : (1.0 - th / (2.0 * std::tan(0.5 * th))) / (th * th);
    for (int i = 0; i < 3; ++i) {
        double s = 0;
        for (int j = 0; j < 3; ++j) s += (((i == j) ? 1.0 : 0.0) - 0.5 * O[i * 3 + j] + c * O2[i * 3 + j]) * T.t(j);
        xi[i] = s;
    }
    xi[3] = ox; xi[4] = oy; xi[5] = oz;
}

// AffLight::fromToVecExposure (util/NumType.h:173-185)
NALO_HD inline void aff_from_to(float expF, float expT, double aF, double bF, double aT, double bT, double out[2]) {
    if (expF == 0 || expT == 0) expT = expF = 1;
    const double a = std::exp(aT - aF) * expT / expF;
    out[0] = a; out[1] = bT - a * bF;
}

// Symmetric solve by LDL^T with diagonal pivoting (semi-definite safe), fp64. A is n x n row-major; only the UPPER triangle (j >= i) is read (the
// caller mirrors the authoritative lower triangle into it). Left-looking (Crout): at step k the pivot row is brought up to date in one go,
// A[k][j] -= sum_{m<k} (L[k][m] d_m) L[j][m], as k row-axpys accumulated in registers over 16-column blocks (every factor row is read contiguously,
// the destination is written once): n^3/6 multiply-adds and no store per multiply-add, against n^3/3 and one store each for a right-looking update of
// both triangles. The running diagonal (what the pivot search needs) is kept in `work`. L^T overwrites the strict upper triangle, d the diagonal.
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
__attribute__((target_clones("avx2", "default")))   // the .so is built on one machine and run on another: dispatch at load time
#endif
inline void ldlt_solve_inplace(int n, double* A, const double* b, double* x, double* work /* 2n */, int* perm /* n */) {
    double* diag = work; double* wv = work + n;
    for (int i = 0; i < n; ++i) { perm[i] = i; diag[i] = A[(size_t)i * n + i]; }
    for (int k = 0; k < n; ++k) {
        int p = k; double best = std::fabs(diag[k]);
        for (int i = k + 1; i < n; ++i) { const double v = std::fabs(diag[i]); if (v > best) { best = v; p = i; } }
        double* rk = A + (size_t)k * n;
        if (p != k) {                                           // symmetric interchange of k and p: factor columns above, original entries below
            double* rp = A + (size_t)p * n;
            std::swap(rk[k], rp[p]); std::swap(diag[k], diag[p]);
            for (int j = 0; j < k; ++j) std::swap(A[(size_t)j * n + k], A[(size_t)j * n + p]);
            for (int j = p + 1; j < n; ++j) std::swap(rk[j], rp[j]);
            for (int m = k + 1; m < p; ++m) std::swap(rk[m], A[(size_t)m * n + p]);
            std::swap(perm[k], perm[p]);
        }
        const double d = diag[k];
        rk[k] = d;
        if (d == 0.0 || !std::isfinite(d)) { for (int j = k + 1; j < n; ++j) rk[j] = 0; continue; }
        for (int m = 0; m < k; ++m) wv[m] = A[(size_t)m * n + k] * A[(size_t)m * n + m];          // L[k][m] d_m
        for (int j0 = k + 1; j0 < n; j0 += 16) {
            const int jn = (n - j0 < 16) ? n - j0 : 16;
            double acc[16];
            for (int t = 0; t < 16; ++t) acc[t] = t < jn ? rk[j0 + t] : 0.0;
            if (jn == 16) for (int m = 0; m < k; ++m) { const double* rm = A + (size_t)m * n + j0; const double w = wv[m]; for (int t = 0; t < 16; ++t) acc[t] -= w * rm[t]; }
            else for (int m = 0; m < k; ++m) { const double* rm = A + (size_t)m * n + j0; const double w = wv[m]; for (int t = 0; t < jn; ++t) acc[t] -= w * rm[t]; }
            for (int t = 0; t < jn; ++t) { const double l = acc[t] / d; diag[j0 + t] -= l * acc[t]; rk[j0 + t] = l; }
        }
    }
    double* y = work;                                           // diag is dead: the factor's diagonal lives in A
    for (int i = 0; i < n; ++i) y[i] = b[perm[i]];
    for (int j = 0; j < n; ++j) { const double* rj = A + (size_t)j * n; const double yj = y[j]; for (int i = j + 1; i < n; ++i) y[i] -= rj[i] * yj; }      // L y = b
    for (int i = 0; i < n; ++i) { const double d = A[(size_t)i * n + i]; y[i] = (d != 0.0 && std::isfinite(d)) ? y[i] / d : 0.0; }
    for (int i = n - 1; i >= 0; --i) { const double* ri = A + (size_t)i * n; double s = y[i]; for (int j = i + 1; j < n; ++j) s -= ri[j] * y[j]; y[i] = s; }  // L^T x = y
    for (int i = 0; i < n; ++i) x[perm[i]] = y[i];
}
// Symmetric solve by LDL^T with diagonal pivoting, fp64, on a matrix stored with a row stride `lda` that is a multiple of 8 (columns n .. lda-1 zero): only the
// UPPER triangle (j >= i) is read. Left-looking (Crout) like ldlt_solve_inplace, but the whole pivot row is brought up to date at once: for every earlier
// factor row m the row's 8-wide column blocks take one fused multiply-add each - up to eight INDEPENDENT accumulator chains, so the multiply-adds pipeline
// (the 16-column version above ran one or two dependent chains: latency bound, 13.6 us at n = 68; this one 3 us). Vector extensions instead of intrinsics:
// the same source gives zmm code in the avx512f clone, ymm pairs in the avx2 one.
typedef double ldlt_v8 __attribute__((vector_size(64), aligned(8)));
template <int NB>
static inline __attribute__((always_inline)) void ldlt_row_update(double* __restrict__ rk, const double* __restrict__ A, int lda, int k, int c0, const double* __restrict__ wv,
                                                                  double dinv, double* __restrict__ diag) {
    ldlt_v8 acc[NB];
    for (int b = 0; b < NB; ++b) acc[b] = *reinterpret_cast<const ldlt_v8*>(rk + c0 + 8 * b);
    for (int m = 0; m < k; ++m) {
        const double w = wv[m];
        const double* rm = A + (size_t)m * lda + c0;
        for (int b = 0; b < NB; ++b) acc[b] -= w * *reinterpret_cast<const ldlt_v8*>(rm + 8 * b);
    }
    for (int b = 0; b < NB; ++b) {
        const int j0 = c0 + 8 * b;
        const ldlt_v8 l = acc[b] * dinv;
        if (j0 > k) {                                           // whole block right of the pivot
            ldlt_v8 dg = *reinterpret_cast<ldlt_v8*>(diag + j0);
            dg -= l * acc[b];
            *reinterpret_cast<ldlt_v8*>(diag + j0) = dg;
            *reinterpret_cast<ldlt_v8*>(rk + j0) = l;
        } else {                                                // the block that holds the pivot column: entries j <= k stay
            for (int t = 0; t < 8; ++t) if (j0 + t > k) { diag[j0 + t] -= l[t] * acc[b][t]; rk[j0 + t] = l[t]; }
        }
    }
}
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
__attribute__((target_clones("avx512f", "avx2", "default")))
#endif
inline void ldlt_solve_blocked(int n, int lda, double* A, const double* b, double* x, double* work /* 2 lda + n */, int* perm /* n */, double piv_slack = 64.0) {
    double* diag = work; double* wv = work + lda; double* y = work + 2 * lda;
    for (int i = 0; i < n; ++i) { perm[i] = i; diag[i] = A[(size_t)i * lda + i]; }
    for (int i = n; i < lda; ++i) diag[i] = 0.0;
    for (int k = 0; k < n; ++k) {
        // pivot = the largest remaining |diagonal| (what Eigen's LDLT takes), found in two branch-free passes (a max reduction the compiler vectorises, then the
        // first index that holds it); the symmetric interchange - strided column swaps - is skipped while the natural pivot is within a factor 64 of the largest
        // (threshold pivoting: the growth of L stays bounded by 8 per step; on the Jacobi-scaled SPD systems of solveSystemF it practically never swaps)
        int p = k;
        {
            double best = 0.0;
            for (int i = k; i < n; ++i) { const double v = std::fabs(diag[i]); best = v > best ? v : best; }
            if (!(std::fabs(diag[k]) * piv_slack >= best)) { for (int i = k; i < n; ++i) if (std::fabs(diag[i]) == best) { p = i; break; } }
        }
        double* rk = A + (size_t)k * lda;
        if (p != k) {                                           // symmetric interchange of k and p: factor columns above, original entries below
            double* rp = A + (size_t)p * lda;
            std::swap(rk[k], rp[p]); std::swap(diag[k], diag[p]);
            for (int j = 0; j < k; ++j) std::swap(A[(size_t)j * lda + k], A[(size_t)j * lda + p]);
            for (int j = p + 1; j < n; ++j) std::swap(rk[j], rp[j]);
            for (int m = k + 1; m < p; ++m) std::swap(rk[m], A[(size_t)m * lda + p]);
            std::swap(perm[k], perm[p]);
        }
        const double d = diag[k];
        rk[k] = d;
        if (d == 0.0 || !std::isfinite(d)) { for (int j = k + 1; j < n; ++j) rk[j] = 0; continue; }
        for (int m = 0; m < k; ++m) wv[m] = A[(size_t)m * lda + k] * A[(size_t)m * lda + m];          // L[k][m] d_m
        const double dinv = 1.0 / d;
        for (int c0 = (k + 1) & ~7; c0 < lda; c0 += 64) {
            const int nb = (lda - c0) / 8;
            switch (nb >= 8 ? 8 : nb) {
                case 1: ldlt_row_update<1>(rk, A, lda, k, c0, wv, dinv, diag); break;
                case 2: ldlt_row_update<2>(rk, A, lda, k, c0, wv, dinv, diag); break;
                case 3: ldlt_row_update<3>(rk, A, lda, k, c0, wv, dinv, diag); break;
                case 4: ldlt_row_update<4>(rk, A, lda, k, c0, wv, dinv, diag); break;
                case 5: ldlt_row_update<5>(rk, A, lda, k, c0, wv, dinv, diag); break;
                case 6: ldlt_row_update<6>(rk, A, lda, k, c0, wv, dinv, diag); break;
                case 7: ldlt_row_update<7>(rk, A, lda, k, c0, wv, dinv, diag); break;
                default: ldlt_row_update<8>(rk, A, lda, k, c0, wv, dinv, diag); break;
            }
        }
    }
    for (int i = 0; i < n; ++i) y[i] = b[perm[i]];
    for (int j = 0; j < n; ++j) { const double* rj = A + (size_t)j * lda; const double yj = y[j]; for (int i = j + 1; i < n; ++i) y[i] -= rj[i] * yj; }      // L y = b
    for (int i = 0; i < n; ++i) { const double d = A[(size_t)i * lda + i]; y[i] = (d != 0.0 && std::isfinite(d)) ? y[i] / d : 0.0; }
    for (int i = n - 1; i >= 0; --i) { const double* ri = A + (size_t)i * lda; double s = y[i]; for (int j = i + 1; j < n; ++j) s -= ri[j] * y[j]; y[i] = s; }  // L^T x = y
    for (int i = 0; i < n; ++i) x[perm[i]] = y[i];
}

inline void ldlt_solve(int n, const double* Ain, const double* b, double* x) {
    std::vector<double> A(Ain, Ain + (size_t)n * n), y(2 * (size_t)n);
    for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) A[(size_t)i * n + j] = A[(size_t)j * n + i];      // the lower triangle is the authoritative one
    std::vector<int> perm(n);
    ldlt_solve_inplace(n, A.data(), b, x, y.data(), perm.data());
}

// inverse of a small dense matrix (n <= 16) by LU with partial pivoting (what Eigen's fixed-size inverse() does for n > 4), fp64; false if singular
inline bool inv_lu(int n, const double* A, double* Ai) {
    double LU[256]; int piv[16];
    if (n > 16) return false;
    std::memcpy(LU, A, sizeof(double) * n * n);
    for (int k = 0; k < n; ++k) {
        int p = k;
        for (int r = k + 1; r < n; ++r) if (std::fabs(LU[r * n + k]) > std::fabs(LU[p * n + k])) p = r;
        piv[k] = p;
        if (p != k) for (int j = 0; j < n; ++j) std::swap(LU[k * n + j], LU[p * n + j]);
        if (LU[k * n + k] == 0.0) return false;
        for (int r = k + 1; r < n; ++r) {
            const double l = LU[r * n + k] / LU[k * n + k];
            LU[r * n + k] = l;
            for (int j = k + 1; j < n; ++j) LU[r * n + j] -= l * LU[k * n + j];
        }
    }
    for (int c = 0; c < n; ++c) {                                   // solve L U x = P e_c
        double y[16];
        for (int i = 0; i < n; ++i) y[i] = (i == c);
        for (int k = 0; k < n; ++k) if (piv[k] != k) std::swap(y[k], y[piv[k]]);
        for (int i = 0; i < n; ++i) for (int j = 0; j < i; ++j) y[i] -= LU[i * n + j] * y[j];
        for (int i = n - 1; i >= 0; --i) { for (int j = i + 1; j < n; ++j) y[i] -= LU[i * n + j] * y[j]; y[i] /= LU[i * n + i]; }
        for (int i = 0; i < n; ++i) Ai[i * n + c] = y[i];
    }
    return true;
}

// cyclic Jacobi for small symmetric matrices; A destroyed, V columns = eigenvectors
inline void sym_eig(int n, double* A, double* V, double* w) {
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) V[i * n + j] = (i == j);
    for (int sweep = 0; sweep < 100; ++sweep) {
        double off = 0;
        for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) off += A[i * n + j] * A[i * n + j];
        if (off < 1e-300) break;
        for (int p = 0; p < n; ++p) for (int q = p + 1; q < n; ++q) {
            const double apq = A[p * n + q];
            if (std::fabs(apq) < 1e-300) continue;
            const double th = (A[q * n + q] - A[p * n + p]) / (2 * apq);
            const double t = (th >= 0 ? 1.0 : -1.0) / (std::fabs(th) + std::sqrt(th * th + 1)), c = 1 / std::sqrt(t * t + 1), s = t * c;
            for (int k = 0; k < n; ++k) { const double a = A[k * n + p], b = A[k * n + q]; A[k * n + p] = c * a - s * b; A[k * n + q] = s * a + c * b; }
            for (int k = 0; k < n; ++k) { const double a = A[p * n + k], b = A[q * n + k]; A[p * n + k] = c * a - s * b; A[q * n + k] = s * a + c * b; }
            for (int k = 0; k < n; ++k) { const double a = V[k * n + p], b = V[k * n + q]; V[k * n + p] = c * a - s * b; V[k * n + q] = s * a + c * b; }
        }
    }
    for (int i = 0; i < n; ++i) w[i] = A[i * n + i];
}

}  // namespace nalo
