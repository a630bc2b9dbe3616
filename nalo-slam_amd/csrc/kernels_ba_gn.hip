// Device-side Gauss-Newton iteration of FullSystem::optimize (reference src/FullSystem/FullSystemOptimize.cpp:482-545): everything the host mirror
// does between two linearisations, in ONE workgroup launch, so that optimize() enqueues all its iterations without waiting on the device once:
//   solveSystemF      OptimizationBackend/EnergyFunctional.cpp:776-914  H = HL + HM + HA (diag x (1+lambda)) - Hsc/(1+lambda), Jacobi scaling, LDL^T with
//                                                                        diagonal pivoting (Eigen's ldlt), orthogonalize(x) from iteration 2 (:898-902)
//   resubstituteF_MT  :263-289 (frame part)                              steps, xAd for the point kernel
//   doStepFromBackup  FullSystemOptimize.cpp:217-299 (frame part)        states = backup + step, FrameHessian::setState (SE3::exp x evalPT), calibration, step norms
//   setPrecalcValues  HessianBlocks.cpp:192-222, EnergyFunctional.cpp:171-194   FrameFramePrecalc records, adHTdeltaF, cDeltaF
// and the termination test of :544 (canbreak): when it fires, a device flag makes every later kernel of the queued iterations return at once.
// fp64 throughout, the same operation order as the host mirror in host_ba.hip (same SE3 source, host_math.h), so both drivers give the same poses up to
// the libm difference of sin/cos/exp (1 ulp). The (8W+4)^2 system lives in LDS; the factorisation runs on all four waves with two workgroup barriers
// per pivot (the rank-1 update is the only O(n^2) part), pivot search / swaps / substitutions on wave 0.
#include "nalo_internal.h"
#include "ba_device.h"

namespace nalo {

constexpr int kGnThreads = 256;

__device__ __forceinline__ void gn_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__global__ __launch_bounds__(kGnThreads) void ba_gn_kernel(GNDev G, int iteration, int never_break, double lambda) {
    extern __shared__ double sh[];
    const int W = G.W, n = G.n, n1 = n + 1, tid = threadIdx.x, NT = kGnThreads;
    double* HF = sh;                       // [n][n]
    double* bF = HF + (size_t)n * n;       // [n]
    double* sv = bF + n;                   // [n]
    double* xs = sv + n;                   // [n]
    double* dl = xs + n;                   // [n] delta
    double* lk = dl + n;                   // [n] column of L being applied
    double* cf = lk + n;                   // [8] projector coefficients
    int* perm = (int*)(cf + 8);            // [n]
    __shared__ int s_piv, s_stop;
    __shared__ double s_d;
    __shared__ float s_sum[4];
    if (tid == 0) s_stop = *G.stop;
    __syncthreads();
    if (s_stop) return;
    // ---- termination test of the previous step (FullSystemOptimize.cpp:286-296 + :544); the point sums were finished by ba_reduce
    if (iteration >= 2 && !never_break) {
        if (tid == 0) {
            const double* s3 = G.stitched + 2 * (size_t)n1 * n1 + 2 * W * W;
            const float numID = (float)s3[2];
            const float sumNID = numID > 0 ? (float)(s3[1] / numID) : 0.f;
            const float th = 1.2f;                                             // setting_thOptIterations
            const float sA = G.sums[0], sB = G.sums[1], sT = G.sums[2], sR = G.sums[3];
            const bool cb = sqrtf(sA) < 0.0005 * th && sqrtf(sB) < 0.00005 * th && sqrtf(sR) < 0.00005 * th && sqrtf(sT) * sumNID < 0.00005 * th;
            if (cb) { *G.stop = 1; s_stop = 1; }
        }
        __syncthreads();
        if (s_stop) return;
    }
    const double* HA = G.stitched;
    const double* HS = G.stitched + (size_t)n1 * n1;
    // ---- delta = state - state_zero (EFFrame::takeData), calibration delta through its float (setDeltaF)
    for (int r = tid; r < n; r += NT) {
        if (r < 4) dl[r] = (double)(float)(G.c_value[r] - G.c_zero[r]);
        else { const int f = (r - 4) >> 3, i = (r - 4) & 7; dl[r] = G.state[f * 10 + i] - G.state_zero[f * 10 + i]; }
    }
    __syncthreads();
    // ---- H and b of :795-868, one row per thread for the b dot product (ascending order like the host), all threads for H
    const double fsc = 1.0 / (1 + lambda);
    for (int e = tid; e < n * n; e += NT) {
        const int r = e / n, cc = e - r * n;
        double v;
        if (r == cc) {
            const double HLd = r < 4 ? kInitialCalibHessian : G.prior[((r - 4) >> 3) * 8 + ((r - 4) & 7)];
            v = (HLd + G.HM[e]) + HA[(size_t)r * n1 + cc];
            v *= (1 + lambda);
        } else v = (0.0 + G.HM[e]) + HA[(size_t)r * n1 + cc];
        v -= HS[(size_t)r * n1 + cc] * fsc;
        HF[e] = v;
    }
    for (int r = tid; r < n; r += NT) {
        double bLr;
        if (r < 4) bLr = kInitialCalibHessian * dl[r];
        else { const int f = (r - 4) >> 3, i = (r - 4) & 7; bLr = G.prior[f * 8 + i] * G.state[f * 10 + i]; }     // delta_prior = state (state_prior = 0)
        double sdot = 0;
        const double* hm = G.HM + (size_t)r * n;
        for (int cc = 0; cc < n; ++cc) sdot += hm[cc] * dl[cc];
        bF[r] = bLr + (G.bM[r] + sdot) + HA[(size_t)r * n1 + n] - HS[(size_t)r * n1 + n];
    }
    __syncthreads();
    // ---- Jacobi scaling (:872-877), lower triangle mirrored (Eigen's LDLT reads the lower triangle only)
    for (int i = tid; i < n; i += NT) sv[i] = 1.0 / sqrt(HF[(size_t)i * n + i] + 10);
    __syncthreads();
    for (int e = tid; e < n * n; e += NT) { const int i = e / n, j = e - i * n; HF[e] = sv[i] * HF[e] * sv[j]; }
    for (int i = tid; i < n; i += NT) { bF[i] *= sv[i]; perm[i] = i; }
    __syncthreads();
    for (int e = tid; e < n * n; e += NT) { const int i = e / n, j = e - i * n; if (j > i) HF[e] = HF[(size_t)j * n + i]; }
    __syncthreads();
    // ---- LDL^T with diagonal pivoting (right-looking; the host's ldlt_solve_inplace is the left-looking form of the same factorisation: same pivots,
    //      results equal to rounding)
    for (int k = 0; k < n; ++k) {
        if (tid < 64) {
            // pivot: largest |diagonal| of the trailing block, first one on ties
            double best = -1.0; int bi = n;
            for (int i = k + tid; i < n; i += 64) { const double v = fabs(HF[(size_t)i * n + i]); if (v > best) { best = v; bi = i; } }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const double ob = __shfl_xor(best, off); const int oi = __shfl_xor(bi, off);
                if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
            }
            const int p = bi < n ? bi : k;                                     // all-NaN diagonal: keep k like the host
            if (p != k) {                                                      // symmetric swap: rows, then columns
                for (int j = tid; j < n; j += 64) { const double a = HF[(size_t)k * n + j]; HF[(size_t)k * n + j] = HF[(size_t)p * n + j]; HF[(size_t)p * n + j] = a; }
                gn_wave_sync();
                for (int j = tid; j < n; j += 64) { const double a = HF[(size_t)j * n + k]; HF[(size_t)j * n + k] = HF[(size_t)j * n + p]; HF[(size_t)j * n + p] = a; }
                if (tid == 0) { const int t = perm[k]; perm[k] = perm[p]; perm[p] = t; }
                gn_wave_sync();
            }
            const double d = HF[(size_t)k * n + k];
            const bool ok = d != 0.0 && isfinite(d);
            for (int i = k + 1 + tid; i < n; i += 64) {
                const double l = ok ? HF[(size_t)i * n + k] / d : 0.0;
                HF[(size_t)i * n + k] = l; lk[i] = l;
            }
            if (tid == 0) s_piv = ok ? 1 : 0;
        }
        __syncthreads();
        if (s_piv) {                                                           // trailing update A[i][j] -= l_i * A[k][j], all waves
            const int m = n - k - 1;
            for (int e = tid; e < m * m; e += NT) {
                const int i = k + 1 + e / m, j = k + 1 + e % m;
                HF[(size_t)i * n + j] -= lk[i] * HF[(size_t)k * n + j];
            }
        }
        __syncthreads();
    }
    if (tid < 64) {
        // L y = P b (column oriented, ascending), D, L^T (descending): the substitution order of the host code
        for (int i = tid; i < n; i += 64) xs[i] = bF[perm[i]];
        gn_wave_sync();
        for (int j = 0; j < n; ++j) {
            const double yj = xs[j];
            for (int i = j + 1 + tid; i < n; i += 64) xs[i] -= HF[(size_t)i * n + j] * yj;
            gn_wave_sync();
        }
        for (int i = tid; i < n; i += 64) { const double d = HF[(size_t)i * n + i]; xs[i] = (d != 0.0 && isfinite(d)) ? xs[i] / d : 0.0; }
        gn_wave_sync();
        for (int i = n - 1; i >= 0; --i) {
            const double yi = xs[i];
            for (int j = tid; j < i; j += 64) xs[j] -= HF[(size_t)i * n + j] * yi;
            gn_wave_sync();
        }
        for (int i = tid; i < n; i += 64) bF[perm[i]] = xs[i];                 // x = P^T y (bF reused)
        gn_wave_sync();
        for (int i = tid; i < n; i += 64) xs[i] = bF[i] * sv[i];
        gn_wave_sync();
        if (iteration >= 2) {                                                  // SOLVER_ORTHOGONALIZE_X_LATER (:898-902): x -= U U^T x
            if (tid < 7) { double s = 0; for (int r = 0; r < n; ++r) s += G.Sproj[(size_t)r * 7 + tid] * xs[r]; cf[tid] = s; }
            gn_wave_sync();
            for (int r = tid; r < n; r += 64) { double s = 0; for (int k2 = 0; k2 < 7; ++k2) s += G.Sproj[(size_t)r * 7 + k2] * cf[k2]; xs[r] -= s; }
            gn_wave_sync();
        }
    }
    __syncthreads();
    // ---- steps (resubstituteF_MT :263-289), xAd for the point kernel, backup + new states (doStepFromBackup with unit step factors)
    for (int i = tid; i < n; i += NT) G.x[i] = xs[i];
    float* xc = G.xad;                                                          // [xc (64) | xAd (W*W*8)]
    if (tid < 4) { xc[tid] = (float)xs[tid]; G.c_backup[tid] = G.c_value[tid]; G.c_value[tid] = G.c_value[tid] + 1.0 * (-xs[tid]); }
    for (int e = tid; e < W * W * 8; e += NT) {
        const int pair = e >> 3, j = e & 7, h = pair / W, t = pair - h * W;
        const float *AH = G.adHostF + (size_t)(h + W * t) * 64, *AT = G.adTargetF + (size_t)(h + W * t) * 64;
        float s1 = 0, s2 = 0;
        for (int i = 0; i < 8; ++i) { s1 += (float)xs[4 + 8 * h + i] * AH[i * 8 + j]; s2 += (float)xs[4 + 8 * t + i] * AT[i * 8 + j]; }
        xc[64 + (size_t)(W * h + t) * 8 + j] = s1 + s2;
    }
    for (int e = tid; e < W * 10; e += NT) {
        const int f = e / 10, i = e - f * 10;
        const double st = G.state[e];
        const double step = i < 8 ? -xs[4 + 8 * f + i] : 0.0;
        G.backup[e] = st; G.step[e] = step;
        G.state[e] = st + 1.0 * step;
    }
    __syncthreads();
    if (tid == 0) {                                                             // step norms of the break test (:279-296), float accumulation like the host
        float sumA = 0, sumB = 0, sumT = 0, sumR = 0;
        for (int f = 0; f < W; ++f) {
            const double* sp = G.step + f * 10;
            sumA += sp[6] * sp[6]; sumB += sp[7] * sp[7];
            sumT += sp[0] * sp[0] + sp[1] * sp[1] + sp[2] * sp[2];
            sumR += sp[3] * sp[3] + sp[4] * sp[4] + sp[5] * sp[5];
        }
        G.sums[0] = sumA / W; G.sums[1] = sumB / W; G.sums[2] = sumT / W; G.sums[3] = sumR / W;
        *G.iters_done = iteration + 1;
    }
    // ---- CalibHessian::setValue + FrameHessian::setState (HessianBlocks.h:208-222, 381-395)
    if (tid < W) {
        const double* st = G.state + tid * 10;
        double sc[6];
        for (int i = 0; i < 3; ++i) sc[i] = kScaleXiTrans * st[i];
        for (int i = 3; i < 6; ++i) sc[i] = kScaleXiRot * st[i];
        const SE3 w2c = se3_exp(sc) * SE3::from(G.evalPT + tid * 12);
        const SE3 c2w = w2c.inverse();
        for (int i = 0; i < 12; ++i) { G.w2c[tid * 12 + i] = w2c.m[i]; G.c2w[tid * 12 + i] = c2w.m[i]; }
    }
    __syncthreads();
    // ---- FrameFramePrecalc::set for every pair + adHTdeltaF (setDeltaF) + the calibration floats
    float cs[4];
    {
        const double v0 = kScaleF * G.c_value[0], v1 = kScaleF * G.c_value[1], v2 = kScaleC * G.c_value[2], v3 = kScaleC * G.c_value[3];
        cs[0] = (float)v0; cs[1] = (float)v1; cs[2] = (float)v2; cs[3] = (float)v3;
    }
    const float fx = cs[0], fy = cs[1], cx = cs[2], cy = cs[3];
    const size_t nfl = (size_t)W * W * kPreStride;
    if (tid == 0) {
        float* cal = G.pre + nfl;
        cal[0] = fx; cal[1] = fy; cal[2] = cx; cal[3] = cy; cal[4] = 1.0f / fx; cal[5] = 1.0f / fy;
        for (int i = 0; i < 4; ++i) cal[6 + i] = (float)(G.c_value[i] - G.c_zero[i]);
    }
    for (int pair = tid; pair < W * W; pair += NT) {
        const int h = pair / W, t = pair - h * W;
        float* o = G.pre + (size_t)pair * kPreStride;
        const float K[9] = {fx, 0, cx, 0, fy, cy, 0, 0, 1}, Ki[9] = {1.0f / fx, 0, -cx / fx, 0, 1.0f / fy, -cy / fy, 0, 0, 1};
        const SE3 ll = SE3::from(G.w2c + t * 12) * SE3::from(G.c2w + h * 12);
        float R[9], tt[3], KR[9];
        for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) R[i * 3 + j] = (float)ll.R(i, j); tt[i] = (float)ll.t(i); }
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) KR[i * 3 + j] = K[i * 3] * R[j] + K[i * 3 + 1] * R[3 + j] + K[i * 3 + 2] * R[6 + j];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) o[i * 3 + j] = KR[i * 3] * Ki[j] + KR[i * 3 + 1] * Ki[3 + j] + KR[i * 3 + 2] * Ki[6 + j];
        for (int i = 0; i < 3; ++i) o[9 + i] = K[i * 3] * tt[0] + K[i * 3 + 1] * tt[1] + K[i * 3 + 2] * tt[2];
        // entries 12..23 (PRE_RTll_0 / PRE_tTll_0 at the linearisation point) and 26 (b0) do not change during optimize(): written by the host
        double a[2];
        aff_from_to(G.ab_exposure[h], G.ab_exposure[t], kScaleA * G.state[h * 10 + 6], kScaleB * G.state[h * 10 + 7], kScaleA * G.state[t * 10 + 6], kScaleB * G.state[t * 10 + 7], a);
        o[24] = (float)a[0]; o[25] = (float)a[1];
        const int idx = h + t * W;
        float dh[8], dt[8];
        for (int i = 0; i < 8; ++i) { dh[i] = (float)(G.state[h * 10 + i] - G.state_zero[h * 10 + i]); dt[i] = (float)(G.state[t * 10 + i] - G.state_zero[t * 10 + i]); }
        for (int j = 0; j < 8; ++j) {
            float s1 = 0, s2 = 0;
            for (int i = 0; i < 8; ++i) { s1 += dh[i] * G.adHostF[(size_t)idx * 64 + i * 8 + j]; s2 += dt[i] * G.adTargetF[(size_t)idx * 64 + i * 8 + j]; }
            o[27 + j] = s1 + s2;
        }
    }
}

int ba_launch_gn(hipStream_t s, const GNDev& G, int iteration, int never_break, double lambda) {
    const size_t n = (size_t)G.n;
    const size_t lds = (n * n + 5 * n + 8) * 8 + n * 4 + 64;
    static size_t lds_allowed = 48 * 1024;
    if (lds > lds_allowed) {
        if (hipFuncSetAttribute((const void*)ba_gn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 1;
        lds_allowed = lds;
    }
    ba_gn_kernel<<<1, kGnThreads, lds, s>>>(G, iteration, never_break, lambda);
    return 0;
}

}  // namespace nalo
