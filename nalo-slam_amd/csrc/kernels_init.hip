// CoarseInitializer::calcResAndGS (reference src/FullSystem/CoarseInitializer.cpp:338-610) for one pyramid level as ONE pass:
// per point the 8-pixel residual / Jacobian rows, the per-point Schur buffer JbBuffer_new (:468-477), the unweighted Accumulator9 of [dp0..dp7, r]
// (:497-513) and, because the reference's EAlpha accumulator never receives anything (:560-572 feed E), alphaOpt is known before the launch, so the
// weighted Schur accumulator acc9SC (:590-612) joins the same pass. 91 per-lane fp32 sums (45 + 45 + E) go through the usual quad-DPP -> LDS rows ->
// fp64 column sums -> block partial -> fp64 finish. doStep (:910-938) is a thread-per-point kernel.
#include "nalo_internal.h"
#include "reduce.h"

namespace nalo {

constexpr int kInitVals = 94, kInitStride = 96;      // 45 (acc9) + 45 (acc9SC) + E + calcEC's {sum rOld^2, sum rNew^2, count}
#define NALO_PAT(i) {kPatternDx[i], kPatternDy[i]}
__constant__ int kInitPattern[8][2] = {NALO_PAT(0), NALO_PAT(1), NALO_PAT(2), NALO_PAT(3), NALO_PAT(4), NALO_PAT(5), NALO_PAT(6), NALO_PAT(7)};   // settings.cpp:297 (ref_constants.h)
#undef NALO_PAT

__device__ __forceinline__ float3 init_interp33(const float4* __restrict__ img, float x, float y, int width) {
    const int ix = (int)x, iy = (int)y;
    const float dx = x - ix, dy = y - iy, dxdy = dx * dy;
    const float4* bp = img + ix + iy * width;
    const float4 p00 = bp[0], p10 = bp[1], p01 = bp[width], p11 = bp[1 + width];
    const float w11 = dxdy, w01 = dy - dxdy, w10 = dx - dxdy, w00 = 1 - dx - dy + dxdy;
    return make_float3(w11 * p11.x + w01 * p01.x + w10 * p10.x + w00 * p00.x, w11 * p11.y + w01 * p01.y + w10 * p10.y + w00 * p00.y,
                       w11 * p11.z + w01 * p01.z + w10 * p10.z + w00 * p00.z);
}

__global__ __launch_bounds__(256) void init_calc_kernel(InitParams P, double* __restrict__ partial) {
    __shared__ float smem[64 * (kInitVals + 1)];
    __shared__ float bsum[kInitStride];
    float acc[kInitVals];
#pragma unroll
    for (int k = 0; k < kInitVals; ++k) acc[k] = 0.f;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < P.n) {
        float maxstep = 1e10f;
        if (!P.isGood[i]) {
            acc[90] += P.energy[i * 2];
            P.energy_new[i * 2] = P.energy[i * 2]; P.energy_new[i * 2 + 1] = P.energy[i * 2 + 1]; P.isGood_new[i] = 0; P.maxstep[i] = maxstep;
        } else {
            const float u = P.u[i], v = P.v[i], idn = P.idepth_new[i];
            float dp[8][8], r[8], jb[10];
#pragma unroll
            for (int k = 0; k < 10; ++k) jb[k] = 0.f;
            bool good = true; float en = 0.f;
#pragma unroll
            for (int idx = 0; idx < 8; ++idx) {
                if (!good) continue;                                          // `break` of the reference: later pixels contribute nothing
                const int dx = kInitPattern[idx][0], dy = kInitPattern[idx][1];
                float pt[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) pt[k] = P.RKi[k * 3] * (u + dx) + P.RKi[k * 3 + 1] * (v + dy) + P.RKi[k * 3 + 2] * 1 + P.t[k] * idn;
                const float uu = pt[0] / pt[2], vv = pt[1] / pt[2], Ku = P.fx * uu + P.cx, Kv = P.fy * vv + P.cy;
                const float new_idepth = idn / pt[2];
                if (!(Ku > 1 && Kv > 1 && Ku < P.wl - 2 && Kv < P.hl - 2 && new_idepth > 0)) { good = false; continue; }
                const float3 hit = init_interp33(P.colorNew, Ku, Kv, P.wl);
                const float rlR = init_interp33(P.colorRef, u + dx, v + dy, P.wl).x;
                if (!isfinite(rlR) || !isfinite(hit.x)) { good = false; continue; }
                const float residual = hit.x - P.r2new0 * rlR - P.r2new1;
                const float ar = fabsf(residual);
                float hw = ar < kHuberTH ? 1 : kHuberTH / ar;
                en += hw * residual * residual * (2 - hw);
                const float dxdd = (P.t[0] - P.t[2] * uu) / pt[2], dydd = (P.t[1] - P.t[2] * vv) / pt[2];
                if (hw < 1) hw = sqrtf(hw);
                const float dxInterp = hw * hit.y * P.fx, dyInterp = hw * hit.z * P.fy;
                dp[0][idx] = new_idepth * dxInterp;
                dp[1][idx] = new_idepth * dyInterp;
                dp[2][idx] = -new_idepth * (uu * dxInterp + vv * dyInterp);
                dp[3][idx] = -uu * vv * dxInterp - (1 + vv * vv) * dyInterp;
                dp[4][idx] = (1 + uu * uu) * dxInterp + uu * vv * dyInterp;
                dp[5][idx] = -vv * dxInterp + uu * dyInterp;
                dp[6][idx] = -hw * P.r2new0 * rlR;
                dp[7][idx] = -hw * 1;
                const float dd = dxInterp * dxdd + dyInterp * dydd;
                r[idx] = hw * residual;
                const float nx = dxdd * P.fx, ny = dydd * P.fy;
                const float ms = 1.0f / sqrtf(nx * nx + ny * ny);
                if (ms < maxstep) maxstep = ms;
#pragma unroll
                for (int k = 0; k < 8; ++k) jb[k] += dp[k][idx] * dd;
                jb[8] += r[idx] * dd;
                jb[9] += dd * dd;
            }
            P.maxstep[i] = maxstep;
            if (!good || en > P.outlierTH[i] * 20) {
                acc[90] += P.energy[i * 2];
                P.isGood_new[i] = 0; P.energy_new[i * 2] = P.energy[i * 2]; P.energy_new[i * 2 + 1] = P.energy[i * 2 + 1];
#pragma unroll
                for (int k = 0; k < 10; ++k) P.Jb[i * 10 + k] = jb[k];
            } else {
                acc[90] += en;
                P.isGood_new[i] = 1; P.energy_new[i * 2] = en; P.energy_new[i * 2 + 1] = (idn - 1) * (idn - 1);
                if (P.idepth) {                                               // calcEC (:634-655) reads exactly what this pass has at hand: the regulariser's old / new energy
                    const float iR = P.iR[i], rOld = P.idepth[i] - iR, rNew = idn - iR;
                    acc[91] += rOld * rOld; acc[92] += rNew * rNew; acc[93] += 1.f;
                }
#pragma unroll
                for (int idx = 0; idx < 8; ++idx) {                          // acc9: upper triangle of [dp0..dp7, r] outer products (constant indices)
                    float J[9];
#pragma unroll
                    for (int k = 0; k < 8; ++k) J[k] = dp[k][idx];
                    J[8] = r[idx];
#pragma unroll
                    for (int a = 0; a < 9; ++a)
#pragma unroll
                        for (int b = a; b < 9; ++b) acc[a * 9 - a * (a - 1) / 2 + (b - a)] += J[a] * J[b];
                }
                // per-point Schur complement (:590-612)
                P.lastHessian_new[i] = jb[9];
                jb[8] += P.alphaOpt * (idn - 1);
                jb[9] += P.alphaOpt;
                if (P.alphaOpt == 0) { jb[8] += P.couplingWeight * (idn - P.iR[i]); jb[9] += P.couplingWeight; }
                jb[9] = 1 / (1 + jb[9]);
#pragma unroll
                for (int k = 0; k < 10; ++k) P.Jb[i * 10 + k] = jb[k];
                float J[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) J[k] = jb[k];
                const float w = jb[9];
#pragma unroll
                for (int a = 0; a < 9; ++a) {                                // updateSingleWeighted (MatrixAccumulators.h:1242-1315)
                    acc[45 + a * 9 - a * (a - 1) / 2] += J[a] * J[a] * w;
                    J[a] *= w;
#pragma unroll
                    for (int b = a + 1; b < 9; ++b) acc[45 + a * 9 - a * (a - 1) / 2 + (b - a)] += J[b] * J[a];
                }
            }
        }
    }
    block_reduce_cols<kInitVals, 256>(acc, smem, bsum);
    if (threadIdx.x < kInitVals) partial[(size_t)blockIdx.x * kInitStride + threadIdx.x] = (double)bsum[threadIdx.x];
}
__global__ __launch_bounds__(1024) void init_finish_kernel(const double* __restrict__ partial, int nblocks, double* __restrict__ out) {
    __shared__ double part[8][128];
    const int j = threadIdx.x & 127, g = threadIdx.x >> 7;
    double s = 0;
    if (j < kInitVals) for (int b = g; b < nblocks; b += 8) s += partial[(size_t)b * kInitStride + j];
    part[g][j] = s;
    __syncthreads();
    if (g == 0 && j < kInitVals) { double t = 0; for (int k = 0; k < 8; ++k) t += part[k][j]; out[j] = t; }
}
// CoarseInitializer::doStep (:910-938)
__global__ __launch_bounds__(256) void init_do_step_kernel(int n, const uint8_t* __restrict__ isGood, const float* __restrict__ Jb, const float* __restrict__ maxstep,
                                                          const float* __restrict__ idepth, float lambda, InitInc inc, float* __restrict__ idepth_new) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !isGood[i]) return;
    const float* jb = Jb + (size_t)i * 10;
    float dot = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) dot += jb[k] * inc.v[k];
    const float b = jb[8] + dot;
    float step = -b * jb[9] / (1 + lambda);
    float ms = 0.25f * maxstep[i];
    if (ms > 1e10f) ms = 1e10f;
    if (step > ms) step = ms;
    if (step < -ms) step = -ms;
    float nid = idepth[i] + step;
    if (nid < 1e-3f) nid = 1e-3f;
    if (nid > 50) nid = 50;
    idepth_new[i] = nid;
}

// CoarseInitializer::applyStep (:939-956) on the resident arrays (the JbBuffer swap is a pointer swap on the host)
__global__ __launch_bounds__(256) void init_apply_step_kernel(int n, uint8_t* __restrict__ isGood, const uint8_t* __restrict__ isGood_new, float* __restrict__ idepth,
                                                             float* __restrict__ idepth_new, const float* __restrict__ iR, float* __restrict__ energy,
                                                             const float* __restrict__ energy_new, float* __restrict__ lastHessian, const float* __restrict__ lastHessian_new) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (!isGood[i]) { const float r = iR[i]; idepth[i] = r; idepth_new[i] = r; return; }
    energy[2 * i] = energy_new[2 * i]; energy[2 * i + 1] = energy_new[2 * i + 1];
    isGood[i] = isGood_new[i];
    idepth[i] = idepth_new[i];
    lastHessian[i] = lastHessian_new[i];
}

int init_calc_launch(nalo_ctx* c, InitParams& P, int lvl, double* sums) {
    P.wl = c->wl[lvl]; P.hl = c->hl[lvl];
    const int nb = (P.n + 255) / 256;
    NALO_HIP(c, c->trk_partial.reserve(((size_t)nb * kInitStride + 128) * 2));          // doubles in a float buffer
    double* partial = (double*)c->trk_partial.p;
    init_calc_kernel<<<nb, 256, 0, c->stream>>>(P, partial);
    init_finish_kernel<<<1, 1024, 0, c->stream>>>(partial, nb, sums);
    NALO_HIP(c, hipGetLastError());
    return NALO_OK;
}
int init_do_step_launch(nalo_ctx* c, int n, const uint8_t* isGood, const float* Jb, const float* maxstep, const float* idepth, float lambda, const float inc[8], float* idepth_new) {
    InitInc I; for (int k = 0; k < 8; ++k) I.v[k] = inc[k];
    if (n > 0) init_do_step_kernel<<<(n + 255) / 256, 256, 0, c->stream>>>(n, isGood, Jb, maxstep, idepth, lambda, I, idepth_new);
    NALO_HIP(c, hipGetLastError());
    return NALO_OK;
}
int init_apply_step_launch(nalo_ctx* c, int n, uint8_t* isGood, const uint8_t* isGood_new, float* idepth, float* idepth_new, const float* iR, float* energy, const float* energy_new,
                           float* lastHessian, const float* lastHessian_new) {
    if (n > 0) init_apply_step_kernel<<<(n + 255) / 256, 256, 0, c->stream>>>(n, isGood, isGood_new, idepth, idepth_new, iR, energy, energy_new, lastHessian, lastHessian_new);
    NALO_HIP(c, hipGetLastError());
    return NALO_OK;
}

}  // namespace nalo
