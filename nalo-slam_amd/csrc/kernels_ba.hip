// Sliding-window photometric BA kernels for gfx950 (reference paths relative to src/).
//
//  ba_linearize_kernel   a5+a6+a7  PointFrameResidual::linearize + applyRes + EFResidual::takeDataF
//                                  (FullSystem/Residuals.cpp:78-274,306-328; OptimizationBackend/EnergyFunctionalStructs.cpp:39-50)
//                                  fused with AccumulatedTopHessianSSE::addPoint<0|2> + AccumulatorApprox
//                                  (OptimizationBackend/AccumulatedTopHessian.cpp:39-162, MatrixAccumulators.h:754-915)
//                                  and, in marginalisation mode, EFResidual::fixLinearizationF (EnergyFunctionalStructs.cpp:89-115).
//                                  The 296-byte RawResidualJacobian never leaves registers.
//  ba_sc_kernel          a9        AccumulatedSCHessianSSE::addPoint (OptimizationBackend/AccumulatedSCHessian.cpp:34-77) as a
//                                  per-host weighted SYRK  G_h = sum_p HdiF_p a_p a_p^T,  a_p = [JpJdF(t) for t != h | Hcd | bdSum]:
//                                  accD = the 8x8 blocks, accE = the Hcd columns, accEB = the bdSum column, accHcc/accbc = the corner.
//  ba_reduce_*           fp64 finish of the per-block fp32 partials (replaces the per-thread replicas summed in stitchDoubleInternal)
//  ba_stitch_*           a8+a10    stitchDouble for both systems as  H~ = sum_b S_b M_b S_b^T  with S_b built from adHost/adTarget
//                                  (AccumulatedTopHessian.cpp:171-303, AccumulatedSCHessian.cpp:78-219)
//  ba_resub_kernel       a12       EnergyFunctional::resubstituteFPt (OptimizationBackend/EnergyFunctional.cpp:291-317)
//  ba_step_kernel                  point part of FullSystem::doStepFromBackup (FullSystem/FullSystemOptimize.cpp:269-277)
//  ba_energy_th_kernel             FullSystem::setNewFrameEnergyTH (FullSystemOptimize.cpp:95-143): exact order statistic by radix select
//
// All kernels are HBM/latency bound (no MFMA): 16-byte texel gathers, coalesced [target][point] slot arrays,
// block-uniform precalc through scalar loads, DPP + LDS reductions, no float atomics on any sum.
#include "nalo_internal.h"
#include "ba_device.h"
#include "reduce.h"

namespace nalo {

__device__ __forceinline__ float4 ba_bilinear(const float4* __restrict__ img, float x, float y, int width) {
    const int ix = (int)x, iy = (int)y;                 // util/globalFuncs.h:75-89
    const float dx = x - ix, dy = y - iy, dxdy = dx * dy;
    const float4* bp = img + ix + iy * width;
    const float4 p00 = bp[0], p10 = bp[1], p01 = bp[width], p11 = bp[1 + width];
    const float w11 = dxdy, w01 = dy - dxdy, w10 = dx - dxdy, w00 = 1 - dx - dy + dxdy;
    float4 r;
    r.x = w11 * p11.x + w01 * p01.x + w10 * p10.x + w00 * p00.x;
    r.y = w11 * p11.y + w01 * p01.y + w10 * p10.y + w00 * p00.y;
    r.z = w11 * p11.z + w01 * p01.z + w10 * p10.z + w00 * p00.z;
    r.w = 0.f;
    return r;
}

// MODE 0: active residuals (optimize). MODE 2: marginalisation of the flagged points (resApprox = res_toZeroF).
// FIX: linearizeAll(true) semantics — residuals that do not end IN are dropped, centerProjectedTo/relBS are stored.
template <int MODE, int FIX>
__global__ __launch_bounds__(kBlk) void ba_linearize_kernel(BADev B) {
    __shared__ float smem[(kBlk / 4) * (kTopVals + 1)];
    const int b = blockIdx.x, tid = threadIdx.x, d = b * kBlk + tid;
    const int h = B.blk_host[b], W = B.W;
    const uint8_t pf = B.pt_flags[d];
    const bool pvalid = (pf & PT_VALID) && (MODE == 0 || (pf & PT_MARG));
    const float4 geo = B.pt_geo[d];
    const float pu = geo.x, pv = geo.y, idepth = geo.z, idz = geo.w;
    float color[8], wgt[8];
    {
        const float4 c0 = B.pt_col0[d], c1 = B.pt_col1[d], w0 = B.pt_w0[d], w1 = B.pt_w1[d];
        color[0] = c0.x; color[1] = c0.y; color[2] = c0.z; color[3] = c0.w; color[4] = c1.x; color[5] = c1.y; color[6] = c1.z; color[7] = c1.w;
        wgt[0] = w0.x; wgt[1] = w0.y; wgt[2] = w0.z; wgt[3] = w0.w; wgt[4] = w1.x; wgt[5] = w1.y; wgt[6] = w1.z; wgt[7] = w1.w;
    }
    const float wM3G = (float)(B.w - 3), hM3G = (float)(B.h - 3);
    const float thH = B.frameTH[h];
    float Hdd_acc = 0.f, bd_acc = 0.f, Hcd_acc[4] = {0.f, 0.f, 0.f, 0.f}, relbs_max = 0.f;
    int ngood = 0;
    const float KliP0 = (pu - B.cxl) * B.fxli, KliP1 = (pv - B.cyl) * B.fyli;     // ResidualProjections.h:70-73
    const float dd = idepth - idz;                                                  // EFPoint::deltaF

    for (int t = 0; t < W; ++t) {
        if (t == h) continue;
        const size_t si = (size_t)t * B.Ppad + d;
        const float* pc = B.pre + (size_t)(h * W + t) * kPreStride;                 // block-uniform: scalar loads
        uint8_t st = B.rs_state[si];
        const bool exists = pvalid && (st & RS_EXISTS) && (MODE == 2 || !(st & RS_LINEARIZED));
        float v[kTopVals];
#pragma unroll
        for (int i = 0; i < kTopVals; ++i) v[i] = 0.f;
        float enew = -1.f;
        if (exists) {
            float2 en = B.rs_energy[si];
            int state = st & RS_STATE_MASK;
            if (MODE == 2) { en.x = 0.f; en.y = 0.f; state = 0; st &= ~RS_LINEARIZED; }        // resetOOB + isLinearized=false (FullSystem.cpp:978-981)
            int newState = 2;
            float energy = en.x;
            bool full = false;
            float Jpdxi0[6], Jpdxi1[6], Jpdc0[4], Jpdc1[4], Jpdd0 = 0.f, Jpdd1 = 0.f, cKu = 0.f, cKv = 0.f, cId = 0.f;
            float a = 0.f, bb = 0.f, c = 0.f, jab00 = 0.f, jab01 = 0.f, jab10 = 0.f, jab11 = 0.f, ab00 = 0.f, ab01 = 0.f, ab11 = 0.f;
            float JIr0 = 0.f, JIr1 = 0.f, Jabr0 = 0.f, Jabr1 = 0.f, rr = 0.f;
            if (state == 1) { newState = 1; }                                                     // Residuals.cpp:82-83
            else {
                // ---- centre projection at idepth_zero (projectPoint, ResidualProjections.h:61-87)
                const float p0 = pc[12] * KliP0 + pc[13] * KliP1 + pc[14] + pc[21] * idz;
                const float p1 = pc[15] * KliP0 + pc[16] * KliP1 + pc[17] + pc[22] * idz;
                const float p2 = pc[18] * KliP0 + pc[19] * KliP1 + pc[20] + pc[23] * idz;
                const float drescale = 1.0f / p2, new_idepth = idz * drescale;
                const float u = p0 * drescale, vv = p1 * drescale;
                const float Ku0 = u * B.fxl + B.cxl, Kv0 = vv * B.fyl + B.cyl;
                bool ok = (drescale > 0.f) && Ku0 > 1.1f && Kv0 > 1.1f && Ku0 < wM3G && Kv0 < hM3G;
                // ---- the 8 pattern pixels at the current idepth (projectPoint, ResidualProjections.h:47-57)
                float Kus[8], Kvs[8];
                const int pdx[8] = {0, -1, 1, -2, 0, 2, -1, 0}, pdy[8] = {-2, -1, -1, 0, 0, 0, 1, 2};   // util/settings.cpp:297
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float x = pu + (float)pdx[k], y = pv + (float)pdy[k];
                    const float q0 = pc[0] * x + pc[1] * y + pc[2] + pc[9] * idepth;
                    const float q1 = pc[3] * x + pc[4] * y + pc[5] + pc[10] * idepth;
                    const float q2 = pc[6] * x + pc[7] * y + pc[8] + pc[11] * idepth;
                    Kus[k] = q0 / q2; Kvs[k] = q1 / q2;
                    ok = ok && Kus[k] > 1.1f && Kvs[k] > 1.1f && Kus[k] < wM3G && Kvs[k] < hM3G;
                }
                if (!ok) { newState = 1; }
                else {
                    cKu = Ku0; cKv = Kv0; cId = new_idepth;
                    const float t0x = pc[21], t0y = pc[22], t0z = pc[23];
                    Jpdd0 = drescale * (t0x - t0z * u) * kScaleIdepth * B.fxl;                         // Residuals.cpp:116-117
                    Jpdd1 = drescale * (t0y - t0z * vv) * kScaleIdepth * B.fyl;
                    Jpdc0[2] = drescale * (pc[18] * u - pc[12]);                                     // :123-131
                    Jpdc0[3] = B.fxl * drescale * (pc[19] * u - pc[13]) * B.fyli;
                    Jpdc0[0] = KliP0 * Jpdc0[2]; Jpdc0[1] = KliP1 * Jpdc0[3];
                    Jpdc1[2] = B.fyl * drescale * (pc[18] * vv - pc[15]) * B.fxli;
                    Jpdc1[3] = drescale * (pc[19] * vv - pc[16]);
                    Jpdc1[0] = KliP0 * Jpdc1[2]; Jpdc1[1] = KliP1 * Jpdc1[3];
                    Jpdc0[0] = (Jpdc0[0] + u) * kScaleF; Jpdc0[1] *= kScaleF; Jpdc0[2] = (Jpdc0[2] + 1) * kScaleC; Jpdc0[3] *= kScaleC;   // :133-141
                    Jpdc1[0] *= kScaleF; Jpdc1[1] = (Jpdc1[1] + vv) * kScaleF; Jpdc1[2] *= kScaleC; Jpdc1[3] = (Jpdc1[3] + 1) * kScaleC;
                    Jpdxi0[0] = new_idepth * B.fxl; Jpdxi0[1] = 0.f; Jpdxi0[2] = -new_idepth * u * B.fxl;   // :144-156
                    Jpdxi0[3] = -u * vv * B.fxl; Jpdxi0[4] = (1 + u * u) * B.fxl; Jpdxi0[5] = -vv * B.fxl;
                    Jpdxi1[0] = 0.f; Jpdxi1[1] = new_idepth * B.fyl; Jpdxi1[2] = -new_idepth * vv * B.fyl;
                    Jpdxi1[3] = -(1 + vv * vv) * B.fyl; Jpdxi1[4] = u * vv * B.fyl; Jpdxi1[5] = u * B.fyl;
                    float jx = 0.f, jy = 0.f;
                    if (MODE == 2) {                                                                 // Jp*delta (EnergyFunctionalStructs.cpp:94-99)
#pragma unroll
                        for (int i = 0; i < 6; ++i) { jx += Jpdxi0[i] * pc[27 + i]; jy += Jpdxi1[i] * pc[27 + i]; }
                        float cxs = 0.f, cys = 0.f;
#pragma unroll
                        for (int i = 0; i < 4; ++i) { cxs += Jpdc0[i] * B.cDelta[i]; cys += Jpdc1[i] * B.cDelta[i]; }
                        jx = jx + cxs + Jpdd0 * dd; jy = jy + cys + Jpdd1 * dd;
                    }
                    const float affLL0 = pc[24], affLL1 = pc[25], b0 = pc[26];
                    const float4* img = B.img[t];
                    float energyLeft = 0.f, wJI2_sum = 0.f;
                    bool finite_ok = true;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {                                                    // :183-245
                        float4 hit = ba_bilinear(img, Kus[k], Kvs[k], B.w);
                        const float residual = hit.x - (affLL0 * color[k] + affLL1);
                        const float drdA = color[k] - b0;
                        finite_ok = finite_ok && isfinite(hit.x);
                        float wgt_k = sqrtf(kOutlierTHSumComponent / (kOutlierTHSumComponent + (hit.y * hit.y + hit.z * hit.z)));
                        wgt_k = 0.5f * (wgt_k + wgt[k]);
                        const float ar = fabsf(residual);
                        float hw = ar < kHuberTH ? 1.f : kHuberTH / ar;
                        energyLeft += wgt_k * wgt_k * hw * residual * residual * (2.f - hw);
                        if (hw < 1.f) hw = sqrtf(hw);
                        hw = hw * wgt_k;
                        const float jI0 = hit.y * hw, jI1 = hit.z * hw, resF = residual * hw, jA = drdA * hw, jB = hw;
                        a += jI0 * jI0; c += jI1 * jI1; bb += jI0 * jI1;
                        jab00 += jA * jI0; jab01 += jA * jI1; jab10 += jB * jI0; jab11 += jB * jI1;
                        ab00 += jA * jA; ab01 += jA * jB; ab11 += jB * jB;
                        wJI2_sum += hw * hw * (jI0 * jI0 + jI1 * jI1);      // on the already hw-scaled gradient, as Residuals.cpp:215-239
                        float ra = resF;                                                             // mode 2: res_toZeroF (:103-111)
                        if (MODE == 2) ra = resF - jI0 * jx - jI1 * jy - jA * pc[33] - jB * pc[34];
                        JIr0 += ra * jI0; JIr1 += ra * jI1; Jabr0 += ra * jA; Jabr1 += ra * jB; rr += ra * ra;
                    }
                    if (!finite_ok) { newState = 1; }
                    else {
                        full = true;
                        enew = energyLeft;
                        const float th = fmaxf(thH, B.frameTH[t]);
                        if (energyLeft > th || wJI2_sum < 2.f) { energyLeft = th; newState = 2; } else newState = 0;   // :262-270
                        en.y = energyLeft;
                        energy = energyLeft;
                    }
                }
            }
            // ---- applyRes(true) (Residuals.cpp:306-328)
            bool active = false;
            if (state != 1) {
                active = (newState == 0);
                state = newState;
                en.x = en.y;
            }
            st = (uint8_t)((st & ~(RS_STATE_MASK | RS_ACTIVE)) | state | (active ? RS_ACTIVE : 0));
            if (MODE == 2 && active) st |= RS_LINEARIZED;
            if (FIX && !active) st &= ~(RS_EXISTS | RS_ACTIVE);                                      // toRemove (FullSystemOptimize.cpp:81-84,184-205)
            B.rs_state[si] = st;
            B.rs_energy[si] = en;
            v[92] = energy;
            if (active && full) {
                ngood++;
                // ---- takeDataF (EnergyFunctionalStructs.cpp:39-50)
                const float a0 = a * Jpdd0 + bb * Jpdd1, a1 = bb * Jpdd0 + c * Jpdd1;
                float4 j0, j1;
                j0.x = Jpdxi0[0] * a0 + Jpdxi1[0] * a1; j0.y = Jpdxi0[1] * a0 + Jpdxi1[1] * a1; j0.z = Jpdxi0[2] * a0 + Jpdxi1[2] * a1;
                j0.w = Jpdxi0[3] * a0 + Jpdxi1[3] * a1; j1.x = Jpdxi0[4] * a0 + Jpdxi1[4] * a1; j1.y = Jpdxi0[5] * a0 + Jpdxi1[5] * a1;
                j1.z = jab00 * Jpdd0 + jab01 * Jpdd1; j1.w = jab10 * Jpdd0 + jab11 * Jpdd1;
                B.rs_jp0[si] = j0; B.rs_jp1[si] = j1;
                // ---- AccumulatorApprox::update / updateTopRight / updateBotRight (AccumulatedTopHessian.cpp:115-129)
                float x[10], y[10], ax[10], cy[10];
#pragma unroll
                for (int i = 0; i < 4; ++i) { x[i] = Jpdc0[i]; y[i] = Jpdc1[i]; }
#pragma unroll
                for (int i = 0; i < 6; ++i) { x[4 + i] = Jpdxi0[i]; y[4 + i] = Jpdxi1[i]; }
#pragma unroll
                for (int i = 0; i < 10; ++i) { ax[i] = a * x[i] + bb * y[i]; cy[i] = bb * x[i] + c * y[i]; }
                int idx = 0;
#pragma unroll
                for (int r = 0; r < 10; ++r)
#pragma unroll
                    for (int cc = r; cc < 10; ++cc) { v[idx] = ax[r] * x[cc] + cy[r] * y[cc]; ++idx; }
#pragma unroll
                for (int i = 0; i < 10; ++i) {
                    v[55 + 3 * i + 0] = x[i] * jab00 + y[i] * jab01;
                    v[55 + 3 * i + 1] = x[i] * jab10 + y[i] * jab11;
                    v[55 + 3 * i + 2] = x[i] * JIr0 + y[i] * JIr1;
                }
                v[85] = ab00; v[86] = ab01; v[87] = Jabr0; v[88] = ab11; v[89] = Jabr1; v[90] = rr;
                v[91] = 1.f;
                bd_acc += JIr0 * Jpdd0 + JIr1 * Jpdd1;                                               // :132-135
                Hdd_acc += a0 * Jpdd0 + a1 * Jpdd1;
#pragma unroll
                for (int i = 0; i < 4; ++i) Hcd_acc[i] += Jpdc0[i] * a0 + Jpdc1[i] * a1;
                if (FIX || MODE == 2) {
                    // relBS of FullSystemOptimize.cpp:69-71 + centerProjectedTo (makeCoarseDepthL0 input)
                    const float i0 = pc[0] * pu + pc[1] * pv + pc[2], i1 = pc[3] * pu + pc[4] * pv + pc[5], i2 = pc[6] * pu + pc[7] * pv + pc[8];
                    const float q0 = i0 + pc[9] * idepth, q1 = i1 + pc[10] * idepth, q2 = i2 + pc[11] * idepth;
                    const float ex = i0 / i2 - q0 / q2, ey = i1 / i2 - q1 / q2;
                    const float relBS = 0.01f * sqrtf(ex * ex + ey * ey);
                    relbs_max = fmaxf(relbs_max, relBS);
                    B.rs_cpt[si] = make_float4(cKu, cKv, cId, relBS);
                }
            }
        }
        if (t == W - 1) {
            B.en_new[d] = enew;
            if (MODE == 0 && enew >= 0.f) atomicAdd(&B.th_hist_hi[__float_as_uint(enew) >> 16], 1u);   // integer atomics: order independent
        }
        block_reduce_cols<kTopVals, kBlk>(v, smem, B.top_partial + ((size_t)b * W + t) * kTopStride);
    }
    if (h == W - 1) B.en_new[d] = -1.f;
    // per-point sums: EFPoint::{Hdd,bd,Hcd}_accAF (mode 0) / _accLF (mode 2, AF zeroed: AccumulatedTopHessian.cpp:140-157)
    if (pvalid) {
        B.pt_acc[d] = make_float4(Hdd_acc, bd_acc, 0.f, 0.f);
        B.pt_hcd[d] = make_float4(Hcd_acc[0], Hcd_acc[1], Hcd_acc[2], Hcd_acc[3]);
        B.pt_ngood[d] = (uint8_t)ngood;
        if (FIX || MODE == 2) B.pt_relbs[d] = relbs_max;
    }
}

// resetOOB for every active residual at the start of optimize() (FullSystemOptimize.cpp:412-429, Residuals.h:88-94)
__global__ __launch_bounds__(256) void ba_reset_oob_kernel(uint8_t* __restrict__ rs_state, float2* __restrict__ rs_energy, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t st = rs_state[i];
    if ((st & RS_EXISTS) && !(st & RS_LINEARIZED)) { rs_state[i] = (uint8_t)(st & ~RS_STATE_MASK); rs_energy[i] = make_float2(0.f, 0.f); }
}

// ------------------------------------------------------------------------------------------------ a9: per-host weighted SYRK
// Block = the same 256 points as the linearize block. Thread (ty,tx) of a 16x16 grid owns a TxT tile of G (NPL = 16T columns).
template <int T>
__global__ __launch_bounds__(256) void ba_sc_kernel(BADev B, int shiftPriorToZero, float priorScaleMarg, int margOnly) {
    constexpr int NPL = 16 * T, SUB = 64;
    __shared__ float A[SUB * NPL];
    __shared__ float Wt[SUB];
    const int b = blockIdx.x, tid = threadIdx.x, h = B.blk_host[b], W = B.W;
    const int ty = tid >> 4, tx = tid & 15;
    // fp32 products and short fp32 runs (16 points), flushed into fp64: keeps the block partial good to ~1e-7 so the
    // cancellation in H_A - H_sc does not amplify summation noise into the poses
    constexpr int RUN = 16;
    float acc[T][T];
    double acc64[T][T];
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < T; ++j) { acc[i][j] = 0.f; acc64[i][j] = 0.0; }
    for (int sub = 0; sub < kBlk / SUB; ++sub) {
        const int d0 = b * kBlk + sub * SUB;
        // ---- stage SUB operand rows: [JpJdF(t != h) (8 each) | Hcd (4) | bdSum | 0..]
        for (int e = tid; e < SUB * (NPL / 4); e += 256) {
            const int r = e / (NPL / 4), q = e - r * (NPL / 4);       // q-th float4 of row r
            const int d = d0 + r;
            const uint8_t pf = B.pt_flags[d];
            const bool pvalid = (pf & PT_VALID) && (!margOnly || (pf & PT_MARG));
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            const int g = q >> 1;                                     // compact target slot
            if (pvalid && g < W - 1) {
                const int t = g < h ? g : g + 1;
                const size_t si = (size_t)t * B.Ppad + d;
                if (B.rs_state[si] & RS_ACTIVE) val = (q & 1) ? B.rs_jp1[si] : B.rs_jp0[si];
            } else if (pvalid && q == 2 * (W - 1)) {
                val = B.pt_hcd[d];
            }
            *reinterpret_cast<float4*>(&A[r * NPL + 4 * q]) = val;
        }
        __syncthreads();
        if (tid < SUB) {                                              // per point: AccumulatedSCHessian.cpp:36-57
            const int d = d0 + tid;
            const uint8_t pf = B.pt_flags[d];
            const bool pvalid = (pf & PT_VALID) && (!margOnly || (pf & PT_MARG));
            float wgt = 0.f;
            if (pvalid) {
                float4 pa = B.pt_acc[d];
                if (B.pt_ngood[d] == 0) { pa.z = 0.f; pa.w = 0.f; }
                else {
                    float prior = B.pt_prior[d];
                    if (margOnly) { prior *= priorScaleMarg; B.pt_prior[d] = prior; }          // EnergyFunctional.cpp:630
                    float Hs = pa.x + prior;                          // Hdd_accAF + Hdd_accLF + priorF (only one of AF/LF is live)
                    if (Hs < 1e-10f) Hs = 1e-10f;
                    pa.z = (float)(1.0 / (double)Hs);
                    pa.w = pa.y;
                    if (shiftPriorToZero) { const float4 geo = B.pt_geo[d]; pa.w += prior * (geo.z - geo.w); }
                    wgt = pa.z;
                }
                B.pt_acc[d] = pa;
                A[tid * NPL + 8 * (W - 1) + 4] = pa.w;                // bdSum column
            }
            Wt[tid] = wgt;
        }
        __syncthreads();
        for (int k0 = 0; k0 < SUB; k0 += RUN) {
#pragma unroll 4
            for (int k = k0; k < k0 + RUN; ++k) {
                const float wk = Wt[k];
                float ai[T], aj[T];
#pragma unroll
                for (int i = 0; i < T; ++i) { ai[i] = wk * A[k * NPL + ty * T + i]; aj[i] = A[k * NPL + tx * T + i]; }
#pragma unroll
                for (int i = 0; i < T; ++i)
#pragma unroll
                    for (int j = 0; j < T; ++j) acc[i][j] += ai[i] * aj[j];
            }
#pragma unroll
            for (int i = 0; i < T; ++i)
#pragma unroll
                for (int j = 0; j < T; ++j) { acc64[i][j] += (double)acc[i][j]; acc[i][j] = 0.f; }
        }
        __syncthreads();
    }
    float* out = B.sc_partial + (size_t)b * NPL * NPL;
#pragma unroll
    for (int i = 0; i < T; ++i)
#pragma unroll
        for (int j = 0; j < T; ++j) out[(ty * T + i) * NPL + tx * T + j] = (float)acc64[i][j];
}

void ba_launch_sc(hipStream_t s, const BADev& B, int T, int shift, float priorScaleMarg, int margOnly) {
    switch (T) {
        case 1: ba_sc_kernel<1><<<B.nblocks, 256, 0, s>>>(B, shift, priorScaleMarg, margOnly); break;
        case 2: ba_sc_kernel<2><<<B.nblocks, 256, 0, s>>>(B, shift, priorScaleMarg, margOnly); break;
        case 3: ba_sc_kernel<3><<<B.nblocks, 256, 0, s>>>(B, shift, priorScaleMarg, margOnly); break;
        case 4: ba_sc_kernel<4><<<B.nblocks, 256, 0, s>>>(B, shift, priorScaleMarg, margOnly); break;
        case 5: ba_sc_kernel<5><<<B.nblocks, 256, 0, s>>>(B, shift, priorScaleMarg, margOnly); break;
        case 6: ba_sc_kernel<6><<<B.nblocks, 256, 0, s>>>(B, shift, priorScaleMarg, margOnly); break;
        case 7: ba_sc_kernel<7><<<B.nblocks, 256, 0, s>>>(B, shift, priorScaleMarg, margOnly); break;
        default: ba_sc_kernel<8><<<B.nblocks, 256, 0, s>>>(B, shift, priorScaleMarg, margOnly); break;
    }
}
void ba_launch_linearize(hipStream_t s, const BADev& B, int mode, int fix) {
    if (mode == 2) ba_linearize_kernel<2, 0><<<B.nblocks, kBlk, 0, s>>>(B);
    else if (fix) ba_linearize_kernel<0, 1><<<B.nblocks, kBlk, 0, s>>>(B);
    else ba_linearize_kernel<0, 0><<<B.nblocks, kBlk, 0, s>>>(B);
}
void ba_launch_reset_oob(hipStream_t s, const BADev& B) {
    const size_t n = (size_t)B.W * B.Ppad;
    ba_reset_oob_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(B.rs_state, B.rs_energy, n);
}

// ------------------------------------------------------------------------------------------------ fp64 finish of the partials
// acc13[(h + t*W)][169] (full symmetric 13x13, AccumulatorApprox::finish layout MatrixAccumulators.h:626-647), misc[h+t*W] = {count, energy}
__global__ __launch_bounds__(128) void ba_reduce_top_kernel(const float* __restrict__ top_partial, const int* __restrict__ host_blk /* [W+1] */,
                                                            int W, double* __restrict__ acc13, double* __restrict__ misc) {
    __shared__ double sums[kTopVals];
    const int h = blockIdx.x % W, t = blockIdx.x / W, j = threadIdx.x;
    if (j < kTopVals) {
        double s = 0;
        if (h != t) for (int b = host_blk[h]; b < host_blk[h + 1]; ++b) s += (double)top_partial[((size_t)b * W + t) * kTopStride + j];
        sums[j] = s;
    }
    __syncthreads();
    double* H = acc13 + (size_t)(h + t * W) * 169;
    for (int e = j; e < 169; e += blockDim.x) {
        int r = e / 13, c = e % 13;
        if (r > c) { const int tmp = r; r = c; c = tmp; }
        int idx;
        if (c < 10) idx = r * 10 - r * (r - 1) / 2 + (c - r);                 // upper triangle of the 10x10, row-major r<=c
        else if (r < 10) idx = 55 + 3 * r + (c - 10);                         // TopRight 10x3
        else { static const int br[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}}; idx = 85 + br[r - 10][c - 10]; }
        H[e] = sums[idx];
    }
    if (j == 0) { misc[2 * (h + t * W)] = sums[91]; misc[2 * (h + t * W) + 1] = sums[92]; }
}
__global__ __launch_bounds__(256) void ba_reduce_sc_kernel(const float* __restrict__ sc_partial, const int* __restrict__ host_blk, int NPL2, double* __restrict__ G) {
    const int h = blockIdx.y, e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= NPL2) return;
    double s = 0;
    for (int b = host_blk[h]; b < host_blk[h + 1]; ++b) s += (double)sc_partial[(size_t)b * NPL2 + e];
    G[(size_t)h * NPL2 + e] = s;
}
void ba_launch_reduce(hipStream_t s, const BADev& B, const int* host_blk, int NPL, double* acc13, double* misc, double* G, bool top, bool sc) {
    if (top) ba_reduce_top_kernel<<<B.W * B.W, 128, 0, s>>>(B.top_partial, host_blk, B.W, acc13, misc);
    if (sc) ba_reduce_sc_kernel<<<dim3((NPL * NPL + 255) / 256, B.W), 256, 0, s>>>(B.sc_partial, host_blk, NPL * NPL, G);
}

// ------------------------------------------------------------------------------------------------ stitch:  H~ = sum_b S_b M_b S_b^T  (fp64)
// S_b ((8W+5) x m) is sparse: a frame row carries one 8-wide adjoint block per slot it takes part in. It is kept in CSR
// (rowptr over [b][row], col, val), built on the host from adHost/adTarget. The last row (index 8W+4) selects the
// residual / bdSum column, so column 8W+4 of H~ is the b vector.
//   step A  T_b[k][c] = sum_{(l,v) in row c of S_b} M_b[k][l] * v        one thread per (b,k,c)
//   step B  H~[r][c]  = sum_b sum_{(k,v) in row r of S_b} v * T_b[k][c]    one block per row r, fixed order: deterministic
__global__ __launch_bounds__(256) void ba_stitch_a_kernel(const int* __restrict__ rowptr, const int* __restrict__ col, const double* __restrict__ val,
                                                          const double* __restrict__ M, int n1, int m, double* __restrict__ Tm) {
    const int b = blockIdx.y, e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= m * n1) return;
    const int k = e / n1, c = e - k * n1;
    const double* Mk = M + ((size_t)b * m + k) * m;
    double s = 0;
    for (int q = rowptr[b * n1 + c]; q < rowptr[b * n1 + c + 1]; ++q) s += Mk[col[q]] * val[q];
    Tm[(size_t)b * m * n1 + e] = s;
}
__global__ __launch_bounds__(128) void ba_stitch_b_kernel(const int* __restrict__ rowptr, const int* __restrict__ col, const double* __restrict__ val,
                                                          const double* __restrict__ Tm, int nb, int n1, int m, double* __restrict__ H) {
    const int r = blockIdx.x;
    for (int c = threadIdx.x; c < n1; c += blockDim.x) {
        double s = 0;
        for (int b = 0; b < nb; ++b) {
            const double* Tb = Tm + (size_t)b * m * n1 + c;
            for (int q = rowptr[b * n1 + r]; q < rowptr[b * n1 + r + 1]; ++q) s += val[q] * Tb[(size_t)col[q] * n1];
        }
        H[(size_t)r * n1 + c] = s;
    }
}
void ba_launch_stitch(hipStream_t s, const int* rowptr, const int* col, const double* val, const double* M, int nb, int n1, int m, double* Tm, double* H) {
    ba_stitch_a_kernel<<<dim3((m * n1 + 255) / 256, nb), 256, 0, s>>>(rowptr, col, val, M, n1, m, Tm);
    ba_stitch_b_kernel<<<n1, 128, 0, s>>>(rowptr, col, val, Tm, nb, n1, m, H);
}

// ------------------------------------------------------------------------------------------------ a12 + step
// xAd: [W*W][8] index h*W + t (EnergyFunctional.cpp:270-280), xc: cstep(4)
__global__ __launch_bounds__(256) void ba_resub_kernel(BADev B, const float* __restrict__ xAd, const float* __restrict__ xc) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= B.Ppad) return;
    if (!(B.pt_flags[d] & PT_VALID)) return;
    const int h = B.blk_host[d / kBlk], W = B.W;
    if (B.pt_ngood[d] == 0) { B.pt_step[d] = 0.f; return; }
    const float4 pa = B.pt_acc[d], hc = B.pt_hcd[d];
    float bsum = pa.w;
    bsum -= xc[0] * hc.x + xc[1] * hc.y + xc[2] * hc.z + xc[3] * hc.w;
    for (int t = 0; t < W; ++t) {
        if (t == h) continue;
        const size_t si = (size_t)t * B.Ppad + d;
        if (!(B.rs_state[si] & RS_ACTIVE)) continue;
        const float4 j0 = B.rs_jp0[si], j1 = B.rs_jp1[si];
        const float* xa = xAd + (size_t)(h * W + t) * 8;
        bsum -= xa[0] * j0.x + xa[1] * j0.y + xa[2] * j0.z + xa[3] * j0.w + xa[4] * j1.x + xa[5] * j1.y + xa[6] * j1.z + xa[7] * j1.w;
    }
    B.pt_step[d] = -bsum * pa.z;
}
// idepth = idepth_backup + stepfacD*step; idepth_zero = idepth (FullSystemOptimize.cpp:271-276). sums: {step^2, |idepth_backup|, count}
__global__ __launch_bounds__(256) void ba_step_kernel(BADev B, float stepfacD, float* __restrict__ partial /* [blocks][4] */) {
    __shared__ float smem[64 * 4];
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    float v[3] = {0.f, 0.f, 0.f};
    if (d < B.Ppad && (B.pt_flags[d] & PT_VALID)) {
        float4 geo = B.pt_geo[d];
        const float idb = geo.z, stp = B.pt_step[d];
        B.pt_backup[d] = idb;
        geo.z = idb + stepfacD * stp; geo.w = geo.z;
        B.pt_geo[d] = geo;
        v[0] = stp * stp; v[1] = fabsf(idb); v[2] = 1.f;
    }
    block_reduce_cols<3, 256>(v, smem, partial + (size_t)blockIdx.x * 4);
}
// out[j] = sum_b partial[b*stride + j] in fp64: 16 groups of 64 lanes stride over the blocks, then a fixed-order LDS combine
__global__ __launch_bounds__(1024) void ba_sum_partials_kernel(const float* __restrict__ partial, int nblocks, int stride, int nvals, double* __restrict__ out) {
    __shared__ double part[16][64];
    const int j = threadIdx.x & 63, g = threadIdx.x >> 6;
    double s = 0;
    if (j < nvals) for (int b = g; b < nblocks; b += 16) s += (double)partial[(size_t)b * stride + j];
    part[g][j] = s;
    __syncthreads();
    if (g == 0 && j < nvals) { double t = 0; for (int k = 0; k < 16; ++k) t += part[k][j]; out[j] = t; }
}
void ba_launch_resub(hipStream_t s, const BADev& B, const float* xAd, const float* xc) {
    ba_resub_kernel<<<(B.Ppad + 255) / 256, 256, 0, s>>>(B, xAd, xc);
}
void ba_launch_step(hipStream_t s, const BADev& B, float stepfacD, float* partial, double* out3) {
    const int nb = (B.Ppad + 255) / 256;
    ba_step_kernel<<<nb, 256, 0, s>>>(B, stepfacD, partial);
    ba_sum_partials_kernel<<<1, 1024, 0, s>>>(partial, nb, 4, 3, out3);
}

// ------------------------------------------------------------------------------------------------ setNewFrameEnergyTH
// Exact n-th element (nthIdx = (int)(0.7f * n), FullSystemOptimize.cpp:117-122) of the non-negative energies of the residuals
// that target the newest frame, by a two-level 16+16-bit radix select on the float bit patterns (monotone for x >= 0):
// the high-half histogram is filled by ba_linearize_kernel itself (integer atomics), then find-hi -> low-half histogram of the
// matching entries -> find-lo + the threshold formula (:130-133). Runs on the side stream, overlapped with SC/reduce/stitch.
__global__ __launch_bounds__(1024) void ba_th_find_kernel(unsigned* __restrict__ hist, unsigned* __restrict__ state, int level, float* __restrict__ frameTH_new) {
    __shared__ unsigned part[1024];
    __shared__ unsigned s_sel, s_run;
    const int tid = threadIdx.x;
    unsigned loc[64], sum = 0;
#pragma unroll
    for (int i = 0; i < 64; ++i) { loc[i] = hist[tid * 64 + i]; sum += loc[i]; }
    part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
        unsigned total = 0;
        for (int i = 0; i < 1024; ++i) total += part[i];
        unsigned k;
        if (level == 0) { state[0] = total; k = (unsigned)(int)(0.7f * (float)total); } else k = state[1];
        unsigned run = 0; int sel = 1023;
        for (int i = 0; i < 1024; ++i) { if (run + part[i] > k) { sel = i; break; } run += part[i]; }
        s_sel = (unsigned)sel; s_run = run;
        if (level == 0 && total == 0) state[3] = 1; else if (level == 0) state[3] = 0;
    }
    __syncthreads();
    if (tid == (int)s_sel) {
        const unsigned k = (level == 0) ? (unsigned)(int)(0.7f * (float)state[0]) : state[1];
        unsigned run = s_run; int bsel = 63;
        for (int i = 0; i < 64; ++i) { if (run + loc[i] > k) { bsel = i; break; } run += loc[i]; }
        const unsigned bin = (unsigned)(tid * 64 + bsel);
        if (level == 0) { state[1] = k - run; state[2] = bin; }
        else {
            float th;
            if (state[3]) th = 12.f * 12.f * (float)kPatternNum;                        // no residual on the newest frame (:110-114)
            else {
                const float nthElement = sqrtf(__uint_as_float((state[2] << 16) | bin));
                th = nthElement * 1.5f;                                                 // setting_frameEnergyTHFacMedian
                th = 26.0f * 0.5f + th * (1.f - 0.5f);                                  // setting_frameEnergyTHConstWeight
                th = th * th;                                                           // setting_overallEnergyTHWeight = 1
            }
            *frameTH_new = th;
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 64; ++i) hist[tid * 64 + i] = 0;                                // ready for the next pass
}
__global__ __launch_bounds__(256) void ba_th_lo_kernel(const float* __restrict__ en, int n, const unsigned* __restrict__ state, unsigned* __restrict__ hist_lo) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float f = en[i];
    if (!(f >= 0.f)) return;
    const unsigned u = __float_as_uint(f);
    if ((u >> 16) == state[2]) atomicAdd(&hist_lo[u & 0xFFFFu], 1u);
}
void ba_launch_energy_th(hipStream_t s, const BADev& B) {
    ba_th_find_kernel<<<1, 1024, 0, s>>>(B.th_hist_hi, B.th_state, 0, nullptr);
    ba_th_lo_kernel<<<(B.Ppad + 255) / 256, 256, 0, s>>>(B.en_new, B.Ppad, B.th_state, B.th_hist_lo);
    ba_th_find_kernel<<<1, 1024, 0, s>>>(B.th_hist_lo, B.th_state, 1, B.frameTH + (B.W - 1));
}

}  // namespace nalo
