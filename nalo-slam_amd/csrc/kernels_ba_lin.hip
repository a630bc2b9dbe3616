// ba_linearize_kernel / ba_linearize_coop_kernel — a5+a6+a7 (+a13 in MODE 2) for gfx950. The two kernels share every formula; the second (default)
// fetches the 32 texels of a residual with its quad so that neighbouring lanes read neighbouring texels (see its phase B).
//
//   PointFrameResidual::linearize + applyRes + EFResidual::takeDataF   (reference src/FullSystem/Residuals.cpp:78-274,306-328;
//                                                                       src/OptimizationBackend/EnergyFunctionalStructs.cpp:39-50)
//   AccumulatedTopHessianSSE::addPoint<0|2> + AccumulatorApprox        (src/OptimizationBackend/AccumulatedTopHessian.cpp:39-162,
//                                                                       MatrixAccumulators.h:754-915)
//   EFResidual::fixLinearizationF (MODE 2)                             (EnergyFunctionalStructs.cpp:89-115)
//
// Grid = (point block, target), TARGET-MAJOR and XCD-aware: all CUs gather from the same target image at a time, so its 16-byte texels
// stay resident in the XCD L2s / Infinity Cache. One block = 256 points of ONE host and ONE target: the FrameFramePrecalc
// record is block-uniform (scalar loads), and the 91-entry AccumulatorApprox block of the (host,target) bin is reduced once
// per block. Register diet for 2 waves/SIMD: the 93 reduced values are streamed (4 at a time: 2 DPP quad adds + one 16-byte LDS
// store by one lane per quad) instead of being held, and the 8x4 texel gather is issued in two halves of 16 loads.
// Per-point sums over the targets (Hdd, bd, Hcd) are written per slot and summed in a fixed order by ba_sc_kernel.
#include "nalo_internal.h"
#include <hip/hip_ext.h>
#include <cstdlib>
#include <type_traits>
#include "ba_device.h"
#include "reduce.h"

#ifndef NALO_LIN_WAVES
#define NALO_LIN_WAVES 3      // waves per SIMD the register allocator must leave room for (142 VGPRs, no scratch)
#endif

namespace nalo {

__device__ __forceinline__ float4 lin_bilinear(const float4* __restrict__ img, float x, float y, int width) {
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

// streams values 0..N-1 (in order) into the block's LDS rows: every 4 values -> quad DPP adds -> one float4 store per quad
struct QuadStream {
    float4* row4;
    bool writer;
    float b0, b1, b2;
    int k;
    __device__ __forceinline__ QuadStream(float* smem) : row4(reinterpret_cast<float4*>(smem + (threadIdx.x >> 2) * kTopStride)), writer((threadIdx.x & 3) == 0), b0(0.f), b1(0.f), b2(0.f), k(0) {}
    __device__ __forceinline__ void put(float v) {
        v += dpp_quad_xor1(v); v += dpp_quad_xor2(v);
        const int m = k & 3;
        if (m == 0) b0 = v; else if (m == 1) b1 = v; else if (m == 2) b2 = v;
        else if (writer) row4[k >> 2] = make_float4(b0, b1, b2, v);
        ++k;
    }
    __device__ __forceinline__ void flush() {            // pad the last group with zeros
        while (k & 3) put(0.f);
    }
};

// MODE 0: active residuals (optimize). MODE 2: marginalisation of the flagged points (resApprox = res_toZeroF).
// FIX: linearizeAll(true) — residuals that do not end IN are dropped; centerProjectedTo / relBS are stored.
template <int MODE, int FIX>
__global__ __launch_bounds__(kBlk, NALO_LIN_WAVES) void ba_linearize_kernel(BADev B) {
    __shared__ __attribute__((aligned(16))) float smem[(kBlk / 4) * kTopStride];
    if (B.stop && B.stop[0]) return;                                                // the queued GN loop has terminated (kernels_ba_gn.hip)
    const int W = B.W, tid = threadIdx.x;
    // CalibHessian::value_scaledf / value_scaledi live in device memory (block-uniform scalar loads): the GN step may be taken on the device
    const float cal_fxl = B.calib[0], cal_fyl = B.calib[1], cal_cxl = B.calib[2], cal_cyl = B.calib[3], cal_fxli = B.calib[4], cal_fyli = B.calib[5];
    // XCD-aware mapping: workgroups are dealt round-robin over the 8 XCDs, so launch index j of a target runs on the XCD group
    // j % 8. Group x walks blk_order[x][*] = the x-th spatial eighth (Morton range) of every host's points, so each XCD's
    // private 4 MiB L2 only ever sees ~1/8 of the target image instead of all of it (speed only, never correctness).
    const int per_t = 8 * B.xcd_len, t = blockIdx.x / per_t, j = blockIdx.x - t * per_t;
    const int b = B.blk_order[(j & 7) * B.xcd_len + (j >> 3)];
    if (b < 0) return;
    const int d = b * kBlk + tid, h = B.blk_host[b];
    if (t == h) {                                                       // no self residuals; the newest frame's own points have no entry
        if (MODE == 0 && t == W - 1) B.en_new[d] = -1.f;
        return;
    }
    const size_t si = (size_t)t * B.Ppad + d;
    const float* pc = B.pre + (size_t)(h * W + t) * kPreStride;         // block-uniform: scalar loads
    const uint8_t pf = B.pt_flags[d];
    const bool pvalid = (pf & PT_VALID) && (MODE == 0 || (pf & PT_MARG));
    uint8_t st = B.rs_state[si];
    const bool exists = pvalid && (st & RS_EXISTS) && (MODE == 2 || !(st & RS_LINEARIZED));

    // everything the accumulation needs; stays zero unless this lane ends with an active (IN) residual
    float x[10], y[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) { x[i] = 0.f; y[i] = 0.f; }
    float a = 0.f, bb = 0.f, c = 0.f, jab00 = 0.f, jab01 = 0.f, jab10 = 0.f, jab11 = 0.f, ab00 = 0.f, ab01 = 0.f, ab11 = 0.f;
    float JIr0 = 0.f, JIr1 = 0.f, Jabr0 = 0.f, Jabr1 = 0.f, rr = 0.f, cnt = 0.f, energy = 0.f, enew = -1.f;

    if (exists) {
        const float4 geo = B.pt_geo[d];
        const float pu = geo.x, pv = geo.y, idepth = geo.z, idz = geo.w;
        float2 en = B.rs_energy[si];
        int state = st & RS_STATE_MASK;
        if (MODE == 2) { en.x = 0.f; en.y = 0.f; state = 0; st &= ~RS_LINEARIZED; }            // resetOOB + isLinearized=false (FullSystem.cpp:978-981)
        int newState = 2;
        energy = en.x;
        bool full = false;
        float Jpdd0 = 0.f, Jpdd1 = 0.f, cKu = 0.f, cKv = 0.f, cId = 0.f;
        if (state == 1) { newState = 1; }                                                       // Residuals.cpp:82-83
        else {
            const float wM3G = (float)(B.w - 3), hM3G = (float)(B.h - 3);
            const float KliP0 = (pu - cal_cxl) * cal_fxli, KliP1 = (pv - cal_cyl) * cal_fyli;           // ResidualProjections.h:70-73
            // ---- centre projection at idepth_zero (projectPoint, ResidualProjections.h:61-87)
            const float p0 = pc[12] * KliP0 + pc[13] * KliP1 + pc[14] + pc[21] * idz;
            const float p1 = pc[15] * KliP0 + pc[16] * KliP1 + pc[17] + pc[22] * idz;
            const float p2 = pc[18] * KliP0 + pc[19] * KliP1 + pc[20] + pc[23] * idz;
            const float drescale = 1.0f / p2, new_idepth = idz * drescale;
            const float u = p0 * drescale, vv = p1 * drescale;
            const float Ku0 = u * cal_fxl + cal_cxl, Kv0 = vv * cal_fyl + cal_cyl;
            bool ok = (drescale > 0.f) && Ku0 > 1.1f && Kv0 > 1.1f && Ku0 < wM3G && Kv0 < hM3G;
            // ---- the 8 pattern pixels at the current idepth (projectPoint, ResidualProjections.h:47-57)
            float Kus[8], Kvs[8];
            constexpr int pdx[8] = {0, -1, 1, -2, 0, 2, -1, 0}, pdy[8] = {-2, -1, -1, 0, 0, 0, 1, 2};   // util/settings.cpp:297
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float xx = pu + (float)pdx[k], yy = pv + (float)pdy[k];
                const float q0 = pc[0] * xx + pc[1] * yy + pc[2] + pc[9] * idepth;
                const float q1 = pc[3] * xx + pc[4] * yy + pc[5] + pc[10] * idepth;
                const float q2 = pc[6] * xx + pc[7] * yy + pc[8] + pc[11] * idepth;
                Kus[k] = q0 / q2; Kvs[k] = q1 / q2;
                ok = ok && Kus[k] > 1.1f && Kvs[k] > 1.1f && Kus[k] < wM3G && Kvs[k] < hM3G;
            }
            if (!ok) { newState = 1; }
            else {
                cKu = Ku0; cKv = Kv0; cId = new_idepth;
                const float t0x = pc[21], t0y = pc[22], t0z = pc[23];
                Jpdd0 = drescale * (t0x - t0z * u) * kScaleIdepth * cal_fxl;                       // Residuals.cpp:116-117
                Jpdd1 = drescale * (t0y - t0z * vv) * kScaleIdepth * cal_fyl;
                // x = (Jpdc[0], Jpdxi[0]), y = (Jpdc[1], Jpdxi[1])                              :123-156
                x[2] = drescale * (pc[18] * u - pc[12]);
                x[3] = cal_fxl * drescale * (pc[19] * u - pc[13]) * cal_fyli;
                x[0] = KliP0 * x[2]; x[1] = KliP1 * x[3];
                y[2] = cal_fyl * drescale * (pc[18] * vv - pc[15]) * cal_fxli;
                y[3] = drescale * (pc[19] * vv - pc[16]);
                y[0] = KliP0 * y[2]; y[1] = KliP1 * y[3];
                x[0] = (x[0] + u) * kScaleF; x[1] *= kScaleF; x[2] = (x[2] + 1) * kScaleC; x[3] *= kScaleC;
                y[0] *= kScaleF; y[1] = (y[1] + vv) * kScaleF; y[2] *= kScaleC; y[3] = (y[3] + 1) * kScaleC;
                x[4] = new_idepth * cal_fxl; x[5] = 0.f; x[6] = -new_idepth * u * cal_fxl;
                x[7] = -u * vv * cal_fxl; x[8] = (1 + u * u) * cal_fxl; x[9] = -vv * cal_fxl;
                y[4] = 0.f; y[5] = new_idepth * cal_fyl; y[6] = -new_idepth * vv * cal_fyl;
                y[7] = -(1 + vv * vv) * cal_fyl; y[8] = u * vv * cal_fyl; y[9] = u * cal_fyl;
                float jx = 0.f, jy = 0.f;
                if (MODE == 2) {                                                                 // Jp*delta (EnergyFunctionalStructs.cpp:94-99)
                    const float dd = idepth - idz;
#pragma unroll
                    for (int i = 0; i < 6; ++i) { jx += x[4 + i] * pc[27 + i]; jy += y[4 + i] * pc[27 + i]; }
                    float cxs = 0.f, cys = 0.f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) { cxs += x[i] * B.calib[6 + i]; cys += y[i] * B.calib[6 + i]; }
                    jx = jx + cxs + Jpdd0 * dd; jy = jy + cys + Jpdd1 * dd;
                }
                const float affLL0 = pc[24], affLL1 = pc[25], b0 = pc[26];
                const float4* img = B.img[t];
                float energyLeft = 0.f, wJI2_sum = 0.f;
                bool finite_ok = true;
                float color[8], wgt[8];
                {
                    const float4 c0 = B.pt_col0[d], c1 = B.pt_col1[d], w0 = B.pt_w0[d], w1 = B.pt_w1[d];
                    color[0] = c0.x; color[1] = c0.y; color[2] = c0.z; color[3] = c0.w; color[4] = c1.x; color[5] = c1.y; color[6] = c1.z; color[7] = c1.w;
                    wgt[0] = w0.x; wgt[1] = w0.y; wgt[2] = w0.z; wgt[3] = w0.w; wgt[4] = w1.x; wgt[5] = w1.y; wgt[6] = w1.z; wgt[7] = w1.w;
                }
#pragma unroll
                for (int half = 0; half < 2; ++half) {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) {                                             // Residuals.cpp:183-245
                        const int k = half * 4 + kk;
                        const float4 hit = lin_bilinear(img, Kus[k], Kvs[k], B.w);
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
                        wJI2_sum += hw * hw * (jI0 * jI0 + jI1 * jI1);                           // on the hw-scaled gradient, as :215-239
                        float ra = resF;                                                         // mode 2: res_toZeroF (:103-111)
                        if (MODE == 2) ra = resF - jI0 * jx - jI1 * jy - jA * pc[33] - jB * pc[34];
                        JIr0 += ra * jI0; JIr1 += ra * jI1; Jabr0 += ra * jA; Jabr1 += ra * jB; rr += ra * ra;
                    }
                    asm volatile("" ::: "memory");                    // keep the second half's 16 gathers behind the first half (register diet)
                }
                if (!finite_ok) { newState = 1; }
                else {
                    full = true;
                    enew = energyLeft;
                    const float th = fmaxf(B.frameTH[h], B.frameTH[t]);
                    if (energyLeft > th || wJI2_sum < 2.f) { energyLeft = th; newState = 2; } else newState = 0;   // :262-270
                    en.y = energyLeft;
                    energy = energyLeft;
                }
            }
        }
        // ---- applyRes(true) (Residuals.cpp:306-328)
        bool active = false;
        if (state != 1) { active = (newState == 0); state = newState; en.x = en.y; }
        st = (uint8_t)((st & ~(RS_STATE_MASK | RS_ACTIVE)) | state | (active ? RS_ACTIVE : 0));
        if (MODE == 2 && active) st |= RS_LINEARIZED;
        if (FIX && !active) st &= ~(RS_EXISTS | RS_ACTIVE);                                      // toRemove (FullSystemOptimize.cpp:81-84,184-205)
        B.rs_state[si] = st;
        B.rs_energy[si] = en;
        if (active && full) {
            cnt = 1.f;
            // ---- takeDataF (EnergyFunctionalStructs.cpp:39-50)
            const float a0 = a * Jpdd0 + bb * Jpdd1, a1 = bb * Jpdd0 + c * Jpdd1;
            float4 j0, j1;
            j0.x = x[4] * a0 + y[4] * a1; j0.y = x[5] * a0 + y[5] * a1; j0.z = x[6] * a0 + y[6] * a1; j0.w = x[7] * a0 + y[7] * a1;
            j1.x = x[8] * a0 + y[8] * a1; j1.y = x[9] * a0 + y[9] * a1;
            j1.z = jab00 * Jpdd0 + jab01 * Jpdd1; j1.w = jab10 * Jpdd0 + jab11 * Jpdd1;
            B.rs_jp0[si] = j0; B.rs_jp1[si] = j1;
            // per-slot share of EFPoint::{bd,Hdd,Hcd}_acc (AccumulatedTopHessian.cpp:132-135); summed over the targets by ba_sc_kernel
            B.rs_pp0[si] = make_float4(JIr0 * Jpdd0 + JIr1 * Jpdd1, a0 * Jpdd0 + a1 * Jpdd1, x[0] * a0 + y[0] * a1, x[1] * a0 + y[1] * a1);
            B.rs_pp1[si] = make_float2(x[2] * a0 + y[2] * a1, x[3] * a0 + y[3] * a1);
            if (FIX || MODE == 2) {
                // relBS of FullSystemOptimize.cpp:69-71 + centerProjectedTo (makeCoarseDepthL0 input)
                const float i0 = pc[0] * pu + pc[1] * pv + pc[2], i1 = pc[3] * pu + pc[4] * pv + pc[5], i2 = pc[6] * pu + pc[7] * pv + pc[8];
                const float q0 = i0 + pc[9] * idepth, q1 = i1 + pc[10] * idepth, q2 = i2 + pc[11] * idepth;
                const float ex = i0 / i2 - q0 / q2, ey = i1 / i2 - q1 / q2;
                const float relBS = 0.01f * sqrtf(ex * ex + ey * ey);
                atomicMax(reinterpret_cast<unsigned*>(&B.pt_relbs[d]), __float_as_uint(relBS));   // non-negative floats order like their bit patterns
                B.rs_cpt[si] = make_float4(cKu, cKv, cId, relBS);
            }
        } else {
            // not IN: contributes nothing to the Hessian block
            a = bb = c = jab00 = jab01 = jab10 = jab11 = ab00 = ab01 = ab11 = JIr0 = JIr1 = Jabr0 = Jabr1 = rr = 0.f;
        }
    }
    if (MODE == 0 && t == W - 1) {
        B.en_new[d] = enew;
        if (enew >= 0.f) atomicAdd(&B.th_hist_hi[__float_as_uint(enew) >> 16], 1u);               // integer atomics: order independent
    }

    // ---- AccumulatorApprox::update / updateTopRight / updateBotRight (AccumulatedTopHessian.cpp:115-129), streamed into the reduction
    QuadStream qs(smem);
    {
        float ax[10], cy[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) { ax[i] = a * x[i] + bb * y[i]; cy[i] = bb * x[i] + c * y[i]; }
#pragma unroll
        for (int r = 0; r < 10; ++r)
#pragma unroll
            for (int cc = r; cc < 10; ++cc) qs.put(ax[r] * x[cc] + cy[r] * y[cc]);                  // 55: upper triangle of the 10x10
    }
#pragma unroll
    for (int i = 0; i < 10; ++i) {                                                                 // 30: TopRight 10x3
        qs.put(x[i] * jab00 + y[i] * jab01);
        qs.put(x[i] * jab10 + y[i] * jab11);
        qs.put(x[i] * JIr0 + y[i] * JIr1);
    }
    qs.put(ab00); qs.put(ab01); qs.put(Jabr0); qs.put(ab11); qs.put(Jabr1); qs.put(rr);           // 6: BotRight
    qs.put(cnt); qs.put(energy);                                                                   // [91] residual count, [92] energy
    qs.flush();
    __syncthreads();
    if (tid < kTopVals) {
        double s = 0.0;
#pragma unroll 8
        for (int r = 0; r < kBlk / 4; ++r) s += (double)smem[r * kTopStride + tid];                // all lanes read one row: conflict free
        B.top_partial[((size_t)b * W + t) * kTopStride + tid] = s;
    }
}

#ifndef NALO_LIN_COOP_NPB
#define NALO_LIN_COOP_NPB 3      // pattern pixels per batch: 3+3+2 (12 loads in flight per lane, 158 VGPRs) measured best: 198 us on stress250k; 2: 202, 4: 226 (spills)
#endif
#ifndef NALO_LIN_COOP_WAVES
#define NALO_LIN_COOP_WAVES 3      // 4 fits (126 VGPRs, 24.5 KB of LDS) but measures slower: 215 vs 203 us on stress250k
#endif
// ---------------------------------------------------------------------------------------------------------------- cooperative-gather variant
// Same kernel, same arithmetic; only the way the 32 texels of a residual reach its lane differs (see phase B).
template <int R> __device__ __forceinline__ int lin_quad_bcast(int v) {        // value of lane R of this lane's quad (DPP quad_perm [R,R,R,R])
    return __builtin_amdgcn_update_dpp(0, v, R * 0x55, 0xF, 0xF, true);
}
template <int MODE, int FIX>
__global__ __launch_bounds__(kBlk, NALO_LIN_COOP_WAVES) void ba_linearize_coop_kernel(BADev B) {
    __shared__ __attribute__((aligned(16))) float smem[(kBlk / 4) * kTopStride];
    if (B.stop && B.stop[0]) return;                                                // the queued GN loop has terminated (kernels_ba_gn.hip)
    const int W = B.W, tid = threadIdx.x;
    // CalibHessian::value_scaledf / value_scaledi live in device memory (block-uniform scalar loads): the GN step may be taken on the device
    const float cal_fxl = B.calib[0], cal_fyl = B.calib[1], cal_cxl = B.calib[2], cal_cyl = B.calib[3], cal_fxli = B.calib[4], cal_fyli = B.calib[5];
    // XCD-aware mapping: workgroups are dealt round-robin over the 8 XCDs, so launch index j of a target runs on the XCD group
    // j % 8. Group x walks blk_order[x][*] = the x-th spatial eighth (Morton range) of every host's points, so each XCD's
    // private 4 MiB L2 only ever sees ~1/8 of the target image instead of all of it (speed only, never correctness).
    const int per_t = 8 * B.xcd_len, t = blockIdx.x / per_t, j = blockIdx.x - t * per_t;
    const int b = B.blk_order[(j & 7) * B.xcd_len + (j >> 3)];
    if (b < 0) return;
    const int d = b * kBlk + tid, h = B.blk_host[b];
    if (t == h) {                                                       // no self residuals; the newest frame's own points have no entry
        if (MODE == 0 && t == W - 1) B.en_new[d] = -1.f;
        return;
    }
    const size_t si = (size_t)t * B.Ppad + d;
    const float* pc = B.pre + (size_t)(h * W + t) * kPreStride;         // block-uniform: scalar loads
    const uint8_t pf = B.pt_flags[d];
    const bool pvalid = (pf & PT_VALID) && (MODE == 0 || (pf & PT_MARG));
    uint8_t st = B.rs_state[si];
    const bool exists = pvalid && (st & RS_EXISTS) && (MODE == 2 || !(st & RS_LINEARIZED));

    // everything the accumulation needs; stays zero unless this lane ends with an active (IN) residual
    float x[10], y[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) { x[i] = 0.f; y[i] = 0.f; }
    float a = 0.f, bb = 0.f, c = 0.f, jab00 = 0.f, jab01 = 0.f, jab10 = 0.f, jab11 = 0.f, ab00 = 0.f, ab01 = 0.f, ab11 = 0.f;
    float JIr0 = 0.f, JIr1 = 0.f, Jabr0 = 0.f, Jabr1 = 0.f, rr = 0.f, cnt = 0.f, energy = 0.f, enew = -1.f;

    // ---- phase A (per lane): geometry of the residual; `need` = it takes its 32 taps
    float pu = 0.f, pv = 0.f, idepth = 0.f, idz = 0.f;
    float2 en = make_float2(0.f, 0.f);
    int state = 0, newState = 2;
    bool full = false, need = false;
    float Jpdd0 = 0.f, Jpdd1 = 0.f, cKu = 0.f, cKv = 0.f, cId = 0.f, jx = 0.f, jy = 0.f;
    float Kus[8], Kvs[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { Kus[k] = 2.f; Kvs[k] = 2.f; }
    if (exists) {
        const float4 geo = B.pt_geo[d];
        pu = geo.x; pv = geo.y; idepth = geo.z; idz = geo.w;
        en = B.rs_energy[si];
        state = st & RS_STATE_MASK;
        if (MODE == 2) { en.x = 0.f; en.y = 0.f; state = 0; st &= ~RS_LINEARIZED; }            // resetOOB + isLinearized=false (FullSystem.cpp:978-981)
        energy = en.x;
        if (state == 1) { newState = 1; }                                                       // Residuals.cpp:82-83
        else {
            const float wM3G = (float)(B.w - 3), hM3G = (float)(B.h - 3);
            const float KliP0 = (pu - cal_cxl) * cal_fxli, KliP1 = (pv - cal_cyl) * cal_fyli;           // ResidualProjections.h:70-73
            // ---- centre projection at idepth_zero (projectPoint, ResidualProjections.h:61-87)
            const float p0 = pc[12] * KliP0 + pc[13] * KliP1 + pc[14] + pc[21] * idz;
            const float p1 = pc[15] * KliP0 + pc[16] * KliP1 + pc[17] + pc[22] * idz;
            const float p2 = pc[18] * KliP0 + pc[19] * KliP1 + pc[20] + pc[23] * idz;
            const float drescale = 1.0f / p2, new_idepth = idz * drescale;
            const float u = p0 * drescale, vv = p1 * drescale;
            const float Ku0 = u * cal_fxl + cal_cxl, Kv0 = vv * cal_fyl + cal_cyl;
            bool ok = (drescale > 0.f) && Ku0 > 1.1f && Kv0 > 1.1f && Ku0 < wM3G && Kv0 < hM3G;
            // ---- the 8 pattern pixels at the current idepth (projectPoint, ResidualProjections.h:47-57)
            constexpr int pdx[8] = {0, -1, 1, -2, 0, 2, -1, 0}, pdy[8] = {-2, -1, -1, 0, 0, 0, 1, 2};   // util/settings.cpp:297
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float xx = pu + (float)pdx[k], yy = pv + (float)pdy[k];
                const float q0 = pc[0] * xx + pc[1] * yy + pc[2] + pc[9] * idepth;
                const float q1 = pc[3] * xx + pc[4] * yy + pc[5] + pc[10] * idepth;
                const float q2 = pc[6] * xx + pc[7] * yy + pc[8] + pc[11] * idepth;
                Kus[k] = q0 / q2; Kvs[k] = q1 / q2;
                ok = ok && Kus[k] > 1.1f && Kvs[k] > 1.1f && Kus[k] < wM3G && Kvs[k] < hM3G;
            }
            if (!ok) { newState = 1; }
            else {
                need = true;
                cKu = Ku0; cKv = Kv0; cId = new_idepth;
                const float t0x = pc[21], t0y = pc[22], t0z = pc[23];
                Jpdd0 = drescale * (t0x - t0z * u) * kScaleIdepth * cal_fxl;                       // Residuals.cpp:116-117
                Jpdd1 = drescale * (t0y - t0z * vv) * kScaleIdepth * cal_fyl;
                // x = (Jpdc[0], Jpdxi[0]), y = (Jpdc[1], Jpdxi[1])                              :123-156
                x[2] = drescale * (pc[18] * u - pc[12]);
                x[3] = cal_fxl * drescale * (pc[19] * u - pc[13]) * cal_fyli;
                x[0] = KliP0 * x[2]; x[1] = KliP1 * x[3];
                y[2] = cal_fyl * drescale * (pc[18] * vv - pc[15]) * cal_fxli;
                y[3] = drescale * (pc[19] * vv - pc[16]);
                y[0] = KliP0 * y[2]; y[1] = KliP1 * y[3];
                x[0] = (x[0] + u) * kScaleF; x[1] *= kScaleF; x[2] = (x[2] + 1) * kScaleC; x[3] *= kScaleC;
                y[0] *= kScaleF; y[1] = (y[1] + vv) * kScaleF; y[2] *= kScaleC; y[3] = (y[3] + 1) * kScaleC;
                x[4] = new_idepth * cal_fxl; x[5] = 0.f; x[6] = -new_idepth * u * cal_fxl;
                x[7] = -u * vv * cal_fxl; x[8] = (1 + u * u) * cal_fxl; x[9] = -vv * cal_fxl;
                y[4] = 0.f; y[5] = new_idepth * cal_fyl; y[6] = -new_idepth * vv * cal_fyl;
                y[7] = -(1 + vv * vv) * cal_fyl; y[8] = u * vv * cal_fyl; y[9] = u * cal_fyl;
                if (MODE == 2) {                                                                 // Jp*delta (EnergyFunctionalStructs.cpp:94-99)
                    const float dd = idepth - idz;
#pragma unroll
                    for (int i = 0; i < 6; ++i) { jx += x[4 + i] * pc[27 + i]; jy += y[4 + i] * pc[27 + i]; }
                    float cxs = 0.f, cys = 0.f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) { cxs += x[i] * B.calib[6 + i]; cys += y[i] * B.calib[6 + i]; }
                    jx = jx + cxs + Jpdd0 * dd; jy = jy + cys + Jpdd1 * dd;
                }
            }
        }
    }
    // ---- phase B (uniform over the wave): the taps of the FOUR residuals of a quad are fetched together. In round r every lane of the quad loads tap q
    // (its lane index: 0,1 = the two texels of row iy, 2,3 = row iy+1) of residual r's pixel, so neighbouring lanes read neighbouring 16-byte texels
    // and the texture path sees half the cache-line accesses of one-residual-per-lane gathers (scripts/ubench/gather.hip: 247 -> 175 us for the taps of
    // 1.75 M residuals). The texels go back to their owner through a per-wave LDS buffer; the bilinear formula and everything after it is the
    // per-lane code of ba_linearize_kernel, operation for operation: results are bit-identical.
    const float affLL0 = pc[24], affLL1 = pc[25], b0 = pc[26];
    float energyLeft = 0.f, wJI2_sum = 0.f;
    bool finite_ok = true;
    float color[8], wgt[8];
    if (need) {
        const float4 c0 = B.pt_col0[d], c1 = B.pt_col1[d], w0 = B.pt_w0[d], w1 = B.pt_w1[d];
        color[0] = c0.x; color[1] = c0.y; color[2] = c0.z; color[3] = c0.w; color[4] = c1.x; color[5] = c1.y; color[6] = c1.z; color[7] = c1.w;
        wgt[0] = w0.x; wgt[1] = w0.y; wgt[2] = w0.z; wgt[3] = w0.w; wgt[4] = w1.x; wgt[5] = w1.y; wgt[6] = w1.z; wgt[7] = w1.w;
    }
    {
        const float4* __restrict__ img = B.img[t];
        const int lane = tid & 63, q = lane & 3, Q = lane >> 2;
        // exchange buffer of this wave: [residual r of the quad][quad Q][tap q], rows padded by one texel (bank spread). It aliases the 16 reduction rows
        // this wave alone writes at the end of the kernel (QuadStream: row = tid >> 2), so it costs no LDS of its own
        float4* xb = reinterpret_cast<float4*>(smem + (tid >> 6) * 16 * kTopStride);
        static_assert(16 * kTopStride * 4 >= 4 * 65 * 16, "a wave's reduction rows must hold its exchange buffer");
        const int tapoff = (q & 1) + (q >> 1) * B.w;
        // one pixel's exchange: the four texels this lane loaded (tap q of residuals 0..3 of its quad) go out, the four taps of its own residual come back
        auto exchange = [&](const float4& t0, const float4& t1, const float4& t2, const float4& t3, float xk, float yk, float& oI, float& oX, float& oY) __attribute__((always_inline)) {
            // lanes exchange through LDS inside one wave: the hardware keeps a wave's LDS operations in order, the fences keep the COMPILER from
            // moving this lane's reads above the (for this lane provably non-aliasing) writes, or the next pixel's writes above these reads
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            xb[0 * 65 + Q * 4 + q] = t0; xb[1 * 65 + Q * 4 + q] = t1; xb[2 * 65 + Q * 4 + q] = t2; xb[3 * 65 + Q * 4 + q] = t3;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const float4 p00 = xb[q * 65 + Q * 4], p10 = xb[q * 65 + Q * 4 + 1], p01 = xb[q * 65 + Q * 4 + 2], p11 = xb[q * 65 + Q * 4 + 3];
            const int ix = (int)xk, iy = (int)yk;                 // util/globalFuncs.h:75-89
            const float dx = xk - ix, dy = yk - iy, dxdy = dx * dy;
            const float w11 = dxdy, w01 = dy - dxdy, w10 = dx - dxdy, w00 = 1 - dx - dy + dxdy;
            oI = w11 * p11.x + w01 * p01.x + w10 * p10.x + w00 * p00.x;
            oX = w11 * p11.y + w01 * p01.y + w10 * p10.y + w00 * p00.y;
            oY = w11 * p11.z + w01 * p01.z + w10 * p10.z + w00 * p00.z;
        };
        // the photometric part of one pattern pixel (Residuals.cpp:183-245), per lane
        auto pixel = [&](float hitI, float hitX, float hitY, float col, float wg) __attribute__((always_inline)) {
            const float residual = hitI - (affLL0 * col + affLL1);
            const float drdA = col - b0;
            finite_ok = finite_ok && isfinite(hitI);
            float wgt_k = sqrtf(kOutlierTHSumComponent / (kOutlierTHSumComponent + (hitX * hitX + hitY * hitY)));
            wgt_k = 0.5f * (wgt_k + wg);
            const float ar = fabsf(residual);
            float hw = ar < kHuberTH ? 1.f : kHuberTH / ar;
            energyLeft += wgt_k * wgt_k * hw * residual * residual * (2.f - hw);
            if (hw < 1.f) hw = sqrtf(hw);
            hw = hw * wgt_k;
            const float jI0 = hitX * hw, jI1 = hitY * hw, resF = residual * hw, jA = drdA * hw, jB = hw;
            a += jI0 * jI0; c += jI1 * jI1; bb += jI0 * jI1;
            jab00 += jA * jI0; jab01 += jA * jI1; jab10 += jB * jI0; jab11 += jB * jI1;
            ab00 += jA * jA; ab01 += jA * jB; ab11 += jB * jB;
            wJI2_sum += hw * hw * (jI0 * jI0 + jI1 * jI1);                                   // on the hw-scaled gradient, as :215-239
            float ra = resF;                                                                 // mode 2: res_toZeroF (:103-111)
            if (MODE == 2) ra = resF - jI0 * jx - jI1 * jy - jA * pc[33] - jB * pc[34];
            JIr0 += ra * jI0; JIr1 += ra * jI1; Jabr0 += ra * jA; Jabr1 += ra * jB; rr += ra * ra;
        };
        // a batch = two pattern pixels: 8 sixteen-byte loads in flight per lane, then the two exchanges, then (per lane) their photometric part
        auto batch = [&](auto HALF) __attribute__((always_inline)) {
            constexpr int k0 = 2 * decltype(HALF)::value, k1 = k0 + 1;
            const int o0 = need ? ((int)Kus[k0] + (int)Kvs[k0] * B.w) : 0, o1 = need ? ((int)Kus[k1] + (int)Kvs[k1] * B.w) : 0;
            const float4 a0 = img[lin_quad_bcast<0>(o0) + tapoff], a1 = img[lin_quad_bcast<1>(o0) + tapoff], a2 = img[lin_quad_bcast<2>(o0) + tapoff], a3 = img[lin_quad_bcast<3>(o0) + tapoff];
            const float4 b0_ = img[lin_quad_bcast<0>(o1) + tapoff], b1_ = img[lin_quad_bcast<1>(o1) + tapoff], b2_ = img[lin_quad_bcast<2>(o1) + tapoff], b3_ = img[lin_quad_bcast<3>(o1) + tapoff];
            float h0I, h0X, h0Y, h1I, h1X, h1Y;
            exchange(a0, a1, a2, a3, Kus[k0], Kvs[k0], h0I, h0X, h0Y);
            exchange(b0_, b1_, b2_, b3_, Kus[k1], Kvs[k1], h1I, h1X, h1Y);
            if (need) { pixel(h0I, h0X, h0Y, color[k0], wgt[k0]); pixel(h1I, h1X, h1Y, color[k1], wgt[k1]); }
        };
        // the same with four pattern pixels per batch: 16 loads in flight per lane (NALO_LIN_COOP_NPB=4)
        auto batch4 = [&](auto HALF) __attribute__((always_inline)) {
            constexpr int k0 = 4 * decltype(HALF)::value, k1 = k0 + 1, k2 = k0 + 2, k3 = k0 + 3;
            const int o0 = need ? ((int)Kus[k0] + (int)Kvs[k0] * B.w) : 0, o1 = need ? ((int)Kus[k1] + (int)Kvs[k1] * B.w) : 0;
            const int o2 = need ? ((int)Kus[k2] + (int)Kvs[k2] * B.w) : 0, o3 = need ? ((int)Kus[k3] + (int)Kvs[k3] * B.w) : 0;
            const float4 a0 = img[lin_quad_bcast<0>(o0) + tapoff], a1 = img[lin_quad_bcast<1>(o0) + tapoff], a2 = img[lin_quad_bcast<2>(o0) + tapoff], a3 = img[lin_quad_bcast<3>(o0) + tapoff];
            const float4 b0_ = img[lin_quad_bcast<0>(o1) + tapoff], b1_ = img[lin_quad_bcast<1>(o1) + tapoff], b2_ = img[lin_quad_bcast<2>(o1) + tapoff], b3_ = img[lin_quad_bcast<3>(o1) + tapoff];
            const float4 c0_ = img[lin_quad_bcast<0>(o2) + tapoff], c1_ = img[lin_quad_bcast<1>(o2) + tapoff], c2_ = img[lin_quad_bcast<2>(o2) + tapoff], c3_ = img[lin_quad_bcast<3>(o2) + tapoff];
            const float4 d0_ = img[lin_quad_bcast<0>(o3) + tapoff], d1_ = img[lin_quad_bcast<1>(o3) + tapoff], d2_ = img[lin_quad_bcast<2>(o3) + tapoff], d3_ = img[lin_quad_bcast<3>(o3) + tapoff];
            float h0I, h0X, h0Y, h1I, h1X, h1Y, h2I, h2X, h2Y, h3I, h3X, h3Y;
            exchange(a0, a1, a2, a3, Kus[k0], Kvs[k0], h0I, h0X, h0Y);
            exchange(b0_, b1_, b2_, b3_, Kus[k1], Kvs[k1], h1I, h1X, h1Y);
            exchange(c0_, c1_, c2_, c3_, Kus[k2], Kvs[k2], h2I, h2X, h2Y);
            exchange(d0_, d1_, d2_, d3_, Kus[k3], Kvs[k3], h3I, h3X, h3Y);
            if (need) { pixel(h0I, h0X, h0Y, color[k0], wgt[k0]); pixel(h1I, h1X, h1Y, color[k1], wgt[k1]); pixel(h2I, h2X, h2Y, color[k2], wgt[k2]); pixel(h3I, h3X, h3Y, color[k3], wgt[k3]); }
        };
#if NALO_LIN_COOP_NPB == 3
        // 3 + 3 + 2 pattern pixels: 12 loads in flight per lane
        auto batch3 = [&](auto FIRST) __attribute__((always_inline)) {
            constexpr int k0 = decltype(FIRST)::value, k1 = k0 + 1, k2 = k0 + 2;
            const int o0 = need ? ((int)Kus[k0] + (int)Kvs[k0] * B.w) : 0, o1 = need ? ((int)Kus[k1] + (int)Kvs[k1] * B.w) : 0, o2 = need ? ((int)Kus[k2] + (int)Kvs[k2] * B.w) : 0;
            const float4 a0 = img[lin_quad_bcast<0>(o0) + tapoff], a1 = img[lin_quad_bcast<1>(o0) + tapoff], a2 = img[lin_quad_bcast<2>(o0) + tapoff], a3 = img[lin_quad_bcast<3>(o0) + tapoff];
            const float4 b0_ = img[lin_quad_bcast<0>(o1) + tapoff], b1_ = img[lin_quad_bcast<1>(o1) + tapoff], b2_ = img[lin_quad_bcast<2>(o1) + tapoff], b3_ = img[lin_quad_bcast<3>(o1) + tapoff];
            const float4 c0_ = img[lin_quad_bcast<0>(o2) + tapoff], c1_ = img[lin_quad_bcast<1>(o2) + tapoff], c2_ = img[lin_quad_bcast<2>(o2) + tapoff], c3_ = img[lin_quad_bcast<3>(o2) + tapoff];
            float h0I, h0X, h0Y, h1I, h1X, h1Y, h2I, h2X, h2Y;
            exchange(a0, a1, a2, a3, Kus[k0], Kvs[k0], h0I, h0X, h0Y);
            exchange(b0_, b1_, b2_, b3_, Kus[k1], Kvs[k1], h1I, h1X, h1Y);
            exchange(c0_, c1_, c2_, c3_, Kus[k2], Kvs[k2], h2I, h2X, h2Y);
            if (need) { pixel(h0I, h0X, h0Y, color[k0], wgt[k0]); pixel(h1I, h1X, h1Y, color[k1], wgt[k1]); pixel(h2I, h2X, h2Y, color[k2], wgt[k2]); }
        };
        batch3(std::integral_constant<int, 0>{}); batch3(std::integral_constant<int, 3>{}); batch(std::integral_constant<int, 3>{});
#elif NALO_LIN_COOP_NPB == 4
        batch4(std::integral_constant<int, 0>{}); batch4(std::integral_constant<int, 1>{});
#else
        batch(std::integral_constant<int, 0>{}); batch(std::integral_constant<int, 1>{}); batch(std::integral_constant<int, 2>{}); batch(std::integral_constant<int, 3>{});
#endif
    }
    if (need) {
        if (!finite_ok) { newState = 1; }
        else {
            full = true;
            enew = energyLeft;
            const float th = fmaxf(B.frameTH[h], B.frameTH[t]);
            if (energyLeft > th || wJI2_sum < 2.f) { energyLeft = th; newState = 2; } else newState = 0;   // :262-270
            en.y = energyLeft;
            energy = energyLeft;
        }
    }
    if (exists) {
        // ---- applyRes(true) (Residuals.cpp:306-328)
        bool active = false;
        if (state != 1) { active = (newState == 0); state = newState; en.x = en.y; }
        st = (uint8_t)((st & ~(RS_STATE_MASK | RS_ACTIVE)) | state | (active ? RS_ACTIVE : 0));
        if (MODE == 2 && active) st |= RS_LINEARIZED;
        if (FIX && !active) st &= ~(RS_EXISTS | RS_ACTIVE);                                      // toRemove (FullSystemOptimize.cpp:81-84,184-205)
        B.rs_state[si] = st;
        B.rs_energy[si] = en;
        if (active && full) {
            cnt = 1.f;
            // ---- takeDataF (EnergyFunctionalStructs.cpp:39-50)
            const float a0 = a * Jpdd0 + bb * Jpdd1, a1 = bb * Jpdd0 + c * Jpdd1;
            float4 j0, j1;
            j0.x = x[4] * a0 + y[4] * a1; j0.y = x[5] * a0 + y[5] * a1; j0.z = x[6] * a0 + y[6] * a1; j0.w = x[7] * a0 + y[7] * a1;
            j1.x = x[8] * a0 + y[8] * a1; j1.y = x[9] * a0 + y[9] * a1;
            j1.z = jab00 * Jpdd0 + jab01 * Jpdd1; j1.w = jab10 * Jpdd0 + jab11 * Jpdd1;
            B.rs_jp0[si] = j0; B.rs_jp1[si] = j1;
            // per-slot share of EFPoint::{bd,Hdd,Hcd}_acc (AccumulatedTopHessian.cpp:132-135); summed over the targets by ba_sc_kernel
            B.rs_pp0[si] = make_float4(JIr0 * Jpdd0 + JIr1 * Jpdd1, a0 * Jpdd0 + a1 * Jpdd1, x[0] * a0 + y[0] * a1, x[1] * a0 + y[1] * a1);
            B.rs_pp1[si] = make_float2(x[2] * a0 + y[2] * a1, x[3] * a0 + y[3] * a1);
            if (FIX || MODE == 2) {
                // relBS of FullSystemOptimize.cpp:69-71 + centerProjectedTo (makeCoarseDepthL0 input)
                const float i0 = pc[0] * pu + pc[1] * pv + pc[2], i1 = pc[3] * pu + pc[4] * pv + pc[5], i2 = pc[6] * pu + pc[7] * pv + pc[8];
                const float q0 = i0 + pc[9] * idepth, q1 = i1 + pc[10] * idepth, q2 = i2 + pc[11] * idepth;
                const float ex = i0 / i2 - q0 / q2, ey = i1 / i2 - q1 / q2;
                const float relBS = 0.01f * sqrtf(ex * ex + ey * ey);
                atomicMax(reinterpret_cast<unsigned*>(&B.pt_relbs[d]), __float_as_uint(relBS));   // non-negative floats order like their bit patterns
                B.rs_cpt[si] = make_float4(cKu, cKv, cId, relBS);
            }
        } else {
            // not IN: contributes nothing to the Hessian block
            a = bb = c = jab00 = jab01 = jab10 = jab11 = ab00 = ab01 = ab11 = JIr0 = JIr1 = Jabr0 = Jabr1 = rr = 0.f;
        }
    }
    if (MODE == 0 && t == W - 1) {
        B.en_new[d] = enew;
        if (enew >= 0.f) atomicAdd(&B.th_hist_hi[__float_as_uint(enew) >> 16], 1u);               // integer atomics: order independent
    }

    // ---- AccumulatorApprox::update / updateTopRight / updateBotRight (AccumulatedTopHessian.cpp:115-129), streamed into the reduction
    QuadStream qs(smem);
    {
        float ax[10], cy[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) { ax[i] = a * x[i] + bb * y[i]; cy[i] = bb * x[i] + c * y[i]; }
#pragma unroll
        for (int r = 0; r < 10; ++r)
#pragma unroll
            for (int cc = r; cc < 10; ++cc) qs.put(ax[r] * x[cc] + cy[r] * y[cc]);                  // 55: upper triangle of the 10x10
    }
#pragma unroll
    for (int i = 0; i < 10; ++i) {                                                                 // 30: TopRight 10x3
        qs.put(x[i] * jab00 + y[i] * jab01);
        qs.put(x[i] * jab10 + y[i] * jab11);
        qs.put(x[i] * JIr0 + y[i] * JIr1);
    }
    qs.put(ab00); qs.put(ab01); qs.put(Jabr0); qs.put(ab11); qs.put(Jabr1); qs.put(rr);           // 6: BotRight
    qs.put(cnt); qs.put(energy);                                                                   // [91] residual count, [92] energy
    qs.flush();
    __syncthreads();
    if (tid < kTopVals) {
        double s = 0.0;
#pragma unroll 8
        for (int r = 0; r < kBlk / 4; ++r) s += (double)smem[r * kTopStride + tid];                // all lanes read one row: conflict free
        B.top_partial[((size_t)b * W + t) * kTopStride + tid] = s;
    }
}


void ba_launch_linearize(hipStream_t s, const BADev& B, int mode, int fix, hipEvent_t ev_start, hipEvent_t ev_stop) {
    const unsigned grid = 8u * (unsigned)B.xcd_len * (unsigned)B.W;
    static const int coop = [] { const char* e = std::getenv("NALO_LIN_COOP"); return e ? std::atoi(e) : 1; }();   // default: quad-cooperative gathers; 0 = one residual per lane
#define NALO_LIN_LAUNCH(KERNEL)                                                                                                     \
    do {                                                                                                                            \
        if (ev_start) hipExtLaunchKernelGGL((KERNEL), dim3(grid), dim3(kBlk), 0, s, ev_start, ev_stop, 0, B);                       \
        else KERNEL<<<grid, kBlk, 0, s>>>(B);                                                                                       \
    } while (0)
    if (coop) {
        if (mode == 2) NALO_LIN_LAUNCH((ba_linearize_coop_kernel<2, 0>));
        else if (fix) NALO_LIN_LAUNCH((ba_linearize_coop_kernel<0, 1>));
        else NALO_LIN_LAUNCH((ba_linearize_coop_kernel<0, 0>));
    } else {
        if (mode == 2) NALO_LIN_LAUNCH((ba_linearize_kernel<2, 0>));
        else if (fix) NALO_LIN_LAUNCH((ba_linearize_kernel<0, 1>));
        else NALO_LIN_LAUNCH((ba_linearize_kernel<0, 0>));
    }
#undef NALO_LIN_LAUNCH
}

}  // namespace nalo
