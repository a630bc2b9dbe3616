// Immature-point kernels for gfx950 (reference paths relative to src/):
//   imm_create_kernel    ImmaturePoint::ImmaturePoint          FullSystem/ImmaturePoint.cpp:32-60
//   imm_trace8_kernel    ImmaturePoint::traceOn                FullSystem/ImmaturePoint.cpp:76-435   (eight lanes per immature point)
//   imm_optimize8_kernel FullSystem::optimizeImmaturePoint     FullSystem/FullSystemOptPoint.cpp:51-206 with
//                        ImmaturePoint::linearizeResidual      FullSystem/ImmaturePoint.cpp:497-564
// These are branchy per-point searches (up to 99 line steps x 8 taps, then <= 3 GN steps), thousands of independent points per frame:
// the reference chunks them 50 per CPU thread (FullSystem.cpp:886). Every decision (status, best step, accept/reject) depends on fp32
// comparisons, so this file is built with -ffp-contract=off and written operation for operation like the scalar reference: results are
// bit-identical to the fp32 restatement (tests/test_imm_gpu.py). The per-step energies of the discrete search and the per-residual
// energies of the point optimisation live in LDS, transposed ([step][lane]): no scratch memory.
#include "nalo_internal.h"

namespace nalo {

// setting_maxPixSearch, setting_trace_*, setting_minTraceTestRadius, setting_GNItsOnPointActivation, setting_minIdepthH_act, setting_outlierTH: ref_constants.h
constexpr float kImmOutlierTH = kOutlierTH, kImmMinIdepthHAct = kMinIdepthHAct;
enum { IPS_GOOD = 0, IPS_OOB, IPS_OUTLIER, IPS_SKIPPED, IPS_BADCONDITION, IPS_UNINITIALIZED };   // ImmaturePoint.h:47-53
enum { IRS_IN = 0, IRS_OOB = 1, IRS_OUTLIER = 2 };
#define NALO_PAT(i) {kPatternDx[i], kPatternDy[i]}
__constant__ int kImmPattern[8][2] = {NALO_PAT(0), NALO_PAT(1), NALO_PAT(2), NALO_PAT(3), NALO_PAT(4), NALO_PAT(5), NALO_PAT(6), NALO_PAT(7)};   // settings.cpp:297 (ref_constants.h)
#undef NALO_PAT

__device__ __forceinline__ float imm_interp31(const float4* __restrict__ img, float x, float y, int width) {          // globalFuncs.h:126-140
    const int ix = (int)x, iy = (int)y;
    const float dx = x - ix, dy = y - iy, dxdy = dx * dy;
    const float4* bp = img + ix + iy * width;
    return dxdy * bp[1 + width].x + (dy - dxdy) * bp[width].x + (dx - dxdy) * bp[1].x + (1 - dx - dy + dxdy) * bp[0].x;
}
__device__ __forceinline__ float3 imm_interp33(const float4* __restrict__ img, float x, float y, int width) {         // globalFuncs.h:75-89
    const int ix = (int)x, iy = (int)y;
    const float dx = x - ix, dy = y - iy, dxdy = dx * dy;
    const float4* bp = img + ix + iy * width;
    const float4 p00 = bp[0], p10 = bp[1], p01 = bp[width], p11 = bp[1 + width];
    float3 r;
    r.x = dxdy * p11.x + (dy - dxdy) * p01.x + (dx - dxdy) * p10.x + (1 - dx - dy + dxdy) * p00.x;
    r.y = dxdy * p11.y + (dy - dxdy) * p01.y + (dx - dxdy) * p10.y + (1 - dx - dy + dxdy) * p00.y;
    r.z = dxdy * p11.z + (dy - dxdy) * p01.z + (dx - dxdy) * p10.z + (1 - dx - dy + dxdy) * p00.z;
    return r;
}

__global__ __launch_bounds__(256) void imm_create_kernel(const float4* __restrict__ dI, int w, int n, const int* __restrict__ u, const int* __restrict__ v,
                                                         float* __restrict__ color, float* __restrict__ weights, float* __restrict__ gradH, float* __restrict__ energyTH) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    float gxx = 0, gxy = 0, gyy = 0;
    bool bad = false;
    for (int idx = 0; idx < 8; ++idx) {
        const float x = (float)(u[p] + kImmPattern[idx][0]), y = (float)(v[p] + kImmPattern[idx][1]);
        const int ix = (int)x, iy = (int)y;
        const float4* bp = dI + ix + iy * w;                                   // getInterpolatedElement33BiLin, globalFuncs.h:166-188
        const float tl = bp[0].x, tr = bp[1].x, bl = bp[w].x, br = bp[w + 1].x;
        const float dx = x - ix, dy = y - iy;
        const float topInt = dx * tr + (1 - dx) * tl, botInt = dx * br + (1 - dx) * bl, leftInt = dy * bl + (1 - dy) * tl, rightInt = dy * br + (1 - dy) * tr;
        const float c0 = dx * rightInt + (1 - dx) * leftInt, g0 = rightInt - leftInt, g1 = botInt - topInt;
        color[p * 8 + idx] = c0;
        if (!isfinite(c0)) { energyTH[p] = NAN; bad = true; break; }
        gxx += g0 * g0; gxy += g0 * g1; gyy += g1 * g1;
        weights[p * 8 + idx] = sqrtf(kOutlierTHSumComponent / (kOutlierTHSumComponent + (g0 * g0 + g1 * g1)));
    }
    gradH[p * 3] = gxx; gradH[p * 3 + 1] = gxy; gradH[p * 3 + 2] = gyy;
    if (bad) return;
    float eth = 8 * kImmOutlierTH;
    eth *= kOverallEnergyTHWeight * kOverallEnergyTHWeight;
    energyTH[p] = eth;
}

struct ImmTraceParams {
    const float4* dI; int w, h, n;
    const float *u, *v, *color, *weights, *gradH, *energyTH;
    const int* host_idx;
    const float *KRKi, *Kt, *aff;                    // per host: [nh][9], [nh][3], [nh][2]
    float *idmin, *idmax; int* status; float *quality, *lastUV, *lastInterval;
};

// ordered sum over the 8 lanes of a point group: init + t_0 + t_1 + ... + t_7, left to right (the rounding sequence of the scalar loop); every lane
// of the group receives the total. Lane j takes the running sum of lane j-1 through DPP row_shr:1 (groups of 8 never straddle a 16-lane DPP row).
__device__ __forceinline__ float imm_group_sum(float init, float t, int l) {
    float s = init + t;                                                        // meaningful on lane 0
#pragma unroll
    for (int j = 1; j < 8; ++j) {
        const float prev = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0x111, 0xF, 0xF, true));
        if (l == j) s = prev + t;
    }
    return __shfl(s, (threadIdx.x & 56) | 7, 64);
}

// Same algorithm with EIGHT LANES PER POINT (lane l = pattern pixel l): the taps of neighbouring lanes fall on neighbouring texels (a wave-wide gather
// touches ~16 cache lines instead of 64), a step's 32 taps are one round of loads, and 8x more waves are in flight. Every scalar of the control flow is
// computed redundantly by the 8 lanes (identical values); the per-step sums over the pattern are ORDERED scans through the group (lane j adds its
// term to the running sum of lane j-1: DPP row_shr:1), so the fp32 sums are those of the sequential loop bit for bit.
__global__ __launch_bounds__(256) void imm_trace8_kernel(ImmTraceParams P) {
    __shared__ float errors[100 * 32];                                         // errors[step][group]
    const int l = threadIdx.x & 7, tid = threadIdx.x >> 3;                     // tid = group (point) inside the block
    const int p = blockIdx.x * 32 + tid;
    if (p >= P.n) return;
    const int lastStatus = P.status[p];
    if (lastStatus == IPS_OOB) return;
    const int w = P.w, h = P.h;
    const float u = P.u[p], v = P.v[p];
    const int hi = P.host_idx[p];
    float KRKi[9], Kt[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) KRKi[i] = P.KRKi[hi * 9 + i];
#pragma unroll
    for (int i = 0; i < 3; ++i) Kt[i] = P.Kt[hi * 3 + i];
    const float aff0 = P.aff[hi * 2], aff1 = P.aff[hi * 2 + 1];
    const float color_l = P.color[p * 8 + l], weight_l = P.weights[p * 8 + l];
    float idepth_min = P.idmin[p], idepth_max = P.idmax[p];
    const float energyTH = P.energyTH[p];
    const float maxPixSearch = (w + h) * kImmMaxPixSearch;
    float pr[3], ptpMin[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) pr[i] = KRKi[i * 3] * u + KRKi[i * 3 + 1] * v + KRKi[i * 3 + 2] * 1;
#pragma unroll
    for (int i = 0; i < 3; ++i) ptpMin[i] = pr[i] + Kt[i] * idepth_min;
    const float uMin = ptpMin[0] / ptpMin[2], vMin = ptpMin[1] / ptpMin[2];
    auto ret_oob = [&]() { if (l == 0) { P.lastUV[p * 2] = -1; P.lastUV[p * 2 + 1] = -1; P.lastInterval[p] = 0; P.status[p] = IPS_OOB; } };
    if (!(uMin > 4 && vMin > 4 && uMin < w - 5 && vMin < h - 5)) { ret_oob(); return; }
    float dist, uMax, vMax, ptpMax[3];
    if (isfinite(idepth_max)) {
#pragma unroll
        for (int i = 0; i < 3; ++i) ptpMax[i] = pr[i] + Kt[i] * idepth_max;
        uMax = ptpMax[0] / ptpMax[2]; vMax = ptpMax[1] / ptpMax[2];
        if (!(uMax > 4 && vMax > 4 && uMax < w - 5 && vMax < h - 5)) { ret_oob(); return; }
        dist = (uMin - uMax) * (uMin - uMax) + (vMin - vMax) * (vMin - vMax);
        dist = sqrtf(dist);
        if (dist < kImmSlackInterval) {                                        // :139-146
            if (l == 0) { P.lastUV[p * 2] = (uMax + uMin) * 0.5f; P.lastUV[p * 2 + 1] = (vMax + vMin) * 0.5f; P.lastInterval[p] = dist; }
            if (l == 0) { P.status[p] = IPS_SKIPPED; }
            return;
        }
    } else {
        dist = maxPixSearch;
#pragma unroll
        for (int i = 0; i < 3; ++i) ptpMax[i] = pr[i] + Kt[i] * 0.01f;         // :152-161
        uMax = ptpMax[0] / ptpMax[2]; vMax = ptpMax[1] / ptpMax[2];
        const float ddx = uMax - uMin, ddy = vMax - vMin;
        const float d = 1.0f / sqrtf(ddx * ddx + ddy * ddy);
        uMax = uMin + dist * ddx * d; vMax = vMin + dist * ddy * d;
        if (!(uMax > 4 && vMax > 4 && uMax < w - 5 && vMax < h - 5)) { ret_oob(); return; }
    }
    if (!(idepth_min < 0 || (ptpMin[2] > 0.75f && ptpMin[2] < 1.5f))) { ret_oob(); return; }      // :178-184
    float dx = kImmStepsize * (uMax - uMin), dy = kImmStepsize * (vMax - vMin);
    const float gxx = P.gradH[p * 3], gxy = P.gradH[p * 3 + 1], gyy = P.gradH[p * 3 + 2];
    const float a = (dx * gxx + dy * gxy) * dx + (dx * gxy + dy * gyy) * dy;                          // (v^T gradH) v, :187-188
    const float b = (dy * gxx + (-dx) * gxy) * dy + (dy * gxy + (-dx) * gyy) * (-dx);
    float errorInPixel = 0.2f + 0.2f * (a + b) / a;
    if (errorInPixel * kImmMinImprovement > dist && isfinite(idepth_max)) {
        if (l == 0) { P.lastUV[p * 2] = (uMax + uMin) * 0.5f; P.lastUV[p * 2 + 1] = (vMax + vMin) * 0.5f; P.lastInterval[p] = dist; }
        if (l == 0) { P.status[p] = IPS_BADCONDITION; }
        return;
    }
    if (errorInPixel > 10) errorInPixel = 10;
    dx /= dist; dy /= dist;
    if (dist > maxPixSearch) { uMax = uMin + maxPixSearch * dx; vMax = vMin + maxPixSearch * dy; dist = maxPixSearch; }
    int numSteps = (int)(1.9999f + dist / kImmStepsize);
    const float randShift = uMin * 1000 - floorf(uMin * 1000);
    float ptx = uMin - randShift * dx, pty = vMin - randShift * dy;
    const float rot0 = KRKi[0] * kImmPattern[l][0] + KRKi[1] * kImmPattern[l][1];
    const float rot1 = KRKi[3] * kImmPattern[l][0] + KRKi[4] * kImmPattern[l][1];
    if (!isfinite(dx) || !isfinite(dy)) { if (l == 0) { P.lastInterval[p] = 0; P.lastUV[p * 2] = -1; P.lastUV[p * 2 + 1] = -1; P.status[p] = IPS_OOB; } return; }
    float bestU = 0, bestV = 0, bestEnergy = 1e10f;
    int bestIdx = -1;
    if (numSteps >= 100) numSteps = 99;
    for (int i = 0; i < numSteps; ++i) {                                       // discrete search, :275-304
        float energy;
        {
            const float hit = imm_interp31(P.dI, ptx + rot0, pty + rot1, w);
            const float residual = hit - (aff0 * color_l + aff1);
            const float ar = fabsf(residual);
            const float hw = ar < kHuberTH ? 1 : kHuberTH / ar;
            energy = imm_group_sum(0.f, isfinite(hit) ? hw * residual * residual * (2 - hw) : 1e5f, l);
        }
        errors[i * 32 + tid] = energy;
        if (energy < bestEnergy) { bestU = ptx; bestV = pty; bestEnergy = energy; bestIdx = i; }
        ptx += dx; pty += dy;
    }
    float secondBest = 1e10f;                                                  // :308-316
    for (int i = 0; i < numSteps; ++i) {
        const float e = errors[i * 32 + tid];
        if ((i < bestIdx - kImmMinTraceTestRadius || i > bestIdx + kImmMinTraceTestRadius) && e < secondBest) secondBest = e;
    }
    const float newQuality = secondBest / bestEnergy;
    const float q0 = P.quality[p];
    if ((newQuality < q0 || numSteps > 10) && l == 0) P.quality[p] = newQuality;
    float uBak = bestU, vBak = bestV, stepBack = 0;                            // GN refinement along the line, :320-380
    const float gnstepsize = 1;
    if (kImmGNIts > 0) bestEnergy = 1e5f;
    for (int it = 0; it < kImmGNIts; ++it) {
        float H, bb, energy;
        {
            const float3 hit = imm_interp33(P.dI, bestU + rot0, bestV + rot1, w);
            const bool fin = isfinite(hit.x);
            const float residual = hit.x - (aff0 * color_l + aff1);
            const float dResdDist = dx * hit.y + dy * hit.z;
            const float ar = fabsf(residual);
            const float hw = ar < kHuberTH ? 1 : kHuberTH / ar;
            // a non-finite pixel adds 1e5 to the energy and nothing to H, b (x + 0 is exact)
            H = imm_group_sum(1.f, fin ? hw * dResdDist * dResdDist : 0.f, l);
            bb = imm_group_sum(0.f, fin ? hw * residual * dResdDist : 0.f, l);
            energy = imm_group_sum(0.f, fin ? weight_l * weight_l * hw * residual * residual * (2 - hw) : 1e5f, l);
        }
        if (energy > bestEnergy) { stepBack *= 0.5f; bestU = uBak + stepBack * dx; bestV = vBak + stepBack * dy; }
        else {
            float step = -gnstepsize * bb / H;
            if (step < -0.5f) step = -0.5f; else if (step > 0.5f) step = 0.5f;
            if (!isfinite(step)) step = 0;
            uBak = bestU; vBak = bestV; stepBack = step;
            bestU += step * dx; bestV += step * dy; bestEnergy = energy;
        }
        if (fabsf(stepBack) < kImmGNTh) break;
    }
    if (!(bestEnergy < energyTH * kImmExtraSlack)) {                           // :384-394
        if (l == 0) { P.lastInterval[p] = 0; P.lastUV[p * 2] = -1; P.lastUV[p * 2 + 1] = -1; }
        if (l == 0) { P.status[p] = lastStatus == IPS_OUTLIER ? IPS_OOB : IPS_OUTLIER; }
        return;
    }
    if (dx * dx > dy * dy) {                                                   // new interval, :398-408
        idepth_min = (pr[2] * (bestU - errorInPixel * dx) - pr[0]) / (Kt[0] - Kt[2] * (bestU - errorInPixel * dx));
        idepth_max = (pr[2] * (bestU + errorInPixel * dx) - pr[0]) / (Kt[0] - Kt[2] * (bestU + errorInPixel * dx));
    } else {
        idepth_min = (pr[2] * (bestV - errorInPixel * dy) - pr[1]) / (Kt[1] - Kt[2] * (bestV - errorInPixel * dy));
        idepth_max = (pr[2] * (bestV + errorInPixel * dy) - pr[1]) / (Kt[1] - Kt[2] * (bestV + errorInPixel * dy));
    }
    if (idepth_min > idepth_max) { const float tmp = idepth_min; idepth_min = idepth_max; idepth_max = tmp; }
    if (l == 0) { P.idmin[p] = idepth_min; P.idmax[p] = idepth_max; }                        // the members are assigned before the validity test
    if (!isfinite(idepth_min) || !isfinite(idepth_max) || (idepth_max < 0)) {
        if (l == 0) { P.lastInterval[p] = 0; P.lastUV[p * 2] = -1; P.lastUV[p * 2 + 1] = -1; }
        if (l == 0) { P.status[p] = IPS_OUTLIER; }
        return;
    }
    if (l == 0) { P.lastInterval[p] = 2 * errorInPixel; P.lastUV[p * 2] = bestU; P.lastUV[p * 2 + 1] = bestV; }
    if (l == 0) { P.status[p] = IPS_GOOD; }
}

struct ImmOptParams {
    const float4* dI[NALO_MAX_WINDOW]; int W, w, h, n, minObs;
    float fx, fy, cx, cy;
    const float *Rt, *aff;                           // [W*W][12] = PRE_RTll | PRE_tTll, [W*W][2] = PRE_aff_mode, index host*W + target
    const int* host; const float *u, *v, *color, *weights, *energyTH, *idmin, *idmax;
    const int* sel;                                  // NULL: point p reads entry p of the input arrays; else entry sel[p] (the device-resident set, nalo_imm_resident_optimize)
    int* result; float* idepth_out; uint8_t* res_in;
};


// ---- eight lanes per point (lane l = pattern pixel l), like imm_trace8_kernel. ImmaturePoint::linearizeResidual returns at the FIRST pattern pixel that
// fails (behind the camera / out of bounds / non-finite), keeping the Hdd / bd contributions of the pixels before it: the group finds that pixel with a
// ballot and the ordered scans add only the terms of the lanes in front of it.
constexpr int kImmGroups = 32;
__device__ __forceinline__ float imm_group_sum_n(float init, float t, int l, int nvalid) {
    float s = nvalid > 0 ? init + t : init;                                    // meaningful on lane 0
#pragma unroll
    for (int j = 1; j < 8; ++j) {
        const float prev = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(s), 0x111, 0xF, 0xF, true));
        if (l == j) s = j < nvalid ? prev + t : prev;
    }
    return __shfl(s, (threadIdx.x & 56) | 7, 64);
}
__device__ __forceinline__ double imm_linearize8(const ImmOptParams& P, int hf, int t, float u_pt, float v_pt, float color_l, float weight_l, int l,
                                                 float energyTH, float outlierTHSlack, unsigned& st, unsigned& nst, double* en, double* nen, int i, float& Hdd, float& bd, float idepth) {
    const int sh = 2 * i;
    if (((st >> sh) & 3u) == IRS_OOB) { nst = (nst & ~(3u << sh)) | ((unsigned)IRS_OOB << sh); return en[i * kImmGroups]; }
    const float fxl = P.fx, fyl = P.fy, cxl = P.cx, cyl = P.cy, fxli = 1.0f / P.fx, fyli = 1.0f / P.fy;
    const float wM3G = P.w - 3, hM3G = P.h - 3;
    const float* Rt = P.Rt + (size_t)(hf * P.W + t) * 12;
    const float affLL0 = P.aff[(hf * P.W + t) * 2], affLL1 = P.aff[(hf * P.W + t) * 2 + 1];
    float R[9], tt[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = Rt[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) tt[k] = Rt[9 + k];
    const int dx = kImmPattern[l][0], dy = kImmPattern[l][1];
    const float k0 = (u_pt + dx - cxl) * fxli, k1 = (v_pt + dy - cyl) * fyli;                        // projectPoint, ResidualProjections.h:61-87
    float ptp[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) ptp[k] = R[k * 3] * k0 + R[k * 3 + 1] * k1 + R[k * 3 + 2] * 1 + tt[k] * idepth;
    const float drescale = 1.0f / ptp[2];
    const float uu = ptp[0] * drescale, vv = ptp[1] * drescale, Ku = uu * fxl + cxl, Kv = vv * fyl + cyl;
    bool ok = (drescale > 0) && (Ku > 1.1f && Kv > 1.1f && Ku < wM3G && Kv < hM3G);
    const float3 hit = imm_interp33(P.dI[t], ok ? Ku : 2.f, ok ? Kv : 2.f, P.w);                     // a failing lane reads a harmless texel
    ok = ok && isfinite(hit.x);
    const unsigned long long bad = __ballot(!ok);
    const unsigned gb = (unsigned)(bad >> (threadIdx.x & 56)) & 0xFFu;
    const int nvalid = gb ? __ffs((int)gb) - 1 : 8;                                                  // pattern pixels in front of the first failure
    const float residual = hit.x - (affLL0 * color_l + affLL1);
    const float ar = fabsf(residual);
    float hw = ar < kHuberTH ? 1 : kHuberTH / ar;
    const float e_t = weight_l * weight_l * hw * residual * residual * (2 - hw);
    const float dxInterp = hit.y * fxl, dyInterp = hit.z * fyl;
    const float d_idepth = (dxInterp * drescale * (tt[0] - tt[2] * uu) + dyInterp * drescale * (tt[1] - tt[2] * vv)) * kScaleIdepth;   // derive_idepth :36-45
    hw *= weight_l * weight_l;
    Hdd = imm_group_sum_n(Hdd, (hw * d_idepth) * d_idepth, l, nvalid);
    bd = imm_group_sum_n(bd, (hw * residual) * d_idepth, l, nvalid);
    float energyLeft = imm_group_sum_n(0.f, e_t, l, nvalid);
    if (nvalid < 8) { nst = (nst & ~(3u << sh)) | ((unsigned)IRS_OOB << sh); return en[i * kImmGroups]; }
    unsigned ns;
    if (energyLeft > energyTH * outlierTHSlack) { energyLeft = energyTH * outlierTHSlack; ns = IRS_OUTLIER; } else ns = IRS_IN;
    nst = (nst & ~(3u << sh)) | (ns << sh);
    nen[i * kImmGroups] = (double)energyLeft;
    return (double)energyLeft;
}

__global__ __launch_bounds__(256) void imm_optimize8_kernel(ImmOptParams P) {
    __shared__ double en_s[(NALO_MAX_WINDOW - 1) * kImmGroups], nen_s[(NALO_MAX_WINDOW - 1) * kImmGroups];
    const int l = threadIdx.x & 7, tid = threadIdx.x >> 3;                     // tid = group (point) inside the block
    const int p = blockIdx.x * kImmGroups + tid;
    if (p >= P.n) return;
    double* en = en_s + tid; double* nen = nen_s + tid;
    const int q = P.sel ? P.sel[p] : p;                                        // where the point's inputs are; the outputs are indexed by p
    const int W = P.W, hf = P.host[q], nres = W - 1;
    const float color_l = P.color[(size_t)q * 8 + l], weight_l = P.weights[(size_t)q * 8 + l];
    const float u = P.u[q], v = P.v[q], energyTH = P.energyTH[q];
    unsigned st = 0, nst = 0;                                                  // state = IN (0) for every residual; newState = OUTLIER
    for (int i = 0; i < nres; ++i) { en[i * kImmGroups] = 0; nen[i * kImmGroups] = 0; nst |= (unsigned)IRS_OUTLIER << (2 * i); }
    if (l == 0) for (int t = 0; t < W; ++t) P.res_in[(size_t)p * W + t] = 0;
    if (l == 0) P.idepth_out[p] = NAN;
    auto tgt = [&](int i) { return i < hf ? i : i + 1; };                      // residual i <-> the i-th frame that is not the host
    float lastEnergy = 0, lastHdd = 0, lastbd = 0;
    float currentIdepth = (P.idmax[q] + P.idmin[q]) * 0.5f;
    for (int i = 0; i < nres; ++i) {
        // `float += double`: formed in double, rounded once (FullSystemOptPoint.cpp:79)
        lastEnergy = (float)((double)lastEnergy + imm_linearize8(P, hf, tgt(i), u, v, color_l, weight_l, l, energyTH, 1000.f, st, nst, en, nen, i, lastHdd, lastbd, currentIdepth));
        st = (st & ~(3u << (2 * i))) | (((nst >> (2 * i)) & 3u) << (2 * i));
        en[i * kImmGroups] = nen[i * kImmGroups];
    }
    if (!isfinite(lastEnergy) || lastHdd < kImmMinIdepthHAct) { if (l == 0) P.result[p] = 0; return; }
    float lambda = 0.1f;
    for (int it = 0; it < kImmGNItsActivation; ++it) {
        float H = lastHdd; H *= 1 + lambda;
        const float step = (float)((1.0 / (double)H) * (double)lastbd);        // `(1.0/H) * lastbd` is a double expression, :99
        const float newIdepth = currentIdepth - step;
        float newHdd = 0, newbd = 0, newEnergy = 0;
        for (int i = 0; i < nres; ++i)
            newEnergy = (float)((double)newEnergy + imm_linearize8(P, hf, tgt(i), u, v, color_l, weight_l, l, energyTH, 1.f, st, nst, en, nen, i, newHdd, newbd, newIdepth));
        if (!isfinite(lastEnergy) || newHdd < kImmMinIdepthHAct) { if (l == 0) P.result[p] = 0; return; }
        if (newEnergy < lastEnergy) {
            currentIdepth = newIdepth; lastHdd = newHdd; lastbd = newbd; lastEnergy = newEnergy;
            st = nst;
            for (int i = 0; i < nres; ++i) en[i * kImmGroups] = nen[i * kImmGroups];
            lambda *= 0.5f;
        } else lambda *= 5;
        if ((double)fabsf(step) < 0.0001 * (double)currentIdepth) break;
    }
    if (!isfinite(currentIdepth)) { if (l == 0) P.result[p] = -1; return; }
    int numGood = 0;
    for (int i = 0; i < nres; ++i) if (((st >> (2 * i)) & 3u) == IRS_IN) numGood++;
    if (numGood < P.minObs) { if (l == 0) P.result[p] = -1; return; }
    if (!isfinite(energyTH)) { if (l == 0) P.result[p] = -1; return; }                     // PointHessian inherits energyTH, :158
    if (l == 0) for (int i = 0; i < nres; ++i) if (((st >> (2 * i)) & 3u) == IRS_IN) P.res_in[(size_t)p * W + tgt(i)] = 1;
    if (l == 0) P.idepth_out[p] = currentIdepth;
    if (l == 0) P.result[p] = 1;
}

// ---------------------------------------------------------------------------------------------------------------- CoarseDistanceMap
// CoarseDistanceMap::makeDistanceMap + growDistBFS (reference src/FullSystem/CoarseTracker.cpp:1410-1561): the window's active points (already in HBM:
// pt_geo) are projected to level 1 of the newest frame, seeds get 0, then 39 breadth-first rounds (odd rounds over 8 neighbours, even rounds over 4,
// border pixels never expand). A BFS level is a pure function of the seeds within 39 pixels, so every workgroup grows its own 32x32 output tile plus a
// 39-pixel halo entirely in LDS (one byte per cell, 39 barriers, no global synchronisation); the result is the integer map of the sequential queue.
constexpr int kDistTile = 32, kDistHalo = 39, kDistSide = kDistTile + 2 * kDistHalo;     // 110
__global__ __launch_bounds__(256) void dist_seed_kernel(const float4* __restrict__ pt_geo, const uint8_t* __restrict__ pt_flags, const int* __restrict__ blk_host, int Ppad, int frame,
                                                        const float* __restrict__ KRKi, const float* __restrict__ Kt, int w1, int h1, uint8_t* __restrict__ seed) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= Ppad || !(pt_flags[d] & 1)) return;
    const int h = blk_host[d >> 8];
    if (h == frame) return;
    const float4 g = pt_geo[d];                                                 // {u, v, idepth_scaled, idepth_zero}
    const float* M = KRKi + h * 9; const float* T = Kt + h * 3;
    float ptp[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) ptp[k] = M[k * 3] * g.x + M[k * 3 + 1] * g.y + M[k * 3 + 2] * 1 + T[k] * g.z;
    const int u = (int)(ptp[0] / ptp[2] + 0.5f), v = (int)(ptp[1] / ptp[2] + 0.5f);
    if (!(u > 0 && v > 0 && u < w1 && v < h1)) return;
    seed[u + w1 * v] = 1;
}
__global__ __launch_bounds__(256) void dist_bfs_kernel(const uint8_t* __restrict__ seed, int w1, int h1, float* __restrict__ out) {
    __shared__ uint8_t cell[kDistSide * kDistSide];
    const int tiles_x = (w1 + kDistTile - 1) / kDistTile;
    const int x0 = (blockIdx.x % tiles_x) * kDistTile - kDistHalo, y0 = (blockIdx.x / tiles_x) * kDistTile - kDistHalo;
    for (int e = threadIdx.x; e < kDistSide * kDistSide; e += blockDim.x) {
        const int lx = e % kDistSide, ly = e / kDistSide, x = x0 + lx, y = y0 + ly;
        cell[e] = (x >= 0 && y >= 0 && x < w1 && y < h1 && seed[x + w1 * y]) ? 0 : 255;
    }
    __syncthreads();
    for (int k = 1; k < 40; ++k) {
        const bool eight = (k & 1) != 0;
        for (int e = threadIdx.x; e < kDistSide * kDistSide; e += blockDim.x) {
            if (cell[e] != 255) continue;
            const int lx = e % kDistSide, ly = e / kDistSide, x = x0 + lx, y = y0 + ly;
            if (x < 0 || y < 0 || x >= w1 || y >= h1) continue;
            bool hit = false;
            // a neighbour q at level k-1 reaches this cell unless q lies on the image border (:1455, :1491)
#define NALO_DIST_TAP(dx, dy) { const int qx = lx + (dx), qy = ly + (dy); if (qx >= 0 && qy >= 0 && qx < kDistSide && qy < kDistSide && cell[qx + qy * kDistSide] == k - 1) { \
                const int gx = x + (dx), gy = y + (dy); if (!(gx == 0 || gy == 0 || gx == w1 - 1 || gy == h1 - 1)) hit = true; } }
            NALO_DIST_TAP(-1, 0) NALO_DIST_TAP(1, 0) NALO_DIST_TAP(0, -1) NALO_DIST_TAP(0, 1)
            if (eight) { NALO_DIST_TAP(-1, -1) NALO_DIST_TAP(1, -1) NALO_DIST_TAP(-1, 1) NALO_DIST_TAP(1, 1) }
#undef NALO_DIST_TAP
            if (hit) cell[e] = (uint8_t)k;                                      // cells written this round hold k, never k-1: in place is race free
        }
        __syncthreads();
    }
    for (int e = threadIdx.x; e < kDistTile * kDistTile; e += blockDim.x) {
        const int lx = e % kDistTile, ly = e / kDistTile, x = x0 + kDistHalo + lx, y = y0 + kDistHalo + ly;
        if (x < w1 && y < h1) { const uint8_t c = cell[(lx + kDistHalo) + (ly + kDistHalo) * kDistSide]; out[x + w1 * y] = c == 255 ? 1000.f : (float)c; }
    }
}
int dist_make_launch(nalo_ctx* c, const float4* pt_geo, const uint8_t* pt_flags, const int* blk_host, int Ppad, int frame, const float* KRKi, const float* Kt, uint8_t* seed, float* out) {
    const int w1 = c->wl[1], h1 = c->hl[1];
    NALO_HIP(c, hipMemsetAsync(seed, 0, (size_t)w1 * h1, c->stream));
    if (Ppad > 0) dist_seed_kernel<<<(Ppad + 255) / 256, 256, 0, c->stream>>>(pt_geo, pt_flags, blk_host, Ppad, frame, KRKi, Kt, w1, h1, seed);
    {
        ProfScope ps(c, "dist_bfs");
        dist_bfs_kernel<<<((w1 + kDistTile - 1) / kDistTile) * ((h1 + kDistTile - 1) / kDistTile), 256, 0, c->stream>>>(seed, w1, h1, out);
    }
    NALO_HIP(c, hipGetLastError());
    return NALO_OK;
}

// ---------------------------------------------------------------------------------------------------------------- PixelSelector::makeHists
// PixelSelector2.cpp:78-142: one workgroup per 32x32 block builds the 50-bin histogram of (int)sqrtf(absSquaredGrad) in LDS (integer atomics), lane 0
// takes computeHistQuantil(0.5) + setting_minGradHistAdd; a second tiny launch smooths the block thresholds (3x3 mean, squared). One read of the image.
__global__ __launch_bounds__(256) void pixsel_hist_kernel(const float* __restrict__ absg0, int w, int h, int w32, float* __restrict__ ths) {
    __shared__ int hist[50];
    const int bx = blockIdx.x % w32, by = blockIdx.x / w32;
    if (threadIdx.x < 50) hist[threadIdx.x] = 0;
    __syncthreads();
    for (int e = threadIdx.x; e < 1024; e += blockDim.x) {
        const int i = e & 31, j = e >> 5, it = i + 32 * bx, jt = j + 32 * by;
        if (it > w - 2 || jt > h - 2 || it < 1 || jt < 1) continue;
        int g = (int)sqrtf(absg0[it + jt * w]);
        if (g > 48) g = 48;
        atomicAdd(&hist[g + 1], 1); atomicAdd(&hist[0], 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int th = (int)(hist[0] * kMinGradHistCut + 0.5f), q = 90;                              // computeHistQuantil (:67-76); bins above 49 are empty
        for (int i = 0; i < 90; ++i) { th -= (i + 1 < 50 ? hist[i + 1] : 0); if (th < 0) { q = i; break; } }
        ths[blockIdx.x] = q + kMinGradHistAdd;
    }
}
__global__ __launch_bounds__(256) void pixsel_smooth_kernel(const float* __restrict__ ths, int w32, int h32, float* __restrict__ thsSmoothed) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= w32 * h32) return;
    const int x = e % w32, y = e / w32;
    float sum = 0, num = 0;
    if (x > 0) { if (y > 0) { num++; sum += ths[x - 1 + (y - 1) * w32]; } if (y < h32 - 1) { num++; sum += ths[x - 1 + (y + 1) * w32]; } num++; sum += ths[x - 1 + y * w32]; }
    if (x < w32 - 1) { if (y > 0) { num++; sum += ths[x + 1 + (y - 1) * w32]; } if (y < h32 - 1) { num++; sum += ths[x + 1 + (y + 1) * w32]; } num++; sum += ths[x + 1 + y * w32]; }
    if (y > 0) { num++; sum += ths[x + (y - 1) * w32]; }
    if (y < h32 - 1) { num++; sum += ths[x + (y + 1) * w32]; }
    num++; sum += ths[x + y * w32];
    thsSmoothed[e] = (sum / num) * (sum / num);
}
int pixsel_hists_launch(nalo_ctx* c, const float* absg0, float* ths, float* thsSmoothed) {
    const int w32 = c->w / 32, h32 = c->h / 32;
    if (w32 * h32 > 0) {
        pixsel_hist_kernel<<<w32 * h32, 256, 0, c->stream>>>(absg0, c->w, c->h, w32, ths);
        pixsel_smooth_kernel<<<(w32 * h32 + 255) / 256, 256, 0, c->stream>>>(ths, w32, h32, thsSmoothed);
    }
    NALO_HIP(c, hipGetLastError());
    return NALO_OK;
}

// a few hundred floats as kernel arguments -> device memory: stream ordered, no staging buffer to protect (per-frame KRKi / Kt / aff of the resident trace)
struct ImmPutArg { float v[224]; };
__global__ void imm_put_kernel(float* __restrict__ dst, ImmPutArg a, int n) { if ((int)threadIdx.x < n) dst[threadIdx.x] = a.v[threadIdx.x]; }
int imm_put_launch(nalo_ctx* c, float* dst, const float* src, int n) {
    if (n > 224) return fail(c, NALO_ERR_ARG, "imm_put_launch: more than 224 floats");
    ImmPutArg a;
    std::memcpy(a.v, src, (size_t)n * 4);
    imm_put_kernel<<<1, 256, 0, c->stream>>>(dst, a, n);
    NALO_HIP(c, hipGetLastError());
    return NALO_OK;
}

// ---------------------------------------------------------------------------------------------------------------- launchers
int imm_stage(nalo_ctx* c, size_t words) {
    if (c->imm_cap >= words) return NALO_OK;
    if (c->imm_host) (void)hipHostFree(c->imm_host);
    c->imm_host = nullptr; c->imm_cap = 0;
    const size_t cap = words + words / 2 + 1024;
    NALO_HIP(c, hipHostMalloc((void**)&c->imm_host, cap * 4));
    NALO_HIP(c, c->imm_dev.reserve(cap));
    c->imm_cap = cap;
    return NALO_OK;
}

int imm_create_launch(nalo_ctx* c, const float4* dI, int n, const int* u, const int* v, float* color, float* weights, float* gradH, float* energyTH) {
    if (n > 0) imm_create_kernel<<<(n + 255) / 256, 256, 0, c->stream>>>(dI, c->w, n, u, v, color, weights, gradH, energyTH);
    NALO_HIP(c, hipGetLastError());
    return NALO_OK;
}
int imm_trace_launch(nalo_ctx* c, const float4* dI, int n, const float* base /* device block, see host_api.hip */, const int* host_idx, const float* KRKi, const float* Kt, const float* aff,
                     float* idmin, float* idmax, int* status, float* quality, float* lastUV, float* lastInterval) {
    ImmTraceParams P;
    P.dI = dI; P.w = c->w; P.h = c->h; P.n = n;
    P.u = base; P.v = base + n; P.color = base + 2 * (size_t)n; P.weights = base + 10 * (size_t)n; P.gradH = base + 18 * (size_t)n; P.energyTH = base + 21 * (size_t)n;
    P.host_idx = host_idx; P.KRKi = KRKi; P.Kt = Kt; P.aff = aff;
    P.idmin = idmin; P.idmax = idmax; P.status = status; P.quality = quality; P.lastUV = lastUV; P.lastInterval = lastInterval;
    if (n > 0) {
        ProfScope ps(c, "imm_trace");
        imm_trace8_kernel<<<(n + 31) / 32, 256, 0, c->stream>>>(P);
    }
    NALO_HIP(c, hipGetLastError());
    return NALO_OK;
}
// the resident set's layout (nalo_imm_resident_set): u | v | color8 | weights8 | gradH3 | energyTH | host | idmin | idmax | ..., N entries each; sel picks n of them
int imm_optimize_resident_launch(nalo_ctx* c, const float4* const* dI, int W, const float K[4], const float* Rt, const float* aff, int n, const int* sel, const float* res, size_t N,
                                 int minObs, int* result, float* idepth_out, uint8_t* res_in) {
    ImmOptParams P;
    std::memset(&P, 0, sizeof(P));
    for (int i = 0; i < W; ++i) P.dI[i] = dI[i];
    P.W = W; P.w = c->w; P.h = c->h; P.n = n; P.minObs = minObs;
    P.fx = K[0]; P.fy = K[1]; P.cx = K[2]; P.cy = K[3];
    P.Rt = Rt; P.aff = aff; P.sel = sel;
    P.u = res; P.v = res + N; P.color = res + 2 * N; P.weights = res + 10 * N; P.energyTH = res + 21 * N; P.host = (const int*)(res + 22 * N); P.idmin = res + 23 * N; P.idmax = res + 24 * N;
    P.result = result; P.idepth_out = idepth_out; P.res_in = res_in;
    if (n > 0) {
        ProfScope ps(c, "imm_optimize");
        imm_optimize8_kernel<<<(n + kImmGroups - 1) / kImmGroups, 256, 0, c->stream>>>(P);
    }
    NALO_HIP(c, hipGetLastError());
    return NALO_OK;
}
int imm_optimize_launch(nalo_ctx* c, const float4* const* dI, int W, const float K[4], const float* Rt, const float* aff, int n, const int* host, const float* base,
                        int minObs, int* result, float* idepth_out, uint8_t* res_in) {
    ImmOptParams P;
    std::memset(&P, 0, sizeof(P));
    for (int i = 0; i < W; ++i) P.dI[i] = dI[i];
    P.W = W; P.w = c->w; P.h = c->h; P.n = n; P.minObs = minObs;
    P.fx = K[0]; P.fy = K[1]; P.cx = K[2]; P.cy = K[3];
    P.Rt = Rt; P.aff = aff; P.host = host;
    P.u = base; P.v = base + n; P.color = base + 2 * (size_t)n; P.weights = base + 10 * (size_t)n; P.energyTH = base + 18 * (size_t)n; P.idmin = base + 19 * (size_t)n; P.idmax = base + 20 * (size_t)n;
    P.result = result; P.idepth_out = idepth_out; P.res_in = res_in;
    if (n > 0) {
        ProfScope ps(c, "imm_optimize");
        imm_optimize8_kernel<<<(n + kImmGroups - 1) / kImmGroups, 256, 0, c->stream>>>(P);
    }
    NALO_HIP(c, hipGetLastError());
    return NALO_OK;
}

}  // namespace nalo
