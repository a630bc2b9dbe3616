/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  *** parity unpinned *** (see orc_common.h)
 *
 * the immature-point depth filter and the 1-D point optimisation that sit either side of the BA.
 *   orc_imm_create    ImmaturePoint::ImmaturePoint          FullSystem/ImmaturePoint.cpp:32-60
 *   orc_imm_trace     ImmaturePoint::traceOn                FullSystem/ImmaturePoint.cpp:76-435 (called per host by
 *                     FullSystem::traceNewCoarse, FullSystem.cpp:702-744, which supplies KRKi, Kt and the affine pair)
 *   orc_imm_optimize  FullSystem::optimizeImmaturePoint     FullSystem/FullSystemOptPoint.cpp:51-206, with
 *                     ImmaturePoint::linearizeResidual      FullSystem/ImmaturePoint.cpp:497-564 and the 9-argument
 *                     projectPoint / derive_idepth          FullSystem/ResidualProjections.h:36-45,61-87
 * Pointwise arithmetic in `real` (fp32 in the f32 build, left-to-right products as written), images are [w*h][3] texels
 * {I, dx, dy} of level 0 like everywhere else in the oracle. Settings: util/settings.cpp:76,99-100,136,146,165-174.
 */
#include "orc_common.h"

#define IMM_MAX_PIX_SEARCH SETTING_MAX_PIX_SEARCH
#define IMM_MIN_TRACE_TEST_RADIUS SETTING_MIN_TRACE_TEST_RADIUS
#define IMM_GN_ITS_ACTIVATION SETTING_GN_ITS_ON_POINT_ACTIVATION
#define IMM_TRACE_STEPSIZE SETTING_TRACE_STEPSIZE
#define IMM_TRACE_GN_ITS SETTING_TRACE_GN_ITERATIONS
#define IMM_TRACE_GN_TH SETTING_TRACE_GN_THRESHOLD
#define IMM_TRACE_EXTRA_SLACK SETTING_TRACE_EXTRA_SLACK_ON_TH
#define IMM_TRACE_SLACK_INTERVAL SETTING_TRACE_SLACK_INTERVAL
#define IMM_TRACE_MIN_IMPROVEMENT SETTING_TRACE_MIN_IMPROVEMENT_FACTOR
#define IMM_OUTLIER_TH SETTING_OUTLIER_TH
#define IMM_MIN_IDEPTH_H_ACT SETTING_MIN_IDEPTH_H_ACT

enum { IPS_GOOD = 0, IPS_OOB, IPS_OUTLIER, IPS_SKIPPED, IPS_BADCONDITION, IPS_UNINITIALIZED };   /* ImmaturePoint.h:47-53 */
enum { RS_IN = 0, RS_OOB = 1, RS_OUTLIER = 2 };                                                  /* Residuals.h ResState */

static inline real imm_interp31(const float* mat, real x, real y, int width) {                   /* globalFuncs.h:126-140 */
    int ix = (int)x, iy = (int)y; real dx = x - ix, dy = y - iy, dxdy = dx * dy; const float* bp = mat + 3 * (ix + iy * width);
    return dxdy * (real)bp[3 * (1 + width)] + (dy - dxdy) * (real)bp[3 * width] + (dx - dxdy) * (real)bp[3] + (1 - dx - dy + dxdy) * (real)bp[0];
}
static inline void imm_interp33(const float* mat, real x, real y, int width, real out[3]) {      /* globalFuncs.h:75-89 */
    int ix = (int)x, iy = (int)y; real dx = x - ix, dy = y - iy, dxdy = dx * dy; const float* bp = mat + 3 * (ix + iy * width);
    for (int c = 0; c < 3; c++) out[c] = dxdy * (real)bp[3 * (1 + width) + c] + (dy - dxdy) * (real)bp[3 * width + c] + (dx - dxdy) * (real)bp[3 + c] + (1 - dx - dy + dxdy) * (real)bp[c];
}

/* ImmaturePoint.cpp:32-60. u, v are integer pixel positions (PixelSelector output): getInterpolatedElement33BiLin at integer
 * coordinates (globalFuncs.h:166-188) returns {I, right-left, bottom-top} of the 2x2 neighbourhood, i.e. forward differences. */
void orc_imm_create(const float* dI_host, int w, int h, int n, const int* u, const int* v, float* color, float* weights, float* gradH, float* energyTH) {
    (void)h;
    for (int p = 0; p < n; p++) {
        real gxx = 0, gxy = 0, gyy = 0; int bad = 0;
        for (int idx = 0; idx < 8 && !bad; idx++) {
            const real x = (real)(u[p] + orc_patternP[idx][0]), y = (real)(v[p] + orc_patternP[idx][1]);
            const int ix = (int)x, iy = (int)y; const float* bp = dI_host + 3 * (ix + iy * w);
            const real tl = bp[0], tr = bp[3], bl = bp[3 * w], br = bp[3 * (w + 1)];
            const real dx = x - ix, dy = y - iy;
            const real topInt = dx * tr + (1 - dx) * tl, botInt = dx * br + (1 - dx) * bl, leftInt = dy * bl + (1 - dy) * tl, rightInt = dy * br + (1 - dy) * tr;
            const real c0 = dx * rightInt + (1 - dx) * leftInt, g0 = rightInt - leftInt, g1 = botInt - topInt;
            color[p * 8 + idx] = (float)c0;
            if (!isfinite((float)c0)) { energyTH[p] = NAN; bad = 1; break; }
            gxx += g0 * g0; gxy += g0 * g1; gyy += g1 * g1;
            weights[p * 8 + idx] = sqrtf((float)(SETTING_OUTLIER_TH_SUMCOMP / (SETTING_OUTLIER_TH_SUMCOMP + (g0 * g0 + g1 * g1))));
        }
        gradH[p * 3] = (float)gxx; gradH[p * 3 + 1] = (float)gxy; gradH[p * 3 + 2] = (float)gyy;
        if (bad) continue;
        float eth = 8 * IMM_OUTLIER_TH;
        eth *= SETTING_OVERALL_ENERGY_TH_WEIGHT * SETTING_OVERALL_ENERGY_TH_WEIGHT;
        energyTH[p] = eth;
    }
}

/* ImmaturePoint.cpp:76-435 for one point. Returns the new lastTraceStatus. */
static int trace_one(const float* dI, int w, int h, real u, real v, const float* color, const float* weights, const float* gradH, real energyTH,
                     const float* KRKi_, const float* Kt_, const float* aff_, float* idepth_min_io, float* idepth_max_io, int lastStatus, float* quality_io,
                     float* lastUV, float* lastInterval) {
    if (lastStatus == IPS_OOB) return lastStatus;
    real KRKi[9], Kt[3]; for (int i = 0; i < 9; i++) KRKi[i] = KRKi_[i]; for (int i = 0; i < 3; i++) Kt[i] = Kt_[i];
    const real aff0 = aff_[0], aff1 = aff_[1];
    real idepth_min = *idepth_min_io, idepth_max = *idepth_max_io;
    const real maxPixSearch = (w + h) * IMM_MAX_PIX_SEARCH;
    real pr[3], ptpMin[3];
    for (int i = 0; i < 3; i++) pr[i] = KRKi[i * 3] * u + KRKi[i * 3 + 1] * v + KRKi[i * 3 + 2] * 1;
    for (int i = 0; i < 3; i++) ptpMin[i] = pr[i] + Kt[i] * idepth_min;
    real uMin = ptpMin[0] / ptpMin[2], vMin = ptpMin[1] / ptpMin[2];
#define IMM_RET_OOB do { lastUV[0] = -1; lastUV[1] = -1; *lastInterval = 0; return IPS_OOB; } while (0)
    if (!(uMin > 4 && vMin > 4 && uMin < w - 5 && vMin < h - 5)) IMM_RET_OOB;
    real dist, uMax, vMax, ptpMax[3];
    if (isfinite((float)idepth_max)) {
        for (int i = 0; i < 3; i++) ptpMax[i] = pr[i] + Kt[i] * idepth_max;
        uMax = ptpMax[0] / ptpMax[2]; vMax = ptpMax[1] / ptpMax[2];
        if (!(uMax > 4 && vMax > 4 && uMax < w - 5 && vMax < h - 5)) IMM_RET_OOB;
        dist = (uMin - uMax) * (uMin - uMax) + (vMin - vMax) * (vMin - vMax);
        dist = sqrtf((float)dist);
        if (dist < IMM_TRACE_SLACK_INTERVAL) {                       /* :139-146 */
            lastUV[0] = (float)((uMax + uMin) * 0.5f); lastUV[1] = (float)((vMax + vMin) * 0.5f); *lastInterval = (float)dist;
            return IPS_SKIPPED;
        }
    } else {
        dist = maxPixSearch;
        for (int i = 0; i < 3; i++) ptpMax[i] = pr[i] + Kt[i] * 0.01f;     /* :152-161: direction towards idepth 0.01, capped length */
        uMax = ptpMax[0] / ptpMax[2]; vMax = ptpMax[1] / ptpMax[2];
        real dx = uMax - uMin, dy = vMax - vMin;
        real d = 1.0f / sqrtf((float)(dx * dx + dy * dy));
        uMax = uMin + dist * dx * d; vMax = vMin + dist * dy * d;
        if (!(uMax > 4 && vMax > 4 && uMax < w - 5 && vMax < h - 5)) IMM_RET_OOB;
    }
    if (!(idepth_min < 0 || (ptpMin[2] > 0.75f && ptpMin[2] < 1.5f))) IMM_RET_OOB;       /* :178-184: scale change */
    /* error bound from the gradient along / across the epipolar line, :187-197 */
    real dx = IMM_TRACE_STEPSIZE * (uMax - uMin), dy = IMM_TRACE_STEPSIZE * (vMax - vMin);
    const real gxx = gradH[0], gxy = gradH[1], gyy = gradH[2];
    /* (v^T gradH) v as Eigen evaluates it, v = (dx,dy) and (dy,-dx) */
    real a = (dx * gxx + dy * gxy) * dx + (dx * gxy + dy * gyy) * dy;
    real b = (dy * gxx + (-dx) * gxy) * dy + (dy * gxy + (-dx) * gyy) * (-dx);
    real errorInPixel = 0.2f + 0.2f * (a + b) / a;
    if (errorInPixel * IMM_TRACE_MIN_IMPROVEMENT > dist && isfinite((float)idepth_max)) {
        lastUV[0] = (float)((uMax + uMin) * 0.5f); lastUV[1] = (float)((vMax + vMin) * 0.5f); *lastInterval = (float)dist;
        return IPS_BADCONDITION;
    }
    if (errorInPixel > 10) errorInPixel = 10;
    dx /= dist; dy /= dist;
    if (dist > maxPixSearch) { uMax = uMin + maxPixSearch * dx; vMax = vMin + maxPixSearch * dy; dist = maxPixSearch; }
    int numSteps = (int)(1.9999f + dist / IMM_TRACE_STEPSIZE);
    const real randShift = uMin * 1000 - floorf((float)(uMin * 1000));
    real ptx = uMin - randShift * dx, pty = vMin - randShift * dy;
    real rot[8][2];
    for (int idx = 0; idx < 8; idx++) {
        rot[idx][0] = KRKi[0] * orc_patternP[idx][0] + KRKi[1] * orc_patternP[idx][1];
        rot[idx][1] = KRKi[3] * orc_patternP[idx][0] + KRKi[4] * orc_patternP[idx][1];
    }
    if (!isfinite((float)dx) || !isfinite((float)dy)) { *lastInterval = 0; lastUV[0] = -1; lastUV[1] = -1; return IPS_OOB; }
    real errors[100];
    real bestU = 0, bestV = 0, bestEnergy = 1e10f; int bestIdx = -1;
    if (numSteps >= 100) numSteps = 99;
    for (int i = 0; i < numSteps; i++) {                             /* discrete search, :275-304 */
        real energy = 0;
        for (int idx = 0; idx < 8; idx++) {
            const real hit = imm_interp31(dI, ptx + rot[idx][0], pty + rot[idx][1], w);
            if (!isfinite((float)hit)) { energy += 1e5f; continue; }
            const real residual = hit - (aff0 * color[idx] + aff1);
            const real ar = fabsf((float)residual);
            const real hw = ar < SETTING_HUBER_TH ? 1 : SETTING_HUBER_TH / ar;
            energy += hw * residual * residual * (2 - hw);
        }
        errors[i] = energy;
        if (energy < bestEnergy) { bestU = ptx; bestV = pty; bestEnergy = energy; bestIdx = i; }
        ptx += dx; pty += dy;
    }
    real secondBest = 1e10f;                                         /* :308-316 */
    for (int i = 0; i < numSteps; i++)
        if ((i < bestIdx - IMM_MIN_TRACE_TEST_RADIUS || i > bestIdx + IMM_MIN_TRACE_TEST_RADIUS) && errors[i] < secondBest) secondBest = errors[i];
    const real newQuality = secondBest / bestEnergy;
    if (newQuality < *quality_io || numSteps > 10) *quality_io = (float)newQuality;
    real uBak = bestU, vBak = bestV, gnstepsize = 1, stepBack = 0;   /* GN refinement along the line, :320-380 */
    if (IMM_TRACE_GN_ITS > 0) bestEnergy = 1e5f;
    for (int it = 0; it < IMM_TRACE_GN_ITS; it++) {
        real H = 1, bb = 0, energy = 0;
        for (int idx = 0; idx < 8; idx++) {
            real hit[3]; imm_interp33(dI, bestU + rot[idx][0], bestV + rot[idx][1], w, hit);
            if (!isfinite((float)hit[0])) { energy += 1e5f; continue; }
            const real residual = hit[0] - (aff0 * color[idx] + aff1);
            const real dResdDist = dx * hit[1] + dy * hit[2];
            const real ar = fabsf((float)residual);
            const real hw = ar < SETTING_HUBER_TH ? 1 : SETTING_HUBER_TH / ar;
            H += hw * dResdDist * dResdDist;
            bb += hw * residual * dResdDist;
            energy += weights[idx] * weights[idx] * hw * residual * residual * (2 - hw);
        }
        if (energy > bestEnergy) { stepBack *= 0.5f; bestU = uBak + stepBack * dx; bestV = vBak + stepBack * dy; }
        else {
            real step = -gnstepsize * bb / H;
            if (step < -0.5f) step = -0.5f; else if (step > 0.5f) step = 0.5f;
            if (!isfinite((float)step)) step = 0;
            uBak = bestU; vBak = bestV; stepBack = step;
            bestU += step * dx; bestV += step * dy; bestEnergy = energy;
        }
        if (fabsf((float)stepBack) < IMM_TRACE_GN_TH) break;
    }
    if (!(bestEnergy < energyTH * IMM_TRACE_EXTRA_SLACK)) {          /* :384-394 */
        *lastInterval = 0; lastUV[0] = -1; lastUV[1] = -1;
        return lastStatus == IPS_OUTLIER ? IPS_OOB : IPS_OUTLIER;
    }
    if (dx * dx > dy * dy) {                                         /* new interval, :398-408 */
        idepth_min = (pr[2] * (bestU - errorInPixel * dx) - pr[0]) / (Kt[0] - Kt[2] * (bestU - errorInPixel * dx));
        idepth_max = (pr[2] * (bestU + errorInPixel * dx) - pr[0]) / (Kt[0] - Kt[2] * (bestU + errorInPixel * dx));
    } else {
        idepth_min = (pr[2] * (bestV - errorInPixel * dy) - pr[1]) / (Kt[1] - Kt[2] * (bestV - errorInPixel * dy));
        idepth_max = (pr[2] * (bestV + errorInPixel * dy) - pr[1]) / (Kt[1] - Kt[2] * (bestV + errorInPixel * dy));
    }
    if (idepth_min > idepth_max) { const real tmp = idepth_min; idepth_min = idepth_max; idepth_max = tmp; }
    *idepth_min_io = (float)idepth_min; *idepth_max_io = (float)idepth_max;      /* the members are assigned before the validity test, :398-410 */
    if (!isfinite((float)idepth_min) || !isfinite((float)idepth_max) || (idepth_max < 0)) {
        *lastInterval = 0; lastUV[0] = -1; lastUV[1] = -1;
        return IPS_OUTLIER;
    }
    *lastInterval = (float)(2 * errorInPixel); lastUV[0] = (float)bestU; lastUV[1] = (float)bestV;
    return IPS_GOOD;
#undef IMM_RET_OOB
}

void orc_imm_trace(const float* dI_new, int w, int h, int n, const float* u, const float* v, const float* color, const float* weights, const float* gradH,
                   const float* energyTH, const int* host_idx, const float* KRKi, const float* Kt, const float* aff,
                   float* idepth_min, float* idepth_max, int* status, float* quality, float* lastUV, float* lastInterval) {
    for (int p = 0; p < n; p++) {
        const int hi = host_idx[p];
        status[p] = trace_one(dI_new, w, h, u[p], v[p], color + p * 8, weights + p * 8, gradH + p * 3, energyTH[p], KRKi + hi * 9, Kt + hi * 3, aff + hi * 2,
                              idepth_min + p, idepth_max + p, status[p], quality + p, lastUV + p * 2, lastInterval + p);
    }
}

/* ImmaturePoint::linearizeResidual, ImmaturePoint.cpp:497-564 */
typedef struct { int state, newState; double energy, newEnergy; } ImmRes;
static double imm_linearize(const float* dIl, int w, int h, const float K[4], const float* Rt, const float* affLL_, real u_pt, real v_pt, const float* color, const float* weights,
                            real energyTH, real outlierTHSlack, ImmRes* r, real* Hdd, real* bd, real idepth) {
    if (r->state == RS_OOB) { r->newState = RS_OOB; return r->energy; }
    const real fxl = K[0], fyl = K[1], cxl = K[2], cyl = K[3], fxli = 1.0f / K[0], fyli = 1.0f / K[1];
    const real wM3G = w - 3, hM3G = h - 3, affLL0 = affLL_[0], affLL1 = affLL_[1];
    real R[9], t[3]; for (int i = 0; i < 9; i++) R[i] = Rt[i]; for (int i = 0; i < 3; i++) t[i] = Rt[9 + i];
    real energyLeft = 0;
    for (int idx = 0; idx < 8; idx++) {
        const int dx = orc_patternP[idx][0], dy = orc_patternP[idx][1];
        const real k0 = (u_pt + dx - cxl) * fxli, k1 = (v_pt + dy - cyl) * fyli;                      /* projectPoint, ResidualProjections.h:61-87 */
        real ptp[3]; for (int i = 0; i < 3; i++) ptp[i] = R[i * 3] * k0 + R[i * 3 + 1] * k1 + R[i * 3 + 2] * 1 + t[i] * idepth;
        const real drescale = 1.0f / ptp[2];
        if (!(drescale > 0)) { r->newState = RS_OOB; return r->energy; }
        const real uu = ptp[0] * drescale, vv = ptp[1] * drescale, Ku = uu * fxl + cxl, Kv = vv * fyl + cyl;
        if (!(Ku > 1.1f && Kv > 1.1f && Ku < wM3G && Kv < hM3G)) { r->newState = RS_OOB; return r->energy; }
        real hit[3]; imm_interp33(dIl, Ku, Kv, w, hit);
        if (!isfinite((float)hit[0])) { r->newState = RS_OOB; return r->energy; }
        const real residual = hit[0] - (affLL0 * color[idx] + affLL1);
        const real ar = fabsf((float)residual);
        real hw = ar < SETTING_HUBER_TH ? 1 : SETTING_HUBER_TH / ar;
        energyLeft += weights[idx] * weights[idx] * hw * residual * residual * (2 - hw);
        const real dxInterp = hit[1] * fxl, dyInterp = hit[2] * fyl;
        const real d_idepth = (dxInterp * drescale * (t[0] - t[2] * uu) + dyInterp * drescale * (t[1] - t[2] * vv)) * SCALE_IDEPTH;   /* derive_idepth :36-45 */
        hw *= weights[idx] * weights[idx];
        *Hdd += (hw * d_idepth) * d_idepth;
        *bd += (hw * residual) * d_idepth;
    }
    if (energyLeft > energyTH * outlierTHSlack) { energyLeft = energyTH * outlierTHSlack; r->newState = RS_OUTLIER; }
    else r->newState = RS_IN;
    r->newEnergy = energyLeft;
    return energyLeft;
}

/* FullSystem::optimizeImmaturePoint, FullSystemOptPoint.cpp:51-206. result: 0 = not well constrained (stays immature),
 * -1 = dropped (nan idepth / too few inlier residuals / non-finite energyTH), 1 = activated with idepth_out and res_in[t] = residual created. */
void orc_imm_optimize(int W, const float* const* dI, int w, int h, const float K[4], const float* Rt /*[W*W][12], index host*W+target*/, const float* aff /*[W*W][2]*/,
                      int n, const int* host, const float* u, const float* v, const float* color, const float* weights, const float* energyTH,
                      const float* idepth_min, const float* idepth_max, int minObs, int* result, float* idepth_out, uint8_t* res_in /*[n][W]*/) {
    for (int p = 0; p < n; p++) {
        ImmRes res[ORC_MAXW]; int tgt[ORC_MAXW]; int nres = 0;
        const int hf = host[p];
        for (int t = 0; t < W; t++) if (t != hf) { res[nres].newEnergy = res[nres].energy = 0; res[nres].newState = RS_OUTLIER; res[nres].state = RS_IN; tgt[nres] = t; nres++; }
        for (int t = 0; t < W; t++) res_in[p * W + t] = 0;
        idepth_out[p] = NAN;
        real lastEnergy = 0, lastHdd = 0, lastbd = 0;
        real currentIdepth = (idepth_max[p] + idepth_min[p]) * 0.5f;
#define IMM_LIN(i, slack, Hp, bp, idp) imm_linearize(dI[tgt[i]], w, h, K, Rt + (hf * W + tgt[i]) * 12, aff + (hf * W + tgt[i]) * 2, u[p], v[p], color + p * 8, weights + p * 8, energyTH[p], slack, &res[i], Hp, bp, idp)
        /* `float += double`: the sum is formed in double and rounded once (FullSystemOptPoint.cpp:79) */
        for (int i = 0; i < nres; i++) { lastEnergy = (real)((double)lastEnergy + IMM_LIN(i, 1000, &lastHdd, &lastbd, currentIdepth)); res[i].state = res[i].newState; res[i].energy = res[i].newEnergy; }
        if (!isfinite((float)lastEnergy) || lastHdd < IMM_MIN_IDEPTH_H_ACT) { result[p] = 0; continue; }
        real lambda = 0.1f;
        for (int it = 0; it < IMM_GN_ITS_ACTIVATION; it++) {
            real H = lastHdd; H *= 1 + lambda;
            const real step = (real)((1.0 / (double)H) * (double)lastbd);            /* `(1.0/H) * lastbd` is a double expression, :99 */
            const real newIdepth = currentIdepth - step;
            real newHdd = 0, newbd = 0, newEnergy = 0;
            for (int i = 0; i < nres; i++) newEnergy = (real)((double)newEnergy + IMM_LIN(i, 1, &newHdd, &newbd, newIdepth));
            if (!isfinite((float)lastEnergy) || newHdd < IMM_MIN_IDEPTH_H_ACT) { result[p] = 0; goto next_point; }
            if (newEnergy < lastEnergy) {
                currentIdepth = newIdepth; lastHdd = newHdd; lastbd = newbd; lastEnergy = newEnergy;
                for (int i = 0; i < nres; i++) { res[i].state = res[i].newState; res[i].energy = res[i].newEnergy; }
                lambda *= 0.5f;
            } else lambda *= 5;
            if ((double)fabsf((float)step) < 0.0001 * (double)currentIdepth) break;
        }
        if (!isfinite((float)currentIdepth)) { result[p] = -1; continue; }
        {
            int numGood = 0;
            for (int i = 0; i < nres; i++) if (res[i].state == RS_IN) numGood++;
            if (numGood < minObs) { result[p] = -1; continue; }
            if (!isfinite(energyTH[p])) { result[p] = -1; continue; }          /* PointHessian inherits energyTH, :158 */
            for (int i = 0; i < nres; i++) if (res[i].state == RS_IN) res_in[p * W + tgt[i]] = 1;
            idepth_out[p] = (float)currentIdepth;
            result[p] = 1;
        }
next_point:;
#undef IMM_LIN
    }
}

/* CoarseDistanceMap::makeDistanceMap + growDistBFS (FullSystem/CoarseTracker.cpp:1410-1561). Points of every window frame
 * except `frame` are projected to level 1 of `frame` with the caller's KRKi = K[1] R Ki[0] and Kt = K[1] t (per host, floats), seeds get 0, then 39
 * BFS rounds: odd rounds grow over 8 neighbours, even rounds over 4; pixels on the image border never expand. out = fwdWarpedIDDistFinal [w1*h1]. */
void orc_dist_make_map(int w1, int h1, int frame, int n, const int* host, const float* u, const float* v, const float* idepth,
                       const float* KRKi, const float* Kt, float* out) {
    const int wh1 = w1 * h1;
    /* every pixel enters a frontier at most once (plus the seeds) */
    int* bufA = (int*)malloc(sizeof(int) * 2 * ((size_t)wh1 + n + 16)); int* bufB = (int*)malloc(sizeof(int) * 2 * ((size_t)wh1 + n + 16));
    for (int i = 0; i < wh1; i++) out[i] = 1000;
    int num = 0;
    for (int p = 0; p < n; p++) {
        if (host[p] == frame) continue;
        const float* M = KRKi + host[p] * 9; const float* T = Kt + host[p] * 3;
        real ptp[3]; for (int k = 0; k < 3; k++) ptp[k] = (real)M[k * 3] * u[p] + (real)M[k * 3 + 1] * v[p] + (real)M[k * 3 + 2] * 1 + (real)T[k] * idepth[p];
        const int uu = (int)(ptp[0] / ptp[2] + 0.5f), vv = (int)(ptp[1] / ptp[2] + 0.5f);
        if (!(uu > 0 && vv > 0 && uu < w1 && vv < h1)) continue;
        out[uu + w1 * vv] = 0;
        bufA[2 * num] = uu; bufA[2 * num + 1] = vv; num++;
    }
    int* cur = bufA; int* nxt = bufB;
    for (int k = 1; k < 40; k++) {
        int num2 = num; int* t = cur; cur = nxt; nxt = t;          /* swap: nxt holds the frontier, cur receives the new one */
        num = 0;
        static const int d4[4][2] = {{1, 0}, {-1, 0}, {0, 1}, {0, -1}};
        static const int d8[8][2] = {{1, 0}, {-1, 0}, {0, 1}, {0, -1}, {1, 1}, {-1, 1}, {-1, -1}, {1, -1}};
        const int nd = (k % 2 == 0) ? 4 : 8;
        for (int i = 0; i < num2; i++) {
            const int x = nxt[2 * i], y = nxt[2 * i + 1];
            if (x == 0 || y == 0 || x == w1 - 1 || y == h1 - 1) continue;
            for (int d = 0; d < nd; d++) {
                const int xx = x + (nd == 4 ? d4[d][0] : d8[d][0]), yy = y + (nd == 4 ? d4[d][1] : d8[d][1]);
                if (out[xx + yy * w1] > k) { out[xx + yy * w1] = (float)k; cur[2 * num] = xx; cur[2 * num + 1] = yy; num++; }
            }
        }
    }
    free(bufA); free(bufB);
}

/* PixelSelector::makeHists (FullSystem/PixelSelector2.cpp:78-142) with computeHistQuantil (:67-76): per 32x32 block the
 * histogram of (int)sqrtf(absSquaredGrad) (capped at 48) over the pixels with 1 <= x <= w-2, 1 <= y <= h-2, ths = quantile(0.5) + 7
 * (setting_minGradHistCut / setting_minGradHistAdd, settings.cpp:154-155), thsSmoothed = (3x3 box mean of ths)^2. */
void orc_pixsel_make_hists(const float* absg0, int w, int h, float* ths, float* thsSmoothed) {
    const int w32 = w / 32, h32 = h / 32;
    for (int y = 0; y < h32; y++) for (int x = 0; x < w32; x++) {
        int hist[50]; memset(hist, 0, sizeof(hist));
        for (int j = 0; j < 32; j++) for (int i = 0; i < 32; i++) {
            const int it = i + 32 * x, jt = j + 32 * y;
            if (it > w - 2 || jt > h - 2 || it < 1 || jt < 1) continue;
            int g = (int)sqrtf(absg0[it + jt * w]);
            if (g > 48) g = 48;
            hist[g + 1]++; hist[0]++;
        }
        int th = (int)(hist[0] * SETTING_MIN_GRAD_HIST_CUT + 0.5f), q = 90;
        for (int i = 0; i < 90; i++) { th -= (i + 1 < 50 ? hist[i + 1] : 0); if (th < 0) { q = i; break; } }
        ths[x + y * w32] = q + SETTING_MIN_GRAD_HIST_ADD;
    }
    for (int y = 0; y < h32; y++) for (int x = 0; x < w32; x++) {
        float sum = 0, num = 0;
        if (x > 0) { if (y > 0) { num++; sum += ths[x - 1 + (y - 1) * w32]; } if (y < h32 - 1) { num++; sum += ths[x - 1 + (y + 1) * w32]; } num++; sum += ths[x - 1 + y * w32]; }
        if (x < w32 - 1) { if (y > 0) { num++; sum += ths[x + 1 + (y - 1) * w32]; } if (y < h32 - 1) { num++; sum += ths[x + 1 + (y + 1) * w32]; } num++; sum += ths[x + 1 + y * w32]; }
        if (y > 0) { num++; sum += ths[x + (y - 1) * w32]; }
        if (y < h32 - 1) { num++; sum += ths[x + (y + 1) * w32]; }
        num++; sum += ths[x + y * w32];
        thsSmoothed[x + y * w32] = (sum / num) * (sum / num);
    }
}
