/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  *** parity unpinned *** (see orc_common.h)
 *
 * CoarseInitializer::calcResAndGS (FullSystem/CoarseInitializer.cpp:338-610) for one pyramid level, with
 * doStep (:910-938) and applyStep (:939-956). Pointwise arithmetic in `real` as written; the Accumulator9 / Accumulator11 sums
 * (unweighted updateSSE, MatrixAccumulators.h:1020-1085; updateSingleWeighted :1242-1315) are fp32 products summed in fp64.
 * Reference quirks kept: the second loop feeds E instead of EAlpha (:560-572), so EAlpha.A = 0, E.A is the value at E.finish()
 * and E.num = 2 npts; JbBuffer_new of a point that fails at pattern pixel idx keeps the partial sums of the pixels before idx.
 */
#include "orc_common.h"

static inline real init_interp31(const float* mat, real x, real y, int width) {                   /* globalFuncs.h:126-140 */
    int ix = (int)x, iy = (int)y; real dx = x - ix, dy = y - iy, dxdy = dx * dy; const float* bp = mat + 3 * (ix + iy * width);
    return dxdy * (real)bp[3 * (1 + width)] + (dy - dxdy) * (real)bp[3 * width] + (dx - dxdy) * (real)bp[3] + (1 - dx - dy + dxdy) * (real)bp[0];
}
static inline void init_interp33(const float* mat, real x, real y, int width, real out[3]) {      /* globalFuncs.h:75-89 */
    int ix = (int)x, iy = (int)y; real dx = x - ix, dy = y - iy, dxdy = dx * dy; const float* bp = mat + 3 * (ix + iy * width);
    for (int c = 0; c < 3; c++) out[c] = dxdy * (real)bp[3 * (1 + width) + c] + (dy - dxdy) * (real)bp[3 * width + c] + (dx - dxdy) * (real)bp[3 + c] + (1 - dx - dy + dxdy) * (real)bp[c];
}

/* colorRef / colorNew: [wl*hl][3] texels of level lvl. K4 = fx, fy, cx, cy of that level. RKi (3x3 row-major, float) = R * K^-1, t (float),
 * aff2 = {exp(a), b} as floats, tlog3 = refToNew.log().head<3>() as floats, t_sqnorm = refToNew.translation().squaredNorm() (double).
 * Per-point in: u, v, idepth_new, iR, isGood, energy[n][2], outlierTH. Per-point out: isGood_new, energy_new[n][2], maxstep, lastHessian_new,
 * JbBuffer_new[n][10] (rows of points with isGood == 0 are left untouched). H/b out: 8x8 + 8 each (row-major doubles), E3 = {E.A, alphaEnergy, E.num}. */
void orc_init_calc_res_and_gs(const float* colorRef, const float* colorNew, int wl, int hl, const float K4[4], const float RKi_[9], const float t_[3], const float aff2[2],
                              const float tlog3[3], double t_sqnorm, float alphaW, float alphaK, float couplingWeight,
                              int n, const float* u, const float* v, const float* idepth_new, const float* iR, const uint8_t* isGood, const float* energy, const float* outlierTH,
                              uint8_t* isGood_new, float* energy_new, float* maxstep_out, float* lastHessian_new, float* Jb,
                              double* H_out, double* b_out, double* Hsc_out, double* bsc_out, double* E3) {
    real RKi[9], t[3]; for (int i = 0; i < 9; i++) RKi[i] = RKi_[i]; for (int i = 0; i < 3; i++) t[i] = t_[i];
    const real r2new0 = aff2[0], r2new1 = aff2[1], fxl = K4[0], fyl = K4[1], cxl = K4[2], cyl = K4[3];
    double acc9[45], acc9sc[45]; memset(acc9, 0, sizeof(acc9)); memset(acc9sc, 0, sizeof(acc9sc));
    double EA = 0;
    for (int i = 0; i < n; i++) {
        real maxstep = 1e10f;
        if (!isGood[i]) {
            EA += (double)(float)energy[i * 2];
            energy_new[i * 2] = energy[i * 2]; energy_new[i * 2 + 1] = energy[i * 2 + 1]; isGood_new[i] = 0; maxstep_out[i] = (float)maxstep;
            continue;
        }
        real dp[8][8], dd[8], r[8];                           /* dp[k][idx] */
        real jb[10]; for (int k = 0; k < 10; k++) jb[k] = 0;
        int good = 1; real en = 0;
        for (int idx = 0; idx < 8; idx++) {
            const int dx = orc_patternP[idx][0], dy = orc_patternP[idx][1];
            real pt[3]; for (int k = 0; k < 3; k++) pt[k] = RKi[k * 3] * (u[i] + dx) + RKi[k * 3 + 1] * (v[i] + dy) + RKi[k * 3 + 2] * 1 + t[k] * idepth_new[i];
            const real uu = pt[0] / pt[2], vv = pt[1] / pt[2], Ku = fxl * uu + cxl, Kv = fyl * vv + cyl;
            const real new_idepth = idepth_new[i] / pt[2];
            if (!(Ku > 1 && Kv > 1 && Ku < wl - 2 && Kv < hl - 2 && new_idepth > 0)) { good = 0; break; }
            real hit[3]; init_interp33(colorNew, Ku, Kv, wl, hit);
            const real rlR = init_interp31(colorRef, u[i] + dx, v[i] + dy, wl);
            if (!isfinite((float)rlR) || !isfinite((float)hit[0])) { good = 0; break; }
            const real residual = hit[0] - r2new0 * rlR - r2new1;
            const real ar = fabsf((float)residual);
            real hw = ar < SETTING_HUBER_TH ? 1 : SETTING_HUBER_TH / ar;
            en += hw * residual * residual * (2 - hw);
            const real dxdd = (t[0] - t[2] * uu) / pt[2], dydd = (t[1] - t[2] * vv) / pt[2];
            if (hw < 1) hw = sqrtf((float)hw);
            const real dxInterp = hw * hit[1] * fxl, dyInterp = hw * hit[2] * fyl;
            dp[0][idx] = new_idepth * dxInterp;
            dp[1][idx] = new_idepth * dyInterp;
            dp[2][idx] = -new_idepth * (uu * dxInterp + vv * dyInterp);
            dp[3][idx] = -uu * vv * dxInterp - (1 + vv * vv) * dyInterp;
            dp[4][idx] = (1 + uu * uu) * dxInterp + uu * vv * dyInterp;
            dp[5][idx] = -vv * dxInterp + uu * dyInterp;
            dp[6][idx] = -hw * r2new0 * rlR;
            dp[7][idx] = -hw * 1;
            dd[idx] = dxInterp * dxdd + dyInterp * dydd;
            r[idx] = hw * residual;
            const real nx = dxdd * fxl, ny = dydd * fyl;
            const real ms = 1.0f / sqrtf((float)(nx * nx + ny * ny));
            if (ms < maxstep) maxstep = ms;
            for (int k = 0; k < 8; k++) jb[k] += dp[k][idx] * dd[idx];
            jb[8] += r[idx] * dd[idx];
            jb[9] += dd[idx] * dd[idx];
        }
        maxstep_out[i] = (float)maxstep;
        for (int k = 0; k < 10; k++) Jb[i * 10 + k] = (float)jb[k];
        if (!good || en > outlierTH[i] * 20) {
            EA += (double)(float)energy[i * 2];
            isGood_new[i] = 0; energy_new[i * 2] = energy[i * 2]; energy_new[i * 2 + 1] = energy[i * 2 + 1];
            continue;
        }
        EA += (double)(float)en;
        isGood_new[i] = 1; energy_new[i * 2] = (float)en;
        for (int idx = 0; idx < 8; idx++) {                   /* acc9.updateSSE: upper triangle of [dp0..dp7, r] outer products */
            real J[9]; for (int k = 0; k < 8; k++) J[k] = dp[k][idx]; J[8] = r[idx];
            int e = 0; for (int a = 0; a < 9; a++) for (int b = a; b < 9; b++) acc9[e++] += (double)(float)(J[a] * J[b]);
        }
    }
    /* second loop (:560-572): energy_new[1]; (E gets the regulariser terms after its finish(): A unchanged, num = 2 npts; EAlpha stays 0) */
    for (int i = 0; i < n; i++) if (isGood_new[i]) energy_new[i * 2 + 1] = (float)((idepth_new[i] - 1) * (idepth_new[i] - 1));
    float alphaEnergy = (float)((double)alphaW * (0.0 + t_sqnorm * n));
    float alphaOpt;
    if (alphaEnergy > alphaK * n) { alphaOpt = 0; alphaEnergy = alphaK * n; } else alphaOpt = alphaW;
    for (int i = 0; i < n; i++) {                             /* third loop (:590-612): per-point Schur complement */
        if (!isGood_new[i]) continue;
        float* jb = Jb + i * 10;
        lastHessian_new[i] = jb[9];
        jb[8] += alphaOpt * (idepth_new[i] - 1);
        jb[9] += alphaOpt;
        if (alphaOpt == 0) { jb[8] += couplingWeight * (idepth_new[i] - iR[i]); jb[9] += couplingWeight; }
        jb[9] = 1 / (1 + jb[9]);
        float J[9]; for (int k = 0; k < 9; k++) J[k] = jb[k];
        const float w = jb[9];
        int e = 0;
        for (int a = 0; a < 9; a++) {                         /* updateSingleWeighted: J_a*J_a*w, then J_a *= w, then J_b*J_a */
            acc9sc[e++] += (double)(float)(J[a] * J[a] * w);
            J[a] *= w;
            for (int b = a + 1; b < 9; b++) acc9sc[e++] += (double)(float)(J[b] * J[a]);
        }
    }
    /* unpack like Accumulator9::finish + topLeftCorner<8,8> / topRightCorner<8,1> */
    int e = 0;
    for (int a = 0; a < 9; a++) for (int b = a; b < 9; b++, e++) {
        const double va = (double)(float)acc9[e], vs = (double)(float)acc9sc[e];       /* acc9.H is a float matrix */
        if (b < 8) { H_out[a * 8 + b] = H_out[b * 8 + a] = va; Hsc_out[a * 8 + b] = Hsc_out[b * 8 + a] = vs; }
        else if (a < 8) { b_out[a] = va; bsc_out[a] = vs; }
    }
    for (int k = 0; k < 3; k++) { H_out[k * 8 + k] = (double)(float)((float)H_out[k * 8 + k] + alphaOpt * n); b_out[k] = (double)(float)((float)b_out[k] + tlog3[k] * alphaOpt * n); }
    E3[0] = (double)(float)EA; E3[1] = alphaEnergy; E3[2] = 2.0 * n;
}

/* CoarseInitializer::doStep (:910-938): idepth_new from the per-point Schur back-substitution */
void orc_init_do_step(int n, const uint8_t* isGood, const float* Jb /*[n][10], the applied buffer*/, const float* maxstep, const float* idepth, float lambda, const float inc[8], float* idepth_new) {
    const float maxPixelStep = 0.25f, idMaxStep = 1e10f;
    for (int i = 0; i < n; i++) {
        if (!isGood[i]) continue;
        const float* jb = Jb + i * 10;
        float dot = 0; for (int k = 0; k < 8; k++) dot += jb[k] * inc[k];
        const float b = jb[8] + dot;
        float step = -b * jb[9] / (1 + lambda);
        float ms = maxPixelStep * maxstep[i];
        if (ms > idMaxStep) ms = idMaxStep;
        if (step > ms) step = ms;
        if (step < -ms) step = -ms;
        float nid = idepth[i] + step;
        if (nid < 1e-3f) nid = 1e-3f;
        if (nid > 50) nid = 50;
        idepth_new[i] = nid;
    }
}
