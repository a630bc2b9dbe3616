// ba_linearize — a5+a6+a7 (+a13 in MODE 2) for gfx950: one lane = one residual (point d of a 256-point block of ONE host x ONE target frame).
//
//   PointFrameResidual::linearize + applyRes + EFResidual::takeDataF   (reference src/FullSystem/Residuals.cpp:78-274,306-328;
//                                                                       src/OptimizationBackend/EnergyFunctionalStructs.cpp:39-50)
//   AccumulatedTopHessianSSE::addPoint<0|2> + AccumulatorApprox        (src/OptimizationBackend/AccumulatedTopHessian.cpp:39-162,
//                                                                       MatrixAccumulators.h:754-915)
//   EFResidual::fixLinearizationF (MODE 2)                             (EnergyFunctionalStructs.cpp:89-115)
//
// The arithmetic of a residual is ONE set of __device__ functions (lin_setup / lin_pixel / lin_commit / lin_stream); the kernel around them fetches the
// 32 bilinear taps of a residual COOPERATIVELY: in round r lane q of a quad loads tap q (0,1 = the two texels of row iy, 2,3 = row iy+1) of residual
// r's pixel, so neighbouring lanes read neighbouring texels (half the cache-line accesses of one-residual-per-lane gathers), and the texels return to their
// owner through a per-wave LDS exchange. The texels come from a second copy of level 0: 12-byte {I,dx,dy} texels in 5x2 tiles of one 128-byte cache line
// (frame_tile_level0, kernels_pyramid.hip): ~7.0 lines per residual footprint instead of ~9.8 with row-major 16-byte texels. Points are Hilbert-ordered per host (host_ba.hip), so a wave's residuals project into a
// compact patch of the target image.
//
// Grid = (target, point block [, quarter]), TARGET-MAJOR and XCD-aware: all CUs gather from the same target image at a time. The FrameFramePrecalc
// record is workgroup-uniform (scalar loads). Two workgroup shapes share the code (template WG):
//   WG = 256  a whole 256-point block: the 93 reduced values of the four waves meet in 64 LDS rows and ONE fp64 partial per (block, target) is written
//             (large windows: a quarter of the partial traffic);
//   WG = 64   one wave = a quarter block with its own partial (small windows: a KITTI-sized window has ~12 blocks x 8 targets, four times as many
//             workgroups fill more of the 256 CUs: 14.0 -> 11.7 us per launch).
// No float atomics anywhere; ba_reduce_kernel adds the partials in a fixed order.
//
// Tried and dropped in round 2 (numbers in profiles/README.md): staging the wave's window of the PLANAR irradiance in LDS (coalesced 16-byte
// loads, taps as LDS reads, gradients recomputed as the pyramid computes them). 86-93 % of the (wave, target) windows of the 250k-point stress window
// fit 36-48 KB, but a 36 KB window per wave leaves ONE wave per SIMD, and the kernel issues ~3000 vector instructions per wave: alone on its SIMD a
// wave issues one per ~4 cycles, so the pass took 387-560 us against 207 us for the gathers at three waves per SIMD.
#include "nalo_internal.h"
#include <hip/hip_ext.h>
#include <cstdlib>
#include <type_traits>
#include "ba_device.h"
#include "reduce.h"

#ifndef NALO_LIN_COOP_NPB
#define NALO_LIN_COOP_NPB 2        // pattern pixels per gather batch: 2 (8 loads in flight per lane) x 4 waves per SIMD. Round-2 sweep (profiles/r02_tune_lin.log):
#endif                             // 3 waves x 3+3+2 pixels 207.6 / 919.9 us (stress250k / shard1m), 4 x 2: 208.4 / 871.4, 4 x 3 (spills) 228.6 / 958.7, 5 x 2: 237.1 / 975.5
#ifndef NALO_LIN_COOP_WAVES
#define NALO_LIN_COOP_WAVES 4      // waves per SIMD the gather kernel's register allocation leaves room for (128 VGPRs)
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

// streams values 0..N-1 (in order) into the LDS row of the lane's quad: every 4 values -> quad DPP adds -> one float4 store per quad
struct QuadStream {
    float4* row4;
    bool writer;
    float b0, b1, b2;
    int k;
    __device__ __forceinline__ QuadStream(float* rows, int tid) : row4(reinterpret_cast<float4*>(rows + (tid >> 2) * kTopStride)), writer((tid & 3) == 0), b0(0.f), b1(0.f), b2(0.f), k(0) {}
    __device__ __forceinline__ void put(float v) {
        v = dpp_quad_sum(v);
        const int m = k & 3;
        if (m == 0) b0 = v; else if (m == 1) b1 = v; else if (m == 2) b2 = v;
        else if (writer) row4[k >> 2] = make_float4(b0, b1, b2, v);
        ++k;
    }
    __device__ __forceinline__ void flush() {            // pad the last group with zeros
        while (k & 3) put(0.f);
    }
};

struct alignas(16) LinTexel { float x, y, z; };           // {I, dx, dy}: a 12-byte texel in a 16-byte LDS slot

// wave-scope ordering of LDS traffic: the hardware keeps one wave's LDS operations in order; the fences keep the COMPILER from moving a lane's reads above
// other lanes' writes (for this lane provably non-aliasing) or the next writes above these reads
__device__ __forceinline__ void lin_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---------------------------------------------------------------------------------------------------------------- the residual, per lane
struct LinRes {
    // everything the accumulation needs; stays zero unless this lane ends with an active (IN) residual
    float a, bb, c, jab00, jab01, jab10, jab11, ab00, ab01, ab11;
    float JIr0, JIr1, Jabr0, Jabr1, rr, cnt, energy, enew;
    // geometry / state
    float pu, pv, idepth, idz;
    float2 en;
    int state, newState;
    bool exists, need, full;
    uint8_t st;
    float Jpdd0, Jpdd1, cKu, cKv, cId, jx, jy;
    float Kus[8], Kvs[8];
    float energyLeft, wJI2_sum;
    bool finite_ok;
};

// workgroup -> (target t, point block b, quarter q): workgroups are dealt round-robin over the 8 XCDs, so launch index j of a target runs on XCD group
// j % 8. Group x walks blk_order[x][*] = the x-th spatial eighth (Hilbert range) of every host's points, so each XCD's private 4 MiB L2 only ever
// sees ~1/8 of the target image instead of all of it (speed only, never correctness). WG = 64: the four quarters of a block stay on one XCD.
struct LinWhere { int t, b, q, h, d; size_t si; const float* pc; bool skip; };
template <int WG>
__device__ __forceinline__ LinWhere lin_where(const BADev& B, int tid) {
    LinWhere w;
    constexpr int SUB = kBlk / WG;                                      // workgroups per point block
    const int per_t = 8 * SUB * B.xcd_len, j = blockIdx.x % per_t, pos = j >> 3;
    w.t = blockIdx.x / per_t;
    w.b = B.blk_order[(j & 7) * B.xcd_len + pos / SUB];
    w.q = pos % SUB;
    w.skip = w.b < 0;
    const int b = w.skip ? 0 : w.b;
    w.h = B.blk_host[b];
    w.d = b * kBlk + w.q * WG + tid;
    w.si = (size_t)w.t * B.Ppad + w.d;
    w.pc = B.pre + (size_t)(w.h * B.W + w.t) * kPreStride;             // workgroup-uniform: scalar loads
    return w;
}

// The per-residual outputs (JpJdF, the point-sum shares, the energies: 64 B per residual, 114 MB on stress250k) are written once here and read once by the
// next kernels: stored with the nontemporal hint they do not evict image lines from the L2 the gathers live in (stress250k: 203.0 -> 193.3 us on one box, back to
// back; shard1m unchanged). The same hint on the point-record LOADS costs 3-5 % (they are re-read per target frame and do hit) - not used.
#ifndef NALO_LIN_NT_STORES
#define NALO_LIN_NT_STORES 1
#endif
typedef float lin_f4 __attribute__((ext_vector_type(4)));
typedef float lin_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void lin_store(float4* p, const float4& v) {
#if NALO_LIN_NT_STORES
    lin_f4 t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w; __builtin_nontemporal_store(t, reinterpret_cast<lin_f4*>(p));
#else
    *p = v;
#endif
}
__device__ __forceinline__ void lin_store(float2* p, const float2& v) {
#if NALO_LIN_NT_STORES
    lin_f2 t; t.x = v.x; t.y = v.y; __builtin_nontemporal_store(t, reinterpret_cast<lin_f2*>(p));
#else
    *p = v;
#endif
}

// x = (Jpdc[0], Jpdxi[0]), y = (Jpdc[1], Jpdxi[1]) of a residual (Residuals.cpp:108-156) from the point and the precalc record. A function of its own so that the
// twenty values are NOT alive across the gather phase: they are (re)computed where they are consumed (takeDataF and the accumulation, after the last tap),
// which is what lets the kernel fit 128 vector registers = four waves per SIMD (the gathers are latency bound: more waves, more loads in flight).
__device__ __forceinline__ void lin_geometry(const BADev& B, const LinWhere& w, float pu, float pv, float idz, float* __restrict__ x, float* __restrict__ y) {
    const float* pc = w.pc;
    const float cal_fxl = B.calib[0], cal_fyl = B.calib[1], cal_cxl = B.calib[2], cal_cyl = B.calib[3], cal_fxli = B.calib[4], cal_fyli = B.calib[5];
    const float KliP0 = (pu - cal_cxl) * cal_fxli, KliP1 = (pv - cal_cyl) * cal_fyli;
    const float p0 = pc[12] * KliP0 + pc[13] * KliP1 + pc[14] + pc[21] * idz;
    const float p1 = pc[15] * KliP0 + pc[16] * KliP1 + pc[17] + pc[22] * idz;
    const float p2 = pc[18] * KliP0 + pc[19] * KliP1 + pc[20] + pc[23] * idz;
    const float drescale = 1.0f / p2, new_idepth = idz * drescale;
    const float u = p0 * drescale, vv = p1 * drescale;
    x[2] = drescale * (pc[18] * u - pc[12]);                                                        // :123-156
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
}

// phase A: state, the two projections, geometric Jacobians (Residuals.cpp:78-170, ResidualProjections.h:47-87). Sets R.need when the residual takes its taps.
template <int MODE>
__device__ __forceinline__ void lin_setup(const BADev& B, const LinWhere& w, LinRes& R) {
    const float* pc = w.pc;
    // CalibHessian::value_scaledf / value_scaledi live in device memory (uniform scalar loads): the GN step may be taken on the device
    const float cal_fxl = B.calib[0], cal_fyl = B.calib[1], cal_cxl = B.calib[2], cal_cyl = B.calib[3], cal_fxli = B.calib[4], cal_fyli = B.calib[5];
    R.a = R.bb = R.c = R.jab00 = R.jab01 = R.jab10 = R.jab11 = R.ab00 = R.ab01 = R.ab11 = 0.f;
    R.JIr0 = R.JIr1 = R.Jabr0 = R.Jabr1 = R.rr = R.cnt = R.energy = 0.f; R.enew = -1.f;
    R.pu = R.pv = R.idepth = R.idz = 0.f; R.en = make_float2(0.f, 0.f);
    R.state = 0; R.newState = 2; R.full = false; R.need = false;
    R.Jpdd0 = R.Jpdd1 = R.cKu = R.cKv = R.cId = R.jx = R.jy = 0.f;
    R.energyLeft = 0.f; R.wJI2_sum = 0.f; R.finite_ok = true;
#pragma unroll
    for (int k = 0; k < 8; ++k) { R.Kus[k] = 2.f; R.Kvs[k] = 2.f; }
    const uint8_t pf = B.pt_flags[w.d];
    const bool pvalid = (pf & PT_VALID) && (MODE == 0 || (pf & PT_MARG));
    R.st = B.rs_state[w.si];
    R.exists = pvalid && (R.st & RS_EXISTS) && (MODE == 2 || !(R.st & RS_LINEARIZED));
    if (!R.exists) return;
    const float4 geo = B.pt_geo[w.d];
    R.pu = geo.x; R.pv = geo.y; R.idepth = geo.z; R.idz = geo.w;
    R.en = B.rs_energy[w.si];
    R.state = R.st & RS_STATE_MASK;
    if (MODE == 2) { R.en.x = 0.f; R.en.y = 0.f; R.state = 0; R.st &= ~RS_LINEARIZED; }            // resetOOB + isLinearized=false (FullSystem.cpp:978-981)
    else if (B.reset_oob) { R.en.x = 0.f; R.en.y = 0.f; R.state = 0; }                             // resetOOB of every active residual at the start of optimize() (FullSystemOptimize.cpp:412-429): was a launch of its own
    R.energy = R.en.x;
    if (R.state == 1) { R.newState = 1; return; }                                                   // Residuals.cpp:82-83
    const float pu = R.pu, pv = R.pv, idepth = R.idepth, idz = R.idz;
    const float wM3G = (float)(B.w - 3), hM3G = (float)(B.h - 3);
    const float KliP0 = (pu - cal_cxl) * cal_fxli, KliP1 = (pv - cal_cyl) * cal_fyli;               // ResidualProjections.h:70-73
    // ---- centre projection at idepth_zero (projectPoint, ResidualProjections.h:61-87)
    const float p0 = pc[12] * KliP0 + pc[13] * KliP1 + pc[14] + pc[21] * idz;
    const float p1 = pc[15] * KliP0 + pc[16] * KliP1 + pc[17] + pc[22] * idz;
    const float p2 = pc[18] * KliP0 + pc[19] * KliP1 + pc[20] + pc[23] * idz;
    const float drescale = 1.0f / p2, new_idepth = idz * drescale;
    const float u = p0 * drescale, vv = p1 * drescale;
    const float Ku0 = u * cal_fxl + cal_cxl, Kv0 = vv * cal_fyl + cal_cyl;
    bool ok = (drescale > 0.f) && Ku0 > 1.1f && Kv0 > 1.1f && Ku0 < wM3G && Kv0 < hM3G;
    // ---- the 8 pattern pixels at the current idepth (projectPoint, ResidualProjections.h:47-57)
    constexpr int pdx[8] = NALO_PATTERN_DX, pdy[8] = NALO_PATTERN_DY;                               // util/settings.cpp:297 (ref_constants.h)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float xx = pu + (float)pdx[k], yy = pv + (float)pdy[k];
        const float q0 = pc[0] * xx + pc[1] * yy + pc[2] + pc[9] * idepth;
        const float q1 = pc[3] * xx + pc[4] * yy + pc[5] + pc[10] * idepth;
        const float q2 = pc[6] * xx + pc[7] * yy + pc[8] + pc[11] * idepth;
        R.Kus[k] = q0 / q2; R.Kvs[k] = q1 / q2;
        ok = ok && R.Kus[k] > 1.1f && R.Kvs[k] > 1.1f && R.Kus[k] < wM3G && R.Kvs[k] < hM3G;
    }
    if (!ok) { R.newState = 1; return; }
    R.need = true;
    R.cKu = Ku0; R.cKv = Kv0; R.cId = new_idepth;
    const float t0x = pc[21], t0y = pc[22], t0z = pc[23];
    R.Jpdd0 = drescale * (t0x - t0z * u) * kScaleIdepth * cal_fxl;                                   // Residuals.cpp:116-117
    R.Jpdd1 = drescale * (t0y - t0z * vv) * kScaleIdepth * cal_fyl;
    if (MODE == 2) {                                                                                // Jp*delta (EnergyFunctionalStructs.cpp:94-99)
        float x[10], y[10];
        lin_geometry(B, w, R.pu, R.pv, R.idz, x, y);
        const float dd = idepth - idz;
        float jx = 0.f, jy = 0.f;
#pragma unroll
        for (int i = 0; i < 6; ++i) { jx += x[4 + i] * pc[27 + i]; jy += y[4 + i] * pc[27 + i]; }
        float cxs = 0.f, cys = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { cxs += x[i] * B.calib[6 + i]; cys += y[i] * B.calib[6 + i]; }
        R.jx = jx + cxs + R.Jpdd0 * dd; R.jy = jy + cys + R.Jpdd1 * dd;
    }
}

// the photometric part of one pattern pixel (Residuals.cpp:183-245)
template <int MODE>
__device__ __forceinline__ void lin_pixel(LinRes& R, const float* pc, float hitI, float hitX, float hitY, float col, float wg, bool fixA, bool fixB) {
    const float affLL0 = pc[24], affLL1 = pc[25], b0 = pc[26];
    const float residual = hitI - (affLL0 * col + affLL1);
    const float drdA = col - b0;
    R.finite_ok = R.finite_ok && isfinite(hitI);
    float wgt_k = sqrtf(kOutlierTHSumComponent / (kOutlierTHSumComponent + (hitX * hitX + hitY * hitY)));
    wgt_k = 0.5f * (wgt_k + wg);
    const float ar = fabsf(residual);
    float hw = ar < kHuberTH ? 1.f : kHuberTH / ar;
    R.energyLeft += wgt_k * wgt_k * hw * residual * residual * (2.f - hw);
    if (hw < 1.f) hw = sqrtf(hw);
    hw = hw * wgt_k;
    const float jI0 = hitX * hw, jI1 = hitY * hw, resF = residual * hw, jA = drdA * hw, jB = hw;
    R.a += jI0 * jI0; R.c += jI1 * jI1; R.bb += jI0 * jI1;
    R.jab00 += jA * jI0; R.jab01 += jA * jI1; R.jab10 += jB * jI0; R.jab11 += jB * jI1;
    R.ab00 += jA * jA; R.ab01 += jA * jB; R.ab11 += jB * jB;
    R.wJI2_sum += hw * hw * (jI0 * jI0 + jI1 * jI1);                                               // on the hw-scaled gradient, as :215-239
    // setting_affineOptModeA / B < 0: J->JabF[0] / [1] are zeroed AFTER the sums above (Residuals.cpp:241-242), so only what reads JabF itself sees it:
    // res_toZeroF (EnergyFunctionalStructs.cpp:107-108) and Jab_r (AccumulatedTopHessian.cpp:108-109)
    const float jAf = fixA ? 0.f : jA, jBf = fixB ? 0.f : jB;
    float ra = resF;                                                                                // mode 2: res_toZeroF (:103-111)
    if (MODE == 2) ra = resF - jI0 * R.jx - jI1 * R.jy - jAf * pc[33] - jBf * pc[34];
    R.JIr0 += ra * jI0; R.JIr1 += ra * jI1; R.Jabr0 += ra * jAf; R.Jabr1 += ra * jBf; R.rr += ra * ra;
}

// energy threshold, applyRes(true), takeDataF, the per-slot stores (Residuals.cpp:260-273, 306-328; EnergyFunctionalStructs.cpp:39-50)
template <int MODE, int FIX>
__device__ __forceinline__ void lin_commit(const BADev& B, const LinWhere& w, LinRes& R) {
    const float* pc = w.pc;
    if (R.need) {
        if (!R.finite_ok) { R.newState = 1; }
        else {
            R.full = true;
            R.enew = R.energyLeft;
            const float th = fmaxf(B.frameTH[w.h], B.frameTH[w.t]);
            if (R.energyLeft > th || R.wJI2_sum < 2.f) { R.energyLeft = th; R.newState = 2; } else R.newState = 0;   // :262-270
            R.en.y = R.energyLeft;
            R.energy = R.energyLeft;
        }
    }
    if (FIX == 2) {
        // linearizeAll(false) WITHOUT applyRes (setting_forceAceptStep = false, FullSystemOptimize.cpp:511-541): only state_NewEnergy is kept; state, Jacobian
        // products and accumulators stay those of the last applied linearisation. An accepted step re-runs the pass with FIX = 0 at the same threshold.
        if (R.exists) B.rs_energy[w.si] = R.en;
    } else if (R.exists) {
        // ---- applyRes(true) (Residuals.cpp:306-328)
        bool active = false;
        if (R.state != 1) { active = (R.newState == 0); R.state = R.newState; R.en.x = R.en.y; }
        uint8_t st = (uint8_t)((R.st & ~(RS_STATE_MASK | RS_ACTIVE)) | R.state | (active ? RS_ACTIVE : 0));
        if (MODE == 2 && active) st |= RS_LINEARIZED;
        if (FIX == 1 && !active) st &= ~(RS_EXISTS | RS_ACTIVE);                                      // toRemove (FullSystemOptimize.cpp:81-84,184-205)
        B.rs_state[w.si] = st;
        lin_store(&B.rs_energy[w.si], R.en);
        if (active && R.full) {
            R.cnt = 1.f;
            float x[10], y[10];
            lin_geometry(B, w, R.pu, R.pv, R.idz, x, y);
            // ---- takeDataF (EnergyFunctionalStructs.cpp:39-50)
            const float a0 = R.a * R.Jpdd0 + R.bb * R.Jpdd1, a1 = R.bb * R.Jpdd0 + R.c * R.Jpdd1;
            float4 j0, j1;
            j0.x = x[4] * a0 + y[4] * a1; j0.y = x[5] * a0 + y[5] * a1; j0.z = x[6] * a0 + y[6] * a1; j0.w = x[7] * a0 + y[7] * a1;
            j1.x = x[8] * a0 + y[8] * a1; j1.y = x[9] * a0 + y[9] * a1;
            j1.z = R.jab00 * R.Jpdd0 + R.jab01 * R.Jpdd1; j1.w = R.jab10 * R.Jpdd0 + R.jab11 * R.Jpdd1;
            lin_store(&B.rs_jp0[w.si], j0); lin_store(&B.rs_jp1[w.si], j1);
            // per-slot share of EFPoint::{bd,Hdd,Hcd}_acc (AccumulatedTopHessian.cpp:132-135); summed over the targets inside ba_sc_kernel
            lin_store(&B.rs_pp0[w.si], make_float4(R.JIr0 * R.Jpdd0 + R.JIr1 * R.Jpdd1, a0 * R.Jpdd0 + a1 * R.Jpdd1, x[0] * a0 + y[0] * a1, x[1] * a0 + y[1] * a1));
            lin_store(&B.rs_pp1[w.si], make_float2(x[2] * a0 + y[2] * a1, x[3] * a0 + y[3] * a1));
            if (FIX == 1 || MODE == 2) {
                // relBS of FullSystemOptimize.cpp:69-71 + centerProjectedTo (makeCoarseDepthL0 input)
                const float pu = R.pu, pv = R.pv, idepth = R.idepth;
                const float i0 = pc[0] * pu + pc[1] * pv + pc[2], i1 = pc[3] * pu + pc[4] * pv + pc[5], i2 = pc[6] * pu + pc[7] * pv + pc[8];
                const float q0 = i0 + pc[9] * idepth, q1 = i1 + pc[10] * idepth, q2 = i2 + pc[11] * idepth;
                const float ex = i0 / i2 - q0 / q2, ey = i1 / i2 - q1 / q2;
                const float relBS = 0.01f * sqrtf(ex * ex + ey * ey);
                atomicMax(reinterpret_cast<unsigned*>(&B.pt_relbs[w.d]), __float_as_uint(relBS));   // non-negative floats order like their bit patterns
                B.rs_cpt[w.si] = make_float4(R.cKu, R.cKv, R.cId, relBS);
            }
        } else {
            // not IN: contributes nothing to the Hessian block
            R.a = R.bb = R.c = R.jab00 = R.jab01 = R.jab10 = R.jab11 = R.ab00 = R.ab01 = R.ab11 = R.JIr0 = R.JIr1 = R.Jabr0 = R.Jabr1 = R.rr = 0.f;
        }
    }
    if (MODE == 0 && w.t == B.W - 1 && !B.no_th) {
        B.en_new[w.d] = R.enew;                          // the input of setNewFrameEnergyTH's radix select (kernels_ba.hip: ba_th_fill_kernel)
    }
}

// AccumulatorApprox::update / updateTopRight / updateBotRight (AccumulatedTopHessian.cpp:115-129) of this lane's residual, streamed into the LDS row of its quad
__device__ __forceinline__ void lin_stream(const BADev& B, const LinWhere& w, const LinRes& R, float* rows, int tid) {
    QuadStream qs(rows, tid);
    float x[10], y[10];
    if (R.cnt != 0.f) lin_geometry(B, w, R.pu, R.pv, R.idz, x, y);                                 // only lanes that end with an active residual contribute
    else {
#pragma unroll
        for (int i = 0; i < 10; ++i) { x[i] = 0.f; y[i] = 0.f; }
    }
    {
        float ax[10], cy[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) { ax[i] = R.a * x[i] + R.bb * y[i]; cy[i] = R.bb * x[i] + R.c * y[i]; }
#pragma unroll
        for (int r = 0; r < 10; ++r)
#pragma unroll
            for (int cc = r; cc < 10; ++cc) qs.put(ax[r] * x[cc] + cy[r] * y[cc]);                  // 55: upper triangle of the 10x10
    }
#pragma unroll
    for (int i = 0; i < 10; ++i) {                                                                 // 30: TopRight 10x3
        qs.put(x[i] * R.jab00 + y[i] * R.jab01);
        qs.put(x[i] * R.jab10 + y[i] * R.jab11);
        qs.put(x[i] * R.JIr0 + y[i] * R.JIr1);
    }
    qs.put(R.ab00); qs.put(R.ab01); qs.put(R.Jabr0); qs.put(R.ab11); qs.put(R.Jabr1); qs.put(R.rr);   // 6: BotRight
    qs.put(R.cnt); qs.put(R.energy);                                                               // [91] residual count, [92] energy
    qs.flush();
}

template <int R_> __device__ __forceinline__ int lin_quad_bcast(int v) {        // value of lane R_ of this lane's quad (DPP quad_perm [R_,R_,R_,R_])
    return __builtin_amdgcn_update_dpp(0, v, R_ * 0x55, 0xF, 0xF, true);
}

// MODE 0: active residuals (optimize). MODE 2: marginalisation of the flagged points (resApprox = res_toZeroF).
// FIX: linearizeAll(true) — residuals that do not end IN are dropped; centerProjectedTo / relBS are stored.
template <int MODE, int FIX, int WG>
__global__ __launch_bounds__(WG, NALO_LIN_COOP_WAVES) void ba_linearize_kernel(BADev B) {
    __shared__ __attribute__((aligned(16))) float smem[(WG / 4) * kTopStride];
    static_assert(16 * kTopStride * 4 >= 4 * 65 * 16, "a wave's reduction rows must hold its exchange buffer");
    const int tid = threadIdx.x;
    if (B.gate_p) {                                                     // enqueued ahead of its precalc records (ba_device.h: GateBlock): wait for the host's word, bounded
        if constexpr (WG == 64) {
            int ok = 1;
            if (tid == 0) ok = gate_wait(B.gate_p, B.gate_p_want, B.gate_err) ? 1 : 0;
            if (!__builtin_amdgcn_readfirstlane(ok)) return;
        } else {
            __shared__ int gate_ok;
            if (tid == 0) gate_ok = gate_wait(B.gate_p, B.gate_p_want, B.gate_err) ? 1 : 0;
            __syncthreads();
            if (!gate_ok) return;
        }
    }
    const LinWhere w = lin_where<WG>(B, tid);
    if (w.skip) return;
    if (w.t == w.h) {                                                   // no self residuals; the newest frame's own points have no entry
        if (MODE == 0 && w.t == B.W - 1 && !B.no_th) B.en_new[w.d] = -1.f;
        if (FIX == 1 || MODE == 2) B.pt_relbs_next[w.d] = 0.f;          // every point has exactly one such workgroup lane: the other buffer is clean for the next fix pass
        return;
    }
    const bool fixA = B.fix_a != 0, fixB = B.fix_b != 0;
    LinRes R;
    lin_setup<MODE>(B, w, R);
    float color[8], wgt[8];
    if (R.need) {
        const float4 c0 = B.pt_col0[w.d], c1 = B.pt_col1[w.d], w0 = B.pt_w0[w.d], w1 = B.pt_w1[w.d];
        color[0] = c0.x; color[1] = c0.y; color[2] = c0.z; color[3] = c0.w; color[4] = c1.x; color[5] = c1.y; color[6] = c1.z; color[7] = c1.w;
        wgt[0] = w0.x; wgt[1] = w0.y; wgt[2] = w0.z; wgt[3] = w0.w; wgt[4] = w1.x; wgt[5] = w1.y; wgt[6] = w1.z; wgt[7] = w1.w;
    }
    {
        // the target frame's level 0 as 12-byte texels in 5x2 tiles of 128 bytes (frame_tile_level0): fewer cache lines per bilinear footprint, 25 % more image per line
        const float* __restrict__ img = B.img_t[w.t];
        const int lane = tid & 63, q = lane & 3, Q = lane >> 2;
        const int wt = B.wt;
        // The lane that owns a residual resolves its pixel once: float index of tap (0,0) plus two flags - does x + 1 / y + 1 leave the tile -, packed into one word
        // (index < 2^30); the quad's lanes then only add their tap's offset to the word a quad lane broadcast (6 instead of 13 vector instructions per tap).
        auto pack = [&](float Ku, float Kv) __attribute__((always_inline)) -> int {
            const int x = (int)Ku, y = (int)Kv, tx = (x * 52429) >> 18, rx = x - tx * 5, ry = y & 1;         // x / 5, exact below 43690
            return ((((y >> 1) * wt + tx) << 5) + (ry * 5 + rx) * 3) | (rx == 4 ? 1 << 30 : 0) | (int)((unsigned)ry << 31);
        };
        const int ox0 = (q & 1) ? 3 : 0, ox1 = (q & 1) ? 20 : 0;                   // x + 1: next texel (+3 floats), or texel 0 of the next tile (+32 - 12)
        const int oy0 = (q >> 1) ? 15 : 0, oy1 = (q >> 1) ? wt * 32 - 15 : 0;      // y + 1: second row of the tile (+15), or first row of the tile below
        auto tap = [&](int packed) __attribute__((always_inline)) -> LinTexel {
            const float* p = img + ((packed & 0x3fffffff) + ((packed & (1 << 30)) ? ox1 : ox0) + (packed < 0 ? oy1 : oy0));
            return LinTexel{p[0], p[1], p[2]};
        };
        // exchange buffer of this wave: [residual r of the quad][quad Q][tap q], rows padded by one texel (bank spread). It aliases the 16 reduction rows
        // this wave alone writes at the end of the kernel (QuadStream: row = tid >> 2), so it costs no LDS of its own
        LinTexel* xb = reinterpret_cast<LinTexel*>(smem + (tid >> 6) * 16 * kTopStride);       // 16-byte slots, 12 bytes used: ds_write_b96 / ds_read_b96 straight from / to the load's registers
        // one pixel's exchange: the four texels this lane loaded (tap q of residuals 0..3 of its quad) go out, the four taps of its own residual come back
        auto exchange = [&](const LinTexel& t0, const LinTexel& t1, const LinTexel& t2, const LinTexel& t3, float xk, float yk, float& oI, float& oX, float& oY) __attribute__((always_inline)) {
            lin_wave_sync();
            xb[0 * 65 + Q * 4 + q] = t0; xb[1 * 65 + Q * 4 + q] = t1; xb[2 * 65 + Q * 4 + q] = t2; xb[3 * 65 + Q * 4 + q] = t3;
            lin_wave_sync();
            const LinTexel p00 = xb[q * 65 + Q * 4], p10 = xb[q * 65 + Q * 4 + 1], p01 = xb[q * 65 + Q * 4 + 2], p11 = xb[q * 65 + Q * 4 + 3];
            const int ix = (int)xk, iy = (int)yk;                 // util/globalFuncs.h:75-89
            const float dx = xk - ix, dy = yk - iy, dxdy = dx * dy;
            const float w11 = dxdy, w01 = dy - dxdy, w10 = dx - dxdy, w00 = 1 - dx - dy + dxdy;
            oI = w11 * p11.x + w01 * p01.x + w10 * p10.x + w00 * p00.x;
            oX = w11 * p11.y + w01 * p01.y + w10 * p10.y + w00 * p00.y;
            oY = w11 * p11.z + w01 * p01.z + w10 * p10.z + w00 * p00.z;
        };
        // a batch = two pattern pixels: 8 sixteen-byte loads in flight per lane, then the two exchanges, then (per lane) their photometric part
        auto batch = [&](auto HALF) __attribute__((always_inline)) {
            constexpr int k0 = 2 * decltype(HALF)::value, k1 = k0 + 1;
            const int o0 = R.need ? pack(R.Kus[k0], R.Kvs[k0]) : 0, o1 = R.need ? pack(R.Kus[k1], R.Kvs[k1]) : 0;
            const LinTexel a0 = tap(lin_quad_bcast<0>(o0)), a1 = tap(lin_quad_bcast<1>(o0)), a2 = tap(lin_quad_bcast<2>(o0)), a3 = tap(lin_quad_bcast<3>(o0));
            const LinTexel b0_ = tap(lin_quad_bcast<0>(o1)), b1_ = tap(lin_quad_bcast<1>(o1)), b2_ = tap(lin_quad_bcast<2>(o1)), b3_ = tap(lin_quad_bcast<3>(o1));
            float h0I, h0X, h0Y, h1I, h1X, h1Y;
            exchange(a0, a1, a2, a3, R.Kus[k0], R.Kvs[k0], h0I, h0X, h0Y);
            exchange(b0_, b1_, b2_, b3_, R.Kus[k1], R.Kvs[k1], h1I, h1X, h1Y);
            if (R.need) { lin_pixel<MODE>(R, w.pc, h0I, h0X, h0Y, color[k0], wgt[k0], fixA, fixB); lin_pixel<MODE>(R, w.pc, h1I, h1X, h1Y, color[k1], wgt[k1], fixA, fixB); }
        };
#if NALO_LIN_COOP_NPB == 3
        // 3 + 3 + 2 pattern pixels: 12 loads in flight per lane
        auto batch3 = [&](auto FIRST) __attribute__((always_inline)) {
            constexpr int k0 = decltype(FIRST)::value, k1 = k0 + 1, k2 = k0 + 2;
            const int o0 = R.need ? pack(R.Kus[k0], R.Kvs[k0]) : 0, o1 = R.need ? pack(R.Kus[k1], R.Kvs[k1]) : 0, o2 = R.need ? pack(R.Kus[k2], R.Kvs[k2]) : 0;
            const LinTexel a0 = tap(lin_quad_bcast<0>(o0)), a1 = tap(lin_quad_bcast<1>(o0)), a2 = tap(lin_quad_bcast<2>(o0)), a3 = tap(lin_quad_bcast<3>(o0));
            const LinTexel b0_ = tap(lin_quad_bcast<0>(o1)), b1_ = tap(lin_quad_bcast<1>(o1)), b2_ = tap(lin_quad_bcast<2>(o1)), b3_ = tap(lin_quad_bcast<3>(o1));
            const LinTexel c0_ = tap(lin_quad_bcast<0>(o2)), c1_ = tap(lin_quad_bcast<1>(o2)), c2_ = tap(lin_quad_bcast<2>(o2)), c3_ = tap(lin_quad_bcast<3>(o2));
            float h0I, h0X, h0Y, h1I, h1X, h1Y, h2I, h2X, h2Y;
            exchange(a0, a1, a2, a3, R.Kus[k0], R.Kvs[k0], h0I, h0X, h0Y);
            exchange(b0_, b1_, b2_, b3_, R.Kus[k1], R.Kvs[k1], h1I, h1X, h1Y);
            exchange(c0_, c1_, c2_, c3_, R.Kus[k2], R.Kvs[k2], h2I, h2X, h2Y);
            if (R.need) { lin_pixel<MODE>(R, w.pc, h0I, h0X, h0Y, color[k0], wgt[k0], fixA, fixB); lin_pixel<MODE>(R, w.pc, h1I, h1X, h1Y, color[k1], wgt[k1], fixA, fixB); lin_pixel<MODE>(R, w.pc, h2I, h2X, h2Y, color[k2], wgt[k2], fixA, fixB); }
        };
        batch3(std::integral_constant<int, 0>{}); batch3(std::integral_constant<int, 3>{}); batch(std::integral_constant<int, 3>{});
#else
        batch(std::integral_constant<int, 0>{}); batch(std::integral_constant<int, 1>{}); batch(std::integral_constant<int, 2>{}); batch(std::integral_constant<int, 3>{});
#endif
    }
    lin_commit<MODE, FIX>(B, w, R);
    if (FIX == 2) {
        // not applied: only the energy sum (stats[0] of linearizeAll_Reductor, a double sum of the float returns) leaves the workgroup
        double e = (double)R.energy;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) e += __shfl_down(e, o);
        double* out = B.noapply_E + ((size_t)w.b * (kBlk / WG) + w.q) * B.W + w.t;
        if (WG > 64) {
            __syncthreads();                                                                       // the exchange buffer is dead
            double* sd = reinterpret_cast<double*>(smem);
            if ((tid & 63) == 0) sd[tid >> 6] = e;
            __syncthreads();
            if (tid == 0) { double s = 0; for (int i = 0; i < WG / 64; ++i) s += sd[i]; *out = s; }
        } else if (tid == 0) *out = e;
        return;
    }
    // ---- the 93 reduced values: quad DPP adds -> LDS rows (one per quad) -> fp64 column sums -> this workgroup's partial
    lin_wave_sync();                                                                               // the rows alias the exchange buffer this wave has just read
    lin_stream(B, w, R, smem, tid);
    if (WG > 64) __syncthreads(); else lin_wave_sync();
    double* out = B.top_partial + (((size_t)w.b * (kBlk / WG) + w.q) * B.W + w.t) * kTopStride;
    constexpr int ROWS = WG / 4;
#pragma unroll
    for (int j0 = 0; j0 < kTopVals; j0 += WG) {
        const int j = j0 + tid;
        if (j < kTopVals) {
            double s = 0.0;
#pragma unroll 8
            for (int r = 0; r < ROWS; ++r) s += (double)smem[r * kTopStride + j];                  // all lanes read one row: conflict free
            out[j] = s;
        }
    }
}

void ba_launch_linearize(hipStream_t s, const BADev& B, int mode, int fix, hipEvent_t ev_start, hipEvent_t ev_stop) {
    const int sub = B.lin_sub;                                       // workgroups (= partials) per point block: 1 (256 threads) or 4 (one wave each)
    const unsigned grid = 8u * (unsigned)sub * (unsigned)B.xcd_len * (unsigned)B.W;
#define NALO_LIN_LAUNCH(MODE_, FIX_, WG_)                                                                                                          \
    do {                                                                                                                                           \
        if (ev_start || ev_stop) hipExtLaunchKernelGGL((ba_linearize_kernel<MODE_, FIX_, WG_>), dim3(grid), dim3(WG_), 0, s, ev_start, ev_stop, 0, B);         \
        else ba_linearize_kernel<MODE_, FIX_, WG_><<<grid, WG_, 0, s>>>(B);                                                                        \
    } while (0)
    // fix: 0 = linearizeAll(false) + applyRes, 1 = linearizeAll(true), 2 = linearizeAll(false) without applyRes (energy only)
    if (sub == 4) {
        if (mode == 2) NALO_LIN_LAUNCH(2, 0, 64); else if (fix == 1) NALO_LIN_LAUNCH(0, 1, 64); else if (fix == 2) NALO_LIN_LAUNCH(0, 2, 64); else NALO_LIN_LAUNCH(0, 0, 64);
    } else {
        if (mode == 2) NALO_LIN_LAUNCH(2, 0, 256); else if (fix == 1) NALO_LIN_LAUNCH(0, 1, 256); else if (fix == 2) NALO_LIN_LAUNCH(0, 2, 256); else NALO_LIN_LAUNCH(0, 0, 256);
    }
#undef NALO_LIN_LAUNCH
}

}  // namespace nalo
