// a1 — FrameHessian::makeImages (reference src/FullSystem/HessianBlocks.cpp:127-190) on gfx950.
// Level l>0: 2x2 box mean of level l-1 (:158-165). Every level: central differences over the FLAT pixel index
// in [w, w(h-1)) (:168-181) — the x=0 / x=w-1 columns difference across row ends exactly as the reference
// does; rows 0 and h-1 (uninitialised heap in the reference) are zero here.
// Output texel = float4 {I, dx, dy, 0}: one aligned 16-byte load per bilinear tap in the gather kernels.
// Bandwidth-bound streaming kernels: 16 B/lane stores, 256-thread blocks, grid-stride.
#include "nalo_internal.h"
#include <hip/hip_ext.h>

namespace nalo {

__global__ __launch_bounds__(256) void pyr_down_kernel(const float* __restrict__ src, float* __restrict__ dst, int wl, int hl, int wlm1) {
    const int n = wl * hl;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int y = i / wl, x = i - y * wl;
        const float2 a = *reinterpret_cast<const float2*>(src + 2 * x + 2 * y * wlm1);
        const float2 b = *reinterpret_cast<const float2*>(src + 2 * x + 2 * y * wlm1 + wlm1);
        // reference order: 0.25f * (p00 + p10 + p01 + p11)
        dst[i] = 0.25f * (((a.x + a.y) + b.x) + b.y);
    }
}

__global__ __launch_bounds__(256) void pyr_grad_kernel(const float* __restrict__ I, float4* __restrict__ dI, float* __restrict__ absg,
                                                       const float* __restrict__ gammaB, int wl, int hl) {
    const int n = wl * hl;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += gridDim.x * blockDim.x) {
        const float c = I[idx];
        float dx = 0.f, dy = 0.f, ab = 0.f;
        if (idx >= wl && idx < wl * (hl - 1)) {
            dx = 0.5f * (I[idx + 1] - I[idx - 1]);
            dy = 0.5f * (I[idx + wl] - I[idx - wl]);
            if (!isfinite(dx)) dx = 0.f;
            if (!isfinite(dy)) dy = 0.f;
            ab = dx * dx + dy * dy;   // this file is built with -ffp-contract=off: rounds as the reference's scalar code
            if (gammaB) {                                   // HessianBlocks.h:400-406 getBGradOnly
                int ci = (int)(c + 0.5f);
                ci = ci < 5 ? 5 : (ci > 250 ? 250 : ci);
                const float gw = gammaB[ci + 1] - gammaB[ci];
                ab *= gw * gw;
            }
        }
        dI[idx] = make_float4(c, dx, dy, 0.f);
        absg[idx] = ab;
    }
}

// All levels' gradients in one launch (the level-by-level path of pyramids with an odd parent level; round 3: even pyramids take pyr_one_pass_kernel below).
struct PyrLevels {
    float* I[NALO_MAX_LEVELS]; float4* dI[NALO_MAX_LEVELS]; float* absg[NALO_MAX_LEVELS];
    int wl[NALO_MAX_LEVELS], hl[NALO_MAX_LEVELS], blk0[NALO_MAX_LEVELS + 1], L;
};
__global__ __launch_bounds__(256) void pyr_grad_all_kernel(PyrLevels P, const float* __restrict__ gammaB) {
    int l = 0;
    while (l + 1 < P.L && (int)blockIdx.x >= P.blk0[l + 1]) ++l;
    const int wl = P.wl[l], hl = P.hl[l], n = wl * hl;
    const int idx = (blockIdx.x - P.blk0[l]) * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const float* __restrict__ I = P.I[l];
    const float c = I[idx];
    float dx = 0.f, dy = 0.f, ab = 0.f;
    if (idx >= wl && idx < wl * (hl - 1)) {
        dx = 0.5f * (I[idx + 1] - I[idx - 1]);
        dy = 0.5f * (I[idx + wl] - I[idx - wl]);
        if (!isfinite(dx)) dx = 0.f;
        if (!isfinite(dy)) dy = 0.f;
        ab = dx * dx + dy * dy;   // this file is built with -ffp-contract=off: rounds as the reference's scalar code
        if (gammaB) {                                   // HessianBlocks.h:400-406 getBGradOnly
            int ci = (int)(c + 0.5f);
            ci = ci < 5 ? 5 : (ci > 250 ? 250 : ci);
            const float gw = gammaB[ci + 1] - gammaB[ci];
            ab *= gw * gw;
        }
    }
    P.dI[l][idx] = make_float4(c, dx, dy, 0.f);
    P.absg[l][idx] = ab;
}

// ---- round 3: ONE pass over level 0 for the whole pyramid. pyr_down_all + pyr_grad_all read level 0 twice and every upper level twice more
// (37 B of traffic per level-0 pixel against 25.3 algorithmic: 0.30 of the HBM roofline at 1920x1072). Here a workgroup owns a 64x16 tile (round 3, late: was 64x32) of level 0 and
// stages it in LDS WITH a halo of H = 2^(NL-1) pixels, walks the box pyramid up inside LDS (the same 0.25f * (((p00 + p10) + p01) + p11) nesting: the values
// of the level-by-level loop bit for bit) and writes {I, dx, dy} + absSquaredGrad of the levels 0 .. NL-1 from there; the halo is what the central differences of
// level l need of the neighbouring tiles (one pixel at level l = 2^l at level 0). The levels >= NL only get their planar box values here (the tile still
// holds whole pixels of them: pyramid_build checks) and their gradients in one small second launch (pyr_grad_all over those levels: 1/64 of the pixels). NL = all levels for
// pyramids of <= 4 levels (a KITTI frame: ONE launch), 3 otherwise (read amplification (72 x 24) / (64 x 16) = 1.7 on 4 of ~30 bytes per pixel).
// The reference's gradient runs over the FLAT index (HessianBlocks.cpp:168-181): column 0 differences against the last pixel of the row above, column w-1
// against the first pixel of the row below. Those two neighbours lie in no tile halo: the edge lanes rebuild them from level 0 (pyr_box_at, same nesting).
template <int L>
__device__ __forceinline__ float pyr_box_at(const float* __restrict__ I0, int w0, int x, int y) {
    if constexpr (L == 0) return I0[x + y * w0];
    else {
        const float a = pyr_box_at<L - 1>(I0, w0, 2 * x, 2 * y), b = pyr_box_at<L - 1>(I0, w0, 2 * x + 1, 2 * y);
        const float c = pyr_box_at<L - 1>(I0, w0, 2 * x, 2 * y + 1), d = pyr_box_at<L - 1>(I0, w0, 2 * x + 1, 2 * y + 1);
        return 0.25f * (((a + b) + c) + d);
    }
}
#ifndef NALO_PYR_TW
#define NALO_PYR_TW 64      // tile of level 0 per workgroup. Same box, 1920x1072 (5 levels) / 1224x368 (4 levels, one launch): 64x32 17.9 / 8.3 us, 32x32 19.0 / 7.2,
#define NALO_PYR_TH 16      // **64x16 17.7 / 6.6**, 32x16 21.2 / 7.1, 128x16 18.2 / 8.4: twice the workgroups of 64x32 for the KITTI frame's 240 tiles (< 256 CUs)
#endif
#ifndef NALO_PYR_NT
#define NALO_PYR_NT 1      // nontemporal stores of the texels: 14.6 -> 12.9 us at 1920x1072 (the pyramid is written once, read by later kernels)
#endif
constexpr int kPyrTW = NALO_PYR_TW, kPyrTH = NALO_PYR_TH;
typedef float pyr_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void pyr_store(float4* p, float c, float dx, float dy) {
#if NALO_PYR_NT
    pyr_f4 t; t.x = c; t.y = dx; t.z = dy; t.w = 0.f; __builtin_nontemporal_store(t, reinterpret_cast<pyr_f4*>(p));
#else
    *p = make_float4(c, dx, dy, 0.f);
#endif
}
__device__ __forceinline__ void pyr_store(float* p, float v) {
#if NALO_PYR_NT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
template <int NL> struct PyrLds {                              // LDS layout of the staged levels 0 .. NL-1 (with halo) and of the halo-free levels above
    static constexpr int H = 1 << (NL - 1);
    static constexpr int rw(int l) { return (kPyrTW >> l) + 2 * (H >> l); }
    static constexpr int rh(int l) { return (kPyrTH >> l) + 2 * (H >> l); }
    static constexpr int off(int l) { int o = 0; for (int k = 0; k < l; ++k) o += k < NL ? rw(k) * rh(k) : (kPyrTW >> k) * (kPyrTH >> k); return o; }
    static constexpr int total = off(NALO_MAX_LEVELS);
};
// flat-index neighbours of the image's first / last column, per staged level and tile row: edge[2 * off(l) + ty] = I_l(w_l - 1, y - 1) (left neighbour of
// x = 0), edge[2 * off(l) + (TH >> l) + ty] = I_l(0, y + 1) (right neighbour of x = w_l - 1); rebuilt from level 0 by a few lanes DURING the tile load (their
// dependent global loads - 4^l per value - would otherwise sit on the critical path of every border tile, i.e. of the whole launch)
__device__ constexpr int kPyrEdgeOff[5] = {0, kPyrTH, kPyrTH + (kPyrTH >> 1), kPyrTH + (kPyrTH >> 1) + (kPyrTH >> 2), kPyrTH + (kPyrTH >> 1) + (kPyrTH >> 2) + (kPyrTH >> 3)};
template <int LV>
__device__ __forceinline__ void pyr_edge_level(const PyrLevels& P, float* __restrict__ edge, int x0, int y0, int k) {     // k = lane of the wave that owns level LV
    constexpr int TH = kPyrTH >> LV, TW = kPyrTW >> LV;
    if (LV >= P.L || k >= 2 * TH) return;
    const int e = 2 * kPyrEdgeOff[LV] + k, right = k >= TH, ty = right ? k - TH : k;
    const int wl = P.wl[LV], hl = P.hl[LV], bx = x0 >> LV, y = (y0 >> LV) + ty;
    float v = 0.f;
    if (y >= 1 && y <= hl - 2) {
        if (!right && bx == 0) v = pyr_box_at<LV>(P.I[0], P.wl[0], wl - 1, y - 1);
        if (right && bx <= wl - 1 && wl - 1 < bx + TW) v = pyr_box_at<LV>(P.I[0], P.wl[0], 0, y + 1);
    }
    edge[e] = v;
}
template <int NL, int LV>
__device__ __forceinline__ void pyr_emit_level(const PyrLevels& P, const float* __restrict__ lds, const float* __restrict__ edge, const float* __restrict__ gammaB, int x0, int y0, int tid) {
    using Lay = PyrLds<NL>;
    if (LV >= P.L) return;
    constexpr int RW = Lay::rw(LV), HL = Lay::H >> LV, TW = kPyrTW >> LV, TH = kPyrTH >> LV;
    const float* __restrict__ S = lds + Lay::off(LV);
    const int wl = P.wl[LV], hl = P.hl[LV], bx = x0 >> LV, by = y0 >> LV;
    for (int i = tid; i < TW * TH; i += 256) {
        const int tx = i % TW, ty = i / TW, x = bx + tx, y = by + ty;
        if (x >= wl || y >= hl) continue;
        const float* sp = S + (ty + HL) * RW + tx + HL;
        const float c = sp[0];
        float dx = 0.f, dy = 0.f, ab = 0.f;
        if (y >= 1 && y <= hl - 2) {                            // idx >= w && idx < w (h - 1)
            const float left = x > 0 ? sp[-1] : edge[2 * kPyrEdgeOff[LV] + ty];
            const float right = x < wl - 1 ? sp[1] : edge[2 * kPyrEdgeOff[LV] + TH + ty];
            dx = 0.5f * (right - left);
            dy = 0.5f * (sp[RW] - sp[-RW]);
            if (!isfinite(dx)) dx = 0.f;
            if (!isfinite(dy)) dy = 0.f;
            ab = dx * dx + dy * dy;                             // -ffp-contract=off: rounds as the reference's scalar code
            if (gammaB) {                                       // HessianBlocks.h:400-406 getBGradOnly
                int ci = (int)(c + 0.5f);
                ci = ci < 5 ? 5 : (ci > 250 ? 250 : ci);
                const float gw = gammaB[ci + 1] - gammaB[ci];
                ab *= gw * gw;
            }
        }
        const int idx = x + y * wl;
        pyr_store(&P.dI[LV][idx], c, dx, dy);
        pyr_store(&P.absg[LV][idx], ab);
    }
}
// ---- round 4: the levels 3 and 4 of a five-level pyramid in the SAME launch. They used to take a second launch over planar box values the fine tiles
// left behind (pyr_grad_tail_kernel: 3.9 us + a kernel boundary behind a 12 us kernel, for 1/64 + 1/256 of the pixels). Now the FIRST workgroups of the one launch
// rebuild them from level 0 on their own: a coarse tile is kCW x kCH pixels of level 3 (= 8 kCW x 8 kCH of level 0) with a halo of two level-3 pixels (one pixel of
// level 4); a lane folds a 4 x 4 block of level 0 (four 16-byte loads) into one value of level 2 - the same 0.25f * (((a + b) + c) + d) nesting, twice -, the levels
// 3 and 4 follow in LDS. The flat-index neighbours of the border columns (I_l(w_l - 1, y - 1) left of x = 0, I_l(0, y + 1) right of x = w_l - 1) are rebuilt the
// same way by the border tiles: 4 (level 3) or 16 (level 4) extra level-2 values each. Read amplification (kCW + 4)(kCH + 4) / (kCW kCH) on 4 of ~25 bytes per pixel.
#ifndef NALO_PYR_CW
#define NALO_PYR_CW 16
#define NALO_PYR_CH 4
#endif
struct PyrCoarse {
    static constexpr int kCW = NALO_PYR_CW, kCH = NALO_PYR_CH;              // tile of level 3 (level 4: kCW / 2 x kCH / 2)
    static constexpr int R3W = kCW + 4, R3H = kCH + 4, R2W = 2 * R3W, R2H = 2 * R3H, R4W = kCW / 2 + 2, R4H = kCH / 2 + 2;
    static constexpr int N2 = R2W * R2H, N3 = R3W * R3H, N4 = R4W * R4H;
    static constexpr int NX3 = 4 * kCH, NX4 = 16 * (kCH / 2);              // extra level-2 values per border side: level 3 / level 4 edge values
    static constexpr int NX = 2 * (NX3 + NX4);
    static_assert(N3 <= 192 && 2 * kCH <= 32 && kCH <= 32 && NX <= 256 && kCW % 2 == 0 && kCH % 2 == 0, "lane ranges of pyr_coarse_tile");
    static constexpr int oL2 = 0, oX2 = N2, oL3 = oX2 + NX, oL4 = oL3 + N3, oE3 = oL4 + N4, oE4 = oE3 + 2 * kCH, total = oE4 + kCH;
};
__device__ __forceinline__ float pyr_box4(float a, float b, float c, float d) { return 0.25f * (((a + b) + c) + d); }
// level-2 value (x2, y2) from its 4 x 4 block of level 0: four aligned 16-byte loads
__device__ __forceinline__ void pyr_l2_load(const float* __restrict__ I0, int w0, int x2, int y2, bool ok, pyr_f4 (&r)[4]) {
    const float* p = I0 + (ok ? (size_t)(4 * y2) * w0 + 4 * x2 : 0);
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = *reinterpret_cast<const pyr_f4*>(p + (ok ? (size_t)k * w0 : 0));
}
__device__ __forceinline__ float pyr_l2_fold(const pyr_f4 (&r)[4]) {
    const float a = pyr_box4(r[0].x, r[0].y, r[1].x, r[1].y), b = pyr_box4(r[0].z, r[0].w, r[1].z, r[1].w);
    const float c = pyr_box4(r[2].x, r[2].y, r[3].x, r[3].y), d = pyr_box4(r[2].z, r[2].w, r[3].z, r[3].w);
    return pyr_box4(a, b, c, d);
}
__device__ __forceinline__ void pyr_grad_store(const PyrLevels& P, int lv, int x, int y, float c, float left, float right, float up, float down, bool inner, const float* __restrict__ gammaB) {
    float dx = 0.f, dy = 0.f, ab = 0.f;
    if (inner) {                                                    // idx >= w && idx < w (h - 1)
        dx = 0.5f * (right - left);
        dy = 0.5f * (down - up);
        if (!isfinite(dx)) dx = 0.f;
        if (!isfinite(dy)) dy = 0.f;
        ab = dx * dx + dy * dy;                                     // -ffp-contract=off: rounds as the reference's scalar code
        if (gammaB) {                                               // HessianBlocks.h:400-406 getBGradOnly
            int ci = (int)(c + 0.5f);
            ci = ci < 5 ? 5 : (ci > 250 ? 250 : ci);
            const float gw = gammaB[ci + 1] - gammaB[ci];
            ab *= gw * gw;
        }
    }
    const int idx = x + y * P.wl[lv];
    pyr_store(&P.dI[lv][idx], c, dx, dy);
    pyr_store(&P.absg[lv][idx], ab);
}
__device__ __forceinline__ void pyr_coarse_tile(const PyrLevels& P, const float* __restrict__ gammaB, int tile, float* __restrict__ lds, int tid) {
    using C = PyrCoarse;
    const int w0 = P.wl[0], w2 = P.wl[2], h2 = P.hl[2], w3 = P.wl[3], h3 = P.hl[3], w4 = P.wl[4], h4 = P.hl[4];
    const int ctx = (w3 + C::kCW - 1) / C::kCW, X3 = (tile % ctx) * C::kCW, Y3 = (tile / ctx) * C::kCH, X4 = X3 >> 1, Y4 = Y3 >> 1;
    const bool left = X3 == 0, right = X3 <= w3 - 1 && w3 - 1 < X3 + C::kCW;
    const float* __restrict__ I0 = P.I[0];
    // ---- level 2 of the haloed region (and of the border columns' flat-index neighbours), straight from level 0
    constexpr int R = (C::N2 + 255) / 256;
    pyr_f4 r[R][4]; bool ok[R];
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int e = tid + 256 * k, x2 = 2 * X3 - 4 + e % C::R2W, y2 = 2 * Y3 - 4 + e / C::R2W;
        ok[k] = e < C::N2 && x2 >= 0 && x2 < w2 && y2 >= 0 && y2 < h2;
        pyr_l2_load(I0, w0, x2, y2, ok[k], r[k]);
    }
    pyr_f4 rx[4]; bool okx = false;
    if ((left || right) && tid < C::NX) {                       // border tiles: extra values, [left: level 3 | level 4][right: level 3 | level 4]
        int k = tid;
        const bool rs = k >= C::NX3 + C::NX4;                   // right side: the neighbour of x = w_l - 1 is I_l(0, y + 1); left side: I_l(w_l - 1, y - 1) for x = 0
        if (rs) k -= C::NX3 + C::NX4;
        int x2, y2;
        if (k < C::NX3) {                                       // level 3: entry = ty * 4 + (sy * 2 + sx)
            const int ty = k >> 2, q2 = k & 3, y = Y3 + ty, x3 = rs ? 0 : w3 - 1, y3 = rs ? y + 1 : y - 1;
            okx = (rs ? right : left) && y >= 1 && y <= h3 - 2;
            x2 = 2 * x3 + (q2 & 1); y2 = 2 * y3 + (q2 >> 1);
        } else {                                                // level 4: entry = ty * 16 + (level-3 sub-pixel) * 4 + (level-2 sub-pixel)
            k -= C::NX3;
            const int ty = k >> 4, q3 = (k >> 2) & 3, q2 = k & 3, y = Y4 + ty, x4 = rs ? 0 : w4 - 1, y4 = rs ? y + 1 : y - 1;
            okx = (rs ? right : left) && y >= 1 && y <= h4 - 2;
            const int x3 = 2 * x4 + (q3 & 1), y3 = 2 * y4 + (q3 >> 1);
            x2 = 2 * x3 + (q2 & 1); y2 = 2 * y3 + (q2 >> 1);
        }
        pyr_l2_load(I0, w0, x2, y2, okx, rx);
    }
#pragma unroll
    for (int k = 0; k < R; ++k) { const int e = tid + 256 * k; if (e < C::N2) lds[C::oL2 + e] = ok[k] ? pyr_l2_fold(r[k]) : 0.f; }
    if (tid < C::NX) lds[C::oX2 + tid] = okx ? pyr_l2_fold(rx) : 0.f;
    __syncthreads();
    // ---- level 3 (region), and the edge values of both levels from the extra level-2 values
    if (tid < C::N3) {
        const float* q = lds + C::oL2 + 2 * (tid / C::R3W) * C::R2W + 2 * (tid % C::R3W);
        lds[C::oL3 + tid] = pyr_box4(q[0], q[1], q[C::R2W], q[C::R2W + 1]);
    } else if (tid >= 192 && tid < 192 + 2 * C::kCH) {          // level-3 edge values: [left kCH | right kCH]
        const int k = tid - 192, side = k >= C::kCH, ty = side ? k - C::kCH : k;
        const float* q = lds + C::oX2 + side * (C::NX3 + C::NX4) + ty * 4;
        lds[C::oE3 + k] = pyr_box4(q[0], q[1], q[2], q[3]);
    } else if (tid >= 224 && tid < 224 + C::kCH) {              // level-4 edge values: [left kCH / 2 | right kCH / 2]
        const int k = tid - 224, side = k >= C::kCH / 2, ty = side ? k - C::kCH / 2 : k;
        const float* q = lds + C::oX2 + side * (C::NX3 + C::NX4) + C::NX3 + ty * 16;
        lds[C::oE4 + k] = pyr_box4(pyr_box4(q[0], q[1], q[2], q[3]), pyr_box4(q[4], q[5], q[6], q[7]), pyr_box4(q[8], q[9], q[10], q[11]), pyr_box4(q[12], q[13], q[14], q[15]));
    }
    __syncthreads();
    if (tid < C::N4) {                                          // ---- level 4 (region)
        const float* q = lds + C::oL3 + 2 * (tid / C::R4W) * C::R3W + 2 * (tid % C::R4W);
        lds[C::oL4 + tid] = pyr_box4(q[0], q[1], q[C::R3W], q[C::R3W + 1]);
    }
    __syncthreads();
    // ---- texels of both levels
    for (int t = tid; t < C::kCW * C::kCH; t += 256) {
        const int tx = t % C::kCW, ty = t / C::kCW, x = X3 + tx, y = Y3 + ty;
        if (x < w3 && y < h3) {
            const float* sp = lds + C::oL3 + (ty + 2) * C::R3W + tx + 2;
            pyr_grad_store(P, 3, x, y, sp[0], x > 0 ? sp[-1] : lds[C::oE3 + ty], x < w3 - 1 ? sp[1] : lds[C::oE3 + C::kCH + ty], sp[-C::R3W], sp[C::R3W], y >= 1 && y <= h3 - 2, gammaB);
        }
    }
    for (int t = tid; t < (C::kCW / 2) * (C::kCH / 2); t += 256) {
        const int tx = t % (C::kCW / 2), ty = t / (C::kCW / 2), x = X4 + tx, y = Y4 + ty;
        if (x < w4 && y < h4) {
            const float* sp = lds + C::oL4 + (ty + 1) * C::R4W + tx + 1;
            pyr_grad_store(P, 4, x, y, sp[0], x > 0 ? sp[-1] : lds[C::oE4 + ty], x < w4 - 1 ? sp[1] : lds[C::oE4 + C::kCH / 2 + ty], sp[-C::R4W], sp[C::R4W], y >= 1 && y <= h4 - 2, gammaB);
        }
    }
}
template <int NL>
__device__ __forceinline__ void pyr_box_level(const PyrLevels& P, float* __restrict__ lds, int l, int x0, int y0, int tid) {
    using Lay = PyrLds<NL>;
    constexpr int H = Lay::H;
    const bool halo = l < NL, phalo = l - 1 < NL;
    const int RW = halo ? Lay::rw(l) : (kPyrTW >> l), RH = halo ? Lay::rh(l) : (kPyrTH >> l);
    const int PW = phalo ? Lay::rw(l - 1) : (kPyrTW >> (l - 1));
    // a halo-free level above a haloed parent (l == NL) starts at the parent's interior
    const int po = (!halo && phalo) ? (H >> (l - 1)) * PW + (H >> (l - 1)) : 0;
    const float* __restrict__ src = lds + Lay::off(l - 1) + po;
    float* __restrict__ dst = lds + Lay::off(l);
    for (int i = tid; i < RW * RH; i += 256) {
        const int rx = i % RW, ry = i / RW;
        const float* q = src + 2 * ry * PW + 2 * rx;
        const float v = 0.25f * (((q[0] + q[1]) + q[PW]) + q[PW + 1]);
        dst[i] = v;
        if (!halo) {                                    // planar box value of an upper level: input of the second launch
            const int x = (x0 >> l) + rx, y = (y0 >> l) + ry;
            if (x < P.wl[l] && y < P.hl[l]) P.I[l][x + y * P.wl[l]] = v;
        }
    }
}
#ifndef NALO_PYR_EARLY
#define NALO_PYR_EARLY 1
#endif
#ifndef NALO_PYR_LD4
#define NALO_PYR_LD4 1
#endif
#ifndef NALO_PYR_COARSE
#define NALO_PYR_COARSE 1
#endif
template <int NL, int LV>
__device__ __forceinline__ void pyr_walk(const PyrLevels& P, float* __restrict__ lds, const float* __restrict__ edge, const float* __restrict__ gammaB, int x0, int y0, int tid, int Lbox) {
    if constexpr (LV <= NALO_MAX_LEVELS) {                      // step LV: emit level LV - 1 (complete since the last barrier), box level LV out of it
        if constexpr (LV - 1 < NL) pyr_emit_level<NL, LV - 1>(P, lds, edge, gammaB, x0, y0, tid);
        if constexpr (LV < NALO_MAX_LEVELS) {
            if (LV < Lbox) pyr_box_level<NL>(P, lds, LV, x0, y0, tid);
            __syncthreads();
            pyr_walk<NL, LV + 1>(P, lds, edge, gammaB, x0, y0, tid, Lbox);
        }
    }
}
template <int NL>
__global__ __launch_bounds__(256) void pyr_one_pass_kernel(PyrLevels P, const float* __restrict__ gammaB, int n_coarse) {
    using Lay = PyrLds<NL>;
    __shared__ __attribute__((aligned(16))) float lds[Lay::total];
    __shared__ float edge[4 * kPyrTH];                          // 2 x (TH + TH/2 + TH/4 + TH/8) <= 4 TH
    constexpr int H = Lay::H;
    const int tid = threadIdx.x, tiles_x = (P.wl[0] + kPyrTW - 1) / kPyrTW;
    if constexpr (NL == 3) {                                    // a five-level pyramid: the first n_coarse workgroups build the levels 3 and 4 (pyr_coarse_tile)
        static_assert(Lay::total >= PyrCoarse::total, "the coarse tile's LDS image fits the fine tile's");
        if ((int)blockIdx.x < n_coarse) { pyr_coarse_tile(P, gammaB, (int)blockIdx.x, lds, tid); return; }
    }
    const int bid = (int)blockIdx.x - n_coarse, Lbox = n_coarse > 0 ? NL : P.L;     // with coarse workgroups the fine tiles stop at level NL - 1 (no planar values of the levels above)
    const int x0 = (bid % tiles_x) * kPyrTW, y0 = (bid / tiles_x) * kPyrTH;
    {                                                           // level 0 with its halo; outside the image: 0 (never used by a pixel that is written)
        constexpr int RW = Lay::rw(0), RH = Lay::rh(0);
        const int w0 = P.wl[0], h0 = P.hl[0];
        const float* __restrict__ I0 = P.I[0];
        // all loads of the tile first (registers), then the LDS stores: a rolled loop waits for every load before its own LDS store, i.e. a dozen
        // dependent HBM round trips per tile (measured: 8 us for ONE tile of a KITTI frame)
#if NALO_PYR_LD4
        // round 4: the haloed tile starts H = 4 or 8 pixels left of a multiple of 64, so with w0 % 4 == 0 every aligned group of four pixels lies wholly inside or
        // wholly outside the image: 16-byte loads (432 / 640 per tile instead of 1728 / 2560 4-byte ones); any other width takes the scalar loop
        static_assert(RW % 4 == 0 && H % 4 == 0 || NL < 3, "haloed row in whole float4s");
        if ((NL >= 3) && (w0 & 3) == 0) {
            constexpr int RW4 = RW / 4, N4 = (RW4 * RH + 255) / 256;
            pyr_f4 v4[N4];
            bool in4[N4];
#pragma unroll
            for (int k = 0; k < N4; ++k) {
                const int i = tid + 256 * k, rx = (i % RW4) * 4, ry = i / RW4, gx = x0 - H + rx, gy = y0 - H + ry;
                in4[k] = i < RW4 * RH && gx >= 0 && gx < w0 && gy >= 0 && gy < h0;
                v4[k] = *reinterpret_cast<const pyr_f4*>(I0 + (in4[k] ? gx + gy * w0 : 0));
            }
#pragma unroll
            for (int k = 0; k < N4; ++k) { const int i = tid + 256 * k; if (i < RW4 * RH) { pyr_f4 t = v4[k]; if (!in4[k]) t = pyr_f4{0.f, 0.f, 0.f, 0.f}; *reinterpret_cast<pyr_f4*>(lds + 4 * i) = t; } }
        } else
#endif
        {
        constexpr int NLD = (RW * RH + 255) / 256;
        float v[NLD];
        bool inb[NLD];
#pragma unroll
        for (int k = 0; k < NLD; ++k) {                         // branch-free: an out-of-image lane loads pixel 0 and discards it (a load under a divergent
            const int i = tid + 256 * k, rx = i % RW, ry = i / RW, gx = x0 - H + rx, gy = y0 - H + ry;   // branch is waited for at the join: one round trip each)
            inb[k] = i < RW * RH && gx >= 0 && gx < w0 && gy >= 0 && gy < h0;
            v[k] = I0[inb[k] ? gx + gy * w0 : 0];
        }
#pragma unroll
        for (int k = 0; k < NLD; ++k) v[k] = inb[k] ? v[k] : 0.f;
#pragma unroll
        for (int k = 0; k < NLD; ++k) { const int i = tid + 256 * k; if (i < RW * RH) lds[i] = v[k]; }
        }
        // the flat-index neighbours of the border columns (only border tiles load anything): wave l owns level l, so the four levels' dependent loads run
        // side by side instead of as four masked branches of one wave
        static_assert(2 * kPyrTH <= 64, "one wave holds a level's edge values");
        switch (tid >> 6) {
            case 0: pyr_edge_level<0>(P, edge, x0, y0, tid & 63); break;
            case 1: if constexpr (NL > 1) pyr_edge_level<1>(P, edge, x0, y0, tid & 63); break;
            case 2: if constexpr (NL > 2) pyr_edge_level<2>(P, edge, x0, y0, tid & 63); break;
            default: if constexpr (NL > 3) pyr_edge_level<3>(P, edge, x0, y0, tid & 63); break;
        }

    }
    __syncthreads();
#if NALO_PYR_EARLY
    // round 4: level l leaves for HBM while level l + 1 is being boxed out of it (both only READ level l's LDS image): the texel stores drain under the rest of
    // the LDS walk instead of forming one store phase at the end of every (lockstep) workgroup
    pyr_walk<NL, 1>(P, lds, edge, gammaB, x0, y0, tid, Lbox);
#else
#pragma unroll
    for (int l = 1; l < NALO_MAX_LEVELS; ++l) {                 // the box pyramid inside LDS: levels < NL over the haloed region, the ones above over the tile only
        if (l < Lbox) pyr_box_level<NL>(P, lds, l, x0, y0, tid);
        __syncthreads();
    }
    pyr_emit_level<NL, 0>(P, lds, edge, gammaB, x0, y0, tid);
    if constexpr (NL > 1) pyr_emit_level<NL, 1>(P, lds, edge, gammaB, x0, y0, tid);
    if constexpr (NL > 2) pyr_emit_level<NL, 2>(P, lds, edge, gammaB, x0, y0, tid);
    if constexpr (NL > 3) pyr_emit_level<NL, 3>(P, lds, edge, gammaB, x0, y0, tid);
#endif
}
// gradients of the levels >= l0 from their planar box values (the second launch of a pyramid with more than four levels)
__global__ __launch_bounds__(256) void pyr_grad_tail_kernel(PyrLevels P, const float* __restrict__ gammaB, int l0) {
    int l = l0;
    const int b = blockIdx.x + P.blk0[l0];
    while (l + 1 < P.L && b >= P.blk0[l + 1]) ++l;
    const int wl = P.wl[l], hl = P.hl[l], n = wl * hl;
    const int idx = (b - P.blk0[l]) * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const float* __restrict__ I = P.I[l];
    const float c = I[idx];
    float dx = 0.f, dy = 0.f, ab = 0.f;
    if (idx >= wl && idx < wl * (hl - 1)) {
        dx = 0.5f * (I[idx + 1] - I[idx - 1]);
        dy = 0.5f * (I[idx + wl] - I[idx - wl]);
        if (!isfinite(dx)) dx = 0.f;
        if (!isfinite(dy)) dy = 0.f;
        ab = dx * dx + dy * dy;
        if (gammaB) {
            int ci = (int)(c + 0.5f);
            ci = ci < 5 ? 5 : (ci > 250 ? 250 : ci);
            const float gw = gammaB[ci + 1] - gammaB[ci];
            ab *= gw * gw;
        }
    }
    P.dI[l][idx] = make_float4(c, dx, dy, 0.f);
    P.absg[l][idx] = ab;
}

// Raw-frame ingest: PhotometricUndistorter::processFrame (reference src/util/Undistort.cpp:214-251) + the remap loop of Undistort::undistort (:435-530) +
// the INTER_NEAREST resizes of Undistort::undistort_mask (:385-433; IOWrapper/OpenCV/ImageRW_OpenCV.cpp:55-85) in ONE pass over the rectified image:
// the 8- or 16-bit sensor frame crosses PCIe as it is (1-2 B/px instead of the 4 B/px float image the host path uploads) and
//   data[p] = G[raw[p]] * vignetteMapInv[p]      (photometricCalibration 2; 1: no vignette; disabled: factor * raw[p])
// is evaluated at the (up to) four taps of every output pixel instead of being materialised: out = bilinear(data, remapX, remapY), 0 where remapX < 0
// (passthrough: out = data). Each product and the weighted sum are rounded as the reference's float code does (this file: -ffp-contract=off).
struct IngestParams {
    const void* raw; int bpp;                       // 1 or 2 bytes per pixel
    int wOrg, hOrg, w, h;
    const float *G, *vinv; const float2* remapXY;   // G [GDepth]; vinv [wOrg*hOrg] or null; remap [w*h] {x, y} interleaved, or null (passthrough)
    int photometric;                                // 0: data = factor * raw, 1: G only, 2: G * vignetteMapInv
    float factor;
    const uint8_t *mask_org, *bgr_org;              // [wOrg*hOrg], [wOrg*hOrg*3] or null
    float* out_I; float* out_mask; uint8_t* out_bgr;
    double ifx, ify;                                // cv::resize: 1 / ((double)w / wOrg)
};
// Round 3: the pass is bound by the NUMBER of memory instructions per pixel (15 in the first version: two remap loads, four byte taps, four vignette
// taps, four response look-ups, one store - 0.19 of the HBM roofline), not by bytes. Now 6: the remap table is interleaved {x, y} (one 8-byte load, built by
// nalo_undist_set), the two taps of a row come with ONE load (2 or 4 raw bytes; 8 bytes of the vignette map), the 8-bit response lives in LDS, and nothing sits
// under a divergent branch (an outside pixel computes on pixel 0 and is zeroed at the end). The arithmetic is the reference's, operation for operation.
template <int BPP>
__device__ __forceinline__ void ingest_pair(const void* __restrict__ raw, int p, unsigned& a, unsigned& b) {      // raw[p], raw[p + 1]
    if constexpr (BPP == 1) { uint16_t v; __builtin_memcpy(&v, reinterpret_cast<const uint8_t*>(raw) + p, 2); a = v & 0xFFu; b = v >> 8; }
    else { uint32_t v; __builtin_memcpy(&v, reinterpret_cast<const uint16_t*>(raw) + p, 4); a = v & 0xFFFFu; b = v >> 16; }
}
// Round 4: a pixel is a chain of two dependent round trips (remap entry -> taps), and with one pixel per lane a 1920x1072 frame is four rounds of the
// resident grid: ~8 round trips end to end (15 us at 0.30 of the roofline). Now every lane owns NALO_INGEST_U pixels (stride 256 inside its workgroup's run of consecutive
// pixels: every instruction still covers one contiguous run per wave): the U remap loads leave together, then the 4 U taps, then the stores - the whole frame is resident at once.
#ifndef NALO_INGEST_U
#define NALO_INGEST_U 4
#endif
template <int BPP, int PHOTO>
__global__ __launch_bounds__(256) void ingest_kernel(IngestParams P) {
    constexpr int U = NALO_INGEST_U;
    __shared__ float sG[BPP == 1 && PHOTO > 0 ? 256 : 1];
    // the 8-bit response table goes to LDS BEHIND the issue of the pixel loads (its own load leaves first, the LDS store and the barrier follow the taps' issue): staged
    // in front of them it was a third round trip at the head of every workgroup
    float g_own = 0.f;
    if constexpr (BPP == 1 && PHOTO > 0) g_own = P.G[threadIdx.x];
    auto stage_G = [&]() { if constexpr (BPP == 1 && PHOTO > 0) { sG[threadIdx.x] = g_own; __syncthreads(); } };
    const float* __restrict__ Gt = (BPP == 1 && PHOTO > 0) ? sG : P.G;
    const int n = P.w * P.h, base = blockIdx.x * (256 * U) + threadIdx.x;
    auto data = [&](unsigned v, float vi) -> float {                       // PhotometricUndistorter::processFrame (Undistort.cpp:224-251) at one original pixel
        if constexpr (PHOTO == 0) return P.factor * (float)v;
        float d = Gt[v];
        if constexpr (PHOTO == 2) d *= vi;
        return d;
    };
    float o[U];
    if (!P.remapXY) {
        unsigned v[U]; float vi[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = base + 256 * u, ii = idx < n ? idx : 0;
            if constexpr (BPP == 1) v[u] = reinterpret_cast<const uint8_t*>(P.raw)[ii]; else v[u] = reinterpret_cast<const uint16_t*>(P.raw)[ii];
            vi[u] = PHOTO == 2 ? P.vinv[ii] : 1.f;
        }
        stage_G();
#pragma unroll
        for (int u = 0; u < U; ++u) o[u] = data(v[u], vi[u]);
    } else {
        float2 r[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int idx = base + 256 * u; r[u] = P.remapXY[idx < n ? idx : 0]; }
        unsigned v00[U], v10[U], v01[U], v11[U];
        float2 i0[U], i1[U];
        float xx[U], yy[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            xx[u] = r[u].x; yy[u] = r[u].y;
            if (xx[u] < 0) { xx[u] = 0.f; yy[u] = 0.f; }
            const int xxi = (int)xx[u], yyi = (int)yy[u];
            xx[u] -= xxi; yy[u] -= yyi;
            const int p = xxi + yyi * P.wOrg;
            ingest_pair<BPP>(P.raw, p, v00[u], v10[u]);
            ingest_pair<BPP>(P.raw, p + P.wOrg, v01[u], v11[u]);
            i0[u] = make_float2(1.f, 1.f); i1[u] = i0[u];
            if constexpr (PHOTO == 2) { __builtin_memcpy(&i0[u], P.vinv + p, 8); __builtin_memcpy(&i1[u], P.vinv + p + P.wOrg, 8); }
        }
        stage_G();
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float xxyy = xx[u] * yy[u];
            o[u] = xxyy * data(v11[u], i1[u].y) + (yy[u] - xxyy) * data(v01[u], i1[u].x) + (xx[u] - xxyy) * data(v10[u], i0[u].y) + (1 - xx[u] - yy[u] + xxyy) * data(v00[u], i0[u].x);
            if (r[u].x < 0) o[u] = 0.f;
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int idx = base + 256 * u;
        if (idx >= n) break;
        P.out_I[idx] = o[u];
        if (P.mask_org || P.bgr_org) {                     // cv::resize(.., INTER_NEAREST): sx = min(floor(x * ifx), wOrg - 1)
            const int y = idx / P.w, x = idx - y * P.w;
            int sx = (int)floor(x * P.ifx), sy = (int)floor(y * P.ify);
            sx = sx < P.wOrg - 1 ? sx : P.wOrg - 1; sy = sy < P.hOrg - 1 ? sy : P.hOrg - 1;
            const int sp = sx + sy * P.wOrg;
            if (P.mask_org) P.out_mask[idx] = (float)P.mask_org[sp] * 1.0f;
            if (P.bgr_org) { P.out_bgr[3 * idx] = P.bgr_org[3 * sp]; P.out_bgr[3 * idx + 1] = P.bgr_org[3 * sp + 1]; P.out_bgr[3 * idx + 2] = P.bgr_org[3 * sp + 2]; }
        }
    }
}
int ingest_launch(nalo_ctx* c, hipStream_t st, const void* raw, int bpp, int wOrg, int hOrg, const float* G, const float* vinv, const float2* remapXY, int photometric,
                  float factor, const uint8_t* mask_org, const uint8_t* bgr_org, float* out_I, float* out_mask, uint8_t* out_bgr) {
    IngestParams P;
    P.raw = raw; P.bpp = bpp; P.wOrg = wOrg; P.hOrg = hOrg; P.w = c->w; P.h = c->h; P.G = G; P.vinv = vinv; P.remapXY = remapXY; P.photometric = photometric;
    P.factor = factor; P.mask_org = mask_org; P.bgr_org = bgr_org; P.out_I = out_I; P.out_mask = out_mask; P.out_bgr = out_bgr;
    P.ifx = 1.0 / ((double)c->w / wOrg); P.ify = 1.0 / ((double)c->h / hOrg);
    const int n = c->w * c->h, grid = (n + 256 * NALO_INGEST_U - 1) / (256 * NALO_INGEST_U);   // NALO_INGEST_U pixels per lane, consecutive workgroups on consecutive memory
    ProfScope ps(c, "ingest", true);                                       // dispatch-attached timestamps
#define NALO_INGEST(B_, P_) do { if (ps.a) hipExtLaunchKernelGGL((ingest_kernel<B_, P_>), dim3(grid), dim3(256), 0, st, ps.a, ps.b, 0, P); else ingest_kernel<B_, P_><<<grid, 256, 0, st>>>(P); } while (0)
    if (bpp == 1) { if (photometric == 0) NALO_INGEST(1, 0); else if (photometric == 1) NALO_INGEST(1, 1); else NALO_INGEST(1, 2); }
    else { if (photometric == 0) NALO_INGEST(2, 0); else if (photometric == 1) NALO_INGEST(2, 1); else NALO_INGEST(2, 2); }
#undef NALO_INGEST
    NALO_HIP(c, hipGetLastError());
    return NALO_OK;
}

// Level 0 once more, for ba_linearize's gathers: 12-byte texels {I,dx,dy} (the 16-byte texel spends a quarter of every cache line on padding) in tiles of
// 5 wide x 2 high texels = 120 bytes, padded to one 128-byte cache line. A line then holds a 5x2 block of texels instead of 8 texels of one row: a 2x2 bilinear
// footprint lies in one line with probability 2/5 instead of never, the 6x6 footprint of a residual in ~7.0 lines instead of ~9.8, and the same lines hold 25 % more
// image. Pure gathers of the stress250k residual list (scripts/ubench/gather.hip, profiles/r02_ubench_gather_12B.log): row-major 16-B texels 174.6 us, 4x2 tiles
// of 16-B texels 153.3, row-major 12-B texels 151.6, **5x2 tiles of 12-B texels 131.5**.
// float index of texel (x, y): ((y >> 1) * wt + x / 5) * 32 + ((y & 1) * 5 + x % 5) * 3, wt = ceil(w / 5); any w, h (the last tile column / row is partly unused).
__global__ __launch_bounds__(256) void tile_level0_kernel(const float4* __restrict__ src, float* __restrict__ dst, int w, int h, int wt) {
    const int o = blockIdx.x * blockDim.x + threadIdx.x;                   // one source texel per lane, consecutive workgroups on consecutive memory
    if (o >= w * h) return;
    const int y = o / w, x = o - y * w, tx = x / 5;
    const float4 t = src[o];
    float* d = dst + ((size_t)(y >> 1) * wt + tx) * 32 + ((y & 1) * 5 + (x - tx * 5)) * 3;
    d[0] = t.x; d[1] = t.y; d[2] = t.z;
}
int frame_tile_level0(nalo_ctx* c, FrameSlot& s) {
    const int wt = (c->w + 4) / 5, ht = (c->h + 1) / 2;
    const size_t n = (size_t)c->w * c->h;
    if (!s.dI0t) NALO_HIP(c, hipMalloc((void**)&s.dI0t, (size_t)wt * ht * 128 + 16));         // + 16: the 12-byte load of a tile's last texel may be issued as 16
    tile_level0_kernel<<<(unsigned)((n + 255) / 256), 256, 0, c->stream>>>(s.dI[0], s.dI0t, c->w, c->h, wt);
    NALO_HIP(c, hipGetLastError());
    s.tiled_valid = true;
    return NALO_OK;
}

// nalo_hbm_calibrate: the plain streaming kernels the roofline's denominator is measured with (copy: 1 read + 1 write, triad: 2 reads + 1 write per element).
// Shape from scripts/ubench/copy.hip on MI355X: ONE 16-byte element per lane, consecutive workgroups on consecutive memory, nontemporal loads and stores:
// 6.66 TB/s copy (6.21 without the nontemporal hint; a grid-stride loop over 4-32 workgroups per CU reaches only 4.2-5.6, hipMemcpyAsync D2D 5.1; the guide
// quotes 6.29 for a float4 copy).
typedef float nalo_f4 __attribute__((ext_vector_type(4)));
template <int TRIAD>
__global__ __launch_bounds__(256) void hbm_stream_kernel(const nalo_f4* __restrict__ a, const nalo_f4* __restrict__ b, nalo_f4* __restrict__ d, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    nalo_f4 v = __builtin_nontemporal_load(a + i);
    if (TRIAD) v += 3.f * __builtin_nontemporal_load(b + i);
    __builtin_nontemporal_store(v, d + i);
}
void hbm_stream_launch(hipStream_t st, const float4* a, const float4* b, float4* d, size_t n, int triad) {
    const unsigned grid = (unsigned)((n + 255) / 256);
    if (triad) hbm_stream_kernel<1><<<grid, 256, 0, st>>>((const nalo_f4*)a, (const nalo_f4*)b, (nalo_f4*)d, n);
    else hbm_stream_kernel<0><<<grid, 256, 0, st>>>((const nalo_f4*)a, (const nalo_f4*)b, (nalo_f4*)d, n);
}

int pyramid_build(nalo_ctx* c, FrameSlot& s, const float* gammaB_dev) {
    s.tiled_valid = false;
    ProfScope ps(c, "pyramid", true);                                      // dispatch-attached timestamps: start of the first launch .. end of the last
    PyrLevels P;
    P.L = c->levels;
    int nb = 0;
    for (int l = 0; l < c->levels; ++l) {
        P.I[l] = s.I[l]; P.dI[l] = s.dI[l]; P.absg[l] = s.absg[l]; P.wl[l] = c->wl[l]; P.hl[l] = c->hl[l];
        P.blk0[l] = nb; nb += (c->wl[l] * c->hl[l] + 255) / 256;
    }
    P.blk0[c->levels] = nb;
    // the hierarchical pass needs even parents all the way up (true for DSO pyramids: a level is only added while w and h are even) and a tile that still holds a
    // whole pixel of the coarsest level (2^(L-1) <= min(tile width, tile height): 5 levels with the 64x16 tile; a 6-level pyramid - 4096x2048 and beyond - takes
    // the level-by-level path below)
    bool fused = c->levels >= 2 && c->levels <= 6 && (1 << (c->levels - 1)) <= (kPyrTW < kPyrTH ? kPyrTW : kPyrTH);
    for (int l = 1; l < c->levels && fused; ++l) fused = (c->wl[l - 1] % 2 == 0) && (c->hl[l - 1] % 2 == 0);
    if (fused) {
        // one pass over level 0 (pyr_one_pass_kernel): every level of a pyramid of <= 4 levels, the three finest + the planar values of the rest otherwise
        const int tiles = ((c->wl[0] + kPyrTW - 1) / kPyrTW) * ((c->hl[0] + kPyrTH - 1) / kPyrTH);
#define NALO_PYR1(NL_, NC_, E0_, E1_) do { if (ps.a) hipExtLaunchKernelGGL((pyr_one_pass_kernel<NL_>), dim3(tiles + (NC_)), dim3(256), 0, c->stream, E0_, E1_, 0, P, gammaB_dev, (NC_)); \
                                      else pyr_one_pass_kernel<NL_><<<tiles + (NC_), 256, 0, c->stream>>>(P, gammaB_dev, (NC_)); } while (0)
        switch (c->levels) {
            case 2: NALO_PYR1(2, 0, ps.a, ps.b); break;
            case 3: NALO_PYR1(3, 0, ps.a, ps.b); break;
            case 4: NALO_PYR1(4, 0, ps.a, ps.b); break;
            default: {
                // five levels: the levels 3 and 4 by the first workgroups of the same launch (pyr_coarse_tile; the 16-byte loads want w0 % 16 == 0, which five even levels imply)
                const int nc = ((c->wl[3] + PyrCoarse::kCW - 1) / PyrCoarse::kCW) * ((c->hl[3] + PyrCoarse::kCH - 1) / PyrCoarse::kCH);
                if (NALO_PYR_COARSE && c->levels == 5) { NALO_PYR1(3, nc, ps.a, ps.b); break; }
                NALO_PYR1(3, 0, ps.a, nullptr);
                if (ps.a) hipExtLaunchKernelGGL(pyr_grad_tail_kernel, dim3(nb - P.blk0[3]), dim3(256), 0, c->stream, nullptr, ps.b, 0, P, gammaB_dev, 3);
                else pyr_grad_tail_kernel<<<nb - P.blk0[3], 256, 0, c->stream>>>(P, gammaB_dev, 3);
            }
        }
#undef NALO_PYR1
        NALO_HIP(c, hipGetLastError());
        return NALO_OK;
    }
    if (ps.a) (void)hipEventRecord(ps.a, c->stream);
    for (int l = 1; l < c->levels; ++l) {                     // a pyramid with an odd parent level (explicit `levels`): level by level
        const int n = c->wl[l] * c->hl[l];
        pyr_down_kernel<<<std::min((n + 255) / 256, 2048), 256, 0, c->stream>>>(s.I[l - 1], s.I[l], c->wl[l], c->hl[l], c->wl[l - 1]);
    }
    pyr_grad_all_kernel<<<nb, 256, 0, c->stream>>>(P, gammaB_dev);
    if (ps.b) (void)hipEventRecord(ps.b, c->stream);
    NALO_HIP(c, hipGetLastError());
    return NALO_OK;
}

}  // namespace nalo
