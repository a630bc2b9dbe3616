// a14 — DenseMapping::updateMap bounding-box scan (reference src/FullSystem/MapPoint.cpp:300-310) and
// DenseMapping::makeMap pixel loop (MapPoint.cpp:362-404) on gfx950.
//   bbox     : integer atomicMin/Max over the mask (order independent)
//   make_map : per pixel of the rect with mask == value and (i%3==0 || j%3==0): plane depth, colour, world point;
//              ordered (raster) stream compaction = count / scan / write, same scheme as the tracker point clouds
//   extent   : the accept test of :403 INCLUDING the reference's order-dependent typos:
//              maxy/maxz are "the last point whose y (z) is strictly above the running minimum", maxima are seeded with
//              FLT_MIN. Done with an in-order prefix-min scan in one block, so the decision equals the serial loop's.
#include <cfloat>
#include <climits>
#include "nalo_internal.h"
#include <hip/hip_ext.h>

namespace nalo {

// Three launches, no host round trip between them (round 3; was seven launches and two synchronisations, ~105 us at 1920x1072):
//   dense_rows_kernel   one workgroup per image row: {min x, max x} of the pixels with mask == value, eight loads in flight per lane, NO atomics (thousands of
//                       atomicMin/Max on four words were what the first versions spent their time on)
//   dense_count_kernel  every workgroup folds the row table into the box (block 0 stores it), then the aggregate of its 2048-pixel chunk of the box:
//                       {kept points, min x, min y, min z, max x of their world coordinates}
//   dense_write_kernel  exclusive prefix of the aggregates in front of the chunk (a block-wide fold of <= nb entries: cheaper than a spinning look-back, whose
//                       agent-scope polling of 380 workgroups cost 100 us here), the points written in raster order, and the accept test of MapPoint.cpp:403:
// the reference's loop keeps float extrema but compares the double coordinate against them, and its maxy / maxz are "the last point whose y (z) is strictly
// above the running minimum" (SURVEY App. C.6). The running minimum after point k is min_{i<=k} (float) y_i (rounding is monotone), so point k qualifies iff
// y_k > (double) min(prefix, (float) y_k), prefix = the chunks in front + an in-order scan inside the workgroup; the largest qualifying index wins (one 64-bit
// atomicMax on {index + 1, float bits} per workgroup; the host decodes the two words).
__global__ __launch_bounds__(256) void dense_rows_kernel(const float* __restrict__ mask, int w, int h, float value, int2* __restrict__ rows /* [h]: min x, max x */) {
    const int y = 2 + blockIdx.x;
    int minx = INT_MAX, maxx = INT_MIN;
    const float* row = mask + (size_t)y * w;
    for (int x0 = 2; x0 < w - 2; x0 += 8 * 256) {
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { const int x = x0 + k * 256 + threadIdx.x; v[k] = row[x < w - 2 ? x : 2]; }
#pragma unroll
        for (int k = 0; k < 8; ++k) { const int x = x0 + k * 256 + threadIdx.x; if (x < w - 2 && v[k] == value) { minx = min(minx, x); maxx = max(maxx, x); } }
    }
    __shared__ int sm[4][2];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { minx = min(minx, __shfl_down(minx, o)); maxx = max(maxx, __shfl_down(maxx, o)); }
    if ((threadIdx.x & 63) == 0) { const int wv = threadIdx.x >> 6; sm[wv][0] = minx; sm[wv][1] = maxx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k) { minx = min(minx, sm[k][0]); maxx = max(maxx, sm[k][1]); }
        rows[y] = make_int2(minx, maxx);
    }
}

struct DenseParams {
    const float* mask; const float* I0; const uint8_t* bgr;
    int w, h;
    const int2* rows;                            // dense_rows_kernel
    int* rect;                                   // {minx, maxx, miny, maxy}: written by block 0 of dense_count_kernel, read by dense_write_kernel and the host
    float p0, p1, p2, p3, pcolor, fxi, fyi, cx, cy;
    double c2w[12];
    float* agg;                                  // [nblocks][8]: {count (int bits), min x, min y, min z, max x} of a chunk's kept points
    unsigned long long* last;                    // [2]: packed (index + 1, float bits) of the last point that qualifies for maxy / maxz (decoded by the host)
    float* out6;                                 // {minx, maxx, miny, maxy, minz, maxz} of the serial loop
    int* n_out;
};
#ifndef NALO_DENSE_CHUNK
#define NALO_DENSE_CHUNK 1024    // pixels of the box per workgroup. Round 4: 2048 -> 1024 (twice the workgroups for the 256 CUs: 764 instead of 382 on the 460 x 1700 box of the bench)
#endif
constexpr int kDenseChunk = NALO_DENSE_CHUNK;
// e / rw and e % rw for 0 <= e < 2^24 without the ~40-instruction integer division: float quotient, one correction step either way
struct DnDiv { int rw; float rcp; };
__device__ __forceinline__ void dn_divmod(const DnDiv& d, int e, int& q, int& r) {
    if (d.rcp == 0.f) { q = e / d.rw; r = e - q * d.rw; return; }          // a box of >= 2^24 pixels: (float) e is no longer exact
    q = (int)((float)e * d.rcp); r = e - q * d.rw;
    if (r < 0) { --q; r += d.rw; } else if (r >= d.rw) { ++q; r -= d.rw; }
}

__device__ __forceinline__ void dn_world(const DenseParams& P, int i, int j, float idepth, double (&m)[3]) {
    float c0 = P.fxi * j + (-P.cx * P.fxi), c1 = P.fyi * i + (-P.cy * P.fyi), c2 = 1.f;       // cP = Ki * (j,i,1) / idepth (float), mP = camToWorld * cP (double)  (:390-394)
    c0 /= idepth; c1 /= idepth; c2 /= idepth;
#pragma unroll
    for (int q = 0; q < 3; ++q) m[q] = P.c2w[q * 4] * (double)c0 + P.c2w[q * 4 + 1] * (double)c1 + P.c2w[q * 4 + 2] * (double)c2 + P.c2w[q * 4 + 3];
}
// which pixels of this lane's R rounds are kept (MapPoint.cpp:366-380); the mask loads are issued together, an out-of-range lane reads pixel (rx0, ry0) and drops it
// Round 4: the chunks run over the CANDIDATES of the box in raster order - the pixels with (i % 3 == 0 || j % 3 == 0), MapPoint.cpp:366 -, not over all its pixels:
// both kernels are bound by the number of vector instructions they issue, and 4 of 9 lanes spent theirs on pixels that can never be kept. A band of three rows
// starting at a row i = 0 (mod 3) holds rw candidates (the whole first row) + cj + cj (the columns j = 0 (mod 3) of the other two); the virtual bands start at
// i0 = ry0 - ry0 % 3, `pre` candidates of the first band lie above the box.
struct DnMap {
    int rx0, rw, cj, j0, i0, B, pre, total; DnDiv dv;
    __device__ __forceinline__ DnMap(int rx0_, int rx1, int ry0, int ry1, bool any) {
        rx0 = any ? rx0_ : 0; rw = any ? rx1 - rx0 : 1;
        j0 = rx0 + (3 - rx0 % 3) % 3;
        cj = any && j0 < rx1 ? (rx1 - 1 - j0) / 3 + 1 : 0;
        B = rw + 2 * cj;
        const int m = any ? ry0 % 3 : 0;
        i0 = (any ? ry0 : 0) - m;
        pre = m == 0 ? 0 : (m == 1 ? rw : rw + cj);
        const int nrows = any ? ry1 - i0 : 0, full = nrows / 3, r = nrows - 3 * full;
        total = any ? full * B + (r >= 1 ? rw : 0) + (r >= 2 ? cj : 0) - pre : 0;
        dv = DnDiv{B, total + pre < (1 << 24) ? 1.0f / (float)B : 0.f};
    }
    __device__ __forceinline__ void at(int e, int& i, int& j) const {         // candidate e (0 <= e < total) -> pixel (i, j)
        int band, rem; dn_divmod(dv, e + pre, band, rem);
        i = i0 + 3 * band;
        if (rem < rw) j = rx0 + rem;
        else if (rem < rw + cj) { i += 1; j = j0 + 3 * (rem - rw); }
        else { i += 2; j = j0 + 3 * (rem - rw - cj); }
    }
};
// which candidates of this lane's R rounds are kept (MapPoint.cpp:366-380); the mask loads are issued together, an out-of-range lane reads candidate 0 and drops it
template <int R>
__device__ __forceinline__ void dn_keep(const DenseParams& P, const DnMap& M, int b, int tid, bool (&keep)[R], float (&idp)[R], int (&pi)[R], int (&pj)[R]) {
    float mv[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int e = b * kDenseChunk + r * 256 + tid;
        M.at(e < M.total ? e : 0, pi[r], pj[r]);
        mv[r] = P.mask[pj[r] + pi[r] * P.w];
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int e = b * kDenseChunk + r * 256 + tid;
        keep[r] = false; idp[r] = 0.f;
        if (e < M.total) {
            const int i = pi[r], j = pj[r];
            if (mv[r] == P.pcolor) {                                                                                  // (i % 3 == 0 || j % 3 == 0) holds for every candidate
                const float ddepth = P.p0 * (j * P.fxi - P.cx * P.fxi) + P.p1 * (i * P.fyi - P.cy * P.fyi) + P.p2;   // MapPoint.cpp:377
                if (ddepth != 0.f) { const float depth = -P.p3 / ddepth; if (depth != 0.f) { idp[r] = 1.f / depth; keep[r] = true; } }
            }
        }
    }
}
__global__ __launch_bounds__(256) void dense_count_kernel(DenseParams P) {
    constexpr int R = kDenseChunk / 256;
    __shared__ int sr[4][4];
    __shared__ float red[4][5];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
    // ---- the box from the row table (every workgroup on its own: a few hundred entries from L2, all loads in flight; block 0 publishes it and re-arms `last`).
    // No "last workgroup folds it" scheme: an agent-scope fence per workgroup (__threadfence = L2 write-back) cost 20 us in the row kernel and 80 us here.
    int minx = INT_MAX, maxx = INT_MIN, miny = INT_MAX, maxy = INT_MIN;
    {
        int2 rr[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { const int y = 2 + tid + 256 * k; rr[k] = P.rows[y < P.h - 2 ? y : 2]; }
#pragma unroll
        for (int k = 0; k < 8; ++k) { const int y = 2 + tid + 256 * k; if (y < P.h - 2 && rr[k].x != INT_MAX) { minx = min(minx, rr[k].x); maxx = max(maxx, rr[k].y); miny = min(miny, y); maxy = max(maxy, y); } }
        for (int y = 2 + tid + 2048; y < P.h - 2; y += 256) { const int2 r = P.rows[y]; if (r.x != INT_MAX) { minx = min(minx, r.x); maxx = max(maxx, r.y); miny = min(miny, y); maxy = max(maxy, y); } }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { minx = min(minx, __shfl_xor(minx, o)); maxx = max(maxx, __shfl_xor(maxx, o)); miny = min(miny, __shfl_xor(miny, o)); maxy = max(maxy, __shfl_xor(maxy, o)); }
    if (lane == 0) { sr[wave][0] = minx; sr[wave][1] = maxx; sr[wave][2] = miny; sr[wave][3] = maxy; }
    __syncthreads();
    const int rx0 = min(min(sr[0][0], sr[1][0]), min(sr[2][0], sr[3][0])), rx1 = max(max(sr[0][1], sr[1][1]), max(sr[2][1], sr[3][1]));
    const int ry0 = min(min(sr[0][2], sr[1][2]), min(sr[2][2], sr[3][2])), ry1 = max(max(sr[0][3], sr[1][3]), max(sr[2][3], sr[3][3]));
    if (b == 0 && tid == 0) { P.rect[0] = rx0; P.rect[1] = rx1; P.rect[2] = ry0; P.rect[3] = ry1; P.last[0] = 0ull; P.last[1] = 0ull; P.n_out[0] = 0; }
    const bool any = rx0 != INT_MAX && rx1 > rx0 && ry1 > ry0;
    const DnMap M(rx0, rx1, ry0, ry1, any);
    const int nb = (M.total + kDenseChunk - 1) / kDenseChunk;
    if (b >= nb) return;
    bool keep[R]; float idp[R]; int pi[R], pj[R];
    dn_keep<R>(P, M, b, tid, keep, idp, pi, pj);
    int cnt = 0;
    float mnx = FLT_MAX, mny = FLT_MAX, mnz = FLT_MAX, mxx = FLT_MIN;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (!keep[r]) continue;
        double m[3]; dn_world(P, pi[r], pj[r], idp[r], m);
        const float fx = (float)m[0];
        mnx = fminf(mnx, fx); mxx = fmaxf(mxx, fx); mny = fminf(mny, (float)m[1]); mnz = fminf(mnz, (float)m[2]);
        ++cnt;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { cnt += __shfl_xor(cnt, o); mnx = fminf(mnx, __shfl_xor(mnx, o)); mny = fminf(mny, __shfl_xor(mny, o)); mnz = fminf(mnz, __shfl_xor(mnz, o)); mxx = fmaxf(mxx, __shfl_xor(mxx, o)); }
    if (lane == 0) { red[wave][0] = __int_as_float(cnt); red[wave][1] = mnx; red[wave][2] = mny; red[wave][3] = mnz; red[wave][4] = mxx; }
    __syncthreads();
    if (tid == 0) {
        float* o = P.agg + (size_t)b * 8;
        o[0] = __int_as_float(__float_as_int(red[0][0]) + __float_as_int(red[1][0]) + __float_as_int(red[2][0]) + __float_as_int(red[3][0]));
        o[1] = fminf(fminf(red[0][1], red[1][1]), fminf(red[2][1], red[3][1])); o[2] = fminf(fminf(red[0][2], red[1][2]), fminf(red[2][2], red[3][2]));
        o[3] = fminf(fminf(red[0][3], red[1][3]), fminf(red[2][3], red[3][3])); o[4] = fmaxf(fmaxf(red[0][4], red[1][4]), fmaxf(red[2][4], red[3][4]));
    }
}
// inclusive prefix minimum over the 64 lanes of a wave, in lane order, without LDS traffic: Hillis-Steele inside the rows of 16 lanes (DPP row_shr), then the row
// tails handed on (row_bcast:15 into the rows 1 and 3, row_bcast:31 into the rows 2 and 3). Lanes without a source keep FLT_MAX, the identity. Round 4: the write
// kernel is bound by the NUMBER of vector instructions (3 waves per SIMD of ~2600 instructions each = 13 us); the ds_bpermute shuffles (150 per wave) and their
// address arithmetic were the largest removable part.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dn_dpp_min(float v) {
    const float o = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(FLT_MAX), __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
    return fminf(v, o);
}
__device__ __forceinline__ float dn_wave_prefix_min(float v) {
    v = dn_dpp_min<0x111, 0xF>(v); v = dn_dpp_min<0x112, 0xF>(v); v = dn_dpp_min<0x114, 0xF>(v); v = dn_dpp_min<0x118, 0xF>(v);     // row_shr:1, 2, 4, 8
    v = dn_dpp_min<0x142, 0xA>(v);                                                                                                  // row_bcast:15 -> rows 1, 3
    v = dn_dpp_min<0x143, 0xC>(v);                                                                                                  // row_bcast:31 -> rows 2, 3
    return v;
}
__device__ __forceinline__ float dn_lane63(float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63)); }
__global__ __launch_bounds__(256) void dense_write_kernel(DenseParams P, int cap, int* __restrict__ ou, int* __restrict__ ov, float* __restrict__ oid, float* __restrict__ ocol,
                                                          uint8_t* __restrict__ obgr) {
    constexpr int R = kDenseChunk / 256;
    __shared__ int cnt[R][4];
    __shared__ float gmin[R][4][2];              // min of (float) y, (float) z per (round, wave) group of kept points
    __shared__ float red[4][5];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
    const int rx0 = P.rect[0], rx1 = P.rect[1], ry0 = P.rect[2], ry1 = P.rect[3];
    const bool any = rx0 != INT_MAX && rx1 > rx0 && ry1 > ry0;
    const DnMap M(rx0, rx1, ry0, ry1, any);
    const int nb = (M.total + kDenseChunk - 1) / kDenseChunk;
    if (b < nb) {
        // ---- exclusive prefix of the chunks in front: count, min x / y / z, max x (order-free folds; the in-order part happens inside the workgroup).
        // The first four strides are loaded together (a rolled loop waits for every load: three dependent L2 round trips for the last chunks of the bench's box)
        int ecnt = 0; float e1 = FLT_MAX, e2 = FLT_MAX, e3 = FLT_MAX, e4 = FLT_MIN;
        {
            float4 a[4]; float a4[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { const int p = tid + 256 * k, pp = p < b ? p : 0; a[k] = *reinterpret_cast<const float4*>(P.agg + (size_t)pp * 8); a4[k] = P.agg[(size_t)pp * 8 + 4]; }
#pragma unroll
            for (int k = 0; k < 4; ++k) if (tid + 256 * k < b) { ecnt += __float_as_int(a[k].x); e1 = fminf(e1, a[k].y); e2 = fminf(e2, a[k].z); e3 = fminf(e3, a[k].w); e4 = fmaxf(e4, a4[k]); }
            for (int p = tid + 1024; p < b; p += 256) {
                const float4 q = *reinterpret_cast<const float4*>(P.agg + (size_t)p * 8); const float q4 = P.agg[(size_t)p * 8 + 4];
                ecnt += __float_as_int(q.x); e1 = fminf(e1, q.y); e2 = fminf(e2, q.z); e3 = fminf(e3, q.w); e4 = fmaxf(e4, q4);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { ecnt += __shfl_xor(ecnt, o); e1 = fminf(e1, __shfl_xor(e1, o)); e2 = fminf(e2, __shfl_xor(e2, o)); e3 = fminf(e3, __shfl_xor(e3, o)); e4 = fmaxf(e4, __shfl_xor(e4, o)); }
        if (lane == 0) { red[wave][0] = __int_as_float(ecnt); red[wave][1] = e1; red[wave][2] = e2; red[wave][3] = e3; red[wave][4] = e4; }
        bool keep[R]; float idp[R]; int rank[R], pi[R], pj[R];
        dn_keep<R>(P, M, b, tid, keep, idp, pi, pj);
        double wy[R], wz[R];                     // world y, z of the kept points (compared as doubles against the float running minima)
        float fyv[R], fzv[R], py[R], pz[R];      // their float values (FLT_MAX: no point) and the inclusive prefix minima over the wave in lane order
#pragma unroll
        for (int r = 0; r < R; ++r) {
            fyv[r] = FLT_MAX; fzv[r] = FLT_MAX; wy[r] = 0; wz[r] = 0;
            if (keep[r]) { double m[3]; dn_world(P, pi[r], pj[r], idp[r], m); wy[r] = m[1]; wz[r] = m[2]; fyv[r] = (float)m[1]; fzv[r] = (float)m[2]; }
            const unsigned long long mk = __ballot(keep[r]);
            rank[r] = __builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u));
            py[r] = dn_wave_prefix_min(fyv[r]); pz[r] = dn_wave_prefix_min(fzv[r]);
            const float gy = dn_lane63(py[r]), gz = dn_lane63(pz[r]);          // the wave's minima = the last lane's prefix
            if (lane == 0) { cnt[r][wave] = __popcll(mk); gmin[r][wave][0] = gy; gmin[r][wave][1] = gz; }
        }
        __syncthreads();
        ecnt = __float_as_int(red[0][0]) + __float_as_int(red[1][0]) + __float_as_int(red[2][0]) + __float_as_int(red[3][0]);
        e1 = fminf(fminf(red[0][1], red[1][1]), fminf(red[2][1], red[3][1])); e2 = fminf(fminf(red[0][2], red[1][2]), fminf(red[2][2], red[3][2]));
        e3 = fminf(fminf(red[0][3], red[1][3]), fminf(red[2][3], red[3][3])); e4 = fmaxf(fmaxf(red[0][4], red[1][4]), fmaxf(red[2][4], red[3][4]));
        if (b == nb - 1 && tid == 0) {           // the last chunk: prefix + its own aggregate = the totals of the serial loop (maxy / maxz follow from the last workgroup)
            const float* a = P.agg + (size_t)b * 8;
            P.n_out[0] = ecnt + __float_as_int(a[0]);
            P.out6[0] = fminf(e1, a[1]); P.out6[1] = fmaxf(e4, a[4]); P.out6[2] = fminf(e2, a[2]); P.out6[4] = fminf(e3, a[3]);
        }
        // ---- the points in raster order; the in-order prefix minima decide who may be "the last point above the running minimum"
        int off = ecnt;
        float runy = e2, runz = e3;
        // output index o grows with (round, wave, lane): the last qualifying point of a wave is the highest lane of its highest round with one (two ballots per
        // round instead of a 64-bit max reduction over the lanes)
        unsigned long long q1 = 0ull, q2 = 0ull;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            int goff = off; float gy = runy, gz = runz;                     // offset / running minima in front of this (round, wave) group
            for (int k = 0; k < wave; ++k) { goff += cnt[r][k]; gy = fminf(gy, gmin[r][k][0]); gz = fminf(gz, gmin[r][k][1]); }
            bool t1 = false, t2 = false;
            const int o = goff + rank[r];
            if (keep[r]) {
                const int i = pi[r], j = pj[r];
                if (o < cap) {
                    const int px = j + i * P.w;
                    ou[o] = j; ov[o] = i; oid[o] = idp[r]; ocol[o] = P.I0[px];                             // = dI[px][0] (makeImages keeps level 0's planar image): 4 instead of 64 bytes of sector traffic per point
                    if (P.bgr) { obgr[3 * o] = P.bgr[3 * px]; obgr[3 * o + 1] = P.bgr[3 * px + 1]; obgr[3 * o + 2] = P.bgr[3 * px + 2]; }
                }
                const float ry_ = fminf(gy, py[r]), rz_ = fminf(gz, pz[r]);       // the running minima AFTER this point's own update
                t1 = wy[r] > (double)ry_; t2 = wz[r] > (double)rz_;
            }
            const unsigned long long m1 = __ballot(t1), m2 = __ballot(t2);
            if (m1) { const int l = 63 - __clzll(m1); q1 = ((unsigned long long)(unsigned)(__builtin_amdgcn_readlane(o, l) + 1) << 32) | (unsigned)__builtin_amdgcn_readlane(__float_as_int(fyv[r]), l); }
            if (m2) { const int l = 63 - __clzll(m2); q2 = ((unsigned long long)(unsigned)(__builtin_amdgcn_readlane(o, l) + 1) << 32) | (unsigned)__builtin_amdgcn_readlane(__float_as_int(fzv[r]), l); }
            for (int k = 0; k < 4; ++k) { off += cnt[r][k]; runy = fminf(runy, gmin[r][k][0]); runz = fminf(runz, gmin[r][k][1]); }   // this round is behind us
        }
        // the largest qualifying index of the workgroup (nearly every point qualifies: one atomic per LANE was 380 k serialised updates of two words, 88 us)
        __shared__ unsigned long long sq[4][2];
        if (lane == 0) { sq[wave][0] = q1; sq[wave][1] = q2; }
        __syncthreads();
        if (tid == 0) {
            for (int k = 1; k < 4; ++k) { q1 = sq[k][0] > q1 ? sq[k][0] : q1; q2 = sq[k][1] > q2 ? sq[k][1] : q2; }
            if (q1) atomicMax(&P.last[0], q1);
            if (q2) atomicMax(&P.last[1], q2);
        }
    }
}

}  // namespace nalo

using namespace nalo;

extern "C" int nalo_dense_make_map(nalo_ctx* c, int slot, const float plane[4], float mask_value, const double camToWorld[12], int cap,
                                   int rect_out[4], int* out_u, int* out_v, float* out_idepth, float* out_color, uint8_t* out_bgr, int* n_out, int* accept) {
    if (!c || !plane || !camToWorld || !n_out || !accept || cap < 0 || slot < 0 || slot >= (int)c->slots.size())
        return fail(c, NALO_ERR_ARG, "nalo_dense_make_map: bad argument");
    FrameSlot& s = c->slots[slot];
    if (!s.valid || !s.mask) return fail(c, NALO_ERR_STATE, "nalo_dense_make_map: slot has no pyramid / mask (nalo_frame_upload with mask)");
    NALO_HIP(c, hipSetDevice(c->device));
    *n_out = 0; *accept = 0;
    // ---- bbox scan (MapPoint.cpp:287-310), makeMap and the accept test: three launches back to back, the box never leaves the device in between
    const int nblk = std::max(1, ((c->w - 4) * (c->h - 4) + kDenseChunk - 1) / kDenseChunk);
    const size_t hrows = ((size_t)c->h + 1) & ~(size_t)1;                  // the aggregates behind the row table stay 16-byte aligned
    const size_t lbw = hrows + (size_t)nblk * 4 + 8;                       // int2 rows[h] | float agg[nblk][8] | last[2] | ticket  (in 8-byte words)
    if (c->dense_lb.cap < lbw) {
        NALO_HIP(c, hipStreamSynchronize(c->stream));
        NALO_HIP(c, c->dense_lb.reserve(lbw));
        NALO_HIP(c, hipMemsetAsync(c->dense_lb.p, 0, lbw * 8, c->stream));   // `last` and the ticket start at zero and are re-armed by the kernel
    }
    int2* rows = reinterpret_cast<int2*>(c->dense_lb.p);
    NALO_HIP(c, c->scan_tmp.reserve(8));
    {   // dispatch-attached timestamps (no barrier packets around the launch: an event pair recorded on the stream adds several microseconds to a 5 us kernel)
        ProfScope ps(c, "dense_bbox", true);
        if (ps.a) hipExtLaunchKernelGGL(dense_rows_kernel, dim3(std::max(1, c->h - 4)), dim3(256), 0, c->stream, ps.a, ps.b, 0, s.mask, c->w, c->h, mask_value, rows);
        else dense_rows_kernel<<<std::max(1, c->h - 4), 256, 0, c->stream>>>(s.mask, c->w, c->h, mask_value, rows);
    }
    int rect[4];
    float ext[6] = {FLT_MAX, FLT_MIN, FLT_MAX, FLT_MIN, FLT_MAX, FLT_MIN};
    int n = 0;
    const size_t capz = (size_t)std::max(cap, 1);
    const size_t words = capz * 5 + 64;                                      // scratch: u, v (int), idepth, colour (float), bgr carved from upload_tmp
    NALO_HIP(c, c->upload_tmp.reserve(words));
    int* du = (int*)c->upload_tmp.p; int* dv = du + capz; float* did = (float*)(dv + capz); float* dcol = did + capz;
    uint8_t* dbgr = (uint8_t*)(dcol + capz);
    NALO_HIP(c, c->trk_partial.reserve(16));
    DenseParams P;
    P.mask = s.mask; P.I0 = s.I[0]; P.bgr = s.bgr; P.w = c->w; P.h = c->h; P.rows = rows; P.rect = c->scan_tmp.p;
    P.p0 = plane[0]; P.p1 = plane[1]; P.p2 = plane[2]; P.p3 = plane[3]; P.pcolor = mask_value;
    P.fxi = 1.0f / c->fx[0]; P.fyi = 1.0f / c->fy[0]; P.cx = c->cx[0]; P.cy = c->cy[0];            // DenseMapping::makeK level 0
    std::memcpy(P.c2w, camToWorld, sizeof(P.c2w));
    P.agg = reinterpret_cast<float*>(c->dense_lb.p + hrows);
    P.last = c->dense_lb.p + hrows + (size_t)nblk * 4;
    P.out6 = c->trk_partial.p; P.n_out = reinterpret_cast<int*>(c->trk_partial.p + 8);
    if (mask_value == 0.f) {                                                 // `if(pcolor==0) return;` (:355-357): only the box is wanted
        dense_count_kernel<<<1, 256, 0, c->stream>>>(P);
        NALO_HIP(c, hipMemcpyAsync(rect, c->scan_tmp.p, 16, hipMemcpyDeviceToHost, c->stream));
        NALO_HIP(c, hipStreamSynchronize(c->stream));
        if (rect_out) std::memcpy(rect_out, rect, 16);
        return NALO_OK;
    }
    {
        ProfScope ps(c, "dense_map", true);                    // start of the first dispatch .. end of the second
        if (ps.a) {
            hipExtLaunchKernelGGL(dense_count_kernel, dim3(nblk), dim3(256), 0, c->stream, ps.a, nullptr, 0, P);
            hipExtLaunchKernelGGL(dense_write_kernel, dim3(nblk), dim3(256), 0, c->stream, nullptr, ps.b, 0, P, cap, du, dv, did, dcol, dbgr);
        } else {
            dense_count_kernel<<<nblk, 256, 0, c->stream>>>(P);
            dense_write_kernel<<<nblk, 256, 0, c->stream>>>(P, cap, du, dv, did, dcol, dbgr);
        }
    }
    float res[9];
    unsigned long long lastq[2] = {0, 0};
    NALO_HIP(c, hipMemcpyAsync(rect, c->scan_tmp.p, 16, hipMemcpyDeviceToHost, c->stream));
    NALO_HIP(c, hipMemcpyAsync(res, c->trk_partial.p, sizeof(res), hipMemcpyDeviceToHost, c->stream));
    NALO_HIP(c, hipMemcpyAsync(lastq, P.last, 16, hipMemcpyDeviceToHost, c->stream));
    NALO_HIP(c, hipStreamSynchronize(c->stream));
    for (int k = 0; k < 2; ++k) {                                            // maxy / maxz: the coordinate of the last qualifying point, FLT_MIN if none (the reference's seed)
        const unsigned bits = (unsigned)lastq[k]; float v = FLT_MIN;
        if (lastq[k] >> 32) std::memcpy(&v, &bits, 4);
        res[k == 0 ? 3 : 5] = v;
    }
    if (rect_out) std::memcpy(rect_out, rect, 16);
    std::memcpy(&n, &res[8], 4);
    if (n > cap) return fail(c, NALO_ERR_ARG, "nalo_dense_make_map: cap too small");
    if (n > 0) {
        std::memcpy(ext, res, 24);
        if (out_u) NALO_HIP(c, hipMemcpyAsync(out_u, du, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
        if (out_v) NALO_HIP(c, hipMemcpyAsync(out_v, dv, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
        if (out_idepth) NALO_HIP(c, hipMemcpyAsync(out_idepth, did, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
        if (out_color) NALO_HIP(c, hipMemcpyAsync(out_color, dcol, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
        if (out_bgr && s.bgr) NALO_HIP(c, hipMemcpyAsync(out_bgr, dbgr, (size_t)n * 3, hipMemcpyDeviceToHost, c->stream));
        NALO_HIP(c, hipStreamSynchronize(c->stream));
    }
    NALO_HIP(c, hipGetLastError());
    *n_out = n;
    *accept = (ext[1] - ext[0] < 30 && ext[3] - ext[2] < 30 && ext[5] - ext[4] < 30) ? 1 : 0;       // :403
    return NALO_OK;
}
