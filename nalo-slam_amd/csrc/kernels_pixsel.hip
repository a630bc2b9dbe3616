// Candidate-pixel selection for gfx950 (reference paths relative to src/FullSystem/).
//
//  nalo_pixsel_set_random        PixelSelector ctor's randomPattern (PixelSelector2.cpp:40-45) + FusedWithMask's srand(3141592)/rand() stream (:496-501):
//                                libc streams, owned by the caller, resident in HBM afterwards
//  nalo_pixsel_select            PixelSelector::select (:564-711)
//  nalo_pixsel_make_maps         PixelSelector::makeMaps (:144-291)
//  nalo_pixsel_make_maps_lidar   PixelSelector::makeMaps_lidar (:293-428) = makeHists + select + FusedWithMask (:431-560)
//  nalo_pixsel_get_selected      the raster-ordered compact list of the last map (what makeNewTraces walks the map for, FullSystem.cpp:1672-1690)
//
// select() looks sequential: every cell's gradient direction is directions[randomPattern[n2] & 15] with n2 = the number of level-1 selections made
// so far in scan order. But whether a cell selects at all almost never depends on its direction (it needs one pixel above the block threshold with a
// non-zero projected gradient), so the kernels run it as count -> exclusive scan -> select, and then CHECK: a cell whose selection flag differs from
// the counted one (all its candidates have exactly zero gradient along the drawn direction) raises a mismatch flag and the scan/select pair is
// repeated with the corrected flags. Every round fixes at least the first wrong cell in scan order, the fixed point is the sequential result.
// The -2 flags of the reference loop reduce to closed forms (see pixsel_cells_kernel): a 2pot block selects its level-2 pixel iff none of its cells
// selected, a 4pot block its level-3 pixel iff nothing below selected; "first maximum in scan order" = strict > combined in lane order.
//
// Thread <-> cell (one wave = four 4pot blocks); slot index = scan order, so the scan runs over thread ids. Compiled without FMA contraction, like
// the oracle: the argmax compares |gx*dx + gy*dy| values that differ in the last bit under contraction.
#include "nalo_internal.h"

namespace nalo {

struct PixSel {
    DevBuf<uint8_t> rp, map, has, sel;
    DevBuf<int> draws, pre, chunk, list, cnt;   // cnt: [0] mismatch [1..3] n2 n3 n4 [4] list length [5] kept (sub-select) [6,7] fused n1 n2 [8..] mask histogram (257)
    DevBuf<float> ths;                           // [ths (nb + 100) | thsSmoothed (nb + 100)], zero tails (see pixsel_state)
    int* host = nullptr;                         // pinned: [0..15] counters, [16..] list
    size_t host_cap = 0;
    int hist_slot = -1;
    bool have_rp = false, have_draws = false, have_map = false;
    int list_n = 0;
};

__constant__ float c_dirs[16][2] = {{0.f, 1.0000f},     {0.3827f, 0.9239f},  {0.1951f, 0.9808f},  {0.9239f, 0.3827f}, {0.7071f, 0.7071f},  {0.3827f, -0.9239f},
                                    {0.8315f, 0.5556f}, {0.8315f, -0.5556f}, {0.5556f, -0.8315f}, {0.9808f, 0.1951f}, {0.9239f, -0.3827f}, {0.7071f, -0.7071f},
                                    {0.5556f, 0.8315f}, {0.9808f, -0.1951f}, {1.0000f, 0.0000f},  {0.1951f, -0.9808f}};   // PixelSelector2.cpp:581-597

struct PixSelArgs {
    const float4* dI; const float *ag0, *ag1, *ag2, *thsSmoothed;
    const uint8_t* rp;
    int w, h, pot, nb4x, nslots;
    float thFactor;
};

// SELECT = false: has[slot] = the cell holds a pixel above its block threshold (the speculative "this cell selects").
// SELECT = true : the selection proper, given pre[slot] = number of selecting cells before slot in scan order.
// L lanes share a cell (1 for pot <= 3, 4 up to 7, 16 above: a cell of potential 16 has 256 pixels): each walks a contiguous raster run of the cell's
// pixels, "first strict maximum" is then combined in lane order, cells of a 2pot block and of a 4pot block likewise (4L and 16L consecutive lanes; with
// L = 16 a 4pot block is the whole 256-thread workgroup and its four waves meet in LDS).
template <int G>
__device__ __forceinline__ void pixsel_first_max(float& v, int& k, int lane, int stride) {       // over the G lanes lane0, lane0 + stride, ... of this lane's group
    const int lane0 = lane & ~(G * stride - 1);
    float bv = 0.f; int bk = -1;
#pragma unroll
    for (int j = 0; j < G; ++j) { const float ov = __shfl(v, lane0 + j * stride); const int ok = __shfl(k, lane0 + j * stride); if (ov > bv) { bv = ov; bk = ok; } }
    v = bv; k = bk;
}
template <bool SELECT, int L>
__global__ __launch_bounds__(256) void pixsel_cells_kernel(PixSelArgs P, uint8_t* __restrict__ has, const int* __restrict__ pre, uint8_t* __restrict__ sel,
                                                           uint8_t* __restrict__ map, int* __restrict__ cnt) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63, s = t / L, sub = t % L;
    const bool live = s < P.nslots;
    const int b4 = s >> 4, c = s & 15, b3 = c >> 2, c2 = c & 3, pot = P.pot, w = P.w, h = P.h;
    const int x0 = (b4 % P.nb4x) * 4 * pot + (b3 & 1) * 2 * pot + (c2 & 1) * pot, y0 = (b4 / P.nb4x) * 4 * pot + (b3 >> 1) * 2 * pot + (c2 >> 1) * pot;
    const bool cell = live && x0 < w && y0 < h;
    const int mx = cell ? min(pot, w - x0) : 0, my = cell ? min(pot, h - y0) : 0, tot = mx * my, per = (tot + L - 1) / L;
    const int p0 = min(sub * per, tot), p1 = min(p0 + per, tot);                         // this lane's raster run of the cell
    const int w1 = w / 2, w2 = w / 4, thsStep = w / 32;
    const float dw1 = kGradDownweightPerLevel, dw2 = dw1 * dw1;  // settings.cpp:156
    constexpr unsigned long long kCellMask = L >= 64 ? ~0ull : ((1ull << L) - 1);
    if constexpr (!SELECT) {
        bool any = false;
        for (int p = p0; p < p1; ++p) {
            const int xf = x0 + p % mx, yf = y0 + p / mx;
            if (xf < 4 || xf >= w - 5 || yf < 4 || yf > h - 4) continue;
            any |= P.ag0[xf + w * yf] > P.thsSmoothed[(xf >> 5) + (yf >> 5) * thsStep] * P.thFactor;
        }
        const unsigned long long m = __ballot(any);
        if (live && sub == 0) has[s] = ((m >> (lane & ~(L - 1))) & kCellMask) ? 1 : 0;
        return;
    } else {
        const int sl = live ? s : 0;
        const int i2 = P.rp[pre[sl]] & 15, i3 = P.rp[pre[sl & ~3]] & 15, i4 = P.rp[pre[sl & ~15]] & 15;      // n2 when the cell / 2pot block / 4pot block was entered
        const float d2x = c_dirs[i2][0], d2y = c_dirs[i2][1], d3x = c_dirs[i3][0], d3y = c_dirs[i3][1], d4x = c_dirs[i4][0], d4y = c_dirs[i4][1];
        float v2 = 0.f, v3 = 0.f, v4 = 0.f;
        int k2 = -1, k3 = -1, k4 = -1;
        for (int p = p0; p < p1; ++p) {
            const int xf = x0 + p % mx, yf = y0 + p / mx, idx = xf + w * yf;
            if (xf < 4 || xf >= w - 5 || yf < 4 || yf > h - 4) continue;
            const float th0 = P.thsSmoothed[(xf >> 5) + (yf >> 5) * thsStep], th1 = th0 * dw1, th2 = th1 * dw2;
            const float4 tx = P.dI[idx];
            if (P.ag0[idx] > th0 * P.thFactor) { const float dn = fabsf(tx.y * d2x + tx.z * d2y); if (dn > v2) { v2 = dn; k2 = idx; } }
            if (P.ag1[(int)(xf * 0.5f + 0.25f) + (int)(yf * 0.5f + 0.25f) * w1] > th1 * P.thFactor) { const float dn = fabsf(tx.y * d3x + tx.z * d3y); if (dn > v3) { v3 = dn; k3 = idx; } }
            if (P.ag2[(int)(xf * 0.25f + 0.125) + (int)(yf * 0.25f + 0.125) * w2] > th2 * P.thFactor) { const float dn = fabsf(tx.y * d4x + tx.z * d4y); if (dn > v4) { v4 = dn; k4 = idx; } }
        }
        if constexpr (L > 1) { pixsel_first_max<L>(v2, k2, lane, 1); pixsel_first_max<L>(v3, k3, lane, 1); pixsel_first_max<L>(v4, k4, lane, 1); }     // the cell
        const bool s2 = k2 > 0;                                 // identical on the L lanes of the cell
        const unsigned long long m2 = __ballot(s2);
        // level 2 (map value 2): first strict maximum over the 2pot block's cells in scan order; alive iff none of its cells selected
        pixsel_first_max<4>(v3, k3, lane, L);
        constexpr int G2 = 4 * L;                               // lanes of a 2pot block (<= 64)
        constexpr unsigned long long kG2Mask = G2 >= 64 ? ~0ull : ((1ull << G2) - 1);
        const bool s3 = ((m2 >> (lane & ~(G2 - 1))) & kG2Mask) == 0 && k3 > 0;
        bool s4;
        if constexpr (L <= 4) {                                 // the 4pot block sits inside the wave
            constexpr int G4 = 16 * L;
            constexpr unsigned long long kG4Mask = G4 >= 64 ? ~0ull : ((1ull << G4) - 1);
            const unsigned long long m3 = __ballot(s3);
            pixsel_first_max<16>(v4, k4, lane, L);
            s4 = ((m2 >> (lane & ~(G4 - 1))) & kG4Mask) == 0 && ((m3 >> (lane & ~(G4 - 1))) & kG4Mask) == 0 && k4 > 0;
        } else {                                                // L = 16: wave = 2pot block, workgroup = 4pot block
            __shared__ float sv[4]; __shared__ int sk[4], sany[4];
            pixsel_first_max<4>(v4, k4, lane, L);
            if (lane == 0) { sv[threadIdx.x >> 6] = v4; sk[threadIdx.x >> 6] = k4; sany[threadIdx.x >> 6] = (m2 != 0) || s3; }
            __syncthreads();
            float bv = 0.f; int bk = -1; bool below = false;
#pragma unroll
            for (int j = 0; j < 4; ++j) { if (sv[j] > bv) { bv = sv[j]; bk = sk[j]; } below |= sany[j] != 0; }
            k4 = bk;
            s4 = !below && bk > 0;
        }
        const bool lead = sub == 0;
        if (s2 && lead) map[k2] = 1;
        if (s3 && lead && c2 == 0) map[k3] = 2;
        if (s4 && lead && c == 0) map[k4] = 4;
        if (live && lead) { sel[s] = s2 ? 1 : 0; if ((has[s] != 0) != s2) cnt[0] = 1; }
        // counts: one integer atomic per wave and level
        const int n2 = __popcll(__ballot(s2 && lead)), n3 = __popcll(__ballot(s3 && lead && c2 == 0)), n4 = __popcll(__ballot(s4 && lead && c == 0));
        if (lane == 0) { if (n2) atomicAdd(&cnt[1], n2); if (n3) atomicAdd(&cnt[2], n3); if (n4) atomicAdd(&cnt[3], n4); }
    }
}

// exclusive scan of n byte flags (single workgroup: contiguous runs per thread, wave shuffles, 16 wave totals)
__global__ __launch_bounds__(1024) void pixsel_scan_bytes_kernel(const uint8_t* __restrict__ f, int n, int* __restrict__ out, int* __restrict__ zero8) {
    __shared__ int wtot[16];
    if (zero8 && threadIdx.x < 8) zero8[threadIdx.x] = 0;          // the counters of the selection pass behind this launch (was a fill launch of its own)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, per = (n + 1023) / 1024, a = min(tid * per, n), b = min(a + per, n);
    int s = 0;
    for (int i = a; i < b; ++i) s += f[i];
    int v = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(v, o); if (lane >= o) v += t; }
    if (lane == 63) wtot[wave] = v;
    __syncthreads();
    int base = v - s;
    for (int k = 0; k < wave; ++k) base += wtot[k];
    for (int i = a; i < b; ++i) { out[i] = base; base += f[i]; }
}
__global__ __launch_bounds__(1024) void pixsel_scan_ints_kernel(int* __restrict__ v, int n, int* __restrict__ total, int* __restrict__ zero1) {       // in place, exclusive; n <= 1024 * per
    __shared__ int wtot[16];
    if (zero1 && threadIdx.x == 0) *zero1 = 0;                     // the compaction's kept counter (was a fill launch of its own)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, per = (n + 1023) / 1024, a = min(tid * per, n), b = min(a + per, n);
    int s = 0;
    for (int i = a; i < b; ++i) s += v[i];
    int x = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(x, o); if (lane >= o) x += t; }
    if (lane == 63) wtot[wave] = x;
    __syncthreads();
    int base = x - s;
    for (int k = 0; k < wave; ++k) base += wtot[k];
    for (int i = a; i < b; ++i) { const int t = v[i]; v[i] = base; base += t; }
    if (tid == 1023) *total = base;
}

// raster-order compaction of the status map: chunks of 2048 pixels
constexpr int kChunk = 2048;
__global__ __launch_bounds__(256) void pixsel_count_kernel(const uint8_t* __restrict__ map, int n, int* __restrict__ chunk) {
    __shared__ int wsum[4];
    const int i0 = blockIdx.x * kChunk + threadIdx.x * 8;
    int s = 0;
    if (i0 + 8 <= n) { const uint2 q = *reinterpret_cast<const uint2*>(map + i0); s = __popc((q.x | (q.x >> 1) | (q.x >> 2)) & 0x01010101u) + __popc((q.y | (q.y >> 1) | (q.y >> 2)) & 0x01010101u); }
    else for (int i = i0; i < n; ++i) s += map[i] != 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) chunk[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
// list[rank] = idx | status << 28. charTH >= 0: makeMaps' random sub-selection (:232-247): the rank-th non-zero pixel is dropped when
// randomPattern[rank] > charTH (status 0 in the list, map cleared); cnt[5] counts the kept ones.
__global__ __launch_bounds__(256) void pixsel_compact_kernel(uint8_t* __restrict__ map, int n, const int* __restrict__ chunk, const uint8_t* __restrict__ rp, int charTH,
                                                             int* __restrict__ list, int* __restrict__ cnt) {
    __shared__ int wsum[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i0 = blockIdx.x * kChunk + tid * 8;
    uint8_t v[8];
    int s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { v[k] = (i0 + k < n) ? map[i0 + k] : 0; s += v[k] != 0; }
    int x = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(x, o); if (lane >= o) x += t; }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    int rank = chunk[blockIdx.x] + x - s;
    for (int k = 0; k < wave; ++k) rank += wsum[k];
    int kept = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (!v[k]) continue;
        int st = v[k];
        if (charTH >= 0 && (int)rp[rank] > charTH) { st = 0; map[i0 + k] = 0; }
        kept += st != 0;
        list[rank++] = (i0 + k) | (st << 28);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) kept += __shfl_down(kept, o);
    if (lane == 0 && kept) atomicAdd(&cnt[5], kept);
}

// FusedWithMask: histogram of the non-zero mask values ([0] = their count), then the per-pixel status changes
__global__ __launch_bounds__(256) void pixsel_mask_hist_kernel(const float* __restrict__ mask, int n, int* __restrict__ hist /* [257] */) {
    __shared__ int lh[257];
    for (int i = threadIdx.x; i < 257; i += blockDim.x) lh[i] = 0;
    __syncthreads();
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float m = mask[i];
        if (m != 0) { atomicAdd(&lh[0], 1); atomicAdd(&lh[min(max((int)m, 0), 256)], 1); }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 257; i += blockDim.x) if (lh[i]) atomicAdd(&hist[i], lh[i]);
}
__global__ __launch_bounds__(256) void pixsel_fuse_kernel(const float* __restrict__ mask, const int* __restrict__ draws, int n, uint8_t* __restrict__ map, int* __restrict__ cnt) {
    __shared__ int s_q, s_m;
    const int* hist = cnt + 8;
    if (threadIdx.x == 0) {                                     // :453-477; hist[256] is the entry the reference reads one past its array: defined as 0 here
        int th_ = (int)(hist[0] * 0.5 + 0.5f), quantile = 255, max_mas = 0;
        for (int i = 0; i < 256; ++i) { th_ -= (i + 1 < 256 ? hist[i + 1] : 0); if (th_ < 0) { quantile = i; break; } }
        for (int i = 255; i > 0; --i) { max_mas = i; if (hist[i] != 0) break; }
        s_q = quantile; s_m = max_mas;
    }
    __syncthreads();
    const int quantile = s_q, max_mas = s_m, i = blockIdx.x * blockDim.x + threadIdx.x;
    int st = 0;
    if (i < n) {
        st = map[i];
        const float rs = draws[i] % 1000 / (float)(1000.0), m = mask[i];
        if (st == 1) { if (rs > 0.5 && m < quantile / 3) st = 2; }
        else if (st == 2) { if (rs < 0.6 && m > quantile + (max_mas - quantile) / 2) st = 1; }
        else if (rs < 0.01 && m > quantile) st = 1;
        map[i] = (uint8_t)st;
    }
    const int n1 = __popcll(__ballot(st == 1)), n2 = __popcll(__ballot(st == 2));
    if ((threadIdx.x & 63) == 0) { if (n1) atomicAdd(&cnt[6], n1); if (n2) atomicAdd(&cnt[7], n2); }
}

void pixsel_destroy(nalo_ctx* c) {
    PixSel* p = c->pixsel;
    if (!p) return;
    p->rp.release(); p->map.release(); p->has.release(); p->sel.release(); p->draws.release(); p->pre.release(); p->chunk.release(); p->list.release(); p->cnt.release(); p->ths.release();
    if (p->host) (void)hipHostFree(p->host);
    delete p;
    c->pixsel = nullptr;
}
void pixsel_invalidate_hists(nalo_ctx* c, int slot) { if (c->pixsel && c->pixsel->hist_slot == slot) c->pixsel->hist_slot = -1; }

static int pixsel_state(nalo_ctx* c, PixSel** out) {
    if (!c->pixsel) c->pixsel = new PixSel();
    PixSel* p = c->pixsel;
    const size_t n = (size_t)c->w * c->h;
    NALO_HIP(c, p->map.reserve(n + 16)); NALO_HIP(c, p->list.reserve(n)); NALO_HIP(c, p->cnt.reserve(8 + 264));
    NALO_HIP(c, p->chunk.reserve((n + kChunk - 1) / kChunk + 1));
    // PixelSelector allocates (w/32)*(h/32)+100 thresholds and fills (w/32)*(h/32); select() indexes (x>>5) + (y>>5)*(w/32), which on sizes that are
    // not multiples of 32 (KITTI 1224x368) reaches the next row or the uninitialised tail. Same indexing here; the tail is DEFINED as 0.
    const size_t nbp = (size_t)(c->w / 32) * (c->h / 32) + 100;
    if (p->ths.cap < 2 * nbp) { NALO_HIP(c, p->ths.reserve(2 * nbp)); NALO_HIP(c, hipMemsetAsync(p->ths.p, 0, 2 * nbp * sizeof(float), c->stream)); }
    if (p->host_cap < n + 16) {
        if (p->host) (void)hipHostFree(p->host);
        p->host = nullptr; p->host_cap = 0;
        NALO_HIP(c, hipHostMalloc((void**)&p->host, (n + 16) * sizeof(int)));
        p->host_cap = n + 16;
    }
    *out = p;
    return NALO_OK;
}
static int pixsel_hists(nalo_ctx* c, PixSel* p, int slot) {
    const size_t nbp = (size_t)(c->w / 32) * (c->h / 32) + 100;
    int rc = pixsel_hists_launch(c, c->slots[slot].absg[0], p->ths.p, p->ths.p + nbp);
    if (rc) return rc;
    p->hist_slot = slot;
    return NALO_OK;
}
// select on the device: map (bytes) + counters. n[3] = {n2, n3, n4} of PixelSelector::select.
static int pixsel_select_dev(nalo_ctx* c, PixSel* p, int slot, int pot, float thFactor, int n[3]) {
    const FrameSlot& s = c->slots[slot];
    const size_t npx = (size_t)c->w * c->h, nbp = (size_t)(c->w / 32) * (c->h / 32) + 100;
    PixSelArgs A;
    A.dI = s.dI[0]; A.ag0 = s.absg[0]; A.ag1 = s.absg[1]; A.ag2 = s.absg[2]; A.thsSmoothed = p->ths.p + nbp; A.rp = p->rp.p;
    A.w = c->w; A.h = c->h; A.pot = pot; A.nb4x = (c->w + 4 * pot - 1) / (4 * pot);
    A.nslots = A.nb4x * ((c->h + 4 * pot - 1) / (4 * pot)) * 16;
    A.thFactor = thFactor;
    NALO_HIP(c, p->has.reserve(A.nslots)); NALO_HIP(c, p->sel.reserve(A.nslots)); NALO_HIP(c, p->pre.reserve(A.nslots));
    const int L = pot <= 3 ? 1 : (pot <= 7 ? 4 : 16), grid = (int)(((size_t)A.nslots * L + 255) / 256);
    auto cells = [&](bool select, uint8_t* fl, uint8_t* out) {
        if (!select) {
            if (L == 1) pixsel_cells_kernel<false, 1><<<grid, 256, 0, c->stream>>>(A, fl, nullptr, nullptr, nullptr, nullptr);
            else if (L == 4) pixsel_cells_kernel<false, 4><<<grid, 256, 0, c->stream>>>(A, fl, nullptr, nullptr, nullptr, nullptr);
            else pixsel_cells_kernel<false, 16><<<grid, 256, 0, c->stream>>>(A, fl, nullptr, nullptr, nullptr, nullptr);
        } else {
            if (L == 1) pixsel_cells_kernel<true, 1><<<grid, 256, 0, c->stream>>>(A, fl, p->pre.p, out, p->map.p, p->cnt.p);
            else if (L == 4) pixsel_cells_kernel<true, 4><<<grid, 256, 0, c->stream>>>(A, fl, p->pre.p, out, p->map.p, p->cnt.p);
            else pixsel_cells_kernel<true, 16><<<grid, 256, 0, c->stream>>>(A, fl, p->pre.p, out, p->map.p, p->cnt.p);
        }
    };
    cells(false, p->has.p, nullptr);
    uint8_t *flags = p->has.p, *other = p->sel.p;
    for (int round = 0;; ++round) {
        if (round > A.nslots) return fail(c, NALO_ERR_STATE, "nalo_pixsel_select: the selection did not reach its fixed point");
        pixsel_scan_bytes_kernel<<<1, 1024, 0, c->stream>>>(flags, A.nslots, p->pre.p, p->cnt.p);
        NALO_HIP(c, hipMemsetAsync(p->map.p, 0, npx, c->stream));
        cells(true, flags, other);
        NALO_HIP(c, hipMemcpyAsync(p->host, p->cnt.p, 4 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        NALO_HIP(c, hipStreamSynchronize(c->stream));
        if (!p->host[0]) break;
        std::swap(flags, other);                                // a cell's flag was wrong: redo the scan with the flags the selection produced
    }
    n[0] = p->host[1]; n[1] = p->host[2]; n[2] = p->host[3];
    p->have_map = true;
    return NALO_OK;
}
// compaction (+ optional sub-selection) and the host copies: list -> pinned, map_out filled from the list
// expect: the number of selected pixels the caller already knows (n2 + n3 + n4 of the selection pass), or < 0: the list then travels with the counters, ONE completion
// round trip instead of two (round 4)
static int pixsel_fetch(nalo_ctx* c, PixSel* p, int charTH, float* map_out, int* kept, int expect = -1) {
    const int npx = c->w * c->h, nch = (npx + kChunk - 1) / kChunk;
    pixsel_count_kernel<<<nch, 256, 0, c->stream>>>(p->map.p, npx, p->chunk.p);
    pixsel_scan_ints_kernel<<<1, 1024, 0, c->stream>>>(p->chunk.p, nch, p->cnt.p + 4, p->cnt.p + 5);
    pixsel_compact_kernel<<<nch, 256, 0, c->stream>>>(p->map.p, npx, p->chunk.p, p->rp.p, charTH, p->list.p, p->cnt.p);
    NALO_HIP(c, hipMemcpyAsync(p->host, p->cnt.p, 8 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    if (expect > 0) NALO_HIP(c, hipMemcpyAsync(p->host + 16, p->list.p, (size_t)expect * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    NALO_HIP(c, hipStreamSynchronize(c->stream));
    const int total = p->host[4];
    if (kept) *kept = p->host[5];
    if (total > 0 && total != expect) {                     // no expectation, or not the count the device found (never seen): the list at its real length
        NALO_HIP(c, hipMemcpyAsync(p->host + 16, p->list.p, (size_t)total * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        NALO_HIP(c, hipStreamSynchronize(c->stream));
    }
    // the list keeps only live entries (sub-selected ones are dropped here)
    int m = 0;
    for (int i = 0; i < total; ++i) { const int e = p->host[16 + i]; if (e >> 28) p->host[16 + m++] = e; }
    p->list_n = m;
    if (map_out) {
        std::memset(map_out, 0, sizeof(float) * (size_t)npx);
        for (int i = 0; i < m; ++i) { const int e = p->host[16 + i]; map_out[e & 0x0FFFFFFF] = (float)(e >> 28); }
    }
    return NALO_OK;
}
static int pixsel_check(nalo_ctx* c, int slot, const char* who, bool need_draws) {
    if (!c) return NALO_ERR_ARG;
    if (slot < 0 || slot >= (int)c->slots.size() || !c->slots[slot].valid) return fail(c, NALO_ERR_STATE, std::string(who) + ": frame slot has no pyramid");
    if (c->levels < 3) return fail(c, NALO_ERR_STATE, std::string(who) + ": needs 3 pyramid levels");
    if (!c->pixsel || !c->pixsel->have_rp) return fail(c, NALO_ERR_STATE, std::string(who) + ": nalo_pixsel_set_random has not been called");
    if (need_draws && !c->pixsel->have_draws) return fail(c, NALO_ERR_STATE, std::string(who) + ": no mask_draws were given to nalo_pixsel_set_random");
    if (need_draws && !c->slots[slot].mask) return fail(c, NALO_ERR_STATE, std::string(who) + ": the frame was uploaded without a mask");
    return NALO_OK;
}

}  // namespace nalo

using namespace nalo;

extern "C" {

int nalo_pixsel_set_random(nalo_ctx* c, const uint8_t* randomPattern, const int* mask_draws) {
    if (!c || !randomPattern) return fail(c, NALO_ERR_ARG, "nalo_pixsel_set_random: bad argument");
    NALO_HIP(c, hipSetDevice(c->device));
    PixSel* p = nullptr;
    int rc = pixsel_state(c, &p); if (rc) return rc;
    const size_t n = (size_t)c->w * c->h;
    NALO_HIP(c, p->rp.reserve(n + 16));
    NALO_HIP(c, hipMemcpyAsync(p->rp.p, randomPattern, n, hipMemcpyHostToDevice, c->stream));
    p->have_rp = true;
    if (mask_draws) {
        NALO_HIP(c, p->draws.reserve(n));
        NALO_HIP(c, hipMemcpyAsync(p->draws.p, mask_draws, n * sizeof(int), hipMemcpyHostToDevice, c->stream));
        p->have_draws = true;
    }
    NALO_HIP(c, hipStreamSynchronize(c->stream));
    return NALO_OK;
}

int nalo_pixsel_make_hists(nalo_ctx* c, int slot, float* ths, float* thsSmoothed) {
    if (!c) return NALO_ERR_ARG;
    if (slot < 0 || slot >= (int)c->slots.size() || !c->slots[slot].valid) return fail(c, NALO_ERR_STATE, "nalo_pixsel_make_hists: frame slot has no pyramid");
    NALO_HIP(c, hipSetDevice(c->device));
    PixSel* p = nullptr;
    int rc = pixsel_state(c, &p); if (rc) return rc;
    const size_t nb = (size_t)(c->w / 32) * (c->h / 32);
    if (nb == 0) return NALO_OK;
    rc = pixsel_hists(c, p, slot); if (rc) return rc;
    if (ths || thsSmoothed) {
        NALO_HIP(c, hipMemcpyAsync(p->host, p->ths.p, 2 * (nb + 100) * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        NALO_HIP(c, hipStreamSynchronize(c->stream));
        if (ths) std::memcpy(ths, p->host, nb * sizeof(float));
        if (thsSmoothed) std::memcpy(thsSmoothed, reinterpret_cast<float*>(p->host) + nb + 100, nb * sizeof(float));
    }
    return NALO_OK;
}

int nalo_pixsel_select(nalo_ctx* c, int slot, int pot, float thFactor, float* map_out, int n[3]) {
    int rc = pixsel_check(c, slot, "nalo_pixsel_select", false); if (rc) return rc;
    if (pot < 1 || !n) return fail(c, NALO_ERR_ARG, "nalo_pixsel_select: bad argument");
    NALO_HIP(c, hipSetDevice(c->device));
    PixSel* p = nullptr;
    rc = pixsel_state(c, &p); if (rc) return rc;
    if (p->hist_slot != slot) return fail(c, NALO_ERR_STATE, "nalo_pixsel_select: nalo_pixsel_make_hists has not run on this frame (PixelSelector::gradHistFrame)");
    ProfScope ps(c, "pixsel");
    rc = pixsel_select_dev(c, p, slot, pot, thFactor, n); if (rc) return rc;
    return pixsel_fetch(c, p, -1, map_out, nullptr, n[0] + n[1] + n[2]);
}

int nalo_pixsel_make_maps(nalo_ctx* c, int slot, float density, int recursionsLeft, float thFactor, int* currentPotential, float* map_out, int* numHaveSub) {
    int rc = pixsel_check(c, slot, "nalo_pixsel_make_maps", false); if (rc) return rc;
    if (!currentPotential || *currentPotential < 1 || !numHaveSub) return fail(c, NALO_ERR_ARG, "nalo_pixsel_make_maps: bad argument");
    NALO_HIP(c, hipSetDevice(c->device));
    PixSel* p = nullptr;
    rc = pixsel_state(c, &p); if (rc) return rc;
    ProfScope ps(c, "pixsel");
    if (p->hist_slot != slot) { rc = pixsel_hists(c, p, slot); if (rc) return rc; }          // if (fh != gradHistFrame) makeHists(fh)
    int cur = *currentPotential, ideal = cur, n[3];
    float numHave = 0, quotia = 0;
    for (;;) {                                                  // PixelSelector2.cpp:180-228 (the recursion as a loop)
        rc = pixsel_select_dev(c, p, slot, cur, thFactor, n); if (rc) return rc;
        numHave = (float)(n[0] + n[1] + n[2]);
        quotia = density / numHave;
        const float K = numHave * (cur + 1) * (cur + 1);
        ideal = (int)(sqrtf(K / density) - 1);
        if (ideal < 1) ideal = 1;
        if (recursionsLeft > 0 && quotia > 1.25 && cur > 1) { if (ideal >= cur) ideal = cur - 1; cur = ideal; --recursionsLeft; continue; }
        if (recursionsLeft > 0 && quotia < 0.25) { if (ideal <= cur) ideal = cur + 1; cur = ideal; --recursionsLeft; continue; }
        break;
    }
    const bool sub = quotia < 0.95;
    int kept = 0;
    rc = pixsel_fetch(c, p, sub ? (int)(unsigned char)(255 * quotia) : -1, map_out, &kept, (int)numHave); if (rc) return rc;
    *numHaveSub = sub ? kept : (int)numHave;
    *currentPotential = ideal;
    return NALO_OK;
}

int nalo_pixsel_make_maps_lidar(nalo_ctx* c, int slot, float thFactor, int currentPotential, float* map_out, int* numHave) {
    int rc = pixsel_check(c, slot, "nalo_pixsel_make_maps_lidar", true); if (rc) return rc;
    if (currentPotential < 1 || !numHave) return fail(c, NALO_ERR_ARG, "nalo_pixsel_make_maps_lidar: bad argument");
    NALO_HIP(c, hipSetDevice(c->device));
    PixSel* p = nullptr;
    rc = pixsel_state(c, &p); if (rc) return rc;
    ProfScope ps(c, "pixsel");
    if (p->hist_slot != slot) { rc = pixsel_hists(c, p, slot); if (rc) return rc; }
    int n[3];
    rc = pixsel_select_dev(c, p, slot, currentPotential, thFactor, n); if (rc) return rc;
    const int npx = c->w * c->h;
    NALO_HIP(c, hipMemsetAsync(p->cnt.p + 6, 0, (2 + 257) * sizeof(int), c->stream));
    pixsel_mask_hist_kernel<<<std::min((npx + 255) / 256, 512), 256, 0, c->stream>>>(c->slots[slot].mask, npx, p->cnt.p + 8);
    pixsel_fuse_kernel<<<(npx + 255) / 256, 256, 0, c->stream>>>(c->slots[slot].mask, p->draws.p, npx, p->map.p, p->cnt.p);
    rc = pixsel_fetch(c, p, -1, map_out, nullptr); if (rc) return rc;
    *numHave = p->host[6] + p->host[7];                         // m[0] + m[1] + m[2] with m[2] = 0 (:313)
    return NALO_OK;
}

int nalo_pixsel_get_selected(nalo_ctx* c, int cap, int* idx, uint8_t* status, int* n) {
    if (!c || !n || cap < 0 || (cap > 0 && (!idx || !status))) return fail(c, NALO_ERR_ARG, "nalo_pixsel_get_selected: bad argument");
    if (!c->pixsel || !c->pixsel->have_map) return fail(c, NALO_ERR_STATE, "nalo_pixsel_get_selected: no selection has been made");
    const PixSel* p = c->pixsel;
    *n = p->list_n;
    for (int i = 0; i < std::min(cap, p->list_n); ++i) { const int e = p->host[16 + i]; idx[i] = e & 0x0FFFFFFF; status[i] = (uint8_t)(e >> 28); }
    return NALO_OK;
}

}  // extern "C"
