// a1 — FrameHessian::makeImages (reference src/FullSystem/HessianBlocks.cpp:127-190) on gfx950.
// Level l>0: 2x2 box mean of level l-1 (:158-165). Every level: central differences over the FLAT pixel index
// in [w, w(h-1)) (:168-181) — the x=0 / x=w-1 columns difference across row ends exactly as the reference
// does; rows 0 and h-1 (uninitialised heap in the reference) are zero here.
// Output texel = float4 {I, dx, dy, 0}: one aligned 16-byte load per bilinear tap in the gather kernels.
// Bandwidth-bound streaming kernels: 16 B/lane stores, 256-thread blocks, grid-stride.
#include "nalo_internal.h"

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

// All levels in TWO launches (the per-level version was 2L-1 dependent launches of a few microseconds each: 56 us per KITTI frame, now ~12).
struct PyrLevels {
    float* I[NALO_MAX_LEVELS]; float4* dI[NALO_MAX_LEVELS]; float* absg[NALO_MAX_LEVELS];
    int wl[NALO_MAX_LEVELS], hl[NALO_MAX_LEVELS], blk0[NALO_MAX_LEVELS + 1], L;
};
// Box pyramid of every level from one pass over level 0: a block owns a 32x32 level-0 tile = 16x16 level-1 pixels and walks up through LDS
// (8x8, 4x4, 2x2, 1). Each parent is 0.25f * (((p00 + p10) + p01) + p11) of its children: the values of the level-by-level loop bit for bit.
__global__ __launch_bounds__(256) void pyr_down_all_kernel(PyrLevels P) {
    __shared__ float sI[2][16 * 16];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int tiles_x = (P.wl[1] + 15) / 16;
    const int bx = blockIdx.x % tiles_x, by = blockIdx.x / tiles_x;
    {
        const int x = bx * 16 + tx, y = by * 16 + ty, w0 = P.wl[0];
        float v = 0.f;
        if (x < P.wl[1] && y < P.hl[1]) {
            const float2 a = *reinterpret_cast<const float2*>(P.I[0] + 2 * x + 2 * y * w0);
            const float2 b = *reinterpret_cast<const float2*>(P.I[0] + 2 * x + 2 * y * w0 + w0);
            v = 0.25f * (((a.x + a.y) + b.x) + b.y);
            P.I[1][x + y * P.wl[1]] = v;
        }
        sI[0][ty * 16 + tx] = v;
    }
    int side = 16, cur = 0;
    for (int l = 2; l < P.L; ++l) {
        __syncthreads();
        const int ps = side; side >>= 1;
        if (side == 0) break;
        if (tx < side && ty < side) {
            const int x = bx * side + tx, y = by * side + ty, o = 2 * tx + 2 * ty * ps;
            const float v = 0.25f * (((sI[cur][o] + sI[cur][o + 1]) + sI[cur][o + ps]) + sI[cur][o + ps + 1]);
            if (x < P.wl[l] && y < P.hl[l]) P.I[l][x + y * P.wl[l]] = v;
            sI[cur ^ 1][ty * side + tx] = v;
        }
        cur ^= 1;
    }
}
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

// Raw-frame ingest: PhotometricUndistorter::processFrame (reference src/util/Undistort.cpp:214-251) + the remap loop of Undistort::undistort (:435-530) +
// the INTER_NEAREST resizes of Undistort::undistort_mask (:385-433; IOWrapper/OpenCV/ImageRW_OpenCV.cpp:55-85) in ONE pass over the rectified image:
// the 8- or 16-bit sensor frame crosses PCIe as it is (1-2 B/px instead of the 4 B/px float image the host path uploads) and
//   data[p] = G[raw[p]] * vignetteMapInv[p]      (photometricCalibration 2; 1: no vignette; disabled: factor * raw[p])
// is evaluated at the (up to) four taps of every output pixel instead of being materialised: out = bilinear(data, remapX, remapY), 0 where remapX < 0
// (passthrough: out = data). Each product and the weighted sum are rounded as the reference's float code does (this file: -ffp-contract=off).
struct IngestParams {
    const void* raw; int bpp;                       // 1 or 2 bytes per pixel
    int wOrg, hOrg, w, h;
    const float *G, *vinv, *remapX, *remapY;        // G [GDepth]; vinv [wOrg*hOrg] or null; remap [w*h] or null (passthrough)
    int photometric;                                // 0: data = factor * raw, 1: G only, 2: G * vignetteMapInv
    float factor;
    const uint8_t *mask_org, *bgr_org;              // [wOrg*hOrg], [wOrg*hOrg*3] or null
    float* out_I; float* out_mask; uint8_t* out_bgr;
    double ifx, ify;                                // cv::resize: 1 / ((double)w / wOrg)
};
__device__ __forceinline__ float ingest_tap(const IngestParams& P, int p) {
    const unsigned v = P.bpp == 1 ? (unsigned)reinterpret_cast<const uint8_t*>(P.raw)[p] : (unsigned)reinterpret_cast<const uint16_t*>(P.raw)[p];
    if (P.photometric == 0) return P.factor * (float)v;
    float d = P.G[v];
    if (P.photometric == 2) d *= P.vinv[p];
    return d;
}
__global__ __launch_bounds__(256) void ingest_kernel(IngestParams P) {
    const int n = P.w * P.h;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += gridDim.x * blockDim.x) {
        float o;
        if (!P.remapX) o = ingest_tap(P, idx);
        else {
            float xx = P.remapX[idx], yy = P.remapY[idx];
            if (xx < 0) o = 0.f;
            else {
                const int xxi = (int)xx, yyi = (int)yy;
                xx -= xxi; yy -= yyi;
                const float xxyy = xx * yy;
                const int p = xxi + yyi * P.wOrg;
                o = xxyy * ingest_tap(P, p + 1 + P.wOrg) + (yy - xxyy) * ingest_tap(P, p + P.wOrg) + (xx - xxyy) * ingest_tap(P, p + 1) + (1 - xx - yy + xxyy) * ingest_tap(P, p);
            }
        }
        P.out_I[idx] = o;
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
int ingest_launch(nalo_ctx* c, hipStream_t st, const void* raw, int bpp, int wOrg, int hOrg, const float* G, const float* vinv, const float* remapX, const float* remapY, int photometric,
                  float factor, const uint8_t* mask_org, const uint8_t* bgr_org, float* out_I, float* out_mask, uint8_t* out_bgr) {
    IngestParams P;
    P.raw = raw; P.bpp = bpp; P.wOrg = wOrg; P.hOrg = hOrg; P.w = c->w; P.h = c->h; P.G = G; P.vinv = vinv; P.remapX = remapX; P.remapY = remapY; P.photometric = photometric;
    P.factor = factor; P.mask_org = mask_org; P.bgr_org = bgr_org; P.out_I = out_I; P.out_mask = out_mask; P.out_bgr = out_bgr;
    P.ifx = 1.0 / ((double)c->w / wOrg); P.ify = 1.0 / ((double)c->h / hOrg);
    const int n = c->w * c->h;
    ProfScope ps(c, "ingest");
    ingest_kernel<<<(n + 255) / 256, 256, 0, st>>>(P);                  // one pixel per lane, consecutive workgroups on consecutive memory (scripts/ubench/copy.hip)
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
    ProfScope ps(c, "pyramid");
    PyrLevels P;
    P.L = c->levels;
    int nb = 0;
    for (int l = 0; l < c->levels; ++l) {
        P.I[l] = s.I[l]; P.dI[l] = s.dI[l]; P.absg[l] = s.absg[l]; P.wl[l] = c->wl[l]; P.hl[l] = c->hl[l];
        P.blk0[l] = nb; nb += (c->wl[l] * c->hl[l] + 255) / 256;
    }
    P.blk0[c->levels] = nb;
    // the hierarchical pass needs even parents all the way up (true for DSO pyramids: a level is only added while w and h are even) and <= 6 levels
    bool fused = c->levels >= 2 && c->levels <= 6;
    for (int l = 1; l < c->levels && fused; ++l) fused = (c->wl[l - 1] % 2 == 0) && (c->hl[l - 1] % 2 == 0);
    if (fused) pyr_down_all_kernel<<<((c->wl[1] + 15) / 16) * ((c->hl[1] + 15) / 16), 256, 0, c->stream>>>(P);
    else for (int l = 1; l < c->levels; ++l) {
        const int n = c->wl[l] * c->hl[l];
        pyr_down_kernel<<<std::min((n + 255) / 256, 2048), 256, 0, c->stream>>>(s.I[l - 1], s.I[l], c->wl[l], c->hl[l], c->wl[l - 1]);
    }
    pyr_grad_all_kernel<<<nb, 256, 0, c->stream>>>(P, gammaB_dev);
    NALO_HIP(c, hipGetLastError());
    return NALO_OK;
}

}  // namespace nalo
