// Micro-benchmark for the texel-gather layout of ba_linearize: how fast can gfx950 serve 32 x 16-byte bilinear taps per residual
//   A: one residual per lane, the 8 pattern pixels in a loop (the current kernel's layout)
//   B: one pattern pixel per lane, 8 lanes per residual (neighbouring lanes touch neighbouring texels)
//   C: one bilinear TAP per lane, 4 lanes per residual (lanes 0,1 = the two texels of row iy: 32 contiguous bytes; 2,3 = row iy+1), 8 pixels in a loop
//   D: one (pattern pixel, tap) per lane, 32 lanes per residual: a residual is ONE load instruction of half a wave
// Points are Morton-sorted inside each host frame, like the BA window. Prints microseconds per pass and effective TB/s of taps.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__constant__ int pat[8][2] = {{0, -2}, {-1, -1}, {1, -1}, {-2, 0}, {0, 0}, {2, 0}, {-1, 1}, {0, 2}};

__global__ __launch_bounds__(256) void gatherA(const float4* __restrict__ img, const float2* __restrict__ uv, int n, int w, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float2 p = uv[i];
    float acc = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float x = p.x + pat[k][0] * 1.1f, y = p.y + pat[k][1] * 1.1f;
        const int ix = (int)x, iy = (int)y;
        const float4* b = img + ix + iy * w;
        const float4 a0 = b[0], a1 = b[1], a2 = b[w], a3 = b[w + 1];
        acc += a0.x + a1.y + a2.z + a3.x;
    }
    out[i] = acc;
}
__global__ __launch_bounds__(256) void gatherB(const float4* __restrict__ img, const float2* __restrict__ uv, int n, int w, float* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = t >> 3, k = t & 7;
    if (i >= n) return;
    const float2 p = uv[i];
    const float x = p.x + pat[k][0] * 1.1f, y = p.y + pat[k][1] * 1.1f;
    const int ix = (int)x, iy = (int)y;
    const float4* b = img + ix + iy * w;
    const float4 a0 = b[0], a1 = b[1], a2 = b[w], a3 = b[w + 1];
    float acc = a0.x + a1.y + a2.z + a3.x;
    acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2); acc += __shfl_xor(acc, 4);
    if (k == 0) out[i] = acc;
}
__global__ __launch_bounds__(256) void gatherC(const float4* __restrict__ img, const float2* __restrict__ uv, int n, int w, float* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = t >> 2, q = t & 3;
    if (i >= n) return;
    const float2 p = uv[i];
    float acc = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float x = p.x + pat[k][0] * 1.1f, y = p.y + pat[k][1] * 1.1f;
        const int ix = (int)x, iy = (int)y;
        const float4 a = img[ix + (q & 1) + (iy + (q >> 1)) * w];
        acc += a.x + a.y;
    }
    acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2);
    if (q == 0) out[i] = acc;
}
__global__ __launch_bounds__(256) void gatherD(const float4* __restrict__ img, const float2* __restrict__ uv, int n, int w, float* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = t >> 5, k = (t >> 2) & 7, q = t & 3;
    if (i >= n) return;
    const float2 p = uv[i];
    const float x = p.x + pat[k][0] * 1.1f, y = p.y + pat[k][1] * 1.1f;
    const int ix = (int)x, iy = (int)y;
    const float4 a = img[ix + (q & 1) + (iy + (q >> 1)) * w];
    float acc = a.x + a.y;
    acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2); acc += __shfl_xor(acc, 4); acc += __shfl_xor(acc, 8); acc += __shfl_xor(acc, 16);
    if ((t & 31) == 0) out[i] = acc;
}
// E / F: 64-byte FOOTPRINT RECORDS — for every pixel the four {I,dx,dy} texels of its bilinear footprint stored together (4 x 16 B, one cache line half):
//   E: quad-cooperative like C, lane q loads 16-byte chunk q of the pixel's record: the quad reads ONE contiguous 64-byte block (1 line instead of 2 half rows)
//   F: one residual per lane like A, its four taps are four 16-byte loads from ONE record (same line)
__global__ __launch_bounds__(256) void gatherE(const float4* __restrict__ rec, const float2* __restrict__ uv, int n, int w, float* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = t >> 2, q = t & 3;
    if (i >= n) return;
    const float2 p = uv[i];
    float acc = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float x = p.x + pat[k][0] * 1.1f, y = p.y + pat[k][1] * 1.1f;
        const int ix = (int)x, iy = (int)y;
        const float4 a = rec[(size_t)(ix + iy * w) * 4 + q];
        acc += a.x + a.y;
    }
    acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2);
    if (q == 0) out[i] = acc;
}
__global__ __launch_bounds__(256) void gatherF(const float4* __restrict__ rec, const float2* __restrict__ uv, int n, int w, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float2 p = uv[i];
    float acc = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float x = p.x + pat[k][0] * 1.1f, y = p.y + pat[k][1] * 1.1f;
        const int ix = (int)x, iy = (int)y;
        const float4* b = rec + (size_t)(ix + iy * w) * 4;
        const float4 a0 = b[0], a1 = b[1], a2 = b[2], a3 = b[3];
        acc += a0.x + a1.y + a2.z + a3.x;
    }
    out[i] = acc;
}
// G: layout C (tap per lane, quad-cooperative) on a TILED image: 4x2-texel tiles of 128 bytes (one cache line holds a 4 wide x 2 high block instead of 8
// texels of one row), so a 2x2 bilinear footprint touches 1-2 lines more often and a 6x6 pattern footprint ~7.9 lines instead of ~9.8
__device__ __forceinline__ size_t tiled_index(int x, int y, int wt) { return ((size_t)(y >> 1) * wt + (x >> 2)) * 8 + ((y & 1) << 2) + (x & 3); }
__global__ __launch_bounds__(256) void gatherG(const float4* __restrict__ img, const float2* __restrict__ uv, int n, int w, float* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = t >> 2, q = t & 3;
    if (i >= n) return;
    const float2 p = uv[i];
    const int wt = w >> 2;
    float acc = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float x = p.x + pat[k][0] * 1.1f, y = p.y + pat[k][1] * 1.1f;
        const int ix = (int)x + (q & 1), iy = (int)y + (q >> 1);
        const float4 a = img[tiled_index(ix, iy, wt)];
        acc += a.x + a.y;
    }
    acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2);
    if (q == 0) out[i] = acc;
}
// H: the same on 2x4 tiles (2 wide x 4 high)
__device__ __forceinline__ size_t tiled_index24(int x, int y, int wt) { return ((size_t)(y >> 2) * wt + (x >> 1)) * 8 + ((y & 3) << 1) + (x & 1); }
__global__ __launch_bounds__(256) void gatherH(const float4* __restrict__ img, const float2* __restrict__ uv, int n, int w, float* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = t >> 2, q = t & 3;
    if (i >= n) return;
    const float2 p = uv[i];
    const int wt = w >> 1;
    float acc = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float x = p.x + pat[k][0] * 1.1f, y = p.y + pat[k][1] * 1.1f;
        const int ix = (int)x + (q & 1), iy = (int)y + (q >> 1);
        const float4 a = img[tiled_index24(ix, iy, wt)];
        acc += a.x + a.y;
    }
    acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2);
    if (q == 0) out[i] = acc;
}
// I: 4x2 tiles as in G, the tiles themselves in 8x8-tile super-blocks (32 x 16 pixels = 8 KB contiguous): locality beyond the line (pages, L2 sets)
__device__ __forceinline__ size_t tiled_index_sb(int x, int y, int wsb) {
    const int tx = x >> 2, ty = y >> 1;
    return (((size_t)(ty >> 3) * wsb + (tx >> 3)) * 64 + ((ty & 7) << 3) + (tx & 7)) * 8 + ((y & 1) << 2) + (x & 3);
}
__global__ __launch_bounds__(256) void gatherI(const float4* __restrict__ img, const float2* __restrict__ uv, int n, int w, float* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = t >> 2, q = t & 3;
    if (i >= n) return;
    const float2 p = uv[i];
    const int wsb = (w >> 2) >> 3;
    float acc = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float x = p.x + pat[k][0] * 1.1f, y = p.y + pat[k][1] * 1.1f;
        const int ix = (int)x + (q & 1), iy = (int)y + (q >> 1);
        const float4 a = img[tiled_index_sb(ix, iy, wsb)];
        acc += a.x + a.y;
    }
    acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2);
    if (q == 0) out[i] = acc;
}
// J: 12-byte texels {I,dx,dy} (the 16-byte texel wastes a quarter of every cache line on padding) in 5x2-texel tiles padded to 128 bytes: a line holds 10 useful
// texels instead of 8, a 6x6 footprint ~7.0 lines instead of ~7.9. Loads are global_load_dwordx3 (4-byte aligned). x / 5 by multiply-shift.
struct f3 { float x, y, z; };
__device__ __forceinline__ size_t tiled_index52(int x, int y, int wt) { const int tx = (x * 52429) >> 18; return ((size_t)(y >> 1) * wt + tx) * 32 + ((y & 1) * 5 + (x - tx * 5)) * 3; }   // in floats
__global__ __launch_bounds__(256) void gatherJ(const float4* __restrict__ img, const float2* __restrict__ uv, int n, int w, float* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = t >> 2, q = t & 3;
    if (i >= n) return;
    const float2 p = uv[i];
    const int wt = (w + 4) / 5;
    const float* base = reinterpret_cast<const float*>(img);
    float acc = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float x = p.x + pat[k][0] * 1.1f, y = p.y + pat[k][1] * 1.1f;
        const int ix = (int)x + (q & 1), iy = (int)y + (q >> 1);
        const f3 a = *reinterpret_cast<const f3*>(base + tiled_index52(ix, iy, wt));
        acc += a.x + a.y;
    }
    acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2);
    if (q == 0) out[i] = acc;
}
// K: 12-byte texels, row major (no tiles): how much of J is the texel size alone
__global__ __launch_bounds__(256) void gatherK(const float4* __restrict__ img, const float2* __restrict__ uv, int n, int w, float* __restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = t >> 2, q = t & 3;
    if (i >= n) return;
    const float2 p = uv[i];
    const float* base = reinterpret_cast<const float*>(img);
    float acc = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float x = p.x + pat[k][0] * 1.1f, y = p.y + pat[k][1] * 1.1f;
        const int ix = (int)x + (q & 1), iy = (int)y + (q >> 1);
        const f3 a = *reinterpret_cast<const f3*>(base + ((size_t)iy * w + ix) * 3);
        acc += a.x + a.y;
    }
    acc += __shfl_xor(acc, 1); acc += __shfl_xor(acc, 2);
    if (q == 0) out[i] = acc;
}
static uint32_t part1by1(uint32_t x) { x &= 0xffff; x = (x | (x << 8)) & 0x00FF00FF; x = (x | (x << 4)) & 0x0F0F0F0F; x = (x | (x << 2)) & 0x33333333; x = (x | (x << 1)) & 0x55555555; return x; }
int main(int argc, char** argv) {
    const int w = 1920, h = 1072, W = 8, P = argc > 1 ? atoi(argv[1]) : 250000;
    const int per = P / W;
    std::vector<float4*> imgs(W);
    for (int f = 0; f < W; ++f) { CK(hipMalloc(&imgs[f], (size_t)w * h * 16)); CK(hipMemset(imgs[f], 0, (size_t)w * h * 16)); }
    std::vector<float4*> recs(W);
    for (int f = 0; f < W; ++f) { CK(hipMalloc(&recs[f], (size_t)w * h * 64)); CK(hipMemset(recs[f], 0, (size_t)w * h * 64)); }
    // residual list: for each target t, for each host h != t, the host's points (Morton order) reprojected with a small shift
    std::vector<float2> uv; std::vector<int> tgt_start(W + 1, 0);
    srand(1);
    std::vector<std::vector<float2>> pts(W);
    for (int f = 0; f < W; ++f) {
        std::vector<std::pair<uint32_t, float2>> v(per);
        for (auto& e : v) { const float x = 8 + (rand() / (float)RAND_MAX) * (w - 16), y = 8 + (rand() / (float)RAND_MAX) * (h - 16); e.second = make_float2(x, y); e.first = part1by1((uint32_t)x) | (part1by1((uint32_t)y) << 1); }
        std::sort(v.begin(), v.end(), [](auto& a, auto& b) { return a.first < b.first; });
        for (auto& e : v) pts[f].push_back(e.second);
    }
    std::vector<std::vector<float2>> per_target(W);
    for (int t = 0; t < W; ++t) for (int f = 0; f < W; ++f) if (f != t) for (auto p : pts[f]) per_target[t].push_back(make_float2(std::min(std::max(p.x + 3.3f * (t - f), 4.f), w - 5.f), p.y));
    float2* duv; float* dout;
    const size_t nmax = per_target[0].size();
    CK(hipMalloc(&duv, nmax * W * 8)); CK(hipMalloc(&dout, nmax * W * 4));
    for (int t = 0; t < W; ++t) CK(hipMemcpy(duv + t * nmax, per_target[t].data(), nmax * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int variant = 0; variant < 11; ++variant) {
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0));
            for (int t = 0; t < W; ++t) {
                const int n = (int)nmax;
                if (variant == 0) gatherA<<<(n + 255) / 256, 256>>>(imgs[t], duv + t * nmax, n, w, dout + t * nmax);
                else if (variant == 1) gatherB<<<(n * 8 + 255) / 256, 256>>>(imgs[t], duv + t * nmax, n, w, dout + t * nmax);
                else if (variant == 2) gatherC<<<(n * 4 + 255) / 256, 256>>>(imgs[t], duv + t * nmax, n, w, dout + t * nmax);
                else if (variant == 3) gatherD<<<(unsigned)(((size_t)n * 32 + 255) / 256), 256>>>(imgs[t], duv + t * nmax, n, w, dout + t * nmax);
                else if (variant == 4) gatherE<<<(n * 4 + 255) / 256, 256>>>(recs[t], duv + t * nmax, n, w, dout + t * nmax);
                else if (variant == 5) gatherF<<<(n + 255) / 256, 256>>>(recs[t], duv + t * nmax, n, w, dout + t * nmax);
                else if (variant == 6) gatherG<<<(n * 4 + 255) / 256, 256>>>(imgs[t], duv + t * nmax, n, w, dout + t * nmax);
                else if (variant == 7) gatherH<<<(n * 4 + 255) / 256, 256>>>(imgs[t], duv + t * nmax, n, w, dout + t * nmax);
                else if (variant == 8) gatherI<<<(n * 4 + 255) / 256, 256>>>(imgs[t], duv + t * nmax, n, w, dout + t * nmax);
                else if (variant == 9) gatherJ<<<(n * 4 + 255) / 256, 256>>>(imgs[t], duv + t * nmax, n, w, dout + t * nmax);
                else gatherK<<<(n * 4 + 255) / 256, 256>>>(imgs[t], duv + t * nmax, n, w, dout + t * nmax);
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
        }
        const double taps = (double)nmax * W * 32;
        printf("variant %c: %zu residuals, %.1f us, %.2f TB/s of 16-B taps, %.2f Gtaps/s\n", "ABCDEFGHIJK"[variant], nmax * W, best * 1e3, taps * 16 / (best * 1e-3) / 1e12, taps / (best * 1e-3) / 1e9);
    }
    // occupancy sweep of layout C: dynamic LDS per 256-thread block caps the blocks per CU (160 KB of LDS), i.e. waves per SIMD, with 8 loads in flight per lane:
    // does the gather rate depend on the loads in flight (latency bound) or not (bound by the miss path of the memory system)?
    for (int wps : {1, 2, 3, 4, 6, 8}) {
        const size_t lds = (size_t)(160 * 1024 / wps) - 2048;
        CK(hipFuncSetAttribute(reinterpret_cast<const void*>(gatherC), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        float best = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0));
            for (int t = 0; t < W; ++t) { const int n = (int)nmax; gatherC<<<(n * 4 + 255) / 256, 256, lds>>>(imgs[t], duv + t * nmax, n, w, dout + t * nmax); }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
        }
        printf("layout C at %d waves/SIMD (8 loads in flight per lane): %.1f us\n", wps, best * 1e3);
    }
    return 0;
}
