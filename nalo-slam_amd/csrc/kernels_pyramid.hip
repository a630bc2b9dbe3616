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

int pyramid_build(nalo_ctx* c, FrameSlot& s, const float* gammaB_dev) {
    ProfScope ps(c, "pyramid");
    for (int l = 0; l < c->levels; ++l) {
        const int wl = c->wl[l], hl = c->hl[l], n = wl * hl;
        const int grid = std::min((n + 255) / 256, 2048);
        if (l > 0) pyr_down_kernel<<<grid, 256, 0, c->stream>>>(s.I[l - 1], s.I[l], wl, hl, c->wl[l - 1]);
        pyr_grad_kernel<<<grid, 256, 0, c->stream>>>(s.I[l], s.dI[l], s.absg[l], gammaB_dev, wl, hl);
    }
    NALO_HIP(c, hipGetLastError());
    return NALO_OK;
}

}  // namespace nalo
