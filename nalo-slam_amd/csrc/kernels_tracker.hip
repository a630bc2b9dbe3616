// Front-end tracker kernels for gfx950.
//   a2  makeCoarseDepthL0 steps 1-5   (reference src/FullSystem/CoarseTracker.cpp:382-538)
//   a3  calcRes                        (CoarseTracker.cpp:891-1049)      } fused into ONE pass: no buf_warped_*
//   a4  calcGSSSE + Accumulator9       (CoarseTracker.cpp:828-885,       } round trip through memory
//                                       OptimizationBackend/MatrixAccumulators.h:1091-1166)
// Reduction: per-lane fp32 accumulators over a grid-stride loop -> quad DPP adds -> 64 LDS rows per block ->
// column sums -> one fp32 partial per block -> fp64 finish kernel (deterministic: no float atomics on sums).
#include "nalo_internal.h"
#include "reduce.h"

namespace nalo {

// ------------------------------------------------------------------------------------------------ a3 + a4
struct TrkEvalParams {
    const float *u, *v, *id, *col;
    const float4* dI;
    int n, wl, hl, lvl;
    float fx, fy, cx, cy;
    float RKi[9], t[3], Ki[9];
    float affa, affb, b0, cutoff, maxEnergy;
};
constexpr int kTrkVals = 52;     // 45 upper-tri H entries + E, numTermsInE, numSaturated, numTermsInWarped, sT, sRT, sNum

__device__ __forceinline__ float4 bilinear4(const float4* __restrict__ img, float x, float y, int width) {
    // getInterpolatedElement33 (util/globalFuncs.h:75-89) on 16-byte texels
    const int ix = (int)x, iy = (int)y;
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

__global__ __launch_bounds__(256) void trk_eval_kernel(TrkEvalParams P, float* __restrict__ partial) {
    __shared__ float smem[64 * (kTrkVals + 1)];
    float acc[kTrkVals];
#pragma unroll
    for (int k = 0; k < kTrkVals; ++k) acc[k] = 0.f;
    const float wlm3 = (float)(P.wl - 3), hlm3 = (float)(P.hl - 3);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P.n; i += gridDim.x * blockDim.x) {
        const float id = P.id[i], x = P.u[i], y = P.v[i];
        const float pt0 = P.RKi[0] * x + P.RKi[1] * y + P.RKi[2] + P.t[0] * id;
        const float pt1 = P.RKi[3] * x + P.RKi[4] * y + P.RKi[5] + P.t[1] * id;
        const float pt2 = P.RKi[6] * x + P.RKi[7] * y + P.RKi[8] + P.t[2] * id;
        const float u = pt0 / pt2, v = pt1 / pt2;
        const float Ku = P.fx * u + P.cx, Kv = P.fy * v + P.cy;
        const float new_idepth = id / pt2;
        if (P.lvl == 0 && (i & 31) == 0) {                       // flow indicators, CoarseTracker.cpp:948-979
            const float a0 = P.Ki[0] * x + P.Ki[1] * y + P.Ki[2], a1 = P.Ki[3] * x + P.Ki[4] * y + P.Ki[5], a2 = P.Ki[6] * x + P.Ki[7] * y + P.Ki[8];
            const float T2 = a2 + P.t[2] * id, U2 = a2 - P.t[2] * id, r2 = P.RKi[6] * x + P.RKi[7] * y + P.RKi[8] - P.t[2] * id;
            const float KuT = P.fx * ((a0 + P.t[0] * id) / T2) + P.cx, KvT = P.fy * ((a1 + P.t[1] * id) / T2) + P.cy;
            const float KuT2 = P.fx * ((a0 - P.t[0] * id) / U2) + P.cx, KvT2 = P.fy * ((a1 - P.t[1] * id) / U2) + P.cy;
            const float Ku3 = P.fx * ((P.RKi[0] * x + P.RKi[1] * y + P.RKi[2] - P.t[0] * id) / r2) + P.cx;
            const float Kv3 = P.fy * ((P.RKi[3] * x + P.RKi[4] * y + P.RKi[5] - P.t[1] * id) / r2) + P.cy;
            acc[49] += (KuT - x) * (KuT - x) + (KvT - y) * (KvT - y);
            acc[49] += (KuT2 - x) * (KuT2 - x) + (KvT2 - y) * (KvT2 - y);
            acc[50] += (Ku - x) * (Ku - x) + (Kv - y) * (Kv - y);
            acc[50] += (Ku3 - x) * (Ku3 - x) + (Kv3 - y) * (Kv3 - y);
            acc[51] += 2.f;
        }
        if (!(Ku > 2.f && Kv > 2.f && Ku < wlm3 && Kv < hlm3 && new_idepth > 0.f)) continue;      // :981
        const float refColor = P.col[i];
        const float4 hit = bilinear4(P.dI, Ku, Kv, P.wl);
        if (!isfinite(hit.x)) continue;
        const float residual = hit.x - (P.affa * refColor + P.affb);
        const float ar = fabsf(residual);
        const float hw = ar < kHuberTH ? 1.f : kHuberTH / ar;
        acc[46] += 1.f;                                          // numTermsInE
        if (ar > P.cutoff) { acc[45] += P.maxEnergy; acc[47] += 1.f; }
        else {
            acc[45] += hw * residual * residual * (2.f - hw);
            acc[48] += 1.f;                                      // numTermsInWarped
            const float dx = hit.y * P.fx, dy = hit.z * P.fy;
            float J[9];
            J[0] = new_idepth * dx;
            J[1] = new_idepth * dy;
            J[2] = -(new_idepth * (u * dx + v * dy));
            J[3] = -(u * v * dx + dy * (1.f + v * v));
            J[4] = u * v * dy + dx * (1.f + u * u);
            J[5] = u * dy - v * dx;
            J[6] = P.affa * (P.b0 - refColor);
            J[7] = -1.f;
            J[8] = residual;
#pragma unroll
            for (int r = 0; r < 9; ++r) {                    // constant indices after unrolling: acc[] stays in VGPRs (a running `k++` index sent it to scratch)
                const float Jw = J[r] * hw;
#pragma unroll
                for (int c2 = r; c2 < 9; ++c2) acc[r * 9 - r * (r - 1) / 2 + (c2 - r)] += Jw * J[c2];
            }
        }
    }
    block_reduce_cols<kTrkVals, 256>(acc, smem, partial + (size_t)blockIdx.x * 64);
}

// fp64 finish: out[j] = sum_b partial[b][j]; written straight into host-mapped pinned memory
// out = host-mapped pinned memory; out[63] carries the sequence number of this evaluation, published after the data with
// system-scope fences so the host can poll it instead of paying a stream synchronisation per LM iteration
__global__ __launch_bounds__(1024) void trk_finish_kernel(const float* __restrict__ partial, int nblocks, double* __restrict__ out, double seq) {
    __shared__ double part[16][64];
    const int j = threadIdx.x & 63, g = threadIdx.x >> 6;
    double s = 0;
    if (j < kTrkVals) for (int b = g; b < nblocks; b += 16) s += (double)partial[(size_t)b * 64 + j];
    part[g][j] = s;
    __syncthreads();
    if (g == 0 && j < kTrkVals) { double t = 0; for (int k = 0; k < 16; ++k) t += part[k][j]; out[j] = t; __threadfence_system(); }
    __syncthreads();
    if (threadIdx.x == 0) { __hip_atomic_store(&out[63], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
}

int trk_eval_launch(nalo_ctx* c, int slot_new, int lvl, const float RKi[9], const float t[3], const float Ki[9],
                    float affa, float affb, float b0, float cutoff, float maxEnergy, double out64[64]) {
    TrkEvalParams P;
    P.u = c->pc_u[lvl].p; P.v = c->pc_v[lvl].p; P.id = c->pc_id[lvl].p; P.col = c->pc_col[lvl].p;
    P.dI = c->slots[slot_new].dI[lvl];
    P.n = c->pc_n[lvl]; P.wl = c->wl[lvl]; P.hl = c->hl[lvl]; P.lvl = lvl;
    P.fx = c->fx[lvl]; P.fy = c->fy[lvl]; P.cx = c->cx[lvl]; P.cy = c->cy[lvl];
    for (int i = 0; i < 9; ++i) { P.RKi[i] = RKi[i]; P.Ki[i] = Ki[i]; }
    for (int i = 0; i < 3; ++i) P.t[i] = t[i];
    P.affa = affa; P.affb = affb; P.b0 = b0; P.cutoff = cutoff; P.maxEnergy = maxEnergy;
    int nblocks = (P.n + 255) / 256;
    nblocks = nblocks < 1 ? 1 : (nblocks > 512 ? 512 : nblocks);
    NALO_HIP(c, c->trk_partial.reserve((size_t)2048 * 64));
    {
        ProfScope ps(c, "trk_eval");
        trk_eval_kernel<<<nblocks, 256, 0, c->stream>>>(P, c->trk_partial.p);
    }
    double* dout = nullptr;
    NALO_HIP(c, hipHostGetDevicePointer((void**)&dout, c->trk_out_host, 0));
    const double seq = (double)(++c->trk_seq);
    trk_finish_kernel<<<1, 1024, 0, c->stream>>>(c->trk_partial.p, nblocks, dout, seq);
    NALO_HIP(c, hipGetLastError());
    if (!poll_flag(c, &c->trk_out_host[63], seq)) return NALO_ERR_HIP;
    std::memcpy(out64, c->trk_out_host, sizeof(double) * kTrkVals);
    return NALO_OK;
}

// ------------------------------------------------------------------------------------------------ a2
// step 1 (CoarseTracker.cpp:388-405): weighted scatter. Two points on one pixel commute exactly; three or more
// (rare) make the fp32 sum order-dependent, as in any parallel scatter.
__global__ __launch_bounds__(256) void trk_scatter_kernel(const float* __restrict__ Ku, const float* __restrict__ Kv, const float* __restrict__ nid,
                                                          const float* __restrict__ HdiF, int n, int w0, int h0, float* __restrict__ idepth, float* __restrict__ wsum) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int u = (int)(Ku[i] + 0.5f), v = (int)(Kv[i] + 0.5f);
    if (u < 0 || v < 0 || u >= w0 || v >= h0) return;
    const float weight = sqrtf((float)(1e-3 / ((double)HdiF[i] + 1e-12)));
    atomicAdd(idepth + u + w0 * v, nid[i] * weight);
    atomicAdd(wsum + u + w0 * v, weight);
}
// step 2 (:408-433): 2x2 SUM pyramid
__global__ __launch_bounds__(256) void trk_sum_down_kernel(const float* __restrict__ idm, const float* __restrict__ wsm, float* __restrict__ id, float* __restrict__ ws,
                                                           int wl, int hl, int wlm1) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= wl * hl) return;
    const int y = i / wl, x = i - y * wl, b = 2 * x + 2 * y * wlm1;
    id[i] = idm[b] + idm[b + 1] + idm[b + wlm1] + idm[b + wlm1 + 1];
    ws[i] = wsm[b] + wsm[b + 1] + wsm[b + wlm1] + wsm[b + wlm1 + 1];
}
// steps 3/4 (:437-489): 1-px dilation, diagonal (levels 0,1) or axis (levels >= 2), flat-index neighbours.
// Reads idepth only where bak>0 and writes only where bak<=0: race-free in place, as the reference notes.
__global__ __launch_bounds__(256) void trk_dilate_kernel(float* __restrict__ id, float* __restrict__ ws, const float* __restrict__ bak, int wl, int hl, int diag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x + wl;
    if (i >= wl * hl - wl) return;
    if (bak[i] > 0) return;
    const int o0 = diag ? 1 + wl : 1, o1 = diag ? -1 - wl : -1, o2 = diag ? wl - 1 : wl, o3 = diag ? -wl + 1 : -wl;
    // the reference reads one element before/after the array at the first/last pixel of this range
    // (i-1-wl = -1, i+1+wl = w*h): those two taps are outside the image and are skipped here.
    const int npx = wl * hl;
    float sum = 0, num = 0, numn = 0;
#define NALO_TAP(o) { const int j = i + (o); if (j >= 0 && j < npx) { const float b = bak[j]; if (b > 0) { sum += id[j]; num += b; numn++; } } }
    NALO_TAP(o0) NALO_TAP(o1) NALO_TAP(o2) NALO_TAP(o3)
#undef NALO_TAP
    if (numn > 0) { id[i] = sum / numn; ws[i] = num / numn; }
}
// step 5 (:493-538): normalise + ordered (raster) compaction into pc_*. Pass 0 counts per block, pass 1 writes.
constexpr int kCompactChunk = 2048;          // interior elements per block (8 rounds of 256)
template <int WRITE>
__global__ __launch_bounds__(256) void trk_compact_kernel(float* __restrict__ id, float* __restrict__ ws, const float4* __restrict__ dIref, int wl, int hl,
                                                          int* __restrict__ counts, const int* __restrict__ offsets,
                                                          float* __restrict__ pu, float* __restrict__ pv, float* __restrict__ pid, float* __restrict__ pcol) {
    __shared__ int wave_cnt[4];
    __shared__ int running;
    const int iw = wl - 4, ih = hl - 4, total = iw * ih;            // interior y in [2,hl-2), x in [2,wl-2)
    const int base = blockIdx.x * kCompactChunk;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) running = WRITE ? offsets[blockIdx.x] : 0;
    __syncthreads();
    for (int r = 0; r < kCompactChunk / 256; ++r) {
        const int e = base + r * 256 + threadIdx.x;
        bool keep = false; float val_id = 0.f, col = 0.f; int x = 0, y = 0, i = 0;
        if (e < total) {
            y = e / iw; x = e - y * iw; y += 2; x += 2; i = x + y * wl;
            const float wsv = ws[i];
            if (wsv > 0) {
                val_id = id[i] / wsv; col = dIref[i].x;
                keep = isfinite(col) && (val_id > 0);
                if (WRITE) { id[i] = keep ? val_id : -1.f; if (keep) ws[i] = 1.f; }   // `continue` skips weightSums=1 (:524-528)
            } else if (WRITE) { id[i] = -1.f; ws[i] = 1.f; }
        }
        const unsigned long long m = __ballot(keep);
        const int rank = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wave] = __popcll(m);
        __syncthreads();
        int off = running;
        for (int k = 0; k < wave; ++k) off += wave_cnt[k];
        if (WRITE && keep) { const int o = off + rank; pu[o] = (float)x; pv[o] = (float)y; pid[o] = val_id; pcol[o] = col; }
        __syncthreads();
        if (threadIdx.x == 0) running += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        __syncthreads();
    }
    if (!WRITE && threadIdx.x == 0) counts[blockIdx.x] = running;
}
// exclusive scan of block counts (nb <= 65536) by one block; total appended at offsets[nb]
__global__ __launch_bounds__(1024) void scan_counts_kernel(const int* __restrict__ counts, int* __restrict__ offsets, int nb) {
    __shared__ int part[1024];
    const int per = (nb + 1023) / 1024, lo = threadIdx.x * per, hi = min(lo + per, nb);
    int s = 0;
    for (int i = lo; i < hi; ++i) s += counts[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) { int run = 0; for (int i = 0; i < 1024; ++i) { const int v = part[i]; part[i] = run; run += v; } offsets[nb] = run; }
    __syncthreads();
    int run = part[threadIdx.x];
    for (int i = lo; i < hi; ++i) { offsets[i] = run; run += counts[i]; }
}

int trk_build_ref(nalo_ctx* c, int n, const float* dKu, const float* dKv, const float* dId, const float* dHdi) {
    const int L = c->levels;
    for (int l = 0; l < L; ++l) {
        const size_t npx = (size_t)c->wl[l] * c->hl[l];
        NALO_HIP(c, c->trk_idepth[l].reserve(npx)); NALO_HIP(c, c->trk_wsum[l].reserve(npx)); NALO_HIP(c, c->trk_wbak[l].reserve(npx));
        NALO_HIP(c, c->pc_u[l].reserve(npx)); NALO_HIP(c, c->pc_v[l].reserve(npx)); NALO_HIP(c, c->pc_id[l].reserve(npx)); NALO_HIP(c, c->pc_col[l].reserve(npx));
    }
    const size_t n0 = (size_t)c->wl[0] * c->hl[0];
    NALO_HIP(c, hipMemsetAsync(c->trk_idepth[0].p, 0, n0 * 4, c->stream));
    NALO_HIP(c, hipMemsetAsync(c->trk_wsum[0].p, 0, n0 * 4, c->stream));
    if (n > 0) trk_scatter_kernel<<<(n + 255) / 256, 256, 0, c->stream>>>(dKu, dKv, dId, dHdi, n, c->wl[0], c->hl[0], c->trk_idepth[0].p, c->trk_wsum[0].p);
    for (int l = 1; l < L; ++l) {
        const int npx = c->wl[l] * c->hl[l];
        trk_sum_down_kernel<<<(npx + 255) / 256, 256, 0, c->stream>>>(c->trk_idepth[l - 1].p, c->trk_wsum[l - 1].p, c->trk_idepth[l].p, c->trk_wsum[l].p, c->wl[l], c->hl[l], c->wl[l - 1]);
    }
    for (int l = 0; l < L; ++l) {
        const int npx = c->wl[l] * c->hl[l];
        NALO_HIP(c, hipMemcpyAsync(c->trk_wbak[l].p, c->trk_wsum[l].p, (size_t)npx * 4, hipMemcpyDeviceToDevice, c->stream));
        const int cnt = npx - 2 * c->wl[l];
        if (cnt > 0) trk_dilate_kernel<<<(cnt + 255) / 256, 256, 0, c->stream>>>(c->trk_idepth[l].p, c->trk_wsum[l].p, c->trk_wbak[l].p, c->wl[l], c->hl[l], l < 2 ? 1 : 0);
    }
    size_t scan_off[NALO_MAX_LEVELS + 1] = {0};
    for (int l = 0; l < L; ++l) scan_off[l + 1] = scan_off[l] + 2 * (size_t)(((c->wl[l] - 4) * (c->hl[l] - 4) + kCompactChunk - 1) / kCompactChunk) + 2;
    NALO_HIP(c, c->scan_tmp.reserve(scan_off[L]));
    for (int l = 0; l < L; ++l) {
        const int total = (c->wl[l] - 4) * (c->hl[l] - 4);
        const int nb = (total + kCompactChunk - 1) / kCompactChunk;
        int* counts = c->scan_tmp.p + scan_off[l]; int* offsets = counts + nb;
        const float4* dIref = c->slots[c->slot_ref].dI[l];
        trk_compact_kernel<0><<<nb, 256, 0, c->stream>>>(c->trk_idepth[l].p, c->trk_wsum[l].p, dIref, c->wl[l], c->hl[l], counts, nullptr, nullptr, nullptr, nullptr, nullptr);
        scan_counts_kernel<<<1, 1024, 0, c->stream>>>(counts, offsets, nb);
        trk_compact_kernel<1><<<nb, 256, 0, c->stream>>>(c->trk_idepth[l].p, c->trk_wsum[l].p, dIref, c->wl[l], c->hl[l], nullptr, offsets, c->pc_u[l].p, c->pc_v[l].p, c->pc_id[l].p, c->pc_col[l].p);
        NALO_HIP(c, hipMemcpyAsync(&c->pc_n[l], offsets + nb, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    }
    NALO_HIP(c, hipStreamSynchronize(c->stream));         // one synchronisation for all levels (pc_n is needed on the host)
    NALO_HIP(c, hipGetLastError());
    return NALO_OK;
}

}  // namespace nalo
