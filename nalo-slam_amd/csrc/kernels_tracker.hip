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
    int i0, i1;                     // this rank's share [i0, i1) of the level's points (all of them unless the tracker is sharded, nalo_trk_set_shard)
    float fx, fy, cx, cy;
    float RKi[9], t[3], Ki[9];
    float affa, affb, b0, cutoff, maxEnergy;
};
typedef float trk_f4 __attribute__((ext_vector_type(4)));
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

#ifndef NALO_TRK_EVAL_NT
#define NALO_TRK_EVAL_NT 512
#endif
constexpr int kTrkEvalNT = NALO_TRK_EVAL_NT;      // lanes per workgroup of the evaluation. 256 -> 512 (round 4, same box): half as many grid-stride rounds at 250 k points and half as many
                                                  // partial rows for the last workgroup: 21.0 -> 19.0 us at 250 k points, 40.0 -> 35.7 us at full density (1920x1072)
constexpr int kTrkEvalNG = kTrkEvalNT / 13;        // lane groups of the last workgroup's sum: 13 lanes x 16 bytes = one block's 52 partials
__global__ __launch_bounds__(kTrkEvalNT) void trk_eval_kernel(TrkEvalParams P, float* __restrict__ partial, unsigned* __restrict__ ticket, double* __restrict__ out, double seq) {
    __shared__ __attribute__((aligned(16))) float smem[(kTrkEvalNT / 4) * (kTrkVals + 1)];
    float acc[kTrkVals];
#pragma unroll
    for (int k = 0; k < kTrkVals; ++k) acc[k] = 0.f;
    const float wlm3 = (float)(P.wl - 3), hlm3 = (float)(P.hl - 3);
    for (int i = P.i0 + blockIdx.x * blockDim.x + threadIdx.x; i < P.i1; i += gridDim.x * blockDim.x) {
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
    // Round 4: the fp64 finish rides in this launch (it was a second, 10 us launch behind a 14 us one). A workgroup's 52 partials leave as
    // agent-scope words, acknowledged before its ticket (the idiom of ba_reduce_kernel's tail); the workgroup that draws the last ticket sums the partials in a fixed order (fp64) and
    // publishes them behind the sequence number the host polls.
    __shared__ float blk[64];
    block_reduce_cols<kTrkVals, kTrkEvalNT>(acc, smem, blk);
    if (threadIdx.x < kTrkVals) __hip_atomic_store(&partial[(size_t)blockIdx.x * 64 + threadIdx.x], blk[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __shared__ bool is_last;
    __syncthreads();
    if (threadIdx.x == 0) is_last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u;
    __syncthreads();
    if (!is_last) return;
    // 13 lanes x 16 bytes cover a block's 52 partials; kTrkEvalNG (19 of a 256-lane workgroup) such lane groups stride over the blocks with fourteen coherent (sc0 sc1: past this XCD's L2, where the
    // other XCDs' write-through stores are not seen) 16-byte loads in flight each - one relaxed atomic word per lane and trip was a chain of ~130 dependent
    // round trips (50 us). Fixed order: rows b = rg, rg + NG, ... in fp64 per group, then the groups in ascending order.
    double (*part)[kTrkVals] = reinterpret_cast<double (*)[kTrkVals]>(smem);       // NG x 52 doubles (7.9 KB of the 13.6 KB reduction buffer at 256 lanes)
    const int cq = threadIdx.x % 13, rg = threadIdx.x / 13, nblocks = (int)gridDim.x;
    static_assert(kTrkEvalNG * kTrkVals * 2 <= (kTrkEvalNT / 4) * (kTrkVals + 1), "the groups' fp64 sums fit the reduction buffer");
    if (rg < kTrkEvalNG) {
        double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        for (int b0 = rg; b0 < nblocks; b0 += kTrkEvalNG * 14) {
            trk_f4 v[14];
#pragma unroll
            for (int k = 0; k < 14; ++k) {
                const int b2 = b0 + kTrkEvalNG * k;
                const float* q = partial + (size_t)(b2 < nblocks ? b2 : b0) * 64 + cq * 4;
                asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v[k]) : "v"(q) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]) : : "memory");
#pragma unroll
            for (int k = 0; k < 14; ++k) if (b0 + kTrkEvalNG * k < nblocks) { a0 += (double)v[k].x; a1 += (double)v[k].y; a2 += (double)v[k].z; a3 += (double)v[k].w; }
        }
        part[rg][cq * 4] = a0; part[rg][cq * 4 + 1] = a1; part[rg][cq * 4 + 2] = a2; part[rg][cq * 4 + 3] = a3;
    }
    __syncthreads();
    if (threadIdx.x < kTrkVals) { double t = 0; for (int k = 0; k < kTrkEvalNG; ++k) t += part[k][threadIdx.x]; __hip_atomic_store(&out[threadIdx.x], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                // re-armed for the next (stream-ordered) launch
        __hip_atomic_store(&out[63], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// out = host-mapped pinned memory; out[63] carries the sequence number of this evaluation, published after the data so the host can poll it instead of paying a
// stream synchronisation per LM iteration
// sharded tracker ("all-reduce of 45 + 6 floats per evaluation"): the evaluation's last workgroup leaves this rank's 52 fp64 sums in DEVICE memory (out = device buffer,
// seq unused), the caller's hook sums them over the ranks in place, and this kernel publishes the result the way the evaluation's tail does
__global__ __launch_bounds__(64) void trk_publish_kernel(const double* __restrict__ src, double* __restrict__ out, double seq) {
    if (threadIdx.x < kTrkVals) __hip_atomic_store(&out[threadIdx.x], src[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __syncthreads();
    if (threadIdx.x == 0) { __threadfence_system(); __hip_atomic_store(&out[63], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
}

int trk_eval_launch(nalo_ctx* c, int slot_new, int lvl, const float RKi[9], const float t[3], const float Ki[9],
                    float affa, float affb, float b0, float cutoff, float maxEnergy, double out64[64]) {
    TrkEvalParams P;
    P.u = c->pc_u[lvl].p; P.v = c->pc_v[lvl].p; P.id = c->pc_id[lvl].p; P.col = c->pc_col[lvl].p;
    P.dI = c->slots[slot_new].dI[lvl];
    P.n = c->pc_n[lvl]; P.wl = c->wl[lvl]; P.hl = c->hl[lvl]; P.lvl = lvl;
    const bool sharded = c->trk_world > 1 && c->trk_hook;
    P.i0 = sharded ? (int)((long long)P.n * c->trk_rank / c->trk_world) : 0;
    P.i1 = sharded ? (int)((long long)P.n * (c->trk_rank + 1) / c->trk_world) : P.n;
    P.fx = c->fx[lvl]; P.fy = c->fy[lvl]; P.cx = c->cx[lvl]; P.cy = c->cy[lvl];
    for (int i = 0; i < 9; ++i) { P.RKi[i] = RKi[i]; P.Ki[i] = Ki[i]; }
    for (int i = 0; i < 3; ++i) P.t[i] = t[i];
    P.affa = affa; P.affb = affb; P.b0 = b0; P.cutoff = cutoff; P.maxEnergy = maxEnergy;
    int nblocks = (P.i1 - P.i0 + kTrkEvalNT - 1) / kTrkEvalNT;
    nblocks = nblocks < 1 ? 1 : (nblocks > 512 ? 512 : nblocks);
    NALO_HIP(c, c->trk_partial.reserve((size_t)2048 * 64));
    if (!c->trk_ticket.p) { NALO_HIP(c, c->trk_ticket.reserve(4)); NALO_HIP(c, hipMemsetAsync(c->trk_ticket.p, 0, 16, c->stream)); }
    double* dout = nullptr;
    NALO_HIP(c, hipHostGetDevicePointer((void**)&dout, c->trk_out_host, 0));
    const double seq = (double)(++c->trk_seq);
    if (sharded) NALO_HIP(c, c->trk_shard_sums.reserve(64));
    {
        ProfScope ps(c, "trk_eval");
        trk_eval_kernel<<<nblocks, kTrkEvalNT, 0, c->stream>>>(P, c->trk_partial.p, c->trk_ticket.p, sharded ? c->trk_shard_sums.p : dout, sharded ? 0.0 : seq);
    }
    if (sharded) {
        // every rank evaluated its share: the 52 sums meet in the hook (in place, device memory), every rank then reads the same totals and runs the same LM step
        if (!c->trk_hook_stream_ordered) NALO_HIP(c, hipStreamSynchronize(c->stream));
        c->trk_hook(c->trk_hook_user, c->trk_shard_sums.p, kTrkVals);
        // the same latch as the BA's call_hook (host_ba.hip): a rank whose collective failed must not take an LM step on rank-local sums - the ranks would take
        // different steps and evaluation counts, and the next collectives would mismatch or hang
        if (c->xchg_failed) return NALO_ERR_HIP;                        // message already in c->err (nalo_ba_exchange_failed / the RCCL hooks)
        trk_publish_kernel<<<1, 64, 0, c->stream>>>(c->trk_shard_sums.p, dout, seq);
    }
    NALO_HIP(c, hipGetLastError());
    if (!poll_flag(c, &c->trk_out_host[63], seq)) return NALO_ERR_HIP;
    std::memcpy(out64, c->trk_out_host, sizeof(double) * kTrkVals);
    return NALO_OK;
}

// ------------------------------------------------------------------------------------------------ a2
// step 1 (CoarseTracker.cpp:388-405): weighted scatter. Two points on one pixel commute exactly in fp32; three or more (rare) would make the sum depend on
// the arrival order of the atomics while the reference adds them serially, in residual order. The scatter therefore also counts the hits per pixel, and
// trk_scatter_fix_kernel redoes the pixels with >= 3 hits in ascending residual index: the result is the serial loop's, bit for bit, run to run.
__global__ __launch_bounds__(256) void trk_scatter_kernel(const float* __restrict__ Ku, const float* __restrict__ Kv, const float* __restrict__ nid,
                                                          const float* __restrict__ HdiF, int n, int w0, int h0, float* __restrict__ idepth, float* __restrict__ wsum, int* __restrict__ cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int u = (int)(Ku[i] + 0.5f), v = (int)(Kv[i] + 0.5f);
    if (u < 0 || v < 0 || u >= w0 || v >= h0) return;
    const float weight = sqrtf((float)(1e-3 / ((double)HdiF[i] + 1e-12)));
    atomicAdd(idepth + u + w0 * v, nid[i] * weight);
    atomicAdd(wsum + u + w0 * v, weight);
    atomicAdd(cnt + u + w0 * v, 1);
}
// Pass 1 (trk_scatter_hot_kernel, the whole grid) lists the residuals whose pixel took three or more hits - in any order: the rounds below order by INDEX,
// not by list position; pass 2 (trk_scatter_fix_kernel, ONE workgroup) zeroes those pixels and adds the listed residuals back one per pixel and round, always
// the lowest remaining index first (the pixel's count word is reused as the "next index allowed" gate). Usually the list is empty and pass 2 is one read.
// (The list used to be built by the single workgroup itself, striding all n residuals: 530 us at n = 250 000.)
constexpr int kScatterFixCap = 4096;
template <int NT>
__device__ __forceinline__ void trk_scatter_fix_body(const float* __restrict__ Ku, const float* __restrict__ Kv, const float* __restrict__ nid, const float* __restrict__ HdiF,
                                                     int n, int w0, int h0, float* __restrict__ idepth, float* __restrict__ wsum, int* __restrict__ cnt, int* __restrict__ glist, int* list);
// Round 4: ONE launch for both passes (the fix pass was a launch of its own, 4.6 us for what is almost always one read): every workgroup lists its residuals on hot pixels
// (agent-scope stores, acknowledged before its ticket), the workgroup that draws the last ticket runs the fix.
__global__ __launch_bounds__(256) void trk_scatter_hot_kernel(const float* __restrict__ Ku, const float* __restrict__ Kv, const float* __restrict__ nid, const float* __restrict__ HdiF,
                                                              int n, int w0, int h0, float* __restrict__ idepth, float* __restrict__ wsum, int* __restrict__ cnt,
                                                              int* __restrict__ list /* [kScatterFixCap] + counter + ticket */) {
    __shared__ int slist[kScatterFixCap];
    __shared__ int is_last;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const int u = (int)(Ku[i] + 0.5f), v = (int)(Kv[i] + 0.5f);
        if (!(u < 0 || v < 0 || u >= w0 || v >= h0) && __hip_atomic_load(cnt + u + w0 * v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= 3) {       // written by the scatter's atomics: read at L2
            const int at = atomicAdd(list + kScatterFixCap, 1);
            if (at < kScatterFixCap) __hip_atomic_store(list + at, i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0) is_last = __hip_atomic_fetch_add((unsigned*)(list + kScatterFixCap + 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u;
    __syncthreads();
    if (!is_last) return;
    if (threadIdx.x == 0) __hip_atomic_store(list + kScatterFixCap + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ticket re-armed for the next (stream-ordered) call
    trk_scatter_fix_body<256>(Ku, Kv, nid, HdiF, n, w0, h0, idepth, wsum, cnt, list, slist);
}
template <int NT>
__device__ __forceinline__ void trk_scatter_fix_body(const float* __restrict__ Ku, const float* __restrict__ Kv, const float* __restrict__ nid, const float* __restrict__ HdiF,
                                                     int n, int w0, int h0, float* __restrict__ idepth, float* __restrict__ wsum, int* __restrict__ cnt, int* __restrict__ glist, int* list) {
    const int tid = threadIdx.x;
    const int total = __hip_atomic_load(glist + kScatterFixCap, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (total == 0) return;
    __syncthreads();
    if (tid == 0) __hip_atomic_store(glist + kScatterFixCap, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);          // re-armed for the next (stream-ordered) call
    for (int k = tid; k < total && k < kScatterFixCap; k += NT) list[k] = __hip_atomic_load(glist + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int m = total;
    if (m == 0 || m > kScatterFixCap) return;            // (more than 4096 colliding residuals — never seen —: the atomic sums stay)
    for (int k = tid; k < m; k += NT) {
        const int i = list[k], p = (int)(Ku[i] + 0.5f) + w0 * (int)(Kv[i] + 0.5f);
        // every access to the pixel's words below goes to L2 (atomics / agent-scope loads): the atomics of the scatter and of the gate bypass this CU's L1,
        // a plain load could return a stale line
        atomicExch(idepth + p, 0.f); atomicExch(wsum + p, 0.f); atomicExch(cnt + p, 0x7fffffff);       // gate = lowest listed index of the pixel, found next
    }
    __syncthreads();
    for (int k = tid; k < m; k += NT) { const int i = list[k]; atomicMin(cnt + (int)(Ku[i] + 0.5f) + w0 * (int)(Kv[i] + 0.5f), i); }
    __syncthreads();
    // rounds: a listed residual adds itself when the gate of its pixel shows its index, then passes the gate to the next listed index of that pixel
    for (int round = 0; round < m; ++round) {
        bool any = false;
        for (int k = tid; k < m; k += NT) {
            const int i = list[k];
            if (i < 0) continue;
            const int p = (int)(Ku[i] + 0.5f) + w0 * (int)(Kv[i] + 0.5f);
            if (__hip_atomic_load(cnt + p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == i) {
                const float weight = sqrtf((float)(1e-3 / ((double)HdiF[i] + 1e-12)));
                atomicAdd(idepth + p, __fmul_rn(nid[i], weight)); atomicAdd(wsum + p, weight);         // one adder per pixel and round: the order is the index order
                list[k] = -1 - i;                                                                   // done (kept negative so that the gate search can skip it)
            } else any = true;
        }
        __syncthreads();
        if (!__syncthreads_or(any)) break;
        // next gate per pixel: the lowest not-yet-added listed index
        for (int k = tid; k < m; k += NT) { const int i = list[k]; if (i < 0) { const int j = -1 - i, p = (int)(Ku[j] + 0.5f) + w0 * (int)(Kv[j] + 0.5f); atomicCAS(cnt + p, j, 0x7fffffff); } }
        __syncthreads();
        for (int k = tid; k < m; k += NT) { const int i = list[k]; if (i >= 0) atomicMin(cnt + (int)(Ku[i] + 0.5f) + w0 * (int)(Kv[i] + 0.5f), i); }
        __syncthreads();
    }
}
// All levels of steps 2-5 go through ONE launch per step (the per-level launches were ~30 dependent kernels of a few microseconds each).
struct TrkLevels {
    float *id[NALO_MAX_LEVELS], *ws[NALO_MAX_LEVELS], *wb[NALO_MAX_LEVELS];      // idepth, weight sums, dilated weight sums (output of step 3/4)
    const float4* dI[NALO_MAX_LEVELS];                                            // reference frame texels
    float *pu[NALO_MAX_LEVELS], *pv[NALO_MAX_LEVELS], *pid[NALO_MAX_LEVELS], *pcol[NALO_MAX_LEVELS];
    int wl[NALO_MAX_LEVELS], hl[NALO_MAX_LEVELS];
    int blk0[NALO_MAX_LEVELS + 1];                                                // first block of each level in a level-partitioned grid
    int scan0[NALO_MAX_LEVELS + 1];                                               // counts/offsets base per level inside scan_tmp ([nb counts | nb+1 offsets])
    int L;
};
__global__ __launch_bounds__(256) void trk_zero2_kernel(float* __restrict__ a, float* __restrict__ b, int* __restrict__ cnt, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { a[i] = 0.f; b[i] = 0.f; cnt[i] = 0; }
    if (i == 0) { cnt[n + kScatterFixCap] = 0; cnt[n + kScatterFixCap + 1] = 0; }   // the hot list's counter and the ticket of trk_scatter_hot_kernel
}
// step 2 (:408-433): 2x2 SUM pyramid, every level from one pass over level 0. A block owns a 32x32 level-0 tile = 16x16 level-1 pixels, and
// walks up through LDS (8x8, 4x4, 2x2, 1); each parent is a + b + c + d of its four children in the reference's order, so the values
// are those of the level-by-level loop bit for bit.
__global__ __launch_bounds__(256) void trk_sum_down_all_kernel(TrkLevels P) {
    __shared__ float sid[2][16 * 16], sws[2][16 * 16];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int tiles_x = (P.wl[1] + 15) / 16;
    const int bx = blockIdx.x % tiles_x, by = blockIdx.x / tiles_x;
    {
        const int x = bx * 16 + tx, y = by * 16 + ty, w0 = P.wl[0];
        float a = 0.f, b = 0.f;
        if (x < P.wl[1] && y < P.hl[1]) {
            const int o = 2 * x + 2 * y * w0;
            a = P.id[0][o] + P.id[0][o + 1] + P.id[0][o + w0] + P.id[0][o + w0 + 1];
            b = P.ws[0][o] + P.ws[0][o + 1] + P.ws[0][o + w0] + P.ws[0][o + w0 + 1];
            P.id[1][x + y * P.wl[1]] = a; P.ws[1][x + y * P.wl[1]] = b;
        }
        sid[0][ty * 16 + tx] = a; sws[0][ty * 16 + tx] = b;
    }
    int side = 16, cur = 0;
    for (int l = 2; l < P.L; ++l) {
        __syncthreads();
        const int ps = side; side >>= 1;
        if (side == 0) break;                                  // more than 6 levels would need a larger tile (NALO_MAX_LEVELS = 6)
        if (tx < side && ty < side) {
            const int x = bx * side + tx, y = by * side + ty, o = 2 * tx + 2 * ty * ps;
            const float a = sid[cur][o] + sid[cur][o + 1] + sid[cur][o + ps] + sid[cur][o + ps + 1];
            const float b = sws[cur][o] + sws[cur][o + 1] + sws[cur][o + ps] + sws[cur][o + ps + 1];
            if (x < P.wl[l] && y < P.hl[l]) { P.id[l][x + y * P.wl[l]] = a; P.ws[l][x + y * P.wl[l]] = b; }
            sid[cur ^ 1][ty * side + tx] = a; sws[cur ^ 1][ty * side + tx] = b;
        }
        cur ^= 1;
    }
}
// steps 3/4 (:437-489): 1-px dilation, diagonal (levels 0,1) or axis (levels >= 2), flat-index neighbours. The reference dilates in place
// against a backup copy of the weights; here the undilated weights ws stay read-only and the result goes to wb (the buffers swap roles
// afterwards), so no copy is needed. idepth is written in place: it is read only where ws > 0 and written only where ws <= 0.
__global__ __launch_bounds__(256) void trk_dilate_all_kernel(TrkLevels P) {
    int l = 0;
    while (l + 1 < P.L && (int)blockIdx.x >= P.blk0[l + 1]) ++l;
    const int wl = P.wl[l], npx = wl * P.hl[l];
    const int i = (blockIdx.x - P.blk0[l]) * blockDim.x + threadIdx.x;
    if (i >= npx) return;
    float* __restrict__ id = P.id[l]; const float* __restrict__ bak = P.ws[l]; float* __restrict__ out = P.wb[l];
    const float own = bak[i];
    float res = own;
    if (i >= wl && i < npx - wl && !(own > 0)) {
        const int diag = l < 2;
        const int o0 = diag ? 1 + wl : 1, o1 = diag ? -1 - wl : -1, o2 = diag ? wl - 1 : wl, o3 = diag ? -wl + 1 : -wl;
        // the reference reads one element before/after the array at the first/last pixel of this range
        // (i-1-wl = -1, i+1+wl = w*h): those two taps are outside the image and are skipped here.
        float sum = 0, num = 0, numn = 0;
#define NALO_TAP(o) { const int j = i + (o); if (j >= 0 && j < npx) { const float b = bak[j]; if (b > 0) { sum += id[j]; num += b; numn++; } } }
        NALO_TAP(o0) NALO_TAP(o1) NALO_TAP(o2) NALO_TAP(o3)
#undef NALO_TAP
        if (numn > 0) { id[i] = sum / numn; res = num / numn; }
    }
    out[i] = res;
}
// step 5 (:493-538): normalise + ordered (raster) compaction into pc_*. Pass 0 counts per block, pass 1 writes. (ws = the dilated weights.)
constexpr int kCompactChunk = 2048;          // interior elements per block (8 rounds of 256)
// A block takes its 8 rounds of 256 elements in ONE sweep: all loads of a lane's eight elements are in flight together, the eight ballots give the per-(round,
// wave) counts, one LDS scan of those 32 counts gives every lane its output slot - element order (round, then lane) as in the reference's double loop. (The
// rounds used to be serial, each a dependent load, a ballot and three barriers: 9.5 + 11.6 us for the two passes on a 1224x368 frame.)
template <int WRITE>
__global__ __launch_bounds__(256) void trk_compact_all_kernel(TrkLevels P, int* __restrict__ scan) {
    constexpr int R = kCompactChunk / 256;
    __shared__ int cnt[R * 4 + 1];
    int l = 0;
    while (l + 1 < P.L && (int)blockIdx.x >= P.blk0[l + 1]) ++l;
    const int chunk = blockIdx.x - P.blk0[l], nb = P.blk0[l + 1] - P.blk0[l];
    const int wl = P.wl[l], hl = P.hl[l];
    float* __restrict__ id = P.id[l]; float* __restrict__ ws = P.wb[l]; const float4* __restrict__ dIref = P.dI[l];
    int* counts = scan + P.scan0[l]; const int* offsets = counts + nb;
    const int iw = wl - 4, ih = hl - 4, total = iw * ih;            // interior y in [2,hl-2), x in [2,wl-2)
    const int base = chunk * kCompactChunk;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int idx[R]; float wsv[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int e = base + r * 256 + threadIdx.x;
        idx[r] = -1; wsv[r] = 0.f;
        if (e < total) { const int y = e / iw, x = e - y * iw; idx[r] = (x + 2) + (y + 2) * wl; wsv[r] = ws[idx[r]]; }
    }
    float val[R], col[R]; bool keep[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        val[r] = 0.f; col[r] = 0.f; keep[r] = false;
        if (idx[r] >= 0 && wsv[r] > 0) { val[r] = id[idx[r]]; col[r] = dIref[idx[r]].x; }
    }
    int rank[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (idx[r] >= 0 && wsv[r] > 0) { val[r] = val[r] / wsv[r]; keep[r] = isfinite(col[r]) && (val[r] > 0); }
        const unsigned long long m = __ballot(keep[r]);
        rank[r] = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) cnt[r * 4 + wave] = __popcll(m);
    }
    __syncthreads();
    if (!WRITE) {
        if (threadIdx.x == 0) { int t = 0; for (int k = 0; k < R * 4; ++k) t += cnt[k]; counts[chunk] = t; }
        return;
    }
    if (threadIdx.x < 64) {                                         // exclusive scan of the R x 4 counts (round-major), one wave
        const int v = lane < R * 4 ? cnt[lane] : 0;
        int inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
        if (lane < R * 4) cnt[lane] = inc - v;
    }
    __syncthreads();
    const int first = offsets[chunk];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = idx[r];
        if (i < 0) continue;
        if (wsv[r] > 0) {
            id[i] = keep[r] ? val[r] : -1.f; if (keep[r]) ws[i] = 1.f;                 // `continue` skips weightSums=1 (:524-528)
            if (keep[r]) {
                const int o = first + cnt[r * 4 + wave] + rank[r];
                const int y = i / wl, x = i - y * wl;
                P.pu[l][o] = (float)x; P.pv[l][o] = (float)y; P.pid[l][o] = val[r]; P.pcol[l][o] = col[r];
            }
        } else { id[i] = -1.f; ws[i] = 1.f; }
    }
}
// exclusive scan of the block counts of every level by one block (a few hundred counts); the per-level totals go to host-mapped memory
// followed by the sequence number the host polls on: pc_n reaches the host without a stream synchronisation
__global__ __launch_bounds__(1024) void trk_scan_all_kernel(TrkLevels P, int* __restrict__ scan, double* __restrict__ out, double seq) {
    __shared__ int part[4][4];
    // four levels at a time, 256 lanes each (the levels used to be scanned one after the other by the whole block)
    const int grp = threadIdx.x >> 8, t = threadIdx.x & 255, lane = threadIdx.x & 63, wave = t >> 6;
    for (int l0 = 0; l0 < P.L; l0 += 4) {
        const int l = l0 + grp;
        const bool on = l < P.L;
        const int nb = on ? P.blk0[l + 1] - P.blk0[l] : 0;
        const int* counts = scan + (on ? P.scan0[l] : 0); int* offsets = scan + (on ? P.scan0[l] : 0) + nb;
        const int per = (nb + 255) / 256, lo = min(t * per, nb), hi = min(lo + per, nb);
        int s = 0;
        for (int i = lo; i < hi; ++i) s += counts[i];
        int v = s;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const int nbv = __shfl_up(v, off); if (lane >= off) v += nbv; }
        if (lane == 63) part[grp][wave] = v;
        __syncthreads();
        int wpre = 0, tot = 0;
        for (int k = 0; k < 4; ++k) { const int pv = part[grp][k]; if (k < wave) wpre += pv; tot += pv; }
        int run = wpre + v - s;
        for (int i = lo; i < hi; ++i) { offsets[i] = run; run += counts[i]; }
        if (on && t == 0) { offsets[nb] = tot; out[l] = (double)tot; }
        __syncthreads();
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(&out[NALO_MAX_LEVELS], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// dense=1: the plane-sampled points CoarseTracker::makeCoarseDepthL0 appends to the level-0 cloud (reference src/FullSystem/CoarseTracker.cpp:628-655). For one
// mask cluster with fitted plane (dir, dis): every pixel of its bounding box [minx,maxx) x [miny,maxy) with x % 5 == 0, y % 5 == 0 and mask == refMaskColor
// becomes a point with new_idepth = (dir^T Ki (x, y, 1)) / -dis, colour = I_ref(x, y), in the reference's loop order (x outer, y inner). The reference stores
// point k at index pc_n + 1 + k and then counts pc_n += 1 per point (:646-650): slot [old pc_n] is never written (stale heap there, ZERO here) and the last
// sampled point falls just outside the count. Reproduced. ONE workgroup: the candidates (multiples of 5: <= w h / 25) are compacted in order, chunk by
// chunk, with ballot counts and an LDS scan over the 16 waves.
__global__ __launch_bounds__(1024) void trk_append_plane_kernel(const float* __restrict__ mask, const float4* __restrict__ dIref, int w, float d0, float d1, float d2, float dis,
                                                                float Ki00, float Ki02, float Ki11, float Ki12, float refColor, int x0, int nx, int y0, int ny, int n0,
                                                                float* __restrict__ pu, float* __restrict__ pv, float* __restrict__ pid, float* __restrict__ pcol, int* __restrict__ n_out) {
    __shared__ int wave_cnt[16];
    __shared__ int base_s;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) { base_s = 0; pu[n0] = 0.f; pv[n0] = 0.f; pid[n0] = 0.f; pcol[n0] = 0.f; }
    __syncthreads();
    // dir^T * Ki as Eigen evaluates it (row vector times matrix, then times the point): t = dir^T Ki
    const float t0 = d0 * Ki00 + d1 * 0.f + d2 * 0.f, t1 = d0 * 0.f + d1 * Ki11 + d2 * 0.f, t2 = d0 * Ki02 + d1 * Ki12 + d2 * 1.f;
    const int total = nx * ny;
    for (int c0 = 0; c0 < total; c0 += 1024) {
        const int j = c0 + tid;
        bool take = false; int x = 0, y = 0;
        if (j < total) { x = x0 + 5 * (j / ny); y = y0 + 5 * (j % ny); take = mask[x + y * w] == refColor; }
        const unsigned long long b = __ballot(take);
        if (lane == 0) wave_cnt[wv] = __popcll(b);
        __syncthreads();
        int off = base_s;
        for (int k = 0; k < wv; ++k) off += wave_cnt[k];
        if (take) {
            const int at = n0 + 1 + off + __popcll(b & ((1ull << lane) - 1ull));
            float nid = t0 * (float)x + t1 * (float)y + t2 * 1.f;
            nid /= -dis;
            pu[at] = (float)x; pv[at] = (float)y; pid[at] = nid; pcol[at] = dIref[x + y * w].x;
        }
        __syncthreads();
        if (tid == 0) { int s = 0; for (int k = 0; k < 16; ++k) s += wave_cnt[k]; base_s += s; }
        __syncthreads();
    }
    if (tid == 0) *n_out = base_s;
}
int trk_append_plane_launch(nalo_ctx* c, const float* mask, const float4* dIref, const float dir[3], float dis, float refColor, int x0, int nx, int y0, int ny, int n0, int* n_dev) {
    const float fx = c->fx[0], fy = c->fy[0], cx = c->cx[0], cy = c->cy[0];
    trk_append_plane_kernel<<<1, 1024, 0, c->stream>>>(mask, dIref, c->w, dir[0], dir[1], dir[2], dis, 1.0f / fx, -cx / fx, 1.0f / fy, -cy / fy, refColor, x0, nx, y0, ny, n0,
                                                       c->pc_u[0].p, c->pc_v[0].p, c->pc_id[0].p, c->pc_col[0].p, n_dev);
    NALO_HIP(c, hipGetLastError());
    return NALO_OK;
}

int trk_build_ref(nalo_ctx* c, int n, const float* dKu, const float* dKv, const float* dId, const float* dHdi) {
    const int L = c->levels;
    TrkLevels P;
    std::memset(&P, 0, sizeof(P));
    P.L = L;
    int dil_blocks = 0, cmp_blocks = 0, scan_total = 0;
    int dil0[NALO_MAX_LEVELS + 1] = {0}, cmp0[NALO_MAX_LEVELS + 1] = {0};
    for (int l = 0; l < L; ++l) {
        const size_t npx = (size_t)c->wl[l] * c->hl[l];
        NALO_HIP(c, c->trk_idepth[l].reserve(npx)); NALO_HIP(c, c->trk_wsum[l].reserve(npx)); NALO_HIP(c, c->trk_wbak[l].reserve(npx));
        NALO_HIP(c, c->pc_u[l].reserve(npx)); NALO_HIP(c, c->pc_v[l].reserve(npx)); NALO_HIP(c, c->pc_id[l].reserve(npx)); NALO_HIP(c, c->pc_col[l].reserve(npx));
        P.id[l] = c->trk_idepth[l].p; P.ws[l] = c->trk_wsum[l].p; P.wb[l] = c->trk_wbak[l].p; P.dI[l] = c->slots[c->slot_ref].dI[l];
        P.pu[l] = c->pc_u[l].p; P.pv[l] = c->pc_v[l].p; P.pid[l] = c->pc_id[l].p; P.pcol[l] = c->pc_col[l].p;
        P.wl[l] = c->wl[l]; P.hl[l] = c->hl[l];
        dil0[l] = dil_blocks; dil_blocks += (int)((npx + 255) / 256);
        const int total = (c->wl[l] - 4) * (c->hl[l] - 4);
        const int nb = total > 0 ? (total + kCompactChunk - 1) / kCompactChunk : 0;
        cmp0[l] = cmp_blocks; cmp_blocks += nb;
        P.scan0[l] = scan_total; scan_total += 2 * nb + 2;
    }
    dil0[L] = dil_blocks; cmp0[L] = cmp_blocks; P.scan0[L] = scan_total;
    NALO_HIP(c, c->scan_tmp.reserve((size_t)scan_total));
    const int n0 = c->wl[0] * c->hl[0];
    NALO_HIP(c, c->trk_cnt.reserve((size_t)n0 + kScatterFixCap + 16));        // hits per level-0 pixel, then the hot list and its counter
    trk_zero2_kernel<<<(n0 + 255) / 256, 256, 0, c->stream>>>(P.id[0], P.ws[0], c->trk_cnt.p, n0);
    if (n > 0) {
        trk_scatter_kernel<<<(n + 255) / 256, 256, 0, c->stream>>>(dKu, dKv, dId, dHdi, n, c->wl[0], c->hl[0], P.id[0], P.ws[0], c->trk_cnt.p);
        trk_scatter_hot_kernel<<<(n + 255) / 256, 256, 0, c->stream>>>(dKu, dKv, dId, dHdi, n, c->wl[0], c->hl[0], P.id[0], P.ws[0], c->trk_cnt.p, c->trk_cnt.p + n0);
    }
    if (L > 1) trk_sum_down_all_kernel<<<((c->wl[1] + 15) / 16) * ((c->hl[1] + 15) / 16), 256, 0, c->stream>>>(P);
    for (int l = 0; l <= L; ++l) P.blk0[l] = dil0[l];
    trk_dilate_all_kernel<<<dil_blocks, 256, 0, c->stream>>>(P);
    for (int l = 0; l <= L; ++l) P.blk0[l] = cmp0[l];
    double* dout = nullptr;
    NALO_HIP(c, hipHostGetDevicePointer((void**)&dout, c->trk_out_host, 0));
    const double seq = (double)(++c->trk_seq);
    if (cmp_blocks > 0) trk_compact_all_kernel<0><<<cmp_blocks, 256, 0, c->stream>>>(P, c->scan_tmp.p);
    trk_scan_all_kernel<<<1, 1024, 0, c->stream>>>(P, c->scan_tmp.p, dout + 96, seq);
    if (cmp_blocks > 0) trk_compact_all_kernel<1><<<cmp_blocks, 256, 0, c->stream>>>(P, c->scan_tmp.p);
    NALO_HIP(c, hipGetLastError());
    for (int l = 0; l < L; ++l) std::swap(c->trk_wsum[l], c->trk_wbak[l]);      // the dilated (then normalised) weights are the level's weightSums now
    if (!poll_flag(c, &c->trk_out_host[96 + NALO_MAX_LEVELS], seq)) return NALO_ERR_HIP;   // pc_n for the host; the point clouds follow in stream order
    for (int l = 0; l < L; ++l) c->pc_n[l] = (int)c->trk_out_host[96 + l];
    return NALO_OK;
}

}  // namespace nalo
