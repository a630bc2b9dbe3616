// CoarseInitializer::calcResAndGS (reference src/FullSystem/CoarseInitializer.cpp:338-610) for one pyramid level as ONE pass:
// per point the 8-pixel residual / Jacobian rows, the per-point Schur buffer JbBuffer_new (:468-477), the unweighted Accumulator9 of [dp0..dp7, r]
// (:497-513) and, because the reference's EAlpha accumulator never receives anything (:560-572 feed E), alphaOpt is known before the launch, so the
// weighted Schur accumulator acc9SC (:590-612) joins the same pass. 91 per-lane fp32 sums (45 + 45 + E) go through the usual quad-DPP -> LDS rows ->
// fp64 column sums -> block partial -> fp64 finish. doStep (:910-938) is a thread-per-point kernel.
#include "nalo_internal.h"
#include <mutex>
#include "reduce.h"
#include <type_traits>

namespace nalo {

constexpr int kInitVals = 94, kInitStride = 96;      // 45 (acc9) + 45 (acc9SC) + E + calcEC's {sum rOld^2, sum rNew^2, count}
#define NALO_PAT(i) {kPatternDx[i], kPatternDy[i]}
__constant__ int kInitPattern[8][2] = {NALO_PAT(0), NALO_PAT(1), NALO_PAT(2), NALO_PAT(3), NALO_PAT(4), NALO_PAT(5), NALO_PAT(6), NALO_PAT(7)};   // settings.cpp:297 (ref_constants.h)
#undef NALO_PAT

__device__ __forceinline__ float3 init_interp33(const float4* __restrict__ img, float x, float y, int width) {
    const int ix = (int)x, iy = (int)y;
    const float dx = x - ix, dy = y - iy, dxdy = dx * dy;
    const float4* bp = img + ix + iy * width;
    const float4 p00 = bp[0], p10 = bp[1], p01 = bp[width], p11 = bp[1 + width];
    const float w11 = dxdy, w01 = dy - dxdy, w10 = dx - dxdy, w00 = 1 - dx - dy + dxdy;
    return make_float3(w11 * p11.x + w01 * p01.x + w10 * p10.x + w00 * p00.x, w11 * p11.y + w01 * p01.y + w10 * p10.y + w00 * p00.y,
                       w11 * p11.z + w01 * p01.z + w10 * p10.z + w00 * p00.z);
}

__global__ __launch_bounds__(256) void init_calc_kernel(InitParams P, double* __restrict__ partial) {
    __shared__ float smem[64 * (kInitVals + 1)];
    __shared__ float bsum[kInitStride];
    float acc[kInitVals];
#pragma unroll
    for (int k = 0; k < kInitVals; ++k) acc[k] = 0.f;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < P.n) {
        float maxstep = 1e10f;
        if (!P.isGood[i]) {
            acc[90] += P.energy[i * 2];
            P.energy_new[i * 2] = P.energy[i * 2]; P.energy_new[i * 2 + 1] = P.energy[i * 2 + 1]; P.isGood_new[i] = 0; P.maxstep[i] = maxstep;
        } else {
            const float u = P.u[i], v = P.v[i], idn = P.idepth_new[i];
            float dp[8][8], r[8], jb[10];
#pragma unroll
            for (int k = 0; k < 10; ++k) jb[k] = 0.f;
            bool good = true; float en = 0.f;
#pragma unroll
            for (int idx = 0; idx < 8; ++idx) {
                if (!good) continue;                                          // `break` of the reference: later pixels contribute nothing
                const int dx = kInitPattern[idx][0], dy = kInitPattern[idx][1];
                float pt[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) pt[k] = P.RKi[k * 3] * (u + dx) + P.RKi[k * 3 + 1] * (v + dy) + P.RKi[k * 3 + 2] * 1 + P.t[k] * idn;
                const float uu = pt[0] / pt[2], vv = pt[1] / pt[2], Ku = P.fx * uu + P.cx, Kv = P.fy * vv + P.cy;
                const float new_idepth = idn / pt[2];
                if (!(Ku > 1 && Kv > 1 && Ku < P.wl - 2 && Kv < P.hl - 2 && new_idepth > 0)) { good = false; continue; }
                const float3 hit = init_interp33(P.colorNew, Ku, Kv, P.wl);
                const float rlR = init_interp33(P.colorRef, u + dx, v + dy, P.wl).x;
                if (!isfinite(rlR) || !isfinite(hit.x)) { good = false; continue; }
                const float residual = hit.x - P.r2new0 * rlR - P.r2new1;
                const float ar = fabsf(residual);
                float hw = ar < kHuberTH ? 1 : kHuberTH / ar;
                en += hw * residual * residual * (2 - hw);
                const float dxdd = (P.t[0] - P.t[2] * uu) / pt[2], dydd = (P.t[1] - P.t[2] * vv) / pt[2];
                if (hw < 1) hw = sqrtf(hw);
                const float dxInterp = hw * hit.y * P.fx, dyInterp = hw * hit.z * P.fy;
                dp[0][idx] = new_idepth * dxInterp;
                dp[1][idx] = new_idepth * dyInterp;
                dp[2][idx] = -new_idepth * (uu * dxInterp + vv * dyInterp);
                dp[3][idx] = -uu * vv * dxInterp - (1 + vv * vv) * dyInterp;
                dp[4][idx] = (1 + uu * uu) * dxInterp + uu * vv * dyInterp;
                dp[5][idx] = -vv * dxInterp + uu * dyInterp;
                dp[6][idx] = -hw * P.r2new0 * rlR;
                dp[7][idx] = -hw * 1;
                const float dd = dxInterp * dxdd + dyInterp * dydd;
                r[idx] = hw * residual;
                const float nx = dxdd * P.fx, ny = dydd * P.fy;
                const float ms = 1.0f / sqrtf(nx * nx + ny * ny);
                if (ms < maxstep) maxstep = ms;
#pragma unroll
                for (int k = 0; k < 8; ++k) jb[k] += dp[k][idx] * dd;
                jb[8] += r[idx] * dd;
                jb[9] += dd * dd;
            }
            P.maxstep[i] = maxstep;
            if (!good || en > P.outlierTH[i] * 20) {
                acc[90] += P.energy[i * 2];
                P.isGood_new[i] = 0; P.energy_new[i * 2] = P.energy[i * 2]; P.energy_new[i * 2 + 1] = P.energy[i * 2 + 1];
#pragma unroll
                for (int k = 0; k < 10; ++k) P.Jb[i * 10 + k] = jb[k];
            } else {
                acc[90] += en;
                P.isGood_new[i] = 1; P.energy_new[i * 2] = en; P.energy_new[i * 2 + 1] = (idn - 1) * (idn - 1);
                if (P.idepth) {                                               // calcEC (:634-655) reads exactly what this pass has at hand: the regulariser's old / new energy
                    const float iR = P.iR[i], rOld = P.idepth[i] - iR, rNew = idn - iR;
                    acc[91] += rOld * rOld; acc[92] += rNew * rNew; acc[93] += 1.f;
                }
#pragma unroll
                for (int idx = 0; idx < 8; ++idx) {                          // acc9: upper triangle of [dp0..dp7, r] outer products (constant indices)
                    float J[9];
#pragma unroll
                    for (int k = 0; k < 8; ++k) J[k] = dp[k][idx];
                    J[8] = r[idx];
#pragma unroll
                    for (int a = 0; a < 9; ++a)
#pragma unroll
                        for (int b = a; b < 9; ++b) acc[a * 9 - a * (a - 1) / 2 + (b - a)] += J[a] * J[b];
                }
                // per-point Schur complement (:590-612)
                P.lastHessian_new[i] = jb[9];
                jb[8] += P.alphaOpt * (idn - 1);
                jb[9] += P.alphaOpt;
                if (P.alphaOpt == 0) { jb[8] += P.couplingWeight * (idn - P.iR[i]); jb[9] += P.couplingWeight; }
                jb[9] = 1 / (1 + jb[9]);
#pragma unroll
                for (int k = 0; k < 10; ++k) P.Jb[i * 10 + k] = jb[k];
                float J[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) J[k] = jb[k];
                const float w = jb[9];
#pragma unroll
                for (int a = 0; a < 9; ++a) {                                // updateSingleWeighted (MatrixAccumulators.h:1242-1315)
                    acc[45 + a * 9 - a * (a - 1) / 2] += J[a] * J[a] * w;
                    J[a] *= w;
#pragma unroll
                    for (int b = a + 1; b < 9; ++b) acc[45 + a * 9 - a * (a - 1) / 2 + (b - a)] += J[b] * J[a];
                }
            }
        }
    }
    block_reduce_cols<kInitVals, 256>(acc, smem, bsum);
    if (threadIdx.x < kInitVals) partial[(size_t)blockIdx.x * kInitStride + threadIdx.x] = (double)bsum[threadIdx.x];
}
// out: device memory, or (mapped != 0) host-mapped memory the host polls: the 94 sums, then the sequence number in slot 95 behind a system-scope release
__global__ __launch_bounds__(1024) void init_finish_kernel(const double* __restrict__ partial, int nblocks, double* __restrict__ out, int mapped, double seq) {
    __shared__ double part[8][128];
    const int j = threadIdx.x & 127, g = threadIdx.x >> 7;
    double s = 0;
    if (j < kInitVals) for (int b = g; b < nblocks; b += 8) s += partial[(size_t)b * kInitStride + j];
    part[g][j] = s;
    __syncthreads();
    if (g == 0 && j < kInitVals) {
        double t = 0; for (int k = 0; k < 8; ++k) t += part[k][j];
        if (mapped) __hip_atomic_store(&out[j], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); else out[j] = t;
    }
    if (mapped) {
        __syncthreads();
        if (threadIdx.x == 0) { __threadfence_system(); __hip_atomic_store(&out[95], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
    }
}
// CoarseInitializer::doStep (:910-938)
__global__ __launch_bounds__(256) void init_do_step_kernel(int n, const uint8_t* __restrict__ isGood, const float* __restrict__ Jb, const float* __restrict__ maxstep,
                                                          const float* __restrict__ idepth, float lambda, InitInc inc, float* __restrict__ idepth_new) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !isGood[i]) return;
    const float* jb = Jb + (size_t)i * 10;
    float dot = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) dot += jb[k] * inc.v[k];
    const float b = jb[8] + dot;
    float step = -b * jb[9] / (1 + lambda);
    float ms = 0.25f * maxstep[i];
    if (ms > 1e10f) ms = 1e10f;
    if (step > ms) step = ms;
    if (step < -ms) step = -ms;
    float nid = idepth[i] + step;
    if (nid < 1e-3f) nid = 1e-3f;
    if (nid > 50) nid = 50;
    idepth_new[i] = nid;
}

// CoarseInitializer::applyStep (:939-956) on the resident arrays (the JbBuffer swap is a pointer swap on the host)
__global__ __launch_bounds__(256) void init_apply_step_kernel(int n, uint8_t* __restrict__ isGood, const uint8_t* __restrict__ isGood_new, float* __restrict__ idepth,
                                                             float* __restrict__ idepth_new, const float* __restrict__ iR, float* __restrict__ energy,
                                                             const float* __restrict__ energy_new, float* __restrict__ lastHessian, const float* __restrict__ lastHessian_new) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (!isGood[i]) { const float r = iR[i]; idepth[i] = r; idepth_new[i] = r; return; }
    energy[2 * i] = energy_new[2 * i]; energy[2 * i + 1] = energy_new[2 * i + 1];
    isGood[i] = isGood_new[i];
    idepth[i] = idepth_new[i];
    lastHessian[i] = lastHessian_new[i];
}

// ---------------------------------------------------------------------------------------------------------------- the "sequential" sweeps
// optReg (:656-691) and the top level's resetPoints (:882-909) are Gauss-Seidel sweeps: point i reads the values its lower-index neighbours were just given and the
// OLD values of its higher-index neighbours. Ordered by dependency instead of by index they are exact and parallel: i and j are ordered (lower index first) whenever one
// is among the other's 10 neighbours, every point's step is 1 + the largest step among the points it is ordered after, and the points of one step touch nothing of each
// other. The host builds that schedule once per setFirst (host_init.hip: a few hundred steps of ~40 points on the KITTI frame, cut to <= kSweepNT points). One workgroup
// walks it with the level's iR in LDS (+inf = not good, which is also how resetPoints' newly good points become visible to later ones; a missing neighbour is the
// index n, a slot that always holds +inf); the records of the next four
// steps are in flight while four steps compute. Levels too large for LDS keep the values in a global scratch array, read and written past the L1 (same kernel, LDS = false).
struct SweepAccessLds { static __device__ __forceinline__ float ld(const float* p) { return *p; } static __device__ __forceinline__ void st(float* p, float x) { *p = x; } };
struct SweepAccessGlobal {
    static __device__ __forceinline__ float ld(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    static __device__ __forceinline__ void st(float* p, float x) { __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
};
// compare-exchange of two non-NaN values: the bare instructions (fminf / fmaxf would first quiet every input with a v_max_f32 x, x)
__device__ __forceinline__ void sweep_ce(float& a, float& b) {
    float lo, hi;
    asm("v_min_f32 %0, %1, %2" : "=v"(lo) : "v"(a), "v"(b));
    asm("v_max_f32 %0, %1, %2" : "=v"(hi) : "v"(a), "v"(b));
    a = lo; b = hi;
}
// ascending sort of 10 values: 29 compare-exchanges in 8 layers (checked exhaustively on 0/1 inputs, scripts/check_sort10.py)
__device__ __forceinline__ void sweep_sort10(float (&v)[10]) {
#define CE(a, b) sweep_ce(v[a], v[b]);
    CE(0, 8) CE(1, 9) CE(2, 7) CE(3, 5) CE(4, 6)
    CE(0, 2) CE(1, 4) CE(5, 8) CE(7, 9)
    CE(0, 3) CE(2, 4) CE(5, 7) CE(6, 9)
    CE(0, 1) CE(3, 6) CE(8, 9)
    CE(1, 5) CE(2, 3) CE(4, 8) CE(6, 7)
    CE(1, 2) CE(3, 5) CE(4, 6) CE(7, 8)
    CE(2, 3) CE(4, 5) CE(6, 7)
    CE(3, 4) CE(5, 6)
#undef CE
}
// MODE 0: optReg with snapped == true. MODE 1: the `lvl == top && !isGood` part of resetPoints.   rec: [n][3] int4 = {point, nn0..nn9, -} in schedule order.
// A grid-wide pass before (init_sweep_pre_kernel) writes the values the walk starts from (iR, NaN where the point is not good) and, for optReg, the points' inverse
// depths in schedule order, so that the walk's loads depend on nothing; a grid-wide pass after it (init_sweep_post_kernel) writes the results back under the
// reference's conditions. The single workgroup in between only fills its LDS (float4, eight loads in flight), walks the schedule and stores the values.
struct SweepRec { int4 a, b, c; float idp; };
template <int MODE>
__global__ __launch_bounds__(256) void init_sweep_pre_kernel(int n, int npad, const int4* __restrict__ rec, const float* __restrict__ iR, const uint8_t* __restrict__ isGood,
                                                            const float* __restrict__ idepth, float* __restrict__ val, float* __restrict__ idp_s) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npad) return;
    if (i >= n) { val[i] = __builtin_inff(); return; }                              // slot n: what a missing neighbour (index n in rec) reads
    val[i] = isGood[i] ? iR[i] : __builtin_inff();
    if (MODE == 0) idp_s[i] = idepth[rec[(size_t)i * 3].x];
}
template <int MODE>
__global__ __launch_bounds__(256) void init_sweep_post_kernel(int n, const float* __restrict__ val, float* __restrict__ iR, uint8_t* __restrict__ isGood, float* __restrict__ idepth,
                                                             float* __restrict__ idepth_new) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = val[i];
    if (MODE == 0) { if (isGood[i]) iR[i] = x; }
    else if (!isGood[i] && x < __builtin_inff()) { isGood[i] = 1; iR[i] = x; idepth[i] = x; idepth_new[i] = x; }
}
// gval: the values in global memory (n + 1 floats rounded up to float4s, 16-byte aligned), start values in, results out; idp_s as above
template <int MODE, bool LDS>
__global__ __launch_bounds__(kSweepNT) void init_sweep_kernel(int n, int nsteps, const int* __restrict__ off, const int4* __restrict__ rec, float w, float one_minus_w,
                                                             float* __restrict__ gval, const float* __restrict__ idp_s) {
    using A = typename std::conditional<LDS, SweepAccessLds, SweepAccessGlobal>::type;
    extern __shared__ float4 sweep_sm4[];
    float* sweep_sm = (float*)sweep_sm4;
    const int n4 = (n + 4) >> 2;
    float* val = LDS ? sweep_sm : gval;
    int* soff = (int*)(sweep_sm + (LDS ? 4 * n4 : 0));                               // off[], then n up to entry nsteps4 + 4 (empty steps: the walk is unrolled by four)
    const int nsteps4 = (nsteps + 3) & ~3;
    const int tid = threadIdx.x;
    if (LDS) {
        const float4* g4 = (const float4*)gval;
        for (int j0 = 0; j0 < n4; j0 += 8 * kSweepNT) {
            float4 x[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) { const int j = j0 + k * kSweepNT + tid; x[k] = g4[j < n4 ? j : n4 - 1]; }
#pragma unroll
            for (int k = 0; k < 8; ++k) { const int j = j0 + k * kSweepNT + tid; if (j < n4) sweep_sm4[j] = x[k]; }
        }
    }
    for (int s = tid; s <= nsteps4 + 4; s += kSweepNT) soff[s] = off[s < nsteps ? s : nsteps];
    __syncthreads();
    // (round 4: whether the lane has a point in step s travels with the record - soff[] is read when the record is requested, a group of four ahead -, and the
    // point's own value is asked for together with its neighbours': one LDS round trip per step on the walk's critical path instead of three)
    auto load = [&](int s) {
        const int p0 = soff[s] + tid, p = p0 < n ? p0 : n - 1;
        const int4* r = rec + (size_t)p * 3;
        SweepRec R; R.a = r[0]; R.b = r[1]; R.c = r[2]; R.idp = MODE == 0 ? idp_s[p] : 0.f;
        R.c.w = p0 < soff[s + 1];                                                   // the record's spare word: active in this step
        return R;
    };
    auto step = [&](int s, const SweepRec& R) {
        const bool act = R.c.w != 0;
        const int i = R.a.x;
        const int nn[10] = {R.a.y, R.a.z, R.a.w, R.b.x, R.b.y, R.b.z, R.b.w, R.c.x, R.c.y, R.c.z};
        const float inf = __builtin_inff();
        float v[10];
        const float self = A::ld(val + i);
#pragma unroll
        for (int k = 0; k < 10; ++k) v[k] = A::ld(val + nn[k]);
        const bool self_good = self < inf;
        if (act && (MODE == 0 ? self_good : !self_good)) {
            if (MODE == 0) {
                sweep_sort10(v);                                                    // the m good neighbours' values first, +inf behind them
                if (v[2] < inf) {                                                   // nnn > 2; nth_element(idnn, idnn + nnn/2, idnn + nnn): the nnn/2-th smallest
                    const float med = v[9] < inf ? v[5] : (v[7] < inf ? v[4] : (v[5] < inf ? v[3] : (v[3] < inf ? v[2] : v[1])));
                    A::st(val + i, __fadd_rn(__fmul_rn(one_minus_w, R.idp), __fmul_rn(w, med)));
                }
            } else {
                float sum = 0.f, cnt = 0.f;
#pragma unroll
                for (int k = 0; k < 10; ++k) if (v[k] < inf) { sum = __fadd_rn(sum, v[k]); cnt += 1.f; }
                if (cnt > 0) A::st(val + i, sum / cnt);
            }
        }
        // the step's LDS writes before anyone's next reads. __syncthreads() would also drain the records in flight (its fence covers global memory): in LDS mode only
        // the LDS counter has to reach zero before the barrier
        if (LDS) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); else __syncthreads();
    };
    // the records of the next four steps are requested when a group of four starts and land while it computes
    SweepRec c0 = load(0), c1 = load(1), c2 = load(2), c3 = load(3);
    for (int s = 0; s < nsteps4; s += 4) {
        const SweepRec n0 = load(s + 4), n1 = load(s + 5), n2 = load(s + 6), n3 = load(s + 7);
        step(s, c0); step(s + 1, c1); step(s + 2, c2); step(s + 3, c3);
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    }
    if (LDS) {
        float4* g4 = (float4*)gval;
        for (int j = tid; j < n4; j += kSweepNT) g4[j] = sweep_sm4[j];
    }
}

// resetPoints' per-point part (:885-888); the unsnapped start of trackFrame (:100-110); optReg with snapped == false (:659-664)
__global__ __launch_bounds__(256) void init_reset_kernel(int n, float* __restrict__ energy, float* __restrict__ idepth_new, const float* __restrict__ idepth) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    energy[2 * i] = 0.f; energy[2 * i + 1] = 0.f; idepth_new[i] = idepth[i];
}
__global__ __launch_bounds__(256) void init_fill_kernel(int n, float* __restrict__ iR, float* __restrict__ idepth_new, float* __restrict__ lastHessian) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    iR[i] = 1.f;
    if (idepth_new) idepth_new[i] = 1.f;
    if (lastHessian) lastHessian[i] = 0.f;
}
// propagateDown (:736-766) without its optReg: one lane per point of the finer level
__global__ __launch_bounds__(256) void init_propagate_down_kernel(int n, const int* __restrict__ parent, const uint8_t* __restrict__ pGood, const float* __restrict__ pLastHessian,
                                                                 const float* __restrict__ pIR, uint8_t* __restrict__ isGood, float* __restrict__ iR, float* __restrict__ idepth,
                                                                 float* __restrict__ idepth_new, float* __restrict__ lastHessian) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int par = parent[i];
    const float pH = pLastHessian[par], pR = pIR[par];
    if (!pGood[par] || pH < 0.1f) return;
    if (!isGood[i]) { iR[i] = pR; idepth[i] = pR; idepth_new[i] = pR; isGood[i] = 1; lastHessian[i] = 0.f; }
    else {
        const float r = iR[i], lh = lastHessian[i];
        const float fused = __fadd_rn(__fmul_rn(__fmul_rn(r, lh), 2.f), __fmul_rn(pR, pH)) / __fadd_rn(__fmul_rn(lh, 2.f), pH);
        iR[i] = fused; idepth[i] = fused; idepth_new[i] = fused;
    }
}
// propagateUp (:695-734) without its optReg: one lane per parent, its children in index order (child_off / child_idx: the host's counting sort of `parent`)
__global__ __launch_bounds__(256) void init_propagate_up_kernel(int nT, const int* __restrict__ child_off, const int* __restrict__ child_idx, const uint8_t* __restrict__ cGood,
                                                               const float* __restrict__ cIR, const float* __restrict__ cLastHessian, uint8_t* __restrict__ isGood,
                                                               float* __restrict__ iR, float* __restrict__ idepth) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nT) return;
    float acc = 0.f, num = 0.f;
    for (int k = child_off[p]; k < child_off[p + 1]; ++k) {
        const int ch = child_idx[k];
        if (!cGood[ch]) continue;
        const float lh = cLastHessian[ch];
        acc = __fadd_rn(acc, __fmul_rn(cIR[ch], lh));
        num = __fadd_rn(num, lh);
    }
    if (num > 0) { const float r = acc / num; iR[p] = r; idepth[p] = r; isGood[p] = 1; }
    else iR[p] = 0.f;
}

int init_sweep_launch(nalo_ctx* c, int mode, int n, int nsteps, const int* off, const int* rec, float* iR, uint8_t* isGood, float* idepth, float* idepth_new, float regWeight, float* scratch) {
    if (n <= 0 || nsteps <= 0) return NALO_OK;
    const size_t npad = ((size_t)n + 4) & ~(size_t)3;                               // n values + the +inf slot, in float4s
    const size_t soff_words = (size_t)((nsteps + 3) & ~3) + 8;
    const size_t lds_bytes = (npad + soff_words) * 4;
    const bool lds = lds_bytes <= kSweepLdsBytes;
    const size_t sm = lds ? lds_bytes : soff_words * 4;
    if (!scratch) return fail(c, NALO_ERR_STATE, "init_sweep_launch: no scratch block");
    float *gval = scratch, *idp_s = scratch + npad;                                 // scratch: 2 * npad floats, 16-byte aligned
    {   // the LDS opt-in of the sweep kernels is a property of the (device, function) pair: latched per device under a lock (contexts on several devices / host threads)
        static std::mutex mu; static bool attr_done[64] = {};
        std::lock_guard<std::mutex> lk(mu);
        const int dv = c->device >= 0 && c->device < 64 ? c->device : 0;
        if (!attr_done[dv]) {
            NALO_HIP(c, hipFuncSetAttribute((const void*)init_sweep_kernel<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSweepLdsBytes));
            NALO_HIP(c, hipFuncSetAttribute((const void*)init_sweep_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSweepLdsBytes));
            attr_done[dv] = true;
        }
    }
    const float w = regWeight, omw = 1 - regWeight;
    const int4* r4 = (const int4*)rec;
    const int nb = (n + 4 + 255) / 256;
    if (mode == 0) {
        init_sweep_pre_kernel<0><<<nb, 256, 0, c->stream>>>(n, (int)npad, r4, iR, isGood, idepth, gval, idp_s);
        if (lds) init_sweep_kernel<0, true><<<1, kSweepNT, sm, c->stream>>>(n, nsteps, off, r4, w, omw, gval, idp_s);
        else init_sweep_kernel<0, false><<<1, kSweepNT, sm, c->stream>>>(n, nsteps, off, r4, w, omw, gval, idp_s);
        init_sweep_post_kernel<0><<<nb, 256, 0, c->stream>>>(n, gval, iR, isGood, idepth, idepth_new);
    } else {
        init_sweep_pre_kernel<1><<<nb, 256, 0, c->stream>>>(n, (int)npad, r4, iR, isGood, idepth, gval, idp_s);
        if (lds) init_sweep_kernel<1, true><<<1, kSweepNT, sm, c->stream>>>(n, nsteps, off, r4, w, omw, gval, idp_s);
        else init_sweep_kernel<1, false><<<1, kSweepNT, sm, c->stream>>>(n, nsteps, off, r4, w, omw, gval, idp_s);
        init_sweep_post_kernel<1><<<nb, 256, 0, c->stream>>>(n, gval, iR, isGood, idepth, idepth_new);
    }
    NALO_HIP(c, hipGetLastError());
    return NALO_OK;
}
int init_reset_launch(nalo_ctx* c, int n, float* energy, float* idepth_new, const float* idepth) {
    if (n > 0) init_reset_kernel<<<(n + 255) / 256, 256, 0, c->stream>>>(n, energy, idepth_new, idepth);
    NALO_HIP(c, hipGetLastError());
    return NALO_OK;
}
int init_fill_launch(nalo_ctx* c, int n, float* iR, float* idepth_new, float* lastHessian) {
    if (n > 0) init_fill_kernel<<<(n + 255) / 256, 256, 0, c->stream>>>(n, iR, idepth_new, lastHessian);
    NALO_HIP(c, hipGetLastError());
    return NALO_OK;
}
int init_propagate_down_launch(nalo_ctx* c, int n, const int* parent, const uint8_t* pGood, const float* pLastHessian, const float* pIR, uint8_t* isGood, float* iR, float* idepth,
                               float* idepth_new, float* lastHessian) {
    if (n > 0) init_propagate_down_kernel<<<(n + 255) / 256, 256, 0, c->stream>>>(n, parent, pGood, pLastHessian, pIR, isGood, iR, idepth, idepth_new, lastHessian);
    NALO_HIP(c, hipGetLastError());
    return NALO_OK;
}
int init_propagate_up_launch(nalo_ctx* c, int nT, const int* child_off, const int* child_idx, const uint8_t* cGood, const float* cIR, const float* cLastHessian, uint8_t* isGood,
                             float* iR, float* idepth) {
    if (nT > 0) init_propagate_up_kernel<<<(nT + 255) / 256, 256, 0, c->stream>>>(nT, child_off, child_idx, cGood, cIR, cLastHessian, isGood, iR, idepth);
    NALO_HIP(c, hipGetLastError());
    return NALO_OK;
}

int init_calc_launch(nalo_ctx* c, InitParams& P, int lvl, double* sums, int mapped, double seq) {
    P.wl = c->wl[lvl]; P.hl = c->hl[lvl];
    const int nb = (P.n + 255) / 256;
    NALO_HIP(c, c->trk_partial.reserve(((size_t)nb * kInitStride + 128) * 2));          // doubles in a float buffer
    double* partial = (double*)c->trk_partial.p;
    init_calc_kernel<<<nb, 256, 0, c->stream>>>(P, partial);
    init_finish_kernel<<<1, 1024, 0, c->stream>>>(partial, nb, sums, mapped, seq);
    NALO_HIP(c, hipGetLastError());
    return NALO_OK;
}
int init_do_step_launch(nalo_ctx* c, int n, const uint8_t* isGood, const float* Jb, const float* maxstep, const float* idepth, float lambda, const float inc[8], float* idepth_new) {
    InitInc I; for (int k = 0; k < 8; ++k) I.v[k] = inc[k];
    if (n > 0) init_do_step_kernel<<<(n + 255) / 256, 256, 0, c->stream>>>(n, isGood, Jb, maxstep, idepth, lambda, I, idepth_new);
    NALO_HIP(c, hipGetLastError());
    return NALO_OK;
}
int init_apply_step_launch(nalo_ctx* c, int n, uint8_t* isGood, const uint8_t* isGood_new, float* idepth, float* idepth_new, const float* iR, float* energy, const float* energy_new,
                           float* lastHessian, const float* lastHessian_new) {
    if (n > 0) init_apply_step_kernel<<<(n + 255) / 256, 256, 0, c->stream>>>(n, isGood, isGood_new, idepth, idepth_new, iR, energy, energy_new, lastHessian, lastHessian_new);
    NALO_HIP(c, hipGetLastError());
    return NALO_OK;
}

}  // namespace nalo
