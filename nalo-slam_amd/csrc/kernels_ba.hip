// Sliding-window photometric BA kernels for gfx950 (reference paths relative to src/).
//
//  ba_linearize_kernel   a5+a6+a7  PointFrameResidual::linearize + applyRes + EFResidual::takeDataF
//                                  (FullSystem/Residuals.cpp:78-274,306-328; OptimizationBackend/EnergyFunctionalStructs.cpp:39-50)
//                                  fused with AccumulatedTopHessianSSE::addPoint<0|2> + AccumulatorApprox
//                                  (OptimizationBackend/AccumulatedTopHessian.cpp:39-162, MatrixAccumulators.h:754-915)
//                                  and, in marginalisation mode, EFResidual::fixLinearizationF (EnergyFunctionalStructs.cpp:89-115).
//                                  The 296-byte RawResidualJacobian never leaves registers.
//  ba_sc_kernel          a9        AccumulatedSCHessianSSE::addPoint (OptimizationBackend/AccumulatedSCHessian.cpp:34-77): the per-point sums
//                                  (Hdd/bd/Hcd over the active residuals, HdiF, bdSumF) and a
//                                  per-host weighted SYRK  G_h = sum_p HdiF_p a_p a_p^T,  a_p = [JpJdF(t) for t != h | Hcd | bdSum]:
//                                  accD = the 8x8 blocks, accE = the Hcd columns, accEB = the bdSum column, accHcc/accbc = the corner.
//                                  The one GEMM-shaped kernel of the path: fp32 MFMA (v_mfma_f32_16x16x4_f32) over the upper-triangular tiles.
//  ba_reduce_*           fp64 finish of the per-block fp32 partials (replaces the per-thread replicas summed in stitchDoubleInternal)
//  ba_stitch_*           a8+a10    stitchDouble for both systems as  H~ = sum_b S_b M_b S_b^T  with S_b built from adHost/adTarget
//                                  (AccumulatedTopHessian.cpp:171-303, AccumulatedSCHessian.cpp:78-219)
//  ba_resub_kernel       a12       EnergyFunctional::resubstituteFPt (OptimizationBackend/EnergyFunctional.cpp:291-317)
//  ba_step_kernel                  point part of FullSystem::doStepFromBackup (FullSystem/FullSystemOptimize.cpp:269-277)
//  ba_energy_th_kernel             FullSystem::setNewFrameEnergyTH (FullSystemOptimize.cpp:95-143): exact order statistic by radix select
//
// Everything but ba_sc_kernel is HBM/latency bound integer/byte/gather work (no MFMA): 16-byte texel gathers, coalesced [target][point] slot arrays,
// block-uniform precalc through scalar loads, DPP + LDS reductions, no float atomics on any sum.
#include "nalo_internal.h"
#include "ba_device.h"
#include "reduce.h"

namespace nalo {

// ba_linearize_kernel lives in kernels_ba_lin.hip

// resetOOB for every active residual at the start of optimize() (FullSystemOptimize.cpp:412-429, Residuals.h:88-94)
__global__ __launch_bounds__(256) void ba_reset_oob_kernel(uint8_t* __restrict__ rs_state, float2* __restrict__ rs_energy, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t st = rs_state[i];
    if ((st & RS_EXISTS) && !(st & RS_LINEARIZED)) { rs_state[i] = (uint8_t)(st & ~RS_STATE_MASK); rs_energy[i] = make_float2(0.f, 0.f); }
}

// ------------------------------------------------------------------------------------------------ a9: per-point sums, per-host weighted SYRK
// The per-point part of AccumulatedSCHessianSSE::addPoint (AccumulatedSCHessian.cpp:36-57): EFPoint::{Hdd,bd,Hcd}_acc = the sum over the point's active
// residuals IN TARGET ORDER of the 24-byte shares ba_linearize_kernel left per residual slot (AccumulatedTopHessian.cpp:132-157), then HdiF and bdSumF.
// Rounds 1-3 ran it as its own launch (ba_pt_acc_kernel: 14 us at 250 k points, 60 us at 1 M - as much HBM time again as the SYRK has). It now lives INSIDE
// ba_sc_kernel, pipelined with the SYRK passes (below): sc_point_finish is the arithmetic both forms of the kernel share.
struct ScPoint { float4 pa; float4 hc; int ngood; };                // pa = {Hdd, bd, HdiF, bdSumF} (what ba_resub_kernel reads back as pt_acc)
__device__ __forceinline__ void sc_point_finish(const BADev& B, int d, ScPoint& P, float prior, float idepth, float idepth_zero, int margOnly, float priorScaleMarg, int shiftPriorToZero) {
    B.pt_hcd[d] = P.hc; B.pt_ngood[d] = (uint8_t)P.ngood;
    if (P.ngood == 0) { P.pa.z = 0.f; P.pa.w = 0.f; }
    else {
        if (margOnly) { prior *= priorScaleMarg; B.pt_prior[d] = prior; }          // EnergyFunctional.cpp:630
        float Hs = P.pa.x + prior;                                                  // Hdd_accAF + Hdd_accLF + priorF (only one of AF/LF is live)
        if (Hs < 1e-10f) Hs = 1e-10f;
        P.pa.z = (float)(1.0 / (double)Hs);
        P.pa.w = P.pa.y;
        if (shiftPriorToZero) P.pa.w += prior * (idepth - idepth_zero);
    }
    B.pt_acc[d] = P.pa;
}

// ba_sc_kernel: G_h = sum over the host's points of HdiF * row row^T, row = [JpJdF(t != h) (8 each) | Hcd (4) | bdSum | 0..] (NPL = 16T columns).
// A workgroup walks a group of up to sc_bpw consecutive point blocks of ONE host in passes of SUB points: the rows of a pass are staged in LDS, and
// the global loads of the NEXT pass are issued into registers before the SYRK of the current one, so HBM latency hides under the matrix work. The fp64
// partial (NPL^2 x 8 B, as large as a block's operands) is written once per group.
// KS > 1 splits a block's points over KS workgroups (small windows: a KITTI-sized window has ~12 point blocks for 256 CUs, and the
// k-loop of a whole block is a 20 us serial chain); the partials are [groups * KS][NPL * NPL].
// The SYRK runs on the matrix cores. G is symmetric, so only the T(T+1)/2 upper 16x16 tiles are computed, dealt round-robin to the four waves; per
// step of 4 points a wave feeds v_mfma_f32_16x16x4_f32 with A[i][k] = a_pk[16 ti + i] and B[k][j] = w_pk a_pk[16 tj + j] (one f32 VGPR each, read
// from the staged LDS rows): an exact k-ordered fp32 fma chain per entry, flushed into fp64 every RUN points. The partial of a workgroup is the
// compact list of its upper tiles in MFMA register order ([tile][reg][lane]: every store is 512 contiguous bytes); ba_reduce_kernel mirrors.
// (The first version computed all T*T tiles with a 16x16 thread grid on the vector ALU: 748 us on the 1M-point / 12-frame window, this one 175.)
typedef float sc_f32x4 __attribute__((ext_vector_type(4)));
// upper-triangular 16x16 tile enumeration (row-major over ti <= tj) of the MFMA SYRK
template <int T> __device__ constexpr int sc_tile_i(int idx) { int ti = 0; while (idx >= T - ti) { idx -= T - ti; ++ti; } return ti; }
template <int T> __device__ constexpr int sc_tile_j(int idx) { int ti = 0; while (idx >= T - ti) { idx -= T - ti; ++ti; } return ti + idx; }
// the MFMAs of one 4-point step: tile indices are template constants, so x[ti] / xw[tj] are plain registers
template <int T, int WAVE, int NW, int M, int MAXM>
__device__ __forceinline__ void sc_mfma_tiles(const float (&x)[T], const float (&xw)[T], sc_f32x4 (&macc)[MAXM]) {
    if constexpr (M < MAXM) {
        constexpr int idx = WAVE + NW * M;
        if constexpr (idx < T * (T + 1) / 2) {
            constexpr int ti = sc_tile_i<T>(idx), tj = sc_tile_j<T>(idx);
            macc[M] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[ti], xw[tj], macc[M], 0, 0, 0);
        }
        sc_mfma_tiles<T, WAVE, NW, M + 1, MAXM>(x, xw, macc);
    }
}
// One staged pass (SUB points) of wave WAVE's tiles {WAVE, WAVE + NW, ...}. The tile list is a compile-time property of the wave, so per step of
// 4 points the wave reads the column blocks its tiles touch ONCE (x[c]; the others are never loaded) and every a / w*b operand is a register pick.
template <int T, int WAVE, int NW, int SUB, int NPLP, int RUN, int MAXM>
__device__ __forceinline__ void sc_mfma_pass(const float* __restrict__ A, const float* __restrict__ Wt, int lane, sc_f32x4 (&macc)[MAXM], double (&macc64)[MAXM][4]) {
    constexpr int NTILES = T * (T + 1) / 2;
    const float* rowp = A + (lane >> 4) * NPLP + (lane & 15);  // the 4 points of an MFMA step: k = lane >> 4
    const float* wp = Wt + (lane >> 4);
    for (int k0 = 0; k0 < SUB; k0 += RUN) {
#pragma unroll
        for (int st4 = 0; st4 < RUN / 4; ++st4) {
            const int k = k0 + 4 * st4;
            const float wk = wp[k];
            float x[T], xw[T];
#pragma unroll
            for (int c = 0; c < T; ++c) x[c] = rowp[k * NPLP + 16 * c];
#pragma unroll
            for (int c = 0; c < T; ++c) xw[c] = wk * x[c];
            sc_mfma_tiles<T, WAVE, NW, 0, MAXM>(x, xw, macc);
        }
#pragma unroll
        for (int m = 0; m < MAXM; ++m) {
            if (WAVE + NW * m >= NTILES) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) macc64[m][r] += (double)macc[m][r];
            macc[m] = (sc_f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
}
// The per-point sums ride in the same pipeline (round 4). KS == 4 (small windows, one pass of 64 points per workgroup): wave 0 sums its 64 points under the
// operand loads of the other columns. KS == 1 (large windows): the 256 / SUB threads that share a point split its W - 1 targets' shares between them - fetched
// with the pass' operand rows, i.e. in flight during the SYRK of the pass before -, park them in LDS (the region of the operand rows, which is dead between two
// passes), and the point's seven sums {bd, Hdd, Hcd[4], count} are then taken IN TARGET ORDER from there (an inactive slot parks +0: bit-identical to skipping
// it), one or two sums per thread; the threads of part 0 finish the point (HdiF, bdSumF), store what the back-substitution reads and write the point's two
// operand columns and its weight. Four barriers per pass instead of two, one launch and a pipeline drain less, and the shares' bytes move under the matrix work.
template <int T, int KS, int NW>
__global__ __launch_bounds__(64 * NW) void ba_sc_kernel(BADev B, int margOnly, int shiftPriorToZero, float priorScaleMarg) {
    constexpr int NPL = 16 * T, ROWS = kBlk / KS, NT = 64 * NW;
    constexpr int SUB0 = T <= 2 ? 256 : (T <= 4 ? 128 : 64), SUB = SUB0 < ROWS ? SUB0 : ROWS, NSUB = ROWS / SUB;
    // padded LDS row: a wave reads 4 consecutive rows x 16 columns per operand, so the stride is an odd multiple of 16 floats (the 4 rows land on
    // disjoint bank quarters)
    constexpr int NPLP = NPL + ((T % 2 == 0) ? 16 : 32);
    __shared__ __attribute__((aligned(16))) float A[SUB * NPLP];
    __shared__ float Wt[SUB];
    constexpr bool FUSE = KS == 4;                             // small windows: wave 0 sums its 64 points up front
    constexpr bool PIPE = KS == 1;                             // large windows: the point sums ride in the pass pipeline
    static_assert(FUSE || PIPE, "ba_sc_kernel: KS is 1 or 4");
    __shared__ __attribute__((aligned(16))) float4 Hc4[FUSE ? SUB : 1];
    __shared__ float Bd[FUSE ? SUB : 1];
    constexpr int MAXG = 2 * T - 1;                            // targets other than the host: 8 (W - 1) + 5 <= 16 T
    __shared__ float Rs[PIPE ? 7 * SUB : 1];                   // the seven sums of a pass' points
    static_assert(!PIPE || MAXG * 7 <= NPLP, "the parked shares of a pass must fit the operand rows they alias");
    const int grp = blockIdx.x / KS, ks = blockIdx.x - grp * KS, tid = threadIdx.x, W = B.W;
    int h = 0;
    while (h + 1 < W && grp >= B.sc_grp[h + 1]) ++h;
    const int b0 = B.host_blk[h] + (grp - B.sc_grp[h]) * B.sc_bpw, b1 = min(b0 + B.sc_bpw, B.host_blk[h + 1]);
    // fp32 products in short runs of 16 points flushed into fp64: the block partial is good to ~1e-8 relative (the reference's AccumulatorXX runs its
    // fp32 levels for 1000 adds, MatrixAccumulators.h), so the ~100x cancellation in H_A - H_sc does not amplify summation noise into the poses.
    // Round 4 measured 64-point runs: 4 % (250 k points) to 9 % (1 M) faster on this kernel, but the solved step of the well-conditioned toy window moves from
    // 1.0x to 1.6x the fp32 oracle's own distance to the fp64 truth (tests/test_ba_gpu.py::test_solve_and_step_on_the_well_conditioned_window): not taken.
#ifdef SC_RUN
    constexpr int RUN = SC_RUN < SUB ? SC_RUN : SUB;
#else
    constexpr int RUN = 16;
#endif
    // this wave's upper-triangular tiles (row-major enumeration of ti <= tj): wave, wave + 4, ...
    constexpr int NTILES = T * (T + 1) / 2, MAXM = (NTILES + NW - 1) / NW;
    const int wave = tid >> 6, lane = tid & 63;
    sc_f32x4 macc[MAXM];
    double macc64[MAXM][4];
#pragma unroll
    for (int m = 0; m < MAXM; ++m) {
        macc[m] = (sc_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) macc64[m][r] = 0.0;
    }
    // ---- staging: lane <-> point (coalesced 16-byte loads from the [target][point] arrays), the NQ float4 columns of a row are split over the
    // 256/SUB threads that share a point; fully unrolled, so every load of a pass is in flight at once.
    constexpr int NQ = NPL / 4, PARTS = NT / SUB, NV = (NQ + PARTS - 1) / PARTS;
    constexpr int NTG = PIPE ? (MAXG + PARTS - 1) / PARTS : 1;  // targets whose shares one thread fetches
    const int r = tid % SUB, part = __builtin_amdgcn_readfirstlane(tid / SUB);      // SUB >= 64: the column set of a thread is wave-uniform (scalar selects)
    const int qH = 2 * (W - 1);                                // float4 column of Hcd; qH + 1 = {bdSum, 0, 0, 0} (and carries HdiF to Wt)
    // fetch = loads only (one 16-byte + one state byte per column, source picked by pointer selects: no branches, no waits); the selects that need the
    // loaded bytes happen at commit time, after the SYRK of the previous pass
    float4 raw[NV];
    uint8_t rst[NV], rpf = 0;
    float4 sh0[NTG]; float2 sh1[NTG]; uint8_t shs[NTG];         // PIPE: this thread's part of the point's shares
    float pprior = 0.f; float4 pgeo = make_float4(0.f, 0.f, 0.f, 0.f); int dcur = 0;
    auto fetch = [&](int p) {
        const int pb = p / NSUB, d = (b0 + pb) * kBlk + ks * ROWS + (p - pb * NSUB) * SUB + r;
        rpf = B.pt_flags[d]; dcur = d;
#pragma unroll
        for (int qi = 0; qi < NV; ++qi) {
            const int q = part + qi * PARTS, g = q >> 1;          // q-th float4 of the row, compact target slot g
            const bool isj = q < qH;
            const size_t si = (size_t)(isj ? (g < h ? g : g + 1) : 0) * B.Ppad + d;
            const float4* src = isj ? ((q & 1) ? B.rs_jp1 : B.rs_jp0) + si : (q == qH ? B.pt_hcd : B.pt_acc) + d;
            rst[qi] = RS_ACTIVE;
            if (!isj) { raw[qi] = make_float4(0.f, 0.f, 0.f, 0.f); continue; }      // the point columns come from this workgroup's own sums (LDS)
            raw[qi] = *src;
            rst[qi] = B.rs_state[si];                              // scalar branch (q is wave-uniform)
        }
        if constexpr (PIPE) {
#pragma unroll
            for (int gi = 0; gi < NTG; ++gi) {
                const int g = part + gi * PARTS;                   // wave-uniform: a scalar branch around the loads, no wait
                shs[gi] = 0;
                if (g < W - 1) { const size_t si = (size_t)(g < h ? g : g + 1) * B.Ppad + d; shs[gi] = B.rs_state[si]; sh0[gi] = B.rs_pp0[si]; sh1[gi] = B.rs_pp1[si]; }
            }
            if (part == 0) { pprior = B.pt_prior[d]; if (shiftPriorToZero) pgeo = B.pt_geo[d]; }
        }
    };
    const int npass = (b1 - b0) * NSUB;
    if (npass > 0) fetch(0);
    if constexpr (FUSE) {
        // per point (AccumulatedSCHessian.cpp:36-57) for the 64 points of this workgroup: all slots of 8 targets are fetched before any is used (latency matters here)
        if (tid < SUB && npass > 0) {
            const int d = b0 * kBlk + ks * ROWS + tid;
            const uint8_t pf = B.pt_flags[d];
            float wgt = 0.f, bds = 0.f;
            ScPoint P; P.pa = make_float4(0.f, 0.f, 0.f, 0.f); P.hc = make_float4(0.f, 0.f, 0.f, 0.f); P.ngood = 0;
            if ((pf & PT_VALID) && (!margOnly || (pf & PT_MARG))) {
                for (int t0 = 0; t0 < W; t0 += 8) {
                    uint8_t rs[8]; float4 q0[8]; float2 q1[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        rs[i] = 0;
                        if (t0 + i < W) { const size_t si = (size_t)(t0 + i) * B.Ppad + d; rs[i] = B.rs_state[si]; q0[i] = B.rs_pp0[si]; q1[i] = B.rs_pp1[si]; }
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int t = t0 + i;
                        if (t >= W || t == h || !(rs[i] & RS_ACTIVE)) continue;
                        P.pa.y += q0[i].x; P.pa.x += q0[i].y; P.hc.x += q0[i].z; P.hc.y += q0[i].w; P.hc.z += q1[i].x; P.hc.w += q1[i].y;
                        ++P.ngood;
                    }
                }
                const float prior = B.pt_prior[d];
                float4 geo = make_float4(0.f, 0.f, 0.f, 0.f);
                if (shiftPriorToZero && P.ngood) geo = B.pt_geo[d];
                sc_point_finish(B, d, P, prior, geo.z, geo.w, margOnly, priorScaleMarg, shiftPriorToZero);
                wgt = P.pa.z; bds = P.pa.w;
            }
            Wt[tid] = wgt; Bd[tid] = bds; Hc4[tid] = P.hc;
        }
    }
    for (int p = 0; p < npass; ++p) {
        __syncthreads();                                       // the previous pass' rows / weights have been consumed
        const bool pvalid = (rpf & PT_VALID) && (!margOnly || (rpf & PT_MARG));
        if constexpr (PIPE) {
            // park this thread's shares ([target][component][point]: conflict-free), then the seven sums in target order
            float* S = A;
#pragma unroll
            for (int gi = 0; gi < NTG; ++gi) {
                const int g = part + gi * PARTS;
                if (g >= MAXG) continue;
                const bool act = (shs[gi] & RS_ACTIVE) != 0;           // shs = 0 beyond the window's last target
                float* sp = S + (g * 7) * SUB + r;
                sp[0] = act ? sh0[gi].x : 0.f; sp[SUB] = act ? sh0[gi].y : 0.f; sp[2 * SUB] = act ? sh0[gi].z : 0.f; sp[3 * SUB] = act ? sh0[gi].w : 0.f;
                sp[4 * SUB] = act ? sh1[gi].x : 0.f; sp[5 * SUB] = act ? sh1[gi].y : 0.f; sp[6 * SUB] = act ? 1.f : 0.f;
            }
            __syncthreads();
            for (int cc = part; cc < 7; cc += PARTS) {
                float sum = 0.f;
                for (int g = 0; g < W - 1; ++g) sum += S[(g * 7 + cc) * SUB + r];
                Rs[cc * SUB + r] = sum;
            }
            __syncthreads();                                   // the parked shares are consumed: the operand rows may overwrite them
        }
        {
#pragma unroll
            for (int qi = 0; qi < NV; ++qi) {
                const int q = part + qi * PARTS;
                float4 v = raw[qi];
                if constexpr (FUSE) { if (q == qH) v = Hc4[r]; else if (q == qH + 1) v = make_float4(Bd[r], 0.f, 0.f, 0.f); }
                if constexpr (PIPE) { if (q == qH || q == qH + 1) continue; }       // written by the point's part-0 thread below
                const bool keep = pvalid && (rst[qi] & RS_ACTIVE) != 0 && q <= qH + 1;
                if (!keep) v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (q < NQ) *reinterpret_cast<float4*>(&A[r * NPLP + 4 * q]) = v;
            }
            if constexpr (PIPE) {
                if (part == 0) {
                    ScPoint P;
                    P.pa = make_float4(Rs[SUB + r], Rs[r], 0.f, 0.f);                 // {Hdd, bd}
                    P.hc = make_float4(Rs[2 * SUB + r], Rs[3 * SUB + r], Rs[4 * SUB + r], Rs[5 * SUB + r]);
                    P.ngood = (int)Rs[6 * SUB + r];
                    float wgt = 0.f; float4 c0 = make_float4(0.f, 0.f, 0.f, 0.f), c1 = c0;
                    if (pvalid) {
                        sc_point_finish(B, dcur, P, pprior, pgeo.z, pgeo.w, margOnly, priorScaleMarg, shiftPriorToZero);
                        wgt = P.pa.z; c0 = P.hc; c1.x = P.pa.w;
                    }
                    Wt[r] = wgt;
                    *reinterpret_cast<float4*>(&A[r * NPLP + 4 * qH]) = c0;
                    *reinterpret_cast<float4*>(&A[r * NPLP + 4 * (qH + 1)]) = c1;
                }
            }
        }
        __syncthreads();
        if (p + 1 < npass) fetch(p + 1);                       // in flight during the SYRK below
        switch (wave) {
            case 0: sc_mfma_pass<T, 0, NW, SUB, NPLP, RUN, MAXM>(A, Wt, lane, macc, macc64); break;
            case 1: sc_mfma_pass<T, 1, NW, SUB, NPLP, RUN, MAXM>(A, Wt, lane, macc, macc64); break;
            case 2: sc_mfma_pass<T, 2, NW, SUB, NPLP, RUN, MAXM>(A, Wt, lane, macc, macc64); break;
            case 3: sc_mfma_pass<T, 3, NW, SUB, NPLP, RUN, MAXM>(A, Wt, lane, macc, macc64); break;
            case 4: if constexpr (NW > 4) sc_mfma_pass<T, 4, NW, SUB, NPLP, RUN, MAXM>(A, Wt, lane, macc, macc64); break;
            case 5: if constexpr (NW > 4) sc_mfma_pass<T, 5, NW, SUB, NPLP, RUN, MAXM>(A, Wt, lane, macc, macc64); break;
            case 6: if constexpr (NW > 4) sc_mfma_pass<T, 6, NW, SUB, NPLP, RUN, MAXM>(A, Wt, lane, macc, macc64); break;
            default: if constexpr (NW > 4) sc_mfma_pass<T, 7, NW, SUB, NPLP, RUN, MAXM>(A, Wt, lane, macc, macc64); break;
        }
    }
    // C/D layout of a 16x16 tile: col = lane & 15, row = (lane >> 4) * 4 + reg
    double* out = B.sc_partial + (size_t)blockIdx.x * (NTILES * 256);
#pragma unroll
    for (int m = 0; m < MAXM; ++m) {
        const int idx = wave + NW * m;
        if (idx >= NTILES) continue;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) out[idx * 256 + rr * 64 + lane] = macc64[m][rr];
    }
}

#ifndef SC_NW
#define SC_NW 8
#endif
// large windows run the SYRK with SC_NW waves per workgroup: eight waves hold two or three 16x16 tiles each instead of five or six (the fp32 + fp64 accumulators
// of six tiles cost 72 registers: two waves per SIMD), so four waves per SIMD take turns on the matrix pipe
template <int KS>
static void launch_sc_ks(hipStream_t s, const BADev& B, int T, int margOnly, int shift, float priorScaleMarg) {
    constexpr int NW = KS == 1 ? SC_NW : 4;
    const int grid = B.sc_groups * KS;
    switch (T) {
        case 1: ba_sc_kernel<1, KS, 4><<<grid, 256, 0, s>>>(B, margOnly, shift, priorScaleMarg); break;        // one tile: nothing to deal out
        case 2: ba_sc_kernel<2, KS, 4><<<grid, 256, 0, s>>>(B, margOnly, shift, priorScaleMarg); break;
        case 3: ba_sc_kernel<3, KS, NW><<<grid, 64 * NW, 0, s>>>(B, margOnly, shift, priorScaleMarg); break;
        case 4: ba_sc_kernel<4, KS, NW><<<grid, 64 * NW, 0, s>>>(B, margOnly, shift, priorScaleMarg); break;
        case 5: ba_sc_kernel<5, KS, NW><<<grid, 64 * NW, 0, s>>>(B, margOnly, shift, priorScaleMarg); break;
        case 6: ba_sc_kernel<6, KS, NW><<<grid, 64 * NW, 0, s>>>(B, margOnly, shift, priorScaleMarg); break;
        case 7: ba_sc_kernel<7, KS, NW><<<grid, 64 * NW, 0, s>>>(B, margOnly, shift, priorScaleMarg); break;
        default: ba_sc_kernel<8, KS, NW><<<grid, 64 * NW, 0, s>>>(B, margOnly, shift, priorScaleMarg); break;
    }
}
void ba_launch_sc(hipStream_t s, const BADev& B, int T, int shift, float priorScaleMarg, int margOnly) {
    if (B.sc_split == 4) launch_sc_ks<4>(s, B, T, margOnly, shift, priorScaleMarg);     // either way ONE launch: the per-point sums are part of it
    else launch_sc_ks<1>(s, B, T, margOnly, shift, priorScaleMarg);
}
// frameEnergyTH of every window frame as kernel arguments: no staging buffer, no synchronisation (window setup / restore)
struct ThArg { float v[16]; };
__global__ void ba_set_th_kernel(float* __restrict__ dst, ThArg a, int W) { if ((int)threadIdx.x < W) dst[threadIdx.x] = a.v[threadIdx.x]; }
void ba_launch_set_th(hipStream_t s, float* dst, const float* th, int W) {
    ThArg a;
    for (int i = 0; i < 16; ++i) a.v[i] = i < W ? th[i] : 0.f;
    ba_set_th_kernel<<<1, 16, 0, s>>>(dst, a, W);
}
// nalo_ba_restore (bench / test utility) in ONE launch: the mutable device state of the window back from its snapshot, the residual energies zeroed, the frames'
// energy thresholds installed (was four device-to-device copies, a fill and a 16-lane kernel: ~60 us of launches and gaps per replayed keyframe)
__global__ __launch_bounds__(256) void ba_restore_kernel(BADev B, const float4* __restrict__ geo, const uint8_t* __restrict__ state, const uint8_t* __restrict__ flags,
                                                         const float* __restrict__ prior, ThArg th) {
    const size_t N = (size_t)B.Ppad, NS = (size_t)B.W * N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < NS; i += (size_t)gridDim.x * blockDim.x) {
        B.rs_state[i] = state[i];
        B.rs_energy[i] = make_float2(0.f, 0.f);
        if (i < N) { B.pt_geo[i] = geo[i]; B.pt_flags[i] = flags[i]; B.pt_prior[i] = prior[i]; }
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < B.W) B.frameTH[threadIdx.x] = th.v[threadIdx.x];
}
void ba_launch_restore(hipStream_t s, const BADev& B, const float4* geo, const uint8_t* state, const uint8_t* flags, const float* prior, const float* th) {
    ThArg a;
    for (int i = 0; i < 16; ++i) a.v[i] = i < B.W ? th[i] : 0.f;
    const size_t NS = (size_t)B.W * B.Ppad;
    ba_restore_kernel<<<(unsigned)std::min<size_t>((NS + 255) / 256, 16384), 256, 0, s>>>(B, geo, state, flags, prior, a);
}
void ba_launch_reset_oob(hipStream_t s, const BADev& B) {
    const size_t n = (size_t)B.W * B.Ppad;
    ba_reset_oob_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(B.rs_state, B.rs_energy, n);
}

// Small windows (<= 16384 point slots, a KITTI-sized window has 2048): the same order statistic in ONE launch. The workgroup keeps the energies in
// registers and runs a most-significant-first radix select with four 256-bin LDS histograms (5 us instead of the 40 us of the two 65536-bin
// searches of round 1, which sat between the publish and the next back-substitution).
__device__ __forceinline__ void th_small_body(const float* __restrict__ en, int n, float* __restrict__ frameTH_new) {
    __shared__ unsigned hist[256], wtot[16];
    __shared__ unsigned s_bin, s_before;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned v[16]; bool ok[16];
    unsigned cnt = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int i = tid + 1024 * j;
        const float f = i < n ? en[i] : -1.f;
        ok[j] = f >= 0.f; v[j] = __float_as_uint(f);
        if (ok[j]) ++cnt;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o);
    if (lane == 0) wtot[wave] = cnt;
    __syncthreads();
    unsigned total = 0;
    for (int i = 0; i < 16; ++i) total += wtot[i];
    if (total == 0) { if (tid == 0) __hip_atomic_store(frameTH_new, 12.f * 12.f * (float)kPatternNum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }       // no residual on the newest frame (:110-114)
    unsigned k = (unsigned)(int)(kFrameEnergyTHN * (float)total), prefix = 0, mask = 0;
    for (int shift = 24; shift >= 0; shift -= 8) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j) if (ok[j] && (v[j] & mask) == prefix) atomicAdd(&hist[(v[j] >> shift) & 255u], 1u);
        __syncthreads();
        if (wave == 0) {                                        // 4 bins per lane, shuffle scan, the lane and then the bin whose running count passes k
            const unsigned h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3], sum = h0 + h1 + h2 + h3;
            unsigned incl = sum;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(incl, o); if (lane >= o) incl += t; }
            const unsigned excl = incl - sum;
            if (excl <= k && k < incl) {
                unsigned run = excl; int b = 3;
                if (run + h0 > k) b = 0; else { run += h0; if (run + h1 > k) b = 1; else { run += h1; if (run + h2 > k) b = 2; else run += h2; } }
                s_bin = 4u * lane + b; s_before = run;
            }
        }
        __syncthreads();
        prefix |= s_bin << shift; mask |= 255u << shift; k -= s_before;
        __syncthreads();
    }
    if (tid == 0) {
        const float nthElement = sqrtf(__uint_as_float(prefix));
        float th = nthElement * kFrameEnergyTHFacMedian;        // FullSystemOptimize.cpp:130-133
        th = 26.0f * kFrameEnergyTHConstWeight + th * (1.f - kFrameEnergyTHConstWeight);
        th = th * th; th *= kOverallEnergyTHWeight * kOverallEnergyTHWeight;
        __hip_atomic_store(frameTH_new, th, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // agent scope: ba_reduce_kernel's publishing workgroup may read it in the same launch
    }
}
__global__ __launch_bounds__(1024) void ba_th_small_kernel(const float* __restrict__ en, int n, float* __restrict__ frameTH_new) { th_small_body(en, n, frameTH_new); }

// ------------------------------------------------------------------------------------------------ fp64 finish of the partials
// ONE launch for both systems (blocks [0, W*W) = top bins, the rest = 64 entries of one upper SC tile of one host):
//   acc13[(h + t*W)][169] (full symmetric 13x13, AccumulatorApprox::finish layout MatrixAccumulators.h:626-647), misc[h+t*W] = {count, energy}
//   G[h][row][col] = G[h][col][row] = sum over the host's workgroups of the compact SYRK partials (MFMA register order, see ba_sc_kernel)
// Lane groups stride over the host's blocks (8 loads in flight per lane) and are combined in a fixed order: deterministic.
__global__ __launch_bounds__(1024) void ba_reduce_kernel(const double* __restrict__ top_partial, const double* __restrict__ sc_partial,
                                                         const int* __restrict__ host_blk /* [W+1] */, const int* __restrict__ sc_grp /* [W+1] */, int W, int NPL, int sc_tiles, int mask, int KS, int lin_sub,
                                                         double* __restrict__ acc13, double* __restrict__ misc, double* __restrict__ G,
                                                         const float* __restrict__ step_partial, int step_blocks, double* __restrict__ step_out, RedExtra X) {
    __shared__ double part[16][64];
    __shared__ double sums[128];
    __shared__ int is_last;
    const int nsc = W * sc_tiles, b_step = W * W + nsc, b_th = b_step + (step_partial ? 1 : 0);
    if (X.th_en && (int)blockIdx.x == b_th) th_small_body(X.th_en, X.th_n, X.th_out);      // the newest frame's threshold rides along (small windows)
    else if ((int)blockIdx.x < W * W) {
        if (mask & 1) {
        double (*part8)[128] = reinterpret_cast<double (*)[128]>(&part[0][0]);
        const int h = blockIdx.x % W, t = blockIdx.x / W, j = threadIdx.x & 127, g = threadIdx.x >> 7;     // 8 groups stride over the blocks
        double s = 0;
        if (j < kTopVals && h != t) {
            const int bend = host_blk[h + 1] * lin_sub;         // lin_sub partials per point block (kernels_ba_lin.hip)
            int b = host_blk[h] * lin_sub + g;
            double u[8];
            for (; b + 56 < bend; b += 64) {                    // 8 independent loads per round: the host's blocks are a long latency-bound walk
#pragma unroll
                for (int k = 0; k < 8; ++k) u[k] = top_partial[((size_t)(b + 8 * k) * W + t) * kTopStride + j];
                s += ((u[0] + u[1]) + (u[2] + u[3])) + ((u[4] + u[5]) + (u[6] + u[7]));
            }
            for (; b < bend; b += 8) s += top_partial[((size_t)b * W + t) * kTopStride + j];
        }
        part8[g][j] = s;
        __syncthreads();
        if (g == 0) { double tt = 0; for (int k = 0; k < 8; ++k) tt += part8[k][j]; sums[j] = tt; }
        __syncthreads();
        double* H = acc13 + (size_t)(h + t * W) * 169;
        for (int e = threadIdx.x; e < 169; e += blockDim.x) {
            int r = e / 13, c = e % 13;
            if (r > c) { const int tmp = r; r = c; c = tmp; }
            int idx;
            if (c < 10) idx = r * 10 - r * (r - 1) / 2 + (c - r);                 // upper triangle of the 10x10, row-major r<=c
            else if (r < 10) idx = 55 + 3 * r + (c - 10);                         // TopRight 10x3
            else { const int rr = r - 10, cc = c - 10; idx = 85 + (rr == 0 ? cc : rr + cc + 1); }   // BotRight: 00 01 02 11 12 22
            H[e] = sums[idx];
        }
        // (agent-scope stores: the workgroup that publishes the tail of a misc-only fetch reads them below, without a device-wide fence = an L2 write-back per workgroup)
        if (threadIdx.x == 0) { __hip_atomic_store(&misc[2 * (h + t * W)], sums[91], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __hip_atomic_store(&misc[2 * (h + t * W) + 1], sums[92], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        }
    } else if (step_partial && (int)blockIdx.x == b_step) {     // the deferred sums of doStepFromBackup's break test ride along (optimize())
        // all 1024 lanes stride over the blocks (a 1M-point window has 3907: 16 lanes walking them was an 85 us serial chain), fixed-order tree
        const int j = threadIdx.x & 63, g = threadIdx.x >> 6;
        double s0 = 0, s1 = 0, s2 = 0;
        for (int b = threadIdx.x; b < step_blocks; b += 1024) {
            const float4 v = reinterpret_cast<const float4*>(step_partial)[b];
            s0 += (double)v.x; s1 += (double)v.y; s2 += (double)v.z;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { s0 += __shfl_down(s0, o); s1 += __shfl_down(s1, o); s2 += __shfl_down(s2, o); }
        if (j == 0) { part[g][0] = s0; part[g][1] = s1; part[g][2] = s2; }
        __syncthreads();
        if (threadIdx.x < 3) { double t = 0; for (int k = 0; k < 16; ++k) t += part[k][threadIdx.x]; __hip_atomic_store(&step_out[threadIdx.x], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    } else if (mask & 2) {
    const int q = blockIdx.x - W * W, h = q / sc_tiles, tile = q - h * sc_tiles;      // tile = 4 * (upper tile index) + MFMA register
    const int j = threadIdx.x & 63, g = threadIdx.x >> 6, e = tile * 64 + j, psz = sc_tiles * 64;
    double s = 0;
    for (int b = sc_grp[h] * KS + g; b < sc_grp[h + 1] * KS; b += 16) s += sc_partial[(size_t)b * psz + e];
    part[g][j] = s;
    __syncthreads();
    if (g == 0) {
        double tt = 0;
        for (int k = 0; k < 16; ++k) tt += part[k][j];
        const int T = NPL >> 4, rr = tile & 3;
        int ti = 0, tj = tile >> 2;
        while (tj >= T - ti) { tj -= T - ti; ++ti; }
        tj += ti;
        const int row = 16 * ti + (j >> 4) * 4 + rr, col = 16 * tj + (j & 15);
        double* Gh = G + (size_t)h * NPL * NPL;
        Gh[row * NPL + col] = tt;
        if (ti != tj) Gh[col * NPL + row] = tt;
    }
    }
    // misc-only fetch (the last pass of optimize(): energy, residual count, threshold - no systems): the last workgroup to finish publishes the tail
    // {misc (2 W^2), step sums (3), TH, 1.0} and the sequence number the host polls, instead of a th_tail + a publish launch behind this one
    if (!X.pub) return;
    // the stores above are agent-scope atomics, acknowledged before this workgroup's ticket (release fence = s_waitcnt; the idiom of ba_stitch_kernel's tail)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0) is_last = __hip_atomic_fetch_add(X.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u;
    __syncthreads();
    if (!is_last) return;
    const int ntail = 2 * W * W + 5;
    for (int i = threadIdx.x; i < ntail; i += blockDim.x) {
        double v;
        if (i == ntail - 2) v = (double)__hip_atomic_load(X.th_src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // what tail_th() reads: {TH, 1.0}
        else if (i == ntail - 1) v = 1.0;
        else v = __hip_atomic_load(&misc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        misc[i] = v;                                           // the device copy keeps the same tail (a later publish re-sends it)
        __hip_atomic_store(&X.pub[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_store(X.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                  // re-armed for the next (stream-ordered) launch
        __hip_atomic_store(&X.pub[ntail], X.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// with_th: the window is small and a linearisation's threshold is pending - it is computed by one more workgroup of this launch. pub: misc-only fetch, the tail is
// published from here (the Schur-complement blocks are not launched at all then)
void ba_launch_reduce(hipStream_t s, const BADev& B, const int* host_blk, int NPL, double* acc13, double* misc, double* G, bool top, bool sc,
                      const float* step_partial, int step_blocks, double* step_out, bool with_th, double* pub, double seq, unsigned* ticket) {
    const int T = NPL / 16, tiles = pub ? 0 : T * (T + 1) / 2 * 4;
    RedExtra X{};
    if (with_th) { X.th_en = B.en_new; X.th_n = B.Ppad; X.th_out = B.frameTH + (B.W - 1); }
    X.pub = pub; X.seq = seq; X.ticket = ticket; X.th_src = B.frameTH + (B.W - 1);
    ba_reduce_kernel<<<B.W * B.W + B.W * tiles + (step_partial ? 1 : 0) + (with_th ? 1 : 0), 1024, 0, s>>>(B.top_partial, B.sc_partial, host_blk, B.sc_grp, B.W, NPL, tiles,
                                                                                       (top ? 1 : 0) | (sc && !pub ? 2 : 0), B.sc_split, B.lin_sub, acc13, misc, G, step_partial, step_blocks, step_out, X);
}

// ------------------------------------------------------------------------------------------------ stitch:  H~ = sum_b S_b M_b S_b^T  (fp64)
// AccumulatedTopHessian::stitchDoubleInternal (AccumulatedTopHessian.cpp:243-330) and AccumulatedSCHessian::stitchDoubleInternal
// (AccumulatedSCHessian.cpp:117-180) as ONE launch. H~ is (8W+5)^2: row/col 0-3 calib, 4+8f.. frame f, and 8W+4 = the b vector.
//   top: bin b = (h,t) holds M_b (13x13: calib 4 | local 8 | residual 1); rows of frame h see it through adHost[b], rows of t through adTarget[b]
//   SC : host i holds G_i (NPL x NPL: 8 columns per compact target slot | Hcd 4 | bdSum 1); rows of i see slot g through adHost[i,j_g], rows of
//        target j see their slot through adTarget[i,j]
// One workgroup per (system, frame a) computes the 8 rows of frame a: phase 1 U = L_a M (the frame's adjoints times the accumulator rows, into
// LDS), phase 2 H~[rows a][c] = U L_c^T with direct block addressing (fixed-trip 8-wide loops: no index lists, no dependent loads). The calib/b
// rows are the mirror of the calib/b columns (H~ is symmetric, as in the reference which copies the transposed blocks) plus a 5x5 corner done by
// one more workgroup per system. Fixed summation order: deterministic. The kernel also publishes: every workgroup stores its rows into
// host-mapped pinned memory, the last one to take a ticket adds the scalar tail and the sequence number the host polls on.
// SC rows of frame a. FAST: every operand is staged in LDS with coalesced single-pass loads (phase 1: G_a, the slot rows of the other hosts'
// G_i and frame a's adjoints; phase 2: all adjoints, over the same region) so the inner 8-wide loops run on ds_read with no global latency
// in the dependency chain. The generic path (large windows, LDS too small) reads G and the adjoints through L2.
#ifdef NALO_STITCH_TICKS
#define STITCH_TICK(i) do { __syncthreads(); tk[i] = clock64(); } while (0)
#else
#define STITCH_TICK(i) do { } while (0)
#endif
constexpr int kAdRow = 9, kAdMat = 73;        // padded 8x8 adjoint in LDS
// WC: the window size as a compile-time constant (0 = run time). With it the 2 x (W - 1) x 8-term sums of phase 2 and the 56-term sums of phase 1 unroll, and
// their LDS loads are issued in batches instead of one dependent group per loop trip (W = 8, headline window: phase 2 9.4 -> see DESIGN 5).
template <bool FAST, int WC, typename Put>
__device__ __forceinline__ void stitch_sc_rows(const StitchDev& D, double* lds, int a, int r0, int nr, Put put) {      // rows r0 .. r0 + nr - 1 of frame a's eight
    const int W = WC ? WC : D.W, n1 = WC ? 8 * WC + 5 : D.n1, n = n1 - 1, NPL = WC ? 16 * ((8 * (WC - 1) + 5 + 15) / 16) : D.NPL, tid = threadIdx.x, NT = blockDim.x, cb = 8 * (W - 1);
#ifdef NALO_STITCH_TICKS
    long long tk[5] = {0, 0, 0, 0, 0};
#endif
    double* U = lds;                                           // [W][8][NPL]
    double* R = U + W * 8 * NPL;                               // staging region (FAST)
    STITCH_TICK(0);
    if constexpr (FAST) {
        double* Ga = R;                                        // [NPL][NPL]
        double* Gi = Ga + NPL * NPL;                           // [W][8][NPL]: rows of slot g_i(a) of host i != a
        double* Ah = Gi + W * 8 * NPL;                         // [W][64] adHost(a, j)
        double* At = Ah + W * 64;                              // [W][64] adTarget(i, a)
        const double* __restrict__ Gsrc = D.M_sc;
        for (int e = tid; e < NPL * NPL; e += NT) Ga[e] = Gsrc[(size_t)a * NPL * NPL + e];
        for (int e = tid; e < W * 8 * NPL; e += NT) {
            const int i = e / (8 * NPL), o = e - i * 8 * NPL;
            if (i != a) Gi[e] = Gsrc[(size_t)i * NPL * NPL + (size_t)(8 * (a < i ? a : a - 1)) * NPL + o];
        }
        for (int e = tid; e < W * 64; e += NT) {
            const int f = e >> 6, o = e & 63;
            if (f != a) { Ah[e] = D.AD[(size_t)(a + f * W) * 64 + o]; At[e] = D.AD[(size_t)W * W * 64 + (size_t)(f + a * W) * 64 + o]; }
        }
        __syncthreads();
        STITCH_TICK(1);
        for (int e = tid; e < W * nr * NPL; e += NT) {
            const int i = e / (nr * NPL), o = e - i * nr * NPL, r = r0 + o / NPL, l = o % NPL;
            double s = 0;
            if (i == a) {
#pragma unroll
                for (int g2 = 0; g2 < W - 1; ++g2) {
                    const double* A = Ah + (g2 < a ? g2 : g2 + 1) * 64 + r * 8;
#pragma unroll
                    for (int k = 0; k < 8; ++k) s += A[k] * Ga[(8 * g2 + k) * NPL + l];
                }
            } else {
                const double* A = At + i * 64 + r * 8;
#pragma unroll
                for (int k = 0; k < 8; ++k) s += A[k] * Gi[(i * 8 + k) * NPL + l];
            }
            U[(i * 8 + r) * NPL + l] = s;
        }
        __syncthreads();
        STITCH_TICK(2);
        // padded copy of all adjoints: row stride 9, matrix stride 73 doubles -> the (cp, j) pattern of phase 2 spreads over the LDS banks
        for (int e = tid; e < 2 * W * W * 64; e += NT) R[(e >> 6) * kAdMat + ((e >> 3) & 7) * kAdRow + (e & 7)] = D.AD[e];
        __syncthreads();
        STITCH_TICK(3);
    } else {
        for (int e = tid; e < W * nr * NPL; e += NT) {
            const int i = e / (nr * NPL), o = e - i * nr * NPL, r = r0 + o / NPL, l = o % NPL;
            const double* __restrict__ G = D.M_sc + (size_t)i * NPL * NPL;
            double s = 0;
            if (i == a) {
                for (int g2 = 0; g2 < W - 1; ++g2) {
                    const double* __restrict__ A = D.AD + (size_t)(a + (g2 < a ? g2 : g2 + 1) * W) * 64 + r * 8;
#pragma unroll
                    for (int k = 0; k < 8; ++k) s += A[k] * G[(size_t)(8 * g2 + k) * NPL + l];
                }
            } else {
                const double* __restrict__ A = D.AD + (size_t)W * W * 64 + (size_t)(i + a * W) * 64 + r * 8;
#pragma unroll
                for (int k = 0; k < 8; ++k) s += A[k] * G[(size_t)(8 * (a < i ? a : a - 1) + k) * NPL + l];
            }
            U[(i * 8 + r) * NPL + l] = s;
        }
        __syncthreads();
    }
    auto phase2 = [&](const double* adH, const double* adT, const int MS, const int RS) {
        for (int e = tid; e < nr * n1; e += NT) {
            const int r = r0 + e / n1, c = e % n1;
            double s = 0;
            if (c < 4) { for (int i = 0; i < W; ++i) s += U[(i * 8 + r) * NPL + cb + c]; }
            else if (c == n) { for (int i = 0; i < W; ++i) s += U[(i * 8 + r) * NPL + cb + 4]; }
            else {
                // two uniform passes (no lane divergence between the 56-term own-host sum and the 8-term sums of the other hosts)
                const int j = (c - 4) >> 3, cp = (c - 4) & 7;
                const double* Uj = U + (j * 8 + r) * NPL;
                double s4[4] = {0, 0, 0, 0};                                   // four chains: fp64 FMA latency is 4x its issue rate
#pragma unroll
                for (int g2 = 0; g2 < W - 1; ++g2) {                           // host j itself: every slot of its G, through adHost(j, .)
                    const double* A = adH + (j + (g2 < j ? g2 : g2 + 1) * W) * MS + cp * RS;
#pragma unroll
                    for (int k = 0; k < 8; ++k) s4[k & 3] += Uj[8 * g2 + k] * A[k];
                }
#pragma unroll
                for (int i = 0; i < W; ++i) {                                  // the other hosts: frame j's slot, through adTarget(i, j)
                    if (i == j) continue;
                    const double* Ui = U + (i * 8 + r) * NPL + 8 * (j < i ? j : j - 1);
                    const double* A = adT + (i + j * W) * MS + cp * RS;
#pragma unroll
                    for (int k = 0; k < 8; ++k) s4[k & 3] += Ui[k] * A[k];
                }
                s = (s4[0] + s4[1]) + (s4[2] + s4[3]);
            }
            put(4 + 8 * a + r, c, s);
            if (c < 4 || c == n) put(c, 4 + 8 * a + r, s);
        }
    };
    if constexpr (FAST) phase2(R, R + W * W * kAdMat, kAdMat, kAdRow); else phase2(D.AD, D.AD + (size_t)W * W * 64, 64, 8);
    STITCH_TICK(4);
#ifdef NALO_STITCH_TICKS
    if (tid == 0) printf("sc a=%d stage1=%lld ph1=%lld stage2=%lld ph2=%lld\n", a, tk[1] - tk[0], tk[2] - tk[1], tk[3] - tk[2], tk[4] - tk[3]);
#endif
}

#ifndef NALO_STITCH_SC_SPLIT
#define NALO_STITCH_SC_SPLIT 8
#endif
#ifndef NALO_STITCH_TOP_SPLIT
#define NALO_STITCH_TOP_SPLIT 8
#endif
constexpr int kTopSplit = NALO_STITCH_TOP_SPLIT;     // workgroups per frame for the rows of the top system (8 / kTopSplit rows each): with the Schur-complement rows on
                                                     // eight workgroups the top system's one workgroup per frame was the long pole. 1 / 2 / 4 / 8: 11.2 / 10.3 / 11.3 / 10.7 us
                                                     // at W = 8 (noise ~0.5), 14.3 / 13.1 / 12.0 / 10.9 us at W = 12
constexpr int kScSplit = NALO_STITCH_SC_SPLIT;       // workgroups per frame for the Schur-complement rows (8 / kScSplit rows each). Round 3, kernel trace: 2 / 4 / 8 workgroups
                                                     // = 13.6 / 12.1 / 11.3 us at W = 8 and 22.7 / 17.0 / 14.4 us at W = 12 (phase 2 is LDS-bandwidth bound; the operand
                                                     // staging every workgroup repeats is the smaller part). Same loops per output element: bit-identical results
__global__ __launch_bounds__(1024) void ba_stitch_kernel(StitchDev D, int mask, int ad_in_lds, double* mapped, int ntail, double seq) {
    extern __shared__ double lds[];
    __shared__ int is_last;
    const int W = D.W, n1 = D.n1, n = n1 - 1, NPL = D.NPL, tid = threadIdx.x, NT = blockDim.x;
    // workgroups: [0, W] the top system (W frame rows + the corner), then kScSplit W for the Schur-complement system - kScSplit per frame, 8 / kScSplit of its eight
    // rows each (phase 2 is bound by LDS bandwidth: 2 x 112 fp64 operands per output; more workgroups on more CUs divide it) - then its corner
    const int ntop = kTopSplit * W + 1;                        // top-system workgroups: kTopSplit per frame, then the corner
    const int sys = (int)blockIdx.x >= ntop ? 1 : 0;
    const int gb = blockIdx.x - ntop;
    const int g = sys ? (gb < kScSplit * W ? gb / kScSplit : W) : ((int)blockIdx.x < kTopSplit * W ? (int)blockIdx.x / kTopSplit : W);
    const int r0 = sys ? (gb % kScSplit) * (8 / kScSplit) : ((int)blockIdx.x % kTopSplit) * (8 / kTopSplit);
    const int nr_top = 8 / kTopSplit;
    const double* __restrict__ adH = D.AD;
    const double* __restrict__ adT = D.AD + (size_t)W * W * 64;
    double* Hs = D.H + (size_t)sys * n1 * n1;
    double* Ms = mapped ? mapped + (size_t)sys * n1 * n1 : nullptr;
    auto put = [&](int r, int c, double v) {
        Hs[(size_t)r * n1 + c] = v;
        if (Ms) __hip_atomic_store(&Ms[(size_t)r * n1 + c], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    };
    if ((mask >> sys) & 1) {
        if (sys == 0 && g < W) {
            const int a = g, nb = 2 * (W - 1);                 // bins with frame a: q < W-1 -> (a,t) [a is host], q >= W-1 -> (h,a) [a is target]
            double* Mb = lds;                                  // [nb][169]
            double* As = Mb + nb * 169;                        // [nb][kAdMat] adjoint of a's own role in the bin (padded rows: bank spread)
            double* Ao = As + nb * kAdMat;                     // [nb][kAdMat] adjoint of the other frame's role
            double* U = Ao + nb * kAdMat;                      // [nb][8][13]
            auto bin_of = [&](int q) { const int o = q < W - 1 ? q : q - (W - 1), f = o < a ? o : o + 1; return q < W - 1 ? a + f * W : f + a * W; };
            for (int e = tid; e < nb * 169; e += NT) { const int q = e / 169; Mb[e] = D.M_top[(size_t)bin_of(q) * 169 + (e - q * 169)]; }
            for (int e = tid; e < nb * 64; e += NT) {
                const int q = e >> 6, o = e & 63, bin = bin_of(q);
                const bool host_role = q < W - 1;
                As[q * kAdMat + (o >> 3) * kAdRow + (o & 7)] = (host_role ? adH : adT)[(size_t)bin * 64 + o];
                Ao[q * kAdMat + (o >> 3) * kAdRow + (o & 7)] = (host_role ? adT : adH)[(size_t)bin * 64 + o];
            }
            __syncthreads();
            for (int e = tid; e < nb * nr_top * 13; e += NT) {          // rows r0 .. r0 + nr_top - 1 of this workgroup only
                const int q = e / (nr_top * 13), o = e - q * (nr_top * 13), r = r0 + o / 13, l = o % 13;
                double s = 0;
#pragma unroll
                for (int k = 0; k < 8; ++k) s += As[q * kAdMat + r * kAdRow + k] * Mb[q * 169 + (4 + k) * 13 + l];
                U[q * 104 + r * 13 + l] = s;
            }
            __syncthreads();
            for (int e = tid; e < nr_top * n1; e += NT) {
                const int r = r0 + e / n1, c = e % n1;
                double s = 0;
                if (c < 4) { for (int q = 0; q < nb; ++q) s += U[q * 104 + r * 13 + c]; }
                else if (c == n) { for (int q = 0; q < nb; ++q) s += U[q * 104 + r * 13 + 12]; }
                else {
                    const int j = (c - 4) >> 3, cp = (c - 4) & 7;
                    if (j == a) {
                        for (int q = 0; q < nb; ++q)
#pragma unroll
                            for (int k = 0; k < 8; ++k) s += U[q * 104 + r * 13 + 4 + k] * As[q * kAdMat + cp * kAdRow + k];
                    } else {
                        const int jj = j < a ? j : j - 1, q1 = jj, q2 = W - 1 + jj;          // bins (a,j) and (j,a)
#pragma unroll
                        for (int k = 0; k < 8; ++k) s += U[q1 * 104 + r * 13 + 4 + k] * Ao[q1 * kAdMat + cp * kAdRow + k];
#pragma unroll
                        for (int k = 0; k < 8; ++k) s += U[q2 * 104 + r * 13 + 4 + k] * Ao[q2 * kAdMat + cp * kAdRow + k];
                    }
                }
                put(4 + 8 * a + r, c, s);
                if (c < 4 || c == n) put(c, 4 + 8 * a + r, s);
            }
        } else if (sys == 0) {                                 // 5x5 corner {calib, b}: all bins' entries in one parallel pass, then a fixed-order sum
            for (int e = tid; e < W * W * 25; e += NT) {
                const int bin = e / 25, o = e - bin * 25, ri = o / 5, ci = o - ri * 5;
                lds[e] = (bin % W != bin / W) ? D.M_top[(size_t)bin * 169 + (ri < 4 ? ri : 12) * 13 + (ci < 4 ? ci : 12)] : 0.0;
            }
            __syncthreads();
            for (int e = tid; e < 25; e += NT) {
                const int ri = e / 5, ci = e - ri * 5;
                double s = 0;
                for (int bin = 0; bin < W * W; ++bin) s += lds[bin * 25 + e];
                put(ri < 4 ? ri : n, ci < 4 ? ci : n, s);
            }
        } else if (g < W) {
            if (ad_in_lds) { if (D.W == 8) stitch_sc_rows<true, 8>(D, lds, g, r0, 8 / kScSplit, put); else stitch_sc_rows<true, 0>(D, lds, g, r0, 8 / kScSplit, put); }
            else stitch_sc_rows<false, 0>(D, lds, g, r0, 8 / kScSplit, put);
        } else {
            const int cb = 8 * (W - 1);
            for (int e = tid; e < W * 25; e += NT) {
                const int i = e / 25, o = e - i * 25, ri = o / 5, ci = o - ri * 5;
                lds[e] = D.M_sc[(size_t)i * NPL * NPL + (size_t)(cb + ri) * NPL + cb + ci];
            }
            __syncthreads();
            for (int e = tid; e < 25; e += NT) {
                const int ri = e / 5, ci = e - ri * 5;
                double s = 0;
                for (int i = 0; i < W; ++i) s += lds[i * 25 + e];
                put(ri < 4 ? ri : n, ci < 4 ? ci : n, s);
            }
        }
    }
    if (!mapped) return;
    // the row stores have been acknowledged (release fence = s_waitcnt) before this workgroup's ticket; the flag follows the last ticket
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (tid == 0) is_last = __hip_atomic_fetch_add(D.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u;
    __syncthreads();
    if (!is_last) return;
    const size_t t0 = (size_t)2 * n1 * n1;
    for (int i = tid; i < ntail; i += NT) __hip_atomic_store(&mapped[t0 + i], D.H[t0 + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_store(D.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                 // re-armed for the next (stream-ordered) launch
        __hip_atomic_store(&mapped[t0 + ntail], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
int ba_launch_stitch(hipStream_t s, const StitchDev& D, bool top, bool sc, double* mapped, int ntail, double seq) {
    const int mask = (top ? 1 : 0) | (sc ? 2 : 0);
    const size_t lds_top = (size_t)2 * (D.W - 1) * (169 + 2 * kAdMat + 104) * 8, lds_ad = (size_t)2 * D.W * D.W * kAdMat * 8;
    size_t lds_sc = (size_t)D.W * 8 * D.NPL * 8;
    const size_t lds_p1 = ((size_t)D.NPL * D.NPL + (size_t)D.W * 8 * D.NPL + (size_t)2 * D.W * 64) * 8;
    const size_t lds_stage = lds_p1 > lds_ad ? lds_p1 : lds_ad;
    const int ad_in_lds = lds_sc + lds_stage <= 156 * 1024;                 // FAST path of stitch_sc_rows
    if (ad_in_lds) lds_sc += lds_stage;
    size_t lds = lds_top > lds_sc ? lds_top : lds_sc;
    if (lds < (size_t)D.W * D.W * 25 * 8) lds = (size_t)D.W * D.W * 25 * 8;
    static size_t lds_allowed = 48 * 1024;
    if (lds > lds_allowed) {
        if (hipFuncSetAttribute((const void*)ba_stitch_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 1;
        lds_allowed = lds;
    }
    ba_stitch_kernel<<<(kScSplit + kTopSplit) * D.W + 2, 1024, lds, s>>>(D, mask, ad_in_lds, mapped, ntail, seq);
    return 0;
}

// ------------------------------------------------------------------------------------------------ a12 + step
// xAd: [W*W][8] index h*W + t (EnergyFunctional.cpp:270-280), xc: cstep(4)
// STEP: optimize() applies the step right away (stepfacD; FullSystemOptimize.cpp:271-276) and leaves the block sums {step^2, |idepth_backup|, count}
// for the break test in `partial` - one launch instead of resubstitute, doStep and a sum kernel.
// KARG: windows of up to 8 frames get {xc, xAd} (2 KB) as kernel ARGUMENTS (block-uniform: scalar loads from the kernarg segment) - no H2D copy, and
// its few microseconds of blit + bubble, between the host's solve and this launch.
// XMODE 2 (round 4): windows of more than 8 frames get x itself (8W + 4 floats) as kernel arguments and every workgroup builds the 8W entries of xAd its host
// needs from the device copy of the float adjoints (EnergyFunctional.cpp:268-280: xAd[h][t] = x_h^T adHostF + x_t^T adTargetF, the same mul / add sequence as the
// host loop, uncontracted): the two or three one-block ba_put launches in front of every back-substitution of a 12-frame window are gone.
template <bool STEP, int XMODE>
__global__ __launch_bounds__(256) void ba_resub_kernel(BADev B, XadArg X, GateArg G, float stepfacD, float* __restrict__ partial) {
    __shared__ float smem[64 * 4];
    __shared__ float xrow[XMODE >= 2 ? NALO_MAX_WINDOW * 8 + 4 : 1];
    constexpr bool KARG = XMODE == 1;
    static_assert(XMODE >= 1 && XMODE <= 3, "x or xAd arrive as kernel arguments, or (3) through the gate block");
    const float* xc = XMODE == 3 ? xrow + NALO_MAX_WINDOW * 8 : X.v;
    const float* xAd = X.v + 4;                                // XMODE 1 only
    if constexpr (XMODE == 3) {
        // enqueued while the host still solves the system (ba_device.h: GateBlock): wait for {xc, xAd} in host-mapped memory, then this host's row of xAd -> LDS
        __shared__ int gate_ok;
        const int tid = threadIdx.x, W = B.W;
        if (tid == 0) gate_ok = gate_wait(G.flag, G.want, G.err) ? 1 : 0;
        __syncthreads();
        if (!gate_ok) return;
        const int hb = B.blk_host[blockIdx.x];
        // ordinary coalesced loads (first touch of these lines in this kernel, after the gate: what the host wrote); cache-bypassing loads are one host request per lane
        if (tid < W * 8) xrow[tid] = G.x[4 + (size_t)hb * W * 8 + tid];
        else if (tid >= 252) xrow[NALO_MAX_WINDOW * 8 + tid - 252] = G.x[tid - 252];
        __syncthreads();
    }
    if constexpr (XMODE == 2) {
        const int hb = B.blk_host[blockIdx.x], W = B.W, tid = threadIdx.x;
        if (tid < W * 8) {
            const int t = tid >> 3, j = tid & 7;
            const float* AH = B.adF + (size_t)(hb + W * t) * 64 + j;
            const float* AT = B.adF + (size_t)W * W * 64 + (size_t)(hb + W * t) * 64 + j;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) { s1 = __fadd_rn(s1, __fmul_rn(X.v[4 + 8 * hb + i], AH[i * 8])); s2 = __fadd_rn(s2, __fmul_rn(X.v[4 + 8 * t + i], AT[i * 8])); }
            xrow[tid] = __fadd_rn(s1, s2);
        }
        __syncthreads();
    }
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    float v[3] = {0.f, 0.f, 0.f};
    if (d < B.Ppad && (B.pt_flags[d] & PT_VALID)) {
        const int h = __builtin_amdgcn_readfirstlane(B.blk_host[blockIdx.x]), W = B.W;          // blockDim = kBlk: one host per block, so xAd rows are scalar loads
        float stp = 0.f;
        if (B.pt_ngood[d] != 0) {
            const float4 pa = B.pt_acc[d], hc = B.pt_hcd[d];
            float bsum = pa.w;
            bsum -= xc[0] * hc.x + xc[1] * hc.y + xc[2] * hc.z + xc[3] * hc.w;
            if constexpr (KARG || XMODE == 3) {                 // small window: 8 targets' slots in flight at once, subtracted in target order
                for (int t0 = 0; t0 < W; t0 += 8) {
                    uint8_t rs[8]; float4 j0[8], j1[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        rs[i] = 0; j0[i] = make_float4(0.f, 0.f, 0.f, 0.f); j1[i] = j0[i];
                        if (t0 + i < W) {                       // wave-uniform: a scalar branch around the loads, no wait
                            const size_t si = (size_t)(t0 + i) * B.Ppad + d;
                            rs[i] = B.rs_state[si]; j0[i] = B.rs_jp0[si]; j1[i] = B.rs_jp1[si];
                        }
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int t = t0 + i;
                        const float* xa = XMODE == 3 ? xrow + min(t, W - 1) * 8 : xAd + (size_t)(h * W + min(t, W - 1)) * 8;           // uniform address, loaded whatever the lane's residual state
                        const float term = xa[0] * j0[i].x + xa[1] * j0[i].y + xa[2] * j0[i].z + xa[3] * j0[i].w + xa[4] * j1[i].x + xa[5] * j1[i].y + xa[6] * j1[i].z + xa[7] * j1[i].w;
                        if (t < W && t != h && (rs[i] & RS_ACTIVE)) bsum -= term;
                    }
                }
            } else {
                // large window: every state byte first, then the JpJdF pairs of the ACTIVE slots with all loads in flight (an inactive slot re-reads the
                // point's first slot instead of branching: a load under a lane-divergent branch is waited for at the join, which made this loop a chain of
                // 2 (W - 1) dependent round trips: 16.7 us for 62 MB at 1.75 M residuals)
                for (int t0 = 0; t0 < W; t0 += 8) {
                    uint8_t rs[8]; float4 j0[8], j1[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) { const int t = t0 + i; rs[i] = (t < W && t != h) ? B.rs_state[(size_t)t * B.Ppad + d] : 0; }
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int t = t0 + i;
                        const bool on = t < W && (rs[i] & RS_ACTIVE);
                        const size_t si = (size_t)(on ? t : 0) * B.Ppad + d;
                        j0[i] = B.rs_jp0[si]; j1[i] = B.rs_jp1[si];
                        if (!on) rs[i] = 0;
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int t = t0 + i;
                        const float* xa = XMODE >= 2 ? xrow + min(t, W - 1) * 8 : xAd + (size_t)(h * W + min(t, W - 1)) * 8;
                        const float term = xa[0] * j0[i].x + xa[1] * j0[i].y + xa[2] * j0[i].z + xa[3] * j0[i].w + xa[4] * j1[i].x + xa[5] * j1[i].y + xa[6] * j1[i].z + xa[7] * j1[i].w;
                        if (rs[i] & RS_ACTIVE) bsum -= term;
                    }
                }
            }
            stp = -bsum * pa.z;
        }
        B.pt_step[d] = stp;
        if (STEP) {
            float4 geo = B.pt_geo[d];
            const float idb = geo.z;
            B.pt_backup[d] = idb;
            geo.z = idb + stepfacD * stp; geo.w = geo.z;
            B.pt_geo[d] = geo;
            v[0] = stp * stp; v[1] = fabsf(idb); v[2] = 1.f;
        }
    }
    if (STEP) block_reduce_cols<3, 256>(v, smem, partial + (size_t)blockIdx.x * 4);
}
// idepth = idepth_backup + stepfacD*step; idepth_zero = idepth (FullSystemOptimize.cpp:271-276). sums: {step^2, |idepth_backup|, count}
__global__ __launch_bounds__(256) void ba_step_kernel(BADev B, float stepfacD, float* __restrict__ partial /* [blocks][4] */) {
    __shared__ float smem[64 * 4];
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    float v[3] = {0.f, 0.f, 0.f};
    if (d < B.Ppad && (B.pt_flags[d] & PT_VALID)) {
        float4 geo = B.pt_geo[d];
        const float idb = geo.z, stp = B.pt_step[d];
        B.pt_backup[d] = idb;
        geo.z = idb + stepfacD * stp; geo.w = geo.z;
        B.pt_geo[d] = geo;
        v[0] = stp * stp; v[1] = fabsf(idb); v[2] = 1.f;
    }
    block_reduce_cols<3, 256>(v, smem, partial + (size_t)blockIdx.x * 4);
}
// out[j] = sum_b partial[b*stride + j] in fp64: 16 groups of 64 lanes stride over the blocks, then a fixed-order LDS combine
__global__ __launch_bounds__(1024) void ba_sum_partials_kernel(const float* __restrict__ partial, int nblocks, int stride, int nvals, double* __restrict__ out) {
    __shared__ double part[1024];
    const int L = nvals <= 4 ? 4 : 64, NG = 1024 / L;          // few values: 256 lane groups walk the blocks (a 1M-point window has 3907 of them)
    const int j = threadIdx.x % L, g = threadIdx.x / L;
    double s = 0;
    if (j < nvals) for (int b = g; b < nblocks; b += NG) s += (double)partial[(size_t)b * stride + j];
    part[g * L + j] = s;
    __syncthreads();
    if (g == 0 && j < nvals) { double t = 0; for (int k = 0; k < NG; ++k) t += part[k * L + j]; out[j] = t; }
}
// copies the stitched systems into host-mapped pinned memory and publishes a sequence number the host polls on (the path after a cross-rank
// all-reduce; a single GPU publishes from the stitch kernel). Several workgroups copy slices; the last one to take a ticket sets the flag.
__global__ __launch_bounds__(256) void ba_publish_kernel(const double* __restrict__ src, double* __restrict__ dst, int n, double seq, unsigned* __restrict__ ticket) {
    __shared__ int is_last;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) __hip_atomic_store(&dst[i], src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0) is_last = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u;
    __syncthreads();
    if (is_last && threadIdx.x == 0) {
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&dst[n], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// sharded window, after the all-reduce: tail2 = {sum of the per-rank thresholds, rank count} -> every rank installs the mean
__global__ void ba_th_install_kernel(const double* __restrict__ tail2, float* __restrict__ th) { if (tail2[1] > 1.5) th[0] = (float)(tail2[0] / tail2[1]); }
void ba_launch_th_install(hipStream_t s, const double* tail2, float* th) { ba_th_install_kernel<<<1, 1, 0, s>>>(tail2, th); }
__global__ void ba_th_tail_kernel(const float* __restrict__ th, double* __restrict__ tail2) { tail2[0] = (double)th[0]; tail2[1] = 1.0; }
void ba_launch_th_tail(hipStream_t s, const float* th, double* tail2) { ba_th_tail_kernel<<<1, 1, 0, s>>>(th, tail2); }
void ba_launch_publish(hipStream_t s, const double* src, double* dst_mapped, int n, double seq, unsigned* ticket) {
    const int nb = n > 32768 ? 32 : (n + 1023) / 1024;
    ba_publish_kernel<<<nb < 1 ? 1 : nb, 256, 0, s>>>(src, dst_mapped, n, seq, ticket);
}
// mapped host memory -> device memory by ONE workgroup: the precalc records of a large window (10-23 KB). A copy packet of that size costs 4 us plus ~6 us of
// pipeline bubble before the next kernel of the stream; the workgroup pulls the block over PCIe in one round of 16-byte loads
__global__ __launch_bounds__(1024) void ba_pull_kernel(float4* __restrict__ dst, const float4* __restrict__ src_mapped, int n4) {
    for (int i = threadIdx.x; i < n4; i += 1024) dst[i] = src_mapped[i];
}
void ba_launch_pull(hipStream_t s, float* dst, const float* src_mapped, int n) { ba_pull_kernel<<<1, 1024, 0, s>>>((float4*)dst, (const float4*)src_mapped, (n + 3) / 4); }
// karg: {xc, xAd} of a window of <= 8 frames, or (karg_is_x) the solution x itself for larger ones
void ba_launch_resub(hipStream_t s, const BADev& B, const XadArg& karg, bool karg_is_x) {
    const GateArg none{};
    if (karg_is_x) ba_resub_kernel<false, 2><<<(B.Ppad + 255) / 256, 256, 0, s>>>(B, karg, none, 0.f, nullptr);
    else ba_resub_kernel<false, 1><<<(B.Ppad + 255) / 256, 256, 0, s>>>(B, karg, none, 0.f, nullptr);
}
void ba_launch_resub_step(hipStream_t s, const BADev& B, float stepfacD, float* partial, const XadArg& karg, bool karg_is_x) {
    const GateArg none{};
    if (karg_is_x) ba_resub_kernel<true, 2><<<(B.Ppad + 255) / 256, 256, 0, s>>>(B, karg, none, stepfacD, partial);
    else ba_resub_kernel<true, 1><<<(B.Ppad + 255) / 256, 256, 0, s>>>(B, karg, none, stepfacD, partial);
}
// the same back-substitution + step, enqueued AHEAD of the solve: {xc, xAd} arrive through the gate block (small windows, <= 8 frames)
void ba_launch_resub_step_gated(hipStream_t s, const BADev& B, float stepfacD, float* partial, const GateArg& gate) {
    static const XadArg none{};
    ba_resub_kernel<true, 3><<<(B.Ppad + 255) / 256, 256, 0, s>>>(B, none, gate, stepfacD, partial);
}
// the three step sums of a back-substitution that stepped the points itself (ba_resub_kernel<true, ..>), for a caller that wants the break test NOW
void ba_launch_step_sums(hipStream_t s, const BADev& B, const float* partial, double* out3) { ba_sum_partials_kernel<<<1, 1024, 0, s>>>(partial, (B.Ppad + 255) / 256, 4, 3, out3); }
void ba_launch_step(hipStream_t s, const BADev& B, float stepfacD, float* partial, double* out3) {
    const int nb = (B.Ppad + 255) / 256;
    ba_step_kernel<<<nb, 256, 0, s>>>(B, stepfacD, partial);
    ba_sum_partials_kernel<<<1, 1024, 0, s>>>(partial, nb, 4, 3, out3);
}

// ------------------------------------------------------------------------------------------------ setNewFrameEnergyTH
// Exact n-th element (nthIdx = (int)(0.7f * n), FullSystemOptimize.cpp:117-122) of the non-negative energies of the residuals that target the newest frame
// (en_new[], written by ba_linearize_kernel; -1 = no residual), by a THREE-level radix select on the float bit patterns (monotone for x >= 0):
//   level A = bits 30..20 (2048 bins), level B = bits 19..9 (2048 bins), level C = bits 8..0 (512 bins).
// Every level is one pass over en_new[] (4 bytes per point slot) into a histogram that IS the cross-rank payload: two bins per double (a + b * 2^26: exact while
// a bin's global count stays below 2^26 = 67 M residuals towards the newest frame and the sum below 2^52), built with fp64 atomics after an LDS pre-aggregation
// - 1024 + 1024 + 256 doubles. A sharded window sums each of them over the ranks before the next level's search (host_ba.hip), so every rank holds the order
// statistic of the WHOLE window, bit for bit what one GPU computes; level C is written behind the stitched systems and rides in their all-reduce.
// (Rounds 1-3: 16 + 16 bits, two 65536-bin histograms = 16384 + 32768 doubles per pass = two thirds of a sharded iteration's 560 KB exchange, four conversion
// kernels, a 16-party arrival counter per search and the hi histogram's atomics inside ba_linearize_kernel. Now 18 KB, no conversion, no inter-workgroup wait:
// every workgroup of a fill kernel repeats the (tiny) search of the previous level itself.)
// state[0] = count, [1] = k below level A's bin, [2] = bin A, [3] = empty flag, [4] = k below level B's bin, [5] = bin B.
constexpr int kThBinsAB = 2048, kThBinsC = 512;
constexpr double kThPack = 67108864.0;                        // 2^26
// the bin of a packed histogram whose running count passes k (all 256 lanes call; s = 8 shared words). first: k = (int)(0.7f * total) of THIS histogram.
template <int NBINS>
__device__ __forceinline__ void th_search(const double* __restrict__ buf, bool first, unsigned k_in, unsigned* s, unsigned& total, unsigned& k, unsigned& bin, unsigned& before) {
    constexpr int NB = NBINS / 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned cnt[NB], sum = 0;
#pragma unroll
    for (int j = 0; j < NB / 2; ++j) {
        const unsigned long long v = (unsigned long long)(buf[tid * (NB / 2) + j] + 0.5);
        cnt[2 * j] = (unsigned)(v & 0x3FFFFFFull); cnt[2 * j + 1] = (unsigned)(v >> 26);
        sum += cnt[2 * j] + cnt[2 * j + 1];
    }
    unsigned incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(incl, o); if (lane >= o) incl += t; }
    if (lane == 63) s[wave] = incl;
    if (tid == 0) { s[4] = (unsigned)(NBINS - 1); s[5] = 0u; }                   // k beyond the last entry (empty histogram): last bin
    __syncthreads();
    unsigned wpre = 0;
    total = s[0] + s[1] + s[2] + s[3];
    for (int i = 0; i < wave; ++i) wpre += s[i];
    k = first ? (unsigned)(int)(kFrameEnergyTHN * (float)total) : k_in;
    const unsigned excl = wpre + incl - sum;
    if (excl <= k && k < excl + sum) {                                          // exactly one lane
        unsigned run = excl; int b = 0; bool found = false;
#pragma unroll
        for (int j = 0; j < NB; ++j) { if (!found && k < run + cnt[j]) { b = j; found = true; } if (!found) run += cnt[j]; }
        s[4] = (unsigned)(tid * NB + b); s[5] = run;
    }
    __syncthreads();
    bin = s[4]; before = s[5];
}
template <int LEVEL>
__global__ __launch_bounds__(256) void ba_th_fill_kernel(const float* __restrict__ en, int n, const double* __restrict__ prev, unsigned* __restrict__ state, double* __restrict__ out) {
    constexpr int NB = LEVEL == 2 ? kThBinsC : kThBinsAB;
    __shared__ unsigned hist[NB];
    __shared__ unsigned s[8];
    const int tid = threadIdx.x;
    unsigned prefix = 0;
    if (LEVEL == 1) {                                                            // the search of level A on its (summed) histogram; workgroup 0 keeps the result
        unsigned total, k, bin, before;
        th_search<kThBinsAB>(prev, true, 0u, s, total, k, bin, before);
        if (blockIdx.x == 0 && tid == 0) { state[0] = total; state[3] = total == 0 ? 1u : 0u; state[1] = k - before; state[2] = bin; }
        prefix = bin;
    }
    if (LEVEL == 2) {
        unsigned total, k, bin, before;
        const unsigned binA = state[2];                                         // written by the previous launch of this stream
        th_search<kThBinsAB>(prev, false, state[1], s, total, k, bin, before);
        if (blockIdx.x == 0 && tid == 0) { state[4] = k - before; state[5] = bin; }
        prefix = (binA << 11) | bin;
    }
    for (int b = tid; b < NB; b += 256) hist[b] = 0u;
    __syncthreads();
    for (int i = blockIdx.x * 256 + tid; i < n; i += gridDim.x * 256) {
        const float f = en[i];
        if (!(f >= 0.f)) continue;
        const unsigned u = __float_as_uint(f);
        if (LEVEL == 0) atomicAdd(&hist[u >> 20], 1u);
        if (LEVEL == 1) { if ((u >> 20) == prefix) atomicAdd(&hist[(u >> 9) & 2047u], 1u); }
        if (LEVEL == 2) { if ((u >> 9) == prefix) atomicAdd(&hist[u & 511u], 1u); }
    }
    __syncthreads();
    for (int b = tid; b < NB / 2; b += 256) {
        const unsigned c0 = hist[2 * b], c1 = hist[2 * b + 1];
        if (c0 | c1) unsafeAtomicAdd(&out[b], (double)c0 + (double)c1 * kThPack);          // integers in doubles: exact, order independent
    }
}
// level C's search + the threshold formula (:130-133); leaves all three histograms zeroed for the next pass (every reader of A and B has finished: stream order)
__global__ __launch_bounds__(256) void ba_th_final_kernel(double* __restrict__ bufC, const unsigned* __restrict__ state, double* __restrict__ bufAB, float* __restrict__ frameTH_new) {
    __shared__ unsigned s[8];
    const int tid = threadIdx.x;
    unsigned total, k, bin, before;
    th_search<kThBinsC>(bufC, false, state[4], s, total, k, bin, before);
    if (tid == 0) {
        float th;
        if (state[3]) th = 12.f * 12.f * (float)kPatternNum;                     // no residual on the newest frame (:110-114)
        else {
            const float nthElement = sqrtf(__uint_as_float((state[2] << 20) | (state[5] << 9) | bin));
            th = nthElement * kFrameEnergyTHFacMedian;                           // FullSystemOptimize.cpp:130-133
            th = 26.0f * kFrameEnergyTHConstWeight + th * (1.f - kFrameEnergyTHConstWeight);
            th = th * th; th *= kOverallEnergyTHWeight * kOverallEnergyTHWeight;
        }
        *frameTH_new = th;
    }
    for (int i = tid; i < kThBinsAB; i += 256) bufAB[i] = 0.0;                   // A | B: 1024 doubles each
    bufC[tid] = 0.0;                                                            // 256 doubles (th_search has read them: two barriers ago)
}
static int th_grid(int n) { return std::min(64, std::max(1, (n + 4095) / 4096)); }
// step 0 / 1 / 2: fill level A / B / C (B and C search the previous level first), 3: level C's search + the threshold. A sharded window sums A, B, C over the
// ranks between the steps (host_ba.hip); a single GPU runs the four launches back to back behind the publish, under the host's solve.
void ba_launch_energy_th_step(hipStream_t s, const BADev& B, int step) {
    double *A = B.th_bufAB, *Bb = B.th_bufAB + kThDblAB, *Cc = B.th_bufC;
    if (step == 0) ba_th_fill_kernel<0><<<th_grid(B.Ppad), 256, 0, s>>>(B.en_new, B.Ppad, nullptr, B.th_state, A);
    else if (step == 1) ba_th_fill_kernel<1><<<th_grid(B.Ppad), 256, 0, s>>>(B.en_new, B.Ppad, A, B.th_state, Bb);
    else if (step == 2) ba_th_fill_kernel<2><<<th_grid(B.Ppad), 256, 0, s>>>(B.en_new, B.Ppad, Bb, B.th_state, Cc);
    else ba_th_final_kernel<<<1, 256, 0, s>>>(Cc, B.th_state, A, B.frameTH + (B.W - 1));
}
void ba_launch_energy_th(hipStream_t s, const BADev& B) {
    if (B.Ppad <= 16384) { ba_th_small_kernel<<<1, 1024, 0, s>>>(B.en_new, B.Ppad, B.frameTH + (B.W - 1)); return; }
    for (int step = 0; step < 4; ++step) ba_launch_energy_th_step(s, B, step);
}

// EnergyFunctional::calcLEnergyPt (EnergyFunctional.cpp:332-392), the per-point part that can be non-zero when FullSystem::optimize calls it: deltaF^2 priorF
// (:388). Its inner loop runs over LINEARISED residuals, which exist only between flagPointsForRemoval and marginalizePointsF (FullSystem.cpp:975-990,1453),
// never while optimize() runs. One fp64 partial per block, summed by the host in block order.
__global__ __launch_bounds__(256) void ba_lenergy_kernel(BADev B, double* __restrict__ partial) {
    __shared__ double sd[4];
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    double e = 0.0;
    if (d < B.Ppad && (B.pt_flags[d] & PT_VALID)) {
        const float4 geo = B.pt_geo[d];
        const float deltaF = geo.z - geo.w;
        e = (double)(deltaF * deltaF * B.pt_prior[d]);
    }
    for (int o = 32; o > 0; o >>= 1) e += __shfl_down(e, o);
    if ((threadIdx.x & 63) == 0) sd[threadIdx.x >> 6] = e;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (sd[0] + sd[1]) + (sd[2] + sd[3]);
}
void ba_launch_lenergy(hipStream_t s, const BADev& B, double* partial) { ba_lenergy_kernel<<<(B.Ppad + 255) / 256, 256, 0, s>>>(B, partial); }

// the point part of FullSystem::loadSateBackup (FullSystemOptimize.cpp:352-369): idepth = idepth_backup AND idepth_zero = idepth_backup
__global__ __launch_bounds__(256) void ba_load_backup_kernel(BADev B) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d < B.Ppad && (B.pt_flags[d] & PT_VALID)) { float4 geo = B.pt_geo[d]; geo.z = geo.w = B.pt_backup[d]; B.pt_geo[d] = geo; }
}
// FullSystem::SWGrayOptimize_J (reference src/FullSystem/PlaneOptimize.cpp:307-454) without Ceres. Its cost functor GrayTHFactor_TH (PlaneOptimize.h:348-457)
// evaluates ONE residual per (point, target != host): the centre pixel (pattern index 4) of the point projected with T_ij = T_j T_i^-1 (poses rebuilt from
// {translation, so3 log} parameter blocks, fp64; the projection in the mixed float/double arithmetic of projectPoint, PlaneOptimize.h:277-297), r = I_target(Ku,Kv)
// - color[4], or r = 100 when the projection leaves the image. Its analytic Jacobians are multiplied by an image gradient that is ALWAYS ZERO: the functor
// reads the gradient from an outer `hitColor` that an inner declaration shadows (PlaneOptimize.h:378-381, 400-401). Ceres therefore sees a
// zero gradient at the initial point and returns it unchanged (gradient tolerance reached in iteration 0). What remains of the call is (a) the value of
// the Huber(100) cost, computed here, and (b) the state changes made AFTER the solve with the unchanged parameters (host_ba.hip: nalo_ba_sw_gray_optimize).
// Block = 256 points of one host x one target; partial {cost sum, residual blocks} per (block, target) in fp64.
__global__ __launch_bounds__(256) void ba_swgray_kernel(BADev B, const double* __restrict__ Rt /* [W*W][12] row h*W+t: R (9), t (3) of T_ij */, double* __restrict__ partial) {
    __shared__ double sd[4][2];
    const int b = blockIdx.x, t = blockIdx.y, tid = threadIdx.x, d = b * kBlk + tid, h = B.blk_host[b];
    double cost = 0.0, cnt = 0.0;
    if (t != h && (B.pt_flags[d] & PT_VALID)) {
        const float4 geo = B.pt_geo[d];
        const double inv_dep = (double)(kScaleIdepth * geo.z);                                        // idepth_scaled
        if (!(inv_dep < 1e-4 || inv_dep > 1e3)) {                                                    // PlaneOptimize.cpp:375-376
            const double* M = Rt + (size_t)(h * B.W + t) * 12;
            const float fxl = B.calib[0], fyl = B.calib[1], cxl = B.calib[2], cyl = B.calib[3], fxli = B.calib[4], fyli = B.calib[5];
            const float idf = (float)inv_dep;
            const double K0 = (double)((geo.x + 0 - cxl) * fxli), K1 = (double)((geo.y + 0 - cyl) * fyli);
            const double p0 = M[0] * K0 + M[1] * K1 + M[2] * 1.0 + M[9] * (double)idf, p1 = M[3] * K0 + M[4] * K1 + M[5] * 1.0 + M[10] * (double)idf,
                         p2 = M[6] * K0 + M[7] * K1 + M[8] * 1.0 + M[11] * (double)idf;
            const float drescale = (float)(1.0f / p2);
            const float u = (float)(p0 * (double)drescale), v = (float)(p1 * (double)drescale);
            const float Ku = u * fxl + cxl, Kv = v * fyl + cyl;
            double r;
            if (!(Ku > 1.1f && Kv > 1.1f && Ku < (float)(B.w - 3) && Kv < (float)(B.h - 3)) || inv_dep < 0) r = 100.0;
            else {
                const float4* img = B.img[t];
                const int ix = (int)Ku, iy = (int)Kv;
                const float dx = Ku - ix, dy = Kv - iy, dxdy = dx * dy;
                const float4* bp = img + ix + iy * B.w;
                const float I = dxdy * bp[1 + B.w].x + (dy - dxdy) * bp[B.w].x + (dx - dxdy) * bp[1].x + (1 - dx - dy + dxdy) * bp[0].x;
                r = (double)(I - B.pt_col1[d].x);                                                    // color[4]
            }
            const double s = r * r;
            cost = 0.5 * (s <= 1e4 ? s : 2.0 * 100.0 * sqrt(s) - 1e4);                               // ceres::HuberLoss(100): rho(s) = s, or 2 a sqrt(s) - a^2
            cnt = 1.0;
        }
    }
    for (int o = 32; o > 0; o >>= 1) { cost += __shfl_down(cost, o); cnt += __shfl_down(cnt, o); }
    if ((tid & 63) == 0) { sd[tid >> 6][0] = cost; sd[tid >> 6][1] = cnt; }
    __syncthreads();
    if (tid == 0) { double* o = partial + ((size_t)b * B.W + t) * 2; o[0] = (sd[0][0] + sd[1][0]) + (sd[2][0] + sd[3][0]); o[1] = (sd[0][1] + sd[1][1]) + (sd[2][1] + sd[3][1]); }
}
void ba_launch_swgray(hipStream_t s, const BADev& B, const double* Rt, double* partial) { ba_swgray_kernel<<<dim3(B.nblocks, B.W), 256, 0, s>>>(B, Rt, partial); }

// idepth (and idepth_zero) of the points hosted by frames with index < first_kept_host are (re)set: mode 0 idepth_zero = idepth (SWGrayOptimize_J's
// setIdepth / setIdepthZero with the unchanged parameter, PlaneOptimize.cpp:413-419: hosts 0 .. W-3); mode 1: idepth = idepth_zero = (float)(idepth / scale) for
// the points of ONE host (planeOptimize's scale fix, :268-273)
__global__ __launch_bounds__(256) void ba_set_idepth_kernel(BADev B, int mode, int host_sel, double scale) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= B.Ppad || !(B.pt_flags[d] & PT_VALID)) return;
    const int h = B.blk_host[d / kBlk];
    float4 geo = B.pt_geo[d];
    if (mode == 0) { if (h < host_sel) { geo.z = (float)(double)(kScaleIdepth * geo.z); geo.w = geo.z; B.pt_geo[d] = geo; } }
    else if (h == host_sel) { geo.z = (float)((double)(kScaleIdepth * geo.z) / scale); geo.w = geo.z; B.pt_geo[d] = geo; }
}
void ba_launch_set_idepth(hipStream_t s, const BADev& B, int mode, int host_sel, double scale) { ba_set_idepth_kernel<<<(B.Ppad + 255) / 256, 256, 0, s>>>(B, mode, host_sel, scale); }

void ba_launch_load_backup(hipStream_t s, const BADev& B) { ba_load_backup_kernel<<<(B.Ppad + 255) / 256, 256, 0, s>>>(B); }

}  // namespace nalo
