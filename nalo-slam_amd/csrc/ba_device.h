// Device-side layout of the BA window (shared by kernels_ba.hip and host_ba.hip).
//
// Points are sorted by host frame; every host segment is padded to a multiple of kBlk so that one 256-thread
// block only holds points of one host (blk_host[b]). One residual slot per (target t, point d): arrays are
// [W][Ppad], t-major, so a wave reads consecutive points of one target: coalesced.
#pragma once
#include <cstddef>
#include <hip/hip_runtime.h>
#include <cstdint>

namespace nalo {

constexpr int kBlk = 256;            // points per linearize block
constexpr int kTopVals = 93;         // 91 AccumulatorApprox entries + residual count + energy
constexpr int kTopStride = 96;       // floats per (block, target) partial
constexpr int kPreStride = 40;       // floats per (host,target) precalc record

// residual slot state byte
enum : uint8_t { RS_STATE_MASK = 3, RS_EXISTS = 4, RS_ACTIVE = 8, RS_LINEARIZED = 16 };
// point flags
enum : uint8_t { PT_VALID = 1, PT_MARG = 2, PT_HAS_PRIOR = 4 };

// precalc record (FrameFramePrecalc, reference src/FullSystem/HessianBlocks.h:80-107 + adHTdeltaF):
//  [0..8] PRE_KRKiTll  [9..11] PRE_KtTll  [12..20] PRE_RTll_0  [21..23] PRE_tTll_0  [24,25] PRE_aff_mode  [26] PRE_b0_mode
//  [27..34] adHTdeltaF[h + t*W]  (EnergyFunctional.cpp:175-181)
struct BADev {
    int W, P, Ppad, nblocks, w, h;
    const float* calib;                         // [10] {fxl, fyl, cxl, cyl, fxli, fyli (CalibHessian::value_scaledf / value_scaledi), cDeltaF[4] (EnergyFunctional)}
    const float4* img[16];                      // level-0 {I,dx,dy,0} of every window frame (row major)
    const float* img_t[16];                     // the same texels, 12 bytes each, in 5x2 tiles of 128 bytes (frame_tile_level0): what ba_linearize gathers from
    int wt;                                     // tiles per tile row = ceil(w / 5)
    int lin_sub;                                // ba_linearize workgroups (= fp64 partials) per point block: 1 = 256 threads, 4 = one wave each (small windows)
    const float* pre;                           // [W*W][kPreStride], index h*W + t
    float* frameTH;                             // [W] frameEnergyTH (device resident, updated by the quantile kernel)
    const int* blk_host;                        // [nblocks]
    const int* blk_order;                       // [8][xcd_len]: point blocks an XCD walks (spatial eighth of every host), -1 = none
    int xcd_len;
    // points
    float4* pt_geo;                             // {u, v, idepth, idepth_zero}
    const float4 *pt_col0, *pt_col1, *pt_w0, *pt_w1;
    float* pt_prior;                            // EFPoint::priorF
    uint8_t* pt_flags;
    float4* pt_acc;                             // {Hdd_accAF, bd_accAF, HdiF, bdSumF}
    float4* pt_hcd;                             // Hcd_accAF
    uint8_t* pt_ngood;
    float* pt_step;
    float* pt_backup;
    const unsigned* gate_p; unsigned* gate_err; unsigned gate_p_want;   // ba_linearize_kernel: NULL, or the gate of its precalc records (GateBlock::p_seq)
    const float* adF;                           // float adjoints [adHostF (W*W*64) | adTargetF (W*W*64)], index (h + t*W)*64 + i*8 + j (ba_resub_kernel, XMODE 2)
    float* pt_relbs;                            // max relBS over this pass' active residuals (fix mode): the pass' atomicMax target, all zero when the pass starts
    float* pt_relbs_next;                       // the buffer of the NEXT fix pass: zeroed by this pass' idle (target == host) workgroups - no fill launch on the path
    // residual slots [W][Ppad]
    uint8_t* rs_state;
    float2* rs_energy;                          // {state_energy, state_NewEnergy}
    float4 *rs_jp0, *rs_jp1;                    // EFResidual::JpJdF
    float4* rs_pp0; float2* rs_pp1;             // per-slot share of the point sums: {bd, Hdd, Hcd0, Hcd1}, {Hcd2, Hcd3}
    float4* rs_cpt;                             // {Ku, Kv, new_idepth, relBS} (centerProjectedTo), written when fix/marg
    float* en_new;                              // [Ppad] state_NewEnergyWithOutlier of residuals targeting frame W-1 (-1 = none)
    double *th_bufAB, *th_bufC;                 // radix-select histograms of setNewFrameEnergyTH, two bins per double: A | B (1024 doubles each), C (256, behind the stitched systems)
    unsigned* th_state;                         // {count, k below bin A, bin A, empty, k below bin B, bin B}
    // partials
    double* top_partial;                        // [nblocks*lin_sub][W][kTopStride]: one partial per ba_linearize workgroup and target (fp64: one rounding less before the cancelling H_A - H_sc)
    double* sc_partial;                         // [sc_groups * sc_split][T(T+1)/2 upper tiles][256] (MFMA register order)
    int sc_split;                               // 1 or 4 workgroups per point block in ba_sc_kernel (4 for small windows)
    const int* host_blk;                        // [W+1] point-block range of every host
    const int* sc_grp;                          // [W+1] ba_sc workgroup-group range of every host (groups of sc_bpw blocks)
    int sc_bpw, sc_groups;                      // blocks per group (1 when sc_split > 1), total groups
    // settings (nalo_set_settings): setting_affineOptModeA / B < 0 zero JabF[0] / JabF[1] (Residuals.cpp:241-242)
    int fix_a, fix_b;
    int reset_oob;                              // 1: the pass starts from resetOOB'ed residuals (the first linearisation of optimize(), FullSystemOptimize.cpp:412-429): state = IN, energies 0
    int no_th;                                  // 1: this pass does not feed setNewFrameEnergyTH (the re-run that applies an accepted step, setting_forceAceptStep = false)
    double* noapply_E;                          // [nblocks*lin_sub][W] energy partials of a linearisation that is NOT applied (FIX = 2)
};

// {xc (4) | xAd [W*W][8]} of resubstituteFPt for windows of up to 8 frames, passed by value as kernel arguments (ba_resub_kernel)
struct XadArg { float v[4 + 8 * 64]; };

// Gate of a kernel that is enqueued BEFORE its inputs exist (round 4, small single-GPU windows): the host still solves the system while the back-substitution and
// the next linearisation already sit in the stream; each spins (one lane per workgroup, bounded) on a word of host-mapped memory until the host has written the
// inputs - the step {xc, xAd}, the precalc records - and then the sequence number. What it saves per Gauss-Newton iteration is the launch latency of the two
// kernels on the critical path (the device starts ~1.5 us after the host's store instead of 5-6 us after its launch call). 0xFFFFFFFF = cancelled: the kernel returns
// without touching anything (the host's error paths), as it does when the bound expires (and then raises err).
// (the payload starts on its own 128-byte line: a gated kernel's first look at a flag is a cached scalar load made BEFORE the host writes the payload, and a device
// cache line that held both would hand the payload's loads the bytes of the iteration before)
struct GateBlock { unsigned x_seq, p_seq, err, pad[29]; float x[4 + NALO_MAX_WINDOW * NALO_MAX_WINDOW * 8]; };
static_assert(offsetof(GateBlock, x) == 128, "flags and payload of the gate block on separate cache lines");
constexpr unsigned kGateCancel = 0xFFFFFFFFu;
struct GateArg { const unsigned* flag; unsigned* err; const float* x; unsigned want; };
__device__ __forceinline__ bool gate_wait(const unsigned* flag, unsigned want, unsigned* err) {      // one lane; true = the inputs are there
    // RELAXED system-scope loads (always served by the host's memory) + a compiler barrier: an ACQUIRE at system scope invalidates the L2 on every poll, and the
    // kernel behind the gate then finds its points, residuals and texels gone (measured: 46 -> 96 us per iteration). Nothing the gate orders lives in a device
    // cache: the inputs are in host-mapped memory, read after the loop in program order (the hardware does not issue loads past an unresolved branch).
    // First a SCALAR load - the path the precalc records themselves take (scalar cache + L2, shared by the workgroups): a gate that is already open (the
    // linearisation's usually is: the host wrote its records while the back-substitution ran) then costs what one more record word costs. 256 workgroups asking
    // the host for one word with cache-bypassing loads are served one after the other, ~75 ns each (measured: 15 -> 34 us per gated linearisation; a volatile load is
    // the same sc0 sc1 access). The value is unique per launch: a cached copy can only be an older, closed one, and falls through to the polling loop.
    {
        unsigned v0;
        asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v0) : "s"(flag) : "memory");
        if (v0 == want) { __atomic_signal_fence(__ATOMIC_ACQUIRE); return true; }
    }
    for (unsigned spins = 0; spins < (1u << 21); ++spins) {
        const unsigned v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (v == want) { __atomic_signal_fence(__ATOMIC_ACQUIRE); return true; }
        if (v == kGateCancel) return false;
        __builtin_amdgcn_s_sleep(2);
    }
    if (err) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return false;
}

// What may ride along with ba_reduce_kernel: the newest frame's energy threshold of a small window (one more workgroup: ba_th_small's body) and, on a misc-only
// fetch, the publication of the tail {misc, step sums, TH, 1.0} + sequence number into host-mapped memory by the last workgroup to finish
struct RedExtra {
    const float* th_en; int th_n; float* th_out;             // th_en != NULL: one more workgroup computes setNewFrameEnergyTH from en_new[0..th_n)
    double* pub; double seq; unsigned* ticket; const float* th_src;   // pub != NULL: mapped tail block; th_src = frameTH of the newest frame
};

// Stitch operands (kernels_ba.hip ba_stitch_kernel)
struct StitchDev {
    const double* AD;                           // [adHost (W*W*64) | adTarget (W*W*64)], index (h + t*W)*64 + i*8 + k
    const double *M_top, *M_sc;                 // acc13 [W*W][169], G [W][NPL*NPL]
    double* H;                                  // [H~_A (n1*n1) | H~_sc (n1*n1) | tail]
    unsigned* ticket;
    int W, n1, NPL;
};

}  // namespace nalo
