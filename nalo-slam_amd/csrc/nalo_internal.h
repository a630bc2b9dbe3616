// Internal state of a nalo_ctx: HBM-resident frame pyramids, tracker point clouds, BA window arrays.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>
#include <chrono>

#include "../../include/nalo_gpu.h"
#include "host_math.h"
#include "ref_constants.h"

namespace nalo {

template <typename T>
struct DevBuf {                  // owning device buffer, grows on demand
    T* p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        hipError_t e = hipMalloc((void**)&p, n * sizeof(T));
        if (e == hipSuccess) cap = n;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct FrameSlot {
    float* I[NALO_MAX_LEVELS] = {};          // planar irradiance per level
    float4* dI[NALO_MAX_LEVELS] = {};        // {I, dx, dy, 0} per level: one 16-B load per bilinear tap
    float* absg[NALO_MAX_LEVELS] = {};       // absSquaredGrad
    float* mask = nullptr;                   // level 0 (densemap only)
    uint8_t* bgr = nullptr;
    bool valid = false;
    hipEvent_t ev_up = nullptr;              // nalo_frame_upload_async: the slot's H2D copies (copy stream) have completed
    uint8_t* raw = nullptr; size_t raw_cap = 0;   // nalo_frame_upload_raw_async: this slot's sensor frame as uploaded (several frames may be in flight)
    float* dI0t = nullptr; bool tiled_valid = false;    // level 0 again as 12-byte texels in 5x2 tiles of 128 bytes (ba_linearize's gathers), made on demand (frame_tile_level0)
};

struct ProfEntry { double ms = 0; int n = 0; std::vector<std::pair<hipEvent_t, hipEvent_t>> pending; std::vector<float> samples; };   // samples: every bracketed launch, in launch order (nalo_profile_samples)

// precalc record per (host,target), 32 floats: KRKi(9) Kt(3) R0(9) t0(3) aff(2) b0 thmax dp(8)... see kernels_ba.hip
struct BAWindow;
struct PixSel;
struct Initializer;

}  // namespace nalo

struct nalo_ctx {
    int device = 0;
    int w = 0, h = 0, levels = 0;
    int wl[NALO_MAX_LEVELS], hl[NALO_MAX_LEVELS];
    float fx[NALO_MAX_LEVELS], fy[NALO_MAX_LEVELS], cx[NALO_MAX_LEVELS], cy[NALO_MAX_LEVELS];   // tracker pyramid intrinsics
    float K0[4];
    hipStream_t stream = nullptr, side = nullptr, copy = nullptr;   // copy: H2D frame uploads of nalo_frame_upload_async (overlap the kernels of `stream`)
    hipEvent_t ev_main = nullptr;            // main-stream marker the copy stream waits on before it overwrites a slot that has been used
    float* gamma_dev = nullptr;              // 256-entry gamma table of the asynchronous upload path
    float gamma_last[256]; bool gamma_have = false;   // what gamma_dev holds: the table is re-sent only when the caller's differs (and then behind every kernel that may read it)
    // raw-frame ingest (nalo_undist_set / nalo_frame_upload_raw): photometric + geometric undistortion tables, raw staging
    int und_wOrg = 0, und_hOrg = 0, und_photometric = 0, und_GDepth = 0; bool und_set = false, und_remap = false, und_vig = false;
    nalo::DevBuf<float> und_G, und_vinv, und_rxy; nalo::DevBuf<uint8_t> und_raw, und_mask, und_bgr;   // und_rxy: the remap table interleaved {x, y} (one 8-byte load per pixel in the ingest pass)
    std::string err;
    std::vector<nalo::FrameSlot> slots;

    // ---- tracker
    int slot_ref = -1;
    nalo::DevBuf<float> trk_idepth[NALO_MAX_LEVELS], trk_wsum[NALO_MAX_LEVELS], trk_wbak[NALO_MAX_LEVELS];
    nalo::DevBuf<float> pc_u[NALO_MAX_LEVELS], pc_v[NALO_MAX_LEVELS], pc_id[NALO_MAX_LEVELS], pc_col[NALO_MAX_LEVELS];
    int pc_n[NALO_MAX_LEVELS] = {};
    nalo::DevBuf<float> trk_partial;         // [blocks][64]
    nalo::DevBuf<unsigned> trk_ticket;       // trk_eval_kernel's arrival counter (zero between launches)
    nalo::DevBuf<float> ref_res; int ref_res_n = -1;   // nalo_trk_ref_upload: {Ku, Kv, new_idepth, HdiF} of the tracking reference, resident (n = -1: none)
    nalo::DevBuf<double> trk_out;            // 64 doubles
    double* trk_out_host = nullptr;          // pinned, host-mapped: results + sequence flag
    unsigned long long trk_seq = 0;
    nalo::DevBuf<unsigned long long> lm_partial;   // persistent LM kernel: [2][blocks][64] block partials {fp32, tag}
    unsigned long long lm_launches = 0;
    int lm_evals_lvl[5] = {};                // LM evaluations per pyramid level of the last persistent-kernel launch (nalo_trk_last_evals)
    int trk_rank = 0, trk_world = 1; nalo_allreduce_fn trk_hook = nullptr; void* trk_hook_user = nullptr; bool trk_hook_stream_ordered = false;   // nalo_trk_set_shard
    nalo::DevBuf<double> trk_shard_sums;     // a sharded evaluation's 52 sums on the device, summed over the ranks in place by the hook
    bool lm_host_only = false;                 // latched when a trk_lm launch lost a workgroup (CUs taken by another context): the host-driven LM loop from then on
    nalo::DevBuf<int> scan_tmp;              // compaction counts
    nalo::DevBuf<unsigned long long> dense_lb;   // nalo_dense_make_map scratch: row table | chunk aggregates | last[2] | ticket
    nalo::DevBuf<int> trk_cnt;               // hits per level-0 pixel of the reference scatter (ordered redo of pixels with >= 3 hits)
    nalo::DevBuf<float> upload_tmp;
    float* pinned_f = nullptr; size_t pinned_f_cap = 0;
    float* imm_host = nullptr; nalo::DevBuf<float> imm_dev; size_t imm_cap = 0;   // immature-point staging (pinned / device)
    nalo::DevBuf<float> imm_res; int imm_res_n = 0, imm_res_maxhost = -1;         // device-resident immature points (nalo_imm_resident_*)

    // ---- BA (opaque; defined in host_ba.cpp)
    nalo::BAWindow* ba = nullptr;
    nalo::PixSel* pixsel = nullptr;          // pixel selector state (kernels_pixsel.hip)
    void* rccl = nullptr;                    // RCCL communicators of the sharded BA (host_rccl.hip)
    bool xchg_failed = false;                // a cross-rank sum failed (host_rccl.hip): the ranks' systems may differ, every later BA call of this context fails
    nalo::Initializer* init = nullptr;       // two-frame initialiser state (host_init.hip)
    nalo_settings set = {1, nalo::kAffineOptModeA, nalo::kAffineOptModeB, 1};   // util/settings.cpp:71,128-129,74

    // ---- host wall-clock accounting (NALO_HOST_TIMING=1 prints it at nalo_destroy)
    std::map<std::string, std::pair<double, long>> host_t;

    // ---- profiling
    bool prof_on = false;
    std::string prof_only;                   // empty = every scope; else only the scope of that name is bracketed
    int prof_every = 1; unsigned prof_tick = 0;   // nalo_profile_sample: bracket one launch in prof_every (the brackets perturb a latency-bound pipeline)
    std::vector<hipEvent_t> prof_pool;       // idle events
    std::map<std::string, nalo::ProfEntry> prof;
};

namespace nalo {

inline int fail(nalo_ctx* c, int code, const std::string& msg) { if (c) c->err = msg; return code; }
#define NALO_HIP(ctx, expr)                                                                          \
    do {                                                                                             \
        hipError_t e__ = (expr);                                                                     \
        if (e__ != hipSuccess)                                                                       \
            return nalo::fail(ctx, NALO_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)); \
    } while (0)

// Wait for a kernel to publish `seq` into host-mapped memory (system-scope release on the device side). Spinning on the
// flag costs a few microseconds; hipStreamSynchronize costs tens. Every 2^20 spins the stream is queried for a fault and the wall clock is
// checked: after 10 s without the flag the call fails with NALO_ERR_HIP (a faulted kernel or a collective
// that never completes must not stall the caller for minutes).
inline bool poll_flag(nalo_ctx* c, volatile double* flag, double seq) {
    constexpr double timeout_s = 10.0;
    std::chrono::steady_clock::time_point t0;
    bool timing = false;
    for (unsigned long long spins = 0;; ++spins) {
        if (*flag == seq) { __atomic_thread_fence(__ATOMIC_ACQUIRE); return true; }
        if ((spins & 0xFFFFF) == 0xFFFFF) {
            const hipError_t e = hipStreamQuery(c->stream);
            if (e != hipSuccess && e != hipErrorNotReady) { c->err = std::string("kernel failed: ") + hipGetErrorString(e); return false; }
            const auto now = std::chrono::steady_clock::now();
            if (!timing) { t0 = now; timing = true; }
            else if (std::chrono::duration<double>(now - t0).count() > timeout_s) { c->err = "timeout waiting for the device"; return false; }
        }
        __builtin_ia32_pause();
    }
}

struct HostTimer {                // wall-clock scope, accumulated per name; a no-op unless NALO_HOST_TIMING is set (two clock reads, a std::string and a map
    nalo_ctx* c; const char* name; std::chrono::steady_clock::time_point t0;      // lookup per scope, ~50 scopes per keyframe, are not free on a 1.3 ms step)
    static bool on() { static const bool v = std::getenv("NALO_HOST_TIMING") != nullptr; return v; }
    HostTimer(nalo_ctx* ctx, const char* n) : c(ctx), name(n) { if (on()) t0 = std::chrono::steady_clock::now(); }
    ~HostTimer() { if (!on()) return; auto& e = c->host_t[name]; e.first += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); e.second++; }
};

struct ProfScope {               // HIP-event bracket on the ctx stream (only when profiling is enabled, and selected: nalo_profile_select)
    // external = true: the events are not recorded here but handed to hipExtLaunchKernelGGL, which timestamps the dispatch itself: no barrier
    // packets around the kernel (an event pair recorded on the stream costs ~10 us of bubbles per bracket on a latency-bound pipeline)
    nalo_ctx* c; const char* name; hipEvent_t a = nullptr, b = nullptr; bool external;
    ProfScope(nalo_ctx* ctx, const char* n, bool ext = false) : c(ctx), name(n), external(ext) {
        if (c->prof_on && (c->prof_only.empty() || c->prof_only == n) && (c->prof_every <= 1 || (c->prof_tick++ % (unsigned)c->prof_every) == 0)) {
            // events come from a pool (filled by nalo_profile_enable, refilled as brackets are drained): no hipEventCreate on the measured path
            for (hipEvent_t* e : {&a, &b}) { if (c->prof_pool.empty()) (void)hipEventCreate(e); else { *e = c->prof_pool.back(); c->prof_pool.pop_back(); } }
            if (!external) (void)hipEventRecord(a, c->stream);
        }
    }
    ~ProfScope() {
        if (a) { if (!external) (void)hipEventRecord(b, c->stream); c->prof[name].pending.emplace_back(a, b); }
    }
};

// kernels_imm.hip
int imm_create_launch(nalo_ctx* c, const float4* dI, int n, const int* u, const int* v, float* color, float* weights, float* gradH, float* energyTH);
int imm_trace_launch(nalo_ctx* c, const float4* dI, int n, const float* base, const int* host_idx, const float* KRKi, const float* Kt, const float* aff,
                     float* idmin, float* idmax, int* status, float* quality, float* lastUV, float* lastInterval);
int imm_optimize_resident_launch(nalo_ctx* c, const float4* const* dI, int W, const float K[4], const float* Rt, const float* aff, int n, const int* sel, const float* res, size_t N,
                                 int minObs, int* result, float* idepth_out, uint8_t* res_in);
int imm_optimize_launch(nalo_ctx* c, const float4* const* dI, int W, const float K[4], const float* Rt, const float* aff, int n, const int* host, const float* base,
                        int minObs, int* result, float* idepth_out, uint8_t* res_in);
int dist_make_launch(nalo_ctx* c, const float4* pt_geo, const uint8_t* pt_flags, const int* blk_host, int Ppad, int frame, const float* KRKi, const float* Kt, uint8_t* seed, float* out);
int pixsel_hists_launch(nalo_ctx* c, const float* absg0, float* ths, float* thsSmoothed);
// host_rccl.hip
void rccl_release(nalo_ctx* c);
// host_init.hip
void init_destroy(nalo_ctx* c);
// kernels_pixsel.hip
void pixsel_destroy(nalo_ctx* c);
void pixsel_invalidate_hists(nalo_ctx* c, int slot);
// staging for the immature-point entry points: pinned host block + device block of `floats` 4-byte words (grown on demand)
int imm_stage(nalo_ctx* c, size_t words);
int imm_put_launch(nalo_ctx* c, float* dst, const float* src, int n);
// kernels_init.hip: one calcResAndGS pass over one level. Every per-point array is a device pointer; idepth (the current inverse depths) may be NULL, then calcEC's
// three sums (slots 91..93) stay zero. sums: 96 doubles, on the device or (mapped) in host-mapped memory the caller polls.
struct InitParams {
    const float4 *colorRef, *colorNew; int wl, hl, n;
    float fx, fy, cx, cy, RKi[9], t[3], r2new0, r2new1, alphaOpt, couplingWeight;
    const float *u, *v, *idepth, *idepth_new, *iR, *energy, *outlierTH; const uint8_t* isGood;
    uint8_t* isGood_new; float *energy_new, *maxstep, *lastHessian_new, *Jb;
};
struct InitInc { float v[8]; };
int init_calc_launch(nalo_ctx* c, InitParams& P, int lvl, double* sums, int mapped = 0, double seq = 0);   // mapped: sums is host-mapped, [95] receives seq last
int init_do_step_launch(nalo_ctx* c, int n, const uint8_t* isGood, const float* Jb, const float* maxstep, const float* idepth, float lambda, const float inc[8], float* idepth_new);
int init_apply_step_launch(nalo_ctx* c, int n, uint8_t* isGood, const uint8_t* isGood_new, float* idepth, float* idepth_new, const float* iR, float* energy, const float* energy_new,
                           float* lastHessian, const float* lastHessian_new);
// the dependency-ordered sweeps (optReg: mode 0, the top level's resetPoints: mode 1) and the per-point parts of resetPoints / propagateDown / propagateUp
constexpr int kThDblAB = 1024, kThDblC = 256;   // setNewFrameEnergyTH: the three radix histograms (2048 + 2048 + 512 bins) as doubles, two bins per double = the cross-rank payloads of a sharded window
#ifndef NALO_SWEEP_NT
#define NALO_SWEEP_NT 128
#endif
constexpr int kSweepNT = NALO_SWEEP_NT;                // lanes of the sweep workgroup = the most points a schedule step may hold. trackFrame at 1224x368, same box:
                                                       // 64 lanes (no second wave at the barrier, 37 % more steps) 8.1 ms, 128: 7.4 ms, 256: 8.0 ms
constexpr size_t kSweepLdsBytes = 158 * 1024;          // of the 160 KB per workgroup
int init_sweep_launch(nalo_ctx* c, int mode, int n, int nsteps, const int* off, const int* rec, float* iR, uint8_t* isGood, float* idepth, float* idepth_new, float regWeight, float* scratch);
int init_reset_launch(nalo_ctx* c, int n, float* energy, float* idepth_new, const float* idepth);
int init_fill_launch(nalo_ctx* c, int n, float* iR, float* idepth_new, float* lastHessian);
int init_propagate_down_launch(nalo_ctx* c, int n, const int* parent, const uint8_t* pGood, const float* pLastHessian, const float* pIR, uint8_t* isGood, float* iR, float* idepth,
                               float* idepth_new, float* lastHessian);
int init_propagate_up_launch(nalo_ctx* c, int nT, const int* child_off, const int* child_idx, const uint8_t* cGood, const float* cIR, const float* cLastHessian, uint8_t* isGood,
                             float* iR, float* idepth);
// host_init.hip: the host arithmetic either side of that pass (pose / affine / alpha terms in, Accumulator9 sums -> H, b, Hsc, bsc, E out), shared by the staged
// entry point (nalo_init_calc_res_and_gs) and the resident trackFrame
struct InitPose { float alphaEnergy; };
void init_pose_setup(const nalo_ctx* c, int lvl, int n, const nalo::SE3& T, const double aff[2], float alphaW, float alphaK, float couplingWeight, InitParams& P, InitPose& X);
void init_sums_to_system(const double* sums, const nalo::SE3& T, int n, const InitParams& P, const InitPose& X, double* H, double* b, double* Hsc, double* bsc, double E3[3]);
// kernels_pyramid.hip
constexpr int NALO_LM_LOST_BLOCK = 1000;     // trk_lm_launch only (never crosses the C ABI): the persistent kernel's workgroups were not co-resident
int pyramid_build(nalo_ctx* c, nalo::FrameSlot& s, const float* gammaB_dev);
int frame_tile_level0(nalo_ctx* c, nalo::FrameSlot& s);
void hbm_stream_launch(hipStream_t st, const float4* a, const float4* b, float4* d, size_t n, int triad);
int ingest_launch(nalo_ctx* c, hipStream_t st, const void* raw, int bpp, int wOrg, int hOrg, const float* G, const float* vinv, const float2* remapXY, int photometric,
                  float factor, const uint8_t* mask_org, const uint8_t* bgr_org, float* out_I, float* out_mask, uint8_t* out_bgr);
// kernels_tracker.hip
int trk_build_ref(nalo_ctx* c, int n, const float* dKu, const float* dKv, const float* dId, const float* dHdi);
int trk_append_plane_launch(nalo_ctx* c, const float* mask, const float4* dIref, const float dir[3], float dis, float refColor, int x0, int nx, int y0, int ny, int n0, int* n_dev);
int trk_eval_launch(nalo_ctx* c, int slot_new, int lvl, const float RKi[9], const float t[3], const float Ki[9],
                    float affa, float affb, float b0, float cutoff, float maxEnergy, double out64[64]);

}  // namespace nalo
