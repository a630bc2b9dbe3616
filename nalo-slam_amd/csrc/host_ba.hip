// Host mirror of the back-end around the BA kernels: the parts of FullSystem::optimize / EnergyFunctional that the
// reference keeps as tiny fp64 host work stay on the host here too (frame states, SE3, precalc, adjoints, (8W+4)^2
// solve, nullspace projection); everything that scales with points/residuals runs in kernels_ba.hip.
// Reference paths relative to src/.
#include "nalo_internal.h"
#include "ba_device.h"

namespace nalo {

void ba_launch_sc(hipStream_t s, const BADev& B, int T, int shift, float priorScaleMarg, int margOnly);
void ba_launch_linearize(hipStream_t s, const BADev& B, int mode, int fix, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
void ba_launch_reset_oob(hipStream_t s, const BADev& B);
void ba_launch_restore(hipStream_t s, const BADev& B, const float4* geo, const uint8_t* state, const uint8_t* flags, const float* prior, const float* th);
void ba_launch_reduce(hipStream_t s, const BADev& B, const int* host_blk, int NPL, double* acc13, double* misc, double* G, bool top, bool sc,
                      const float* step_partial, int step_blocks, double* step_out, bool with_th, double* pub, double seq, unsigned* ticket);
void ba_launch_resub_step(hipStream_t s, const BADev& B, float stepfacD, float* partial, const XadArg& karg, bool karg_is_x);
int ba_launch_stitch(hipStream_t s, const StitchDev& D, bool top, bool sc, double* mapped, int ntail, double seq);
void ba_launch_resub(hipStream_t s, const BADev& B, const XadArg& karg, bool karg_is_x);
void ba_launch_resub_step_gated(hipStream_t s, const BADev& B, float stepfacD, float* partial, const GateArg& gate);
void ba_launch_pull(hipStream_t s, float* dst, const float* src_mapped, int n);
void ba_launch_step(hipStream_t s, const BADev& B, float stepfacD, float* partial, double* out3);
void ba_launch_step_sums(hipStream_t s, const BADev& B, const float* partial, double* out3);
void ba_launch_publish(hipStream_t s, const double* src, double* dst_mapped, int n, double seq, unsigned* ticket);
void ba_launch_th_install(hipStream_t s, const double* tail2, float* th);
void ba_launch_th_tail(hipStream_t s, const float* th, double* tail2);
void ba_launch_energy_th(hipStream_t s, const BADev& B);
void ba_launch_set_th(hipStream_t s, float* dst, const float* th, int W);
void ba_launch_energy_th_step(hipStream_t s, const BADev& B, int step);
void ba_launch_lenergy(hipStream_t s, const BADev& B, double* partial);
void ba_launch_load_backup(hipStream_t s, const BADev& B);
void ba_launch_swgray(hipStream_t s, const BADev& B, const double* Rt, double* partial);
void ba_launch_set_idepth(hipStream_t s, const BADev& B, int mode, int host_sel, double scale);

struct HostFrame {
    int slot = 0, frameID = 0;
    SE3 evalPT, PRE_worldToCam, PRE_camToWorld;
    double state[10] = {}, state_zero[10] = {}, state_scaled[10] = {}, step[10] = {}, state_backup[10] = {};
    float ab_exposure = 1.f, frameEnergyTH = 8 * 8 * kPatternNum;
    double prior[8] = {}, delta[8] = {}, delta_prior[8] = {};
    double ns_pose[6][6] = {}, ns_scale[6] = {};
};

struct BAWindow {
    int W = 0, P = 0, Ppad = 0, nblocks = 0, n = 0, n1 = 0, T = 1, NPL = 16;
    // CalibHessian (FullSystem/HessianBlocks.h:364-395)
    double c_value[4] = {}, c_value_zero[4] = {}, c_value_scaled[4] = {}, c_step[4] = {}, c_value_backup[4] = {};
    float c_scaledf[4] = {}, c_scaledi[4] = {};
    std::vector<HostFrame> frames;
    std::vector<double> adHost, adTarget, HM, bM, lastX, Sproj, solve_scratch;
    std::vector<double> ad_key; int ad_key_W = 0;                   // what the adjoints of a frame were last computed from (set_adjoints recomputes the pairs of changed frames only)
    std::vector<int> solve_perm;
    std::vector<float> adHostF, adTargetF, adHTdeltaF;
    float cDeltaF[4] = {};
    bool proj_valid = false;
    bool ad_pending = false;               // the host adjoint tables are newer than the device copy (upload_adjoints)
    int resInA = 0, resInL = 0, resInM = 0;
    // permutation: device slot d -> caller point (or -1), caller point -> device slot
    std::vector<int> d2p, p2d, blk_host_h, host_blk_h;
    std::vector<uint8_t> flags_h;
    // device
    BADev dev{};
    DevBuf<float> pre, frameTH, pt_prior, pt_step, pt_backup, pt_relbs, pt_relbs2, en_new, step_partial;
    DevBuf<double> top_partial, sc_partial;
    unsigned long long pub_seq = 0; bool step_pending = false; int last_canbreak = 0; float st_sumA = 0, st_sumB = 0, st_sumT = 0, st_sumR = 0;
    DevBuf<float4> pt_geo, pt_col0, pt_col1, pt_w0, pt_w1, pt_acc, pt_hcd, rs_jp0, rs_jp1, rs_cpt;
    DevBuf<float2> rs_energy, rs_pp1;
    DevBuf<float4> rs_pp0;
    DevBuf<uint8_t> pt_flags, pt_ngood, rs_state;
    DevBuf<int> blk_host, host_blk, blk_order, sc_grp;
    DevBuf<unsigned> th_hist;                                        // the radix select's state words (kernels_ba.hip: ba_th_fill_kernel)
    bool step_fused = false, step_sums_deferred = false;            // optimize(): resubstitute + point step in one kernel, its sums finished by the reduce launch
    bool th_pending = false;                                        // a linearize pass whose frameEnergyTH quantile has not been launched yet
    DevBuf<double> acc13, G, AD, stitched;                      // stitched: [H~_A ((n1)^2) | H~_sc ((n1)^2) | misc (2 W^2) | step sums (3) | TH sum, ranks]
    DevBuf<unsigned> st_ticket;
    StitchDev sd{};
    double* ad_host = nullptr; size_t ad_cap = 0;                   // pinned staging of AD
    hipEvent_t ev_ad = nullptr;
    double* stitched_host = nullptr;                                // pinned mirror
    float* up_host = nullptr;                                       // pinned staging of the step: [- | calib 16 | xc 64 | xAd] (the precalc records live in pre_map)
    size_t up_cap = 0;
    // small windows: the precalc records are not copied at all - ba_linearize reads them (scalar loads, one record per workgroup) straight from mapped host
    // memory; four rotating blocks, a block is rewritten only after the host has seen the stream drain past its last reader (pre_synced)
    float* pre_map[4] = {nullptr, nullptr, nullptr, nullptr}; float* pre_map_dev[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t pre_map_cap = 0; long pre_pos = 0, pre_synced = 0;
    bool have_lin = false, have_sc = false, stitched_top = false, stitched_sc = false, points_set = false, res_set = false;
    bool pt_acc_on_read = false;                                    // nalo_ba_get_points may run the per-point accumulation of an explicit nalo_ba_linearize
    int sc_shift = -1;
    // snapshot of the mutable window state (bench/test utility: replay the same synthetic keyframe)
    DevBuf<float4> snap_geo; DevBuf<uint8_t> snap_state, snap_flags; DevBuf<float> snap_prior;
    std::vector<HostFrame> snap_frames; std::vector<double> snap_HM, snap_bM; std::vector<uint8_t> snap_flags_h; bool have_snap = false;
    double snap_calib[4] = {}, snap_calib_scaled[4] = {}; float snap_scaledf[4] = {}, snap_scaledi[4] = {};   // CalibHessian as it is: a fresh window holds value_scaled = K exactly,
                                                                    // setValue(value) would round it through SCALE_F * (SCALE_F_INVERSE * K)
    // small single-GPU windows: the back-substitution and the next linearisation of optimize() are enqueued BEFORE the host has solved the system (ba_device.h: GateBlock)
    GateBlock* gate_h = nullptr; GateBlock* gate_d = nullptr; unsigned gate_x_seq = 0, gate_p_seq = 0;
    bool want_prelaunch = false, resub_pre = false, lin_pre = false;
    nalo_allreduce_fn hook = nullptr;
    void* hook_user = nullptr;
    bool hook_stream_ordered = false;                               // the hook enqueues its collective on nalo_stream(ctx): no host synchronisation around it
    nalo_allreduce_fn hook_side = nullptr; void* hook_side_user = nullptr;   // same sum, enqueued on nalo_side_stream(ctx): the threshold's histograms
    DevBuf<double> th_buf;                                          // level A | level B histograms of the radix select as doubles (2 x kThDblAB: the side stream's all-reduce payloads)
    hipEvent_t ev_lin = nullptr, ev_th = nullptr; bool th_side_inflight = false;   // sharded windows run the quantile kernels on the side stream, under the SC kernel
    bool th_lo_pending = false;                                     // level C's histogram sits behind the stitched systems (lo_off), not yet summed over the ranks: the next
                                                                    // stitch sums it WITH the systems in one all-reduce, anything else that needs the threshold sums it alone
    size_t lo_off = 0;
    bool never_break = false;
    DevBuf<double> small_sum;                                        // a few scalars on their way through the all-reduce hook (sum_over_ranks)
    DevBuf<double> noapply_E; std::vector<double> noapply_h;          // energy partials of a linearisation that is not applied (forceAcceptStep = false)
    int opt_iterations = 0, opt_rejected = 0;
    bool prior_next = false;                                        // nalo_ba_marginalize_frame has left HM / bM for the NEXT nalo_ba_set_window (kept or extended there)
    bool prior_carry = false;                                       // nalo_ba_set_prior_carry: the context is ONE continuing EnergyFunctional, every set_window keeps / extends
};

void ba_destroy(nalo_ctx* c) {
    BAWindow* w = c->ba;
    if (!w) return;
    w->pre.release(); w->frameTH.release(); w->pt_prior.release(); w->pt_step.release(); w->pt_backup.release(); w->pt_relbs.release(); w->pt_relbs2.release();
    w->small_sum.release(); w->th_buf.release(); w->en_new.release(); w->top_partial.release(); w->sc_partial.release(); w->step_partial.release();
    w->pt_geo.release(); w->pt_col0.release(); w->pt_col1.release(); w->pt_w0.release(); w->pt_w1.release(); w->pt_acc.release(); w->pt_hcd.release();
    w->rs_jp0.release(); w->rs_jp1.release(); w->rs_cpt.release(); w->rs_energy.release(); w->rs_pp0.release(); w->rs_pp1.release(); w->pt_flags.release(); w->pt_ngood.release(); w->rs_state.release();
    w->blk_host.release(); w->host_blk.release(); w->sc_grp.release(); w->blk_order.release(); w->acc13.release(); w->G.release(); w->AD.release(); w->st_ticket.release();
    w->stitched.release(); w->th_hist.release();
    if (w->ev_lin) (void)hipEventDestroy(w->ev_lin);
    if (w->ev_th) (void)hipEventDestroy(w->ev_th);
    w->snap_geo.release(); w->snap_state.release(); w->snap_flags.release(); w->snap_prior.release();
    if (w->gate_h) (void)hipHostFree(w->gate_h);
    if (w->stitched_host) (void)hipHostFree(w->stitched_host);
    if (w->up_host) (void)hipHostFree(w->up_host);
    for (float* p : w->pre_map) if (p) (void)hipHostFree(p);
    if (w->ad_host) (void)hipHostFree(w->ad_host);
    if (w->ev_ad) (void)hipEventDestroy(w->ev_ad);
    delete w;
    c->ba = nullptr;
}

// ---------------------------------------------------------------------------------------------- frame / calib state
static void calib_set_value(BAWindow& w, const double v[4]) {            // CalibHessian::setValue
    for (int i = 0; i < 4; ++i) w.c_value[i] = v[i];
    w.c_value_scaled[0] = kScaleF * v[0]; w.c_value_scaled[1] = kScaleF * v[1]; w.c_value_scaled[2] = kScaleC * v[2]; w.c_value_scaled[3] = kScaleC * v[3];
    for (int i = 0; i < 4; ++i) w.c_scaledf[i] = (float)w.c_value_scaled[i];
    w.c_scaledi[0] = 1.0f / w.c_scaledf[0]; w.c_scaledi[1] = 1.0f / w.c_scaledf[1];
    w.c_scaledi[2] = -w.c_scaledf[2] / w.c_scaledf[0]; w.c_scaledi[3] = -w.c_scaledf[3] / w.c_scaledf[1];
}
static void frame_set_state(HostFrame& f, const double st[10]) {          // FrameHessian::setState (HessianBlocks.h:208-222)
    std::memcpy(f.state, st, sizeof(f.state));
    for (int i = 0; i < 3; ++i) f.state_scaled[i] = kScaleXiTrans * st[i];
    for (int i = 3; i < 6; ++i) f.state_scaled[i] = kScaleXiRot * st[i];
    f.state_scaled[6] = kScaleA * st[6]; f.state_scaled[7] = kScaleB * st[7]; f.state_scaled[8] = kScaleA * st[8]; f.state_scaled[9] = kScaleB * st[9];
    f.PRE_worldToCam = se3_exp(f.state_scaled) * f.evalPT;
    f.PRE_camToWorld = f.PRE_worldToCam.inverse();
}
static void frame_set_state_zero(HostFrame& f, const double sz[10]) {     // FrameHessian::setStateZero (HessianBlocks.cpp:73-106)
    std::memcpy(f.state_zero, sz, sizeof(f.state_zero));
    const SE3 inv = f.evalPT.inverse();
    for (int i = 0; i < 6; ++i) {
        double eps[6] = {0, 0, 0, 0, 0, 0}, lp[6], lm[6];
        eps[i] = 1e-3; const SE3 P = (f.evalPT * se3_exp(eps)) * inv;
        eps[i] = -1e-3; const SE3 M = (f.evalPT * se3_exp(eps)) * inv;
        se3_log(P, lp); se3_log(M, lm);
        for (int r = 0; r < 6; ++r) f.ns_pose[i][r] = (lp[r] - lm[r]) / (2e-3);
    }
    SE3 P = f.evalPT, M = f.evalPT;
    for (int i = 0; i < 3; ++i) { P.m[i * 4 + 3] *= 1.00001; M.m[i * 4 + 3] /= 1.00001; }
    double lp[6], lm[6];
    se3_log(P * inv, lp); se3_log(M * inv, lm);
    for (int r = 0; r < 6; ++r) f.ns_scale[r] = (lp[r] - lm[r]) / (2e-3);
}
static void frame_take_data(const nalo_settings& set, HostFrame& f) {     // EFFrame::takeData + FrameHessian::getPrior (HessianBlocks.h:286-312)
    double p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (f.frameID == 0) { for (int i = 0; i < 3; ++i) p[i] = kInitialTransPrior; for (int i = 3; i < 6; ++i) p[i] = kInitialRotPrior; p[6] = kInitialAffPrior; p[7] = kInitialAffPrior; }
    else { p[6] = set.affineOptModeA < 0 ? kInitialAffPrior : set.affineOptModeA; p[7] = set.affineOptModeB < 0 ? kInitialAffPrior : set.affineOptModeB; }
    for (int i = 0; i < 8; ++i) { f.prior[i] = p[i]; f.delta[i] = f.state[i] - f.state_zero[i]; f.delta_prior[i] = f.state[i]; }
}

// EnergyFunctional::setAdjointsF (OptimizationBackend/EnergyFunctional.cpp:46-106); the values feed the stitch matrices
static int upload_adjoints(nalo_ctx* c);
// defer = true (the epilogue of optimize()): the host tables are rebuilt now, the device copy follows behind the epilogue's own kernels (none of them reads it)
static int set_adjoints(nalo_ctx* c, bool defer = false) {
    BAWindow& w = *c->ba;
    HostTimer ht(c, "ba.set_adjoints");
    const int W = w.W, n1 = w.n1;
    // A pair (h, t) depends on the two frames' linearisation points, exposures and affine zero states only. The calls on the per-keyframe path change ONE frame (the
    // epilogue of optimize() moves the newest frame's evalPT, FullSystemOptimize.cpp:550-557; a snapshot restore moves it back): 2W - 1 of the W^2 pairs, not all.
    constexpr int KEY = 15;
    std::vector<uint8_t> dirty(W, 1);
    if (w.ad_key_W == W && w.ad_key.size() == (size_t)W * KEY && w.adHost.size() == (size_t)W * W * 64) {
        for (int f = 0; f < W; ++f) {
            double k[KEY]; std::memcpy(k, w.frames[f].evalPT.m, 12 * sizeof(double)); k[12] = w.frames[f].ab_exposure; k[13] = w.frames[f].state_zero[6]; k[14] = w.frames[f].state_zero[7];
            dirty[f] = std::memcmp(k, &w.ad_key[(size_t)f * KEY], sizeof(k)) != 0;
        }
    } else {
        w.adHost.assign((size_t)W * W * 64, 0.0); w.adTarget.assign((size_t)W * W * 64, 0.0);
        w.adHostF.resize((size_t)W * W * 64); w.adTargetF.resize((size_t)W * W * 64);
    }
    w.ad_key.resize((size_t)W * KEY); w.ad_key_W = W;
    for (int f = 0; f < W; ++f) { double* k = &w.ad_key[(size_t)f * KEY]; std::memcpy(k, w.frames[f].evalPT.m, 12 * sizeof(double)); k[12] = w.frames[f].ab_exposure; k[13] = w.frames[f].state_zero[6]; k[14] = w.frames[f].state_zero[7]; }
    bool any = false;
    for (int h = 0; h < W; ++h) for (int t = 0; t < W; ++t) {
        if (!dirty[h] && !dirty[t]) continue;
        any = true;
        const HostFrame &host = w.frames[h], &target = w.frames[t];
        const SE3 h2t = target.evalPT * host.evalPT.inverse();
        double Ad[36]; h2t.adjoint(Ad);
        double* AH = &w.adHost[(size_t)(h + t * W) * 64];
        double* AT = &w.adTarget[(size_t)(h + t * W) * 64];
        std::fill(AH, AH + 64, 0.0); std::fill(AT, AT + 64, 0.0);
        for (int i = 0; i < 8; ++i) { AH[i * 8 + i] = 1; AT[i * 8 + i] = 1; }
        for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) AH[i * 8 + j] = -Ad[j * 6 + i];
        double a[2];
        aff_from_to(host.ab_exposure, target.ab_exposure, host.state_zero[6] * kScaleA, host.state_zero[7] * kScaleB,
                    target.state_zero[6] * kScaleA, target.state_zero[7] * kScaleB, a);
        const float affLL0 = (float)a[0];
        AT[6 * 8 + 6] = -affLL0; AH[6 * 8 + 6] = affLL0; AT[7 * 8 + 7] = -1; AH[7 * 8 + 7] = affLL0;
        for (int j = 0; j < 8; ++j) {
            for (int i = 0; i < 3; ++i) { AH[i * 8 + j] *= kScaleXiTrans; AT[i * 8 + j] *= kScaleXiTrans; }
            for (int i = 3; i < 6; ++i) { AH[i * 8 + j] *= kScaleXiRot; AT[i * 8 + j] *= kScaleXiRot; }
            AH[6 * 8 + j] *= kScaleA; AT[6 * 8 + j] *= kScaleA; AH[7 * 8 + j] *= kScaleB; AT[7 * 8 + j] *= kScaleB;
        }
        for (int k = 0; k < 64; ++k) { w.adHostF[(size_t)(h + t * W) * 64 + k] = (float)AH[k]; w.adTargetF[(size_t)(h + t * W) * 64 + k] = (float)AT[k]; }
    }
    w.proj_valid = false;
    if (!w.st_ticket.p) { NALO_HIP(c, w.st_ticket.reserve(4)); NALO_HIP(c, hipMemset(w.st_ticket.p, 0, 16)); }
    w.sd.ticket = w.st_ticket.p; w.sd.W = W; w.sd.n1 = n1; w.sd.NPL = w.NPL;
    if (!any && w.sd.AD == w.AD.p && w.AD.p && !w.ad_pending) return NALO_OK;        // nothing moved: the device copy is current
    w.ad_pending = true;
    return defer ? NALO_OK : upload_adjoints(c);
}
static int upload_adjoints(nalo_ctx* c) {
    BAWindow& w = *c->ba;
    if (!w.ad_pending) return NALO_OK;
    const int W = w.W;
    // the stitch kernel reads the fp64 adjoints: AD = [adHost | adTarget]
    // (behind them, in the same upload: the float adjoints the back-substitution of a window of more than 8 frames builds its xAd rows from)
    const size_t nad = (size_t)2 * W * W * 64;
    if (w.ad_cap < nad) { if (w.ad_host) (void)hipHostFree(w.ad_host); NALO_HIP(c, hipHostMalloc((void**)&w.ad_host, nad * 12)); w.ad_cap = nad; }
    if (!w.ev_ad) NALO_HIP(c, hipEventCreateWithFlags(&w.ev_ad, hipEventDisableTiming));
    else NALO_HIP(c, hipEventSynchronize(w.ev_ad));                  // the previous upload has left the staging buffer
    std::memcpy(w.ad_host, w.adHost.data(), (size_t)W * W * 64 * 8);
    std::memcpy(w.ad_host + (size_t)W * W * 64, w.adTarget.data(), (size_t)W * W * 64 * 8);
    float* adf = reinterpret_cast<float*>(w.ad_host + nad);
    std::memcpy(adf, w.adHostF.data(), (size_t)W * W * 64 * 4);
    std::memcpy(adf + (size_t)W * W * 64, w.adTargetF.data(), (size_t)W * W * 64 * 4);
    NALO_HIP(c, w.AD.reserve(nad + nad / 2));
    NALO_HIP(c, hipMemcpyAsync(w.AD.p, w.ad_host, nad * 12, hipMemcpyHostToDevice, c->stream));
    w.dev.adF = reinterpret_cast<const float*>(w.AD.p + nad);
    NALO_HIP(c, hipEventRecord(w.ev_ad, c->stream));
    w.sd.AD = w.AD.p;
    w.ad_pending = false;
    return NALO_OK;
}

// FullSystem::setPrecalcValues = FrameFramePrecalc::set for all pairs (HessianBlocks.cpp:192-222) + EnergyFunctional::setDeltaF (:171-194)
static int set_precalc(nalo_ctx* c) {
    BAWindow& w = *c->ba;
    HostTimer ht(c, "ba.set_precalc");
    const int W = w.W;
    w.adHTdeltaF.resize((size_t)W * W * 8);
    for (int h = 0; h < W; ++h) for (int t = 0; t < W; ++t) {
        const int idx = h + t * W;
        float dh[8], dt[8];
        for (int i = 0; i < 8; ++i) { dh[i] = (float)(w.frames[h].state[i] - w.frames[h].state_zero[i]); dt[i] = (float)(w.frames[t].state[i] - w.frames[t].state_zero[i]); }
        for (int j = 0; j < 8; ++j) {
            float s1 = 0, s2 = 0;
            for (int i = 0; i < 8; ++i) { s1 += dh[i] * w.adHostF[(size_t)idx * 64 + i * 8 + j]; s2 += dt[i] * w.adTargetF[(size_t)idx * 64 + i * 8 + j]; }
            w.adHTdeltaF[(size_t)idx * 8 + j] = s1 + s2;
        }
    }
    for (int i = 0; i < 4; ++i) w.cDeltaF[i] = (float)(w.c_value[i] - w.c_value_zero[i]);
    for (auto& f : w.frames) frame_take_data(c->set, f);
    const size_t nfl = (size_t)W * W * kPreStride;
    if (w.up_cap < nfl + 80) {                                             // [pre | calib 16 | xc 64 | xAd]; hipHostFree waits for copies in flight
        if (w.up_host) (void)hipHostFree(w.up_host);
        w.up_host = nullptr; w.up_cap = 0;
        NALO_HIP(c, hipHostMalloc((void**)&w.up_host, (nfl + 80 + (size_t)W * W * 8) * 4));
        w.up_cap = nfl + 80;
    }
    // Small windows (the KITTI-sized ones, latency bound): no copy. The H2D blit of these 8 KB sat between the back-substitution and the next linearisation
    // with ~10 us of pipeline gaps around it plus ~5 us of runtime calls on the host; the kernels read the records from mapped host memory instead (scalar
    // loads, a few hundred bytes per workgroup over PCIe: +2 us inside ba_linearize). Large windows keep the device copy (thousands of workgroups).
    // Round 4, measured and NOT kept for large windows: read in place there too, the records are fetched from the host once per XCD (its L2 keeps them for the launch),
    // W^2 x 160 B x 8 over PCIe per linearisation = 3 us at W = 8, 7 us at W = 12, INSIDE ba_linearize (the roofline kernel: 161 -> 164-171 us on stress250k), against ~10 us
    // of pull kernel + boundary beside it: stress250k 2.064 -> 2.002 ms per keyframe, the emulated N = 8 shard of the 12-frame window 1.72 -> 1.76 ms (same-box A/B).
    const bool direct = w.points_set && w.Ppad <= 32768;
    // Either way the records are written into a ring of four mapped, coherent host blocks (a block's last readers may still run: the ring is what lets do_step ->
    // set_precalc and the epilogue's set_precalc follow each other without a wait). Large windows: ONE workgroup pulls the block into device memory (ba_pull_kernel)
    // instead of a copy packet + its event.
    float* rec;
    {
        const size_t nfl4 = (nfl + 16 + 3) & ~(size_t)3;
        if (w.pre_map_cap < nfl4) {
            NALO_HIP(c, hipStreamSynchronize(c->stream));
            for (int i = 0; i < 4; ++i) {
                if (w.pre_map[i]) (void)hipHostFree(w.pre_map[i]);
                w.pre_map[i] = nullptr;
                NALO_HIP(c, hipHostMalloc((void**)&w.pre_map[i], nfl4 * 4, hipHostMallocMapped | hipHostMallocCoherent));
                NALO_HIP(c, hipHostGetDevicePointer((void**)&w.pre_map_dev[i], w.pre_map[i], 0));
            }
            w.pre_map_cap = nfl4; w.pre_synced = w.pre_pos;
        }
        ++w.pre_pos;
        if (w.pre_pos - 4 > w.pre_synced) { NALO_HIP(c, hipStreamSynchronize(c->stream)); w.pre_synced = w.pre_pos - 1; }   // the block's last readers may still run
        rec = w.pre_map[w.pre_pos & 3];
    }
    const float fx = w.c_scaledf[0], fy = w.c_scaledf[1], cx = w.c_scaledf[2], cy = w.c_scaledf[3];
    const float K[9] = {fx, 0, cx, 0, fy, cy, 0, 0, 1}, Ki[9] = {1.0f / fx, 0, -cx / fx, 0, 1.0f / fy, -cy / fy, 0, 0, 1};
    SE3 evalInv[NALO_MAX_WINDOW];                                          // W inverses, not W^2 (this runs between the back-substitution and the next linearisation of every iteration)
    for (int h = 0; h < W; ++h) evalInv[h] = w.frames[h].evalPT.inverse();
    for (int h = 0; h < W; ++h) for (int t = 0; t < W; ++t) {
        float* o = rec + (size_t)(h * W + t) * kPreStride;
        std::memset(o, 0, kPreStride * 4);
        const HostFrame &host = w.frames[h], &target = w.frames[t];
        const SE3 l0 = target.evalPT * evalInv[h];
        const SE3 ll = target.PRE_worldToCam * host.PRE_camToWorld;
        float R[9], tt[3], KR[9];
        for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) { o[12 + i * 3 + j] = (float)l0.R(i, j); R[i * 3 + j] = (float)ll.R(i, j); } o[21 + i] = (float)l0.t(i); tt[i] = (float)ll.t(i); }
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) KR[i * 3 + j] = K[i * 3] * R[j] + K[i * 3 + 1] * R[3 + j] + K[i * 3 + 2] * R[6 + j];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) o[i * 3 + j] = KR[i * 3] * Ki[j] + KR[i * 3 + 1] * Ki[3 + j] + KR[i * 3 + 2] * Ki[6 + j];
        for (int i = 0; i < 3; ++i) o[9 + i] = K[i * 3] * tt[0] + K[i * 3 + 1] * tt[1] + K[i * 3 + 2] * tt[2];
        double a[2];
        aff_from_to(host.ab_exposure, target.ab_exposure, host.state_scaled[6], host.state_scaled[7], target.state_scaled[6], target.state_scaled[7], a);
        o[24] = (float)a[0]; o[25] = (float)a[1]; o[26] = (float)(host.state_zero[7] * kScaleB);
        for (int k = 0; k < 8; ++k) o[27 + k] = w.adHTdeltaF[(size_t)(h + t * W) * 8 + k];
    }
    float* cal = rec + nfl;                                               // CalibHessian::value_scaledf / value_scaledi + cDeltaF travel with the records
    cal[0] = fx; cal[1] = fy; cal[2] = cx; cal[3] = cy; cal[4] = w.c_scaledi[0]; cal[5] = w.c_scaledi[1];
    for (int i = 0; i < 4; ++i) cal[6 + i] = w.cDeltaF[i];
    NALO_HIP(c, w.pre.reserve(nfl + 32));
    if (direct) {
        w.dev.pre = w.pre_map_dev[w.pre_pos & 3]; w.dev.calib = w.dev.pre + nfl;
        // a linearisation that was enqueued ahead of these records waits for this word (x86 stores are not reordered with earlier stores; the fence keeps the compiler honest)
        if (w.lin_pre && w.gate_h) { __atomic_thread_fence(__ATOMIC_RELEASE); __atomic_store_n(&w.gate_h->p_seq, w.gate_p_seq, __ATOMIC_RELEASE); }
        return NALO_OK;
    }
    ba_launch_pull(c->stream, w.pre.p, w.pre_map_dev[w.pre_pos & 3], (int)(nfl + 16));
    NALO_HIP(c, hipGetLastError());
    w.dev.pre = w.pre.p; w.dev.calib = w.pre.p + nfl;

    return NALO_OK;
}

// every cross-rank sum goes through here: a failed collective (host_rccl.hip latches it) must not be followed by a solve on rank-local sums
static int call_hook(nalo_ctx* c, nalo_allreduce_fn fn, void* user, double* buf, int n) {
    fn(user, buf, n);
    if (c->xchg_failed) return NALO_ERR_HIP;                          // message already in c->err
    return NALO_OK;
}
// n (<= 8) host scalars summed over the ranks of a sharded window, in place; a single-GPU window returns at once. What the energy test of
// setting_forceAceptStep = false needs beyond the systems: the energy of a linearisation that is not applied, the point part of calcLEnergyPt and the step sums of
// doStepFromBackup's break test are all sums over the active points, i.e. over the shards (FullSystemOptimize.cpp:511-541, EnergyFunctional.cpp:332-392)
static int sum_over_ranks(nalo_ctx* c, double* v, int n) {
    BAWindow& w = *c->ba;
    if (!w.hook) return NALO_OK;
    NALO_HIP(c, w.small_sum.reserve(8));
    NALO_HIP(c, hipMemcpyAsync(w.small_sum.p, v, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    if (!w.hook_stream_ordered) NALO_HIP(c, hipStreamSynchronize(c->stream));
    { int rh = call_hook(c, w.hook, w.hook_user, w.small_sum.p, n); if (rh) return rh; }
    NALO_HIP(c, hipMemcpyAsync(v, w.small_sum.p, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    NALO_HIP(c, hipStreamSynchronize(c->stream));
    return NALO_OK;
}
static int flush_th(nalo_ctx* c);
static int upload_frame_th(nalo_ctx* c) {
    BAWindow& w = *c->ba;
    if (w.points_set && w.dev.th_bufAB) { int rc = flush_th(c); if (rc) return rc; }   // a pending quantile pass also clears its histograms
    std::vector<float> th(w.W);
    for (int i = 0; i < w.W; ++i) th[i] = w.frames[i].frameEnergyTH;
    NALO_HIP(c, w.frameTH.reserve(16));
    ba_launch_set_th(c->stream, w.frameTH.p, th.data(), w.W);              // W <= 16 values as kernel arguments: stream ordered, nothing to wait for
    w.th_pending = false;
    w.dev.frameTH = w.frameTH.p;
    return NALO_OK;
}

// ---------------------------------------------------------------------------------------------- pipeline pieces
// setNewFrameEnergyTH of the last linearize pass (FullSystemOptimize.cpp:95-143): three small kernels on the main stream. They are launched
// lazily: after the stitch has been published (so they run while the host solves), or at the latest before the next pass that reads the value.
static int flush_th(nalo_ctx* c) {
    BAWindow& w = *c->ba;
    if (w.th_side_inflight) { NALO_HIP(c, hipStreamWaitEvent(c->stream, w.ev_th, 0)); w.th_side_inflight = false; }
    if (w.th_lo_pending) {                                            // no stitch took level C's histogram along: its own all-reduce, then the search
        w.th_lo_pending = false;
        if (!w.hook_stream_ordered) NALO_HIP(c, hipStreamSynchronize(c->stream));
        { int rh = call_hook(c, w.hook, w.hook_user, w.stitched.p + w.lo_off, kThDblC); if (rh) return rh; }
        ba_launch_energy_th_step(c->stream, w.dev, 3);
        NALO_HIP(c, hipGetLastError());
    }
    if (!w.th_pending) return NALO_OK;
    ba_launch_energy_th(c->stream, w.dev);
    w.th_pending = false;
    NALO_HIP(c, hipGetLastError());
    return NALO_OK;
}
// keep_th: the pass that APPLIES a step accepted by the energy test re-runs the linearisation the test was made on: same frameEnergyTH (no flush), and it
// does not feed setNewFrameEnergyTH again (the untested pass already did)
// the tiled level-0 copies ba_linearize gathers from: made on demand, remade when a window frame's pyramid was rebuilt since
static int ensure_tiled(nalo_ctx* c) {
    BAWindow& w = *c->ba;
    for (int i = 0; i < w.W; ++i) {
        FrameSlot& s = c->slots[w.frames[i].slot];
        if (!s.tiled_valid) { int rc = frame_tile_level0(c, s); if (rc) return rc; }
        w.dev.img_t[i] = s.dI0t;
    }
    w.dev.wt = (c->w + 4) / 5;
    return NALO_OK;
}
static int linearize_async(nalo_ctx* c, int mode, int fix, bool keep_th = false) {
    BAWindow& w = *c->ba;
    if (w.lin_pre) {
        // the kernel of this pass is already in the stream (prelaunch_iteration), and set_precalc has just opened its gate: what is left is the bookkeeping
        w.lin_pre = false;
        if (mode != 0 || fix != 0 || keep_th) return fail(c, NALO_ERR_STATE, "linearize_async: a pre-enqueued pass is a plain linearizeAll(false)");
        w.th_pending = true;
        w.have_lin = true; w.have_sc = false; w.stitched_top = false; w.stitched_sc = false;
        return NALO_OK;
    }
    { int rc = ensure_tiled(c); if (rc) return rc; }
    if (keep_th) {
        w.dev.no_th = 1;
        { ProfScope ps(c, "ba_linearize", true); ba_launch_linearize(c->stream, w.dev, mode, fix, ps.a, ps.b); }
        w.dev.no_th = 0;
        w.have_lin = true; w.have_sc = false; w.stitched_top = false; w.stitched_sc = false;
        NALO_HIP(c, hipGetLastError());
        return NALO_OK;
    }
    int rc = flush_th(c); if (rc) return rc;                          // frameEnergyTH of the previous pass feeds this one
    // a pass that records relBS (fix / marginalisation) takes the clean buffer of the pair and leaves the other one clean (its idle workgroups zero it): no fill launch
    if (fix == 1 || mode == 2) std::swap(w.dev.pt_relbs, w.dev.pt_relbs_next);
    const bool th_sharded = mode == 0 && w.hook;
    const bool on_side = th_sharded && !(w.hook_stream_ordered && !w.hook_side);
    // (device-scope events: both sides of these dependencies are kernels of this context; a system-scope release behind the linearisation writes its output back first)
    if (th_sharded && !w.ev_lin) { NALO_HIP(c, hipEventCreateWithFlags(&w.ev_lin, hipEventDisableTiming | hipEventDisableSystemFence)); NALO_HIP(c, hipEventCreateWithFlags(&w.ev_th, hipEventDisableTiming | hipEventDisableSystemFence)); }
    hipEvent_t lin_done = nullptr;
    {
        // the side stream's dependency on this pass is the dispatch's own completion signal (hipExtLaunchKernelGGL's stop event): an event RECORDED behind the kernel
        // is a marker packet of its own, and the next kernel of the main stream stood 14 us behind it
        ProfScope ps(c, "ba_linearize", true);
        lin_done = ps.b ? ps.b : (on_side ? w.ev_lin : nullptr);
        ba_launch_linearize(c->stream, w.dev, mode, fix, ps.a, lin_done);
    }
    if (th_sharded) {
        // sharded window: the threshold is the EXACT order statistic over all ranks' residuals (what one GPU holding the whole window computes): each level's
        // histogram of the three-level radix select (kernels_ba.hip) is summed across the ranks before the next level's search. Levels A and B (8 KB each) are
        // filled, summed and searched here - on the side stream under pt_acc / SC / reduce / stitch when there is a side hook or the main hook blocks, in line
        // otherwise -; level C (2 KB) is left behind the stitched systems and summed together with them (stitch_and_fetch): three small kernels, two 8 KB
        // collectives beside the main stream and ONE collective of the systems on it per pass.
        hipStream_t st = on_side ? c->side : c->stream;
        nalo_allreduce_fn fn = (on_side && w.hook_side) ? w.hook_side : w.hook;
        void* user = (on_side && w.hook_side) ? w.hook_side_user : w.hook_user;
        if (on_side) NALO_HIP(c, hipStreamWaitEvent(c->side, lin_done, 0));
        for (int lvl = 0; lvl < 2; ++lvl) {
            ba_launch_energy_th_step(st, w.dev, lvl);
            if (!w.hook_stream_ordered) NALO_HIP(c, hipStreamSynchronize(st));
            { int rh = call_hook(c, fn, user, w.th_buf.p + lvl * kThDblAB, kThDblAB); if (rh) return rh; }
        }
        ba_launch_energy_th_step(st, w.dev, 2);
        if (on_side) { NALO_HIP(c, hipEventRecord(w.ev_th, c->side)); w.th_side_inflight = true; }
        w.th_lo_pending = true;
    } else if (mode == 0) w.th_pending = true;
    if (fix != 2) { w.have_lin = true; w.have_sc = false; w.stitched_top = false; w.stitched_sc = false; }
    NALO_HIP(c, hipGetLastError());
    return NALO_OK;
}
// linearizeAll(false) without applyRes: the energy of the current states (lastEnergyP), nothing else changes but state_NewEnergy and the threshold input
static int linearize_noapply(nalo_ctx* c, double* E) {
    BAWindow& w = *c->ba;
    const size_t n = (size_t)w.nblocks * w.dev.lin_sub * w.W;
    NALO_HIP(c, w.noapply_E.reserve(n));
    w.dev.noapply_E = w.noapply_E.p;
    NALO_HIP(c, hipMemsetAsync(w.noapply_E.p, 0, n * sizeof(double), c->stream));
    int rc = linearize_async(c, 0, 2); if (rc) return rc;
    w.noapply_h.resize(n);
    NALO_HIP(c, hipMemcpyAsync(w.noapply_h.data(), w.noapply_E.p, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    NALO_HIP(c, hipStreamSynchronize(c->stream));
    double s = 0; for (double v : w.noapply_h) s += v;
    { int rs = sum_over_ranks(c, &s, 1); if (rs) return rs; }
    *E = s;
    return NALO_OK;
}
// calcLEnergyF_MT (EnergyFunctional.cpp:397-415) + calcLEnergyPt (:332-392); calcMEnergyF (:320-329)
static int calc_l_energy(nalo_ctx* c, double* E) {
    BAWindow& w = *c->ba;
    double e = 0;
    for (const auto& f : w.frames) for (int i = 0; i < 8; ++i) e += f.delta_prior[i] * f.prior[i] * f.delta_prior[i];
    { float ec = 0; for (int i = 0; i < 4; ++i) ec += (w.cDeltaF[i] * (float)kInitialCalibHessian) * w.cDeltaF[i]; e += ec; }       // cDeltaF.cwiseProduct(cPriorF).dot(cDeltaF): floats
    const int nb = (w.Ppad + 255) / 256;
    NALO_HIP(c, w.noapply_E.reserve((size_t)std::max<size_t>(nb, (size_t)w.nblocks * w.dev.lin_sub * w.W)));
    ba_launch_lenergy(c->stream, w.dev, w.noapply_E.p);
    std::vector<double> part(nb);
    NALO_HIP(c, hipMemcpyAsync(part.data(), w.noapply_E.p, nb * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    NALO_HIP(c, hipStreamSynchronize(c->stream));
    double ep = 0; for (double v : part) ep += v;
    { int rs = sum_over_ranks(c, &ep, 1); if (rs) return rs; }         // the frame / calibration part above is the same on every rank (frames replicate)
    *E = e + (double)(float)ep;                                       // E.finish(): Accumulator11 holds a float
    return NALO_OK;
}
static double calc_m_energy(const BAWindow& w) {
    const int n = w.n;
    std::vector<double> d(n);
    for (int i = 0; i < 4; ++i) d[i] = (double)w.cDeltaF[i];
    for (int h = 0; h < w.W; ++h) for (int i = 0; i < 8; ++i) d[4 + 8 * h + i] = w.frames[h].delta[i];
    double E = 0;
    for (int i = 0; i < n; ++i) { double s = 0; for (int j = 0; j < n; ++j) s += w.HM[(size_t)i * n + j] * d[j]; E += d[i] * (2 * w.bM[i] + s); }
    return E;
}
// FullSystem::loadSateBackup (FullSystemOptimize.cpp:352-369)
static int load_state_backup(nalo_ctx* c) {
    BAWindow& w = *c->ba;
    calib_set_value(w, w.c_value_backup);
    for (auto& f : w.frames) frame_set_state(f, f.state_backup);
    ba_launch_load_backup(c->stream, w.dev);
    NALO_HIP(c, hipGetLastError());
    return set_precalc(c);
}
static int sc_async(nalo_ctx* c, int shift, float margScale, int margOnly) {
    BAWindow& w = *c->ba;
    ProfScope ps(c, "ba_sc");
    ba_launch_sc(c->stream, w.dev, w.T, shift, margScale, margOnly);
    w.have_sc = true; w.sc_shift = shift; w.stitched_sc = false;
    NALO_HIP(c, hipGetLastError());
    return NALO_OK;
}
// optimize() on a small single-GPU window: with the systems of iteration k on their way to the host, the two kernels that will follow the host's solve - the
// back-substitution + point step and the linearisation at the stepped states - are enqueued NOW, each behind a gate in host-mapped memory (ba_device.h: GateBlock).
// The host then solves, writes {xc, xAd} and opens the first gate (solve_system), steps its frame states, writes the precalc records and opens the second
// (set_precalc): the device picks both up ~1.5 us after the stores instead of 5-6 us after a launch call - the two launch latencies that sat on the critical
// path of every Gauss-Newton iteration (FullSystemOptimize.cpp:478-545 is one serial chain of such steps).
static bool prelaunch_eligible(const nalo_ctx* c, const BAWindow& w) {
    return !w.hook && c->set.forceAcceptStep && w.points_set && w.Ppad <= 32768 && w.W <= 8;
}
static int prelaunch_iteration(nalo_ctx* c) {
    BAWindow& w = *c->ba;
    if (!w.gate_h) {
        NALO_HIP(c, hipHostMalloc((void**)&w.gate_h, sizeof(GateBlock), hipHostMallocMapped | hipHostMallocCoherent));
        std::memset(w.gate_h, 0, sizeof(GateBlock));
        NALO_HIP(c, hipHostGetDevicePointer((void**)&w.gate_d, w.gate_h, 0));
    }
    if (w.pre_pos + 1 - 4 > w.pre_synced || !w.pre_map_dev[0]) return NALO_OK;    // the ring slot of the next records may still have readers: this iteration launches the ordinary way
    // a cancelled pass (prelaunch_cancel) leaves kGateCancel in both words and nothing waiting on them (it drained the stream): a kernel enqueued now must find
    // them merely closed
    if (w.gate_h->x_seq == kGateCancel) __atomic_store_n(&w.gate_h->x_seq, 0u, __ATOMIC_RELEASE);
    if (w.gate_h->p_seq == kGateCancel) __atomic_store_n(&w.gate_h->p_seq, 0u, __ATOMIC_RELEASE);
    if (++w.gate_x_seq == kGateCancel) w.gate_x_seq = 1;
    if (++w.gate_p_seq == kGateCancel) w.gate_p_seq = 1;
    {
        ProfScope ps(c, "ba_resub");
        NALO_HIP(c, w.step_partial.reserve((size_t)(w.Ppad / 256 + 1) * 4));
        const GateArg g{&w.gate_d->x_seq, &w.gate_d->err, w.gate_d->x, w.gate_x_seq};
        ba_launch_resub_step_gated(c->stream, w.dev, 1.f, w.step_partial.p, g);
    }
    w.resub_pre = true;
    {
        BADev D = w.dev;                                       // the records this pass will read: the ring slot set_precalc takes next
        const size_t nfl = (size_t)w.W * w.W * kPreStride;
        D.pre = w.pre_map_dev[(w.pre_pos + 1) & 3]; D.calib = D.pre + nfl;
        D.gate_p = &w.gate_d->p_seq; D.gate_err = &w.gate_d->err; D.gate_p_want = w.gate_p_seq;
        ProfScope ps(c, "ba_linearize", true);
        ba_launch_linearize(c->stream, D, 0, 0, ps.a, ps.b);
    }
    w.lin_pre = true;
    NALO_HIP(c, hipGetLastError());
    return NALO_OK;
}
// error paths: no kernel may be left spinning on a gate nobody will open
static void prelaunch_cancel(nalo_ctx* c) {
    BAWindow& w = *c->ba;
    if (!(w.resub_pre || w.lin_pre) || !w.gate_h) return;
    __atomic_store_n(&w.gate_h->x_seq, kGateCancel, __ATOMIC_RELEASE); __atomic_store_n(&w.gate_h->p_seq, kGateCancel, __ATOMIC_RELEASE);
    (void)hipStreamSynchronize(c->stream);
    w.resub_pre = w.lin_pre = false; w.want_prelaunch = false;
    w.have_lin = false; w.have_sc = false;
}

// fp64 finish + stitch of whichever system is not stitched yet; then (optionally) the cross-rank sum and the D2H copy
// misc_only: the caller wants the per-bin {count, energy} and the threshold, not the systems: finish the partials, skip the stitch, publish the tail
static int stitch_and_fetch(nalo_ctx* c, bool want_top, bool want_sc, bool th_to_host = false, bool misc_only = false) {
    BAWindow& w = *c->ba;
    if (w.ad_pending && !misc_only) { int ra = upload_adjoints(c); if (ra) return ra; }   // the stitch (and a large window's back-substitution behind it) reads the device copy of the adjoints
    const int n1 = w.n1, NPL = w.NPL, W = w.W;
    const size_t blk = (size_t)n1 * n1;
    bool did = false;
    const int npub = (int)(2 * blk + 2 * W * W + 5);                 // [H~_A | H~_sc | misc (2 W^2) | step sums (3) | TH sum, rank count]
    const double seq = (double)(w.pub_seq + 1);
    double* dmap = nullptr;
    NALO_HIP(c, hipHostGetDevicePointer((void**)&dmap, w.stitched_host, 0));
    const bool top = want_top && !w.stitched_top, sc = want_sc && !w.stitched_sc;
    const bool sc_was_summed = w.stitched_sc;                        // a hooked window: its block already holds the sum over the ranks
    // this stitch's all-reduce takes level C's histogram of the newest frame's threshold along - when the summed range reaches the tail it sits behind
    const bool fuse_lo = w.hook && w.th_lo_pending && (top || misc_only);
    bool th_after_publish = false;
    {
        // small single-GPU windows: a pending threshold (setNewFrameEnergyTH of the last linearisation) is computed by one more workgroup of the reduce launch, and a
        // misc-only fetch publishes its tail from the reduce launch too: the last pass of optimize() is [linearize | reduce] instead of [linearize | th | tail | reduce | publish]
        // (only where the host waits for the threshold anyway: behind a regular iteration's publish the separate 7 us launch runs under the host's solve, inside the reduce
        // launch it would sit on the critical path in front of the stitch)
        const bool small_th = !w.hook && w.th_pending && th_to_host && w.dev.Ppad <= 16384 && (top || sc);
        const bool pub_in_reduce = misc_only && !w.hook && (top || sc) && (small_th || !w.th_pending);
        if ((top || sc) && (th_to_host || w.hook) && !fuse_lo && !pub_in_reduce) {      // the threshold rides in the tail {TH, 1.0}: compute it before the publish
            if (!small_th) { int rc = flush_th(c); if (rc) return rc; }
            if (!small_th) ba_launch_th_tail(c->stream, w.frameTH.p + (W - 1), w.stitched.p + 2 * blk + 2 * W * W + 3);
        }
        ProfScope ps(c, "ba_reduce");                                 // reduce + stitch only: the threshold search above is its own chain (and holds collectives)
        if (top || sc) {
            // misc {count, energy} per bin lands in the tail of the stitched buffer; without a cross-rank hook step B publishes
            // rows + tail + sequence number straight into host-mapped memory
            ba_launch_reduce(c->stream, w.dev, w.host_blk.p, NPL, w.acc13.p, w.stitched.p + 2 * blk, w.G.p, top, sc,
                             w.step_sums_deferred ? w.step_partial.p : nullptr, (w.Ppad + 255) / 256, w.stitched.p + 2 * blk + 2 * W * W, small_th,
                             pub_in_reduce ? dmap + 2 * blk : nullptr, seq, w.st_ticket.p + 2);
            if (small_th) w.th_pending = false;
            w.step_sums_deferred = false;
            if (misc_only) { if (!w.hook && !pub_in_reduce) ba_launch_publish(c->stream, w.stitched.p + 2 * blk, dmap + 2 * blk, npub - (int)(2 * blk), seq, w.st_ticket.p + 1); }
            else {
                if (small_th && th_to_host) ba_launch_th_tail(c->stream, w.frameTH.p + (W - 1), w.stitched.p + 2 * blk + 2 * W * W + 3);   // the stitch publishes the tail: {TH, 1.0} must be in it
                if (ba_launch_stitch(c->stream, w.sd, top, sc, w.hook ? nullptr : dmap, npub - (int)(2 * blk), seq)) return fail(c, NALO_ERR_HIP, "ba_stitch_kernel: LDS size rejected");
                if (top) w.stitched_top = true;
                if (sc) w.stitched_sc = true;
            }
            did = true;
        }
    }
    NALO_HIP(c, hipGetLastError());
    if (did) {
        ++w.pub_seq;
        if (w.hook) {
            // sharded window: tail = {step sums (3), frameEnergyTH of the newest frame, 1.0}. Every rank already holds the SAME threshold (the order
            // statistic over all ranks' residuals, linearize_async), so sum / count below re-installs that value; the tail keeps its layout.
            const size_t off = misc_only ? 2 * blk : 0;               // misc_only: only the tail is summed and published
            // What is summed over the ranks is what THIS call produced, once: the freshly stitched block(s) and - with the top system, whose reduce writes it - the
            // tail {per-bin count / energy, step sums, threshold pair}. A second fetch of the same linearisation (nalo_ba_linearize, then nalo_ba_accumulate_sc or
            // nalo_ba_solve_system: the step-by-step mapping of ef->solveSystemF) stitches the Schur complement alone and sums ITS block alone:
            // H_A, b_A and the tail already hold the window's totals (rounds 2-3 summed them again: world x H_A).
            const size_t tail_end = fuse_lo ? w.lo_off + kThDblC : (size_t)npub;
            size_t ra[2], rb[2]; int nr = 0;
            if (misc_only) { ra[nr] = 2 * blk; rb[nr++] = tail_end; }
            else if (top && (sc || !sc_was_summed)) { ra[nr] = 0; rb[nr++] = tail_end; }     // the SC block in between is fresh, or dead (re-stitched before anyone reads it)
            else if (top) { ra[nr] = 0; rb[nr++] = blk; ra[nr] = 2 * blk; rb[nr++] = tail_end; }
            else { ra[nr] = blk; rb[nr++] = 2 * blk; }
            if (fuse_lo) {
                // [systems | tail | level C]: one sum over the ranks, then the search on the summed histogram gives every rank the same threshold
                if (w.th_side_inflight) { NALO_HIP(c, hipStreamWaitEvent(c->stream, w.ev_th, 0)); w.th_side_inflight = false; }
                w.th_lo_pending = false;
            }
            if (!w.hook_stream_ordered) NALO_HIP(c, hipStreamSynchronize(c->stream));
            for (int i = 0; i < nr; ++i) { int rh = call_hook(c, w.hook, w.hook_user, w.stitched.p + ra[i], (int)(rb[i] - ra[i])); if (rh) return rh; }
            if (fuse_lo) {
                if (th_to_host) {                                    // the caller reads the threshold from the published tail: search first
                    ba_launch_energy_th_step(c->stream, w.dev, 3);
                    ba_launch_th_tail(c->stream, w.frameTH.p + (W - 1), w.stitched.p + 2 * blk + 2 * W * W + 3);    // what tail_th() reads: {TH, 1.0}
                } else th_after_publish = true;                      // optimize(): the host only waits for the systems; the search runs under its solve
            } else if (top || misc_only) ba_launch_th_install(c->stream, w.stitched.p + 2 * blk + 2 * W * W + 3, w.frameTH.p + (W - 1));   // the tail's {sum of TH, ranks}: the common threshold
            ba_launch_publish(c->stream, w.stitched.p + off, dmap + off, npub - (int)off, seq, w.st_ticket.p + 1);
            if (th_after_publish) ba_launch_energy_th_step(c->stream, w.dev, 3);
            NALO_HIP(c, hipGetLastError());
        }
        { int rc = flush_th(c); if (rc) return rc; }                  // behind the publish: overlaps the host's solve
        if (w.want_prelaunch) { w.want_prelaunch = false; int rc = prelaunch_iteration(c); if (rc) return rc; }
        if (!poll_flag(c, &w.stitched_host[npub], seq)) return NALO_ERR_HIP;
        if (w.gate_h && w.gate_h->err) return fail(c, NALO_ERR_HIP, "a pre-enqueued kernel gave up waiting for its inputs (gate timeout)");
        w.pre_synced = w.pre_pos;                               // everything launched on the main stream so far has run (the side stream reads no precalc record)
        if (w.step_pending) {                                   // finish doStepFromBackup's break test with the sums of the last step
            const double* s3 = w.stitched_host + 2 * blk + 2 * W * W;
            const float numID = (float)s3[2];
            const float sumNID = numID > 0 ? (float)(s3[1] / numID) : 0.f;
            const float th = kThOptIterations;
            w.last_canbreak = std::sqrt(w.st_sumA) < 0.0005 * th && std::sqrt(w.st_sumB) < 0.00005 * th && std::sqrt(w.st_sumR) < 0.00005 * th &&
                              std::sqrt(w.st_sumT) * sumNID < 0.00005 * th;
            w.step_pending = false;
        }
        return NALO_OK;
    }
    NALO_HIP(c, hipStreamSynchronize(c->stream));
    return NALO_OK;
}
static int stitch_and_fetch_for_break(nalo_ctx* c) {
    BAWindow& w = *c->ba;
    if (!w.have_sc || w.sc_shift != 1) { int rc = sc_async(c, 1, 1.f, 0); if (rc) return rc; }
    return stitch_and_fetch(c, true, true);
}
// frameEnergyTH of the newest frame as published in the tail {sum over ranks, rank count} (stitch_and_fetch with th_to_host)
static float tail_th(const BAWindow& w) {
    const double* tl = w.stitched_host + 2 * (size_t)w.n1 * w.n1 + 2 * w.W * w.W + 3;
    return (float)(tl[0] / tl[1]);
}
// a threshold search that lost a workgroup at its arrival counter reports NaN from then on (ba_th_find_kernel, sticky): fail instead of installing it
static int check_th(nalo_ctx* c, float th) {
    if (std::isfinite(th)) return NALO_OK;
    return fail(c, NALO_ERR_HIP, "setNewFrameEnergyTH: the device search timed out (a workgroup never arrived); re-issue the window with nalo_ba_set_window");
}
static void unpack_system(const BAWindow& w, const double* Ht, double* H, double* b) {
    const int n = w.n, n1 = w.n1;
    for (int r = 0; r < n; ++r) { for (int cc = 0; cc < n; ++cc) H[(size_t)r * n + cc] = Ht[(size_t)r * n1 + cc]; b[r] = Ht[(size_t)r * n1 + n]; }
}
static void misc_totals(const BAWindow& w, double* energy, int* nres) {
    const double* m = w.stitched_host + 2 * (size_t)w.n1 * w.n1;
    double e = 0, cnt = 0;
    for (int i = 0; i < w.W * w.W; ++i) { cnt += m[2 * i]; e += m[2 * i + 1]; }
    if (energy) *energy = e;
    if (nres) *nres = (int)(cnt + 0.5);
}

// orthogonalize(x) (EnergyFunctional.cpp:719-773) with the nullspaces of FullSystem::getNullspaces (FullSystemOptimize.cpp:658-712):
// x -= U U^T x, U = left singular vectors of the normalised [pose(6) | scale] nullspace matrix above delta * sigma_max.
static void build_projector(BAWindow& w) {
    const int W = w.W, n = w.n, m = 7;
    std::vector<double> N((size_t)n * m, 0.0);
    for (int i = 0; i < 6; ++i) for (int f = 0; f < W; ++f) for (int r = 0; r < 6; ++r)
        N[(size_t)(4 + f * 8 + r) * m + i] = w.frames[f].ns_pose[i][r] * (r < 3 ? (1.0f / kScaleXiTrans) : (1.0f / kScaleXiRot));
    for (int f = 0; f < W; ++f) for (int r = 0; r < 6; ++r) N[(size_t)(4 + f * 8 + r) * m + 6] = w.frames[f].ns_scale[r] * (r < 3 ? (1.0f / kScaleXiTrans) : (1.0f / kScaleXiRot));
    for (int cc = 0; cc < m; ++cc) { double s = 0; for (int r = 0; r < n; ++r) s += N[(size_t)r * m + cc] * N[(size_t)r * m + cc]; s = std::sqrt(s); if (s > 0) for (int r = 0; r < n; ++r) N[(size_t)r * m + cc] /= s; }
    double G[49], V[49], wv[7];
    for (int a = 0; a < m; ++a) for (int cc = 0; cc < m; ++cc) { double s = 0; for (int r = 0; r < n; ++r) s += N[(size_t)r * m + a] * N[(size_t)r * m + cc]; G[a * m + cc] = s; }
    sym_eig(m, G, V, wv);
    double maxSv = 0;
    for (int i = 0; i < m; ++i) maxSv = std::max(maxSv, wv[i] > 0 ? std::sqrt(wv[i]) : 0.0);
    w.Sproj.assign((size_t)n * m, 0.0);                 // columns u_i (zero column if dropped)
    for (int i = 0; i < m; ++i) {
        const double sv = wv[i] > 0 ? std::sqrt(wv[i]) : 0.0;
        if (!(sv > kSolverModeDelta * maxSv)) continue;
        for (int r = 0; r < n; ++r) { double s = 0; for (int cc = 0; cc < m; ++cc) s += N[(size_t)r * m + cc] * V[cc * m + i]; w.Sproj[(size_t)r * m + i] = s / sv; }
    }
    w.proj_valid = true;
}

static int do_accumulate_top(nalo_ctx* c) {
    BAWindow& w = *c->ba;
    if (!w.have_lin) return fail(c, NALO_ERR_STATE, "accumulate before linearize");
    return stitch_and_fetch(c, true, false);
}
static int do_accumulate_sc(nalo_ctx* c, int shift) {
    BAWindow& w = *c->ba;
    if (!w.have_lin) return fail(c, NALO_ERR_STATE, "accumulate_sc before linearize");
    if (!w.have_sc || w.sc_shift != shift) { int rc = sc_async(c, shift, 1.f, 0); if (rc) return rc; }
    return stitch_and_fetch(c, false, true);
}
static void prior_system(const BAWindow& w, double* H, double* b) {         // accumulateLF with usePrior (AccumulatedTopHessian.cpp:292-302)
    const int n = w.n;
    std::fill(H, H + (size_t)n * n, 0.0); std::fill(b, b + n, 0.0);
    for (int i = 0; i < 4; ++i) { H[(size_t)i * n + i] += kInitialCalibHessian; b[i] += kInitialCalibHessian * (double)w.cDeltaF[i]; }
    for (int h = 0; h < w.W; ++h) for (int i = 0; i < 8; ++i) { const int d = 4 + h * 8 + i; H[(size_t)d * n + d] += w.frames[h].prior[i]; b[d] += w.frames[h].prior[i] * w.frames[h].delta_prior[i]; }
}

static int solve_system(nalo_ctx* c, int iteration, double lambda, double* x_out, bool fuse_step = false) {
    BAWindow& w = *c->ba;
    HostTimer ht(c, "ba.solve_system");
    const int W = w.W, n = w.n, n1 = w.n1;
    (void)lambda; lambda = 1e-5;                                            // SOLVER_FIX_LAMBDA (EnergyFunctional.cpp:779)
    if (!w.have_lin) return fail(c, NALO_ERR_STATE, "solve_system before linearize");
    if (!w.have_sc || w.sc_shift != 1) { int rc = sc_async(c, 1, 1.f, 0); if (rc) return rc; }
    // What the assembly needs from the HOST side only - delta, bL + (bM + HM delta), HL + diag(HM) - is computed here, while the device still works on the systems
    // (the products HM delta are n chains of n dependent multiply-adds: 5 of the 8 us the assembly took at n = 68 when it ran behind the wait, round 3)
    const int lda = (n + 7) & ~7;
    w.solve_scratch.resize((size_t)lda * lda + 10 * (size_t)lda + n); w.solve_perm.resize(n);
    double* HF = w.solve_scratch.data(); double* bF = HF + (size_t)lda * lda; double* sv = bF + lda; double* delta = sv + lda; double* yv = delta + lda;   // yv: 3 lda
    double* rhs0 = yv + 3 * (size_t)lda; double* dg0 = rhs0 + lda;
    float* xF = reinterpret_cast<float*>(dg0 + lda);
    {
        HostTimer hp(c, "ba.solve.host_pre");
        for (int i = 0; i < 4; ++i) delta[i] = (double)w.cDeltaF[i];
        for (int h = 0; h < W; ++h) for (int i = 0; i < 8; ++i) delta[4 + 8 * h + i] = w.frames[h].delta[i];
        for (int r = 0; r < n; ++r) {                                       // accumulateLF with usePrior (AccumulatedTopHessian.cpp:292-302): diagonal only
            double HLd, bLr;
            if (r < 4) { HLd = kInitialCalibHessian; bLr = kInitialCalibHessian * (double)w.cDeltaF[r]; }
            else { const HostFrame& f = w.frames[(r - 4) >> 3]; const int i = (r - 4) & 7; HLd = f.prior[i]; bLr = f.prior[i] * f.delta_prior[i]; }
            const double* hm = &w.HM[(size_t)r * n];
            double sdot = 0;
            for (int cc = 0; cc < n; ++cc) sdot += hm[cc] * delta[cc];
            dg0[r] = HLd + hm[r];
            rhs0[r] = bLr + (w.bM[r] + sdot);
        }
    }
    int rc;
    w.want_prelaunch = fuse_step && W <= 8 && prelaunch_eligible(c, w);
    { HostTimer h2(c, "ba.solve.fetch_wait"); rc = stitch_and_fetch(c, true, true); }
    w.want_prelaunch = false;
    if (rc) return rc;
    {   // tests: the error path BETWEEN a pre-launch and its gates (tests/test_ba_gpu.py, child process): the second solve of the process fails here, once
        static const bool test_cancel = std::getenv("NALO_BA_TEST_GATE_CANCEL") != nullptr;
        static int solves = 0;
        if (test_cancel && w.resub_pre && ++solves == 2) return fail(c, NALO_ERR_HIP, "test: failure between a pre-launch and its gates");
    }
    HostTimer h3(c, "ba.solve.host_math");
    // H = (HL + HM + HA) with the diagonal * (1+lambda), minus Hsc/(1+lambda); b = bL + (bM + HM delta) + bA - bsc   (:795-868), then the Jacobi scaling
    // (:872-885): TWO passes over the published systems (the diagonal first: the scaling needs it), written straight into the padded, scaled matrix the
    // factorisation works on. Eigen's LDLT (the reference, :880) reads the LOWER triangle only; H_sc carries fp32-rounding asymmetry (w*a_j*a_k vs
    // w*a_k*a_j), so the upper triangle the factorisation reads is filled from the lower one: entry (r, c >= r) is computed from the sources' (c, r).
    // Scratch lives in the window: no allocation per iteration.
    std::vector<double>& x = w.lastX; x.resize(n);
    const double* HAp = w.stitched_host; const double* HSp = w.stitched_host + (size_t)n1 * n1;
    misc_totals(w, nullptr, &w.resInA);
    const double fsc = 1.0 / (1 + lambda);
    { HostTimer hl(c, "ba.solve.math.assemble");
    for (int r = 0; r < n; ++r) {                                           // diagonal + right-hand side (the host-only terms were prepared above, in the same operation order)
        const double* ha = HAp + (size_t)r * n1; const double* hs = HSp + (size_t)r * n1;
        double dg = dg0[r] + ha[r];
        dg *= (1 + lambda);
        dg -= hs[r] * fsc;
        sv[r] = 1.0 / std::sqrt(dg + 10);
        HF[(size_t)r * lda + r] = dg;
        bF[r] = (rhs0[r] + ha[n] - hs[n]) * sv[r];
    }
    for (int r = 0; r < n; ++r) {                                           // lower-triangle entries (r, cc < r), stored mirrored at (cc, r) and scaled
        const double* hm = &w.HM[(size_t)r * n]; const double* ha = HAp + (size_t)r * n1; const double* hs = HSp + (size_t)r * n1;
        const double si = sv[r];
        for (int cc = 0; cc < r; ++cc) HF[(size_t)cc * lda + r] = si * (((0.0 + hm[cc]) + ha[cc]) - hs[cc] * fsc) * sv[cc];
        HF[(size_t)r * lda + r] = si * HF[(size_t)r * lda + r] * si;
        for (int cc = n; cc < lda; ++cc) HF[(size_t)r * lda + cc] = 0.0;   // the padding columns the blocked factorisation sweeps over
    }
    }
    { HostTimer hl(c, "ba.solve.math.ldlt");
    ldlt_solve_blocked(n, lda, HF, bF, x.data(), yv, w.solve_perm.data());
    for (int i = 0; i < n; ++i) x[i] *= sv[i];
    }
    if (iteration >= 2) {                                                   // SOLVER_ORTHOGONALIZE_X_LATER (:898-902)
        if (!w.proj_valid) build_projector(w);
        double coef[7];
        for (int k = 0; k < 7; ++k) { double s = 0; for (int r = 0; r < n; ++r) s += w.Sproj[(size_t)r * 7 + k] * x[r]; coef[k] = s; }
        for (int r = 0; r < n; ++r) { double s = 0; for (int k = 0; k < 7; ++k) s += w.Sproj[(size_t)r * 7 + k] * coef[k]; x[r] -= s; }
    }
    if (x_out) std::memcpy(x_out, x.data(), n * 8);
    // resubstituteF_MT (:263-289)
    for (int i = 0; i < 4; ++i) w.c_step[i] = -x[i];
    float* xAd = w.up_host + (size_t)W * W * kPreStride + 16 + 64;
    float* xc = w.up_host + (size_t)W * W * kPreStride + 16;
    for (int i = 0; i < n; ++i) xF[i] = (float)x[i];
    for (int i = 0; i < 4; ++i) xc[i] = xF[i];
    for (int h = 0; h < W; ++h) {
        for (int i = 0; i < 8; ++i) w.frames[h].step[i] = -x[4 + 8 * h + i];
        w.frames[h].step[8] = w.frames[h].step[9] = 0;
        if (W > 8) continue;                                   // larger windows: every back-substitution workgroup builds its host's rows of xAd from x (ba_resub_kernel, XMODE 2)
        for (int t = 0; t < W; ++t) {
            const float *AH = &w.adHostF[(size_t)(h + W * t) * 64], *AT = &w.adTargetF[(size_t)(h + W * t) * 64];
            // xAd = x_h^T adHost + x_t^T adTarget: rows of the adjoints are contiguous in j, so j is the inner (vector) loop; per entry the same
            // i-ordered sums as the scalar form
            float s1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int i = 0; i < 8; ++i) { const float xh = xF[4 + 8 * h + i], xt = xF[4 + 8 * t + i]; for (int j = 0; j < 8; ++j) { s1[j] += xh * AH[i * 8 + j]; s2[j] += xt * AT[i * 8 + j]; } }
            for (int j = 0; j < 8; ++j) xAd[(size_t)(W * h + t) * 8 + j] = s1[j] + s2[j];
        }
    }
    if (w.resub_pre) {                                          // the kernel is already in the stream, spinning on its gate: hand it {xc, xAd} and open
        w.resub_pre = false;
        std::memcpy(w.gate_h->x, xc, 16); std::memcpy(w.gate_h->x + 4, xAd, (size_t)W * W * 8 * 4);
        __atomic_thread_fence(__ATOMIC_RELEASE);
        __atomic_store_n(&w.gate_h->x_seq, w.gate_x_seq, __ATOMIC_RELEASE);
        w.step_fused = true;
        return NALO_OK;
    }
    XadArg karg;
    if (W <= 8) { std::memcpy(karg.v, xc, 16); std::memcpy(karg.v + 4, xAd, (size_t)W * W * 8 * 4); }      // small window: {xc, xAd} travel as kernel arguments, no copy
    else std::memcpy(karg.v, xF, (size_t)n * 4);                 // larger windows: x itself (8W + 4 floats); xAd is built on the device from the float adjoints
    {
        ProfScope ps(c, "ba_resub");
        if (fuse_step) {
            NALO_HIP(c, w.step_partial.reserve((size_t)(w.Ppad / 256 + 1) * 4));
            ba_launch_resub_step(c->stream, w.dev, 1.f, w.step_partial.p, karg, W > 8);
            w.step_fused = true;
        } else ba_launch_resub(c->stream, w.dev, karg, W > 8);
    }
    NALO_HIP(c, hipGetLastError());
    return NALO_OK;
}

static void backup_state(BAWindow& w) {
    std::memcpy(w.c_value_backup, w.c_value, sizeof(w.c_value));
    for (auto& f : w.frames) std::memcpy(f.state_backup, f.state, sizeof(f.state));
}
static int do_step(nalo_ctx* c, float fC, float fT, float fR, float fA, float fD, int* canbreak) {
    BAWindow& w = *c->ba;
    HostTimer ht(c, "ba.do_step");
    const double pf[10] = {fT, fT, fT, fR, fR, fR, fA, fA, fA, fA};
    float sumA = 0, sumB = 0, sumT = 0, sumR = 0;
    double v[4];
    for (int i = 0; i < 4; ++i) v[i] = w.c_value_backup[i] + fC * w.c_step[i];
    calib_set_value(w, v);
    for (auto& fh : w.frames) {
        double st[10];
        for (int i = 0; i < 10; ++i) st[i] = fh.state_backup[i] + pf[i] * fh.step[i];
        frame_set_state(fh, st);
        sumA += fh.step[6] * fh.step[6]; sumB += fh.step[7] * fh.step[7];
        sumT += fh.step[0] * fh.step[0] + fh.step[1] * fh.step[1] + fh.step[2] * fh.step[2];
        sumR += fh.step[3] * fh.step[3] + fh.step[4] * fh.step[4] + fh.step[5] * fh.step[5];
    }
    NALO_HIP(c, w.step_partial.reserve((size_t)(w.Ppad / 256 + 1) * 4));
    double* out3 = w.stitched.p + 2 * (size_t)w.n1 * w.n1 + 2 * w.W * w.W;      // scratch tail of the stitched buffer
    if (w.step_fused) {                                                        // points already stepped by solve_system (fD = 1)
        w.step_fused = false;
        if (canbreak) ba_launch_step_sums(c->stream, w.dev, w.step_partial.p, out3);     // the caller wants the break test now (round 4: it read the sums of the step before)
        else w.step_sums_deferred = true;                                      // optimize(): the sums follow with the next reduce launch
    } else ba_launch_step(c->stream, w.dev, fD, w.step_partial.p, out3);
    int rc = set_precalc(c);
    if (rc) return rc;
    sumA /= w.W; sumB /= w.W; sumR /= w.W; sumT /= w.W;
    w.st_sumA = sumA; w.st_sumB = sumB; w.st_sumT = sumT; w.st_sumR = sumR;
    w.have_lin = false; w.have_sc = false;
    if (canbreak) {                                                         // the API call wants the answer now: fetch the point sums
        double s3[3];
        NALO_HIP(c, hipMemcpyAsync(s3, out3, 24, hipMemcpyDeviceToHost, c->stream));
        NALO_HIP(c, hipStreamSynchronize(c->stream));
        { int rs = sum_over_ranks(c, s3, 3); if (rs) return rs; }
        const float numID = (float)s3[2];
        const float sumNID = numID > 0 ? (float)(s3[1] / numID) : 0.f;
        const float th = kThOptIterations;
        *canbreak = std::sqrt(sumA) < 0.0005 * th && std::sqrt(sumB) < 0.00005 * th && std::sqrt(sumR) < 0.00005 * th && std::sqrt(sumT) * sumNID < 0.00005 * th;
    } else w.step_pending = true;                                           // optimize(): the sums ride along with the next fetch (no extra sync)
    return NALO_OK;
}

}  // namespace nalo

using namespace nalo;

extern "C" {

int nalo_ba_set_window(nalo_ctx* c, int W, const nalo_frame_state* frames, const double calib[4], const double calib_zero[4]) {
    if (!c || !frames || !calib || W < 2 || W > NALO_MAX_WINDOW) return fail(c, NALO_ERR_ARG, "nalo_ba_set_window: bad argument");
    NALO_HIP(c, hipSetDevice(c->device));
    if (!c->ba) c->ba = new BAWindow();
    BAWindow& w = *c->ba;
    if (w.W != W) { w.points_set = false; w.res_set = false; }
    w.W = W; w.n = 8 * W + 4; w.n1 = w.n + 1;
    w.T = (8 * (W - 1) + 5 + 15) / 16; w.NPL = 16 * w.T;
    // CalibHessian(): setValueScaled(K) then value_zero (HessianBlocks.h:347-361, 381-395)
    double v[4], vz[4];
    const double* cz = calib_zero ? calib_zero : calib;
    v[0] = (1.0f / kScaleF) * calib[0]; v[1] = (1.0f / kScaleF) * calib[1]; v[2] = (1.0f / kScaleC) * calib[2]; v[3] = (1.0f / kScaleC) * calib[3];
    vz[0] = (1.0f / kScaleF) * cz[0]; vz[1] = (1.0f / kScaleF) * cz[1]; vz[2] = (1.0f / kScaleC) * cz[2]; vz[3] = (1.0f / kScaleC) * cz[3];
    calib_set_value(w, v);
    for (int i = 0; i < 4; ++i) { w.c_value_scaled[i] = calib[i]; w.c_scaledf[i] = (float)calib[i]; w.c_value_zero[i] = vz[i]; }
    w.c_scaledi[0] = 1.0f / w.c_scaledf[0]; w.c_scaledi[1] = 1.0f / w.c_scaledf[1];
    w.c_scaledi[2] = -w.c_scaledf[2] / w.c_scaledf[0]; w.c_scaledi[3] = -w.c_scaledf[3] / w.c_scaledf[1];
    w.frames.assign(W, HostFrame());
    for (int i = 0; i < W; ++i) {
        HostFrame& f = w.frames[i];
        const nalo_frame_state& s = frames[i];
        if (s.slot < 0 || s.slot >= (int)c->slots.size() || !c->slots[s.slot].valid) return fail(c, NALO_ERR_STATE, "nalo_ba_set_window: frame slot has no pyramid");
        f.slot = s.slot; f.frameID = s.frame_id; f.evalPT = SE3::from(s.worldToCam_evalPT);
        f.ab_exposure = s.ab_exposure; f.frameEnergyTH = s.frameEnergyTH;
        double z6[10]; std::memcpy(z6, s.state_zero, sizeof(z6));
        frame_set_state(f, s.state_zero);
        frame_set_state_zero(f, z6);
        frame_set_state(f, s.state);
        frame_take_data(c->set, f);
        w.dev.img[i] = c->slots[s.slot].dI[0];
    }
    w.dev.W = W; w.dev.w = c->w; w.dev.h = c->h;
    w.dev.fix_a = c->set.affineOptModeA < 0; w.dev.fix_b = c->set.affineOptModeB < 0; w.dev.no_th = 0;
    NALO_HIP(c, w.th_hist.reserve(64)); NALO_HIP(c, hipMemset(w.th_hist.p, 0, 64 * 4));
    NALO_HIP(c, w.th_buf.reserve(2 * kThDblAB)); NALO_HIP(c, hipMemset(w.th_buf.p, 0, 2 * kThDblAB * 8));      // zero = ready; ba_th_final_kernel leaves them zeroed again
    w.dev.th_state = w.th_hist.p; w.dev.th_bufAB = w.th_buf.p;
    // HM / bM survive a nalo_ba_set_window only when that is asked for: right after nalo_ba_marginalize_frame (whose result is meant for this very call) or on a
    // context declared continuing (nalo_ba_set_prior_carry). Any other window starts from a zero prior, whatever an earlier, unrelated window left behind.
    const bool keep_prior = w.prior_next || w.prior_carry;
    w.prior_next = false;
    if (!keep_prior) { w.HM.assign((size_t)w.n * w.n, 0.0); w.bM.assign(w.n, 0.0); }
    else if (w.HM.size() == (size_t)(w.n - 8) * (w.n - 8) && w.n > 12) {
        // one frame appended to a window that carries a prior = EnergyFunctional::insertFrame (EnergyFunctional.cpp:437-442): conservativeResize, the new
        // frame's rows / columns zero
        const int no = w.n - 8;
        std::vector<double> H2((size_t)w.n * w.n, 0.0), b2(w.n, 0.0);
        for (int r = 0; r < no; ++r) { std::memcpy(&H2[(size_t)r * w.n], &w.HM[(size_t)r * no], (size_t)no * 8); b2[r] = w.bM[r]; }
        w.HM.swap(H2); w.bM.swap(b2);
    } else if (w.HM.size() != (size_t)w.n * w.n) { w.HM.assign((size_t)w.n * w.n, 0.0); w.bM.assign(w.n, 0.0); }
    w.lastX.assign(w.n, 0.0);
    const size_t blk = (size_t)w.n1 * w.n1;
    NALO_HIP(c, w.acc13.reserve((size_t)W * W * 169));
    NALO_HIP(c, w.G.reserve((size_t)W * w.NPL * w.NPL));
    w.lo_off = (2 * blk + 2 * W * W + 5 + 15) & ~(size_t)15;                       // behind the published doubles: level C's histogram of the threshold's radix select (a sharded window sums it with the systems) search
    NALO_HIP(c, w.stitched.reserve(w.lo_off + kThDblC));
    NALO_HIP(c, hipMemsetAsync(w.stitched.p, 0, (w.lo_off + kThDblC) * 8, c->stream));
    w.dev.th_bufC = w.stitched.p + w.lo_off;
    w.th_lo_pending = false; w.th_pending = false; w.th_side_inflight = false;
    w.sd.M_top = w.acc13.p; w.sd.M_sc = w.G.p; w.sd.H = w.stitched.p;
    if (w.stitched_host) { (void)hipHostFree(w.stitched_host); w.stitched_host = nullptr; }
    NALO_HIP(c, hipHostMalloc((void**)&w.stitched_host, (2 * blk + 2 * W * W + 16) * 8, hipHostMallocMapped));
    w.stitched_host[2 * blk + 2 * W * W + 5] = -1.0; w.pub_seq = 0;
    int rc = upload_frame_th(c); if (rc) return rc;
    rc = set_adjoints(c); if (rc) return rc;
    rc = set_precalc(c); if (rc) return rc;
    NALO_HIP(c, hipStreamSynchronize(c->stream));
    w.have_lin = w.have_sc = false;
    return NALO_OK;
}

int nalo_ba_set_points(nalo_ctx* c, int P, const int* host, const float* u, const float* v, const float* idepth, const float* idepth_zero,
                       const float* color, const float* weights, const int* has_prior) {
    if (!c || !c->ba || c->ba->W < 2) return fail(c, NALO_ERR_STATE, "nalo_ba_set_points: set the window first");
    if (P < 0 || (P > 0 && (!host || !u || !v || !idepth || !color || !weights))) return fail(c, NALO_ERR_ARG, "nalo_ba_set_points: bad argument");
    BAWindow& w = *c->ba;
    const int W = w.W;
    NALO_HIP(c, hipSetDevice(c->device));
    // sort by host (stable: keeps the caller's order inside a host, i.e. the reference's frame->points iteration order), pad each host to kBlk
    std::vector<int> cnt(W, 0);
    for (int p = 0; p < P; ++p) { if (host[p] < 0 || host[p] >= W) return fail(c, NALO_ERR_ARG, "nalo_ba_set_points: host index out of range"); cnt[host[p]]++; }
    w.host_blk_h.assign(W + 1, 0);
    for (int h = 0; h < W; ++h) w.host_blk_h[h + 1] = w.host_blk_h[h] + (cnt[h] + kBlk - 1) / kBlk;
    w.nblocks = std::max(w.host_blk_h[W], 1);
    if (w.host_blk_h[W] == 0) w.host_blk_h[W] = 0;
    w.Ppad = w.nblocks * kBlk; w.P = P;
    w.blk_host_h.assign(w.nblocks, 0);
    for (int h = 0; h < W; ++h) for (int b = w.host_blk_h[h]; b < w.host_blk_h[h + 1]; ++b) w.blk_host_h[b] = h;
    w.d2p.assign(w.Ppad, -1); w.p2d.assign(P, -1);
    // inside a host: HILBERT order of the 8x8-pixel cell. Any run of consecutive points then covers a compact, connected patch of the host image (a
    // Morton range can jump across a quadrant boundary), so the 64 residuals of a wave project into a small window of the target image: that window is
    // what ba_linearize_tile_kernel stages in LDS, and what keeps the gather kernel's texels in L1/L2. Sums are order independent up to rounding.
    unsigned hn = 1; while ((int)hn * 8 < std::max(c->w, c->h)) hn <<= 1;
    auto hilbert = [hn](unsigned x, unsigned y) {
        unsigned long long d = 0;
        for (unsigned s = hn / 2; s > 0; s /= 2) {
            const unsigned rx = (x & s) ? 1u : 0u, ry = (y & s) ? 1u : 0u;
            d += (unsigned long long)s * s * ((3u * rx) ^ ry);
            if (ry == 0) { if (rx == 1) { x = hn - 1 - x; y = hn - 1 - y; } const unsigned tmp = x; x = y; y = tmp; }
        }
        return d;
    };
    std::vector<int> order(P);
    for (int p = 0; p < P; ++p) order[p] = p;
    std::vector<unsigned long long> key(P);
    for (int p = 0; p < P; ++p) key[p] = ((unsigned long long)host[p] << 40) | hilbert(std::min(hn - 1, (unsigned)std::max(0.f, u[p]) >> 3), std::min(hn - 1, (unsigned)std::max(0.f, v[p]) >> 3));
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return key[a] < key[b]; });
    std::vector<int> fill(W);
    for (int h = 0; h < W; ++h) fill[h] = w.host_blk_h[h] * kBlk;
    for (int q = 0; q < P; ++q) { const int p = order[q]; const int d = fill[host[p]]++; w.d2p[d] = p; w.p2d[p] = d; }
    const size_t N = w.Ppad;
    std::vector<float4> geo(N, make_float4(8.f, 8.f, 1.f, 1.f)), c0(N, make_float4(0, 0, 0, 0)), c1(N, make_float4(0, 0, 0, 0)), w0(N, make_float4(0, 0, 0, 0)), w1(N, make_float4(0, 0, 0, 0));
    std::vector<float> prior(N, 0.f);
    w.flags_h.assign(N, 0);
    for (size_t d = 0; d < N; ++d) {
        const int p = w.d2p[d];
        if (p < 0) continue;
        geo[d] = make_float4(u[p], v[p], idepth[p], idepth_zero ? idepth_zero[p] : idepth[p]);
        c0[d] = make_float4(color[8 * p], color[8 * p + 1], color[8 * p + 2], color[8 * p + 3]); c1[d] = make_float4(color[8 * p + 4], color[8 * p + 5], color[8 * p + 6], color[8 * p + 7]);
        w0[d] = make_float4(weights[8 * p], weights[8 * p + 1], weights[8 * p + 2], weights[8 * p + 3]); w1[d] = make_float4(weights[8 * p + 4], weights[8 * p + 5], weights[8 * p + 6], weights[8 * p + 7]);
        const bool hp = has_prior && has_prior[p];
        prior[d] = hp ? kIdepthFixPrior * kScaleIdepth * kScaleIdepth : 0.f;        // EFPoint::takeData (EnergyFunctionalStructs.cpp:79-85)
        w.flags_h[d] = PT_VALID | (hp ? PT_HAS_PRIOR : 0);
    }
    NALO_HIP(c, w.pt_geo.reserve(N)); NALO_HIP(c, w.pt_col0.reserve(N)); NALO_HIP(c, w.pt_col1.reserve(N)); NALO_HIP(c, w.pt_w0.reserve(N)); NALO_HIP(c, w.pt_w1.reserve(N));
    NALO_HIP(c, w.pt_acc.reserve(N)); NALO_HIP(c, w.pt_hcd.reserve(N)); NALO_HIP(c, w.pt_prior.reserve(N)); NALO_HIP(c, w.pt_step.reserve(N)); NALO_HIP(c, w.pt_backup.reserve(N));
    NALO_HIP(c, w.pt_relbs.reserve(N)); NALO_HIP(c, w.pt_relbs2.reserve(N)); NALO_HIP(c, w.en_new.reserve(N)); NALO_HIP(c, w.pt_flags.reserve(N)); NALO_HIP(c, w.pt_ngood.reserve(N));
    NALO_HIP(c, w.blk_host.reserve(w.nblocks)); NALO_HIP(c, w.host_blk.reserve(W + 1));
    const size_t NS = (size_t)W * N;
    NALO_HIP(c, w.rs_state.reserve(NS)); NALO_HIP(c, w.rs_energy.reserve(NS)); NALO_HIP(c, w.rs_jp0.reserve(NS)); NALO_HIP(c, w.rs_jp1.reserve(NS)); NALO_HIP(c, w.rs_cpt.reserve(NS)); NALO_HIP(c, w.rs_pp0.reserve(NS)); NALO_HIP(c, w.rs_pp1.reserve(NS));
    w.dev.lin_sub = w.nblocks >= 128 ? 1 : 4;             // small windows: one workgroup per wave fills more CUs (kernels_ba_lin.hip)
    NALO_HIP(c, w.top_partial.reserve((size_t)w.nblocks * w.dev.lin_sub * W * kTopStride)); {   // ba_sc work distribution: ~1000+ workgroups whatever the window size. Small windows split a point block over 4 (2) workgroups; large
        // ones put up to 8 blocks of a host through one workgroup so that the NPL^2 fp64 partial is written once per group.
        w.dev.sc_split = w.nblocks <= 256 ? 4 : 1;          // measured (scripts/tune_sc.sh): beyond ~256 blocks more workgroups only add partial traffic
        w.dev.sc_bpw = w.dev.sc_split > 1 ? 1 : (w.nblocks >= 2048 ? 8 : (w.nblocks >= 1024 ? 4 : 2));
        std::vector<int> grp(W + 1, 0);
        for (int h = 0; h < W; ++h) grp[h + 1] = grp[h] + (w.host_blk_h[h + 1] - w.host_blk_h[h] + w.dev.sc_bpw - 1) / w.dev.sc_bpw;
        w.dev.sc_groups = std::max(grp[W], 1);
        NALO_HIP(c, w.sc_grp.reserve(W + 1));
        NALO_HIP(c, hipMemcpy(w.sc_grp.p, grp.data(), (size_t)(W + 1) * 4, hipMemcpyHostToDevice));
        w.dev.sc_grp = w.sc_grp.p;
    }
    NALO_HIP(c, w.sc_partial.reserve((size_t)w.dev.sc_groups * w.dev.sc_split * (w.T * (w.T + 1) / 2) * 256));
    NALO_HIP(c, hipMemcpy(w.pt_geo.p, geo.data(), N * 16, hipMemcpyHostToDevice)); NALO_HIP(c, hipMemcpy(w.pt_col0.p, c0.data(), N * 16, hipMemcpyHostToDevice));
    NALO_HIP(c, hipMemcpy(w.pt_col1.p, c1.data(), N * 16, hipMemcpyHostToDevice)); NALO_HIP(c, hipMemcpy(w.pt_w0.p, w0.data(), N * 16, hipMemcpyHostToDevice));
    NALO_HIP(c, hipMemcpy(w.pt_w1.p, w1.data(), N * 16, hipMemcpyHostToDevice)); NALO_HIP(c, hipMemcpy(w.pt_prior.p, prior.data(), N * 4, hipMemcpyHostToDevice));
    NALO_HIP(c, hipMemcpy(w.pt_flags.p, w.flags_h.data(), N, hipMemcpyHostToDevice));
    NALO_HIP(c, hipMemcpy(w.blk_host.p, w.blk_host_h.data(), (size_t)w.nblocks * 4, hipMemcpyHostToDevice));
    NALO_HIP(c, hipMemcpy(w.host_blk.p, w.host_blk_h.data(), (size_t)(W + 1) * 4, hipMemcpyHostToDevice));
    {   // XCD walk lists: list x = the x-th eighth (in Morton = spatial order) of every host's blocks
        std::vector<std::vector<int>> lists(8);
        for (int h = 0; h < W; ++h) {
            const int b0 = w.host_blk_h[h], nb = w.host_blk_h[h + 1] - b0;
            for (int k = 0; k < nb; ++k) lists[std::min(7, (int)((long long)k * 8 / std::max(nb, 1)))].push_back(b0 + k);
        }
        // balance: an XCD with a short list would idle; move tail blocks from the longest to the shortest list
        for (;;) {
            int lo = 0, hi = 0;
            for (int x = 1; x < 8; ++x) { if (lists[x].size() < lists[lo].size()) lo = x; if (lists[x].size() > lists[hi].size()) hi = x; }
            if (lists[hi].size() <= lists[lo].size() + 1) break;
            lists[lo].push_back(lists[hi].back()); lists[hi].pop_back();
        }
        size_t len = 1;
        for (auto& l : lists) len = std::max(len, l.size());
        std::vector<int> order(8 * len, -1);
        for (int x = 0; x < 8; ++x) for (size_t k = 0; k < lists[x].size(); ++k) order[x * len + k] = lists[x][k];
        NALO_HIP(c, w.blk_order.reserve(order.size()));
        NALO_HIP(c, hipMemcpy(w.blk_order.p, order.data(), order.size() * 4, hipMemcpyHostToDevice));
        w.dev.blk_order = w.blk_order.p; w.dev.xcd_len = (int)len;
    }
    NALO_HIP(c, hipMemset(w.pt_acc.p, 0, N * 16)); NALO_HIP(c, hipMemset(w.pt_hcd.p, 0, N * 16)); NALO_HIP(c, hipMemset(w.pt_step.p, 0, N * 4));
    NALO_HIP(c, hipMemset(w.pt_backup.p, 0, N * 4)); NALO_HIP(c, hipMemset(w.pt_relbs.p, 0, N * 4)); NALO_HIP(c, hipMemset(w.pt_relbs2.p, 0, N * 4)); NALO_HIP(c, hipMemset(w.pt_ngood.p, 0, N));
    NALO_HIP(c, hipMemset(w.rs_state.p, 0, NS)); NALO_HIP(c, hipMemset(w.rs_energy.p, 0, NS * 8)); NALO_HIP(c, hipMemset(w.rs_jp0.p, 0, NS * 16)); NALO_HIP(c, hipMemset(w.rs_jp1.p, 0, NS * 16));
    NALO_HIP(c, hipMemset(w.rs_cpt.p, 0, NS * 16)); NALO_HIP(c, hipMemset(w.top_partial.p, 0, (size_t)w.nblocks * w.dev.lin_sub * W * kTopStride * 8));
    BADev& D = w.dev;
    D.P = P; D.Ppad = w.Ppad; D.nblocks = w.nblocks; D.blk_host = w.blk_host.p; D.host_blk = w.host_blk.p;
    D.pt_geo = w.pt_geo.p; D.pt_col0 = w.pt_col0.p; D.pt_col1 = w.pt_col1.p; D.pt_w0 = w.pt_w0.p; D.pt_w1 = w.pt_w1.p; D.pt_prior = w.pt_prior.p;
    D.pt_flags = w.pt_flags.p; D.pt_acc = w.pt_acc.p; D.pt_hcd = w.pt_hcd.p; D.pt_ngood = w.pt_ngood.p; D.pt_step = w.pt_step.p; D.pt_backup = w.pt_backup.p; D.pt_relbs = w.pt_relbs.p; D.pt_relbs_next = w.pt_relbs2.p;
    D.rs_state = w.rs_state.p; D.rs_energy = w.rs_energy.p; D.rs_jp0 = w.rs_jp0.p; D.rs_jp1 = w.rs_jp1.p; D.rs_cpt = w.rs_cpt.p; D.rs_pp0 = w.rs_pp0.p; D.rs_pp1 = w.rs_pp1.p; D.en_new = w.en_new.p;
    D.top_partial = w.top_partial.p; D.sc_partial = w.sc_partial.p;
    w.points_set = true; w.res_set = false; w.have_lin = w.have_sc = false;
    return NALO_OK;
}

int nalo_ba_set_residuals(nalo_ctx* c, const uint8_t* exists) {
    if (!c || !c->ba || !c->ba->points_set || !exists) return fail(c, NALO_ERR_STATE, "nalo_ba_set_residuals: set window and points first");
    BAWindow& w = *c->ba;
    const int W = w.W;
    std::vector<uint8_t> st((size_t)W * w.Ppad, 0);
    for (int p = 0; p < w.P; ++p) { const int d = w.p2d[p]; for (int t = 0; t < W; ++t) if (exists[(size_t)p * W + t]) st[(size_t)t * w.Ppad + d] = RS_EXISTS; }   // state IN (0), resetOOB
    NALO_HIP(c, hipMemcpy(w.rs_state.p, st.data(), st.size(), hipMemcpyHostToDevice));
    NALO_HIP(c, hipMemset(w.rs_energy.p, 0, st.size() * 8));
    w.res_set = true; w.have_lin = w.have_sc = false;
    return NALO_OK;
}

int nalo_ba_set_prior(nalo_ctx* c, const double* HM, const double* bM) {
    if (!c || !c->ba || c->ba->W < 2) return fail(c, NALO_ERR_STATE, "nalo_ba_set_prior: set the window first");
    BAWindow& w = *c->ba;
    if (HM) w.HM.assign(HM, HM + (size_t)w.n * w.n); else w.HM.assign((size_t)w.n * w.n, 0.0);
    if (bM) w.bM.assign(bM, bM + w.n); else w.bM.assign(w.n, 0.0);
    return NALO_OK;
}
int nalo_ba_get_prior(nalo_ctx* c, double* HM, double* bM) {
    if (!c || !c->ba || c->ba->W < 2) return fail(c, NALO_ERR_STATE, "nalo_ba_get_prior: set the window first");
    BAWindow& w = *c->ba;
    if (HM) std::memcpy(HM, w.HM.data(), w.HM.size() * 8);
    if (bM) std::memcpy(bM, w.bM.data(), w.bM.size() * 8);
    return NALO_OK;
}

#define NALO_BA_READY(name)                                                                                       \
    if (!c || !c->ba || !c->ba->points_set || !c->ba->res_set) return fail(c, NALO_ERR_STATE, name ": window/points/residuals not set"); \
    if (c->xchg_failed) return fail(c, NALO_ERR_HIP, name ": a cross-rank sum of this context failed earlier; rebuild the window on a new context"); \
    NALO_HIP(c, hipSetDevice(c->device));                                                                         \
    BAWindow& w = *c->ba;

int nalo_ba_linearize(nalo_ctx* c, int fix, double* energy) {
    NALO_BA_READY("nalo_ba_linearize")
    int rc = linearize_async(c, 0, fix); if (rc) return rc;
    w.pt_acc_on_read = true;
    rc = stitch_and_fetch(c, true, false, true); if (rc) return rc;
    { const float th = tail_th(w); rc = check_th(c, th); if (rc) return rc; w.frames[w.W - 1].frameEnergyTH = th; }
    double e = 0; misc_totals(w, &e, &w.resInA);
    if (energy) *energy = e;
    return NALO_OK;
}
int nalo_ba_accumulate(nalo_ctx* c, int mode, double* H, double* b) {
    NALO_BA_READY("nalo_ba_accumulate")
    if (!H || !b) return fail(c, NALO_ERR_ARG, "nalo_ba_accumulate: H/b required");
    if (mode == 0) { int rc = do_accumulate_top(c); if (rc) return rc; unpack_system(w, w.stitched_host, H, b); misc_totals(w, nullptr, &w.resInA); return NALO_OK; }
    if (mode == 1) { prior_system(w, H, b); w.resInL = 0; return NALO_OK; }
    return fail(c, NALO_ERR_UNSUPPORTED, "nalo_ba_accumulate: mode 2 runs inside nalo_ba_marginalize_points");
}
int nalo_ba_accumulate_sc(nalo_ctx* c, int shiftPriorToZero, double* H, double* b) {
    NALO_BA_READY("nalo_ba_accumulate_sc")
    if (!H || !b) return fail(c, NALO_ERR_ARG, "nalo_ba_accumulate_sc: H/b required");
    int rc = do_accumulate_sc(c, shiftPriorToZero ? 1 : 0); if (rc) return rc;
    unpack_system(w, w.stitched_host + (size_t)w.n1 * w.n1, H, b);
    return NALO_OK;
}
int nalo_ba_solve_system(nalo_ctx* c, int iteration, double lambda, double* x_out) {
    NALO_BA_READY("nalo_ba_solve_system")
    (void)w;
    int rc = solve_system(c, iteration, lambda, x_out); if (rc) return rc;
    NALO_HIP(c, hipStreamSynchronize(c->stream));
    return NALO_OK;
}
int nalo_ba_backup_state(nalo_ctx* c) {
    NALO_BA_READY("nalo_ba_backup_state")
    backup_state(w);
    return NALO_OK;
}
int nalo_ba_do_step(nalo_ctx* c, float fC, float fT, float fR, float fA, float fD, int* canbreak) {
    NALO_BA_READY("nalo_ba_do_step")
    (void)w;
    return do_step(c, fC, fT, fR, fA, fD, canbreak);
}

// FullSystemOptimize.cpp:550-602: fix the newest frame's linearisation point, linearizeAll(true), rmse
static int optimize_epilogue(nalo_ctx* c, double* rmse) {
    BAWindow& w = *c->ba;
    const int W = w.W;
    HostTimer hte(c, "ba.opt.epilogue");
    HostFrame& nf = w.frames[W - 1];                                        // :550-557
    const double nsz[10] = {0, 0, 0, 0, 0, 0, nf.state[6], nf.state[7], 0, 0};
    nf.evalPT = nf.PRE_worldToCam;
    frame_set_state(nf, nsz); frame_set_state_zero(nf, nsz);
    int rc = set_adjoints(c, true); if (rc) return rc;                      // host tables only: the device copy follows when somebody stitches (round 4: the copy packet and its event stood ~25 us
    rc = set_precalc(c); if (rc) return rc;                                 // between the last linearisation of the loop and this one; nothing in the epilogue reads the adjoints)
    rc = linearize_async(c, 0, 1); if (rc) return rc;                       // :562 linearizeAll(true)
    w.pt_acc_on_read = false;                                               // the per-point sums stay those of the last solve
    rc = stitch_and_fetch(c, true, false, true, true); if (rc) return rc;  // energy, residual count and the threshold: no stitch
    { const float th = tail_th(w); rc = check_th(c, th); if (rc) return rc; nf.frameEnergyTH = th; }
    // (the device copy of the adjoints is brought up to date by whoever stitches next - stitch_and_fetch -, or by the next set_adjoints: the epilogue has no reader for it)
    double e = 0; int nres = 0; misc_totals(w, &e, &nres);
    // the reference reports sqrt(E / (patternNum * resInA)) with resInA from the last accumulateAF (the last solve)
    if (rmse) *rmse = std::sqrt((float)(e / (kPatternNum * (double)w.resInA)));
    return NALO_OK;
}

static int optimize_impl(nalo_ctx* c, int mnumOptIts, int never_break, double* rmse);
int nalo_ba_optimize(nalo_ctx* c, int mnumOptIts, int never_break, double* rmse) {
    const int rc = optimize_impl(c, mnumOptIts, never_break, rmse);
    if (c && c->ba && (c->ba->resub_pre || c->ba->lin_pre)) prelaunch_cancel(c);      // an error between the pre-launch and its gates: nothing stays behind spinning
    return rc;
}
static int optimize_impl(nalo_ctx* c, int mnumOptIts, int never_break, double* rmse) {
    NALO_BA_READY("nalo_ba_optimize")
    HostTimer ht(c, "ba_optimize");
    const int W = w.W;
    if (W < 3) mnumOptIts = 20;                                             // FullSystemOptimize.cpp:401-403
    if (W < 4) mnumOptIts = 15;
    w.opt_iterations = 0; w.opt_rejected = 0;
    if (!c->set.forceAcceptStep) {
        // setting_forceAceptStep = false: every linearisation is an energy evaluation first (FIX = 2: nothing but state_NewEnergy and the threshold input
        // changes) and is applied only if E + E_L + E_M decreased; a rejected step restores the backup (:511-541). Host-driven, one sync per evaluation.
        // (a sharded window: the three scalars of the energy test are summed over the ranks, sum_over_ranks; every rank then takes the same accept / reject branch)
        ba_launch_reset_oob(c->stream, w.dev);
        double lastE, lastL, lastM;
        int rc = linearize_noapply(c, &lastE); if (rc) return rc;            // :436-438
        rc = calc_l_energy(c, &lastL); if (rc) return rc;
        lastM = calc_m_energy(w);
        rc = linearize_async(c, 0, 0, true); if (rc) return rc;              // applyRes (:459-462)
        double lambda = 1e-1;
        for (int it = 0; it < mnumOptIts; ++it) {
            backup_state(w);
            rc = solve_system(c, it, lambda, nullptr, true); if (rc) return rc;
            int canbreak = 0;
            rc = do_step(c, 1, 1, 1, 1, 1, &canbreak); if (rc) return rc;
            double newE, newL, newM;
            rc = linearize_noapply(c, &newE); if (rc) return rc;             // :511-513
            rc = calc_l_energy(c, &newL); if (rc) return rc;
            newM = calc_m_energy(w);
            ++w.opt_iterations;
            if (newE + newL + newM < lastE + lastL + lastM) {                // :519-532
                rc = linearize_async(c, 0, 0, true); if (rc) return rc;
                lastE = newE; lastL = newL; lastM = newM;
                lambda *= 0.25;
            } else {                                                         // :534-541
                rc = load_state_backup(c); if (rc) return rc;
                rc = linearize_noapply(c, &lastE); if (rc) return rc;
                rc = calc_l_energy(c, &lastL); if (rc) return rc;
                lastM = calc_m_energy(w);
                lambda *= 1e2;
                ++w.opt_rejected;
                // the accumulators, Jacobian products and residual states are still those of the last APPLIED linearisation (made at the restored
                // states); only the systems stitched from them have to be rebuilt for the next solve
                w.have_lin = true; w.have_sc = false; w.stitched_top = false; w.stitched_sc = false;
            }
            if (canbreak && it >= c->set.minOptIterations && !never_break) break;
        }
        return optimize_epilogue(c, rmse);
    }
    // :412-429 resetOOB rides in the first linearisation (every residual it touches is one the pass reads and writes back anyway; round 4: it was a 2.7 us launch
    // with its boundary in front of the 14 us pass). The energy-test variant above keeps the kernel: its first pass (FIX = 2) writes no state.
    w.dev.reset_oob = 1;
    int rc = linearize_async(c, 0, 0);                                      // :436 (+ applyRes :459-462)
    w.dev.reset_oob = 0;
    if (rc) return rc;
    double lambda = 1e-1;
    for (int it = 0; it < mnumOptIts; ++it) {
        ++w.opt_iterations;
        backup_state(w);                                                    // :482
        rc = solve_system(c, it, lambda, nullptr, true); if (rc) return rc; // :485 (+ the point part of doStepFromBackup)
        rc = do_step(c, 1, 1, 1, 1, 1, nullptr); if (rc) return rc;         // :501 (stepsize 1: no SOLVER_STEPMOMENTUM)
        rc = linearize_async(c, 0, 0); if (rc) return rc;                   // :511, accepted unconditionally (:519-532)
        lambda *= 0.25;
        // :544 `if(canbreak && iteration >= setting_minOptIterations) break;` — the step sums arrive with the next fetch, so the
        // test is made there; a loop that must break only costs one discarded accumulate+stitch, the state is untouched
        if (!never_break && it >= c->set.minOptIterations && it + 1 < mnumOptIts) {
            rc = stitch_and_fetch_for_break(c); if (rc) return rc;
            if (w.last_canbreak) break;
        }
    }
    return optimize_epilogue(c, rmse);
}

// the newest frame takes a new pose as its linearisation point: setEvalPT(PRE_worldToCam, {0.., a, b}) + setAdjointsF + the precalc values
// (PlaneOptimize.cpp:275-289, 421-436; the same steps as the end of FullSystem::optimize, FullSystemOptimize.cpp:550-557)
static int relinearize_newest(nalo_ctx* c, const SE3& worldToCam) {
    BAWindow& w = *c->ba;
    HostFrame& nf = w.frames[w.W - 1];
    const double nsz[10] = {0, 0, 0, 0, 0, 0, nf.state[6], nf.state[7], 0, 0};
    nf.evalPT = worldToCam;
    frame_set_state(nf, nsz); frame_set_state_zero(nf, nsz);
    w.proj_valid = false;
    int rc = set_adjoints(c); if (rc) return rc;
    rc = set_precalc(c); if (rc) return rc;
    w.have_lin = false; w.have_sc = false; w.stitched_top = false; w.stitched_sc = false;
    return NALO_OK;
}

int nalo_ba_plane_scale_fix(nalo_ctx* c, double localscale, const double camToTrackingRef[12], const double trackingRef_camToWorld[12]) {
    NALO_BA_READY("nalo_ba_plane_scale_fix")
    if (!camToTrackingRef || !trackingRef_camToWorld || !(localscale > 0) || !std::isfinite(localscale)) return fail(c, NALO_ERR_ARG, "nalo_ba_plane_scale_fix: bad argument");
    SE3 cam2ref = SE3::from(camToTrackingRef);                                // :259-261
    cam2ref.m[3] *= localscale; cam2ref.m[7] *= localscale; cam2ref.m[11] *= localscale;
    const SE3 camToWorld = SE3::from(trackingRef_camToWorld) * cam2ref;       // :267
    ba_launch_set_idepth(c->stream, w.dev, 1, w.W - 1, localscale);           // :270-275: the newest frame's own points
    NALO_HIP(c, hipGetLastError());
    return relinearize_newest(c, camToWorld.inverse());
}

int nalo_ba_sw_gray_optimize(nalo_ctx* c, double* cost, int* n_residual_blocks) {
    NALO_BA_READY("nalo_ba_sw_gray_optimize")
    const int W = w.W;
    // parameter blocks: per frame {translation, so3 log} of camToWorld.inverse() (:321-331); the functor rebuilds T = [exp(omega) | t] (PlaneOptimize.h:350-360)
    std::vector<SE3> T(W);
    for (int i = 0; i < W; ++i) {
        double xi[6]; se3_log(w.frames[i].PRE_worldToCam, xi);
        const double om[6] = {0, 0, 0, xi[3], xi[4], xi[5]};
        T[i] = se3_exp(om);
        T[i].m[3] = w.frames[i].PRE_worldToCam.t(0); T[i].m[7] = w.frames[i].PRE_worldToCam.t(1); T[i].m[11] = w.frames[i].PRE_worldToCam.t(2);
    }
    std::vector<double> Rt((size_t)W * W * 12, 0.0);
    for (int h = 0; h < W; ++h) for (int t = 0; t < W; ++t) {
        const SE3 Tij = T[t] * T[h].inverse();
        double* o = &Rt[(size_t)(h * W + t) * 12];
        for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) o[i * 3 + j] = Tij.R(i, j); o[9 + i] = Tij.t(i); }
    }
    const size_t npart = (size_t)w.nblocks * W * 2;
    NALO_HIP(c, w.noapply_E.reserve(npart + Rt.size()));
    double* dRt = w.noapply_E.p + npart;
    NALO_HIP(c, hipMemcpyAsync(dRt, Rt.data(), Rt.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
    ba_launch_swgray(c->stream, w.dev, dRt, w.noapply_E.p);
    std::vector<double> part(npart);
    NALO_HIP(c, hipMemcpyAsync(part.data(), w.noapply_E.p, npart * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    NALO_HIP(c, hipStreamSynchronize(c->stream));
    double cs = 0, cn = 0;
    for (size_t i = 0; i < npart; i += 2) { cs += part[i]; cn += part[i + 1]; }
    if (cost) *cost = cs;
    if (n_residual_blocks) *n_residual_blocks = (int)(cn + 0.5);
    // zero Jacobians -> Ceres returns the parameters it was given. After the solve (:396-440):
    //  * the newest frame's PRE_worldToCam is rebuilt from its parameter block: rotation = exp(log(R)) (a round trip through so3, not the identical matrix)
    //  * points hosted by frames 0 .. W-3 get setIdepth / setIdepthZero of their (unchanged) parameter: their linearisation point moves to the current value
    //  * the newest frame is re-linearised at that pose, adjoints and precalc values are recomputed
    ba_launch_set_idepth(c->stream, w.dev, 0, W - 2, 1.0);
    NALO_HIP(c, hipGetLastError());
    return relinearize_newest(c, T[W - 1]);
}

int nalo_ba_calc_l_energy(nalo_ctx* c, double* E) {
    NALO_BA_READY("nalo_ba_calc_l_energy")
    if (!E) return fail(c, NALO_ERR_ARG, "nalo_ba_calc_l_energy: bad argument");
    return calc_l_energy(c, E);
}
int nalo_ba_calc_m_energy(nalo_ctx* c, double* E) {
    NALO_BA_READY("nalo_ba_calc_m_energy")
    if (!E) return fail(c, NALO_ERR_ARG, "nalo_ba_calc_m_energy: bad argument");
    *E = calc_m_energy(w);
    return NALO_OK;
}
int nalo_ba_optimize_stats(nalo_ctx* c, int* iterations, int* rejected) {
    if (!c || !c->ba) return fail(c, NALO_ERR_STATE, "nalo_ba_optimize_stats: no window");
    if (iterations) *iterations = c->ba->opt_iterations;
    if (rejected) *rejected = c->ba->opt_rejected;
    return NALO_OK;
}
int nalo_get_settings(nalo_ctx* c, nalo_settings* out) {
    if (!c || !out) return fail(c, NALO_ERR_ARG, "nalo_get_settings: bad argument");
    *out = c->set;
    return NALO_OK;
}
int nalo_set_settings(nalo_ctx* c, const nalo_settings* in) {
    if (!c || !in) return fail(c, NALO_ERR_ARG, "nalo_set_settings: bad argument");
    if (in->minOptIterations < 0 || !std::isfinite(in->affineOptModeA) || !std::isfinite(in->affineOptModeB)) return fail(c, NALO_ERR_ARG, "nalo_set_settings: bad value");
    c->set = *in;
    if (c->ba) {                                                             // a window that is already set picks the new priors / flags up at once
        for (auto& f : c->ba->frames) frame_take_data(c->set, f);
        c->ba->dev.fix_a = c->set.affineOptModeA < 0; c->ba->dev.fix_b = c->set.affineOptModeB < 0;
    }
    return NALO_OK;
}

int nalo_ba_marginalize_points(nalo_ctx* c, const uint8_t* flags, double* M, double* Mb, double* Msc, double* Mbsc) {
    NALO_BA_READY("nalo_ba_marginalize_points")
    if (!flags) return fail(c, NALO_ERR_ARG, "nalo_ba_marginalize_points: flags required");
    const int n = w.n, n1 = w.n1, W = w.W;
    for (int p = 0; p < w.P; ++p) { const int d = w.p2d[p]; if (flags[p] && (w.flags_h[d] & PT_VALID)) w.flags_h[d] |= PT_MARG; }
    NALO_HIP(c, hipMemcpyAsync(w.pt_flags.p, w.flags_h.data(), w.Ppad, hipMemcpyHostToDevice, c->stream));
    int rc = linearize_async(c, 2, 0); if (rc) return rc;                                   // relinearise + fixLinearizationF + addPoint<2>
    rc = sc_async(c, 0, kIdepthFixPriorMargFac, 1); if (rc) return rc;                      // priorF *= margFac; addPoint(p, false)
    rc = stitch_and_fetch(c, true, true); if (rc) return rc;
    std::vector<double> m((size_t)n * n), mb(n), ms((size_t)n * n), mbs(n);
    unpack_system(w, w.stitched_host, m.data(), mb.data());
    unpack_system(w, w.stitched_host + (size_t)n1 * n1, ms.data(), mbs.data());
    int nres = 0; misc_totals(w, nullptr, &nres); w.resInM += nres;
    for (size_t i = 0; i < (size_t)n * n; ++i) w.HM[i] += kMargWeightFac * (m[i] - ms[i]);  // EnergyFunctional.cpp:654-669
    for (int i = 0; i < n; ++i) w.bM[i] += kMargWeightFac * (mb[i] - mbs[i]);
    if (M) std::memcpy(M, m.data(), m.size() * 8); if (Mb) std::memcpy(Mb, mb.data(), n * 8);
    if (Msc) std::memcpy(Msc, ms.data(), ms.size() * 8); if (Mbsc) std::memcpy(Mbsc, mbs.data(), n * 8);
    // removePoint: drop the points and all their residuals
    std::vector<uint8_t> st((size_t)W * w.Ppad);
    NALO_HIP(c, hipMemcpy(st.data(), w.rs_state.p, st.size(), hipMemcpyDeviceToHost));
    for (int d = 0; d < w.Ppad; ++d) if (w.flags_h[d] & PT_MARG) { w.flags_h[d] = 0; for (int t = 0; t < W; ++t) st[(size_t)t * w.Ppad + d] = 0; }
    NALO_HIP(c, hipMemcpy(w.rs_state.p, st.data(), st.size(), hipMemcpyHostToDevice));
    NALO_HIP(c, hipMemcpy(w.pt_flags.p, w.flags_h.data(), w.Ppad, hipMemcpyHostToDevice));
    w.have_lin = w.have_sc = false;
    return NALO_OK;
}

// EnergyFunctional::marginalizeFrame (EnergyFunctional.cpp:498-610): host fp64, (8W+4)^2. The frame is permuted to the end of HM/bM, its prior is
// added, the system is scaled by 1/sqrt(|diag|+10), the frame's 8x8 block is inverted and Schur-complemented away, unscaled, symmetrised.
int nalo_ba_marginalize_frame(nalo_ctx* c, int idx) {
    if (!c || !c->ba || c->ba->W < 2) return fail(c, NALO_ERR_STATE, "nalo_ba_marginalize_frame: set the window first");
    BAWindow& w = *c->ba;
    if (idx < 0 || idx >= w.W) return fail(c, NALO_ERR_ARG, "nalo_ba_marginalize_frame: frame index out of range");
    if (w.W < 3) return fail(c, NALO_ERR_STATE, "nalo_ba_marginalize_frame: a window needs two frames");
    if (w.points_set) for (int d = 0; d < w.Ppad; ++d)              // assert((int)fh->points.size()==0) at :505
        if ((w.flags_h[d] & PT_VALID) && w.blk_host_h[d / kBlk] == idx) return fail(c, NALO_ERR_STATE, "nalo_ba_marginalize_frame: the frame still hosts active points (marginalise or drop them first)");
    const int odim = w.n, ndim = odim - 8;
    std::vector<int> perm; perm.reserve(odim);
    for (int i = 0; i < odim; ++i) if (i < 4 || (i - 4) / 8 != idx) perm.push_back(i);
    for (int i = 0; i < 8; ++i) perm.push_back(4 + 8 * idx + i);
    std::vector<double> H((size_t)odim * odim), b(odim), S(odim), Si(odim);
    for (int i = 0; i < odim; ++i) { b[i] = w.bM[perm[i]]; for (int j = 0; j < odim; ++j) H[(size_t)i * odim + j] = w.HM[(size_t)perm[i] * odim + perm[j]]; }
    const HostFrame& fh = w.frames[idx];
    for (int i = 0; i < 8; ++i) { H[(size_t)(ndim + i) * odim + ndim + i] += fh.prior[i]; b[ndim + i] += fh.prior[i] * fh.delta_prior[i]; }   // :544-545
    for (int i = 0; i < odim; ++i) { S[i] = std::sqrt(std::fabs(H[(size_t)i * odim + i]) + 10); Si[i] = 1.0 / S[i]; }                     // :552-553
    for (int i = 0; i < odim; ++i) { for (int j = 0; j < odim; ++j) H[(size_t)i * odim + j] = Si[i] * H[(size_t)i * odim + j] * Si[j]; b[i] *= Si[i]; }
    double hp[64], hpi[64];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) hp[i * 8 + j] = H[(size_t)(ndim + i) * odim + ndim + j];
    if (!inv_lu(8, hp, hpi)) return fail(c, NALO_ERR_STATE, "nalo_ba_marginalize_frame: singular frame block");                            // :564-567 (0.5f*(hpi+hpi) is the identity)
    std::vector<double> bli((size_t)ndim * 8);
    for (int r = 0; r < ndim; ++r) for (int cc = 0; cc < 8; ++cc) { double s = 0; for (int k = 0; k < 8; ++k) s += H[(size_t)(ndim + k) * odim + r] * hpi[k * 8 + cc]; bli[(size_t)r * 8 + cc] = s; }
    for (int r = 0; r < ndim; ++r) {                                                                                                            // :570-572
        for (int cc = 0; cc < ndim; ++cc) { double s = 0; for (int k = 0; k < 8; ++k) s += bli[(size_t)r * 8 + k] * H[(size_t)(ndim + k) * odim + cc]; H[(size_t)r * odim + cc] -= s; }
        double s = 0; for (int k = 0; k < 8; ++k) s += bli[(size_t)r * 8 + k] * b[ndim + k];
        b[r] -= s;
    }
    for (int i = 0; i < odim; ++i) { for (int j = 0; j < odim; ++j) H[(size_t)i * odim + j] = S[i] * H[(size_t)i * odim + j] * S[j]; b[i] *= S[i]; }   // :575-576
    std::vector<double> HMn((size_t)ndim * ndim), bMn(ndim);
    for (int i = 0; i < ndim; ++i) { for (int j = 0; j < ndim; ++j) HMn[(size_t)i * ndim + j] = 0.5 * (H[(size_t)i * odim + j] + H[(size_t)j * odim + i]); bMn[i] = b[i]; }   // :579-580
    w.HM.swap(HMn); w.bM.swap(bMn);
    // the frame leaves the window (:583-590; FullSystem::marginalizeFrame drops every residual that targets it and re-runs setPrecalcValues / setAdjointsF,
    // FullSystemMarginalize.cpp:155-212): the device arrays are laid out per window size, so the caller re-issues nalo_ba_set_window (+ points, residuals)
    // for the frames that remain — with the next keyframe appended, set_window extends HM/bM like insertFrame does.
    w.frames.erase(w.frames.begin() + idx);
    w.W -= 1; w.n = 8 * w.W + 4; w.n1 = w.n + 1;
    w.points_set = false; w.res_set = false; w.have_lin = w.have_sc = false; w.proj_valid = false; w.have_snap = false;
    w.lastX.assign(w.n, 0.0);
    w.prior_next = true;
    return NALO_OK;
}

int nalo_ba_set_prior_carry(nalo_ctx* c, int on) {
    if (!c) return NALO_ERR_ARG;
    if (!c->ba) c->ba = new BAWindow();
    c->ba->prior_carry = on != 0;
    return NALO_OK;
}

int nalo_ba_get_frames(nalo_ctx* c, nalo_frame_state* frames, double* worldToCam, double calib[4]) {
    if (!c || !c->ba || c->ba->W < 2) return fail(c, NALO_ERR_STATE, "nalo_ba_get_frames: set the window first");
    BAWindow& w = *c->ba;
    for (int i = 0; i < w.W; ++i) {
        const HostFrame& f = w.frames[i];
        if (frames) {
            nalo_frame_state& s = frames[i];
            s.slot = f.slot; s.frame_id = f.frameID; std::memcpy(s.worldToCam_evalPT, f.evalPT.m, 96);
            std::memcpy(s.state, f.state, 80); std::memcpy(s.state_zero, f.state_zero, 80); s.ab_exposure = f.ab_exposure; s.frameEnergyTH = f.frameEnergyTH;
        }
        if (worldToCam) std::memcpy(worldToCam + 12 * i, f.PRE_worldToCam.m, 96);
    }
    if (calib) std::memcpy(calib, w.c_value_scaled, 32);
    return NALO_OK;
}

int nalo_ba_get_points(nalo_ctx* c, float* idepth, float* step, float* HdiF, float* bdSumF, float* Hdd, float* bd, float* Hcd, float* maxRelBaseline) {
    if (!c || !c->ba || !c->ba->points_set) return fail(c, NALO_ERR_STATE, "nalo_ba_get_points: no points");
    BAWindow& w = *c->ba;
    const size_t N = w.Ppad;
    // Hdd/bd/Hcd/HdiF/bdSumF are values of the last ACCUMULATION (addPoint<0> / AccumulatedSCHessianSSE::addPoint), as in the reference: after
    // nalo_ba_optimize they are those of the last solveSystemF, which is what CoarseTracker::makeCoarseDepthL0 reads (HdiF, CoarseTracker.cpp:396).
    // After an explicit nalo_ba_linearize the accumulation of that linearisation is run here if the caller has not asked for it yet.
    if (w.have_lin && !w.have_sc && w.pt_acc_on_read) { int rc = sc_async(c, 1, 1.f, 0); if (rc) return rc; }
    NALO_HIP(c, hipStreamSynchronize(c->stream));
    std::vector<float4> geo(N), acc(N), hcd(N);
    std::vector<float> stp(N), rel(N);
    NALO_HIP(c, hipMemcpy(geo.data(), w.pt_geo.p, N * 16, hipMemcpyDeviceToHost)); NALO_HIP(c, hipMemcpy(acc.data(), w.pt_acc.p, N * 16, hipMemcpyDeviceToHost));
    NALO_HIP(c, hipMemcpy(hcd.data(), w.pt_hcd.p, N * 16, hipMemcpyDeviceToHost)); NALO_HIP(c, hipMemcpy(stp.data(), w.pt_step.p, N * 4, hipMemcpyDeviceToHost));
    NALO_HIP(c, hipMemcpy(rel.data(), w.dev.pt_relbs, N * 4, hipMemcpyDeviceToHost));      // the buffer of the last pass that recorded relBS
    for (int p = 0; p < w.P; ++p) {
        const int d = w.p2d[p];
        if (idepth) idepth[p] = geo[d].z; if (step) step[p] = stp[d]; if (HdiF) HdiF[p] = acc[d].z; if (bdSumF) bdSumF[p] = acc[d].w;
        if (Hdd) Hdd[p] = acc[d].x; if (bd) bd[p] = acc[d].y;
        if (Hcd) { Hcd[4 * p] = hcd[d].x; Hcd[4 * p + 1] = hcd[d].y; Hcd[4 * p + 2] = hcd[d].z; Hcd[4 * p + 3] = hcd[d].w; }
        if (maxRelBaseline) maxRelBaseline[p] = rel[d];
    }
    return NALO_OK;
}

int nalo_ba_get_idepth_zero(nalo_ctx* c, float* idepth_zero) {
    if (!c || !c->ba || !c->ba->points_set || !idepth_zero) return fail(c, NALO_ERR_STATE, "nalo_ba_get_idepth_zero: no points");
    BAWindow& w = *c->ba;
    std::vector<float4> geo(w.Ppad);
    NALO_HIP(c, hipStreamSynchronize(c->stream));
    NALO_HIP(c, hipMemcpy(geo.data(), w.pt_geo.p, (size_t)w.Ppad * 16, hipMemcpyDeviceToHost));
    for (int p = 0; p < w.P; ++p) idepth_zero[p] = geo[w.p2d[p]].w;
    return NALO_OK;
}

int nalo_ba_get_residuals(nalo_ctx* c, int8_t* state, uint8_t* active, float* JpJdF, float* energy_new, float* center) {
    if (!c || !c->ba || !c->ba->points_set) return fail(c, NALO_ERR_STATE, "nalo_ba_get_residuals: no points");
    BAWindow& w = *c->ba;
    const int W = w.W; const size_t NS = (size_t)W * w.Ppad;
    NALO_HIP(c, hipStreamSynchronize(c->stream));
    std::vector<uint8_t> st(NS); std::vector<float4> j0(NS), j1(NS), cp(NS); std::vector<float> en(w.Ppad);
    NALO_HIP(c, hipMemcpy(st.data(), w.rs_state.p, NS, hipMemcpyDeviceToHost));
    if (JpJdF) { NALO_HIP(c, hipMemcpy(j0.data(), w.rs_jp0.p, NS * 16, hipMemcpyDeviceToHost)); NALO_HIP(c, hipMemcpy(j1.data(), w.rs_jp1.p, NS * 16, hipMemcpyDeviceToHost)); }
    if (center) NALO_HIP(c, hipMemcpy(cp.data(), w.rs_cpt.p, NS * 16, hipMemcpyDeviceToHost));
    if (energy_new) NALO_HIP(c, hipMemcpy(en.data(), w.en_new.p, (size_t)w.Ppad * 4, hipMemcpyDeviceToHost));
    for (int p = 0; p < w.P; ++p) for (int t = 0; t < W; ++t) {
        const size_t si = (size_t)t * w.Ppad + w.p2d[p], o = (size_t)p * W + t;
        const uint8_t s = st[si];
        if (state) state[o] = (s & RS_EXISTS) ? (int8_t)(s & RS_STATE_MASK) : (int8_t)-1;
        if (active) active[o] = (s & RS_ACTIVE) ? 1 : 0;
        if (JpJdF) { float* q = JpJdF + o * 8; q[0] = j0[si].x; q[1] = j0[si].y; q[2] = j0[si].z; q[3] = j0[si].w; q[4] = j1[si].x; q[5] = j1[si].y; q[6] = j1[si].z; q[7] = j1[si].w; }
        if (energy_new) energy_new[o] = (t == W - 1) ? en[w.p2d[p]] : -1.f;
        if (center) { center[o * 3] = cp[si].x; center[o * 3 + 1] = cp[si].y; center[o * 3 + 2] = cp[si].z; }
    }
    return NALO_OK;
}

int nalo_ba_get_acc13(nalo_ctx* c, double* H13) {
    if (!c || !c->ba || !c->ba->have_lin || !H13) return fail(c, NALO_ERR_STATE, "nalo_ba_get_acc13: linearize first");
    BAWindow& w = *c->ba;
    int rc = stitch_and_fetch(c, true, false); if (rc) return rc;
    NALO_HIP(c, hipMemcpy(H13, w.acc13.p, (size_t)w.W * w.W * 169 * 8, hipMemcpyDeviceToHost));
    return NALO_OK;
}
int nalo_ba_counts(nalo_ctx* c, int* a, int* l, int* m) {
    if (!c || !c->ba) return fail(c, NALO_ERR_STATE, "nalo_ba_counts: no window");
    if (a) *a = c->ba->resInA; if (l) *l = c->ba->resInL; if (m) *m = c->ba->resInM;
    return NALO_OK;
}
int nalo_ba_set_allreduce_mode(nalo_ctx* c, int stream_ordered) {
    if (!c || !c->ba) return fail(c, NALO_ERR_STATE, "nalo_ba_set_allreduce_mode: set the window first");
    c->ba->hook_stream_ordered = stream_ordered != 0;
    return NALO_OK;
}
int nalo_ba_set_allreduce(nalo_ctx* c, nalo_allreduce_fn hook, void* user) {
    if (!c) return NALO_ERR_ARG;
    if (!c->ba) c->ba = new BAWindow();
    // a pass whose level-C histogram has not been summed yet (the last linearisation of optimize(), say) is finished with the hook it was started under: every rank
    // changes its hook at the same point of the program, so the collective still matches
    // (also when the same function is registered again - with another user / communicator pointer, or idempotently per keyframe: the pending sum belongs to the OLD pair)
    if (c->ba->th_lo_pending && c->ba->hook && !c->xchg_failed) { NALO_HIP(c, hipSetDevice(c->device)); const int rc = flush_th(c); if (rc) return rc; }
    c->ba->th_lo_pending = false;
    c->ba->hook = hook; c->ba->hook_user = user;
    return NALO_OK;
}
// a caller-supplied hook has no return value: this is how it reports that its collective failed (the built-in RCCL hooks set the same latch, host_rccl.hip)
int nalo_ba_exchange_failed(nalo_ctx* c, const char* what) {
    if (!c) return NALO_ERR_ARG;
    if (!c->xchg_failed) { c->xchg_failed = true; c->err = std::string("cross-rank sum: ") + (what ? what : "reported as failed by the hook"); }
    return NALO_OK;
}
int nalo_ba_set_allreduce_side(nalo_ctx* c, nalo_allreduce_fn hook, void* user) {
    if (!c) return NALO_ERR_ARG;
    if (!c->ba) c->ba = new BAWindow();
    c->ba->hook_side = hook; c->ba->hook_side_user = user;
    return NALO_OK;
}

// CoarseDistanceMap::makeDistanceMap for the window's active points (CoarseTracker.cpp:1410-1444)
int nalo_dist_make_map(nalo_ctx* c, int frame, const float* KRKi, const float* Kt, float* out) {
    if (!c || !c->ba || !c->ba->points_set) return fail(c, NALO_ERR_STATE, "nalo_dist_make_map: set the window and its points first");
    BAWindow& w = *c->ba;
    if (!KRKi || !Kt || !out || frame < 0 || frame >= w.W || c->levels < 2) return fail(c, NALO_ERR_ARG, "nalo_dist_make_map: bad argument");
    NALO_HIP(c, hipSetDevice(c->device));
    const int W = w.W, w1 = c->wl[1], h1 = c->hl[1];
    const size_t npx = (size_t)w1 * h1, words = 12 * (size_t)W + npx + (npx + 3) / 4 + 8;
    int rc = imm_stage(c, words); if (rc) return rc;
    float* hst = c->imm_host; float* d = c->imm_dev.p;
    std::memcpy(hst, KRKi, 9 * (size_t)W * 4); std::memcpy(hst + 9 * W, Kt, 3 * (size_t)W * 4);
    NALO_HIP(c, hipMemcpyAsync(d, hst, 12 * (size_t)W * 4, hipMemcpyHostToDevice, c->stream));
    float* dout = d + 12 * W; uint8_t* seed = (uint8_t*)(dout + npx);
    rc = dist_make_launch(c, w.pt_geo.p, w.pt_flags.p, w.blk_host.p, w.Ppad, frame, d, d + 9 * W, seed, dout);
    if (rc) return rc;
    NALO_HIP(c, hipMemcpyAsync(hst + 12 * W, dout, npx * 4, hipMemcpyDeviceToHost, c->stream));
    NALO_HIP(c, hipStreamSynchronize(c->stream));
    std::memcpy(out, hst + 12 * W, npx * 4);
    return NALO_OK;
}

// FullSystem::optimizeImmaturePoint for a batch of immature points against the current window (FullSystemOptPoint.cpp:51-206)
int nalo_imm_optimize(nalo_ctx* c, int n, const int* host, const float* u, const float* v, const float* color, const float* weights,
                      const float* energyTH, const float* idepth_min, const float* idepth_max, int minObs, int* result, float* idepth_out, uint8_t* res_in) {
    if (!c || !c->ba || c->ba->W < 2) return fail(c, NALO_ERR_STATE, "nalo_imm_optimize: set the window first (nalo_ba_set_window)");
    if (n < 0 || (n > 0 && (!host || !u || !v || !color || !weights || !energyTH || !idepth_min || !idepth_max || !result || !idepth_out || !res_in)))
        return fail(c, NALO_ERR_ARG, "nalo_imm_optimize: bad argument");
    if (n == 0) return NALO_OK;
    BAWindow& w = *c->ba;
    const int W = w.W;
    for (int i = 0; i < n; ++i) if (host[i] < 0 || host[i] >= W) return fail(c, NALO_ERR_ARG, "nalo_imm_optimize: host out of range");
    NALO_HIP(c, hipSetDevice(c->device));
    HostTimer ht(c, "imm_optimize");
    // words: [0,21n) u v color weights energyTH idmin idmax | [21n,22n) host | [22n, +14 W^2) Rt aff | outputs: result(n) idepth(n) res_in(n*W bytes)
    const size_t N = (size_t)n, PW = (size_t)W * W, out0 = 22 * N + 14 * PW, outw = 2 * N + (N * W + 3) / 4;
    int rc = imm_stage(c, out0 + outw); if (rc) return rc;
    float* hst = c->imm_host;
    std::memcpy(hst, u, N * 4); std::memcpy(hst + N, v, N * 4); std::memcpy(hst + 2 * N, color, 8 * N * 4); std::memcpy(hst + 10 * N, weights, 8 * N * 4);
    std::memcpy(hst + 18 * N, energyTH, N * 4); std::memcpy(hst + 19 * N, idepth_min, N * 4); std::memcpy(hst + 20 * N, idepth_max, N * 4); std::memcpy(hst + 21 * N, host, N * 4);
    float* Rt = hst + 22 * N; float* af = Rt + 12 * PW;
    for (int h = 0; h < W; ++h) for (int t = 0; t < W; ++t) {               // FrameFramePrecalc::set (HessianBlocks.cpp:203-221) at the current states
        const HostFrame &hf = w.frames[h], &tf = w.frames[t];
        const SE3 ll = tf.PRE_worldToCam * hf.PRE_camToWorld;
        float* o = Rt + (size_t)(h * W + t) * 12;
        for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) o[i * 3 + j] = (float)ll.R(i, j); o[9 + i] = (float)ll.t(i); }
        double a[2];
        aff_from_to(hf.ab_exposure, tf.ab_exposure, hf.state_scaled[6], hf.state_scaled[7], tf.state_scaled[6], tf.state_scaled[7], a);
        af[(h * W + t) * 2] = (float)a[0]; af[(h * W + t) * 2 + 1] = (float)a[1];
    }
    float* d = c->imm_dev.p;
    NALO_HIP(c, hipMemcpyAsync(d, hst, out0 * 4, hipMemcpyHostToDevice, c->stream));
    const float K[4] = {w.c_scaledf[0], w.c_scaledf[1], w.c_scaledf[2], w.c_scaledf[3]};
    rc = imm_optimize_launch(c, w.dev.img, W, K, d + 22 * N, d + 22 * N + 12 * PW, n, (const int*)(d + 21 * N), d, minObs, (int*)(d + out0), d + out0 + N, (uint8_t*)(d + out0 + 2 * N));
    if (rc) return rc;
    NALO_HIP(c, hipMemcpyAsync(hst + out0, d + out0, outw * 4, hipMemcpyDeviceToHost, c->stream));
    NALO_HIP(c, hipStreamSynchronize(c->stream));
    std::memcpy(result, hst + out0, N * 4); std::memcpy(idepth_out, hst + out0 + N, N * 4); std::memcpy(res_in, hst + out0 + 2 * N, N * W);
    return NALO_OK;
}

// The same for points of the DEVICE-RESIDENT set (nalo_imm_resident_set / _trace, round 4): the caller names them by index, their pattern, weights, host frame and -
// as the device's last trace left them - inverse-depth interval are read where they are. Per call 4 bytes per point go down instead of 88, the results come back as
// in nalo_imm_optimize. sel == NULL: all resident points, n = their number.
int nalo_imm_resident_optimize(nalo_ctx* c, int n, const int* sel, int minObs, int* result, float* idepth_out, uint8_t* res_in) {
    if (!c || !c->ba || c->ba->W < 2) return fail(c, NALO_ERR_STATE, "nalo_imm_resident_optimize: set the window first (nalo_ba_set_window)");
    if (n < 0 || (n > 0 && (!result || !idepth_out || !res_in))) return fail(c, NALO_ERR_ARG, "nalo_imm_resident_optimize: bad argument");
    if (n == 0) return NALO_OK;
    BAWindow& w = *c->ba;
    const int W = w.W;
    if (c->imm_res_n <= 0) return fail(c, NALO_ERR_STATE, "nalo_imm_resident_optimize: no resident points (nalo_imm_resident_set)");
    if (c->imm_res_maxhost >= W) return fail(c, NALO_ERR_ARG, "nalo_imm_resident_optimize: a resident point's host_idx is outside the window");
    if (!sel && n != c->imm_res_n) return fail(c, NALO_ERR_ARG, "nalo_imm_resident_optimize: sel == NULL means all resident points");
    if (sel) for (int i = 0; i < n; ++i) if (sel[i] < 0 || sel[i] >= c->imm_res_n) return fail(c, NALO_ERR_ARG, "nalo_imm_resident_optimize: index outside the resident set");
    NALO_HIP(c, hipSetDevice(c->device));
    HostTimer ht(c, "imm_optimize");
    // words: [0, n) sel | [n, n + 14 W^2) Rt aff | outputs: result(n) idepth(n) res_in(n*W bytes)
    const size_t N = (size_t)n, PW = (size_t)W * W, out0 = N + 14 * PW, outw = 2 * N + (N * W + 3) / 4;
    int rc = imm_stage(c, out0 + outw); if (rc) return rc;
    float* hst = c->imm_host;
    if (sel) std::memcpy(hst, sel, N * 4);
    float* Rt = hst + N; float* af = Rt + 12 * PW;
    for (int h = 0; h < W; ++h) for (int t = 0; t < W; ++t) {               // FrameFramePrecalc::set (HessianBlocks.cpp:203-221) at the current states
        const HostFrame &hf = w.frames[h], &tf = w.frames[t];
        const SE3 ll = tf.PRE_worldToCam * hf.PRE_camToWorld;
        float* o = Rt + (size_t)(h * W + t) * 12;
        for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) o[i * 3 + j] = (float)ll.R(i, j); o[9 + i] = (float)ll.t(i); }
        double a[2];
        aff_from_to(hf.ab_exposure, tf.ab_exposure, hf.state_scaled[6], hf.state_scaled[7], tf.state_scaled[6], tf.state_scaled[7], a);
        af[(h * W + t) * 2] = (float)a[0]; af[(h * W + t) * 2 + 1] = (float)a[1];
    }
    float* d = c->imm_dev.p;
    NALO_HIP(c, hipMemcpyAsync(d, hst, out0 * 4, hipMemcpyHostToDevice, c->stream));
    const float K[4] = {w.c_scaledf[0], w.c_scaledf[1], w.c_scaledf[2], w.c_scaledf[3]};
    rc = imm_optimize_resident_launch(c, w.dev.img, W, K, d + N, d + N + 12 * PW, n, sel ? (const int*)d : nullptr, c->imm_res.p, (size_t)c->imm_res_n, minObs,
                                      (int*)(d + out0), d + out0 + N, (uint8_t*)(d + out0 + 2 * N));
    if (rc) return rc;
    NALO_HIP(c, hipMemcpyAsync(hst + out0, d + out0, outw * 4, hipMemcpyDeviceToHost, c->stream));
    NALO_HIP(c, hipStreamSynchronize(c->stream));
    std::memcpy(result, hst + out0, N * 4); std::memcpy(idepth_out, hst + out0 + N, N * 4); std::memcpy(res_in, hst + out0 + 2 * N, N * W);
    return NALO_OK;
}

int nalo_ba_snapshot(nalo_ctx* c) {
    NALO_BA_READY("nalo_ba_snapshot")
    const size_t N = w.Ppad, NS = (size_t)w.W * N;
    NALO_HIP(c, w.snap_geo.reserve(N)); NALO_HIP(c, w.snap_state.reserve(NS)); NALO_HIP(c, w.snap_flags.reserve(N)); NALO_HIP(c, w.snap_prior.reserve(N));
    NALO_HIP(c, hipMemcpyAsync(w.snap_geo.p, w.pt_geo.p, N * 16, hipMemcpyDeviceToDevice, c->stream));
    NALO_HIP(c, hipMemcpyAsync(w.snap_state.p, w.rs_state.p, NS, hipMemcpyDeviceToDevice, c->stream));
    NALO_HIP(c, hipMemcpyAsync(w.snap_flags.p, w.pt_flags.p, N, hipMemcpyDeviceToDevice, c->stream));
    NALO_HIP(c, hipMemcpyAsync(w.snap_prior.p, w.pt_prior.p, N * 4, hipMemcpyDeviceToDevice, c->stream));
    NALO_HIP(c, hipStreamSynchronize(c->stream));
    w.snap_frames = w.frames; w.snap_HM = w.HM; w.snap_bM = w.bM; w.snap_flags_h = w.flags_h;
    std::memcpy(w.snap_calib, w.c_value, sizeof(w.snap_calib)); std::memcpy(w.snap_calib_scaled, w.c_value_scaled, sizeof(w.snap_calib_scaled));
    std::memcpy(w.snap_scaledf, w.c_scaledf, sizeof(w.snap_scaledf)); std::memcpy(w.snap_scaledi, w.c_scaledi, sizeof(w.snap_scaledi));
    w.have_snap = true;
    return NALO_OK;
}
int nalo_ba_restore(nalo_ctx* c) {
    NALO_BA_READY("nalo_ba_restore")
    HostTimer ht(c, "ba_restore");
    if (!w.have_snap) return fail(c, NALO_ERR_STATE, "nalo_ba_restore: no snapshot");
    const size_t N = w.Ppad, NS = (size_t)w.W * N;
    (void)N; (void)NS;
    w.frames = w.snap_frames; w.HM = w.snap_HM; w.bM = w.snap_bM; w.flags_h = w.snap_flags_h;
    {   // one launch: device state from the snapshot, energies zeroed, thresholds installed (a pending quantile pass is flushed first: it clears its histogram)
        int rf = flush_th(c); if (rf) return rf;
        float th[16] = {};
        for (int i = 0; i < w.W; ++i) th[i] = w.frames[i].frameEnergyTH;
        NALO_HIP(c, w.frameTH.reserve(16));
        w.dev.frameTH = w.frameTH.p;
        ba_launch_restore(c->stream, w.dev, w.snap_geo.p, w.snap_state.p, w.snap_flags.p, w.snap_prior.p, th);
        w.th_pending = false;
        NALO_HIP(c, hipGetLastError());
    }
    std::memcpy(w.c_value, w.snap_calib, sizeof(w.snap_calib)); std::memcpy(w.c_value_scaled, w.snap_calib_scaled, sizeof(w.snap_calib_scaled));
    std::memcpy(w.c_scaledf, w.snap_scaledf, sizeof(w.snap_scaledf)); std::memcpy(w.c_scaledi, w.snap_scaledi, sizeof(w.snap_scaledi));
    int rc = set_adjoints(c); if (rc) return rc;
    rc = set_precalc(c); if (rc) return rc;
    w.have_lin = w.have_sc = false;
    return NALO_OK;
}


}  // extern "C"
