// C-ABI: context, frame pyramids, front-end tracker (host mirror of CoarseTracker, reference
// src/FullSystem/CoarseTracker.{h,cpp}). The LM loop of trackNewestCoarse stays on the host exactly as in the
// reference (8x8 LDL^T, SE3::exp); every calcRes/calcGSSSE pair is one fused kernel launch.
#include "nalo_internal.h"
#include <cstdlib>

using namespace nalo;

namespace nalo { int trk_lm_launch(nalo_ctx* c, int slot_new, const double T0[12], const double aff0[2], const double ref_aff[2], const float exposures[2], int coarsest, int stop_lvl, const double* minRes, double out24[32]); }

namespace nalo { void ba_destroy(nalo_ctx* c); }

static int pyr_levels_rule(int w, int h) {           // util/globalCalib.cpp:50-55
    int wl = w, hl = h, lv = 1;
    while (wl % 2 == 0 && hl % 2 == 0 && wl * hl > 5000 && lv < NALO_MAX_LEVELS) { wl /= 2; hl /= 2; lv++; }
    return lv;
}
static void set_pyr_calib(nalo_ctx* c, float fx, float fy, float cx, float cy) {   // globalCalib.cpp:74-104 / CoarseTracker::makeK
    c->fx[0] = fx; c->fy[0] = fy; c->cx[0] = cx; c->cy[0] = cy;
    for (int l = 1; l < c->levels; ++l) {
        c->fx[l] = c->fx[l - 1] * 0.5; c->fy[l] = c->fy[l - 1] * 0.5;
        c->cx[l] = (c->cx[0] + 0.5) / ((int)1 << l) - 0.5; c->cy[l] = (c->cy[0] + 0.5) / ((int)1 << l) - 0.5;
    }
}

extern "C" {

int nalo_create(nalo_ctx** out, int device, int w, int h, int levels, const float K[4], int n_slots) {
    if (!out || !K || w < 16 || h < 16 || n_slots < 1) return NALO_ERR_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return NALO_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return NALO_ERR_NO_DEVICE;
    nalo_ctx* c = new nalo_ctx();
    c->device = device; c->w = w; c->h = h;
    c->levels = levels > 0 ? levels : pyr_levels_rule(w, h);
    if (c->levels > NALO_MAX_LEVELS) { delete c; return NALO_ERR_ARG; }
    for (int l = 0; l < c->levels; ++l) { c->wl[l] = w >> l; c->hl[l] = h >> l; }
    for (int i = 0; i < 4; ++i) c->K0[i] = K[i];
    set_pyr_calib(c, K[0], K[1], K[2], K[3]);
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess) { delete c; return NALO_ERR_HIP; }
    c->slots.resize(n_slots);
    for (auto& s : c->slots)
        for (int l = 0; l < c->levels; ++l) {
            const size_t npx = (size_t)c->wl[l] * c->hl[l];
            if (hipMalloc((void**)&s.I[l], npx * 4) != hipSuccess || hipMalloc((void**)&s.dI[l], npx * 16) != hipSuccess ||
                hipMalloc((void**)&s.absg[l], npx * 4) != hipSuccess) { nalo_destroy(c); return NALO_ERR_HIP; }
        }
    if (hipHostMalloc((void**)&c->trk_out_host, 128 * sizeof(double), hipHostMallocMapped) != hipSuccess) { nalo_destroy(c); return NALO_ERR_HIP; }
    *out = c;
    return NALO_OK;
}

void nalo_destroy(nalo_ctx* c) {
    if (!c) return;
    if (std::getenv("NALO_HOST_TIMING")) for (auto& kv : c->host_t) fprintf(stderr, "[nalo host] %-28s calls=%6ld total=%10.1f us  avg=%8.2f us\n", kv.first.c_str(), kv.second.second, kv.second.first, kv.second.first / std::max(1L, kv.second.second));
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    rccl_release(c);
    ba_destroy(c);
    pixsel_destroy(c);
    init_destroy(c);
    if (c->copy) (void)hipStreamSynchronize(c->copy);
    for (auto& s : c->slots) {
        if (s.ev_up) (void)hipEventDestroy(s.ev_up);
        for (int l = 0; l < NALO_MAX_LEVELS; ++l) { if (s.I[l]) (void)hipFree(s.I[l]); if (s.dI[l]) (void)hipFree(s.dI[l]); if (s.absg[l]) (void)hipFree(s.absg[l]); }
        if (s.mask) (void)hipFree(s.mask);
        if (s.bgr) (void)hipFree(s.bgr);
        if (s.raw) (void)hipFree(s.raw);
        if (s.dI0t) (void)hipFree(s.dI0t);
    }
    for (int l = 0; l < NALO_MAX_LEVELS; ++l) {
        c->trk_idepth[l].release(); c->trk_wsum[l].release(); c->trk_wbak[l].release();
        c->pc_u[l].release(); c->pc_v[l].release(); c->pc_id[l].release(); c->pc_col[l].release();
    }
    c->dense_lb.release(); c->trk_partial.release(); c->trk_ticket.release(); c->ref_res.release(); c->trk_out.release(); c->lm_partial.release(); c->trk_shard_sums.release(); c->scan_tmp.release(); c->trk_cnt.release(); c->upload_tmp.release();
    if (c->trk_out_host) (void)hipHostFree(c->trk_out_host);
    if (c->pinned_f) (void)hipHostFree(c->pinned_f);
    if (c->imm_host) (void)hipHostFree(c->imm_host);
    c->imm_dev.release(); c->imm_res.release();
    c->und_G.release(); c->und_vinv.release(); c->und_rxy.release(); c->und_raw.release(); c->und_mask.release(); c->und_bgr.release();
    for (auto& kv : c->prof) for (auto& ev : kv.second.pending) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
    for (hipEvent_t e : c->prof_pool) (void)hipEventDestroy(e);
    if (c->ev_main) (void)hipEventDestroy(c->ev_main);
    if (c->gamma_dev) (void)hipFree(c->gamma_dev);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->side) (void)hipStreamDestroy(c->side);
    if (c->copy) (void)hipStreamDestroy(c->copy);
    delete c;
}

// The reference constants this library was compiled with (ref_constants.h: the table the constexpr values of host and device code are generated from),
// the pattern offsets and the defaults of nalo_settings. Needs no device.
int nalo_constants(int cap, const char** names, double* values) {
    static const struct { const char* name; double value; } table[] = {
#define NALO_REF_ROW(type, name, refname, value) {refname, (double)(nalo::name)},
        NALO_REF_CONSTANTS(NALO_REF_ROW)
#undef NALO_REF_ROW
        {"patternP[0].x", (double)nalo::kPatternDx[0]}, {"patternP[0].y", (double)nalo::kPatternDy[0]}, {"patternP[1].x", (double)nalo::kPatternDx[1]}, {"patternP[1].y", (double)nalo::kPatternDy[1]},
        {"patternP[2].x", (double)nalo::kPatternDx[2]}, {"patternP[2].y", (double)nalo::kPatternDy[2]}, {"patternP[3].x", (double)nalo::kPatternDx[3]}, {"patternP[3].y", (double)nalo::kPatternDy[3]},
        {"patternP[4].x", (double)nalo::kPatternDx[4]}, {"patternP[4].y", (double)nalo::kPatternDy[4]}, {"patternP[5].x", (double)nalo::kPatternDx[5]}, {"patternP[5].y", (double)nalo::kPatternDy[5]},
        {"patternP[6].x", (double)nalo::kPatternDx[6]}, {"patternP[6].y", (double)nalo::kPatternDy[6]}, {"patternP[7].x", (double)nalo::kPatternDx[7]}, {"patternP[7].y", (double)nalo::kPatternDy[7]},
    };
    const int n = (int)(sizeof(table) / sizeof(table[0]));
    for (int i = 0; i < n && i < cap; ++i) { if (names) names[i] = table[i].name; if (values) values[i] = table[i].value; }
    return n;
}

// the same table as DEVICE code evaluates it: one lane writes every constant (+ the pattern the kernels index) from inside a kernel
__global__ void nalo_constants_kernel(double* out) {
    int i = 0;
#define NALO_REF_ROW(type, name, refname, value) out[i++] = (double)(nalo::name);
    NALO_REF_CONSTANTS(NALO_REF_ROW)
#undef NALO_REF_ROW
    constexpr int dx[8] = NALO_PATTERN_DX, dy[8] = NALO_PATTERN_DY;
    for (int k = 0; k < 8; ++k) { out[i++] = (double)dx[k]; out[i++] = (double)dy[k]; }
}
int nalo_constants_device(nalo_ctx* c, int cap, double* values) {
    if (!c || !values) return fail(c, NALO_ERR_ARG, "nalo_constants_device: bad argument");
    const int n = nalo_constants(0, nullptr, nullptr);
    NALO_HIP(c, hipSetDevice(c->device));
    double* d = nullptr;
    NALO_HIP(c, hipMalloc((void**)&d, (size_t)n * 8));
    nalo_constants_kernel<<<1, 1, 0, c->stream>>>(d);
    std::vector<double> h(n);
    hipError_t e = hipMemcpyAsync(h.data(), d, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    NALO_HIP(c, e);
    for (int i = 0; i < n && i < cap; ++i) values[i] = h[i];
    return n;
}

const char* nalo_last_error(nalo_ctx* c) { return c ? c->err.c_str() : "null ctx"; }
int nalo_levels(nalo_ctx* c) { return c ? c->levels : NALO_ERR_ARG; }
int nalo_sync(nalo_ctx* c) { if (!c) return NALO_ERR_ARG; if (c->copy) NALO_HIP(c, hipStreamSynchronize(c->copy)); NALO_HIP(c, hipStreamSynchronize(c->stream)); NALO_HIP(c, hipStreamSynchronize(c->side)); return NALO_OK; }
void* nalo_stream(nalo_ctx* c) { return c ? (void*)c->stream : nullptr; }
void* nalo_side_stream(nalo_ctx* c) { return c ? (void*)c->side : nullptr; }

int nalo_frame_upload(nalo_ctx* c, int slot, const float* irradiance, const float* mask, const uint8_t* bgr, const float* gammaB) {
    if (!c || !irradiance || slot < 0 || slot >= (int)c->slots.size()) return fail(c, NALO_ERR_ARG, "nalo_frame_upload: bad argument");
    NALO_HIP(c, hipSetDevice(c->device));
    FrameSlot& s = c->slots[slot];
    const size_t n0 = (size_t)c->w * c->h;
    NALO_HIP(c, hipMemcpyAsync(s.I[0], irradiance, n0 * 4, hipMemcpyHostToDevice, c->stream));
    if (mask) { if (!s.mask) NALO_HIP(c, hipMalloc((void**)&s.mask, n0 * 4)); NALO_HIP(c, hipMemcpyAsync(s.mask, mask, n0 * 4, hipMemcpyHostToDevice, c->stream)); }
    if (bgr) { if (!s.bgr) NALO_HIP(c, hipMalloc((void**)&s.bgr, n0 * 3)); NALO_HIP(c, hipMemcpyAsync(s.bgr, bgr, n0 * 3, hipMemcpyHostToDevice, c->stream)); }
    const float* gdev = nullptr;
    if (gammaB) { NALO_HIP(c, c->upload_tmp.reserve(256)); NALO_HIP(c, hipMemcpyAsync(c->upload_tmp.p, gammaB, 256 * 4, hipMemcpyHostToDevice, c->stream)); gdev = c->upload_tmp.p; }
    int rc = pyramid_build(c, s, gdev);
    if (rc) return rc;
    pixsel_invalidate_hists(c, slot);
    NALO_HIP(c, hipStreamSynchronize(c->stream));      // host buffers are caller-owned: safe to reuse on return
    s.valid = true;
    return NALO_OK;
}

// Asynchronous form of nalo_frame_upload for a per-frame pipeline (FullSystem::addActiveFrame -> makeImages, FullSystem.cpp:1053-1065): the H2D copies run
// on the context's copy stream, under whatever the main stream is executing (the previous frame's tracking), the pyramid kernels are queued on the main
// stream behind an event. Returns at once: the host buffers must stay untouched until nalo_frame_wait(ctx, slot) (or nalo_sync) returns. Truly
// asynchronous only from pinned host memory (nalo_host_alloc / hipHostMalloc); pageable buffers work but are staged by the HIP runtime.
// The gamma table is shared by all slots of the context: another slot's pyramid kernel (main stream) may still be reading it when the next frame arrives. A running
// system passes the same CalibHessian::B every frame, so the table is sent once; a DIFFERENT table is copied only after the copy stream has waited for everything
// queued on the main stream so far.
static int gamma_upload_async(nalo_ctx* c, const float* gammaB) {
    if (!gammaB) return NALO_OK;
    if (c->gamma_have && std::memcmp(c->gamma_last, gammaB, sizeof(c->gamma_last)) == 0) return NALO_OK;
    NALO_HIP(c, hipEventRecord(c->ev_main, c->stream));
    NALO_HIP(c, hipStreamWaitEvent(c->copy, c->ev_main, 0));
    std::memcpy(c->gamma_last, gammaB, sizeof(c->gamma_last));
    NALO_HIP(c, hipMemcpyAsync(c->gamma_dev, c->gamma_last, 256 * 4, hipMemcpyHostToDevice, c->copy));   // from the context's copy: the caller's table may change after the call
    c->gamma_have = true;
    return NALO_OK;
}
int nalo_frame_upload_async(nalo_ctx* c, int slot, const float* irradiance, const float* mask, const uint8_t* bgr, const float* gammaB) {
    if (!c || !irradiance || slot < 0 || slot >= (int)c->slots.size()) return fail(c, NALO_ERR_ARG, "nalo_frame_upload_async: bad argument");
    NALO_HIP(c, hipSetDevice(c->device));
    FrameSlot& s = c->slots[slot];
    if (!c->copy) { NALO_HIP(c, hipStreamCreateWithFlags(&c->copy, hipStreamNonBlocking)); NALO_HIP(c, hipEventCreateWithFlags(&c->ev_main, hipEventDisableTiming)); }
    if (!s.ev_up) NALO_HIP(c, hipEventCreateWithFlags(&s.ev_up, hipEventDisableTiming));
    const size_t n0 = (size_t)c->w * c->h;
    if (mask && !s.mask) NALO_HIP(c, hipMalloc((void**)&s.mask, n0 * 4));
    if (bgr && !s.bgr) NALO_HIP(c, hipMalloc((void**)&s.bgr, n0 * 3));
    if (gammaB && !c->gamma_dev) NALO_HIP(c, hipMalloc((void**)&c->gamma_dev, 256 * 4));
    if (s.valid) {                                                   // kernels already queued on the main stream may still read this slot
        NALO_HIP(c, hipEventRecord(c->ev_main, c->stream));
        NALO_HIP(c, hipStreamWaitEvent(c->copy, c->ev_main, 0));
    }
    NALO_HIP(c, hipMemcpyAsync(s.I[0], irradiance, n0 * 4, hipMemcpyHostToDevice, c->copy));
    if (mask) NALO_HIP(c, hipMemcpyAsync(s.mask, mask, n0 * 4, hipMemcpyHostToDevice, c->copy));
    if (bgr) NALO_HIP(c, hipMemcpyAsync(s.bgr, bgr, n0 * 3, hipMemcpyHostToDevice, c->copy));
    { int rg = gamma_upload_async(c, gammaB); if (rg) return rg; }
    NALO_HIP(c, hipEventRecord(s.ev_up, c->copy));
    NALO_HIP(c, hipStreamWaitEvent(c->stream, s.ev_up, 0));
    int rc = pyramid_build(c, s, gammaB ? c->gamma_dev : nullptr);
    if (rc) return rc;
    pixsel_invalidate_hists(c, slot);
    s.valid = true;
    return NALO_OK;
}
// Undistort / PhotometricUndistorter tables (util/Undistort.cpp:49-190 builds G and vignetteMapInv, :637-1018 remapX / remapY): device resident after this call
int nalo_undist_set(nalo_ctx* c, int wOrg, int hOrg, const float* G, int GDepth, const float* vignetteMapInv, int photometricCalibration, const float* remapX, const float* remapY) {
    if (!c || wOrg <= 1 || hOrg <= 1 || photometricCalibration < 0 || photometricCalibration > 2 || (!remapX) != (!remapY)) return fail(c, NALO_ERR_ARG, "nalo_undist_set: bad argument");
    if (photometricCalibration > 0 && (!G || GDepth < 256)) return fail(c, NALO_ERR_ARG, "nalo_undist_set: the response G needs >= 256 entries (Undistort.cpp:83-87)");
    if (photometricCalibration == 2 && !vignetteMapInv) return fail(c, NALO_ERR_ARG, "nalo_undist_set: photometricCalibration 2 needs the vignette");
    if (!remapX && (wOrg != c->w || hOrg != c->h)) return fail(c, NALO_ERR_ARG, "nalo_undist_set: passthrough needs wOrg x hOrg = w x h");
    NALO_HIP(c, hipSetDevice(c->device));
    const size_t no = (size_t)wOrg * hOrg, n = (size_t)c->w * c->h;
    if (G) { NALO_HIP(c, c->und_G.reserve(GDepth)); NALO_HIP(c, hipMemcpy(c->und_G.p, G, (size_t)GDepth * 4, hipMemcpyHostToDevice)); }
    if (vignetteMapInv) { NALO_HIP(c, c->und_vinv.reserve(no)); NALO_HIP(c, hipMemcpy(c->und_vinv.p, vignetteMapInv, no * 4, hipMemcpyHostToDevice)); }
    if (remapX) {
        // the four taps of every output pixel must lie inside the original image (the reference's makeOptimalK_crop / the remap construction guarantee it:
        // out-of-image entries are -1, Undistort.cpp:998-1010); checked here because a violated table is an out-of-bounds device read
        for (size_t i = 0; i < n; ++i) if (!(remapX[i] < 0) && !(remapX[i] >= 0 && remapY[i] >= 0 && (int)remapX[i] + 1 < wOrg && (int)remapY[i] + 1 < hOrg))
            return fail(c, NALO_ERR_ARG, "nalo_undist_set: remap entry outside the original image");
        std::vector<float> xy(2 * n);
        for (size_t i = 0; i < n; ++i) { xy[2 * i] = remapX[i]; xy[2 * i + 1] = remapY[i]; }
        NALO_HIP(c, c->und_rxy.reserve(2 * n));
        NALO_HIP(c, hipMemcpy(c->und_rxy.p, xy.data(), 2 * n * 4, hipMemcpyHostToDevice));
    }
    c->und_wOrg = wOrg; c->und_hOrg = hOrg; c->und_photometric = photometricCalibration; c->und_GDepth = GDepth; c->und_remap = remapX != nullptr; c->und_vig = vignetteMapInv != nullptr;
    c->und_set = true;
    return NALO_OK;
}

int nalo_frame_upload_raw(nalo_ctx* c, int slot, const void* raw, int bytes_per_px, float exposure_time, float factor, const uint8_t* mask_org, const uint8_t* bgr_org,
                          const float* gammaB) {
    if (!c || !raw || slot < 0 || slot >= (int)c->slots.size() || (bytes_per_px != 1 && bytes_per_px != 2)) return fail(c, NALO_ERR_ARG, "nalo_frame_upload_raw: bad argument");
    if (!c->und_set) return fail(c, NALO_ERR_STATE, "nalo_frame_upload_raw: nalo_undist_set has not run");
    NALO_HIP(c, hipSetDevice(c->device));
    FrameSlot& s = c->slots[slot];
    const size_t no = (size_t)c->und_wOrg * c->und_hOrg, n0 = (size_t)c->w * c->h;
    // processFrame: `if(!valid || exposure_time <= 0 || setting_photometricCalibration==0)` -> data = factor * image_in (:224-231)
    int photometric = c->und_photometric;
    if (exposure_time <= 0) photometric = 0;
    if (photometric > 0 && bytes_per_px == 2 && c->und_GDepth < 65536) return fail(c, NALO_ERR_ARG, "nalo_frame_upload_raw: 16-bit frames index G beyond its depth");
    NALO_HIP(c, c->und_raw.reserve(no * 2));
    NALO_HIP(c, hipMemcpyAsync(c->und_raw.p, raw, no * bytes_per_px, hipMemcpyHostToDevice, c->stream));
    if (mask_org) { NALO_HIP(c, c->und_mask.reserve(no)); NALO_HIP(c, hipMemcpyAsync(c->und_mask.p, mask_org, no, hipMemcpyHostToDevice, c->stream)); if (!s.mask) NALO_HIP(c, hipMalloc((void**)&s.mask, n0 * 4)); }
    if (bgr_org) { NALO_HIP(c, c->und_bgr.reserve(no * 3)); NALO_HIP(c, hipMemcpyAsync(c->und_bgr.p, bgr_org, no * 3, hipMemcpyHostToDevice, c->stream)); if (!s.bgr) NALO_HIP(c, hipMalloc((void**)&s.bgr, n0 * 3)); }
    const float* gdev = nullptr;
    if (gammaB) { NALO_HIP(c, c->upload_tmp.reserve(256)); NALO_HIP(c, hipMemcpyAsync(c->upload_tmp.p, gammaB, 256 * 4, hipMemcpyHostToDevice, c->stream)); gdev = c->upload_tmp.p; }
    int rc = ingest_launch(c, c->stream, c->und_raw.p, bytes_per_px, c->und_wOrg, c->und_hOrg, c->und_G.p, c->und_vig ? c->und_vinv.p : nullptr, c->und_remap ? reinterpret_cast<const float2*>(c->und_rxy.p) : nullptr,
                           photometric, factor, mask_org ? c->und_mask.p : nullptr, bgr_org ? c->und_bgr.p : nullptr, s.I[0], s.mask, s.bgr);
    if (rc) return rc;
    rc = pyramid_build(c, s, gdev);
    if (rc) return rc;
    pixsel_invalidate_hists(c, slot);
    NALO_HIP(c, hipStreamSynchronize(c->stream));
    s.valid = true;
    return NALO_OK;
}

// The per-frame entry of a running pipeline for SENSOR frames: the 1-2 B/px raw image is copied on the copy stream (under the tracking of the previous frame),
// the ingest + pyramid kernels are queued on the main stream behind the copy's event. raw must stay untouched until nalo_frame_wait (pinned memory for a truly
// asynchronous copy). Mask / colour are not taken here (keyframe-only inputs: use nalo_frame_upload_raw for those frames).
int nalo_frame_upload_raw_async(nalo_ctx* c, int slot, const void* raw, int bytes_per_px, float exposure_time, float factor, const float* gammaB) {
    if (!c || !raw || slot < 0 || slot >= (int)c->slots.size() || (bytes_per_px != 1 && bytes_per_px != 2)) return fail(c, NALO_ERR_ARG, "nalo_frame_upload_raw_async: bad argument");
    if (!c->und_set) return fail(c, NALO_ERR_STATE, "nalo_frame_upload_raw_async: nalo_undist_set has not run");
    NALO_HIP(c, hipSetDevice(c->device));
    FrameSlot& s = c->slots[slot];
    const size_t no = (size_t)c->und_wOrg * c->und_hOrg;
    int photometric = c->und_photometric;
    if (exposure_time <= 0) photometric = 0;
    if (photometric > 0 && bytes_per_px == 2 && c->und_GDepth < 65536) return fail(c, NALO_ERR_ARG, "nalo_frame_upload_raw_async: 16-bit frames index G beyond its depth");
    if (!c->copy) { NALO_HIP(c, hipStreamCreateWithFlags(&c->copy, hipStreamNonBlocking)); NALO_HIP(c, hipEventCreateWithFlags(&c->ev_main, hipEventDisableTiming)); }
    if (!s.ev_up) NALO_HIP(c, hipEventCreateWithFlags(&s.ev_up, hipEventDisableTiming));
    if (s.raw_cap < no * 2) { if (s.raw) { NALO_HIP(c, hipStreamSynchronize(c->stream)); (void)hipFree(s.raw); } s.raw = nullptr; s.raw_cap = 0; NALO_HIP(c, hipMalloc((void**)&s.raw, no * 2)); s.raw_cap = no * 2; }
    if (gammaB && !c->gamma_dev) NALO_HIP(c, hipMalloc((void**)&c->gamma_dev, 256 * 4));
    if (s.valid) {                                                   // kernels already queued on the main stream may still read this slot (incl. its raw buffer)
        NALO_HIP(c, hipEventRecord(c->ev_main, c->stream));
        NALO_HIP(c, hipStreamWaitEvent(c->copy, c->ev_main, 0));
    }
    NALO_HIP(c, hipMemcpyAsync(s.raw, raw, no * bytes_per_px, hipMemcpyHostToDevice, c->copy));
    { int rg = gamma_upload_async(c, gammaB); if (rg) return rg; }
    NALO_HIP(c, hipEventRecord(s.ev_up, c->copy));
    NALO_HIP(c, hipStreamWaitEvent(c->stream, s.ev_up, 0));
    int rc = ingest_launch(c, c->stream, s.raw, bytes_per_px, c->und_wOrg, c->und_hOrg, c->und_G.p, c->und_vig ? c->und_vinv.p : nullptr, c->und_remap ? reinterpret_cast<const float2*>(c->und_rxy.p) : nullptr,
                           photometric, factor, nullptr, nullptr, s.I[0], nullptr, nullptr);
    if (rc) return rc;
    rc = pyramid_build(c, s, gammaB ? c->gamma_dev : nullptr);
    if (rc) return rc;
    pixsel_invalidate_hists(c, slot);
    s.valid = true;
    return NALO_OK;
}

int nalo_frame_wait(nalo_ctx* c, int slot) {
    if (!c || slot < 0 || slot >= (int)c->slots.size()) return fail(c, NALO_ERR_ARG, "nalo_frame_wait: bad slot");
    if (c->slots[slot].ev_up) NALO_HIP(c, hipEventSynchronize(c->slots[slot].ev_up));
    return NALO_OK;
}
void* nalo_host_alloc(size_t bytes) { void* p = nullptr; return hipHostMalloc(&p, bytes) == hipSuccess ? p : nullptr; }
void nalo_host_free(void* p) { if (p) (void)hipHostFree(p); }

int nalo_frame_rebuild(nalo_ctx* c, int slot) {
    if (!c || slot < 0 || slot >= (int)c->slots.size() || !c->slots[slot].valid) return fail(c, NALO_ERR_ARG, "nalo_frame_rebuild: bad slot");
    NALO_HIP(c, hipSetDevice(c->device));
    HostTimer ht(c, "frame_rebuild");
    return pyramid_build(c, c->slots[slot], nullptr);
}

int nalo_frame_download(nalo_ctx* c, int slot, int lvl, float* dI3, float* absg) {
    if (!c || slot < 0 || slot >= (int)c->slots.size() || lvl < 0 || lvl >= c->levels) return fail(c, NALO_ERR_ARG, "nalo_frame_download: bad argument");
    FrameSlot& s = c->slots[slot];
    if (!s.valid) return fail(c, NALO_ERR_STATE, "nalo_frame_download: empty slot");
    const size_t npx = (size_t)c->wl[lvl] * c->hl[lvl];
    NALO_HIP(c, hipStreamSynchronize(c->stream));
    if (dI3) {
        std::vector<float4> tmp(npx);
        NALO_HIP(c, hipMemcpy(tmp.data(), s.dI[lvl], npx * 16, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < npx; ++i) { dI3[3 * i] = tmp[i].x; dI3[3 * i + 1] = tmp[i].y; dI3[3 * i + 2] = tmp[i].z; }
    }
    if (absg) NALO_HIP(c, hipMemcpy(absg, s.absg[lvl], npx * 4, hipMemcpyDeviceToHost));
    return NALO_OK;
}

// ------------------------------------------------------------------------------------------------ tracker
int nalo_trk_make_k(nalo_ctx* c, float fx, float fy, float cx, float cy) {
    if (!c) return NALO_ERR_ARG;
    set_pyr_calib(c, fx, fy, cx, cy);
    return NALO_OK;
}

static int upload4(nalo_ctx* c, int n, const float* a, const float* b, const float* d, const float* e, float** dev) {
    // one pinned staging buffer, one H2D copy (four pageable copies cost four blocking round trips); the previous upload has been consumed:
    // trk_build_ref ends on a flag published after it in stream order
    NALO_HIP(c, c->upload_tmp.reserve((size_t)4 * n + 256));
    if (c->pinned_f_cap < (size_t)4 * n) {
        if (c->pinned_f) (void)hipHostFree(c->pinned_f);
        c->pinned_f = nullptr; c->pinned_f_cap = 0;
        NALO_HIP(c, hipHostMalloc((void**)&c->pinned_f, (size_t)4 * n * 4 + 1024));
        c->pinned_f_cap = (size_t)4 * n + 256;
    }
    float* base = c->upload_tmp.p + 256;
    const float* src[4] = {a, b, d, e};
    for (int k = 0; k < 4; ++k) { std::memcpy(c->pinned_f + (size_t)k * n, src[k], (size_t)n * 4); dev[k] = base + (size_t)k * n; }
    NALO_HIP(c, hipMemcpyAsync(base, c->pinned_f, (size_t)4 * n * 4, hipMemcpyHostToDevice, c->stream));
    return NALO_OK;
}

int nalo_trk_set_ref(nalo_ctx* c, int slot_ref, int n, const float* Ku, const float* Kv, const float* new_idepth, const float* HdiF) {
    if (!c || slot_ref < 0 || slot_ref >= (int)c->slots.size() || n < 0 || (n > 0 && (!Ku || !Kv || !new_idepth || !HdiF)))
        return fail(c, NALO_ERR_ARG, "nalo_trk_set_ref: bad argument");
    if (!c->slots[slot_ref].valid) return fail(c, NALO_ERR_STATE, "nalo_trk_set_ref: reference slot has no pyramid");
    NALO_HIP(c, hipSetDevice(c->device));
    HostTimer ht(c, "trk_set_ref");
    c->slot_ref = slot_ref;
    float* dev[4] = {};
    if (n > 0) { int rc = upload4(c, n, Ku, Kv, new_idepth, HdiF, dev); if (rc) return rc; }
    return trk_build_ref(c, n, dev[0], dev[1], dev[2], dev[3]);
}

int nalo_trk_ref_upload(nalo_ctx* c, int n, const float* Ku, const float* Kv, const float* new_idepth, const float* HdiF) {
    if (!c || n < 0 || (n > 0 && (!Ku || !Kv || !new_idepth || !HdiF))) return fail(c, NALO_ERR_ARG, "nalo_trk_ref_upload: bad argument");
    NALO_HIP(c, hipSetDevice(c->device));
    c->ref_res_n = -1;
    if (n > 0) {
        float* dev[4] = {};
        int rc = upload4(c, n, Ku, Kv, new_idepth, HdiF, dev); if (rc) return rc;                  // pinned staging + ONE H2D copy into the shared scratch ...
        NALO_HIP(c, c->ref_res.reserve((size_t)4 * n));
        NALO_HIP(c, hipMemcpyAsync(c->ref_res.p, dev[0], (size_t)4 * n * 4, hipMemcpyDeviceToDevice, c->stream));   // ... and from there into the block that stays
        NALO_HIP(c, hipStreamSynchronize(c->stream));                                              // the staging buffers are free again when this returns
    }
    c->ref_res_n = n;
    return NALO_OK;
}
int nalo_trk_set_ref_resident(nalo_ctx* c, int slot_ref) {
    if (!c || slot_ref < 0 || slot_ref >= (int)c->slots.size()) return fail(c, NALO_ERR_ARG, "nalo_trk_set_ref_resident: bad argument");
    if (!c->slots[slot_ref].valid) return fail(c, NALO_ERR_STATE, "nalo_trk_set_ref_resident: reference slot has no pyramid");
    if (c->ref_res_n < 0) return fail(c, NALO_ERR_STATE, "nalo_trk_set_ref_resident: no resident inputs (nalo_trk_ref_upload)");
    NALO_HIP(c, hipSetDevice(c->device));
    HostTimer ht(c, "trk_set_ref");
    c->slot_ref = slot_ref;
    const size_t n = (size_t)c->ref_res_n;
    const float* b = c->ref_res.p;
    return trk_build_ref(c, c->ref_res_n, b, b + n, b + 2 * n, b + 3 * n);
}

int nalo_trk_set_pc(nalo_ctx* c, int slot_ref, int lvl, int n, const float* u, const float* v, const float* idepth, const float* color) {
    if (!c || slot_ref < 0 || slot_ref >= (int)c->slots.size() || lvl < 0 || lvl >= c->levels || n < 0) return fail(c, NALO_ERR_ARG, "nalo_trk_set_pc: bad argument");
    NALO_HIP(c, hipSetDevice(c->device));
    c->slot_ref = slot_ref;
    const size_t cap = std::max((size_t)n, (size_t)c->wl[lvl] * c->hl[lvl]);
    NALO_HIP(c, c->pc_u[lvl].reserve(cap)); NALO_HIP(c, c->pc_v[lvl].reserve(cap)); NALO_HIP(c, c->pc_id[lvl].reserve(cap)); NALO_HIP(c, c->pc_col[lvl].reserve(cap));
    NALO_HIP(c, hipMemcpy(c->pc_u[lvl].p, u, (size_t)n * 4, hipMemcpyHostToDevice));
    NALO_HIP(c, hipMemcpy(c->pc_v[lvl].p, v, (size_t)n * 4, hipMemcpyHostToDevice));
    NALO_HIP(c, hipMemcpy(c->pc_id[lvl].p, idepth, (size_t)n * 4, hipMemcpyHostToDevice));
    NALO_HIP(c, hipMemcpy(c->pc_col[lvl].p, color, (size_t)n * 4, hipMemcpyHostToDevice));
    c->pc_n[lvl] = n;
    return NALO_OK;
}

// dense=1 glue: append the plane-sampled points of one mask cluster to the level-0 cloud on the device (CoarseTracker.cpp:628-655)
int nalo_trk_append_plane_points(nalo_ctx* c, const float dir[3], float dis_plane, int refMaskColor, const int rect[4], int* n_added) {
    if (!c || !dir || !rect) return fail(c, NALO_ERR_ARG, "nalo_trk_append_plane_points: bad argument");
    if (c->slot_ref < 0 || !c->slots[c->slot_ref].valid || !c->pc_u[0].p) return fail(c, NALO_ERR_STATE, "nalo_trk_append_plane_points: no tracking reference");
    const FrameSlot& s = c->slots[c->slot_ref];
    if (!s.mask) return fail(c, NALO_ERR_STATE, "nalo_trk_append_plane_points: the reference frame was uploaded without a mask");
    const int minx = rect[0], maxx = rect[1], miny = rect[2], maxy = rect[3];
    // `if(maxx>w[0]-1||minx<1||maxy>h[0]-1||miny<1) continue;` and `if (refMaskColor==0) continue;` (:621-630): nothing is appended
    if (n_added) *n_added = 0;
    if (maxx > c->w - 1 || minx < 1 || maxy > c->h - 1 || miny < 1 || refMaskColor == 0 || dis_plane == 0.f) return NALO_OK;
    NALO_HIP(c, hipSetDevice(c->device));
    const int x0 = ((minx + 4) / 5) * 5, y0 = ((miny + 4) / 5) * 5;
    const int nx = x0 < maxx ? (maxx - 1 - x0) / 5 + 1 : 0, ny = y0 < maxy ? (maxy - 1 - y0) / 5 + 1 : 0;
    const int n0 = c->pc_n[0];
    if ((size_t)n0 + 2 + (size_t)nx * ny > c->pc_u[0].cap) return fail(c, NALO_ERR_STATE, "nalo_trk_append_plane_points: the level-0 cloud would outgrow its w*h buffer");
    if (nx == 0 || ny == 0) return NALO_OK;
    NALO_HIP(c, c->scan_tmp.reserve(64));
    int rc = trk_append_plane_launch(c, s.mask, s.dI[0], dir, dis_plane, (float)refMaskColor, x0, nx, y0, ny, n0, c->scan_tmp.p); if (rc) return rc;
    int added = 0;
    NALO_HIP(c, hipMemcpyAsync(&added, c->scan_tmp.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    NALO_HIP(c, hipStreamSynchronize(c->stream));
    c->pc_n[0] = n0 + added;
    if (n_added) *n_added = added;
    return NALO_OK;
}

int nalo_trk_get_pc(nalo_ctx* c, int lvl, int* n, float* u, float* v, float* idepth, float* color) {
    if (!c || lvl < 0 || lvl >= c->levels || !n) return fail(c, NALO_ERR_ARG, "nalo_trk_get_pc: bad argument");
    *n = c->pc_n[lvl];
    NALO_HIP(c, hipStreamSynchronize(c->stream));
    const size_t b = (size_t)c->pc_n[lvl] * 4;
    if (u && b) NALO_HIP(c, hipMemcpy(u, c->pc_u[lvl].p, b, hipMemcpyDeviceToHost));
    if (v && b) NALO_HIP(c, hipMemcpy(v, c->pc_v[lvl].p, b, hipMemcpyDeviceToHost));
    if (idepth && b) NALO_HIP(c, hipMemcpy(idepth, c->pc_id[lvl].p, b, hipMemcpyDeviceToHost));
    if (color && b) NALO_HIP(c, hipMemcpy(color, c->pc_col[lvl].p, b, hipMemcpyDeviceToHost));
    return NALO_OK;
}

int nalo_trk_get_depth(nalo_ctx* c, int lvl, float* idepth, float* wsum) {
    if (!c || lvl < 0 || lvl >= c->levels) return fail(c, NALO_ERR_ARG, "nalo_trk_get_depth: bad argument");
    if (!c->trk_idepth[lvl].p) return fail(c, NALO_ERR_STATE, "nalo_trk_get_depth: no reference set");
    const size_t b = (size_t)c->wl[lvl] * c->hl[lvl] * 4;
    NALO_HIP(c, hipStreamSynchronize(c->stream));
    if (idepth) NALO_HIP(c, hipMemcpy(idepth, c->trk_idepth[lvl].p, b, hipMemcpyDeviceToHost));
    if (wsum) NALO_HIP(c, hipMemcpy(wsum, c->trk_wsum[lvl].p, b, hipMemcpyDeviceToHost));
    return NALO_OK;
}

// SURVEY 8(e), tracker: every rank holds the whole reference (nalo_trk_set_ref is replicated: makeCoarseDepthL0's dilation and normalisation need every point's
// neighbours) and evaluates rank/world of every level's point cloud; the 45 + 7 sums of an evaluation are all-reduced through the hook, so every rank takes the same LM
// step. Worth it for level sizes of ~1e5 points and more (CoarseTracker.cpp:828-885 sums per thread in the reference).
int nalo_trk_set_shard(nalo_ctx* c, int rank, int world, nalo_allreduce_fn hook, void* user, int stream_ordered) {
    if (!c || world < 1 || rank < 0 || rank >= world) return fail(c, NALO_ERR_ARG, "nalo_trk_set_shard: bad argument");
    if (world > 1 && !hook) return fail(c, NALO_ERR_ARG, "nalo_trk_set_shard: a sharded tracker needs the all-reduce hook");
    c->trk_rank = rank; c->trk_world = world; c->trk_hook = world > 1 ? hook : nullptr; c->trk_hook_user = user; c->trk_hook_stream_ordered = stream_ordered != 0;
    return NALO_OK;
}

int nalo_trk_eval(nalo_ctx* c, int slot_new, int lvl, const double R[9], const double t[3], const float affLL[2], float b0,
                  float cutoffTH, int want_gs, double stats6[6], double H[64], double b[8]) {
    if (!c || !R || !t || !affLL || !stats6 || lvl < 0 || lvl >= c->levels || slot_new < 0 || slot_new >= (int)c->slots.size())
        return fail(c, NALO_ERR_ARG, "nalo_trk_eval: bad argument");
    if (c->xchg_failed) return fail(c, NALO_ERR_HIP, "nalo_trk_eval: a cross-rank sum of this context failed earlier; rebuild on a new context");
    if (c->slot_ref < 0 || !c->slots[slot_new].valid) return fail(c, NALO_ERR_STATE, "nalo_trk_eval: no reference / empty frame slot");
    if (want_gs && (!H || !b)) return fail(c, NALO_ERR_ARG, "nalo_trk_eval: H/b required");
    // RKi = R.cast<float>() * Ki[lvl], t = translation.cast<float>() (CoarseTracker.cpp:907-908)
    const float fx = c->fx[lvl], fy = c->fy[lvl], cx = c->cx[lvl], cy = c->cy[lvl];
    const float Ki[9] = {1.0f / fx, 0, -cx / fx, 0, 1.0f / fy, -cy / fy, 0, 0, 1};
    float Rf[9], RKi[9], tf[3];
    for (int i = 0; i < 9; ++i) Rf[i] = (float)R[i];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) RKi[i * 3 + j] = Rf[i * 3] * Ki[j] + Rf[i * 3 + 1] * Ki[3 + j] + Rf[i * 3 + 2] * Ki[6 + j];
    for (int i = 0; i < 3; ++i) tf[i] = (float)t[i];
    const float maxEnergy = 2 * kHuberTH * cutoffTH - kHuberTH * kHuberTH;          // :916
    double o[64];
    int rc = trk_eval_launch(c, slot_new, lvl, RKi, tf, Ki, affLL[0], affLL[1], b0, cutoffTH, maxEnergy, o);
    if (rc) return rc;
    const double E = o[45], nE = o[46], nSat = o[47], nW = o[48], sT = o[49], sRT = o[50], sN = o[51];
    stats6[0] = E; stats6[1] = nE; stats6[2] = sT / (sN + 0.1); stats6[3] = 0; stats6[4] = sRT / (sN + 0.1);
    stats6[5] = (double)((float)nSat / (float)nE);
    if (want_gs) {
        const double npad = (double)(((long)nW + 3) & ~3L);          // divided by the padded count
        const double inv = 1.0 / npad;
        static const double sc[8] = {kScaleXiRot, kScaleXiRot, kScaleXiRot, kScaleXiTrans, kScaleXiTrans, kScaleXiTrans, kScaleA, kScaleB};
        double Hf[81]; int k = 0;
        for (int r = 0; r < 9; ++r) for (int cc = r; cc < 9; ++cc) { Hf[r * 9 + cc] = Hf[cc * 9 + r] = o[k]; ++k; }
        for (int r = 0; r < 8; ++r) { for (int cc = 0; cc < 8; ++cc) H[r * 8 + cc] = Hf[r * 9 + cc] * inv * sc[r] * sc[cc]; b[r] = Hf[r * 9 + 8] * inv * sc[r]; }
    }
    return NALO_OK;
}

// evaluations per pyramid level and point-cloud sizes of the last nalo_trk_track that ran in the persistent kernel: the algorithmic bytes of that launch are
// sum_l evals[l] * n[l] * 64 (16 B point + four 12-B taps per point and evaluation)
int nalo_trk_last_evals(nalo_ctx* c, int evals[5], int n[5]) {
    if (!c) return NALO_ERR_ARG;
    for (int i = 0; i < 5; ++i) { if (evals) evals[i] = c->lm_evals_lvl[i]; if (n) n[i] = i < c->levels ? c->pc_n[i] : 0; }
    return NALO_OK;
}
int nalo_trk_track(nalo_ctx* c, int slot_new, double T_io[12], double aff_io[2], const double ref_aff[2], const float exposures[2],
                   int coarsestLvl, const double minResForAbort[5], double lastResiduals[5], double lastFlow[3], int* ok, int* n_evals) {
    if (!c || !T_io || !aff_io || !ref_aff || !exposures || !ok) return fail(c, NALO_ERR_ARG, "nalo_trk_track: bad argument");
    if (!(coarsestLvl >= 0 && coarsestLvl < 5 && coarsestLvl < c->levels)) return fail(c, NALO_ERR_ARG, "nalo_trk_track: coarsestLvl out of range");
    if (slot_new < 0 || slot_new >= (int)c->slots.size()) return fail(c, NALO_ERR_ARG, "nalo_trk_track: frame slot out of range");
    if (c->slot_ref < 0 || !c->slots[slot_new].valid) return fail(c, NALO_ERR_STATE, "nalo_trk_track: no reference (nalo_trk_set_ref) / empty frame slot");
    if (c->xchg_failed) return fail(c, NALO_ERR_HIP, "nalo_trk_track: a cross-rank sum of this context failed earlier; rebuild on a new context");
    HostTimer ht(c, "trk_track");   // assert at :1083
    double lastRes[5] = {NAN, NAN, NAN, NAN, NAN}, flow[3] = {1000, 1000, 1000};
    SE3 cur = SE3::from(T_io);
    double aff_cur[2] = {aff_io[0], aff_io[1]};
    bool haveRepeated = false, good = true;
    int evals = 0, start_lvl = coarsestLvl;
    // The whole pyramid descent runs in ONE persistent multi-block kernel (kernels_trk_lm.hip): ~10 us per LM evaluation against ~19 us for
    // the host-driven loop below (a launch, a finish kernel and a polled flag per evaluation). NALO_TRK_HOST_LM=1 selects the host loop,
    // which is also what a caller gets by driving nalo_trk_eval itself.
    {
        // a fixed affine parameter changes the system the LM solves (:1140-1162): those variants live in the host loop below
        static const bool env_host = std::getenv("NALO_TRK_HOST_LM") != nullptr;
        const bool sharded = c->trk_world > 1 && c->trk_hook;          // the persistent kernel cannot exchange sums with other GPUs: a sharded tracker runs the host-driven loop
        const bool force_host = env_host || sharded || c->lm_host_only || c->set.affineOptModeA < 0 || c->set.affineOptModeB < 0;
        const int stop = 0;                                            // levels coarsestLvl..0 on the device
        if (!force_host) {
            if (c->slot_ref < 0 || slot_new < 0 || slot_new >= (int)c->slots.size() || !c->slots[slot_new].valid)
                return fail(c, NALO_ERR_STATE, "nalo_trk_track: no reference / empty frame slot");
            double o[32];
            int rc = trk_lm_launch(c, slot_new, T_io, aff_io, ref_aff, exposures, coarsestLvl, stop, minResForAbort, o);
            if (rc == NALO_LM_LOST_BLOCK) {
                // the persistent kernel's workgroups were not co-resident (another context holds the CUs): nothing was written back, so the frame is
                // redone by the host-driven loop below (same kernels per evaluation, same results to 1e-5), and so is every later frame of this context
                c->lm_host_only = true;
                std::fprintf(stderr, "[nalo] trk_lm_kernel: a workgroup's partial never arrived; this context now drives the tracker's LM loop from the host\n");
            } else {
            if (rc) return rc;
            for (int l = stop; l <= coarsestLvl && l < 5; ++l) lastRes[l] = o[14 + l];
            flow[0] = o[19]; flow[1] = o[20]; flow[2] = o[21];
            evals = (int)o[23];
            if ((int)o[22] == 0) {                                          // aborted on minResForAbort (:1227): outputs untouched
                if (lastResiduals) std::memcpy(lastResiduals, lastRes, sizeof(lastRes));
                if (lastFlow) std::memcpy(lastFlow, flow, sizeof(flow));
                if (n_evals) *n_evals = evals;
                *ok = 0;
                return NALO_OK;
            }
            cur = SE3::from(o); aff_cur[0] = o[12]; aff_cur[1] = o[13];
            haveRepeated = o[25] != 0.0;
            start_lvl = (int)o[24];                                         // next level to process (-1: pyramid finished)
            }
        }
    }
    static const int maxIterations[5] = {10, 20, 50, 50, 50};
    const float lambdaExtrapolationLimit = 0.001f;
    auto eval = [&](int lvl, const SE3& T, const double aff[2], float cutoff, double st[6], double* H, double* b, double* aLL0) -> int {
        double aLL[2];
        aff_from_to(exposures[0], exposures[1], ref_aff[0], ref_aff[1], aff[0], aff[1], aLL);
        const float aLLf[2] = {(float)aLL[0], (float)aLL[1]};
        const double R[9] = {T.R(0, 0), T.R(0, 1), T.R(0, 2), T.R(1, 0), T.R(1, 1), T.R(1, 2), T.R(2, 0), T.R(2, 1), T.R(2, 2)};
        const double tt[3] = {T.t(0), T.t(1), T.t(2)};
        if (aLL0) *aLL0 = aLL[0];
        ++evals;
        return nalo_trk_eval(c, slot_new, lvl, R, tt, aLLf, (float)ref_aff[1], cutoff, 1, st, H, b);
    };
    for (int lvl = start_lvl; lvl >= 0; --lvl) {
        double H[64], b[8], Hn[64], bn[8], resOld[6], resNew[6];
        float levelCutoffRepeat = 1;
        int rc = eval(lvl, cur, aff_cur, kCoarseCutoffTH * levelCutoffRepeat, resOld, H, b, nullptr);
        if (rc) return rc;
        while (resOld[5] > 0.6 && levelCutoffRepeat < 50) {
            levelCutoffRepeat *= 2;
            rc = eval(lvl, cur, aff_cur, kCoarseCutoffTH * levelCutoffRepeat, resOld, H, b, nullptr);
            if (rc) return rc;
        }
        float lambda = 0.01f;
        for (int it = 0; it < maxIterations[lvl]; ++it) {
            double Hl[64], nb[8], inc[8];
            std::memcpy(Hl, H, sizeof(Hl));
            for (int i = 0; i < 8; ++i) { Hl[i * 8 + i] *= (1 + lambda); nb[i] = -b[i]; }
            ldlt_solve(8, Hl, nb, inc);
            {                                                                  // :1140-1162: a and/or b fixed
                const bool fa = c->set.affineOptModeA < 0, fb = c->set.affineOptModeB < 0;
                if (fa && fb) { double H6[36], x6[6]; for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) H6[i * 6 + j] = Hl[i * 8 + j];
                    ldlt_solve(6, H6, nb, x6); for (int i = 0; i < 6; ++i) inc[i] = x6[i]; inc[6] = inc[7] = 0; }
                if (!fa && fb) { double H7[49], x7[7]; for (int i = 0; i < 7; ++i) for (int j = 0; j < 7; ++j) H7[i * 7 + j] = Hl[i * 8 + j];
                    ldlt_solve(7, H7, nb, x7); for (int i = 0; i < 7; ++i) inc[i] = x7[i]; inc[7] = 0; }
                if (fa && !fb) {                                               // b takes a's slot: column 6 = column 7, then row 6 = row 7
                    double Hs[64], bs[8], H7[49], x7[7]; std::memcpy(Hs, Hl, sizeof(Hs)); std::memcpy(bs, nb, sizeof(bs));
                    for (int i = 0; i < 8; ++i) Hs[i * 8 + 6] = Hs[i * 8 + 7];
                    for (int j = 0; j < 8; ++j) Hs[6 * 8 + j] = Hs[7 * 8 + j];
                    bs[6] = bs[7];
                    for (int i = 0; i < 7; ++i) for (int j = 0; j < 7; ++j) H7[i * 7 + j] = Hs[i * 8 + j];
                    ldlt_solve(7, H7, bs, x7); for (int i = 0; i < 6; ++i) inc[i] = x7[i]; inc[6] = 0; inc[7] = x7[6];
                }
            }
            float extrapFac = 1;
            if (lambda < lambdaExtrapolationLimit) extrapFac = std::sqrt(std::sqrt(lambdaExtrapolationLimit / lambda));
            for (double& v : inc) v *= extrapFac;
            double incS[8];
            std::memcpy(incS, inc, sizeof(incS));
            for (int i = 0; i < 3; ++i) incS[i] *= kScaleXiRot;      // labels swapped vs. tangent order in the reference (:1172-1173)
            for (int i = 3; i < 6; ++i) incS[i] *= kScaleXiTrans;
            incS[6] *= kScaleA; incS[7] *= kScaleB;
            double s = 0; for (double v : incS) s += v;
            if (!std::isfinite(s)) std::memset(incS, 0, sizeof(incS));
            const SE3 Tn = se3_exp(incS) * cur;
            const double aff_new[2] = {aff_cur[0] + incS[6], aff_cur[1] + incS[7]};
            rc = eval(lvl, Tn, aff_new, kCoarseCutoffTH * levelCutoffRepeat, resNew, Hn, bn, nullptr);
            if (rc) return rc;
            const bool accept = (resNew[0] / resNew[1]) < (resOld[0] / resOld[1]);
            if (accept) {
                std::memcpy(H, Hn, sizeof(H)); std::memcpy(b, bn, sizeof(b)); std::memcpy(resOld, resNew, sizeof(resOld));
                aff_cur[0] = aff_new[0]; aff_cur[1] = aff_new[1]; cur = Tn;
                lambda *= 0.5f;
            } else { lambda *= 4; if (lambda < lambdaExtrapolationLimit) lambda = lambdaExtrapolationLimit; }
            double nrm = 0; for (double v : inc) nrm += v * v;
            if (!(std::sqrt(nrm) > 1e-3)) break;
        }
        lastRes[lvl] = std::sqrt((float)(resOld[0] / resOld[1]));
        flow[0] = resOld[2]; flow[1] = resOld[3]; flow[2] = resOld[4];
        if (minResForAbort && lastRes[lvl] > 1.5 * minResForAbort[lvl]) { good = false; break; }
        if (levelCutoffRepeat > 1 && !haveRepeated) { lvl++; haveRepeated = true; }
    }
    if (lastResiduals) std::memcpy(lastResiduals, lastRes, sizeof(lastRes));
    if (lastFlow) std::memcpy(lastFlow, flow, sizeof(flow));
    if (n_evals) *n_evals = evals;
    if (!good) { *ok = 0; return NALO_OK; }
    std::memcpy(T_io, cur.m, sizeof(cur.m)); aff_io[0] = aff_cur[0]; aff_io[1] = aff_cur[1];
    {                                                                          // :1243-1256
        const double mA = c->set.affineOptModeA, mB = c->set.affineOptModeB;
        bool fine = !((mA != 0 && std::fabs((float)aff_io[0]) > 1.2f) || (mB != 0 && std::fabs((float)aff_io[1]) > 200.f));
        double rel[2]; aff_from_to(exposures[0], exposures[1], ref_aff[0], ref_aff[1], aff_io[0], aff_io[1], rel);
        if ((mA == 0 && std::fabs(logf((float)rel[0])) > 1.5f) || (mB == 0 && std::fabs((float)rel[1]) > 200.f)) fine = false;
        if (fine) { if (mA < 0) aff_io[0] = 0; if (mB < 0) aff_io[1] = 0; }
        *ok = fine;
    }
    return NALO_OK;
}

// ------------------------------------------------------------------------------------------------ immature points
int nalo_imm_create(nalo_ctx* c, int slot_host, int n, const int* u, const int* v, float* color, float* weights, float* gradH, float* energyTH) {
    if (!c || n < 0 || (n > 0 && (!u || !v || !color || !weights || !gradH || !energyTH))) return fail(c, NALO_ERR_ARG, "nalo_imm_create: bad argument");
    if (slot_host < 0 || slot_host >= (int)c->slots.size() || !c->slots[slot_host].valid) return fail(c, NALO_ERR_STATE, "nalo_imm_create: host slot has no pyramid");
    if (n == 0) return NALO_OK;
    NALO_HIP(c, hipSetDevice(c->device));
    for (int i = 0; i < n; ++i) if (u[i] < 2 || v[i] < 2 || u[i] >= c->w - 3 || v[i] >= c->h - 3) return fail(c, NALO_ERR_ARG, "nalo_imm_create: pattern leaves the image");
    // words: u(n) v(n) | color(8n) weights(8n) gradH(3n) energyTH(n)
    const size_t N = (size_t)n;
    int rc = imm_stage(c, 22 * N); if (rc) return rc;
    std::memcpy(c->imm_host, u, N * 4); std::memcpy(c->imm_host + N, v, N * 4);
    float* d = c->imm_dev.p;
    NALO_HIP(c, hipMemcpyAsync(d, c->imm_host, 2 * N * 4, hipMemcpyHostToDevice, c->stream));
    rc = imm_create_launch(c, c->slots[slot_host].dI[0], n, (const int*)d, (const int*)(d + N), d + 2 * N, d + 10 * N, d + 18 * N, d + 21 * N);
    if (rc) return rc;
    NALO_HIP(c, hipMemcpyAsync(c->imm_host + 2 * N, d + 2 * N, 20 * N * 4, hipMemcpyDeviceToHost, c->stream));
    NALO_HIP(c, hipStreamSynchronize(c->stream));
    std::memcpy(color, c->imm_host + 2 * N, 8 * N * 4); std::memcpy(weights, c->imm_host + 10 * N, 8 * N * 4);
    std::memcpy(gradH, c->imm_host + 18 * N, 3 * N * 4); std::memcpy(energyTH, c->imm_host + 21 * N, N * 4);
    return NALO_OK;
}

int nalo_imm_trace(nalo_ctx* c, int slot_new, int n, const float* u, const float* v, const float* color, const float* weights, const float* gradH,
                   const float* energyTH, const int* host_idx, int nh, const float* KRKi, const float* Kt, const float* aff,
                   float* idepth_min, float* idepth_max, int* status, float* quality, float* lastTraceUV, float* lastTracePixelInterval) {
    if (!c || n < 0 || nh < 0 || (n > 0 && (!u || !v || !color || !weights || !gradH || !energyTH || !host_idx || !KRKi || !Kt || !aff || !idepth_min || !idepth_max ||
                                             !status || !quality || !lastTraceUV || !lastTracePixelInterval)))
        return fail(c, NALO_ERR_ARG, "nalo_imm_trace: bad argument");
    if (slot_new < 0 || slot_new >= (int)c->slots.size() || !c->slots[slot_new].valid) return fail(c, NALO_ERR_STATE, "nalo_imm_trace: frame slot has no pyramid");
    if (n == 0) return NALO_OK;
    for (int i = 0; i < n; ++i) if (host_idx[i] < 0 || host_idx[i] >= nh) return fail(c, NALO_ERR_ARG, "nalo_imm_trace: host_idx out of range");
    NALO_HIP(c, hipSetDevice(c->device));
    HostTimer ht(c, "imm_trace");
    // words: [0,22n) u v color weights gradH energyTH | [22n,23n) host_idx | [23n,30n) idmin idmax status quality lastUV(2) lastInterval | [30n, +14nh) KRKi Kt aff
    const size_t N = (size_t)n, H = (size_t)nh;
    int rc = imm_stage(c, 30 * N + 14 * H); if (rc) return rc;
    float* hst = c->imm_host;
    std::memcpy(hst, u, N * 4); std::memcpy(hst + N, v, N * 4); std::memcpy(hst + 2 * N, color, 8 * N * 4); std::memcpy(hst + 10 * N, weights, 8 * N * 4);
    std::memcpy(hst + 18 * N, gradH, 3 * N * 4); std::memcpy(hst + 21 * N, energyTH, N * 4); std::memcpy(hst + 22 * N, host_idx, N * 4);
    std::memcpy(hst + 23 * N, idepth_min, N * 4); std::memcpy(hst + 24 * N, idepth_max, N * 4); std::memcpy(hst + 25 * N, status, N * 4); std::memcpy(hst + 26 * N, quality, N * 4);
    std::memcpy(hst + 27 * N, lastTraceUV, 2 * N * 4); std::memcpy(hst + 29 * N, lastTracePixelInterval, N * 4);     // untouched entries keep the caller's values
    std::memcpy(hst + 30 * N, KRKi, 9 * H * 4); std::memcpy(hst + 30 * N + 9 * H, Kt, 3 * H * 4); std::memcpy(hst + 30 * N + 12 * H, aff, 2 * H * 4);
    float* d = c->imm_dev.p;
    NALO_HIP(c, hipMemcpyAsync(d, hst, (30 * N + 14 * H) * 4, hipMemcpyHostToDevice, c->stream));
    rc = imm_trace_launch(c, c->slots[slot_new].dI[0], n, d, (const int*)(d + 22 * N), d + 30 * N, d + 30 * N + 9 * H, d + 30 * N + 12 * H,
                          d + 23 * N, d + 24 * N, (int*)(d + 25 * N), d + 26 * N, d + 27 * N, d + 29 * N);
    if (rc) return rc;
    NALO_HIP(c, hipMemcpyAsync(hst + 23 * N, d + 23 * N, 7 * N * 4, hipMemcpyDeviceToHost, c->stream));
    NALO_HIP(c, hipStreamSynchronize(c->stream));
    std::memcpy(idepth_min, hst + 23 * N, N * 4); std::memcpy(idepth_max, hst + 24 * N, N * 4); std::memcpy(status, hst + 25 * N, N * 4); std::memcpy(quality, hst + 26 * N, N * 4);
    std::memcpy(lastTraceUV, hst + 27 * N, 2 * N * 4); std::memcpy(lastTracePixelInterval, hst + 29 * N, N * 4);
    return NALO_OK;
}

// Device-resident immature points: the set is uploaded when it changes (makeNewTraces / activation, once per keyframe), every new frame then only sends
// its 14 floats per host frame and the trace updates the state in place; the state comes back when the host wants it (activatePointsMT).
int nalo_imm_resident_set(nalo_ctx* c, int n, const float* u, const float* v, const float* color, const float* weights, const float* gradH, const float* energyTH,
                          const int* host_idx, const float* idepth_min, const float* idepth_max, const int* status, const float* quality) {
    if (!c || n < 0 || (n > 0 && (!u || !v || !color || !weights || !gradH || !energyTH || !host_idx || !idepth_min || !idepth_max || !status || !quality)))
        return fail(c, NALO_ERR_ARG, "nalo_imm_resident_set: bad argument");
    NALO_HIP(c, hipSetDevice(c->device));
    c->imm_res_n = n; c->imm_res_maxhost = -1;
    if (n == 0) return NALO_OK;
    const size_t N = (size_t)n;
    int rc = imm_stage(c, 30 * N); if (rc) return rc;
    NALO_HIP(c, c->imm_res.reserve(30 * N + 256));
    float* hst = c->imm_host;
    for (int i = 0; i < n; ++i) { if (host_idx[i] < 0) return fail(c, NALO_ERR_ARG, "nalo_imm_resident_set: negative host_idx"); c->imm_res_maxhost = std::max(c->imm_res_maxhost, host_idx[i]); }
    std::memcpy(hst, u, N * 4); std::memcpy(hst + N, v, N * 4); std::memcpy(hst + 2 * N, color, 8 * N * 4); std::memcpy(hst + 10 * N, weights, 8 * N * 4);
    std::memcpy(hst + 18 * N, gradH, 3 * N * 4); std::memcpy(hst + 21 * N, energyTH, N * 4); std::memcpy(hst + 22 * N, host_idx, N * 4);
    std::memcpy(hst + 23 * N, idepth_min, N * 4); std::memcpy(hst + 24 * N, idepth_max, N * 4); std::memcpy(hst + 25 * N, status, N * 4); std::memcpy(hst + 26 * N, quality, N * 4);
    for (size_t i = 0; i < 2 * N; ++i) hst[27 * N + i] = -1.f;                // ImmaturePoint ctor: lastTraceUV = (-1,-1), lastTracePixelInterval = 0
    std::memset(hst + 29 * N, 0, N * 4);
    NALO_HIP(c, hipMemcpyAsync(c->imm_res.p, hst, 30 * N * 4, hipMemcpyHostToDevice, c->stream));
    NALO_HIP(c, hipStreamSynchronize(c->stream));                             // the staging buffer is shared with the other immature-point calls
    return NALO_OK;
}
int nalo_imm_resident_trace(nalo_ctx* c, int slot_new, int nh, const float* KRKi, const float* Kt, const float* aff) {
    if (!c || nh < 1 || nh > 16 || !KRKi || !Kt || !aff) return fail(c, NALO_ERR_ARG, "nalo_imm_resident_trace: bad argument");
    if (slot_new < 0 || slot_new >= (int)c->slots.size() || !c->slots[slot_new].valid) return fail(c, NALO_ERR_STATE, "nalo_imm_resident_trace: frame slot has no pyramid");
    if (c->imm_res_n == 0) return NALO_OK;
    if (c->imm_res_maxhost >= nh) return fail(c, NALO_ERR_ARG, "nalo_imm_resident_trace: a resident point's host_idx is out of range");
    NALO_HIP(c, hipSetDevice(c->device));
    HostTimer ht(c, "imm_resident_trace");
    const size_t N = (size_t)c->imm_res_n, H = (size_t)nh;
    float hk[224];
    std::memcpy(hk, KRKi, 9 * H * 4); std::memcpy(hk + 9 * H, Kt, 3 * H * 4); std::memcpy(hk + 12 * H, aff, 2 * H * 4);
    float* d = c->imm_res.p;
    int rc = imm_put_launch(c, d + 30 * N, hk, (int)(14 * H)); if (rc) return rc;
    return imm_trace_launch(c, c->slots[slot_new].dI[0], c->imm_res_n, d, (const int*)(d + 22 * N), d + 30 * N, d + 30 * N + 9 * H, d + 30 * N + 12 * H,
                            d + 23 * N, d + 24 * N, (int*)(d + 25 * N), d + 26 * N, d + 27 * N, d + 29 * N);
}
int nalo_imm_resident_get(nalo_ctx* c, float* idepth_min, float* idepth_max, int* status, float* quality, float* lastTraceUV, float* lastTracePixelInterval) {
    if (!c) return NALO_ERR_ARG;
    if (c->imm_res_n == 0) return NALO_OK;
    if (!idepth_min || !idepth_max || !status || !quality) return fail(c, NALO_ERR_ARG, "nalo_imm_resident_get: bad argument");
    NALO_HIP(c, hipSetDevice(c->device));
    const size_t N = (size_t)c->imm_res_n;
    int rc = imm_stage(c, 7 * N); if (rc) return rc;
    float* hst = c->imm_host;
    NALO_HIP(c, hipMemcpyAsync(hst, c->imm_res.p + 23 * N, 7 * N * 4, hipMemcpyDeviceToHost, c->stream));
    NALO_HIP(c, hipStreamSynchronize(c->stream));
    std::memcpy(idepth_min, hst, N * 4); std::memcpy(idepth_max, hst + N, N * 4); std::memcpy(status, hst + 2 * N, N * 4); std::memcpy(quality, hst + 3 * N, N * 4);
    if (lastTraceUV) std::memcpy(lastTraceUV, hst + 4 * N, 2 * N * 4);
    if (lastTracePixelInterval) std::memcpy(lastTracePixelInterval, hst + 6 * N, N * 4);
    return NALO_OK;
}

// ------------------------------------------------------------------------------------------------ two-frame initialiser
int nalo_init_calc_res_and_gs(nalo_ctx* c, int slot_first, int slot_new, int lvl, int n, const float* u, const float* v, const float* idepth_new, const float* iR,
                              const uint8_t* isGood, const float* energy, const float* outlierTH, const double refToNew[12], const double aff[2],
                              float alphaW, float alphaK, float couplingWeight,
                              uint8_t* isGood_new, float* energy_new, float* maxstep, float* lastHessian_new, float* JbBuffer_new,
                              double* H_out, double* b_out, double* H_out_sc, double* b_out_sc, double E3[3]) {
    if (!c || n < 0 || !refToNew || !aff || !H_out || !b_out || !H_out_sc || !b_out_sc || !E3 ||
        (n > 0 && (!u || !v || !idepth_new || !iR || !isGood || !energy || !outlierTH || !isGood_new || !energy_new || !maxstep || !lastHessian_new || !JbBuffer_new)))
        return fail(c, NALO_ERR_ARG, "nalo_init_calc_res_and_gs: bad argument");
    if (lvl < 0 || lvl >= c->levels) return fail(c, NALO_ERR_ARG, "nalo_init_calc_res_and_gs: level out of range");
    for (int s : {slot_first, slot_new}) if (s < 0 || s >= (int)c->slots.size() || !c->slots[s].valid) return fail(c, NALO_ERR_STATE, "nalo_init_calc_res_and_gs: frame slot has no pyramid");
    NALO_HIP(c, hipSetDevice(c->device));
    const SE3 T = SE3::from(refToNew);
    InitParams P; InitPose X;
    init_pose_setup(c, lvl, n, T, aff, alphaW, alphaK, couplingWeight, P, X);
    double sums[96] = {};
    if (n > 0) {
        // words in: [u | v | idepth_new | iR | energy(2n) | outlierTH | isGood bytes]; out: [energy_new(2n) | maxstep | lastHessian_new | Jb(10n) | isGood_new bytes]; then 96 doubles
        const size_t N = (size_t)n, NB = (N + 3) / 4, in_w = 7 * N + NB, out_w = 14 * N + NB, tot = in_w + out_w + 2 * 96 + 2;
        int rc = imm_stage(c, tot); if (rc) return rc;
        float* hst = c->imm_host;
        std::memcpy(hst, u, N * 4); std::memcpy(hst + N, v, N * 4); std::memcpy(hst + 2 * N, idepth_new, N * 4); std::memcpy(hst + 3 * N, iR, N * 4);
        std::memcpy(hst + 4 * N, energy, 2 * N * 4); std::memcpy(hst + 6 * N, outlierTH, N * 4); std::memcpy(hst + 7 * N, isGood, N);
        float* ho = hst + in_w;
        std::memcpy(ho + 3 * N, lastHessian_new, N * 4); std::memcpy(ho + 4 * N, JbBuffer_new, 10 * N * 4);          // in/out members
        float* d = c->imm_dev.p;
        const size_t sums_off = (in_w + out_w + 1) & ~(size_t)1;                                                      // 8-byte aligned
        NALO_HIP(c, hipMemcpyAsync(d, hst, (in_w + out_w) * 4, hipMemcpyHostToDevice, c->stream));
        P.colorRef = c->slots[slot_first].dI[lvl]; P.colorNew = c->slots[slot_new].dI[lvl];
        P.u = d; P.v = d + N; P.idepth = nullptr; P.idepth_new = d + 2 * N; P.iR = d + 3 * N; P.energy = d + 4 * N; P.outlierTH = d + 6 * N; P.isGood = (const uint8_t*)(d + 7 * N);
        float* outw = d + in_w;
        P.energy_new = outw; P.maxstep = outw + 2 * N; P.lastHessian_new = outw + 3 * N; P.Jb = outw + 4 * N; P.isGood_new = (uint8_t*)(outw + 14 * N);
        rc = init_calc_launch(c, P, lvl, (double*)(d + sums_off));
        if (rc) return rc;
        NALO_HIP(c, hipMemcpyAsync(ho, d + in_w, (sums_off - in_w + 2 * 96) * 4, hipMemcpyDeviceToHost, c->stream));
        NALO_HIP(c, hipStreamSynchronize(c->stream));
        std::memcpy(energy_new, ho, 2 * N * 4); std::memcpy(maxstep, ho + 2 * N, N * 4); std::memcpy(lastHessian_new, ho + 3 * N, N * 4);
        std::memcpy(JbBuffer_new, ho + 4 * N, 10 * N * 4); std::memcpy(isGood_new, ho + 14 * N, N);
        std::memcpy(sums, hst + sums_off, 91 * 8);
    }
    init_sums_to_system(sums, T, n, P, X, H_out, b_out, H_out_sc, b_out_sc, E3);
    return NALO_OK;
}

int nalo_init_do_step(nalo_ctx* c, int n, const uint8_t* isGood, const float* JbBuffer, const float* maxstep, const float* idepth, float lambda, const float inc[8], float* idepth_new) {
    if (!c || n < 0 || !inc || (n > 0 && (!isGood || !JbBuffer || !maxstep || !idepth || !idepth_new))) return fail(c, NALO_ERR_ARG, "nalo_init_do_step: bad argument");
    if (n == 0) return NALO_OK;
    NALO_HIP(c, hipSetDevice(c->device));
    // words: [Jb(10n) | maxstep | idepth | idepth_new (in/out) | isGood bytes]
    const size_t N = (size_t)n, tot = 13 * N + (N + 3) / 4;
    int rc = imm_stage(c, tot); if (rc) return rc;
    float* hst = c->imm_host;
    std::memcpy(hst, JbBuffer, 10 * N * 4); std::memcpy(hst + 10 * N, maxstep, N * 4); std::memcpy(hst + 11 * N, idepth, N * 4); std::memcpy(hst + 12 * N, idepth_new, N * 4);
    std::memcpy(hst + 13 * N, isGood, N);
    float* d = c->imm_dev.p;
    NALO_HIP(c, hipMemcpyAsync(d, hst, tot * 4, hipMemcpyHostToDevice, c->stream));
    rc = init_do_step_launch(c, n, (const uint8_t*)(d + 13 * N), d, d + 10 * N, d + 11 * N, lambda, inc, d + 12 * N);
    if (rc) return rc;
    NALO_HIP(c, hipMemcpyAsync(hst + 12 * N, d + 12 * N, N * 4, hipMemcpyDeviceToHost, c->stream));
    NALO_HIP(c, hipStreamSynchronize(c->stream));
    std::memcpy(idepth_new, hst + 12 * N, N * 4);
    return NALO_OK;
}

// ------------------------------------------------------------------------------------------------ profiling
static void prof_drain(nalo_ctx* c) {
    for (auto& kv : c->prof) {
        for (auto& ev : kv.second.pending) {
            float ms = 0;
            if (hipEventSynchronize(ev.second) == hipSuccess && hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) { kv.second.ms += ms; kv.second.n++; if (kv.second.samples.size() < (size_t)1 << 20) kv.second.samples.push_back(ms); }
            c->prof_pool.push_back(ev.first); c->prof_pool.push_back(ev.second);
        }
        kv.second.pending.clear();
    }
}
int nalo_profile_enable(nalo_ctx* c, int on) {
    if (!c) return NALO_ERR_ARG;
    c->prof_on = on != 0;
    if (on) while (c->prof_pool.size() < 4096) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) break; c->prof_pool.push_back(e); }
    return NALO_OK;
}
int nalo_profile_select(nalo_ctx* c, const char* kernel) { if (!c) return NALO_ERR_ARG; c->prof_only = kernel ? kernel : ""; return NALO_OK; }
int nalo_profile_sample(nalo_ctx* c, int every) { if (!c || every < 1) return NALO_ERR_ARG; c->prof_every = every; c->prof_tick = 0; return NALO_OK; }
int nalo_profile_reset(nalo_ctx* c) { if (!c) return NALO_ERR_ARG; prof_drain(c); c->prof.clear(); return NALO_OK; }
int nalo_profile_get(nalo_ctx* c, const char* kernel, double* total_ms, int* launches) {
    if (!c || !kernel) return NALO_ERR_ARG;
    NALO_HIP(c, hipStreamSynchronize(c->stream));
    prof_drain(c);
    auto it = c->prof.find(kernel);
    if (total_ms) *total_ms = it == c->prof.end() ? 0.0 : it->second.ms;
    if (launches) *launches = it == c->prof.end() ? 0 : it->second.n;
    return NALO_OK;
}

// every bracketed launch of a scope since the last reset, in launch order (milliseconds): what a mean hides - the spread, and the position inside a keyframe
int nalo_profile_samples(nalo_ctx* c, const char* kernel, float* ms, int cap, int* n) {
    if (!c || !kernel || cap < 0 || (cap > 0 && !ms)) return NALO_ERR_ARG;
    NALO_HIP(c, hipStreamSynchronize(c->stream));
    prof_drain(c);
    auto it = c->prof.find(kernel);
    const int have = it == c->prof.end() ? 0 : (int)it->second.samples.size();
    if (n) *n = have;
    if (ms && have) std::memcpy(ms, it->second.samples.data(), (size_t)std::min(cap, have) * sizeof(float));
    return NALO_OK;
}

int nalo_hbm_calibrate(nalo_ctx* c, size_t bytes, int iters, double* copy_GBs, double* triad_GBs) {
    bytes &= ~(size_t)15;
    if (!c || bytes < ((size_t)1 << 20) || iters < 1) return fail(c, NALO_ERR_ARG, "nalo_hbm_calibrate: bad argument");
    NALO_HIP(c, hipSetDevice(c->device));
    float4 *a = nullptr, *b = nullptr, *d = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto done = [&](int rc) { if (a) (void)hipFree(a); if (b) (void)hipFree(b); if (d) (void)hipFree(d); if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); return rc; };
    if (hipMalloc((void**)&a, bytes) != hipSuccess || hipMalloc((void**)&b, bytes) != hipSuccess || hipMalloc((void**)&d, bytes) != hipSuccess ||
        hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return done(fail(c, NALO_ERR_HIP, "nalo_hbm_calibrate: allocation failed"));
    if (hipMemsetAsync(a, 0, bytes, c->stream) != hipSuccess || hipMemsetAsync(b, 0, bytes, c->stream) != hipSuccess) return done(fail(c, NALO_ERR_HIP, "nalo_hbm_calibrate: memset failed"));
    const size_t n = bytes / 16;
    for (int which = 0; which < 2; ++which) {
        double* out = which ? triad_GBs : copy_GBs;
        if (!out) continue;
        hbm_stream_launch(c->stream, a, b, d, n, which);                               // untimed: first touch, clocks
        (void)hipEventRecord(e0, c->stream);
        for (int i = 0; i < iters; ++i) hbm_stream_launch(c->stream, a, b, d, n, which);
        (void)hipEventRecord(e1, c->stream);
        if (hipEventSynchronize(e1) != hipSuccess) return done(fail(c, NALO_ERR_HIP, "nalo_hbm_calibrate: kernel failed"));
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        *out = (which ? 3.0 : 2.0) * (double)bytes * iters / ((double)ms * 1e-3) / 1e9;
    }
    return done(NALO_OK);
}

}  // extern "C"
