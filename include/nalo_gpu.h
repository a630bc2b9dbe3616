/*
 * nalo_gpu.h — C ABI of the MI355X (gfx950) hot path of the NALO-SLAM direct photometric core.
 *
 * Drop-in boundary. The reference has no FFI: its hot path is plain C++ member calls.
 * Every entry point below names the reference call site it replaces (paths relative to the reference's
 * src/). A maintainer binds these from the reference's own FullSystem / CoarseTracker / EnergyFunctional
 * (see INTEGRATION.md for the stubs).
 *
 * Conventions
 *   - opaque nalo_ctx*, one per thread of use (the reference keeps two CoarseTracker instances and one
 *     EnergyFunctional, FullSystem/FullSystem.h:310-311; use one ctx per such owner);
 *   - every call returns 0 on success, <0 on error (nalo_last_error gives the message); no exceptions
 *     cross the ABI; all calls are synchronous on return unless suffixed _async;
 *   - plain pointers and sizes only; host buffers are caller-owned; matrices are row-major;
 *   - SE(3) is a row-major 3x4 [R|t] of doubles; the tangent order is Sophus' [translation(3), rotation(3)];
 *   - arithmetic is IEEE fp32 on the device exactly where the reference uses float, partial sums are
 *     finished in fp64, stitched systems and solves are fp64 (reference: Eigen double).
 *   - the library needs a gfx950 device; there is no CPU fallback (calls fail with NALO_ERR_NO_DEVICE).
 */
#ifndef NALO_GPU_H
#define NALO_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nalo_ctx nalo_ctx;

enum {
    NALO_OK = 0,
    NALO_ERR_ARG = -1,
    NALO_ERR_NO_DEVICE = -2,
    NALO_ERR_HIP = -3,
    NALO_ERR_STATE = -4,
    NALO_ERR_UNSUPPORTED = -5
};

#define NALO_MAX_LEVELS 6     /* PYR_LEVELS, util/settings.h:52 (tracker uses <= 5, CoarseTracker.cpp:1083) */
#define NALO_MAX_WINDOW 16    /* frames in the BA window; reference: setting_maxFrames+1 = 8 (util/settings.cpp:88) */

/* ------------------------------------------------------------------------------------------------
 * Context.  Replaces: setGlobalCalib (util/globalCalib.cpp:45-105) + the allocations in
 * CoarseTracker::CoarseTracker (FullSystem/CoarseTracker.cpp:58-95) / EnergyFunctional().
 * levels <= 0 -> use the reference's level rule (halve while both dims even and area > 5000).
 * K = {fx, fy, cx, cy} at level 0. n_slots = number of frame slots (pyramids resident in HBM).
 * ------------------------------------------------------------------------------------------------ */
int nalo_create(nalo_ctx** out, int device, int w, int h, int levels, const float K[4], int n_slots);
void nalo_destroy(nalo_ctx* ctx);
const char* nalo_last_error(nalo_ctx* ctx);
int nalo_levels(nalo_ctx* ctx);
int nalo_sync(nalo_ctx* ctx);
void* nalo_stream(nalo_ctx* ctx);                 /* hipStream_t every kernel of this ctx is launched on */

/* ------------------------------------------------------------------------------------------------
 * Settings of the reference that change the arithmetic of this path (util/settings.cpp). Defaults = the reference's; nalo_set_settings before
 * nalo_ba_set_window / nalo_trk_track. Everything else in settings.cpp that this path reads is a compile-time constant here, as it is never changed by
 * the reference's CLI (main_dso_pangolin.cpp:400-460 changes exactly these: mode=1 sets the affine modes to 0, mode=2 to -1).
 *   forceAcceptStep   setting_forceAceptStep (:71, default 1). 0: FullSystem::optimize linearises WITHOUT applyRes, compares
 *                     E + calcLEnergy + calcMEnergy against the last accepted values and either applies the step or restores the backup
 *                     (FullSystemOptimize.cpp:511-541). On a sharded window the scalars of that test are summed over the ranks through the all-reduce hook
 *                     (three more calls of 1, 1 and 3 doubles per evaluation), so every rank takes the same branch.
 *   affineOptModeA/B  setting_affineOptModeA / B (:128-129, defaults 1e12 / 1e8): >= 0 = prior on a / b of every frame but the first
 *                     (FrameHessian::getPrior, HessianBlocks.h:286-312), < 0 = fixed: the prior becomes setting_initialAffAPrior / BPrior, JabF[0] / JabF[1]
 *                     are zeroed (Residuals.cpp:241-242), the tracker solves the reduced 6x6 / 7x7 system (CoarseTracker.cpp:1140-1162, host LM loop) and
 *                     zeroes the fixed output (:1255-1256); == 0 switches the tracker's plausibility test to relAff (:1249-1250).
 *   minOptIterations  setting_minOptIterations (:74, default 1).
 * ------------------------------------------------------------------------------------------------ */
/* The reference constants the library was compiled with (util/settings.cpp:56-160,297, util/settings.h:52,232-234, FullSystem/HessianBlocks.h:61-68,268,
 * util/NumType.h:41-53), under the reference's own names ("setting_huberTH", "SCALE_XI_TRANS", "patternP[3].x", ...). They come from the one table the host and
 * device code take their constexpr values from (csrc/ref_constants.h). Float settings carry the value the reference's `float` holds (setting_initialRotPrior =
 * 1e11 is 99999997952). Returns the number of entries; the first min(cap, n) names (static strings) / values are written; either array may be NULL. No device needed. */
int nalo_constants(int cap, const char** names, double* values);
/* the same table as the DEVICE code evaluates it (a one-lane kernel writes every entry): values[i] belongs to names[i] of nalo_constants. Returns the number of entries. */
int nalo_constants_device(nalo_ctx* ctx, int cap, double* values);
typedef struct nalo_settings {
    int forceAcceptStep;
    double affineOptModeA, affineOptModeB;
    int minOptIterations;
} nalo_settings;
int nalo_get_settings(nalo_ctx* ctx, nalo_settings* out);
int nalo_set_settings(nalo_ctx* ctx, const nalo_settings* in);

/* ------------------------------------------------------------------------------------------------
 * a1  FrameHessian::makeImages (FullSystem/HessianBlocks.cpp:127-190), called at FullSystem.cpp:1065.
 * Uploads level-0 irradiance (w*h floats, 0..255), builds all pyramid levels {I,dx,dy} + absSquaredGrad
 * in HBM. mask (w*h floats) and bgr (3*w*h bytes) are optional (densemap=1 only); gammaB[256] optional
 * (CalibHessian::B, HessianBlocks.h:397-406; NULL = identity).
 * ------------------------------------------------------------------------------------------------ */
int nalo_frame_upload(nalo_ctx* ctx, int slot, const float* irradiance, const float* mask, const uint8_t* bgr,
                      const float* gammaB);
/* The per-frame entry of a running pipeline (FullSystem::addActiveFrame -> makeImages, FullSystem.cpp:1053-1065): same as nalo_frame_upload, but the H2D
 * copies run on the context's copy stream — under the kernels the main stream is executing for the previous frame — and the pyramid kernels are queued
 * behind them. Returns immediately; irradiance / mask / bgr / gammaB must stay untouched until nalo_frame_wait(ctx, slot) or nalo_sync returns.
 * Asynchronous only from pinned host memory: nalo_host_alloc / nalo_host_free (= hipHostMalloc / hipHostFree) hand out such buffers. */
int nalo_frame_upload_async(nalo_ctx* ctx, int slot, const float* irradiance, const float* mask, const uint8_t* bgr,
                            const float* gammaB);
int nalo_frame_wait(nalo_ctx* ctx, int slot);
/* The sensor frame as the dataset holds it: Undistort::undistort<T> (util/Undistort.cpp:435-530) = PhotometricUndistorter::processFrame
 * (:214-251) + the bilinear remap, and the INTER_NEAREST resizes of Undistort::undistort_mask (:385-433, IOWrapper/OpenCV/ImageRW_OpenCV.cpp:55-85), fused
 * in front of makeImages; call site ImageFolderReader::getImage_internal (util/DatasetReader.h:270-298) -> FullSystem::addActiveFrame.
 * nalo_undist_set          the tables of the Undistort object, uploaded once: the response G (GDepth >= 256 entries, already rescaled to 0..255 as
 *                          Undistort.cpp:101-103 does: nalo_io_read_pcalib), vignetteMapInv [wOrg*hOrg] (nalo_io_make_vignette; NULL without a vignette),
 *                          photometricCalibration = setting_photometricCalibration (0 none, 1 response only, 2 response + vignette; util/settings.cpp:40),
 *                          remapX / remapY [w*h] in original-image pixels, -1 = outside (Undistort.cpp:998-1010); NULL = passthrough (wOrg x hOrg = w x h).
 * nalo_frame_upload_raw    raw = wOrg*hOrg pixels of bytes_per_px 1 (uchar) or 2 (ushort); exposure_time <= 0 disables the photometric part for this frame
 *                          (data = factor * raw, :224-231); mask_org [wOrg*hOrg] / bgr_org [wOrg*hOrg*3] optional (dense=1 / densemap=1 inputs);
 *                          gammaB as in nalo_frame_upload. The frame crosses PCIe at 1-2 B/px instead of 4. Synchronous like nalo_frame_upload. */
int nalo_undist_set(nalo_ctx* ctx, int wOrg, int hOrg, const float* G, int GDepth, const float* vignetteMapInv, int photometricCalibration, const float* remapX,
                    const float* remapY);
int nalo_frame_upload_raw(nalo_ctx* ctx, int slot, const void* raw, int bytes_per_px, float exposure_time, float factor, const uint8_t* mask_org, const uint8_t* bgr_org,
                          const float* gammaB);
/* asynchronous form for tracked (non-key) frames, as nalo_frame_upload_async: the raw copy runs on the copy stream, ingest + pyramid are queued behind it; returns
 * at once, raw must stay untouched until nalo_frame_wait(ctx, slot) (pinned memory: nalo_host_alloc). No mask / colour (keyframe inputs). */
int nalo_frame_upload_raw_async(nalo_ctx* ctx, int slot, const void* raw, int bytes_per_px, float exposure_time, float factor, const float* gammaB);
void* nalo_host_alloc(size_t bytes);
void nalo_host_free(void* p);
/* makeImages again from the level-0 irradiance already resident in the slot (asynchronous on the ctx stream): the
 * HBM-resident form of a1, used when the caller's frames already live on the device */
int nalo_frame_rebuild(nalo_ctx* ctx, int slot);
/* test/inspection: level image as AoS {I,dx,dy} (3 floats/px) and absSquaredGrad (1 float/px); either may be NULL */
int nalo_frame_download(nalo_ctx* ctx, int slot, int lvl, float* dI3, float* abs_sq_grad);

/* ------------------------------------------------------------------------------------------------
 * Front-end tracker.
 * ------------------------------------------------------------------------------------------------ */
/* CoarseTracker::makeK (CoarseTracker.cpp:97-141): pyramid intrinsics from the current calibration */
int nalo_trk_make_k(nalo_ctx* ctx, float fx, float fy, float cx, float cy);

/* a2  CoarseTracker::setCoarseTrackingRef -> makeCoarseDepthL0 steps 1-5 (CoarseTracker.cpp:1053-1067, 382-538),
 * called at FullSystem.cpp:1404. One entry per IN residual targeting the reference keyframe:
 * centerProjectedTo = (Ku, Kv, new_idepth) and the point's HdiF. Builds idepth/weight pyramids and the
 * per-level point clouds pc_u/pc_v/pc_idepth/pc_color in raster order. */
int nalo_trk_set_ref(nalo_ctx* ctx, int slot_ref, int n, const float* Ku, const float* Kv,
                     const float* new_idepth, const float* HdiF);
/* The same with the four input arrays RESIDENT on the device (round 4): nalo_trk_ref_upload copies them into a block the context owns (one pinned staging copy + one
 * H2D copy, as nalo_trk_set_ref does on every call) and nalo_trk_set_ref_resident runs makeCoarseDepthL0 (CoarseTracker.cpp:382-538) from that block - for a caller whose
 * inputs do not change between calls, or who fills them ahead of time: the per-keyframe call then pays no host copy, no staging and no copy packet in front of its kernels.
 * The block stays valid until the next nalo_trk_ref_upload; nalo_trk_set_ref leaves it untouched. Results are those of nalo_trk_set_ref on the same arrays, bit for bit. */
int nalo_trk_ref_upload(nalo_ctx* ctx, int n, const float* Ku, const float* Kv, const float* new_idepth, const float* HdiF);
int nalo_trk_set_ref_resident(nalo_ctx* ctx, int slot_ref);
/* direct injection of one level's point cloud (synthetic stress windows) */
int nalo_trk_set_pc(nalo_ctx* ctx, int slot_ref, int lvl, int n, const float* u, const float* v,
                    const float* idepth, const float* color);
int nalo_trk_get_pc(nalo_ctx* ctx, int lvl, int* n, float* u, float* v, float* idepth, float* color);
/* dense=1: the plane-sampled points makeCoarseDepthL0 appends to the LEVEL-0 cloud after step 5 (CoarseTracker.cpp:600-655), one call per mask cluster, on
 * the device (no round trip of the cloud): dir / dis_plane = the cluster's fitted plane (fitPlane, a PCL RANSAC on the caller's side), refMaskColor =
 * clusters[i][0][3], rect = {minx, maxx, miny, maxy} of the cluster's pixels. Uses the mask and I of the tracking reference's slot (frameHessians.back() is
 * lastRef). Appends, in the reference's x-outer / y-inner order, every (x % 5 == 0, y % 5 == 0) pixel of [minx,maxx) x [miny,maxy) whose mask equals
 * refMaskColor with new_idepth = dir^T Ki (x,y,1) / -dis_plane and colour I_ref(x,y) — INCLUDING the reference's off-by-one (the k-th point is stored at
 * pc_n + 1 + k while pc_n grows by one per point, :646-650): slot [old pc_n] is left unwritten by the reference (defined as zeros here) and the last
 * sampled point stays outside the count. Nothing is appended when the box touches the border, refMaskColor is 0 (:621-630). *n_added = growth of pc_n[0]. */
int nalo_trk_append_plane_points(nalo_ctx* ctx, const float dir[3], float dis_plane, int refMaskColor, const int rect[4], int* n_added);
int nalo_trk_get_depth(nalo_ctx* ctx, int lvl, float* idepth, float* weight_sums);

/* Sharded tracker (the reference's analogue is the per-thread sum of calcGSSSE, CoarseTracker.cpp:828-885): with world > 1 every rank evaluates
 * points [n rank / world, n (rank + 1) / world) of every pyramid level and the 52 sums of an evaluation (45 H entries, E and the six counters, as doubles) are summed
 * over the ranks by hook(user, device_buf, 52) - same contract as nalo_ba_set_allreduce; stream_ordered != 0: the hook enqueues on nalo_stream(ctx) and returns.
 * nalo_trk_set_ref stays replicated (every rank passes all reference points). nalo_trk_eval and nalo_trk_track both honour it; nalo_trk_track then runs its
 * host-driven LM loop (one launch + one all-reduce per evaluation), every rank ends with the same pose. world = 1 switches it off. */
typedef void (*nalo_allreduce_fn)(void* user, double* device_buf, int n);
int nalo_trk_set_shard(nalo_ctx* ctx, int rank, int world, nalo_allreduce_fn hook, void* user, int stream_ordered);

/* a3+a4 fused  CoarseTracker::calcRes (CoarseTracker.cpp:891-1049) + calcGSSSE (:828-885), call sites
 * CoarseTracker.cpp:1104,1109,1115,1184,1204. R,t = refToNew; affLL = fromToVecExposure(ref,new) as float;
 * b0 = lastRef_aff_g2l.b. stats6 = {E, numTermsInE, shiftT/(n+.1), 0, shiftRT/(n+.1), saturatedRatio}.
 * If want_gs: H (8x8) and b (8) as calcGSSSE returns them (divided by the PADDED count, scaled by SCALE_*). */
int nalo_trk_eval(nalo_ctx* ctx, int slot_new, int lvl, const double R[9], const double t[3],
                  const float affLL[2], float b0, float cutoffTH, int want_gs,
                  double stats6[6], double H[64], double b[8]);

/* CoarseTracker::trackNewestCoarse (CoarseTracker.cpp:1073-1259), call site FullSystem.cpp:594-597.
 * T_io = lastToNew_out, aff_io = aff_g2l_out (a,b), ref_aff = lastRef_aff_g2l, exposures = {ref,new}.
 * *ok = the bool the reference returns. n_evals (optional) = number of fused evaluations launched. */
int nalo_trk_track(nalo_ctx* ctx, int slot_new, double T_io[12], double aff_io[2], const double ref_aff[2],
                   const float exposures[2], int coarsestLvl, const double minResForAbort[5],
                   double lastResiduals[5], double lastFlowIndicators[3], int* ok, int* n_evals);

/* measurement aid: LM evaluations per pyramid level (evals[l]) and point-cloud sizes (n[l]) of the last nalo_trk_track; the algorithmic bytes of that frame are
 * sum_l evals[l] * n[l] * 64 B. Either array may be NULL. */
int nalo_trk_last_evals(nalo_ctx* ctx, int evals[5], int n[5]);

/* ------------------------------------------------------------------------------------------------
 * Back-end: sliding-window photometric bundle adjustment.
 * ------------------------------------------------------------------------------------------------ */
typedef struct nalo_frame_state {
    int slot;                 /* frame slot holding the pyramid (nalo_frame_upload) */
    int frame_id;             /* FrameHessian::frameID (0 gets the gauge priors, HessianBlocks.h:321-350) */
    double worldToCam_evalPT[12];
    double state[10];         /* FrameHessian::state (unscaled); [0:6] relative to evalPT, [6:8] = a,b / SCALE */
    double state_zero[10];
    float ab_exposure;
    float frameEnergyTH;
} nalo_frame_state;

/* EnergyFunctional::insertFrame / FullSystem::setPrecalcValues / EnergyFunctional::setAdjointsF, setDeltaF
 * (OptimizationBackend/EnergyFunctional.cpp:429-461, 46-106, 171-194; FullSystem.cpp:1694-1704).
 * The newest keyframe must be the last entry (frameHessians.back()). calib = {fx,fy,cx,cy} value_scaled. */
int nalo_ba_set_window(nalo_ctx* ctx, int W, const nalo_frame_state* frames, const double calib[4],
                       const double calib_zero[4]);
/* EnergyFunctional::insertPoint (EnergyFunctional.cpp:462-474) for all active points: SoA, P entries.
 * color/weights are P x 8 (PointHessian::color/weights, HessianBlocks.h:424-425). idepth_zero may be NULL (= idepth). */
int nalo_ba_set_points(nalo_ctx* ctx, int P, const int* host, const float* u, const float* v,
                       const float* idepth, const float* idepth_zero, const float* color, const float* weights,
                       const int* has_depth_prior);
/* EnergyFunctional::insertResidual (EnergyFunctional.cpp:417-428): exists[p*W + t] != 0 creates the
 * PointFrameResidual (point p -> target frame t) in the resetOOB state (FullSystem/Residuals.h:88-94). */
int nalo_ba_set_residuals(nalo_ctx* ctx, const uint8_t* exists);
/* marginalisation prior HM/bM ((8W+4)^2, 8W+4), EnergyFunctional.h; NULL = zero */
int nalo_ba_set_prior(nalo_ctx* ctx, const double* HM, const double* bM);
int nalo_ba_get_prior(nalo_ctx* ctx, double* HM, double* bM);

/* a5+a6 (+ the accumulation side of a7, a9)  FullSystem::linearizeAll(fix) + applyRes_Reductor
 * (FullSystem/FullSystemOptimize.cpp:144-211, 90-94, call sites :436,:460,:511,:524,:562) =
 * PointFrameResidual::linearize + applyRes + EFResidual::takeDataF (Residuals.cpp:78-274,306-328,
 * EnergyFunctionalStructs.cpp:39-50) + setNewFrameEnergyTH (:95-143). Valid because
 * setting_forceAceptStep=true (util/settings.cpp:71): every linearisation is committed.
 * energy = lastEnergyP (the stats[0] sum). fix != 0 drops residuals that are not IN, as linearizeAll(true). */
int nalo_ba_linearize(nalo_ctx* ctx, int fix, double* energy);
/* a7+a8  EnergyFunctional::accumulateAF_MT (mode 0) / accumulateLF_MT (mode 1)  (EnergyFunctional.cpp:197-238,
 * call sites :788,:791): stitched H ((8W+4)^2) and b. Mode 1 adds the priors (usePrior=true); linearised
 * residuals never exist at this call in the reference flow (they live only inside flagPointsForRemoval ->
 * marginalizePointsF, FullSystem.cpp:975-990,1453) so mode 1 carries priors only. */
int nalo_ba_accumulate(nalo_ctx* ctx, int mode, double* H, double* b);
/* a9+a10  EnergyFunctional::accumulateSCF_MT (EnergyFunctional.cpp:244-261, call site :795) */
int nalo_ba_accumulate_sc(nalo_ctx* ctx, int shiftPriorToZero, double* H_sc, double* b_sc);
/* a11+a12  EnergyFunctional::solveSystemF (EnergyFunctional.cpp:776-914, call site FullSystemOptimize.cpp:616):
 * accumulate A, L, SC; assemble; Jacobi-scaled LDL^T; orthogonalise x for iteration >= 2; resubstituteF_MT.
 * x_out (8W+4) optional. */
int nalo_ba_solve_system(nalo_ctx* ctx, int iteration, double lambda, double* x_out);
/* FullSystem::backupState + doStepFromBackup (FullSystemOptimize.cpp:304-349, 217-299) */
int nalo_ba_backup_state(nalo_ctx* ctx);
int nalo_ba_do_step(nalo_ctx* ctx, float stepfacC, float stepfacT, float stepfacR, float stepfacA, float stepfacD,
                    int* canbreak);
/* FullSystem::optimize (FullSystemOptimize.cpp:398-602, call site FullSystem.cpp:1362). never_break != 0 disables
 * the early exit at :544 so a benchmark keyframe always runs mnumOptIts iterations. */
int nalo_ba_optimize(nalo_ctx* ctx, int mnumOptIts, int never_break, double* rmse);
/* a13  EnergyFunctional::calcLEnergyF_MT + calcLEnergyPt (EnergyFunctional.cpp:332-415) and calcMEnergyF (:320-329) at the CURRENT states, as
 * FullSystem::calcLEnergy / calcMEnergy return them when setting_forceAceptStep is false (FullSystemOptimize.cpp:351, 371-379; they return 0 otherwise —
 * these two entry points always compute). L = frame priors + calibration prior + per point deltaF^2 priorF (device reduction); the inner loop of
 * calcLEnergyPt runs over linearised residuals, which never exist while optimize() runs (they live only inside nalo_ba_marginalize_points).
 * M = delta . (2 bM + HM delta) on the host in fp64. *rejected (nalo_ba_optimize_stats) = steps the last nalo_ba_optimize restored from the backup. */
int nalo_ba_calc_l_energy(nalo_ctx* ctx, double* E);
/* planeOpt=1, call sites FullSystem.cpp:1440-1441, without Ceres:
 * nalo_ba_plane_scale_fix    FullSystem::planeOptimize's active part (FullSystem/PlaneOptimize.cpp:183-301) for the newest keyframe: camToWorld =
 *     trackingRef.camToWorld * [R | localscale * t](camToTrackingRef), its own points' idepth /= localscale (idepth_zero alike), setEvalPT at the new pose, adjoints,
 *     precalc. localscale = getlocalgh() / groundP[3] is the caller's (ground plane from its RANSAC); the caller skips the call when scaleFixed / no ground.
 * nalo_ba_sw_gray_optimize   FullSystem::SWGrayOptimize_J (:307-454). The reference's Ceres cost functor multiplies every Jacobian by an image gradient it never
 *     reads (a shadowed variable, PlaneOptimize.h:378-381): the gradient is identically zero, Ceres returns its initial point. The call therefore
 *     (a) evaluates the Huber(100) cost 1/2 sum rho(r^2) over all (point, target != host) centre-pixel residuals with 1e-4 <= idepth <= 1e3 on the device
 *     (*cost, *n_residual_blocks: what Ceres' summary reports as initial = final cost), and (b) applies the post-solve state changes with the unchanged
 *     parameters: newest frame PRE_worldToCam = [exp(log R) | t] and re-linearised there, idepth_zero = idepth for the points of frames 0 .. W-3, adjoints and
 *     precalc values recomputed. Read the result with nalo_ba_get_frames / nalo_ba_get_points. */
int nalo_ba_plane_scale_fix(nalo_ctx* ctx, double localscale, const double camToTrackingRef[12], const double trackingRef_camToWorld[12]);
int nalo_ba_sw_gray_optimize(nalo_ctx* ctx, double* cost, int* n_residual_blocks);
int nalo_ba_calc_m_energy(nalo_ctx* ctx, double* E);
int nalo_ba_optimize_stats(nalo_ctx* ctx, int* iterations, int* rejected);
/* a13 + a7<2> + a9  flagPointsForRemoval's relinearise/fixLinearizationF (FullSystem.cpp:975-990,
 * EnergyFunctionalStructs.cpp:89-115) + EnergyFunctional::marginalizePointsF (EnergyFunctional.cpp:615-676).
 * flags[p] != 0 marks PS_MARGINALIZE. Adds 0.25*(M - Msc) into HM/bM and removes the points.
 * M, Mb, Msc, Mbsc (optional outputs) are the stitched systems. */
int nalo_ba_marginalize_points(nalo_ctx* ctx, const uint8_t* flags, double* M, double* Mb, double* Msc, double* Mbsc);

/* EnergyFunctional::marginalizeFrame (OptimizationBackend/EnergyFunctional.cpp:498-610), call site FullSystem::marginalizeFrame
 * (FullSystem/FullSystemMarginalize.cpp:155). Host fp64 on HM/bM: the frame `idx` (window index) is permuted to the end, its prior is added,
 * the scaled 8x8 block is inverted and eliminated by a Schur complement; HM/bM shrink to 8(W-1)+4. The frame must not host active points any more
 * (the reference asserts it: marginalise or drop them with nalo_ba_marginalize_points / by not re-submitting them). The frame leaves the window:
 * W decreases by one, nalo_ba_get_frames / nalo_ba_get_prior return the remaining frames, and the device window must be re-issued with
 * nalo_ba_set_window (+ set_points, set_residuals) before the next linearisation — FullSystem::marginalizeFrame likewise drops every residual
 * that targets the frame and recomputes the precalc values and adjoints (:161-212). The shrunk HM/bM are handed to exactly that NEXT nalo_ba_set_window:
 * kept if it names the remaining frames, extended by a zero block if it appends one keyframe (EnergyFunctional::insertFrame, :437-442); any other size
 * resets the prior to zero.
 * Prior ownership in general: nalo_ba_set_window starts from a ZERO prior unless (a) it directly follows nalo_ba_marginalize_frame (above) or (b) the context
 * was declared one continuing EnergyFunctional with nalo_ba_set_prior_carry(ctx, 1): then every nalo_ba_set_window keeps HM/bM (same frames) or extends them
 * (one frame appended), as the reference's single EnergyFunctional does over a session. A caller that keeps HM/bM itself needs neither:
 * it calls nalo_ba_set_prior after every nalo_ba_set_window. */
int nalo_ba_marginalize_frame(nalo_ctx* ctx, int idx);
int nalo_ba_set_prior_carry(nalo_ctx* ctx, int on);

/* read-back of window state (host pointers, any may be NULL) */
int nalo_ba_get_frames(nalo_ctx* ctx, nalo_frame_state* frames /* W */, double* worldToCam /* W x 12 PRE_worldToCam */,
                       double calib[4]);
/* Hdd_accAF / bd_accAF / Hcd_accAF / HdiF / bdSumF are values of the last ACCUMULATION, as in the reference (addPoint<0>, AccumulatedSCHessianSSE::addPoint):
 * after nalo_ba_optimize those of its last solveSystemF — what CoarseTracker::makeCoarseDepthL0 reads as HdiF (CoarseTracker.cpp:396), so read them
 * BEFORE nalo_ba_marginalize_points (which re-accumulates); after an explicit nalo_ba_linearize the accumulation of that linearisation is run on demand. */
int nalo_ba_get_points(nalo_ctx* ctx, float* idepth, float* step, float* HdiF, float* bdSumF, float* Hdd_accAF,
                       float* bd_accAF, float* Hcd_accAF /* P x 4 */, float* maxRelBaseline);
/* PointHessian::idepth_zero of every point (the linearisation point SWGrayOptimize_J / loadSateBackup / doStepFromBackup move) */
int nalo_ba_get_idepth_zero(nalo_ctx* ctx, float* idepth_zero);
/* per residual slot [p*W + t]: state (-1 none, 0 IN, 1 OOB, 2 OUTLIER), active flag, JpJdF (8), state_NewEnergyWithOutlier,
 * centerProjectedTo (3) */
int nalo_ba_get_residuals(nalo_ctx* ctx, int8_t* state, uint8_t* active, float* JpJdF, float* energy_new,
                          float* center_projected);
/* per (host,target) bin 13x13 accumulator of the last linearisation (AccumulatorApprox::H, MatrixAccumulators.h:600),
 * index h + t*W, fp64 */
int nalo_ba_get_acc13(nalo_ctx* ctx, double* H13 /* W*W x 169 */);
int nalo_ba_counts(nalo_ctx* ctx, int* resInA, int* resInL, int* resInM);

/* bench/test utility (no reference counterpart): snapshot / restore the mutable window state (idepths, residual states,
 * frame states, calibration, HM/bM) on the device, so the same synthetic keyframe can be replayed without host uploads */
int nalo_ba_snapshot(nalo_ctx* ctx);
int nalo_ba_restore(nalo_ctx* ctx);

/* multi-GPU: the active-point set is sharded by the caller (each rank sets only its points); the stitched
 * buffers {H_A, b_A, H_sc, b_sc, energy, counters} are summed across ranks through this hook before every solve (the replica sum of
 * AccumulatedTopHessian.h:144-149 across GPUs), and so are the three radix histograms of setNewFrameEnergyTH (FullSystemOptimize.cpp:95-143) after every
 * linearisation: the newest frame's energy threshold is the exact order statistic over ALL ranks' residuals, i.e. what one GPU holding the whole window
 * computes. Per pass the hook is called THREE times, in the same order on every rank: level A and level B of the radix select (1024 doubles each: 2048 bins, two
 * per double as a + b * 2^26; on the side stream when a side hook is set), then one buffer [stitched systems | tail | level C] (2 (8W+5)^2 + 2 W^2 + 5 doubles
 * padded to a multiple of 16, + 256) = 166 KB + 2 KB at W = 12; a pass whose systems are never fetched sums level C alone (256) when the threshold is next
 * needed; a second fetch of a pass (the Schur complement after the top system, or the reverse) sums ONLY the block it stitched - everything else already holds
 * the window's totals. buf is a DEVICE pointer to n doubles. hook == NULL = single GPU.
 * A hook returns nothing: when its collective fails it calls nalo_ba_exchange_failed(ctx, message) before returning (the built-in RCCL hooks of
 * nalo_ba_rccl_init do the same). The call that issued the hook and every later nalo_ba_* call of the context then return NALO_ERR_HIP - a rank must not
 * solve with sums the others never received; the window has to be rebuilt on a new context. */
/* nalo_allreduce_fn: declared with nalo_trk_set_shard above */
int nalo_ba_set_allreduce(nalo_ctx* ctx, nalo_allreduce_fn hook, void* user);
int nalo_ba_exchange_failed(nalo_ctx* ctx, const char* what);
/* stream_ordered = 1: the hook ENQUEUES its collective on nalo_stream(ctx) (e.g. ncclAllReduce(..., (hipStream_t)nalo_stream(ctx))) and returns
 * without waiting; the library then neither synchronises before nor after the hook. Default 0: the library synchronises the stream the buffer was
 * produced on before the call and the hook returns when the sum is complete. */
int nalo_ba_set_allreduce_mode(nalo_ctx* ctx, int stream_ordered);
/* optional, stream-ordered mode: a second hook that enqueues the same sum on nalo_side_stream(ctx). The histogram sums then run on the side stream,
 * under the Schur-complement / reduce / stitch kernels of the main stream, instead of in line before them. */
int nalo_ba_set_allreduce_side(nalo_ctx* ctx, nalo_allreduce_fn hook, void* user);
void* nalo_side_stream(nalo_ctx* ctx);
/* The native form of the exchange: the library calls ncclAllReduce (RCCL over xGMI) itself, stream-ordered on nalo_stream / nalo_side_stream — no
 * callback into the caller per Gauss-Newton iteration. librccl is opened with dlopen when the first of these is called (NALO_ERR_UNSUPPORTED if absent).
 *   nalo_rccl_unique_id    ncclGetUniqueId: rank 0 draws one id per communicator and ships it to the other ranks (MPI, a file, torch.distributed ...)
 *   nalo_ba_rccl_init      ncclCommInitRank for the main and (id_side != NULL) the side communicator on this context's device; collective over the ranks;
 *                          the context owns and destroys the communicators
 *   nalo_ba_set_rccl_comm  the same with ncclComm_t handles the caller owns (comm_side may be NULL: the histogram sums then run in line on the main
 *                          stream); comm_main == NULL returns to single-GPU operation
 * Two communicators because the stitched systems (main stream) and the two radix histograms of setNewFrameEnergyTH (side stream, under the
 * Schur-complement kernels) are in flight at the same time, and RCCL serialises the operations of one communicator. Call after nalo_ba_set_window. */
int nalo_rccl_unique_id(char id[128]);
int nalo_ba_rccl_init(nalo_ctx* ctx, int nranks, int rank, const char id_main[128], const char id_side[128]);
int nalo_ba_set_rccl_comm(nalo_ctx* ctx, void* comm_main, void* comm_side);
/* ncclCommCount of the context's two communicators (0 = none installed): the number of ranks the exchange REALLY spans, for a launcher to print beside the
 * number of processes it started (the per-thread replica count the reference sums over at stitch, AccumulatedTopHessian.h:144-149, is NUM_THREADS there). */
int nalo_ba_rccl_ranks(nalo_ctx* ctx, int* ranks_main, int* ranks_side);
/* The partition a sharded window uses: which of the P active points rank `rank` of `world` keeps. Every rank gets the same share of every
 * host frame (all (host, target) bins stay evenly populated) as a contiguous range of the host's points in Hilbert order of their 8x8-pixel cells — a
 * spatially compact part of the image, so a rank's texel gathers keep the locality of the unsharded window. keep[] (capacity P) receives the ascending
 * point indices; the return value is their number (< 0: NALO_ERR_ARG). Host code, needs no device. */
int nalo_shard_points(int P, int W, const int* host, const float* u, const float* v, int img_w, int img_h, int rank, int world, int* keep);

/* ------------------------------------------------------------------------------------------------
 * a14  DenseMapping::updateMap bbox scan + makeMap (FullSystem/MapPoint.cpp:300-310, 334-407), call site
 * FullSystem.cpp:1494. plane = (pi1..pi4) from the caller's RANSAC. rect_out = {minx,maxx,miny,maxy}.
 * Outputs at most cap points in raster order; *n = count; *accept = the extent test of MapPoint.cpp:403.
 * ------------------------------------------------------------------------------------------------ */
int nalo_dense_make_map(nalo_ctx* ctx, int slot, const float plane[4], float mask_value, const double camToWorld[12],
                        int cap, int rect_out[4], int* out_u, int* out_v, float* out_idepth, float* out_color,
                        uint8_t* out_bgr, int* n, int* accept);

/* ------------------------------------------------------------------------------------------------
 * the immature-point depth filter and point activation that run either side of the BA.
 * Point state is caller-owned (the reference keeps it in ImmaturePoint objects): arrays of n points in, updated arrays out.
 * color/weights are [n][8] (pattern order), gradH is [n][3] = {xx, xy, yy}.
 *
 * nalo_imm_create    ImmaturePoint::ImmaturePoint (FullSystem/ImmaturePoint.cpp:32-60), call site FullSystem::makeNewTraces
 *                    (FullSystem.cpp:1596-1625): color, weights, gradH, energyTH (NaN = point rejected) from the host frame's slot.
 * nalo_imm_trace     ImmaturePoint::traceOn (ImmaturePoint.cpp:76-435) for every immature point of every host against the frame in
 *                    slot_new, replaces the loops of FullSystem::traceNewCoarse (FullSystem.cpp:702-744), which computes per host
 *                    KRKi = K R K^-1 (row-major 3x3), Kt = K t and the affine pair (AffLight::fromToVecExposure): pass them as [nh][9],
 *                    [nh][3], [nh][2] with host_idx[n] selecting the host of each point.
 *                    In/out: idepth_min, idepth_max, status (ImmaturePointStatus: 0 GOOD, 1 OOB, 2 OUTLIER, 3 SKIPPED, 4 BADCONDITION,
 *                    5 UNINITIALIZED), quality. Out: lastTraceUV [n][2], lastTracePixelInterval [n].
 * nalo_imm_optimize  FullSystem::optimizeImmaturePoint (FullSystem/FullSystemOptPoint.cpp:51-206) incl. ImmaturePoint::linearizeResidual
 *                    (ImmaturePoint.cpp:497-564), call site FullSystem::activatePointsMT_Reductor (FullSystem.cpp:748-762). Uses the
 *                    frames, current states (PRE_RTll / PRE_tTll / PRE_aff_mode) and calibration of the window set by nalo_ba_set_window;
 *                    host[n] = window index of each point's host frame.
 *                    result[n]: 0 = not well constrained (the point stays immature), -1 = drop the point, 1 = activated with
 *                    idepth_out[n]; res_in[n][W] = 1 where a PointFrameResidual is created (state IN).
 * ------------------------------------------------------------------------------------------------ */
/* CoarseDistanceMap::makeDistanceMap + growDistBFS (FullSystem/CoarseTracker.cpp:1410-1561), call site FullSystem::activatePointsMT
 * (FullSystem.cpp:797-798). Uses the ACTIVE POINTS of the window set by nalo_ba_set_points (already on the device); frame = window index of the newest frame
 * (its own points are skipped); KRKi[W][9] = K[1] R Ki[0] and Kt[W][3] = K[1] t per host (floats, as :1424-1425). out = fwdWarpedIDDistFinal [w1*h1]
 * (level-1 size), 1000 = farther than 39. addIntoDistFinal (:1556-1561, one seed per newly activated point, sequential) stays on the caller's copy. */
/* ---- PixelSelector (FullSystem/PixelSelector2.cpp), SURVEY 8(f) rank 3. The selector's state (randomPattern, the block thresholds of
 * gradHistFrame, the last status map) lives in the context, as it lives in the PixelSelector object.
 *  nalo_pixsel_set_random       the constructor's randomPattern[w*h] (:40-45: srand(3141592); rand() & 0xFF) and, for FusedWithMask, the first w*h
 *                               rand() values after srand(3141592) (:496-501; NULL if makeMaps_lidar is not used). Both are libc streams, so the
 *                               caller draws them and the selection stays identical to the reference built on the same libc.
 *  nalo_pixsel_make_hists       makeHists (:78-142): block thresholds of the frame in `slot`; ths / thsSmoothed [(w/32)*(h/32)] may be NULL.
 *  nalo_pixsel_select           select (:564-711) with potential `pot` on the frame make_hists ran on: n = {n2, n3, n4} (its return value),
 *                               map_out[w*h] (may be NULL) = 0 / 1 / 2 / 4 per pixel like PixelSelectorStatus.
 *  nalo_pixsel_make_maps        makeMaps (:144-291): makeHists if the frame changed, select, up to recursionsLeft re-selections with the adapted
 *                               potential, the random sub-selection; *currentPotential is PixelSelector::currentPotential (in/out),
 *                               *numHaveSub the return value. Call sites: CoarseInitializer.cpp:811, FullSystem.cpp:1663.
 *  nalo_pixsel_make_maps_lidar  makeMaps_lidar (:293-428) = makeHists + select + FusedWithMask (:431-560) with the mask the frame was uploaded
 *                               with; *numHave the return value. Call site FullSystem.cpp:1668. (The reference reads mhist[256], one past its
 *                               array, in the last quantile iteration: taken as 0 here.)
 *  nalo_pixsel_get_selected     the non-zero pixels of the last map in raster order (idx = x + y*w, status), i.e. what makeNewTraces' loop over
 *                               the map visits (FullSystem.cpp:1672-1690): *n = their number, the first min(cap, *n) are written. */
int nalo_pixsel_set_random(nalo_ctx* ctx, const uint8_t* randomPattern, const int* mask_draws);
int nalo_pixsel_make_hists(nalo_ctx* ctx, int slot, float* ths, float* thsSmoothed);
int nalo_pixsel_select(nalo_ctx* ctx, int slot, int pot, float thFactor, float* map_out, int n[3]);
int nalo_pixsel_make_maps(nalo_ctx* ctx, int slot, float density, int recursionsLeft, float thFactor, int* currentPotential, float* map_out, int* numHaveSub);
int nalo_pixsel_make_maps_lidar(nalo_ctx* ctx, int slot, float thFactor, int currentPotential, float* map_out, int* numHave);
int nalo_pixsel_get_selected(nalo_ctx* ctx, int cap, int* idx, uint8_t* status, int* n);
int nalo_dist_make_map(nalo_ctx* ctx, int frame, const float* KRKi, const float* Kt, float* out);
int nalo_imm_create(nalo_ctx* ctx, int slot_host, int n, const int* u, const int* v, float* color, float* weights, float* gradH, float* energyTH);
int nalo_imm_trace(nalo_ctx* ctx, int slot_new, int n, const float* u, const float* v, const float* color, const float* weights, const float* gradH,
                   const float* energyTH, const int* host_idx, int nh, const float* KRKi, const float* Kt, const float* aff,
                   float* idepth_min, float* idepth_max, int* status, float* quality, float* lastTraceUV, float* lastTracePixelInterval);
/* Device-resident form of the tracing state (what a running system uses: ImmaturePoints live across frames, traceNewCoarse touches all of them on
 * every frame, the host only looks at them when it activates points). nalo_imm_resident_set uploads the whole set (after makeNewTraces / activation,
 * once per keyframe; lastTraceUV = (-1,-1), lastTracePixelInterval = 0 as the constructor leaves them), nalo_imm_resident_trace = traceNewCoarse for one
 * new frame: only the nh x {KRKi, Kt, aff} cross PCIe, the call returns without waiting; nalo_imm_resident_get brings the state back (sync inside;
 * lastTraceUV / lastTracePixelInterval may be NULL). Arrays as in nalo_imm_trace. */
int nalo_imm_resident_set(nalo_ctx* ctx, int n, const float* u, const float* v, const float* color, const float* weights, const float* gradH, const float* energyTH,
                          const int* host_idx, const float* idepth_min, const float* idepth_max, const int* status, const float* quality);
int nalo_imm_resident_trace(nalo_ctx* ctx, int slot_new, int nh, const float* KRKi, const float* Kt, const float* aff);
int nalo_imm_resident_get(nalo_ctx* ctx, float* idepth_min, float* idepth_max, int* status, float* quality, float* lastTraceUV, float* lastTracePixelInterval);
int nalo_imm_optimize(nalo_ctx* ctx, int n, const int* host, const float* u, const float* v, const float* color, const float* weights,
                      const float* energyTH, const float* idepth_min, const float* idepth_max, int minObs,
                      int* result, float* idepth_out, uint8_t* res_in);
/* The same for points of the device-resident set (nalo_imm_resident_set, kept up to date by nalo_imm_resident_trace): sel[n] = their indices in that set (NULL: all of
 * them, n = the set's size). Pattern colours, weights, energyTH, host frame (host_idx must index the window's frames) and the inverse-depth interval [idepth_min,
 * idepth_max] - as the device's last trace left it, FullSystem.cpp:700-760 activates right after traceNewCoarse - are read on the device; 4 bytes per point cross PCIe
 * on the way down instead of 88. Outputs as nalo_imm_optimize. */
int nalo_imm_resident_optimize(nalo_ctx* ctx, int n, const int* sel, int minObs, int* result, float* idepth_out, uint8_t* res_in);

/* ------------------------------------------------------------------------------------------------
 * the two-frame initialiser's Gauss-Newton pass.
 * nalo_init_calc_res_and_gs  CoarseInitializer::calcResAndGS (FullSystem/CoarseInitializer.cpp:338-610), call sites :129,163 (trackFrame), for one
 *     pyramid level lvl between the pyramids in slot_first (firstFrame) and slot_new. Pnt members are caller-owned arrays of n points:
 *     in  u, v, idepth_new, iR, isGood[n] (0/1), energy[n][2], outlierTH;
 *     out isGood_new, energy_new[n][2], maxstep; in/out lastHessian_new, JbBuffer_new[n][10] (entries the reference leaves untouched stay).
 *     refToNew = 3x4 row-major [R|t], aff = {a, b} of refToNew_aff; alphaW, alphaK, couplingWeight as in the constructor (:92-95: 150^2, 2.5^2, 1).
 *     H_out/H_out_sc are 8x8 row-major, b_out/b_out_sc 8, E3 = {E.A, alphaEnergy, E.num} exactly as returned (:609), including the reference's
 *     behaviour that the regulariser loop feeds E instead of EAlpha (:560-572).
 * nalo_init_do_step          CoarseInitializer::doStep (:910-938): idepth_new[i] for the good points from JbBuffer (the applied buffer), inc[8], lambda.
 * applyStep (:939-956) is a member copy on the caller's arrays (or use nalo_init_track_frame below, which runs the whole loop).
 * ------------------------------------------------------------------------------------------------ */
int nalo_init_calc_res_and_gs(nalo_ctx* ctx, int slot_first, int slot_new, int lvl, int n, const float* u, const float* v, const float* idepth_new, const float* iR,
                              const uint8_t* isGood, const float* energy, const float* outlierTH, const double refToNew[12], const double aff[2],
                              float alphaW, float alphaK, float couplingWeight,
                              uint8_t* isGood_new, float* energy_new, float* maxstep, float* lastHessian_new, float* JbBuffer_new,
                              double* H_out, double* b_out, double* H_out_sc, double* b_out_sc, double E3[3]);
int nalo_init_do_step(nalo_ctx* ctx, int n, const uint8_t* isGood, const float* JbBuffer, const float* maxstep, const float* idepth, float lambda,
                      const float inc[8], float* idepth_new);
/* The whole initialiser behind the boundary (the CoarseInitializer object lives in the context; Pnt arrays on the host side of the library, the two
 * per-point image passes and both point selections on the device):
 * nalo_init_set_first    CoarseInitializer::setFirst (FullSystem/CoarseInitializer.cpp:785-880), call site FullSystem::addActiveFrame (FullSystem.cpp:1101):
 *     makeK from the context's calibration, level 0 selected by a fresh PixelSelector (makeMaps(., 0.03 w h, 1, false, 2), currentPotential 3: needs
 *     nalo_pixsel_set_random), levels >= 1 by makePixelStatus / gridMaxSelection (FullSystem/PixelSelector.h:38-253) with densities {0.05, 0.15, 0.5, 1} w h,
 *     Pnt construction, makeNN (:992-1069; the reference's nanoflann k-d tree, tie order included). *sparsityFactor is the reference's GLOBAL of that name
 *     (util/settings.cpp:223, initially 5; makePixelStatus keeps adapting it from call to call): in/out. numPoints[lvl] (optional) = points per level.
 * nalo_init_track_frame  CoarseInitializer::trackFrame (:81-285), call site FullSystem.cpp:1106: the coarse-to-fine LM over pose, (fixed) affine and the
 *     inverse depths with propagateDown / resetPoints / calcResAndGS / doStep / calcEC / applyStep / optReg / propagateUp; exposures are
 *     firstFrame->ab_exposure and newFrame->ab_exposure. *ok = its return value (snapped && frameID > snappedAt + 5).
 * nalo_init_get_state    thisToNext (3x4), thisToNext_aff {a, b}, snapped, frameID, snappedAt (+ the number of calcResAndGS evaluations so far); any may be NULL.
 * nalo_init_get_points   the Pnt members FullSystem::initializeFromInitializer reads (FullSystem.cpp:1601-1660: u, v, iR, my_type of level 0) and the rest
 *     of the state, for one level: *n = numPoints[lvl], the first min(cap, *n) entries are written; energy2 is [n][2], neighbours / neighboursDist [n][10];
 *     any output may be NULL. */
int nalo_init_set_first(nalo_ctx* ctx, int slot_first, int* sparsityFactor, int numPoints[NALO_MAX_LEVELS]);
int nalo_init_track_frame(nalo_ctx* ctx, int slot_new, float exposure_first, float exposure_new, int* ok);
int nalo_init_get_state(nalo_ctx* ctx, double thisToNext[12], double aff[2], int* snapped, int* frameID, int* snappedAt, int* n_evals);
/* nalo_init_set_state / nalo_init_set_points: write back what trackFrame carries from frame to frame (thisToNext, thisToNext_aff, snapped, frameID, snappedAt; per
 *     level the Pnt members idepth, idepth_new, iR, isGood, lastHessian, energy, maxstep and the *_new / iRSumNum members stale entries of which survive a frame; NULL = leave as is): resume of a checkpointed initialisation, and what
 *     the teacher-forced parity test uses to start every frame from the oracle's state. n must equal the level's point count. */
int nalo_init_set_state(nalo_ctx* ctx, const double thisToNext[12], const double aff[2], int snapped, int frameID, int snappedAt);
int nalo_init_set_points(nalo_ctx* ctx, int lvl, int n, const float* idepth, const float* idepth_new, const float* iR, const uint8_t* isGood, const float* lastHessian,
                         const float* energy2, const float* maxstep, const float* lastHessian_new, const float* energy_new2, const uint8_t* isGood_new, const float* iRSumNum);
/* nalo_init_sweep: one of trackFrame's per-level sweeps on its own, on the initialiser's current state (replaces CoarseInitializer::optReg :656-691, ::propagateUp
 *     :695-734 (lvl = srcLvl, 0 .. levels-2), ::propagateDown :736-766 (lvl = srcLvl, 1 .. levels-1), ::resetPoints :882-909; optReg reads `snapped` as set by
 *     nalo_init_set_state / trackFrame). trackFrame calls the same code; the entry point exists so that each sweep can be checked against the reference's alone. */
/* nalo_init_get_carried: the read side of nalo_init_set_points for the members nalo_init_get_points does not return (iRSumNum is recomputed by propagateUp before
 *     anything reads it and is not kept); the first min(cap, n) entries, any output may be NULL. */
int nalo_init_get_carried(nalo_ctx* ctx, int lvl, int cap, float* idepth_new, float* maxstep, float* lastHessian_new, float* energy_new2, uint8_t* isGood_new);
#define NALO_INIT_SWEEP_OPT_REG 0
#define NALO_INIT_SWEEP_PROPAGATE_UP 1
#define NALO_INIT_SWEEP_PROPAGATE_DOWN 2
#define NALO_INIT_SWEEP_RESET_POINTS 3
int nalo_init_sweep(nalo_ctx* ctx, int which, int lvl);
int nalo_init_get_points(nalo_ctx* ctx, int lvl, int cap, int* n, float* u, float* v, float* idepth, float* iR, uint8_t* isGood, float* lastHessian, float* energy2,
                         float* my_type, float* outlierTH, int* parent, float* parentDist, int* neighbours, float* neighboursDist);

/* ------------------------------------------------------------------------------------------------
 * Profiling: per-kernel HIP-event timing on the ctx stream. Names: "trk_eval", "ba_linearize",
 * "ba_sc", "ba_reduce", "ba_resub", "pyramid", "trk_lm", "imm_trace", "imm_optimize", "pixsel", "dist_bfs", "dense_bbox", "dense_map", "dense_extent", "ingest". Enable, run, then query (sync inside).
 * nalo_profile_select(ctx, name) restricts the brackets to ONE scope (NULL = all): a recorded event pair costs ~10 us of pipeline bubbles on a
 * latency-bound window, so a timed run brackets only the kernel it reports ("ba_linearize" carries its timestamps in the dispatch itself).
 * ------------------------------------------------------------------------------------------------ */
int nalo_profile_enable(nalo_ctx* ctx, int on);
int nalo_profile_select(nalo_ctx* ctx, const char* kernel);
int nalo_profile_reset(nalo_ctx* ctx);
/* bracket only one launch in `every` of the selected scopes (default 1 = all). The event pairs cost a few microseconds of pipeline each: on the KITTI-sized
 * window 8 bracketed launches per keyframe are 4-6 % of a step; a sampled average (every = 3: co-prime with the 8 launches of a keyframe) measures the same kernel
 * with a third of the perturbation. nalo_profile_get reports the bracketed launches only. */
int nalo_profile_sample(nalo_ctx* ctx, int every);
int nalo_profile_get(nalo_ctx* ctx, const char* kernel, double* total_ms, int* launches);
/* every bracketed launch of a scope since the last nalo_profile_reset, in launch order (milliseconds; at most cap values are copied, *n = how many exist):
 * the spread and the position inside a keyframe that the mean of nalo_profile_get hides */
int nalo_profile_samples(nalo_ctx* ctx, const char* kernel, float* ms, int cap, int* n);

/* Calibration of the roofline's denominator on THIS device ("fraction of the box's measured device-copy / triad bandwidth from a calibration
 * kernel in the same run"; no reference counterpart). Runs `iters` timed passes (after one untimed) of a streaming kernel over `bytes`-sized buffers on the
 * context's stream, timed with HIP events: copy_GBs = 2 * bytes / t (b[i] = a[i], 16 B per lane, grid-stride), triad_GBs = 3 * bytes / t (c[i] = a[i] + s * b[i]).
 * bytes is rounded down to a multiple of 16; >= 1 MiB. Either output may be NULL. */
int nalo_hbm_calibrate(nalo_ctx* ctx, size_t bytes, int iters, double* copy_GBs, double* triad_GBs);

#ifdef __cplusplus
}
#endif
#endif
