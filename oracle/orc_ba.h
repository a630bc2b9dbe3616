/* ORACLE — TEST INFRASTRUCTURE ONLY (see orc_common.h). parity unpinned.
 * Flat data model for the sliding-window photometric BA restatement. One residual slot per
 * (point, target frame): the reference creates at most one PointFrameResidual per such pair
 * (FullSystem/FullSystem.cpp:1335-1348, FullSystemOptPoint.cpp activation). */
#ifndef ORC_BA_H
#define ORC_BA_H
#include "orc_common.h"

typedef struct {                       /* OptimizationBackend/RawResidualJacobian.h:32-61 */
    real resF[8];
    real Jpdxi[2][6];
    real Jpdc[2][4];
    real Jpdd[2];
    real JIdx[2][8];
    real JabF[2][8];
    real JIdx2[4], JabJIdx[4], Jab2[4];    /* 2x2 row-major */
} OrcJ;

enum { ORC_IN=0, ORC_OOB=1, ORC_OUTLIER=2 };   /* FullSystem/Residuals.h:47 */

typedef struct {
    uint8_t exists, isNew, isLinearized, isActive;
    int state_state, state_NewState;
    double state_energy, state_NewEnergy, state_NewEnergyWithOutlier;
    OrcJ Jnew;                         /* PointFrameResidual::J */
    OrcJ J;                            /* EFResidual::J (after takeDataF swap) */
    real res_toZeroF[8];
    real JpJdF[8];
    float centerProjectedTo[3];
    float projectedTo[8][2];
} OrcRes;

typedef struct {
    const float* dI;                   /* level-0 AoS {I,dx,dy}, borrowed */
    double evalPT[12];                 /* worldToCam_evalPT */
    double state[10], state_zero[10], state_scaled[10], step[10], state_backup[10];
    double PRE_worldToCam[12], PRE_camToWorld[12];
    float ab_exposure, frameEnergyTH;
    int frameID;
    double prior[8], delta[8], delta_prior[8];
    double ns_pose[6][6];              /* [col i][row] HessianBlocks.cpp:79-87 */
    double ns_scale[6];
    double ns_affine[2][2];            /* [col][row 0..1] */
} OrcFrame;

typedef struct {                       /* HessianBlocks.h:80-107 */
    float PRE_RTll_0[9], PRE_tTll_0[3], PRE_KRKiTll[9], PRE_KtTll[3], PRE_RTll[9], PRE_tTll[3];
    float PRE_aff_mode[2], PRE_b0_mode;
} OrcPrecalc;

typedef struct {
    int host;
    float u, v;
    float idepth, idepth_scaled, idepth_zero, idepth_zero_scaled, step, idepth_backup;
    float color[8], weights[8];
    int hasDepthPrior;
    float priorF, deltaF;
    float bdSumF, HdiF, Hdd_accLF, Hcd_accLF[4], bd_accLF, Hdd_accAF, Hcd_accAF[4], bd_accAF;
    float idepth_hessian, maxRelBaseline; int numGoodResiduals;
    int removed;
} OrcPoint;

#define ORC_NTHREADS 6                 /* util/NumType.h:42: the reference's NUM_THREADS (default replica / worker count) */
#define ORC_MAXTHREADS 64              /* upper bound for the all-cores baseline line (orc_ba_set_options) */

typedef struct OrcBA {
    int W, P, w, h;
    /* CalibHessian (HessianBlocks.h:364-395) */
    double c_value[4], c_value_zero[4], c_value_scaled[4], c_step[4], c_value_backup[4];
    float c_scaledf[4], c_scaledi[4];
    OrcFrame* frames; OrcPoint* pts; OrcRes* res; OrcPrecalc* pre;
    /* EnergyFunctional */
    double *adHost, *adTarget;         /* [W*W][64] index h + t*W */
    float *adHostF, *adTargetF, *adHTdeltaF; /* adHTdeltaF [W*W][8] */
    float cDeltaF[4]; double cPrior[4];
    double *HM, *bM;                   /* (8W+4)^2 */
    double *lastX; int resInA, resInL, resInM;
    /* accumulators: [tid][...] */
    OrcTier (*accTopA)[ORC_MAXW*ORC_MAXW], (*accTopL)[ORC_MAXW*ORC_MAXW];
    OrcTier *accD[ORC_MAXTHREADS], *accE[ORC_MAXTHREADS], *accEB[ORC_MAXTHREADS], accHcc[ORC_MAXTHREADS], accbc[ORC_MAXTHREADS];
    int nres[ORC_MAXTHREADS];
    int nt_alloc;                      /* accumulator replicas allocated (>= nthreads_used) */
    int nthreads_used;                 /* 1 .. nt_alloc; 6 = the reference */
    int linearize_mt;                  /* 0 = single-threaded linearizeAll as in this fork (FullSystemOptimize.cpp:154-164); 1 = chunked over the workers (upstream DSO) */
    int never_break;
    /* settings a caller may change (util/settings.cpp:71,128-129,74): defaults 1, 1e12, 1e8, 1 */
    int force_accept_step; double affine_opt_mode_a, affine_opt_mode_b; int min_opt_iterations; int n_rejected;
    double t_linearize, t_accumulate, t_solve, t_other;   /* wall seconds, for the baseline report */
} OrcBA;

#endif
