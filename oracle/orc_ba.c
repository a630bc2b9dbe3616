/* ORACLE — TEST INFRASTRUCTURE ONLY (see orc_common.h). parity unpinned.
 * Restates the sliding-window photometric BA:
 *   PointFrameResidual::linearize / applyRes        FullSystem/Residuals.cpp:78-274, 306-328
 *   projectPoint x2                                 FullSystem/ResidualProjections.h:47-87
 *   EFResidual::takeDataF / fixLinearizationF       OptimizationBackend/EnergyFunctionalStructs.cpp:39-50, 89-115
 *   FrameFramePrecalc::set                          FullSystem/HessianBlocks.cpp:192-222
 *   EnergyFunctional::setAdjointsF / setDeltaF      OptimizationBackend/EnergyFunctional.cpp:46-106, 171-194
 *   AccumulatedTopHessianSSE::addPoint<0/1/2>       OptimizationBackend/AccumulatedTopHessian.cpp:39-162
 *   AccumulatorApprox                               OptimizationBackend/MatrixAccumulators.h:595-972
 *   stitchDouble (top)                              AccumulatedTopHessian.cpp:171-303, .h:91-139
 *   AccumulatedSCHessianSSE::addPoint / stitch      OptimizationBackend/AccumulatedSCHessian.cpp:34-219
 *   solveSystemF / orthogonalize / resubstitute     EnergyFunctional.cpp:776-914, 719-773, 263-317
 *   marginalizePointsF                              EnergyFunctional.cpp:615-676
 *   linearizeAll / setNewFrameEnergyTH / doStepFromBackup / optimize
 *                                                   FullSystem/FullSystemOptimize.cpp:52-211, 217-299, 398-602
 */
#include "orc_ba.h"
#include <time.h>
#include <pthread.h>

static double now_s(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC,&ts); return ts.tv_sec+1e-9*ts.tv_nsec; }
#define NF(ba) ((ba)->W)
#define NDIM(ba) ((ba)->W*8+ORC_CPARS)
#define RES(ba,p,t) (&(ba)->res[(size_t)(p)*(ba)->W+(t)])
#define RSQRT(x) ((sizeof(real)==4) ? (real)sqrtf((float)(x)) : (real)sqrt((double)(x)))

/* ---------------------------------------------------------------- calib: HessianBlocks.h:364-395 */
static void calib_set_value(OrcBA* ba, const double v[4]) {
    for (int i=0;i<4;i++) ba->c_value[i]=v[i];
    ba->c_value_scaled[0]=SCALE_F*v[0]; ba->c_value_scaled[1]=SCALE_F*v[1]; ba->c_value_scaled[2]=SCALE_C*v[2]; ba->c_value_scaled[3]=SCALE_C*v[3];
    for (int i=0;i<4;i++) ba->c_scaledf[i]=(float)ba->c_value_scaled[i];
    ba->c_scaledi[0]=1.0f/ba->c_scaledf[0]; ba->c_scaledi[1]=1.0f/ba->c_scaledf[1];
    ba->c_scaledi[2]=-ba->c_scaledf[2]/ba->c_scaledf[0]; ba->c_scaledi[3]=-ba->c_scaledf[3]/ba->c_scaledf[1];
}

/* ---------------------------------------------------------------- frames: HessianBlocks.h:194-242 */
static void frame_set_state(OrcFrame* f, const double st[10]) {
    memcpy(f->state, st, sizeof(double)*10);
    for (int i=0;i<3;i++) f->state_scaled[i]=SCALE_XI_TRANS*st[i];
    for (int i=3;i<6;i++) f->state_scaled[i]=SCALE_XI_ROT*st[i];
    f->state_scaled[6]=SCALE_A*st[6]; f->state_scaled[7]=SCALE_B*st[7]; f->state_scaled[8]=SCALE_A*st[8]; f->state_scaled[9]=SCALE_B*st[9];
    double E[12]; orc_se3_exp(f->state_scaled, E);
    orc_se3_mul(E, f->evalPT, f->PRE_worldToCam); orc_se3_inv(f->PRE_worldToCam, f->PRE_camToWorld);
}
/* FrameHessian::setStateZero, HessianBlocks.cpp:73-106 */
static void frame_set_state_zero(OrcFrame* f, const double sz[10]) {
    memcpy(f->state_zero, sz, sizeof(double)*10);
    double inv[12]; orc_se3_inv(f->evalPT, inv);
    for (int i=0;i<6;i++) {
        double eps[6]={0,0,0,0,0,0}, Ep[12], Em[12], A[12], B[12], lp[6], lm[6];
        eps[i]=1e-3; orc_se3_exp(eps,Ep); eps[i]=-1e-3; orc_se3_exp(eps,Em);
        orc_se3_mul(f->evalPT,Ep,A); orc_se3_mul(A,inv,A); orc_se3_mul(f->evalPT,Em,B); orc_se3_mul(B,inv,B);
        orc_se3_log(A,lp); orc_se3_log(B,lm);
        for (int r=0;r<6;r++) f->ns_pose[i][r]=(lp[r]-lm[r])/(2e-3);
    }
    { double P[12], M[12], lp[6], lm[6]; memcpy(P,f->evalPT,sizeof(P)); memcpy(M,f->evalPT,sizeof(M));
      P[3]*=1.00001; P[7]*=1.00001; P[11]*=1.00001; M[3]/=1.00001; M[7]/=1.00001; M[11]/=1.00001;
      orc_se3_mul(P,inv,P); orc_se3_mul(M,inv,M); orc_se3_log(P,lp); orc_se3_log(M,lm);
      for (int r=0;r<6;r++) f->ns_scale[r]=(lp[r]-lm[r])/(2e-3); }
    f->ns_affine[0][0]=1; f->ns_affine[0][1]=0;
    f->ns_affine[1][0]=0; f->ns_affine[1][1]=expf((float)(sz[6]*SCALE_A))*f->ab_exposure;
}
/* FrameHessian::getPrior, HessianBlocks.h:321-350 */
static void frame_take_data(const OrcBA* ba, OrcFrame* f) {
    double p[8]={0,0,0,0,0,0,0,0};
    if (f->frameID==0) { for (int i=0;i<3;i++) p[i]=SETTING_INITIAL_TRANS_PRIOR; for (int i=3;i<6;i++) p[i]=SETTING_INITIAL_ROT_PRIOR;
        p[6]=SETTING_INITIAL_AFFA_PRIOR; p[7]=SETTING_INITIAL_AFFB_PRIOR; }
    else { p[6]=ba->affine_opt_mode_a<0 ? SETTING_INITIAL_AFFA_PRIOR : ba->affine_opt_mode_a;            /* :292-302 */
           p[7]=ba->affine_opt_mode_b<0 ? SETTING_INITIAL_AFFB_PRIOR : ba->affine_opt_mode_b; }
    for (int i=0;i<8;i++) { f->prior[i]=p[i]; f->delta[i]=f->state[i]-f->state_zero[i]; f->delta_prior[i]=f->state[i]; }
}

/* ---------------------------------------------------------------- create / setters */
static void alloc_replicas(OrcBA* ba, int upto) {          /* per-worker accumulator replicas (AccumulatedTopHessian.h:144-149, AccumulatedSCHessian.h) */
    int W=ba->W;
    for (int t=ba->nt_alloc;t<upto;t++) {
        for (int i=0;i<W*W;i++) { orc_tier_init(&ba->accTopA[t][i],91); orc_tier_init(&ba->accTopL[t][i],91); }
        ba->accD[t]=(OrcTier*)calloc(W*W*W,sizeof(OrcTier)); ba->accE[t]=(OrcTier*)calloc(W*W,sizeof(OrcTier)); ba->accEB[t]=(OrcTier*)calloc(W*W,sizeof(OrcTier));
        for (int i=0;i<W*W*W;i++) orc_tier_init(&ba->accD[t][i],64);
        for (int i=0;i<W*W;i++) { orc_tier_init(&ba->accE[t][i],32); orc_tier_init(&ba->accEB[t][i],8); }
        orc_tier_init(&ba->accHcc[t],16); orc_tier_init(&ba->accbc[t],4);
    }
    if (upto>ba->nt_alloc) ba->nt_alloc=upto;
}
OrcBA* orc_ba_create(int W, int P, int w, int h, double fx, double fy, double cx, double cy) {
    OrcBA* ba=(OrcBA*)calloc(1,sizeof(OrcBA));
    ba->W=W; ba->P=P; ba->w=w; ba->h=h;
    double vs[4]={fx,fy,cx,cy}, v[4]={fx/SCALE_F, fy/SCALE_F, cx/SCALE_C, cy/SCALE_C};
    /* CalibHessian(): setValueScaled(initial) then value_zero=value (HessianBlocks.h:347-361) */
    v[0]=(1.0f/SCALE_F)*vs[0]; v[1]=(1.0f/SCALE_F)*vs[1]; v[2]=(1.0f/SCALE_C)*vs[2]; v[3]=(1.0f/SCALE_C)*vs[3];
    calib_set_value(ba, v);
    for (int i=0;i<4;i++) { ba->c_value_scaled[i]=vs[i]; ba->c_scaledf[i]=(float)vs[i]; ba->c_value_zero[i]=ba->c_value[i]; }
    ba->c_scaledi[0]=1.0f/ba->c_scaledf[0]; ba->c_scaledi[1]=1.0f/ba->c_scaledf[1];
    ba->c_scaledi[2]=-ba->c_scaledf[2]/ba->c_scaledf[0]; ba->c_scaledi[3]=-ba->c_scaledf[3]/ba->c_scaledf[1];
    ba->frames=(OrcFrame*)calloc(W,sizeof(OrcFrame)); ba->pts=(OrcPoint*)calloc(P,sizeof(OrcPoint));
    ba->res=(OrcRes*)calloc((size_t)P*W,sizeof(OrcRes)); ba->pre=(OrcPrecalc*)calloc(W*W,sizeof(OrcPrecalc));
    ba->adHost=(double*)calloc(W*W*64,8); ba->adTarget=(double*)calloc(W*W*64,8);
    ba->adHostF=(float*)calloc(W*W*64,4); ba->adTargetF=(float*)calloc(W*W*64,4); ba->adHTdeltaF=(float*)calloc(W*W*8,4);
    int n=W*8+ORC_CPARS; ba->HM=(double*)calloc(n*n,8); ba->bM=(double*)calloc(n,8); ba->lastX=(double*)calloc(n,8);
    ba->accTopA=calloc(ORC_MAXTHREADS,sizeof(*ba->accTopA)); ba->accTopL=calloc(ORC_MAXTHREADS,sizeof(*ba->accTopL));
    ba->nt_alloc=0; alloc_replicas(ba, ORC_NTHREADS);
    for (int i=0;i<4;i++) ba->cPrior[i]=SETTING_INITIAL_CALIB_HESSIAN;
    ba->nthreads_used=ORC_NTHREADS;
    ba->force_accept_step=1; ba->affine_opt_mode_a=SETTING_AFFINE_OPT_MODE_A; ba->affine_opt_mode_b=SETTING_AFFINE_OPT_MODE_B; ba->min_opt_iterations=1;
    return ba;
}
void orc_ba_destroy(OrcBA* ba) {
    int W=ba->W;
    for (int t=0;t<ba->nt_alloc;t++) {
        for (int i=0;i<W*W;i++) { orc_tier_free(&ba->accTopA[t][i]); orc_tier_free(&ba->accTopL[t][i]); orc_tier_free(&ba->accE[t][i]); orc_tier_free(&ba->accEB[t][i]); }
        for (int i=0;i<W*W*W;i++) orc_tier_free(&ba->accD[t][i]);
        free(ba->accD[t]); free(ba->accE[t]); free(ba->accEB[t]); orc_tier_free(&ba->accHcc[t]); orc_tier_free(&ba->accbc[t]);
    }
    free(ba->accTopA); free(ba->accTopL);
    free(ba->frames); free(ba->pts); free(ba->res); free(ba->pre); free(ba->adHost); free(ba->adTarget);
    free(ba->adHostF); free(ba->adTargetF); free(ba->adHTdeltaF); free(ba->HM); free(ba->bM); free(ba->lastX); free(ba);
}
/* FrameHessian::setEvalPT_scaled(worldToCam_evalPT, aff_g2l) (HessianBlocks.h:247-255) followed by an optional
 * non-zero unscaled state[0:6] (older window frames keep their evalPT and carry a state, FEJ). */
void orc_ba_set_frame(OrcBA* ba, int i, const float* dI, const double evalPT[12], double aff_a, double aff_b,
                      float exposure, float frameEnergyTH, int frameID, const double state6[6]) {
    OrcFrame* f=&ba->frames[i];
    f->dI=dI; memcpy(f->evalPT,evalPT,sizeof(double)*12); f->ab_exposure=exposure; f->frameEnergyTH=frameEnergyTH; f->frameID=frameID;
    double st[10]={0,0,0,0,0,0, (1.0f/SCALE_A)*aff_a, (1.0f/SCALE_B)*aff_b, 0,0};
    frame_set_state(f, st);
    frame_set_state_zero(f, f->state);
    if (state6) { for (int k=0;k<6;k++) st[k]=state6[k]; frame_set_state(f, st); }
    frame_take_data(ba,f);
}
/* General form for a window that carries state between keyframes: evalPT (FEJ point), state and state_zero as the running system holds them
 * (FrameHessian::setStateZero / setState, HessianBlocks.h:208-255, HessianBlocks.cpp:73-106). Both are the UNSCALED 10-vectors. */
void orc_ba_set_frame_full(OrcBA* ba, int i, const float* dI, const double evalPT[12], const double state[10], const double state_zero[10],
                           float exposure, float frameEnergyTH, int frameID) {
    OrcFrame* f=&ba->frames[i];
    f->dI=dI; memcpy(f->evalPT,evalPT,sizeof(double)*12); f->ab_exposure=exposure; f->frameEnergyTH=frameEnergyTH; f->frameID=frameID;
    frame_set_state(f, state_zero);
    frame_set_state_zero(f, state_zero);
    frame_set_state(f, state);
    frame_take_data(ba,f);
}
/* PointHessian::idepth_zero differing from idepth (setIdepthZero / setIdepth are separate setters, HessianBlocks.h:449-460) */
void orc_ba_set_idepth_zero(OrcBA* ba, const float* idepth_zero) {
    for (int p=0;p<ba->P;p++) { OrcPoint* pt=&ba->pts[p]; pt->idepth_zero=idepth_zero[p]; pt->idepth_zero_scaled=SCALE_IDEPTH*idepth_zero[p]; pt->deltaF=pt->idepth-pt->idepth_zero; }
}
/* CalibHessian::value_zero differing from value (HessianBlocks.h:364-395): vs = {fx,fy,cx,cy} of the linearisation point, scaled units */
void orc_ba_set_calib_zero(OrcBA* ba, const double vs[4]) {
    ba->c_value_zero[0]=(1.0f/SCALE_F)*vs[0]; ba->c_value_zero[1]=(1.0f/SCALE_F)*vs[1]; ba->c_value_zero[2]=(1.0f/SCALE_C)*vs[2]; ba->c_value_zero[3]=(1.0f/SCALE_C)*vs[3];
}
void orc_ba_set_points(OrcBA* ba, const int* host, const float* u, const float* v, const float* idepth,
                       const float* color, const float* weights, const int* hasDepthPrior) {
    for (int p=0;p<ba->P;p++) {
        OrcPoint* pt=&ba->pts[p]; memset(pt,0,sizeof(*pt));
        pt->host=host[p]; pt->u=u[p]; pt->v=v[p];
        pt->idepth=idepth[p]; pt->idepth_scaled=SCALE_IDEPTH*idepth[p]; pt->idepth_zero=idepth[p]; pt->idepth_zero_scaled=SCALE_IDEPTH*idepth[p];
        memcpy(pt->color,color+8*p,32); memcpy(pt->weights,weights+8*p,32);
        pt->hasDepthPrior = hasDepthPrior ? hasDepthPrior[p] : 0;
        pt->priorF = pt->hasDepthPrior ? SETTING_IDEPTH_FIX_PRIOR*SCALE_IDEPTH*SCALE_IDEPTH : 0;   /* EFPoint::takeData */
        pt->deltaF = pt->idepth-pt->idepth_zero;
    }
}
/* exists[p*W+t] != 0 -> PointFrameResidual(point p, host, target t) with resetOOB state (Residuals.h:88-94) */
void orc_ba_set_residuals(OrcBA* ba, const uint8_t* exists) {
    for (size_t i=0;i<(size_t)ba->P*ba->W;i++) {
        OrcRes* r=&ba->res[i]; memset(r,0,sizeof(*r));
        r->exists = exists[i] ? 1 : 0; r->isNew=1; r->state_state=ORC_IN; r->state_NewState=ORC_OUTLIER;
    }
}

/* ---------------------------------------------------------------- precalc / adjoints / delta */
static void mat3f_mul(const float* A, const float* B, float* C) {
    for (int i=0;i<3;i++) for (int j=0;j<3;j++) C[i*3+j]=A[i*3]*B[j]+A[i*3+1]*B[3+j]+A[i*3+2]*B[6+j];
}
/* FrameFramePrecalc::set, HessianBlocks.cpp:192-222 */
static void precalc_set(OrcBA* ba, int h, int t) {
    OrcPrecalc* pc=&ba->pre[h*ba->W+t]; OrcFrame *host=&ba->frames[h], *target=&ba->frames[t];
    double inv[12], l0[12], ll[12];
    orc_se3_inv(host->evalPT,inv); orc_se3_mul(target->evalPT,inv,l0);
    for (int i=0;i<3;i++) { for (int j=0;j<3;j++) pc->PRE_RTll_0[i*3+j]=(float)l0[i*4+j]; pc->PRE_tTll_0[i]=(float)l0[i*4+3]; }
    orc_se3_mul(target->PRE_worldToCam, host->PRE_camToWorld, ll);
    for (int i=0;i<3;i++) { for (int j=0;j<3;j++) pc->PRE_RTll[i*3+j]=(float)ll[i*4+j]; pc->PRE_tTll[i]=(float)ll[i*4+3]; }
    float fx=ba->c_scaledf[0], fy=ba->c_scaledf[1], cx=ba->c_scaledf[2], cy=ba->c_scaledf[3];
    float K[9]={fx,0,cx, 0,fy,cy, 0,0,1}, Ki[9]={1.0f/fx,0,-cx/fx, 0,1.0f/fy,-cy/fy, 0,0,1}, KR[9];
    mat3f_mul(K,pc->PRE_RTll,KR); mat3f_mul(KR,Ki,pc->PRE_KRKiTll);
    for (int i=0;i<3;i++) pc->PRE_KtTll[i]=K[i*3]*pc->PRE_tTll[0]+K[i*3+1]*pc->PRE_tTll[1]+K[i*3+2]*pc->PRE_tTll[2];
    double a[2]; orc_aff_from_to(host->ab_exposure,target->ab_exposure,host->state_scaled[6],host->state_scaled[7],target->state_scaled[6],target->state_scaled[7],a);
    pc->PRE_aff_mode[0]=(float)a[0]; pc->PRE_aff_mode[1]=(float)a[1];
    pc->PRE_b0_mode=(float)(host->state_zero[7]*SCALE_B);
}
/* EnergyFunctional::setAdjointsF, EnergyFunctional.cpp:46-106 */
void orc_ba_set_adjoints(OrcBA* ba) {
    int W=ba->W;
    for (int h=0;h<W;h++) for (int t=0;t<W;t++) {
        OrcFrame *host=&ba->frames[h], *target=&ba->frames[t];
        double inv[12], h2t[12], Ad[36], AH[64], AT[64];
        orc_se3_inv(host->evalPT,inv); orc_se3_mul(target->evalPT,inv,h2t); orc_se3_adj(h2t,Ad);
        memset(AH,0,sizeof(AH)); memset(AT,0,sizeof(AT));
        for (int i=0;i<8;i++) { AH[i*8+i]=1; AT[i*8+i]=1; }
        for (int i=0;i<6;i++) for (int j=0;j<6;j++) AH[i*8+j] = -Ad[j*6+i];
        double a[2]; orc_aff_from_to(host->ab_exposure,target->ab_exposure,host->state_zero[6]*SCALE_A,host->state_zero[7]*SCALE_B,
                                    target->state_zero[6]*SCALE_A,target->state_zero[7]*SCALE_B,a);
        float affLL0=(float)a[0];
        AT[6*8+6]=-affLL0; AH[6*8+6]=affLL0; AT[7*8+7]=-1; AH[7*8+7]=affLL0;
        for (int j=0;j<8;j++) {
            for (int i=0;i<3;i++) { AH[i*8+j]*=SCALE_XI_TRANS; AT[i*8+j]*=SCALE_XI_TRANS; }
            for (int i=3;i<6;i++) { AH[i*8+j]*=SCALE_XI_ROT; AT[i*8+j]*=SCALE_XI_ROT; }
            AH[6*8+j]*=SCALE_A; AT[6*8+j]*=SCALE_A; AH[7*8+j]*=SCALE_B; AT[7*8+j]*=SCALE_B;
        }
        int idx=h+t*W;
        memcpy(ba->adHost+idx*64,AH,sizeof(AH)); memcpy(ba->adTarget+idx*64,AT,sizeof(AT));
        for (int k=0;k<64;k++) { ba->adHostF[idx*64+k]=(float)AH[k]; ba->adTargetF[idx*64+k]=(float)AT[k]; }
    }
}
/* EnergyFunctional::setDeltaF, EnergyFunctional.cpp:171-194 */
static void set_delta(OrcBA* ba) {
    int W=ba->W;
    for (int h=0;h<W;h++) for (int t=0;t<W;t++) {
        int idx=h+t*W; float dh[8], dt[8];
        for (int i=0;i<8;i++) { dh[i]=(float)(ba->frames[h].state[i]-ba->frames[h].state_zero[i]); dt[i]=(float)(ba->frames[t].state[i]-ba->frames[t].state_zero[i]); }
        for (int j=0;j<8;j++) { float s1=0, s2=0; for (int i=0;i<8;i++) { s1+=dh[i]*ba->adHostF[idx*64+i*8+j]; s2+=dt[i]*ba->adTargetF[idx*64+i*8+j]; }
            ba->adHTdeltaF[idx*8+j]=s1+s2; }
    }
    for (int i=0;i<4;i++) ba->cDeltaF[i]=(float)(ba->c_value[i]-ba->c_value_zero[i]);
    for (int f=0;f<W;f++) frame_take_data(ba,&ba->frames[f]);
    for (int p=0;p<ba->P;p++) ba->pts[p].deltaF = ba->pts[p].idepth-ba->pts[p].idepth_zero;
}
/* FullSystem::setPrecalcValues, FullSystem.cpp:1694-1704 */
void orc_ba_set_precalc(OrcBA* ba) {
    for (int h=0;h<ba->W;h++) for (int t=0;t<ba->W;t++) precalc_set(ba,h,t);
    set_delta(ba);
}

/* ---------------------------------------------------------------- a5: linearize, Residuals.cpp:78-274 */
static inline void interp33(const float* mat, real x, real y, int width, real out[3]) {   /* globalFuncs.h:75-89 */
    int ix=(int)x, iy=(int)y; real dx=x-ix, dy=y-iy, dxdy=dx*dy; const float* bp=mat+3*(ix+iy*width);
    for (int c=0;c<3;c++) out[c]=dxdy*(real)bp[3*(1+width)+c]+(dy-dxdy)*(real)bp[3*width+c]+(dx-dxdy)*(real)bp[3+c]+(1-dx-dy+dxdy)*(real)bp[c];
}
static double linearize(OrcBA* ba, int p, int t) {
    OrcRes* r=RES(ba,p,t); OrcPoint* pt=&ba->pts[p]; OrcJ* J=&r->Jnew;
    r->state_NewEnergyWithOutlier=-1;
    if (r->state_state==ORC_OOB) { r->state_NewState=ORC_OOB; return r->state_energy; }
    const OrcPrecalc* pc=&ba->pre[pt->host*ba->W+t];
    OrcFrame *host=&ba->frames[pt->host], *target=&ba->frames[t];
    const float* dIl=target->dI;
    real KRKi[9], Kt[3], R0[9], t0[3];
    for (int i=0;i<9;i++) { KRKi[i]=pc->PRE_KRKiTll[i]; R0[i]=pc->PRE_RTll_0[i]; }
    for (int i=0;i<3;i++) { Kt[i]=pc->PRE_KtTll[i]; t0[i]=pc->PRE_tTll_0[i]; }
    real affLL0=pc->PRE_aff_mode[0], affLL1=pc->PRE_aff_mode[1], b0=pc->PRE_b0_mode;
    real fxl=ba->c_scaledf[0], fyl=ba->c_scaledf[1], cxl=ba->c_scaledf[2], cyl=ba->c_scaledf[3], fxli=ba->c_scaledi[0], fyli=ba->c_scaledi[1];
    real wM3G=ba->w-3, hM3G=ba->h-3;
    real d_xi_x[6], d_xi_y[6], d_C_x[4], d_C_y[4], d_d_x, d_d_y;
    {   /* projectPoint, ResidualProjections.h:61-87 at idepth_zero_scaled */
        real KliP[3]={((real)pt->u+0-cxl)*fxli, ((real)pt->v+0-cyl)*fyli, 1};
        real idz=pt->idepth_zero_scaled;
        real ptp0=R0[0]*KliP[0]+R0[1]*KliP[1]+R0[2]*KliP[2]+t0[0]*idz, ptp1=R0[3]*KliP[0]+R0[4]*KliP[1]+R0[5]*KliP[2]+t0[1]*idz,
             ptp2=R0[6]*KliP[0]+R0[7]*KliP[1]+R0[8]*KliP[2]+t0[2]*idz;
        real drescale=(real)1.0/ptp2, new_idepth=idz*drescale;
        if (!(drescale>0)) { r->state_NewState=ORC_OOB; return r->state_energy; }
        real u=ptp0*drescale, v=ptp1*drescale, Ku=u*fxl+cxl, Kv=v*fyl+cyl;
        if (!(Ku>(real)1.1f && Kv>(real)1.1f && Ku<wM3G && Kv<hM3G)) { r->state_NewState=ORC_OOB; return r->state_energy; }
        r->centerProjectedTo[0]=(float)Ku; r->centerProjectedTo[1]=(float)Kv; r->centerProjectedTo[2]=(float)new_idepth;
        d_d_x = drescale*(t0[0]-t0[2]*u)*SCALE_IDEPTH*fxl;                         /* :116-117 */
        d_d_y = drescale*(t0[1]-t0[2]*v)*SCALE_IDEPTH*fyl;
        d_C_x[2]=drescale*(R0[6]*u-R0[0]); d_C_x[3]=fxl*drescale*(R0[7]*u-R0[1])*fyli;   /* :123-131 */
        d_C_x[0]=KliP[0]*d_C_x[2]; d_C_x[1]=KliP[1]*d_C_x[3];
        d_C_y[2]=fyl*drescale*(R0[6]*v-R0[3])*fxli; d_C_y[3]=drescale*(R0[7]*v-R0[4]);
        d_C_y[0]=KliP[0]*d_C_y[2]; d_C_y[1]=KliP[1]*d_C_y[3];
        d_C_x[0]=(d_C_x[0]+u)*SCALE_F; d_C_x[1]*=SCALE_F; d_C_x[2]=(d_C_x[2]+1)*SCALE_C; d_C_x[3]*=SCALE_C;   /* :133-141 */
        d_C_y[0]*=SCALE_F; d_C_y[1]=(d_C_y[1]+v)*SCALE_F; d_C_y[2]*=SCALE_C; d_C_y[3]=(d_C_y[3]+1)*SCALE_C;
        d_xi_x[0]=new_idepth*fxl; d_xi_x[1]=0; d_xi_x[2]=-new_idepth*u*fxl; d_xi_x[3]=-u*v*fxl; d_xi_x[4]=(1+u*u)*fxl; d_xi_x[5]=-v*fxl;   /* :144-156 */
        d_xi_y[0]=0; d_xi_y[1]=new_idepth*fyl; d_xi_y[2]=-new_idepth*v*fyl; d_xi_y[3]=-(1+v*v)*fyl; d_xi_y[4]=u*v*fyl; d_xi_y[5]=u*fyl;
    }
    for (int i=0;i<6;i++) { J->Jpdxi[0][i]=d_xi_x[i]; J->Jpdxi[1][i]=d_xi_y[i]; }
    for (int i=0;i<4;i++) { J->Jpdc[0][i]=d_C_x[i]; J->Jpdc[1][i]=d_C_y[i]; }
    J->Jpdd[0]=d_d_x; J->Jpdd[1]=d_d_y;
    real JIdxJIdx_00=0,JIdxJIdx_11=0,JIdxJIdx_10=0, JabJIdx_00=0,JabJIdx_01=0,JabJIdx_10=0,JabJIdx_11=0, JabJab_00=0,JabJab_01=0,JabJab_11=0, wJI2_sum=0;
    real energyLeft=0, huber=SETTING_HUBER_TH;
    for (int idx=0; idx<ORC_PATTERN_NUM; idx++) {
        real x=(real)pt->u+orc_patternP[idx][0], y=(real)pt->v+orc_patternP[idx][1], id=pt->idepth_scaled;   /* projectPoint :47-57 */
        real q0=KRKi[0]*x+KRKi[1]*y+KRKi[2]+Kt[0]*id, q1=KRKi[3]*x+KRKi[4]*y+KRKi[5]+Kt[1]*id, q2=KRKi[6]*x+KRKi[7]*y+KRKi[8]+Kt[2]*id;
        real Ku=q0/q2, Kv=q1/q2;
        if (!(Ku>(real)1.1f && Kv>(real)1.1f && Ku<wM3G && Kv<hM3G)) { r->state_NewState=ORC_OOB; return r->state_energy; }
        r->projectedTo[idx][0]=(float)Ku; r->projectedTo[idx][1]=(float)Kv;
        real hit[3]; interp33(dIl,Ku,Kv,ba->w,hit);
        real residual = hit[0]-(real)(float)(affLL0*(real)pt->color[idx]+affLL1);
        real drdA=((real)pt->color[idx]-b0);
        if (!isfinite((float)hit[0])) { r->state_NewState=ORC_OOB; return r->state_energy; }
        real w=RSQRT((real)SETTING_OUTLIER_TH_SUMCOMP/((real)SETTING_OUTLIER_TH_SUMCOMP+(hit[1]*hit[1]+hit[2]*hit[2])));
        w=(real)0.5f*(w+(real)pt->weights[idx]);
        real ar=residual<0?-residual:residual;
        real hw = ar<huber ? 1 : huber/ar;
        energyLeft += w*w*hw*residual*residual*(2-hw);
        if (hw<1) hw = RSQRT(hw);
        hw=hw*w;
        hit[1]*=hw; hit[2]*=hw;
        J->resF[idx]=residual*hw; J->JIdx[0][idx]=hit[1]; J->JIdx[1][idx]=hit[2]; J->JabF[0][idx]=drdA*hw; J->JabF[1][idx]=hw;
        JIdxJIdx_00+=hit[1]*hit[1]; JIdxJIdx_11+=hit[2]*hit[2]; JIdxJIdx_10+=hit[1]*hit[2];
        JabJIdx_00+=drdA*hw*hit[1]; JabJIdx_01+=drdA*hw*hit[2]; JabJIdx_10+=hw*hit[1]; JabJIdx_11+=hw*hit[2];
        JabJab_00+=drdA*drdA*hw*hw; JabJab_01+=drdA*hw*hw; JabJab_11+=hw*hw;
        wJI2_sum += hw*hw*(hit[1]*hit[1]+hit[2]*hit[2]);
        if (ba->affine_opt_mode_a<0) J->JabF[0][idx]=0;                            /* Residuals.cpp:241-242 */
        if (ba->affine_opt_mode_b<0) J->JabF[1][idx]=0;
    }
    J->JIdx2[0]=JIdxJIdx_00; J->JIdx2[1]=JIdxJIdx_10; J->JIdx2[2]=JIdxJIdx_10; J->JIdx2[3]=JIdxJIdx_11;
    J->JabJIdx[0]=JabJIdx_00; J->JabJIdx[1]=JabJIdx_01; J->JabJIdx[2]=JabJIdx_10; J->JabJIdx[3]=JabJIdx_11;
    J->Jab2[0]=JabJab_00; J->Jab2[1]=JabJab_01; J->Jab2[2]=JabJab_01; J->Jab2[3]=JabJab_11;
    r->state_NewEnergyWithOutlier=energyLeft;
    float th = host->frameEnergyTH > target->frameEnergyTH ? host->frameEnergyTH : target->frameEnergyTH;
    if ((float)energyLeft > th || wJI2_sum < 2) { energyLeft=th; r->state_NewState=ORC_OUTLIER; }
    else r->state_NewState=ORC_IN;
    r->state_NewEnergy=energyLeft;
    return energyLeft;
}
/* EFResidual::takeDataF, EnergyFunctionalStructs.cpp:39-50 */
static void take_data(OrcRes* r) {
    OrcJ tmp=r->J; r->J=r->Jnew; r->Jnew=tmp;
    OrcJ* J=&r->J;
    real a0=J->JIdx2[0]*J->Jpdd[0]+J->JIdx2[1]*J->Jpdd[1], a1=J->JIdx2[2]*J->Jpdd[0]+J->JIdx2[3]*J->Jpdd[1];
    for (int i=0;i<6;i++) r->JpJdF[i]=J->Jpdxi[0][i]*a0+J->Jpdxi[1][i]*a1;
    r->JpJdF[6]=J->JabJIdx[0]*J->Jpdd[0]+J->JabJIdx[1]*J->Jpdd[1];
    r->JpJdF[7]=J->JabJIdx[2]*J->Jpdd[0]+J->JabJIdx[3]*J->Jpdd[1];
}
/* PointFrameResidual::applyRes(true), Residuals.cpp:306-328 */
static void apply_res(OrcRes* r) {
    if (r->state_state==ORC_OOB) return;
    if (r->state_NewState==ORC_IN) { r->isActive=1; take_data(r); } else r->isActive=0;
    r->state_state=r->state_NewState; r->state_energy=r->state_NewEnergy;
}
/* EFResidual::fixLinearizationF, EnergyFunctionalStructs.cpp:89-115 */
static void fix_linearization(OrcBA* ba, int p, int t) {
    OrcRes* r=RES(ba,p,t); OrcJ* J=&r->J; OrcPoint* pt=&ba->pts[p];
    const float* dp=ba->adHTdeltaF+(pt->host+ba->W*t)*8;
    real jx=0, jy=0;
    for (int i=0;i<6;i++) { jx+=J->Jpdxi[0][i]*(real)dp[i]; jy+=J->Jpdxi[1][i]*(real)dp[i]; }
    { real cx_=0, cy_=0; for (int i=0;i<4;i++) { cx_+=J->Jpdc[0][i]*(real)ba->cDeltaF[i]; cy_+=J->Jpdc[1][i]*(real)ba->cDeltaF[i]; }
      jx = jx + cx_ + J->Jpdd[0]*(real)pt->deltaF; jy = jy + cy_ + J->Jpdd[1]*(real)pt->deltaF; }
    for (int i=0;i<8;i++) {
        real rtz=J->resF[i];
        rtz -= J->JIdx[0][i]*jx; rtz -= J->JIdx[1][i]*jy; rtz -= J->JabF[0][i]*(real)dp[6]; rtz -= J->JabF[1][i]*(real)dp[7];
        r->res_toZeroF[i]=rtz;
    }
    r->isLinearized=1;
}

/* FullSystem::setNewFrameEnergyTH, FullSystemOptimize.cpp:95-143 */
static int cmp_float(const void* a, const void* b) { float x=*(const float*)a, y=*(const float*)b; return (x>y)-(x<y); }
static void set_new_frame_energy_th(OrcBA* ba) {
    int W=ba->W, t=W-1, n=0; float* v=(float*)malloc(sizeof(float)*(ba->P+1));
    for (int p=0;p<ba->P;p++) { OrcRes* r=RES(ba,p,t); if (r->exists && !r->isLinearized && !ba->pts[p].removed && r->state_NewEnergyWithOutlier>=0) v[n++]=(float)r->state_NewEnergyWithOutlier; }
    OrcFrame* nf=&ba->frames[t];
    if (n==0) { nf->frameEnergyTH=12*12*ORC_PATTERN_NUM; free(v); return; }
    int nth=(int)(SETTING_FRAME_ENERGY_TH_N*n);
    qsort(v,n,sizeof(float),cmp_float);                 /* nth_element: same order statistic */
    float nthElement=sqrtf(v[nth]);
    nf->frameEnergyTH = nthElement*SETTING_FRAME_ENERGY_TH_FAC_MEDIAN;
    nf->frameEnergyTH = 26.0f*SETTING_FRAME_ENERGY_TH_CONST_WEIGHT + nf->frameEnergyTH*(1-SETTING_FRAME_ENERGY_TH_CONST_WEIGHT);
    nf->frameEnergyTH = nf->frameEnergyTH*nf->frameEnergyTH;
    nf->frameEnergyTH *= SETTING_OVERALL_ENERGY_TH_WEIGHT*SETTING_OVERALL_ENERGY_TH_WEIGHT;
    free(v);
}
/* FullSystem::linearizeAll(fixLinearization), FullSystemOptimize.cpp:52-87, 144-211.
 * activeResiduals = existing, non-linearized residuals (FullSystemOptimize.cpp:412-429). */
static double linearize_points(OrcBA* ba, int fix, int lo, int hi) {
    double E=0; int W=ba->W;
    for (int p=lo;p<hi;p++) { if (ba->pts[p].removed) continue;
        for (int t=0;t<W;t++) { OrcRes* r=RES(ba,p,t); if (!r->exists || r->isLinearized) continue;
            E += linearize(ba,p,t);
            if (fix) {
                apply_res(r);
                if (r->isActive) {
                    if (r->isNew) {                     /* :66-79 */
                        OrcPoint* pt=&ba->pts[p]; const OrcPrecalc* pc=&ba->pre[pt->host*W+t];
                        float i0=pc->PRE_KRKiTll[0]*pt->u+pc->PRE_KRKiTll[1]*pt->v+pc->PRE_KRKiTll[2], i1=pc->PRE_KRKiTll[3]*pt->u+pc->PRE_KRKiTll[4]*pt->v+pc->PRE_KRKiTll[5], i2=pc->PRE_KRKiTll[6]*pt->u+pc->PRE_KRKiTll[7]*pt->v+pc->PRE_KRKiTll[8];
                        float q0=i0+pc->PRE_KtTll[0]*pt->idepth_scaled, q1=i1+pc->PRE_KtTll[1]*pt->idepth_scaled, q2=i2+pc->PRE_KtTll[2]*pt->idepth_scaled;
                        float ex=i0/i2-q0/q2, ey=i1/i2-q1/q2;
                        float relBS=0.01*sqrtf(ex*ex+ey*ey);
                        if (relBS>pt->maxRelBaseline) pt->maxRelBaseline=relBS;
                        pt->numGoodResiduals++;
                    }
                } else r->exists=2;                     /* toRemove -> dropResidual after the TH update (:184-205) */
            }
        } }
    return E;
}
#ifdef ORC_FAST
/* Persistent worker pool of the timed baseline (the reference keeps its IndexThreadReduce workers alive and hands them chunk ranges, util/IndexThreadReduce.h:76-137;
 * rounds 1-2 created and joined the threads per call, which made the all-cores line SLOWER than the 6-thread one). One pool per process, grown on demand;
 * workers sleep on a condition variable between jobs. */
typedef struct { pthread_t th[ORC_MAXTHREADS]; int n; pthread_mutex_t mu; pthread_cond_t cv_go, cv_done; unsigned gen; int pending; void* (*fn)(void*); void* arg[ORC_MAXTHREADS]; int nrun; } OrcPool;
static OrcPool g_pool = { .n = 0, .mu = PTHREAD_MUTEX_INITIALIZER, .cv_go = PTHREAD_COND_INITIALIZER, .cv_done = PTHREAD_COND_INITIALIZER };
static void* pool_worker(void* a) {
    const int id = (int)(size_t)a; unsigned seen = 0;
    pthread_mutex_lock(&g_pool.mu);
    for (;;) {
        while (g_pool.gen == seen) pthread_cond_wait(&g_pool.cv_go, &g_pool.mu);
        seen = g_pool.gen;
        if (id < g_pool.nrun) {
            void* (*fn)(void*) = g_pool.fn; void* arg = g_pool.arg[id];
            pthread_mutex_unlock(&g_pool.mu);
            fn(arg);
            pthread_mutex_lock(&g_pool.mu);
            if (--g_pool.pending == 0) pthread_cond_signal(&g_pool.cv_done);
        }
    }
    return 0;
}
/* runs fn(args[t]) for t = 0 .. nt-1: t = 0 on the caller, the rest on the pool */
static void pool_run(int nt, void* (*fn)(void*), void** args) {
    if (nt > ORC_MAXTHREADS) nt = ORC_MAXTHREADS;
    pthread_mutex_lock(&g_pool.mu);
    while (g_pool.n < nt - 1) { pthread_create(&g_pool.th[g_pool.n], 0, pool_worker, (void*)(size_t)g_pool.n); g_pool.n++; }
    g_pool.fn = fn; g_pool.nrun = nt - 1; g_pool.pending = nt - 1;
    for (int t = 1; t < nt; t++) g_pool.arg[t - 1] = args[t];
    g_pool.gen++;
    pthread_cond_broadcast(&g_pool.cv_go);
    pthread_mutex_unlock(&g_pool.mu);
    fn(args[0]);
    pthread_mutex_lock(&g_pool.mu);
    while (g_pool.pending > 0) pthread_cond_wait(&g_pool.cv_done, &g_pool.mu);
    pthread_mutex_unlock(&g_pool.mu);
}
#endif
typedef struct { OrcBA* ba; int fix, tid, nt; double E; } LinJob;
static void* lin_job_run(void* arg) {
    LinJob* j=(LinJob*)arg; OrcBA* ba=j->ba; j->E=0;
    for (int c=j->tid; c*50<ba->P; c+=j->nt) { int lo=c*50, hi=lo+50; if (hi>ba->P) hi=ba->P; j->E+=linearize_points(ba,j->fix,lo,hi); }
    return 0;
}
double orc_ba_linearize_all(OrcBA* ba, int fix) {
    double t0=now_s(), E=0; int W=ba->W;
#ifdef ORC_FAST
    if (ba->linearize_mt && ba->nthreads_used>1) {     /* upstream DSO's linearizeAll_Reductor over chunks of points; residuals are independent */
        LinJob jobs[ORC_MAXTHREADS]; void* args[ORC_MAXTHREADS]; int nt=ba->nthreads_used;
        for (int t=0;t<nt;t++) { jobs[t].ba=ba; jobs[t].fix=fix; jobs[t].tid=t; jobs[t].nt=nt; args[t]=&jobs[t]; }
        pool_run(nt, lin_job_run, args);
        for (int t=0;t<nt;t++) E+=jobs[t].E;
    } else
#endif
    E=linearize_points(ba,fix,0,ba->P);
    set_new_frame_energy_th(ba);
    if (fix) for (size_t i=0;i<(size_t)ba->P*W;i++) if (ba->res[i].exists==2) { ba->res[i].exists=0; ba->res[i].isActive=0; }
    ba->t_linearize += now_s()-t0;
    return E;
}
void orc_ba_apply_res(OrcBA* ba) {
    for (int p=0;p<ba->P;p++) { if (ba->pts[p].removed) continue;
        for (int t=0;t<ba->W;t++) { OrcRes* r=RES(ba,p,t); if (r->exists && !r->isLinearized) apply_res(r); } }
}

/* ---------------------------------------------------------------- a7: AccumulatedTopHessianSSE::addPoint<mode> */
static inline void tier_add(OrcTier* T, int i, real v) {
    T->A[i] += (float)v;
#ifndef ORC_FAST
    T->D[i] += (double)v;
#endif
}
static void top_add_point(OrcBA* ba, OrcTier* acc /*[W*W]*/, int p, int mode, int* nres) {
    OrcPoint* pt=&ba->pts[p]; int W=ba->W;
    real dd=pt->deltaF, bd_acc=0, Hdd_acc=0, Hcd_acc[4]={0,0,0,0};
    for (int t=0;t<W;t++) {
        OrcRes* r=RES(ba,p,t); if (!r->exists) continue;
        if (mode==0) { if (r->isLinearized || !r->isActive) continue; }
        if (mode==1) { if (!r->isLinearized || !r->isActive) continue; }
        if (mode==2) { if (!r->isActive) continue; }
        OrcJ* rJ=&r->J; int htIDX=pt->host+t*W; const float* dp=ba->adHTdeltaF+htIDX*8;
        real resApprox[8];
        if (mode==0) for (int i=0;i<8;i++) resApprox[i]=rJ->resF[i];
        if (mode==2) for (int i=0;i<8;i++) resApprox[i]=r->res_toZeroF[i];
        if (mode==1) {                                   /* :81-99 */
            real jx=0, jy=0, cx_=0, cy_=0;
            for (int i=0;i<6;i++) { jx+=rJ->Jpdxi[0][i]*(real)dp[i]; jy+=rJ->Jpdxi[1][i]*(real)dp[i]; }
            for (int i=0;i<4;i++) { cx_+=rJ->Jpdc[0][i]*(real)ba->cDeltaF[i]; cy_+=rJ->Jpdc[1][i]*(real)ba->cDeltaF[i]; }
            jx=jx+cx_+rJ->Jpdd[0]*dd; jy=jy+cy_+rJ->Jpdd[1]*dd;
            for (int i=0;i<8;i++) { real rtz=r->res_toZeroF[i]; rtz+=rJ->JIdx[0][i]*jx; rtz+=rJ->JIdx[1][i]*jy; rtz+=rJ->JabF[0][i]*(real)dp[6]; rtz+=rJ->JabF[1][i]*(real)dp[7]; resApprox[i]=rtz; }
        }
        real JI_r[2]={0,0}, Jab_r[2]={0,0}, rr=0;
        for (int i=0;i<8;i++) { JI_r[0]+=resApprox[i]*rJ->JIdx[0][i]; JI_r[1]+=resApprox[i]*rJ->JIdx[1][i];
            Jab_r[0]+=resApprox[i]*rJ->JabF[0][i]; Jab_r[1]+=resApprox[i]*rJ->JabF[1][i]; rr+=resApprox[i]*resApprox[i]; }
        OrcTier* A=&acc[htIDX];
        /* AccumulatorApprox::update (MatrixAccumulators.h:754-847): x=(Jpdc[0],Jpdxi[0]) y=(Jpdc[1],Jpdxi[1]) a,b,c = JIdx2 */
        real x[10], y[10]; for (int i=0;i<4;i++) { x[i]=rJ->Jpdc[0][i]; y[i]=rJ->Jpdc[1][i]; } for (int i=0;i<6;i++) { x[4+i]=rJ->Jpdxi[0][i]; y[4+i]=rJ->Jpdxi[1][i]; }
        real a=rJ->JIdx2[0], b=rJ->JIdx2[1], c=rJ->JIdx2[3];
        int idx=0;
        for (int rr_=0;rr_<10;rr_++) for (int cc=rr_;cc<10;cc++) { tier_add(A,idx, a*x[cc]*x[rr_] + c*y[cc]*y[rr_] + b*(x[cc]*y[rr_]+y[cc]*x[rr_])); idx++; }
        A->num++; A->numIn1++; orc_tier_shift(A,0);
        /* updateBotRight (:901-915) */
        tier_add(A,85,rJ->Jab2[0]); tier_add(A,86,rJ->Jab2[1]); tier_add(A,87,Jab_r[0]); tier_add(A,88,rJ->Jab2[3]); tier_add(A,89,Jab_r[1]); tier_add(A,90,rr);
        /* updateTopRight (:850-899): TR00=JabJIdx(0,0) TR10=JabJIdx(0,1) TR01=JabJIdx(1,0) TR11=JabJIdx(1,1) TR02=JI_r[0] TR12=JI_r[1] */
        for (int i=0;i<10;i++) {
            tier_add(A,55+3*i+0, x[i]*rJ->JabJIdx[0]+y[i]*rJ->JabJIdx[1]);
            tier_add(A,55+3*i+1, x[i]*rJ->JabJIdx[2]+y[i]*rJ->JabJIdx[3]);
            tier_add(A,55+3*i+2, x[i]*JI_r[0]+y[i]*JI_r[1]);
        }
        real Ji2_Jpdd[2]={rJ->JIdx2[0]*rJ->Jpdd[0]+rJ->JIdx2[1]*rJ->Jpdd[1], rJ->JIdx2[2]*rJ->Jpdd[0]+rJ->JIdx2[3]*rJ->Jpdd[1]};
        bd_acc += JI_r[0]*rJ->Jpdd[0]+JI_r[1]*rJ->Jpdd[1];
        Hdd_acc += Ji2_Jpdd[0]*rJ->Jpdd[0]+Ji2_Jpdd[1]*rJ->Jpdd[1];
        for (int i=0;i<4;i++) Hcd_acc[i] += rJ->Jpdc[0][i]*Ji2_Jpdd[0]+rJ->Jpdc[1][i]*Ji2_Jpdd[1];
        (*nres)++;
    }
    if (mode==0) { pt->Hdd_accAF=Hdd_acc; pt->bd_accAF=bd_acc; for (int i=0;i<4;i++) pt->Hcd_accAF[i]=Hcd_acc[i]; }
    if (mode==1 || mode==2) { pt->Hdd_accLF=Hdd_acc; pt->bd_accLF=bd_acc; for (int i=0;i<4;i++) pt->Hcd_accLF[i]=Hcd_acc[i]; }
    if (mode==2) { pt->Hdd_accAF=0; pt->bd_accAF=0; for (int i=0;i<4;i++) pt->Hcd_accAF[i]=0; }
}
/* AccumulatorApprox::finish (MatrixAccumulators.h:619-651) -> 13x13 */
static void top_finish13(OrcTier* A, double H13[169]) {
    orc_tier_shift(A,1);
    int idx=0; memset(H13,0,sizeof(double)*169);
    for (int r=0;r<10;r++) for (int c=r;c<10;c++) { H13[r*13+c]=H13[c*13+r]=orc_tier_get(A,idx); idx++; }
    idx=0; for (int r=0;r<10;r++) for (int c=0;c<3;c++) { H13[r*13+c+10]=H13[(c+10)*13+r]=orc_tier_get(A,55+idx); idx++; }
    H13[10*13+10]=orc_tier_get(A,85); H13[10*13+11]=H13[11*13+10]=orc_tier_get(A,86); H13[10*13+12]=H13[12*13+10]=orc_tier_get(A,87);
    H13[11*13+11]=orc_tier_get(A,88); H13[11*13+12]=H13[12*13+11]=orc_tier_get(A,89); H13[12*13+12]=orc_tier_get(A,90);
}
/* C(8x8) += A(8x8) * M(8x8 view, ld) * B^T(8x8) */
static void amb8(const double* A, const double* M, int ldm, const double* B, double* C, int ldc) {
    double T[64];
    for (int i=0;i<8;i++) for (int j=0;j<8;j++) { double s=0; for (int k=0;k<8;k++) s+=A[i*8+k]*M[k*ldm+j]; T[i*8+j]=s; }
    for (int i=0;i<8;i++) for (int j=0;j<8;j++) { double s=0; for (int k=0;k<8;k++) s+=T[i*8+k]*B[j*8+k]; C[i*ldc+j]+=s; }
}
/* stitchDoubleInternal + stitchDoubleMT tail (AccumulatedTopHessian.cpp:241-303, .h:91-139); sums the thread replicas in fp64 */
static void top_stitch(OrcBA* ba, OrcTier (*acc)[ORC_MAXW*ORC_MAXW], int usePrior, double* H, double* b, double* H13_all /* optional [W*W][169] */) {
    int W=ba->W, n=NDIM(ba);
    memset(H,0,sizeof(double)*n*n); memset(b,0,sizeof(double)*n);
    for (int k=0;k<W*W;k++) {
        int h=k%W, t=k/W, hIdx=ORC_CPARS+h*8, tIdx=ORC_CPARS+t*8;
        double accH[169]; memset(accH,0,sizeof(accH));
        for (int tid=0;tid<ba->nt_alloc;tid++) { double H13[169]; OrcTier* A=&acc[tid][k]; top_finish13(A,H13); if (A->num==0) continue; for (int i=0;i<169;i++) accH[i]+=H13[i]; }
        if (H13_all) memcpy(H13_all+k*169,accH,sizeof(accH));
        const double *AH=ba->adHost+k*64, *AT=ba->adTarget+k*64, *M=accH+4*13+4;
        amb8(AH,M,13,AH,H+hIdx*n+hIdx,n); amb8(AT,M,13,AT,H+tIdx*n+tIdx,n); amb8(AH,M,13,AT,H+hIdx*n+tIdx,n);
        for (int i=0;i<8;i++) for (int c=0;c<4;c++) { double s1=0,s2=0; for (int kk=0;kk<8;kk++) { s1+=AH[i*8+kk]*accH[(4+kk)*13+c]; s2+=AT[i*8+kk]*accH[(4+kk)*13+c]; }
            H[(hIdx+i)*n+c]+=s1; H[(tIdx+i)*n+c]+=s2; }
        for (int i=0;i<4;i++) for (int c=0;c<4;c++) H[i*n+c]+=accH[i*13+c];
        for (int i=0;i<8;i++) { double s1=0,s2=0; for (int kk=0;kk<8;kk++) { s1+=AH[i*8+kk]*accH[(4+kk)*13+12]; s2+=AT[i*8+kk]*accH[(4+kk)*13+12]; } b[hIdx+i]+=s1; b[tIdx+i]+=s2; }
        for (int i=0;i<4;i++) b[i]+=accH[i*13+12];
    }
    if (usePrior) {                                        /* :292-302 */
        for (int i=0;i<4;i++) { H[i*n+i]+=ba->cPrior[i]; b[i]+=ba->cPrior[i]*(double)ba->cDeltaF[i]; }
        for (int h=0;h<W;h++) for (int i=0;i<8;i++) { int d=ORC_CPARS+h*8+i; H[d*n+d]+=ba->frames[h].prior[i]; b[d]+=ba->frames[h].prior[i]*ba->frames[h].delta_prior[i]; }
    }
    for (int h=0;h<W;h++) {                                /* .h:127-138 */
        int hIdx=ORC_CPARS+h*8;
        for (int i=0;i<8;i++) for (int c=0;c<4;c++) H[c*n+hIdx+i]=H[(hIdx+i)*n+c];
        for (int t=h+1;t<W;t++) { int tIdx=ORC_CPARS+t*8;
            for (int i=0;i<8;i++) for (int j=0;j<8;j++) H[(hIdx+i)*n+tIdx+j]+=H[(tIdx+j)*n+hIdx+i];
            for (int i=0;i<8;i++) for (int j=0;j<8;j++) H[(tIdx+j)*n+hIdx+i]=H[(hIdx+i)*n+tIdx+j]; }
    }
}

/* ---------------------------------------------------------------- a9: AccumulatedSCHessianSSE::addPoint, AccumulatedSCHessian.cpp:34-77 */
static inline void tierxx_update(OrcTier* T, int ni, int nj, const real* L, const real* Rv, real w) {   /* AccumulatorXX::update :62-67: A += w*L*R^T */
    for (int i=0;i<ni;i++) { real wl=w*L[i]; for (int j=0;j<nj;j++) tier_add(T,i*nj+j, wl*Rv[j]); }
    T->numIn1++; orc_tier_shift(T,0);
}
static inline void tierx_update(OrcTier* T, int ni, const real* L, real w) {   /* AccumulatorX::update :203-208 */
    for (int i=0;i<ni;i++) tier_add(T,i,w*L[i]);
    T->numIn1++; orc_tier_shift(T,0);
}
static void sc_add_point(OrcBA* ba, int tid, int p, int shiftPriorToZero) {
    OrcPoint* pt=&ba->pts[p]; int W=ba->W, ngood=0;
    for (int t=0;t<W;t++) { OrcRes* r=RES(ba,p,t); if (r->exists && r->isActive) ngood++; }
    if (ngood==0) { pt->HdiF=0; pt->bdSumF=0; pt->idepth_hessian=0; pt->maxRelBaseline=0; return; }
    float H=pt->Hdd_accAF+pt->Hdd_accLF+pt->priorF; if (H<1e-10) H=1e-10;
    pt->idepth_hessian=H; pt->HdiF=1.0/H; pt->bdSumF=pt->bd_accAF+pt->bd_accLF;
    if (shiftPriorToZero) pt->bdSumF += pt->priorF*pt->deltaF;
    real Hcd[4]; for (int i=0;i<4;i++) Hcd[i]=pt->Hcd_accAF[i]+pt->Hcd_accLF[i];
    tierxx_update(&ba->accHcc[tid],4,4,Hcd,Hcd,pt->HdiF);
    tierx_update(&ba->accbc[tid],4,Hcd,(real)(pt->bdSumF*pt->HdiF));
    int nF2=W*W;
    for (int t1=0;t1<W;t1++) { OrcRes* r1=RES(ba,p,t1); if (!r1->exists || !r1->isActive) continue;
        int r1ht=pt->host+t1*W;
        for (int t2=0;t2<W;t2++) { OrcRes* r2=RES(ba,p,t2); if (!r2->exists || !r2->isActive) continue;
            tierxx_update(&ba->accD[tid][r1ht+t2*nF2],8,8,r1->JpJdF,r2->JpJdF,pt->HdiF); }
        tierxx_update(&ba->accE[tid][r1ht],8,4,r1->JpJdF,Hcd,pt->HdiF);
        tierx_update(&ba->accEB[tid][r1ht],8,r1->JpJdF,(real)(pt->HdiF*pt->bdSumF));
    }
}
static void tier_sum_threads(int nt, OrcTier** arr, int idx, int n, double* out) {
    for (int i=0;i<n;i++) out[i]=0;
    for (int tid=0;tid<nt;tid++) { OrcTier* T=&arr[tid][idx]; orc_tier_shift(T,1); if (T->numIn1m==0) continue; for (int i=0;i<n;i++) out[i]+=orc_tier_get(T,i); }
}
/* AccumulatedSCHessianSSE::stitchDoubleInternal + MT tail, AccumulatedSCHessian.cpp:78-157, .h:93-133 */
static void sc_stitch(OrcBA* ba, double* H, double* b) {
    int W=ba->W, n=NDIM(ba), nf2=W*W;
    memset(H,0,sizeof(double)*n*n); memset(b,0,sizeof(double)*n);
    for (int k0=0;k0<W*W;k0++) {
        int i=k0%W, j=k0/W, iIdx=ORC_CPARS+i*8, jIdx=ORC_CPARS+j*8, ijIdx=i+W*j;
        double Hpc[32], bp[8]; tier_sum_threads(ba->nt_alloc,ba->accE,ijIdx,32,Hpc); tier_sum_threads(ba->nt_alloc,ba->accEB,ijIdx,8,bp);
        const double *AHij=ba->adHost+ijIdx*64, *ATij=ba->adTarget+ijIdx*64;
        for (int r=0;r<8;r++) { for (int c=0;c<4;c++) { double s1=0,s2=0; for (int kk=0;kk<8;kk++) { s1+=AHij[r*8+kk]*Hpc[kk*4+c]; s2+=ATij[r*8+kk]*Hpc[kk*4+c]; } H[(iIdx+r)*n+c]+=s1; H[(jIdx+r)*n+c]+=s2; }
            double s1=0,s2=0; for (int kk=0;kk<8;kk++) { s1+=AHij[r*8+kk]*bp[kk]; s2+=ATij[r*8+kk]*bp[kk]; } b[iIdx+r]+=s1; b[jIdx+r]+=s2; }
        for (int k=0;k<W;k++) {
            int kIdx=ORC_CPARS+k*8, ijk=ijIdx+k*nf2, ik=i+W*k; double D[64]; tier_sum_threads(ba->nt_alloc,ba->accD,ijk,64,D);
            const double *AHik=ba->adHost+ik*64, *ATik=ba->adTarget+ik*64;
            amb8(AHij,D,8,AHik,H+iIdx*n+iIdx,n); amb8(ATij,D,8,ATik,H+jIdx*n+kIdx,n);
            amb8(ATij,D,8,AHik,H+jIdx*n+iIdx,n); amb8(AHij,D,8,ATik,H+iIdx*n+kIdx,n);
        }
    }
    double Hcc[16], bc[4];
    { OrcTier* a[ORC_MAXTHREADS]; for (int t=0;t<ba->nt_alloc;t++) a[t]=&ba->accHcc[t]; tier_sum_threads(ba->nt_alloc,a,0,16,Hcc);
      for (int t=0;t<ba->nt_alloc;t++) a[t]=&ba->accbc[t]; tier_sum_threads(ba->nt_alloc,a,0,4,bc); }
    for (int r=0;r<4;r++) { for (int c=0;c<4;c++) H[r*n+c]+=Hcc[r*4+c]; b[r]+=bc[r]; }
    for (int h=0;h<W;h++) { int hIdx=ORC_CPARS+h*8; for (int r=0;r<8;r++) for (int c=0;c<4;c++) H[c*n+hIdx+r]=H[(hIdx+r)*n+c]; }
}

/* ---------------------------------------------------------------- threaded point loops (IndexThreadReduce, chunks of 50: EnergyFunctional.cpp:202-203) */
typedef struct { OrcBA* ba; int tid, kind, mode, shift; const float* xc; const float* xAd; } Job;
static void resub_point(OrcBA* ba, int p, const float* xc, const float* xAd);
static void* job_run(void* arg) {
    Job* j=(Job*)arg; OrcBA* ba=j->ba; int nt=ba->nthreads_used;
    for (int c=j->tid; c*50<ba->P; c+=nt) {
        int lo=c*50, hi=lo+50; if (hi>ba->P) hi=ba->P;
        for (int p=lo;p<hi;p++) { if (ba->pts[p].removed) continue;
            if (j->kind==0) top_add_point(ba, j->mode==1 ? ba->accTopL[j->tid] : ba->accTopA[j->tid], p, j->mode, &ba->nres[j->tid]);
            else if (j->kind==1) sc_add_point(ba,j->tid,p,j->shift);
            else resub_point(ba,p,j->xc,j->xAd); }
    }
    return 0;
}
static void run_jobs(OrcBA* ba, int kind, int mode, int shift, const float* xc, const float* xAd) {
    Job jobs[ORC_MAXTHREADS]; int nt=ba->nthreads_used;
    for (int t=0;t<nt;t++) { jobs[t].ba=ba; jobs[t].tid=t; jobs[t].kind=kind; jobs[t].mode=mode; jobs[t].shift=shift; jobs[t].xc=xc; jobs[t].xAd=xAd; }
#ifdef ORC_FAST
    void* args[ORC_MAXTHREADS];
    for (int t=0;t<nt;t++) args[t]=&jobs[t];
    pool_run(nt, job_run, args);
#else
    for (int t=0;t<nt;t++) job_run(&jobs[t]);     /* serial emulation of the static chunk split: deterministic */
#endif
}
/* accumulateAF_MT / accumulateLF_MT, EnergyFunctional.cpp:197-238 */
void orc_ba_accumulate(OrcBA* ba, int mode, double* H, double* b, double* H13_all) {
    double t0=now_s(); int W=ba->W;
    OrcTier (*acc)[ORC_MAXW*ORC_MAXW] = mode==1 ? ba->accTopL : ba->accTopA;
    for (int t=0;t<ba->nt_alloc;t++) { for (int i=0;i<W*W;i++) orc_tier_zero(&acc[t][i]); ba->nres[t]=0; }
    run_jobs(ba,0,mode,0,0,0);
    top_stitch(ba,acc,mode==1,H,b,H13_all);
    int nr=0; for (int t=0;t<ba->nt_alloc;t++) nr+=ba->nres[t];
    if (mode==0) ba->resInA=nr; else if (mode==1) ba->resInL=nr;
    ba->t_accumulate += now_s()-t0;
}
static void sc_zero(OrcBA* ba) {
    int W=ba->W;
    for (int t=0;t<ba->nt_alloc;t++) { for (int i=0;i<W*W*W;i++) orc_tier_zero(&ba->accD[t][i]);
        for (int i=0;i<W*W;i++) { orc_tier_zero(&ba->accE[t][i]); orc_tier_zero(&ba->accEB[t][i]); } orc_tier_zero(&ba->accHcc[t]); orc_tier_zero(&ba->accbc[t]); }
}
/* accumulateSCF_MT, EnergyFunctional.cpp:244-261 */
void orc_ba_accumulate_sc(OrcBA* ba, int shiftPriorToZero, double* H, double* b) {
    double t0=now_s();
    sc_zero(ba); run_jobs(ba,1,0,shiftPriorToZero,0,0); sc_stitch(ba,H,b);
    ba->t_accumulate += now_s()-t0;
}

/* ---------------------------------------------------------------- orthogonalize, EnergyFunctional.cpp:719-773 + getNullspaces FullSystemOptimize.cpp:658-712 */
static void build_projector(OrcBA* ba, double* Pm /* n x n */) {
    int W=ba->W, n=NDIM(ba), m=7;
    double* N=(double*)calloc(n*m,8);
    for (int i=0;i<6;i++) for (int f=0;f<W;f++) for (int r=0;r<6;r++)
        N[(ORC_CPARS+f*8+r)*m+i] = ba->frames[f].ns_pose[i][r] * (r<3 ? (1.0f/SCALE_XI_TRANS) : (1.0f/SCALE_XI_ROT));
    for (int f=0;f<W;f++) for (int r=0;r<6;r++) N[(ORC_CPARS+f*8+r)*m+6] = ba->frames[f].ns_scale[r] * (r<3 ? (1.0f/SCALE_XI_TRANS) : (1.0f/SCALE_XI_ROT));
    for (int c=0;c<m;c++) { double s=0; for (int r=0;r<n;r++) s+=N[r*m+c]*N[r*m+c]; s=sqrt(s); if (s>0) for (int r=0;r<n;r++) N[r*m+c]/=s; }
    double G[49], V[49], wv[7];
    for (int a=0;a<m;a++) for (int c=0;c<m;c++) { double s=0; for (int r=0;r<n;r++) s+=N[r*m+a]*N[r*m+c]; G[a*m+c]=s; }
    orc_sym_eig_jacobi(m,G,V,wv);
    double maxSv=0; for (int i=0;i<m;i++) { double sv=wv[i]>0?sqrt(wv[i]):0; if (sv>maxSv) maxSv=sv; }
    memset(Pm,0,sizeof(double)*n*n);
    double* u=(double*)malloc(8*n);
    for (int i=0;i<m;i++) { double sv=wv[i]>0?sqrt(wv[i]):0; if (!(sv>SETTING_SOLVER_MODE_DELTA*maxSv)) continue;
        for (int r=0;r<n;r++) { double s=0; for (int c=0;c<m;c++) s+=N[r*m+c]*V[c*m+i]; u[r]=s/sv; }
        for (int r=0;r<n;r++) for (int c=0;c<n;c++) Pm[r*n+c]+=u[r]*u[c]; }
    free(u); free(N);
}

/* ---------------------------------------------------------------- a12: resubstituteFPt, EnergyFunctional.cpp:291-317 */
static void resub_point(OrcBA* ba, int p, const float* xc, const float* xAd) {
    OrcPoint* pt=&ba->pts[p]; int W=ba->W, ngood=0;
    for (int t=0;t<W;t++) { OrcRes* r=RES(ba,p,t); if (r->exists && r->isActive) ngood++; }
    if (ngood==0) { pt->step=0; return; }
    float b=pt->bdSumF;
    { float s=0; for (int i=0;i<4;i++) s+=xc[i]*(pt->Hcd_accAF[i]+pt->Hcd_accLF[i]); b-=s; }
    for (int t=0;t<W;t++) { OrcRes* r=RES(ba,p,t); if (!r->exists || !r->isActive) continue;
        const float* xa=xAd+(pt->host*W+t)*8; float s=0; for (int i=0;i<8;i++) s+=xa[i]*(float)r->JpJdF[i]; b-=s; }
    pt->step=-b*pt->HdiF;
}
/* solveSystemF, EnergyFunctional.cpp:776-914 (solverMode = FIX_LAMBDA | ORTHOGONALIZE_X_LATER) */
void orc_ba_solve_system(OrcBA* ba, int iteration, double lambda, double* x_out, double* dbg_HA, double* dbg_bA, double* dbg_Hsc, double* dbg_bsc) {
    int W=ba->W, n=NDIM(ba); lambda=1e-5;
    double *HA=(double*)malloc(8*n*n), *HL=(double*)malloc(8*n*n), *Hsc=(double*)malloc(8*n*n), *HF=(double*)malloc(8*n*n);
    double *bA=(double*)malloc(8*n), *bL=(double*)malloc(8*n), *bsc=(double*)malloc(8*n), *bF=(double*)malloc(8*n), *x=(double*)malloc(8*n);
    orc_ba_accumulate(ba,0,HA,bA,0);
    orc_ba_accumulate(ba,1,HL,bL,0);
    orc_ba_accumulate_sc(ba,1,Hsc,bsc);
    double t0=now_s();
    if (dbg_HA) { memcpy(dbg_HA,HA,8*n*n); memcpy(dbg_bA,bA,8*n); memcpy(dbg_Hsc,Hsc,8*n*n); memcpy(dbg_bsc,bsc,8*n); }
    double* delta=(double*)malloc(8*n);
    for (int i=0;i<4;i++) delta[i]=(double)ba->cDeltaF[i];
    for (int h=0;h<W;h++) for (int i=0;i<8;i++) delta[ORC_CPARS+8*h+i]=ba->frames[h].delta[i];
    for (int i=0;i<n;i++) { double s=0; for (int j=0;j<n;j++) s+=ba->HM[i*n+j]*delta[j]; bF[i]=bL[i]+(ba->bM[i]+s)+bA[i]-bsc[i]; }
    for (int i=0;i<n*n;i++) HF[i]=HL[i]+ba->HM[i]+HA[i];
    for (int i=0;i<n;i++) HF[i*n+i]*=(1+lambda);
    { double f=1.0/(1+lambda); for (int i=0;i<n*n;i++) HF[i]-=Hsc[i]*f; }
    double* sv=(double*)malloc(8*n); double* Hs=(double*)malloc(8*n*n); double* bs=(double*)malloc(8*n);
    for (int i=0;i<n;i++) sv[i]=1.0/sqrt(HF[i*n+i]+10);
    for (int i=0;i<n;i++) { for (int j=0;j<n;j++) Hs[i*n+j]=sv[i]*HF[i*n+j]*sv[j]; bs[i]=sv[i]*bF[i]; }
    orc_ldlt_solve(n,Hs,bs,x);
    for (int i=0;i<n;i++) x[i]*=sv[i];
    if (iteration>=2) {                                   /* SOLVER_ORTHOGONALIZE_X_LATER, :898-902 */
        double* Pm=(double*)malloc(8*n*n); build_projector(ba,Pm);
        double* y=(double*)malloc(8*n); for (int i=0;i<n;i++) { double s=0; for (int j=0;j<n;j++) s+=Pm[i*n+j]*x[j]; y[i]=s; }
        for (int i=0;i<n;i++) x[i]-=y[i];
        free(Pm); free(y);
    }
    memcpy(ba->lastX,x,8*n); if (x_out) memcpy(x_out,x,8*n);
    /* resubstituteF_MT, :263-289 */
    float* xF=(float*)malloc(4*n); for (int i=0;i<n;i++) xF[i]=(float)x[i];
    for (int i=0;i<4;i++) ba->c_step[i]=-x[i];
    float* xAd=(float*)malloc(4*W*W*8);
    for (int h=0;h<W;h++) { for (int i=0;i<8;i++) ba->frames[h].step[i]=-x[ORC_CPARS+8*h+i]; ba->frames[h].step[8]=ba->frames[h].step[9]=0;
        for (int t=0;t<W;t++) { const float *AH=ba->adHostF+(h+W*t)*64, *AT=ba->adTargetF+(h+W*t)*64;
            for (int j=0;j<8;j++) { float s1=0,s2=0; for (int i=0;i<8;i++) { s1+=xF[ORC_CPARS+8*h+i]*AH[i*8+j]; s2+=xF[ORC_CPARS+8*t+i]*AT[i*8+j]; } xAd[(W*h+t)*8+j]=s1+s2; } } }
    ba->t_solve += now_s()-t0; t0=now_s();
    run_jobs(ba,2,0,0,xF,xAd);
    ba->t_accumulate += now_s()-t0;
    free(HA);free(HL);free(Hsc);free(HF);free(bA);free(bL);free(bsc);free(bF);free(x);free(delta);free(sv);free(Hs);free(bs);free(xF);free(xAd);
}

/* ---------------------------------------------------------------- doStepFromBackup / backupState, FullSystemOptimize.cpp:217-349 */
static void backup_state(OrcBA* ba) {
    memcpy(ba->c_value_backup,ba->c_value,sizeof(double)*4);
    for (int f=0;f<ba->W;f++) memcpy(ba->frames[f].state_backup,ba->frames[f].state,sizeof(double)*10);
    for (int p=0;p<ba->P;p++) ba->pts[p].idepth_backup=ba->pts[p].idepth;
}
int orc_ba_do_step(OrcBA* ba, float stepfacC, float stepfacT, float stepfacR, float stepfacA, float stepfacD) {
    double t0=now_s();
    double pf[10]={stepfacT,stepfacT,stepfacT,stepfacR,stepfacR,stepfacR,stepfacA,stepfacA,stepfacA,stepfacA};
    float sumA=0,sumB=0,sumT=0,sumR=0,sumID=0,numID=0,sumNID=0;
    double v[4]; for (int i=0;i<4;i++) v[i]=ba->c_value_backup[i]+stepfacC*ba->c_step[i];
    calib_set_value(ba,v);
    for (int f=0;f<ba->W;f++) { OrcFrame* fh=&ba->frames[f]; double st[10];
        for (int i=0;i<10;i++) st[i]=fh->state_backup[i]+pf[i]*fh->step[i];
        frame_set_state(fh,st);
        sumA+=fh->step[6]*fh->step[6]; sumB+=fh->step[7]*fh->step[7];
        sumT+=fh->step[0]*fh->step[0]+fh->step[1]*fh->step[1]+fh->step[2]*fh->step[2];
        sumR+=fh->step[3]*fh->step[3]+fh->step[4]*fh->step[4]+fh->step[5]*fh->step[5]; }
    for (int p=0;p<ba->P;p++) { OrcPoint* ph=&ba->pts[p]; if (ph->removed) continue;
        ph->idepth=ph->idepth_backup+stepfacD*ph->step; ph->idepth_scaled=SCALE_IDEPTH*ph->idepth;
        sumID+=ph->step*ph->step; sumNID+=fabsf(ph->idepth_backup); numID++;
        ph->idepth_zero=ph->idepth; ph->idepth_zero_scaled=SCALE_IDEPTH*ph->idepth; }
    sumA/=ba->W; sumB/=ba->W; sumR/=ba->W; sumT/=ba->W; sumID/=numID; sumNID/=numID;
    orc_ba_set_precalc(ba);
    ba->t_other += now_s()-t0;
    const float th=SETTING_TH_OPT_ITERATIONS;
    return sqrtf(sumA)<0.0005*th && sqrtf(sumB)<0.00005*th && sqrtf(sumR)<0.00005*th && sqrtf(sumT)*sumNID<0.00005*th;
}

/* EnergyFunctional::calcLEnergyF_MT + calcLEnergyPt (EnergyFunctional.cpp:332-415): frame priors, calibration prior, and per point the linearised
 * residuals' (2 res_toZeroF + J delta) . J delta plus deltaF^2 priorF. FullSystem::calcLEnergy returns 0 while setting_forceAceptStep (FullSystemOptimize.cpp:351). */
double orc_ba_calc_l_energy(OrcBA* ba) {
    int W=ba->W; double E=0;
    for (int f=0;f<W;f++) for (int i=0;i<8;i++) E += ba->frames[f].delta_prior[i]*ba->frames[f].prior[i]*ba->frames[f].delta_prior[i];
    { float e=0; for (int i=0;i<4;i++) e += (ba->cDeltaF[i]*(float)ba->cPrior[i])*ba->cDeltaF[i]; E += e; }      /* cDeltaF.cwiseProduct(cPriorF).dot(cDeltaF): floats */
    double Ept=0;
    for (int p=0;p<ba->P;p++) { OrcPoint* pt=&ba->pts[p]; if (pt->removed) continue;
        real dd=pt->deltaF;
        for (int t=0;t<W;t++) { OrcRes* r=RES(ba,p,t); if (!r->exists || !r->isLinearized || !r->isActive) continue;
            const float* dp=ba->adHTdeltaF+(pt->host+W*t)*8; OrcJ* J=&r->J;
            real jx=0, jy=0, cx_=0, cy_=0;
            for (int i=0;i<6;i++) { jx+=J->Jpdxi[0][i]*(real)dp[i]; jy+=J->Jpdxi[1][i]*(real)dp[i]; }
            for (int i=0;i<4;i++) { cx_+=J->Jpdc[0][i]*(real)ba->cDeltaF[i]; cy_+=J->Jpdc[1][i]*(real)ba->cDeltaF[i]; }
            jx = jx + cx_ + J->Jpdd[0]*dd; jy = jy + cy_ + J->Jpdd[1]*dd;
            for (int i=0;i<8;i++) {
                real Jdelta = J->JIdx[0][i]*jx + J->JIdx[1][i]*jy + J->JabF[0][i]*(real)dp[6] + J->JabF[1][i]*(real)dp[7];
                real r0 = r->res_toZeroF[i]; r0 = r0 + r0; r0 = r0 + Jdelta;
                Ept += (double)(float)(Jdelta*r0);
            } }
        Ept += (double)(float)(pt->deltaF*pt->deltaF*pt->priorF); }
    return E + (double)(float)Ept;
}
/* EnergyFunctional::calcMEnergyF (:320-329): delta . (2 bM + HM delta) with the stitched delta of getStitchedDeltaF (:938-945) */
double orc_ba_calc_m_energy(OrcBA* ba) {
    int W=ba->W, n=NDIM(ba); double* d=(double*)malloc(8*n); double E=0;
    for (int i=0;i<4;i++) d[i]=(double)ba->cDeltaF[i];
    for (int h=0;h<W;h++) for (int i=0;i<8;i++) d[ORC_CPARS+8*h+i]=ba->frames[h].delta[i];
    for (int i=0;i<n;i++) { double s=0; for (int j=0;j<n;j++) s+=ba->HM[i*n+j]*d[j]; E += d[i]*(2*ba->bM[i]+s); }
    free(d); return E;
}
/* FullSystem::loadSateBackup (FullSystemOptimize.cpp:352-369): note setIdepthZero(idepth_backup) */
static void load_state_backup(OrcBA* ba) {
    calib_set_value(ba,ba->c_value_backup);
    for (int f=0;f<ba->W;f++) frame_set_state(&ba->frames[f],ba->frames[f].state_backup);
    for (int p=0;p<ba->P;p++) { OrcPoint* ph=&ba->pts[p]; if (ph->removed) continue;
        ph->idepth=ph->idepth_backup; ph->idepth_scaled=SCALE_IDEPTH*ph->idepth; ph->idepth_zero=ph->idepth_backup; ph->idepth_zero_scaled=SCALE_IDEPTH*ph->idepth_backup; }
    orc_ba_set_precalc(ba);
}

/* ---------------------------------------------------------------- planeOpt=1 without Ceres
 * FullSystem::planeOptimize's scale fix (PlaneOptimize.cpp:183-301) and FullSystem::SWGrayOptimize_J (:307-454). GrayTHFactor_TH::Evaluate (PlaneOptimize.h:348-457)
 * multiplies its Jacobians by d_uv = the gradient of an OUTER hitColor that stays zero (an inner declaration shadows it, :378-381): zero gradient, Ceres returns
 * the initial point. Restated: the Huber(100) cost it reports, and the state changes after the solve with unchanged parameters. */
static void relinearize_newest(OrcBA* ba, const double w2c[12]) {
    OrcFrame* nf=&ba->frames[ba->W-1];
    double nsz[10]={0,0,0,0,0,0,nf->state[6],nf->state[7],0,0};
    memcpy(nf->evalPT,w2c,sizeof(double)*12);
    frame_set_state(nf,nsz); frame_set_state_zero(nf,nsz);
    orc_ba_set_adjoints(ba); orc_ba_set_precalc(ba);
}
void orc_ba_plane_scale_fix(OrcBA* ba, double localscale, const double camToTrackingRef[12], const double trackingRef_camToWorld[12]) {
    double c2r[12], c2w[12], w2c[12]; memcpy(c2r,camToTrackingRef,sizeof(c2r));
    c2r[3]*=localscale; c2r[7]*=localscale; c2r[11]*=localscale;
    orc_se3_mul(trackingRef_camToWorld,c2r,c2w); orc_se3_inv(c2w,w2c);
    for (int p=0;p<ba->P;p++) { OrcPoint* ph=&ba->pts[p]; if (ph->removed || ph->host!=ba->W-1) continue;
        double idp=ph->idepth_scaled/localscale;
        ph->idepth=(float)idp; ph->idepth_scaled=SCALE_IDEPTH*ph->idepth; ph->idepth_zero=(float)idp; ph->idepth_zero_scaled=SCALE_IDEPTH*ph->idepth_zero; }
    relinearize_newest(ba,w2c);
}
double orc_ba_sw_gray_optimize(OrcBA* ba, int* n_blocks) {
    int W=ba->W; double T[ORC_MAXW][12]; double cost=0; int nb=0;
    for (int i=0;i<W;i++) { double xi[6], om[6]={0,0,0,0,0,0}; orc_se3_log(ba->frames[i].PRE_worldToCam,xi); om[3]=xi[3]; om[4]=xi[4]; om[5]=xi[5];
        orc_se3_exp(om,T[i]); T[i][3]=ba->frames[i].PRE_worldToCam[3]; T[i][7]=ba->frames[i].PRE_worldToCam[7]; T[i][11]=ba->frames[i].PRE_worldToCam[11]; }
    float fxl=ba->c_scaledf[0], fyl=ba->c_scaledf[1], cxl=ba->c_scaledf[2], cyl=ba->c_scaledf[3], fxli=ba->c_scaledi[0], fyli=ba->c_scaledi[1];
    for (int h=0;h<W;h++) for (int t=0;t<W;t++) { if (h==t) continue;
        double Ti[12], Tij[12]; orc_se3_inv(T[h],Ti); orc_se3_mul(T[t],Ti,Tij);
        const float* dIl=ba->frames[t].dI;
        for (int p=0;p<ba->P;p++) { OrcPoint* ph=&ba->pts[p]; if (ph->removed || ph->host!=h) continue;
            double inv_dep=ph->idepth_scaled;
            if (inv_dep<1e-4 || inv_dep>1e3) continue;
            float idf=(float)inv_dep;
            double K0=(double)((ph->u+0-cxl)*fxli), K1=(double)((ph->v+0-cyl)*fyli);
            double p0=Tij[0]*K0+Tij[1]*K1+Tij[2]*1.0+Tij[3]*(double)idf, p1=Tij[4]*K0+Tij[5]*K1+Tij[6]*1.0+Tij[7]*(double)idf, p2=Tij[8]*K0+Tij[9]*K1+Tij[10]*1.0+Tij[11]*(double)idf;
            float drescale=(float)(1.0f/p2);
            float u=(float)(p0*(double)drescale), v=(float)(p1*(double)drescale);
            float Ku=u*fxl+cxl, Kv=v*fyl+cyl;
            double r;
            if (!(Ku>1.1f && Kv>1.1f && Ku<(float)(ba->w-3) && Kv<(float)(ba->h-3)) || inv_dep<0) r=100;
            else { int ix=(int)Ku, iy=(int)Kv; float dx=Ku-ix, dy=Kv-iy, dxdy=dx*dy; const float* bp=dIl+3*(ix+iy*ba->w);
                float I=dxdy*bp[3*(1+ba->w)]+(dy-dxdy)*bp[3*ba->w]+(dx-dxdy)*bp[3]+(1-dx-dy+dxdy)*bp[0];
                r=(double)(I-ph->color[4]); }
            double s=r*r; cost+=0.5*(s<=1e4 ? s : 2.0*100.0*sqrt(s)-1e4); nb++;
        } }
    for (int p=0;p<ba->P;p++) { OrcPoint* ph=&ba->pts[p]; if (ph->removed || ph->host>=W-2) continue;
        double idp=ph->idepth_scaled; ph->idepth=(float)idp; ph->idepth_scaled=SCALE_IDEPTH*ph->idepth; ph->idepth_zero=(float)idp; ph->idepth_zero_scaled=SCALE_IDEPTH*ph->idepth_zero; }
    relinearize_newest(ba,T[W-1]);
    if (n_blocks) *n_blocks=nb;
    return cost;
}
void orc_ba_get_idepth_zero(OrcBA* ba, float* out) { for (int p=0;p<ba->P;p++) out[p]=ba->pts[p].idepth_zero; }

/* ---------------------------------------------------------------- FullSystem::optimize, FullSystemOptimize.cpp:398-602 */
double orc_ba_optimize(OrcBA* ba, int mnumOptIts) {
    if (ba->W<2) return 0; if (ba->W<3) mnumOptIts=20; if (ba->W<4) mnumOptIts=15;
    for (int p=0;p<ba->P;p++) for (int t=0;t<ba->W;t++) { OrcRes* r=RES(ba,p,t);        /* :412-429 resetOOB */
        if (r->exists && !r->isLinearized) { r->state_NewEnergy=r->state_energy=0; r->state_NewState=ORC_OUTLIER; r->state_state=ORC_IN; } }
    const int force=ba->force_accept_step;
    double lastEnergy=orc_ba_linearize_all(ba,0);
    double lastEnergyL = force ? 0 : orc_ba_calc_l_energy(ba), lastEnergyM = force ? 0 : orc_ba_calc_m_energy(ba);
    orc_ba_apply_res(ba);
    double lambda=1e-1;
    ba->n_rejected=0;
    for (int it=0; it<mnumOptIts; it++) {
        backup_state(ba);
        orc_ba_solve_system(ba,it,lambda,0,0,0,0,0);
        int canbreak=orc_ba_do_step(ba,1,1,1,1,1);
        double newEnergy=orc_ba_linearize_all(ba,0);
        double newEnergyL = force ? 0 : orc_ba_calc_l_energy(ba), newEnergyM = force ? 0 : orc_ba_calc_m_energy(ba);
        if (force || (newEnergy + newEnergyL + newEnergyM < lastEnergy + lastEnergyL + lastEnergyM)) {      /* :519-532 */
            orc_ba_apply_res(ba);
            lastEnergy=newEnergy; lastEnergyL=newEnergyL; lastEnergyM=newEnergyM; lambda*=0.25;
        } else {                                                                                          /* :534-541 */
            load_state_backup(ba);
            lastEnergy=orc_ba_linearize_all(ba,0);
            lastEnergyL=orc_ba_calc_l_energy(ba); lastEnergyM=orc_ba_calc_m_energy(ba);
            lambda*=1e2; ba->n_rejected++;
        }
        if (canbreak && it>=ba->min_opt_iterations && !ba->never_break) break;  /* setting_minOptIterations = 1 */
    }
    OrcFrame* nf=&ba->frames[ba->W-1];                     /* :550-557 */
    double nsz[10]={0,0,0,0,0,0,nf->state[6],nf->state[7],0,0};
    memcpy(nf->evalPT,nf->PRE_worldToCam,sizeof(double)*12);
    frame_set_state(nf,nsz); frame_set_state_zero(nf,nsz);
    orc_ba_set_adjoints(ba); orc_ba_set_precalc(ba);
    lastEnergy=orc_ba_linearize_all(ba,1);
    return sqrtf((float)(lastEnergy/(ORC_PATTERN_NUM*ba->resInA)));
}

/* ---------------------------------------------------------------- marginalisation of points
 * FullSystem::flagPointsForRemoval inner loop (FullSystem.cpp:975-990) + EnergyFunctional::marginalizePointsF (:615-676).
 * flags[p] != 0 -> PS_MARGINALIZE. Outputs the stitched M, Mb, Msc, Mbsc and adds 0.25*(M-Msc) into HM/bM. */
void orc_ba_marginalize_points(OrcBA* ba, const uint8_t* flags, double* M, double* Mb, double* Msc, double* Mbsc) {
    int W=ba->W, n=NDIM(ba);
    for (int p=0;p<ba->P;p++) { if (!flags[p] || ba->pts[p].removed) continue;
        for (int t=0;t<W;t++) { OrcRes* r=RES(ba,p,t); if (!r->exists) continue;
            r->state_NewEnergy=r->state_energy=0; r->state_NewState=ORC_OUTLIER; r->state_state=ORC_IN;
            linearize(ba,p,t); r->isLinearized=0; apply_res(r); if (r->isActive) fix_linearization(ba,p,t); } }
    for (int t=0;t<ba->nt_alloc;t++) { for (int i=0;i<W*W;i++) orc_tier_zero(&ba->accTopA[t][i]); ba->nres[t]=0; }
    sc_zero(ba);
    for (int p=0;p<ba->P;p++) { if (!flags[p] || ba->pts[p].removed) continue;
        ba->pts[p].priorF *= SETTING_IDEPTH_FIX_PRIOR_MARGFAC;
        top_add_point(ba,ba->accTopA[0],p,2,&ba->nres[0]); sc_add_point(ba,0,p,0);
        ba->pts[p].removed=1; for (int t=0;t<W;t++) RES(ba,p,t)->exists=0; }
    top_stitch(ba,ba->accTopA,0,M,Mb,0); sc_stitch(ba,Msc,Mbsc);
    ba->resInM += ba->nres[0];
    for (int i=0;i<n*n;i++) ba->HM[i]+=SETTING_MARG_WEIGHT_FAC*(M[i]-Msc[i]);
    for (int i=0;i<n;i++) ba->bM[i]+=SETTING_MARG_WEIGHT_FAC*(Mb[i]-Mbsc[i]);
}

/* ---------------------------------------------------------------- EnergyFunctional::marginalizeFrame, EnergyFunctional.cpp:498-610
 * Works on HM/bM only (the frame has no points left: assert at :505). Writes the (n-8)^2 prior and (n-8) vector to HM_out/bM_out; the caller
 * rebuilds the window without frame `idx` (FullSystem::marginalizeFrame, FullSystemMarginalize.cpp:148-213, drops the residuals targeting it). */
static void inv8_lu(const double* A, double* Ai) {           /* Eigen's fixed-size inverse() for 8x8 = partial-pivot LU */
    double M[8][16];
    for (int i=0;i<8;i++) for (int j=0;j<8;j++) { M[i][j]=A[i*8+j]; M[i][8+j]=(i==j); }
    for (int c=0;c<8;c++) {
        int piv=c; for (int r=c+1;r<8;r++) if (fabs(M[r][c])>fabs(M[piv][c])) piv=r;
        if (piv!=c) for (int j=0;j<16;j++) { double t=M[c][j]; M[c][j]=M[piv][j]; M[piv][j]=t; }
        double d=1.0/M[c][c];
        for (int j=0;j<16;j++) M[c][j]*=d;
        for (int r=0;r<8;r++) if (r!=c) { double f=M[r][c]; if (f!=0) for (int j=0;j<16;j++) M[r][j]-=f*M[c][j]; }
    }
    for (int i=0;i<8;i++) for (int j=0;j<8;j++) Ai[i*8+j]=M[i][8+j];
}
void orc_ba_marginalize_frame(OrcBA* ba, int idx, double* HM_out, double* bM_out) {
    int W=ba->W, odim=NDIM(ba), ndim=odim-8;
    double* H=(double*)malloc(8*(size_t)odim*odim); double* b=(double*)malloc(8*odim);
    /* permutation: frame idx moves to the end, the frames behind it move up (:519-540) */
    int* perm=(int*)malloc(sizeof(int)*odim); int k=0;
    for (int i=0;i<odim;i++) { int f=(i<ORC_CPARS)?-1:(i-ORC_CPARS)/8; if (f!=idx) perm[k++]=i; }
    for (int i=0;i<8;i++) perm[k++]=ORC_CPARS+8*idx+i;
    for (int i=0;i<odim;i++) { b[i]=ba->bM[perm[i]]; for (int j=0;j<odim;j++) H[(size_t)i*odim+j]=ba->HM[(size_t)perm[i]*odim+perm[j]]; }
    OrcFrame* fh=&ba->frames[idx];
    for (int i=0;i<8;i++) { H[(size_t)(ndim+i)*odim+ndim+i]+=fh->prior[i]; b[ndim+i]+=fh->prior[i]*fh->delta_prior[i]; }   /* :544-545 */
    double* S=(double*)malloc(8*odim);
    for (int i=0;i<odim;i++) S[i]=sqrt(fabs(H[(size_t)i*odim+i])+10);                                                         /* :552-553 */
    for (int i=0;i<odim;i++) { for (int j=0;j<odim;j++) H[(size_t)i*odim+j]=(1.0/S[i])*H[(size_t)i*odim+j]*(1.0/S[j]); b[i]=(1.0/S[i])*b[i]; }
    double hp[64], hpi[64];
    for (int i=0;i<8;i++) for (int j=0;j<8;j++) hp[i*8+j]=H[(size_t)(ndim+i)*odim+ndim+j];
    for (int i=0;i<64;i++) hp[i]=0.5f*(hp[i]+hp[i]);                                                                            /* :564-567 (sic: hpi+hpi, not the transpose) */
    inv8_lu(hp,hpi);
    for (int i=0;i<64;i++) hpi[i]=0.5f*(hpi[i]+hpi[i]);
    /* bli = BL^T * hpi (ndim x 8); TL -= bli * BL; bTop -= bli * bTail (:570-572) */
    double* bli=(double*)malloc(8*(size_t)ndim*8);
    for (int r=0;r<ndim;r++) for (int c=0;c<8;c++) { double s=0; for (int kk=0;kk<8;kk++) s+=H[(size_t)(ndim+kk)*odim+r]*hpi[kk*8+c]; bli[r*8+c]=s; }
    for (int r=0;r<ndim;r++) { for (int c=0;c<ndim;c++) { double s=0; for (int kk=0;kk<8;kk++) s+=bli[r*8+kk]*H[(size_t)(ndim+kk)*odim+c]; H[(size_t)r*odim+c]-=s; }
        double s=0; for (int kk=0;kk<8;kk++) s+=bli[r*8+kk]*b[ndim+kk]; b[r]-=s; }
    for (int i=0;i<odim;i++) { for (int j=0;j<odim;j++) H[(size_t)i*odim+j]=S[i]*H[(size_t)i*odim+j]*S[j]; b[i]=S[i]*b[i]; }   /* :575-576 */
    for (int i=0;i<ndim;i++) { for (int j=0;j<ndim;j++) HM_out[(size_t)i*ndim+j]=0.5*(H[(size_t)i*odim+j]+H[(size_t)j*odim+i]); bM_out[i]=b[i]; }   /* :579-580 */
    free(H); free(b); free(perm); free(S); free(bli); (void)W;
}

/* ---------------------------------------------------------------- getters for the tests */
void orc_ba_get_residual(OrcBA* ba, int p, int t, double* J74, double* JpJdF8, double* rtz8, int* state3, double* energy3, float* proj19) {
    OrcRes* r=RES(ba,p,t); const OrcJ* J=&r->J; int k=0;
    if (J74) { for (int i=0;i<8;i++) J74[k++]=J->resF[i]; for (int a=0;a<2;a++) for (int i=0;i<6;i++) J74[k++]=J->Jpdxi[a][i];
        for (int a=0;a<2;a++) for (int i=0;i<4;i++) J74[k++]=J->Jpdc[a][i]; J74[k++]=J->Jpdd[0]; J74[k++]=J->Jpdd[1];
        for (int a=0;a<2;a++) for (int i=0;i<8;i++) J74[k++]=J->JIdx[a][i]; for (int a=0;a<2;a++) for (int i=0;i<8;i++) J74[k++]=J->JabF[a][i];
        for (int i=0;i<4;i++) J74[k++]=J->JIdx2[i]; for (int i=0;i<4;i++) J74[k++]=J->JabJIdx[i]; for (int i=0;i<4;i++) J74[k++]=J->Jab2[i]; }
    if (JpJdF8) for (int i=0;i<8;i++) JpJdF8[i]=r->JpJdF[i];
    if (rtz8) for (int i=0;i<8;i++) rtz8[i]=r->res_toZeroF[i];
    if (state3) { state3[0]=r->exists ? r->state_state : -1; state3[1]=r->isActive; state3[2]=r->isLinearized; }
    if (energy3) { energy3[0]=r->state_energy; energy3[1]=r->state_NewEnergy; energy3[2]=r->state_NewEnergyWithOutlier; }
    if (proj19) { for (int i=0;i<3;i++) proj19[i]=r->centerProjectedTo[i]; for (int i=0;i<8;i++) { proj19[3+2*i]=r->projectedTo[i][0]; proj19[4+2*i]=r->projectedTo[i][1]; } }
}
/* bulk: per-slot arrays [P*W] */
void orc_ba_get_slots(OrcBA* ba, int8_t* state, uint8_t* active, float* JpJdF /*[P*W*8]*/, float* energyNew) {
    for (size_t i=0;i<(size_t)ba->P*ba->W;i++) { OrcRes* r=&ba->res[i];
        if (state) state[i]=r->exists ? (int8_t)r->state_state : -1; if (active) active[i]=r->exists && r->isActive;
        if (JpJdF) for (int k=0;k<8;k++) JpJdF[i*8+k]=(float)r->JpJdF[k];
        if (energyNew) energyNew[i]=(float)r->state_NewEnergyWithOutlier; }
}
/* centerProjectedTo of every residual slot [P*W][3] (what CoarseTracker::makeCoarseDepthL0 reads, CoarseTracker.cpp:388-405) */
void orc_ba_get_center_projected(OrcBA* ba, float* out) {
    for (size_t i=0;i<(size_t)ba->P*ba->W;i++) for (int k=0;k<3;k++) out[3*i+k]=ba->res[i].centerProjectedTo[k];
}
void orc_ba_get_points(OrcBA* ba, float* idepth, float* step, float* HdiF, float* bdSumF, float* Hdd, float* bd, float* Hcd4, float* maxRelBaseline) {
    for (int p=0;p<ba->P;p++) { OrcPoint* pt=&ba->pts[p];
        if (idepth) idepth[p]=pt->idepth; if (step) step[p]=pt->step; if (HdiF) HdiF[p]=pt->HdiF; if (bdSumF) bdSumF[p]=pt->bdSumF;
        if (Hdd) Hdd[p]=pt->Hdd_accAF; if (bd) bd[p]=pt->bd_accAF; if (Hcd4) for (int i=0;i<4;i++) Hcd4[4*p+i]=pt->Hcd_accAF[i];
        if (maxRelBaseline) maxRelBaseline[p]=pt->maxRelBaseline; }
}
void orc_ba_get_frame(OrcBA* ba, int f, double* state10, double* worldToCam12, double* evalPT12, float* frameEnergyTH) {
    OrcFrame* fr=&ba->frames[f];
    if (state10) memcpy(state10,fr->state,80); if (worldToCam12) memcpy(worldToCam12,fr->PRE_worldToCam,96);
    if (evalPT12) memcpy(evalPT12,fr->evalPT,96); if (frameEnergyTH) *frameEnergyTH=fr->frameEnergyTH;
}
void orc_ba_get_frame_state_zero(OrcBA* ba, int f, double* sz10) { memcpy(sz10,ba->frames[f].state_zero,80); }
void orc_ba_get_calib(OrcBA* ba, double* value_scaled4) { memcpy(value_scaled4,ba->c_value_scaled,32); }
void orc_ba_get_precalc(OrcBA* ba, float* out /*[W*W][32]*/) {
    for (int i=0;i<ba->W*ba->W;i++) { const OrcPrecalc* pc=&ba->pre[i]; float* o=out+i*32; memset(o,0,128);
        memcpy(o,pc->PRE_KRKiTll,36); memcpy(o+9,pc->PRE_KtTll,12); memcpy(o+12,pc->PRE_RTll_0,36); memcpy(o+21,pc->PRE_tTll_0,12);
        o[24]=pc->PRE_aff_mode[0]; o[25]=pc->PRE_aff_mode[1]; o[26]=pc->PRE_b0_mode; }
}
/* PRE_RTll | PRE_tTll of the current states (FrameFramePrecalc::set, HessianBlocks.cpp:203-209): what ImmaturePoint::linearizeResidual projects with */
void orc_ba_get_precalc_rt(OrcBA* ba, float* out /*[W*W][12]*/, float* aff /*[W*W][2]*/) {
    for (int i=0;i<ba->W*ba->W;i++) { const OrcPrecalc* pc=&ba->pre[i]; memcpy(out+i*12,pc->PRE_RTll,36); memcpy(out+i*12+9,pc->PRE_tTll,12); aff[i*2]=pc->PRE_aff_mode[0]; aff[i*2+1]=pc->PRE_aff_mode[1]; }
}
void orc_ba_get_adjoints(OrcBA* ba, double* adHost, double* adTarget, float* adHTdeltaF) {
    int n=ba->W*ba->W; if (adHost) memcpy(adHost,ba->adHost,8*64*n); if (adTarget) memcpy(adTarget,ba->adTarget,8*64*n); if (adHTdeltaF) memcpy(adHTdeltaF,ba->adHTdeltaF,4*8*n);
}
void orc_ba_get_prior(OrcBA* ba, double* HM, double* bM) { int n=NDIM(ba); memcpy(HM,ba->HM,8*n*n); memcpy(bM,ba->bM,8*n); }
void orc_ba_set_prior(OrcBA* ba, const double* HM, const double* bM) { int n=NDIM(ba); memcpy(ba->HM,HM,8*n*n); memcpy(ba->bM,bM,8*n); }
void orc_ba_set_options(OrcBA* ba, int nthreads, int never_break) {
    if (nthreads<1) nthreads=1; if (nthreads>ORC_MAXTHREADS) nthreads=ORC_MAXTHREADS;
    if (nthreads>ba->nt_alloc) alloc_replicas(ba,nthreads);
    ba->nthreads_used=nthreads; ba->never_break=never_break;
}
/* 1 = linearizeAll chunked over the workers (upstream DSO; this fork runs it single-threaded, FullSystemOptimize.cpp:154-164): only for the
 * all-cores baseline line of bench.py */
void orc_ba_set_linearize_mt(OrcBA* ba, int on) { ba->linearize_mt=on; }
void orc_ba_set_settings(OrcBA* ba, int force_accept_step, double affA, double affB, int min_opt_iterations) {
    ba->force_accept_step=force_accept_step; ba->affine_opt_mode_a=affA; ba->affine_opt_mode_b=affB; ba->min_opt_iterations=min_opt_iterations;
    for (int f=0;f<ba->W;f++) frame_take_data(ba,&ba->frames[f]);
}
int orc_ba_n_rejected(OrcBA* ba) { return ba->n_rejected; }
void orc_ba_get_timers(OrcBA* ba, double* t4) { t4[0]=ba->t_linearize; t4[1]=ba->t_accumulate; t4[2]=ba->t_solve; t4[3]=ba->t_other; }
int orc_ba_counts(OrcBA* ba, int which) { return which==0?ba->resInA: which==1?ba->resInL: ba->resInM; }
void orc_ba_set_idepth(OrcBA* ba, const float* idepth) { for (int p=0;p<ba->P;p++) { OrcPoint* pt=&ba->pts[p]; pt->idepth=idepth[p]; pt->idepth_scaled=idepth[p]; pt->idepth_zero=idepth[p]; pt->idepth_zero_scaled=idepth[p]; pt->deltaF=0; } }
