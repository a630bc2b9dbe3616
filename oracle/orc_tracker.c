/* ORACLE — TEST INFRASTRUCTURE ONLY (see orc_common.h). parity unpinned.
 * Restates: FrameHessian::makeImages (FullSystem/HessianBlocks.cpp:127-190),
 * CoarseTracker::makeCoarseDepthL0 steps 1-5 (FullSystem/CoarseTracker.cpp:382-538),
 * CoarseTracker::calcRes (:891-1049), calcGSSSE (:828-885) with Accumulator9
 * (OptimizationBackend/MatrixAccumulators.h:982-1166), trackNewestCoarse (:1073-1259),
 * getInterpolatedElement33 (util/globalFuncs.h:75-89), setGlobalCalib level rule (util/globalCalib.cpp:45-105). */
#include "orc_common.h"

/* ---------------- pyramid geometry: util/globalCalib.cpp:45-105 ---------------- */
int orc_pyr_levels(int w, int h) {
    int wl=w, hl=h, lv=1;
    while (wl%2==0 && hl%2==0 && wl*hl>5000 && lv<ORC_PYR_MAX) { wl/=2; hl/=2; lv++; }
    return lv;
}
void orc_pyr_calib(int w, int h, int levels, float fx, float fy, float cx, float cy,
                   int* wl, int* hl, float* fxl, float* fyl, float* cxl, float* cyl) {
    wl[0]=w; hl[0]=h; fxl[0]=fx; fyl[0]=fy; cxl[0]=cx; cyl[0]=cy;
    for (int l=1;l<levels;l++) {
        wl[l]=w>>l; hl[l]=h>>l;
        fxl[l]=fxl[l-1]*0.5; fyl[l]=fyl[l-1]*0.5;
        cxl[l]=(cxl[0]+0.5)/((int)1<<l)-0.5; cyl[l]=(cyl[0]+0.5)/((int)1<<l)-0.5;
    }
}
long orc_pyr_offset(int w, int h, int lvl) { long o=0; for (int l=0;l<lvl;l++) o += (long)(w>>l)*(h>>l); return o; }

/* ---------------- a1: makeImages, HessianBlocks.cpp:127-190 ----------------
 * dI_all: AoS 3 floats/pixel, levels concatenated (offset orc_pyr_offset). abs_all: 1 float/pixel.
 * Rows 0 and h-1 of dx,dy,abs are uninitialised heap in the reference: zero here.
 * Bg: gamma table B[256] (HessianBlocks.h:400-406) or NULL -> gw=1. */
void orc_make_images(const float* color, int w, int h, int levels, const float* Bg, float* dI_all, float* abs_all) {
    long tot = orc_pyr_offset(w,h,levels);
    memset(dI_all, 0, sizeof(float)*3*tot); memset(abs_all, 0, sizeof(float)*tot);
    for (long i=0;i<(long)w*h;i++) dI_all[3*i] = color[i];
    for (int lvl=0; lvl<levels; lvl++) {
        int wl=w>>lvl, hl=h>>lvl;
        float* dI_l = dI_all + 3*orc_pyr_offset(w,h,lvl);
        float* dabs = abs_all + orc_pyr_offset(w,h,lvl);
        if (lvl>0) {
            int wlm1 = w>>(lvl-1);
            const float* dI_lm = dI_all + 3*orc_pyr_offset(w,h,lvl-1);
            for (int y=0;y<hl;y++) for (int x=0;x<wl;x++)
                dI_l[3*(x+y*wl)] = 0.25f * (dI_lm[3*(2*x+2*y*wlm1)] + dI_lm[3*(2*x+1+2*y*wlm1)] +
                                            dI_lm[3*(2*x+2*y*wlm1+wlm1)] + dI_lm[3*(2*x+1+2*y*wlm1+wlm1)]);
        }
        for (int idx=wl; idx<wl*(hl-1); idx++) {
            float dx = 0.5f*(dI_l[3*(idx+1)] - dI_l[3*(idx-1)]);
            float dy = 0.5f*(dI_l[3*(idx+wl)] - dI_l[3*(idx-wl)]);
            if (!isfinite(dx)) dx=0;
            if (!isfinite(dy)) dy=0;
            dI_l[3*idx+1]=dx; dI_l[3*idx+2]=dy;
            dabs[idx] = dx*dx+dy*dy;
            if (Bg) { int c = dI_l[3*idx]+0.5f; if (c<5) c=5; if (c>250) c=250; float gw = Bg[c+1]-Bg[c]; dabs[idx] *= gw*gw; }
        }
    }
}

/* util/globalFuncs.h:75-89 — pointwise in `real` */
static inline void interp33(const float* mat, real x, real y, int width, real out[3]) {
    int ix=(int)x, iy=(int)y;
    real dx=x-ix, dy=y-iy, dxdy=dx*dy;
    const float* bp = mat + 3*(ix+iy*width);
    for (int c=0;c<3;c++)
        out[c] = dxdy*(real)bp[3*(1+width)+c] + (dy-dxdy)*(real)bp[3*width+c] + (dx-dxdy)*(real)bp[3+c] + (1-dx-dy+dxdy)*(real)bp[c];
}

/* ---------------- tracker state ---------------- */
typedef struct {
    int levels, w[ORC_PYR_MAX], h[ORC_PYR_MAX];
    float fx[ORC_PYR_MAX], fy[ORC_PYR_MAX], cx[ORC_PYR_MAX], cy[ORC_PYR_MAX];
    float Ki[ORC_PYR_MAX][9];
    float *idepth[ORC_PYR_MAX], *weightSums[ORC_PYR_MAX], *weightSums_bak[ORC_PYR_MAX];
    float *pc_u[ORC_PYR_MAX], *pc_v[ORC_PYR_MAX], *pc_idepth[ORC_PYR_MAX], *pc_color[ORC_PYR_MAX];
    int pc_n[ORC_PYR_MAX];
    /* buf_warped_* (CoarseTracker.h) */
    real *bw_idepth, *bw_u, *bw_v, *bw_dx, *bw_dy, *bw_res, *bw_w, *bw_ref; int bw_n;
    const float* dI_ref;    /* borrowed: reference frame pyramid (AoS 3) */
    double lastResiduals[5], lastFlow[3];
    long n_calcres, n_calcgs;   /* call counters for the baseline report */
    double affA, affB;          /* setting_affineOptModeA / B (util/settings.cpp:128-129); set by orc_trk_set_affine_modes, default 1e12 / 1e8 */
    int aff_set;
} OrcTracker;

OrcTracker* orc_trk_create(int w, int h, int levels, float fx, float fy, float cx, float cy) {
    OrcTracker* t = (OrcTracker*)calloc(1,sizeof(OrcTracker));
    t->levels = levels;
    orc_pyr_calib(w,h,levels,fx,fy,cx,cy,t->w,t->h,t->fx,t->fy,t->cx,t->cy);
    for (int l=0;l<levels;l++) {
        /* K^-1 of a pinhole K (CoarseTracker::makeK, CoarseTracker.cpp:97-141) */
        float* Ki=t->Ki[l]; memset(Ki,0,sizeof(float)*9);
        Ki[0]=1.0f/t->fx[l]; Ki[4]=1.0f/t->fy[l]; Ki[2]=-t->cx[l]/t->fx[l]; Ki[5]=-t->cy[l]/t->fy[l]; Ki[8]=1;
        size_t n=(size_t)t->w[l]*t->h[l];
        t->idepth[l]=(float*)calloc(n,4); t->weightSums[l]=(float*)calloc(n,4); t->weightSums_bak[l]=(float*)calloc(n,4);
        t->pc_u[l]=(float*)calloc(n,4); t->pc_v[l]=(float*)calloc(n,4); t->pc_idepth[l]=(float*)calloc(n,4); t->pc_color[l]=(float*)calloc(n,4);
    }
    size_t n0=(size_t)w*h+8;
    t->bw_idepth=(real*)calloc(n0,sizeof(real)); t->bw_u=(real*)calloc(n0,sizeof(real)); t->bw_v=(real*)calloc(n0,sizeof(real));
    t->bw_dx=(real*)calloc(n0,sizeof(real)); t->bw_dy=(real*)calloc(n0,sizeof(real)); t->bw_res=(real*)calloc(n0,sizeof(real));
    t->bw_w=(real*)calloc(n0,sizeof(real)); t->bw_ref=(real*)calloc(n0,sizeof(real));
    return t;
}
void orc_trk_destroy(OrcTracker* t) {
    for (int l=0;l<t->levels;l++) { free(t->idepth[l]); free(t->weightSums[l]); free(t->weightSums_bak[l]);
        free(t->pc_u[l]); free(t->pc_v[l]); free(t->pc_idepth[l]); free(t->pc_color[l]); }
    free(t->bw_idepth); free(t->bw_u); free(t->bw_v); free(t->bw_dx); free(t->bw_dy); free(t->bw_res); free(t->bw_w); free(t->bw_ref);
    free(t);
}

/* ---------------- a2: makeCoarseDepthL0 steps 1-5, CoarseTracker.cpp:382-538 ----------------
 * inputs per IN residual targeting the newest KF: centerProjectedTo (Ku,Kv,new_idepth) and the point's HdiF. */
void orc_trk_set_ref(OrcTracker* t, const float* dI_ref, int n, const float* Ku, const float* Kv,
                     const float* new_idepth, const float* HdiF) {
    t->dI_ref = dI_ref;
    int w0=t->w[0], h0=t->h[0];
    memset(t->idepth[0],0,sizeof(float)*w0*h0); memset(t->weightSums[0],0,sizeof(float)*w0*h0);
    for (int i=0;i<n;i++) {                                         /* step 1, :388-405 */
        int u = Ku[i]+0.5f, v = Kv[i]+0.5f;
        float weight = sqrtf(1e-3 / (HdiF[i]+1e-12));
        t->idepth[0][u+w0*v] += new_idepth[i]*weight;
        t->weightSums[0][u+w0*v] += weight;
    }
    for (int lvl=1; lvl<t->levels; lvl++) {                         /* step 2, :408-433 */
        int wl=t->w[lvl], hl=t->h[lvl], wlm1=t->w[lvl-1];
        float *il=t->idepth[lvl], *wsl=t->weightSums[lvl], *ilm=t->idepth[lvl-1], *wslm=t->weightSums[lvl-1];
        for (int y=0;y<hl;y++) for (int x=0;x<wl;x++) {
            int b=2*x+2*y*wlm1;
            il[x+y*wl] = ilm[b]+ilm[b+1]+ilm[b+wlm1]+ilm[b+wlm1+1];
            wsl[x+y*wl] = wslm[b]+wslm[b+1]+wslm[b+wlm1]+wslm[b+wlm1+1];
        }
    }
    for (int lvl=0; lvl<2 && lvl<t->levels; lvl++) {                /* step 3, :437-464 (diagonal dilate) */
        int wl=t->w[lvl], wh=t->w[lvl]*t->h[lvl]-t->w[lvl];
        float *ws=t->weightSums[lvl], *bak=t->weightSums_bak[lvl], *id=t->idepth[lvl];
        memcpy(bak, ws, sizeof(float)*t->w[lvl]*t->h[lvl]);
        for (int i=wl;i<wh;i++) if (bak[i] <= 0) {
            float sum=0,num=0,numn=0;
            /* reference reads bak[w*h] / bak[-1] at the range ends (one past / before the array): skipped here */
            if (i+1+wl<t->w[lvl]*t->h[lvl] && bak[i+1+wl]>0) { sum+=id[i+1+wl]; num+=bak[i+1+wl]; numn++; }
            if (i-1-wl>=0 && bak[i-1-wl]>0) { sum+=id[i-1-wl]; num+=bak[i-1-wl]; numn++; }
            if (bak[i+wl-1]>0) { sum+=id[i+wl-1]; num+=bak[i+wl-1]; numn++; }
            if (bak[i-wl+1]>0) { sum+=id[i-wl+1]; num+=bak[i-wl+1]; numn++; }
            if (numn>0) { id[i]=sum/numn; ws[i]=num/numn; }
        }
    }
    for (int lvl=2; lvl<t->levels; lvl++) {                         /* step 4, :468-489 (axis dilate) */
        int wl=t->w[lvl], wh=t->w[lvl]*t->h[lvl]-t->w[lvl];
        float *ws=t->weightSums[lvl], *bak=t->weightSums_bak[lvl], *id=t->idepth[lvl];
        memcpy(bak, ws, sizeof(float)*t->w[lvl]*t->h[lvl]);
        for (int i=wl;i<wh;i++) if (bak[i] <= 0) {
            float sum=0,num=0,numn=0;
            if (bak[i+1]>0) { sum+=id[i+1]; num+=bak[i+1]; numn++; }
            if (bak[i-1]>0) { sum+=id[i-1]; num+=bak[i-1]; numn++; }
            if (bak[i+wl]>0) { sum+=id[i+wl]; num+=bak[i+wl]; numn++; }
            if (bak[i-wl]>0) { sum+=id[i-wl]; num+=bak[i-wl]; numn++; }
            if (numn>0) { id[i]=sum/numn; ws[i]=num/numn; }
        }
    }
    for (int lvl=0; lvl<t->levels; lvl++) {                         /* step 5, :493-538 */
        float *ws=t->weightSums[lvl], *id=t->idepth[lvl];
        const float* dIRef = dI_ref + 3*orc_pyr_offset(t->w[0],t->h[0],lvl);
        int wl=t->w[lvl], hl=t->h[lvl], n_=0;
        for (int y=2;y<hl-2;y++) for (int x=2;x<wl-2;x++) {
            int i=x+y*wl;
            if (ws[i] > 0) {
                id[i] /= ws[i];
                t->pc_u[lvl][n_]=x; t->pc_v[lvl][n_]=y; t->pc_idepth[lvl][n_]=id[i]; t->pc_color[lvl][n_]=dIRef[3*i];
                if (!isfinite(t->pc_color[lvl][n_]) || !(id[i]>0)) { id[i]=-1; continue; }
                n_++;
            } else id[i]=-1;
            ws[i]=1;
        }
        t->pc_n[lvl]=n_;
    }
}
/* direct injection of level point clouds (synthetic stress: n_0 = 250k by construction) */
void orc_trk_set_pc(OrcTracker* t, const float* dI_ref, int lvl, int n, const float* u, const float* v, const float* idepth, const float* color) {
    t->dI_ref = dI_ref; t->pc_n[lvl]=n;
    memcpy(t->pc_u[lvl],u,4*n); memcpy(t->pc_v[lvl],v,4*n); memcpy(t->pc_idepth[lvl],idepth,4*n); memcpy(t->pc_color[lvl],color,4*n);
}
/* dense=1 sampling of one mask cluster, CoarseTracker.cpp:621-655 (incl. the off-by-one store index; the never-written slot [old pc_n] is zeroed) */
int orc_trk_append_plane_points(OrcTracker* t, const float* mask, const float dir[3], float dis_plane, int refMaskColor, const int rect[4]) {
    const int w=t->w[0], h=t->h[0], minx=rect[0], maxx=rect[1], miny=rect[2], maxy=rect[3];
    if (maxx>w-1||minx<1||maxy>h-1||miny<1) return 0;
    if (refMaskColor==0) return 0;
    const float* Ki=t->Ki[0];
    const int last=t->pc_n[0];
    t->pc_u[0][last]=t->pc_v[0][last]=t->pc_idepth[0][last]=t->pc_color[0][last]=0;
    for (int x=minx;x<maxx;x++) for (int y=miny;y<maxy;y++) {
        if (mask[x+y*w]!=refMaskColor) continue;
        if (x%5==0&&y%5==0) {
            float tv[3]; for (int j=0;j<3;j++) tv[j]=dir[0]*Ki[j]+dir[1]*Ki[3+j]+dir[2]*Ki[6+j];      /* (dir^T Ki) first, then times (x,y,1) */
            float nid=tv[0]*x+tv[1]*y+tv[2]*1;
            nid/=-dis_plane;
            t->pc_u[0][t->pc_n[0]+1]=x; t->pc_v[0][t->pc_n[0]+1]=y; t->pc_idepth[0][t->pc_n[0]+1]=nid; t->pc_color[0][t->pc_n[0]+1]=t->dI_ref[3*(x+y*w)];
            t->pc_n[0]++;
        }
    }
    return t->pc_n[0]-last;
}
int orc_trk_get_pc(OrcTracker* t, int lvl, float* u, float* v, float* idepth, float* color) {
    int n=t->pc_n[lvl];
    if (u) { memcpy(u,t->pc_u[lvl],4*n); memcpy(v,t->pc_v[lvl],4*n); memcpy(idepth,t->pc_idepth[lvl],4*n); memcpy(color,t->pc_color[lvl],4*n); }
    return n;
}
void orc_trk_get_depth(OrcTracker* t, int lvl, float* idepth, float* weightSums) {
    size_t n=(size_t)t->w[lvl]*t->h[lvl]; memcpy(idepth,t->idepth[lvl],4*n); memcpy(weightSums,t->weightSums[lvl],4*n);
}

/* ---------------- a3: calcRes, CoarseTracker.cpp:891-1049 ----------------
 * R,t = refToNew (double); affLL = fromToVecExposure(ref,new) cast to float (:909). out6 as :1040-1046. */
void orc_trk_calc_res(OrcTracker* T, const float* dI_new, int lvl, const double R[9], const double tr[3],
                      const float affLL[2], float cutoffTH, double out6[6]) {
    T->n_calcres++;
    int wl=T->w[lvl], hl=T->h[lvl];
    const float* dINewl = dI_new + 3*orc_pyr_offset(T->w[0],T->h[0],lvl);
    real fxl=T->fx[lvl], fyl=T->fy[lvl], cxl=T->cx[lvl], cyl=T->cy[lvl];
    real RKi[9], Ki[9], t[3];
    for (int i=0;i<9;i++) Ki[i]=T->Ki[lvl][i];
    { float Rf[9]; for (int i=0;i<9;i++) Rf[i]=(float)R[i];      /* (R.cast<float>() * Ki[lvl]) :907 */
      for (int i=0;i<3;i++) for (int j=0;j<3;j++) { real s=0; for (int k=0;k<3;k++) s += (real)Rf[i*3+k]*Ki[k*3+j]; RKi[i*3+j]=s; } }
    for (int i=0;i<3;i++) t[i]=(real)(float)tr[i];
    real aL=affLL[0], bL=affLL[1];
    /* E and the shift sums are `float` scalars in the reference; sum mode 0 keeps fp64 shadows */
    float E=0; double E64=0; int numTermsInE=0, numTermsInWarped=0, numSaturated=0;
    float sT=0, sRT=0, sN=0; double sT64=0, sRT64=0;
    real huber=SETTING_HUBER_TH;
    real maxEnergy = 2*huber*(real)cutoffTH - huber*huber;
    int nl=T->pc_n[lvl];
    const float *pu=T->pc_u[lvl], *pv=T->pc_v[lvl], *pid=T->pc_idepth[lvl], *pcol=T->pc_color[lvl];
    for (int i=0;i<nl;i++) {
        real id=pid[i], x=pu[i], y=pv[i];
        real pt0=RKi[0]*x+RKi[1]*y+RKi[2]+t[0]*id, pt1=RKi[3]*x+RKi[4]*y+RKi[5]+t[1]*id, pt2=RKi[6]*x+RKi[7]*y+RKi[8]+t[2]*id;
        real u=pt0/pt2, v=pt1/pt2, Ku=fxl*u+cxl, Kv=fyl*v+cyl, new_idepth=id/pt2;
        if (lvl==0 && i%32==0) {                                     /* :948-979 */
            real a0=Ki[0]*x+Ki[1]*y+Ki[2], a1=Ki[3]*x+Ki[4]*y+Ki[5], a2=Ki[6]*x+Ki[7]*y+Ki[8];
            real T0=a0+t[0]*id, T1=a1+t[1]*id, T2=a2+t[2]*id;
            real KuT=fxl*(T0/T2)+cxl, KvT=fyl*(T1/T2)+cyl;
            real U0=a0-t[0]*id, U1=a1-t[1]*id, U2=a2-t[2]*id;
            real KuT2=fxl*(U0/U2)+cxl, KvT2=fyl*(U1/U2)+cyl;
            real r0=RKi[0]*x+RKi[1]*y+RKi[2]-t[0]*id, r1=RKi[3]*x+RKi[4]*y+RKi[5]-t[1]*id, r2=RKi[6]*x+RKi[7]*y+RKi[8]-t[2]*id;
            real Ku3=fxl*(r0/r2)+cxl, Kv3=fyl*(r1/r2)+cyl;
            real d1=(KuT-x)*(KuT-x)+(KvT-y)*(KvT-y), d2=(KuT2-x)*(KuT2-x)+(KvT2-y)*(KvT2-y);
            real d3=(Ku-x)*(Ku-x)+(Kv-y)*(Kv-y), d4=(Ku3-x)*(Ku3-x)+(Kv3-y)*(Kv3-y);
            sT += (float)d1; sT += (float)d2; sRT += (float)d3; sRT += (float)d4; sN += 2;
            sT64 += (double)d1 + (double)d2; sRT64 += (double)d3 + (double)d4;
        }
        if (!(Ku>2 && Kv>2 && Ku<wl-3 && Kv<hl-3 && new_idepth>0)) continue;      /* :981 */
        real refColor=pcol[i], hit[3];
        interp33(dINewl, Ku, Kv, wl, hit);
        if (!isfinite((float)hit[0])) continue;
        real residual = hit[0] - (real)(float)(aL*refColor + bL);
        real ar = residual<0 ? -residual : residual;
        real hw = ar < huber ? 1 : huber/ar;
        if (ar > (real)cutoffTH) { E += (float)maxEnergy; E64 += (double)maxEnergy; numTermsInE++; numSaturated++; }
        else {
            real e = hw*residual*residual*(2-hw);
            E += (float)e; E64 += (double)e; numTermsInE++;
            int k=numTermsInWarped++;
            T->bw_idepth[k]=new_idepth; T->bw_u[k]=u; T->bw_v[k]=v; T->bw_dx[k]=hit[1]; T->bw_dy[k]=hit[2];
            T->bw_res[k]=residual; T->bw_w[k]=hw; T->bw_ref[k]=pcol[i];
        }
    }
    while (numTermsInWarped%4!=0) {                                   /* :1018-1029 */
        int k=numTermsInWarped++;
        T->bw_idepth[k]=0; T->bw_u[k]=0; T->bw_v[k]=0; T->bw_dx[k]=0; T->bw_dy[k]=0; T->bw_res[k]=0; T->bw_w[k]=0; T->bw_ref[k]=0;
    }
    T->bw_n = numTermsInWarped;
#ifdef ORC_FAST
    int ref=1;
#else
    int ref=orc_sum_mode;
#endif
    out6[0] = ref ? (double)E : E64;
    out6[1] = numTermsInE;
    out6[2] = ref ? (double)(sT/(sN+0.1f)) : sT64/((double)sN+0.1);
    out6[3] = 0;
    out6[4] = ref ? (double)(sRT/(sN+0.1f)) : sRT64/((double)sN+0.1);
    out6[5] = numSaturated/(float)numTermsInE;
}

/* ---------------- a4: calcGSSSE + Accumulator9, CoarseTracker.cpp:828-885 ----------------
 * Accumulator9::updateSSE_eighted (MatrixAccumulators.h:1091-1166): 45 upper-tri entries x 4 SSE lanes,
 * shiftUp every >1000 updates (:1325-1344), finish adds the 4 lanes of tier 1m (:1001-1017).
 * H,b divided by the PADDED count, then scaled (:873-884). */
#ifdef ORC_FAST
#include <xmmintrin.h>
/* The timed baseline's calcGSSSE as the reference writes it: four points per step in __m128 lanes, the nine J rows as in CoarseTracker.cpp:846-867,
 * Accumulator9::updateSSE_eighted (MatrixAccumulators.h:1091-1166: 45 x {load, mul, add, store} on SSEData) and the three-tier shiftUp. Same lane arithmetic as
 * the portable loop below (which the strict builds run). */
static void calc_gs_sse(OrcTracker* T, int lvl, float aff_a, float b0, float* S /* 45*4, aligned */, float* S1k, float* S1m) {
    const __m128 fxl = _mm_set1_ps(T->fx[lvl]), fyl = _mm_set1_ps(T->fy[lvl]), b0v = _mm_set1_ps(b0), a = _mm_set1_ps(aff_a);
    const __m128 one = _mm_set1_ps(1.f), minusOne = _mm_set1_ps(-1.f), zero = _mm_set1_ps(0.f);
    float numIn1 = 0, numIn1k = 0;
    const int n = T->bw_n;
    for (int i = 0; i < n; i += 4) {
        const __m128 dx = _mm_mul_ps(_mm_loadu_ps(T->bw_dx + i), fxl), dy = _mm_mul_ps(_mm_loadu_ps(T->bw_dy + i), fyl);
        const __m128 u = _mm_loadu_ps(T->bw_u + i), v = _mm_loadu_ps(T->bw_v + i), id = _mm_loadu_ps(T->bw_idepth + i);
        __m128 J[9];
        J[0] = _mm_mul_ps(id, dx);
        J[1] = _mm_mul_ps(id, dy);
        J[2] = _mm_sub_ps(zero, _mm_mul_ps(id, _mm_add_ps(_mm_mul_ps(u, dx), _mm_mul_ps(v, dy))));
        J[3] = _mm_sub_ps(zero, _mm_add_ps(_mm_mul_ps(_mm_mul_ps(u, v), dx), _mm_mul_ps(dy, _mm_add_ps(one, _mm_mul_ps(v, v)))));
        J[4] = _mm_add_ps(_mm_mul_ps(_mm_mul_ps(u, v), dy), _mm_mul_ps(dx, _mm_add_ps(one, _mm_mul_ps(u, u))));
        J[5] = _mm_sub_ps(_mm_mul_ps(u, dy), _mm_mul_ps(v, dx));
        J[6] = _mm_mul_ps(a, _mm_sub_ps(b0v, _mm_loadu_ps(T->bw_ref + i)));
        J[7] = minusOne;
        J[8] = _mm_loadu_ps(T->bw_res + i);
        const __m128 w = _mm_loadu_ps(T->bw_w + i);
        float* pt = S;
        for (int r = 0; r < 9; r++) {
            const __m128 Jw = _mm_mul_ps(J[r], w);
            for (int c = r; c < 9; c++) { _mm_store_ps(pt, _mm_add_ps(_mm_load_ps(pt), _mm_mul_ps(Jw, J[c]))); pt += 4; }
        }
        numIn1++;
        if (numIn1 > 1000) { for (int k = 0; k < 180; k += 4) { _mm_store_ps(S1k + k, _mm_add_ps(_mm_load_ps(S1k + k), _mm_load_ps(S + k))); _mm_store_ps(S + k, zero); } numIn1k += numIn1; numIn1 = 0; }
        if (numIn1k > 1000) { for (int k = 0; k < 180; k += 4) { _mm_store_ps(S1m + k, _mm_add_ps(_mm_load_ps(S1m + k), _mm_load_ps(S1k + k))); _mm_store_ps(S1k + k, zero); } numIn1k = 0; }
    }
    for (int k = 0; k < 180; k++) { S1k[k] += S[k]; S[k] = 0; }
    for (int k = 0; k < 180; k++) { S1m[k] += S1k[k]; S1k[k] = 0; }
}
#endif
void orc_trk_calc_gs(OrcTracker* T, int lvl, float aff_a /* (float)affLL[0] */, float b0, double H_out[64], double b_out[8]) {
    T->n_calcgs++;
    static float S[45*4] __attribute__((aligned(16))), S1k[45*4] __attribute__((aligned(16))), S1m[45*4] __attribute__((aligned(16))); static double D[45];
    memset(S,0,sizeof(S)); memset(S1k,0,sizeof(S1k)); memset(S1m,0,sizeof(S1m)); memset(D,0,sizeof(D));
    float numIn1=0, numIn1k=0;
    real fxl=T->fx[lvl], fyl=T->fy[lvl];
    int n=T->bw_n;
#ifdef ORC_FAST
    calc_gs_sse(T, lvl, aff_a, b0, S, S1k, S1m);
    n = 0;                                                       /* the portable loop below is the strict builds' */
#endif
    for (int i=0;i<n;i+=4) {
        real J[9][4], w[4];
        for (int l=0;l<4;l++) {
            int k=i+l;
            real dx=T->bw_dx[k]*fxl, dy=T->bw_dy[k]*fyl, u=T->bw_u[k], v=T->bw_v[k], id=T->bw_idepth[k];
            J[0][l]=id*dx; J[1][l]=id*dy; J[2][l]=0-(id*((u*dx)+(v*dy)));
            J[3][l]=0-(((u*v)*dx)+(dy*(1+(v*v)))); J[4][l]=((u*v)*dy)+(dx*(1+(u*u))); J[5][l]=(u*dy)-(v*dx);
            J[6][l]=(real)aff_a*((real)b0-T->bw_ref[k]); J[7][l]=-1; J[8][l]=T->bw_res[k]; w[l]=T->bw_w[k];
        }
        int idx=0;
        for (int r=0;r<9;r++) for (int c=r;c<9;c++) {
            for (int l=0;l<4;l++) {
                real Jw = J[r][l]*w[l];
                real p = Jw*J[c][l];
                S[idx*4+l] += (float)p;
#ifndef ORC_FAST
                D[idx] += (double)p;
#endif
            }
            idx++;
        }
        numIn1++;
        if (numIn1 > 1000) { for (int k=0;k<180;k++) { S1k[k]+=S[k]; S[k]=0; } numIn1k+=numIn1; numIn1=0; }
        if (numIn1k > 1000) { for (int k=0;k<180;k++) { S1m[k]+=S1k[k]; S1k[k]=0; } numIn1k=0; }
    }
    for (int k=0;k<180;k++) { S1k[k]+=S[k]; S[k]=0; }
    for (int k=0;k<180;k++) { S1m[k]+=S1k[k]; S1k[k]=0; }
    double Hf[81]; int idx=0;
#ifdef ORC_FAST
    int ref=1;
#else
    int ref=orc_sum_mode;
#endif
    for (int r=0;r<9;r++) for (int c=r;c<9;c++) {
        double d = ref ? (double)(float)(S1m[idx*4]+S1m[idx*4+1]+S1m[idx*4+2]+S1m[idx*4+3]) : D[idx];
        Hf[r*9+c]=Hf[c*9+r]=d; idx++;
    }
    n = T->bw_n;
    double inv = ref ? (double)(1.0f/n) : 1.0/(double)n;
    static const double sc[8] = {SCALE_XI_ROT,SCALE_XI_ROT,SCALE_XI_ROT,SCALE_XI_TRANS,SCALE_XI_TRANS,SCALE_XI_TRANS,SCALE_A,SCALE_B};
    for (int r=0;r<8;r++) { for (int c=0;c<8;c++) H_out[r*8+c] = Hf[r*9+c]*inv*sc[r]*sc[c]; b_out[r] = Hf[r*9+8]*inv*sc[r]; }
}

/* ---------------- trackNewestCoarse, CoarseTracker.cpp:1073-1259 ----------------
 * T_io: refToNew (3x4). aff_io: aff_g2l of the new frame (a,b). ref_aff: lastRef_aff_g2l. exposures {ref,new}.
 * returns 1 on success. */
int orc_trk_track(OrcTracker* T, const float* dI_new, double T_io[12], double aff_io[2], const double ref_aff[2],
                  const float exposures[2], int coarsestLvl, const double minResForAbort[5],
                  double lastResiduals_out[5], double lastFlow_out[3]) {
    for (int i=0;i<5;i++) T->lastResiduals[i]=NAN;
    for (int i=0;i<3;i++) T->lastFlow[i]=1000;
    static const int maxIterations[5] = {10,20,50,50,50};
    const float lambdaExtrapolationLimit = 0.001f;
    double cur[12]; memcpy(cur,T_io,sizeof(cur));
    double aff_cur[2]={aff_io[0],aff_io[1]};
    int haveRepeated=0, ok=1;
    for (int lvl=coarsestLvl; lvl>=0; lvl--) {
        double H[64], b[8], resOld[6], resNew[6], aLL[2]; float affLLf[2];
        float levelCutoffRepeat=1;
        double R[9]={cur[0],cur[1],cur[2],cur[4],cur[5],cur[6],cur[8],cur[9],cur[10]}, tr[3]={cur[3],cur[7],cur[11]};
        orc_aff_from_to(exposures[0],exposures[1],ref_aff[0],ref_aff[1],aff_cur[0],aff_cur[1],aLL);
        affLLf[0]=(float)aLL[0]; affLLf[1]=(float)aLL[1];
        orc_trk_calc_res(T,dI_new,lvl,R,tr,affLLf,SETTING_COARSE_CUTOFF_TH*levelCutoffRepeat,resOld);
        while (resOld[5] > 0.6 && levelCutoffRepeat < 50) {
            levelCutoffRepeat*=2;
            orc_trk_calc_res(T,dI_new,lvl,R,tr,affLLf,SETTING_COARSE_CUTOFF_TH*levelCutoffRepeat,resOld);
        }
        orc_trk_calc_gs(T,lvl,(float)aLL[0],(float)ref_aff[1],H,b);
        float lambda=0.01f;
        for (int it=0; it<maxIterations[lvl]; it++) {
            double Hl[64], nb[8], inc[8];
            memcpy(Hl,H,sizeof(Hl)); for (int i=0;i<8;i++) { Hl[i*8+i]*=(1+lambda); nb[i]=-b[i]; }
            orc_ldlt_solve(8,Hl,nb,inc);
            {   /* :1140-1162: a and/or b fixed */
                const double mA = T->aff_set ? T->affA : 1e12, mB = T->aff_set ? T->affB : 1e8;
                if (mA < 0 && mB < 0) { double H6[36], x6[6]; for (int i=0;i<6;i++) for (int j=0;j<6;j++) H6[i*6+j]=Hl[i*8+j];
                    orc_ldlt_solve(6,H6,nb,x6); for (int i=0;i<6;i++) inc[i]=x6[i]; inc[6]=inc[7]=0; }
                if (!(mA < 0) && mB < 0) { double H7[49], x7[7]; for (int i=0;i<7;i++) for (int j=0;j<7;j++) H7[i*7+j]=Hl[i*8+j];
                    orc_ldlt_solve(7,H7,nb,x7); for (int i=0;i<7;i++) inc[i]=x7[i]; inc[7]=0; }
                if (mA < 0 && !(mB < 0)) { double Hs[64], bs[8], H7[49], x7[7]; memcpy(Hs,Hl,sizeof(Hs)); memcpy(bs,nb,sizeof(bs));
                    for (int i=0;i<8;i++) Hs[i*8+6]=Hs[i*8+7];                       /* HlStitch.col(6) = col(7); then row(6) = row(7) */
                    for (int j=0;j<8;j++) Hs[6*8+j]=Hs[7*8+j];
                    bs[6]=bs[7];
                    for (int i=0;i<7;i++) for (int j=0;j<7;j++) H7[i*7+j]=Hs[i*8+j];
                    orc_ldlt_solve(7,H7,bs,x7); for (int i=0;i<6;i++) inc[i]=x7[i]; inc[6]=0; inc[7]=x7[6]; }
            }
            float extrapFac=1;
            if (lambda < lambdaExtrapolationLimit) extrapFac = sqrt(sqrt(lambdaExtrapolationLimit/lambda));
            for (int i=0;i<8;i++) inc[i]*=extrapFac;
            double incS[8]; memcpy(incS,inc,sizeof(incS));
            for (int i=0;i<3;i++) incS[i]*=SCALE_XI_ROT;      /* labels swapped vs order, SURVEY App. C.2 */
            for (int i=3;i<6;i++) incS[i]*=SCALE_XI_TRANS;
            incS[6]*=SCALE_A; incS[7]*=SCALE_B;
            double s=0; for (int i=0;i<8;i++) s+=incS[i];
            if (!isfinite(s)) memset(incS,0,sizeof(incS));
            double E[12], Tn[12]; orc_se3_exp(incS,E); orc_se3_mul(E,cur,Tn);
            double aff_new[2]={aff_cur[0]+incS[6], aff_cur[1]+incS[7]};
            double Rn[9]={Tn[0],Tn[1],Tn[2],Tn[4],Tn[5],Tn[6],Tn[8],Tn[9],Tn[10]}, tn[3]={Tn[3],Tn[7],Tn[11]};
            double aLLn[2]; float aLLnf[2];
            orc_aff_from_to(exposures[0],exposures[1],ref_aff[0],ref_aff[1],aff_new[0],aff_new[1],aLLn);
            aLLnf[0]=(float)aLLn[0]; aLLnf[1]=(float)aLLn[1];
            orc_trk_calc_res(T,dI_new,lvl,Rn,tn,aLLnf,SETTING_COARSE_CUTOFF_TH*levelCutoffRepeat,resNew);
            int accept = (resNew[0]/resNew[1]) < (resOld[0]/resOld[1]);
            if (accept) {
                orc_trk_calc_gs(T,lvl,(float)aLLn[0],(float)ref_aff[1],H,b);
                memcpy(resOld,resNew,sizeof(resOld)); aff_cur[0]=aff_new[0]; aff_cur[1]=aff_new[1]; memcpy(cur,Tn,sizeof(cur));
                lambda*=0.5f;
            } else { lambda*=4; if (lambda<lambdaExtrapolationLimit) lambda=lambdaExtrapolationLimit; }
            double nrm=0; for (int i=0;i<8;i++) nrm+=inc[i]*inc[i];
            if (!(sqrt(nrm) > 1e-3)) break;
        }
        T->lastResiduals[lvl] = sqrtf((float)(resOld[0]/resOld[1]));
        T->lastFlow[0]=resOld[2]; T->lastFlow[1]=resOld[3]; T->lastFlow[2]=resOld[4];
        if (T->lastResiduals[lvl] > 1.5*minResForAbort[lvl]) { ok=0; break; }
        if (levelCutoffRepeat > 1 && !haveRepeated) { lvl++; haveRepeated=1; }
    }
    for (int i=0;i<5;i++) lastResiduals_out[i]=T->lastResiduals[i];
    for (int i=0;i<3;i++) lastFlow_out[i]=T->lastFlow[i];
    if (!ok) return 0;
    memcpy(T_io,cur,sizeof(cur)); aff_io[0]=aff_cur[0]; aff_io[1]=aff_cur[1];
    {   /* :1243-1256 */
        const double mA = T->aff_set ? T->affA : 1e12, mB = T->aff_set ? T->affB : 1e8;
        if ((mA != 0 && fabsf((float)aff_io[0]) > 1.2f) || (mB != 0 && fabsf((float)aff_io[1]) > 200)) return 0;
        double rel[2]; orc_aff_from_to(exposures[0],exposures[1],ref_aff[0],ref_aff[1],aff_io[0],aff_io[1],rel);
        if ((mA == 0 && fabsf(logf((float)rel[0])) > 1.5f) || (mB == 0 && fabsf((float)rel[1]) > 200)) return 0;
        if (mA < 0) aff_io[0]=0;
        if (mB < 0) aff_io[1]=0;
    }
    return 1;
}
void orc_trk_set_affine_modes(OrcTracker* T, double affA, double affB) { T->affA=affA; T->affB=affB; T->aff_set=1; }
long orc_trk_counter(OrcTracker* T, int which) { return which ? T->n_calcgs : T->n_calcres; }
