/* ORACLE — TEST INFRASTRUCTURE ONLY (see orc_common.h). parity unpinned. */
#include "orc_common.h"

const int orc_patternP[8][2] = {{0,-2},{-1,-1},{1,-1},{-2,0},{0,0},{2,0},{-1,1},{0,2}}; /* util/settings.cpp:297 */
int orc_sum_mode = 0;
void orc_set_sum_mode(int m) { orc_sum_mode = m; }

/* ------------------------------------------------------------------------------------------
 * SE3 — thirdparty/Sophus/sophus/se3.hpp:407-428 (exp), :560-586 (log), :131-139 (Adj);
 * so3.hpp:343-369 (expAndTheta, quaternion form), :491-531 (logAndTheta, atan form).
 * ------------------------------------------------------------------------------------------ */
static void quat_to_R(const double q[4], double R[9]) {   /* q = (w,x,y,z) */
    double w=q[0],x=q[1],y=q[2],z=q[3];
    R[0]=1-2*(y*y+z*z); R[1]=2*(x*y-w*z);   R[2]=2*(x*z+w*y);
    R[3]=2*(x*y+w*z);   R[4]=1-2*(x*x+z*z); R[5]=2*(y*z-w*x);
    R[6]=2*(x*z-w*y);   R[7]=2*(y*z+w*x);   R[8]=1-2*(x*x+y*y);
}
static void R_to_quat(const double R[9], double q[4]) {
    double tr = R[0]+R[4]+R[8];
    if (tr > 0) { double s = sqrt(tr+1.0)*2; q[0]=0.25*s; q[1]=(R[7]-R[5])/s; q[2]=(R[2]-R[6])/s; q[3]=(R[3]-R[1])/s; }
    else if (R[0]>R[4] && R[0]>R[8]) { double s=sqrt(1.0+R[0]-R[4]-R[8])*2; q[0]=(R[7]-R[5])/s; q[1]=0.25*s; q[2]=(R[1]+R[3])/s; q[3]=(R[2]+R[6])/s; }
    else if (R[4]>R[8]) { double s=sqrt(1.0+R[4]-R[0]-R[8])*2; q[0]=(R[2]-R[6])/s; q[1]=(R[1]+R[3])/s; q[2]=0.25*s; q[3]=(R[5]+R[7])/s; }
    else { double s=sqrt(1.0+R[8]-R[0]-R[4])*2; q[0]=(R[3]-R[1])/s; q[1]=(R[2]+R[6])/s; q[2]=(R[5]+R[7])/s; q[3]=0.25*s; }
    double n = sqrt(q[0]*q[0]+q[1]*q[1]+q[2]*q[2]+q[3]*q[3]);
    for (int i=0;i<4;i++) q[i]/=n;
}
static void hat(const double w[3], double O[9]) {
    O[0]=0; O[1]=-w[2]; O[2]=w[1]; O[3]=w[2]; O[4]=0; O[5]=-w[0]; O[6]=-w[1]; O[7]=w[0]; O[8]=0;
}
static void mat3mul(const double A[9], const double B[9], double C[9]) {
    for (int i=0;i<3;i++) for (int j=0;j<3;j++) { double s=0; for (int k=0;k<3;k++) s+=A[i*3+k]*B[k*3+j]; C[i*3+j]=s; }
}
void orc_se3_exp(const double xi[6], double T[12]) {
    const double* om = xi+3;
    double theta_sq = om[0]*om[0]+om[1]*om[1]+om[2]*om[2];
    double theta = sqrt(theta_sq), half = 0.5*theta, imag, re;
    if (theta < 1e-10) { double t4=theta_sq*theta_sq; imag = 0.5 - theta_sq/48.0 + t4/3840.0; re = 1 - 0.5*theta_sq + t4/384.0; }
    else { imag = sin(half)/theta; re = cos(half); }
    double q[4] = {re, imag*om[0], imag*om[1], imag*om[2]};
    double n = sqrt(q[0]*q[0]+q[1]*q[1]+q[2]*q[2]+q[3]*q[3]); for (int i=0;i<4;i++) q[i]/=n;
    double R[9]; quat_to_R(q, R);
    double O[9], O2[9], V[9]; hat(om, O); mat3mul(O, O, O2);
    if (theta < 1e-10) { memcpy(V, R, sizeof(V)); }
    else {
        double a = (1-cos(theta))/theta_sq, b = (theta-sin(theta))/(theta_sq*theta);
        for (int i=0;i<9;i++) V[i] = ((i%4==0)?1.0:0.0) + a*O[i] + b*O2[i];
    }
    for (int i=0;i<3;i++) {
        T[i*4+0]=R[i*3+0]; T[i*4+1]=R[i*3+1]; T[i*4+2]=R[i*3+2];
        T[i*4+3]=V[i*3+0]*xi[0]+V[i*3+1]*xi[1]+V[i*3+2]*xi[2];
    }
}
void orc_se3_log(const double T[12], double xi[6]) {
    double R[9] = {T[0],T[1],T[2],T[4],T[5],T[6],T[8],T[9],T[10]}, t[3]={T[3],T[7],T[11]};
    double q[4]; R_to_quat(R, q);
    double sq = q[1]*q[1]+q[2]*q[2]+q[3]*q[3], n = sqrt(sq), w = q[0], f;
    if (n < 1e-10) { f = 2.0/w - 2.0*sq/(w*w*w); }
    else if (fabs(w) < 1e-10) { f = (w>0 ? M_PI : -M_PI)/n; }
    else f = 2.0*atan(n/w)/n;
    double theta = f*n;
    double om[3] = {f*q[1], f*q[2], f*q[3]};
    double O[9], O2[9], Vi[9]; hat(om, O); mat3mul(O,O,O2);
    if (fabs(theta) < 1e-10) { for (int i=0;i<9;i++) Vi[i] = ((i%4==0)?1.0:0.0) - 0.5*O[i] + (1.0/12.0)*O2[i]; }
    else { double c = (1.0 - theta/(2.0*tan(theta/2.0)))/(theta*theta); for (int i=0;i<9;i++) Vi[i] = ((i%4==0)?1.0:0.0) - 0.5*O[i] + c*O2[i]; }
    for (int i=0;i<3;i++) xi[i] = Vi[i*3]*t[0]+Vi[i*3+1]*t[1]+Vi[i*3+2]*t[2];
    xi[3]=om[0]; xi[4]=om[1]; xi[5]=om[2];
}
void orc_se3_mul(const double A[12], const double B[12], double C[12]) {
    double Cc[12];
    for (int i=0;i<3;i++) {
        for (int j=0;j<3;j++) Cc[i*4+j] = A[i*4]*B[j]+A[i*4+1]*B[4+j]+A[i*4+2]*B[8+j];
        Cc[i*4+3] = A[i*4]*B[3]+A[i*4+1]*B[7]+A[i*4+2]*B[11]+A[i*4+3];
    }
    memcpy(C, Cc, sizeof(Cc));
}
void orc_se3_inv(const double A[12], double C[12]) {
    double Cc[12];
    for (int i=0;i<3;i++) for (int j=0;j<3;j++) Cc[i*4+j] = A[j*4+i];
    for (int i=0;i<3;i++) Cc[i*4+3] = -(Cc[i*4]*A[3]+Cc[i*4+1]*A[7]+Cc[i*4+2]*A[11]);
    memcpy(C, Cc, sizeof(Cc));
}
void orc_se3_adj(const double T[12], double Ad[36]) {   /* se3.hpp:131-139: [R, hat(t)R; 0, R] */
    double R[9] = {T[0],T[1],T[2],T[4],T[5],T[6],T[8],T[9],T[10]}, t[3]={T[3],T[7],T[11]}, H[9], HR[9];
    hat(t, H); mat3mul(H, R, HR);
    memset(Ad, 0, 36*sizeof(double));
    for (int i=0;i<3;i++) for (int j=0;j<3;j++) { Ad[i*6+j]=R[i*3+j]; Ad[(i+3)*6+j+3]=R[i*3+j]; Ad[i*6+j+3]=HR[i*3+j]; }
}
/* util/NumType.h:173-185 */
void orc_aff_from_to(float expF, float expT, double aF, double bF, double aT, double bT, double out[2]) {
    if (expF==0 || expT==0) { expT = expF = 1; }
    double a = exp(aT-aF) * expT / expF;
    out[0] = a; out[1] = bT - a*bF;
}

/* ------------------------------------------------------------------------------------------
 * fp64 LDL^T with diagonal pivoting (restates what Eigen's ldlt() provides at
 * CoarseTracker.cpp:1138 and EnergyFunctional.cpp:893: a robust Cholesky for semi-definite A).
 * ------------------------------------------------------------------------------------------ */
int orc_ldlt_solve(int n, const double* Ain, const double* b, double* x) {
    double* A = (double*)malloc(sizeof(double)*n*n); int* perm = (int*)malloc(sizeof(int)*n);
    double* y = (double*)malloc(sizeof(double)*n);
    memcpy(A, Ain, sizeof(double)*n*n);
    for (int i=0;i<n;i++) perm[i]=i;
    for (int k=0;k<n;k++) {
        int p=k; double best=fabs(A[k*n+k]);
        for (int i=k+1;i<n;i++) if (fabs(A[i*n+i])>best) { best=fabs(A[i*n+i]); p=i; }
        if (p!=k) {
            for (int j=0;j<n;j++) { double tmp=A[k*n+j]; A[k*n+j]=A[p*n+j]; A[p*n+j]=tmp; }
            for (int j=0;j<n;j++) { double tmp=A[j*n+k]; A[j*n+k]=A[j*n+p]; A[j*n+p]=tmp; }
            int ti=perm[k]; perm[k]=perm[p]; perm[p]=ti;
        }
        double d = A[k*n+k];
        if (d==0.0 || !isfinite(d)) { for (int i=k+1;i<n;i++) A[i*n+k]=0; continue; }
        for (int i=k+1;i<n;i++) A[i*n+k] /= d;
        for (int i=k+1;i<n;i++) { double lik=A[i*n+k]; if (lik==0) continue;
            for (int j=k+1;j<=i;j++) A[i*n+j] -= lik*d*A[j*n+k]; }
        for (int i=k+1;i<n;i++) for (int j=i+1;j<n;j++) A[i*n+j]=A[j*n+i];
    }
    for (int i=0;i<n;i++) y[i]=b[perm[i]];
    for (int i=0;i<n;i++) for (int j=0;j<i;j++) y[i]-=A[i*n+j]*y[j];
    for (int i=0;i<n;i++) { double d=A[i*n+i]; y[i] = (d!=0.0 && isfinite(d)) ? y[i]/d : 0.0; }
    for (int i=n-1;i>=0;i--) for (int j=i+1;j<n;j++) y[i]-=A[j*n+i]*y[j];
    for (int i=0;i<n;i++) x[perm[i]]=y[i];
    free(A); free(perm); free(y);
    return 0;
}
/* cyclic Jacobi eigen-decomposition of a small symmetric matrix (used for the nullspace projector,
 * EnergyFunctional.cpp:747-761: N (N^T N)^-1 N^T with singular values thresholded at delta*max) */
void orc_sym_eig_jacobi(int n, double* A, double* V, double* w) {
    for (int i=0;i<n;i++) for (int j=0;j<n;j++) V[i*n+j]=(i==j);
    for (int sweep=0; sweep<100; sweep++) {
        double off=0; for (int i=0;i<n;i++) for (int j=i+1;j<n;j++) off+=A[i*n+j]*A[i*n+j];
        if (off < 1e-300) break;
        for (int p=0;p<n;p++) for (int q=p+1;q<n;q++) {
            double apq=A[p*n+q]; if (fabs(apq) < 1e-300) continue;
            double th=(A[q*n+q]-A[p*n+p])/(2*apq);
            double t=(th>=0?1.0:-1.0)/(fabs(th)+sqrt(th*th+1)), c=1/sqrt(t*t+1), s=t*c;
            for (int k=0;k<n;k++) { double akp=A[k*n+p], akq=A[k*n+q]; A[k*n+p]=c*akp-s*akq; A[k*n+q]=s*akp+c*akq; }
            for (int k=0;k<n;k++) { double apk=A[p*n+k], aqk=A[q*n+k]; A[p*n+k]=c*apk-s*aqk; A[q*n+k]=s*apk+c*aqk; }
            for (int k=0;k<n;k++) { double vkp=V[k*n+p], vkq=V[k*n+q]; V[k*n+p]=c*vkp-s*vkq; V[k*n+q]=s*vkp+c*vkq; }
        }
    }
    for (int i=0;i<n;i++) w[i]=A[i*n+i];
}

/* ------------------------------------------------------------------------------------------
 * Tiered accumulators — OptimizationBackend/MatrixAccumulators.h:69-88 (AccumulatorXX::shiftUp),
 * :937-971 (AccumulatorApprox), :1325-1344 (Accumulator9): tier-0 flushed when numIn1 > 1000,
 * tier-1 when numIn1k > 1000, finish() forces both.
 * ------------------------------------------------------------------------------------------ */
void orc_tier_init(OrcTier* t, int n) {
    t->n=n; t->A=(float*)calloc(n,sizeof(float)); t->A1k=(float*)calloc(n,sizeof(float)); t->A1m=(float*)calloc(n,sizeof(float));
    t->D=(double*)calloc(n,sizeof(double)); t->numIn1=t->numIn1k=t->numIn1m=0; t->num=0;
}
void orc_tier_free(OrcTier* t) { free(t->A); free(t->A1k); free(t->A1m); free(t->D); memset(t,0,sizeof(*t)); }
void orc_tier_zero(OrcTier* t) {
    memset(t->A,0,sizeof(float)*t->n); memset(t->A1k,0,sizeof(float)*t->n); memset(t->A1m,0,sizeof(float)*t->n);
    memset(t->D,0,sizeof(double)*t->n); t->numIn1=t->numIn1k=t->numIn1m=0; t->num=0;
}
void orc_tier_shift(OrcTier* t, int force) {
    if (t->numIn1 > 1000 || force) {
        for (int i=0;i<t->n;i++) { t->A1k[i] += t->A[i]; t->A[i]=0; }
        t->numIn1k += t->numIn1; t->numIn1 = 0;
    }
    if (t->numIn1k > 1000 || force) {
        for (int i=0;i<t->n;i++) { t->A1m[i] += t->A1k[i]; t->A1k[i]=0; }
        t->numIn1m += t->numIn1k; t->numIn1k = 0;
    }
}
double orc_tier_get(const OrcTier* t, int i) {
#ifdef ORC_FAST
    return (double)t->A1m[i];
#else
    return orc_sum_mode ? (double)t->A1m[i] : t->D[i];
#endif
}

/* the constants of orc_common.h under the reference's own names (tests/test_constants_cpu.py compares them with tests/golden/constants_ref.json) */
int orc_constants(int cap, const char** names, double* values) {
    static const struct { const char* name; double value; } t[] = {
        {"SCALE_IDEPTH", SCALE_IDEPTH}, {"SCALE_XI_ROT", SCALE_XI_ROT}, {"SCALE_XI_TRANS", SCALE_XI_TRANS}, {"SCALE_F", SCALE_F}, {"SCALE_C", SCALE_C},
        {"SCALE_A", SCALE_A}, {"SCALE_B", SCALE_B}, {"PYR_LEVELS", ORC_PYR_MAX}, {"patternNum", ORC_PATTERN_NUM}, {"patternPadding", ORC_PATTERN_PADDING},
        {"MAX_RES_PER_POINT", ORC_MAX_RES_PER_POINT}, {"NUM_THREADS", ORC_NUM_THREADS}, {"CPARS", ORC_CPARS},
        {"setting_huberTH", SETTING_HUBER_TH}, {"setting_coarseCutoffTH", SETTING_COARSE_CUTOFF_TH}, {"setting_outlierTH", SETTING_OUTLIER_TH},
        {"setting_outlierTHSumComponent", SETTING_OUTLIER_TH_SUMCOMP}, {"setting_idepthFixPrior", SETTING_IDEPTH_FIX_PRIOR},
        {"setting_idepthFixPriorMargFac", SETTING_IDEPTH_FIX_PRIOR_MARGFAC}, {"setting_initialRotPrior", SETTING_INITIAL_ROT_PRIOR},
        {"setting_initialTransPrior", SETTING_INITIAL_TRANS_PRIOR}, {"setting_initialAffBPrior", SETTING_INITIAL_AFFB_PRIOR},
        {"setting_initialAffAPrior", SETTING_INITIAL_AFFA_PRIOR}, {"setting_initialCalibHessian", SETTING_INITIAL_CALIB_HESSIAN},
        {"setting_solverModeDelta", SETTING_SOLVER_MODE_DELTA}, {"setting_affineOptModeA", SETTING_AFFINE_OPT_MODE_A}, {"setting_affineOptModeB", SETTING_AFFINE_OPT_MODE_B},
        {"setting_margWeightFac", SETTING_MARG_WEIGHT_FAC}, {"setting_frameEnergyTHConstWeight", SETTING_FRAME_ENERGY_TH_CONST_WEIGHT},
        {"setting_frameEnergyTHN", SETTING_FRAME_ENERGY_TH_N}, {"setting_frameEnergyTHFacMedian", SETTING_FRAME_ENERGY_TH_FAC_MEDIAN},
        {"setting_overallEnergyTHWeight", SETTING_OVERALL_ENERGY_TH_WEIGHT}, {"setting_thOptIterations", SETTING_TH_OPT_ITERATIONS},
        {"setting_minIdepthH_act", SETTING_MIN_IDEPTH_H_ACT}, {"setting_minGradHistCut", SETTING_MIN_GRAD_HIST_CUT}, {"setting_minGradHistAdd", SETTING_MIN_GRAD_HIST_ADD},
        {"setting_gradDownweightPerLevel", SETTING_GRAD_DOWNWEIGHT_PER_LEVEL}, {"setting_maxPixSearch", SETTING_MAX_PIX_SEARCH},
        {"setting_minTraceTestRadius", SETTING_MIN_TRACE_TEST_RADIUS}, {"setting_GNItsOnPointActivation", SETTING_GN_ITS_ON_POINT_ACTIVATION},
        {"setting_trace_stepsize", SETTING_TRACE_STEPSIZE}, {"setting_trace_GNIterations", SETTING_TRACE_GN_ITERATIONS},
        {"setting_trace_GNThreshold", SETTING_TRACE_GN_THRESHOLD}, {"setting_trace_extraSlackOnTH", SETTING_TRACE_EXTRA_SLACK_ON_TH},
        {"setting_trace_slackInterval", SETTING_TRACE_SLACK_INTERVAL}, {"setting_trace_minImprovementFactor", SETTING_TRACE_MIN_IMPROVEMENT_FACTOR},
        {"frameEnergyTH_init", FRAME_ENERGY_TH_INIT},
        {"patternP[0].x", 0}, {"patternP[0].y", 0}, {"patternP[1].x", 0}, {"patternP[1].y", 0}, {"patternP[2].x", 0}, {"patternP[2].y", 0}, {"patternP[3].x", 0}, {"patternP[3].y", 0},
        {"patternP[4].x", 0}, {"patternP[4].y", 0}, {"patternP[5].x", 0}, {"patternP[5].y", 0}, {"patternP[6].x", 0}, {"patternP[6].y", 0}, {"patternP[7].x", 0}, {"patternP[7].y", 0},
    };
    const int n = (int)(sizeof(t) / sizeof(t[0]));
    for (int i = 0; i < n && i < cap; ++i) {
        if (names) names[i] = t[i].name;
        if (values) values[i] = i >= n - 16 ? (double)orc_patternP[(i - (n - 16)) / 2][(i - (n - 16)) % 2] : t[i].value;
    }
    return n;
}
