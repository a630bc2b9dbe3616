/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  *** parity unpinned *** (see orc_common.h), except orc_kdtree_knn, which IS pinned: tests/test_oracle_cpu.py checks it
 * against the reference's own nanoflann.h compiled into oracle/_ref/libref_nanoflann.so (oracle/ref_nanoflann.cpp).
 *
 * SURVEY 8(f) rank 2, the rest of the two-frame initialiser around calcResAndGS / doStep (orc_init.c):
 *   orc_grid_max_selection / orc_make_pixel_status   gridMaxSelection<pot>, makePixelStatus      FullSystem/PixelSelector.h:38-117, 199-253
 *   orc_kdtree_knn                                    nanoflann KDTreeSingleIndexAdaptor (leaf 5, L2_Simple, float) build + KNN search,
 *                                                     util/nanoflann.h:876-884, 1056-1195 (divideTree, middleSplit_, planeSplit), :1198-1270 (search), :92-151 (result set)
 *   orc_initf_set_first                               CoarseInitializer::setFirst                 FullSystem/CoarseInitializer.cpp:785-880 (+ makeK :958-988, makeNN :992-1069)
 *   orc_initf_track_frame                             CoarseInitializer::trackFrame               :81-285
 *     with resetPoints :882-909, applyStep :939-956, calcEC :634-655, optReg :656-691, propagateUp :695-734, propagateDown :736-766
 * All of these are sequential by construction (Gauss-Seidel sweeps, index-ordered sums, the k-d tree's traversal order decides ties between
 * equidistant grid points), so the restatement keeps the reference's loop order statement for statement.
 */
#include "orc_common.h"
#include <float.h>

void orc_init_calc_res_and_gs(const float* colorRef, const float* colorNew, int wl, int hl, const float K4[4], const float RKi_[9], const float t_[3], const float aff2[2],
                              const float tlog3[3], double t_sqnorm, float alphaW, float alphaK, float couplingWeight,
                              int n, const float* u, const float* v, const float* idepth_new, const float* iR, const uint8_t* isGood, const float* energy, const float* outlierTH,
                              uint8_t* isGood_new, float* energy_new, float* maxstep_out, float* lastHessian_new, float* Jb,
                              double* H_out, double* b_out, double* Hsc_out, double* bsc_out, double* E3);
void orc_init_do_step(int n, const uint8_t* isGood, const float* Jb, const float* maxstep, const float* idepth, float lambda, const float inc[8], float* idepth_new);

/* ------------------------------------------------------------------------------------------------ gridMaxSelection / makePixelStatus */
/* grads: [w*h][3] = {I, dx, dy}. map_out: w*h bytes. PixelSelector.h:38-117 (the template and the runtime-pot version are the same loop). */
int orc_grid_max_selection(const float* grads, uint8_t* map_out, int w, int h, int pot, float THFac) {
    memset(map_out, 0, (size_t)w * h);
    int numGood = 0;
    for (int y = 1; y < h - pot; y += pot)
        for (int x = 1; x < w - pot; x += pot) {
            int bestXXID = -1, bestYYID = -1, bestXYID = -1, bestYXID = -1;
            float bestXX = 0, bestYY = 0, bestXY = 0, bestYX = 0;
            const float* grads0 = grads + 3 * (x + y * w);
            for (int dx = 0; dx < pot; dx++)
                for (int dy = 0; dy < pot; dy++) {
                    const int idx = dx + dy * w;
                    const float g1 = grads0[3 * idx + 1], g2 = grads0[3 * idx + 2];
                    const float sqgd = g1 * g1 + g2 * g2;
                    const float TH = THFac * 10.0f * (0.75f);                                    /* minUseGrad_pixsel = 10 (:34) */
                    if (sqgd > TH * TH) {
                        const float agx = fabsf(g1); if (agx > bestXX) { bestXX = agx; bestXXID = idx; }
                        const float agy = fabsf(g2); if (agy > bestYY) { bestYY = agy; bestYYID = idx; }
                        const float gxpy = fabsf(g1 - g2); if (gxpy > bestXY) { bestXY = gxpy; bestXYID = idx; }
                        const float gxmy = fabsf(g1 + g2); if (gxmy > bestYX) { bestYX = gxmy; bestYXID = idx; }
                    }
                }
            uint8_t* map0 = map_out + x + y * w;
            const int ids[4] = {bestXXID, bestYYID, bestXYID, bestYXID};
            for (int k = 0; k < 4; k++) if (ids[k] >= 0) { if (!map0[ids[k]]) numGood++; map0[ids[k]] = 1; }
        }
    return numGood;
}

/* PixelSelector.h:199-253. *sparsityFactor is the reference's GLOBAL (util/settings.cpp:223, initial value 5): it carries over from call to call. */
int orc_make_pixel_status(const float* grads, uint8_t* map, int w, int h, float desiredDensity, int recsLeft, float THFac, int* sparsityFactor) {
    for (;;) {
        if (*sparsityFactor < 1) *sparsityFactor = 1;
        const int numGoodPoints = orc_grid_max_selection(grads, map, w, h, *sparsityFactor, THFac);
        const float quotia = numGoodPoints / (float)(desiredDensity);
        int newSparsity = (int)((*sparsityFactor * sqrtf(quotia)) + 0.7f);
        if (newSparsity < 1) newSparsity = 1;
        const float oldTHFac = THFac;
        if (newSparsity == 1 && *sparsityFactor == 1) THFac = 0.5;
        if ((abs(newSparsity - *sparsityFactor) < 1 && THFac == oldTHFac) || (quotia > 0.8 && 1.0f / quotia > 0.8) || recsLeft == 0) {
            *sparsityFactor = newSparsity;
            return numGoodPoints;
        }
        *sparsityFactor = newSparsity;
        recsLeft--;
    }
}

/* ------------------------------------------------------------------------------------------------ k-d tree (nanoflann restated) */
typedef struct { int child1, child2; int left, right; int divfeat; float divlow, divhigh; } KdNode;
typedef struct {
    int n; const float* pt[2]; int* vind; KdNode* nodes; int n_nodes, cap; float root_lo[2], root_hi[2]; int leaf_max;
    /* search state */
    int k, count; int* ri; float* rd;
} Kd;

static inline float kd_get(const Kd* T, int idx, int dim) { return T->pt[dim][idx]; }

static void kd_min_max(const Kd* T, const int* ind, int count, int element, float* mn, float* mx) {   /* computeMinMax :1107 */
    *mn = *mx = kd_get(T, ind[0], element);
    for (int i = 1; i < count; ++i) { const float val = kd_get(T, ind[i], element); if (val < *mn) *mn = val; if (val > *mx) *mx = val; }
}

static void kd_plane_split(const Kd* T, int* ind, int count, int cutfeat, float cutval, int* lim1, int* lim2) {   /* planeSplit :1169-1195 (IndexType is unsigned there) */
    int left = 0, right = count - 1;
    for (;;) {
        while (left <= right && kd_get(T, ind[left], cutfeat) < cutval) ++left;
        while (right && left <= right && kd_get(T, ind[right], cutfeat) >= cutval) --right;
        if (left > right || !right) break;
        { const int tmp = ind[left]; ind[left] = ind[right]; ind[right] = tmp; }
        ++left; --right;
    }
    *lim1 = left;
    right = count - 1;
    for (;;) {
        while (left <= right && kd_get(T, ind[left], cutfeat) <= cutval) ++left;
        while (right && left <= right && kd_get(T, ind[right], cutfeat) > cutval) --right;
        if (left > right || !right) break;
        { const int tmp = ind[left]; ind[left] = ind[right]; ind[right] = tmp; }
        ++left; --right;
    }
    *lim2 = left;
}

static void kd_middle_split(const Kd* T, int* ind, int count, int* index, int* cutfeat, float* cutval, const float lo[2], const float hi[2]) {   /* middleSplit_ :1118-1158 */
    const float EPS = 0.00001f;
    float max_span = hi[0] - lo[0];
    for (int i = 1; i < 2; ++i) { const float span = hi[i] - lo[i]; if (span > max_span) max_span = span; }
    float max_spread = -1;
    *cutfeat = 0;
    for (int i = 0; i < 2; ++i) {
        const float span = hi[i] - lo[i];
        if (span > (1 - EPS) * max_span) {
            float mn, mx;
            kd_min_max(T, ind, count, *cutfeat, &mn, &mx);               /* sic: the reference passes cutfeat, not i (:1133) */
            const float spread = mx - mn;
            if (spread > max_spread) { *cutfeat = i; max_spread = spread; }
        }
    }
    const float split_val = (lo[*cutfeat] + hi[*cutfeat]) / 2;
    float mn, mx;
    kd_min_max(T, ind, count, *cutfeat, &mn, &mx);
    if (split_val < mn) *cutval = mn; else if (split_val > mx) *cutval = mx; else *cutval = split_val;
    int lim1, lim2;
    kd_plane_split(T, ind, count, *cutfeat, *cutval, &lim1, &lim2);
    if (lim1 > count / 2) *index = lim1; else if (lim2 < count / 2) *index = lim2; else *index = count / 2;
}

static int kd_divide(Kd* T, int left, int right, float lo[2], float hi[2]) {      /* divideTree :1056-1104; lo/hi = bbox, in/out */
    if (T->n_nodes == T->cap) { T->cap *= 2; T->nodes = (KdNode*)realloc(T->nodes, sizeof(KdNode) * T->cap); }
    const int me = T->n_nodes++;
    if ((right - left) <= T->leaf_max) {
        T->nodes[me].child1 = T->nodes[me].child2 = -1; T->nodes[me].left = left; T->nodes[me].right = right;
        for (int i = 0; i < 2; ++i) lo[i] = hi[i] = kd_get(T, T->vind[left], i);
        for (int k = left + 1; k < right; ++k)
            for (int i = 0; i < 2; ++i) { const float x = kd_get(T, T->vind[k], i); if (lo[i] > x) lo[i] = x; if (hi[i] < x) hi[i] = x; }
    } else {
        int idx, cutfeat; float cutval;
        kd_middle_split(T, T->vind + left, right - left, &idx, &cutfeat, &cutval, lo, hi);
        float llo[2] = {lo[0], lo[1]}, lhi[2] = {hi[0], hi[1]}, rlo[2] = {lo[0], lo[1]}, rhi[2] = {hi[0], hi[1]};
        lhi[cutfeat] = cutval;
        const int c1 = kd_divide(T, left, left + idx, llo, lhi);
        rlo[cutfeat] = cutval;
        const int c2 = kd_divide(T, left + idx, right, rlo, rhi);
        KdNode* nd = T->nodes + me;                                         /* (re-read: the array may have moved) */
        nd->child1 = c1; nd->child2 = c2; nd->divfeat = cutfeat; nd->divlow = lhi[cutfeat]; nd->divhigh = rlo[cutfeat];
        for (int i = 0; i < 2; ++i) { lo[i] = llo[i] < rlo[i] ? llo[i] : rlo[i]; hi[i] = lhi[i] > rhi[i] ? lhi[i] : rhi[i]; }
    }
    return me;
}

static void kd_build(Kd* T, int n, const float* u, const float* v) {
    T->n = n; T->pt[0] = u; T->pt[1] = v; T->leaf_max = 5;
    T->vind = (int*)malloc(sizeof(int) * (n > 0 ? n : 1));
    for (int i = 0; i < n; i++) T->vind[i] = i;
    T->cap = 64; T->n_nodes = 0; T->nodes = (KdNode*)malloc(sizeof(KdNode) * T->cap);
    if (n == 0) return;
    for (int i = 0; i < 2; ++i) T->root_lo[i] = T->root_hi[i] = kd_get(T, 0, i);                 /* computeBoundingBox :1024-1046 */
    for (int k = 1; k < n; ++k)
        for (int i = 0; i < 2; ++i) { const float x = kd_get(T, k, i); if (x < T->root_lo[i]) T->root_lo[i] = x; if (x > T->root_hi[i]) T->root_hi[i] = x; }
    kd_divide(T, 0, n, T->root_lo, T->root_hi);                                                     /* root_bbox is passed by reference and tightened */
}
static void kd_free(Kd* T) { free(T->vind); free(T->nodes); }

static inline float kd_worst(const Kd* T) { return T->rd[T->k - 1]; }
static void kd_add(Kd* T, float dist, int index) {                                                 /* KNNResultSet::addPoint :124-145 (NANOFLANN_FIRST_MATCH undefined) */
    int i;
    for (i = T->count; i > 0; --i) {
        if (T->rd[i - 1] > dist) { if (i < T->k) { T->rd[i] = T->rd[i - 1]; T->ri[i] = T->ri[i - 1]; } }
        else break;
    }
    if (i < T->k) { T->rd[i] = dist; T->ri[i] = index; }
    if (T->count < T->k) T->count++;
}
static void kd_search(Kd* T, const float vec[2], int node, float mindistsq, float dists[2], const float epsError) {   /* searchLevel :1222-1270 */
    const KdNode* nd = T->nodes + node;
    if (nd->child1 < 0 && nd->child2 < 0) {
        const float worst_dist = kd_worst(T);
        for (int i = nd->left; i < nd->right; ++i) {
            const int index = T->vind[i];
            const float d0 = vec[0] - T->pt[0][index], d1 = vec[1] - T->pt[1][index];
            const float dist = d0 * d0 + d1 * d1;
            if (dist < worst_dist) kd_add(T, dist, index);
        }
        return;
    }
    const int idx = nd->divfeat;
    const float val = vec[idx], diff1 = val - nd->divlow, diff2 = val - nd->divhigh;
    int best, other; float cut_dist;
    if ((diff1 + diff2) < 0) { best = nd->child1; other = nd->child2; cut_dist = (val - nd->divhigh) * (val - nd->divhigh); }
    else { best = nd->child2; other = nd->child1; cut_dist = (val - nd->divlow) * (val - nd->divlow); }
    kd_search(T, vec, best, mindistsq, dists, epsError);
    const float dst = dists[idx];
    mindistsq = mindistsq + cut_dist - dst;
    dists[idx] = cut_dist;
    if (mindistsq * epsError <= kd_worst(T)) kd_search(T, vec, other, mindistsq, dists, epsError);
    dists[idx] = dst;
}
static void kd_knn(Kd* T, const float vec[2], int k, int* ri, float* rd) {                          /* findNeighbors :919-933 + KNNResultSet::init :104-111 */
    T->k = k; T->count = 0; T->ri = ri; T->rd = rd;
    rd[k - 1] = FLT_MAX;
    if (T->n == 0) return;
    float dists[2] = {0, 0}, distsq = 0;
    for (int i = 0; i < 2; ++i) {                                                                    /* computeInitialDistances :1198-1216 */
        if (vec[i] < T->root_lo[i]) { dists[i] = (vec[i] - T->root_lo[i]) * (vec[i] - T->root_lo[i]); distsq += dists[i]; }
        if (vec[i] > T->root_hi[i]) { dists[i] = (vec[i] - T->root_hi[i]) * (vec[i] - T->root_hi[i]); distsq += dists[i]; }
    }
    kd_search(T, vec, 0, distsq, dists, 1.0f);
}

/* same contract as ref_nanoflann_knn (oracle/ref_nanoflann.cpp) */
int orc_kdtree_knn(int n, const float* u, const float* v, int nq, const float* qu, const float* qv, int k, int* idx_out, float* dist_out) {
    if (n <= 0 || k <= 0) return -1;
    Kd T; kd_build(&T, n, u, v);
    for (int i = 0; i < nq; ++i) {
        int* ri = idx_out + (size_t)i * k; float* rd = dist_out + (size_t)i * k;
        for (int j = 0; j < k; ++j) { ri[j] = -1; rd[j] = FLT_MAX; }
        const float pt[2] = {qu[i], qv[i]};
        kd_knn(&T, pt, k, ri, rd);
    }
    kd_free(&T);
    return 0;
}

/* ------------------------------------------------------------------------------------------------ the initialiser object */
typedef struct {
    int n;
    float *u, *v, *idepth, *idepth_new, *iR, *iRSumNum, *lastHessian, *lastHessian_new, *maxstep, *energy, *energy_new, *outlierTH, *my_type, *neighboursDist, *parentDist;
    uint8_t *isGood, *isGood_new;
    int *parent, *neighbours;
} OrcInitLvl;
typedef struct {
    int levels, w[ORC_PYR_MAX], h[ORC_PYR_MAX];
    float fx[ORC_PYR_MAX], fy[ORC_PYR_MAX], cx[ORC_PYR_MAX], cy[ORC_PYR_MAX];
    OrcInitLvl L[ORC_PYR_MAX];
    float *Jb, *Jb_new;
    float* first[ORC_PYR_MAX];            /* firstFrame->dIp[lvl], [wl*hl][3] */
    double thisToNext[12], aff[2];
    int snapped, frameID, snappedAt, fixAffine;
    float alphaK, alphaW, regWeight, couplingWeight;
    int n_evals;
} OrcInit;

static void lvl_free(OrcInitLvl* l) {
    free(l->u); free(l->v); free(l->idepth); free(l->idepth_new); free(l->iR); free(l->iRSumNum); free(l->lastHessian); free(l->lastHessian_new); free(l->maxstep);
    free(l->energy); free(l->energy_new); free(l->outlierTH); free(l->my_type); free(l->neighboursDist); free(l->parentDist); free(l->isGood); free(l->isGood_new);
    free(l->parent); free(l->neighbours); memset(l, 0, sizeof(*l));
}
static void lvl_alloc(OrcInitLvl* l, int n) {
    const size_t N = n > 0 ? n : 1;
#define FA(x, m) l->x = (float*)calloc(N * (m), sizeof(float))
    FA(u, 1); FA(v, 1); FA(idepth, 1); FA(idepth_new, 1); FA(iR, 1); FA(iRSumNum, 1); FA(lastHessian, 1); FA(lastHessian_new, 1); FA(maxstep, 1);
    FA(energy, 2); FA(energy_new, 2); FA(outlierTH, 1); FA(my_type, 1); FA(neighboursDist, 10); FA(parentDist, 1);
#undef FA
    l->isGood = (uint8_t*)calloc(N, 1); l->isGood_new = (uint8_t*)calloc(N, 1);
    l->parent = (int*)calloc(N, sizeof(int)); l->neighbours = (int*)calloc(N * 10, sizeof(int));
    l->n = n;
}

void* orc_initf_create(int w, int h, int levels, float fx, float fy, float cx, float cy) {        /* ctor :47-68 + makeK :958-988 */
    OrcInit* I = (OrcInit*)calloc(1, sizeof(OrcInit));
    I->levels = levels;
    I->w[0] = w; I->h[0] = h; I->fx[0] = fx; I->fy[0] = fy; I->cx[0] = cx; I->cy[0] = cy;
    for (int l = 1; l < levels; ++l) {
        I->w[l] = w >> l; I->h[l] = h >> l;
        I->fx[l] = I->fx[l - 1] * 0.5; I->fy[l] = I->fy[l - 1] * 0.5;                             /* double product, stored to float (:972-973) */
        I->cx[l] = (I->cx[0] + 0.5) / ((int)1 << l) - 0.5; I->cy[l] = (I->cy[0] + 0.5) / ((int)1 << l) - 0.5;
    }
    I->Jb = (float*)calloc((size_t)w * h * 10, sizeof(float)); I->Jb_new = (float*)calloc((size_t)w * h * 10, sizeof(float));
    const double id[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    memcpy(I->thisToNext, id, sizeof(id));
    I->frameID = -1; I->fixAffine = 1;
    return I;
}
void orc_initf_destroy(void* p) {
    OrcInit* I = (OrcInit*)p; if (!I) return;
    for (int l = 0; l < ORC_PYR_MAX; ++l) { lvl_free(&I->L[l]); free(I->first[l]); }
    free(I->Jb); free(I->Jb_new); free(I);
}

static void initf_make_nn(OrcInit* I) {                                                              /* makeNN :992-1069 */
    const float NNDistFactor = 0.05f;
    Kd T[ORC_PYR_MAX];
    for (int l = 0; l < I->levels; ++l) kd_build(&T[l], I->L[l].n, I->L[l].u, I->L[l].v);
    const int nn = 10;
    for (int lvl = 0; lvl < I->levels; ++lvl) {
        OrcInitLvl* P = &I->L[lvl];
        int ret_index[10]; float ret_dist[10];
        for (int k = 0; k < nn; ++k) { ret_index[k] = -1; ret_dist[k] = FLT_MAX; }               /* (uninitialised stack in the reference; only matters for < 10 points) */
        for (int i = 0; i < P->n; ++i) {
            float pt[2] = {P->u[i], P->v[i]};
            kd_knn(&T[lvl], pt, nn, ret_index, ret_dist);
            float sumDF = 0;
            for (int k = 0; k < nn; ++k) {
                P->neighbours[i * 10 + k] = ret_index[k];
                const float df = expf(-ret_dist[k] * NNDistFactor);
                sumDF += df;
                P->neighboursDist[i * 10 + k] = df;
            }
            for (int k = 0; k < nn; ++k) P->neighboursDist[i * 10 + k] *= 10 / sumDF;
            if (lvl < I->levels - 1) {
                pt[0] = pt[0] * 0.5f - 0.25f; pt[1] = pt[1] * 0.5f - 0.25f;
                kd_knn(&T[lvl + 1], pt, 1, ret_index, ret_dist);
                P->parent[i] = ret_index[0];
                P->parentDist[i] = expf(-ret_dist[0] * NNDistFactor);
            } else { P->parent[i] = -1; P->parentDist[i] = -1; }
        }
    }
    for (int l = 0; l < I->levels; ++l) kd_free(&T[l]);
}

/* setFirst :785-880. dI[l] = the first frame's {I,dx,dy} image of level l; statusMap0 = PixelSelector::makeMaps(firstFrame, ., 0.03*w*h, 1, false, 2) output for
 * level 0 (a fresh selector with currentPotential = 3; orc_pixsel_make_maps); *sparsityFactor = the global makePixelStatus keeps adapting (in/out). */
void orc_initf_set_first(void* p, const float* const* dI, const float* statusMap0, int* sparsityFactor) {
    OrcInit* I = (OrcInit*)p;
    const float densities[] = {0.03f, 0.05f, 0.15f, 0.5f, 1};
    const int pad = 2;                                                                               /* patternPadding, util/settings.h:234 */
    uint8_t* mapB = (uint8_t*)malloc((size_t)I->w[0] * I->h[0]);
    for (int lvl = 0; lvl < I->levels; ++lvl) {
        const int wl = I->w[lvl], hl = I->h[lvl];
        free(I->first[lvl]); I->first[lvl] = (float*)malloc(sizeof(float) * 3 * wl * hl); memcpy(I->first[lvl], dI[lvl], sizeof(float) * 3 * wl * hl);
        int npts = 0;
        if (lvl != 0) npts = orc_make_pixel_status(dI[lvl], mapB, wl, hl, densities[lvl] * I->w[0] * I->h[0], 5, 1, sparsityFactor);
        (void)npts;
        int nl = 0;
        for (int pass = 0; pass < 2; ++pass) {                                                      /* count, then fill (the reference sizes by npts, an upper bound) */
            if (pass == 1) { lvl_free(&I->L[lvl]); lvl_alloc(&I->L[lvl], nl); nl = 0; }
            OrcInitLvl* P = &I->L[lvl];
            for (int y = pad + 1; y < hl - pad - 2; y++)
                for (int x = pad + 1; x < wl - pad - 2; x++) {
                    if ((lvl != 0 && mapB[x + y * wl]) || (lvl == 0 && statusMap0[x + y * wl] != 0)) {
                        if (pass == 1) {
                            P->u[nl] = x + 0.1; P->v[nl] = y + 0.1; P->idepth[nl] = 1; P->iR[nl] = 1; P->isGood[nl] = 1;
                            P->energy[2 * nl] = P->energy[2 * nl + 1] = 0; P->lastHessian[nl] = 0; P->lastHessian_new[nl] = 0;
                            P->my_type[nl] = (lvl != 0) ? 1 : statusMap0[x + y * wl];
                            P->outlierTH[nl] = ORC_PATTERN_NUM * SETTING_OUTLIER_TH;                                  /* patternNum*setting_outlierTH, settings.cpp:99 */
                        }
                        nl++;
                    }
                }
        }
    }
    free(mapB);
    initf_make_nn(I);
    const double id[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    memcpy(I->thisToNext, id, sizeof(id));
    I->snapped = 0; I->frameID = I->snappedAt = 0;
}

static void initf_opt_reg(OrcInit* I, int lvl) {                                                     /* optReg :656-691 */
    OrcInitLvl* P = &I->L[lvl];
    if (!I->snapped) { for (int i = 0; i < P->n; i++) P->iR[i] = 1; return; }
    for (int i = 0; i < P->n; i++) {
        if (!P->isGood[i]) continue;
        float idnn[10]; int nnn = 0;
        for (int j = 0; j < 10; j++) {
            const int o = P->neighbours[i * 10 + j];
            if (o == -1) continue;
            if (!P->isGood[o]) continue;
            idnn[nnn++] = P->iR[o];
        }
        if (nnn > 2) {
            /* std::nth_element(idnn, idnn+nnn/2, idnn+nnn): only the VALUE at position nnn/2 is used = the (nnn/2)-th order statistic */
            for (int a = 1; a < nnn; a++) { const float x = idnn[a]; int b = a - 1; while (b >= 0 && idnn[b] > x) { idnn[b + 1] = idnn[b]; b--; } idnn[b + 1] = x; }
            P->iR[i] = (1 - I->regWeight) * P->idepth[i] + I->regWeight * idnn[nnn / 2];
        }
    }
}
static void initf_propagate_up(OrcInit* I, int srcLvl) {                                            /* propagateUp :695-734 */
    OrcInitLvl *S = &I->L[srcLvl], *T = &I->L[srcLvl + 1];
    for (int i = 0; i < T->n; i++) { T->iR[i] = 0; T->iRSumNum[i] = 0; }
    for (int i = 0; i < S->n; i++) {
        if (!S->isGood[i]) continue;
        const int par = S->parent[i];
        T->iR[par] += S->iR[i] * S->lastHessian[i];
        T->iRSumNum[par] += S->lastHessian[i];
    }
    for (int i = 0; i < T->n; i++)
        if (T->iRSumNum[i] > 0) { T->idepth[i] = T->iR[i] = (T->iR[i] / T->iRSumNum[i]); T->isGood[i] = 1; }
    initf_opt_reg(I, srcLvl + 1);
}
static void initf_propagate_down(OrcInit* I, int srcLvl) {                                          /* propagateDown :736-766 */
    OrcInitLvl *S = &I->L[srcLvl], *T = &I->L[srcLvl - 1];
    for (int i = 0; i < T->n; i++) {
        const int par = T->parent[i];
        if (!S->isGood[par] || S->lastHessian[par] < 0.1) continue;
        if (!T->isGood[i]) { T->iR[i] = T->idepth[i] = T->idepth_new[i] = S->iR[par]; T->isGood[i] = 1; T->lastHessian[i] = 0; }
        else {
            const float newiR = (T->iR[i] * T->lastHessian[i] * 2 + S->iR[par] * S->lastHessian[par]) / (T->lastHessian[i] * 2 + S->lastHessian[par]);
            T->iR[i] = T->idepth[i] = T->idepth_new[i] = newiR;
        }
    }
    initf_opt_reg(I, srcLvl - 1);
}
static void initf_reset_points(OrcInit* I, int lvl) {                                               /* resetPoints :882-909 */
    OrcInitLvl* P = &I->L[lvl];
    for (int i = 0; i < P->n; i++) {
        P->energy[2 * i] = P->energy[2 * i + 1] = 0;
        P->idepth_new[i] = P->idepth[i];
        if (lvl == I->levels - 1 && !P->isGood[i]) {
            float snd = 0, sn = 0;
            for (int n = 0; n < 10; n++) {
                const int o = P->neighbours[i * 10 + n];
                if (o == -1 || !P->isGood[o]) continue;
                snd += P->iR[o]; sn += 1;
            }
            if (sn > 0) { P->isGood[i] = 1; P->iR[i] = P->idepth[i] = P->idepth_new[i] = snd / sn; }
        }
    }
}
static void initf_apply_step(OrcInit* I, int lvl) {                                                 /* applyStep :939-956 */
    OrcInitLvl* P = &I->L[lvl];
    for (int i = 0; i < P->n; i++) {
        if (!P->isGood[i]) { P->idepth[i] = P->idepth_new[i] = P->iR[i]; continue; }
        P->energy[2 * i] = P->energy_new[2 * i]; P->energy[2 * i + 1] = P->energy_new[2 * i + 1];
        P->isGood[i] = P->isGood_new[i];
        P->idepth[i] = P->idepth_new[i];
        P->lastHessian[i] = P->lastHessian_new[i];
    }
    float* t = I->Jb; I->Jb = I->Jb_new; I->Jb_new = t;
}
static void initf_calc_ec(OrcInit* I, int lvl, float out[3]) {                                      /* calcEC :634-655 (AccumulatorX<2>: fp32 products, summed in fp64 here) */
    OrcInitLvl* P = &I->L[lvl];
    if (!I->snapped) { out[0] = 0; out[1] = 0; out[2] = P->n; return; }
    double e0 = 0, e1 = 0; int num = 0;
    for (int i = 0; i < P->n; i++) {
        if (!P->isGood_new[i]) continue;
        const float rOld = (P->idepth[i] - P->iR[i]), rNew = (P->idepth_new[i] - P->iR[i]);
        e0 += (double)(float)(rOld * rOld); e1 += (double)(float)(rNew * rNew); num++;
    }
    out[0] = I->couplingWeight * (float)e0; out[1] = I->couplingWeight * (float)e1; out[2] = num;
}

static void initf_calc(OrcInit* I, int lvl, const float* colorNew, const double T[12], const double aff[2], double H[64], double b[8], double Hsc[64], double bsc[8], float res[3]) {
    OrcInitLvl* P = &I->L[lvl];
    /* RKi = (R * Ki[lvl]).cast<float>(), Ki = K^-1 in double (CoarseInitializer.h:103-112: K, Ki, fx.. are doubles; the halvings of makeK are exact, so the
     * float copies kept here hold the same values) (:347-349) */
    const double fx = I->fx[lvl], fy = I->fy[lvl], cx = I->cx[lvl], cy = I->cy[lvl];
    const double Kif[9] = {1.0 / fx, 0, -cx / fx, 0, 1.0 / fy, -cy / fy, 0, 0, 1};
    float RKi[9], t[3];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) RKi[i * 3 + j] = (float)(T[i * 4 + 0] * Kif[j] + T[i * 4 + 1] * Kif[3 + j] + T[i * 4 + 2] * Kif[6 + j]);
        t[i] = (float)T[i * 4 + 3];
    }
    const float aff2[2] = {(float)exp(aff[0]), (float)aff[1]};
    double xi[6]; orc_se3_log(T, xi);
    const float tlog3[3] = {(float)xi[0], (float)xi[1], (float)xi[2]};
    const double tsq = T[3] * T[3] + T[7] * T[7] + T[11] * T[11];
    const float K4[4] = {I->fx[lvl], I->fy[lvl], I->cx[lvl], I->cy[lvl]};
    double E3[3];
    orc_init_calc_res_and_gs(I->first[lvl], colorNew, I->w[lvl], I->h[lvl], K4, RKi, t, aff2, tlog3, tsq, I->alphaW, I->alphaK, I->couplingWeight,
                             P->n, P->u, P->v, P->idepth_new, P->iR, P->isGood, P->energy, P->outlierTH, P->isGood_new, P->energy_new, P->maxstep, P->lastHessian_new, I->Jb_new,
                             H, b, Hsc, bsc, E3);
    res[0] = (float)E3[0]; res[1] = (float)E3[1]; res[2] = (float)E3[2];
    I->n_evals++;
}

/* Hl.ldlt().solve(bl) in `real` (Eigen float LDLT: symmetric pivoting on the largest remaining diagonal entry) */
static void ldlt_solve_real(int n, const real* Ain, const real* bin, real* x) {
    real A[64], bb[8]; int perm[8];
    for (int i = 0; i < n * n; i++) A[i] = Ain[i];
    for (int i = 0; i < n; i++) { bb[i] = bin[i]; perm[i] = i; }
    for (int k = 0; k < n; k++) {
        int piv = k; real best = (real)fabs((double)A[k * n + k]);
        for (int i = k + 1; i < n; i++) { const real a = (real)fabs((double)A[i * n + i]); if (a > best) { best = a; piv = i; } }
        if (piv != k) {
            for (int j = 0; j < n; j++) { const real tt = A[k * n + j]; A[k * n + j] = A[piv * n + j]; A[piv * n + j] = tt; }
            for (int j = 0; j < n; j++) { const real tt = A[j * n + k]; A[j * n + k] = A[j * n + piv]; A[j * n + piv] = tt; }
            { const int tt = perm[k]; perm[k] = perm[piv]; perm[piv] = tt; }
        }
        const real d = A[k * n + k];
        if (d == 0) continue;
        for (int i = k + 1; i < n; i++) A[i * n + k] = A[i * n + k] / d;
        for (int i = k + 1; i < n; i++) for (int j = k + 1; j <= i; j++) { A[i * n + j] -= A[i * n + k] * d * A[j * n + k]; A[j * n + i] = A[i * n + j]; }
    }
    real y[8];
    for (int i = 0; i < n; i++) { real s = bb[perm[i]]; for (int j = 0; j < i; j++) s -= A[i * n + j] * y[j]; y[i] = s; }
    for (int i = 0; i < n; i++) { const real d = A[i * n + i]; y[i] = d != 0 ? y[i] / d : 0; }
    real z[8];
    for (int i = n - 1; i >= 0; i--) { real s = y[i]; for (int j = i + 1; j < n; j++) s -= A[j * n + i] * z[j]; z[i] = s; }
    for (int i = 0; i < n; i++) x[perm[i]] = z[i];
}

/* trackFrame :81-285. dI[l] = the new frame's pyramid. Returns snapped && frameID > snappedAt + 5. */
int orc_initf_track_frame(void* p, const float* const* dI, float exposure_first, float exposure_new) {
    OrcInit* I = (OrcInit*)p;
    const int maxIterations[] = {5, 5, 10, 30, 50};
    I->alphaK = 2.5 * 2.5; I->alphaW = 150 * 150; I->regWeight = 0.8; I->couplingWeight = 1;
    if (!I->snapped) {
        I->thisToNext[3] = I->thisToNext[7] = I->thisToNext[11] = 0;
        for (int lvl = 0; lvl < I->levels; lvl++) { OrcInitLvl* P = &I->L[lvl]; for (int i = 0; i < P->n; i++) { P->iR[i] = 1; P->idepth_new[i] = 1; P->lastHessian[i] = 0; } }
    }
    double T_cur[12]; memcpy(T_cur, I->thisToNext, sizeof(T_cur));
    double aff_cur[2] = {I->aff[0], I->aff[1]};
    if (exposure_first > 0 && exposure_new > 0) { aff_cur[0] = logf(exposure_new / exposure_first); aff_cur[1] = 0; }
    const float wM[8] = {SCALE_XI_ROT, SCALE_XI_ROT, SCALE_XI_ROT, SCALE_XI_TRANS, SCALE_XI_TRANS, SCALE_XI_TRANS, SCALE_A, SCALE_B};   /* :64-67 (labels as in the reference) */
    for (int lvl = I->levels - 1; lvl >= 0; lvl--) {
        if (lvl < I->levels - 1) initf_propagate_down(I, lvl + 1);
        double H[64], b[8], Hsc[64], bsc[8]; float resOld[3];
        initf_reset_points(I, lvl);
        initf_calc(I, lvl, dI[lvl], T_cur, aff_cur, H, b, Hsc, bsc, resOld);
        initf_apply_step(I, lvl);
        float lambda = 0.1f; const float eps = 1e-4f; int fails = 0, iteration = 0;
        for (;;) {
            real Hl[64], bl[8];
            for (int i = 0; i < 64; i++) Hl[i] = (real)(float)H[i];
            for (int i = 0; i < 8; i++) Hl[i * 8 + i] *= (1 + lambda);
            for (int i = 0; i < 64; i++) Hl[i] -= (real)(float)Hsc[i] * (1 / (1 + lambda));
            for (int i = 0; i < 8; i++) bl[i] = (real)(float)b[i] - (real)(float)bsc[i] * (1 / (1 + lambda));
            const real sc = (0.01f / (I->w[lvl] * I->h[lvl]));
            for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) Hl[i * 8 + j] = wM[i] * Hl[i * 8 + j] * wM[j] * sc;
            for (int i = 0; i < 8; i++) bl[i] = wM[i] * bl[i] * sc;
            float inc[8];
            if (I->fixAffine) {
                real H6[36], x6[6];
                for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) H6[i * 6 + j] = Hl[i * 8 + j];
                ldlt_solve_real(6, H6, bl, x6);
                for (int i = 0; i < 6; i++) inc[i] = (float)(-(wM[i] * x6[i]));
                inc[6] = inc[7] = 0;
            } else {
                real x8[8]; ldlt_solve_real(8, Hl, bl, x8);
                for (int i = 0; i < 8; i++) inc[i] = (float)(-(wM[i] * x8[i]));
            }
            double xi[6], Te[12], T_new[12];
            for (int i = 0; i < 6; i++) xi[i] = inc[i];
            orc_se3_exp(xi, Te); orc_se3_mul(Te, T_cur, T_new);
            double aff_new[2] = {aff_cur[0] + inc[6], aff_cur[1] + inc[7]};
            OrcInitLvl* P = &I->L[lvl];
            orc_init_do_step(P->n, P->isGood, I->Jb, P->maxstep, P->idepth, lambda, inc, P->idepth_new);
            double Hn[64], bn[8], Hscn[64], bscn[8]; float resNew[3], regEnergy[3];
            initf_calc(I, lvl, dI[lvl], T_new, aff_new, Hn, bn, Hscn, bscn, resNew);
            initf_calc_ec(I, lvl, regEnergy);
            const float eTotalNew = (resNew[0] + resNew[1] + regEnergy[1]);
            const float eTotalOld = (resOld[0] + resOld[1] + regEnergy[0]);
            const int accept = eTotalOld > eTotalNew;
            if (accept) {
                if (resNew[1] == I->alphaK * P->n) I->snapped = 1;
                memcpy(H, Hn, sizeof(H)); memcpy(b, bn, sizeof(b)); memcpy(Hsc, Hscn, sizeof(Hsc)); memcpy(bsc, bscn, sizeof(bsc));
                resOld[0] = resNew[0]; resOld[1] = resNew[1]; resOld[2] = resNew[2];
                aff_cur[0] = aff_new[0]; aff_cur[1] = aff_new[1]; memcpy(T_cur, T_new, sizeof(T_cur));
                initf_apply_step(I, lvl);
                initf_opt_reg(I, lvl);
                lambda *= 0.5; fails = 0;
                if (lambda < 0.0001) lambda = 0.0001;
            } else {
                fails++; lambda *= 4;
                if (lambda > 10000) lambda = 10000;
            }
            float nrm = 0; for (int i = 0; i < 8; i++) nrm += inc[i] * inc[i]; nrm = sqrtf(nrm);
            if (!(nrm > eps) || iteration >= maxIterations[lvl] || fails >= 2) break;
            iteration++;
        }
    }
    memcpy(I->thisToNext, T_cur, sizeof(T_cur)); I->aff[0] = aff_cur[0]; I->aff[1] = aff_cur[1];
    for (int i = 0; i < I->levels - 1; i++) initf_propagate_up(I, i);
    I->frameID++;
    if (!I->snapped) I->snappedAt = 0;
    if (I->snapped && I->snappedAt == 0) I->snappedAt = I->frameID;
    return I->snapped && I->frameID > I->snappedAt + 5;
}

/* one sweep alone (tests): 0 optReg(lvl), 1 propagateUp(srcLvl = lvl), 2 propagateDown(srcLvl = lvl), 3 resetPoints(lvl) */
void orc_initf_sweep(void* p, int which, int lvl) {
    OrcInit* I = (OrcInit*)p;
    I->regWeight = 0.8f;
    if (which == 0) initf_opt_reg(I, lvl);
    else if (which == 1) initf_propagate_up(I, lvl);
    else if (which == 2) initf_propagate_down(I, lvl);
    else if (which == 3) initf_reset_points(I, lvl);
}
int orc_initf_num(void* p, int lvl) { return ((OrcInit*)p)->L[lvl].n; }
/* field: one of the Pnt members; out must hold n (x2 for energy, x10 for neighbours / neighboursDist) entries of float, int (parent, neighbours) or bytes (isGood) */
int orc_initf_get(void* p, int lvl, const char* field, void* out) {
    OrcInitLvl* P = &((OrcInit*)p)->L[lvl]; const size_t n = P->n;
#define G(name, ptr, mult, sz) if (!strcmp(field, name)) { memcpy(out, ptr, n * (mult) * (sz)); return 0; }
    G("u", P->u, 1, 4) G("v", P->v, 1, 4) G("idepth", P->idepth, 1, 4) G("idepth_new", P->idepth_new, 1, 4) G("iR", P->iR, 1, 4) G("lastHessian", P->lastHessian, 1, 4)
    G("energy", P->energy, 2, 4) G("outlierTH", P->outlierTH, 1, 4) G("my_type", P->my_type, 1, 4) G("neighboursDist", P->neighboursDist, 10, 4) G("parentDist", P->parentDist, 1, 4)
    G("isGood", P->isGood, 1, 1) G("parent", P->parent, 1, 4) G("neighbours", P->neighbours, 10, 4) G("maxstep", P->maxstep, 1, 4)
    G("lastHessian_new", P->lastHessian_new, 1, 4) G("energy_new", P->energy_new, 2, 4) G("isGood_new", P->isGood_new, 1, 1) G("iRSumNum", P->iRSumNum, 1, 4)
#undef G
    return -1;
}
/* the carried state of a Pnt array, written from outside (teacher-forced parity runs: every back-end starts a frame from the same state) */
int orc_initf_set(void* p, int lvl, const char* field, const void* in) {
    OrcInitLvl* P = &((OrcInit*)p)->L[lvl]; const size_t n = P->n;
#define S(name, ptr, mult, sz) if (!strcmp(field, name)) { memcpy(ptr, in, n * (mult) * (sz)); return 0; }
    S("idepth", P->idepth, 1, 4) S("idepth_new", P->idepth_new, 1, 4) S("iR", P->iR, 1, 4) S("lastHessian", P->lastHessian, 1, 4) S("energy", P->energy, 2, 4)
    S("isGood", P->isGood, 1, 1) S("maxstep", P->maxstep, 1, 4)
    S("lastHessian_new", P->lastHessian_new, 1, 4) S("energy_new", P->energy_new, 2, 4) S("isGood_new", P->isGood_new, 1, 1) S("iRSumNum", P->iRSumNum, 1, 4)
#undef S
    return -1;
}
void orc_initf_set_state(void* p, const double T[12], const double aff[2], const int st[3]) {
    OrcInit* I = (OrcInit*)p; memcpy(I->thisToNext, T, sizeof(I->thisToNext)); I->aff[0] = aff[0]; I->aff[1] = aff[1];
    I->snapped = st[0]; I->frameID = st[1]; I->snappedAt = st[2];
}
void orc_initf_get_state(void* p, double T[12], double aff[2], int st[4]) {
    OrcInit* I = (OrcInit*)p; memcpy(T, I->thisToNext, sizeof(I->thisToNext)); aff[0] = I->aff[0]; aff[1] = I->aff[1];
    st[0] = I->snapped; st[1] = I->frameID; st[2] = I->snappedAt; st[3] = I->n_evals;
}
