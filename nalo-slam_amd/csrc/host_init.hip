// The two-frame initialiser behind the C-ABI: CoarseInitializer::setFirst and ::trackFrame with everything they call.
//
//   nalo_init_set_first     setFirst          (reference src/FullSystem/CoarseInitializer.cpp:785-880): point selection of every pyramid level
//                                               (level 0: PixelSelector::makeMaps on the device, kernels_pixsel.hip; levels >= 1: makePixelStatus /
//                                               gridMaxSelection, FullSystem/PixelSelector.h:38-253 — the kernel below), Pnt construction, makeNN (:992-1069)
//   nalo_init_track_frame   trackFrame        (:81-285): per level propagateDown (:736-766), resetPoints (:882-909), the Levenberg-Marquardt loop around
//                                               calcResAndGS (:338-610, kernels_init.hip) with doStep (:910-938, kernels_init.hip), calcEC (:634-655),
//                                               applyStep (:939-956), optReg (:656-691); then propagateUp (:695-734) and the snapped / frameID bookkeeping
//   nalo_init_get_state / nalo_init_get_points   read-back of thisToNext, thisToNext_aff, snapped, frameID, snappedAt and the Pnt arrays
//
// What runs where: the two per-point passes that touch the images (calcResAndGS, doStep) and both selections are kernels; everything else here is SEQUENTIAL
// BY CONSTRUCTION in the reference — optReg is a Gauss-Seidel sweep (point i reads the iR its lower-index neighbours were just given), resetPoints likewise,
// propagateUp adds children into their parent in index order (fp32, order matters), and makeNN's result depends on the traversal order of nanoflann's k-d
// tree wherever neighbours are equidistant (points sit on the integer grid + 0.1, so most 10-NN sets end in a tie). The host code keeps those orders.
//
// Residency: inside a level's Levenberg-Marquardt loop the Pnt arrays (SoA) and both JbBuffers live on the DEVICE (InitDev below): doStep, calcResAndGS (which also
// forms calcEC's three sums) and applyStep are kernels on them, and one evaluation moves 94 doubles plus {idepth_new, isGood_new} down (what the host's mirror of
// applyStep and the next optReg sweep read) and, after an accepted step, the re-regularised iR up. The host mirror is authoritative between levels (propagateDown /
// resetPoints before a level, propagateUp after the last): one packed upload when a level starts, one packed download when it ends. makeNN's 10-NN / parent
// queries are independent per point and run on a few host threads against trees built one level per thread.
#include "nalo_internal.h"
#include <cfloat>
#include <cmath>
#include <numeric>
#include <thread>

namespace nalo {

// ---------------------------------------------------------------------------------------------------------------- gridMaxSelection (PixelSelector.h:38-117)
// One lane per pot x pot cell: the four directional maxima (|dx|, |dy|, |dx-dy|, |dx+dy|) among the cell's pixels whose squared gradient exceeds
// (0.75 * 10 * THFac)^2, first strict maximum in the reference's scan order (dx outer, dy inner). Marks up to four pixels in the byte map and counts the
// distinct ones (cells are disjoint, so the per-cell distinct count sums to the reference's numGood).
__global__ void grid_max_kernel(const float4* __restrict__ dI, uint8_t* __restrict__ map, int w, int h, int pot, int ncx, int ncy, float THFac, int* __restrict__ count) {
    const int cell = blockIdx.x * blockDim.x + threadIdx.x;
    int good = 0;
    if (cell < ncx * ncy) {
        const int x = 1 + (cell % ncx) * pot, y = 1 + (cell / ncx) * pot;
        int id[4] = {-1, -1, -1, -1};
        float best[4] = {0.f, 0.f, 0.f, 0.f};
        const float TH = THFac * 10.0f * (0.75f);
        for (int dx = 0; dx < pot; ++dx)
            for (int dy = 0; dy < pot; ++dy) {
                const int idx = dx + dy * w;
                const float4 g = dI[x + y * w + idx];
                const float sqgd = g.y * g.y + g.z * g.z;
                if (sqgd > TH * TH) {
                    const float v[4] = {fabsf(g.y), fabsf(g.z), fabsf(g.y - g.z), fabsf(g.y + g.z)};
#pragma unroll
                    for (int k = 0; k < 4; ++k) if (v[k] > best[k]) { best[k] = v[k]; id[k] = idx; }
                }
            }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (id[k] < 0) continue;
            bool seen = false;
            for (int j = 0; j < k; ++j) seen = seen || (id[j] == id[k]);
            if (!seen) { ++good; map[x + y * w + id[k]] = 1; }
        }
    }
    // wave sum, one atomic per wave (integer: order independent)
    for (int o = 32; o > 0; o >>= 1) good += __shfl_down(good, o);
    if ((threadIdx.x & 63) == 0 && good) atomicAdd(count, good);
}

// ---------------------------------------------------------------------------------------------------------------- k-d tree with nanoflann's build and search order
// (util/nanoflann.h: KDTreeSingleIndexAdaptor<L2_Simple_Adaptor<float, .>, ., 2>, leaf size 5, KNNResultSet). Exactness includes the ORDER in which
// equidistant points are met, so the split rule (widest bounding-box side among the candidates — evaluated the way the reference evaluates it, see
// pick_split —, middle cut clamped to the data, three-way partition, balance rule) and the search order (near child first, far child only if it can
// still beat the current worst) are the reference's; the data structure (flat node array, index ranges, explicit bounding boxes) is ours.
class GridKdTree {
public:
    GridKdTree(const float* u, const float* v, int n) : n_(n) {
        pt_[0] = u; pt_[1] = v;
        perm_.resize(n);
        std::iota(perm_.begin(), perm_.end(), 0);
        if (n == 0) return;
        Box bb;
        for (int d = 0; d < 2; ++d) bb.lo[d] = bb.hi[d] = pt_[d][0];
        for (int i = 1; i < n; ++i) for (int d = 0; d < 2; ++d) { bb.lo[d] = std::min(bb.lo[d], pt_[d][i]); bb.hi[d] = std::max(bb.hi[d], pt_[d][i]); }
        nodes_.reserve(n / 2 + 8);
        split(0, n, bb);
        root_ = bb;
    }
    // k nearest neighbours of q: idx/dist (ascending squared distance; slots that stay empty keep idx -1 / FLT_MAX)
    void knn(const float q[2], int k, int* idx, float* dist) const {
        for (int j = 0; j < k; ++j) { idx[j] = -1; dist[j] = FLT_MAX; }
        if (n_ == 0) return;
        Query Q{q, k, 0, idx, dist};
        float off[2] = {0.f, 0.f}, d2 = 0.f;
        for (int d = 0; d < 2; ++d) {
            if (q[d] < root_.lo[d]) { off[d] = (q[d] - root_.lo[d]) * (q[d] - root_.lo[d]); d2 += off[d]; }
            if (q[d] > root_.hi[d]) { off[d] = (q[d] - root_.hi[d]) * (q[d] - root_.hi[d]); d2 += off[d]; }
        }
        descend(Q, 0, d2, off);
    }

private:
    struct Box { float lo[2], hi[2]; };
    struct Node { int a, b; int dim; float lo_cut, hi_cut; };      // leaf: dim < 0, points perm_[a..b); inner: children a, b
    struct Query { const float* q; int k, have; int* idx; float* dist; };

    void range(const int* ind, int cnt, int d, float& mn, float& mx) const {
        mn = mx = pt_[d][ind[0]];
        for (int i = 1; i < cnt; ++i) { const float x = pt_[d][ind[i]]; if (x < mn) mn = x; if (x > mx) mx = x; }
    }
    // three-way partition around cut: [< cut | == cut | > cut]; returns the two boundaries. Swap order as in planeSplit (nanoflann.h:1169-1195)
    void partition(int* ind, int cnt, int d, float cut, int& m1, int& m2) const {
        int l = 0, r = cnt - 1;
        for (;;) {
            while (l <= r && pt_[d][ind[l]] < cut) ++l;
            while (r && l <= r && pt_[d][ind[r]] >= cut) --r;
            if (l > r || !r) break;
            std::swap(ind[l], ind[r]); ++l; --r;
        }
        m1 = l; r = cnt - 1;
        for (;;) {
            while (l <= r && pt_[d][ind[l]] <= cut) ++l;
            while (r && l <= r && pt_[d][ind[r]] > cut) --r;
            if (l > r || !r) break;
            std::swap(ind[l], ind[r]); ++l; --r;
        }
        m2 = l;
    }
    // middleSplit_ (nanoflann.h:1118-1158). Note the reference measures the spread along the CURRENT candidate `dim` (not along i) while it scans i.
    void pick_split(int* ind, int cnt, const Box& bb, int& at, int& dim, float& cut) const {
        const float eps = 0.00001f;
        const float span0 = bb.hi[0] - bb.lo[0], span1 = bb.hi[1] - bb.lo[1], widest = span1 > span0 ? span1 : span0;
        float spread_best = -1.f;
        dim = 0;
        for (int i = 0; i < 2; ++i) {
            const float span = bb.hi[i] - bb.lo[i];
            if (span > (1 - eps) * widest) {
                float mn, mx; range(ind, cnt, dim, mn, mx);
                const float spread = mx - mn;
                if (spread > spread_best) { dim = i; spread_best = spread; }
            }
        }
        const float mid = (bb.lo[dim] + bb.hi[dim]) / 2;
        float mn, mx; range(ind, cnt, dim, mn, mx);
        cut = mid < mn ? mn : (mid > mx ? mx : mid);
        int m1, m2; partition(ind, cnt, dim, cut, m1, m2);
        at = m1 > cnt / 2 ? m1 : (m2 < cnt / 2 ? m2 : cnt / 2);
    }
    int split(int a, int b, Box& bb) {                               // bb: in = the cell, out = tight box of the points (divideTree :1056-1104)
        const int me = (int)nodes_.size();
        nodes_.push_back(Node{});
        if (b - a <= kLeaf) {
            for (int d = 0; d < 2; ++d) bb.lo[d] = bb.hi[d] = pt_[d][perm_[a]];
            for (int i = a + 1; i < b; ++i) for (int d = 0; d < 2; ++d) { const float x = pt_[d][perm_[i]]; if (bb.lo[d] > x) bb.lo[d] = x; if (bb.hi[d] < x) bb.hi[d] = x; }
            nodes_[me] = Node{a, b, -1, 0.f, 0.f};
            return me;
        }
        int at, dim; float cut;
        pick_split(perm_.data() + a, b - a, bb, at, dim, cut);
        Box lb = bb, rb = bb;
        lb.hi[dim] = cut; rb.lo[dim] = cut;
        const int c1 = split(a, a + at, lb), c2 = split(a + at, b, rb);
        nodes_[me] = Node{c1, c2, dim, lb.hi[dim], rb.lo[dim]};
        for (int d = 0; d < 2; ++d) { bb.lo[d] = std::min(lb.lo[d], rb.lo[d]); bb.hi[d] = std::max(lb.hi[d], rb.hi[d]); }
        return me;
    }
    static void offer(Query& Q, float d2, int id) {                 // KNNResultSet::addPoint (:124-145): insertion behind equal distances
        int i = Q.have;
        for (; i > 0 && Q.dist[i - 1] > d2; --i) if (i < Q.k) { Q.dist[i] = Q.dist[i - 1]; Q.idx[i] = Q.idx[i - 1]; }
        if (i < Q.k) { Q.dist[i] = d2; Q.idx[i] = id; }
        if (Q.have < Q.k) ++Q.have;
    }
    void descend(Query& Q, int node, float lower, float off[2]) const {   // searchLevel (:1222-1270)
        const Node& nd = nodes_[node];
        if (nd.dim < 0) {
            const float worst = Q.dist[Q.k - 1];                     // sampled once per leaf, as the reference does
            for (int i = nd.a; i < nd.b; ++i) {
                const int id = perm_[i];
                const float d0 = Q.q[0] - pt_[0][id], d1 = Q.q[1] - pt_[1][id];
                const float d2 = d0 * d0 + d1 * d1;
                if (d2 < worst) offer(Q, d2, id);
            }
            return;
        }
        const float x = Q.q[nd.dim], e1 = x - nd.lo_cut, e2 = x - nd.hi_cut;
        const bool left_first = (e1 + e2) < 0;
        const float gap = left_first ? e2 * e2 : e1 * e1;
        descend(Q, left_first ? nd.a : nd.b, lower, off);
        const float keep = off[nd.dim];
        lower = lower + gap - keep;
        off[nd.dim] = gap;
        if (lower * 1.0f <= Q.dist[Q.k - 1]) descend(Q, left_first ? nd.b : nd.a, lower, off);
        off[nd.dim] = keep;
    }

    static constexpr int kLeaf = 5;
    int n_;
    const float* pt_[2];
    std::vector<int> perm_;
    std::vector<Node> nodes_;
    Box root_;
};

// ---------------------------------------------------------------------------------------------------------------- the initialiser's state
struct InitLevel {
    int n = 0;
    std::vector<float> u, v, idepth, idepth_new, iR, iRSumNum, lastHessian, lastHessian_new, maxstep, energy, energy_new, outlierTH, my_type, nnDist, parentDist;
    std::vector<uint8_t> isGood, isGood_new;
    std::vector<int> parent, nn;
    void resize(int n_) {
        n = n_;
        for (auto* a : {&u, &v, &idepth, &idepth_new, &iR, &iRSumNum, &lastHessian, &lastHessian_new, &maxstep, &outlierTH, &my_type, &parentDist}) a->assign(n_, 0.f);
        energy.assign(2 * (size_t)n_, 0.f); energy_new.assign(2 * (size_t)n_, 0.f); nnDist.assign(10 * (size_t)n_, 0.f);
        isGood.assign(n_, 0); isGood_new.assign(n_, 0); parent.assign(n_, -1); nn.assign(10 * (size_t)n_, -1);
    }
};

constexpr int kInitHostThreads = 8;                      // makeNN's query slices (the reference's own pool has NUM_THREADS = 6, util/NumType.h:42)

struct Initializer {
    int levels = 0, slot_first = -1;
    InitLevel L[NALO_MAX_LEVELS];
    SE3 thisToNext = SE3::identity();
    double aff[2] = {0, 0};                              // thisToNext_aff
    bool snapped = false, fixAffine = true;
    int frameID = -1, snappedAt = 0, n_evals = 0;
    float alphaK = 2.5f * 2.5f, alphaW = 150.f * 150.f, regWeight = 0.8f, couplingWeight = 1.f;
    DevBuf<uint8_t> map_dev; DevBuf<int> cnt_dev;
    // device residency of the Pnt arrays: one block per level, word offsets from lvl_off(); JbBuffer / JbBuffer_new are one pair for all levels, as in the reference
    DevBuf<float> lvl_dev[NALO_MAX_LEVELS], jb_dev[2];
    bool static_on_dev = false;
    int jb_cur = 0;
    float* pin = nullptr; size_t pin_half = 0;           // pinned staging: [0, pin_half) host -> device, [pin_half, 2 pin_half) device -> host
};

// word offsets inside a level's device block: the static members, then the members a level's LM loop changes ("dyn": one packed upload / download per level), the last
// two of which (idepth_new, isGood_new) come down after every evaluation together with the 94 sums that follow them
struct LvlOff { size_t u, v, outlierTH, dyn, idepth, iR, lastHessian, lastHessian_new, maxstep, energy, energy_new, isGood, idepth_new, isGood_new, dyn_end, sums, total; };
static LvlOff lvl_off(size_t n) {
    const size_t nb = (n + 3) / 4;
    LvlOff o;
    o.u = 0; o.v = n; o.outlierTH = 2 * n; o.dyn = (3 * n + 1) & ~(size_t)1;
    o.idepth = o.dyn; o.iR = o.idepth + n; o.lastHessian = o.iR + n; o.lastHessian_new = o.lastHessian + n; o.maxstep = o.lastHessian_new + n;
    o.energy = o.maxstep + n; o.energy_new = o.energy + 2 * n; o.isGood = o.energy_new + 2 * n; o.idepth_new = o.isGood + nb; o.isGood_new = o.idepth_new + n;
    o.dyn_end = o.isGood_new + nb; o.sums = (o.dyn_end + 1) & ~(size_t)1; o.total = o.sums + 2 * 96;
    return o;
}

void init_destroy(nalo_ctx* c) {
    if (!c->init) return;
    c->init->map_dev.release(); c->init->cnt_dev.release();
    for (auto& b : c->init->lvl_dev) b.release();
    for (auto& b : c->init->jb_dev) b.release();
    if (c->init->pin) (void)hipHostFree(c->init->pin);
    delete c->init; c->init = nullptr;
}

// Hl.ldlt().solve(bl) for the 6x6 / 8x8 float system of the LM step (Eigen's LDLT: symmetric pivoting on the largest remaining diagonal entry), fp32
template <int N>
static void ldlt_f32(const float* Ain, const float* bin, float* x) {
    float A[N * N], y[N], z[N]; int perm[N];
    for (int i = 0; i < N * N; ++i) A[i] = Ain[i];
    for (int i = 0; i < N; ++i) perm[i] = i;
    for (int k = 0; k < N; ++k) {
        int p = k; float best = std::fabs(A[k * N + k]);
        for (int i = k + 1; i < N; ++i) if (std::fabs(A[i * N + i]) > best) { best = std::fabs(A[i * N + i]); p = i; }
        if (p != k) {
            for (int j = 0; j < N; ++j) std::swap(A[k * N + j], A[p * N + j]);
            for (int j = 0; j < N; ++j) std::swap(A[j * N + k], A[j * N + p]);
            std::swap(perm[k], perm[p]);
        }
        const float d = A[k * N + k];
        if (d == 0.f) continue;
        for (int i = k + 1; i < N; ++i) A[i * N + k] = A[i * N + k] / d;
        for (int i = k + 1; i < N; ++i) for (int j = k + 1; j <= i; ++j) { A[i * N + j] -= A[i * N + k] * d * A[j * N + k]; A[j * N + i] = A[i * N + j]; }
    }
    for (int i = 0; i < N; ++i) { float s = bin[perm[i]]; for (int j = 0; j < i; ++j) s -= A[i * N + j] * y[j]; y[i] = s; }
    for (int i = 0; i < N; ++i) { const float d = A[i * N + i]; y[i] = d != 0.f ? y[i] / d : 0.f; }
    for (int i = N - 1; i >= 0; --i) { float s = y[i]; for (int j = i + 1; j < N; ++j) s -= A[j * N + i] * z[j]; z[i] = s; }
    for (int i = 0; i < N; ++i) x[perm[i]] = z[i];
}

static void opt_reg(Initializer& I, int lvl) {                                                 // optReg :656-691
    InitLevel& P = I.L[lvl];
    if (!I.snapped) { std::fill(P.iR.begin(), P.iR.end(), 1.f); return; }
    for (int i = 0; i < P.n; ++i) {
        if (!P.isGood[i]) continue;
        float vals[10]; int m = 0;
        for (int j = 0; j < 10; ++j) { const int o = P.nn[(size_t)i * 10 + j]; if (o != -1 && P.isGood[o]) vals[m++] = P.iR[o]; }
        if (m > 2) {
            std::nth_element(vals, vals + m / 2, vals + m);
            P.iR[i] = (1 - I.regWeight) * P.idepth[i] + I.regWeight * vals[m / 2];
        }
    }
}
static void propagate_up(Initializer& I, int src) {                                            // propagateUp :695-734
    InitLevel &S = I.L[src], &T = I.L[src + 1];
    std::fill(T.iR.begin(), T.iR.end(), 0.f); std::fill(T.iRSumNum.begin(), T.iRSumNum.end(), 0.f);
    for (int i = 0; i < S.n; ++i) {
        if (!S.isGood[i]) continue;
        const int par = S.parent[i];
        T.iR[par] += S.iR[i] * S.lastHessian[i];
        T.iRSumNum[par] += S.lastHessian[i];
    }
    for (int i = 0; i < T.n; ++i) if (T.iRSumNum[i] > 0) { T.idepth[i] = T.iR[i] = (T.iR[i] / T.iRSumNum[i]); T.isGood[i] = 1; }
    opt_reg(I, src + 1);
}
static void propagate_down(Initializer& I, int src) {                                          // propagateDown :736-766
    InitLevel &S = I.L[src], &T = I.L[src - 1];
    for (int i = 0; i < T.n; ++i) {
        const int par = T.parent[i];
        if (!S.isGood[par] || S.lastHessian[par] < 0.1) continue;
        if (!T.isGood[i]) { T.iR[i] = T.idepth[i] = T.idepth_new[i] = S.iR[par]; T.isGood[i] = 1; T.lastHessian[i] = 0; }
        else {
            const float fused = (T.iR[i] * T.lastHessian[i] * 2 + S.iR[par] * S.lastHessian[par]) / (T.lastHessian[i] * 2 + S.lastHessian[par]);
            T.iR[i] = T.idepth[i] = T.idepth_new[i] = fused;
        }
    }
    opt_reg(I, src - 1);
}
static void reset_points(Initializer& I, int lvl) {                                            // resetPoints :882-909
    InitLevel& P = I.L[lvl];
    for (int i = 0; i < P.n; ++i) {
        P.energy[2 * (size_t)i] = P.energy[2 * (size_t)i + 1] = 0;
        P.idepth_new[i] = P.idepth[i];
        if (lvl == I.levels - 1 && !P.isGood[i]) {
            float sum = 0, cnt = 0;
            for (int j = 0; j < 10; ++j) { const int o = P.nn[(size_t)i * 10 + j]; if (o == -1 || !P.isGood[o]) continue; sum += P.iR[o]; cnt += 1; }
            if (cnt > 0) { P.isGood[i] = 1; P.iR[i] = P.idepth[i] = P.idepth_new[i] = sum / cnt; }
        }
    }
}
// applyStep (:939-956) on the host mirror, for the members the host sweeps read (the device runs init_apply_step_kernel on all of them; the JbBuffer swap is jb_cur)
static void apply_step_host(Initializer& I, int lvl) {
    InitLevel& P = I.L[lvl];
    for (int i = 0; i < P.n; ++i) {
        if (!P.isGood[i]) { P.idepth[i] = P.idepth_new[i] = P.iR[i]; continue; }
        P.isGood[i] = P.isGood_new[i];
        P.idepth[i] = P.idepth_new[i];
    }
}

static int dev_prepare(nalo_ctx* c, Initializer& I) {                                          // device blocks + pinned staging for the current point counts
    size_t maxn = 0, maxtot = 0;
    for (int l = 0; l < I.levels; ++l) { const LvlOff o = lvl_off((size_t)I.L[l].n); NALO_HIP(c, I.lvl_dev[l].reserve(o.total)); maxn = std::max(maxn, (size_t)I.L[l].n); maxtot = std::max(maxtot, o.total); }
    for (auto& b : I.jb_dev) NALO_HIP(c, b.reserve(10 * maxn + 16));
    if (I.pin_half < maxtot) {
        if (I.pin) (void)hipHostFree(I.pin);
        I.pin = nullptr; I.pin_half = 0;
        NALO_HIP(c, hipHostMalloc((void**)&I.pin, 2 * maxtot * sizeof(float)));
        I.pin_half = maxtot;
    }
    if (!I.static_on_dev) {
        for (int l = 0; l < I.levels; ++l) {
            const InitLevel& P = I.L[l]; const size_t n = (size_t)P.n; const LvlOff o = lvl_off(n);
            if (!n) continue;
            std::memcpy(I.pin + o.u, P.u.data(), n * 4); std::memcpy(I.pin + o.v, P.v.data(), n * 4); std::memcpy(I.pin + o.outlierTH, P.outlierTH.data(), n * 4);
            NALO_HIP(c, hipMemcpyAsync(I.lvl_dev[l].p, I.pin, 3 * n * 4, hipMemcpyHostToDevice, c->stream));
            NALO_HIP(c, hipStreamSynchronize(c->stream));
        }
        I.static_on_dev = true;
    }
    return NALO_OK;
}
// host mirror -> device, every member a level's loop reads or keeps (the *_new members hold stale entries that survive, as in the reference)
static int level_upload(nalo_ctx* c, Initializer& I, int lvl) {
    HostTimer ht(c, "init.level_io");
    const InitLevel& P = I.L[lvl]; const size_t n = (size_t)P.n; const LvlOff o = lvl_off(n);
    if (!n) return NALO_OK;
    float* h = I.pin - o.dyn;
    std::memcpy(h + o.idepth, P.idepth.data(), n * 4); std::memcpy(h + o.iR, P.iR.data(), n * 4); std::memcpy(h + o.lastHessian, P.lastHessian.data(), n * 4);
    std::memcpy(h + o.lastHessian_new, P.lastHessian_new.data(), n * 4); std::memcpy(h + o.maxstep, P.maxstep.data(), n * 4); std::memcpy(h + o.energy, P.energy.data(), 2 * n * 4);
    std::memcpy(h + o.energy_new, P.energy_new.data(), 2 * n * 4); std::memcpy(h + o.isGood, P.isGood.data(), n); std::memcpy(h + o.idepth_new, P.idepth_new.data(), n * 4);
    std::memcpy(h + o.isGood_new, P.isGood_new.data(), n);
    NALO_HIP(c, hipMemcpyAsync(I.lvl_dev[lvl].p + o.dyn, I.pin, (o.dyn_end - o.dyn) * 4, hipMemcpyHostToDevice, c->stream));
    return NALO_OK;
}
static int level_download(nalo_ctx* c, Initializer& I, int lvl) {
    HostTimer ht(c, "init.level_io");
    InitLevel& P = I.L[lvl]; const size_t n = (size_t)P.n; const LvlOff o = lvl_off(n);
    if (!n) return NALO_OK;
    float* hd = I.pin + I.pin_half;
    NALO_HIP(c, hipMemcpyAsync(hd, I.lvl_dev[lvl].p + o.dyn, (o.dyn_end - o.dyn) * 4, hipMemcpyDeviceToHost, c->stream));
    NALO_HIP(c, hipStreamSynchronize(c->stream));
    const float* h = hd - o.dyn;
    std::memcpy(P.idepth.data(), h + o.idepth, n * 4); std::memcpy(P.lastHessian.data(), h + o.lastHessian, n * 4); std::memcpy(P.lastHessian_new.data(), h + o.lastHessian_new, n * 4);
    std::memcpy(P.maxstep.data(), h + o.maxstep, n * 4); std::memcpy(P.energy.data(), h + o.energy, 2 * n * 4); std::memcpy(P.energy_new.data(), h + o.energy_new, 2 * n * 4);
    std::memcpy(P.isGood.data(), h + o.isGood, n); std::memcpy(P.idepth_new.data(), h + o.idepth_new, n * 4); std::memcpy(P.isGood_new.data(), h + o.isGood_new, n);
    return NALO_OK;
}
static int iR_upload(nalo_ctx* c, Initializer& I, int lvl) {                                   // after optReg: the only member the host changes inside the loop
    const InitLevel& P = I.L[lvl]; const size_t n = (size_t)P.n; const LvlOff o = lvl_off(n);
    if (!n) return NALO_OK;
    std::memcpy(I.pin, P.iR.data(), n * 4);
    NALO_HIP(c, hipMemcpyAsync(I.lvl_dev[lvl].p + o.iR, I.pin, n * 4, hipMemcpyHostToDevice, c->stream));
    return NALO_OK;
}

// calcResAndGS on the resident level (writes JbBuffer_new = jb_dev[1 - jb_cur]); brings back the sums and {idepth_new, isGood_new}. regEnergy = calcEC (:634-655)
static int calc(nalo_ctx* c, Initializer& I, int lvl, int slot_new, const SE3& T, const double aff[2], double H[64], double b[8], double Hsc[64], double bsc[8], float res[3], float regEnergy[3]) {
    HostTimer ht(c, "init.calc");
    InitLevel& L = I.L[lvl]; const size_t n = (size_t)L.n; const LvlOff o = lvl_off(n);
    InitParams P; InitPose X;
    init_pose_setup(c, lvl, L.n, T, aff, I.alphaW, I.alphaK, I.couplingWeight, P, X);
    double sums[96] = {};
    if (n) {
        float* d = I.lvl_dev[lvl].p;
        P.colorRef = c->slots[I.slot_first].dI[lvl]; P.colorNew = c->slots[slot_new].dI[lvl];
        P.u = d + o.u; P.v = d + o.v; P.outlierTH = d + o.outlierTH; P.idepth = d + o.idepth; P.idepth_new = d + o.idepth_new; P.iR = d + o.iR; P.energy = d + o.energy;
        P.isGood = (const uint8_t*)(d + o.isGood); P.isGood_new = (uint8_t*)(d + o.isGood_new); P.energy_new = d + o.energy_new; P.maxstep = d + o.maxstep;
        P.lastHessian_new = d + o.lastHessian_new; P.Jb = I.jb_dev[1 - I.jb_cur].p;
        int rc = init_calc_launch(c, P, lvl, (double*)(d + o.sums)); if (rc) return rc;
        float* hd = I.pin + I.pin_half;
        NALO_HIP(c, hipMemcpyAsync(hd, d + o.idepth_new, (o.sums + 2 * 94 - o.idepth_new) * 4, hipMemcpyDeviceToHost, c->stream));
        NALO_HIP(c, hipStreamSynchronize(c->stream));
        std::memcpy(L.idepth_new.data(), hd, n * 4); std::memcpy(L.isGood_new.data(), hd + (o.isGood_new - o.idepth_new), n);
        std::memcpy(sums, hd + (o.sums - o.idepth_new), 94 * 8);
    }
    double E3[3];
    init_sums_to_system(sums, T, L.n, P, X, H, b, Hsc, bsc, E3);
    res[0] = (float)E3[0]; res[1] = (float)E3[1]; res[2] = (float)E3[2];
    if (!I.snapped) { regEnergy[0] = 0; regEnergy[1] = 0; regEnergy[2] = (float)L.n; }
    else { regEnergy[0] = I.couplingWeight * (float)sums[91]; regEnergy[1] = I.couplingWeight * (float)sums[92]; regEnergy[2] = (float)sums[93]; }
    ++I.n_evals;
    return NALO_OK;
}
static int apply_step(nalo_ctx* c, Initializer& I, int lvl) {
    HostTimer ht(c, "init.apply_step");
    InitLevel& L = I.L[lvl]; const LvlOff o = lvl_off((size_t)L.n);
    float* d = I.lvl_dev[lvl].p;
    const int rc = init_apply_step_launch(c, L.n, (uint8_t*)(d + o.isGood), (const uint8_t*)(d + o.isGood_new), d + o.idepth, d + o.idepth_new, d + o.iR, d + o.energy, d + o.energy_new,
                                          d + o.lastHessian, d + o.lastHessian_new);
    if (rc) return rc;
    apply_step_host(I, lvl);
    I.jb_cur = 1 - I.jb_cur;
    return NALO_OK;
}

void init_pose_setup(const nalo_ctx* c, int lvl, int n, const SE3& T, const double aff[2], float alphaW, float alphaK, float couplingWeight, InitParams& P, InitPose& X) {
    // RKi = (R * K^-1).cast<float>(), t.cast<float>(), r2new_aff = (exp(a), b) as floats (:347-349)
    const double fxd = c->fx[lvl], fyd = c->fy[lvl], cxd = c->cx[lvl], cyd = c->cy[lvl];
    const double Ki[9] = {1.0 / fxd, 0, -cxd / fxd, 0, 1.0 / fyd, -cyd / fyd, 0, 0, 1};
    for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) P.RKi[i * 3 + j] = (float)(T.R(i, 0) * Ki[j] + T.R(i, 1) * Ki[3 + j] + T.R(i, 2) * Ki[6 + j]); P.t[i] = (float)T.t(i); }
    P.r2new0 = (float)std::exp(aff[0]); P.r2new1 = (float)aff[1];
    P.fx = (float)fxd; P.fy = (float)fyd; P.cx = (float)cxd; P.cy = (float)cyd;
    const double tsq = T.t(0) * T.t(0) + T.t(1) * T.t(1) + T.t(2) * T.t(2);
    float alphaEnergy = (float)((double)alphaW * (0.0 + tsq * n));                     // EAlpha.A is always 0 in the reference (:560-575)
    if (alphaEnergy > alphaK * n) { P.alphaOpt = 0; alphaEnergy = alphaK * n; } else P.alphaOpt = alphaW;
    P.couplingWeight = couplingWeight; P.n = n;
    X.alphaEnergy = alphaEnergy;
}
void init_sums_to_system(const double* sums, const SE3& T, int n, const InitParams& P, const InitPose& X, double* H_out, double* b_out, double* H_out_sc, double* b_out_sc, double E3[3]) {
    // Accumulator9::finish -> float matrix, then topLeftCorner<8,8> / topRightCorner<8,1> (:596-599) and the alpha terms (:601-607)
    int e = 0;
    for (int a = 0; a < 9; ++a) for (int b = a; b < 9; ++b, ++e) {
        const double va = (double)(float)sums[e], vs = (double)(float)sums[45 + e];
        if (b < 8) { H_out[a * 8 + b] = H_out[b * 8 + a] = va; H_out_sc[a * 8 + b] = H_out_sc[b * 8 + a] = vs; }
        else if (a < 8) { b_out[a] = va; b_out_sc[a] = vs; }
    }
    double xi[6]; se3_log(T, xi);
    for (int k = 0; k < 3; ++k) {
        H_out[k * 8 + k] = (double)(float)((float)H_out[k * 8 + k] + P.alphaOpt * n);
        b_out[k] = (double)(float)((float)b_out[k] + (float)xi[k] * P.alphaOpt * n);
    }
    E3[0] = (double)(float)sums[90]; E3[1] = X.alphaEnergy; E3[2] = 2.0 * n;
}

static int make_pixel_status(nalo_ctx* c, Initializer& I, int slot, int lvl, float desiredDensity, int* sparsityFactor, std::vector<uint8_t>& map_host, int* numGood) {
    // makePixelStatus (PixelSelector.h:199-253), the recursion as a loop; sparsityFactor is the reference's global (util/settings.cpp:223)
    const int w = c->wl[lvl], h = c->hl[lvl];
    const size_t npx = (size_t)w * h;
    NALO_HIP(c, I.map_dev.reserve(npx)); NALO_HIP(c, I.cnt_dev.reserve(1));
    int recsLeft = 5; float THFac = 1.f;
    for (;;) {
        if (*sparsityFactor < 1) *sparsityFactor = 1;
        const int pot = *sparsityFactor;
        const int ncx = (w - pot - 1 + pot - 1) / pot, ncy = (h - pot - 1 + pot - 1) / pot;      // x = 1, 1+pot, .. < w-pot
        NALO_HIP(c, hipMemsetAsync(I.map_dev.p, 0, npx, c->stream)); NALO_HIP(c, hipMemsetAsync(I.cnt_dev.p, 0, sizeof(int), c->stream));
        int good = 0;
        if (ncx > 0 && ncy > 0) {
            const int cells = ncx * ncy;
            grid_max_kernel<<<(cells + 255) / 256, 256, 0, c->stream>>>(c->slots[slot].dI[lvl], I.map_dev.p, w, h, pot, ncx, ncy, THFac, I.cnt_dev.p);
            NALO_HIP(c, hipGetLastError());
        }
        NALO_HIP(c, hipMemcpyAsync(&good, I.cnt_dev.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        NALO_HIP(c, hipStreamSynchronize(c->stream));
        const float quotia = good / (float)(desiredDensity);
        int newSparsity = (int)((*sparsityFactor * sqrtf(quotia)) + 0.7f);
        if (newSparsity < 1) newSparsity = 1;
        const float oldTHFac = THFac;
        if (newSparsity == 1 && *sparsityFactor == 1) THFac = 0.5;
        const bool done = (std::abs(newSparsity - *sparsityFactor) < 1 && THFac == oldTHFac) || (quotia > 0.8 && 1.0f / quotia > 0.8) || recsLeft == 0;
        *sparsityFactor = newSparsity;
        if (done) { *numGood = good; break; }
        --recsLeft;
    }
    map_host.resize(npx);
    NALO_HIP(c, hipMemcpy(map_host.data(), I.map_dev.p, npx, hipMemcpyDeviceToHost));
    return NALO_OK;
}

}  // namespace nalo

using namespace nalo;

int nalo_init_set_first(nalo_ctx* c, int slot_first, int* sparsityFactor, int numPoints[NALO_MAX_LEVELS]) {
    if (!c || !sparsityFactor) return fail(c, NALO_ERR_ARG, "nalo_init_set_first: bad argument");
    if (slot_first < 0 || slot_first >= (int)c->slots.size() || !c->slots[slot_first].valid) return fail(c, NALO_ERR_STATE, "nalo_init_set_first: frame slot has no pyramid");
    NALO_HIP(c, hipSetDevice(c->device));
    if (!c->init) c->init = new Initializer();
    Initializer& I = *c->init;
    I.levels = c->levels; I.slot_first = slot_first;
    I.static_on_dev = false; I.jb_cur = 0;
    HostTimer htf(c, "init_set_first");
    const float densities[] = {0.03f, 0.05f, 0.15f, 0.5f, 1.f, 1.f};
    const int pad = 2;                                                                           // patternPadding (util/settings.h:234)
    std::vector<float> status0((size_t)c->w * c->h);
    std::vector<uint8_t> mapB;
    for (int lvl = 0; lvl < I.levels; ++lvl) {
        const int wl = c->wl[lvl], hl = c->hl[lvl];
        HostTimer hs(c, lvl ? "init.select_upper" : "init.select_l0");
        if (lvl == 0) {
            // PixelSelector sel(w,h); sel.currentPotential = 3; sel.makeMaps(firstFrame, statusMap, densities[0]*w*h, 1, false, 2)   (:806-811)
            int pot = 3, have = 0;
            const int rc = nalo_pixsel_make_maps(c, slot_first, densities[0] * c->w * c->h, 1, 2.f, &pot, status0.data(), &have);
            if (rc) return rc;
        } else {
            int good = 0;
            const int rc = make_pixel_status(c, I, slot_first, lvl, densities[lvl] * c->w * c->h, sparsityFactor, mapB, &good);
            if (rc) return rc;
        }
        int nl = 0;
        for (int y = pad + 1; y < hl - pad - 2; ++y) for (int x = pad + 1; x < wl - pad - 2; ++x) nl += lvl ? (mapB[x + (size_t)y * wl] != 0) : (status0[x + (size_t)y * wl] != 0);
        InitLevel& P = I.L[lvl];
        P.resize(nl);
        nl = 0;
        for (int y = pad + 1; y < hl - pad - 2; ++y)
            for (int x = pad + 1; x < wl - pad - 2; ++x) {
                const bool take = lvl ? (mapB[x + (size_t)y * wl] != 0) : (status0[x + (size_t)y * wl] != 0);
                if (!take) continue;
                P.u[nl] = x + 0.1; P.v[nl] = y + 0.1; P.idepth[nl] = 1; P.iR[nl] = 1; P.isGood[nl] = 1;
                P.my_type[nl] = lvl ? 1.f : status0[x + (size_t)y * wl];
                P.outlierTH[nl] = kPatternNum * kOutlierTH;                                   // patternNum * setting_outlierTH (util/settings.cpp:99)
                ++nl;
            }
        if (numPoints) numPoints[lvl] = nl;
    }
    // makeNN (:992-1069): 10 nearest neighbours inside the level, nearest point of (u/2 - 0.25, v/2 - 0.25) one level up. The trees are built one level per thread
    // (each build is sequential, its order decides the ties); the queries only read them and are independent per point, so they are cut into slices for a few threads.
    {
        HostTimer hn(c, "init.make_nn");
        const float NNDistFactor = 0.05f;
        std::vector<GridKdTree*> trees(I.levels, nullptr);
        {
            std::vector<std::thread> th;
            for (int l = 1; l < I.levels; ++l) th.emplace_back([&, l] { trees[l] = new GridKdTree(I.L[l].u.data(), I.L[l].v.data(), I.L[l].n); });
            trees[0] = new GridKdTree(I.L[0].u.data(), I.L[0].v.data(), I.L[0].n);
            for (auto& t : th) t.join();
        }
        auto queries = [&](int lvl, int i0, int i1) {
            InitLevel& P = I.L[lvl];
            for (int i = i0; i < i1; ++i) {
                int ri[10]; float rd[10];
                float q[2] = {P.u[i], P.v[i]};
                trees[lvl]->knn(q, 10, ri, rd);
                float sumDF = 0;
                for (int k = 0; k < 10; ++k) {
                    P.nn[(size_t)i * 10 + k] = ri[k];
                    const float df = expf(-rd[k] * NNDistFactor);
                    sumDF += df;
                    P.nnDist[(size_t)i * 10 + k] = df;
                }
                for (int k = 0; k < 10; ++k) P.nnDist[(size_t)i * 10 + k] *= 10 / sumDF;
                if (lvl < I.levels - 1) {
                    q[0] = q[0] * 0.5f - 0.25f; q[1] = q[1] * 0.5f - 0.25f;
                    trees[lvl + 1]->knn(q, 1, ri, rd);
                    P.parent[i] = ri[0];
                    P.parentDist[i] = expf(-rd[0] * NNDistFactor);
                } else { P.parent[i] = -1; P.parentDist[i] = -1; }
            }
        };
        const int hw = (int)std::thread::hardware_concurrency();
        const int nt = std::max(1, std::min(kInitHostThreads, hw > 0 ? hw : 1));
        auto slice = [&](int t) { for (int lvl = 0; lvl < I.levels; ++lvl) { const long n = I.L[lvl].n; queries(lvl, (int)(n * t / nt), (int)(n * (t + 1) / nt)); } };
        std::vector<std::thread> th;
        for (int t = 1; t < nt; ++t) th.emplace_back(slice, t);
        slice(0);
        for (auto& t : th) t.join();
        for (auto* t : trees) delete t;
    }
    I.thisToNext = SE3::identity();
    I.snapped = false; I.frameID = I.snappedAt = 0;
    return NALO_OK;
}

int nalo_init_track_frame(nalo_ctx* c, int slot_new, float exposure_first, float exposure_new, int* ok) {
    if (!c || !ok) return fail(c, NALO_ERR_ARG, "nalo_init_track_frame: bad argument");
    if (!c->init || c->init->slot_first < 0) return fail(c, NALO_ERR_STATE, "nalo_init_track_frame: nalo_init_set_first has not run");
    if (slot_new < 0 || slot_new >= (int)c->slots.size() || !c->slots[slot_new].valid) return fail(c, NALO_ERR_STATE, "nalo_init_track_frame: frame slot has no pyramid");
    Initializer& I = *c->init;
    HostTimer htf(c, "init_track_frame");
    NALO_HIP(c, hipSetDevice(c->device));
    { const int rc = dev_prepare(c, I); if (rc) return rc; }
    const int maxIterations[] = {5, 5, 10, 30, 50, 50};
    I.alphaK = 2.5 * 2.5; I.alphaW = 150 * 150; I.regWeight = 0.8; I.couplingWeight = 1;
    if (!I.snapped) {
        I.thisToNext.m[3] = I.thisToNext.m[7] = I.thisToNext.m[11] = 0;
        for (int lvl = 0; lvl < I.levels; ++lvl) {
            InitLevel& P = I.L[lvl];
            std::fill(P.iR.begin(), P.iR.end(), 1.f); std::fill(P.idepth_new.begin(), P.idepth_new.end(), 1.f); std::fill(P.lastHessian.begin(), P.lastHessian.end(), 0.f);
        }
    }
    SE3 T_cur = I.thisToNext;
    double aff_cur[2] = {I.aff[0], I.aff[1]};
    if (exposure_first > 0 && exposure_new > 0) { aff_cur[0] = logf(exposure_new / exposure_first); aff_cur[1] = 0; }      // coarse approximation (:123-124)
    const float wM[8] = {kScaleXiRot, kScaleXiRot, kScaleXiRot, kScaleXiTrans, kScaleXiTrans, kScaleXiTrans, kScaleA, kScaleB};   // :64-67, labels as the reference has them
    for (int lvl = I.levels - 1; lvl >= 0; --lvl) {
        double H[64], b[8], Hsc[64], bsc[8]; float resOld[3];
        {
            HostTimer hp(c, "init.propagate");
            if (lvl < I.levels - 1) propagate_down(I, lvl + 1);
            reset_points(I, lvl);
        }
        int rc = level_upload(c, I, lvl); if (rc) return rc;
        float regEnergy[3];
        rc = calc(c, I, lvl, slot_new, T_cur, aff_cur, H, b, Hsc, bsc, resOld, regEnergy); if (rc) return rc;
        rc = apply_step(c, I, lvl); if (rc) return rc;
        float lambda = 0.1f; const float eps = 1e-4f; int fails = 0, iteration = 0;
        InitLevel& P = I.L[lvl];
        for (;;) {
            float Hl[64], bl[8];
            for (int i = 0; i < 64; ++i) Hl[i] = (float)H[i];
            for (int i = 0; i < 8; ++i) Hl[i * 8 + i] *= (1 + lambda);
            for (int i = 0; i < 64; ++i) Hl[i] -= (float)Hsc[i] * (1 / (1 + lambda));
            for (int i = 0; i < 8; ++i) bl[i] = (float)b[i] - (float)bsc[i] * (1 / (1 + lambda));
            const float sc = (0.01f / (c->wl[lvl] * c->hl[lvl]));
            for (int i = 0; i < 8; ++i) for (int j = 0; j < 8; ++j) Hl[i * 8 + j] = wM[i] * Hl[i * 8 + j] * wM[j] * sc;
            for (int i = 0; i < 8; ++i) bl[i] = wM[i] * bl[i] * sc;
            float inc[8];
            if (I.fixAffine) {
                float H6[36], x6[6];
                for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) H6[i * 6 + j] = Hl[i * 8 + j];
                ldlt_f32<6>(H6, bl, x6);
                for (int i = 0; i < 6; ++i) inc[i] = -(wM[i] * x6[i]);
                inc[6] = inc[7] = 0;
            } else {
                float x8[8]; ldlt_f32<8>(Hl, bl, x8);
                for (int i = 0; i < 8; ++i) inc[i] = -(wM[i] * x8[i]);
            }
            double xi[6]; for (int i = 0; i < 6; ++i) xi[i] = inc[i];
            const SE3 T_new = se3_exp(xi) * T_cur;
            const double aff_new[2] = {aff_cur[0] + inc[6], aff_cur[1] + inc[7]};
            {   // doStep (:910-938) on the resident arrays; idepth_new comes back with the evaluation that follows
                const LvlOff o = lvl_off((size_t)P.n); float* d = I.lvl_dev[lvl].p;
                rc = init_do_step_launch(c, P.n, (const uint8_t*)(d + o.isGood), I.jb_dev[I.jb_cur].p, d + o.maxstep, d + o.idepth, lambda, inc, d + o.idepth_new); if (rc) return rc;
            }
            double Hn[64], bn[8], Hscn[64], bscn[8]; float resNew[3];
            rc = calc(c, I, lvl, slot_new, T_new, aff_new, Hn, bn, Hscn, bscn, resNew, regEnergy); if (rc) return rc;
            const float eTotalNew = (resNew[0] + resNew[1] + regEnergy[1]);
            const float eTotalOld = (resOld[0] + resOld[1] + regEnergy[0]);
            if (eTotalOld > eTotalNew) {
                if (resNew[1] == I.alphaK * P.n) I.snapped = true;
                std::memcpy(H, Hn, sizeof(H)); std::memcpy(b, bn, sizeof(b)); std::memcpy(Hsc, Hscn, sizeof(Hsc)); std::memcpy(bsc, bscn, sizeof(bsc));
                resOld[0] = resNew[0]; resOld[1] = resNew[1]; resOld[2] = resNew[2];
                aff_cur[0] = aff_new[0]; aff_cur[1] = aff_new[1]; T_cur = T_new;
                rc = apply_step(c, I, lvl); if (rc) return rc;
                { HostTimer ho(c, "init.opt_reg"); opt_reg(I, lvl); }
                rc = iR_upload(c, I, lvl); if (rc) return rc;
                lambda *= 0.5; fails = 0;
                if (lambda < 0.0001) lambda = 0.0001;
            } else {
                ++fails; lambda *= 4;
                if (lambda > 10000) lambda = 10000;
            }
            float nrm = 0; for (int i = 0; i < 8; ++i) nrm += inc[i] * inc[i]; nrm = sqrtf(nrm);
            if (!(nrm > eps) || iteration >= maxIterations[lvl] || fails >= 2) break;
            ++iteration;
        }
        rc = level_download(c, I, lvl); if (rc) return rc;
    }
    I.thisToNext = T_cur; I.aff[0] = aff_cur[0]; I.aff[1] = aff_cur[1];
    { HostTimer hp(c, "init.propagate"); for (int i = 0; i < I.levels - 1; ++i) propagate_up(I, i); }
    ++I.frameID;
    if (!I.snapped) I.snappedAt = 0;
    if (I.snapped && I.snappedAt == 0) I.snappedAt = I.frameID;
    *ok = (I.snapped && I.frameID > I.snappedAt + 5) ? 1 : 0;
    return NALO_OK;
}

int nalo_init_get_state(nalo_ctx* c, double thisToNext[12], double aff[2], int* snapped, int* frameID, int* snappedAt, int* n_evals) {
    if (!c || !c->init) return fail(c, c ? NALO_ERR_STATE : NALO_ERR_ARG, "nalo_init_get_state: no initialiser");
    const Initializer& I = *c->init;
    if (thisToNext) std::memcpy(thisToNext, I.thisToNext.m, sizeof(I.thisToNext.m));
    if (aff) { aff[0] = I.aff[0]; aff[1] = I.aff[1]; }
    if (snapped) *snapped = I.snapped; if (frameID) *frameID = I.frameID; if (snappedAt) *snappedAt = I.snappedAt; if (n_evals) *n_evals = I.n_evals;
    return NALO_OK;
}

// restore of the carried state (what trackFrame reads from the previous frame): checkpoint / resume of an initialisation, and the teacher-forced parity runs
int nalo_init_set_state(nalo_ctx* c, const double thisToNext[12], const double aff[2], int snapped, int frameID, int snappedAt) {
    if (!c || !c->init || c->init->slot_first < 0) return fail(c, c ? NALO_ERR_STATE : NALO_ERR_ARG, "nalo_init_set_state: nalo_init_set_first has not run");
    if (!thisToNext || !aff) return fail(c, NALO_ERR_ARG, "nalo_init_set_state: bad argument");
    Initializer& I = *c->init;
    I.thisToNext = SE3::from(thisToNext); I.aff[0] = aff[0]; I.aff[1] = aff[1];
    I.snapped = snapped != 0; I.frameID = frameID; I.snappedAt = snappedAt;
    return NALO_OK;
}
int nalo_init_set_points(nalo_ctx* c, int lvl, int n, const float* idepth, const float* idepth_new, const float* iR, const uint8_t* isGood, const float* lastHessian,
                         const float* energy2, const float* maxstep, const float* lastHessian_new, const float* energy_new2, const uint8_t* isGood_new, const float* iRSumNum) {
    if (!c || !c->init || c->init->slot_first < 0) return fail(c, c ? NALO_ERR_STATE : NALO_ERR_ARG, "nalo_init_set_points: nalo_init_set_first has not run");
    if (lvl < 0 || lvl >= c->init->levels || n != c->init->L[lvl].n) return fail(c, NALO_ERR_ARG, "nalo_init_set_points: level / point count do not match the initialiser's");
    InitLevel& P = c->init->L[lvl];
    auto take = [&](auto& dst, const auto* src, size_t mult) { if (src && n) std::memcpy(dst.data(), src, (size_t)n * mult * sizeof(*src)); };
    take(P.idepth, idepth, 1); take(P.idepth_new, idepth_new, 1); take(P.iR, iR, 1); take(P.isGood, isGood, 1); take(P.lastHessian, lastHessian, 1); take(P.energy, energy2, 2);
    take(P.maxstep, maxstep, 1); take(P.lastHessian_new, lastHessian_new, 1); take(P.energy_new, energy_new2, 2); take(P.isGood_new, isGood_new, 1); take(P.iRSumNum, iRSumNum, 1);
    return NALO_OK;
}

int nalo_init_get_points(nalo_ctx* c, int lvl, int cap, int* n, float* u, float* v, float* idepth, float* iR, uint8_t* isGood, float* lastHessian, float* energy2, float* my_type,
                         float* outlierTH, int* parent, float* parentDist, int* neighbours, float* neighboursDist) {
    if (!c || !c->init) return fail(c, c ? NALO_ERR_STATE : NALO_ERR_ARG, "nalo_init_get_points: no initialiser");
    if (lvl < 0 || lvl >= c->init->levels || !n) return fail(c, NALO_ERR_ARG, "nalo_init_get_points: bad argument");
    const InitLevel& P = c->init->L[lvl];
    *n = P.n;
    const size_t m = (size_t)std::min(cap, P.n);
    auto put = [&](auto* dst, const auto& src, size_t mult) { if (dst && m) std::memcpy(dst, src.data(), m * mult * sizeof(*dst)); };
    put(u, P.u, 1); put(v, P.v, 1); put(idepth, P.idepth, 1); put(iR, P.iR, 1); put(isGood, P.isGood, 1); put(lastHessian, P.lastHessian, 1); put(energy2, P.energy, 2);
    put(my_type, P.my_type, 1); put(outlierTH, P.outlierTH, 1); put(parent, P.parent, 1); put(parentDist, P.parentDist, 1); put(neighbours, P.nn, 10); put(neighboursDist, P.nnDist, 10);
    return NALO_OK;
}
