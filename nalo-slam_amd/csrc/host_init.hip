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
// What runs where: everything per point is a kernel. The two passes that touch the images (calcResAndGS, which also forms calcEC's sums, and doStep), applyStep,
// propagateDown / propagateUp (one lane per point / per parent, children summed in index order as the reference's fp32 += does) and resetPoints are plain
// parallel kernels; optReg and the top level's resetPoints are Gauss-Seidel sweeps in the reference (point i reads what its lower-index neighbours were just
// given) and run as dependency-ordered kernels on a schedule this file builds once per setFirst (build_sweep below, kernels_init.hip). makeNN's result depends on
// the traversal order of nanoflann's k-d tree wherever neighbours are equidistant (points sit on the integer grid + 0.1, so most 10-NN sets end in a tie): the
// trees are built on the host in the reference's order, one level per thread, and queried from a few host threads.
//
// Residency: the Pnt arrays (SoA) and both JbBuffers live on the DEVICE for the whole of trackFrame (InitLevel's device block, lvl_off()); one evaluation moves 94
// doubles to the host, where the Levenberg-Marquardt bookkeeping (8x8 solve, accept / reject, lambda) runs. The host mirror of the arrays is refreshed lazily
// (nalo_init_get_points / nalo_init_set_points) and uploaded when it is the newer side (after setFirst or nalo_init_set_points).
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
    DevBuf<int> tab_dev[NALO_MAX_LEVELS];                // per level: the sweep schedule, parent, the children lists (TabOff)
    DevBuf<float> sweep_scratch;                         // 2 x max n floats (padded): the sweep's values | the inverse depths in schedule order
    int nsteps[NALO_MAX_LEVELS] = {};
    bool static_on_dev = false;                          // u, v, outlierTH and the tables are uploaded
    bool dev_valid = false, host_valid = true;           // which side holds the current Pnt state
    int jb_cur = 0;
    float* pin = nullptr; size_t pin_half = 0;           // pinned staging: [0, pin_half) host -> device, [pin_half, 2 pin_half) device -> host
    double* sums_host = nullptr; double* sums_dev = nullptr; double sums_seq = 0;    // an evaluation's 94 sums land in mapped host memory, sequence number last: polled, no copy, no sync
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
    for (auto& b : c->init->tab_dev) b.release();
    c->init->sweep_scratch.release();
    for (auto& b : c->init->jb_dev) b.release();
    if (c->init->pin) (void)hipHostFree(c->init->pin);
    if (c->init->sums_host) (void)hipHostFree(c->init->sums_host);
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

// ---------------------------------------------------------------------------------------------------------------- sweep schedule and tables of one level
// int words of a level's table block: rec [n][12] = {point, its 10 neighbours (n where there is none), -} in schedule order (16-byte records first), off [nsteps + 1], parent [n],
// child_off [n + 1] / child_idx [n_below]: the points of the level below grouped by parent, in index order
struct TabOff { size_t rec, off, parent, child_off, child_idx, total; };
static TabOff tab_off(size_t n, size_t nsteps, size_t n_below) {
    TabOff o; o.rec = 0; o.off = 12 * n; o.parent = o.off + nsteps + 1; o.child_off = o.parent + n; o.child_idx = o.child_off + n + 1; o.total = o.child_idx + n_below + 4;
    return o;
}
// The order optReg / resetPoints impose: i before j (i < j) whenever one is among the other's neighbours. step[i] = 1 + max step of the points ordered before i;
// the points of a step are independent. Steps wider than kSweepNT are cut. Returns off (nsteps + 1 entries) and the points in schedule order.
static void build_sweep(const InitLevel& P, std::vector<int>& off, std::vector<int>& order) {
    const int n = P.n;
    std::vector<int> step(n, 0), pending(n, 0);
    int depth = 0;
    for (int i = 0; i < n; ++i) {
        int st = pending[i];
        for (int k = 0; k < 10; ++k) { const int j = P.nn[(size_t)i * 10 + k]; if (j >= 0 && j < i) st = std::max(st, step[j] + 1); }
        step[i] = st; depth = std::max(depth, st + 1);
        for (int k = 0; k < 10; ++k) { const int j = P.nn[(size_t)i * 10 + k]; if (j > i) pending[j] = std::max(pending[j], st + 1); }
    }
    std::vector<int> cnt(depth + 1, 0);
    for (int i = 0; i < n; ++i) ++cnt[step[i] + 1];
    for (int d = 0; d < depth; ++d) cnt[d + 1] += cnt[d];
    order.assign(n, 0);
    { std::vector<int> at(cnt.begin(), cnt.end() - 1); for (int i = 0; i < n; ++i) order[at[step[i]]++] = i; }
    off.clear();
    for (int d = 0; d < depth; ++d) for (int p = cnt[d]; p < cnt[d + 1]; p += kSweepNT) off.push_back(p);
    off.push_back(n);
}

static int dev_prepare(nalo_ctx* c, Initializer& I) {                                          // device blocks, tables, pinned staging for the current point counts
    size_t maxn = 0, maxtot = 0;
    for (int l = 0; l < I.levels; ++l) { const LvlOff o = lvl_off((size_t)I.L[l].n); NALO_HIP(c, I.lvl_dev[l].reserve(o.total)); maxn = std::max(maxn, (size_t)I.L[l].n); maxtot = std::max(maxtot, o.total); }
    for (auto& b : I.jb_dev) NALO_HIP(c, b.reserve(10 * maxn + 16));
    NALO_HIP(c, I.sweep_scratch.reserve(2 * (maxn + 8) + 32));
    if (I.pin_half < maxtot) {
        if (I.pin) (void)hipHostFree(I.pin);
        I.pin = nullptr; I.pin_half = 0;
        NALO_HIP(c, hipHostMalloc((void**)&I.pin, 2 * maxtot * sizeof(float)));
        I.pin_half = maxtot;
    }
    if (!I.sums_host) {
        NALO_HIP(c, hipHostMalloc((void**)&I.sums_host, 96 * sizeof(double), hipHostMallocMapped));
        std::memset(I.sums_host, 0, 96 * sizeof(double));
        NALO_HIP(c, hipHostGetDevicePointer((void**)&I.sums_dev, I.sums_host, 0));
    }
    if (I.static_on_dev) return NALO_OK;
    HostTimer ht(c, "init.tables");
    std::vector<int> off, order, tab;
    for (int l = 0; l < I.levels; ++l) {
        const InitLevel& P = I.L[l]; const size_t n = (size_t)P.n; const LvlOff o = lvl_off(n);
        I.nsteps[l] = 0;
        if (!n) continue;
        std::memcpy(I.pin + o.u, P.u.data(), n * 4); std::memcpy(I.pin + o.v, P.v.data(), n * 4); std::memcpy(I.pin + o.outlierTH, P.outlierTH.data(), n * 4);
        NALO_HIP(c, hipMemcpyAsync(I.lvl_dev[l].p, I.pin, 3 * n * 4, hipMemcpyHostToDevice, c->stream));
        build_sweep(P, off, order);
        const size_t ns = off.size() - 1, n_below = l > 0 ? (size_t)I.L[l - 1].n : 0;
        I.nsteps[l] = (int)ns;
        const TabOff t = tab_off(n, ns, n_below);
        tab.assign(t.total, 0);
        for (size_t p = 0; p < n; ++p) { int* r = &tab[t.rec + 12 * p]; r[0] = order[p]; for (int k = 0; k < 10; ++k) { const int j = P.nn[(size_t)order[p] * 10 + k]; r[1 + k] = j < 0 ? (int)n : j; } r[11] = 0; }
        std::memcpy(&tab[t.off], off.data(), (ns + 1) * 4);
        std::memcpy(&tab[t.parent], P.parent.data(), n * 4);
        if (l > 0) {                                                                            // counting sort of the level below by parent (stable: index order inside a parent)
            const InitLevel& B = I.L[l - 1];
            int* co = &tab[t.child_off];
            for (int i = 0; i < B.n; ++i) if (B.parent[i] >= 0) ++co[B.parent[i] + 1];
            for (size_t q = 0; q < n; ++q) co[q + 1] += co[q];
            std::vector<int> at(co, co + n);
            for (int i = 0; i < B.n; ++i) if (B.parent[i] >= 0) tab[t.child_idx + at[B.parent[i]]++] = i;
        }
        NALO_HIP(c, I.tab_dev[l].reserve(t.total));
        NALO_HIP(c, hipMemcpyAsync(I.tab_dev[l].p, tab.data(), t.total * 4, hipMemcpyHostToDevice, c->stream));
        NALO_HIP(c, hipStreamSynchronize(c->stream));                                           // pin / tab are reused by the next level
    }
    I.static_on_dev = true; I.dev_valid = false;
    return NALO_OK;
}
// host mirror -> device, every member trackFrame reads or keeps (the *_new members hold stale entries that survive, as in the reference)
static int level_upload(nalo_ctx* c, Initializer& I, int lvl) {
    const InitLevel& P = I.L[lvl]; const size_t n = (size_t)P.n; const LvlOff o = lvl_off(n);
    if (!n) return NALO_OK;
    float* h = I.pin - o.dyn;
    std::memcpy(h + o.idepth, P.idepth.data(), n * 4); std::memcpy(h + o.iR, P.iR.data(), n * 4); std::memcpy(h + o.lastHessian, P.lastHessian.data(), n * 4);
    std::memcpy(h + o.lastHessian_new, P.lastHessian_new.data(), n * 4); std::memcpy(h + o.maxstep, P.maxstep.data(), n * 4); std::memcpy(h + o.energy, P.energy.data(), 2 * n * 4);
    std::memcpy(h + o.energy_new, P.energy_new.data(), 2 * n * 4); std::memcpy(h + o.isGood, P.isGood.data(), n); std::memcpy(h + o.idepth_new, P.idepth_new.data(), n * 4);
    std::memcpy(h + o.isGood_new, P.isGood_new.data(), n);
    NALO_HIP(c, hipMemcpyAsync(I.lvl_dev[lvl].p + o.dyn, I.pin, (o.dyn_end - o.dyn) * 4, hipMemcpyHostToDevice, c->stream));
    NALO_HIP(c, hipStreamSynchronize(c->stream));                                               // the staging block is reused by the next level
    return NALO_OK;
}
static int level_download(nalo_ctx* c, Initializer& I, int lvl) {
    InitLevel& P = I.L[lvl]; const size_t n = (size_t)P.n; const LvlOff o = lvl_off(n);
    if (!n) return NALO_OK;
    float* hd = I.pin + I.pin_half;
    NALO_HIP(c, hipMemcpyAsync(hd, I.lvl_dev[lvl].p + o.dyn, (o.dyn_end - o.dyn) * 4, hipMemcpyDeviceToHost, c->stream));
    NALO_HIP(c, hipStreamSynchronize(c->stream));
    const float* h = hd - o.dyn;
    std::memcpy(P.idepth.data(), h + o.idepth, n * 4); std::memcpy(P.iR.data(), h + o.iR, n * 4); std::memcpy(P.lastHessian.data(), h + o.lastHessian, n * 4);
    std::memcpy(P.lastHessian_new.data(), h + o.lastHessian_new, n * 4);
    std::memcpy(P.maxstep.data(), h + o.maxstep, n * 4); std::memcpy(P.energy.data(), h + o.energy, 2 * n * 4); std::memcpy(P.energy_new.data(), h + o.energy_new, 2 * n * 4);
    std::memcpy(P.isGood.data(), h + o.isGood, n); std::memcpy(P.idepth_new.data(), h + o.idepth_new, n * 4); std::memcpy(P.isGood_new.data(), h + o.isGood_new, n);
    return NALO_OK;
}
// make the device the current side (upload the mirror if it is newer) / make the host mirror current
static int dev_sync(nalo_ctx* c, Initializer& I) {
    int rc = dev_prepare(c, I); if (rc) return rc;
    if (!I.dev_valid) {
        HostTimer ht(c, "init.mirror_io");
        for (int l = 0; l < I.levels; ++l) { rc = level_upload(c, I, l); if (rc) return rc; }
        I.dev_valid = true;
    }
    return NALO_OK;
}
static int host_sync(nalo_ctx* c, Initializer& I) {
    if (I.host_valid) return NALO_OK;
    HostTimer ht(c, "init.mirror_io");
    NALO_HIP(c, hipSetDevice(c->device));
    for (int l = 0; l < I.levels; ++l) { const int rc = level_download(c, I, l); if (rc) return rc; }
    I.host_valid = true;
    return NALO_OK;
}

struct LvlPtr { float *idepth, *idepth_new, *iR, *lastHessian, *lastHessian_new, *maxstep, *energy, *energy_new; uint8_t *isGood, *isGood_new; };
static LvlPtr lvl_ptr(Initializer& I, int lvl) {
    const LvlOff o = lvl_off((size_t)I.L[lvl].n); float* d = I.lvl_dev[lvl].p;
    return LvlPtr{d + o.idepth, d + o.idepth_new, d + o.iR, d + o.lastHessian, d + o.lastHessian_new, d + o.maxstep, d + o.energy, d + o.energy_new,
                  (uint8_t*)(d + o.isGood), (uint8_t*)(d + o.isGood_new)};
}
static int sweep(nalo_ctx* c, Initializer& I, int lvl, int mode) {
    const InitLevel& P = I.L[lvl];
    if (!P.n) return NALO_OK;
    const TabOff t = tab_off((size_t)P.n, (size_t)I.nsteps[lvl], lvl > 0 ? (size_t)I.L[lvl - 1].n : 0);
    const LvlPtr q = lvl_ptr(I, lvl);
    const int* tab = I.tab_dev[lvl].p;
    return init_sweep_launch(c, mode, P.n, I.nsteps[lvl], tab + t.off, tab + t.rec, q.iR, q.isGood, q.idepth, q.idepth_new, I.regWeight, I.sweep_scratch.p);
}
static int opt_reg(nalo_ctx* c, Initializer& I, int lvl) {                                     // optReg :656-691
    if (!I.snapped) return init_fill_launch(c, I.L[lvl].n, lvl_ptr(I, lvl).iR, nullptr, nullptr);
    return sweep(c, I, lvl, 0);
}
static int propagate_up(nalo_ctx* c, Initializer& I, int src) {                                // propagateUp :695-734
    const int nT = I.L[src + 1].n;
    if (nT) {
        const TabOff t = tab_off((size_t)nT, (size_t)I.nsteps[src + 1], (size_t)I.L[src].n);
        const LvlPtr S = lvl_ptr(I, src), T = lvl_ptr(I, src + 1);
        const int* tab = I.tab_dev[src + 1].p;
        const int rc = init_propagate_up_launch(c, nT, tab + t.child_off, tab + t.child_idx, S.isGood, S.iR, S.lastHessian, T.isGood, T.iR, T.idepth); if (rc) return rc;
    }
    return opt_reg(c, I, src + 1);
}
static int propagate_down(nalo_ctx* c, Initializer& I, int src) {                              // propagateDown :736-766
    const int n = I.L[src - 1].n;
    if (n && I.L[src].n) {
        const TabOff t = tab_off((size_t)n, (size_t)I.nsteps[src - 1], src - 1 > 0 ? (size_t)I.L[src - 2].n : 0);
        const LvlPtr S = lvl_ptr(I, src), T = lvl_ptr(I, src - 1);
        const int rc = init_propagate_down_launch(c, n, I.tab_dev[src - 1].p + t.parent, S.isGood, S.lastHessian, S.iR, T.isGood, T.iR, T.idepth, T.idepth_new, T.lastHessian); if (rc) return rc;
    }
    return opt_reg(c, I, src - 1);
}
static int reset_points(nalo_ctx* c, Initializer& I, int lvl) {                                // resetPoints :882-909
    const LvlPtr q = lvl_ptr(I, lvl);
    const int rc = init_reset_launch(c, I.L[lvl].n, q.energy, q.idepth_new, q.idepth); if (rc) return rc;
    return lvl == I.levels - 1 ? sweep(c, I, lvl, 1) : NALO_OK;
}

// calcResAndGS on the resident level (writes JbBuffer_new = jb_dev[1 - jb_cur]); the 94 sums come back, regEnergy = calcEC (:634-655)
static int calc(nalo_ctx* c, Initializer& I, int lvl, int slot_new, const SE3& T, const double aff[2], double H[64], double b[8], double Hsc[64], double bsc[8], float res[3], float regEnergy[3]) {
    HostTimer ht(c, "init.calc");
    InitLevel& L = I.L[lvl]; const size_t n = (size_t)L.n; const LvlOff o = lvl_off(n);
    InitParams P; InitPose X;
    init_pose_setup(c, lvl, L.n, T, aff, I.alphaW, I.alphaK, I.couplingWeight, P, X);
    double sums[96] = {};
    if (n) {
        float* d = I.lvl_dev[lvl].p;
        const LvlPtr q = lvl_ptr(I, lvl);
        P.colorRef = c->slots[I.slot_first].dI[lvl]; P.colorNew = c->slots[slot_new].dI[lvl];
        P.u = d + o.u; P.v = d + o.v; P.outlierTH = d + o.outlierTH; P.idepth = q.idepth; P.idepth_new = q.idepth_new; P.iR = q.iR; P.energy = q.energy;
        P.isGood = q.isGood; P.isGood_new = q.isGood_new; P.energy_new = q.energy_new; P.maxstep = q.maxstep; P.lastHessian_new = q.lastHessian_new; P.Jb = I.jb_dev[1 - I.jb_cur].p;
        I.sums_seq += 1;
        int rc = init_calc_launch(c, P, lvl, I.sums_dev, 1, I.sums_seq); if (rc) return rc;
        if (!poll_flag(c, &I.sums_host[95], I.sums_seq)) return NALO_ERR_HIP;
        std::memcpy(sums, I.sums_host, 94 * 8);
    }
    double E3[3];
    init_sums_to_system(sums, T, L.n, P, X, H, b, Hsc, bsc, E3);
    res[0] = (float)E3[0]; res[1] = (float)E3[1]; res[2] = (float)E3[2];
    if (!I.snapped) { regEnergy[0] = 0; regEnergy[1] = 0; regEnergy[2] = (float)L.n; }
    else { regEnergy[0] = I.couplingWeight * (float)sums[91]; regEnergy[1] = I.couplingWeight * (float)sums[92]; regEnergy[2] = (float)sums[93]; }
    ++I.n_evals;
    return NALO_OK;
}
static int apply_step(nalo_ctx* c, Initializer& I, int lvl) {                                  // applyStep :939-956; the JbBuffer swap is jb_cur
    const LvlPtr q = lvl_ptr(I, lvl);
    const int rc = init_apply_step_launch(c, I.L[lvl].n, q.isGood, q.isGood_new, q.idepth, q.idepth_new, q.iR, q.energy, q.energy_new, q.lastHessian, q.lastHessian_new);
    if (rc) return rc;
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
    I.static_on_dev = false; I.dev_valid = false; I.host_valid = true; I.jb_cur = 0;
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
    { const int rc = dev_sync(c, I); if (rc) return rc; }
    I.host_valid = false;
    const int maxIterations[] = {5, 5, 10, 30, 50, 50};
    I.alphaK = 2.5 * 2.5; I.alphaW = 150 * 150; I.regWeight = 0.8; I.couplingWeight = 1;
    if (!I.snapped) {
        I.thisToNext.m[3] = I.thisToNext.m[7] = I.thisToNext.m[11] = 0;
        for (int lvl = 0; lvl < I.levels; ++lvl) { const LvlPtr q = lvl_ptr(I, lvl); const int rc = init_fill_launch(c, I.L[lvl].n, q.iR, q.idepth_new, q.lastHessian); if (rc) return rc; }
    }
    SE3 T_cur = I.thisToNext;
    double aff_cur[2] = {I.aff[0], I.aff[1]};
    if (exposure_first > 0 && exposure_new > 0) { aff_cur[0] = logf(exposure_new / exposure_first); aff_cur[1] = 0; }      // coarse approximation (:123-124)
    const float wM[8] = {kScaleXiRot, kScaleXiRot, kScaleXiRot, kScaleXiTrans, kScaleXiTrans, kScaleXiTrans, kScaleA, kScaleB};   // :64-67, labels as the reference has them
    for (int lvl = I.levels - 1; lvl >= 0; --lvl) {
        double H[64], b[8], Hsc[64], bsc[8]; float resOld[3], regEnergy[3];
        int rc = NALO_OK;
        if (lvl < I.levels - 1) { rc = propagate_down(c, I, lvl + 1); if (rc) return rc; }
        rc = reset_points(c, I, lvl); if (rc) return rc;
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
            {   // doStep (:910-938)
                const LvlPtr q = lvl_ptr(I, lvl);
                rc = init_do_step_launch(c, P.n, q.isGood, I.jb_dev[I.jb_cur].p, q.maxstep, q.idepth, lambda, inc, q.idepth_new); if (rc) return rc;
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
                rc = opt_reg(c, I, lvl); if (rc) return rc;
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
    }
    I.thisToNext = T_cur; I.aff[0] = aff_cur[0]; I.aff[1] = aff_cur[1];
    for (int i = 0; i < I.levels - 1; ++i) { const int rc = propagate_up(c, I, i); if (rc) return rc; }
    NALO_HIP(c, hipStreamSynchronize(c->stream));
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
    { const int rc = host_sync(c, *c->init); if (rc) return rc; }
    c->init->dev_valid = false;
    InitLevel& P = c->init->L[lvl];
    auto take = [&](auto& dst, const auto* src, size_t mult) { if (src && n) std::memcpy(dst.data(), src, (size_t)n * mult * sizeof(*src)); };
    take(P.idepth, idepth, 1); take(P.idepth_new, idepth_new, 1); take(P.iR, iR, 1); take(P.isGood, isGood, 1); take(P.lastHessian, lastHessian, 1); take(P.energy, energy2, 2);
    take(P.maxstep, maxstep, 1); take(P.lastHessian_new, lastHessian_new, 1); take(P.energy_new, energy_new2, 2); take(P.isGood_new, isGood_new, 1); take(P.iRSumNum, iRSumNum, 1);
    return NALO_OK;
}

int nalo_init_get_carried(nalo_ctx* c, int lvl, int cap, float* idepth_new, float* maxstep, float* lastHessian_new, float* energy_new2, uint8_t* isGood_new) {
    if (!c || !c->init) return fail(c, c ? NALO_ERR_STATE : NALO_ERR_ARG, "nalo_init_get_carried: no initialiser");
    if (lvl < 0 || lvl >= c->init->levels || cap < 0) return fail(c, NALO_ERR_ARG, "nalo_init_get_carried: bad argument");
    { const int rc = host_sync(c, *c->init); if (rc) return rc; }
    const InitLevel& P = c->init->L[lvl];
    const size_t m = (size_t)std::min(cap, P.n);
    auto put = [&](auto* dst, const auto& src, size_t mult) { if (dst && m) std::memcpy(dst, src.data(), m * mult * sizeof(*dst)); };
    put(idepth_new, P.idepth_new, 1); put(maxstep, P.maxstep, 1); put(lastHessian_new, P.lastHessian_new, 1); put(energy_new2, P.energy_new, 2); put(isGood_new, P.isGood_new, 1);
    return NALO_OK;
}

// one of trackFrame's sweeps on its own, on the resident arrays (the parity tests compare each with the oracle's on identical state)
int nalo_init_sweep(nalo_ctx* c, int which, int lvl) {
    if (!c || !c->init || c->init->slot_first < 0) return fail(c, c ? NALO_ERR_STATE : NALO_ERR_ARG, "nalo_init_sweep: nalo_init_set_first has not run");
    Initializer& I = *c->init;
    const bool ok = which == NALO_INIT_SWEEP_PROPAGATE_UP ? (lvl >= 0 && lvl < I.levels - 1) : which == NALO_INIT_SWEEP_PROPAGATE_DOWN ? (lvl >= 1 && lvl < I.levels) : (lvl >= 0 && lvl < I.levels);
    if (!ok) return fail(c, NALO_ERR_ARG, "nalo_init_sweep: level out of range");
    NALO_HIP(c, hipSetDevice(c->device));
    int rc = dev_sync(c, I); if (rc) return rc;
    I.host_valid = false;
    I.regWeight = 0.8;
    switch (which) {
        case NALO_INIT_SWEEP_OPT_REG: rc = opt_reg(c, I, lvl); break;
        case NALO_INIT_SWEEP_PROPAGATE_UP: rc = propagate_up(c, I, lvl); break;
        case NALO_INIT_SWEEP_PROPAGATE_DOWN: rc = propagate_down(c, I, lvl); break;
        case NALO_INIT_SWEEP_RESET_POINTS: rc = reset_points(c, I, lvl); break;
        default: return fail(c, NALO_ERR_ARG, "nalo_init_sweep: unknown sweep");
    }
    if (rc) return rc;
    NALO_HIP(c, hipStreamSynchronize(c->stream));
    return NALO_OK;
}

int nalo_init_get_points(nalo_ctx* c, int lvl, int cap, int* n, float* u, float* v, float* idepth, float* iR, uint8_t* isGood, float* lastHessian, float* energy2, float* my_type,
                         float* outlierTH, int* parent, float* parentDist, int* neighbours, float* neighboursDist) {
    if (!c || !c->init) return fail(c, c ? NALO_ERR_STATE : NALO_ERR_ARG, "nalo_init_get_points: no initialiser");
    if (lvl < 0 || lvl >= c->init->levels || !n) return fail(c, NALO_ERR_ARG, "nalo_init_get_points: bad argument");
    { const int rc = host_sync(c, *c->init); if (rc) return rc; }
    const InitLevel& P = c->init->L[lvl];
    *n = P.n;
    const size_t m = (size_t)std::min(cap, P.n);
    auto put = [&](auto* dst, const auto& src, size_t mult) { if (dst && m) std::memcpy(dst, src.data(), m * mult * sizeof(*dst)); };
    put(u, P.u, 1); put(v, P.v, 1); put(idepth, P.idepth, 1); put(iR, P.iR, 1); put(isGood, P.isGood, 1); put(lastHessian, P.lastHessian, 1); put(energy2, P.energy, 2);
    put(my_type, P.my_type, 1); put(outlierTH, P.outlierTH, 1); put(parent, P.parent, 1); put(parentDist, P.parentDist, 1); put(neighbours, P.nn, 10); put(neighboursDist, P.nnDist, 10);
    return NALO_OK;
}
