// Native multi-GPU exchange of the sharded bundle adjustment: the library enqueues ncclAllReduce (RCCL) itself on its own streams, no
// callback into the caller's language per Gauss-Newton iteration. librccl is opened at run time (dlopen), so a single-GPU deployment needs no RCCL.
//
//   nalo_rccl_unique_id      ncclGetUniqueId (rank 0 draws two ids: main and side communicator; the caller ships them to the other ranks with whatever it
//                            has: MPI, a file, torch.distributed)
//   nalo_ba_rccl_init        ncclCommInitRank x 2 on this context's device; the communicators belong to the context
//   nalo_ba_set_rccl_comm    the same with communicators the caller owns
//
// Two communicators because two streams carry collectives at the same time: the stitched systems on the main stream, the two radix histograms of
// setNewFrameEnergyTH (FullSystemOptimize.cpp:95-143) on the side stream under the Schur-complement kernels (host_ba.hip linearize_async); RCCL
// serialises the operations of ONE communicator.
//
// nalo_shard_points is the partition itself (which points a rank keeps): contiguous Hilbert ranges of every host frame, see below.
#include "nalo_internal.h"
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <algorithm>
#include <numeric>

namespace nalo {

struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};
static RcclApi& rccl() {
    static RcclApi api = [] {
        RcclApi a;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { a.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (a.lib) break; }
        if (!a.lib) { a.err = std::string("librccl not found: ") + (dlerror() ? dlerror() : ""); return a; }
        a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(dlsym(a.lib, "ncclGetUniqueId"));
        a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(dlsym(a.lib, "ncclCommInitRank"));
        a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(a.lib, "ncclCommDestroy"));
        a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(dlsym(a.lib, "ncclAllReduce"));
        a.CommCount = reinterpret_cast<decltype(a.CommCount)>(dlsym(a.lib, "ncclCommCount"));
        a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(a.lib, "ncclGetErrorString"));
        if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllReduce) a.err = "librccl lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce";
        return a;
    }();
    return api;
}

struct RcclState { ncclComm_t main = nullptr, side = nullptr; bool owned = false; nalo_ctx* ctx = nullptr; };

static void rccl_sum(RcclState* st, ncclComm_t comm, hipStream_t stream, double* buf, int n) {
    const ncclResult_t r = rccl().AllReduce(buf, buf, (size_t)n, ncclDouble, ncclSum, comm, stream);       // in place, stream ordered
    if (r != ncclSuccess && !st->ctx->xchg_failed) {
        // the hook itself returns nothing (nalo_allreduce_fn): the latch is read by host_ba.hip right after every hook call, which then returns
        // NALO_ERR_HIP instead of solving with sums that were never reduced (the ranks would diverge silently)
        st->ctx->xchg_failed = true;
        st->ctx->err = std::string("ncclAllReduce: ") + (rccl().GetErrorString ? rccl().GetErrorString(r) : "error");
    }
}
static void rccl_hook_main(void* user, double* buf, int n) { RcclState* st = static_cast<RcclState*>(user); rccl_sum(st, st->main, st->ctx->stream, buf, n); }
static void rccl_hook_side(void* user, double* buf, int n) { RcclState* st = static_cast<RcclState*>(user); rccl_sum(st, st->side, st->ctx->side, buf, n); }

void rccl_release(nalo_ctx* c) {
    RcclState* st = static_cast<RcclState*>(c->rccl);
    if (!st) return;
    if (st->owned) { if (st->main) (void)rccl().CommDestroy(st->main); if (st->side) (void)rccl().CommDestroy(st->side); }
    delete st;
    c->rccl = nullptr;
}

}  // namespace nalo

using namespace nalo;

extern "C" {

int nalo_rccl_unique_id(char id[128]) {
    if (!id) return NALO_ERR_ARG;
    if (!rccl().err.empty()) return NALO_ERR_UNSUPPORTED;
    ncclUniqueId u;
    if (rccl().GetUniqueId(&u) != ncclSuccess) return NALO_ERR_HIP;
    static_assert(sizeof(u) == 128, "ncclUniqueId is 128 bytes");
    std::memcpy(id, &u, 128);
    return NALO_OK;
}

static int install(nalo_ctx* c, ncclComm_t comm_main, ncclComm_t comm_side, bool owned) {
    rccl_release(c);
    RcclState* st = new RcclState();
    st->main = comm_main; st->side = comm_side; st->owned = owned; st->ctx = c;
    c->rccl = st;
    int rc = nalo_ba_set_allreduce(c, rccl_hook_main, st); if (rc) return rc;
    rc = nalo_ba_set_allreduce_mode(c, 1); if (rc) return rc;                       // the hooks only ENQUEUE on the context's streams
    return nalo_ba_set_allreduce_side(c, comm_side ? rccl_hook_side : nullptr, st);
}

int nalo_ba_rccl_init(nalo_ctx* c, int nranks, int rank, const char id_main[128], const char id_side[128]) {
    if (!c || !id_main || nranks < 1 || rank < 0 || rank >= nranks) return fail(c, NALO_ERR_ARG, "nalo_ba_rccl_init: bad argument");
    if (!rccl().err.empty()) return fail(c, NALO_ERR_UNSUPPORTED, "nalo_ba_rccl_init: " + rccl().err);
    if (!c->ba) return fail(c, NALO_ERR_STATE, "nalo_ba_rccl_init: set the window first (nalo_ba_set_window)");
    NALO_HIP(c, hipSetDevice(c->device));
    ncclUniqueId um, us;
    std::memcpy(&um, id_main, 128);
    ncclComm_t cm = nullptr, cs = nullptr;
    ncclResult_t r = rccl().CommInitRank(&cm, nranks, um, rank);
    if (r != ncclSuccess) return fail(c, NALO_ERR_HIP, std::string("ncclCommInitRank (main): ") + (rccl().GetErrorString ? rccl().GetErrorString(r) : "error"));
    if (id_side) {
        std::memcpy(&us, id_side, 128);
        r = rccl().CommInitRank(&cs, nranks, us, rank);
        if (r != ncclSuccess) { (void)rccl().CommDestroy(cm); return fail(c, NALO_ERR_HIP, std::string("ncclCommInitRank (side): ") + (rccl().GetErrorString ? rccl().GetErrorString(r) : "error")); }
    }
    return install(c, cm, cs, true);
}

int nalo_ba_set_rccl_comm(nalo_ctx* c, void* comm_main, void* comm_side) {
    if (!c) return NALO_ERR_ARG;
    if (!comm_main) {                                                               // back to a single GPU
        rccl_release(c);
        int rc = nalo_ba_set_allreduce(c, nullptr, nullptr); if (rc) return rc;
        return nalo_ba_set_allreduce_side(c, nullptr, nullptr);
    }
    if (!rccl().err.empty()) return fail(c, NALO_ERR_UNSUPPORTED, "nalo_ba_set_rccl_comm: " + rccl().err);
    if (!c->ba) return fail(c, NALO_ERR_STATE, "nalo_ba_set_rccl_comm: set the window first (nalo_ba_set_window)");
    return install(c, static_cast<ncclComm_t>(comm_main), static_cast<ncclComm_t>(comm_side), false);
}

// How many ranks the context's communicators really span, read back from RCCL (ncclCommCount): what a launcher prints next to the number of processes it
// started, so that a job whose ranks never joined one communicator cannot pass for an N-GPU run. 0 = no communicator installed.
int nalo_ba_rccl_ranks(nalo_ctx* c, int* ranks_main, int* ranks_side) {
    if (!c) return NALO_ERR_ARG;
    if (ranks_main) *ranks_main = 0;
    if (ranks_side) *ranks_side = 0;
    RcclState* st = static_cast<RcclState*>(c->rccl);
    if (!st) return NALO_OK;
    if (!rccl().CommCount) return fail(c, NALO_ERR_UNSUPPORTED, "nalo_ba_rccl_ranks: librccl lacks ncclCommCount");
    int n = 0;
    if (st->main) { if (rccl().CommCount(st->main, &n) != ncclSuccess) return fail(c, NALO_ERR_HIP, "ncclCommCount (main) failed"); if (ranks_main) *ranks_main = n; }
    if (st->side) { if (rccl().CommCount(st->side, &n) != ncclSuccess) return fail(c, NALO_ERR_HIP, "ncclCommCount (side) failed"); if (ranks_side) *ranks_side = n; }
    return NALO_OK;
}

// The partition of the active-point set over `world` ranks: points keep all their residuals, frames are replicated. Every rank gets the same
// share of EVERY host frame, so all (host, target) bins stay evenly populated, and within a host the share is a contiguous range of the host's points in
// Hilbert order of their 8x8-pixel cells — the order nalo_ba_set_points sorts by —, i.e. a spatially compact part of the image: the texels a rank gathers
// then have the reuse of the unsharded window instead of a 1/N-density sample of every image (a block-cyclic shard ran ba_linearize 1.8x slower per
// residual at N = 8). keep[] receives the indices (ascending) of the points of `rank`; returns their number, or < 0 on a bad argument.
int nalo_shard_points(int P, int W, const int* host, const float* u, const float* v, int img_w, int img_h, int rank, int world, int* keep) {
    if (P < 0 || W < 1 || !host || !u || !v || !keep || world < 1 || rank < 0 || rank >= world) return NALO_ERR_ARG;
    unsigned hn = 1; while ((int)hn * 8 < std::max(img_w, img_h)) hn <<= 1;
    auto hilbert = [hn](unsigned x, unsigned y) {
        unsigned long long d = 0;
        for (unsigned s = hn / 2; s > 0; s /= 2) {
            const unsigned rx = (x & s) ? 1u : 0u, ry = (y & s) ? 1u : 0u;
            d += (unsigned long long)s * s * ((3u * rx) ^ ry);
            if (ry == 0) { if (rx == 1) { x = hn - 1 - x; y = hn - 1 - y; } const unsigned tmp = x; x = y; y = tmp; }
        }
        return d;
    };
    std::vector<unsigned long long> key(P);
    std::vector<int> order(P), cnt(W, 0);
    for (int p = 0; p < P; ++p) {
        if (host[p] < 0 || host[p] >= W) return NALO_ERR_ARG;
        cnt[host[p]]++;
        key[p] = ((unsigned long long)host[p] << 40) | hilbert(std::min(hn - 1, (unsigned)std::max(0.f, u[p]) >> 3), std::min(hn - 1, (unsigned)std::max(0.f, v[p]) >> 3));
    }
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return key[a] < key[b]; });
    int n = 0, base = 0;
    for (int h = 0; h < W; ++h) {                               // numpy.array_split semantics: the first (cnt % world) ranks get one point more
        const int c = cnt[h], q = c / world, r = c % world;
        const int lo = rank * q + std::min(rank, r), hi = lo + q + (rank < r ? 1 : 0);
        for (int k = lo; k < hi; ++k) keep[n++] = order[base + k];
        base += c;
    }
    std::sort(keep, keep + n);
    return n;
}

}  // extern "C"
