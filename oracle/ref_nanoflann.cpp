/*
 * oracle/_ref harness — TEST INFRASTRUCTURE ONLY.
 *
 * The ONE piece of the reference's hot-path neighbourhood that compiles in this image without Eigen/Boost/OpenCV: the header-only k-d tree the
 * two-frame initialiser uses for CoarseInitializer::makeNN (src/FullSystem/CoarseInitializer.cpp:992-1069). This file is OUR driver; the reference's
 * header is compiled from where it lies (-I/root/reference/src/util, see oracle/Makefile target _ref/libref_nanoflann.so) and nothing of it is copied
 * into the repository. The point-cloud adaptor below is the user-side concept nanoflann asks for (kdtree_get_point_count / kdtree_distance /
 * kdtree_get_pt / kdtree_get_bbox), written against plain u/v arrays with the same fp32 arithmetic as FLANNPointcloud (CoarseInitializer.h:167-188:
 * d0*d0+d1*d1 in float, no bounding box).
 *
 * Exports the two queries makeNN issues, with the reference's own template arguments (KDTreeSingleIndexAdaptor<L2_Simple_Adaptor<float, Cloud>, Cloud, 2>,
 * leaf size 5, KNNResultSet<float,int,int>): the 10 nearest neighbours of every point inside its level and the single nearest point of
 * (0.5 u - 0.25, 0.5 v - 0.25) one level up. The tie order of equidistant neighbours (points sit on an integer grid) is whatever this tree's
 * traversal yields — which is exactly what the product and the C restatement have to reproduce.
 */
#include <cstddef>
#include <vector>
#include "nanoflann.h"

namespace {
struct Cloud {
    int num; const float* u; const float* v;
    inline size_t kdtree_get_point_count() const { return (size_t)num; }
    inline float kdtree_distance(const float* p1, const size_t idx_p2, size_t) const {
        const float d0 = p1[0] - u[idx_p2];
        const float d1 = p1[1] - v[idx_p2];
        return d0 * d0 + d1 * d1;
    }
    inline float kdtree_get_pt(const size_t idx, int dim) const { return dim == 0 ? u[idx] : v[idx]; }
    template <class BBOX> bool kdtree_get_bbox(BBOX&) const { return false; }
};
typedef nanoflann::KDTreeSingleIndexAdaptor<nanoflann::L2_Simple_Adaptor<float, Cloud>, Cloud, 2> Tree;
}  // namespace

extern "C" {

/* knn: for each of the nq queries (qu, qv) the k nearest points of the cloud (u, v)[n]: idx_out [nq][k], dist_out [nq][k] (squared distances, ascending).
 * Entries beyond the number of points found keep -1 / FLT_MAX. */
int ref_nanoflann_knn(int n, const float* u, const float* v, int nq, const float* qu, const float* qv, int k, int* idx_out, float* dist_out) {
    if (n <= 0 || k <= 0) return -1;
    Cloud pc{n, u, v};
    Tree index(2, pc, nanoflann::KDTreeSingleIndexAdaptorParams(5));
    index.buildIndex();
    nanoflann::KNNResultSet<float, int, int> rs(k);
    std::vector<int> ri(k); std::vector<float> rd(k);
    for (int i = 0; i < nq; ++i) {
        for (int j = 0; j < k; ++j) { ri[j] = -1; rd[j] = 3.402823466e+38f; }
        rs.init(ri.data(), rd.data());
        const float pt[2] = {qu[i], qv[i]};
        index.findNeighbors(rs, pt, nanoflann::SearchParams());
        for (int j = 0; j < k; ++j) { idx_out[(size_t)i * k + j] = ri[j]; dist_out[(size_t)i * k + j] = rd[j]; }
    }
    return 0;
}

}  // extern "C"
