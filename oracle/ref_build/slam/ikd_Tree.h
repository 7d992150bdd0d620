// ikd_Tree.h — STAND-IN for the ikd-Tree submodule (github.com/Huguet57/ikd-Tree, absent from the reference mount: SURVEY F1).
// oracle/ref_build, TEST INFRASTRUCTURE.  The surface the reference uses (src/Modules/Mapper.cpp:33,65,70,75,86): ctor(3 floats),
// Build, Add_Points, Nearest_Search, size.  The search is EXACT k-NN under the (squared f32 distance, insertion index) order
// [UPSTREAM-RECALL ikd-Tree: calc_dist = dx*dx + dy*dy + dz*dz in f32, results ascending, fewer than k when the tree is smaller]
// through the oracle's brute-force / kd-tree search; Add_Points with downsampling follows the oracle's restatement of the 0.2 m
// box rule (lvo_map_add).  Neither is the fork's code: a comparison against this build pins the reference's callers, not these.
#ifndef LVREF_IKD_TREE_STUB
#define LVREF_IKD_TREE_STUB
#include <cassert>
#include <cmath>
#include <memory>
#include <vector>
#include "lv_oracle.h"

using namespace std;   // [UPSTREAM-RECALL ikd_Tree.h has it; Mapper.cpp:85 relies on it: `vector<float>`]

#ifndef _OPENMP
inline void omp_set_num_threads(int) {}   // (Mapper.cpp:45; this build compiles the loop sequentially: deterministic match order)
#else
#include <omp.h>
#endif

template <typename PointType>
class KD_TREE {
public:
    typedef std::vector<PointType, Eigen::aligned_allocator<PointType>> PointVector;
    typedef std::shared_ptr<KD_TREE<PointType>> Ptr;
    KD_TREE(float delete_param = 0.5f, float balance_param = 0.6f, float box_length = 0.2f) : box_length_(box_length) {
        (void)delete_param; (void)balance_param;
        instances().push_back(this);
    }
    ~KD_TREE() { drop_index(); }
    static std::vector<KD_TREE*>& instances() { static std::vector<KD_TREE*> v; return v; }

    void Build(PointVector points) {
        pts_.assign(points.begin(), points.end());
        rebuild_xyz();
    }
    int Add_Points(PointVector& add, bool downsample_on) {
        std::vector<float> nx(3 * add.size()), out(3 * (pts_.size() + add.size()));
        for (size_t i = 0; i < add.size(); ++i) { nx[3 * i] = add[i].x; nx[3 * i + 1] = add[i].y; nx[3 * i + 2] = add[i].z; }
        const size_t m = lvo_map_add(xyz_.data(), pts_.size(), nx.data(), add.size(), downsample_on ? 1 : 0, box_length_, out.data());
        // lvo_map_add returns [surviving old points in their old order] + [surviving new points in input order]: give every
        // survivor the attributes of the point it is (greedy match by coordinates along the two runs)
        std::vector<PointType> np;
        np.reserve(m);
        size_t k = 0;
        for (size_t i = 0; i < pts_.size() && k < m; ++i)
            if (pts_[i].x == out[3 * k] && pts_[i].y == out[3 * k + 1] && pts_[i].z == out[3 * k + 2]) { np.push_back(pts_[i]); ++k; }
        for (size_t i = 0; i < add.size() && k < m; ++i)
            if (add[i].x == out[3 * k] && add[i].y == out[3 * k + 1] && add[i].z == out[3 * k + 2]) { np.push_back(add[i]); ++k; }
        assert(k == m);
        pts_.swap(np);
        rebuild_xyz();
        return (int)add.size();
    }
    void Nearest_Search(PointType point, int k_nearest, PointVector& Nearest_Points, std::vector<float>& Point_Distance, double max_dist = INFINITY) {
        (void)max_dist;
        Nearest_Points.clear();
        Point_Distance.clear();
        if (pts_.empty() || k_nearest < 1) return;
        if (!index_) index_ = lvo_kdtree_build(xyz_.data(), pts_.size());
        std::vector<uint32_t> idx((size_t)k_nearest);
        std::vector<float> d2((size_t)k_nearest);
        int32_t found = 0;
        const float q[3] = {point.x, point.y, point.z};
        lvo_kdtree_knn(index_, q, 1, k_nearest, idx.data(), d2.data(), &found, 1);
        for (int j = 0; j < found; ++j) { Nearest_Points.push_back(pts_[idx[(size_t)j]]); Point_Distance.push_back(d2[(size_t)j]); }
        last_idx_.assign(idx.begin(), idx.begin() + found);
    }
    int size() { return (int)pts_.size(); }
    // (stand-in only) reset between test cases; the indices of the last search, for the kNN comparison
    void clear() { pts_.clear(); xyz_.clear(); drop_index(); }
    const std::vector<uint32_t>& last_indices() const { return last_idx_; }
    const std::vector<PointType>& points() const { return pts_; }
private:
    void rebuild_xyz() {
        xyz_.resize(3 * pts_.size());
        for (size_t i = 0; i < pts_.size(); ++i) { xyz_[3 * i] = pts_[i].x; xyz_[3 * i + 1] = pts_[i].y; xyz_[3 * i + 2] = pts_[i].z; }
        drop_index();
    }
    void drop_index() { if (index_) { lvo_kdtree_free(index_); index_ = nullptr; } }
    float box_length_;
    std::vector<PointType> pts_;
    std::vector<float> xyz_;
    void* index_ = nullptr;
    std::vector<uint32_t> last_idx_;
};
#endif
