// pcl::VoxelGrid stub [UPSTREAM-RECALL PCL 1.8 VoxelGrid::applyFilter, downsample_all_data_ = true]: leaf index
// floor(p * inverse_leaf_size) - min_b per axis, linear index i + j*dx + k*dx*dy, points sorted by leaf index, per occupied leaf
// the centroid of x, y, z as f32 sums in sorted order divided by the count; leaves in ascending index order.  The sort is STABLE
// here (std::sort upstream: the order of the points of a leaf, hence the last bits of an f32 centroid, is unspecified there) —
// the same statement as oracle/lv_oracle.cpp::lvo_voxelgrid, whose arithmetic this reuses through the C interface.
#ifndef LVREF_PCL_VOXELGRID_STUB
#define LVREF_PCL_VOXELGRID_STUB
#include <pcl/point_types.h>
extern "C" size_t lvo_voxelgrid(const float* xyz, size_t n, float leaf, float* out_xyz);
namespace pcl {
template <typename PointT>
class VoxelGrid {
    typename PointCloud<PointT>::Ptr in_;
    float leaf_ = 0.f;
public:
    void setInputCloud(const typename PointCloud<PointT>::Ptr& c) { in_ = c; }
    void setLeafSize(float lx, float, float) { leaf_ = lx; }   // (the reference passes one precision three times)
    void filter(PointCloud<PointT>& out) {
        const std::size_t n = in_->points.size();
        std::vector<float> xyz(3 * n), o(3 * n);
        for (std::size_t i = 0; i < n; ++i) { xyz[3 * i] = in_->points[i].x; xyz[3 * i + 1] = in_->points[i].y; xyz[3 * i + 2] = in_->points[i].z; }
        const std::size_t m = n ? lvo_voxelgrid(xyz.data(), n, leaf_, o.data()) : 0;
        out.points.assign(m, PointT());
        for (std::size_t i = 0; i < m; ++i) {
            std::memset((void*)&out.points[i], 0, sizeof(PointT));
            out.points[i].x = o[3 * i]; out.points[i].y = o[3 * i + 1]; out.points[i].z = o[3 * i + 2];
        }
        out.header = in_->header;
        out.width = (std::uint32_t)m;
    }
};
}  // namespace pcl
#endif
