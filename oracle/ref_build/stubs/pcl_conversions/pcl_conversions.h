// pcl_conversions stub: pcl::fromROSMsg / toROSMsg  [UPSTREAM-RECALL pcl_conversions / pcl::fromPCLPointCloud2]
#ifndef LVREF_PCL_CONVERSIONS_STUB
#define LVREF_PCL_CONVERSIONS_STUB
#include <pcl/point_types.h>
namespace pcl {
// Header: stamp in microseconds = nanoseconds / 1000 (pcl_conversions::toPCL).  Fields: every registered field of PointT is
// looked up in msg.fields by NAME, DATATYPE and count == 1 (pcl::FieldMatches); a field without a match is reported and left
// value-initialised; matched fields are copied bytewise, point by point.
template <typename PointT>
void fromROSMsg(const sensor_msgs::PointCloud2& msg, PointCloud<PointT>& cloud) {
    cloud.header.seq = msg.header.seq;
    cloud.header.stamp = msg.header.stamp.toNSec() / 1000ull;
    cloud.header.frame_id = msg.header.frame_id;
    cloud.width = msg.width;
    cloud.height = msg.height;
    cloud.is_dense = msg.is_dense;
    const std::size_t n = (std::size_t)msg.width * msg.height;
    cloud.points.assign(n, PointT());
    for (std::size_t i = 0; i < n; ++i) std::memset((void*)&cloud.points[i], 0, sizeof(PointT));
    struct Map { std::size_t src, dst, size; };
    std::vector<Map> maps;
    for (const RegisteredField& f : point_fields<PointT>::get()) {
        bool found = false;
        for (const sensor_msgs::PointField& mf : msg.fields)
            if (mf.name == f.tag && (int)mf.datatype == f.datatype && mf.count == 1) { maps.push_back({mf.offset, f.offset, f.size}); found = true; break; }
        if (!found) std::fprintf(stderr, "Failed to find match for field '%s'.\n", f.tag);
    }
    for (std::size_t i = 0; i < n; ++i) {
        const std::uint8_t* src = msg.data.data() + i * msg.point_step;
        std::uint8_t* dst = reinterpret_cast<std::uint8_t*>(&cloud.points[i]);
        for (const Map& m : maps) std::memcpy(dst + m.dst, src + m.src, m.size);
    }
}
template <typename PointT>
void toROSMsg(const PointCloud<PointT>&, sensor_msgs::PointCloud2&) {}
}  // namespace pcl
#endif
