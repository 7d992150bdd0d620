// pcl stubs (oracle/ref_build; TEST INFRASTRUCTURE): point-type macros, PointCloud, the PointCloud2 conversion and VoxelGrid,
// restated from the published behaviour of PCL 1.8 [UPSTREAM-RECALL PCL 1.8] for the calls the reference makes
// (include/Headers/Common.hpp:109-221, src/Utils/PointCloudProcessor.cpp, src/Modules/Compensator.cpp:148-163).
#ifndef LVREF_PCL_STUB
#define LVREF_PCL_STUB
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#include <ros/ros.h>
#include <sensor_msgs/PointCloud2.h>

// (usable with and without a trailing semicolon: Common.hpp has both)
#define PCL_ADD_POINT4D union alignas(16) { float data[4]; struct { float x; float y; float z; }; };
#define PCL_ADD_RGB union { union { struct { std::uint8_t b; std::uint8_t g; std::uint8_t r; std::uint8_t a; }; float rgb; }; std::uint32_t rgba; };

namespace pcl {
struct PCLHeader { std::uint32_t seq = 0; std::uint64_t stamp = 0; std::string frame_id; };

template <typename T> struct field_datatype;
template <> struct field_datatype<std::int8_t> { enum { value = sensor_msgs::PointField::INT8 }; };
template <> struct field_datatype<std::uint8_t> { enum { value = sensor_msgs::PointField::UINT8 }; };
template <> struct field_datatype<std::int16_t> { enum { value = sensor_msgs::PointField::INT16 }; };
template <> struct field_datatype<std::uint16_t> { enum { value = sensor_msgs::PointField::UINT16 }; };
template <> struct field_datatype<std::int32_t> { enum { value = sensor_msgs::PointField::INT32 }; };
template <> struct field_datatype<std::uint32_t> { enum { value = sensor_msgs::PointField::UINT32 }; };
template <> struct field_datatype<float> { enum { value = sensor_msgs::PointField::FLOAT32 }; };
template <> struct field_datatype<double> { enum { value = sensor_msgs::PointField::FLOAT64 }; };

struct RegisteredField { const char* tag; std::size_t offset; int datatype; std::size_t size; };
template <typename PointT> struct point_fields;   // specialised by POINT_CLOUD_REGISTER_POINT_STRUCT

template <typename PointT>
class PointCloud {
public:
    typedef boost::shared_ptr<PointCloud<PointT>> Ptr;
    typedef boost::shared_ptr<const PointCloud<PointT>> ConstPtr;
    PCLHeader header;
    std::vector<PointT> points;
    std::uint32_t width = 0, height = 1;
    bool is_dense = true;
    std::size_t size() const { return points.size(); }
    bool empty() const { return points.empty(); }
    void clear() { points.clear(); width = 0; }
    void push_back(const PointT& p) { points.push_back(p); width = (std::uint32_t)points.size(); }
};
}  // namespace pcl

// iteration over the (type, name, tag)(type, name, tag)... sequence without Boost.Preprocessor
#define LVREF_REG_A(type, name, tag) v.push_back(pcl::RegisteredField{#tag, offsetof(PT_, name), (int)pcl::field_datatype<type>::value, sizeof(type)}); LVREF_REG_B
#define LVREF_REG_B(type, name, tag) v.push_back(pcl::RegisteredField{#tag, offsetof(PT_, name), (int)pcl::field_datatype<type>::value, sizeof(type)}); LVREF_REG_A
#define LVREF_REG_A_END
#define LVREF_REG_B_END
#define LVREF_CAT_(a, b) a##b
#define LVREF_CAT(a, b) LVREF_CAT_(a, b)
#define POINT_CLOUD_REGISTER_POINT_STRUCT(PointT, seq)                                              \
    namespace pcl {                                                                                   \
    template <> struct point_fields<PointT> {                                                         \
        static const std::vector<RegisteredField>& get() {                                            \
            typedef PointT PT_;                                                                       \
            static const std::vector<RegisteredField> f = [] {                                        \
                std::vector<RegisteredField> v;                                                       \
                _Pragma("GCC diagnostic push") _Pragma("GCC diagnostic ignored \"-Winvalid-offsetof\"") \
                LVREF_CAT(LVREF_REG_A seq, _END)                                                      \
                _Pragma("GCC diagnostic pop")                                                         \
                return v;                                                                             \
            }();                                                                                      \
            return f;                                                                                 \
        }                                                                                             \
    };                                                                                                \
    }
#endif
