// boost/shared_ptr.hpp — stub: ROS message ConstPtr types and pcl::PointCloud::Ptr are boost::shared_ptr upstream
#ifndef LVREF_BOOST_SHARED_PTR_STUB
#define LVREF_BOOST_SHARED_PTR_STUB
#include <memory>
namespace boost {
template <typename T> using shared_ptr = std::shared_ptr<T>;
}
#endif
