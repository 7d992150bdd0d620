#include <pcl/point_types.h>
