#ifndef MOT_SHIM_MARKERARRAY_H
#define MOT_SHIM_MARKERARRAY_H
#include <visualization_msgs/Marker.h>
namespace visualization_msgs { struct MarkerArray { std::vector<Marker> markers; }; }
#endif
