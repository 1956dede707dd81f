#include <geometry_msgs/geometry.h>
