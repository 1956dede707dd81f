#include <sensor_msgs/PointCloud2.h>
