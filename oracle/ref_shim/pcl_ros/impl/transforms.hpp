#include <pcl_ros/transforms.h>
