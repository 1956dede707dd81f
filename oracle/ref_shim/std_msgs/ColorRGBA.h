#include <std_msgs/Header.h>
