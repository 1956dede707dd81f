#include <tf/transform_listener.h>
