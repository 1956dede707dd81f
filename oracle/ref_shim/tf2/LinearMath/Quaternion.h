#include <tf/transform_datatypes.h>
