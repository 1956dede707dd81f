// included by the node sources, never used
#ifndef MOT_SHIM_IMU_H
#define MOT_SHIM_IMU_H
namespace sensor_msgs { struct Imu {}; }
#endif
