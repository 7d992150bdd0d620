/* lvref.h — C interface of oracle/_ref/liblvref.so: the REFERENCE's own in-tree sources (/root/reference/src, compiled in place
 * against the stand-in headers of oracle/ref_build) behind plain C entry points.  TEST INFRASTRUCTURE (see ref_glue.cpp). */
#ifndef LVREF_H
#define LVREF_H
#include <stddef.h>
#include <stdint.h>
#include "lv_oracle.h"
#ifdef __cplusplus
extern "C" {
#endif
typedef struct lvr_config {   /* the keys of struct Params (include/Headers/Common.hpp:56-107) the compiled sources read */
    int estimate_extrinsics, max_num_iters, num_match_points, max_points2match;
    double max_dist_plane;
    float planes_threshold;
    double lidar_noise, degeneracy_threshold;
    double limits[23];
    float initial_gravity[3], I_Rotation_L[9], I_Translation_L[3];
    double cov_acc, cov_gyro, cov_bias_acc, cov_bias_gyro;
    double full_rotation_time, imu_rate, real_time_delay, min_dist;
    int offset_beginning, stamp_beginning, downsample_rate, lidar_type;   /* lidar_type: 0 velodyne, 1 hesai, 2 ouster, 3 custom */
    float downsample_prec;
} lvr_config;
void lvr_set_config(const lvr_config* c);
void lvr_reset(void);
void lvr_state_to_pose(const double x[26], float out[24]);
void lvr_transform(const double x[26], const float* scan_xyz, size_t n, float* out_xyz);
void lvr_map_add(const float* xyz, size_t n, double time, int downsample);
size_t lvr_map_size(void);
void lvr_map_fetch(float* out_xyz);
size_t lvr_match(const double x[26], const float* scan_xyz, size_t n, uint32_t* src_index, float* p_world, float* abcd, float* dist);
int lvr_plane(const float* near_xyz, const float* sq_dists, int found, float abcd[4]);
void lvr_estimate_plane(const float* near_xyz, int npts, float abcd[4]);
void lvr_calculate_H(const double x[26], size_t n, const float* p_world, const float* abcd, double* H, double* h, float* dist_out);
int lvr_update(double x[26], double* P, const float* scan_xyz, size_t n, lvo_iter_out* sums_log, double* state_log);
void lvr_initialize(const float a[3], const float w[3], const float q_xyzw[4], double t, double x[26], double* P);
void lvr_propagate(double x[26], double* P, double last_time_integrated, const float* imu_a, const float* imu_w, const double* imu_t, size_t n_imu, double t);
void lvr_state_integrate(lvo_motion_state* m, const float a[3], const float w[3], double t);
size_t lvr_deskew(const float* xyz, const double* times, size_t n, const lvo_motion_state* states, size_t n_states, const lvo_motion_state* Xt2, float* out_xyz);
size_t lvr_path(const lvo_motion_state* states, size_t n_states, const float* imu_a, const float* imu_w, const double* imu_t, size_t n_imu, double t1, double t2,
                lvo_motion_state* out, size_t cap);
size_t lvr_cloud_ingest(const uint8_t* data, size_t n, uint32_t point_step, int nfields, const char* const* names, const uint32_t* offsets,
                        const uint8_t* datatypes, uint64_t stamp_usec, lvo_point* out);
size_t lvr_buffer_window(const double* times, size_t n, double t1, double t2, double clear_t, double* out);
#ifdef __cplusplus
}
#endif
#endif
