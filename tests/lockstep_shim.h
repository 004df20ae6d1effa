// Test infrastructure: compiles the C++ host mirror's driver (obvi-slam_amd/host/run_offline_ba.cpp) against BOTH backends at once.
// Force-included before include/obvi_ba.h (g++ -include): every entry of the C ABI the host mirror calls is renamed to a lock_* function
// of tests/lockstep_shim.cpp, which forwards it to libobvi_ba.so (the HIP path) AND to the CPU oracle, compares what comes back, and
// keeps the oracle's state equal to the HIP path's after every solve -- so every optimisation of a session starts from the same
// values on both and can be compared one to one (tests/test_lockstep_session.py), without the chaotic drift of two whole sessions.
#ifndef OBVI_TESTS_LOCKSTEP_SHIM_H_
#define OBVI_TESTS_LOCKSTEP_SHIM_H_
#define obvi_ba_create lock_ba_create
#define obvi_ba_destroy lock_ba_destroy
#define obvi_ba_reset lock_ba_reset
#define obvi_ba_last_error lock_ba_last_error
#define obvi_ba_set_cameras lock_ba_set_cameras
#define obvi_ba_set_poses lock_ba_set_poses
#define obvi_ba_set_points lock_ba_set_points
#define obvi_ba_set_objects lock_ba_set_objects
#define obvi_ba_set_const_flags lock_ba_set_const_flags
#define obvi_ba_set_reproj lock_ba_set_reproj
#define obvi_ba_set_bbox lock_ba_set_bbox
#define obvi_ba_set_shape_priors lock_ba_set_shape_priors
#define obvi_ba_set_ltm_priors lock_ba_set_ltm_priors
#define obvi_ba_set_relpose lock_ba_set_relpose
#define obvi_ba_set_active_mask lock_ba_set_active_mask
#define obvi_ba_set_parameter_priors lock_ba_set_parameter_priors
#define obvi_ba_column_sqnorms lock_ba_column_sqnorms
#define obvi_ba_evaluate lock_ba_evaluate
#define obvi_ba_solve lock_ba_solve
#define obvi_ba_get_iterations lock_ba_get_iterations
#define obvi_ba_select_outliers lock_ba_select_outliers
#define obvi_ba_object_covariances lock_ba_object_covariances
#define obvi_ba_snapshot lock_ba_snapshot
#define obvi_ba_restore lock_ba_restore
#define obvi_ba_get_poses lock_ba_get_poses
#define obvi_ba_get_points lock_ba_get_points
#define obvi_ba_get_objects lock_ba_get_objects
#define obvi_ba_get_state lock_ba_get_state
#define obvi_ba_update_points lock_ba_update_points
#define obvi_ba_update_state lock_ba_update_state
#define obvi_ba_prepare lock_ba_prepare
#define obvi_ba_num_residuals lock_ba_num_residuals
#define obvi_ba_num_factors lock_ba_num_factors
#endif
