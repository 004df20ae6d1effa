// Test infrastructure: compiles the C++ host mirror's driver (obvi-slam_amd/host/run_offline_ba.cpp) against the CPU oracle instead
// of libobvi_ba.so, so that a whole session -- window provider, two-phase iterations, PGO stages, final global BA, long-term map --
// can be driven through the oracle and compared with the same session on the HIP path (tests/test_host_mirror.py).  Force-included
// before include/obvi_ba.h (g++ -include): every entry of the C ABI the host mirror calls is renamed to the oracle's export.
#ifndef OBVI_TESTS_ORACLE_ABI_SHIM_H_
#define OBVI_TESTS_ORACLE_ABI_SHIM_H_
#define obvi_ba_create oracle_ba_create
#define obvi_ba_destroy oracle_ba_destroy
#define obvi_ba_reset oracle_ba_reset
#define obvi_ba_last_error oracle_ba_last_error
#define obvi_ba_set_cameras oracle_ba_set_cameras
#define obvi_ba_set_poses oracle_ba_set_poses
#define obvi_ba_set_points oracle_ba_set_points
#define obvi_ba_set_objects oracle_ba_set_objects
#define obvi_ba_set_const_flags oracle_ba_set_const_flags
#define obvi_ba_set_reproj oracle_ba_set_reproj
#define obvi_ba_set_bbox oracle_ba_set_bbox
#define obvi_ba_set_shape_priors oracle_ba_set_shape_priors
#define obvi_ba_set_ltm_priors oracle_ba_set_ltm_priors
#define obvi_ba_set_relpose oracle_ba_set_relpose
#define obvi_ba_set_active_mask oracle_ba_set_active_mask
#define obvi_ba_set_parameter_priors oracle_ba_set_parameter_priors
#define obvi_ba_column_sqnorms oracle_ba_column_sqnorms
#define obvi_ba_evaluate oracle_ba_evaluate
#define obvi_ba_solve oracle_ba_solve
#define obvi_ba_get_iterations oracle_ba_get_iterations
#define obvi_ba_select_outliers oracle_ba_select_outliers
#define obvi_ba_object_covariances oracle_ba_object_covariances
#define obvi_ba_snapshot oracle_ba_snapshot
#define obvi_ba_restore oracle_ba_restore
#define obvi_ba_get_poses oracle_ba_get_poses
#define obvi_ba_get_points oracle_ba_get_points
#define obvi_ba_get_objects oracle_ba_get_objects
#define obvi_ba_get_state oracle_ba_get_state
#define obvi_ba_update_points oracle_ba_update_points
#define obvi_ba_update_state oracle_ba_update_state
#define obvi_ba_prepare oracle_ba_prepare
#define obvi_ba_num_residuals oracle_ba_num_residuals
#define obvi_ba_num_factors oracle_ba_num_factors
#define obvi_frontend_epipolar_votes oracle_frontend_epipolar_votes
#define obvi_frontend_epipolar_errors oracle_frontend_epipolar_errors
#define obvi_frontend_parallax oracle_frontend_parallax
#endif
