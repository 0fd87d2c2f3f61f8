/*
 * calico_hip_testing.h — test hooks of libcalico_hip.so. NOT part of the drop-in surface (include/calico_hip.h): nothing a
 * maintainer binds; tests/ reach these symbols through ctypes.
 */
#ifndef CALICO_HIP_TESTING_H_
#define CALICO_HIP_TESTING_H_

#include "../../include/calico_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* The device's trust-region control stage (the accept / reject decision, the
 * radius schedule and the iteration log of ceres::TrustRegionMinimizer as restated in solve_kernels.hip) driven by a
 * given sequence of step qualities rho[i] (infinite[i] != 0: the candidate's cost could not be evaluated). Row i runs
 * the same control kernel a solve runs, seeded with x_cost = model_cost_change = 1 and candidate cost 1 - rho[i];
 * radius, decrease factor and counters carry over. Out: radius after the row, accepted flag, and the cost column the
 * row shows. tests/test_ceres_log.py replays the iteration table the reference ships
 * (demos/imu_camera_calibration.ipynb) through it. */
int32_t calico_debug_lm_control_replay(int32_t device, int32_t n, const double* rho, const int32_t* infinite,
                                       const calico_solver_options* options, double* radius_out, int32_t* accepted_out,
                                       double* cost_column_out);

/* What calico_problem_finalize decided for the handle's structure (finalizes the handle if it has not been yet), so
 * that a test can assert WHICH evaluation route it is comparing with the oracle. out[0..n) (n <= 9) receives:
 *   [0] fuse_expand (1: eval_cells_kernel -- cell workgroups; 0: eval_jacobian_kernel + expand_cells_kernel + row cells),
 *   [1] camera frames on the frame path, [2] work items of the generic / IMU path, [3] cells,
 *   [4] most frames in one camera cell, [5] most work items in one (layout, segment) of the item path,
 *   [6] 1: tree solver, 0: sequential banded solver, [7] m (tangent size of the calibration blocks). */
int32_t calico_debug_plan_info(calico_problem* problem, int32_t* out, int32_t n);
/* (out[8], n = 9: 1 if every control point is observed.) */

/* Host only (no device needed): the 48 words of the rolling chief's per-lane offset table (bcr_kernels.hip, g_roll_tab) for a spline
 * order 1..6 and a lane 0..63 -- byte offsets of the lane's tile entries in the band's storage and which of them exist. */
int32_t calico_debug_roll_table(int32_t spline_order, int32_t lane, uint32_t* out48);

#ifdef __cplusplus
}
#endif
#endif /* CALICO_HIP_TESTING_H_ */
