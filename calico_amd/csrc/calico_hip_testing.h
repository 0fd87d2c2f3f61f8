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

#ifdef __cplusplus
}
#endif
#endif /* CALICO_HIP_TESTING_H_ */
