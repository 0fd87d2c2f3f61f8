/*
 * calico_hip.h — C ABI of libcalico_hip.so, the MI355X (gfx950) drop-in for the
 * hot path of yangjames/Calico:  BatchOptimizer::Optimize
 * (reference calico/batch_optimizer.cpp:53-81).
 *
 * The reference builds a ceres::Problem (parameter blocks + one residual block
 * per measurement), calls ceres::Solve, then re-evaluates every residual
 * block.  This header is what a maintainer binds instead of Ceres for that
 * path: every entry point names the reference interface it replaces.  The ABI
 * is plain C: opaque handle, caller-owned host buffers, int32 status codes
 * (absl::StatusCode numbering, as the reference's absl::Status uses), no
 * exceptions, no C++/torch types.
 *
 * Conventions
 *  - all floating point is IEEE-754 double (the reference path is double only);
 *  - quaternion blocks are stored x,y,z,w (Eigen::Quaterniond::coeffs(),
 *    reference optimization_utils.h:51-60);
 *  - host buffers are only read during the call that receives them;
 *  - one handle = one HIP device + one stream; a handle is not thread-safe.
 */
#ifndef CALICO_HIP_H_
#define CALICO_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes: absl::StatusCode numbering ------------------------- */
#define CALICO_OK 0
#define CALICO_INVALID_ARGUMENT 3
#define CALICO_FAILED_PRECONDITION 9
#define CALICO_UNIMPLEMENTED 12
#define CALICO_INTERNAL 13

/* ---- enums: integer values identical to the reference ----------------- */
/* ceres manifold kinds used by the reference (optimization_utils.h:51-60) */
#define CALICO_MANIFOLD_EUCLIDEAN 0
#define CALICO_MANIFOLD_EIGEN_QUATERNION 1
/* sensors::CameraIntrinsicsModel (camera_models.h:16-33) */
#define CALICO_CAMERA_NONE 0
#define CALICO_CAMERA_OPENCV5 1
#define CALICO_CAMERA_OPENCV8 2
#define CALICO_CAMERA_KANNALA_BRANDT 3
#define CALICO_CAMERA_DOUBLE_SPHERE 4
#define CALICO_CAMERA_FIELD_OF_VIEW 5
#define CALICO_CAMERA_UNIFIED 6
#define CALICO_CAMERA_EXTENDED_UNIFIED 7
/* sensors::{Gyroscope,Accelerometer}IntrinsicsModel (gyroscope_models.h:16-25) */
#define CALICO_IMU_NONE 0
#define CALICO_IMU_SCALE_ONLY 1
#define CALICO_IMU_SCALE_AND_BIAS 2
#define CALICO_IMU_VECTOR_NAV 3
/* utils::LossFunctionType (optimization_utils.h:15-22) */
#define CALICO_LOSS_NONE 0
#define CALICO_LOSS_HUBER 1
#define CALICO_LOSS_CAUCHY 2
/* ceres::TerminationType */
#define CALICO_CONVERGENCE 0
#define CALICO_NO_CONVERGENCE 1
#define CALICO_FAILURE 2
/* sensor kinds of this ABI */
#define CALICO_SENSOR_CAMERA 0
#define CALICO_SENSOR_GYROSCOPE 1
#define CALICO_SENSOR_ACCELEROMETER 2

typedef struct calico_problem calico_problem;

/* Mirrors the fields of ceres::Solver::Options the reference sets or exposes
 * (batch_optimizer.cpp:10-17, calico.cpp:378-394) plus the Ceres trust-region
 * defaults the path depends on.  calico_default_solver_options() fills it the
 * way DefaultSolverOptions() + Ceres defaults do. */
typedef struct calico_solver_options {
  int32_t max_num_iterations;                /* 50 */
  int32_t num_threads;                       /* 1; honoured by CPU code only */
  int32_t minimizer_progress_to_stdout;      /* reference default: 1 */
  int32_t jacobi_scaling;                    /* 1 */
  int32_t max_num_consecutive_invalid_steps; /* 5 */
  int32_t sync_every; /* HIP: LM iterations enqueued between host syncs (>=1) */
  double function_tolerance;                 /* 1e-8  (batch_optimizer.cpp:14) */
  double gradient_tolerance;                 /* 1e-10 */
  double parameter_tolerance;                /* 1e-10 (batch_optimizer.cpp:15) */
  double initial_trust_region_radius;        /* 1e4 */
  double max_trust_region_radius;            /* 1e16 */
  double min_trust_region_radius;            /* 1e-32 */
  double min_relative_decrease;              /* 1e-3 */
  double min_lm_diagonal;                    /* 1e-6 */
  double max_lm_diagonal;                    /* 1e32 */
} calico_solver_options;

/* The fields of ceres::Solver::Summary the reference reads or binds
 * (calico.cpp:356-375, batch_optimizer_test.cpp:186-187). */
typedef struct calico_summary {
  int32_t termination_type;
  int32_t num_successful_steps;
  int32_t num_unsuccessful_steps;
  int32_t num_iterations; /* LM iterations run, iteration 0 excluded */
  int32_t num_jacobian_evaluations;
  int32_t num_cost_evaluations;
  int32_t num_residual_blocks;
  int32_t num_residuals;
  int32_t num_parameter_blocks;
  int32_t num_parameters;
  int32_t num_effective_parameters;
  int32_t num_residual_blocks_reduced;
  int32_t num_residuals_reduced;
  int32_t num_parameter_blocks_reduced;
  int32_t num_parameters_reduced;
  int32_t num_effective_parameters_reduced;
  double initial_cost;
  double final_cost;
  double total_time_in_seconds;
  double solve_time_in_seconds; /* LM loop only (device-timed on HIP) */
  char message[256];
} calico_summary;

/* One row of Ceres' per-iteration progress table. */
typedef struct calico_iteration {
  int32_t iteration;
  int32_t step_is_valid;
  int32_t step_is_successful;
  int32_t reserved;
  double cost;
  double cost_change;
  double gradient_max_norm;
  double step_norm;
  double relative_decrease;
  double trust_region_radius;
} calico_iteration;

/* ---- lifetime --------------------------------------------------------- */
/* Replaces `ceres::Problem problem;` (batch_optimizer.cpp:57).  device = HIP
 * device ordinal. Fails with CALICO_INTERNAL when no GPU is usable: there is
 * no CPU fallback behind this ABI. */
int32_t calico_problem_create(calico_problem** out, int32_t device);
void calico_problem_destroy(calico_problem* p);
/* Message of the last non-OK status on this handle ("" if none). */
const char* calico_last_error(const calico_problem* p);
void calico_default_solver_options(calico_solver_options* o);

/* ---- parameters ------------------------------------------------------- */
/* Replaces ceres::Problem::AddParameterBlock (+ SetParameterBlockConstant,
 * + EigenQuaternionManifold) as used by world_model.cpp:40-77,
 * bspline.hpp:10-17, camera.cpp:92-113, gyroscope.cpp:10-31,
 * accelerometer.cpp:10-33, optimization_utils.h:51-68.  size must be 4 for
 * the quaternion manifold. */
int32_t calico_problem_add_param_block(calico_problem* p, const double* values,
                                       int32_t size, int32_t manifold,
                                       int32_t is_constant,
                                       int32_t* block_id_out);
/* Bulk form for n blocks of one size and manifold (the model points of a chart, the control points of the spline):
 * values n x size row-major, is_constant one byte per block (NULL: none is constant), ids returned in block_ids_out. */
int32_t calico_problem_add_param_blocks(calico_problem* p, int32_t n, int32_t size, int32_t manifold,
                                        const uint8_t* is_constant, const double* values, int32_t* block_ids_out);
/* Read / overwrite the current value of a block (the reference hands Ceres
 * pointers into the user's objects and reads them back in place). */
int32_t calico_get_param_block(calico_problem* p, int32_t block_id,
                               double* out);
int32_t calico_set_param_block(calico_problem* p, int32_t block_id,
                               const double* values);
/* Bulk forms: n blocks, values concatenated in the order of block_ids. Ceres reads and writes through the pointers
 * Trajectory::AddParametersToProblem / WorldModel::AddParametersToProblem hand it (trajectory.cpp:51-60,
 * world_model.cpp:52-61), so after Optimize() (batch_optimizer.cpp:72-78) the control points and model points are
 * simply there; here they are fetched -- in one call instead of one per control point. */
int32_t calico_set_param_blocks(calico_problem* p, int32_t n,
                                const int32_t* block_ids,
                                const double* values);
int32_t calico_get_param_blocks(calico_problem* p, int32_t n,
                                const int32_t* block_ids,
                                double* values_out);

/* Replaces Trajectory::AddParametersToProblem + GetEvaluationParams
 * (trajectory.cpp:51-79, bspline.hpp:138-161): the uniform knot vector
 * (n_knots entries), the per-segment basis matrices (n_segments × order ×
 * order, row-major, n_segments = n_knots - 2*(order-1) - 1) and the block ids
 * of the n_knots - order control points (6-vectors [axis-angle; position]). */
int32_t calico_problem_set_spline(calico_problem* p, int32_t order,
                                  int32_t n_knots, const double* knots,
                                  const double* basis,
                                  const int32_t* ctrl_block_ids);

/* Rigid body (calibration chart) pose blocks, world_model.h:54-69. */
int32_t calico_problem_add_rigid_body(calico_problem* p, int32_t q_block,
                                      int32_t t_block, int32_t* body_id_out);

/* ---- sensors ---------------------------------------------------------- */
/* One call per Sensor object: what AddParametersToProblem registered plus the
 * per-sensor state AddResidualsToProblem reads (model, sigma -> information
 * 1/sigma, loss type + scale).  kind/model per the enums above.
 * gravity_block is used by accelerometers only (pass -1 otherwise). */
int32_t calico_problem_add_sensor(calico_problem* p, int32_t kind,
                                  int32_t model, int32_t intrinsics_block,
                                  int32_t q_block, int32_t t_block,
                                  int32_t latency_block, int32_t gravity_block,
                                  double sigma, int32_t loss,
                                  double loss_scale, int32_t* sensor_id_out);

/* Replaces Camera::AddResidualsToProblem (camera.cpp:115-153) for the
 * non-outlier measurements of one camera: pixels n×2, stamps n, the rigid
 * body and the model-point parameter block of every observation. */
int32_t calico_problem_add_camera_residuals(calico_problem* p,
                                            int32_t sensor_id, int64_t n,
                                            const double* pixels,
                                            const double* stamps,
                                            const int32_t* body_ids,
                                            const int32_t* point_blocks);
/* Replaces Gyroscope/Accelerometer::AddResidualsToProblem
 * (gyroscope.cpp:33-54, accelerometer.cpp:35-56): measurements n×3. */
int32_t calico_problem_add_imu_residuals(calico_problem* p, int32_t sensor_id,
                                         int64_t n, const double* measurements,
                                         const double* stamps);

/* Flatten the recorded blocks into the device-side problem now (cells, work items, gather lists, elimination plan;
 * uploads) instead of inside the first calico_solve / calico_evaluate. What BatchOptimizer::Optimize does before
 * ceres::Solve on every call (batch_optimizer.cpp:57-70: it rebuilds the ceres::Problem each time), made separately
 * callable so that its cost can be measured (bench.py: config.setup_ms). */
int32_t calico_problem_finalize(calico_problem* p);

/* The plan -- everything calico_problem_finalize derives -- depends on the STRUCTURE of the problem only (block sizes,
 * manifolds and constancy; the spline's knots, basis and control-point blocks; every sensor's model, blocks, sigma and
 * loss; per observation its stamp, rigid body and model point), not on parameter values or measurements. The library
 * keeps the plans of the last few structures it has seen (keyed on a 128-bit hash of exactly those inputs, per device)
 * together with the device workspaces of destroyed handles: BatchOptimizer::Optimize rebuilds its problem on every call
 * (batch_optimizer.cpp:57-70), and a rebuilt problem of known structure then only uploads its values. A structure that
 * differs in any of those inputs is planned afresh. CALICO_PLAN_CACHE=0 in the environment switches the cache off.
 * calico_plan_cache_stats: look-ups served from the cache / planned afresh since the process started, plans held.
 * calico_plan_cache_clear: drops the cached plans and workspaces; the device memory of plans no live handle uses, and
 * every allocation slab of the library nothing lives in any more, goes back to the driver (hipFree). Destroying a handle
 * does the same for all idle slabs but one per device. */
int32_t calico_plan_cache_stats(int64_t* hits_out, int64_t* misses_out, int64_t* entries_out);
int32_t calico_plan_cache_clear(void);

/* ---- solve ------------------------------------------------------------ */
/* Replaces ceres::Solve (batch_optimizer.cpp:72-73): Levenberg–Marquardt
 * trust region on the flattened problem, entirely on the device. On return the
 * summary, the iteration table and the parameter values are complete; a few
 * kernels of iterations enqueued ahead of the device (they exit at once) may
 * still be draining on the handle's stream -- later calls on the handle are
 * ordered behind them. */
int32_t calico_solve(calico_problem* p, const calico_solver_options* options,
                     calico_summary* summary);
/* Per-iteration table of the last solve; returns rows written via *n_out. */
int32_t calico_get_iterations(calico_problem* p, calico_iteration* out,
                              int32_t max_rows, int32_t* n_out);

/* Replaces Sensor::UpdateResiduals (camera.cpp:70-80, gyroscope.cpp:171-182,
 * accelerometer.cpp:58-69): sigma-weighted residuals WITHOUT the loss
 * function, in the order the residuals were added. out is n×dim (dim 2 for a
 * camera, 3 for an IMU sensor); valid[i]=0 where the projection failed
 * (the reference returns kInternal in that case; so does this call, after
 * filling both arrays). */
int32_t calico_get_residuals(calico_problem* p, int32_t sensor_id, double* out,
                             uint8_t* valid);
/* Spline initialisation on the device: BSpline::FitToData / FitSpline (bspline.hpp:19-37, 246-297), behind
 * Trajectory::FitSpline (trajectory.cpp:14-49). Least-squares control points (n_knots - order rows of 6) of the
 * spline on the given knot vector / basis matrices (as for calico_problem_set_spline) through n sorted samples
 * `data6` (n x 6: axis-angle, position) at `stamps`. The reference factors the dense n x n_ctrl design matrix by
 * column-pivoted QR; here the banded normal equations are assembled in a fixed order and solved by a banded Cholesky
 * (control points the samples do not determine, e.g. at an under-sampled trajectory end, are decoupled). kInvalidArgument for unsorted stamps,
 * stamps outside the valid knots or bad sizes; kUnimplemented when the band does not fit the on-chip solve. Needs no
 * problem handle. */
int32_t calico_fit_spline(int32_t device, int32_t order, int32_t n_knots, const double* knots, const double* basis,
                          int64_t n, const double* stamps, const double* data6, double* ctrl_out);

/* Outlier tags (Camera::MarkOutliersById / ClearOutliers, camera.cpp:281-301; outlier_ids_, camera.h:185): tagged
 * observations stay registered but are left out of the problem, as AddResidualsToProblem does (camera.cpp:121-124) --
 * no residual block, no residual, not counted in the summary. `is_outlier` has one byte per observation of the
 * sensor in insertion order; NULL clears all tags. */
int32_t calico_problem_set_outlier_mask(calico_problem* p, int32_t sensor, const uint8_t* is_outlier);
/* One pass of the demos' tagging loop (kalibr_multicam_demo.ipynb:666-674) on the device: residuals at the current
 * estimates without the loss function, every still-untagged observation of `sensor` with ||r|| > threshold (or that
 * cannot be evaluated) is tagged. *n_marked = number of observations tagged by this call. Follow with calico_solve. */
int32_t calico_mark_outliers(calico_problem* p, int32_t sensor, double threshold, int64_t* n_marked);
/* Residual statistics of a camera on the device (utils.py:12-50 ComputeRmseHeatmapAndFeatureCount, the notebooks'
 * per-region RMSE / feature count): the image (image_width x image_height pixels) is divided into num_rows x num_cols
 * bins by the measured pixel; rmse_out[row * num_cols + col] = sqrt(sum ||r||^2 / count) over the untagged
 * observations of the bin (residuals at the current estimates without the loss function, NaN for an empty bin like
 * the reference's 0/0), count_out the number of observations. Fixed-order reductions: results are reproducible. */
int32_t calico_residual_heatmap(calico_problem* p, int32_t sensor, int32_t image_width, int32_t image_height,
                                int32_t num_rows, int32_t num_cols, double* rmse_out, int64_t* count_out);

/* Sensor::Project for the registered observations (camera.cpp:155-208, gyroscope.cpp:56-82,
 * accelerometer.cpp:76-123): the model's prediction -- pixel (2) or IMU reading (3) per observation, in insertion
 * order -- at the current parameter values, i.e. exactly the quantity the residual compares the measurement with
 * (pose at stamp - latency, spline segment of the stamp). Evaluated by the residual kernel in prediction mode, so
 * measurements generated with it give residuals that are exactly zero. Points that do not project (the reference
 * skips them) come back with valid[i] = 0. `valid` may be NULL. */
int32_t calico_project(calico_problem* p, int32_t sensor, double* out, uint8_t* valid);

/* The outlier-tagging step that follows the path in the reference's demos
 * (kalibr_multicam_demo.ipynb:666-674 -> Camera::MarkOutliersById):
 * mask[i] = 1 iff the residual is valid and ||r_i|| <= threshold. */
int32_t calico_get_inlier_mask(calico_problem* p, int32_t sensor_id,
                               double threshold, uint8_t* mask);

/* ---- evaluation entry points (parity tests, benchmarks) --------------- */
/* Number of tangent columns of the reduced problem and their order:
 * [control points (6 each, spline order) | remaining free, used blocks in
 * block-id order]. */
int32_t calico_num_effective_parameters(calico_problem* p, int32_t* n_out);
/* One residual+Jacobian evaluation at the current parameters, as Ceres'
 * Evaluator would do for the LM loop (loss corrected, manifold projected):
 * cost, gradient (n), and the Gauss-Newton matrix JᵀJ expanded to dense
 * row-major n×n.  Any output pointer may be NULL. */
int32_t calico_evaluate(calico_problem* p, double* cost, double* gradient,
                        double* jtj_dense);

/* ---- multi-GPU -------------------------------------------------------- */
/* Observations shard across ranks; the only exchange is the sum of the
 * packed normal-equation buffer (and of the candidate cost).  The host owns
 * the communicator: it registers a callback that must all-reduce (sum)
 * n doubles in place at device address buf, ordered on HIP stream `stream`
 * (e.g. torch.distributed.all_reduce over RCCL). Without a callback the
 * handle is single-rank. */
/* Native exchange (the production path): the handle owns an RCCL communicator and issues ncclAllReduce itself, on its
 * own stream, between the kernels of an iteration -- no host code in the loop. Rank 0 draws the 128-byte id with
 * calico_comm_get_unique_id and hands it to every rank by whatever means the application has (MPI, a file, a
 * torch.distributed broadcast); every rank then calls calico_comm_init_rccl, which also selects its shard (like
 * calico_problem_set_shard). One process per GPU; the communicator lives until the handle is destroyed.
 * LOAD ORDER: librccl is dlopen()ed by the first calico_comm_* call -- an RCCL the process already holds is adopted
 * (RTLD_NOLOAD by soname), otherwise a private copy is loaded (RTLD_LOCAL; CALICO_RCCL_LIB names a particular file).
 * An application that brings its own RCCL (PyTorch does) must therefore load it BEFORE the first calico_comm_* call:
 * two RCCL images in one process have ended in a double free at exit. calico_comm_init_rccl warns on stderr when it
 * finds two.
 * Replaces nothing in the reference (it is single-process); SURVEY.md 8(b),(e). */
#define CALICO_COMM_ID_BYTES 128
int32_t calico_comm_get_unique_id(uint8_t* id_out /* CALICO_COMM_ID_BYTES */);
int32_t calico_comm_init_rccl(calico_problem* p, const uint8_t* id, int32_t rank, int32_t world_size);
/* What the handle's exchange really spans: rank and rank count as the RCCL communicator reports them (ncclCommCount; the
 * shard set by calico_problem_set_shard when there is no communicator) and the number of residual blocks this rank
 * evaluates out of the problem's total (finalises the problem). Any output pointer may be NULL. */
int32_t calico_comm_info(calico_problem* p, int32_t* rank_out, int32_t* world_out, int64_t* local_blocks_out,
                         int64_t* total_blocks_out);
/* Host-side exchange (tests, exotic transports): a callback instead of the communicator. */
typedef int32_t (*calico_allreduce_fn)(void* ctx, void* buf, int64_t n,
                                       void* stream);
int32_t calico_problem_set_allreduce(calico_problem* p, calico_allreduce_fn fn,
                                     void* ctx);
/* Every rank receives the WHOLE problem through the add_* calls above and
 * keeps only its shard on the device: rank r of world_size evaluates the
 * residual blocks whose spline segment lies in its time window (contiguous
 * windows balanced by block count). All ranks solve the same reduced system
 * after the all-reduce, so their parameter estimates stay identical. */
int32_t calico_problem_set_shard(calico_problem* p, int32_t rank,
                                 int32_t world_size);
/* Use an externally owned HIP stream (hipStream_t) for all work of this
 * handle, e.g. torch's current stream so the callback above is ordered. */
int32_t calico_problem_set_stream(calico_problem* p, void* stream);

/* ---- timing ----------------------------------------------------------- */
/* HIP-event timings accumulated since the last calico_set_phase_timing call (over
 * as many solves as followed it), milliseconds summed over launches, and launch
 * counts, for the named phase (the call waits for the stream when brackets are
 * still pending: calico_solve itself returns as soon as the device reports the
 * end of the solve): 0 jacobian evaluation (residual +
 * Jacobian + JᵀJ partials), 1 reduction of partials, 2 linear solve,
 * 3 cost-only evaluation, 4 LM control + update, 5 calibration: the same
 * event bracket around a trivial (~2 us) kernel, i.e. the overhead contained
 * in every per-launch figure of the other phases, 6 the launch inside phase 2
 * that solves the reduced system (dense reduced solve + first back-substitution
 * where they share a launch; not recorded while phase 2's bracket is open). `phase | 0x100` restricts the
 * sums to working launches: kernels of iterations enqueued ahead return at once
 * when the solve has terminated, and brackets shorter than a quarter of the
 * phase's longest one are left out. */
int32_t calico_get_phase_time(calico_problem* p, int32_t phase, double* ms,
                              int64_t* launches);
/* Which phases are bracketed by HIP events (bits 0..6: bit i = phase i; default: none) and, in bits 8..15, a sampling
 * interval N: only every N-th launch of a phase is bracketed (0 or 1: every launch). An event pair costs about 6 us of
 * stream time, so a throughput measurement brackets a sample of the launches, not all of them. The call resets the
 * accumulated times. */
int32_t calico_set_phase_timing(calico_problem* p, int32_t mask);

#ifdef __cplusplus
}
#endif
#endif /* CALICO_HIP_H_ */
