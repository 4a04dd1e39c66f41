/* lmgpu.h — C ABI of the MI355X-native Levenberg-Marquardt inner loop.
 *
 * Drop-in boundary for GTSAM's LM hot path (citations relative to the reference tree):
 *   - LevenbergMarquardtOptimizer::iterate()      gtsam/nonlinear/LevenbergMarquardtOptimizer.h:103, .cpp:273-308
 *   - LevenbergMarquardtOptimizer::linearize()    gtsam/nonlinear/LevenbergMarquardtOptimizer.h:113
 *   - NonlinearOptimizer::solve()                 gtsam/nonlinear/NonlinearOptimizer.h:129-130, .cpp:132-178
 *   - NonlinearFactorGraph::error()/linearize()   gtsam/nonlinear/NonlinearFactorGraph.cpp:170-179, 239-278
 *   - GaussianFactorGraph::optimize()             gtsam/linear/GaussianFactorGraph.cpp:316-319
 * The reference has no FFI of its own: its extension points are C++ virtuals.  INTEGRATION.md shows the
 * thin C++ adapter (a LevenbergMarquardtOptimizer subclass) that binds these entry points.
 *
 * Conventions: plain C, opaque handle, caller-owned host pointers with explicit counts, int status return,
 * no exceptions across the boundary.  One HIP stream per handle; a handle is not thread-safe; handles are
 * independent.  All arithmetic is IEEE double.  The library REQUIRES a HIP device: there is no CPU fallback.
 *
 * Variable order: lmgpu_set_variables() receives the variables in ELIMINATION order (the Ordering of
 * NonlinearOptimizerParams::ordering, gtsam/nonlinear/NonlinearOptimizerParams.h:47); the position in that
 * list is the variable's "slot".  Every packed per-variable vector crossing this boundary (values, delta,
 * hessian diagonal) is the concatenation over slots 0..n-1.
 */
#ifndef LMGPU_H
#define LMGPU_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lmgpu_handle lmgpu_handle;

enum lmgpu_status {
  LMGPU_OK = 0,
  LMGPU_INDETERMINATE = 1, /* IndeterminantLinearSystemException (gtsam/linear/linearExceptions.h); see lmgpu_last_failed_slot */
  LMGPU_INVALID = 2,       /* invalid argument / call order */
  LMGPU_HIP_ERROR = 3      /* HIP or RCCL runtime failure; see lmgpu_last_error */
};

/* Variable types.  dim = tangent dimension, store = doubles per packed value.
 *   POSE2       dim 3  store 3   (x, y, theta)                                   gtsam/geometry/Pose2.h
 *   POSE3       dim 6  store 12  (R row-major 9, t 3)   tangent (omega, v)       gtsam/geometry/Pose3.h
 *   POINT3      dim 3  store 3
 *   CAM_BUNDLER dim 9  store 15  (R 9, t 3, f, k1, k2)  PinholeCamera<Cal3Bundler>; tangent (omega, v, f, k1, k2).
 *               The constant principal point (u0, v0) of Cal3Bundler is folded into the measurement by the host
 *               (z' = z - (u0, v0)); Cal3Bundler::retract keeps it constant (gtsam/geometry/Cal3Bundler.h:134-136).
 *   POINT2      dim 2  store 2                           planar landmark (gtsam/geometry/Point2.h), vector retract
 *   CAL3_S2     dim 5  store 5   (fx, fy, s, u0, v0)     Cal3_S2 as a VARIABLE (gtsam/geometry/Cal3_S2.h:106-119: vector retract / local)
 */
enum lmgpu_var_type {
  LMGPU_POSE2 = 0,
  LMGPU_POSE3 = 1,
  LMGPU_POINT3 = 2,
  LMGPU_CAM_BUNDLER = 3,
  LMGPU_POINT2 = 4,
  LMGPU_CAL3_S2 = 5,
  LMGPU_NUM_VAR_TYPES = 6
};

/* Factor types ("buckets" are keyed by (factor type, noise kind) = fixed block shape).
 *   type             arity rows  measurement doubles
 *   SFM              2     2     2   z                         GeneralSFMFactor<PinholeCamera<Cal3Bundler>,Point3>  gtsam/slam/GeneralSFMFactor.h:141-177
 *   BETWEEN_POSE2    2     3     3   (x,y,theta)               BetweenFactor<Pose2>   gtsam/slam/BetweenFactor.h:111-124
 *   BETWEEN_POSE3    2     6     12  (R row-major, t)          BetweenFactor<Pose3>
 *   PRIOR_POSE2      1     3     3                             PriorFactor<Pose2>     gtsam/nonlinear/PriorFactor.h:98-102
 *   PRIOR_POSE3      1     6     12
 *   PRIOR_POINT3     1     3     3
 *   PRIOR_CAM        1     9     15  (R, t, f, k1, k2)         PriorFactor<PinholeCamera<Cal3Bundler>>
 *   PROJECTION       2     2     7   (z, fx, fy, s, u0, v0)    GenericProjectionFactor<Pose3,Point3,Cal3_S2>  gtsam/slam/ProjectionFactor.h:138-165
 *   PROJECTION_BPS   2     2     19  (z, K, body_P_sensor R t) the same with body_P_sensor (ProjectionFactor.h:142-149: camera =
 *                                                              pose.compose(body_P_sensor), H1 chained with AdjointMap(body_P_sensor^-1))
 *   BEARING_RANGE_2D 2     2     2   (bearing angle, range)    BearingRangeFactor<Pose2,Point2>  gtsam/sam/BearingRangeFactor.h:33-77:
 *                                                              error = (wrap(bearing - measured), range - measured) with Pose2::bearing /
 *                                                              Pose2::range and their Jacobians, gtsam/geometry/Pose2.cpp:283-330
 *   SFM2             3     2     2   z                         GeneralSFMFactor2<Cal3_S2> (Pose3, Point3, Cal3_S2)  gtsam/slam/GeneralSFMFactor.h:208-262:
 *                                                              PinholeCamera<Cal3_S2>(pose, K).project(point, H1, H2, H3) - z; a point behind
 *                                                              the camera gives zero error and zero Jacobians (:251-260)
 *   PRIOR_CAL3_S2    1     5     5   (fx, fy, s, u0, v0)       PriorFactor<Cal3_S2>
 * Keys of a factor: `arity` per factor, in the reference's key order.
 */
enum lmgpu_factor_type {
  LMGPU_F_SFM = 0,
  LMGPU_F_BETWEEN_POSE2 = 1,
  LMGPU_F_BETWEEN_POSE3 = 2,
  LMGPU_F_PRIOR_POSE2 = 3,
  LMGPU_F_PRIOR_POSE3 = 4,
  LMGPU_F_PRIOR_POINT3 = 5,
  LMGPU_F_PRIOR_CAM = 6,
  LMGPU_F_PROJECTION = 7,
  LMGPU_F_PROJECTION_BPS = 8,
  LMGPU_F_BEARING_RANGE_2D = 9,
  LMGPU_F_SFM2 = 10,
  LMGPU_F_PRIOR_CAL3_S2 = 11,
  LMGPU_NUM_FACTOR_TYPES = 12
};

/* Noise models (gtsam/linear/NoiseModel.cpp).  Per-factor noise data, `rows` = factor rows:
 *   UNIT   0 doubles
 *   DIAG   rows doubles: INVERSE sigmas (Isotropic = all equal)          whiten: v .* invsigma   (:314-332, :653-665)
 *   GAUSS  rows*rows doubles: sqrt information R, row-major              whiten: R v             (:160-186)
 */
enum lmgpu_noise_kind { LMGPU_N_UNIT = 0, LMGPU_N_DIAG = 2, LMGPU_N_GAUSS = 3 };

#define LMGPU_FLAG_SPLIT_ROOT 1
typedef struct lmgpu_config {
  int32_t device;     /* HIP device ordinal */
  int32_t rank;       /* this process' rank among the cooperating handles (0 if single) */
  int32_t world_size; /* number of cooperating ranks (1 if single) */
  int32_t flags;      /* bit 0 (LMGPU_FLAG_SPLIT_ROOT, testing aid): take the multi-rank data path for the HBM fronts (partial
                         assembly in its own buffer, chunked all-reduce on the communication stream, fold-in before each panel)
                         even with world_size == 1, so that a one-rank RCCL communicator exercises it on a single GPU */
} lmgpu_config;

/* LevenbergMarquardtParams subset honoured by lmgpu_iterate / lmgpu_optimize
 * (gtsam/nonlinear/LevenbergMarquardtParams.h:35-157; defaults = SetLegacyDefaults :69-82). */
typedef struct lmgpu_lm_params {
  int32_t maxIterations;
  double relativeErrorTol, absoluteErrorTol, errorTol;
  double lambdaInitial, lambdaFactor, lambdaUpperBound, lambdaLowerBound;
  double minModelFidelity;
  int32_t diagonalDamping, useFixedLambdaFactor;
  double minDiagonal, maxDiagonal;
} lmgpu_lm_params;

/* LevenbergMarquardtState (gtsam/nonlinear/internal/LevenbergMarquardtState.h:42-68) */
typedef struct lmgpu_lm_state {
  double error;
  double lambda;
  double currentFactor;
  int32_t iterations;
  int32_t totalNumberInnerIterations;
} lmgpu_lm_state;

/* per-phase device time of the last lmgpu_iterate, milliseconds (HIP events on the handle's stream) */
typedef struct lmgpu_timings {
  double linearize_ms, eliminate_ms, backsub_ms, linear_error_ms, retract_error_ms, total_ms;
  int32_t inner_iterations;
} lmgpu_timings;

/* ---- lifetime ---- */
int lmgpu_create(const lmgpu_config* cfg, lmgpu_handle** out);
int lmgpu_destroy(lmgpu_handle* h);
const char* lmgpu_last_error(const lmgpu_handle* h);
int lmgpu_last_failed_slot(const lmgpu_handle* h); /* slot of the first frontal variable of the failing front */

/* ---- structure (once per graph + ordering; replaces the symbolic work eliminateMultifrontal redoes per solve,
 *      gtsam/inference/EliminateableFactorGraph-inst.h:123-146) ---- */
int lmgpu_set_variables(lmgpu_handle* h, int32_t n_vars, const uint64_t* keys, const int32_t* types);
/* graph_index[i] = index of factor i in the NonlinearFactorGraph (defines VariableIndex order and error-sum order);
 * var_slots: n x arity slots; meas: n x measurement doubles; noise: n x noise doubles (NULL for UNIT). */
int lmgpu_add_factor_bucket(lmgpu_handle* h, int32_t factor_type, int32_t n, const int32_t* graph_index, const int32_t* var_slots,
                            const double* meas, int32_t noise_kind, const double* noise);
/* The same with a noiseModel::Robust wrapped around the Gaussian model of every factor of the bucket
 * (gtsam/linear/NoiseModel.h:663-760): linearize multiplies the whitened [A b] of a factor by sqrt(weight(||b||))
 * (Robust::WhitenSystem NoiseModel.cpp:705-723 -> mEstimator::Base::reweight, Block scheme, LossFunctions.cpp:61-76) and the
 * factor's error is loss(||whitened error||) (NoiseModel.h:717-725).  robust_k = the m-estimator's tuning constant. */
enum lmgpu_robust_kind {
  LMGPU_ROBUST_NONE = 0,
  LMGPU_ROBUST_FAIR = 1,           /* LossFunctions.cpp:146-155 */
  LMGPU_ROBUST_HUBER = 2,          /* :179-191 */
  LMGPU_ROBUST_CAUCHY = 3,         /* :217-224 */
  LMGPU_ROBUST_TUKEY = 4,          /* :250-266 */
  LMGPU_ROBUST_WELSCH = 5,         /* :289-297 */
  LMGPU_ROBUST_GEMAN_MCCLURE = 6,  /* :320-331 */
  LMGPU_ROBUST_DCS = 7,            /* :354-373 */
  LMGPU_ROBUST_L2_WITH_DEAD_ZONE = 8 /* :400-412 */
};
int lmgpu_add_factor_bucket_robust(lmgpu_handle* h, int32_t factor_type, int32_t n, const int32_t* graph_index, const int32_t* var_slots,
                                   const double* meas, int32_t noise_kind, const double* noise, int32_t robust_kind, double robust_k);
int lmgpu_finalize_structure(lmgpu_handle* h);

/* ---- values ---- */
int lmgpu_set_values(lmgpu_handle* h, const double* packed_values);
int lmgpu_get_values(lmgpu_handle* h, double* packed_values);
/* device-side snapshot / restore of the values (e.g. keep the initial estimate around to restart an optimisation) */
int lmgpu_save_values(lmgpu_handle* h);
int lmgpu_restore_values(lmgpu_handle* h);
int lmgpu_total_dim(const lmgpu_handle* h);   /* sum of tangent dims */
int lmgpu_total_store(const lmgpu_handle* h); /* doubles in packed values */

/* ---- the hot path, piecewise (what LevenbergMarquardtOptimizer::tryLambda calls) ---- */
int lmgpu_error(lmgpu_handle* h, double* total);  /* NonlinearFactorGraph::error at the current values */
int lmgpu_linearize(lmgpu_handle* h);             /* NonlinearFactorGraph::linearize -> device-resident whitened [A|b] per factor */
/* buildDampedSystem + solve + linear.error(0), linear.error(delta).  delta_packed may be NULL. */
int lmgpu_solve(lmgpu_handle* h, double lambda, int32_t diagonal_damping, double min_diag, double max_diag, double* delta_packed,
                double* lin_err0, double* lin_err1);
int lmgpu_retract(lmgpu_handle* h, const double* delta_packed); /* values <- values.retract(delta); NULL = last solve's delta */
int lmgpu_hessian_diagonal(lmgpu_handle* h, double* diag_packed); /* GaussianFactorGraph::hessianDiagonal */

/* ---- the hot path, whole (drop-in for LevenbergMarquardtOptimizer::iterate / NonlinearOptimizer::optimize) ---- */
int lmgpu_lm_init(lmgpu_handle* h, const lmgpu_lm_params* p, lmgpu_lm_state* state_out);
int lmgpu_iterate(lmgpu_handle* h, const lmgpu_lm_params* p, lmgpu_lm_state* inout);
int lmgpu_optimize(lmgpu_handle* h, const lmgpu_lm_params* p, lmgpu_lm_state* inout);

/* GaussNewtonOptimizer::iterate (gtsam/nonlinear/GaussNewtonOptimizer.cpp:44-66) on the same device-resident graph:
 * linearize, solve the undamped system, retract, error of the new values; state->error / iterations are updated, lambda
 * is not used.  LMGPU_INDETERMINATE where the reference throws IndeterminantLinearSystemException.
 * lmgpu_gn_optimize = NonlinearOptimizer::defaultOptimize (NonlinearOptimizer.cpp:62-117) around it; of `p` only
 * maxIterations and the three error tolerances are read (NonlinearOptimizerParams). */
int lmgpu_gn_iterate(lmgpu_handle* h, lmgpu_lm_state* inout);
int lmgpu_gn_optimize(lmgpu_handle* h, const lmgpu_lm_params* p, lmgpu_lm_state* inout);

/* DoglegOptimizer::iterate (gtsam/nonlinear/DoglegOptimizer.cpp:84-126, multifrontal elimination, ONE_STEP_PER_ITERATION as there;
 * DoglegOptimizerImpl.h:139-254, DoglegOptimizerImpl.cpp:26-91): undamped solve, steepest-descent point from the Bayes tree
 * (GaussianFactorGraph::optimizeGradientSearch, GaussianFactorGraph.cpp:381-406), dogleg point inside the trust region, gain
 * ratio rho = (f(x) - f(x + dx_d)) / (M(0) - M(dx_d)) with M the Bayes tree's error, radius update.  The trust radius travels in
 * state->lambda (initialise it to DoglegParams::deltaInitial, default 1.0; state->error as for LM).  Single-rank. */
int lmgpu_dl_iterate(lmgpu_handle* h, lmgpu_lm_state* inout);
int lmgpu_dl_optimize(lmgpu_handle* h, const lmgpu_lm_params* p, lmgpu_lm_state* inout);
int lmgpu_get_timings(const lmgpu_handle* h, lmgpu_timings* out);

/* Per-kernel device time (HIP events on the handle's stream around each launch), accumulated since
 * lmgpu_set_kernel_timing(h, 1).  work[] is the ALGORITHMIC work of the launches in the category:
 * bytes for LINEARIZE (232 B per SFM factor + every camera / point row once) and ALLREDUCE,
 * FP64 flop for PANEL and SYRK (kb * m * (m + 1) per trailing update = the n^3/3 of the dense front, plus kb^3/3 + kb^2 cols of a
 * panel factored in the same launch). */
enum lmgpu_kernel_category {
  LMGPU_KT_LINEARIZE = 0,   /* sfm_linearize_kernel */
  LMGPU_KT_LDS_FRONT = 1,   /* lds_front_kernel (assemble + partial Cholesky of one small front per workgroup) */
  LMGPU_KT_HBM_ASSEMBLE = 2,/* memset + factor / child extend-add / Schur gather + damping of an HBM front */
  LMGPU_KT_PANEL = 3,       /* panel factorisations launched on their own: panel_dataflow_kernel, diag_potrf_kernel + panel_trsm_kernel */
  LMGPU_KT_SYRK = 4,        /* step_kernel = trailing update (v_mfma_f64_16x16x4_f64) + the next panel in the same launch (steps outside a chained launch); syrk_mfma_kernel */
  LMGPU_KT_BACKSUB_HBM = 5,
  LMGPU_KT_BACKSUB_LDS = 6,
  LMGPU_KT_LINEAR_ERROR = 7,
  LMGPU_KT_RETRACT_ERROR = 8,
  LMGPU_KT_ALLREDUCE = 9,
  LMGPU_KT_CHAIN = 10,      /* chain_kernel = all fused steps of a dense front (updates + panel factorisations) as ONE launch */
  LMGPU_KT_NUM = 11
};
/* on: 0 off; 1 every category; 2 only LINEARIZE and CHAIN / SYRK (the roofline kernels; SYRK is the per-step form of CHAIN's work): an event between two launches costs a few
 * microseconds of idle device time, and a throughput measurement should not pay that for the eleven categories it does not report */
int lmgpu_set_kernel_timing(lmgpu_handle* h, int32_t on);
int lmgpu_get_kernel_times(const lmgpu_handle* h, double* ms /*[LMGPU_KT_NUM]*/, double* work /*[LMGPU_KT_NUM]*/, int64_t* launches /*[LMGPU_KT_NUM]*/);

/* ---- parity taps ---- */
/* whitened Jacobian of graph factor `graph_index`, column-major rows x (sum dims + 1) like the reference's
 * VerticalBlockMatrix ([A1 A2 b]); out may be NULL to query the shape. */
int lmgpu_get_jacobian(lmgpu_handle* h, int32_t graph_index, double* out, int32_t* rows, int32_t* cols);
/* The whole linearization in one call: what LevenbergMarquardtOptimizer::iterate() RETURNS (the GaussianFactorGraph of
 * gtsam/nonlinear/NonlinearOptimizer.h:136, LevenbergMarquardtOptimizer.cpp:281,307; read by tests/testNonlinearOptimizer.cpp:282).
 * Every factor's whitened [A1 .. Ak b] back to back in ascending graph-index order (index-preserving like
 * NonlinearFactorGraph::linearize), column-major rows[i] x cols[i] at out + offsets[i]; offsets has *n_out + 1 entries.
 * Any output may be NULL (out == NULL: shapes only, no device needed).  One device copy per factor bucket. */
int lmgpu_get_jacobians(lmgpu_handle* h, int32_t* n_out, int32_t* graph_index, int32_t* rows, int32_t* cols, int64_t* offsets, double* out);
int lmgpu_num_fronts(const lmgpu_handle* h);
/* info8: n_keys, n_frontal_keys, nf (rows of [R S d]), n (cols), parent front (-1 root), class (0 = LDS front, 1 = HBM front),
 *        owner rank (-1 = replicated on every rank), level (0 = leaf) */
int lmgpu_front_info(const lmgpu_handle* h, int32_t front, int32_t* info8);
/* slots: the front's variables in Scatter order (gtsam/linear/Scatter.cpp:39-73); RSd: column-major nf x n */
int lmgpu_get_front(lmgpu_handle* h, int32_t front, int32_t* slots, double* RSd_colmajor);

/* ---- multi-GPU (points/subtrees sharded over ranks; separator contributions summed with RCCL over xGMI) ---- */
/* rank 0 fills id[128] (ncclUniqueId bytes); the caller broadcasts it; every rank then calls lmgpu_comm_init. */
int lmgpu_comm_unique_id(char id128[128]);
int lmgpu_comm_init(lmgpu_handle* h, const char id128[128]);

/* Marginals(graph, values).marginalCovariance(variable)  (gtsam/nonlinear/Marginals.cpp:28-33 constructor: linearize at the
 * solution and eliminate into a Bayes tree; :124-127 marginalCovariance = inverse of the marginal information :109-121):
 * the dim x dim covariance (row-major) of the variable in `slot` at the handle's current values, i.e. its diagonal block of
 * (A^T A)^-1.  Computed with the path's own two kernels: per column one elimination (lambda = 0) + back-substitution of the
 * linearized system whose gradient is replaced by a unit vector.  The stored linearization is consumed (the next
 * lmgpu_solve needs lmgpu_linearize again).  LMGPU_INDETERMINATE where the reference throws
 * IndeterminantLinearSystemException (singular information matrix).  One GPU only (world_size == 1). */
int lmgpu_marginal_covariance(lmgpu_handle* h, int32_t slot, double* cov /* dim x dim */);
/* Marginals::jointMarginalCovariance(variables) (gtsam/nonlinear/Marginals.cpp:130-137): the blocks of (A^T A)^-1 for the
 * variables in `slots` (distinct), as one D x D row-major matrix, D = sum of their dims, block order = order of `slots`
 * (the reference's JointMarginal::fullMatrix orders the blocks by Key; the caller chooses here).  Cost: D unit-gradient solves. */
int lmgpu_joint_marginal_covariance(lmgpu_handle* h, int32_t nslots, const int32_t* slots, double* cov /* D x D */);

/* Host-only self-test (no GPU work): the ticket order the library gives the chained factorisation launch of a dense front
 * with n columns (nf frontal) for the steps i0 .. i0 + nsteps - 1 (tile rows beyond far_pct percent scheduled late; far_pct + 1000: the
 * default schedule, whose update tiles apply two steps per pass) is a permutation of all logical workgroups in which every
 * in-launch dependency points to an earlier ticket.  0 = valid.  No reference counterpart (the reference factors a front with
 * one Eigen LLT call, gtsam/base/cholesky.cpp:108-159); it exists so that the CPU test-suite can check the scheduler. */
int lmgpu_selftest_chain_schedule(int n, int nf, int i0, int nsteps, int far_pct);

/* ---- ISAM2 (gtsam/nonlinear/ISAM2.h): incremental smoothing on a device-resident Bayes tree (BASELINE config 5) ----
 * ISAM2::update(newFactors, newTheta) (gtsam/nonlinear/ISAM2.cpp:419-480) = lmgpu_isam2_add_variables + lmgpu_isam2_add_factors
 * (the pending input) + lmgpu_isam2_update: add the variables, (every relinearizeSkip-th update) refresh delta, mark and
 * relinearize the variables whose delta exceeds relinearizeThreshold, relinearize the affected factor subset on the device,
 * remove the top of the Bayes tree, re-eliminate it with the cached boundary factors of the orphaned subtrees under a constrained
 * COLAMD ordering (:250-362; batch fallback when >= 65 % of the variables are affected, :156-159), and keep the tree on the device.
 * ISAM2Params honoured: ISAM2GaussNewtonParams::wildfireThreshold, relinearizeThreshold (one double), relinearizeSkip,
 * enableRelinearization (gtsam/nonlinear/ISAM2Params.h:133-246); Cholesky factorisation; cacheLinearizedFactors semantics.
 * ISAM2UpdateParams (gtsam/nonlinear/ISAM2UpdateParams.h:30-90) through lmgpu_isam2_update_with: removeFactorIndices, constrainedKeys,
 * noRelinKeys, extraReelimKeys, force_relinearize, forceFullSolve.
 * relinearizeThreshold as FastMap<char, Vector> and enablePartialRelinearizationCheck: the two setters below.
 * ISAM2DoglegParams: lmgpu_isam2_set_dogleg.
 * Robust noise models: lmgpu_isam2_add_factors_robust.
 * ISAM2::marginalizeLeaves: lmgpu_isam2_marginalize_leaves; ISAM2Params::findUnusedFactorSlots: lmgpu_isam2_set_find_unused_factor_slots.
 * Not bound: QR, newAffectedKeys (smart factors).
 * updateDelta (ISAM2.cpp:701-719) is scheduled by the library: Gauss-Newton mode runs it BEHIND an update's elimination, under the update's
 * one wait, when delta is going to be read before the next elimination anyway (the next update checks relinearization, or the caller read an
 * estimate after the previous update); otherwise where the reference runs it (at the next reader).  Same delta either way; the one visible
 * difference: an indeterminate back-substitution would be returned by that update instead of the next call (the elimination's own pivot
 * test reports such a system first).
 *
 * The fill-reducing ordering is a boundary input like in the batch path, but here it is needed per update: the caller hands over
 * ITS ccolamd (the reference side: the one Ordering::ColamdConstrained calls, gtsam/inference/Ordering.cpp:50-125) as a callback:
 *   int fn(user, n_rows, n_cols, col_ptr[n_cols + 1], row_idx[nnz], cmember[n_cols], perm_out[n_cols]) -> 1 on success,
 * to be run with GTSAM's knobs (CCOLAMD_DENSE_ROW = CCOLAMD_DENSE_COL = -1, Ordering.cpp:94-97); perm_out[j] = the column eliminated j-th. */
typedef struct lmgpu_isam2 lmgpu_isam2;
typedef int (*lmgpu_ccolamd_fn)(void* user, int32_t n_rows, int32_t n_cols, const int32_t* col_ptr, const int32_t* row_idx, const int32_t* cmember,
                                int32_t* perm_out);
typedef struct lmgpu_isam2_params {
  double relinearizeThreshold;  /* default 0.1 */
  int32_t relinearizeSkip;      /* default 10 */
  int32_t enableRelinearization; /* default 1 */
  double wildfireThreshold;     /* ISAM2GaussNewtonParams, default 0.001 */
} lmgpu_isam2_params;
/* ISAM2Result subset (gtsam/nonlinear/ISAM2Result.h:60-93) + whether the batch fallback ran */
typedef struct lmgpu_isam2_result {
  int32_t variablesRelinearized, variablesReeliminated, factorsRecalculated, cliques, batch;
} lmgpu_isam2_result;
int lmgpu_isam2_create(const lmgpu_config* cfg, const lmgpu_isam2_params* params, lmgpu_ccolamd_fn ccolamd, void* user, lmgpu_isam2** out);
int lmgpu_isam2_destroy(lmgpu_isam2* s);
const char* lmgpu_isam2_last_error(const lmgpu_isam2* s);
uint64_t lmgpu_isam2_last_failed_key(const lmgpu_isam2* s); /* LMGPU_INDETERMINATE: first frontal key of the failing clique */
/* ISAM2Params::relinearizeThreshold as FastMap<char, Vector> (gtsam/nonlinear/ISAM2Params.h:139-141, 169-181): n entries, entry i = the
 * Symbol character chrs[i] with dims[i] per-dof thresholds, the vectors back to back in `values`.  A variable is then relinearized
 * when any |delta_i| > threshold_i (strictly; ISAM2-impl.h:266, 375) of its character's vector; an update that meets a variable
 * without a vector of its dimension returns LMGPU_INVALID (the reference throws, :258-262).  n = 0: the scalar threshold again. */
int lmgpu_isam2_set_relinearize_thresholds(lmgpu_isam2* s, int32_t n, const char* chrs, const int32_t* dims, const double* values);
/* ISAM2Params::optimizationParams = ISAM2DoglegParams(initialDelta, wildfireThreshold, adaptationMode) (ISAM2Params.h:68-110) instead of
 * ISAM2GaussNewtonParams: updateDelta becomes one iteration of Powell's dog leg (ISAM2.cpp:739-779; adaptationMode as
 * DoglegOptimizerImpl::TrustRegionAdaptationMode: 0 SEARCH_EACH_ITERATION (the default), 1 SEARCH_REDUCE_ONLY, 2 ONE_STEP_PER_ITERATION).
 * To be called before the first variable is added.  lmgpu_isam2_get_dogleg_delta: the current trust-region radius (doglegDelta_). */
int lmgpu_isam2_set_dogleg(lmgpu_isam2* s, double initialDelta, double wildfireThreshold, int32_t adaptationMode);
double lmgpu_isam2_get_dogleg_delta(const lmgpu_isam2* s);
/* ISAM2Params::evaluateNonlinearError (ISAM2Params.h:200-203): every update also evaluates the nonlinear error of the whole graph at
 * calculateEstimate() after the new factors have been added (ISAM2Result::errorBefore, ISAM2.cpp:444-446) and at its end (errorAfter,
 * :481-483) -- each through calculateEstimate(), i.e. with the back-substitution brought up to date first, like the reference.
 * lmgpu_isam2_get_errors returns the two of the last update; lmgpu_isam2_error is getFactorsUnsafe().error(values) on demand,
 * which = 0 at calculateEstimate(), 2 at the linearization point (the error kernels of the batch path on every factor bucket). */
int lmgpu_isam2_set_evaluate_nonlinear_error(lmgpu_isam2* s, int32_t enable);
int lmgpu_isam2_get_errors(const lmgpu_isam2* s, double* error_before, double* error_after);
int lmgpu_isam2_error(lmgpu_isam2* s, int32_t which, double* out);
/* ISAM2Params::enablePartialRelinearizationCheck (ISAM2Params.h:214-222): the check walks down from the roots and stops below a
 * clique none of whose variables is above its threshold (CheckRelinearizationPartial, ISAM2-impl.h:246-331) */
int lmgpu_isam2_set_partial_relinearization_check(lmgpu_isam2* s, int32_t enable);
/* newTheta of the next update: packed values like lmgpu_set_values (store doubles per type, in the order given) */
int lmgpu_isam2_add_variables(lmgpu_isam2* s, int32_t n, const uint64_t* keys, const int32_t* types, const double* packed_values);
/* newFactors of the next update, appended in call order (= their order in the NonlinearFactorGraph); keys: n x arity Keys */
int lmgpu_isam2_add_factors(lmgpu_isam2* s, int32_t factor_type, int32_t n, const uint64_t* keys, const double* meas, int32_t noise_kind,
                            const double* noise);
/* the same with noiseModel::Robust(mEstimator(robust_k), model) around the Gaussian model of every one of the n factors (lmgpu_robust_kind;
 * gtsam/linear/NoiseModel.h:663-760): relinearization reweights [A b] by sqrt(weight(||b||)) (Robust::WhitenSystem, Block scheme) and
 * evaluateNonlinearError uses the m-estimator's loss, exactly as in the batch path (lmgpu_add_factor_bucket_robust) */
int lmgpu_isam2_add_factors_robust(lmgpu_isam2* s, int32_t factor_type, int32_t n, const uint64_t* keys, const double* meas, int32_t noise_kind,
                                   const double* noise, int32_t robust_kind, double robust_k);
int lmgpu_isam2_update(lmgpu_isam2* s, int32_t force_relinearize, lmgpu_isam2_result* out);
/* ISAM2::update(newFactors, newTheta, const ISAM2UpdateParams&) (gtsam/nonlinear/ISAM2.h:176-186, ISAM2UpdateParams.h:30-90).
 * removeFactorIndices: positions in the factor list (getFactorsUnsafe(); the factors of update k start at lmgpu_isam2_num_factors()
 *   before it); a removed factor leaves an empty slot (indices never shift), removing an empty slot is a no-op, an index beyond the
 *   list (including the factors this very update adds) is LMGPU_INVALID.  A variable that loses its last factor leaves the system
 *   (ISAM2Result::unusedKeys -> ISAM2::removeVariables, ISAM2.cpp:385-398): see lmgpu_isam2_get_unused_keys.
 * constrainedKeys (has_constrained != 0: the optional is engaged, even with n_constrained == 0): key -> group for the constrained
 *   COLAMD call instead of "the observed keys last" (ISAM2.cpp:205-218, 318-340).
 * noRelinKeys: never relinearized by this update; extraReelimKeys: re-eliminated although no new factor touches them;
 * forceFullSolve: wildfire threshold 0 and every variable relinearized (the reference's debugging switch). */
typedef struct lmgpu_isam2_update_params {
  int32_t n_remove;
  const uint64_t* removeFactorIndices;
  int32_t has_constrained, n_constrained;
  const uint64_t* constrainedKeys;
  const int32_t* constrainedGroups;
  int32_t n_no_relin;
  const uint64_t* noRelinKeys;
  int32_t n_extra_reelim;
  const uint64_t* extraReelimKeys;
  int32_t force_relinearize, forceFullSolve;
} lmgpu_isam2_update_params;
int lmgpu_isam2_update_with(lmgpu_isam2* s, const lmgpu_isam2_update_params* params, lmgpu_isam2_result* out);
/* ISAM2Result::unusedKeys of the last update, ascending; returns their number (keys_out may be NULL) */
int lmgpu_isam2_get_unused_keys(const lmgpu_isam2* s, uint64_t* keys_out);
/* 1 when slot i of the factor list holds a factor (getFactorsUnsafe().exists(i)) */
int lmgpu_isam2_factor_exists(const lmgpu_isam2* s, int32_t i);
/* ISAM2Params::findUnusedFactorSlots (gtsam/nonlinear/ISAM2Params.h:225): new factors -- of an update and of marginalizeLeaves -- fill the
 * empty slots of the factor list from the front before they extend it (FactorGraph::add_factors, gtsam/inference/FactorGraph-inst.h:109-137). */
int lmgpu_isam2_set_find_unused_factor_slots(lmgpu_isam2* s, int32_t enable);
/* ISAM2::marginalizeLeaves(leafKeys, &marginalFactorsIndices, &deletedFactorsIndices) (gtsam/nonlinear/ISAM2.h:198-222, ISAM2.cpp:487-720):
 * the variables must be leaves of the Bayes tree (the caller orders them first with constrainedKeys, as with the reference: see
 * IncrementalFixedLagSmoother).  The marginal on what they were connected to stays in the factor list as LinearContainerFactors (constant
 * Hessian factors in device memory: they enter later eliminations, are never relinearized and add nothing to the nonlinear error), their
 * keys become fixedVariables_ (lmgpu_isam2_get_fixed_variables, ascending; returns their number), the summarised factors' slots empty.
 * n_marginal_out / n_deleted_out: sizes of the two index lists, read with lmgpu_isam2_get_marginalize_result (either may be NULL).
 * A key that is not a leaf is refused BEFORE anything changes (LMGPU_INVALID, lmgpu_isam2_last_failed_key names the variable that is in
 * the way) -- the reference checks this in debug builds only and leaves the object unusable.
 * lmgpu_isam2_get_marginal_factor: parity tap -- slot i as a marginal factor: its keys, their dimensions and the augmented information
 * matrix ((sum dims + 1)^2, column-major, symmetric); returns the number of keys, -1 when slot i holds no marginal factor. */
int lmgpu_isam2_marginalize_leaves(lmgpu_isam2* s, int32_t n, const uint64_t* leaf_keys, int32_t* n_marginal_out, int32_t* n_deleted_out);
int lmgpu_isam2_get_marginalize_result(const lmgpu_isam2* s, uint64_t* marginal_idx_out, uint64_t* deleted_idx_out);
int lmgpu_isam2_get_fixed_variables(const lmgpu_isam2* s, uint64_t* keys_out);
int lmgpu_isam2_get_marginal_factor(lmgpu_isam2* s, int32_t i, uint64_t* keys_out, int32_t* dims_out, double* info_colmajor);
int lmgpu_isam2_num_variables(const lmgpu_isam2* s); /* live variables = theta_.size() */
int lmgpu_isam2_num_factors(const lmgpu_isam2* s);   /* slots of the factor list, removed ones included */
/* which: 0 = calculateEstimate (ISAM2.cpp:748-754), 1 = calculateBestEstimate (:763-766), 2 = getLinearizationPoint.
 * Variables ascending by Key (the reference's Values order); any of the outputs may be NULL. */
int lmgpu_isam2_get_values(lmgpu_isam2* s, int32_t which, uint64_t* keys_out, int32_t* types_out, double* packed_out);
/* calculateEstimate(Key) (ISAM2.cpp:757-760) for which = 0, the linearization point of one variable for which = 2: only this variable
 * is retracted and downloaded (store doubles of its type, lmgpu.h packings); LMGPU_INVALID for an unknown key */
int lmgpu_isam2_get_value(lmgpu_isam2* s, int32_t which, uint64_t key, int32_t* type_out, double* packed_out);
int lmgpu_isam2_get_delta(lmgpu_isam2* s, double* packed); /* getDelta (:776-779), ascending by Key */
/* ISAM2::marginalCovariance(key) (gtsam/nonlinear/ISAM2.h:253-257: the inverse of BayesTree::marginalFactor(key)'s information; the
 * linearization point's covariance block of the variable, dim x dim): two triangular solves per column, on the device, along the path
 * from the variable's clique to its root.  LMGPU_INVALID for a variable that is not in the tree. */
int lmgpu_isam2_marginal_covariance(lmgpu_isam2* s, uint64_t key, double* cov);
/* parity taps: the Bayes tree depth-first from the roots (children in order).  lmgpu_isam2_num_cliques takes the snapshot the
 * other two index; info5: n_keys, n_frontal_keys, nf, n, parent (index in the snapshot, -1 root); RSd column-major nf x n */
int lmgpu_isam2_num_cliques(lmgpu_isam2* s);
int lmgpu_isam2_clique_info(const lmgpu_isam2* s, int32_t i, int32_t* info5);
int lmgpu_isam2_get_clique(lmgpu_isam2* s, int32_t i, uint64_t* keys, double* RSd_colmajor);

#ifdef LMGPU_TEST_HOOKS
/* TEST HOOKS -- exported by liblmgpu_test.so only (csrc/Makefile builds it with -DLMGPU_TEST_HOOKS); the product library has none of them.
 * In-process stand-in for the RCCL communicator: W handles created by W threads of ONE process on one
 * device sum their buffers through a host rendezvous, at exactly the call sites where the RCCL path all-reduces.
 * Lets the sharded LM loop run end to end on a single GPU.  Every rank must drive its handle from its own thread. */
typedef struct lmgpu_local_group lmgpu_local_group;
int lmgpu_local_group_create(int32_t world_size, lmgpu_local_group** out);
int lmgpu_local_group_destroy(lmgpu_local_group* g);
int lmgpu_comm_init_local(lmgpu_handle* h, lmgpu_local_group* g);
#endif

/* ---- micro-benchmarks used by bench.py for roofline peaks (device-only, no graph needed) ---- */
int lmgpu_peak_mfma_f64(int32_t device, int32_t iters, double* tflops);
/* the same loop with n_acc (4, 8, or 16 = the 4 x 4 accumulators and 4 + 4 operand registers of the update tile's 64x64 wave tile, two
 * workgroups per CU) independent accumulators, measured after ~1 s of back-to-back launches, together with the shader
 * clock the chip HOLDS under that load (in-kernel s_memtime / s_memrealtime stamps) and the resulting flop per clock and SIMD:
 * 32 = one v_mfma_f64_16x16x4_f64 per 64 cycles, the issue rate behind the 78.6 TFLOP/s datasheet figure at 2.4 GHz */
int lmgpu_peak_mfma_f64_clock(int32_t device, int32_t iters, int32_t n_acc, double* tflops, double* sclk_mhz, double* flop_per_clk_simd);
int lmgpu_peak_hbm_copy(int32_t device, int64_t bytes, int32_t iters, double* gbps);

#ifdef __cplusplus
}
#endif
#endif /* LMGPU_H */
