/* ORACLE — TEST INFRASTRUCTURE ONLY.  C ABI of the CPU restatement (oracle/ sources).
 * Loaded through ctypes by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg.  Never linked into or called by the product library. */
#ifndef SKERES_ORACLE_H
#define SKERES_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif

/* ceres::LinearSolverType numbering (ceres/types.h [ext]) */
enum { OR_DENSE_NORMAL_CHOLESKY = 0, OR_DENSE_QR = 1, OR_DENSE_SCHUR = 3 };
/* ceres::TerminationType */
enum { OR_CONVERGENCE = 0, OR_NO_CONVERGENCE = 1, OR_FAILURE = 2 };

#define OR_MAX_LOG 256

typedef struct or_options {
  int linear_solver_type;
  int max_num_iterations;
  double initial_trust_region_radius, max_trust_region_radius, min_trust_region_radius;
  double min_relative_decrease, min_lm_diagonal, max_lm_diagonal;
  double function_tolerance, gradient_tolerance, parameter_tolerance;
  int jacobi_scaling;
  int max_num_consecutive_invalid_steps;
  int num_threads; /* <=0: all cores */
  int cholesky_envelope; /* DENSE_SCHUR: skip the structural zeros outside the column envelope of the reduced system
                            (bit-identical factor; the like-for-like CPU baseline of the GPU's block envelope) */
} or_options;

typedef struct or_iteration {
  int iteration;
  double cost, cost_change, gradient_max_norm, step_norm, relative_decrease, trust_region_radius;
  int step_is_valid, step_is_successful;
  double time_s; /* wall clock when the iteration was logged */
} or_iteration;

typedef struct or_summary {
  double initial_cost, final_cost;
  int num_iterations, num_successful_steps, num_unsuccessful_steps, termination_type;
  int num_logged, num_threads_used;
  double total_time_s, t_linear_assemble_s, t_linear_cholesky_s, t_linear_backsub_s;
  char message[256];
  or_iteration iterations[OR_MAX_LOG];
} or_summary;

void or_options_default(or_options* o);
int or_functor_info(int id, int* num_residuals, int* num_blocks, int* num_consts, int* block_sizes);
int or_evaluate(int functor_id, const double* consts, double const* const* parameters,
                double* residuals, double** jacobians);
void or_angle_axis_rotate_point(const double* aa, const double* pt, double* out);
void or_angle_axis_to_rotation_matrix(const double* aa, double* R_rowmajor);
int or_solve(int num_blocks, const int* block_sizes, double* x, int num_res_blocks,
             const int* functor_ids, const double* consts, const int* const_off, const int* pidx,
             const int* pidx_off, const or_options* opt, or_summary* summary);
int or_solve_bal(int C, int P, int N, const int* cam_idx, const int* pt_idx, const double* obs,
                 double* x, const or_options* opt, or_summary* summary);
/* Rotation functions (oracle/rotation.hpp); op ids as sk_rotation_op in include/skeres_amd.h */
int or_rotation_apply(int op, int row_major, int jet_dim, const double* in, int n, double* out);
/* robust losses (oracle/loss.hpp): nodes of 5 doubles {type, a, b, f, g} */
void or_loss_evaluate(const double* loss_nodes, int root, double s, double* rho);
int or_solve_loss(int num_blocks, const int* block_sizes, double* x, int num_res_blocks,
                  const int* functor_ids, const double* consts, const int* const_off, const int* pidx,
                  const int* pidx_off, const double* loss_nodes, const int* block_loss,
                  const or_options* opt, or_summary* summary);
/* local parameterizations (oracle/parameterization.hpp): type 0 identity, 1 subset, 2 quaternion, 3 homogeneous vector,
 * 4 constant; -1 = none */
int or_solve_param(int num_blocks, const int* block_sizes, double* x, int num_res_blocks,
                   const int* functor_ids, const double* consts, const int* const_off, const int* pidx,
                   const int* pidx_off, const double* loss_nodes, const int* block_loss,
                   const int* param_type, const int* param_const_off, const int* param_const,
                   const or_options* opt, or_summary* summary);
int or_parameterization_local_size(int type, int size, int nconst);
void or_parameterization_plus(int type, int size, const int* constant, int nconst, const double* x, const double* delta, double* x_plus);
void or_parameterization_jacobian(int type, int size, const int* constant, int nconst, const double* x, double* J);
int or_solve_bal_loss(int C, int P, int N, const int* cam_idx, const int* pt_idx, const double* obs,
                      const double* loss_nodes, int loss_root, double* x, const or_options* opt, or_summary* summary);
/* ... with parameter-block state (ceres::Problem::SetParameterBlockConstant, SubsetParameterization): bit k of
 * cam_mask[i] / pt_mask[p] holds coordinate k of camera i / point p constant (all 9 / 3 bits: a constant block).
 * Either array may be NULL. */
int or_solve_bal_masks(int C, int P, int N, const int* cam_idx, const int* pt_idx, const double* obs,
                       const double* loss_nodes, int loss_root, const int* cam_mask, const int* pt_mask, double* x,
                       const or_options* opt, or_summary* summary);
int or_bal_evaluate(int C, int P, int N, const int* cam_idx, const int* pt_idx, const double* obs,
                    const double* x, double* r, double* F, double* E, double* cost);
int or_bal_reduced_system(int C, int P, int N, const int* cam_idx, const int* pt_idx, const double* obs,
                          const double* x, const double* D, int add_Dc, double* S, double* rhs);
/* cost of m dense rows (functor 10, consts m x 3 = seed, row, y) at x: cost-only evaluation, no Jacobian */
int or_dense_rows_cost(const double* consts, int m, const double* x, int n, int num_threads, double* cost);
int or_cholesky_lower(double* A, int n, int num_threads);
void or_cholesky_solve(const double* L, int n, double* b);
/* last_row: n entries, non-decreasing, last_row[j] >= j (column envelope) */
int or_cholesky_lower_envelope(double* A, int n, int num_threads, const int* last_row);
void or_cholesky_solve_envelope(const double* L, int n, double* b, const int* last_row);

#ifdef __cplusplus
}
#endif
#endif
