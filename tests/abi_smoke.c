/* Compile-time and link-time check of the C ABI from plain C99: include/skeres_amd.h must be a
 * valid C header and every signature used here must link against libskeres_amd.so — something the hand-written
 * ctypes table of skeres_amd/api.py cannot prove.  Mirrors what the reference's SWIG module generates its JNI thunks
 * from (ceres.i:95-125 DoubleArray / DoubleArraySlice / DoubleMatrix / StdVectorDoublePointer; CORE/Problem.scala:20-27;
 * EX/SimpleBundleAdjuster.scala:134-152).  Built and run by tests/test_capi_cpu.py:
 *   gcc -std=c99 -Wall -Werror -pedantic -Iinclude tests/abi_smoke.c -o <tmp>/abi_smoke -Lskeres_amd -lskeres_amd ...
 * Exit code 0 = every check passed.  Without a GPU sk_solve must fail with SK_ERR_NO_DEVICE (no CPU fallback); with
 * one it must solve the tiny problem. */
#include <stdio.h>
#include <string.h>
#include "skeres_amd.h"

#define CHECK(cond)                                                         \
  do {                                                                      \
    if (!(cond)) { fprintf(stderr, "abi_smoke: %s failed (line %d): %s\n", #cond, __LINE__, sk_last_error()); return 1; } \
  } while (0)

static int evaluate_cb(void* user, double const* const* parameters, double* residuals, double** jacobians) {
  /* r = 10 - x (EX/HelloWorld.scala:11-14 as an analytic cost function) */
  (void)user;
  residuals[0] = 10.0 - parameters[0][0];
  if (jacobians && jacobians[0]) jacobians[0][0] = -1.0;
  return 1;
}

int main(void) {
  /* ---- DoubleArray / DoubleArraySlice ---- */
  double* a = sk_array_new(12);
  double host[3] = {1.5, 2.5, 3.5}, back[3] = {0, 0, 0};
  double* slice;
  sk_ptrvec* v;
  double** pp;
  sk_loss_function *loss, *huber, *scaled;
  sk_cost_function *cost, *cb;
  sk_problem* problem;
  sk_options* options;
  sk_summary* summary;
  sk_local_parameterization* subset;
  sk_residual_block_id id = -1;
  double consts[2] = {0.0, 1.0};
  double m = 0.0, c = 0.0, x = 0.5;
  double* blocks[2];
  double* xb[1];
  int sizes[1] = {1};
  int constant[1] = {0};
  int rc, i;

  CHECK(a != NULL);
  for (i = 0; i < 12; ++i) sk_array_setitem(a, i, (double)i);
  CHECK(sk_array_getitem(a, 7) == 7.0);
  slice = sk_array_slice(a, 9);
  CHECK(slice == a + 9);
  sk_array_copy_in(slice, host, 3);
  sk_array_copy_out(slice, back, 3);
  CHECK(memcmp(host, back, sizeof host) == 0 && sk_array_getitem(a, 10) == 2.5);

  /* ---- StdVectorDoublePointer / DoubleMatrix ---- */
  v = sk_ptrvec_new();
  CHECK(v != NULL && sk_ptrvec_to_pointer_pointer(v) == NULL);
  sk_ptrvec_add(v, a);
  sk_ptrvec_add(v, slice);
  CHECK(sk_ptrvec_size(v) == 2 && sk_ptrvec_get(v, 1) == slice);
  sk_ptrvec_set(v, 0, slice);
  pp = sk_ptrvec_to_pointer_pointer(v);
  CHECK(pp != NULL && !sk_matrix_is_null(pp) && sk_matrix_row(pp, 0) == slice && sk_matrix_is_null(NULL));
  sk_ptrvec_free(v);

  /* ---- losses, cost functions, local parameterizations: construction and argument checks ---- */
  loss = sk_loss_trivial();
  huber = sk_loss_huber(1.0);
  scaled = sk_loss_scaled(huber, 2.0);
  CHECK(loss != NULL && huber != NULL && scaled != NULL);
  cost = sk_cost_function_new_autodiff(SK_FUNCTOR_EXPONENTIAL_RESIDUAL, consts, 2);
  CHECK(cost != NULL && sk_cost_function_num_residuals(cost) == 1 && sk_cost_function_num_parameter_blocks(cost) == 2 &&
        sk_cost_function_parameter_block_size(cost, 1) == 1);
  CHECK(sk_cost_function_new_autodiff(SK_FUNCTOR_EXPONENTIAL_RESIDUAL, consts, 3) == NULL); /* wrong number of captured doubles */
  CHECK(sk_cost_function_new_autodiff(999, consts, 2) == NULL);
  cb = sk_cost_function_new_callback(evaluate_cb, NULL, 1, sizes, 1);
  CHECK(cb != NULL && sk_cost_function_num_residuals(cb) == 1);
  subset = sk_local_parameterization_subset(3, constant, 1);
  CHECK(subset != NULL && sk_local_parameterization_global_size(subset) == 3 && sk_local_parameterization_local_size(subset) == 2);
  constant[0] = 5;
  CHECK(sk_local_parameterization_subset(3, constant, 1) == NULL); /* index out of range */
  sk_local_parameterization_free(subset);

  /* ---- Problem bookkeeping: identity by address, sizes remembered ---- */
  problem = sk_problem_new();
  CHECK(problem != NULL);
  blocks[0] = &m; blocks[1] = &c;
  rc = sk_problem_add_residual_block(problem, cost, loss, blocks, 2, &id);
  CHECK(rc == SK_OK && id == 0);
  rc = sk_problem_add_residual_block(problem, cost, NULL, blocks, 2, &id);
  CHECK(rc == SK_OK && id == 1);
  xb[0] = &x;
  rc = sk_problem_add_residual_block(problem, cb, scaled, xb, 1, &id);
  CHECK(rc == SK_OK && id == 2);
  CHECK(sk_problem_num_residual_blocks(problem) == 3 && sk_problem_num_parameter_blocks(problem) == 3 &&
        sk_problem_num_parameters(problem) == 3 && sk_problem_num_residuals(problem) == 3);
  rc = sk_problem_add_residual_block(problem, cost, loss, blocks, 1, &id); /* wrong number of parameter blocks */
  CHECK(rc == SK_ERR_INVALID_ARGUMENT && strlen(sk_last_error()) > 0);
  CHECK(sk_problem_set_parameter_block_constant(problem, &c) == SK_OK && sk_problem_set_parameter_block_variable(problem, &c) == SK_OK);

  /* ---- Options / Summary / solve ---- */
  options = sk_options_new();
  summary = sk_summary_new();
  CHECK(options != NULL && summary != NULL);
  CHECK(sk_options_set_linear_solver_type(options, SK_DENSE_QR) == SK_OK);
  CHECK(sk_options_set_linear_solver_type(options, SK_ITERATIVE_SCHUR) == SK_ERR_UNSUPPORTED);
  CHECK(sk_options_set_max_num_iterations(options, 25) == SK_OK && sk_options_set_max_num_iterations(options, -1) == SK_ERR_INVALID_ARGUMENT);
  CHECK(sk_options_set_minimizer_progress_to_stdout(options, 0) == SK_OK);
  rc = sk_solve(options, problem, summary);
  if (sk_device_count() == 0) {
    CHECK(rc == SK_ERR_NO_DEVICE);  /* no CPU fallback */
    CHECK(m == 0.0 && c == 0.0 && x == 0.5);  /* caller memory untouched */
    printf("abi_smoke ok (no device: sk_solve -> SK_ERR_NO_DEVICE)\n");
  } else {
    CHECK(rc == SK_OK);
    CHECK(sk_summary_final_cost(summary) < sk_summary_initial_cost(summary) && sk_summary_num_iterations(summary) >= 2);
    CHECK(x > 9.0 && x < 11.0);  /* r = 10 - x under a scaled Huber loss still ends at x = 10 */
    printf("abi_smoke ok (%s)\n", sk_summary_brief_report(summary));
  }
  sk_summary_free(summary);
  sk_options_free(options);
  sk_problem_free(problem);
  sk_cost_function_free(cb);
  sk_cost_function_free(cost);
  sk_loss_free(scaled);
  sk_loss_free(huber);
  sk_loss_free(loss);
  sk_array_free(a);
  return 0;
}
