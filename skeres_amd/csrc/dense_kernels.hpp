// Launchers of dense_kernels.hip (generic dense-Jacobian path).
#pragma once
#include <hip/hip_runtime.h>

#include "common.hpp"
#include "loss.hpp"

namespace sk {

struct DenseEvalArgs {
  int count;              // residual blocks of this functor
  const int* blocks;      // [count] residual block ids
  const double* consts;   // all constants
  const size_t* const_off;  // [num blocks]
  const int* pidx;        // x offsets of the parameter blocks, flattened
  const size_t* pidx_off;   // [num blocks]
  const int* res_off;     // first residual row of each block
  const double* x;        // parameters
  const double* scale;    // Jacobi column scaling
  double* r;              // [m]
  double* J;              // [m][n]
  int n;
  int* fail_flag;
};

struct DenseLossArgs {
  int num_blocks;          // all residual blocks
  const int* res_off;      // [num_blocks + 1] first residual row of each block
  const int* rb_loss;      // [num_blocks] root node of the block's loss, -1 = trivial
  const LossNode* nodes;
  const int* pidx;         // x offsets of the parameter blocks, flattened
  const int* psize;        // their sizes, same indexing
  const size_t* pidx_off;  // [num_blocks + 1]
  double* r;               // [m] residuals, corrected in place
  double* J;               // [m][n] or null (cost-only evaluation)
  double* cterm;           // [m] cost terms: cost = 1/2 sum
  int n;
};

void launch_dense_eval(int functor_id, bool jac, const DenseEvalArgs& a, hipStream_t s);
void launch_dense_loss(const DenseLossArgs& a, hipStream_t s);
void launch_dense_sum(const double* v, int m, double* out, hipStream_t s);
// recorded functors (tape.hpp); false = the tape's register file does not fit the LDS (nothing was launched)
bool launch_dense_eval_tape(const TapeDevBuffers& tb, bool jac, const DenseEvalArgs& a, hipStream_t s);
bool launch_single_eval_tape(const TapeDevBuffers& tb, const double* consts, const double* x, const int* x_off, double* residuals, double* jac,
                             const int* jac_off, int want_jac, unsigned jac_mask, int* ok, hipStream_t s);
void launch_single_eval(int functor_id, const double* consts, const double* x, const int* x_off, double* residuals, double* jac,
                        const int* jac_off, int want_jac, unsigned jac_mask, int* ok, hipStream_t s);
void launch_dense_col_reduce(const double* J, const double* r, int m, int n, double* colsq, double* gs, hipStream_t s);
struct ParamBlock;
void launch_dense_project(const double* Jg, int m, int ng, const ParamBlock* blocks, int nblocks, const double* x, const double* scale, double* Jl, int nl,
                          hipStream_t s);
void launch_dense_plus(const double* y, const double* scale, const double* x, double* step, double* x_new, const ParamBlock* blocks, int nblocks, double* out,
                       hipStream_t s);
void launch_dense_scale(double* J, const double* scale, int m, int n, hipStream_t s);
void launch_dense_sumsq(const double* r, int m, double* out, hipStream_t s);
void launch_dense_normal(const double* J, const double* r, int m, int n, double* H, int ld, int rhs_row, hipStream_t s);
void launch_dense_step(const double* y, const double* scale, const double* x, double* step, double* x_new, int n, double* out, hipStream_t s);
void launch_dense_model(const double* J, const double* r, const double* step, int m, int n, double* out, hipStream_t s);
void launch_dense_gmax(const double* gs, const double* scale, const double* x, int n, double* out, hipStream_t s);
void launch_dense_qr(const double* J, const double* r, const double* D, int m, int n, double* A, double* b, double* y, int* ok, hipStream_t s);

}  // namespace sk
