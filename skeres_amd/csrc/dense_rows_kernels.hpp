// Launchers of dense_rows_kernels.hip (dense-rows path, BASELINE.json config 5).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace sk {

struct DenseRowsArgs {
  int m = 0, n = 0;              // residuals, parameters
  size_t m_pad = 0;              // leading dimension of Jt (multiple of 16)
  const double* consts = nullptr;  // [m][3]: seed, row index, y
  double inv_sqrt_n = 0.0;
  // robust loss shared by every row (loss.hpp); loss_root < 0 = trivial.  With a loss the residual kernel stores the CORRECTED
  // residual and derivative factor (ceres Corrector: for a scalar residual r~ = residual_scaling r, J~ = sqrt(rho') (1 - alpha) J)
  // and the row's cost term rho(r^2) in cterm; without one cterm == nullptr and the cost is 1/2 sum r^2.
  const struct LossNode* loss_nodes = nullptr;
  int loss_root = -1;
  double* cterm = nullptr;
};

void launch_rows_residual(const DenseRowsArgs& a, const double* x, double* r, double* sd, bool want_sd, hipStream_t s);
void launch_rows_jacobian(const DenseRowsArgs& a, const double* sd, const double* scale, double* Jt, hipStream_t s);
void launch_rows_col_reduce(const double* Jt, const double* r, int m, int n, size_t m_pad, double* colsq, double* gs, hipStream_t s);
void launch_rows_scale(double* Jt, const double* scale, int m, int n, size_t m_pad, hipStream_t s);
int launch_rows_model(const double* Jt, const double* r, const double* step, int m, int n, size_t m_pad, double* partial, hipStream_t s);
int launch_rows_sumsq(const double* r, const double* cterm, int m, double* partial, hipStream_t s);
void launch_rows_set_rhs(double* H, long ld, int rhs_row, const double* gs, int n, hipStream_t s);

}  // namespace sk
