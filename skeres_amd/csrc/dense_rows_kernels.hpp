// Launchers of dense_rows_kernels.hip (dense-rows path, BASELINE.json config 5).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace sk {

struct DenseRowsArgs {
  int m, n;              // residuals, parameters
  size_t m_pad;          // leading dimension of Jt (multiple of 16)
  const double* consts;  // [m][3]: seed, row index, y
  double inv_sqrt_n;
};

void launch_rows_residual(const DenseRowsArgs& a, const double* x, double* r, double* sd, bool want_sd, hipStream_t s);
void launch_rows_jacobian(const DenseRowsArgs& a, const double* sd, const double* scale, double* Jt, hipStream_t s);
void launch_rows_col_reduce(const double* Jt, const double* r, int m, int n, size_t m_pad, double* colsq, double* gs, hipStream_t s);
void launch_rows_scale(double* Jt, const double* scale, int m, int n, size_t m_pad, hipStream_t s);
int launch_rows_model(const double* Jt, const double* r, const double* step, int m, int n, size_t m_pad, double* partial, hipStream_t s);
int launch_rows_sumsq(const double* r, int m, double* partial, hipStream_t s);
void launch_rows_set_rhs(double* H, long ld, int rhs_row, const double* gs, int n, hipStream_t s);

}  // namespace sk
