// sk_rotation_apply: the Rotation functions (rotation.hpp; CORE/Rotation.scala:63-522) evaluated ON THE
// DEVICE for batches of inputs, with T = double or T = Jet<K> (K <= 4), so that the reference's own
// RotationSpec can be run against the code device functors call.
#include <hip/hip_runtime.h>

#include "common.hpp"
#include "functors.hpp"
#include "rotation.hpp"

namespace sk {

__host__ __device__ inline int rotation_in_len(int op) {
  const int t[14] = {3, 4, 9, 9, 3, 3, 4, 4, 7, 7, 8, 6, 6, 6};
  return t[op];
}
__host__ __device__ inline int rotation_out_len(int op) {
  const int t[14] = {4, 3, 4, 3, 9, 9, 9, 9, 3, 3, 4, 3, 1, 3};
  return t[op];
}

template <class T>
__device__ bool rotation_apply_one(int op, int row_major, const T* in, T* out) {
  const int rs = row_major ? 3 : 1, cs = row_major ? 1 : 3;
  switch (op) {
    case SK_ROT_ANGLE_AXIS_TO_QUATERNION: angle_axis_to_quaternion(in, out); return true;
    case SK_ROT_QUATERNION_TO_ANGLE_AXIS: quaternion_to_angle_axis(in, out); return true;
    case SK_ROT_ROTATION_MATRIX_TO_QUATERNION: rotation_matrix_to_quaternion(in, rs, cs, out); return true;
    case SK_ROT_ROTATION_MATRIX_TO_ANGLE_AXIS: rotation_matrix_to_angle_axis(in, rs, cs, out); return true;
    case SK_ROT_ANGLE_AXIS_TO_ROTATION_MATRIX: angle_axis_to_rotation_matrix(in, out, rs, cs); return true;
    case SK_ROT_EULER_ANGLES_TO_ROTATION_MATRIX: euler_angles_to_rotation_matrix(in, out, rs, cs); return true;
    case SK_ROT_QUATERNION_TO_SCALED_ROTATION: quaternion_to_scaled_rotation(in, out, rs, cs); return true;
    case SK_ROT_QUATERNION_TO_ROTATION: return quaternion_to_rotation(in, out, rs, cs);
    case SK_ROT_UNIT_QUATERNION_ROTATE_POINT: unit_quaternion_rotate_point(in, in + 4, out); return true;
    case SK_ROT_QUATERNION_ROTATE_POINT: quaternion_rotate_point(in, in + 4, out); return true;
    case SK_ROT_QUATERNION_PRODUCT: quaternion_product(in, in + 4, out); return true;
    case SK_ROT_CROSS_PRODUCT: cross_product(in, in + 3, out); return true;
    case SK_ROT_DOT_PRODUCT: out[0] = dot_product(in, in + 3); return true;
    case SK_ROT_ANGLE_AXIS_ROTATE_POINT: angle_axis_rotate_point(in, in + 3, out); return true;
  }
  return false;
}

__global__ void rotation_kernel_double(int op, int row_major, const double* in, int n, double* out, int* fail) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int ni = rotation_in_len(op), no = rotation_out_len(op);
  double x[9], y[9];
  for (int e = 0; e < ni; ++e) x[e] = in[(size_t)i * ni + e];
  if (!rotation_apply_one<double>(op, row_major, x, y)) { *fail = 1; return; }
  for (int e = 0; e < no; ++e) out[(size_t)i * no + e] = y[e];
}

template <int K>
__global__ void rotation_kernel_jet(int op, int row_major, const double* in, int n, double* out, int* fail) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int ni = rotation_in_len(op), no = rotation_out_len(op);
  Jet<K> x[9], y[9];
  for (int e = 0; e < ni; ++e) {
    const double* p = in + ((size_t)i * ni + e) * (1 + K);
    x[e].a = p[0];
    for (int k = 0; k < K; ++k) x[e].v[k] = p[1 + k];
  }
  if (!rotation_apply_one<Jet<K>>(op, row_major, x, y)) { *fail = 1; return; }
  for (int e = 0; e < no; ++e) {
    double* p = out + ((size_t)i * no + e) * (1 + K);
    p[0] = y[e].a;
    for (int k = 0; k < K; ++k) p[1 + k] = y[e].v[k];
  }
}

int rotation_apply(int op, int row_major, int jet_dim, const double* in, int n, double* out) {
  if (op < 0 || op > 13) { set_error("unknown rotation op %d", op); return SK_ERR_INVALID_ARGUMENT; }
  if (jet_dim < 0 || jet_dim > 4) { set_error("jet dimension %d not supported (0..4)", jet_dim); return SK_ERR_INVALID_ARGUMENT; }
  if (n < 0 || (n > 0 && (!in || !out))) { set_error("sk_rotation_apply: bad arguments"); return SK_ERR_INVALID_ARGUMENT; }
  if (n == 0) return SK_OK;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { (void)hipGetLastError(); set_error("no HIP device available: libskeres_amd has no CPU fallback"); return SK_ERR_NO_DEVICE; }
  const size_t w = 1 + (size_t)jet_dim;
  const size_t n_in = (size_t)n * rotation_in_len(op) * w, n_out = (size_t)n * rotation_out_len(op) * w;
  DevBuf<double> din, dout; DevBuf<int> dfail;
  SK_HIP_TRY(din.upload(std::vector<double>(in, in + n_in), nullptr));
  SK_HIP_TRY(dout.alloc(n_out)); SK_HIP_TRY(dfail.alloc(1)); SK_HIP_TRY(dfail.zero(nullptr));
  const dim3 g((n + 127) / 128), b(128);
  switch (jet_dim) {
    case 0: hipLaunchKernelGGL(rotation_kernel_double, g, b, 0, nullptr, op, row_major, din.p, n, dout.p, dfail.p); break;
    case 1: hipLaunchKernelGGL(rotation_kernel_jet<1>, g, b, 0, nullptr, op, row_major, din.p, n, dout.p, dfail.p); break;
    case 2: hipLaunchKernelGGL(rotation_kernel_jet<2>, g, b, 0, nullptr, op, row_major, din.p, n, dout.p, dfail.p); break;
    case 3: hipLaunchKernelGGL(rotation_kernel_jet<3>, g, b, 0, nullptr, op, row_major, din.p, n, dout.p, dfail.p); break;
    case 4: hipLaunchKernelGGL(rotation_kernel_jet<4>, g, b, 0, nullptr, op, row_major, din.p, n, dout.p, dfail.p); break;
  }
  SK_HIP_TRY(hipGetLastError());
  int fail = 0;
  SK_HIP_TRY(hipMemcpy(&fail, dfail.p, sizeof(int), hipMemcpyDeviceToHost));
  SK_HIP_TRY(hipMemcpy(out, dout.p, n_out * sizeof(double), hipMemcpyDeviceToHost));
  if (fail) { set_error("rotation function failed (quaternionToRotation of the zero quaternion)"); return SK_ERR_EVALUATION_FAILED; }
  return SK_OK;
}

}  // namespace sk
