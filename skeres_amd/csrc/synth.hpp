// Counter-based generator of the synthetic dense problem of BASELINE.json config 5
// ("10k params x 1M residuals": r_i(x) = tanh(a_i . x) - y_i, a_ij ~ N(0, 1/n)).  The 80 GB
// coefficient matrix is never stored: a_ij is a pure function of (seed, i, j).
//   h   = splitmix64(seed ^ (i * n + j) * golden)
//   a_ij = (u0 + u1 + u2 + u3 - 2) * sqrt(3 / n),  u_k = (16-bit field k of h + 0.5) / 65536
// (Irwin-Hall(4): mean 0, variance 1/n; approximately normal.)
#pragma once
#include <stdint.h>
#include "jet.hpp"

namespace sk {

SK_HD uint64_t synth_mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// unit-variance draw for element (i, j); multiply by sqrt(1/n) for a_ij
SK_HD double synth_unit(uint64_t seed, uint64_t i, uint64_t n, uint64_t j) {
  const uint64_t h = synth_mix64(seed ^ ((i * n + j) * 0xD6E8FEB86659FD93ull));
  const double s = (double)((h & 0xffff) + ((h >> 16) & 0xffff) + ((h >> 32) & 0xffff) + ((h >> 48) & 0xffff)) + 2.0;  // 4 * 0.5
  return (s * (1.0 / 65536.0) - 2.0) * 1.7320508075688772;
}

}  // namespace sk
