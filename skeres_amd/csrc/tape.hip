// Host side of recorded functor bodies (tape.hpp): content key, validation, device copy.
#include <hip/hip_runtime.h>
#include <cstring>
#include "common.hpp"
#include "tape.hpp"

namespace sk {

std::string Tape::key() const {
  std::string k;
  auto put = [&k](const void* p, size_t n) { k.append(reinterpret_cast<const char*>(p), n); };
  const int head[3] = {num_residuals, num_registers, num_obs_consts};
  put(head, sizeof(head));
  const int nb = (int)block_sizes.size();
  put(&nb, sizeof(nb)); put(block_sizes.data(), block_sizes.size() * sizeof(int));
  put(ins.data(), ins.size() * sizeof(TapeIns));
  put(consts.data(), consts.size() * sizeof(double));
  put(out.data(), out.size() * sizeof(int32_t));
  return k;
}

static int operand_count(int op) {
  switch (op) {
    case kTapeAdd: case kTapeSub: case kTapeMul: case kTapeDiv: case kTapeAtan2: case kTapeLt: case kTapeLe: return 2;
    case kTapeSelect: return 3;
    default: return 1;
  }
}

std::string tape_validate(const Tape& t) {
  char buf[160];
  if (t.num_residuals <= 0) { snprintf(buf, sizeof(buf), "Nonpositive number of residuals specified: %d", t.num_residuals); return buf; }
  if (t.num_residuals > kTapeMaxResiduals) { snprintf(buf, sizeof(buf), "a recorded functor has at most %d residuals, %d given", kTapeMaxResiduals, t.num_residuals); return buf; }
  if (t.block_sizes.empty()) return "a cost function needs at least one parameter block";
  for (int b : t.block_sizes) if (b <= 0) return "Nonpositive parameter block sizes specified";
  const int dim = t.dim();
  if (dim > kTapeMaxDim) { snprintf(buf, sizeof(buf), "a recorded functor has at most %d parameters in all, %d given", kTapeMaxDim, dim); return buf; }
  if (t.num_registers < 0 || t.num_registers > (1 << 16)) return "invalid number of registers";
  if (t.num_obs_consts < 0 || t.num_obs_consts > 64) return "a recorded functor captures at most 64 doubles";
  if ((int)t.out.size() != t.num_residuals) return "one output operand per residual is needed";
  std::vector<char> written((size_t)t.num_registers, 0);
  auto operand_ok = [&](int32_t code, const char** why) {
    const int idx = tape_index(code);
    if (code < 0) { *why = "negative operand"; return false; }
    switch (tape_kind(code)) {
      case kTapeReg: if (idx >= t.num_registers) { *why = "register out of range"; return false; } if (!written[idx]) { *why = "register read before it is written"; return false; } return true;
      case kTapeParam: if (idx >= dim) { *why = "parameter index out of range"; return false; } return true;
      case kTapeObs: if (idx >= t.num_obs_consts) { *why = "captured-constant index out of range"; return false; } return true;
      case kTapeConst: if (idx >= (int)t.consts.size()) { *why = "tape-constant index out of range"; return false; } return true;
      default: *why = "unknown operand kind"; return false;
    }
  };
  for (size_t i = 0; i < t.ins.size(); ++i) {
    const TapeIns& in = t.ins[i];
    const char* why = nullptr;
    if (in.op < 0 || in.op >= kTapeNumOps) { snprintf(buf, sizeof(buf), "instruction %zu: unknown opcode %d", i, in.op); return buf; }
    const int32_t ops[3] = {in.a, in.b, in.c};
    for (int k = 0; k < operand_count(in.op); ++k)
      if (!operand_ok(ops[k], &why)) { snprintf(buf, sizeof(buf), "instruction %zu: %s", i, why); return buf; }
    if (in.dst < 0 || in.dst >= t.num_registers) { snprintf(buf, sizeof(buf), "instruction %zu: destination register out of range", i); return buf; }
    written[in.dst] = 1;
  }
  for (int r = 0; r < t.num_residuals; ++r) {
    const char* why = nullptr;
    if (!operand_ok(t.out[r], &why)) { snprintf(buf, sizeof(buf), "residual %d: %s", r, why); return buf; }
  }
  return std::string();
}

hipError_t TapeDevBuffers::upload(const Tape& t, hipStream_t s) {
  hipError_t e;
  std::vector<TapeIns> ins_h = t.ins;
  if (ins_h.empty()) ins_h.push_back(TapeIns{kTapeMov, 0, (kTapeConst << 28), 0, 0});  // (never executed: num_ins stays 0)
  std::vector<double> consts_h = t.consts;
  if (consts_h.empty()) consts_h.push_back(0.0);
  std::vector<int> pb, pi;
  for (size_t q = 0; q < t.block_sizes.size(); ++q)
    for (int j = 0; j < t.block_sizes[q]; ++j) { pb.push_back((int)q); pi.push_back(j); }
  if ((e = ins.upload(ins_h, s)) != hipSuccess) return e;
  if ((e = consts.upload(consts_h, s)) != hipSuccess) return e;
  if ((e = out.upload(t.out, s)) != hipSuccess) return e;
  if ((e = param_block.upload(pb, s)) != hipSuccess) return e;
  if ((e = param_index.upload(pi, s)) != hipSuccess) return e;
  view.ins = ins.p; view.num_ins = (int)t.ins.size();
  view.consts = consts.p;
  view.out = out.p; view.num_residuals = t.num_residuals;
  view.num_registers = std::max(1, t.num_registers); view.dim = t.dim(); view.num_blocks = (int)t.block_sizes.size();
  view.param_block = param_block.p; view.param_index = param_index.p;
  host = t;
  return hipSuccess;
}

// W derivative slots per pass for a register file of `threads` threads: the widest of 3, 2, 1 that leaves room for TWO
// workgroups per CU (the interpreter is bound by LDS latency: Ladybug-1723, 11 registers, 256 threads — W = 3, one
// workgroup per CU, 1.28 ms; W = 2, two, 1.08 ms; W = 1, three, 1.92 ms), else the widest that fits at all;
// 0 = not even one slot fits (too many registers)
int tape_pick_width(const Tape& t, int threads) {
  const int top = 3;
  for (int W = top; W >= 1; --W) if (tape_lds_bytes(t, W, threads) <= kTapeLdsBudget / 2) return W;
  for (int W = top; W >= 1; --W) if (tape_lds_bytes(t, W, threads) <= kTapeLdsBudget) return W;
  return 0;
}

}  // namespace sk
