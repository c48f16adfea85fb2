// Trust-region Levenberg-Marquardt driver (host) over device linear algebra.
// Restates the loop native Ceres [ext] runs behind `ceres.solve`
// (EX/SimpleBundleAdjuster.scala:152, EX/CurveFitting.scala:127); constants and
// update rules as published for Ceres 1.x.  Only a
// handful of scalars cross PCIe per iteration.
#pragma once
#include <chrono>
#include <memory>

#include "chol_kernels.hpp"
#include "common.hpp"

namespace sk {

class SolverBase {
 public:
  SolverBase(const Options& o, Problem* p) : opt_(o), problem_(p) {}
  virtual ~SolverBase();
  int create();                 // device setup + iteration 0
  int step(bool* done);         // one trust-region iteration
  int finish(Summary* s);       // parameters back to caller memory + summary
  KernelTimer& kernel_timer() { return kt_; }
  // sk_solver_set_kernel_timing / sk_solver_kernel_seconds (a solver may keep more than one timer: one per enqueueing thread)
  virtual void set_kernel_timing(int on) { kt_.only(on == 2 ? "gemm_syrk" : ""); kt_.enable(on != 0); }
  virtual KernelTimer::Stat kernel_stat(const std::string& name) { return kt_.get_stat(name); }
  virtual double syrk_flops_per_solve() const { return 0.0; }
  virtual double syrk_c_bytes_per_solve() const { return 0.0; }  // C tiles read + written by those launches
  virtual bool stat(const std::string& name, double* value) const { (void)name; (void)value; return false; }  // sk_solver_stat
  // seconds accumulated so far in phase i (the summary's phase_seconds, readable between steps: "phase_seconds_<i>" of sk_solver_stat)
  // (between steps the stream is idle: the all-reduce phase takes in every collective's event pair first)
  double phase_seconds(int i) { if (i == 5) collect_allreduce_time(true); return (i >= 0 && i < 6) ? phase_[i] : 0.0; }
  // how a world > 1 is used (SK_DISTRIBUTION_*), with the estimates behind an automatic choice
  virtual int distribution(double* allreduce_s, double* saved_s) const {
    if (allreduce_s) *allreduce_s = 0.0;
    if (saved_s) *saved_s = 0.0;
    return SK_DISTRIBUTION_REPLICATED;
  }

 protected:
  // --- representation-specific pieces -------------------------------------
  virtual int setup() = 0;                        // build device structures, upload x
  virtual int evaluate_with_jacobian(bool first) = 0;  // at current x: cost_, gmax_, xnorm_
  // D from radius, solve, candidate point, candidate cost:
  virtual int try_step(double radius, bool* valid, double* model_cost_change, double* new_cost, double* step_norm) = 0;
  virtual void accept_candidate() = 0;            // x <- candidate (pointer swap)
  virtual int write_back() = 0;                   // device x -> caller memory
  virtual void describe(Summary* s) = 0;

  int init_device();
  double now() const { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0_).count(); }
  void log_iteration(int it, double cost_change, double step_norm, double rho, int valid, int success, double iter_time);
  int allreduce(double* dev, size_t count);
  void collect_allreduce_time(bool all);  // phase_[5] from the event pairs of collectives that have completed (all: wait for every one)
  std::vector<hipEvent_t> ar_pending_, ar_free_;

  Options opt_;
  Problem* problem_;
  hipStream_t stream_ = nullptr;
  bool own_stream_ = false;
  KernelTimer kt_;
  // phase timing (HIP events on stream_)
  enum { kEvBegin = 0, kEvJac, kEvAssemble, kEvChol, kEvBacksub, kEvCost, kEvCount };
  hipEvent_t ev_[16] = {};
  double phase_[7] = {0, 0, 0, 0, 0, 0, 0};
  // LM state
  double cost_ = 0, gmax_ = 0, xnorm_ = 0;
  double radius_ = 0, decrease_factor_ = 2.0;
  int iteration_ = 0, invalid_ = 0, n_success_ = 0, n_unsuccess_ = 0;
  bool terminated_ = false;
  Summary sum_;
  std::chrono::steady_clock::time_point t0_;
  std::string device_name_;
};

// BAL-shaped problems (2 residuals, one 9-block + one 3-block per residual
// block): Schur elimination of the 3-blocks + dense Cholesky of the reduced system.
std::unique_ptr<SolverBase> make_bal_solver(const Options& o, Problem* p);
bool problem_is_bal_shaped(const Problem& p, std::string* why_not);
void bal_index_problem(const Problem& p, std::vector<int>* cam_block, std::vector<int>* pt_block, std::vector<int>* ocam,
                       std::vector<int>* opt);
void bal_partition_points(const std::vector<int>& opt, int num_points, int world, std::vector<int>* cut);
int bal_segment_plan(const Problem& p, int max_segments, bool forced, std::vector<int>* block_camera_part, std::vector<int>* block_point_owner);
bool bal_block_shape(const Problem& p, int* r, int* c, int* q);
int bal_border_plan(const Problem& p, int mode, std::vector<int>* final_index_of_block, int* gap, double* model_us, double* plain_us, double* fill);
// the retained points as BalSolver::setup chooses them (one process): flag per residual block; returns their number
int bal_retained_plan(const Problem& p, int mode, int max_points, int border_mode, std::vector<int>* retained_of_block, double* model_us, double* model_us_without,
                      bool with_memory_order = true);
// Generic dense Jacobian path: DENSE_QR / DENSE_NORMAL_CHOLESKY.
std::unique_ptr<SolverBase> make_dense_solver(const Options& o, Problem* p);
// Tall dense rows over one parameter block (transposed Jacobian + long-K MFMA SYRK): DENSE_NORMAL_CHOLESKY.
std::unique_ptr<SolverBase> make_dense_rows_solver(const Options& o, Problem* p);
bool problem_is_dense_rows(const Problem& p);

}  // namespace sk
