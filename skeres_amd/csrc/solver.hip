// Host-side trust-region loop shared by the Schur and the dense paths.
#include "solver.hpp"

#include <cmath>
#include <limits>

namespace sk {

static thread_local std::string g_error;
void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_error = buf;
}
const char* get_error() { return g_error.c_str(); }
// the status that goes with a failure reported through a null handle (sk_solver_create): typed, not parsed from the text
static thread_local int g_status = SK_OK;
void set_status(int status) { g_status = status; }
int get_status() { return g_status; }

const char* linear_solver_name(int t) {
  switch (t) {
    case SK_DENSE_NORMAL_CHOLESKY: return "DENSE_NORMAL_CHOLESKY";
    case SK_DENSE_QR: return "DENSE_QR";
    case SK_SPARSE_NORMAL_CHOLESKY: return "SPARSE_NORMAL_CHOLESKY";
    case SK_DENSE_SCHUR: return "DENSE_SCHUR";
    case SK_SPARSE_SCHUR: return "SPARSE_SCHUR";
    case SK_ITERATIVE_SCHUR: return "ITERATIVE_SCHUR";
    case SK_CGNR: return "CGNR";
  }
  return "UNKNOWN";
}
const char* termination_name(int t) {
  switch (t) {
    case SK_CONVERGENCE: return "CONVERGENCE";
    case SK_NO_CONVERGENCE: return "NO_CONVERGENCE";
    case SK_FAILURE: return "FAILURE";
    case SK_USER_SUCCESS: return "USER_SUCCESS";
    case SK_USER_FAILURE: return "USER_FAILURE";
  }
  return "UNKNOWN";
}

SolverBase::~SolverBase() {
  for (auto& e : ev_) if (e) (void)hipEventDestroy(e);
  for (auto& e : ar_pending_) (void)hipEventDestroy(e);
  for (auto& e : ar_free_) (void)hipEventDestroy(e);
  if (own_stream_ && stream_) (void)hipStreamDestroy(stream_);
}

int SolverBase::init_device() {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
    set_error("no HIP device available: libskeres_amd has no CPU fallback");
    return SK_ERR_NO_DEVICE;
  }
  if (opt_.device >= 0) SK_HIP_TRY(hipSetDevice(opt_.device));
  int dev = 0;
  SK_HIP_TRY(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  SK_HIP_TRY(hipGetDeviceProperties(&prop, dev));
  device_name_ = std::string(prop.name) + " (" + prop.gcnArchName + ")";
  if (opt_.stream_set) {
    stream_ = opt_.stream;
  } else {
    SK_HIP_TRY(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    own_stream_ = true;
  }
  for (int i = 0; i < kEvCount; ++i) SK_HIP_TRY(hipEventCreate(&ev_[i]));
  return SK_OK;
}

// The hook runs on the solver's stream and the solver goes on enqueueing behind it: NO host synchronisation here (until round 5 every
// collective ended in one — four per iteration of a segmented world, the device idle while the host caught up with its launches).  The
// time of the all-reduce phase comes from event pairs that are read once they have completed (here, opportunistically, and in finish()).
void SolverBase::collect_allreduce_time(bool all) {
  size_t done = 0;
  for (; done + 1 < ar_pending_.size(); done += 2) {
    if (!all && hipEventQuery(ar_pending_[done + 1]) != hipSuccess) { (void)hipGetLastError(); break; }
    if (all) (void)hipEventSynchronize(ar_pending_[done + 1]);
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ar_pending_[done], ar_pending_[done + 1]) == hipSuccess) phase_[5] += 1e-3 * ms;
    else (void)hipGetLastError();
    ar_free_.push_back(ar_pending_[done]); ar_free_.push_back(ar_pending_[done + 1]);
  }
  ar_pending_.erase(ar_pending_.begin(), ar_pending_.begin() + (long)done);
}

int SolverBase::allreduce(double* dev, size_t count) {
  if (!opt_.allreduce) return SK_OK;  // the hook decides: a world of 1 with a hook still exercises the whole path
  collect_allreduce_time(ar_pending_.size() >= 64);
  hipEvent_t ab[2];
  for (hipEvent_t& e : ab) {
    if (!ar_free_.empty()) { e = ar_free_.back(); ar_free_.pop_back(); }
    else SK_HIP_TRY(hipEventCreate(&e));
  }
  SK_HIP_TRY(hipEventRecord(ab[0], stream_));
  if (opt_.allreduce(opt_.allreduce_user, dev, count, (void*)stream_) != 0) {
    ar_free_.push_back(ab[0]); ar_free_.push_back(ab[1]);
    set_error("allreduce hook failed");
    return SK_ERR_COMM;
  }
  SK_HIP_TRY(hipEventRecord(ab[1], stream_));
  ar_pending_.push_back(ab[0]); ar_pending_.push_back(ab[1]);
  return SK_OK;
}

void SolverBase::log_iteration(int it, double cost_change, double step_norm, double rho, int valid, int success, double iter_time) {
  IterationLog L;
  L.iteration = it; L.cost = cost_; L.cost_change = cost_change; L.gradient_max_norm = gmax_;
  L.step_norm = step_norm; L.relative_decrease = rho; L.trust_region_radius = radius_;
  L.step_is_valid = valid; L.step_is_successful = success; L.iter_time = iter_time; L.total_time = now();
  sum_.iterations.push_back(L);
  if (opt_.progress_to_stdout && opt_.rank == 0) {
    if (it == 0) printf("iter      cost      cost_change  |gradient|   |step|    tr_ratio  tr_radius  ls_iter  iter_time  total_time\n");
    printf("%4d % 8e   % 3.2e   % 3.2e  % 3.2e  % 3.2e % 3.2e     % 4d   % 3.2e   % 3.2e\n", it, L.cost, L.cost_change,
           L.gradient_max_norm, L.step_norm, L.relative_decrease, L.trust_region_radius, 1, L.iter_time, L.total_time);
    fflush(stdout);
  }
}

int SolverBase::create() {
  t0_ = std::chrono::steady_clock::now();
  if (opt_.minimizer_type != SK_TRUST_REGION) { set_error("only TRUST_REGION (Levenberg-Marquardt) is implemented"); return SK_ERR_UNSUPPORTED; }
  int rc = init_device();
  if (rc) return rc;
  rc = setup();
  if (rc) return rc;
  rc = evaluate_with_jacobian(true);
  if (rc == SK_ERR_EVALUATION_FAILED) {
    sum_.termination_type = SK_FAILURE;
    sum_.message = "Initial residual and Jacobian evaluation failed.";
    terminated_ = true;
    return SK_OK;
  }
  if (rc) return rc;
  sum_.initial_cost = cost_;
  radius_ = opt_.initial_trust_region_radius;
  decrease_factor_ = 2.0;
  iteration_ = 0;
  log_iteration(0, 0.0, 0.0, 0.0, 1, 1, now());
  sum_.termination_type = SK_NO_CONVERGENCE;
  return SK_OK;
}

int SolverBase::step(bool* done) {
  char msg[256];
  *done = false;
  if (terminated_) { *done = true; return SK_OK; }
  // FinalizeIterationAndCheckIfMinimizerCanContinue
  if (iteration_ >= opt_.max_num_iterations) {
    sum_.termination_type = SK_NO_CONVERGENCE;
    snprintf(msg, sizeof(msg), "Maximum number of iterations reached. Number of iterations: %d.", iteration_);
    sum_.message = msg; terminated_ = true; *done = true; return SK_OK;
  }
  if (gmax_ <= opt_.gradient_tolerance) {
    sum_.termination_type = SK_CONVERGENCE;
    snprintf(msg, sizeof(msg), "Gradient tolerance reached. Gradient max norm: %e <= %e", gmax_, opt_.gradient_tolerance);
    sum_.message = msg; terminated_ = true; *done = true; return SK_OK;
  }
  if (radius_ < opt_.min_trust_region_radius) {
    sum_.termination_type = SK_CONVERGENCE;
    snprintf(msg, sizeof(msg), "Minimum trust region radius reached. Trust region radius: %e <= %e", radius_, opt_.min_trust_region_radius);
    sum_.message = msg; terminated_ = true; *done = true; return SK_OK;
  }
  const double t_iter = now();
  ++iteration_;
  bool valid = false;
  double mcc = 0.0, new_cost = 0.0, step_norm = 0.0;
  int rc = try_step(radius_, &valid, &mcc, &new_cost, &step_norm);
  if (rc) return rc;
  if (valid && !(mcc > 0.0)) valid = false;
  if (!valid) {
    ++invalid_;
    if (invalid_ >= opt_.max_num_consecutive_invalid_steps) {
      sum_.termination_type = SK_FAILURE;
      snprintf(msg, sizeof(msg), "Number of consecutive invalid steps more than Solver::Options::max_num_consecutive_invalid_steps: %d",
               opt_.max_num_consecutive_invalid_steps);
      sum_.message = msg; terminated_ = true; *done = true;
      log_iteration(iteration_, 0.0, 0.0, 0.0, 0, 0, now() - t_iter);
      return SK_OK;
    }
    radius_ /= decrease_factor_; decrease_factor_ *= 2.0;  // StepIsInvalid == StepRejected
    log_iteration(iteration_, 0.0, 0.0, 0.0, 0, 0, now() - t_iter);
    return SK_OK;
  }
  invalid_ = 0;
  if (!std::isfinite(new_cost)) new_cost = std::numeric_limits<double>::max();
  const double cost_change = cost_ - new_cost;
  if (step_norm <= opt_.parameter_tolerance * (xnorm_ + opt_.parameter_tolerance)) {
    sum_.termination_type = SK_CONVERGENCE;
    snprintf(msg, sizeof(msg), "Parameter tolerance reached. Relative step_norm: %e <= %e.", step_norm / (xnorm_ + opt_.parameter_tolerance), opt_.parameter_tolerance);
    sum_.message = msg; terminated_ = true; *done = true;
    log_iteration(iteration_, cost_change, step_norm, 0.0, 1, 0, now() - t_iter);
    return SK_OK;
  }
  if (std::fabs(cost_change) <= opt_.function_tolerance * cost_) {
    sum_.termination_type = SK_CONVERGENCE;
    snprintf(msg, sizeof(msg), "Function tolerance reached. |cost_change|/cost: %e <= %e", std::fabs(cost_change) / cost_, opt_.function_tolerance);
    sum_.message = msg; terminated_ = true; *done = true;
    log_iteration(iteration_, cost_change, step_norm, 0.0, 1, 0, now() - t_iter);
    return SK_OK;
  }
  const double rho = cost_change / mcc;
  if (rho > opt_.min_relative_decrease) {
    accept_candidate();
    rc = evaluate_with_jacobian(false);
    if (rc == SK_ERR_EVALUATION_FAILED) {
      sum_.termination_type = SK_FAILURE; sum_.message = "Residual and Jacobian evaluation failed.";
      terminated_ = true; *done = true; return SK_OK;
    }
    if (rc) return rc;
    radius_ = radius_ / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * rho - 1.0, 3));
    radius_ = std::min(opt_.max_trust_region_radius, radius_);
    decrease_factor_ = 2.0;
    ++n_success_;
    log_iteration(iteration_, cost_change, step_norm, rho, 1, 1, now() - t_iter);
  } else {
    radius_ /= decrease_factor_; decrease_factor_ *= 2.0;
    ++n_unsuccess_;
    log_iteration(iteration_, cost_change, step_norm, rho, 1, 0, now() - t_iter);
  }
  return SK_OK;
}

int SolverBase::finish(Summary* s) {
  int rc = write_back();
  if (rc) return rc;
  collect_allreduce_time(true);
  sum_.final_cost = cost_;
  sum_.num_successful_steps = n_success_;
  sum_.num_unsuccessful_steps = n_unsuccess_;
  for (int i = 0; i < 6; ++i) sum_.phase_seconds[i] = phase_[i];
  sum_.phase_seconds[6] = now();
  sum_.device_name = device_name_;
  sum_.world = opt_.world;
  sum_.linear_solver_type = opt_.linear_solver_type;
  sum_.linear_solver_type_given = opt_.linear_solver_type_given;
  describe(&sum_);
  sum_.build_reports();
  *s = sum_;
  return SK_OK;
}

void Summary::build_reports() {
  char b[4096];
  const int iters = (int)iterations.size();
  snprintf(b, sizeof(b), "Ceres Solver Report: Iterations: %d, Initial cost: %e, Final cost: %e, Termination: %s", iters,
           initial_cost, final_cost, termination_name(termination_type));
  brief = b;
  std::string f;
  snprintf(b, sizeof(b),
           "\nSolver Summary (skeres_amd, MI355X-native Levenberg-Marquardt)\n\n"
           "Parameter blocks            % 12d\nParameters                  % 12d\nResidual blocks             % 12d\nResiduals                   % 12ld\n\n"
           "Minimizer                        TRUST_REGION\nTrust region strategy     LEVENBERG_MARQUARDT\n\n"
           "Linear solver          %22s\nDevice                 %s\nGPUs                        % 12d\n",
           num_parameter_blocks, num_parameters, num_residual_blocks, num_residuals, linear_solver_name(linear_solver_type),
           device_name.c_str(), world);
  f += b;
  if (linear_solver_type_given >= 0 && linear_solver_type_given != linear_solver_type) {
    // (Ceres reports "Given / Used"; its alternate for a Schur-type solver with nothing to eliminate is DENSE_QR too)
    snprintf(b, sizeof(b), "Linear solver given    %22s   (no 2-residual / 9- and 3-parameter block structure to eliminate: its alternate is used)\n",
             linear_solver_name(linear_solver_type_given));
    f += b;
  }
  if (linear_solver_type == SK_DENSE_SCHUR) {
    snprintf(b, sizeof(b), "Schur structure                        2,3,9\nE blocks (eliminated)       % 12d\nF blocks                    % 12d\n", num_e_blocks, num_f_blocks);
    f += b;
  }
  snprintf(b, sizeof(b),
           "\nCost:\nInitial                   % 14e\nFinal                     % 14e\nChange                    % 14e\n\n"
           "Minimizer iterations        % 12d\nSuccessful steps            % 12d\nUnsuccessful steps          % 12d\n\n"
           "Device time (s):\n  Jacobian evaluation       % 12.6f\n  Linear solver assembly    % 12.6f\n  Linear solver factor      % 12.6f\n"
           "  Back-substitution         % 12.6f\n  Cost evaluation           % 12.6f\n  All-reduce                % 12.6f\nTotal wall time             % 12.6f\n\n"
           "Termination:          %22s (%s)\n",
           initial_cost, final_cost, initial_cost - final_cost, iters, num_successful_steps, num_unsuccessful_steps, phase_seconds[0],
           phase_seconds[1], phase_seconds[2], phase_seconds[3], phase_seconds[4], phase_seconds[5], phase_seconds[6],
           termination_name(termination_type), message.c_str());
  f += b;
  full = f;
}

}  // namespace sk
