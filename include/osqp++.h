// osqp++.h -- the subset of google/osqp-cpp's header that ZPP-Robotics/OSQP-Solver compiles against, implemented on the
// C-ABI of the MI355X-native core (mi_osqp.h) instead of the osqp C library.
//
// Put this directory in front of the include path and link libmi_osqp.so: the reference's headers then compile
// UNCHANGED -- [REF] src/utils.h:7,11 (`#include <osqp++.h>`, `using ExitCode = osqp::OsqpExitCode`),
// src/constraints/constraint-builder.h:10, src/osqp-wrapper.h:6,18-28,36,40,46,52-53, examples/solver-example.cpp:71,94
// (`ToString(exit_code)`).  Like upstream's header this one pulls in Eigen (the reference relies on that for
// Eigen::SparseMatrix / VectorXd / Triplet, [REF] src/utils.h:12-20) and a Status type under namespace absl.
//
// What is provided, and which reference line uses it:
//   osqp::OsqpInstance  {objective_matrix, objective_vector, constraint_matrix, lower_bounds, upper_bounds}   [REF] osqp-wrapper.h:18-24
//   osqp::OsqpSettings  (every field of osqp-cpp's struct; the reference sets only `verbose`)                 [REF] osqp-wrapper.h:26-27
//   osqp::OsqpSolver::Init / UpdateConstraintMatrix / SetBounds / SetPrimalWarmStart / Solve / primal_solution [REF] :28,36,40,46,52,53
//   absl::Status::ok / ToString                                                                               [REF] :30,34-48
//   osqp::OsqpExitCode, osqp::ToString                                                                        [REF] utils.h:11; gomp-solver.h:40,46-49,68,72,79
// plus dual_solution / iterations / objective_value / IsInitialized.  Entry points of osqp-cpp that the C-ABI has no
// counterpart for (SetObjectiveVector, UpdateObjectiveMatrix, SetDualWarmStart, polishing) return kUnimplemented.
//
// Settings that the MI355X core does not implement are validated like upstream and otherwise ignored: polish (the
// reference leaves it off), time_limit, delta, adaptive_rho_fraction (the wall-clock rule; the deterministic interval
// 4 * check_termination stands in for "auto").
#ifndef MI_OSQP_OSQPPP_SHIM_H_
#define MI_OSQP_OSQPPP_SHIM_H_

#include <cstdint>
#include <cstdio>
#include <string>
#include <utility>
#include <vector>

#if !__has_include(<Eigen/Sparse>)
#error "osqp++.h (mi_osqp shim): Eigen is required, exactly as for google/osqp-cpp's own header - add Eigen to the include path"
#endif
#include <Eigen/Core>
#include <Eigen/Sparse>

#include "mi_osqp.h"

// ---- absl::Status: the real one when abseil is on the include path, else the few members the reference touches
#if __has_include("absl/status/status.h")
#include "absl/status/status.h"
#define MI_OSQP_SHIM_STATUS(code, msg) ::absl::Status(::absl::StatusCode::code, msg)
#define MI_OSQP_SHIM_OK() ::absl::OkStatus()
#else
namespace absl {
enum class StatusCode : int { kOk = 0, kUnknown = 2, kInvalidArgument = 3, kFailedPrecondition = 9, kUnimplemented = 12 };
class Status {
 public:
  Status() = default;
  Status(StatusCode code, std::string msg) : code_(code), msg_(std::move(msg)) {}
  bool ok() const { return code_ == StatusCode::kOk; }
  StatusCode code() const { return code_; }
  const std::string &message() const { return msg_; }
  std::string ToString() const {
    if (ok()) return "OK";
    const char *n = code_ == StatusCode::kInvalidArgument ? "INVALID_ARGUMENT"
                  : code_ == StatusCode::kFailedPrecondition ? "FAILED_PRECONDITION"
                  : code_ == StatusCode::kUnimplemented ? "UNIMPLEMENTED" : "UNKNOWN";
    return std::string(n) + ": " + msg_;
  }
 private:
  StatusCode code_ = StatusCode::kOk;
  std::string msg_;
};
inline Status OkStatus() { return Status(); }
}  // namespace absl
#define MI_OSQP_SHIM_STATUS(code, msg) ::absl::Status(::absl::StatusCode::code, msg)
#define MI_OSQP_SHIM_OK() ::absl::Status()
#endif

namespace osqp {

using c_int = long long;            // OSQP built with DLONG, which is what Eigen::SparseMatrix<.., long long> of the reference needs

struct OsqpInstance {
  c_int num_variables() const { return static_cast<c_int>(objective_vector.size()); }
  c_int num_constraints() const { return static_cast<c_int>(lower_bounds.size()); }
  Eigen::SparseMatrix<double, Eigen::ColMajor, c_int> objective_matrix;     // upper triangle is used (both may be given)
  Eigen::VectorXd objective_vector;
  Eigen::SparseMatrix<double, Eigen::ColMajor, c_int> constraint_matrix;
  Eigen::VectorXd lower_bounds;
  Eigen::VectorXd upper_bounds;
};

struct OsqpSettings {               // defaults = osqp 0.6.x (osqp_set_default_settings)
  double rho = 0.1;
  double sigma = 1e-6;
  c_int scaling = 10;
  bool adaptive_rho = true;
  c_int adaptive_rho_interval = 0;
  double adaptive_rho_tolerance = 5.0;
  double adaptive_rho_fraction = 0.4;
  c_int max_iter = 4000;
  double eps_abs = 1e-3;
  double eps_rel = 1e-3;
  double eps_prim_inf = 1e-4;
  double eps_dual_inf = 1e-4;
  double alpha = 1.6;
  double delta = 1e-6;
  bool polish = false;
  c_int polish_refine_iter = 3;
  bool verbose = true;
  bool scaled_termination = false;
  c_int check_termination = 25;
  bool warm_start = true;
  double time_limit = 0.0;
};

enum class OsqpExitCode {
  kOptimal, kPrimalInfeasible, kDualInfeasible, kOptimalInaccurate, kPrimalInfeasibleInaccurate,
  kDualInfeasibleInaccurate, kMaxIterations, kInterrupted, kTimeLimitReached, kNonConvex, kUnknown,
};
inline std::string ToString(OsqpExitCode exitcode) { return mi_osqp_exit_code_name(static_cast<int64_t>(exitcode)); }

class OsqpSolver {
 public:
  OsqpSolver() = default;
  OsqpSolver(OsqpSolver &&o) noexcept { *this = std::move(o); }
  OsqpSolver &operator=(OsqpSolver &&o) noexcept {
    if (this != &o) { reset(); h_ = o.h_; n_ = o.n_; m_ = o.m_; x_ = std::move(o.x_); y_ = std::move(o.y_); info_ = o.info_; o.h_ = nullptr; }
    return *this;
  }
  OsqpSolver(const OsqpSolver &) = delete;
  OsqpSolver &operator=(const OsqpSolver &) = delete;
  ~OsqpSolver() { reset(); }

  // osqp_setup: validates, copies the data (the caller's instance may die right after, [REF] osqp-wrapper.h:18-31),
  // equilibrates, orders and factors the KKT matrix, uploads everything to the GPU.
  absl::Status Init(const OsqpInstance &instance, const OsqpSettings &settings) {
    reset();
    const c_int n = instance.num_variables(), m = instance.num_constraints();
    if (n <= 0) return MI_OSQP_SHIM_STATUS(kInvalidArgument, "The number of variables must be positive");
    if (instance.objective_matrix.rows() != n || instance.objective_matrix.cols() != n ||
        instance.constraint_matrix.rows() != m || instance.constraint_matrix.cols() != n || instance.upper_bounds.size() != m)
      return MI_OSQP_SHIM_STATUS(kInvalidArgument, "The dimensions of the objective / constraint data do not agree");
    if (settings.polish) return MI_OSQP_SHIM_STATUS(kUnimplemented, "polish is not available in the MI355X core");
    Eigen::SparseMatrix<double, Eigen::ColMajor, c_int> P = instance.objective_matrix, A = instance.constraint_matrix;
    P.makeCompressed(); A.makeCompressed();
    mi_osqp_settings s;
    mi_osqp_default_settings(&s);
    s.rho = settings.rho; s.sigma = settings.sigma; s.scaling = settings.scaling; s.adaptive_rho = settings.adaptive_rho;
    s.adaptive_rho_interval = settings.adaptive_rho_interval; s.adaptive_rho_tolerance = settings.adaptive_rho_tolerance;
    s.max_iter = settings.max_iter; s.eps_abs = settings.eps_abs; s.eps_rel = settings.eps_rel;
    s.eps_prim_inf = settings.eps_prim_inf; s.eps_dual_inf = settings.eps_dual_inf; s.alpha = settings.alpha;
    s.scaled_termination = settings.scaled_termination; s.check_termination = settings.check_termination;
    s.warm_start = settings.warm_start; s.verbose = settings.verbose;
    static_assert(sizeof(c_int) == sizeof(int64_t), "CSC index width");
    const int rc = mi_osqp_setup(&h_, n, m, reinterpret_cast<const int64_t *>(P.outerIndexPtr()),
                                 reinterpret_cast<const int64_t *>(P.innerIndexPtr()), P.valuePtr(), instance.objective_vector.data(),
                                 reinterpret_cast<const int64_t *>(A.outerIndexPtr()), reinterpret_cast<const int64_t *>(A.innerIndexPtr()),
                                 A.valuePtr(), instance.lower_bounds.data(), instance.upper_bounds.data(), &s);
    if (rc != MI_OSQP_OK) { h_ = nullptr; return from_error(rc, "osqp_setup"); }
    n_ = n; m_ = m;
    x_.assign(static_cast<size_t>(n), 0.0); y_.assign(static_cast<size_t>(m), 0.0);
    return MI_OSQP_SHIM_OK();
  }
  bool IsInitialized() const { return h_ != nullptr; }

  // osqp_solve + the copy-out of the solution; never throws ([REF] osqp-wrapper.h:51-54)
  OsqpExitCode Solve() {
    if (!h_) return OsqpExitCode::kUnknown;
    last_error_ = mi_osqp_solve(h_, &info_);
    if (last_error_ != MI_OSQP_OK) {
      // osqp-cpp turns a failing osqp_solve into kUnknown, which the reference's driver reads as "not converged, go on"
      // ([REF] src/gomp-solver.h:44-51): a device fault must not pass as that silently
      std::fprintf(stderr, "OsqpSolver::Solve: %s (%s)\n", mi_osqp_error_name(last_error_), mi_osqp_last_error());
      return last_error_ == MI_OSQP_ERR_NONCONVEX ? OsqpExitCode::kNonConvex : OsqpExitCode::kUnknown;
    }
    mi_osqp_get_primal(h_, x_.data());
    if (m_) mi_osqp_get_dual(h_, y_.data());
    return static_cast<OsqpExitCode>(info_.exit_code);
  }
  int last_error() const { return last_error_; }        // mi_osqp_error of the last Solve() (not part of osqp-cpp)
  c_int iterations() const { return info_.iter; }
  double objective_value() const { return info_.obj_val; }
  Eigen::Map<const Eigen::VectorXd> primal_solution() const { return Eigen::Map<const Eigen::VectorXd>(x_.data(), n_); }
  Eigen::Map<const Eigen::VectorXd> dual_solution() const { return Eigen::Map<const Eigen::VectorXd>(y_.data(), m_); }

  absl::Status SetPrimalWarmStart(const Eigen::Ref<const Eigen::VectorXd> &primal_vector) {
    if (!h_) return not_initialized();
    if (primal_vector.size() != n_) return MI_OSQP_SHIM_STATUS(kInvalidArgument, "The warm start has the wrong length");
    const Eigen::VectorXd x = primal_vector;                    // (a Ref may be strided)
    return from_error(mi_osqp_warm_start_x(h_, x.data()), "osqp_warm_start_x");
  }
  absl::Status SetBounds(const Eigen::Ref<const Eigen::VectorXd> &lower_bounds, const Eigen::Ref<const Eigen::VectorXd> &upper_bounds) {
    if (!h_) return not_initialized();
    if (lower_bounds.size() != m_ || upper_bounds.size() != m_) return MI_OSQP_SHIM_STATUS(kInvalidArgument, "The bounds have the wrong length");
    const Eigen::VectorXd l = lower_bounds, u = upper_bounds;
    return from_error(mi_osqp_update_bounds(h_, l.data(), u.data()), "osqp_update_bounds");
  }
  absl::Status UpdateConstraintMatrix(const Eigen::SparseMatrix<double, Eigen::ColMajor, c_int> &constraint_matrix) {
    if (!h_) return not_initialized();
    if (constraint_matrix.rows() != m_ || constraint_matrix.cols() != n_)
      return MI_OSQP_SHIM_STATUS(kInvalidArgument, "The constraint matrix has the wrong shape");
    Eigen::SparseMatrix<double, Eigen::ColMajor, c_int> A = constraint_matrix;
    A.makeCompressed();
    return from_error(mi_osqp_update_A(h_, reinterpret_cast<const int64_t *>(A.outerIndexPtr()),
                                       reinterpret_cast<const int64_t *>(A.innerIndexPtr()), A.valuePtr()), "osqp_update_A");
  }
  // entry points of osqp-cpp without a counterpart in the C-ABI (the reference uses none of them)
  absl::Status SetDualWarmStart(const Eigen::Ref<const Eigen::VectorXd> &) { return unimplemented("SetDualWarmStart"); }
  absl::Status SetObjectiveVector(const Eigen::Ref<const Eigen::VectorXd> &) { return unimplemented("SetObjectiveVector"); }
  absl::Status UpdateObjectiveMatrix(const Eigen::SparseMatrix<double, Eigen::ColMajor, c_int> &) { return unimplemented("UpdateObjectiveMatrix"); }

 private:
  void reset() { if (h_) mi_osqp_free(h_); h_ = nullptr; n_ = m_ = 0; }
  static absl::Status not_initialized() { return MI_OSQP_SHIM_STATUS(kFailedPrecondition, "OsqpSolver is not initialized."); }
  static absl::Status unimplemented(const char *what) { return MI_OSQP_SHIM_STATUS(kUnimplemented, std::string(what) + " is not available in the MI355X core"); }
  static absl::Status from_error(int rc, const char *where) {
    if (rc == MI_OSQP_OK) return MI_OSQP_SHIM_OK();
    std::string msg = std::string(where) + ": " + mi_osqp_error_name(rc);
    const char *extra = mi_osqp_last_error();
    if (extra && *extra && (rc == MI_OSQP_ERR_DEVICE || rc == MI_OSQP_ERR_ALLOC)) msg += std::string(" (") + extra + ")";
    if (rc == MI_OSQP_ERR_INVALID_DATA || rc == MI_OSQP_ERR_INVALID_SETTINGS || rc == MI_OSQP_ERR_PATTERN_CHANGED || rc == MI_OSQP_ERR_NONCONVEX)
      return MI_OSQP_SHIM_STATUS(kInvalidArgument, msg);      // what osqp-cpp reports for rejected data / a changed sparsity pattern
    return MI_OSQP_SHIM_STATUS(kUnknown, msg);
  }
  mi_osqp_solver *h_ = nullptr;
  c_int n_ = 0, m_ = 0;
  std::vector<double> x_, y_;
  mi_osqp_info info_{};
  int last_error_ = 0;
};

}  // namespace osqp

#endif  // MI_OSQP_OSQPPP_SHIM_H_
