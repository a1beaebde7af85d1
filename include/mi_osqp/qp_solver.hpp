// qp_solver.hpp -- header-only C++17 facade with the method set of the
// reference's class QPSolver ([REF] /root/reference/src/osqp-wrapper.h:12-60),
// implemented on the C-ABI of mi_osqp.h instead of google/osqp-cpp.
//
//   QPSolver(const QPConstraints&, const QPMatrixSparse& P)   [REF] :16-31
//   void update(const QPConstraints&)                          [REF] :33-43  (throws std::invalid_argument)
//   void setWarmStart(const QPVector&)                         [REF] :45-49
//   std::pair<OsqpExitCode, QPVector> solve()                  [REF] :51-54
//
// The reference's types are Eigen types ([REF] src/utils.h:12,15;
// src/constraints/constraint-builder.h:16).  Eigen is not part of this
// repository, so the same data contract is expressed with plain containers:
// CSC, column-major, `long long` indices, double values, +-1e30 = unbounded.
// Where <Eigen/Sparse> is available, `from_eigen()` adapts the reference's own
// objects without copying their layout assumptions.
#pragma once

#include <algorithm>
#include <atomic>
#include <cassert>
#include <iostream>
#include <stdexcept>
#include <string>
#include <thread>
#include <tuple>
#include <utility>
#include <vector>

#include "../mi_osqp.h"

namespace miosqp_ref {

constexpr double INF = 1e30;                       // [REF] src/constraints/constraints.h:11

struct QPMatrixSparse {                            // = Eigen::SparseMatrix<double, ColMajor, long long>
  long long rows = 0, cols = 0;
  std::vector<long long> outer;                    // cols+1 column pointers
  std::vector<long long> inner;                    // row indices
  std::vector<double> values;
};
using QPVector = std::vector<double>;              // = Eigen::VectorXd
using QPConstraints = std::tuple<QPVector, QPMatrixSparse, QPVector>;   // <l, A, u>, [REF] constraint-builder.h:16

enum class OsqpExitCode {                          // osqp-cpp's enum, consumed at [REF] src/gomp-solver.h:40,46-49,68,72,79
  kOptimal, kPrimalInfeasible, kDualInfeasible, kOptimalInaccurate, kPrimalInfeasibleInaccurate,
  kDualInfeasibleInaccurate, kMaxIterations, kInterrupted, kTimeLimitReached, kNonConvex, kUnknown
};
using ExitCode = OsqpExitCode;                     // [REF] src/utils.h:11
inline std::string ToString(OsqpExitCode c) { return mi_osqp_exit_code_name(static_cast<int64_t>(c)); }

#if __has_include(<Eigen/Sparse>)
}  // namespace miosqp_ref
#include <Eigen/Sparse>
namespace miosqp_ref {
template <class EigenSparse>
inline QPMatrixSparse from_eigen(const EigenSparse &M_) {
  Eigen::SparseMatrix<double, Eigen::ColMajor, long long> M = M_;
  M.makeCompressed();
  QPMatrixSparse r;
  r.rows = M.rows(); r.cols = M.cols();
  r.outer.assign(M.outerIndexPtr(), M.outerIndexPtr() + M.cols() + 1);
  r.inner.assign(M.innerIndexPtr(), M.innerIndexPtr() + M.nonZeros());
  r.values.assign(M.valuePtr(), M.valuePtr() + M.nonZeros());
  return r;
}
#endif

class QPSolver {
 public:
  // `verbose` mirrors settings.verbose = true of the reference (log lines only).
  QPSolver(const QPConstraints &c, const QPMatrixSparse &P, bool verbose = true,
           const mi_osqp_settings *custom = nullptr) {
    const auto &[l, A, u] = c;
    if (verbose)
      std::cout << l.size() << ", " << u.size() << ", " << A.cols << ", " << A.rows << ", " << P.rows << ", "
                << P.cols << std::endl;                                     // [REF] :19
    mi_osqp_settings s;
    mi_osqp_default_settings(&s);
    if (custom) s = *custom;
    s.verbose = verbose;
    int rc = MI_OSQP_ERR_INVALID_DATA;
    if ((long long)l.size() == A.rows && (long long)u.size() == A.rows && P.rows == A.cols && P.cols == A.cols)
      rc = mi_osqp_setup(&h_, A.cols, A.rows, reinterpret_cast<const int64_t *>(P.outer.data()),
                         reinterpret_cast<const int64_t *>(P.inner.data()), P.values.data(), nullptr /* q = 0, [REF] :22 */,
                         reinterpret_cast<const int64_t *>(A.outer.data()), reinterpret_cast<const int64_t *>(A.inner.data()),
                         A.values.data(), l.data(), u.data(), &s);
    status_ = rc;
    verbose_ = verbose;
    if (rc != MI_OSQP_OK) std::cerr << "QPSolver: setup failed: " << mi_osqp_error_name(rc) << " (" << mi_osqp_last_error() << ")" << std::endl;
    assert(rc == MI_OSQP_OK);                                              // [REF] :30 (assert only)
    n_ = A.cols;
  }
  ~QPSolver() { mi_osqp_free(h_); }
  // Not in the reference: the pattern analysis of a solver that will be constructed later for (c, P), computed now (blocking;
  // meant for a spare host thread).  Values are not looked at, only the patterns.
  static void prefetch(const QPConstraints &c, const QPMatrixSparse &P) {
    const QPMatrixSparse &A = std::get<1>(c);
    (void)mi_osqp_prefetch_analysis(1, A.cols, A.rows, reinterpret_cast<const int64_t *>(P.outer.data()), reinterpret_cast<const int64_t *>(P.inner.data()),
                                    reinterpret_cast<const int64_t *>(A.outer.data()), reinterpret_cast<const int64_t *>(A.inner.data()), -1);
  }
  QPSolver(const QPSolver &) = delete;
  QPSolver &operator=(const QPSolver &) = delete;

  void update(const QPConstraints &qp_constraints) {
    const auto &[low, A, upp] = qp_constraints;
    if ((long long)low.size() != A.rows || (long long)upp.size() != A.rows) throw std::invalid_argument(mi_osqp_error_name(MI_OSQP_ERR_INVALID_DATA));
    // UpdateConstraintMatrix + SetBounds ([REF] :36-42) as one call: one numeric refactorisation instead of two
    int rc = mi_osqp_update_A_bounds(h_, reinterpret_cast<const int64_t *>(A.outer.data()),
                                     reinterpret_cast<const int64_t *>(A.inner.data()), A.values.data(), low.data(), upp.data());
    if (rc != MI_OSQP_OK) throw std::invalid_argument(mi_osqp_error_name(rc));      // [REF] :36-38, :40-42
  }

  void setWarmStart(const QPVector &primal_vector) {
    int rc = (long long)primal_vector.size() == n_ ? mi_osqp_warm_start_x(h_, primal_vector.data())
                                                   : (int)MI_OSQP_ERR_INVALID_DATA;
    if (verbose_) std::cout << "STATUS: " << (rc == MI_OSQP_OK ? "OK" : mi_osqp_error_name(rc)) << std::endl;   // [REF] :47
    assert(rc == MI_OSQP_OK);
  }

  std::pair<OsqpExitCode, QPVector> solve() {
    mi_osqp_info info{};
    QPVector x(n_);
    int rc = mi_osqp_solve(h_, &info);
    if (rc != MI_OSQP_OK) {
      std::cerr << "QPSolver: solve failed: " << mi_osqp_error_name(rc) << " (" << mi_osqp_last_error() << ")" << std::endl;
      return {OsqpExitCode::kUnknown, x};
    }
    mi_osqp_get_primal(h_, x.data());
    last_ = info;
    return {static_cast<OsqpExitCode>(info.exit_code), x};
  }

  int setup_status() const { return status_; }
  const mi_osqp_info &last_info() const { return last_; }

 private:
  mi_osqp_solver *h_ = nullptr;
  long long n_ = 0;
  int status_ = 0;
  bool verbose_ = true;
  mi_osqp_info last_{};
};

// Batched twin of QPSolver on the mi_osqp_batch_* entry points: K QPs that share ONE sparsity pattern (the
// GOMP situation, [REF] src/constraints/constraint-builder.h:112-116) advance in lock-step on one GPU.  Method
// set and error behaviour follow QPSolver; everything is per-QP vectors in the order of construction.
class BatchQPSolver {
 public:
  BatchQPSolver(const std::vector<QPConstraints> &cs, const QPMatrixSparse &P, bool verbose = false,
                const mi_osqp_settings *custom = nullptr)
      : K_((long long)cs.size()) {
    assert(!cs.empty());
    const QPMatrixSparse &A0 = std::get<1>(cs[0]);
    n_ = A0.cols; m_ = A0.rows;
    mi_osqp_settings s;
    mi_osqp_default_settings(&s);
    if (custom) s = *custom;
    s.verbose = verbose;
    std::vector<double> Pv, Av, l, u;
    Pv.reserve(P.values.size() * cs.size()); Av.reserve(A0.values.size() * cs.size());
    l.reserve((size_t)m_ * cs.size()); u.reserve((size_t)m_ * cs.size());
    bool ok = P.rows == n_ && P.cols == n_;
    for (const QPConstraints &c : cs) {
      const auto &[lo, A, up] = c;
      ok = ok && A.rows == m_ && A.cols == n_ && A.outer == A0.outer && A.inner == A0.inner &&
           (long long)lo.size() == m_ && (long long)up.size() == m_;
      if (!ok) break;
      Pv.insert(Pv.end(), P.values.begin(), P.values.end());
      Av.insert(Av.end(), A.values.begin(), A.values.end());
      l.insert(l.end(), lo.begin(), lo.end());
      u.insert(u.end(), up.begin(), up.end());
    }
    if (ok) { P_ = P; A_outer_ = A0.outer; A_inner_ = A0.inner; Av_ = Av; }          // what reinit() compares against
    int rc = MI_OSQP_ERR_INVALID_DATA;
    if (ok)
      rc = mi_osqp_batch_setup(&h_, K_, n_, m_, reinterpret_cast<const int64_t *>(P.outer.data()),
                               reinterpret_cast<const int64_t *>(P.inner.data()), Pv.data(), nullptr,
                               reinterpret_cast<const int64_t *>(A0.outer.data()),
                               reinterpret_cast<const int64_t *>(A0.inner.data()), Av.data(), l.data(), u.data(), &s, -1);
    status_ = rc;
    assert(rc == MI_OSQP_OK);
  }
  ~BatchQPSolver() { mi_osqp_batch_free(h_); }
  BatchQPSolver(const BatchQPSolver &) = delete;
  BatchQPSolver &operator=(const BatchQPSolver &) = delete;

  void update(const std::vector<QPConstraints> &cs) {
    pristine_ = false;                       // (new A values / a snapshot taken with adapted rho: reinit() no longer applies)
    if ((long long)cs.size() != K_) throw std::invalid_argument(mi_osqp_error_name(MI_OSQP_ERR_INVALID_DATA));
    const QPMatrixSparse &A0 = std::get<1>(cs[0]);
    std::vector<double> Av, l, u;
    for (const QPConstraints &c : cs) {
      const auto &[lo, A, up] = c;
      if (A.outer != A0.outer || A.inner != A0.inner) throw std::invalid_argument(mi_osqp_error_name(MI_OSQP_ERR_PATTERN_CHANGED));
      Av.insert(Av.end(), A.values.begin(), A.values.end());
      l.insert(l.end(), lo.begin(), lo.end());
      u.insert(u.end(), up.begin(), up.end());
    }
    int rc = mi_osqp_batch_update_A_bounds(h_, reinterpret_cast<const int64_t *>(A0.outer.data()),
                                           reinterpret_cast<const int64_t *>(A0.inner.data()), Av.data(), l.data(), u.data());
    if (rc != MI_OSQP_OK) throw std::invalid_argument(mi_osqp_error_name(rc));
  }
  // bounds only (the joint-space GOMP rows never change A)
  void updateBounds(const std::vector<QPConstraints> &cs) {
    pristine_ = false;
    std::vector<double> l, u;
    for (const QPConstraints &c : cs) {
      l.insert(l.end(), std::get<0>(c).begin(), std::get<0>(c).end());
      u.insert(u.end(), std::get<2>(c).begin(), std::get<2>(c).end());
    }
    int rc = (long long)cs.size() == K_ ? mi_osqp_batch_update_bounds(h_, l.data(), u.data()) : (int)MI_OSQP_ERR_INVALID_DATA;
    if (rc != MI_OSQP_OK) throw std::invalid_argument(mi_osqp_error_name(rc));
  }

  void setWarmStart(const std::vector<QPVector> &xs) {
    std::vector<double> &x = xbuf_;               // (kept across calls: no fresh pages to fault in on every call)
    x.clear();
    for (const QPVector &v : xs) x.insert(x.end(), v.begin(), v.end());
    int rc = (long long)x.size() == K_ * n_ ? mi_osqp_batch_warm_start_x(h_, x.data()) : (int)MI_OSQP_ERR_INVALID_DATA;
    assert(rc == MI_OSQP_OK);
    (void)rc;
  }

  std::vector<std::pair<OsqpExitCode, QPVector>> solve() {
    std::vector<std::pair<OsqpExitCode, QPVector>> out((size_t)K_, {OsqpExitCode::kUnknown, QPVector((size_t)n_)});
    if (mi_osqp_batch_solve(h_) != MI_OSQP_OK) return out;
    std::vector<double> x((size_t)(K_ * n_));
    infos_.resize((size_t)K_);
    mi_osqp_batch_get_primal(h_, x.data());
    mi_osqp_batch_get_info(h_, infos_.data());
    for (long long k = 0; k < K_; k++) {
      out[(size_t)k].first = static_cast<OsqpExitCode>(infos_[(size_t)k].exit_code);
      std::copy(x.begin() + k * n_, x.begin() + (k + 1) * n_, out[(size_t)k].second.begin());
    }
    return out;
  }

  // A planner builds the same solver again on every run ([REF] src/gomp-solver.h:61: one QPSolver per horizon segment).
  // When P, the pattern of A and the values of A are the ones this solver was built with, the handle is simply put back
  // into its state after construction (mi_osqp_batch_reset) and given the new bounds: bitwise the results of a fresh
  // construction, without analysis, equilibration, upload and factorisation.  false: not applicable, build a new one.
  bool reinit(const std::vector<QPConstraints> &cs, const QPMatrixSparse &P) {
    if (status_ != MI_OSQP_OK || !pristine_ || (long long)cs.size() != K_ || P.outer != P_.outer || P.inner != P_.inner || P.values != P_.values) return false;
    const size_t nnzA = A_inner_.size(), K = cs.size(), m = (size_t)m_;
    lbuf_.resize(m * K); ubuf_.resize(m * K);     // (kept across calls)
    // the K comparisons (pattern and values of A: ~20 MB for 256 trajectories) and copies are independent: a few host threads
    std::atomic<bool> same{true};
    auto body = [&](size_t k) {
      const auto &[lo, A, up] = cs[k];
      if (A.rows != m_ || A.cols != n_ || A.outer != A_outer_ || A.inner != A_inner_ || lo.size() != m || up.size() != m ||
          A.values.size() != nnzA || !std::equal(A.values.begin(), A.values.end(), Av_.begin() + k * nnzA)) { same = false; return; }
      std::copy(lo.begin(), lo.end(), lbuf_.begin() + k * m);
      std::copy(up.begin(), up.end(), ubuf_.begin() + k * m);
    };
    const size_t nt = std::min<size_t>({K, 16, std::max(1u, std::thread::hardware_concurrency())});
    if (nt <= 1) { for (size_t k = 0; k < K; k++) body(k); }
    else {
      std::vector<std::thread> th;
      for (size_t t = 0; t < nt; t++) th.emplace_back([&, t] { for (size_t k = t; k < K; k += nt) body(k); });
      for (auto &x : th) x.join();
    }
    if (!same) return false;
    if (mi_osqp_batch_reset(h_) != MI_OSQP_OK) return false;
    return mi_osqp_batch_update_bounds(h_, lbuf_.data(), ubuf_.data()) == MI_OSQP_OK;
  }

  int setup_status() const { return status_; }
  const std::vector<mi_osqp_info> &last_infos() const { return infos_; }
  long long size() const { return K_; }

 private:
  mi_osqp_batch *h_ = nullptr;
  long long K_ = 0, n_ = 0, m_ = 0;
  int status_ = 0;
  std::vector<mi_osqp_info> infos_;
  QPMatrixSparse P_;
  std::vector<long long> A_outer_, A_inner_;
  std::vector<double> Av_;
  std::vector<double> lbuf_, ubuf_, xbuf_;
  bool pristine_ = true;            // no update() since construction: the handle's setup snapshot is the post-construction state
};

// Continuous twin of BatchQPSolver on the per-QP entry points of mi_osqp.h ("continuous batching"): K slots that share ONE
// sparsity pattern, each slot one QPSolver of the reference in its own phase of life - constructed ([REF]
// src/osqp-wrapper.h:16-31 -> reinit), warm-started (:45-49), solving (:51-54), updated (:33-43) - while the others keep
// iterating.  begin() + advance() + poll() replace the blocking solve(); a finished slot's result() is what solve()
// would have returned for it, bit for bit.  Error behaviour as QPSolver: invalid updates throw std::invalid_argument.
class ContinuousQPSolver {
 public:
  // every slot starts as a copy of (c0, P); P and q = 0 stay, A values and bounds are per slot (reinit / update)
  ContinuousQPSolver(long long slots, const QPConstraints &c0, const QPMatrixSparse &P, bool verbose = false,
                     const mi_osqp_settings *custom = nullptr)
      : K_(slots) {
    assert(slots > 0);
    const auto &[lo, A, up] = c0;
    n_ = A.cols; m_ = A.rows; nnzA_ = (long long)A.values.size();
    mi_osqp_settings s;
    mi_osqp_default_settings(&s);
    if (custom) s = *custom;
    s.verbose = verbose;
    std::vector<double> Pv, Av, l, u;
    for (long long k = 0; k < K_; k++) {
      Pv.insert(Pv.end(), P.values.begin(), P.values.end());
      Av.insert(Av.end(), A.values.begin(), A.values.end());
      l.insert(l.end(), lo.begin(), lo.end());
      u.insert(u.end(), up.begin(), up.end());
    }
    int rc = MI_OSQP_ERR_INVALID_DATA;
    if (P.rows == n_ && P.cols == n_ && (long long)lo.size() == m_ && (long long)up.size() == m_) {
      P_ = P; A_outer_ = A.outer; A_inner_ = A.inner;
      rc = mi_osqp_batch_setup(&h_, K_, n_, m_, reinterpret_cast<const int64_t *>(P.outer.data()),
                               reinterpret_cast<const int64_t *>(P.inner.data()), Pv.data(), nullptr,
                               reinterpret_cast<const int64_t *>(A.outer.data()), reinterpret_cast<const int64_t *>(A.inner.data()),
                               Av.data(), l.data(), u.data(), &s, -1);
    }
    status_ = rc;
    if (rc != MI_OSQP_OK) std::cerr << "ContinuousQPSolver: setup failed: " << mi_osqp_error_name(rc) << " (" << mi_osqp_last_error() << ")" << std::endl;
    assert(rc == MI_OSQP_OK);
  }
  ~ContinuousQPSolver() { mi_osqp_batch_free(h_); }
  ContinuousQPSolver(const ContinuousQPSolver &) = delete;
  ContinuousQPSolver &operator=(const ContinuousQPSolver &) = delete;

  // same objective, same pattern of A: a solver built for one run of a planner serves the next one
  bool compatible(long long slots, const QPConstraints &c0, const QPMatrixSparse &P) const {
    const QPMatrixSparse &A = std::get<1>(c0);
    return status_ == MI_OSQP_OK && slots == K_ && A.rows == m_ && A.cols == n_ && A.outer == A_outer_ && A.inner == A_inner_ &&
           P.outer == P_.outer && P.inner == P_.inner && P.values == P_.values;
  }

  // QPSolver's constructor for the listed slots (new A values and bounds; P as built)
  void reinit(const std::vector<long long> &ids, const std::vector<const QPConstraints *> &cs) { newData(ids, cs, true); }
  // QPSolver::update for the listed slots
  void update(const std::vector<long long> &ids, const std::vector<const QPConstraints *> &cs) { newData(ids, cs, false); }
  void setWarmStart(const std::vector<long long> &ids, const std::vector<const QPVector *> &xs) {
    if (ids.empty()) return;
    xbuf_.clear();
    for (const QPVector *x : xs) { assert((long long)x->size() == n_); xbuf_.insert(xbuf_.end(), x->begin(), x->end()); }
    const int rc = mi_osqp_batch_warm_start_x_some(h_, (int64_t)ids.size(), reinterpret_cast<const int64_t *>(ids.data()), xbuf_.data());
    if (rc != MI_OSQP_OK) std::cerr << "ContinuousQPSolver: warm start failed: " << mi_osqp_error_name(rc) << " (" << mi_osqp_last_error() << ")" << std::endl;
    assert(rc == MI_OSQP_OK);
  }
  // QPSolver::solve, first half: the listed slots start iterating with the next advance()
  void begin(const std::vector<long long> &ids) {
    if (ids.empty()) return;
    const int rc = mi_osqp_batch_solve_begin_some(h_, (int64_t)ids.size(), reinterpret_cast<const int64_t *>(ids.data()));
    if (rc != MI_OSQP_OK) std::cerr << "ContinuousQPSolver: begin failed: " << mi_osqp_error_name(rc) << " (" << mi_osqp_last_error() << ")" << std::endl;
    assert(rc == MI_OSQP_OK);
  }
  // enqueue one segment (25 iterations + checks with default settings) for every slot that is iterating; does not wait
  bool advance(int segments = 1) {
    const int rc = mi_osqp_batch_advance(h_, segments);
    if (rc != MI_OSQP_OK) std::cerr << "ContinuousQPSolver: advance failed: " << mi_osqp_error_name(rc) << " (" << mi_osqp_last_error() << ")" << std::endl;
    return rc == MI_OSQP_OK;
  }
  // slots that finished in the oldest advance not polled yet (waits for it)
  std::vector<long long> poll() {
    std::vector<long long> out((size_t)K_);
    int64_t nf = 0;
    const int rc = mi_osqp_batch_poll(h_, 1, &nf, reinterpret_cast<int64_t *>(out.data()), K_);
    if (rc != MI_OSQP_OK) { std::cerr << "ContinuousQPSolver: poll failed: " << mi_osqp_error_name(rc) << " (" << mi_osqp_last_error() << ")" << std::endl; nf = 0; }
    out.resize((size_t)std::max<int64_t>(nf, 0));
    return out;
  }
  // QPSolver::solve, second half: exit code and primal solution of a finished slot
  std::pair<OsqpExitCode, QPVector> result(long long id) {
    const int64_t i = id;
    mi_osqp_info info{};
    QPVector x((size_t)n_);
    if (mi_osqp_batch_get_info_some(h_, 1, &i, &info) != MI_OSQP_OK || mi_osqp_batch_get_primal_some(h_, 1, &i, x.data()) != MI_OSQP_OK)
      return {OsqpExitCode::kUnknown, x};
    last_ = info;
    return {static_cast<OsqpExitCode>(info.exit_code), x};
  }
  long long running() const { return mi_osqp_batch_running(h_); }
  long long size() const { return K_; }
  int setup_status() const { return status_; }
  const mi_osqp_info &last_info() const { return last_; }
  mi_osqp_batch *handle() const { return h_; }          // for the device-side companions of the solver (mi_gomp_scene)

 private:
  void newData(const std::vector<long long> &ids, const std::vector<const QPConstraints *> &cs, bool fresh) {
    if (ids.empty()) return;
    if (ids.size() != cs.size()) throw std::invalid_argument(mi_osqp_error_name(MI_OSQP_ERR_INVALID_DATA));
    abuf_.clear(); lbuf_.clear(); ubuf_.clear();
    for (const QPConstraints *c : cs) {
      const auto &[lo, A, up] = *c;
      if (A.outer != A_outer_ || A.inner != A_inner_) throw std::invalid_argument(mi_osqp_error_name(MI_OSQP_ERR_PATTERN_CHANGED));
      if ((long long)lo.size() != m_ || (long long)up.size() != m_) throw std::invalid_argument(mi_osqp_error_name(MI_OSQP_ERR_INVALID_DATA));
      abuf_.insert(abuf_.end(), A.values.begin(), A.values.end());
      lbuf_.insert(lbuf_.end(), lo.begin(), lo.end());
      ubuf_.insert(ubuf_.end(), up.begin(), up.end());
    }
    const int64_t *pid = reinterpret_cast<const int64_t *>(ids.data());
    const int rc = fresh ? mi_osqp_batch_reinit_some(h_, (int64_t)ids.size(), pid, abuf_.data(), lbuf_.data(), ubuf_.data())
                         : mi_osqp_batch_update_A_bounds_some(h_, (int64_t)ids.size(), pid, abuf_.data(), lbuf_.data(), ubuf_.data());
    if (rc != MI_OSQP_OK) throw std::invalid_argument(std::string(mi_osqp_error_name(rc)) + " (" + mi_osqp_last_error() + ")");
  }
  mi_osqp_batch *h_ = nullptr;
  long long K_ = 0, n_ = 0, m_ = 0, nnzA_ = 0;
  int status_ = 0;
  QPMatrixSparse P_;
  std::vector<long long> A_outer_, A_inner_;
  std::vector<double> abuf_, lbuf_, ubuf_, xbuf_;
  mi_osqp_info last_{};
};

}  // namespace miosqp_ref
