// gomp.hpp -- the callers of the solver path, SURVEY.md section 8(f) ranks 1-3, as a
// header-only C++17 layer on top of qp_solver.hpp (no Eigen):
//
//   constraints helpers      [REF] /root/reference/src/constraints/constraints.h:9-69
//   HorizontalLine           [REF] src/horizontal-line.h:6-104
//   RobotBall, linspace,
//   triDiagonalMatrix        [REF] src/utils.h:33-42,50-64,72-96
//   ConstraintBuilder<N>     [REF] src/constraints/constraint-builder.h:18-283
//   GOMPSolver<N>            [REF] src/gomp-solver.h:13-201
//
// Same names, argument meaning, row layout and control flow as the reference (the known-answer
// tests of [REF] tests/test.cpp are replayed against it in tests/cpp/), written against plain
// containers.  GOMPSolver takes the QP backend as a template parameter: `QPSolver` (the MI355X
// path) by default; the parity test instantiates it with an oracle-backed twin.
// The reference prints every waypoint; here printing is off unless `verbose` is set.
#pragma once

#include <algorithm>
#include <array>
#include <memory>
#include <cassert>
#include <cmath>
#include <cstdio>
#include <functional>
#include <map>
#include <optional>
#include <tuple>
#include <utility>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <thread>
#include <type_traits>
#include <vector>

#include "qp_solver.hpp"

namespace miosqp_ref {

// ------------------------------------------------------------------ constraints.h
template <size_t N> using Vec = std::array<double, N>;
template <size_t N> using Bound = std::optional<Vec<N>>;
template <size_t N> using Constraint = std::pair<Bound<N>, Bound<N>>;
template <size_t N> using Ctrl = Vec<N>;
using Point = Vec<3>;
enum Axis : size_t { X, Y, Z };
constexpr std::array<Axis, 3> XYZ_AXES = {X, Y, Z};
constexpr double CENTIMETER = 0.01;
constexpr double ERROR = 1e-3;                         // [REF] src/utils.h:31

namespace constraints {
template <size_t N> Vec<N> of(double v) { Vec<N> r; r.fill(v); return r; }
template <size_t N> Constraint<N> inRange(Bound<N> low, Bound<N> upp) { return {low, upp}; }
template <size_t N> Constraint<N> equal(const Vec<N> &v) { return {v, v}; }
template <size_t N> Constraint<N> greaterEq(const Vec<N> &v) { return {v, std::nullopt}; }
template <size_t N> Constraint<N> lessEq(const Vec<N> &v) { return {std::nullopt, v}; }
template <size_t N> Constraint<N> any() { return {std::nullopt, std::nullopt}; }
template <size_t N> Constraint<N> eqZero() { return equal<N>(of<N>(0.0)); }
template <size_t N> Bound<N> scaled(const Bound<N> &b, double f) {
  if (!b) return std::nullopt;
  Vec<N> r = *b;
  for (double &x : r) x *= f;
  return r;
}
template <size_t N> Constraint<N> scaled(const Constraint<N> &c, double f) { return {scaled<N>(c.first, f), scaled<N>(c.second, f)}; }
}  // namespace constraints

// ------------------------------------------------------------------------ utils.h
using ForwardKinematicsFun = std::function<std::tuple<double, double, double>(double *)>;
using JacobianFun = std::function<void(double *, double *)>;          // out: 3 x N row-major, in: q
using InverseKinematics = std::function<int(double *, double, double, double)>;

struct RobotBall {
  RobotBall(ForwardKinematicsFun fk_, JacobianFun jac_, double radius_, bool is_gripper_ = false)
      : fk(std::move(fk_)), jacobian(std::move(jac_)), radius(radius_), is_gripper(is_gripper_) {}
  ForwardKinematicsFun fk;
  JacobianFun jacobian;
  double radius;
  bool is_gripper;
  // Not in the reference: when fk / jacobian are one of the kinematic models the library also has on the device
  // (mi_gomp_model in mi_osqp.h), naming it here lets the continuous planner re-linearise on the GPU (withBuiltin()).
  int builtin_model = 0;
  std::array<double, 12> builtin_param{};
  RobotBall &withBuiltin(int model, std::initializer_list<double> param = {}) {
    builtin_model = model;
    size_t k = 0;
    for (double v : param) if (k < builtin_param.size()) builtin_param[k++] = v;
    return *this;
  }
};

// n x n matrix with a on the diagonal and b on the +-diagonal_num diagonals for rows >= offset
// (both triangles are emitted, as the reference does)
inline QPMatrixSparse triDiagonalMatrix(double a, double b, int n, int offset = 0, int diagonal_num = 1) {
  std::map<std::pair<long long, long long>, double> cells;          // (col, row) -> value : CSC order
  for (int i = offset; i < n; ++i) {
    cells[{i, i}] = a;
    if (i + diagonal_num < n) cells[{i + diagonal_num, i}] = b;
    if (i - diagonal_num >= offset) cells[{i - diagonal_num, i}] = b;
  }
  QPMatrixSparse M;
  M.rows = M.cols = n;
  M.outer.assign(n + 1, 0);
  for (const auto &[cr, v] : cells) { M.outer[cr.first + 1]++; M.inner.push_back(cr.second); M.values.push_back(v); }
  for (int j = 0; j < n; ++j) M.outer[j + 1] += M.outer[j];
  return M;
}

template <size_t N>
QPVector linspace(const Vec<N> &a, const Vec<N> &b, size_t n_steps) {
  QPVector res(N * n_steps);
  for (size_t s = 0; s < n_steps; ++s)
    for (size_t j = 0; j < N; ++j) res[s * N + j] = (b[j] - a[j]) / double(n_steps - 1) * double(s) + a[j];
  return res;
}

// positions of a joint trajectory (first half of the vector) -> xyz, fk called once per waypoint in order
template <size_t N>
QPVector mapJointTrajectoryToXYZ(const QPVector &trajectory, const ForwardKinematicsFun &mapper) {
  const size_t waypoints = trajectory.size() / 2 / N;
  QPVector xyz(3 * waypoints);
  for (size_t w = 0; w < waypoints; ++w) {
    Vec<N> q;
    for (size_t j = 0; j < N; ++j) q[j] = trajectory[w * N + j];
    auto [x, y, z] = mapper(q.data());
    xyz[3 * w] = x; xyz[3 * w + 1] = y; xyz[3 * w + 2] = z;
  }
  return xyz;
}

// -------------------------------------------------------------- horizontal-line.h
class HorizontalLine {
 public:
  HorizontalLine(const std::array<double, 2> &direction, const Point &point, bool bypass_from_below = false)
      : A_(point), below_(bypass_from_below) {
    const double nrm = std::hypot(direction[0], direction[1]);
    D_ = {direction[0] / nrm, direction[1] / nrm, 0.0};
  }
  // perpendicular from P to the line: X - P with X = A + ((P-A).D) D
  Point getDistanceVec(const Point &P) const {
    double t = 0.0;
    for (int k = 0; k < 3; ++k) t += (P[k] - A_[k]) * D_[k];
    return {A_[0] + t * D_[0] - P[0], A_[1] + t * D_[1] - P[1], A_[2] + t * D_[2] - P[2]};
  }
  std::array<double, 2> getDistanceVecXY(const Point &P) const { Point d = getDistanceVec(P); return {d[0], d[1]}; }
  double getDistanceXY(const Point &P) const { auto d = getDistanceVecXY(P); return std::hypot(d[0], d[1]); }
  Point operator[](const Point &P) const { Point d = getDistanceVec(P); return {P[0] + d[0], P[1] + d[1], P[2] + d[2]}; }
  bool areOnOppositeSides(const Point &P, const Point &Q) const {
    auto a = getDistanceVecXY(P), b = getDistanceVecXY(Q);
    return a[0] * b[0] + a[1] * b[1] < 0;
  }
  // (for the device twin of this class, mi_gomp_line)
  std::array<double, 2> directionXY() const { return {D_[0], D_[1]}; }
  const Point &point() const { return A_; }
  bool isClose(const Point &P, const RobotBall &b) const { return getDistanceXY(P) < b.radius; }
  bool hasCollision(int waypoint, const QPVector &xyz, const RobotBall &b) const {
    const int waypoints = (int)xyz.size() / 3;
    auto at = [&](int w) { return Point{xyz[3 * w], xyz[3 * w + 1], xyz[3 * w + 2]}; };
    const Point p = at(waypoint);
    if (isClose(p, b)) return true;
    if (waypoint > 0 && areOnOppositeSides(at(waypoint - 1), p)) return true;
    if (waypoint + 1 < waypoints && areOnOppositeSides(p, at(waypoint + 1))) return true;
    return false;
  }
  bool isAbove(const Point &P, const RobotBall &b) const {
    return below_ ? (P[Z] - A_[Z]) <= -b.radius + ERROR : (P[Z] - A_[Z]) >= b.radius - ERROR;
  }
  bool bypassFromBelow() const { return below_; }

 private:
  Point D_{}, A_;
  bool below_;
};

// ------------------------------------------------------------- constraint-builder.h
// Rows: (W-1)*D velocity<->position links, then per variable boxes (positions W*D, velocities
// (W-1)*D, accelerations (W-2)*D), then D*W*(3 + |obstacles|*|balls|) rows for the linearised
// end-effector / obstacle constraints (allocated even when unused: bounds +-INF).
template <size_t N_DIM>
class ConstraintBuilder {
  using OptPair = std::pair<std::optional<double>, std::optional<double>>;

 public:
  ConstraintBuilder(size_t waypoints, std::vector<RobotBall> m, std::vector<HorizontalLine> obstacles)
      : W_(waypoints), balls_(std::move(m)), lines_(std::move(obstacles)) {
    cells_.reserve(N_DIM * W_ * (10 + 3 * balls_.size() * (1 + lines_.size())));
    lo_.reserve(N_DIM * W_ * (7 + lines_.size() * balls_.size())); up_.reserve(lo_.capacity());
    for (size_t t = 0; t + 1 < W_; ++t)                      // v_t - q_{t+1} + q_t = 0
      for (size_t j = 0; j < N_DIM; ++j) {
        lo_.push_back(-INF); up_.push_back(INF);
        put(lo_.size() - 1, {{nthVelocity(t) + j, 1.0}, {nthPos(t + 1) + j, -1.0}, {nthPos(t) + j, 1.0}}, {0.0, 0.0});
      }
    user_off_ = lo_.size();
    const size_t extra = N_DIM * (W_ + W_ - 1 + W_ - 2 + W_ * (3 + lines_.size() * balls_.size()));
    lo_.resize(user_off_ + extra, -INF);
    up_.resize(user_off_ + extra, INF);
  }

  ConstraintBuilder &position(size_t i, const Constraint<N_DIM> &c) { return positions(i, i, c); }
  ConstraintBuilder &positions(size_t first, size_t last, const Constraint<N_DIM> &c) { return boxes(nthPos(first), nthPos(last), c); }
  ConstraintBuilder &velocity(size_t i, const Constraint<N_DIM> &c) { return velocities(i, i, c); }
  ConstraintBuilder &velocities(size_t first, size_t last, const Constraint<N_DIM> &c) {
    assert(first <= last && last < W_ - 1);
    return boxes(nthVelocity(first), nthVelocity(last), c);
  }
  ConstraintBuilder &accelerations(size_t first, size_t last, const Constraint<N_DIM> &c) {
    for (size_t i = first; i <= last; ++i) acceleration(i, c);
    return *this;
  }
  ConstraintBuilder &acceleration(size_t i, const Constraint<N_DIM> &c) {
    assert(i + 2 < W_);
    for (size_t j = 0; j < N_DIM; ++j)                          // l <= v_{t+1} - v_t <= u
      put(user_off_ + nthAcceleration(i) + j, {{nthVelocity(i + 1) + j, 1.0}, {nthVelocity(i) + j, -1.0}}, dim(j, c));
    return *this;
  }

  // (re-)linearise the 3-D rows around `trajectory`; the pattern never changes (dummy rows)
  ConstraintBuilder &withObstacles(const Constraint<3> &con_3d, const QPVector &trajectory) {
    size_t row = user_off_ + N_DIM * (W_ + W_ - 1 + W_ - 2);
    for (const RobotBall &ball : balls_) {
      const QPVector xyz = mapJointTrajectoryToXYZ<N_DIM>(trajectory, ball.fk);
      for (size_t w = 0; w < W_; ++w) {
        Vec<N_DIM> q;
        for (size_t j = 0; j < N_DIM; ++j) q[j] = trajectory[w * N_DIM + j];
        const Point p{xyz[3 * w], xyz[3 * w + 1], xyz[3 * w + 2]};
        std::array<double, 3 * N_DIM> J{};
        ball.jacobian(J.data(), q.data());
        auto Jq = [&](size_t axis) { double s = 0.0; for (size_t j = 0; j < N_DIM; ++j) s += J[axis * N_DIM + j] * q[j]; return s; };
        if (ball.is_gripper) {
          for (Axis axis : XYZ_AXES) {
            double lo = -INF, up = INF;
            if (con_3d.first) lo = (*con_3d.first)[axis] - p[axis] + Jq(axis);
            if (con_3d.second) up = (*con_3d.second)[axis] - p[axis] + Jq(axis);
            axisRow(row++, ball, axis, J, w, lo, up);
          }
        }
        for (const HorizontalLine &line : lines_) {
          if (line.hasCollision((int)w, xyz, ball)) {
            const double bound = line[p][Z] - p[Z] + Jq(Z);
            if (line.bypassFromBelow()) axisRow(row++, ball, Z, J, w, -INF, bound);
            else axisRow(row++, ball, Z, J, w, bound, INF);
          } else {
            axisRow(row++, ball, Z, J, w, -INF, INF);          // dummy: keeps the pattern constant
          }
        }
      }
    }
    return *this;
  }

  // the cells and rows withObstacles() writes, with zero values and open bounds, without calling the balls' callbacks: the
  // sparsity pattern of every later build() (pattern analysis ahead of time, GOMPSolver::run)
  ConstraintBuilder &withObstaclePattern() {
    size_t row = user_off_ + N_DIM * (W_ + W_ - 1 + W_ - 2);
    const std::array<double, 3 * N_DIM> J{};
    for (const RobotBall &ball : balls_)
      for (size_t w = 0; w < W_; ++w) {
        if (ball.is_gripper) for (Axis axis : XYZ_AXES) axisRow(row++, ball, axis, J, w, -INF, INF);
        for (size_t k = 0; k < lines_.size(); ++k) axisRow(row++, ball, Z, J, w, -INF, INF);
      }
    return *this;
  }

  // sort and de-duplicate the write log now (build() does it on demand): a builder that serves as a template for copies
  ConstraintBuilder &normalised() { normalise(); return *this; }

  QPConstraints build() const {
    QPMatrixSparse A;
    A.rows = (long long)lo_.size(); A.cols = (long long)(2 * N_DIM * W_);
    A.outer.assign(A.cols + 1, 0);
    normalise();
    A.inner.reserve(cells_.size()); A.values.reserve(cells_.size());
    for (const Cell &c : cells_) { A.outer[c.col + 1]++; A.inner.push_back((long long)c.row); A.values.push_back(c.v); }
    for (long long j = 0; j < A.cols; ++j) A.outer[j + 1] += A.outer[j];
    return {lo_, A, up_};
  }

  size_t nthVelocity(size_t i) const { assert(i < W_ - 1); return W_ * N_DIM + i * N_DIM; }
  size_t nthPos(size_t i) const { assert(i < W_); return i * N_DIM; }
  size_t nthAcceleration(size_t i) const { assert(i < W_ - 2); return W_ * N_DIM + (W_ - 1) * N_DIM + i * N_DIM; }

 private:
  size_t W_, user_off_ = 0;
  std::vector<RobotBall> balls_;
  std::vector<HorizontalLine> lines_;
  // (col, row) -> value, last write wins ([REF] constraint-builder.h:129).  A write log that is sorted into CSC
  // order and de-duplicated on demand: the builder runs once per trajectory, segment and re-linearisation,
  // and a std::map made it the bottleneck of the batched driver.
  struct Cell { size_t col, row; double v; };
  mutable std::vector<Cell> cells_;
  mutable bool sorted_ = true;
  void set(size_t col, size_t row, double v) {
    if (sorted_ && !cells_.empty()) {          // a write to an existing cell of a normalised builder (a copied template): in place
      auto it = std::lower_bound(cells_.begin(), cells_.end(), Cell{col, row, 0.0},
                                 [](const Cell &a, const Cell &b) { return a.col != b.col ? a.col < b.col : a.row < b.row; });
      if (it != cells_.end() && it->col == col && it->row == row) { it->v = v; return; }
    }
    cells_.push_back({col, row, v}); sorted_ = false;
  }
  void normalise() const {
    if (sorted_) return;
    // stable order by (col, row): a counting pass over the columns (2 N_DIM W of them, a handful of cells each), then an
    // insertion sort inside every column - the write order among equal (col, row) survives both
    // (scratch kept per thread: the builder runs thousands of times per second in the batched driver, and fresh
    //  buffers of this size come straight from mmap every time)
    const size_t ncol = 2 * N_DIM * W_;
    static thread_local std::vector<size_t> start, at;
    static thread_local std::vector<Cell> by_col;
    start.assign(ncol + 1, 0);
    for (const Cell &c : cells_) { assert(c.col < ncol); start[c.col + 1]++; }
    for (size_t j = 0; j < ncol; ++j) start[j + 1] += start[j];
    if (by_col.size() < cells_.size()) by_col.resize(cells_.size());
    at.assign(start.begin(), start.end() - 1);
    for (const Cell &c : cells_) by_col[at[c.col]++] = c;
    size_t o = 0;
    for (size_t j = 0; j < ncol; ++j) {
      Cell *b = by_col.data() + start[j], *e = by_col.data() + start[j + 1];
      for (Cell *p = b + 1; p < e; ++p) {
        const Cell c = *p;
        Cell *q = p;
        while (q > b && (q - 1)->row > c.row) { *q = *(q - 1); --q; }
        *q = c;
      }
      for (Cell *p = b; p < e; ++p) {
        if (p + 1 < e && (p + 1)->row == p->row) continue;    // a later write wins
        cells_[o++] = *p;
      }
    }
    cells_.resize(o);
    sorted_ = true;
  }
  std::vector<double> lo_, up_;

  static OptPair dim(size_t j, const Constraint<N_DIM> &c) {
    OptPair r;
    if (c.first) r.first = (*c.first)[j];
    if (c.second) r.second = (*c.second)[j];
    return r;
  }
  void put(size_t row, std::initializer_list<std::pair<size_t, double>> eq, OptPair b) {
    for (const auto &[col, coeff] : eq) set(col, row, coeff);
    if (b.first) lo_[row] = *b.first;
    if (b.second) up_[row] = *b.second;
    assert(lo_[row] <= up_[row]);
  }
  ConstraintBuilder &boxes(size_t first_start, size_t last_start, const Constraint<N_DIM> &c) {
    for (size_t s = first_start; s <= last_start; s += N_DIM)
      for (size_t j = 0; j < N_DIM; ++j) put(user_off_ + s + j, {{s + j, 1.0}}, dim(j, c));
    return *this;
  }
  void axisRow(size_t row, const RobotBall &ball, Axis axis, const std::array<double, 3 * N_DIM> &J, size_t waypoint,
               double low, double upp) {
    for (size_t j = 0; j < N_DIM; ++j) set(nthPos(waypoint) + j, row, J[axis * N_DIM + j]);
    lo_[row] = low + ball.radius;
    up_[row] = upp - ball.radius;
    assert(lo_[row] <= up_[row]);
  }
};

// ------------------------------------------------------------------- gomp-solver.h
namespace detail {
template <class S, class = void> struct has_prefetch : std::false_type {};
template <class S>
struct has_prefetch<S, std::void_t<decltype(S::prefetch(std::declval<const QPConstraints &>(), std::declval<const QPMatrixSparse &>()))>> : std::true_type {};
}  // namespace detail
constexpr int MAX_ITERATIONS = 100;
constexpr int SEGMENTS = 10;

template <size_t N_DIM, class SolverT = QPSolver>
class GOMPSolver {
 public:
  GOMPSolver(size_t waypoints, double time_step, const Constraint<N_DIM> &pos_con, const Constraint<N_DIM> &vel_con,
             const Constraint<N_DIM> &acc_con, const Constraint<3> &con_3d, std::vector<HorizontalLine> obstacles,
             std::vector<RobotBall> m, InverseKinematics gripper_ik = nullptr, bool verbose = false)
      : max_waypoints(waypoints), time_step(time_step), pos_con(pos_con),
        vel_con(constraints::scaled<N_DIM>(vel_con, time_step)),
        acc_con(constraints::scaled<N_DIM>(acc_con, time_step * time_step)), con_3d(con_3d),
        obstacles(std::move(obstacles)), mappers(std::move(m)), gripper_ik(std::move(gripper_ik)), verbose(verbose) {
    assert(max_waypoints >= 4);
  }

  // horizon-shrinking outer loop: SEGMENTS QPs chains of W = max_waypoints * i / SEGMENTS waypoints
  std::pair<ExitCode, QPVector> run(Ctrl<N_DIM> start_pos, Ctrl<N_DIM> end_pos) {
    QPVector last_solution = calcWarmStart(start_pos, end_pos);
    ExitCode last_code = ExitCode::kUnknown;
    // Not in the reference: the horizons of this run are known now, and a solver backend that can analyse a sparsity pattern
    // ahead of time (SolverT::prefetch) does so for the later horizons on spare host threads while the first ones are solved.
    PrefetchJoiner prefetching;
    if constexpr (detail::has_prefetch<SolverT>::value) {
      if (prefetch_patterns && max_waypoints * N_DIM >= 600)
        for (int i = SEGMENTS - 1; i >= 1; --i) {
          const size_t waypoints = max_waypoints * i / SEGMENTS;
          if (waypoints < 4) continue;
          prefetching.threads.emplace_back([this, waypoints, start_pos, end_pos] {
            ConstraintBuilder<N_DIM> b = jointSpaceRows(start_pos, end_pos, waypoints);
            SolverT::prefetch(b.withObstaclePattern().build(), triDiagonalMatrix(2, -1, (int)(N_DIM * 2 * waypoints), (int)(waypoints * N_DIM), (int)N_DIM));
          });
        }
    }
    for (int i = SEGMENTS; i >= 1; --i) {
      const size_t waypoints = max_waypoints * i / SEGMENTS;
      QPVector warm_start(waypoints * N_DIM * 2);
      // same slicing as the reference, including its quirk for i < SEGMENTS (the "velocity" half is
      // taken at offset W*D of the previous, longer solution)
      for (size_t k = 0; k < waypoints * N_DIM; ++k) {
        warm_start[k] = last_solution[k];
        warm_start[waypoints * N_DIM + k] = last_solution[waypoints * N_DIM + k];
      }
      auto [exit_code, solution] = run(start_pos, end_pos, waypoints, warm_start);
      ++segments_run;
      if (exit_code != ExitCode::kOptimal && exit_code != ExitCode::kUnknown) break;
      if (exit_code == ExitCode::kOptimal) { last_code = ExitCode::kOptimal; last_solution = solution; }
    }
    for (size_t k = last_solution.size() / 2; k < last_solution.size(); ++k) last_solution[k] /= time_step;
    return {last_code, last_solution};
  }

  // SQP inner loop for one horizon
  std::pair<ExitCode, QPVector> run(Ctrl<N_DIM> start_pos, Ctrl<N_DIM> end_pos, size_t waypoints, const QPVector &warm_start) {
    ConstraintBuilder<N_DIM> builder = initConstraints(start_pos, end_pos, warm_start, waypoints);
    SolverT qp_solver{builder.build(), triDiagonalMatrix(2, -1, (int)(N_DIM * 2 * waypoints), (int)(waypoints * N_DIM), (int)N_DIM), verbose};
    qp_solver.setWarmStart(warm_start);
    QPVector last_solution = warm_start;
    ExitCode last_code = ExitCode::kUnknown;
    int i = 0;
    while (i++ < MAX_ITERATIONS) {
      auto [exit_code, solution] = qp_solver.solve();
      ++qp_solves;
      if (exit_code != ExitCode::kOptimal) { last_solution = solution; break; }      // there are no solutions
      if (isSolutionOK(solution)) { last_solution = solution; last_code = ExitCode::kOptimal; break; }
      qp_solver.update(builder.withObstacles(con_3d, solution).build());
      ++qp_updates;
    }
    return {last_code, last_solution};
  }

  // counters for tests / reporting
  int segments_run = 0, qp_solves = 0, qp_updates = 0;
  bool prefetch_patterns = true;

 private:
  struct PrefetchJoiner { std::vector<std::thread> threads; ~PrefetchJoiner() { for (auto &t : threads) t.join(); } };
  const size_t max_waypoints;
  const double time_step;
  const Constraint<N_DIM> pos_con, vel_con, acc_con;
  const Constraint<3> con_3d;
  const std::vector<HorizontalLine> obstacles;
  const std::vector<RobotBall> mappers;
  const InverseKinematics gripper_ik;
  const bool verbose;

  QPVector calcWarmStart(const Ctrl<N_DIM> &start_pos, const Ctrl<N_DIM> &end_pos) const {
    QPVector w = linspace<N_DIM>(start_pos, end_pos, max_waypoints);      // joint-space line, zero velocities
    w.resize(2 * max_waypoints * N_DIM, 0.0);
    return w;
  }

  ConstraintBuilder<N_DIM> initConstraints(const Ctrl<N_DIM> &start_pos, const Ctrl<N_DIM> &end_pos, const QPVector &warm_start,
                                           size_t waypoints) const {
    ConstraintBuilder<N_DIM> b = jointSpaceRows(start_pos, end_pos, waypoints);
    b.withObstacles(con_3d, warm_start);
    return b;
  }
  ConstraintBuilder<N_DIM> jointSpaceRows(const Ctrl<N_DIM> &start_pos, const Ctrl<N_DIM> &end_pos, size_t waypoints) const {
    assert(waypoints >= 4);
    ConstraintBuilder<N_DIM> b{waypoints, mappers, obstacles};
    b.position(0, constraints::equal<N_DIM>(start_pos))
        .positions(1, waypoints - 2, pos_con)
        .position(waypoints - 3, constraints::equal<N_DIM>(end_pos))
        .velocities(0, waypoints - 4, vel_con)
        .velocity(waypoints - 3, constraints::eqZero<N_DIM>())
        .accelerations(0, waypoints - 4, acc_con)
        .acceleration(waypoints - 3, constraints::eqZero<N_DIM>());
    return b;
  }

  bool isSolutionOK(const QPVector &q_trajectory) const {
    bool res = true;
    for (const RobotBall &ball : mappers) {
      const QPVector xyz = mapJointTrajectoryToXYZ<N_DIM>(q_trajectory, ball.fk);
      const int waypoints = (int)xyz.size() / 3;
      for (int w = 0; w < waypoints; ++w) {
        const Point p{xyz[3 * w], xyz[3 * w + 1], xyz[3 * w + 2]};
        if (verbose) std::printf("(%f, %f, %f)\n", p[X], p[Y], p[Z]);
        if (ball.is_gripper) {
          for (Axis axis : XYZ_AXES) {
            const double lo = con_3d.first ? (*con_3d.first)[axis] : -INF;
            const double up = con_3d.second ? (*con_3d.second)[axis] : INF;
            if (!(lo - ERROR <= p[axis] - ball.radius && p[axis] + ball.radius <= up + ERROR)) res = false;
          }
        }
        for (const HorizontalLine &line : obstacles)
          if (line.hasCollision(w, xyz, ball) && !line.isAbove(p, ball)) res = false;
      }
    }
    return res;
  }
};

// ------------------------------------------------------ batched driver ("many trajectories")
// B independent (start, end) pairs run the same horizon-shrinking / SQP schedule as GOMPSolver::run in
// LOCK-STEP on one BatchQPSolver: all trajectories of a segment share W, hence one sparsity pattern.  Each
// trajectory follows exactly the decisions the sequential driver would take for it (same QPs, same updates,
// same accepted solutions); trajectories that have finished a segment simply keep their constraints while the
// others re-linearise (their extra solves start from the converged iterate and stop at the first check).
template <size_t N_DIM, class BatchSolverT = BatchQPSolver>
class BatchGOMPSolver {
 public:
  BatchGOMPSolver(size_t waypoints, double time_step, const Constraint<N_DIM> &pos_con, const Constraint<N_DIM> &vel_con,
                  const Constraint<N_DIM> &acc_con, const Constraint<3> &con_3d, std::vector<HorizontalLine> obstacles,
                  std::vector<RobotBall> m, bool verbose = false)
      : max_waypoints(waypoints), time_step(time_step), pos_con(pos_con),
        vel_con(constraints::scaled<N_DIM>(vel_con, time_step)),
        acc_con(constraints::scaled<N_DIM>(acc_con, time_step * time_step)), con_3d(con_3d),
        obstacles(std::move(obstacles)), mappers(std::move(m)), verbose(verbose) {
    assert(max_waypoints >= 4);
  }

  std::vector<std::pair<ExitCode, QPVector>> run(const std::vector<Ctrl<N_DIM>> &starts, const std::vector<Ctrl<N_DIM>> &ends) {
    const size_t B = starts.size();
    assert(ends.size() == B && B > 0);
    std::vector<QPVector> last_solution(B);
    std::vector<ExitCode> last_code(B, ExitCode::kUnknown);
    std::vector<char> alive(B, 1);
    segments_run.assign(B, 0); qp_solves.assign(B, 0); qp_updates.assign(B, 0);
    batch_solves = 0;
    seconds_build = seconds_setup = seconds_solve = seconds_update = 0.0;
    using clk_ = std::chrono::steady_clock;
    auto since_ = [](clk_::time_point t) { return std::chrono::duration<double>(clk_::now() - t).count(); };
    for (size_t b = 0; b < B; ++b) {
      last_solution[b] = linspace<N_DIM>(starts[b], ends[b], max_waypoints);      // joint-space line, zero velocities
      last_solution[b].resize(2 * max_waypoints * N_DIM, 0.0);
    }
    for (int i = SEGMENTS; i >= 1; --i) {
      const size_t waypoints = max_waypoints * i / SEGMENTS;
      std::vector<size_t> ids;                                 // batch slot -> trajectory
      for (size_t b = 0; b < B; ++b) if (alive[b]) ids.push_back(b);
      if (ids.empty()) break;
      const size_t K = ids.size();
      std::vector<QPVector> warm(K), seg_solution(K);
      std::vector<ExitCode> seg_code(K, ExitCode::kUnknown);
      std::vector<ConstraintBuilder<N_DIM>> builders;
      std::vector<QPConstraints> cons(K);
      auto tb_ = clk_::now();
      // The joint-space rows of a segment differ between the trajectories only in the bounds of the first and the third-last
      // waypoint: they are laid down once (same calls, same order as initConstraints) and copied; each trajectory then
      // repeats its two position() calls - in the sequential order they are the last writes to those rows as well - and
      // adds its own obstacle rows.  Same QPConstraints as initConstraints(), a third of the time.
      const ConstraintBuilder<N_DIM> tmpl = jointSpaceTemplate(starts[ids[0]], ends[ids[0]], waypoints);
      builders.assign(K, ConstraintBuilder<N_DIM>{4, {}, {}});
      // the K trajectories are independent: their constraints are built by a few host threads
      parallel_for_(K, [&](size_t k) {
        const QPVector &prev = last_solution[ids[k]];
        warm[k].resize(waypoints * N_DIM * 2);
        for (size_t t = 0; t < waypoints * N_DIM; ++t) {        // same slicing as GOMPSolver::run
          warm[k][t] = prev[t];
          warm[k][waypoints * N_DIM + t] = prev[waypoints * N_DIM + t];
        }
        builders[k] = tmpl;
        builders[k].position(0, constraints::equal<N_DIM>(starts[ids[k]]))
            .position(waypoints - 3, constraints::equal<N_DIM>(ends[ids[k]]))
            .withObstacles(con_3d, warm[k]);
        cons[k] = builders[k].build();
        seg_solution[k] = warm[k];
      });
      seconds_build += since_(tb_);
      auto ts_ = clk_::now();
      // One solver per horizon segment, as the reference builds them ([REF] src/gomp-solver.h:61) - but kept across run()
      // calls: when this segment's P and A are those of the solver built for it earlier (joint-space rows: only the bounds
      // depend on the trajectory), the handle is put back into its post-construction state and given the new bounds
      // (BatchQPSolver::reinit: bitwise a fresh construction) instead of being analysed, uploaded and factored again.
      const QPMatrixSparse Pseg = triDiagonalMatrix(2, -1, (int)(N_DIM * 2 * waypoints), (int)(waypoints * N_DIM), (int)N_DIM);
      std::unique_ptr<BatchSolverT> &slot = solver_cache_[(size_t)(SEGMENTS - i)];
      if (reuse_solvers && slot && reinit_(*slot, cons, Pseg)) ++solver_reuses;
      else slot = std::make_unique<BatchSolverT>(cons, Pseg, verbose);
      BatchSolverT &qp = *slot;
      qp.setWarmStart(warm);
      seconds_setup += since_(ts_);
      std::vector<char> running(K, 1);
      size_t n_running = K;
      for (int it = 0; it < MAX_ITERATIONS && n_running; ++it) {
        auto tq_ = clk_::now();
        auto res = qp.solve();
        seconds_solve += since_(tq_);
        ++batch_solves;
        bool need_update = false;
        auto tu_ = clk_::now();
        for (size_t k = 0; k < K; ++k) {
          if (!running[k]) continue;
          ++qp_solves[ids[k]];
          const auto &[exit_code, solution] = res[k];
          if (exit_code != ExitCode::kOptimal) { seg_solution[k] = solution; running[k] = 0; --n_running; continue; }
          if (isSolutionOK(solution)) { seg_solution[k] = solution; seg_code[k] = ExitCode::kOptimal; running[k] = 0; --n_running; continue; }
          if (it + 1 < MAX_ITERATIONS) {
            cons[k] = builders[k].withObstacles(con_3d, solution).build();
            ++qp_updates[ids[k]];
            need_update = true;
          }
        }
        if (need_update && n_running) qp.update(cons);
        seconds_update += since_(tu_);
      }
      for (size_t k = 0; k < K; ++k) {
        const size_t b = ids[k];
        ++segments_run[b];
        if (seg_code[k] != ExitCode::kOptimal && seg_code[k] != ExitCode::kUnknown) { alive[b] = 0; continue; }
        if (seg_code[k] == ExitCode::kOptimal) { last_code[b] = ExitCode::kOptimal; last_solution[b] = seg_solution[k]; }
      }
    }
    std::vector<std::pair<ExitCode, QPVector>> out(B);
    for (size_t b = 0; b < B; ++b) {
      for (size_t t = last_solution[b].size() / 2; t < last_solution[b].size(); ++t) last_solution[b][t] /= time_step;
      out[b] = {last_code[b], last_solution[b]};
    }
    return out;
  }

  // per-trajectory counters (comparable with GOMPSolver's) and the number of batched solves
  std::vector<int> segments_run, qp_solves, qp_updates;
  int batch_solves = 0;
  bool reuse_solvers = true;      // keep the per-segment solvers across run() calls (see run())
  int solver_reuses = 0;          // how often a kept solver was re-initialised instead of a new one being built
  // where the wall time of the last run() went: building constraints, QP setup (analysis + upload + factorisation),
  // batched solves, feasibility checks + re-linearisation + update
  double seconds_build = 0.0, seconds_setup = 0.0, seconds_solve = 0.0, seconds_update = 0.0;

 private:
  const size_t max_waypoints;
  const double time_step;
  const Constraint<N_DIM> pos_con, vel_con, acc_con;
  const Constraint<3> con_3d;
  const std::vector<HorizontalLine> obstacles;
  const std::vector<RobotBall> mappers;
  const bool verbose;

  std::array<std::unique_ptr<BatchSolverT>, SEGMENTS> solver_cache_;
  // (a batch solver type without reinit(), e.g. a test double, is always rebuilt)
  template <class S>
  static auto reinit_(S &s, const std::vector<QPConstraints> &cons, const QPMatrixSparse &P) -> decltype(s.reinit(cons, P)) { return s.reinit(cons, P); }
  static bool reinit_(...) { return false; }

  // body(k) for k in [0, count) on up to 16 host threads (the FK / Jacobian callbacks of the balls must be re-entrant,
  // which the reference's are: pure functions of the joint vector)
  template <class F>
  static void parallel_for_(size_t count, F &&body) {
    const size_t nt = std::min<size_t>({count, 16, std::max(1u, std::thread::hardware_concurrency())});
    if (nt <= 1) { for (size_t k = 0; k < count; ++k) body(k); return; }
    std::vector<std::thread> th;
    for (size_t t = 0; t < nt; ++t) th.emplace_back([&, t] { for (size_t k = t; k < count; k += nt) body(k); });
    for (auto &x : th) x.join();
  }

  // initConstraints of GOMPSolver ([REF] src/gomp-solver.h:98-110) without its last call (the obstacle rows), normalised
  ConstraintBuilder<N_DIM> jointSpaceTemplate(const Ctrl<N_DIM> &start_pos, const Ctrl<N_DIM> &end_pos, size_t waypoints) const {
    ConstraintBuilder<N_DIM> b{waypoints, mappers, obstacles};
    b.position(0, constraints::equal<N_DIM>(start_pos))
        .positions(1, waypoints - 2, pos_con)
        .position(waypoints - 3, constraints::equal<N_DIM>(end_pos))
        .velocities(0, waypoints - 4, vel_con)
        .velocity(waypoints - 3, constraints::eqZero<N_DIM>())
        .accelerations(0, waypoints - 4, acc_con)
        .acceleration(waypoints - 3, constraints::eqZero<N_DIM>())
        .normalised();
    return b;
  }

  bool isSolutionOK(const QPVector &q_trajectory) const {
    bool res = true;
    for (const RobotBall &ball : mappers) {
      const QPVector xyz = mapJointTrajectoryToXYZ<N_DIM>(q_trajectory, ball.fk);
      const int waypoints = (int)xyz.size() / 3;
      for (int w = 0; w < waypoints; ++w) {
        const Point p{xyz[3 * w], xyz[3 * w + 1], xyz[3 * w + 2]};
        if (ball.is_gripper) {
          for (Axis axis : XYZ_AXES) {
            const double lo = con_3d.first ? (*con_3d.first)[axis] : -INF;
            const double up = con_3d.second ? (*con_3d.second)[axis] : INF;
            if (!(lo - ERROR <= p[axis] - ball.radius && p[axis] + ball.radius <= up + ERROR)) res = false;
          }
        }
        for (const HorizontalLine &line : obstacles)
          if (line.hasCollision(w, xyz, ball) && !line.isAbove(p, ball)) res = false;
      }
    }
    return res;
  }
};

// ------------------------------------------------- continuous driver (SURVEY 8(f) rank 1 without the lock-step)
// The reference's loop is per trajectory: solve -> check -> re-linearise -> update -> solve again, one horizon after the
// other ([REF] src/gomp-solver.h:38-91).  BatchGOMPSolver advances B trajectories in lock-step and waits for the slowest
// QP of every round; here every trajectory walks through GOMPSolver::run on its own.  The ten horizons are ten STAGES:
// one ContinuousQPSolver (B slots, slot = trajectory) and one host thread each.  A trajectory entering a stage gets a
// freshly constructed QP in its slot (reinit: what [REF] src/gomp-solver.h:61-65 does), is warm-started and begun; the
// stage thread advances whatever is iterating in its solver, and as soon as ONE QP has finished its trajectory is
// checked, re-linearised and updated ([REF] :70-88) or handed to the next stage - while the other QPs keep iterating
// and the other stages run side by side on their own streams.  Every trajectory takes exactly the decisions of a
// sequential GOMPSolver::run: same QPs, same updates, same accepted solutions (tests/cpp/gomp_parity.cpp `cont`).
template <size_t N_DIM, class ContSolverT = ContinuousQPSolver>
class ContinuousGOMPSolver {
 public:
  ContinuousGOMPSolver(size_t waypoints, double time_step, const Constraint<N_DIM> &pos_con, const Constraint<N_DIM> &vel_con,
                       const Constraint<N_DIM> &acc_con, const Constraint<3> &con_3d, std::vector<HorizontalLine> obstacles,
                       std::vector<RobotBall> m, bool verbose = false)
      : max_waypoints(waypoints), time_step(time_step), pos_con(pos_con),
        vel_con(constraints::scaled<N_DIM>(vel_con, time_step)),
        acc_con(constraints::scaled<N_DIM>(acc_con, time_step * time_step)), con_3d(con_3d),
        obstacles(std::move(obstacles)), mappers(std::move(m)), verbose(verbose) {
    assert(max_waypoints >= 4);
  }

  std::vector<std::pair<ExitCode, QPVector>> run(const std::vector<Ctrl<N_DIM>> &starts, const std::vector<Ctrl<N_DIM>> &ends) {
    const size_t B = starts.size();
    assert(ends.size() == B && B > 0);
    starts_ = &starts; ends_ = &ends;
    traj_.clear(); traj_.resize(B);
    segments_run.assign(B, 0); qp_solves.assign(B, 0); qp_updates.assign(B, 0);
    advances = 0; solver_reuses = 0;
    for (size_t b = 0; b < B; ++b) {
      traj_[b].last_solution = linspace<N_DIM>(starts[b], ends[b], max_waypoints);      // joint-space line, zero velocities
      traj_[b].last_solution.resize(2 * max_waypoints * N_DIM, 0.0);
      traj_[b].builder = std::make_unique<ConstraintBuilder<N_DIM>>(4, std::vector<RobotBall>{}, std::vector<HorizontalLine>{});
    }
    for (int s = 0; s < SEGMENTS; ++s) {
      stages_[s].waypoints = max_waypoints * (size_t)(SEGMENTS - s) / SEGMENTS;
      stages_[s].inbox.clear();
      stages_[s].first_admission = true;
      stages_[s].seconds_admit = stages_[s].seconds_wait = stages_[s].seconds_process = stages_[s].seconds_idle = 0.0;
      stages_[s].n_advances = 0;
    }
    for (size_t b = 0; b < B; ++b) stages_[0].inbox.push_back(b);
    finished_ = 0; failed_ = false;
    std::vector<std::thread> th;
    for (int s = 0; s < SEGMENTS; ++s) th.emplace_back([this, s, B] { stageLoop(s, B); });
    for (auto &t : th) t.join();
    std::vector<std::pair<ExitCode, QPVector>> out(B);
    for (size_t b = 0; b < B; ++b) {
      QPVector &x = traj_[b].last_solution;
      for (size_t t = x.size() / 2; t < x.size(); ++t) x[t] /= time_step;
      out[b] = {traj_[b].last_code, x};
    }
    return out;
  }

  // per-trajectory counters (comparable with GOMPSolver's)
  std::vector<int> segments_run, qp_solves, qp_updates;
  std::atomic<long> advances{0};       // advance() calls of all stages
  std::atomic<int> solver_reuses{0};   // stages whose solver was kept from the previous run()
  // Re-linearise on the device (mi_gomp_scene: FK / Jacobians of built-in kinematic models, row assembly, feasibility check,
  // update from device-resident rows) instead of on the stage's host thread.  Needs every ball to name its model
  // (RobotBall::withBuiltin); the trajectories then agree with the host path to round-off (device sin / cos), not bitwise.
  bool device_assembly = false;
  int pipeline_depth = 1;              // 2: a stage enqueues its next advance before it looks at the previous one's results
  int segments_per_advance = 4;        // a launch runs on until a QP of the stage finishes, at most that many segments
  // per stage: {waypoints, advances, seconds admitting, waiting for the device, processing finished QPs, idle}
  std::vector<std::array<double, 6>> stageProfile() const {
    std::vector<std::array<double, 6>> r;
    for (const Stage &st : stages_) r.push_back({(double)st.waypoints, (double)st.n_advances, st.seconds_admit, st.seconds_wait, st.seconds_process, st.seconds_idle});
    return r;
  }

 private:
  const size_t max_waypoints;
  const double time_step;
  const Constraint<N_DIM> pos_con, vel_con, acc_con;
  const Constraint<3> con_3d;
  const std::vector<HorizontalLine> obstacles;
  const std::vector<RobotBall> mappers;
  const bool verbose;

  struct Traj {
    QPVector last_solution, warm, seg_solution;
    ExitCode last_code = ExitCode::kUnknown, seg_code = ExitCode::kUnknown;
    std::unique_ptr<ConstraintBuilder<N_DIM>> builder;
    QPConstraints cons;
    int sqp_it = 0;
  };
  struct Stage {
    size_t waypoints = 0;
    std::unique_ptr<ContSolverT> qp;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<size_t> inbox;                       // trajectories that have reached this horizon
    mi_gomp_scene *scene = nullptr;                 // device-side re-linearisation of this stage's solver (device_assembly)
    ~Stage() { if (scene) mi_gomp_scene_free(scene); }
    bool first_admission = true;
    // where the stage thread's time went in the last run(): preparing + admitting arrivals, waiting for the device in
    // poll(), checking / re-linearising / updating finished QPs, waiting for arrivals
    double seconds_admit = 0.0, seconds_wait = 0.0, seconds_process = 0.0, seconds_idle = 0.0;
    long n_advances = 0;
  };
  std::vector<Traj> traj_;
  std::array<Stage, SEGMENTS> stages_;
  const std::vector<Ctrl<N_DIM>> *starts_ = nullptr, *ends_ = nullptr;
  std::atomic<size_t> finished_{0};
  std::atomic<bool> failed_{false};

  void wakeAll() { for (Stage &st : stages_) { std::lock_guard<std::mutex> lk(st.mu); st.cv.notify_all(); } }

  void stageLoop(int s, size_t B) {
    Stage &st = stages_[(size_t)s];
    const size_t W = st.waypoints;
    const QPMatrixSparse P = triDiagonalMatrix(2, -1, (int)(N_DIM * 2 * W), (int)(W * N_DIM), (int)N_DIM);
    const ConstraintBuilder<N_DIM> tmpl = jointSpaceTemplate((*starts_)[0], (*ends_)[0], W);
    int in_flight = 0;                               // advances enqueued and not polled
    std::vector<size_t> arrivals;
    using clk_ = std::chrono::steady_clock;
    auto lap = [t = clk_::now()](double &acc) mutable { const auto now = clk_::now(); acc += std::chrono::duration<double>(now - t).count(); t = now; };
    double other = 0.0;
    while (finished_.load() < B && !failed_.load()) {
      // ---- trajectories that have reached this horizon: a freshly constructed QP each ([REF] src/gomp-solver.h:57-65)
      arrivals.clear();
      {
        std::unique_lock<std::mutex> lk(st.mu);
        const bool idle = in_flight == 0 && (!st.qp || st.qp->running() == 0);
        lap(other);
        if (idle) st.cv.wait(lk, [&] { return !st.inbox.empty() || finished_.load() >= B || failed_.load(); });
        lap(st.seconds_idle);
        while (!st.inbox.empty()) { arrivals.push_back(st.inbox.front()); st.inbox.pop_front(); }
      }
      if (!arrivals.empty()) { admit(st, P, tmpl, arrivals, B); lap(st.seconds_admit); }
      if (!st.qp) continue;
      // ---- one segment for everything that iterates here; the finished QPs decide the next step of their trajectories
      if (st.qp->running() > 0 && in_flight < pipeline_depth) {
        if (!st.qp->advance(segments_per_advance)) { failed_ = true; wakeAll(); break; }
        ++in_flight; ++advances; ++st.n_advances;
      }
      if (in_flight == 0) continue;
      if (in_flight < pipeline_depth && st.qp->running() > 0) continue;       // (fill the pipeline first)
      lap(other);
      const std::vector<long long> done = st.qp->poll();
      lap(st.seconds_wait);
      --in_flight;
      if (!done.empty()) { process(s, st, done, B); lap(st.seconds_process); }
    }
    // leave nothing enqueued behind
    while (st.qp && in_flight-- > 0) (void)st.qp->poll();
  }

  void admit(Stage &st, const QPMatrixSparse &P, const ConstraintBuilder<N_DIM> &tmpl, const std::vector<size_t> &arrivals, size_t B) {
    const size_t W = st.waypoints;
    auto prepare = [&](size_t k) {
      Traj &T = traj_[arrivals[k]];
      const QPVector &prev = T.last_solution;
      T.warm.resize(W * N_DIM * 2);
      for (size_t t = 0; t < W * N_DIM; ++t) {               // same slicing as GOMPSolver::run
        T.warm[t] = prev[t];
        T.warm[W * N_DIM + t] = prev[W * N_DIM + t];
      }
      *T.builder = tmpl;
      T.builder->position(0, constraints::equal<N_DIM>((*starts_)[arrivals[k]]))
          .position(W - 3, constraints::equal<N_DIM>((*ends_)[arrivals[k]]))
          .withObstacles(con_3d, T.warm);
      T.cons = T.builder->build();
      T.seg_solution = T.warm; T.seg_code = ExitCode::kUnknown; T.sqp_it = 0;
    };
    parallelFor(arrivals.size(), prepare);
    if (st.qp && !st.qp->compatible((long long)B, traj_[arrivals[0]].cons, P)) { if (st.scene) { mi_gomp_scene_free(st.scene); st.scene = nullptr; } st.qp.reset(); }
    if (st.qp) { if (st.first_admission) ++solver_reuses; }                       // (kept from the previous run(): every slot is re-initialised on arrival)
    else st.qp = std::make_unique<ContSolverT>((long long)B, traj_[arrivals[0]].cons, P, verbose);
    st.first_admission = false;
    if (useDevice() && !st.scene) makeScene(st);
    std::vector<long long> ids;
    std::vector<const QPConstraints *> cs;
    std::vector<const QPVector *> xs;
    for (size_t b : arrivals) { ids.push_back((long long)b); cs.push_back(&traj_[b].cons); xs.push_back(&traj_[b].warm); }
    st.qp->reinit(ids, cs);
    // (with a device scene the library keeps these rows on the device as well: the re-linearisations to come start from them)
    st.qp->setWarmStart(ids, xs);
    st.qp->begin(ids);
  }

  bool useDevice() const {
    if (!device_assembly) return false;
    for (const RobotBall &b : mappers) if (!b.builtin_model) return false;
    return true;
  }
  void makeScene(Stage &st) {
    std::vector<mi_gomp_ball> balls;
    for (const RobotBall &b : mappers) {
      mi_gomp_ball g{};
      g.model = b.builtin_model; g.is_gripper = b.is_gripper ? 1 : 0; g.radius = b.radius;
      for (size_t k = 0; k < 12; ++k) g.param[k] = b.builtin_param[k];
      balls.push_back(g);
    }
    std::vector<mi_gomp_line> lines;
    for (const HorizontalLine &l : obstacles) {
      mi_gomp_line g{};
      g.dir[0] = l.directionXY()[0]; g.dir[1] = l.directionXY()[1];
      for (int k = 0; k < 3; ++k) g.point[k] = l.point()[k];
      g.below = l.bypassFromBelow() ? 1 : 0;
      lines.push_back(g);
    }
    double lo[3], hi[3];
    for (int k = 0; k < 3; ++k) { lo[k] = con_3d.first ? (*con_3d.first)[k] : -INF; hi[k] = con_3d.second ? (*con_3d.second)[k] : INF; }
    const int rc = mi_gomp_scene_create(&st.scene, st.qp->handle(), (int64_t)N_DIM, (int64_t)st.waypoints, (int64_t)balls.size(), balls.data(),
                                        (int64_t)lines.size(), lines.data(), lo, hi);
    if (rc != MI_OSQP_OK) std::fprintf(stderr, "ContinuousGOMPSolver: device scene failed (%s): falling back to the host path\n", mi_osqp_error_name(rc));
  }

  void process(int s, Stage &st, const std::vector<long long> &done, size_t B) {
    std::vector<long long> again;
    std::vector<const QPConstraints *> cs;
    std::vector<size_t> leaving;
    if (st.scene) {
      // the SQP step on the device: feasibility check, re-linearisation and update of the QPs whose trajectory is not accepted
      std::vector<long long> opt;
      std::vector<QPVector> sol;
      std::vector<double> xs;
      for (long long id : done) {
        const size_t b = (size_t)id;
        Traj &T = traj_[b];
        auto [exit_code, solution] = st.qp->result(id);
        ++qp_solves[b];
        if (exit_code != ExitCode::kOptimal) { T.seg_solution = std::move(solution); leaving.push_back(b); continue; }
        opt.push_back(id); xs.insert(xs.end(), solution.begin(), solution.end()); sol.push_back(std::move(solution));
      }
      std::vector<int32_t> ok(opt.size(), 0);
      if (!opt.empty()) {
        const int rc = mi_gomp_relinearise_some(st.scene, (int64_t)opt.size(), reinterpret_cast<const int64_t *>(opt.data()), xs.data(), ok.data());
        if (rc != MI_OSQP_OK) { std::fprintf(stderr, "ContinuousGOMPSolver: device re-linearisation failed: %s (%s)\n", mi_osqp_error_name(rc), mi_osqp_last_error()); failed_ = true; wakeAll(); return; }
      }
      for (size_t k = 0; k < opt.size(); ++k) {
        const size_t b = (size_t)opt[k];
        Traj &T = traj_[b];
        if (ok[k]) { T.seg_solution = std::move(sol[k]); T.seg_code = ExitCode::kOptimal; leaving.push_back(b); continue; }
        ++qp_updates[b];
        if (++T.sqp_it < MAX_ITERATIONS) again.push_back(opt[k]); else leaving.push_back(b);
      }
      if (!again.empty()) st.qp->begin(again);
      again.clear();
    } else
    for (long long id : done) {
      const size_t b = (size_t)id;
      Traj &T = traj_[b];
      auto [exit_code, solution] = st.qp->result(id);
      ++qp_solves[b];
      bool ends_here = true;
      if (exit_code != ExitCode::kOptimal) T.seg_solution = std::move(solution);                          // there are no solutions
      else if (isSolutionOK(solution)) { T.seg_solution = std::move(solution); T.seg_code = ExitCode::kOptimal; }
      else {
        // (the reference re-linearises and updates also in the last of its MAX_ITERATIONS rounds; that update has no reader)
        ++qp_updates[b];
        if (++T.sqp_it < MAX_ITERATIONS) {
          T.cons = T.builder->withObstacles(con_3d, solution).build();
          again.push_back(id); cs.push_back(&T.cons);
          ends_here = false;
        }
      }
      if (ends_here) leaving.push_back(b);
    }
    if (!again.empty()) { st.qp->update(again, cs); st.qp->begin(again); }
    for (size_t b : leaving) {
      Traj &T = traj_[b];
      ++segments_run[b];
      if (T.seg_code == ExitCode::kOptimal) { T.last_code = ExitCode::kOptimal; T.last_solution = T.seg_solution; }
      if (s + 1 < SEGMENTS) {
        Stage &nx = stages_[(size_t)s + 1];
        { std::lock_guard<std::mutex> lk(nx.mu); nx.inbox.push_back(b); }
        nx.cv.notify_one();
      } else if (finished_.fetch_add(1) + 1 >= B) wakeAll();
    }
  }

  template <class F>
  static void parallelFor(size_t count, F &&body) {
    const size_t nt = std::min<size_t>({count / 8, 8, std::max(1u, std::thread::hardware_concurrency())});
    if (nt <= 1) { for (size_t k = 0; k < count; ++k) body(k); return; }
    std::vector<std::thread> th;
    for (size_t t = 0; t < nt; ++t) th.emplace_back([&, t] { for (size_t k = t; k < count; k += nt) body(k); });
    for (auto &x : th) x.join();
  }

  // initConstraints of GOMPSolver ([REF] src/gomp-solver.h:98-110) without its last call (the obstacle rows), normalised
  ConstraintBuilder<N_DIM> jointSpaceTemplate(const Ctrl<N_DIM> &start_pos, const Ctrl<N_DIM> &end_pos, size_t waypoints) const {
    ConstraintBuilder<N_DIM> b{waypoints, mappers, obstacles};
    b.position(0, constraints::equal<N_DIM>(start_pos))
        .positions(1, waypoints - 2, pos_con)
        .position(waypoints - 3, constraints::equal<N_DIM>(end_pos))
        .velocities(0, waypoints - 4, vel_con)
        .velocity(waypoints - 3, constraints::eqZero<N_DIM>())
        .accelerations(0, waypoints - 4, acc_con)
        .acceleration(waypoints - 3, constraints::eqZero<N_DIM>())
        .normalised();
    return b;
  }

  bool isSolutionOK(const QPVector &q_trajectory) const {
    bool res = true;
    for (const RobotBall &ball : mappers) {
      const QPVector xyz = mapJointTrajectoryToXYZ<N_DIM>(q_trajectory, ball.fk);
      const int waypoints = (int)xyz.size() / 3;
      for (int w = 0; w < waypoints; ++w) {
        const Point p{xyz[3 * w], xyz[3 * w + 1], xyz[3 * w + 2]};
        if (ball.is_gripper) {
          for (Axis axis : XYZ_AXES) {
            const double lo = con_3d.first ? (*con_3d.first)[axis] : -INF;
            const double up = con_3d.second ? (*con_3d.second)[axis] : INF;
            if (!(lo - ERROR <= p[axis] - ball.radius && p[axis] + ball.radius <= up + ERROR)) res = false;
          }
        }
        for (const HorizontalLine &line : obstacles)
          if (line.hasCollision(w, xyz, ball) && !line.isAbove(p, ball)) res = false;
      }
    }
    return res;
  }
};

}  // namespace miosqp_ref
