// examples/batch_gomp_example.cpp -- many UR5e planning problems at once on one MI355X.
//
// The reference plans ONE trajectory per GOMPSolver::run ([REF] /root/reference/src/gomp-solver.h:38-91): a chain of QPs per
// horizon, each solved, checked, re-linearised and updated in turn.  A planner that evaluates many start / goal pairs (grasp
// candidates, a roadmap's edges) has that many independent chains; ContinuousGOMPSolver runs them side by side - every
// trajectory on its own schedule, the decisions of a sequential run for each - on the per-QP entry points of mi_osqp.h.
// The scene is the reference example's ([REF] examples/solver-example.cpp:31-70): a UR5e, a collision ball at wrist 3 and
// the gripper ball at the flange, the wall y >= -0.4, plus a bar to pass above.
//
//   usage: batch_gomp_example [trajectories = 64] [waypoints = 60] [SQP step on the device: 0|1 = 1]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "mi_osqp/gomp.hpp"
#include "mi_osqp/ur5e_kinematics.hpp"

namespace ref = miosqp_ref;
constexpr size_t kJoints = 6;

int main(int argc, char **argv) {
  setenv("GPU_MAX_HW_QUEUES", "10", 0);                 // one hardware queue per horizon stage
  const int n_traj = argc > 1 ? std::atoi(argv[1]) : 64;
  const size_t waypoints = argc > 2 ? (size_t)std::atoi(argv[2]) : 60;
  const bool on_device = argc > 3 ? std::atoi(argv[3]) != 0 : true;
  if (n_traj < 1 || waypoints < 10) { std::fprintf(stderr, "usage: batch_gomp_example [trajectories] [waypoints >= 10] [0|1]\n"); return 2; }

  // the balls name the built-in kinematic model next to the host callbacks: the host path uses the callbacks, the device path
  // the model (same DH parameters: include/mi_osqp/ur5e_kinematics.hpp)
  std::vector<ref::RobotBall> balls{
      ref::RobotBall(&forward_kinematics_6_back, &joint_jacobian_6_back, 0.15, false).withBuiltin(MI_GOMP_MODEL_UR5E_WRIST3),
      ref::RobotBall(&forward_kinematics, &joint_jacobian, 0.05, true).withBuiltin(MI_GOMP_MODEL_UR5E_FLANGE)};
  std::vector<ref::HorizontalLine> bars{ref::HorizontalLine({0, 1}, {0.3, 0, 0.35}, /*bypass from below*/ false)};
  const double pi = 3.14159265358979323846;
  auto joint_limits = ref::constraints::inRange<kJoints>(ref::constraints::of<kJoints>(-2 * pi), ref::constraints::of<kJoints>(2 * pi));
  auto speed_limits = ref::constraints::inRange<kJoints>(ref::constraints::of<kJoints>(-pi), ref::constraints::of<kJoints>(pi));
  auto accel_limits = ref::constraints::inRange<kJoints>(ref::constraints::of<kJoints>(-pi * 800 / 180), ref::constraints::of<kJoints>(pi * 800 / 180));
  auto work_space = ref::constraints::inRange<3>(ref::Vec<3>{-ref::INF, -0.4, -ref::INF}, ref::Vec<3>{ref::INF, ref::INF, ref::INF});

  std::vector<ref::Ctrl<kJoints>> from, to;
  std::mt19937_64 rng(2024);
  std::uniform_real_distribution<double> jitter(-0.05, 0.05);
  for (int t = 0; t < n_traj; ++t) {
    ref::Ctrl<kJoints> a{}, b{};
    for (size_t j = 0; j < kJoints; ++j) { a[j] = jitter(rng); b[j] = jitter(rng); }
    b[0] += pi * (0.9 + 2.0 * jitter(rng));
    from.push_back(a); to.push_back(b);
  }

  ref::ContinuousGOMPSolver<kJoints> planner(waypoints, 0.1, joint_limits, speed_limits, accel_limits, work_space, bars, balls);
  planner.device_assembly = on_device;
  using clock = std::chrono::steady_clock;
  auto t0 = clock::now();
  auto plans = planner.run(from, to);                    // first call: builds the ten per-horizon solvers
  const double first = std::chrono::duration<double>(clock::now() - t0).count();
  t0 = clock::now();
  plans = planner.run(from, to);                         // a planner calls run() again and again: the solvers are kept
  const double again = std::chrono::duration<double>(clock::now() - t0).count();

  int ok = 0, solves = 0, relin = 0;
  for (int t = 0; t < n_traj; ++t) { ok += plans[(size_t)t].first == ref::ExitCode::kOptimal; solves += planner.qp_solves[(size_t)t]; relin += planner.qp_updates[(size_t)t]; }
  std::printf("%d of %d trajectories planned (%zu waypoints, SQP step on the %s)\n", ok, n_traj, waypoints, on_device ? "device" : "host threads");
  std::printf("%d QP solves, %d re-linearisations; first run %.3f s, next run %.3f s = %.1f trajectories/s\n", solves, relin, first, again, n_traj / again);
  return ok == n_traj ? 0 : 1;
}
