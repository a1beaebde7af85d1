// examples/gomp_example.cpp -- a UR5e planning run on the MI355X QPSolver, in the scene of the reference's example
// ([REF] /root/reference/examples/solver-example.cpp:12-16,44-70): six joints, the base joint turns by pi, a wrist ball
// (radius 0.15) and a gripper ball (radius 0.05) must stay inside the work space y >= -0.4, optionally above a bar.
//
// What is kept from the reference is the WIRE FORMAT of the two trajectory dumps ([REF] :73-81), because other tools
// read them:
//     output_trajectory_ctrl.data   one waypoint per line: the D joint angles separated by single blanks
//     output_trajectory_xyz.data    one waypoint per line: "(x, y, z)" of the elbow joint
// Everything else - argument handling, the report, the program structure - is this repository's own.  The kinematics are
// include/mi_osqp/ur5e_kinematics.hpp (the reference's kinematics library is not in its tree).
//
//   usage: gomp_example [waypoints = 52] [bar obstacle: 0|1] [output directory]
//          (the reference plans 802 waypoints; that size runs too, in the global-vector mode of the solver)
#include <array>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>

#include "mi_osqp/gomp.hpp"
#include "mi_osqp/ur5e_kinematics.hpp"

namespace ref = miosqp_ref;
constexpr size_t kJoints = 6;
using JointVector = ref::Ctrl<kJoints>;

struct Scene {
  size_t waypoints = 52;
  bool bar = false;
  std::string out_dir;
  double dt = 0.1;                       // seconds between waypoints
  JointVector from{0, 0, 0, 0, 0, 0};
  JointVector to{M_PI, 0, 0, 0, 0, 0};
};

static Scene parseArgs(int argc, char **argv) {
  Scene sc;
  if (argc > 1 && std::atoi(argv[1]) >= 4) sc.waypoints = (size_t)std::atoi(argv[1]);
  if (argc > 2) sc.bar = std::atoi(argv[2]) != 0;
  if (argc > 3) sc.out_dir = std::string(argv[3]) + "/";
  return sc;
}

static ref::Point elbowOf(const double *joints) {
  JointVector q;
  for (size_t j = 0; j < kJoints; ++j) q[j] = joints[j];
  auto [x, y, z] = forward_kinematics_elbow_joint(q.data());
  return {x, y, z};
}

// the two dumps, byte-compatible with the reference's writer
static void writeTrajectory(const ref::QPVector &plan, size_t waypoints, const std::string &dir) {
  std::ofstream ctrl(dir + "output_trajectory_ctrl.data"), xyz(dir + "output_trajectory_xyz.data");
  for (size_t w = 0; w < waypoints; ++w) {
    const double *q = plan.data() + kJoints * w;
    for (size_t j = 0; j < kJoints; ++j) ctrl << q[j] << (j + 1 < kJoints ? " " : "\n");
    const ref::Point p = elbowOf(q);
    xyz << "(" << p[0] << ", " << p[1] << ", " << p[2] << ")\n";
  }
}

int main(int argc, char **argv) {
  const Scene sc = parseArgs(argc, argv);
  using namespace ref::constraints;
  std::vector<ref::RobotBall> balls{ref::RobotBall(&forward_kinematics_6_back, &joint_jacobian_6_back, 0.15, false),
                                    ref::RobotBall(&forward_kinematics, &joint_jacobian, 0.05, true)};
  std::vector<ref::HorizontalLine> bars;
  if (sc.bar) bars.push_back(ref::HorizontalLine({0, 1}, {0.3, 0, 0.35}, false));       // along y, to be passed from above
  const double rad = M_PI / 180.0;
  ref::GOMPSolver<kJoints> planner(sc.waypoints, sc.dt,
                                   inRange<kJoints>(of<kJoints>(-360 * rad), of<kJoints>(360 * rad)),      // joint range
                                   inRange<kJoints>(of<kJoints>(-180 * rad), of<kJoints>(180 * rad)),      // rad / s
                                   inRange<kJoints>(of<kJoints>(-800 * rad), of<kJoints>(800 * rad)),      // rad / s^2
                                   inRange<3>(ref::Vec<3>{-ref::INF, -0.4, -ref::INF}, ref::Vec<3>{ref::INF, ref::INF, ref::INF}),
                                   bars, balls, &inverse_kinematics, false);
  const auto t0 = std::chrono::steady_clock::now();
  const auto [code, plan] = planner.run(sc.from, sc.to);
  const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  std::printf("%s\n", ref::ToString(code).c_str());

  const size_t kept = plan.size() / (2 * kJoints);          // waypoints of the last horizon that was solved to optimality
  writeTrajectory(plan, kept, sc.out_dir);

  const ref::Point a = elbowOf(plan.data()), b = elbowOf(plan.data() + kJoints * (kept - 1));
  const ref::Point a_want = elbowOf(sc.from.data()), b_want = elbowOf(sc.to.data());
  double lowest_y = 1e30;
  for (size_t w = 0; w < kept; ++w) {
    JointVector q;
    for (size_t j = 0; j < kJoints; ++j) q[j] = plan[kJoints * w + j];
    lowest_y = std::min(lowest_y, std::get<1>(forward_kinematics(q.data())) - 0.05);
  }
  std::printf("plan: %zu of %zu waypoints kept, %d horizons, %d QP solves, %d re-linearisations, %.3f s\n", kept, sc.waypoints,
              planner.segments_run, planner.qp_solves, planner.qp_updates, seconds);
  std::printf("elbow at the first waypoint (%.4f, %.4f, %.4f), requested (%.4f, %.4f, %.4f)\n", a[0], a[1], a[2], a_want[0], a_want[1], a_want[2]);
  std::printf("elbow at the last  waypoint (%.4f, %.4f, %.4f), requested (%.4f, %.4f, %.4f)\n", b[0], b[1], b[2], b_want[0], b_want[1], b_want[2]);
  std::printf("gripper ball: lowest y - radius = %.4f (work-space limit -0.4)%s\n", lowest_y, sc.bar ? "; bar at x = 0.3, z = 0.35" : "");
  std::printf("wrote %soutput_trajectory_ctrl.data and %soutput_trajectory_xyz.data\n", sc.out_dir.c_str(), sc.out_dir.c_str());
  return code == ref::ExitCode::kOptimal ? 0 : 1;
}
