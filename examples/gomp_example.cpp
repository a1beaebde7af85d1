// examples/gomp_example.cpp -- the reference's example program ([REF] /root/reference/examples/solver-example.cpp)
// on the MI355X QPSolver: a 6-DOF UR5e moves joint 1 by pi while two collision balls (wrist, radius 0.15; gripper,
// radius 0.05) respect the work-space limit y >= -0.4; the optimised trajectory is written in the reference's two
// text formats ([REF] :73-81):
//     output_trajectory_ctrl.data   D joint values per line, separated by blanks
//     output_trajectory_xyz.data    "(x, y, z)" of the elbow joint per line
// Differences, all forced by what is absent from the reference tree: the kinematics are
// include/mi_osqp/ur5e_kinematics.hpp (own DH model, see there), and the default horizon is 50 + 2 waypoints
// instead of 800 + 2 (pass the number of waypoints as argv[1]; 802 runs in the global-vector mode).
//   usage: gomp_example [waypoints] [obstacle: 0|1] [output directory]
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>

#include "mi_osqp/gomp.hpp"
#include "mi_osqp/ur5e_kinematics.hpp"

using namespace miosqp_ref;

constexpr double TIME_STEP = 0.1;
constexpr size_t DIMS = 6;
constexpr double Q_MIN = -2 * M_PI, Q_MAX = 2 * M_PI;

static Point toPoint(Ctrl<DIMS> c) {
  auto [x, y, z] = forward_kinematics_elbow_joint(c.data());
  return {x, y, z};
}
static std::string str(const Point &p) {
  char b[96];
  std::snprintf(b, sizeof b, "(%g, %g, %g)", p[0], p[1], p[2]);
  return b;
}

int main(int argc, char **argv) {
  const size_t WAYPOINTS = argc > 1 ? (size_t)std::atoi(argv[1]) : 50 + 2;
  const bool with_obstacle = argc > 2 && std::atoi(argv[2]) != 0;
  const std::string dir = argc > 3 ? std::string(argv[3]) + "/" : "";

  std::vector<RobotBall> mappers{
      RobotBall(&forward_kinematics_6_back, &joint_jacobian_6_back, 0.15, false),
      RobotBall(&forward_kinematics, &joint_jacobian, 0.05, true),
  };
  std::vector<HorizontalLine> obstacles;
  if (with_obstacle) obstacles.push_back(HorizontalLine({0, 1}, {0.3, 0, 0.35}, false));   // a bar along y under the path

  GOMPSolver<DIMS> solver(WAYPOINTS, TIME_STEP,
                          constraints::inRange<DIMS>(constraints::of<DIMS>(Q_MIN), constraints::of<DIMS>(Q_MAX)),
                          constraints::inRange<DIMS>(constraints::of<DIMS>(-M_PI), constraints::of<DIMS>(M_PI)),
                          constraints::inRange<DIMS>(constraints::of<DIMS>(-M_PI * 800 / 180), constraints::of<DIMS>(M_PI * 800 / 180)),
                          constraints::inRange<3>(Vec<3>{-INF, -0.4, -INF}, Vec<3>{INF, INF, INF}),
                          obstacles, mappers, &inverse_kinematics, false);

  const Point start_pos_gt = toPoint({0, 0, 0, 0, 0, 0}), end_pos_gt = toPoint({M_PI, 0, 0, 0, 0, 0});
  auto [e, b1] = solver.run({0, 0, 0, 0, 0, 0}, {M_PI, 0, 0, 0, 0, 0});
  std::cout << ToString(e) << std::endl;

  std::ofstream output_file_ctrl(dir + "output_trajectory_ctrl.data"), output_file_xyz(dir + "output_trajectory_xyz.data");
  const size_t n_way = b1.size() / DIMS / 2;
  for (size_t i = 0; i < n_way; i++) {
    for (size_t j = 0; j < DIMS; j++) output_file_ctrl << b1[DIMS * i + j] << (j + 1 < DIMS ? " " : "\n");
    const Point point = toPoint({b1[DIMS * i + 0], b1[DIMS * i + 1], b1[DIMS * i + 2], b1[DIMS * i + 3], b1[DIMS * i + 4], b1[DIMS * i + 5]});
    output_file_xyz << "(" << point[0] << ", " << point[1] << ", " << point[2] << ")" << "\n";
  }
  output_file_ctrl.close();
  output_file_xyz.close();

  auto at = [&](size_t w) { return toPoint({b1[DIMS * w + 0], b1[DIMS * w + 1], b1[DIMS * w + 2], b1[DIMS * w + 3], b1[DIMS * w + 4], b1[DIMS * w + 5]}); };
  std::cout << "\n\nSummary:\n";
  std::cout << "Ground true starting position: " << str(start_pos_gt) << " starting position after optimization: " << str(at(0)) << "\n";
  std::cout << "Middle position after optimization: " << str(at(n_way > 10 ? 10 : n_way / 2)) << "\n";
  std::cout << "Ground true end position: " << str(end_pos_gt) << " end position after optimization: " << str(at(n_way - 1)) << "\n\n";
  std::cout << ToString(e) << std::endl;
  std::cout << b1.size() << std::endl;
  std::printf("segments %d qp solves %d re-linearisations %d\n", solver.segments_run, solver.qp_solves, solver.qp_updates);
  return e == ExitCode::kOptimal ? 0 : 1;
}
