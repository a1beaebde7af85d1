// ur5e_kinematics.hpp -- analytic UR5e forward kinematics / position Jacobians with the call signatures
// the reference's example binds into RobotBall ([REF] /root/reference/examples/solver-example.cpp:31,38-40,
// 54,98; types [REF] src/utils.h:21-22, src/gomp-solver.h:9).
//
// The reference takes these functions from an external library (Kinematics-UR5e-arm, header
// "analytical_ik.h", fetched by CMake from the network: [REF] src/CMakeLists.txt:6-12) that is NOT part of
// the reference tree, so its exact frames cannot be read.  This is an independent model built from Universal
// Robots' published DH parameters of the UR5e,
//     a     = {0, -0.425, -0.3922, 0, 0, 0}           [m]
//     d     = {0.1625, 0, 0, 0.1333, 0.0997, 0.0996}  [m]
//     alpha = {pi/2, 0, 0, pi/2, -pi/2, 0}
// with T_i = Rz(q_i) Tz(d_i) Tx(a_i) Rx(alpha_i).  Same names, same argument order:
//     forward_kinematics(q)             tool flange (origin of frame 6)
//     forward_kinematics_6_back(q)      wrist-3 joint (origin of frame 5 = flange moved back by d6 along its z axis)
//     forward_kinematics_elbow_joint(q) elbow joint (origin of frame 2)
//     joint_jacobian / joint_jacobian_6_back / jacobian_elbow_joint (jac, q)
//                                       3 x 6 position Jacobians of those points, ROW-major like the reference's
//                                       QPMatrix<3, N> ([REF] src/utils.h:13-14)
//     inverse_kinematics(out, x, y, z)  position-only IK by damped least squares from the zero pose; returns the
//                                       number of solutions written (0 or 1).  The reference stores the callback
//                                       and never calls it ([REF] src/gomp-solver.h:25,103).
#pragma once

#include <array>
#include <cmath>
#include <tuple>

namespace ur5e {

constexpr double A[6] = {0.0, -0.425, -0.3922, 0.0, 0.0, 0.0};
constexpr double Dd[6] = {0.1625, 0.0, 0.0, 0.1333, 0.0997, 0.0996};
constexpr double ALPHA[6] = {1.5707963267948966, 0.0, 0.0, 1.5707963267948966, -1.5707963267948966, 0.0};

struct Frame { double R[3][3]; double p[3]; };

inline Frame identity() { return {{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}}, {0, 0, 0}}; }
// F <- F * T_i(q)
inline void advance(Frame &F, int i, double q) {
  const double ct = std::cos(q), st = std::sin(q), ca = std::cos(ALPHA[i]), sa = std::sin(ALPHA[i]);
  const double T[3][4] = {{ct, -st * ca, st * sa, A[i] * ct}, {st, ct * ca, -ct * sa, A[i] * st}, {0.0, sa, ca, Dd[i]}};
  Frame G;
  for (int r = 0; r < 3; r++) {
    for (int c = 0; c < 3; c++) G.R[r][c] = F.R[r][0] * T[0][c] + F.R[r][1] * T[1][c] + F.R[r][2] * T[2][c];
    G.p[r] = F.R[r][0] * T[0][3] + F.R[r][1] * T[1][3] + F.R[r][2] * T[2][3] + F.p[r];
  }
  F = G;
}
// origins o_0..o_6 and joint axes z_0..z_5 (axis of joint i+1 = z of frame i)
struct Chain { double o[7][3]; double z[6][3]; };
inline Chain chain(const double *q) {
  Chain c;
  Frame F = identity();
  for (int i = 0; i < 6; i++) {
    for (int r = 0; r < 3; r++) { c.o[i][r] = F.p[r]; c.z[i][r] = F.R[r][2]; }
    advance(F, i, q[i]);
  }
  for (int r = 0; r < 3; r++) c.o[6][r] = F.p[r];
  return c;
}
// position Jacobian of the origin of frame `frame` (joints >= frame do not move it): column j = z_j x (p - o_j)
inline void point_jacobian(double *jac, const double *q, int frame) {
  const Chain c = chain(q);
  const double *p = c.o[frame];
  for (int j = 0; j < 6; j++) {
    double col[3] = {0, 0, 0};
    if (j < frame) {
      const double r[3] = {p[0] - c.o[j][0], p[1] - c.o[j][1], p[2] - c.o[j][2]};
      col[0] = c.z[j][1] * r[2] - c.z[j][2] * r[1];
      col[1] = c.z[j][2] * r[0] - c.z[j][0] * r[2];
      col[2] = c.z[j][0] * r[1] - c.z[j][1] * r[0];
    }
    for (int axis = 0; axis < 3; axis++) jac[axis * 6 + j] = col[axis];
  }
}

}  // namespace ur5e

inline std::tuple<double, double, double> forward_kinematics(double *q) {
  const ur5e::Chain c = ur5e::chain(q);
  return {c.o[6][0], c.o[6][1], c.o[6][2]};
}
inline std::tuple<double, double, double> forward_kinematics_6_back(double *q) {
  const ur5e::Chain c = ur5e::chain(q);
  return {c.o[5][0], c.o[5][1], c.o[5][2]};
}
inline std::tuple<double, double, double> forward_kinematics_elbow_joint(double *q) {
  const ur5e::Chain c = ur5e::chain(q);
  return {c.o[2][0], c.o[2][1], c.o[2][2]};
}
inline void joint_jacobian(double *jac, double *q) { ur5e::point_jacobian(jac, q, 6); }
inline void joint_jacobian_6_back(double *jac, double *q) { ur5e::point_jacobian(jac, q, 5); }
inline void jacobian_elbow_joint(double *jac, double *q) { ur5e::point_jacobian(jac, q, 2); }

inline int inverse_kinematics(double *out, double x, double y, double z) {
  double q[6] = {0.3, -1.0, 1.0, -1.0, -1.0, 0.0};       // away from the stretched-out singular pose
  const double target[3] = {x, y, z};
  for (int it = 0; it < 200; it++) {
    const auto [px, py, pz] = forward_kinematics(q);
    const double e[3] = {target[0] - px, target[1] - py, target[2] - pz};
    if (std::sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]) < 1e-10) {
      for (int j = 0; j < 6; j++) out[j] = q[j];
      return 1;
    }
    double J[18];
    joint_jacobian(J, q);
    // damped least squares: dq = J' (J J' + lambda I)^-1 e
    double M[3][3];
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) {
        M[a][b] = a == b ? 1e-6 : 0.0;
        for (int j = 0; j < 6; j++) M[a][b] += J[a * 6 + j] * J[b * 6 + j];
      }
    const double det = M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) +
                       M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
    if (std::fabs(det) < 1e-300) return 0;
    double y3[3];
    for (int a = 0; a < 3; a++) {       // Cramer
      double N[3][3];
      for (int r = 0; r < 3; r++) for (int cc = 0; cc < 3; cc++) N[r][cc] = cc == a ? e[r] : M[r][cc];
      y3[a] = (N[0][0] * (N[1][1] * N[2][2] - N[1][2] * N[2][1]) - N[0][1] * (N[1][0] * N[2][2] - N[1][2] * N[2][0]) +
               N[0][2] * (N[1][0] * N[2][1] - N[1][1] * N[2][0])) / det;
    }
    for (int j = 0; j < 6; j++) q[j] += J[0 * 6 + j] * y3[0] + J[1 * 6 + j] * y3[1] + J[2 * 6 + j] * y3[2];
  }
  return 0;
}
