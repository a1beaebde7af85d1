// examples/solver_example.cpp -- BASELINE.json config 1: one small QP through the
// QPSolver facade (plumbing check).  The reference's own example
// ([REF] /root/reference/examples/solver-example.cpp) needs the absent UR5e
// kinematics library, so this drives the same four QPSolver calls
// (ctor / setWarmStart / solve / update) on the textbook QP
//   min 1/2 x'[4 1;1 2]x   s.t.  x0 + x1 = 1, 0 <= x <= 0.7      (q = 0 as in the wrapper)
// and prints exit codes with ToString like [REF] examples/solver-example.cpp:71.
#include <cstdio>
#include <iostream>

#include "mi_osqp/qp_solver.hpp"

using namespace miosqp_ref;

int main() {
  QPMatrixSparse P;                         // both triangles, as triDiagonalMatrix emits them
  P.rows = P.cols = 2; P.outer = {0, 2, 4}; P.inner = {0, 1, 0, 1}; P.values = {4, 1, 1, 2};
  QPMatrixSparse A;
  A.rows = 3; A.cols = 2; A.outer = {0, 2, 4}; A.inner = {0, 1, 0, 2}; A.values = {1, 1, 1, 1};
  QPVector l = {1, 0, 0}, u = {1, 0.7, 0.7};

  QPSolver solver({l, A, u}, P);
  if (solver.setup_status() != 0) { std::printf("setup failed: %d\n", solver.setup_status()); return 2; }
  solver.setWarmStart({0.5, 0.5});
  auto [code, x] = solver.solve();
  std::cout << ToString(code) << std::endl;
  std::printf("x = %.6f %.6f iters %lld\n", x[0], x[1], (long long)solver.last_info().iter);

  // re-linearisation step of the SQP loop: same pattern, new values and bounds
  A.values = {1, 1, 1, 2};
  solver.update({{1, 0, 0}, A, {1, 0.7, 1.0}});
  auto [code2, x2] = solver.solve();
  std::cout << ToString(code2) << std::endl;
  std::printf("x = %.6f %.6f iters %lld\n", x2[0], x2[1], (long long)solver.last_info().iter);

  // a pattern change must throw std::invalid_argument like the reference
  QPMatrixSparse A_bad = A;
  A_bad.inner = {0, 2, 0, 2};
  try { solver.update({l, A_bad, u}); std::printf("ERROR: no throw\n"); return 3; }
  catch (const std::invalid_argument &e) { std::printf("update refused: %s\n", e.what()); }
  return (code == OsqpExitCode::kOptimal && code2 == OsqpExitCode::kOptimal) ? 0 : 1;
}
