// Compile / run check of include/osqp++.h (the osqp-cpp-shaped shim on the MI355X C-ABI).
//   -DMI_REF_WRAPPER="\"/root/reference/src/osqp-wrapper.h\""  (build container only: the reference's own header text is
//       included where it lies, never copied): class QPSolver of the reference must compile UNCHANGED against the shim,
//       and main() drives it through its four methods ([REF] src/osqp-wrapper.h:16,33,45,51).
//   without it: main() drives osqp::OsqpSolver directly with the call sequence of that wrapper.
// Output: one line of JSON (exit code name, iterations, solution) that tests/test_osqp_shim.py compares with the
// ctypes binding; on a machine without a gfx950 GPU Init reports the device error and Solve() returns kUnknown, like an
// uninitialised osqp-cpp solver does.
#include <cassert>
#include <cstdio>
#include <functional>
#include <iostream>
#include <stdexcept>
#include <tuple>

#include <osqp++.h>

using QPMatrixSparse = Eigen::SparseMatrix<double, Eigen::ColMajor, long long>;      // [REF] src/utils.h:12
using QPVector = Eigen::VectorXd;                                                    // [REF] src/utils.h:15
using QPConstraints = std::tuple<QPVector, QPMatrixSparse, QPVector>;                // [REF] src/constraints/constraint-builder.h:16
using ExitCode = osqp::OsqpExitCode;                                                 // [REF] src/utils.h:11

#ifdef MI_REF_WRAPPER
#include MI_REF_WRAPPER
#endif

static QPMatrixSparse from_dense(int rows, int cols, const double *a) {
  std::vector<Eigen::Triplet<double, long long>> t;
  for (int c = 0; c < cols; c++) for (int r = 0; r < rows; r++) if (a[r * cols + c] != 0.0) t.emplace_back(r, c, a[r * cols + c]);
  QPMatrixSparse M(rows, cols);
  M.setFromTriplets(t.begin(), t.end());
  return M;
}

int main() {
  // the QP of upstream's documentation demo: min 1/2 x'Px (q = 0 here, as the reference's wrapper fixes it), 3 constraints
  const double Pd[4] = {4, 1, 1, 2}, Ad[6] = {1, 1, 1, 0, 0, 1};
  QPMatrixSparse P = from_dense(2, 2, Pd), A = from_dense(3, 2, Ad);
  QPVector l(3), u(3), warm(2);
  l[0] = 1; l[1] = 0; l[2] = 0; u[0] = 1; u[1] = 0.7; u[2] = 0.7; warm[0] = 0.3; warm[1] = 0.7;
  QPConstraints c{l, A, u};
  QPVector l2 = l, u2 = u;
  u2[1] = 0.6; u2[2] = 0.9;
  QPConstraints c2{l2, A, u2};
#ifdef MI_REF_WRAPPER
  QPSolver solver(c, P);                                   // the reference's class, compiled from its own header
  bool threw = false;
  try { solver.setWarmStart(warm); } catch (...) { threw = true; }
  auto [code1, x1] = solver.solve();
  try { solver.update(c2); } catch (const std::invalid_argument &e) { threw = true; std::cout << "update: " << e.what() << std::endl; }
  auto [code2, x2] = solver.solve();
  const long long it1 = -1, it2 = -1;
#else
  osqp::OsqpInstance instance;
  instance.constraint_matrix = A; instance.objective_matrix = P; instance.objective_vector.setZero(A.cols());
  instance.lower_bounds = l; instance.upper_bounds = u;
  osqp::OsqpSettings settings;
  settings.verbose = true;
  osqp::OsqpSolver solver;
  absl::Status status = solver.Init(instance, settings);
  std::cout << "Init: " << status.ToString() << std::endl;
  bool threw = !status.ok();
  status = solver.SetPrimalWarmStart(warm);
  std::cout << "STATUS: " << status.ToString() << std::endl;
  ExitCode code1 = solver.Solve();
  QPVector x1 = solver.primal_solution();
  const long long it1 = solver.iterations();
  if (!(status = solver.UpdateConstraintMatrix(A)).ok()) threw = true;
  if (!(status = solver.SetBounds(l2, u2)).ok()) threw = true;
  ExitCode code2 = solver.Solve();
  QPVector x2 = solver.primal_solution();
  const long long it2 = solver.iterations();
  // a changed sparsity pattern must be refused like osqp-cpp does (InvalidArgument -> the reference throws)
  const double Bd[6] = {1, 1, 1, 0, 1, 1};
  const bool refused = solver.IsInitialized() ? !solver.UpdateConstraintMatrix(from_dense(3, 2, Bd)).ok() : true;
  if (!refused) return 3;
#endif
  auto num = [](const QPVector &x, int i) { return x.size() > i ? x[i] : 0.0; };
  std::printf("{\"code1\": \"%s\", \"code2\": \"%s\", \"it1\": %lld, \"it2\": %lld, \"x1\": [%.17g, %.17g], \"x2\": [%.17g, %.17g], \"threw\": %s}\n",
              osqp::ToString(code1).c_str(), osqp::ToString(code2).c_str(), it1, it2, num(x1, 0), num(x1, 1), num(x2, 0), num(x2, 1),
              threw ? "true" : "false");
  return 0;
}
