"""The callers of the path as C++ on top of the QPSolver facade:
include/mi_osqp/gomp.hpp = ConstraintBuilder (with obstacle rows), HorizontalLine, GOMPSolver.

CPU : the reference's own known-answer tests for the 3-D rows / line geometry
      ([REF] tests/test.cpp:82-100,250-448, data embedded in tests/cpp/gomp_parity.cpp) and the
      SQP driver on the oracle backend.
GPU : the same driver on the MI355X QPSolver vs on the oracle backend -- identical exit codes,
      identical numbers of QP solves / re-linearisations, trajectories within 1e-6."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    import osqp_solver_amd as M
    from oracle import oracle as O
    M.lib(); O.lib()
    out = tmp_path_factory.mktemp("gomp") / "gomp_parity"
    libdir, ordir = os.path.join(ROOT, "osqp-solver_amd"), os.path.join(ROOT, "oracle", "_build")
    cmd = ["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "gomp_parity.cpp"),
           "-o", str(out), "-L" + libdir, "-lmi_osqp", "-L" + ordir, "-loracle_osqp", "-fopenmp", "-pthread",
           "-Wl,-rpath," + libdir, "-Wl,-rpath," + ordir]
    subprocess.run(cmd, check=True)
    return str(out)


def test_reference_known_answers_for_builder_and_line(exe):
    r = subprocess.run([exe, "kats"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "KATS OK" in r.stdout, r.stdout + r.stderr


def test_sqp_driver_on_oracle_backend(exe):
    r = subprocess.run([exe, "oracle"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = r.stdout.strip().splitlines()
    assert lines[0].startswith("obstacle=0 kOptimal segments 10 solves 10 updates 0")
    assert lines[1].startswith("obstacle=1 kOptimal segments 10") and "updates 0" not in lines[1]


@pytest.mark.gpu
def test_gomp_driver_gpu_matches_oracle_backend(exe):
    r = subprocess.run([exe, "parity"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "PARITY OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_batched_gomp_driver_matches_sequential_drivers(exe):
    """SURVEY 8(f) rank 1, batched: 6 trajectories in lock-step on the batch API take exactly the decisions
    of 6 sequential GOMPSolver runs (GPU QPSolver and oracle twin)."""
    r = subprocess.run([exe, "batch"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "BATCH OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_continuous_gomp_driver_matches_sequential_drivers(exe):
    """SURVEY 8(f) rank 1 as the reference runs it - per trajectory ([REF] src/gomp-solver.h:70-88): the continuous driver
    (per-QP entry points, one stage per horizon, pipeline depths 1 and 2) takes exactly the decisions of sequential
    GOMPSolver runs: exit codes, segment / solve / update counters, trajectories (1e-9 vs GPU, 1e-6 vs the oracle twin)."""
    r = subprocess.run([exe, "cont"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "CONT OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_device_relinearisation_matches_host_constraint_builder(exe):
    """SURVEY 8(f) rank 2 on the device (mi_gomp_scene): the reference's known answers for the 3-D rows
    ([REF] tests/test.cpp:250-448) exactly; rows / bounds / acceptance test of random 3-link-arm and UR5e trajectories
    against ConstraintBuilder::withObstacles (1e-12: device sin / cos); one SQP step on the device against
    QPSolver::update with host rows (solutions 1e-6); the continuous planner with device assembly against the
    host-assembling one (same exit codes and counters, trajectories 1e-6)."""
    r = subprocess.run([exe, "devasm"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "DEVASM OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("device_assembly", ["0", "1"])
def test_continuous_driver_on_the_ur5e_scene(exe, device_assembly):
    """The reference's robot and scene (UR5e, balls at wrist 3 and flange, the wall y >= -0.4, a bar) through the continuous
    driver, SQP step on the host threads and on the device: repeated runs bitwise equal, the first trajectories against the
    sequential driver on the oracle - same exit codes, same solve / update counts, trajectories within 1e-5."""
    r = subprocess.run([exe, "contbench_ur5e", "24", "60", "3"], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, GOMP_DEVICE_ASSEMBLY=device_assembly))
    assert r.returncode == 0 and "CONTBENCH_UR5E OK" in r.stdout and "Memory access fault" not in r.stdout + r.stderr, r.stdout + r.stderr


def test_ur5e_kinematics_and_example_scenario_on_oracle(exe):
    """own UR5e FK / Jacobians (published DH parameters; the reference's kinematics library is
    absent) -- zero-pose position, Jacobians vs central differences, IK round trip -- and the scenario of
    examples/gomp_example.cpp (two collision balls, y >= -0.4, optional bar obstacle) on the oracle backend."""
    r = subprocess.run([exe, "ur5e", "40"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "UR5E OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_gomp_example_writes_reference_trajectory_files(tmp_path):
    """examples/gomp_example.cpp = [REF] examples/solver-example.cpp on the MI355X QPSolver;
    output_trajectory_ctrl.data holds D joint values per line, output_trajectory_xyz.data one "(x, y, z)" per line."""
    import osqp_solver_amd as M
    M.lib()
    libdir = os.path.join(ROOT, "osqp-solver_amd")
    out = tmp_path / "gomp_example"
    subprocess.run(["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "gomp_example.cpp"),
                    "-o", str(out), "-L" + libdir, "-lmi_osqp", "-pthread", "-Wl,-rpath," + libdir], check=True)
    r = subprocess.run([str(out), "40", "1", str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.splitlines()[0] == "kOptimal", r.stdout + r.stderr
    ctrl = (tmp_path / "output_trajectory_ctrl.data").read_text().strip().splitlines()
    xyz = (tmp_path / "output_trajectory_xyz.data").read_text().strip().splitlines()
    assert len(ctrl) == len(xyz) >= 4
    rows = [[float(v) for v in line.split(" ")] for line in ctrl]
    assert all(len(rw) == 6 for rw in rows)
    assert max(abs(v) for v in rows[0]) < 1e-3                     # starts at the zero pose
    import re
    assert all(re.fullmatch(r"\(-?[0-9.e+-]+, -?[0-9.e+-]+, -?[0-9.e+-]+\)", line) for line in xyz)
    assert "re-linearisations" in r.stdout


@pytest.mark.gpu
def test_batch_gomp_example_plans_every_trajectory(tmp_path):
    """examples/batch_gomp_example.cpp: the continuous planner on the reference's robot and scene as a user would call it
    (SQP step on the device), every trajectory planned."""
    import osqp_solver_amd as M
    M.lib()
    libdir = os.path.join(ROOT, "osqp-solver_amd")
    out = tmp_path / "batch_gomp_example"
    subprocess.run(["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "batch_gomp_example.cpp"),
                    "-o", str(out), "-L" + libdir, "-lmi_osqp", "-pthread", "-Wl,-rpath," + libdir], check=True)
    r = subprocess.run([str(out), "16", "40", "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "16 of 16 trajectories planned" in r.stdout and "Memory access fault" not in r.stdout + r.stderr, r.stdout + r.stderr
