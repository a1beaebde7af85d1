#!/bin/bash
# GPU box: the reference's example scenario (one UR5e trajectory, sequential SQP driver) on the GPU QPSolver and on the
# oracle backend.   scripts/example_bench.sh [waypoints ...]
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
g++ -std=c++17 -O2 -Iinclude tests/cpp/gomp_parity.cpp -o gpurun_out/gomp_parity -Losqp-solver_amd -lmi_osqp -Loracle/_build -loracle_osqp \
    -fopenmp -pthread -Wl,-rpath,$PWD/osqp-solver_amd -Wl,-rpath,$PWD/oracle/_build
for W in ${@:-52 202 802}; do timeout -k 10 900 gpurun_out/gomp_parity example $W 2>&1 | tee -a gpurun_out/example_bench.txt; done
