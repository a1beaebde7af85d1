#!/bin/bash
# GPU box: the batched GOMP driver benchmark of tests/cpp/gomp_parity.cpp (256 trajectories, D = 7, W = 100), three times
# plain and once with the per-entry-point wall times of the C-ABI (MI_OSQP_DEBUG_TIMING).   scripts/gomp_bench.sh [outfile]
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
g++ -std=c++17 -O2 -Iinclude tests/cpp/gomp_parity.cpp -o gpurun_out/gomp_parity -Losqp-solver_amd -lmi_osqp -Loracle/_build -loracle_osqp \
    -fopenmp -pthread -Wl,-rpath,$PWD/osqp-solver_amd -Wl,-rpath,$PWD/oracle/_build
out="${1:-gpurun_out/gomp_bench.txt}"
: > "$out"
for i in 1 2 3; do timeout -k 10 300 gpurun_out/gomp_parity bench 2>&1 | tee -a "$out"; done
timeout -k 10 600 gpurun_out/gomp_parity obstbench 2>&1 | tee -a "$out"
MI_OSQP_DEBUG_TIMING=1 timeout -k 10 300 gpurun_out/gomp_parity bench 2>&1 | grep -v "setup: ordering\|^\[mi_osqp\] setup" | tee -a "$out"
