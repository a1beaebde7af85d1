// developer probe (runs on the GPU box): lane layout and issue rate of v_mfma_f64_16x16x4_f64 on gfx950, next to v_fma_f64.
//   hipcc -O3 --offload-arch=gfx950 scripts/probes/mfma_f64_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
typedef double v4d __attribute__((ext_vector_type(4)));

__global__ void layout_kernel(const double *A /*16x4 row-major*/, const double *B /*4x16 row-major*/, double *D /*64 lanes x 4*/) {
  const int l = threadIdx.x;
  const double a = A[(l & 15) * 4 + (l >> 4)];      // A[i = l&15][k = l>>4]
  const double b = B[(l >> 4) * 16 + (l & 15)];     // B[k = l>>4][j = l&15]
  v4d c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; r++) D[l * 4 + r] = c[r];
}

template <int NACC>
__global__ void rate_kernel(double *out, unsigned long long *cyc, int iters) {
  v4d acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = v4d{0, 0, 0, 0};
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC>
__global__ void fma_kernel(double *out, unsigned long long *cyc, int iters) {
  double acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = threadIdx.x * 1e-3 + i;
  double a = 1.0 + threadIdx.x * 1e-9, b = 1e-9;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = fma(acc[i], a, b);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  // ---- layout
  std::vector<double> A(64), B(64), D(256);
  for (int i = 0; i < 16; i++) for (int k = 0; k < 4; k++) A[i * 4 + k] = 1 + i * 0.5 + k * 7;
  for (int k = 0; k < 4; k++) for (int j = 0; j < 16; j++) B[k * 16 + j] = 2 + k * 3 - j * 0.25 + (k == j ? 11 : 0);
  double *dA, *dB, *dD;
  hipMalloc(&dA, 512); hipMalloc(&dB, 512); hipMalloc(&dD, 2048);
  hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(layout_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost);
  auto ref = [&](int i, int j) { double s = 0; for (int k = 0; k < 4; k++) s += A[i * 4 + k] * B[k * 16 + j]; return s; };
  int okA = 0, okB = 0;       // A: row = (lane>>4) + 4*reg ; B: row = (lane>>4)*4 + reg
  for (int l = 0; l < 64; l++) for (int r = 0; r < 4; r++) {
    if (std::fabs(D[l * 4 + r] - ref((l >> 4) + 4 * r, l & 15)) < 1e-9) okA++;
    if (std::fabs(D[l * 4 + r] - ref((l >> 4) * 4 + r, l & 15)) < 1e-9) okB++;
  }
  printf("layout: row=(lane>>4)+4*reg matches %d/256 ; row=(lane>>4)*4+reg matches %d/256 (col = lane&15)\n", okA, okB);
  // ---- rates
  double *out; unsigned long long *cyc;
  const int iters = 20000;
  hipMalloc(&out, 8 * 1024 * 2048); hipMalloc(&cyc, 8 * 2048);
  std::vector<unsigned long long> hc(2048);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](auto kern, const char *name, int nacc, int blocks, int threads, double flop_per_inst_per_wave) {
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, cyc, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(hc.data(), cyc, 8 * blocks, hipMemcpyDeviceToHost);
    double c = 0; for (int b = 0; b < blocks; b++) c += (double)hc[b];
    c /= blocks;
    const double insts = (double)iters * nacc;
    const double waves = (double)blocks * threads / 64;
    printf("%-10s acc=%d blocks=%4d threads=%4d : %.1f cycles per instruction per wave, %.2f TFLOP/s chip, %.3f ms\n", name, nacc, blocks,
           threads, c / insts, waves * insts * flop_per_inst_per_wave / (ms * 1e-3) / 1e12, ms);
  };
  for (int threads : {256, 512, 1024}) {
    run(rate_kernel<1>, "mfma_f64", 1, 256, threads, 2048);
    run(rate_kernel<2>, "mfma_f64", 2, 256, threads, 2048);
    run(rate_kernel<4>, "mfma_f64", 4, 256, threads, 2048);
    run(rate_kernel<8>, "mfma_f64", 8, 256, threads, 2048);
  }
  run(rate_kernel<4>, "mfma_f64", 4, 1, 256, 2048);
  run(rate_kernel<4>, "mfma_f64", 4, 1, 64, 2048);
  for (int threads : {256, 512, 1024}) {
    run(fma_kernel<4>, "fma_f64", 4, 256, threads, 128);
    run(fma_kernel<8>, "fma_f64", 8, 256, threads, 128);
    run(fma_kernel<16>, "fma_f64", 16, 256, threads, 128);
  }
  run(fma_kernel<16>, "fma_f64", 16, 1, 64, 128);
  return 0;
}
