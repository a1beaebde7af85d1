// developer probe (GPU box): HBM rate of MANY concurrent sequential streams, as iterate_kernel reads them:
// W workgroups x 8 waves, every wave walks its own contiguous stream with a 16-deep register ring,
// (a) 8 B per lane and load (512 B per wave-load, what the kernels issue today), (b) 16 B per lane (1 KiB per wave-load).
//   hipcc -O3 --offload-arch=gfx950 scripts/probes/stream_probe.hip -o /tmp/stream_probe && /tmp/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

template <int WIDE>
__global__ __launch_bounds__(512) void stream_kernel(const double *base, size_t stream_doubles, int steps, double *out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t sid = (size_t)blockIdx.x * 8 + wave;
  const double *p = base + sid * stream_doubles;
  rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(p), 0, (int)(stream_doubles * 8), 0x00020000);
  constexpr int PF = 16;
  double acc = 0.0;
  if (WIDE) {
    u32x4 ring[PF / 2];
#pragma unroll
    for (int s = 0; s < PF / 2; s++) ring[s] = __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, s * 1024, 0);
    for (int pos = PF / 2; pos < steps / 2 + PF / 2; pos += PF / 2) {
#pragma unroll
      for (int s = 0; s < PF / 2; s++) {
        acc += __hiloint2double((int)ring[s].y, (int)ring[s].x) + __hiloint2double((int)ring[s].w, (int)ring[s].z);
        ring[s] = __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, (pos + s) * 1024, 0);
      }
    }
  } else {
    u32x2 ring[PF];
#pragma unroll
    for (int s = 0; s < PF; s++) ring[s] = __builtin_amdgcn_raw_buffer_load_b64(r, lane * 8, s * 512, 0);
    for (int pos = PF; pos < steps + PF; pos += PF) {
#pragma unroll
      for (int s = 0; s < PF; s++) {
        acc += __hiloint2double((int)ring[s].y, (int)ring[s].x);
        ring[s] = __builtin_amdgcn_raw_buffer_load_b64(r, lane * 8, (pos + s) * 512, 0);
      }
    }
  }
  out[(size_t)blockIdx.x * 512 + threadIdx.x] = acc;
}

int main() {
  const int steps = 1568;                         // 100 352 slots of 8 B per stream (the dense-tail stream of config 3), 784 KiB
  const size_t stream_doubles = (size_t)steps * 64;
  for (int W : {256, 512, 1024, 2048}) {
    const size_t total = (size_t)W * 8 * stream_doubles;
    double *buf, *out;
    if (hipMalloc(&buf, total * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&out, (size_t)W * 512 * 8);
    hipMemset(buf, 0, total * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wide = 0; wide < 2; wide++) {
      float best = 1e9f;
      for (int rep = 0; rep < 5; rep++) {
        hipEventRecord(e0);
        if (wide) hipLaunchKernelGGL(stream_kernel<1>, dim3(W), dim3(512), 0, 0, buf, stream_doubles, steps, out);
        else hipLaunchKernelGGL(stream_kernel<0>, dim3(W), dim3(512), 0, 0, buf, stream_doubles, steps, out);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      printf("W=%4d workgroups (%5d streams of %zu KiB, %.2f GB): %2d B per lane: %.3f ms = %.2f TB/s\n", W, W * 8, stream_doubles * 8 / 1024,
             total * 8 / 1e9, wide ? 16 : 8, best, total * 8 / (best * 1e-3) / 1e12);
    }
    hipFree(buf); hipFree(out);
  }
  return 0;
}
