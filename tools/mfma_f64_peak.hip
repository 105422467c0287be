// Peak micro-kernel for the roofline denominator: back-to-back independent v_mfma_f64_16x16x4_f64
// (and v_mfma_f32_16x16x4_f32) from registers, no memory traffic.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak tools/mfma_f64_peak.hip && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __attribute__((amdgpu_flat_work_group_size(256, 256), amdgpu_waves_per_eu(2, 2))) void k_f64(double *out, int iters) {
  v4d acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = (v4d){0, 0, 0, 0};
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  for (int it = 0; it < iters; it++)
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  double s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ __attribute__((amdgpu_flat_work_group_size(256, 256), amdgpu_waves_per_eu(2, 2))) void k_f32(float *out, int iters) {
  v4f acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = (v4f){0, 0, 0, 0};
  float a = 1.0f + threadIdx.x * 1e-6f, b = 1.0f - threadIdx.x * 1e-6f;
  for (int it = 0; it < iters; it++)
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  float s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// the same with operands that differ per lane and per instruction (random mantissas, both signs): the matrix cores'
// power draw, and with it the sustained clock, depends on the data
template <int NACC>
__global__ __attribute__((amdgpu_flat_work_group_size(256, 256), amdgpu_waves_per_eu(2, 2))) void k_f64_random(double *out, int iters) {
  v4d acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = (v4d){0, 0, 0, 0};
  double a[4], b[4];
  unsigned h = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
  for (int i = 0; i < 4; i++) {
    h ^= h >> 15, h *= 2246822519u, h ^= h >> 13;
    a[i] = (double)(int)h * (1.0 / 2147483648.0) * (1.0 + 1e-13 * (h & 1023));
    h ^= h >> 15, h *= 3266489917u, h ^= h >> 16;
    b[i] = (double)(int)h * (1.0 / 2147483648.0) * (1.0 + 1e-13 * (h & 1023));
  }
  for (int it = 0; it < iters; it++)
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
  double s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  const int nblk = 256 * 8, nthr = 256, iters = 4000, NACC = 16;
  void *buf;
  hipMalloc(&buf, (size_t)nblk * nthr * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int pass = 0; pass < 2; pass++) {
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(e0, 0);
      if (pass == 0)
        hipLaunchKernelGGL(k_f64<NACC>, dim3(nblk), dim3(nthr), 0, 0, (double *)buf, iters);
      else
        hipLaunchKernelGGL(k_f32<NACC>, dim3(nblk), dim3(nthr), 0, 0, (float *)buf, iters);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      double flops = (double)nblk * (nthr / 64) * (double)iters * NACC * 2048.0;
      printf("%s 16x16x4 MFMA: %.1f TFLOP/s (%.2f ms)\n", pass == 0 ? "fp64" : "fp32", flops / ms / 1e9, ms);
    }
  }
  // sustained, Legendre-launch-sized runs (~100 ms): uniform against random operand values
  for (int pass = 0; pass < 2; pass++)
    for (int rep = 0; rep < 3; rep++) {
      const int long_iters = 60000;
      hipEventRecord(e0, 0);
      if (pass == 0)
        hipLaunchKernelGGL(k_f64<NACC>, dim3(nblk), dim3(nthr), 0, 0, (double *)buf, long_iters);
      else
        hipLaunchKernelGGL(k_f64_random<NACC>, dim3(nblk), dim3(nthr), 0, 0, (double *)buf, long_iters);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      double flops = (double)nblk * (nthr / 64) * (double)long_iters * NACC * 2048.0;
      printf("fp64 sustained, %s operands: %.1f TFLOP/s (%.1f ms)\n", pass == 0 ? "uniform" : "random ", flops / ms / 1e9, ms);
    }
  return 0;
}
