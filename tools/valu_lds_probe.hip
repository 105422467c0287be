// Do fp64 vector arithmetic and LDS traffic of OTHER waves overlap on a CU?  (The FFT kernels: 16 waves per CU, per radix-8
// pass and wave ~170 vector instructions, 8 ds_read_b128 and 8 ds_write_b128.)  A workgroup of 512 threads (8 waves, 64 KiB of
// LDS: two workgroups per CU) loops over "passes" of NV independent v_fma_f64, NR ds_read_b128 and NW ds_write_b128;
// time per pass for the arithmetic alone, the LDS traffic alone and both.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_lds_probe tools/valu_lds_probe.hip && /tmp/valu_lds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d2 __attribute__((ext_vector_type(2)));

template <int NV, int NR, int NW, int SYNC>
__global__ __attribute__((amdgpu_flat_work_group_size(512, 512), amdgpu_waves_per_eu(4, 4))) void k_probe(double *out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char lds_raw[];
  d2 *lds = (d2 *)lds_raw;
  const int tid = threadIdx.x;
  double acc[8], m = 1.0 + tid * 1e-12, c = 1e-9 * tid;
  d2 v[8];
#pragma unroll
  for (int i = 0; i < 8; i++) acc[i] = tid + i, v[i] = (d2){(double)tid, (double)i};
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < NR; r++) v[r & 7] += lds[(tid + 512 * r) & 4095];
#pragma unroll
    for (int k = 0; k < NV / 8; k++)
#pragma unroll
      for (int i = 0; i < 8; i++) acc[i] = __builtin_fma(acc[i], m, c);
#pragma unroll
    for (int w = 0; w < NW; w++) lds[(tid + 512 * w) & 4095] = (d2){acc[w & 7], v[w & 7].x};
    if (SYNC) __syncthreads();
  }
  double s = 0;
  for (int i = 0; i < 8; i++) s += acc[i] + v[i].x + v[i].y;
  out[blockIdx.x * blockDim.x + tid] = s;
}

template <int NV, int NR, int NW, int SYNC>
static double run(double *buf, int iters) {
  hipFuncSetAttribute((const void *)k_probe<NV, NR, NW, SYNC>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  const int nblk = 256 * 2 * 4;  // 4 rounds of two workgroups per CU
  k_probe<NV, NR, NW, SYNC><<<nblk, 512, 65536>>>(buf, 10);
  hipEventRecord(e0);
  k_probe<NV, NR, NW, SYNC><<<nblk, 512, 65536>>>(buf, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  // cycles per pass and CU-slot at 2.4 GHz: 4 rounds
  return ms * 1e-3 * 2.4e9 / 4.0 / iters;
}

int main() {
  double *buf;
  hipMalloc(&buf, (size_t)256 * 8 * 512 * 8);
  const int it = 2000;
  printf("cycles (2.4 GHz) per pass of a CU holding 16 waves; pass = NV v_fma_f64 + NR ds_read_b128 + NW ds_write_b128 per wave\n");
  printf("NV=160 alone            : %8.0f\n", run<160, 0, 0, 0>(buf, it));
  printf("NR=8 alone              : %8.0f\n", run<0, 8, 0, 0>(buf, it));
  printf("NW=8 alone              : %8.0f\n", run<0, 0, 8, 0>(buf, it));
  printf("NR=8 NW=8               : %8.0f\n", run<0, 8, 8, 0>(buf, it));
  printf("NV=160 NR=8             : %8.0f\n", run<160, 8, 0, 0>(buf, it));
  printf("NV=160 NW=8             : %8.0f\n", run<160, 0, 8, 0>(buf, it));
  printf("NV=160 NR=8 NW=8        : %8.0f\n", run<160, 8, 8, 0>(buf, it));
  printf("NV=160 NR=8 NW=8 barrier: %8.0f\n", run<160, 8, 8, 1>(buf, it));
  printf("NV=80 NR=8 NW=8         : %8.0f\n", run<80, 8, 8, 0>(buf, it));
  printf("NV=160 NR=8 NW=4        : %8.0f\n", run<160, 8, 4, 0>(buf, it));
  printf("NV=320 NR=16 NW=16      : %8.0f\n", run<320, 16, 16, 0>(buf, it));
  return 0;
}
