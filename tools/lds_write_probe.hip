// LDS store path of gfx950: bytes per clock and CU for the store shapes a Legendre loader could use (16 waves per CU, stores only).
//   b128      ds_write_b128, a lane's 16 bytes contiguous (what the loaders do)
//   b64x2     two ds_write_b64, each lane-contiguous (lane l -> base + 8 l), the halves 512 B apart
//   w2st64    ONE ds_write2st64_b64: the same two stores in one instruction
//   w2        ds_write2_b64 offset1 = offset0 + 1: the b128 footprint as a paired store
//   addtid    four ds_write_addtid_b32 (address = M0 + offset + 4 lane): the 16 bytes of a lane as four dword planes
// hipcc --offload-arch=gfx950 -O3 -o tools/lds_write_probe tools/lds_write_probe.hip && ./tools/lds_write_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __attribute__((amdgpu_flat_work_group_size(256, 256), amdgpu_waves_per_eu(4, 4))) void k(double *out, int iters) {
  __shared__ __attribute__((aligned(16))) double lds[4 * 1024 + 64];
  const unsigned l = threadIdx.x & 63, w = threadIdx.x >> 6;
  double x = threadIdx.x, y = 2.0 * threadIdx.x;
  typedef double d2_t __attribute__((ext_vector_type(2)));
  d2_t xy = {x, y};
  unsigned a128 = (w * 1024 + l * 2) * 8, a64 = (w * 1024 + l) * 8;
  const unsigned xl = threadIdx.x, xh = threadIdx.x * 3u, yl = threadIdx.x * 5u, yh = threadIdx.x * 7u;
  const unsigned m0v = __builtin_amdgcn_readfirstlane(w * 8192u);
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
      if (MODE == 0) asm volatile("ds_write_b128 %0, %1" ::"v"(a128), "v"(xy) : "memory");
      if (MODE == 1) asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %0, %2 offset:512" ::"v"(a64), "v"(x), "v"(y) : "memory");
      if (MODE == 2) asm volatile("ds_write2st64_b64 %0, %1, %2 offset0:0 offset1:1" ::"v"(a64), "v"(x), "v"(y) : "memory");
      if (MODE == 3) asm volatile("ds_write2_b64 %0, %1, %2 offset0:0 offset1:1" ::"v"(a128), "v"(x), "v"(y) : "memory");
      if (MODE == 4)  // four dword planes of 256 B: address = M0 + offset + 4 lane, no address register
        asm volatile("s_mov_b32 m0, %4\n\tds_write_addtid_b32 %0\n\tds_write_addtid_b32 %1 offset:256\n\tds_write_addtid_b32 %2 offset:512\n\tds_write_addtid_b32 %3 offset:768"
                     ::"v"(xl), "v"(xh), "v"(yl), "v"(yh), "s"(m0v) : "memory", "m0");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  __syncthreads();
  out[blockIdx.x * 256 + threadIdx.x] = lds[threadIdx.x];
}
int main() {
  double *buf;
  (void)hipMalloc(&buf, 4096 * 256 * 8);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int nblk = 256 * 4 * 4, iters = 20000;
  const char *names[] = {"ds_write_b128 (16 B per lane, contiguous)", "2 x ds_write_b64 (lane-contiguous halves)", "ds_write2st64_b64 (the same, one instruction)",
                         "ds_write2_b64 (b128 footprint)", "4 x ds_write_addtid_b32 (dword planes, M0 base)"};
  for (int m = 0; m < 5; m++) {
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
      (void)hipEventRecord(e0, 0);
      if (m == 0) hipLaunchKernelGGL(k<0>, dim3(nblk), dim3(256), 0, 0, buf, iters);
      if (m == 1) hipLaunchKernelGGL(k<1>, dim3(nblk), dim3(256), 0, 0, buf, iters);
      if (m == 2) hipLaunchKernelGGL(k<2>, dim3(nblk), dim3(256), 0, 0, buf, iters);
      if (m == 3) hipLaunchKernelGGL(k<3>, dim3(nblk), dim3(256), 0, 0, buf, iters);
      if (m == 4) hipLaunchKernelGGL(k<4>, dim3(nblk), dim3(256), 0, 0, buf, iters);
      (void)hipEventRecord(e1, 0);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    const double bytes = (double)nblk * 256 * 16.0 * 8 * iters;
    // per CU and clock at a nominal 2.4 GHz: 16 resident workgroups of 4 waves = 4 per CU at a time
    printf("%-48s %8.2f ms  %7.1f B/clk/CU (at 2.4 GHz)\n", names[m], best, bytes / (best * 1e-3) / 256 / 2.4e9);
  }
  return 0;
}
