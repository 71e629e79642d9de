// Register-only SHA-256 compression rate (no memory traffic): the ALU ceiling of the compression as compiled.
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../zkemail.rs_amd/csrc/sha256.hip.h"
__global__ __launch_bounds__(256) void k(uint32_t* out, int nblk, uint32_t seed) {
  uint32_t st[8] = {1, 2, 3, 4, 5, 6, 7, 8};
  uint32_t w0[16];
  for (int i = 0; i < 16; i++) w0[i] = seed * (threadIdx.x + i + 1);
  for (int b = 0; b < nblk; b++) {
    uint32_t w[16];
#pragma unroll
    for (int i = 0; i < 16; i++) w[i] = w0[i] ^ st[i & 7];
    zke::sha256_compress(st, w);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = st[0] ^ st[5];
}
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  uint32_t* out; hipMalloc(&out, 4 * 256 * 8 * 256 * 4);
  const int nblk = 2048;
  for (int w : {1, 2, 4, 8}) {
    int blocks = p.multiProcessorCount * w;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, nblk, 7u); hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, nblk, 7u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double bytes = (double)blocks * 256 * nblk * 64;
    double cyc_per_block_per_simd = ms * 1e-3 * p.clockRate * 1e3 / ((double)nblk * w);
    printf("waves/SIMD=%d: %.3f ms, %.0f GB/s equivalent, %.0f cycles per block per wave-slot (nominal clock)\n", w, ms, bytes / ms / 1e6, cyc_per_block_per_simd);
  }
}
