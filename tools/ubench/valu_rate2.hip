#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 4096
#define REP8(x) x x x x x x x x
__global__ __launch_bounds__(256) void k_lshr64(uint32_t* out, uint32_t x) {
  uint64_t a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3;
  for (int i = 0; i < ITER; i++)
    asm volatile(REP8("v_lshrrev_b64 %0, 7, %0\n v_lshrrev_b64 %1, 7, %1\n v_lshrrev_b64 %2, 7, %2\n v_lshrrev_b64 %3, 7, %3\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a + b + c + d);
}
__global__ __launch_bounds__(256) void k_lshladd(uint32_t* out, uint32_t x, uint32_t y) {
  uint32_t a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3;
  for (int i = 0; i < ITER; i++)
    asm volatile(REP8("v_lshl_add_u32 %0, %0, 3, %4\n v_lshl_add_u32 %1, %1, 3, %4\n v_lshl_add_u32 %2, %2, 3, %4\n v_lshl_add_u32 %3, %3, 3, %4\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x));
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}
__global__ __launch_bounds__(256) void k_xad(uint32_t* out, uint32_t x, uint32_t y) {
  uint32_t a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3;
  for (int i = 0; i < ITER; i++)
    asm volatile(REP8("v_xad_u32 %0, %0, %4, %5\n v_xad_u32 %1, %1, %4, %5\n v_xad_u32 %2, %2, %4, %5\n v_xad_u32 %3, %3, %4, %5\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x), "v"(y));
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}
__global__ __launch_bounds__(256) void k_alignbit_s(uint32_t* out, uint32_t x, uint32_t y) {   // alignbit with a scalar third operand / two regs
  uint32_t a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3;
  for (int i = 0; i < ITER; i++)
    asm volatile(REP8("v_alignbit_b32 %0, %0, %4, 7\n v_alignbit_b32 %1, %1, %4, 7\n v_alignbit_b32 %2, %2, %4, 7\n v_alignbit_b32 %3, %3, %4, 7\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x));
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}
__global__ __launch_bounds__(256) void k_readlane(uint32_t* out, uint32_t x, uint32_t y) {
  uint32_t a = threadIdx.x; uint32_t s0, s1, s2, s3, acc = 0;
  for (int i = 0; i < ITER; i++) {
    asm volatile(REP8("v_readlane_b32 %0, %4, 3\n v_readlane_b32 %1, %4, 5\n v_readlane_b32 %2, %4, 7\n v_readlane_b32 %3, %4, 9\n") : "=s"(s0), "=s"(s1), "=s"(s2), "=s"(s3) : "v"(a));
    acc += s0 + s1 + s2 + s3;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_dpp(uint32_t* out, uint32_t x, uint32_t y) {
  uint32_t a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3;
  for (int i = 0; i < ITER; i++)
    asm volatile(REP8("v_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 wave_shl:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 wave_shl:1 row_mask:0xf bank_mask:0xf\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}
__global__ __launch_bounds__(256) void k_lshladd64(uint32_t* out, uint32_t x, uint32_t y) {
  uint64_t a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3, z = x;
  for (int i = 0; i < ITER; i++)
    asm volatile(REP8("v_lshl_add_u64 %0, %0, 0, %4\n v_lshl_add_u64 %1, %1, 0, %4\n v_lshl_add_u64 %2, %2, 0, %4\n v_lshl_add_u64 %3, %3, 0, %4\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(z));
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a + b + c + d);
}
template <class K, class... A> void run(const char* name, K k, uint32_t* out, int w, A... args) {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  int blocks = p.multiProcessorCount * w;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, args...); hipDeviceSynchronize();
  hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, args...); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-14s waves/SIMD=%d %.2f cycles per wave-instruction per SIMD\n", name, w, ms * 1e-3 * p.clockRate * 1e3 / ((double)ITER * 32 * w));
}
int main() {
  uint32_t* out; hipMalloc(&out, 256 * 4 * 256 * 64);
  for (int w : {1, 4}) {
    run("lshrrev_b64", k_lshr64, out, w, 3u); run("lshl_add_u32", k_lshladd, out, w, 3u, 5u); run("xad_u32", k_xad, out, w, 3u, 5u);
    run("alignbit(x,y)", k_alignbit_s, out, w, 3u, 5u); run("readlane", k_readlane, out, w, 3u, 5u); run("mov_dpp wshl", k_dpp, out, w, 3u, 5u);
    run("lshl_add_u64", k_lshladd64, out, w, 3u, 5u);
  }
}
