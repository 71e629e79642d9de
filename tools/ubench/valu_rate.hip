// Micro-benchmark: sustained issue rate of the integer VALU instructions the SHA-256 and RSA kernels are made of.
// Each kernel runs ITER x 32 independent instructions per wave; 8 waves per SIMD on every SIMD of the chip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 4096
#define REP8(x) x x x x x x x x
#define BODY(INS) \
  for (int i = 0; i < ITER; i++) { asm volatile(REP8(INS) REP8(INS) REP8(INS) REP8(INS) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e2), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y)); }
#define KERNEL(name, INS) \
__global__ __launch_bounds__(256) void name(uint32_t* out, uint32_t x, uint32_t y) { \
  uint32_t a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3, e2 = a + 4, f = a + 5, g = a + 6, h = a + 7; \
  BODY(INS) \
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e2 + f + g + h; }
// 8 independent destinations rotate so there is no dependent-issue stall
KERNEL(k_add,      "v_add_u32 %0, %8, %0\n v_add_u32 %1, %8, %1\n v_add_u32 %2, %8, %2\n v_add_u32 %3, %8, %3\n")
KERNEL(k_xor,      "v_xor_b32 %0, %8, %0\n v_xor_b32 %1, %8, %1\n v_xor_b32 %2, %8, %2\n v_xor_b32 %3, %8, %3\n")
KERNEL(k_add3,     "v_add3_u32 %0, %8, %0, %9\n v_add3_u32 %1, %8, %1, %9\n v_add3_u32 %2, %8, %2, %9\n v_add3_u32 %3, %8, %3, %9\n")
KERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %0, 7\n v_alignbit_b32 %1, %1, %1, 7\n v_alignbit_b32 %2, %2, %2, 7\n v_alignbit_b32 %3, %3, %3, 7\n")
KERNEL(k_bitop3,   "v_bitop3_b32 %0, %8, %0, %9 bitop3:0x96\n v_bitop3_b32 %1, %8, %1, %9 bitop3:0x96\n v_bitop3_b32 %2, %8, %2, %9 bitop3:0x96\n v_bitop3_b32 %3, %8, %3, %9 bitop3:0x96\n")
KERNEL(k_perm,     "v_perm_b32 %0, %8, %0, %9\n v_perm_b32 %1, %8, %1, %9\n v_perm_b32 %2, %8, %2, %9\n v_perm_b32 %3, %8, %3, %9\n")
KERNEL(k_lshr,     "v_lshrrev_b32 %0, 3, %0\n v_lshrrev_b32 %1, 3, %1\n v_lshrrev_b32 %2, 3, %2\n v_lshrrev_b32 %3, 3, %3\n")
KERNEL(k_mullo,    "v_mul_lo_u32 %0, %8, %0\n v_mul_lo_u32 %1, %8, %1\n v_mul_lo_u32 %2, %8, %2\n v_mul_lo_u32 %3, %8, %3\n")
KERNEL(k_bfi,      "v_bfi_b32 %0, %8, %0, %9\n v_bfi_b32 %1, %8, %1, %9\n v_bfi_b32 %2, %8, %2, %9\n v_bfi_b32 %3, %8, %3, %9\n")
KERNEL(k_mad24,    "v_mad_u32_u24 %0, %8, %0, %9\n v_mad_u32_u24 %1, %8, %1, %9\n v_mad_u32_u24 %2, %8, %2, %9\n v_mad_u32_u24 %3, %8, %3, %9\n")
__global__ __launch_bounds__(256) void k_mad64(uint32_t* out, uint32_t x, uint32_t y) {
  uint64_t a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3;
  for (int i = 0; i < ITER; i++) {
    asm volatile(REP8("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n")
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x), "v"(y) : "vcc");
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a + b + c + d);
}
template <class K> void run(const char* name, K k, int per_iter, uint32_t* out, int waves_per_simd) {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  int cus = p.multiProcessorCount;
  int blocks = cus * waves_per_simd;          // 256-thread blocks = 4 waves: one per SIMD
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 3u, 5u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 3u, 5u);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double wave_instr_per_simd = (double)ITER * per_iter * waves_per_simd;
  double cyc = ms * 1e-3 * p.clockRate * 1e3;   // clockRate in kHz
  printf("%-10s waves/SIMD=%d  %.3f ms  -> %.2f cycles per wave-instruction per SIMD (at %d MHz nominal), %.1f T lane-ops/s chip\n",
         name, waves_per_simd, ms, cyc / wave_instr_per_simd, p.clockRate / 1000,
         wave_instr_per_simd * 64 * 4 * cus / (ms * 1e-3) / 1e12);
}
int main() {
  uint32_t* out; hipMalloc(&out, 256 * 4 * 256 * 64);
  for (int w : {1, 2, 4, 8}) {
    run("add_u32", k_add, 128, out, w); run("xor_b32", k_xor, 128, out, w); run("add3_u32", k_add3, 128, out, w);
    run("alignbit", k_alignbit, 128, out, w); run("bitop3", k_bitop3, 128, out, w); run("perm_b32", k_perm, 128, out, w);
    run("lshrrev", k_lshr, 128, out, w); run("bfi_b32", k_bfi, 128, out, w); run("mad_u32_u24", k_mad24, 128, out, w);
    run("mul_lo_u32", k_mullo, 128, out, w); run("mad_u64_u32", k_mad64, 128, out, w);
    printf("\n");
  }
  return 0;
}
