// coissue.hip — do scalar and vector instructions of DIFFERENT waves on one SIMD issue in the same cycle on gfx950?
// One 16-wave workgroup per CU.  Every wave reads the SIMD it landed on (HW_REG_HW_ID) and takes the next role on THAT SIMD from
// a counter in LDS: the first nV arrivals run a VALU stream, the next nS a SALU stream, the rest leave — so every SIMD holds
// exactly nV vector and nS scalar waves, however the dispatcher dealt the waves.
//   2 V + 1 S ~ 2 V alone -> scalar instructions of another wave ride along;  ~ 3 V -> a SIMD issues one instruction per slot.
//   hipcc --offload-arch=gfx950 -O3 -o coissue tools/ubench/coissue.hip && ./coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 20000
#define REP16(x) x x x x x x x x x x x x x x x x
__global__ __launch_bounds__(1024) void k(uint32_t* out, uint32_t* roles_seen, int nV, int nS, uint32_t x) {
  __shared__ uint32_t cnt[4];
  if (threadIdx.x < 4) cnt[threadIdx.x] = 0;
  __syncthreads();
  uint32_t hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  const uint32_t simd = (hwid >> 4) & 3;
  uint32_t idx = 0;
  if ((threadIdx.x & 63) == 0) idx = atomicAdd(&cnt[simd], 1u);
  idx = __builtin_amdgcn_readfirstlane(idx);
  __syncthreads();
  if (blockIdx.x == 0 && threadIdx.x < 4) roles_seen[threadIdx.x] = cnt[threadIdx.x];
  const bool valu = (int)idx < nV, salu = !valu && (int)idx < nV + nS;
  uint32_t a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3;
  if (valu) {
    for (int i = 0; i < ITER; i++)
      asm volatile(REP16("v_add_u32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_xor_b32 %3, %3, %4\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(x));
  }
  if (salu) {
    uint32_t s0 = x, s1 = x + 1, s2 = x + 2, s3 = x + 3;
    for (int i = 0; i < ITER; i++)
      asm volatile(REP16("s_add_u32 %0, %0, %4\n s_xor_b32 %1, %1, %4\n s_add_u32 %2, %2, %4\n s_xor_b32 %3, %3, %4\n") : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "s"(x) : "scc");
    a += s0 + s1 + s2 + s3;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}
int main() {
  uint32_t *out, *seen;
  (void)hipMalloc(&out, 256 * 1024 * 4);
  (void)hipHostMalloc(&seen, 16, 0);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int cfg[][2] = {{1, 0}, {2, 0}, {3, 0}, {4, 0}, {0, 1}, {0, 2}, {1, 1}, {2, 1}, {2, 2}, {3, 1}, {1, 2}};
  for (auto& c : cfg) {
    hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, 0, out, seen, c[0], c[1], 3u);   // warm
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, 0, out, seen, c[0], c[1], 3u);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double inst = (double)ITER * 64;   // per wave
    printf("%d VALU + %d SALU waves per SIMD   %.3f ms  -> %.2f cycles per instruction of one wave, %.2f per instruction issued on the SIMD (2.4 GHz nominal)   [waves per SIMD in block 0: %u %u %u %u]\n",
           c[0], c[1], ms, ms * 1e-3 * 2.4e9 / inst, ms * 1e-3 * 2.4e9 / inst / (c[0] + c[1]), seen[0], seen[1], seen[2], seen[3]);
  }
  return 0;
}
