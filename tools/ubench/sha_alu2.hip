// Register-only SHA-256 compression variants: which formulation issues fastest on gfx950?
// V0 as shipped; V1 add3 -> two adds; V2 bitop3 -> classic two-input logic; V3 two independent messages per lane (ILP);
// V4 rotates as shift pairs merged by bitop3 (no v_alignbit)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ uint32_t rotr(uint32_t x, int n) { return __builtin_amdgcn_alignbit(x, x, n); }
template <int V> __device__ __forceinline__ uint32_t x3(uint32_t a, uint32_t b, uint32_t c) {
  if (V == 2) return a ^ b ^ c; return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
template <int V> __device__ __forceinline__ uint32_t chf(uint32_t e, uint32_t f, uint32_t g) {
  if (V == 2) return g ^ (e & (f ^ g)); return __builtin_amdgcn_bitop3_b32(e, f, g, 0xCA); }
template <int V> __device__ __forceinline__ uint32_t mjf(uint32_t a, uint32_t b, uint32_t c) {
  if (V == 2) return (a & b) | (c & (a | b)); return __builtin_amdgcn_bitop3_b32(a, b, c, 0xE8); }
template <int V> __device__ __forceinline__ uint32_t add3(uint32_t a, uint32_t b, uint32_t c) {
  if (V == 1) { uint32_t t; asm volatile("v_add_u32 %0, %1, %2" : "=v"(t) : "v"(a), "v"(b)); uint32_t r; asm volatile("v_add_u32 %0, %1, %2" : "=v"(r) : "v"(t), "v"(c)); return r; }
  return a + b + c; }
template <int V> __device__ __forceinline__ uint32_t S3(uint32_t x, int r1, int r2, int r3) {   // xor of three rotations
  if (V == 4) {
    uint32_t t = __builtin_amdgcn_bitop3_b32(x >> r1, x << (32 - r1), x >> r2, 0x96);
    uint32_t u = __builtin_amdgcn_bitop3_b32(t, x << (32 - r2), x >> r3, 0x96);
    return u ^ (x << (32 - r3));
  }
  return x3<V>(rotr(x, r1), rotr(x, r2), rotr(x, r3)); }
template <int V> __device__ __forceinline__ uint32_t s3(uint32_t x, int r1, int r2, int sh) {   // two rotations and a shift
  if (V == 4) {
    uint32_t t = __builtin_amdgcn_bitop3_b32(x >> r1, x << (32 - r1), x >> r2, 0x96);
    return __builtin_amdgcn_bitop3_b32(t, x << (32 - r2), x >> sh, 0x96);
  }
  return x3<V>(rotr(x, r1), rotr(x, r2), x >> sh); }
__constant__ uint32_t Kc[64];
template <int V> __device__ __forceinline__ void compress(uint32_t (&st)[8], uint32_t (&w)[16]) {
  constexpr uint32_t K[64] = {
      0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
      0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
      0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
      0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
      0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
      0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
      0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
      0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
  uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
#pragma unroll
  for (int i = 0; i < 64; i++) {
    uint32_t wi;
    if (i < 16) wi = w[i];
    else {
      uint32_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
      uint32_t s0 = s3<V>(w15, 7, 18, 3), s1 = s3<V>(w2, 17, 19, 10);
      wi = add3<V>(w[i & 15], s0, w[(i - 7) & 15]) + s1;
      w[i & 15] = wi;
    }
    uint32_t S1 = S3<V>(e, 6, 11, 25);
    uint32_t t1 = add3<V>(h, S1, chf<V>(e, f, g)) + (K[i] + wi);
    uint32_t S0 = S3<V>(a, 2, 13, 22);
    uint32_t mj = mjf<V>(a, b, c);
    h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = add3<V>(t1, S0, mj);
  }
  st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}
template <int V> __global__ __launch_bounds__(256) void k(uint32_t* out, int nblk, uint32_t seed) {
  uint32_t st[8] = {1, 2, 3, 4, 5, 6, 7, 8};
  uint32_t w0[16];
  for (int i = 0; i < 16; i++) w0[i] = seed * (threadIdx.x + i + 1);
  for (int b = 0; b < nblk; b++) {
    uint32_t w[16];
#pragma unroll
    for (int i = 0; i < 16; i++) w[i] = w0[i] ^ st[i & 7];
    compress<V>(st, w);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = st[0] ^ st[5];
}
__global__ __launch_bounds__(256) void k2(uint32_t* out, int nblk, uint32_t seed) {     // V3: two messages per lane
  uint32_t sa[8] = {1, 2, 3, 4, 5, 6, 7, 8}, sb[8] = {9, 8, 7, 6, 5, 4, 3, 2};
  uint32_t w0[16];
  for (int i = 0; i < 16; i++) w0[i] = seed * (threadIdx.x + i + 1);
  for (int b = 0; b < nblk; b++) {
    uint32_t wa[16], wb[16];
#pragma unroll
    for (int i = 0; i < 16; i++) { wa[i] = w0[i] ^ sa[i & 7]; wb[i] = w0[i] ^ sb[i & 7]; }
    compress<0>(sa, wa);
    compress<0>(sb, wb);
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = sa[0] ^ sb[5];
}
template <class KF> void run(const char* name, KF kf, int per_lane, uint32_t* out) {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int nblk = 1024;
  for (int w : {2, 4, 8}) {
    int blocks = p.multiProcessorCount * w;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kf, dim3(blocks), dim3(256), 0, 0, out, nblk, 7u); hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kf, dim3(blocks), dim3(256), 0, 0, out, nblk, 7u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double bytes = (double)blocks * 256 * nblk * 64 * per_lane;
    printf("%-28s waves/SIMD=%d: %.3f ms, %.0f GB/s equivalent, %.0f cycles per block per wave-slot\n", name, w, ms, bytes / ms / 1e6,
           ms * 1e-3 * p.clockRate * 1e3 / ((double)nblk * w * per_lane));
  }
}
int main() {
  uint32_t* out; hipMalloc(&out, 4 * 256 * 8 * 256 * 4);
  run("V0 shipped", k<0>, 1, out);
  run("V1 add3 -> 2 adds", k<1>, 1, out);
  run("V2 two-input logic", k<2>, 1, out);
  run("V3 two messages per lane", k2, 2, out);
  run("V4 shifts + bitop3 rotates", k<4>, 1, out);
  return 0;
}
