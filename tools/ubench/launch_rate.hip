// launch_rate.hip — how many dependent kernel launches per second does the chip take with S streams in flight?
// (the per-batch floor of an "empty" pipeline.)   hipcc --offload-arch=gfx950 -O2 launch_rate.hip -o launch_rate
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 12345) *p = 1; }
__global__ void touch_kernel(int* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1; }
int main(int argc, char** argv) {
  const int K = argc > 1 ? atoi(argv[1]) : 2000;
  const size_t BUF_INTS = (size_t)32 * 65536;      // one 65 536-int window per stream, S <= 32
  int* buf; if (hipMalloc(&buf, BUF_INTS * sizeof(int)) != hipSuccess) return 1;
  hipMemset(buf, 0, BUF_INTS * sizeof(int));
  setvbuf(stdout, nullptr, _IOLBF, 0);
  for (int S : {1, 4, 8, 20}) {
    if ((size_t)S * 65536 > BUF_INTS) return 2;
    std::vector<hipStream_t> st(S);
    for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int variant = 0; variant < 3; variant++) {
      const int blocks = variant == 0 ? 1 : 1024;
      for (int rep = 0; rep < 2; rep++) {
        hipDeviceSynchronize();
        auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < K; k++)
          for (int s = 0; s < S; s++) {
            if (variant < 2) hipLaunchKernelGGL(empty_kernel, dim3(blocks), dim3(64), 0, st[s], (int*)nullptr);
            else hipLaunchKernelGGL(touch_kernel, dim3(blocks), dim3(64), 0, st[s], buf + s * 65536, 65536);
          }
        auto t1 = std::chrono::steady_clock::now();
        hipDeviceSynchronize();
        auto t2 = std::chrono::steady_clock::now();
        if (rep == 1) {
          const double sub = std::chrono::duration<double, std::micro>(t1 - t0).count(), tot = std::chrono::duration<double, std::micro>(t2 - t0).count();
          printf("streams %2d  %s  host submit %.2f us/launch   end-to-end %.2f us/launch (all streams: one launch per %.2f us)\n", S,
                 variant == 0 ? "empty 1 block    " : (variant == 1 ? "empty 1024 blocks" : "touch 1024 blocks"), sub / (K * S), tot / K, tot / (K * S));
        }
      }
    }
    for (auto& s : st) hipStreamDestroy(s);
  }
  return 0;
}
