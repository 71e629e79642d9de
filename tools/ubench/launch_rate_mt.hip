// launch_rate_mt.hip — do kernel launches from several host threads add up?  T threads, each with its own S/T streams,
// launch K empty kernels per stream; aggregate launches per second.   hipcc --offload-arch=gfx950 -O2 -pthread launch_rate_mt.hip -o launch_rate_mt
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 12345) *p = 1; }
int main(int argc, char** argv) {
  const int K = argc > 1 ? atoi(argv[1]) : 3000;
  setvbuf(stdout, nullptr, _IOLBF, 0);
  for (int T : {1, 2, 4, 8}) {
    const int S = 20;
    std::vector<hipStream_t> st(S);
    for (auto& s : st) (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (auto& s : st) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s, (int*)nullptr);
    (void)hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++)
      th.emplace_back([&, t] {
        (void)hipSetDevice(0);
        for (int k = 0; k < K; k++)
          for (int s = t; s < S; s += T) hipLaunchKernelGGL(empty_kernel, dim3(1024), dim3(64), 0, st[s], (int*)nullptr);
      });
    for (auto& x : th) x.join();
    auto t1 = std::chrono::steady_clock::now();
    (void)hipDeviceSynchronize();
    auto t2 = std::chrono::steady_clock::now();
    const double sub = std::chrono::duration<double, std::micro>(t1 - t0).count(), tot = std::chrono::duration<double, std::micro>(t2 - t0).count();
    printf("threads %d, %d streams: submit %.2f us per launch (aggregate), end-to-end %.2f us per launch\n", T, S, sub / (K * S), tot / (K * S));
    for (auto& s : st) (void)hipStreamDestroy(s);
  }
  return 0;
}
