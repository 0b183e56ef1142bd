// micro-benchmark: host-side kernel launch rate vs number of host threads (one stream per thread)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
__global__ void k_empty(int* p) { if (p && threadIdx.x == 12345) *p = 1; }
int main() {
  for (int nt : {1, 2, 4, 8}) {
    std::vector<hipStream_t> st(nt);
    for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const int N = 20000;
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t)
      th.emplace_back([&, t] {
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st[t], (int*)nullptr);
        hipStreamSynchronize(st[t]);
      });
    for (auto& x : th) x.join();
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("threads %d: %.2f us per launch per thread, %.0f launches/s total\n", nt, dt / N * 1e6, nt * N / dt);
    for (auto& s : st) hipStreamDestroy(s);
  }
  // single thread round-robin over 4 streams
  {
    std::vector<hipStream_t> st(4);
    for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const int N = 40000;
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st[i & 3], (int*)nullptr);
    for (auto& s : st) hipStreamSynchronize(s);
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("1 thread over 4 streams: %.2f us per launch\n", dt / N * 1e6);
  }
  return 0;
}
