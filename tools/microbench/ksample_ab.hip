#include <hip/hip_runtime.h>
#include <rocrand/rocrand_kernel.h>
#include <stdio.h>
#include "../../ml2048_amd/csrc/board_ops.hpp"
using namespace ml2048;
template <int V>
__global__ __launch_bounds__(256) void ks(const uint8_t* valid, int64_t M, uint64_t seed, uint64_t counter, uint8_t* actions, float* logp) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; if (i >= M) return;
  uint32_t v = reinterpret_cast<const uint32_t*>(valid)[i];
  uint32_t word;
  if (V == 1) { rocrand_state_philox4x32_10 s; rocrand_init(seed, i, counter * 4ull, &s); word = rocrand(&s); }
  else if (V == 2) { rocrand_state_philox4x32_10 s; rocrand_init(seed, i, counter, &s); word = rocrand(&s); }
  else { word = sampler_word(seed, counter, i); }
  uint32_t cnt; uint32_t a = pick_action(v, word, 1, cnt);
  actions[i] = (uint8_t)a; logp[i] = neg_log_count(cnt);
}
template <int V> float run(const uint8_t* valid, int64_t M, uint8_t* a, float* lp) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 10; ++w) hipLaunchKernelGGL(ks<V>, dim3((M + 255) / 256), dim3(256), 0, 0, valid, M, 7ull, (uint64_t)w, a, lp);
  hipEventRecord(e0);
  for (int w = 0; w < 200; ++w) hipLaunchKernelGGL(ks<V>, dim3((M + 255) / 256), dim3(256), 0, 0, valid, M, 7ull, (uint64_t)w, a, lp);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 200 * 1000;
}
int main() {
  int64_t M = 65536; uint8_t *valid, *a; float* lp;
  hipMalloc(&valid, M * 4); hipMalloc(&a, M); hipMalloc(&lp, M * 4); hipMemset(valid, 1, M * 4);
  printf("V1 rocrand offset 4c : %.2f us\n", run<1>(valid, M, a, lp));
  printf("V2 rocrand offset c  : %.2f us\n", run<2>(valid, M, a, lp));
  printf("V3 own philox        : %.2f us\n", run<3>(valid, M, a, lp));
  printf("V1 again             : %.2f us\n", run<1>(valid, M, a, lp));
}
