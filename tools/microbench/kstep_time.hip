// standalone timing of the C-ABI step entry points (back-to-back launches, HIP events)
#include "../../ml2048_amd/csrc/ml2048_hip.hip"
#include <vector>
int main() {
  const int64_t M = 65536;
  ml2048_state st; void* p[11]; size_t sz[11] = {16,16,4,4,4,4,1,1,4,16,4};
  for (int i = 0; i < 11; ++i) hipMalloc(&p[i], M * sz[i]);
  st.board=(uint8_t*)p[0]; st.merged=(uint8_t*)p[1]; st.valid=(uint8_t*)p[2]; st.step=(int32_t*)p[3]; st.score=(float*)p[4]; st.reward=(float*)p[5];
  st.terminated=(uint8_t*)p[6]; st.invalid=(uint8_t*)p[7]; st.id=(int32_t*)p[8]; st.prev_board=(uint8_t*)p[9]; st.prev_valid=(uint8_t*)p[10];
  std::vector<uint8_t> perm(16384); for (int i = 0; i < 16384; ++i) perm[i] = (i * 7 + i / 16) & 15;
  for (int r = 0; r < 1024; ++r) for (int i = 0; i < 16; ++i) perm[r*16+i] = (i * 5 + r) & 15; // a permutation per row (5 coprime with 16)
  std::vector<float> flt(1024, 0.5f);
  uint8_t* dperm; float* dflt; hipMalloc(&dperm, 16384); hipMalloc(&dflt, 4096);
  hipMemcpy(dperm, perm.data(), 16384, hipMemcpyHostToDevice); hipMemcpy(dflt, flt.data(), 4096, hipMemcpyHostToDevice);
  int32_t *gc, *nr, *scratch; int64_t* idx; hipMalloc(&gc, 4); hipMalloc(&nr, 4); hipMalloc(&scratch, 4 * ml2048_scratch_ints(M)); hipMalloc(&idx, 8 * M);
  hipMemset(gc, 0, 4);
  uint8_t* act; hipMalloc(&act, M); float* lp; hipMalloc(&lp, 4 * M);
  ml2048_reset(0, M, &st);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto timeit = [&](const char* name, auto fn) {
    for (int w = 0; w < 5; ++w) fn(w);
    hipDeviceSynchronize(); hipEventRecord(e0);
    for (int w = 0; w < 100; ++w) fn(w + 5);
    hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %.2f us per call\n", name, ms * 10);
  };
  timeit("prepare (3 kernels)", [&](int w) { ml2048_prepare(0, M, 0, &st, dperm, dflt, 0.8, w, gc, nr, idx, scratch); });
  timeit("sample", [&](int w) { ml2048_sample_actions(0, M, 0, st.valid, 7, w, 0, act, lp); });
  timeit("step", [&](int w) { ml2048_step(0, M, 0, &st, act, 0, 0, 0.8, w, dperm, dflt); });
  timeit("prepare+sample+step", [&](int w) { ml2048_prepare(0, M, 0, &st, dperm, dflt, 0.8, w, gc, nr, idx, scratch); ml2048_sample_actions(0, M, 0, st.valid, 7, w, 0, act, lp); ml2048_step(0, M, 0, &st, act, 0, 0, 0.8, w, dperm, dflt); });
  timeit("step_wave", [&](int w) { ml2048_step_wave(0, M, 0, &st, act, 0, 0, 0.8, w, dperm, dflt); });
  timeit("step_fast", [&](int w) { ml2048_step_fast(0, M, 0, &st, act, 0, 0, 0.8, 9, w); });
  printf("%s\n", ml2048_last_error());
}
