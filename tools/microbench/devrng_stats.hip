// diagnostic build of the device schedule kernel with per-phase cycle counters
#define ML2048_DEVRNG_STATS 1
#include "../../ml2048_amd/csrc/devrng.hip"
#include <stdio.h>
#include <vector>
int main() {
  void* ws; hipMalloc(&ws, ml2048_devrng_workspace_bytes()); ml2048_devrng_init(0, ws);
  ml2048_devrng h{0x123456789abcdefull, 0xfedcba9876543210ull, 0x1111, 0x2222333344445555ull | 1, 0, 0, 0, 0}; ml2048_devrng* d; hipMalloc(&d, sizeof(h)); hipMemcpy(d, &h, sizeof(h), hipMemcpyHostToDevice);
  std::vector<uint8_t> perm(16384); for (int r = 0; r < 1024; ++r) for (int i = 0; i < 16; ++i) perm[r*16+i] = i;
  uint8_t* dperm; float* dflt; hipMalloc(&dperm, 16384); hipMalloc(&dflt, 4096); hipMemcpy(dperm, perm.data(), 16384, hipMemcpyHostToDevice); hipMemset(dflt, 0, 4096);
  int N = 128; int32_t* sched; uint8_t* ep; uint32_t* masks; int32_t* ne;
  hipMalloc(&sched, N*12); hipMalloc(&ep, (size_t)(N+1)*16384); hipMalloc(&masks, (N+1)*4); hipMalloc(&ne, 128); hipMemset(ne, 0, 128);
  for (int rep = 0; rep < 4; ++rep) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0);
    ml2048_devrng_schedule(0, ws, d, dperm, dflt, N, 0.8, sched, ep, masks, N+1, ne);
    hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
    int32_t hne[32]; hipMemcpy(hne, ne, 128, hipMemcpyDeviceToHost); long long* dbg = (long long*)(hne + 2);
    printf("%.1f us, epochs=%d sweeps=%lld | cycles: gen %lld relax %lld rows %lld replay %lld float %lld total %lld\n", ms*1e3, hne[0], dbg[0], dbg[1], dbg[2], dbg[3], dbg[4], dbg[5], dbg[6]);
  }
}
