#include <chrono>
#include <stdio.h>
#include "../../ml2048_amd/csrc/hostrng.cpp"
int main() {
    static uint64_t a[8576], b[8576];
    u128 base = ((u128)123 << 64) | 456, inc = (((u128)789 << 64) | 1011) | 1;
    for (int mode = 0; mode < 2; ++mode) {
        auto t0 = std::chrono::steady_clock::now();
        for (int rep = 0; rep < 2000; ++rep) {
            if (mode) gen_outputs_avx512(lcg_advance(base, inc, rep * 8192ull) * kMul + inc, inc, b, 8576);
            else { setenv("ML2048_HOSTRNG_AVX512", "0", 1); gen_outputs(base, inc, rep * 8192ull, a, 8576); }
        }
        auto t1 = std::chrono::steady_clock::now();
        printf("%s: %.2f us per 8576 outputs\n", mode ? "avx512" : "scalar4", std::chrono::duration<double, std::micro>(t1 - t0).count() / 2000);
    }
    int bad = 0; for (int i = 0; i < 8576; ++i) bad += a[i] != b[i];
    printf("mismatch %d\n", bad);
}
