// encoder.hip - the front end of the reference's CNN encoder: the three depthwise
// convolutions over the one-hot board, evaluated straight from the 16-byte board.
//
// CNNEncoder.forward (src/ml2048/policy/_network.py:84-119) one-hot encodes the board (16 classes x 16
// cells, 1 KiB of floats per board), runs three depthwise convolutions over it - the whole board
// (Conv1d(16, 256, 16, groups=16), :30-35, :98-100), its rows (Conv2d (1,4), :42-47, :108-110) and its
// columns (Conv2d (4,1), :54-59, :114-116) - and applies leaky_relu to each.  On a one-hot input a
// depthwise convolution is a WEIGHT GATHER: channel c = 16*k + j belongs to class k, and
//     full[c]      = b_full[c] + sum over cells with board[cell] == k          of w_full[c][cell]
//     rows[c][r]   = b_rows[c] + sum over columns w with board[r][w] == k      of w_rows[c][w]
//     cols[c][w]   = b_cols[c] + sum over rows r with board[r][w] == k         of w_cols[c][r]
// so the 256 + 1024 + 1024 activations (9 KiB) can be produced from the 16-byte board without ever
// materialising the one-hot: 16 B read + 9 216 B written per board instead of ~31 KiB of traffic in seven
// launches (one-hot, three convolutions, three activations).  The pointwise convolutions behind it are
// dense GEMMs and stay on hipBLASLt.  Floating point: every output is a sum of its bias and at most 16
// (full) / 4 (rows, columns) weights, added in ascending cell order; torch's convolution may add them in
// another order, so parity is to a relative tolerance (tests/test_encoder_front.py), not bit-exact.
//
// Mapping: one wavefront per board on the store side (as k_onehot): the 2 304 floats of a board are 576
// float4 units, 9 per lane, each a coalesced 1 KiB global_store_dwordx4 per wave; unit u < 64 is full
// channels 4u..4u+3 (one class), 64 <= u < 320 is rows channel u-64 (its four row positions), the rest
// columns.  The weights (28 KiB) sit in LDS, transposed so that a unit's four values are one ds_read_b128.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/ml2048_hip.h"

namespace {

constexpr int kThreads = 256;
constexpr int kCh = 256; // 16 classes x multiplier 16

// 0x01 in every byte of `word` equal to the class byte (bytes < 0x80); v_cvt_f32_ubyteN then turns one
// such byte into the float 0.0 / 1.0 that multiplies a weight: acc = fma(flag, w, acc) is exactly
// "acc += w if the cell has this class" (1.0 * w is exact), in two instructions per term and without a
// compare / select pair per term
__device__ __forceinline__ uint32_t eq_bytes(uint32_t word, uint32_t kkkk) {
    return (~((word ^ kkkk) + 0x7F7F7F7Fu) & 0x80808080u) >> 7;
}
__device__ __forceinline__ float byte_f32(uint32_t x, int j) { return (float)((x >> (8 * j)) & 0xFFu); } // v_cvt_f32_ubyteN

__device__ __forceinline__ float lrelu(float x, float slope) { return fmaxf(x, x * slope); } // 0 <= slope <= 1

// the 9 KiB of activations per board are written once and read once, by the GEMM behind this kernel: non-temporal
// stores (ML2048_ENC_NT=0: A/B)
__device__ __forceinline__ void store_act(float4 *dst, float a, float b, float c, float d) {
#if !defined(ML2048_ENC_NT) || ML2048_ENC_NT
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f q = {a, b, c, d};
    __builtin_nontemporal_store(q, reinterpret_cast<v4f *>(dst));
#else
    *dst = make_float4(a, b, c, d);
#endif
}

__global__ __launch_bounds__(kThreads) void k_encoder_front(const uint4 *__restrict__ board, int64_t N,
                                                            const float *__restrict__ w_full, const float *__restrict__ b_full,
                                                            const float *__restrict__ w_rows, const float *__restrict__ b_rows,
                                                            const float *__restrict__ w_cols, const float *__restrict__ b_cols,
                                                            float slope, float4 *__restrict__ out_full,
                                                            float4 *__restrict__ out_rows, float4 *__restrict__ out_cols) {
    // s_full[cell][c]: the 4 channels of a unit are contiguous; s_rows[c][w] / s_cols[c][r] as given
    __shared__ float s_full[16][kCh];
    __shared__ float s_rows[kCh][4], s_cols[kCh][4];
    __shared__ float s_bf[kCh], s_br[kCh], s_bc[kCh];
    for (int e = threadIdx.x; e < kCh * 16; e += kThreads) s_full[e & 15][e >> 4] = w_full[e]; // w_full[c][cell]
    for (int e = threadIdx.x; e < kCh * 4; e += kThreads) {
        s_rows[e >> 2][e & 3] = w_rows[e];
        s_cols[e >> 2][e & 3] = w_cols[e];
    }
    for (int e = threadIdx.x; e < kCh; e += kThreads) {
        s_bf[e] = b_full[e];
        s_br[e] = b_rows[e];
        s_bc[e] = b_cols[e];
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * kThreads + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * kThreads) >> 6;
    for (int64_t n = wave; n < N; n += n_waves) {
        const uint4 b = board[n]; // the same 16 bytes in every lane (one broadcast load)
        const uint32_t r[4] = {b.x, b.y, b.z, b.w};
        // ---- full: unit = lane, channels 4*lane .. 4*lane+3, class k = lane >> 2
        {
            const uint32_t kkkk = (uint32_t)(lane >> 2) * 0x01010101u;
            const int c0 = 4 * lane;
            float4 acc = make_float4(s_bf[c0], s_bf[c0 + 1], s_bf[c0 + 2], s_bf[c0 + 3]);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t eq = eq_bytes(r[q], kkkk);
#pragma unroll
                for (int w = 0; w < 4; ++w) { // cells in ascending order
                    const float f = byte_f32(eq, w);
                    const float4 v = *reinterpret_cast<const float4 *>(&s_full[4 * q + w][c0]);
                    acc.x = __builtin_fmaf(f, v.x, acc.x);
                    acc.y = __builtin_fmaf(f, v.y, acc.y);
                    acc.z = __builtin_fmaf(f, v.z, acc.z);
                    acc.w = __builtin_fmaf(f, v.w, acc.w);
                }
            }
            store_act(out_full + n * 64 + lane, lrelu(acc.x, slope), lrelu(acc.y, slope), lrelu(acc.z, slope), lrelu(acc.w, slope));
        }
        // ---- rows and columns: unit = one channel c (class k = c >> 4), its four positions
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int c = lane + 64 * p;
            const uint32_t kkkk = (uint32_t)(c >> 4) * 0x01010101u;
            float f[4][4]; // f[row][col] = 1.0 where the cell has class k
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t eq = eq_bytes(r[q], kkkk);
#pragma unroll
                for (int w = 0; w < 4; ++w) f[q][w] = byte_f32(eq, w);
            }
            const float4 wr = *reinterpret_cast<const float4 *>(s_rows[c]);
            const float4 wc = *reinterpret_cast<const float4 *>(s_cols[c]);
            const float wrv[4] = {wr.x, wr.y, wr.z, wr.w}, wcv[4] = {wc.x, wc.y, wc.z, wc.w};
            float rows[4], cols[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float a = s_br[c]; // row q: columns in ascending order
                float d = s_bc[c]; // column q: rows in ascending order
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    a = __builtin_fmaf(f[q][t], wrv[t], a);
                    d = __builtin_fmaf(f[t][q], wcv[t], d);
                }
                rows[q] = a;
                cols[q] = d;
            }
            store_act(out_rows + n * kCh + c, lrelu(rows[0], slope), lrelu(rows[1], slope), lrelu(rows[2], slope), lrelu(rows[3], slope));
            store_act(out_cols + n * kCh + c, lrelu(cols[0], slope), lrelu(cols[1], slope), lrelu(cols[2], slope), lrelu(cols[3], slope));
        }
    }
}

} // namespace

extern "C" int ml2048_encoder_front(void *stream, int64_t N, const uint8_t *board, const float *w_full,
                                    const float *b_full, const float *w_rows, const float *b_rows, const float *w_cols,
                                    const float *b_cols, float negative_slope, float *out_full, float *out_rows,
                                    float *out_cols) {
    if (N <= 0 || !board || !w_full || !b_full || !w_rows || !b_rows || !w_cols || !b_cols || !out_full || !out_rows || !out_cols)
        return ML2048_E_INVALID;
    if (((uintptr_t)board | (uintptr_t)out_full | (uintptr_t)out_rows | (uintptr_t)out_cols) % 16) return ML2048_E_ALIGN;
    int64_t blocks = (N + 3) / 4; // one wave per board, 4 waves per workgroup, grid-stride: the 28 KiB of weights
    if (blocks > 2048) blocks = 2048; // are staged into LDS once per workgroup
    hipLaunchKernelGGL(k_encoder_front, dim3((unsigned)blocks), dim3(kThreads), 0, (hipStream_t)stream,
                       reinterpret_cast<const uint4 *>(board), N, w_full, b_full, w_rows, b_rows, w_cols, b_cols,
                       negative_slope, reinterpret_cast<float4 *>(out_full), reinterpret_cast<float4 *>(out_rows),
                       reinterpret_cast<float4 *>(out_cols));
    return (int)hipGetLastError();
}
