// board_ops.hpp - per-board 2048 arithmetic for gfx950, one board per lane, SIMD-within-register.
//
// A board is 16 uint8 tile exponents (0 = empty, k = tile 2^k; reference layout
// src/ml2048/game_numba.py:13-20, :542) held as four 32-bit row words: r[i] byte j = cell 4i+j.
// All game logic is branch-free byte-parallel integer work on those four VGPRs (no MFMA: this is
// indexing work, not a contraction).  Exponents are < 0x80, which the carry-free byte tests below
// rely on (the reference's tiles top out at 17, ITEM_VALUES game_numba.py:23-45).
//
// The functions are plain C++ so that tests/csrc/host_ops_harness.cpp can compile this very file
// with g++ and check the bit tricks exhaustively on the CPU (a test harness, not a product path);
// on the device `perm_b32` lowers to one v_perm_b32.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define ML_HD __host__ __device__ __forceinline__
#else
#define ML_HD inline
#endif

namespace ml2048 {

// "does any lane of this wavefront see `cond`" on the device (a wave-uniform answer: the caller's branch is a scalar
// branch with no exec-mask bookkeeping); the condition itself on the host
ML_HD bool any_lane(bool cond) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ballot_w64(cond) != 0ull;
#else
    return cond;
#endif
}

struct Board {
    uint32_t r[4];
};

// ---- byte shuffles ---------------------------------------------------------------------------
// v_perm_b32 semantics: pool = {hi:bytes 4-7, lo:bytes 0-3}; selector byte 0x0c yields 0x00.
ML_HD uint32_t perm_b32(uint32_t hi, uint32_t lo, uint32_t sel) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(hi, lo, sel);
#else
    uint64_t pool = ((uint64_t)hi << 32) | lo;
    uint32_t out = 0;
    for (int i = 0; i < 4; ++i) {
        uint32_t s = (sel >> (8 * i)) & 0xFF;
        uint32_t b = (s < 8) ? (uint32_t)((pool >> (8 * s)) & 0xFF) : (s == 0x0c ? 0u : 0xFFu);
        out |= b << (8 * i);
    }
    return out;
#endif
}

ML_HD uint32_t bswap32(uint32_t x) { return perm_b32(0, x, 0x00010203u); }

// bit-field extract (v_bfe_u32): (x >> offset) & ((1 << width) - 1), offset taken mod 32
ML_HD uint32_t ubfe(uint32_t x, uint32_t offset, uint32_t width) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ubfe(x, offset, width);
#else
    return (x >> (offset & 31u)) & ((1u << width) - 1u);
#endif
}

// 0x80 in every byte of x that is non-zero (bytes must be <= 0x80)
ML_HD uint32_t nz_bytes(uint32_t x) { return (x + 0x7F7F7F7Fu) & 0x80808080u; }
// 0x80 in every byte of x that is zero
ML_HD uint32_t z_bytes(uint32_t x) { return ~(x + 0x7F7F7F7Fu) & 0x80808080u; }
// gather the four 0x80 flags of a byte mask into bits 0..3
ML_HD uint32_t flags4(uint32_t m80) { return (((m80 >> 7) * 0x00204081u) >> 21) & 0xFu; }

// 4x4 byte transpose (an involution): 8 v_perm_b32
ML_HD Board transpose(const Board &b) {
    // interleave bytes of (r0,r1) and (r2,r3)
    uint32_t a_lo = perm_b32(b.r[1], b.r[0], 0x05010400u); // r0.b0 r1.b0 r0.b1 r1.b1
    uint32_t a_hi = perm_b32(b.r[1], b.r[0], 0x07030602u); // r0.b2 r1.b2 r0.b3 r1.b3
    uint32_t c_lo = perm_b32(b.r[3], b.r[2], 0x05010400u);
    uint32_t c_hi = perm_b32(b.r[3], b.r[2], 0x07030602u);
    Board t;
    t.r[0] = perm_b32(c_lo, a_lo, 0x05040100u); // a_lo.lo16 | c_lo.lo16
    t.r[1] = perm_b32(c_lo, a_lo, 0x07060302u); // a_lo.hi16 | c_lo.hi16
    t.r[2] = perm_b32(c_hi, a_hi, 0x05040100u);
    t.r[3] = perm_b32(c_hi, a_hi, 0x07060302u);
    return t;
}

// A two-stage network of eight v_perm_b32 that pairs the rows (0, 2) and (1, 3) in its first stage: what it does to the
// board is decided by its byte selectors alone (a pair per stage), which may differ from lane to lane.
ML_HD Board perm_net(const Board &b, uint32_t s1x, uint32_t s1y, uint32_t s2x, uint32_t s2y) {
    const uint32_t t0 = perm_b32(b.r[2], b.r[0], s1x), t1 = perm_b32(b.r[2], b.r[0], s1y);
    const uint32_t t2 = perm_b32(b.r[3], b.r[1], s1x), t3 = perm_b32(b.r[3], b.r[1], s1y);
    Board o;
    o.r[0] = perm_b32(t2, t0, s2x);
    o.r[1] = perm_b32(t2, t0, s2y);
    o.r[2] = perm_b32(t3, t1, s2x);
    o.r[3] = perm_b32(t3, t1, s2y);
    return o;
}
// The transposition OR the identity: (sx, sy) = (0x05010400, 0x07030602) in both stages transposes, (0x03020100,
// 0x07060504) hands the board back.  `if (horizontal) b = transpose(b)` costs a wave the exec-mask round trip and four
// v_mov for the lanes that keep their board; this form costs two selects per step, off the board's dependency chain
// (the selectors only depend on the action).
ML_HD Board transpose_by(const Board &b, uint32_t sx, uint32_t sy) { return perm_net(b, sx, sy, sx, sy); }

// The same network with a selector pair per stage does the whole orientation of a move - transpose for a horizontal
// move AND the reversal of the cell order for a move toward the far side - and its inverse (checked for every action by
// tests/test_host_ops.py): vertical moves reverse by swapping which source word a selector byte names (^ 0x04 in both
// stages), horizontal ones by reversing the byte order the first stage reads (^ 0x03) or the second stage of the
// inverse writes (^ 0x05).  s[0..1] / s[2..3]: stage 1 / 2 of the forward network; s[2..3] / s[4..5]: of the inverse.
ML_HD void move_selectors(uint32_t action, uint32_t s[6]) {
    const bool horizontal = (action & 2u) == 0;
    const uint32_t rev = (action & 1u) ? 0xFFFFFFFFu : 0u;
    const uint32_t sx = horizontal ? 0x05010400u : 0x03020100u, sy = horizontal ? 0x07030602u : 0x07060504u;
    const uint32_t m_a = rev & (horizontal ? 0x03030303u : 0x04040404u);
    const uint32_t m_b = rev & (horizontal ? 0u : 0x04040404u);
    const uint32_t m_d = rev & (horizontal ? 0x05050505u : 0x04040404u);
    s[0] = sx ^ m_a; s[1] = sy ^ m_a;
    s[2] = sx ^ m_b; s[3] = sy ^ m_b;
    s[4] = sx ^ m_d; s[5] = sy ^ m_d;
}

// ---- one line, pushed toward byte 0 (reference _push_row, game_numba.py:48-90) ---------------
// Returns the new line; mexp gets the pre-merge exponents of the (at most two) merges in
// bytes 0 and 1 (0 = no merge): exactly the indices `buckets[num] += 1` touches (:76).
ML_HD uint32_t push_line(uint32_t x, uint32_t &mexp) {
    // stable compaction of the non-zero bytes toward byte 0, high end first
    uint32_t s2 = perm_b32(0, x, 0x0c030100u); // drop byte 2
    x = (x & 0x00FF0000u) ? x : s2;
    uint32_t s1 = perm_b32(0, x, 0x0c030200u); // drop byte 1
    x = (x & 0x0000FF00u) ? x : s1;
    x = (x & 0x000000FFu) ? x : (x >> 8);
    // equal neighbours among the compacted bytes c0..c3
    uint32_t e = z_bytes(x ^ (x >> 8)) & nz_bytes(x); // 0x80 at byte i  <=>  c_i == c_{i+1} != 0
    bool e01 = (e & 0x00000080u) != 0;
    bool e12 = (e & 0x00008000u) != 0;
    bool e23 = (e & 0x00800000u) != 0;
    uint32_t c0 = x & 0xFFu, c1 = (x >> 8) & 0xFFu, c2 = (x >> 16) & 0xFFu;
    // e01: [c0+1, c2, c3, 0]; and e23 on top: [c0+1, c2+1, 0, 0]
    uint32_t rB = perm_b32(0, x, 0x0c030200u) + 0x00000001u;
    uint32_t rA = (rB & 0x0000FFFFu) + 0x00000100u;
    // e12 alone: [c0, c1+1, c3, 0]
    uint32_t rC = perm_b32(0, x, 0x0c030100u) + 0x00000100u;
    // e23 alone: [c0, c1, c2+1, 0]
    uint32_t rD = (x & 0x00FFFFFFu) + 0x00010000u;
    uint32_t res = e01 ? (e23 ? rA : rB) : (e12 ? rC : (e23 ? rD : x));
    uint32_t m_first = e01 ? c0 : (e12 ? c1 : (e23 ? c2 : 0u));
    uint32_t m_second = (e01 && e23) ? c2 : 0u;
    mexp = m_first | (m_second << 8);
    return res;
}

struct MoveResult {
    Board board;
    uint32_t mexp_first;  // byte j = exponent of the first merge in line j (0 = none)
    uint32_t mexp_second; // byte j = exponent of the second merge in line j
};

// reference _step_kernel (game_numba.py:93-134): L (0,+1) R (3,-1) U (0,+4) D (12,-4)
// Line-at-a-time form: four push_line calls (kept as the cross-check of move_board below; the
// exhaustive 18^4 line test runs on push_line).
ML_HD MoveResult move_board_by_lines(const Board &in, uint32_t action) {
    const bool vertical = (action & 2u) != 0;
    const bool reverse = (action & 1u) != 0;
    Board b = in;
    if (vertical) b = transpose(b);
    MoveResult out;
    out.mexp_first = 0;
    out.mexp_second = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        uint32_t x = b.r[j];
        x = reverse ? bswap32(x) : x;
        uint32_t m;
        x = push_line(x, m);
        x = reverse ? bswap32(x) : x;
        b.r[j] = x;
        out.mexp_first |= (m & 0xFFu) << (8 * j);
        out.mexp_second |= (m >> 8) << (8 * j);
    }
    if (vertical) b = transpose(b);
    out.board = b;
    return out;
}

// 0x7F in every byte of x that is zero / non-zero (bytes < 0x80): a byte-wide blend mask
ML_HD uint32_t zmask7(uint32_t x) {
    const uint32_t m = ~(x + 0x7F7F7F7Fu) & 0x80808080u;
    return m - (m >> 7);
}

// "if cell i is empty, pull cell i+1 into it", for the four lines held bytewise in (lo, hi)
ML_HD void pull_into_empty(uint32_t &lo, uint32_t &hi) {
    const uint32_t z = zmask7(lo);
    lo |= hi & z;
    hi &= ~z;
}

// Stable compaction of the non-empty cells of four lines (cell words a, b, c, d) toward a.  An empty cell adds nothing
// to an OR, so every cell is simply OR-ed into the position it lands on: cell j moves up by the number of empty cells in
// front of it, and "exactly k of the cells in front are empty" is one three-input boolean function of their zero masks
// (one v_bitop3_b32 each on gfx950).  31 VALU instead of 39 for the six-step bubble of pull_into_empty it replaces
// (which tests/csrc/host_ops_harness.cpp keeps as its cross-check).
ML_HD void compact_cells(uint32_t &a, uint32_t &b, uint32_t &c, uint32_t &d) {
    const uint32_t za = zmask7(a), zb = zmask7(b), zc = zmask7(c);
    const uint32_t ab2 = za & zb;                 // both cells in front of c are empty
    const uint32_t abc3 = ab2 & zc;               // all three in front of d
    const uint32_t ab1 = za ^ zb;                 // exactly one of a, b
    const uint32_t abc2 = ((za & zb) | (za & zc) | (zb & zc)) & ~abc3; // exactly two of a, b, c
    const uint32_t abc1 = (za ^ zb ^ zc) & ~abc3; // exactly one
    const uint32_t any_ab = za | zb;
    const uint32_t r0 = a | (b & za) | (c & ab2) | (d & abc3);
    const uint32_t r1 = (b & ~za) | (c & ab1) | (d & abc2);
    const uint32_t r2 = (c & ~any_ab) | (d & abc1);
    const uint32_t r3 = d & ~(any_ab | zc);
    a = r0;
    b = r1;
    c = r2;
    d = r3;
}

// All four lines at once.  The board is held as four CELL words w0..w3: byte j of w_k is cell k
// (counted from the side the tiles move to) of line j, so one pass of the scalar line algorithm
// (_push_row, :48-90) on whole words moves the four lines together:
//   1. stable compaction of the non-empty cells toward w0 (compact_cells);
//   2. merge marks on the compacted cells (a, b, c, d): e01 = (a == b != 0), e12 = (b == c != 0) and
//      not e01, e23 = (c == d != 0) and not e12 - exactly the pairs the reference's left-to-right scan
//      merges; the merged cell is incremented, the consumed one removed and the rest closed up.
// For up/down the cell words are the rows themselves (line j = column j); for left/right they are
// the columns, i.e. the transposed board (line j = row j); the opposite direction reverses the cell
// order.  ~100 VALU operations per move instead of ~250 for four separate line pushes.
ML_HD MoveResult move_board(const Board &in, uint32_t action) {
    const bool horizontal = (action & 2u) == 0;
    const bool reverse = (action & 1u) != 0;
    Board b = in;
    if (horizontal) b = transpose(b);
    uint32_t a = reverse ? b.r[3] : b.r[0], bb = reverse ? b.r[2] : b.r[1];
    uint32_t c = reverse ? b.r[1] : b.r[2], d = reverse ? b.r[0] : b.r[3];
    // 1. compaction
    compact_cells(a, bb, c, d);
    // 2. merge marks (0x80 per line)
    const uint32_t na = (a + 0x7F7F7F7Fu) & 0x80808080u, nb = (bb + 0x7F7F7F7Fu) & 0x80808080u;
    const uint32_t nc = (c + 0x7F7F7F7Fu) & 0x80808080u;
    const uint32_t e01 = ~((a ^ bb) + 0x7F7F7F7Fu) & na;
    const uint32_t e12 = ~((bb ^ c) + 0x7F7F7F7Fu) & nb & ~e01;
    const uint32_t e23 = ~((c ^ d) + 0x7F7F7F7Fu) & nc & ~e12;
    const uint32_t i01 = e01 >> 7, i12 = e12 >> 7, i23 = e23 >> 7; // +1 per merging line
    const uint32_t m01 = e01 - i01, m12 = e12 - i12, m23 = e23 - i23; // 0x7F blend masks
    MoveResult out;
    out.mexp_first = (a & m01) | (bb & m12) | (c & m23 & ~m01);
    out.mexp_second = c & m23 & m01;
    const uint32_t r0 = a + i01;
    const uint32_t v1 = bb + i12;
    const uint32_t v2 = (c + i23) & ~m12; // consumed by e12
    const uint32_t v3 = d & ~m23;         // consumed by e23
    // e01 consumed b: close the gap
    const uint32_t p1 = (v2 & m01) | (v1 & ~m01);
    const uint32_t p2 = (v3 & m01) | (v2 & ~m01);
    const uint32_t p3 = v3 & ~m01;
    // e12 (never together with e01) left a hole at cell 2
    const uint32_t q2 = p2 | (p3 & m12);
    const uint32_t q3 = p3 & ~m12;
    b.r[0] = reverse ? q3 : r0;
    b.r[1] = reverse ? q2 : p1;
    b.r[2] = reverse ? p1 : q2;
    b.r[3] = reverse ? r0 : q3;
    if (horizontal) b = transpose(b);
    out.board = b;
    return out;
}

// ---- the move by table look-up ---------------------------------------------------------------------------------------
// A line of four cells with exponents <= 15 is a 16-bit index (cell k, counted from the side the tiles move to, in
// nibble k) and what _push_row (:48-90) makes of it is a 16-bit value: 65 536 entries x 2 bytes = 128 KiB, which fits
// the 160 KiB of LDS of a CDNA4 compute unit next to the rollout's 16 KiB of parked flags.  A fused rollout that keeps
// nothing of a move but the board (the loops without reward arithmetic: score_potential) then moves four lines with four
// ds_read_u16 and ~20 VALU around them instead of the ~85 VALU of compact_cells + merge marks + gap closing.
// A merge of two 2^15 tiles has no 4-bit result: its entry is 0 and callers must rule it out (lut_guard).
constexpr int kRowLutEntries = 65536;
ML_HD uint16_t row_lut_entry(uint32_t idx) {
    const uint32_t x = (idx & 0xFu) | ((idx & 0xF0u) << 4) | ((idx & 0xF00u) << 8) | ((idx & 0xF000u) << 12);
    uint32_t mexp;
    const uint32_t y = push_line(x, mexp);
    if (y & 0xF0F0F0F0u) return 0;
    return (uint16_t)((y & 0xFu) | ((y >> 4) & 0xF0u) | ((y >> 8) & 0xF00u) | ((y >> 12) & 0xF000u));
}

// Where a line's entry lives in the SWIZZLED form of the table.  LDS banks are picked by the low address bits, and in play
// the cells are small numbers: with the plain index (cell 0 in bits 0-3, cell 1 in bits 4-7) most look-ups of a wave fall
// on a dozen banks (12 of the 15 LDS cycles of a ds_read_u16 were bank conflicts).  XOR-ing the low byte with the high
// byte shifted up by two - a bijection, one more VALU for the four lines of a board - brings the low bits of three cells
// into the six bits that pick bank and half-word.  It pays where a wave is alone on its SIMD and waits out every
// look-up (C2: 66.9 -> 65.1 us); with several waves per SIMD the extra instruction costs what the shorter LDS time gains
// (262 144 boards: 147.8 vs 148.7 us), so the large-M pass keeps the plain index.
ML_HD uint32_t row_lut_slot(uint32_t idx) { return idx ^ ((idx >> 6) & 0xFCu); }

// move_board(in, action).board for boards whose lines are in the table's domain; lut[SWZ ? row_lut_slot(idx) : idx] =
// row_lut_entry(idx) for the 65 536 line indices (in LDS on the device).
// Measured and dropped: the next step's table requests issued between the look-ups and their use (C2: 67.0 vs 66.9 us).
template <bool SWZ = false, class LutPtr>
ML_HD Board move_board_lut(const Board &in, uint32_t action, LutPtr lut) {
    const bool horizontal = (action & 2u) == 0;
    const bool reverse = (action & 1u) != 0;
#if !defined(ML2048_TRANSPOSE_BY) || ML2048_TRANSPOSE_BY // (-DML2048_TRANSPOSE_BY=0: the exec-masked transposes, A/B)
    const uint32_t tsx = horizontal ? 0x05010400u : 0x03020100u, tsy = horizontal ? 0x07030602u : 0x07060504u;
    Board b = transpose_by(in, tsx, tsy);
#else
    Board b = in;
    if (horizontal) b = transpose(b);
#endif
    const uint32_t a = reverse ? b.r[3] : b.r[0], bb = reverse ? b.r[2] : b.r[1];
    const uint32_t c = reverse ? b.r[1] : b.r[2], d = reverse ? b.r[0] : b.r[3];
    // byte j of n01 / n23 = the low / high byte of line j's index (row_lut_slot on all four lines at once)
    const uint32_t n23 = c | (d << 4);
    const uint32_t n01 = SWZ ? (a | (bb << 4)) ^ ((n23 << 2) & 0xFCFCFCFCu) : a | (bb << 4);
    const uint32_t r0 = lut[perm_b32(n23, n01, 0x0c0c0400u)];
    const uint32_t r1 = lut[perm_b32(n23, n01, 0x0c0c0501u)];
    const uint32_t r2 = lut[perm_b32(n23, n01, 0x0c0c0602u)];
    const uint32_t r3 = lut[perm_b32(n23, n01, 0x0c0c0703u)];
    const uint32_t r01 = r0 | (r1 << 16), r23 = r2 | (r3 << 16);
    const uint32_t e = perm_b32(r23, r01, 0x06040200u); // byte j = cells 0, 1 of line j as nibbles
    const uint32_t f = perm_b32(r23, r01, 0x07050301u); // byte j = cells 2, 3
    const uint32_t o0 = e & 0x0F0F0F0Fu, o1 = (e >> 4) & 0x0F0F0F0Fu;
    const uint32_t o2 = f & 0x0F0F0F0Fu, o3 = (f >> 4) & 0x0F0F0F0Fu;
    b.r[0] = reverse ? o3 : o0;
    b.r[1] = reverse ? o2 : o1;
    b.r[2] = reverse ? o1 : o2;
    b.r[3] = reverse ? o0 : o3;
#if !defined(ML2048_TRANSPOSE_BY) || ML2048_TRANSPOSE_BY
    return transpose_by(b, tsx, tsy);
#else
    if (horizontal) b = transpose(b);
    return b;
#endif
}

// move_board_lut with the orientation folded into the byte selectors: `sel` = a table of move_selectors(action) at
// sel[8 * action .. 8 * action + 5] (the rollout kernel keeps it in LDS: two look-ups per step replace the direction
// decode, the two selects of transpose_by's selectors and the eight selects of the cell-order reversal)
template <bool SWZ = false, class LutPtr, class SelPtr>
ML_HD Board move_board_lut_sel(const Board &in, uint32_t action, LutPtr lut, SelPtr sel) {
    const uint32_t *sp = &sel[8u * (action & 3u)];
    const uint32_t s0 = sp[0], s1 = sp[1], s2 = sp[2], s3 = sp[3], s4 = sp[4], s5 = sp[5];
    const Board b = perm_net(in, s0, s1, s2, s3);
    const uint32_t n23 = b.r[2] | (b.r[3] << 4);
    const uint32_t n01 = SWZ ? (b.r[0] | (b.r[1] << 4)) ^ ((n23 << 2) & 0xFCFCFCFCu) : b.r[0] | (b.r[1] << 4);
    const uint32_t r0 = lut[perm_b32(n23, n01, 0x0c0c0400u)];
    const uint32_t r1 = lut[perm_b32(n23, n01, 0x0c0c0501u)];
    const uint32_t r2 = lut[perm_b32(n23, n01, 0x0c0c0602u)];
    const uint32_t r3 = lut[perm_b32(n23, n01, 0x0c0c0703u)];
    const uint32_t r01 = r0 | (r1 << 16), r23 = r2 | (r3 << 16);
    const uint32_t e = perm_b32(r23, r01, 0x06040200u);
    const uint32_t f = perm_b32(r23, r01, 0x07050301u);
    Board o;
    o.r[0] = e & 0x0F0F0F0Fu;
    o.r[1] = (e >> 4) & 0x0F0F0F0Fu;
    o.r[2] = f & 0x0F0F0F0Fu;
    o.r[3] = (f >> 4) & 0x0F0F0F0Fu;
    return perm_net(o, s2, s3, s4, s5);
}

// every tile <= 2^11 and at most 8 191 steps ahead: the tile sum (<= 16 * 2048 now, + 4 per step at most) stays below
// 2^16, so no tile reaches 2^16 and no line leaves the table's domain (two 2^15 tiles would need a sum >= 2^16)
constexpr int kRowLutMaxSteps = 8191;
ML_HD bool lut_guard(const Board &b) {
    return (((b.r[0] + 0x74747474u) | (b.r[1] + 0x74747474u) | (b.r[2] + 0x74747474u) | (b.r[3] + 0x74747474u)) &
            0x80808080u) == 0u && ((b.r[0] | b.r[1] | b.r[2] | b.r[3]) & 0x80808080u) == 0u;
}

// ---- merged[16] bucket counters (the record's `merged` field) ---------------------------------
// merged[num] += 1 for num in 1..15; num >= 16 is the reference's out-of-bounds write (H9) whose
// target byte is rewritten by _compute_valid_actions before anything reads it, so it is dropped.
ML_HD void add_bucket(uint32_t m[4], uint32_t num) {
    uint32_t ok = (num - 1u) < 15u;
    uint32_t inc = ok << ((num & 3u) * 8u);
    uint32_t d = num >> 2;
    m[0] += (d == 0u) ? inc : 0u;
    m[1] += (d == 1u) ? inc : 0u;
    m[2] += (d == 2u) ? inc : 0u;
    m[3] += (d == 3u) ? inc : 0u;
}

struct MergeStats {
    uint32_t merged[4]; // 16 byte counters
    uint32_t sum_value; // sum 2^(num+1)      -> reward_fn_normal (game_numba.py:408-438)
    uint32_t sum_rank;  // sum (num+1)        -> reward_fn_rank   (:469-484)
    uint32_t count;     // number of merges   -> merged.sum() in reward_fn_maxcell (:502)
};

ML_HD MergeStats merge_stats(uint32_t mexp_first, uint32_t mexp_second) {
    MergeStats s;
    s.merged[0] = s.merged[1] = s.merged[2] = s.merged[3] = 0;
    s.sum_value = s.sum_rank = s.count = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        uint32_t word = (k < 4) ? mexp_first : mexp_second;
        uint32_t num = (word >> (8 * (k & 3))) & 0xFFu;
        add_bucket(s.merged, num);
        uint32_t ok = (num - 1u) < 15u;
        s.sum_value += ok ? (2u << num) : 0u;
        s.sum_rank += ok ? (num + 1u) : 0u;
        s.count += ok;
    }
    return s;
}

// sum over the bytes n of (a, b) with 1 <= n <= 15 of 2^(n+1): the value the merges of one move add to the score.
// byte + 31 is, modulo 32, the shift amount n - 1, and 31 for an empty byte - where 4 << 31 drops out of the word -
// so each byte costs one SDWA shift (byte operand in, dword out) and no compare/select.  Bytes >= 16 (H9: the
// reference's out-of-range buckets; tiles of 2^16 and more) take the plain loop - the whole wave does, when one of
// its boards merges such tiles, so that the test is a scalar branch.  The forward declaration is needed because the helper lives with the spawn code.
ML_HD uint32_t shl_byte_by_byte(uint32_t flags, uint32_t amounts, int j);
ML_HD uint32_t merged_value_sum(uint32_t a, uint32_t b) {
    if (__builtin_expect(any_lane(((a | b) & 0xF0F0F0F0u) != 0u), 0)) { // rare, and taken by the whole wave
        uint32_t sum = 0;
        for (int k = 0; k < 8; ++k) {
            const uint32_t num = ((k < 4 ? a : b) >> (8 * (k & 3))) & 0xFFu;
            sum += (num - 1u) < 15u ? (2u << num) : 0u;
        }
        return sum;
    }
    const uint32_t sa = a + 0x1F1F1F1Fu, sb = b + 0x1F1F1F1Fu, four = 0x04040404u;
    return ((shl_byte_by_byte(four, sa, 0) + shl_byte_by_byte(four, sa, 1)) +
            (shl_byte_by_byte(four, sa, 2) + shl_byte_by_byte(four, sa, 3))) +
           ((shl_byte_by_byte(four, sb, 0) + shl_byte_by_byte(four, sb, 1)) +
            (shl_byte_by_byte(four, sb, 2) + shl_byte_by_byte(four, sb, 3)));
}

// the reward-relevant sums only (no bucket array): what a fused rollout needs every step
ML_HD void merge_sums(uint32_t mexp_first, uint32_t mexp_second, int kind, uint32_t &sum_value,
                      uint32_t &sum_rank, uint32_t &count) {
    sum_value = merged_value_sum(mexp_first, mexp_second);
    sum_rank = count = 0;
    if (kind != 2 && kind != 3) return;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t word = (k < 4) ? mexp_first : mexp_second;
        const uint32_t num = ubfe(word, 8u * (k & 3), 8);
        const uint32_t ok = (num - 1u) < 15u;
        if (kind == 2) sum_rank += ok ? (num + 1u) : 0u;
        if (kind == 3) count += ok;
    }
}

// merged[16] byte counters from the merge exponents of one move
ML_HD void merge_buckets(uint32_t mexp_first, uint32_t mexp_second, uint32_t merged[4]) {
    merged[0] = merged[1] = merged[2] = merged[3] = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t word = (k < 4) ? mexp_first : mexp_second;
        add_bucket(merged, ubfe(word, 8u * (k & 3), 8));
    }
}

// ---- valid-action mask (reference _line_movable / _compute_valid_actions, :215-289) -----------
// forward  (toward index 0)  <=> exists i: n[i+1] != 0 and (n[i] == 0 or n[i] == n[i+1])
// backward (toward index 3)  <=> exists i: n[i]   != 0 and (n[i+1] == 0 or n[i] == n[i+1])
// returns bytes {left, right, up, down} as 0/1 in byte 0..3 - the record's valid_actions word.
ML_HD uint32_t valid_actions(const Board &b) {
    uint32_t nz[4], zz[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        nz[i] = nz_bytes(b.r[i]);
        zz[i] = nz[i] ^ 0x80808080u;
    }
    uint32_t left = 0, right = 0, up = 0, down = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { // rows: neighbour of byte j is byte j+1
        uint32_t a = b.r[i], n = a >> 8;
        uint32_t eq = z_bytes(a ^ n);
        uint32_t nz_n = nz[i] >> 8, z_n = (zz[i] >> 8) | 0x80000000u;
        left |= nz_n & (zz[i] | eq);
        right |= nz[i] & (z_n | eq) & 0x00808080u;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) { // columns: neighbour of row i is row i+1, 4 columns at once
        uint32_t eq = z_bytes(b.r[i] ^ b.r[i + 1]);
        up |= nz[i + 1] & (zz[i] | eq);
        down |= nz[i] & (zz[i + 1] | eq);
    }
    left &= 0x00808080u;
    return (left ? 1u : 0u) | (right ? 0x100u : 0u) | (up ? 0x10000u : 0u) | (down ? 0x1000000u : 0u);
}

// bit c set <=> cell c is empty
ML_HD uint32_t empty_mask(const Board &b) {
    uint32_t m = flags4(nz_bytes(b.r[0])) | (flags4(nz_bytes(b.r[1])) << 4) |
                 (flags4(nz_bytes(b.r[2])) << 8) | (flags4(nz_bytes(b.r[3])) << 12);
    return ~m & 0xFFFFu;
}

ML_HD void set_cell(Board &b, uint32_t cell, uint32_t value) {
    uint32_t v = value << ((cell & 3u) * 8u);
    uint32_t d = cell >> 2;
    b.r[0] |= (d == 0u) ? v : 0u;
    b.r[1] |= (d == 1u) ? v : 0u;
    b.r[2] |= (d == 2u) ? v : 0u;
    b.r[3] |= (d == 3u) ? v : 0u;
}

// ---- table-driven spawn, bit-exact with the reference (_spawn2, game_numba.py:172-212) --------
// perm = one row of randperm (16 cell indices); two_mask bit c <=> float64(randfloat[c]) < two_prob
// (randfloat is indexed by the CELL, :207, hazard H1).  Places up to `count` tiles.
ML_HD void spawn_table(Board &b, const uint32_t perm[4], uint32_t two_mask, int count) {
    const uint32_t empties = empty_mask(b);
    uint32_t hits = 0; // bit i <=> cell perm[i] is empty: bfe (cell), bfe (its empty bit), lshl_or
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint32_t c = ubfe(perm[i >> 2], 8u * (i & 3), 4);
        hits |= ubfe(empties, c, 1) << i;
    }
    // NOTE: the chosen entry is extracted with 64-bit shifts, not by indexing perm[] with a runtime
    // index (nor a select chain, which hipcc turns back into one): a runtime-indexed private array is
    // promoted to LDS with a lane stride of 16 B and cost ~14 us per launch in k_step (measured).
    const uint64_t plo = (uint64_t)perm[0] | ((uint64_t)perm[1] << 32);
    const uint64_t phi = (uint64_t)perm[2] | ((uint64_t)perm[3] << 32);
    for (int n = 0; n < count; ++n) {
        if (hits == 0) break;
        const uint32_t i = (uint32_t)__builtin_ctz(hits);
        hits &= hits - 1u;
        const uint64_t half = (i & 8u) ? phi : plo;
        const uint32_t c = (uint32_t)(half >> (8u * (i & 7u))) & 0xFu;
        set_cell(b, c, 2u - ubfe(two_mask, c, 1));
    }
}

// ---- the same spawn from the row's INVERSE permutation ------------------------------------------
// (round 1's form of the plain rollout's spawn; the kernels now use spawn_table_dot below, and this one stays as the
// intermediate step of the host cross-check: probing == inverse-shift == dot products, tests/test_host_ops.py)
// inv[c] = position of cell c in the permutation row, laid out like the board (byte j of word k =
// cell 4k+j).  Then  hits = OR over empty cells c of (1 << inv[c])  needs no per-position probing: one
// SDWA shift per cell - byte j of the emptiness flags (0x80 / 0) shifted left by byte j of inv - and
// an OR tree (32 VALU instead of ~72).  The winning position is ctz(hits); its cell is perm[position].
ML_HD uint32_t shl_byte_by_byte(uint32_t flags, uint32_t amounts, int j) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t out;
    switch (j) { // v_lshlrev_b32 D = S1 << S0[4:0]: S0 = byte j of `amounts`, S1 = byte j of `flags`
    case 0: asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_0" : "=v"(out) : "v"(amounts), "v"(flags)); break;
    case 1: asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1" : "=v"(out) : "v"(amounts), "v"(flags)); break;
    case 2: asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:BYTE_2" : "=v"(out) : "v"(amounts), "v"(flags)); break;
    default: asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:BYTE_3" : "=v"(out) : "v"(amounts), "v"(flags)); break;
    }
    return out;
#else
    return ((flags >> (8 * j)) & 0xFFu) << ((amounts >> (8 * j)) & 0x1Fu);
#endif
}

// one tile (the step's spawn, count = 1); equals spawn_table(b, perm, two_mask, 1) when inv is the
// inverse of perm
// returns the emptiness mask before the spawn (bit i <=> the cell at position i of the row was empty)
ML_HD uint32_t spawn_table_inv(Board &b, const uint32_t perm[4], const uint32_t inv[4], uint32_t two_mask) {
    uint32_t acc = 0; // bit 7 + i  <=>  the cell at position i of the row is empty
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t z = z_bytes(b.r[k]);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc |= shl_byte_by_byte(z, inv[k], j);
    }
    const uint32_t hits = acc >> 7;
    // A full board (hits == 0: cannot follow a valid move) is left alone.  ML2048_LONE_WAVE (the compilation pass for
    // small M, one wave per SIMD): by OR-ing a zero tile in - two selects instead of a divergent branch, whose
    // exec-mask bookkeeping costs a lone wave more (-1.7 % together with the scalar branch in merged_value_sum);
    // otherwise by the branch, which is 1.4 % cheaper at 1 M boards where VALU issue is the limit.
#ifndef ML2048_LONE_WAVE
    if (hits == 0) return 0u;
#endif
    const uint32_t i = hits ? (uint32_t)__builtin_ctz(hits) : 0u;
    const uint64_t plo = (uint64_t)perm[0] | ((uint64_t)perm[1] << 32);
    const uint64_t phi = (uint64_t)perm[2] | ((uint64_t)perm[3] << 32);
    const uint64_t half = (i & 8u) ? phi : plo;
    const uint32_t c = (uint32_t)(half >> (8u * (i & 7u))) & 0xFu;
    set_cell(b, c, hits ? 2u - ubfe(two_mask, c, 1) : 0u);
    return hits;
}

// inverse of one permutation row, in board layout (what the schedule upload computes per row)
ML_HD void invert_perm_row(const uint32_t perm[4], uint32_t inv[4]) {
    inv[0] = inv[1] = inv[2] = inv[3] = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint32_t c = ubfe(perm[i >> 2], 8u * (i & 3), 4);
        const uint32_t v = (uint32_t)i << ((c & 3u) * 8u);
        const uint32_t d = c >> 2;
        inv[0] |= (d == 0u) ? v : 0u;
        inv[1] |= (d == 1u) ? v : 0u;
        inv[2] |= (d == 2u) ? v : 0u;
        inv[3] |= (d == 3u) ? v : 0u;
    }
}

// ---- the same spawn as four dot products --------------------------------------------------------------------------
// hits = sum over the empty cells c of 2^inv[c] is a dot product of the emptiness flags with per-cell weights.  A
// weight has to fit a byte for v_dot4_u32_u8, so the row carries two weight vectors in board layout (byte j of word k =
// cell 4k+j): wlo[c] = 2^inv[c] if inv[c] < 8 else 0, whi[c] = 2^(inv[c]-8) if inv[c] >= 8 else 0; with the flags
// as 0x80 / 0 the two sums are the low and the high byte of hits, times 128.  Eight dot instructions instead of sixteen
// SDWA shifts and an OR tree.  The permutation itself travels as sixteen nibbles (pn_lo | pn_hi << 32, nibble i = the
// cell at position i), so the winning cell is one 64-bit shift away.  This is the EXPANDED ROW the upload kernels and
// the device schedule produce (expand_perm_row), 48 bytes: {pn_lo, pn_hi, 0, 0, wlo[4], whi[4]}.
constexpr int kExpandedRowWords = 12;
ML_HD uint32_t udot4(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_udot4(a, b, c, false);
#else
    for (int j = 0; j < 4; ++j) c += ((a >> (8 * j)) & 0xFFu) * ((b >> (8 * j)) & 0xFFu);
    return c;
#endif
}

ML_HD void expand_perm_row(const uint32_t perm[4], uint32_t out[kExpandedRowWords]) {
    for (int k = 0; k < kExpandedRowWords; ++k) out[k] = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const uint32_t c = ubfe(perm[i >> 2], 8u * (i & 3), 4);
        out[i >> 3] |= c << (4u * (i & 7));
        const uint32_t w = (1u << (i & 7)) << ((c & 3u) * 8u);
        const uint32_t d = c >> 2;
        const int base = i < 8 ? 4 : 8;
        out[base + 0] |= (d == 0u) ? w : 0u;
        out[base + 1] |= (d == 1u) ? w : 0u;
        out[base + 2] |= (d == 2u) ? w : 0u;
        out[base + 3] |= (d == 3u) ? w : 0u;
    }
}

// one tile (the step's spawn); equals spawn_table(b, perm, two_mask, 1) when the row is expand_perm_row(perm).
// Returns the emptiness mask before the spawn (bit i <=> the cell at position i of the row was empty).
// HAS_EMPTY: the caller knows that the board has an empty cell - it always has after a move that changed it (a slide
// keeps the number of empty cells, a merge adds one), which is the lazy loop's validity test; a caller that trusts a
// stored valid-action mask instead (which load_records lets a user make inconsistent) must pass false, and a full
// board is then left alone as _spawn2 leaves it.
template <bool HAS_EMPTY = false>
ML_HD uint32_t spawn_table_dot(Board &b, uint32_t pn_lo, uint32_t pn_hi, const uint32_t wlo[4], const uint32_t whi[4],
                               uint32_t two_mask) {
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t z = z_bytes(b.r[k]);
        lo = udot4(z, wlo[k], lo);
        hi = udot4(z, whi[k], hi);
    }
    const uint32_t hits = ((hi << 8) | lo) >> 7;
    // a full board (hits == 0: cannot follow a valid move) is left alone: a zero tile is OR-ed in (see spawn_table_inv)
#ifndef ML2048_LONE_WAVE
    if (!HAS_EMPTY && hits == 0) return 0u;
#endif
    const bool some = HAS_EMPTY || hits != 0u;
    const uint32_t i = some ? (uint32_t)__builtin_ctz(hits) : 0u;
    // nibble i of the permutation (32-bit forms: the 64-bit shifts are multi-pass instructions)
    const uint32_t c = ubfe((i & 8u) ? pn_hi : pn_lo, 4u * i, 4); // v_bfe_u32 takes the offset modulo 32
    const uint32_t val = some ? 2u - ubfe(two_mask, c, 1) : 0u;
    const uint64_t t = (uint64_t)val << (8u * (c & 7u)); // the tile in its half of the board
    const bool upper = (c & 8u) != 0u;
    const uint32_t tl = (uint32_t)t, th = (uint32_t)(t >> 32);
    b.r[0] |= upper ? 0u : tl;
    b.r[1] |= upper ? 0u : th;
    b.r[2] |= upper ? tl : 0u;
    b.r[3] |= upper ? th : 0u;
    return hits;
}

// the re-deal from the head of an expanded row (nibbles 0 and 1 of pn_lo): each tile is shifted into place inside
// its half of the board as a 64-bit value (two v_lshlrev_b64 and four selects instead of two set_cell chains - the deal
// runs in ~40 % of a wave's steps, whenever one of its 64 games has ended)
ML_HD void deal_table_nibbles(Board &b, uint32_t pn_lo, uint32_t two_mask) {
    const uint32_t c1 = pn_lo & 0xFu, c2 = (pn_lo >> 4) & 0xFu;
    const uint64_t t1 = (uint64_t)(2u - ubfe(two_mask, c1, 1)) << (8u * (c1 & 7u));
    const uint64_t t2 = (uint64_t)(2u - ubfe(two_mask, c2, 1)) << (8u * (c2 & 7u));
    const uint64_t lo = ((c1 & 8u) ? 0ull : t1) | ((c2 & 8u) ? 0ull : t2);
    const uint64_t hi = ((c1 & 8u) ? t1 : 0ull) | ((c2 & 8u) ? t2 : 0ull);
    b.r[0] = (uint32_t)lo;
    b.r[1] = (uint32_t)(lo >> 32);
    b.r[2] = (uint32_t)hi;
    b.r[3] = (uint32_t)(hi >> 32);
}

// the two tiles of a freshly dealt (empty) board: the first two entries of the permutation row
// (VecGame.prepare's `_spawn2(..., 2)` on a zeroed record, game_numba.py:639-655)
ML_HD void deal_table(Board &b, const uint32_t perm[4], uint32_t two_mask) {
    const uint32_t c1 = perm[0] & 0xFu, c2 = (perm[0] >> 8) & 0xFu;
    b.r[0] = b.r[1] = b.r[2] = b.r[3] = 0;
    set_cell(b, c1, 2u - ubfe(two_mask, c1, 1));
    set_cell(b, c2, 2u - ubfe(two_mask, c2, 1));
}

// ---- counter-based spawn (spawn="rocrand"; build-defined, equals oracle orc_step_fast / orc_prepare_fast) ---------
// ONE 32-bit word of the board's Philox sequence per spawned tile: its high half picks the k-th empty cell in ascending
// order, k = (hi16 * n_empty) >> 16 (uniform over the empty cells up to 2^-16), its low half decides 2 vs 4 against
// ceil(two_prob * 2^16).  A step consumes word c of the board's step stream (c = the env's step-call counter), so one
// Philox block serves four consecutive steps - a fused rollout computes it once per four steps; the re-deal consumes
// words 2c and 2c+1 of the deal stream (same key with kDealSeedXor).
constexpr unsigned long long kDealSeedXor = 0x9E3779B97F4A7C15ull;
// returns the emptiness mask before the spawn (bit c <=> cell c was empty; one bit set <=> the board is full now)
ML_HD uint32_t spawn_word(Board &b, uint32_t w, uint32_t thr16) {
    const uint32_t empties = empty_mask(b);
    uint32_t n_empty = (uint32_t)__builtin_popcount(empties);
    if (n_empty == 0) return 0u;
    uint32_t k = ((w >> 16) * n_empty) >> 16;
    // position of the k-th (0-based) set bit of the 16-bit mask, by branch-free halving
    uint32_t m = empties, c = 0, cnt;
    cnt = (uint32_t)__builtin_popcount(m & 0xFFu);
    if (k >= cnt) { k -= cnt; c += 8; m >>= 8; }
    cnt = (uint32_t)__builtin_popcount(m & 0xFu);
    if (k >= cnt) { k -= cnt; c += 4; m >>= 4; }
    cnt = (uint32_t)__builtin_popcount(m & 0x3u);
    if (k >= cnt) { k -= cnt; c += 2; m >>= 2; }
    cnt = m & 1u;
    if (k >= cnt) c += 1;
    set_cell(b, c, ((w & 0xFFFFu) < thr16) ? 1u : 2u);
    return empties;
}

// The two spawns of a re-deal in the counter-based mode (spawn_word twice on an empty board) as an index into the table
// of dealt boards that the fused rollout keeps in LDS (deal_table_nibbles' index: c1 | c2 << 4 | "tile 1 is a 2" << 8 |
// "tile 2 is a 2" << 9).  On an empty board the first word's cell is its top four bits (k = (hi16 * 16) >> 16), the
// second picks the k-th of the remaining fifteen cells.  ~15 VALU instead of two k-th-empty-cell selections (~120).
ML_HD uint32_t deal_words_index(uint32_t w0, uint32_t w1, uint32_t thr16) {
    const uint32_t c1 = w0 >> 28;
    const uint32_t k2 = ((w1 >> 16) * 15u) >> 16;
    const uint32_t c2 = k2 + (k2 >= c1 ? 1u : 0u);
    const uint32_t b1 = (w0 & 0xFFFFu) < thr16 ? 0x100u : 0u, b2 = (w1 & 0xFFFFu) < thr16 ? 0x200u : 0u;
    return c1 | (c2 << 4) | b1 | b2;
}

ML_HD uint32_t max_cell(const Board &b) {
    uint32_t mx = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t v = (b.r[i] >> (8 * j)) & 0xFFu;
            mx = v > mx ? v : mx;
        }
    return mx;
}

enum { REWARD_NORMAL = 0, REWARD_IMPROVED = 1, REWARD_RANK = 2, REWARD_MAXCELL = 3 };

// reference reward functions at numba's typing (int64 sums, then float64), game_numba.py:408-504
ML_HD double reward_value(int kind, const Board &state, const Board &prev, const MergeStats &s) {
    switch (kind) {
    default:
    case REWARD_NORMAL:
        return (double)s.sum_value;
    case REWARD_IMPROVED: {
        uint32_t p0 = prev.r[0] & 0xFFu, s0 = state.r[0] & 0xFFu;
        int64_t extra = 0;
        if (p0 != 0) extra -= (int64_t)64 << p0;
        if (s0 != 0) extra += (int64_t)64 << s0;
        return (double)s.sum_value + (double)extra;
    }
    case REWARD_RANK:
        return (double)s.sum_rank;
    case REWARD_MAXCELL: {
        uint32_t cur = max_cell(state), prv = max_cell(prev);
        double r = 0.0;
        if (cur > prv) r += (double)((int64_t)1 << cur);
        return r + (double)s.count;
    }
    }
}

// ---- one board of the reference's _vec_step loop body (game_numba.py:716-738) ------------------
// Split around the spawn so that the table-driven and counter-based spawns share it.
// Returns false for an invalid move: the caller then writes only invalid=1 (hazard H3).
struct StepRegs {
    Board board;
    uint32_t valid;      // valid_actions word (4 x 0/1 bytes)
    int32_t step;
    float score;
    float reward;        // written only on a valid move
    uint32_t merged[4];  // written only on a valid move
    uint32_t terminated; // written only on a valid move
};

// score accumulates reward_fn_normal whatever reward_fn is (:729-731).  The reference computes f32 + f64 -> f64 -> f32
// (H8).  The added value is an integer < 2^22 (eight merges of at most 2^18) and the score a float32, so the float64 sum
// is exact and its conversion is the one rounding of the exact sum - which is what a float32 addition returns (RNE).
// One v_add_f32 instead of two conversions, a 64-bit add and a conversion back.
ML_HD float add_score(float score, uint32_t sum_value) { return score + (float)sum_value; }

ML_HD bool step_pre(StepRegs &s, uint32_t action, int reward_kind) {
    // H11: an out-of-range action is treated as invalid (the reference would read out of bounds)
    bool ok = action < 4u && ((s.valid >> (8u * (action & 3u))) & 0xFFu) != 0u;
    if (!ok) return false;
    Board prev = s.board;
    MoveResult mv = move_board(prev, action);
    MergeStats st = merge_stats(mv.mexp_first, mv.mexp_second);
    s.board = mv.board;
    s.merged[0] = st.merged[0];
    s.merged[1] = st.merged[1];
    s.merged[2] = st.merged[2];
    s.merged[3] = st.merged[3];
    s.step += 1;
    s.reward = (float)reward_value(reward_kind, mv.board, prev, st);
    s.score = add_score(s.score, st.sum_value);
    return true;
}

// step_pre without the bucket array: s.merged is NOT touched; the caller keeps (mexp_first,
// mexp_second) and derives merged[16] with merge_buckets() when it is actually needed.
ML_HD bool step_pre_lite(StepRegs &s, uint32_t action, int reward_kind, uint32_t &mexp_first,
                         uint32_t &mexp_second) {
    bool ok = action < 4u && ((s.valid >> (8u * (action & 3u))) & 0xFFu) != 0u;
    if (!ok) return false;
    Board prev = s.board;
    MoveResult mv = move_board(prev, action);
    MergeStats st;
    merge_sums(mv.mexp_first, mv.mexp_second, reward_kind, st.sum_value, st.sum_rank, st.count);
    mexp_first = mv.mexp_first;
    mexp_second = mv.mexp_second;
    s.board = mv.board;
    s.step += 1;
    s.reward = (float)reward_value(reward_kind, mv.board, prev, st);
    s.score = add_score(s.score, st.sum_value);
    return true;
}

// step_pre_lite without any reward arithmetic (see step_pre_lazy_pv below): validity from the maintained mask, the
// caller keeps the board and the action of the last valid move
ML_HD bool step_pre_lite_pv(StepRegs &s, uint32_t action, Board &pv, uint32_t &pv_action) {
    const bool ok = action < 4u && ((s.valid >> (8u * (action & 3u))) & 0xFFu) != 0u;
    if (!ok) return false;
    pv = s.board;
    pv_action = action;
    s.board = move_board(pv, action).board;
    s.step += 1;
    return true;
}

// ---- the same step without a maintained valid-action mask ---------------------------------------
// For uniform (unmasked) sampling nothing reads the mask between steps: an action is valid exactly
// when the move changes the board, and `terminated` (= no valid action, :735-736) only needs "board
// full and no two equal neighbours" - a board with a tile and an empty cell always has a move.  A
// fused rollout may therefore skip valid_actions() every step and derive valid / prev_valid once at
// the end, PROVIDED the stored mask is consistent with the stored board when it starts
// (valid_consistent()); callers fall back to the eager form otherwise.
ML_HD bool valid_consistent(const Board &b, uint32_t valid) { return valid_actions(b) == valid; }

ML_HD bool board_differs(const Board &a, const Board &b) {
    uint32_t d = (a.r[0] ^ b.r[0]) | (a.r[1] ^ b.r[1]) | (a.r[2] ^ b.r[2]) | (a.r[3] ^ b.r[3]);
#if defined(__HIP_DEVICE_COMPILE__) && defined(ML2048_LONE_WAVE)
    // keep it one value and one compare: the compiler otherwise tests the four words separately and joins the
    // results with scalar ORs (4 v_cmp + 3 s_or instead of 3-4 VALU + 1 v_cmp) - cheaper when other waves fill the
    // scalar slots (1 M boards: +1.4 % with the barrier), dearer for a lone wave
    asm("" : "+v"(d));
#endif
    return d != 0u;
}

// == (valid_actions(b) == 0) for every board with at least one tile
ML_HD uint32_t no_move_left(const Board &b) {
    const uint32_t all_nz = (b.r[0] + 0x7F7F7F7Fu) & (b.r[1] + 0x7F7F7F7Fu) & (b.r[2] + 0x7F7F7F7Fu) &
                            (b.r[3] + 0x7F7F7F7Fu) & 0x80808080u;
    uint32_t eq = 0; // 0x80 wherever two neighbours hold the same byte (only meaningful on a full board)
#pragma unroll
    for (int i = 0; i < 4; ++i) eq |= z_bytes(b.r[i] ^ (b.r[i] >> 8)) & 0x00808080u;
#pragma unroll
    for (int i = 0; i < 3; ++i) eq |= z_bytes(b.r[i] ^ b.r[i + 1]);
    return (all_nz == 0x80808080u && eq == 0u) ? 1u : 0u;
}

// step_pre_lite whose validity test is "the move changes the board" (s.valid is neither read nor
// written); pair with step_post_lazy
ML_HD bool step_pre_lazy(StepRegs &s, uint32_t action, int reward_kind, uint32_t &mexp_first,
                         uint32_t &mexp_second) {
    const Board prev = s.board;
    const MoveResult mv = move_board(prev, action & 3u);
    if (action >= 4u || !board_differs(mv.board, prev)) return false;
    MergeStats st;
    merge_sums(mv.mexp_first, mv.mexp_second, reward_kind, st.sum_value, st.sum_rank, st.count);
    mexp_first = mv.mexp_first;
    mexp_second = mv.mexp_second;
    s.board = mv.board;
    s.step += 1;
    s.reward = (float)reward_value(reward_kind, mv.board, prev, st);
    s.score = add_score(s.score, st.sum_value);
    return true;
}

// ---- the score as a function of the board ------------------------------------------------------------------------
// `score` accumulates reward_fn_normal = the value of every tile a merge creates (game_numba.py:408-438, :729-731).
// A tile 2^e that was built from 2-tiles alone has collected (e-1)*2^e on the way; every 4 that was spawned instead of
// being merged from two 2s saves one reward of 4.  The 4s follow from the board too: T tiles were spawned since the deal
// (2 + one per valid move = 2 + step), worth sum(2^e) = 2*(T - B) + 4*B together, B of them 4s.  Hence, exactly,
//     score = sum over tiles of (e-3)*2^e  +  8  +  4*step                       (verified move by move against the
// reference-derived oracle on 49 000 games, tests/test_host_ops.py::test_score_is_a_function_of_board_and_step)
// and a rollout that only has to deliver the score AFTER its last step needs no reward arithmetic inside its loop:
// score_end = score_start - potential(start) + potential(end), restarting from 0 at a re-deal.  Integers below 2^24
// throughout (phi_guard), so the float32 running sum the reference keeps (H8) holds the same exact integers.
ML_HD int32_t score_potential(const Board &b, int32_t step) {
    int32_t sum = 8 + 4 * step;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int32_t e = (int32_t)((b.r[q] >> (8 * j)) & 0xFFu);
            sum += e ? (e - 3) * (int32_t)(1u << (e & 31)) : 0;
        }
    }
    return sum;
}
// sum of the tiles on the board (exponents <= 31)
ML_HD uint32_t tile_sum(const Board &b) {
    uint32_t sum = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int j = 0; j < 4; ++j) sum += (1u << ((b.r[q] >> (8 * j)) & 31u)) & ~1u; // an empty cell (1 << 0) adds nothing
    }
    return sum;
}
// When may a rollout of n_steps steps take its score from the board?  The identity above counts EVERY merge, but the
// reference drops the reward of a merge of two 2^16 (or 2^17) tiles: `buckets[num] += 1` with num = 16 lands outside
// merged[16] (H9) and reward_fn_normal never sees it.  So no 2^17 tile may appear during the rollout: a valid move adds
// one spawned tile (<= 4) to the tile sum and merges keep it, hence tile_sum + 4 * n_steps < 2^17 rules it out (and with it
// every exponent above 16).  Tiles <= 2^15 now and an integer score in [0, 2^21) keep every intermediate value an exact
// integer below 2^24 (potential <= 13 * 2^17), as the float32 running sum of the reference (H8) holds them.
// Round 2's guard only bounded the values (tiles <= 2^15, <= 65 536 steps) and was wrong for loaded states that reach two
// 2^16 tiles inside the rollout (tests/test_gpu_plain_rollout.py::test_plain_large_pass_on_boards_outside_the_table_guard
// found it); play from a deal never gets near (a 2^17 tile needs a tile sum of 131 072).
ML_HD bool phi_guard(const Board &b, float score, int32_t n_steps) {
    const bool small = ((b.r[0] | b.r[1] | b.r[2] | b.r[3]) & 0xF0F0F0F0u) == 0u;
    const bool no_dropped_reward = n_steps >= 0 && n_steps < 32768 && tile_sum(b) + 4u * (uint32_t)n_steps < 131072u;
    return small && no_dropped_reward && score >= 0.0f && score < 2097152.0f && (float)(int32_t)score == score;
}
// step_pre_lazy without any reward arithmetic: the caller keeps the board and the action of the last VALID move
// (`pv`, `pv_action`) and restores score (score_potential), reward and merged[] (one more move_board on `pv`) after its
// last step - the record only ever shows the last valid move's reward and buckets (H3)
ML_HD bool step_pre_lazy_pv(StepRegs &s, uint32_t action, Board &pv, uint32_t &pv_action) {
    const Board prev = s.board;
    const MoveResult mv = move_board(prev, action & 3u);
    if (action >= 4u || !board_differs(mv.board, prev)) return false;
    pv = prev;
    pv_action = action;
    s.board = mv.board;
    s.step += 1;
    return true;
}

// the two steps above with the move looked up (move_board_lut; callers establish lut_guard first)
template <bool SWZ = false, class LutPtr>
ML_HD bool step_pre_lazy_pv_lut(StepRegs &s, uint32_t action, Board &pv, uint32_t &pv_action, LutPtr lut) {
    const Board prev = s.board;
    const Board nb = move_board_lut<SWZ>(prev, action & 3u, lut);
    if (action >= 4u || !board_differs(nb, prev)) return false;
    pv = prev;
    pv_action = action;
    s.board = nb;
    s.step += 1;
    return true;
}
template <bool SWZ = false, class LutPtr, class SelPtr>
ML_HD bool step_pre_lazy_pv_lut(StepRegs &s, uint32_t action, Board &pv, uint32_t &pv_action, LutPtr lut, SelPtr sel) {
    const Board prev = s.board;
    const Board nb = move_board_lut_sel<SWZ>(prev, action, lut, sel);
    if (action >= 4u || !board_differs(nb, prev)) return false;
    pv = prev;
    pv_action = action;
    s.board = nb;
    s.step += 1;
    return true;
}
template <bool SWZ = false, class LutPtr, class SelPtr>
ML_HD bool step_pre_lite_pv_lut(StepRegs &s, uint32_t action, Board &pv, uint32_t &pv_action, LutPtr lut, SelPtr sel) {
    const bool ok = action < 4u && ((s.valid >> (8u * (action & 3u))) & 0xFFu) != 0u;
    if (!ok) return false;
    pv = s.board;
    pv_action = action;
    s.board = move_board_lut_sel<SWZ>(pv, action, lut, sel);
    s.step += 1;
    return true;
}
template <bool SWZ = false, class LutPtr>
ML_HD bool step_pre_lite_pv_lut(StepRegs &s, uint32_t action, Board &pv, uint32_t &pv_action, LutPtr lut) {
    const bool ok = action < 4u && ((s.valid >> (8u * (action & 3u))) & 0xFFu) != 0u;
    if (!ok) return false;
    pv = s.board;
    pv_action = action;
    s.board = move_board_lut<SWZ>(pv, action, lut);
    s.step += 1;
    return true;
}

ML_HD void step_post_lazy(StepRegs &s) { s.terminated = no_move_left(s.board); }

// the same when the spawn reported which cells were empty before it (spawn_table_inv): the board is full now exactly
// when one cell was empty then, which replaces the zero test over all 16 cells
ML_HD void step_post_lazy(StepRegs &s, uint32_t empties_before_spawn) {
    const Board &b = s.board;
    uint32_t eq = 0; // 0x80 wherever two neighbours hold the same byte (only meaningful on a full board)
#pragma unroll
    for (int i = 0; i < 4; ++i) eq |= z_bytes(b.r[i] ^ (b.r[i] >> 8)) & 0x00808080u;
#pragma unroll
    for (int i = 0; i < 3; ++i) eq |= z_bytes(b.r[i] ^ b.r[i + 1]);
    const bool full = (empties_before_spawn & (empties_before_spawn - 1u)) == 0u;
    s.terminated = (full && eq == 0u) ? 1u : 0u;
}

ML_HD void step_post(StepRegs &s) {
    s.valid = valid_actions(s.board);
    s.terminated = s.valid == 0u;
}

// a freshly dealt game (VecGame.prepare's loop body, game_numba.py:638-656) before its two spawns
ML_HD void clear_regs(StepRegs &s) {
    s.board.r[0] = s.board.r[1] = s.board.r[2] = s.board.r[3] = 0;
    s.merged[0] = s.merged[1] = s.merged[2] = s.merged[3] = 0;
    s.valid = 0;
    s.step = 0;
    s.score = 0.0f;
    s.reward = 0.0f;
    s.terminated = 0;
}

// ---- Philox4x32-10 (hand-written form; the kernels' default path uses the rocRAND device API,
//      which tests/test_gpu_parity.py checks word-for-word against this and the oracle) --------
struct U4 {
    uint32_t x, y, z, w;
};

ML_HD U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c.x;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c.z;
        U4 n;
        n.x = (uint32_t)(p1 >> 32) ^ c.y ^ k0;
        n.y = (uint32_t)p1;
        n.z = (uint32_t)(p0 >> 32) ^ c.w ^ k1;
        n.w = (uint32_t)p0;
        c = n;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

ML_HD U4 philox_at(uint64_t seed, uint64_t counter, uint64_t idx) {
    U4 c = {(uint32_t)counter, (uint32_t)(counter >> 32), (uint32_t)idx, (uint32_t)(idx >> 32)};
    return philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
}

// The action sampler's random stream of board idx is rocRAND's Philox4x32-10 sequence W(0), W(1), ... of
// subsequence idx (W(j) = word j & 3 of Philox block j >> 2, i.e. rocrand_init(seed, idx, j)).
//   masked-uniform draw of call-counter t: the whole word W(t) (multiply-shift by the number of valid actions);
//   plain uniform draw (4 actions, 2 bits): bits 2(t & 15).. of W(t >> 4) - sixteen draws per word, so a fused
//   rollout evaluates Philox once per 64 steps instead of once per 4 (its ten 64-bit multiply rounds were ~6 %
//   of the rollout kernel).
// sampler_word returns the word with the draw's bits at the bottom: pick_action takes it from there.
ML_HD uint32_t philox_seq_word(uint64_t seed, uint64_t j, uint64_t idx) {
    U4 r = philox_at(seed, j >> 2, idx);
    const uint32_t k = (uint32_t)(j & 3u);
    return k == 0 ? r.x : k == 1 ? r.y : k == 2 ? r.z : r.w;
}
ML_HD uint32_t sampler_word(uint64_t seed, uint64_t t, uint64_t idx, int masked) {
    return masked ? philox_seq_word(seed, t, idx) : philox_seq_word(seed, t >> 4, idx) >> (2u * (uint32_t)(t & 15u));
}

// masked-uniform / uniform action from one random word (oracle orc_sample_actions)
ML_HD uint32_t pick_action(uint32_t valid_word, uint32_t word, int masked, uint32_t &cnt_out) {
    // branch-free on purpose: in the fused rollout a taken branch costs an instruction-buffer refill
    uint32_t bits = flags4(nz_bytes(valid_word));
    const uint32_t cnt = (uint32_t)__builtin_popcount(bits);
    const uint32_t k = (uint32_t)(((uint64_t)word * cnt) >> 32);
    bits = k > 0 ? bits & (bits - 1u) : bits; // drop the k lowest set bits
    bits = k > 1 ? bits & (bits - 1u) : bits;
    bits = k > 2 ? bits & (bits - 1u) : bits;
    const uint32_t kth = bits ? (uint32_t)__builtin_ctz(bits) : 0u; // cnt == 0 -> action 0
    cnt_out = masked ? cnt : 4u;
    return masked ? kth : (word & 3u);
}

ML_HD float neg_log_count(uint32_t cnt) {
    // -ln(cnt) as float32 literals (shared with the oracle so CPU and GPU agree bit for bit)
    return cnt == 4 ? -1.38629436111989f : cnt == 3 ? -1.09861228866811f
         : cnt == 2 ? -0.693147180559945f : 0.0f;
}

// 64-bit order-independent per-board digest (oracle orc_digest)
ML_HD uint64_t mix_word(uint64_t h, uint64_t w) {
    h = (h ^ w) * 0xBF58476D1CE4E5B9ull;
    h ^= h >> 29;
    return h;
}

// id < 0 in `with_id == 0` calls is simply not mixed in
ML_HD uint64_t board_digest(uint64_t global_idx, const Board &b, const uint32_t merged[4], uint32_t valid,
                            uint32_t terminated, uint32_t invalid, int32_t step, int32_t id,
                            uint32_t score_bits, uint32_t reward_bits, int with_id = 1) {
    uint64_t h = 0x9E3779B97F4A7C15ull * (global_idx + 1);
    h = mix_word(h, (uint64_t)b.r[0] | ((uint64_t)b.r[1] << 32));
    h = mix_word(h, (uint64_t)b.r[2] | ((uint64_t)b.r[3] << 32));
    h = mix_word(h, (uint64_t)merged[0] | ((uint64_t)merged[1] << 32));
    h = mix_word(h, (uint64_t)merged[2] | ((uint64_t)merged[3] << 32));
    h = mix_word(h, (uint64_t)valid | ((uint64_t)(terminated & 0xFFu) << 32) | ((uint64_t)(invalid & 0xFFu) << 40));
    h = mix_word(h, (uint64_t)(uint32_t)step);
    h = mix_word(h, (uint64_t)score_bits | ((uint64_t)reward_bits << 32));
    if (with_id) h = mix_word(h, (uint64_t)(uint32_t)id);
    return h;
}

} // namespace ml2048
