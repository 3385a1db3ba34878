// kf_tn_map.h -- index arithmetic of the K-MAJOR ("TN") operand path of kf_pingpong_tn.h as plain functions that compile for
// the host too: tools/tn_map_check.cpp replays them on the CPU (tests/test_tools_cpu.py builds and runs it with g++).
//
// An operand tile of the 256 x 256 x 64 loop is staged in four 16 KB PIECES (kf_pingpong.h: A0, A1, B0, B1), each 128 tile rows
// ("features") x 64 k.  For K-major operands X[k][feature] (the hooked [t][feature] rows of a sequence layer) a 16-byte chunk of
// global memory is 8 FEATURES of one k, so the LDS image keeps such chunks whole and the fragment reads transpose:
// ds_read_b64_tr_b16 hands lane l of a 16-lane group element l & 3 of the 64-bit words addressed by lanes (l >> 2) + 4 j,
// j = 0..3 -- a [4 k][16 feature] block read column-wise when lane s addresses (k0 + (s >> 2), f0 + 4 (s & 3)).  Lane (g, s) of a
// wave (g = lane >> 4) reads
//       k = 16 kk + 8 (g >> 1) + 4 quad + (s >> 2),     feature = f0 + 16 (g & 1) + 4 (s & 3)
// so that two reads (quad 0, 1) give it the 8 consecutive k of tile row f0 + (lane & 31) at k-octet lane >> 5 -- the operand
// layout of v_mfma_f32_32x32x16_bf16.
//
// An LDS-DMA request writes 1 KB lane-linearly (lane j: 16 bytes at request base + 16 j) from per-lane global addresses; a piece
// is 16 requests, request q at q * 1024.  Three images (which 8 features x which k lane j of request q fetches); all three put
// the four k rows a 32-lane half of a transposing read touches into the four 64-byte quarters of the 256-byte bank space:
//   IMG 0   64 rows (k) of 256 bytes (128 features), the 64-byte quarter XORed with k & 3.  Request = 4 k rows x 256 contiguous
//           bytes of global memory each.
//   IMG 1   [feature / 32][k][32 features]: 64-byte rows in k order.  Request = 16 k rows x 64 bytes.
//   IMG 2   as IMG 1 with bits 2 and 3 of k swapped in the row index: the 8 rows one transposing read touches are 512 contiguous
//           bytes (the "[8-key][32-col] subtiles with key bits 2 / 3 swapped" of cdna_hip_programming.md T10).
#pragma once

#if defined(__HIPCC__)
#define KF_TN_HD __host__ __device__ __forceinline__
#else
#define KF_TN_HD inline
#endif

namespace kf {
namespace tnmap {

constexpr int PIECE_BYTES = 16384, STAGE_BYTES = 65536;   // piece p (0 A0, 1 A1, 2 B0, 3 B1) of k-tile t at (t & 1) * 64 KB + p * 16 KB
constexpr int REQUEST_BYTES = 1024;

// tile row (0..255 of the operand tile) of piece-local row fl (0..127)
KF_TN_HD int tile_row(int piece, int fl) { return piece < 2 ? (fl >> 6) * 128 + piece * 64 + (fl & 63) : (piece - 2) * 128 + fl; }

template <int IMG>
struct Image;

template <>
struct Image<0> {
    // byte offset inside a piece of element (k, fl)
    static KF_TN_HD int at(int k, int fl) { return k * 256 + ((((fl >> 5) ^ k) & 3) << 6) + (fl & 31) * 2; }
    // what lane `lane` of request q (0..15) fetches: its k and the first of its 8 piece-local rows
    static KF_TN_HD int dma_k(int q, int lane) { return 4 * q + (lane >> 4); }
    static KF_TN_HD int dma_row(int q, int lane) { return ((((lane & 15) >> 2) ^ (dma_k(q, lane) & 3)) << 5) + (lane & 3) * 8; }
};

template <>
struct Image<1> {
    static KF_TN_HD int at(int k, int fl) { return (fl >> 5) * 4096 + k * 64 + (fl & 31) * 2; }
    static KF_TN_HD int dma_k(int q, int lane) { return 16 * (q & 3) + (lane >> 2); }
    static KF_TN_HD int dma_row(int q, int lane) { return (q >> 2) * 32 + (lane & 3) * 8; }
};

template <>
struct Image<2> {
    static KF_TN_HD int krow(int k) { return (k & ~12) | ((k & 4) << 1) | ((k & 8) >> 1); }   // an involution
    static KF_TN_HD int at(int k, int fl) { return (fl >> 5) * 4096 + krow(k) * 64 + (fl & 31) * 2; }
    static KF_TN_HD int dma_k(int q, int lane) { return krow(16 * (q & 3) + (lane >> 2)); }
    static KF_TN_HD int dma_row(int q, int lane) { return (q >> 2) * 32 + (lane & 3) * 8; }
};

// the two requests of a piece wave w issues: q = w and q = w + 8
KF_TN_HD int request_of(int wave, int h) { return wave + 8 * h; }

// byte offset inside a piece of the 64-bit word lane `lane` addresses for (32-row block at piece-local row fl0, k-slab kk of 16,
// quad 0 / 1 = the first / last four of the lane's eight k)
template <int IMG>
KF_TN_HD int word(int fl0, int kk, int quad, int lane) {
    const int g = lane >> 4, s = lane & 15;
    return Image<IMG>::at(kk * 16 + 8 * (g >> 1) + 4 * quad + (s >> 2), fl0 + 16 * (g & 1) + 4 * (s & 3));
}

// word(f, kk, quad, lane) - word(f, 0, 0, lane): the same constant for every lane and block in all three images (checked by
// tools/tn_map_check.cpp) -- the immediate offset of the read instruction
template <int IMG>
constexpr int word_step(int kk, int quad) { return IMG == 0 ? (16 * kk + 4 * quad) * 256 : IMG == 1 ? (16 * kk + 4 * quad) * 64 : (16 * kk + 8 * quad) * 64; }

// piece-local first row of the blocks a wave reads: A block i (0..3) of wave row wm -> (piece i >> 1, row); B block jn of wave column wn
KF_TN_HD int a_piece(int i) { return i >> 1; }
KF_TN_HD int a_row(int wm, int i) { return wm * 64 + (i & 1) * 32; }
KF_TN_HD int b_piece(int wn) { return 2 + (wn >> 1); }
KF_TN_HD int b_row(int wn, int jn) { return (wn & 1) * 64 + jn * 32; }

}  // namespace tnmap
}  // namespace kf
