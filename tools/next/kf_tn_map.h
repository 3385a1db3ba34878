// kf_tn_map.h -- the index arithmetic of the K-major ("TN") operand path of kf_pingpong_tn.h, as plain functions that compile for
// the host too: tools/next/tn_map_check.cpp replays them on the CPU (tests/test_tools_cpu.py builds and runs it with g++).
// PREPARED FOR ROUND 5, not part of the library.
#pragma once

#if defined(__HIPCC__)
#define KF_TN_HD __host__ __device__ __forceinline__
#else
#define KF_TN_HD inline
#endif

namespace kf {
namespace tnmap {

constexpr int PIECE_BYTES = 16384, STAGE_BYTES = 65536;   // piece p (0 A0, 1 A1, 2 B0, 3 B1) of k-tile t at (t & 1) * 64 KB + p * 16 KB

// tile row (0..255 of the operand tile) of piece-local row fl (0..127)
KF_TN_HD int tile_row(int piece, int fl) { return piece < 2 ? (fl >> 6) * 128 + piece * 64 + (fl & 63) : (piece - 2) * 128 + fl; }

// byte offset inside a piece of element (k, fl): 256-byte rows, the 64-byte quarter XORed with k & 3
KF_TN_HD int image(int k, int fl) { return k * 256 + ((((fl >> 5) ^ k) & 3) << 6) + (fl & 31) * 2; }

// LDS-DMA request `wave + 8 h` of a piece, lane j: the k row and the first of the 8 piece-local rows it fetches, and where the
// request starts in the piece (the hardware adds 16 * lane)
KF_TN_HD int dma_k(int wave, int h, int lane) { return 4 * (wave + 8 * h) + (lane >> 4); }
KF_TN_HD int dma_row(int wave, int h, int lane) {
    return ((((lane & 15) >> 2) ^ (dma_k(wave, h, lane) & 3)) << 5) + (lane & 3) * 8;
}
KF_TN_HD int dma_base(int wave, int h) { return (wave + 8 * h) * 1024; }

// byte offset inside a piece of the 64-bit word lane `lane` addresses for (32-row block at piece-local row fl0, k-slab kk of 16,
// quad 0 / 1 = the first / last four of the lane's eight k)
KF_TN_HD int word(int fl0, int kk, int quad, int lane) {
    const int g = lane >> 4, s = lane & 15;
    const int k = kk * 16 + 8 * (g >> 1) + 4 * quad + (s >> 2);
    return k * 256 + ((((fl0 >> 5) ^ (s >> 2)) & 3) << 6) + (16 * (g & 1) + 4 * (s & 3)) * 2;
}

// piece-local first row of the blocks a wave reads: A block i (0..3) of wave row wm -> (piece i >> 1, row); B block jn of wave column wn
KF_TN_HD int a_piece(int i) { return i >> 1; }
KF_TN_HD int a_row(int wm, int i) { return wm * 64 + (i & 1) * 32; }
KF_TN_HD int b_piece(int wn) { return 2 + (wn >> 1); }
KF_TN_HD int b_row(int wn, int jn) { return (wn & 1) * 64 + jn * 32; }

}  // namespace tnmap
}  // namespace kf
