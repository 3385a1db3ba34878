// kf_pingpong_tn.h -- PREPARED FOR ROUND 5, NOT PART OF THE LIBRARY, never run on a GPU (tools/next/README.md).
//
// The 256 x 256 x 64 wave-role-split main loop of kronfluence_amd/csrc/kf_pingpong.h (request schedule ISSUE = 1) for K-MAJOR
// operands: both operands are given as X[k][row] (the contraction index is the slow axis: the hooked [t][feature] rows of a
// sequence layer), C[m][n] += sum_k A[k][m] B[k][n].  Phases, barriers, the four pieces per k-tile, their liveness, the counted
// waits and the ordering argument are those of kf_pingpong.h (tools/pp_schedule_check.py); what differs is
//
//   the LDS image   a 16 KB piece = 128 rows (features) x 64 k as 64 rows of 256 bytes, one per k: exactly what an LDS-DMA
//                   request writes lane-linearly when its 64 lanes fetch 4 k x 256 contiguous bytes (fully coalesced).  The
//                   64-byte quarter q of the row of k sits at quarter q ^ (k & 3):
//                       piece + k * 256 + (((f >> 5) ^ (k & 3)) * 64) + (f & 31) * 2            f = piece-local row 0..127
//                   so that the four k rows one 32-lane half of a transposing read touches fall into the four bank quarters.
//   the fragments   ds_read_b64_tr_b16 (ASSUMED semantics, to be confirmed by tools/next/tr_probe.py: within a 16-lane group
//                   lane l receives element l & 3 of the 64-bit words addressed by lanes (l >> 2) + 4 j, j = 0..3): lane s of
//                   group g addresses (k = kk * 16 + 8 (g >> 1) + 4 quad + (s >> 2), f = f0 + 16 (g & 1) + 4 (s & 3)); two reads
//                   (quad 0, 1) give the lane the 8 consecutive k of row f0 + (lane & 31) that v_mfma_f32_32x32x16_bf16 wants.
//
// Pieces (piece-local row fl of tile row f):   A0: rows of blocks i = 0, 1 of both wave rows, fl = (f >> 7) * 64 + (f & 63), f & 64 == 0
//                                              A1: blocks i = 2, 3, same fl, f & 64 != 0        B0: f < 128, fl = f        B1: fl = f - 128
// Requests: a piece is 16 requests of 4 k rows; wave w issues requests w and w + 8 of every piece (8 per k-tile, as before).
// Lane j of request q fetches k = 4 q + (j >> 4) and the 16-byte chunk whose image position is j & 15 in that row: logical
// quarter ((j & 15) >> 2) ^ (k & 3), chunk j & 3 of it, i.e. fl = that quarter * 32 + (j & 3) * 8.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../kronfluence_amd/csrc/kf_engine.h"
#include "../../kronfluence_amd/csrc/kf_pingpong.h"
#include "kf_tn_map.h"

namespace kf {
namespace pptn {

using pp::bf16x8;
using pp::barrier;
using pp::glds16;
using pp::wait_lds_reads;
using pp::wait_vmcnt;

constexpr int THREADS = 512;
using tnmap::PIECE_BYTES;
using tnmap::STAGE_BYTES;
constexpr int SMEM_BYTES = 2 * STAGE_BYTES;

typedef short s16x4 __attribute__((ext_vector_type(4)));

// per-lane DMA sources for k-tile 0: p[2 piece + h] = request w + 8 h of the piece
struct Sources {
    const uint16_t* p[8];
};

// row_a(f) / row_b(f): address of (k = 0, tile row f) of the operand; ld_a / ld_b: elements between consecutive k.  Rows are
// fetched in chunks of 8: the functors get the chunk's first row (a multiple of 8) and clamp it themselves.
template <class RowA, class RowB>
__device__ __forceinline__ void make_sources(Sources& s, int wave, int lane, RowA row_a, int64_t ld_a, RowB row_b, int64_t ld_b) {
#pragma unroll
    for (int piece = 0; piece < 4; ++piece)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k = tnmap::dma_k(wave, h, lane);
            const int f = tnmap::tile_row(piece, tnmap::dma_row(wave, h, lane));
            s.p[2 * piece + h] = piece < 2 ? row_a(f) + k * ld_a : row_b(f) + k * ld_b;
        }
}

// acc[i][jn] += A^T B over k-tiles [0, nt); walk_a(t) / walk_b(t): element offset of k-tile t (= t * 64 * ld for plain operands).
template <class WalkA, class WalkB>
__device__ __forceinline__ void mainloop(f32x16 (&acc)[4][2], unsigned char* sm, const Sources& src, int nt, int wave, int lane,
                                         WalkA walk_a, WalkB walk_b) {
    const int wm = wave >> 2, wn = wave & 3;
    const unsigned char* piece_b = sm + tnmap::b_piece(wn) * PIECE_BYTES;

    auto issue_at = [&](int piece, int t, int64_t off) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
            glds16(src.p[2 * piece + h] + off, sm + (t & 1) * STAGE_BYTES + piece * PIECE_BYTES + tnmap::dma_base(wave, h));
    };
    auto issue_piece = [&](int piece, int t) { issue_at(piece, t, piece < 2 ? walk_a(t) : walk_b(t)); };

    bf16x8 a[2][4], b[2][4];
    auto read_tr = [&](const unsigned char* piece, int fl0, int kk) -> bf16x8 {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(piece + tnmap::word(fl0, kk, 0, lane)));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(piece + tnmap::word(fl0, kk, 1, lane)));
        union { s16x4 h[2]; bf16x8 v; } u;
        u.h[0] = lo; u.h[1] = hi;
        return u.v;
    };
    auto read_a = [&](int half, int buf) {   // half 0: blocks 0, 1 (piece A0), half 1: blocks 2, 3 (piece A1)
        const unsigned char* piece = sm + buf * STAGE_BYTES + tnmap::a_piece(2 * half) * PIECE_BYTES;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) a[i][kk] = read_tr(piece, tnmap::a_row(wm, 2 * half + i), kk);
    };
    auto read_b = [&](int buf) {
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) b[jn][kk] = read_tr(piece_b + buf * STAGE_BYTES, tnmap::b_row(wn, jn), kk);
    };
#define KF_TN_GROUP(HALF, KK)                                                                                          \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                      \
        _Pragma("unroll") for (int jn = 0; jn < 2; ++jn)                                                               \
            acc[(HALF) * 2 + i][jn] =                                                                                  \
                __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][KK], b[jn][KK], acc[(HALF) * 2 + i][jn], 0, 0, 0)
    auto ride = [&](bool on, int piece, int t, int64_t off) {
        if (on) {
            __builtin_amdgcn_sched_barrier(0);
            issue_at(piece, t, off);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    issue_piece(0, 0); issue_piece(2, 0); issue_piece(3, 0); issue_piece(1, 0);
    if (nt > 1) { issue_piece(0, 1); issue_piece(2, 1); issue_piece(3, 1); wait_vmcnt<8>(); }
    else wait_vmcnt<2>();
    barrier();
    if (wm == 1) barrier();

#define KF_TN_TILE(T, MORE1, MORE2)                                                                                    \
    do {                                                                                                               \
        const int t_ = (T), buf_ = t_ & 1;                                                                             \
        int64_t oa_ = 0, ob_ = 0;                                                                                      \
        if (MORE1) oa_ = walk_a(t_ + 1);                                                                               \
        read_a(0, buf_);                                                                                               \
        read_b(buf_);                                                                                                  \
        if (MORE1) wait_vmcnt<6>();                                                                                    \
        else wait_vmcnt<0>();                                                                                          \
        wait_lds_reads();                                                                                              \
        barrier();                                                                                                     \
        __builtin_amdgcn_s_setprio(1);                                                                                 \
        KF_TN_GROUP(0, 0);                                                                                             \
        ride(MORE1, 1, t_ + 1, oa_);                                                                                   \
        KF_TN_GROUP(0, 1);                                                                                             \
        KF_TN_GROUP(0, 2);                                                                                             \
        KF_TN_GROUP(0, 3);                                                                                             \
        __builtin_amdgcn_s_setprio(0);                                                                                 \
        barrier();                                                                                                     \
        if (MORE2) { oa_ = walk_a(t_ + 2); ob_ = walk_b(t_ + 2); }                                                     \
        read_a(1, buf_);                                                                                               \
        if (MORE1) wait_vmcnt<2>();                                                                                    \
        wait_lds_reads();                                                                                              \
        barrier();                                                                                                     \
        __builtin_amdgcn_s_setprio(1);                                                                                 \
        KF_TN_GROUP(1, 0);                                                                                             \
        ride(MORE2, 0, t_ + 2, oa_);                                                                                   \
        KF_TN_GROUP(1, 1);                                                                                             \
        ride(MORE2, 2, t_ + 2, ob_);                                                                                   \
        KF_TN_GROUP(1, 2);                                                                                             \
        ride(MORE2, 3, t_ + 2, ob_);                                                                                   \
        KF_TN_GROUP(1, 3);                                                                                             \
        __builtin_amdgcn_s_setprio(0);                                                                                 \
        barrier();                                                                                                     \
    } while (0)

    int t = 0;
    for (; t + 2 < nt; ++t) KF_TN_TILE(t, true, true);
    if (t + 1 < nt) { KF_TN_TILE(t, true, false); ++t; }
    KF_TN_TILE(t, false, false);
    if (wm == 0) barrier();
#undef KF_TN_TILE
#undef KF_TN_GROUP
}

}  // namespace pptn
}  // namespace kf
