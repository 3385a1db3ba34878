// kf_pingpong_tn.h -- the 256 x 256 x 64 wave-role-split main loop of kf_pingpong.h (request schedule ISSUE = 1) for K-MAJOR
// operands (round 5): both operands are X[k][row] -- the contraction index is the slow axis, as in the hooked [t][feature] rows of
// a Linear layer on sequences -- and C[m][n] += sum_k A[k][m] B[k][n].  The K-contiguous loop needs two transposed copies of the
// hooked tensors per call (transpose_rows_kernel: 22 % of a GPT-2 score call, 0.8 of the 3.0 GB it moves); this one reads them
// where autograd left them.
//
// Phases, barriers, the four pieces per k-tile, their liveness, the request schedule, the counted waits and the RAW / WAR argument
// are those of kf_pingpong.h, ISSUE = 1 (tools/pp_schedule_check.py models them; nothing about WHEN a piece is written or read
// changes).  What differs is WHERE the bytes of a piece sit and how a fragment is read:
//
//   the LDS image   keeps 16-byte global chunks (8 rows of one k) whole: kf_tn_map.h, three candidate images (template IMG);
//                   a request is still 64 lanes x 16 bytes written lane-linearly, 16 requests per 16 KB piece, wave w issues
//                   requests w and w + 8 of every piece (8 per wave and k-tile, as before);
//   the fragments   two ds_read_b64_tr_b16 per bf16x8 operand fragment (32 / 16 LDS instructions in the two L segments instead
//                   of 16 / 8, the same LDS-array time: 2 cycles per 8-byte read against 4 per 16-byte read).
//
// The transposing read returns the 8-byte-aligned word's data whatever the low address bits (cdna_hip_programming.md G17): the
// piece base must be 16-byte aligned -- the kernels carve their dynamic LDS from offset 0 and have no static __shared__.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kf_engine.h"
#include "kf_pingpong.h"
#include "kf_tn_map.h"

namespace kf {
namespace pptn {

using pp::barrier;
using pp::bf16x8;
using pp::glds16;
using pp::wait_lds_reads;
using pp::wait_vmcnt;

constexpr int THREADS = 512;
using tnmap::PIECE_BYTES;
using tnmap::STAGE_BYTES;
constexpr int SMEM_BYTES = 2 * STAGE_BYTES;

typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t lds_address(const void* p) {
    return static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) const void*)p));
}

// One transposing read.  Inline asm, not __builtin_amdgcn_ds_read_tr16_b64_*: hipcc (ROCm 7.2) puts an s_waitcnt vmcnt(0) in
// front of the builtin whenever an LDS-DMA request is in flight (it carries no memory operand the waitcnt pass could tell apart
// from the DMA's destination) -- which would drain the pipeline the counted waits keep full.  The compiler does not count an asm
// load: every L segment ends with wait_lds_reads() + barrier() (a sched_barrier) before the first MFMA that consumes a fragment
// (cdna_hip_programming.md section 5.7, form iii).
template <int OFFSET>
__device__ __forceinline__ s16x4 read_tr(uint32_t lds_byte_address) {
    static_assert(OFFSET >= 0 && OFFSET < 65536 && OFFSET % 8 == 0, "16-bit offset field, 8-byte aligned words");
    s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(lds_byte_address), "n"(OFFSET));
    return v;
}

// the lane's bf16x8 operand fragment of k-slab KK: k-quads 0 and 1
template <int IMG, int KK>
__device__ __forceinline__ bf16x8 fragment(uint32_t base) {
    union { s16x4 h[2]; bf16x8 v; } u;
    u.h[0] = read_tr<tnmap::word_step<IMG>(KK, 0)>(base);
    u.h[1] = read_tr<tnmap::word_step<IMG>(KK, 1)>(base);
    return u.v;
}

// per-lane DMA sources for k-tile 0: p[2 piece + h] = request w + 8 h of the piece
struct Sources {
    const uint16_t* p[8];
};

// row_a(f) / row_b(f): address of (k = 0, tile row f) of the operand for the FIRST of a chunk of 8 rows (f % 8 == 0; the functor
// clamps a chunk that starts beyond the operand itself); ld_a / ld_b: elements between consecutive k.
template <int IMG, class RowA, class RowB>
__device__ __forceinline__ void make_sources(Sources& s, int wave, int lane, RowA row_a, int64_t ld_a, RowB row_b, int64_t ld_b) {
    using Img = tnmap::Image<IMG>;
#pragma unroll
    for (int piece = 0; piece < 4; ++piece)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int q = tnmap::request_of(wave, h);
            const int k = Img::dma_k(q, lane);
            const int f = tnmap::tile_row(piece, Img::dma_row(q, lane));
            s.p[2 * piece + h] = piece < 2 ? row_a(f) + k * ld_a : row_b(f) + k * ld_b;
        }
}

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// sum of the eight bf16 of a fragment, added to acc (v_dot2c_f32_bf16 with a pair of ones: beside the MFMAs it costs ~10 cycles each)
__device__ __forceinline__ float sum8(const bf16x8& f, float acc) {
    const bf16x2 one = {static_cast<__bf16>(1.0f), static_cast<__bf16>(1.0f)};
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f, f, 0, 1), one, acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f, f, 2, 3), one, acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f, f, 4, 5), one, acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f, f, 6, 7), one, acc, false);
    return acc;
}

// acc[i][jn] (i = 0..3 row blocks, jn = 0..1 column blocks of the wave's 128 x 64 tile) += A^T B over k-tiles [0, nt); k-tile t of
// an operand sits t * step_a / t * step_b elements behind Sources::p.  `wave` must be wave-uniform.  All 512 threads; sm: SMEM_BYTES
// of LDS at a 16-byte-aligned base.  On return every wave has finished reading LDS and every request has landed.
//
// colsum (wave-uniform): the waves also sum the B operand over k -- the column sums of the K-major B matrix, i.e. the bias column
// of a per-sample gradient (sum_t G[t][o]) or the bias row of a covariance -- from the fragments they hold anyway: wave (wm, wn)
// sums columns wn * 64 + wm * 32 + (lane & 31) of the tile; lanes l and l ^ 32 hold the two k-octets of a column (the caller adds
// them).  Returns this lane's partial sum (0 without colsum).
template <int IMG>
__device__ __forceinline__ float mainloop(f32x16 (&acc)[4][2], unsigned char* sm, const Sources& src, int nt, int wave, int lane,
                                          int64_t step_a, int64_t step_b, bool colsum = false) {
    const int wm = wave >> 2, wn = wave & 3;

    auto issue_at = [&](int piece, int t, int64_t off) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
            glds16(src.p[2 * piece + h] + off,
                   sm + (t & 1) * STAGE_BYTES + piece * PIECE_BYTES + tnmap::request_of(wave, h) * tnmap::REQUEST_BYTES);
    };
    auto issue_piece = [&](int piece, int t) { issue_at(piece, t, t * (piece < 2 ? step_a : step_b)); };

    // per-lane LDS byte addresses of the words of k-slab 0, quad 0 for this wave's blocks in stage 0; k-slab kk / quad add a
    // compile-time constant (tnmap::word_step: folded into the instruction's offset field), stage 1 adds STAGE_BYTES
    const uint32_t sm_lds = lds_address(sm);
    uint32_t wa[4], wb[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) wa[i] = sm_lds + tnmap::a_piece(i) * PIECE_BYTES + tnmap::word<IMG>(tnmap::a_row(wm, i), 0, 0, lane);
#pragma unroll
    for (int jn = 0; jn < 2; ++jn) wb[jn] = sm_lds + tnmap::b_piece(wn) * PIECE_BYTES + tnmap::word<IMG>(tnmap::b_row(wn, jn), 0, 0, lane);

    bf16x8 a[2][4], b[2][4];
    auto read_a = [&](int half, int buf) {   // half 0: blocks 0, 1 (piece A0), half 1: blocks 2, 3 (piece A1)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const uint32_t base = wa[2 * half + i] + buf * STAGE_BYTES;
            a[i][0] = fragment<IMG, 0>(base); a[i][1] = fragment<IMG, 1>(base); a[i][2] = fragment<IMG, 2>(base); a[i][3] = fragment<IMG, 3>(base);
        }
    };
    auto read_b = [&](int buf) {
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) {
            const uint32_t base = wb[jn] + buf * STAGE_BYTES;
            b[jn][0] = fragment<IMG, 0>(base); b[jn][1] = fragment<IMG, 1>(base); b[jn][2] = fragment<IMG, 2>(base); b[jn][3] = fragment<IMG, 3>(base);
        }
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

    // prologue: k-tile 0 complete, A0 / B0 / B1 of k-tile 1 on their way
    issue_piece(0, 0); issue_piece(2, 0); issue_piece(3, 0); issue_piece(1, 0);
    if (nt > 1) { issue_piece(0, 1); issue_piece(2, 1); issue_piece(3, 1); wait_vmcnt<8>(); }   // in flight: A1(0), A0 B0 B1(1)
    else wait_vmcnt<2>();
    barrier();
    if (wm == 1) barrier();   // Y runs half a phase behind X from here on (wave-uniform branch)

    float cs = 0.0f;
    // one k-tile = L(2t) M(2t) L(2t+1) M(2t+1); M(2t) carries A1(t+1), M(2t+1) A0, B0, B1 of t+2 (kf_pingpong.h, ISSUE = 1)
#define KF_TN_TILE(T, MORE1, MORE2)                                                                                    \
    do {                                                                                                               \
        const int t_ = (T), buf_ = t_ & 1;                                                                             \
        read_a(0, buf_);                                                                                               \
        read_b(buf_);                                                                                                  \
        if (MORE1) wait_vmcnt<6>();   /* A1(t) landed; in flight: A0, B0, B1 of t+1 */                                 \
        else wait_vmcnt<0>();                                                                                          \
        wait_lds_reads();                                                                                              \
        barrier();                                                                                                     \
        __builtin_amdgcn_s_setprio(1);                                                                                 \
        KF_TN_GROUP(0, 0);                                                                                             \
        ride(MORE1, 1, t_ + 1, (t_ + 1) * step_a);                                                                     \
        KF_TN_GROUP(0, 1);                                                                                             \
        KF_TN_GROUP(0, 2);                                                                                             \
        KF_TN_GROUP(0, 3);                                                                                             \
        __builtin_amdgcn_s_setprio(0);                                                                                 \
        barrier();                                                                                                     \
        read_a(1, buf_);                                                                                               \
        if (MORE1) wait_vmcnt<2>();   /* A0, B0, B1 of t+1 landed; in flight: A1(t+1) */                               \
        wait_lds_reads();                                                                                              \
        barrier();                                                                                                     \
        __builtin_amdgcn_s_setprio(1);                                                                                 \
        KF_TN_GROUP(1, 0);                                                                                             \
        ride(MORE2, 0, t_ + 2, (t_ + 2) * step_a);                                                                     \
        KF_TN_GROUP(1, 1);                                                                                             \
        ride(MORE2, 2, t_ + 2, (t_ + 2) * step_b);                                                                     \
        KF_TN_GROUP(1, 2);                                                                                             \
        ride(MORE2, 3, t_ + 2, (t_ + 2) * step_b);                                                                     \
        KF_TN_GROUP(1, 3);                                                                                             \
        if (colsum) {   /* the B fragments of this k-tile are still in registers */                                    \
            if (wm == 0) { cs = sum8(b[0][0], cs); cs = sum8(b[0][1], cs); cs = sum8(b[0][2], cs); cs = sum8(b[0][3], cs); } \
            else { cs = sum8(b[1][0], cs); cs = sum8(b[1][1], cs); cs = sum8(b[1][2], cs); cs = sum8(b[1][3], cs); }   \
        }                                                                                                              \
        __builtin_amdgcn_s_setprio(0);                                                                                 \
        barrier();                                                                                                     \
    } while (0)

    int t = 0;
    for (; t + 2 < nt; ++t) KF_TN_TILE(t, true, true);
    if (t + 1 < nt) { KF_TN_TILE(t, true, false); ++t; }
    KF_TN_TILE(t, false, false);
    if (wm == 0) barrier();   // X waits for Y's last segment: barrier counts match, all LDS reads are done
#undef KF_TN_TILE
#undef KF_TN_GROUP
    return cs;
}

}  // namespace pptn
}  // namespace kf
