// kf_pingpong.h -- 256 x 256 x 64 bf16 MFMA main loop with WAVE ROLE SPLIT for gfx950 (round 3).
//
// Why: the round-2 main loop (kf_score_v2.hip: every wave reads fragments, waits for them, issues MFMAs, and all eight
// waves drain vmcnt + barrier once per k-step) left the two waves that share a SIMD in lock-step: both wait for LDS at the
// same time, both want the matrix pipe at the same time (47 % MFMA utilisation, SQ_WAIT_INST_ANY 41 %).  Here the two waves
// of a SIMD run HALF A PHASE APART: waves 0-3 ("X") and waves 4-7 ("Y", which sit on the same four SIMDs) alternate between
//
//     L segment   ds_read the fragments of the next MFMA block, issue LDS-DMA requests for later k-tiles, wait for the
//                 fragments (lgkmcnt) and for exactly the DMA pieces the NEXT segment needs (counted vmcnt, never 0)
//     M segment   16 back-to-back v_mfma_f32_32x32x16_bf16 at raised priority
//
// separated by raw s_barriers.  Y executes one extra barrier up front, so while X is in M(P), Y is in L(P); after the next
// barrier X is in L(P + 1) and Y in M(P).  Per SIMD the matrix pipe is handed back and forth and every LDS / DMA latency of
// one wave is covered by the other wave's MFMAs (cdna_hip_programming.md section 5, "8-phase" template / T3-T5; this is a
// 4-segment-per-k-tile variant of it on the 32x32x16 instruction with the fragment set of a whole half tile in registers).
//
// Tile: 512 threads = 8 wave64, wave (wm, wn) = (wave >> 2, wave & 3) owns rows wm*128 .. +128 and columns wn*64 .. +64
// = 4 x 2 accumulators of 32 x 32.  One k-tile (64 deep) of the wave tile is TWO phases:
//     phase 0: A blocks i = 0, 1 (8 ds_read_b128) + B blocks jn = 0, 1 (8 reads)  ->  acc[0..1][0..1] += ...   16 MFMAs
//     phase 1: A blocks i = 2, 3 (8 reads), B fragments kept in registers          ->  acc[2..3][0..1] += ...   16 MFMAs
// (24 fragment reads per wave and k-tile: the minimum for a 128 x 64 wave tile.)
//
// LDS: two 64 KB buffers (k-tile t lives in buffer t & 1), each A rows 0..255 then B rows 0..255, 128 bytes per row, the
// 16-byte chunk c of row r at position c ^ ((r >> 1) & 7) (XOR on the DMA source address and on the fragment read alike:
// bank-conflict free, measured round 2).  The DMA moves a k-tile in FOUR pieces of 16 KB ordered by liveness -- a piece is
// re-staged for k-tile t + 2 as soon as k-tile t is done with it:
//     A0 = A rows of blocks i = 0, 1 of both wave rows (read in phase 0 only)      re-staged in L(2t + 1)
//     B0 = B rows 0..127, B1 = B rows 128..255 (read in phase 0 only)              re-staged in L(2t + 1) / L(2t + 2)
//     A1 = A rows of blocks i = 2, 3 (read in phase 1 only)                        re-staged in L(2t + 2)
// Issue order per wave (2 requests = 2 KB per piece):  L(2t): B1(t+1), A1(t+1);   L(2t+1): A0(t+2), B0(t+2).
//
// Ordering argument (S_k = the interval between global barrier events E_k and E_k+1; X runs L(P) in S_2P and M(P) in
// S_2P+1, Y runs L(P) in S_2P+1 and M(P) in S_2P+2):
//   RAW  a piece is complete once EVERY wave's part has landed: each wave waits (counted vmcnt) for its own requests at the
//        end of an L segment, i.e. before a barrier, and the piece is first read one phase later.  End of L(2t): A1(t) must
//        be there for L(2t+1) -> everything issued after it may stay in flight: A0, B0, B1, A1 of t+1 = 8 requests
//        (vmcnt(8)).  End of L(2t+1): A0, B0, B1 of t+1 must be there for L(2t+2) -> in flight: A1(t+1), A0(t+2), B0(t+2) = 6
//        (vmcnt(6)).  X's wait finishes in S_4t / S_4t+2, Y's one interval later, both before the barrier that opens the
//        first reading segment (X: S_4t+2 / S_4t+4).
//   WAR  a request issued in L(P) overwrites rows last read in L(P-1) (or earlier).  Every L segment ends with
//        s_waitcnt lgkmcnt(0) BEFORE its barrier, so when any wave passes that barrier all fragment reads of L(P-1) of both
//        groups have returned (Y's L(P-1) is S_2P-1, X's L(P) is S_2P).  Pieces written in L(P) are never read in L(P) or
//        L(P+1) of the other group: L(2t) writes buffer (t+1)&1 while tile t (buffer t&1) is being read; L(2t+1) writes A0 /
//        B0 of buffer t&1 while L(2t+1) reads only A1 of it and L(2t+2) reads the other buffer.
// Nothing else orders an LDS-DMA against a ds_read (MI355X_MICROARCH.md, "Two waves per SIMD", item 7).
//
// ISSUE != 0 (round 4): the requests ride BETWEEN THE MFMA GROUPS of the M segments instead of in the L segments (an LDS-DMA
// request costs ~60 cycles of issue among bare MFMAs but 100-185 inside a segment that already carries 8-16 ds_read_b128; the
// same move paid 3-15 % on the loop of kf_pingpong64.h).  A request issued in M(P) lands while the OTHER group runs L(P) and
// L(P + 1), and it must be waited for at the end of an L segment one phase before its first read, at least two segments
// after its issue -- which fixes the schedule:
//     M(2t)    A1(t+1)                        (2 requests; its rows were last read in L(2t-1))
//     M(2t+1)  A0(t+2), B0(t+2), B1(t+2)      (6 requests; last read in L(2t)).   ISSUE == 2: A0(t+2) stays in L(2t+1).
//   RAW  end of L(2t): A1(t) (issued in M(2t-2)) must be there for L(2t+1) -> in flight: A0, B0, B1 of t+1 = 6 (vmcnt(6)).
//        End of L(2t+1): A0, B0, B1 of t+1 (issued in L / M(2t-1)) must be there for L(2t+2) -> in flight: A1(t+1), and with
//        ISSUE == 2 the A0(t+2) just issued = 2 / 4 (vmcnt(2) / vmcnt(4)).  Each wait is >= 3 segments after the issue.
//   WAR  M(P) of X is S_2P+1, of Y S_2P+2: later than L(P), so every read the request overwrites has returned as argued above;
//        what is read meanwhile -- L(P) of Y in S_2P+1, L(P+1) of X in S_2P+2 -- is, for M(2t): tile t phase 0 / phase 1 (buffer
//        t & 1; the request writes buffer (t+1) & 1), for M(2t+1): A1 of tile t and tile t+1 (the other buffer); the request
//        writes A0 / B0 / B1 of buffer t & 1.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kf_engine.h"

namespace kf {
namespace pp {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int THREADS = 512;
constexpr int STAGE_BYTES = 65536, B_OFFSET = 32768, SMEM_BYTES = 2 * STAGE_BYTES;

// per-lane DMA sources of this wave's 8 requests per k-tile, for k-tile 0; the element offset of k-tile t is supplied by
// the two "walk" functors of mainloop (the same for every request of an operand)
//   p[0], p[1]: A0 (wave row j = 0, 1)   p[2], p[3]: A1   p[4], p[5]: B0 (rows j*128 ..)   p[6], p[7]: B1
struct Sources {
    const uint16_t* p[8];
};

// the k-octet a lane fetches is the SAME for its 8 requests: every request starts at a row that is a multiple of 8 whose
// bit 3 equals wave & 1, so swz(row) = 4 * (wave & 1) + (lane >> 4)
__device__ __forceinline__ int lane_octet(int wave, int lane) { return (lane & 7) ^ (4 * (wave & 1) + (lane >> 4)); }

// row (within the 256-row operand tile) whose 8-row group request `r` (0..7, order of Sources::p) of wave `wave` stages
__device__ __forceinline__ int request_row0(int r, int wave) {
    const int j = r & 1, h = (r >> 1) & 1;
    return r < 4 ? j * 128 + h * 64 + wave * 8                               // A: piece h, wave row j
                 : h * 128 + j * 64 + (wave >> 2) * 32 + (wave & 3) * 8;     // B: piece h = rows h*128.., two requests 64 rows apart
}
__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

__device__ __forceinline__ void glds16(const void* src, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_lds_reads() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void barrier() {
    asm volatile("s_barrier" ::: "memory");   // raw: no vmcnt drain (LDS-DMA requests stay in flight across it)
    __builtin_amdgcn_sched_barrier(0);        // nothing is scheduled across a segment boundary
}

// Fills the per-lane source table for an operand pair whose element (row, k) of the A / B tile lives at
// A + row_a(row) ... given by two functors returning the address of k = 0 of a row.
template <class RowA, class RowB>
__device__ __forceinline__ void make_sources(Sources& s, int wave, int lane, RowA row_a, RowB row_b) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int row = request_row0(r, wave) + (lane >> 3);
        const int chunk = (lane & 7) ^ swz(row);
        s.p[r] = (r < 4 ? row_a(row) : row_b(row)) + chunk * 8;
    }
}

// acc[i][jn] (i = 0..3 row blocks, jn = 0..1 column blocks of the wave's 128 x 64 tile) += A B^T over k-tiles [0, nt).
// walk_a(t) / walk_b(t): element offset of k-tile t relative to Sources::p (any per-lane value; evaluated once per piece).
// `wave` must be wave-uniform (readfirstlane'd).  All 512 threads; sm: SMEM_BYTES of LDS.  On return every wave has
// finished reading LDS (the buffers may be reused after one more barrier).
// ISSUE: where the LDS-DMA requests are issued -- 0: in the L segments (round 3), 1: between the MFMA groups of the M segments,
// 2: as 1 with A0 left in L(2t+1) (the segment with 8 fragment reads); see the header.
template <int ISSUE = 0, class WalkA, class WalkB>
__device__ __forceinline__ void mainloop(f32x16 (&acc)[4][2], unsigned char* sm, const Sources& src, int nt, int wave, int lane,
                                         WalkA walk_a, WalkB walk_b) {
    static_assert(ISSUE >= 0 && ISSUE <= 2, "request schedule");
    const int wm = wave >> 2, wn = wave & 3;
    const int lr = lane & 31, hi = lane >> 5, sw = (lr >> 1) & 7;
    int co[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) co[kk] = ((kk * 2 + hi) ^ sw) * 16;
    const unsigned char* frag_a = sm + (wm * 128 + lr) * 128;
    const unsigned char* frag_b = sm + B_OFFSET + (wn * 64 + lr) * 128;

    // piece (0 A0, 1 A1, 2 B0, 3 B1) of k-tile t = requests 2 piece, 2 piece + 1 -> buffer t & 1
    auto issue_at = [&](int piece, int t, int64_t off) {
#pragma unroll
        for (int r = 2 * piece; r < 2 * piece + 2; ++r) {
            unsigned char* dst = sm + (t & 1) * STAGE_BYTES + (r < 4 ? 0 : B_OFFSET) + request_row0(r, wave) * 128;
            glds16(src.p[r] + off, dst);
        }
    };
    auto issue_piece = [&](int piece, int t) { issue_at(piece, t, piece < 2 ? walk_a(t) : walk_b(t)); };

    bf16x8 a[2][4], b[2][4];
    auto read_a = [&](int half, int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                a[i][kk] = *reinterpret_cast<const bf16x8*>(frag_a + buf * STAGE_BYTES + (half * 2 + i) * 4096 + co[kk]);
    };
    auto read_b = [&](int buf) {
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                b[jn][kk] = *reinterpret_cast<const bf16x8*>(frag_b + buf * STAGE_BYTES + jn * 4096 + co[kk]);
    };
#define KF_PP_MFMA(HALF)                                                                                               \
    do {                                                                                                               \
        __builtin_amdgcn_s_setprio(1);                                                                                 \
        _Pragma("unroll") for (int kk = 0; kk < 4; ++kk)                                                               \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                              \
                _Pragma("unroll") for (int jn = 0; jn < 2; ++jn)                                                       \
                    acc[(HALF) * 2 + i][jn] =                                                                          \
                        __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][kk], b[jn][kk], acc[(HALF) * 2 + i][jn], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                                 \
    } while (0)

    // one MFMA group = the four products of k-slab KK
#define KF_PP_GROUP(HALF, KK)                                                                                          \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                                      \
        _Pragma("unroll") for (int jn = 0; jn < 2; ++jn)                                                               \
            acc[(HALF) * 2 + i][jn] =                                                                                  \
                __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][KK], b[jn][KK], acc[(HALF) * 2 + i][jn], 0, 0, 0)
    // a piece issued between two MFMA groups, pinned there
    auto ride = [&](bool on, int piece, int t, int64_t off) {
        if (on) {
            __builtin_amdgcn_sched_barrier(0);
            issue_at(piece, t, off);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // prologue: k-tile 0 complete, A0 / B0 (ISSUE != 0: and B1) of k-tile 1 on their way
    issue_piece(0, 0); issue_piece(2, 0); issue_piece(3, 0); issue_piece(1, 0);
    if constexpr (ISSUE == 0) {
        if (nt > 1) { issue_piece(0, 1); issue_piece(2, 1); wait_vmcnt<6>(); }   // in flight: A1(0), A0(1), B0(1)
        else wait_vmcnt<2>();                                                     // in flight: A1(0)
    } else {
        if (nt > 1) { issue_piece(0, 1); issue_piece(2, 1); issue_piece(3, 1); wait_vmcnt<8>(); }   // A1(0), A0 B0 B1(1)
        else wait_vmcnt<2>();
    }
    barrier();
    if (wm == 1) barrier();   // Y runs half a phase behind X from here on (wave-uniform branch)

    // one k-tile = L(2t) M(2t) L(2t+1) M(2t+1);  MORE1: k-tile t+1 exists, MORE2: k-tile t+2 exists
#define KF_PP_TILE(T, MORE1, MORE2)                                                                                    \
    do {                                                                                                               \
        const int t_ = (T), buf_ = t_ & 1;                                                                             \
        int64_t oa_ = 0, ob_ = 0;   /* walks first: an offset read from an LDS table is then the oldest LDS request */ \
        if (MORE1) { oa_ = walk_a(t_ + 1); ob_ = walk_b(t_ + 1); }                                                     \
        read_a(0, buf_);                                                                                               \
        read_b(buf_);                                                                                                  \
        if (MORE1) { issue_at(3, t_ + 1, ob_); issue_at(1, t_ + 1, oa_); wait_vmcnt<8>(); }                            \
        else wait_vmcnt<0>();                                                                                          \
        wait_lds_reads();                                                                                              \
        barrier();                                                                                                     \
        KF_PP_MFMA(0);                                                                                                 \
        barrier();                                                                                                     \
        if (MORE2) { oa_ = walk_a(t_ + 2); ob_ = walk_b(t_ + 2); }                                                     \
        read_a(1, buf_);                                                                                               \
        if (MORE2) { issue_at(0, t_ + 2, oa_); issue_at(2, t_ + 2, ob_); wait_vmcnt<6>(); }                            \
        else if (MORE1) wait_vmcnt<2>();                                                                               \
        wait_lds_reads();                                                                                              \
        barrier();                                                                                                     \
        KF_PP_MFMA(1);                                                                                                 \
        barrier();                                                                                                     \
    } while (0)

    // the same k-tile with the requests in the M segments (ISSUE 1 / 2): M(2t) carries A1(t+1), M(2t+1) (A0,) B0, B1 of t+2
#define KF_PP_TILE_M(T, MORE1, MORE2)                                                                                  \
    do {                                                                                                               \
        const int t_ = (T), buf_ = t_ & 1;                                                                             \
        int64_t oa_ = 0, ob_ = 0;                                                                                      \
        if (MORE1) oa_ = walk_a(t_ + 1);                                                                               \
        read_a(0, buf_);                                                                                               \
        read_b(buf_);                                                                                                  \
        if (MORE1) wait_vmcnt<6>();   /* A1(t) landed; in flight: A0, B0, B1 of t+1 */                                 \
        else wait_vmcnt<0>();                                                                                          \
        wait_lds_reads();                                                                                              \
        barrier();                                                                                                     \
        __builtin_amdgcn_s_setprio(1);                                                                                 \
        KF_PP_GROUP(0, 0);                                                                                             \
        ride(MORE1, 1, t_ + 1, oa_);                                                                                   \
        KF_PP_GROUP(0, 1);                                                                                             \
        KF_PP_GROUP(0, 2);                                                                                             \
        KF_PP_GROUP(0, 3);                                                                                             \
        __builtin_amdgcn_s_setprio(0);                                                                                 \
        barrier();                                                                                                     \
        if (MORE2) { oa_ = walk_a(t_ + 2); ob_ = walk_b(t_ + 2); }                                                     \
        read_a(1, buf_);                                                                                               \
        if (ISSUE == 2 && (MORE2)) issue_at(0, t_ + 2, oa_);                                                           \
        if (MORE1) {   /* A0, B0, B1 of t+1 landed; in flight: A1(t+1) (+ the A0(t+2) just issued) */                  \
            if (ISSUE == 2 && (MORE2)) wait_vmcnt<4>();                                                                \
            else wait_vmcnt<2>();                                                                                      \
        }                                                                                                              \
        wait_lds_reads();                                                                                              \
        barrier();                                                                                                     \
        __builtin_amdgcn_s_setprio(1);                                                                                 \
        KF_PP_GROUP(1, 0);                                                                                             \
        ride((MORE2) && ISSUE == 1, 0, t_ + 2, oa_);                                                                   \
        ride((MORE2) && ISSUE == 2, 2, t_ + 2, ob_);                                                                   \
        KF_PP_GROUP(1, 1);                                                                                             \
        ride((MORE2) && ISSUE == 1, 2, t_ + 2, ob_);                                                                   \
        ride((MORE2) && ISSUE == 2, 3, t_ + 2, ob_);                                                                   \
        KF_PP_GROUP(1, 2);                                                                                             \
        ride((MORE2) && ISSUE == 1, 3, t_ + 2, ob_);                                                                   \
        KF_PP_GROUP(1, 3);                                                                                             \
        __builtin_amdgcn_s_setprio(0);                                                                                 \
        barrier();                                                                                                     \
    } while (0)

    int t = 0;
    if constexpr (ISSUE == 0) {
        for (; t + 2 < nt; ++t) KF_PP_TILE(t, true, true);
        if (t + 1 < nt) { KF_PP_TILE(t, true, false); ++t; }
        KF_PP_TILE(t, false, false);
    } else {
        for (; t + 2 < nt; ++t) KF_PP_TILE_M(t, true, true);
        if (t + 1 < nt) { KF_PP_TILE_M(t, true, false); ++t; }
        KF_PP_TILE_M(t, false, false);
    }
    if (wm == 0) barrier();   // X waits for Y's last segment: barrier counts match, all LDS reads are done
#undef KF_PP_TILE
#undef KF_PP_TILE_M
#undef KF_PP_MFMA
#undef KF_PP_GROUP
}

}  // namespace pp

// ------------------------------------------------------------------------------------------------
// Round 4: the SAME loop for other wave grids -- WM x WN waves of 128 x 64, tile (WM 128) x (WN 64): 4 x 2 = 512 x 128 and
// 1 x 8 = 128 x 512 (pp above is the 2 x 4 case, kept as it is: its kernels carry three rounds of verification).  A parameter
// change of that template, not a new synchronisation structure: the phases, the four pieces per k-tile and their liveness
// order (A0 | B0, B1 | A1), the issue order (L(2t): B1, A1 of t + 1; L(2t+1): A0, B0 of t + 2) and the RAW / WAR argument at the
// top of this file carry over verbatim; only the request counts per piece change -- NA = WM per A piece, NB = WN / 2 per B
// piece and wave -- and with them the counted waits:
//     end of L(2t)    in flight: A0, B0, B1, A1 of t + 1     = 2 NA + 2 NB
//     end of L(2t+1)  in flight: A1(t + 1), A0, B0 of t + 2  = 2 NA + NB      (no t + 2: A1(t + 1) = NA)
// Two stages of (WM 128 + WN 64) x 128 B: 80 KB each for 4 x 2 and 1 x 8, i.e. ALL 160 KB of a CU's LDS.
// Why: a train batch of 128 sequences (GPT-2) against >= 512 queries fills a 512 x 128 tile with 8 + 2 = 10 DMA requests per
// wave and k-tile for 32 MFMAs -- the 256 x 128 tile of kf_pingpong64.h needs 12 for the same MFMAs.
// ------------------------------------------------------------------------------------------------
namespace ppw {

using pp::bf16x8;
using pp::glds16;
using pp::swz;

template <int WM, int WN>
struct Geo {
    static_assert(WM * WN == 8 && WN % 2 == 0, "8 waves, an even number of wave columns");
    static constexpr int TA = WM * 128, TB = WN * 64;
    static constexpr int NA = WM, NB = WN / 2;   // requests per wave for one A piece / one B piece
    static constexpr int A_BYTES = TA * 128, STAGE_BYTES = (TA + TB) * 128, SMEM_BYTES = 2 * STAGE_BYTES;
};

// p[0 .. NA): A0, p[NA .. 2 NA): A1, then B0 (NB), B1 (NB)
template <int WM, int WN>
struct Sources {
    const uint16_t* p[2 * Geo<WM, WN>::NA + 2 * Geo<WM, WN>::NB];
};

// first row (within the operand tile) of request r of `wave`; every such row is 8 (8 j' + wave): bit 3 = wave & 1, so
// pp::lane_octet holds here too
template <int WM, int WN>
__device__ __forceinline__ int request_row0(int r, int wave) {
    using G = Geo<WM, WN>;
    if (r < 2 * G::NA) { const int h = r / G::NA, j = r % G::NA; return j * 128 + h * 64 + wave * 8; }
    const int rb = r - 2 * G::NA, h = rb / G::NB, j = rb % G::NB;
    return h * (G::TB / 2) + j * 64 + wave * 8;
}

template <int WM, int WN, class RowA, class RowB>
__device__ __forceinline__ void make_sources(Sources<WM, WN>& s, int wave, int lane, RowA row_a, RowB row_b) {
    using G = Geo<WM, WN>;
#pragma unroll
    for (int r = 0; r < 2 * G::NA + 2 * G::NB; ++r) {
        const int row = request_row0<WM, WN>(r, wave) + (lane >> 3);
        const int chunk = (lane & 7) ^ swz(row);
        s.p[r] = (r < 2 * G::NA ? row_a(row) : row_b(row)) + chunk * 8;
    }
}

template <int WM, int WN, class WalkA, class WalkB>
__device__ __forceinline__ void mainloop(f32x16 (&acc)[4][2], unsigned char* sm, const Sources<WM, WN>& src, int nt, int wave, int lane,
                                         WalkA walk_a, WalkB walk_b) {
    using G = Geo<WM, WN>;
    constexpr int NA = G::NA, NB = G::NB;
    const int wm = wave / WN, wn = wave % WN, role = wave >> 2;
    const int lr = lane & 31, hi = lane >> 5, sw = (lr >> 1) & 7;
    int co[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) co[kk] = ((kk * 2 + hi) ^ sw) * 16;
    const unsigned char* frag_a = sm + (wm * 128 + lr) * 128;
    const unsigned char* frag_b = sm + G::A_BYTES + (wn * 64 + lr) * 128;

    // piece: 0 A0, 1 A1, 2 B0, 3 B1 of k-tile t -> stage t & 1
    auto issue_at = [&](int piece, int t, int64_t off) {
        const int first = piece < 2 ? piece * NA : 2 * NA + (piece - 2) * NB, count = piece < 2 ? NA : NB;
#pragma unroll
        for (int r = first; r < first + count; ++r) {
            unsigned char* dst = sm + (t & 1) * G::STAGE_BYTES + (r < 2 * NA ? 0 : G::A_BYTES) + request_row0<WM, WN>(r, wave) * 128;
            glds16(src.p[r] + off, dst);
        }
    };
    auto issue_piece = [&](int piece, int t) { issue_at(piece, t, piece < 2 ? walk_a(t) : walk_b(t)); };

    bf16x8 a[2][4], b[2][4];
    auto read_a = [&](int half, int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                a[i][kk] = *reinterpret_cast<const bf16x8*>(frag_a + buf * G::STAGE_BYTES + (half * 2 + i) * 4096 + co[kk]);
    };
    auto read_b = [&](int buf) {
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                b[jn][kk] = *reinterpret_cast<const bf16x8*>(frag_b + buf * G::STAGE_BYTES + jn * 4096 + co[kk]);
    };
#define KF_PPW_MFMA(HALF)                                                                                              \
    do {                                                                                                               \
        __builtin_amdgcn_s_setprio(1);                                                                                 \
        _Pragma("unroll") for (int kk = 0; kk < 4; ++kk)                                                               \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                              \
                _Pragma("unroll") for (int jn = 0; jn < 2; ++jn)                                                       \
                    acc[(HALF) * 2 + i][jn] =                                                                          \
                        __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][kk], b[jn][kk], acc[(HALF) * 2 + i][jn], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                                 \
    } while (0)

    // prologue: k-tile 0 complete, A0 / B0 of k-tile 1 on their way
    issue_piece(0, 0); issue_piece(2, 0); issue_piece(3, 0); issue_piece(1, 0);
    if (nt > 1) { issue_piece(0, 1); issue_piece(2, 1); pp::wait_vmcnt<2 * NA + NB>(); }   // in flight: A1(0), A0(1), B0(1)
    else pp::wait_vmcnt<NA>();                                                               // in flight: A1(0)
    pp::barrier();
    if (role == 1) pp::barrier();   // Y runs half a phase behind X from here on (wave-uniform branch)

#define KF_PPW_TILE(T, MORE1, MORE2)                                                                                   \
    do {                                                                                                               \
        const int t_ = (T), buf_ = t_ & 1;                                                                             \
        int64_t oa_ = 0, ob_ = 0;                                                                                      \
        if (MORE1) { oa_ = walk_a(t_ + 1); ob_ = walk_b(t_ + 1); }                                                     \
        read_a(0, buf_);                                                                                               \
        read_b(buf_);                                                                                                  \
        if (MORE1) { issue_at(3, t_ + 1, ob_); issue_at(1, t_ + 1, oa_); pp::wait_vmcnt<2 * NA + 2 * NB>(); }          \
        else pp::wait_vmcnt<0>();                                                                                      \
        pp::wait_lds_reads();                                                                                          \
        pp::barrier();                                                                                                 \
        KF_PPW_MFMA(0);                                                                                                \
        pp::barrier();                                                                                                 \
        if (MORE2) { oa_ = walk_a(t_ + 2); ob_ = walk_b(t_ + 2); }                                                     \
        read_a(1, buf_);                                                                                               \
        if (MORE2) { issue_at(0, t_ + 2, oa_); issue_at(2, t_ + 2, ob_); pp::wait_vmcnt<2 * NA + NB>(); }              \
        else if (MORE1) pp::wait_vmcnt<NA>();                                                                          \
        pp::wait_lds_reads();                                                                                          \
        pp::barrier();                                                                                                 \
        KF_PPW_MFMA(1);                                                                                                \
        pp::barrier();                                                                                                 \
    } while (0)

    int t = 0;
    for (; t + 2 < nt; ++t) KF_PPW_TILE(t, true, true);
    if (t + 1 < nt) { KF_PPW_TILE(t, true, false); ++t; }
    KF_PPW_TILE(t, false, false);
    if (role == 0) pp::barrier();   // X waits for Y's last segment: barrier counts match, all LDS reads are done
#undef KF_PPW_TILE
#undef KF_PPW_MFMA
}

}  // namespace ppw
}  // namespace kf
