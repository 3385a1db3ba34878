// kf_pingpong64.h -- wave-role-split bf16 MFMA main loop for 256 x 128 / 128 x 256 tiles, waves of 64 x 64 (round 4).
//
// Why a second loop: kf_pingpong.h needs 128 accumulator registers per lane (wave tile 128 x 64).  Two users cannot afford
// that: (1) the Lambda product of a sequence layer keeps a SECOND tile-sized register set, the running sum of squares over
// samples -- acc + sumsq of a 128 x 64 wave tile is the whole 256-register budget of a wave at two waves per SIMD; (2) score
// GEMMs whose train batch (GPT-2: 128 sequences) or query count is half a 256-row tile.  Here a wave owns 64 x 64 = 2 x 2
// accumulators of v_mfma_f32_32x32x16_bf16 (64 registers), 8 waves form a TA x TB tile (256 x 128: 4 x 2 waves, or
// 128 x 256: 2 x 4), and ONE k-tile (64 deep) is ONE phase:
//
//     L(t)  16 ds_read_b128 (2 A blocks + 2 B blocks x 4 k-slabs) of k-tile t, LREQ of the 6 LDS-DMA requests of k-tile t + 2,
//           s_waitcnt vmcnt(LREQ) (this wave's pieces of k-tile t + 1 have landed), lgkmcnt(0)
//     M(t)  16 MFMAs at raised priority in four groups, the other 6 - LREQ requests between the groups (+ the caller's per-tile
//           hook: the Lambda fold).  LREQ = 6 is the plain form the ordering argument below is written for; the Lambda kernel
//           runs LREQ = 0 (measured fastest), see the note at `mainloop`.
//
// separated by raw s_barriers; waves 4-7 ("Y", on the same four SIMDs as waves 0-3, "X") run one barrier behind, so per SIMD
// one wave is in L while the other is in M (cdna_hip_programming.md section 5, T3-T5).
//
// LDS: THREE stages of (TA + TB) x 128 B = 48 KB (k-tile t lives in stage t % 3), A rows then B rows, 128 bytes per row, the
// 16-byte chunk c of row r at position c ^ ((r >> 1) & 7) -- on the DMA source address and the fragment read alike, as in
// kf_pingpong.h.  Wave w stages the 8-row groups w, w + 8, ... of A (TA / 64 requests) and of B (TB / 64 requests).
//
// Ordering (S_k = interval between barrier events E_k and E_k+1; X runs L(t) in S_2t and M(t) in S_2t+1, Y one interval later):
//   RAW  k-tile t + 1 is first read in X's L(t + 1) = S_2t+2, after E_2t+2.  Every wave waits (counted vmcnt) for ITS pieces
//        of k-tile t + 1 at the end of ITS L(t): X in S_2t (before E_2t+1), Y in S_2t+1 (before E_2t+2) -- both before the
//        barrier that opens the first reading segment, and a phase after the requests were issued (L(t - 1)), never in the
//        issuing phase.  Prologue: k-tile 0 is waited for by every wave before E_0.
//   WAR  k-tile t + 2 overwrites stage (t + 2) % 3 = (t - 1) % 3, last read in L(t - 1): X in S_2t-2, Y in S_2t-1, each ending
//        with s_waitcnt lgkmcnt(0) BEFORE its barrier (E_2t-1 / E_2t).  The requests are issued in L(t): X in S_2t (after E_2t),
//        Y in S_2t+1.  Two stages would force the wait into the issuing phase (the whole DMA latency exposed): hence three.
// Nothing else orders an LDS-DMA against a ds_read (MI355X_MICROARCH.md, "Two waves per SIMD", item 7).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kf_engine.h"
#include "kf_pingpong.h"

namespace kf {
namespace pp64 {

using pp::bf16x8;
using pp::glds16;
using pp::swz;

constexpr int THREADS = 512;

template <int TA, int TB>
struct Geo {
    static_assert((TA == 256 && TB == 128) || (TA == 128 && TB == 256), "8 waves of 64 x 64");
    static constexpr int WA = TA / 64, WB = TB / 64;        // wave grid
    static constexpr int NA = TA / 64, NB = TB / 64;        // LDS-DMA requests per wave and k-tile
    static constexpr int A_BYTES = TA * 128, STAGE_BYTES = (TA + TB) * 128, SMEM_BYTES = 3 * STAGE_BYTES;
};

// per-lane DMA sources of this wave's requests for k-tile 0: p[0 .. NA) the A row groups wave, wave + 8, ...;
// p[NA .. NA + NB) the B row groups.  As in kf_pingpong.h every request of a lane fetches the same k-octet
// (pp::lane_octet): a group starts at row 8 (8 j + wave), whose bit 3 is wave & 1.
template <int TA, int TB>
struct Sources {
    const uint16_t* p[Geo<TA, TB>::NA + Geo<TA, TB>::NB];
};

template <int TA, int TB, class RowA, class RowB>
__device__ __forceinline__ void make_sources(Sources<TA, TB>& s, int wave, int lane, RowA row_a, RowB row_b) {
    using G = Geo<TA, TB>;
#pragma unroll
    for (int r = 0; r < G::NA + G::NB; ++r) {
        const int row = ((r < G::NA ? r : r - G::NA) * 8 + wave) * 8 + (lane >> 3);
        const int chunk = (lane & 7) ^ swz(row);
        s.p[r] = (r < G::NA ? row_a(row) : row_b(row)) + chunk * 8;
    }
}

// Control of the accumulators across k-tiles:
//   ctl.first(t) (wave-uniform)  k-tile t starts a new accumulation: its first MFMAs take C = 0
//   ctl.done(t, acc)             after the MFMAs of k-tile t (the Lambda kernel folds acc^2 into its sums at a sample's end)
struct PlainCtl {
    __device__ __forceinline__ bool first(int) const { return false; }
    __device__ __forceinline__ void done(int, f32x16 (&)[2][2]) const {}
};

// acc[i][jn] (i: 32-row blocks of the wave's 64 A rows, jn: 32-column blocks of its 64 B rows) (+)= A B^T over k-tiles [0, nt).
// walk_a(t) / walk_b(t): element offset of k-tile t relative to Sources::p.  `wave` must be wave-uniform.  All 512 threads;
// sm: Geo::SMEM_BYTES of LDS.  On return every wave has finished reading LDS.
// LREQ: how many of the NA + NB requests of k-tile t + 2 are issued in L(t); the rest ride between the MFMA groups of M(t)
// (an LDS-DMA request costs ~60 cycles of issue among bare MFMAs but 100-185 inside a segment that already carries 16
// ds_read_b128, MI355X_MICROARCH.md).  Ordering with LREQ < NA + NB: the late requests are issued in M(t) -- X in S_2t+1, Y in
// S_2t+2, still after every read of the stage they overwrite (L(t - 1)) -- and are covered by the issuing wave's counted wait at
// the end of ITS L(t + 1) (vmcnt(LREQ): only the early requests of k-tile t + 3 may still be in flight), i.e. before the barrier
// that opens the first segment reading k-tile t + 2.
template <int TA, int TB, int LREQ, class WalkA, class WalkB, class Ctl>
__device__ __forceinline__ void mainloop(f32x16 (&acc)[2][2], unsigned char* sm, const Sources<TA, TB>& src, int nt, int wave,
                                         int lane, WalkA walk_a, WalkB walk_b, Ctl& ctl) {
    using G = Geo<TA, TB>;
    static_assert(LREQ >= 0 && LREQ <= G::NA + G::NB, "requests issued in the L segment");
    constexpr int NREQ = G::NA + G::NB, LATE = NREQ - LREQ;
    // late requests after MFMA groups kk = 0, 1, 2: as even as possible
    constexpr int LATE0 = (LATE + 2) / 3, LATE1 = (LATE - LATE0 + 1) / 2, LATE2 = LATE - LATE0 - LATE1;
    const int wm = wave / G::WB, wn = wave % G::WB, role = wave >> 2;
    const int lr = lane & 31, hi = lane >> 5, sw = (lr >> 1) & 7;
    int co[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) co[kk] = ((kk * 2 + hi) ^ sw) * 16;
    const unsigned char* frag_a = sm + (wm * 64 + lr) * 128;
    const unsigned char* frag_b = sm + G::A_BYTES + (wn * 64 + lr) * 128;

    auto issue_range = [&](int stage, int64_t oa, int64_t ob, int first, int count) {   // requests [first, first + count)
        unsigned char* base = sm + stage * G::STAGE_BYTES;
#pragma unroll
        for (int r = 0; r < NREQ; ++r) {
            if (r < first || r >= first + count) continue;
            if (r < G::NA) glds16(src.p[r] + oa, base + (r * 8 + wave) * 1024);
            else glds16(src.p[r] + ob, base + G::A_BYTES + ((r - G::NA) * 8 + wave) * 1024);
        }
    };
    auto issue = [&](int stage, int64_t oa, int64_t ob) { issue_range(stage, oa, ob, 0, NREQ); };
    bf16x8 a[2][4], b[2][4];
    auto read_frags = [&](int stage) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) a[i][kk] = *reinterpret_cast<const bf16x8*>(frag_a + stage * G::STAGE_BYTES + i * 4096 + co[kk]);
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) b[jn][kk] = *reinterpret_cast<const bf16x8*>(frag_b + stage * G::STAGE_BYTES + jn * 4096 + co[kk]);
    };

    // prologue: k-tile 0 complete, k-tile 1 on its way
    issue(0, walk_a(0), walk_b(0));
    if (nt > 1) { issue(1, walk_a(1), walk_b(1)); pp::wait_vmcnt<G::NA + G::NB>(); }
    else pp::wait_vmcnt<0>();
    pp::barrier();
    if (role == 1) pp::barrier();   // Y runs one segment behind X from here on (wave-uniform branch)

    int rd = 0, wr = 2;
    for (int t = 0; t < nt; ++t) {
        const bool more2 = t + 2 < nt;
        int64_t oa = 0, ob = 0;   // walks first: an offset read from an LDS table is then the oldest LDS request
        if (more2) { oa = walk_a(t + 2); ob = walk_b(t + 2); }
        read_frags(rd);
        if (more2) { issue_range(wr, oa, ob, 0, LREQ); pp::wait_vmcnt<LREQ>(); }   // in flight: the early requests of k-tile t + 2 only
        else pp::wait_vmcnt<0>();                                                  // k-tile t + 1 (if any) has landed
        pp::wait_lds_reads();
        pp::barrier();
        __builtin_amdgcn_s_setprio(1);
        // four groups of four MFMAs (k-slabs); the late requests of k-tile t + 2 ride between them
        auto late = [&](int first, int count) {
            if (count > 0 && more2) {
                __builtin_amdgcn_sched_barrier(0);
                issue_range(wr, oa, ob, first, count);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (ctl.first(t)) {
            const f32x16 zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jn = 0; jn < 2; ++jn) acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[jn][0], zero, 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jn = 0; jn < 2; ++jn) acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[jn][0], acc[i][jn], 0, 0, 0);
        }
        late(LREQ, LATE0);
#pragma unroll
        for (int kk = 1; kk < 4; ++kk) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jn = 0; jn < 2; ++jn) acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][kk], b[jn][kk], acc[i][jn], 0, 0, 0);
            if (kk == 1) late(LREQ + LATE0, LATE1);
            if (kk == 2) late(LREQ + LATE0 + LATE1, LATE2);
        }
        __builtin_amdgcn_s_setprio(0);
        ctl.done(t, acc);
        pp::barrier();
        rd = rd == 2 ? 0 : rd + 1;
        wr = wr == 2 ? 0 : wr + 1;
    }
    if (role == 0) pp::barrier();   // X waits for Y's last segment: barrier counts match, all LDS reads are done
}

}  // namespace pp64
}  // namespace kf
