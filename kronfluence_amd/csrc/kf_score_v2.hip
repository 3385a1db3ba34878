// kf_score_v2.hip -- second-generation bf16 score path (gfx950): LDS-DMA staged MFMA kernels.
//
// The R > 1 pairwise score of one layer and one train batch is two contractions (reference
// module/conv2d.py:199-209, module/linear.py:112-122, module/tracker/pairwise_score.py:41-45):
//
//   psg[n][o, i]  = sum_r G[n, r, o] A'[n, r, i]                  per-sample gradient, K = R (64 .. 1024)
//   scores[q, n] += scale * sum_{o,i} P[q, o, i] psg[n, o, i]     one GEMM over D = O I' (1e5 .. 2.4e6)
//
// Both kernels here stage their operand tiles with global_load_lds_dwordx4 (16 bytes per lane straight
// into LDS: no staging VGPRs, no ds_write pass), so both operands must be K-CONTIGUOUS:
//
//   * score GEMM: P and psg are k-tile-major, [D/64][rows][64] -- one k-step of a 256-row tile is one
//     contiguous 32 KB block;
//   * per-sample gradient: G is consumed as [n][o][r] -- exactly the NCHW layout autograd hands a
//     convolution's output gradient over -- and A' as [n][i][r].  For convolutions the rows of A' are never
//     materialised (IMPLICIT im2col): row i = (ky, kx, c) of sample n is a strided walk over a zero-padded
//     copy of the layer input, and because the LDS-DMA takes a per-lane source address, every 16-byte chunk
//     (8 consecutive output columns of one output row) is fetched directly from that copy.  For Linear
//     layers on sequences a small transposition kernel produces [n][o][t] / [n][i'][t] (and appends the
//     ones row of the bias and the zero rows that pad I' to a multiple of 8).
//
// LDS image of an operand tile: [rows][64 k] bf16 with 128-byte rows and NO padding (the DMA writes lane-
// linear); the 16-byte chunk c of row r lives at chunk position c ^ ((r >> 1) & 7), applied to the SOURCE
// address of the DMA and to the ds_read_b128 fragment address alike, which makes the fragment reads
// (lane l: row l & 31, k-octet l >> 5) bank-conflict free (cdna_hip_programming.md T2, rule 21).
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>
#include <mutex>

#include "../../include/kronfluence_hip.h"
#include "kf_engine.h"
#include "kf_pingpong.h"
#include "kf_pingpong64.h"
#include "kf_pingpong_tn.h"

using namespace kf;

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline int launch_status() { return hipGetLastError() == hipSuccess ? KF_OK : KF_ERR_LAUNCH_FAILED; }

__device__ __forceinline__ void glds16(const void* src, void* lds_wave_base) {
    // 16 bytes per lane: global (per-lane address) -> LDS at lds_wave_base + lane * 16 (wave-uniform base in M0)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ int lds_swz(int row) { return (row >> 1) & 7; }

// One 64-deep k-step of a wave's 64 x 64 block (2 x 2 accumulators) from an LDS stage; sa / sb = this lane's row in the A / B
// image.  (Requesting the fragments of k-slab kk + 1 before the MFMAs of slab kk -- asm ds_reads with counted lgkmcnt, because
// hipcc drains lgkmcnt to 0 after every global_load_lds -- was measured on all three LDS-DMA kernels: no change; the second
// wave of the SIMD already fills those gaps.)
__device__ __forceinline__ void wave_kstep_64x64(f32x16 (&acc)[2][2], const unsigned char* sa, const unsigned char* sb, int hi, int sw) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int co = ((kk * 2 + hi) ^ sw) * 16;
        const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(sa + co);
        const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(sa + 32 * 128 + co);
        const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(sb + co);
        const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(sb + 32 * 128 + co);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
    }
}

// ------------------------------------------------------------------------------------------------
// Score GEMM: C[m, n] += alpha * sum_k A[m, k] B[n, k], A and B k-tile-major bf16.
// 512 threads = 8 wave64 as 2 (m) x 4 (n); 256 x 256 tile, each wave 128 x 64 = 4 x 2 accumulators of
// v_mfma_f32_32x32x16_bf16; k-step 64; two LDS stages of 64 KB (A 32 KB + B 32 KB).
// ------------------------------------------------------------------------------------------------
struct ScoreV2Args {
    float* C; int64_t ldc;
    const uint16_t* A; const uint16_t* B;
    int M, N, KT;                          // KT = K / 64
    int a_rows;                            // rows of a k-tile of A in memory (>= M: a launch may cover a row range of P)
    int tiles_m, tiles_n, ksplit, kchunk;  // kchunk in k-tiles
    float alpha;
};

constexpr int SV2_THREADS = 512;

// TM x TN output tile, WMW waves along m (8 / WMW along n); every wave owns (TM / WMW) x (TN / (8 / WMW)) as MI x NI
// accumulators of 32 x 32.  256 x 256 (2 x 4 waves of 128 x 64) is the workhorse; 256 x 128 and 128 x 256 (waves of
// 64 x 64) cut the padding when the train batch or the query count is far from a multiple of 256.
template <int TM, int TN, int WMW>
__global__ __launch_bounds__(SV2_THREADS) void score_gemm_v2_kernel(ScoreV2Args a) {
    constexpr int WNW = 8 / WMW, MI = TM / WMW / 32, NI = TN / WNW / 32;
    constexpr int A_BYTES = TM * 128, STAGE_BYTES = (TM + TN) * 128, GA = TM / 64, GB = TN / 64;  // GA / GB: DMA row groups per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave / WNW, wn = wave % WNW;
    // XCD-aware block -> work mapping (workgroup L runs on XCD L % 8; speed only): the tiles of one k-chunk are
    // consecutive items of one XCD, so their shared A / B panels are fetched from HBM once per XCD.
    const int tiles = a.tiles_m * a.tiles_n;
    const int64_t items = static_cast<int64_t>(a.ksplit) * tiles, per_xcd = (items + 7) / 8;
    const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
    const int64_t item = static_cast<int64_t>(xcd) * per_xcd + j;
    if (j >= per_xcd || item >= items) return;
    const int chunk = static_cast<int>(item / tiles), tile = static_cast<int>(item % tiles);
    const int m0 = (tile / a.tiles_n) * TM, n0 = (tile % a.tiles_n) * TN;
    const int kt_begin = chunk * a.kchunk, kt_end = min(a.KT, kt_begin + a.kchunk);

    // DMA assignment: a wave instruction fills 8 rows x 128 B; wave w owns row groups GA w .. of A and GB w .. of B
    int off_a[GA], off_b[GB];
#pragma unroll
    for (int t = 0; t < GA; ++t) {
        const int row = (wave * GA + t) * 8 + (lane >> 3);
        off_a[t] = min(m0 + row, a.M - 1) * 64 + ((lane & 7) ^ lds_swz(row)) * 8;
    }
#pragma unroll
    for (int t = 0; t < GB; ++t) {
        const int row = (wave * GB + t) * 8 + (lane >> 3);
        off_b[t] = min(n0 + row, a.N - 1) * 64 + ((lane & 7) ^ lds_swz(row)) * 8;
    }
    const int64_t a_kt = static_cast<int64_t>(a.a_rows) * 64, b_kt = static_cast<int64_t>(a.N) * 64;
    // part `part` (0..3) of the DMA requests of one stage: 1/4 of this wave's row groups of A and of B.  The requests of
    // the NEXT k-step are spread over the four MFMA groups of the current one -- issuing all of them up front keeps every
    // wave of the CU in its (50-150 cycles per request) issue phase at the same time, with the matrix pipes idle.
    auto stage_part = [&](int buf, int kt, int part) {
        const uint16_t* ap = a.A + kt * a_kt;
        const uint16_t* bp = a.B + kt * b_kt;
        unsigned char* base = sm + buf * STAGE_BYTES;
#pragma unroll
        for (int t = 0; t < GA; ++t)
            if (t * 4 / GA == part || (GA < 4 && t == part)) glds16(ap + off_a[t], base + (wave * GA + t) * 1024);
#pragma unroll
        for (int t = 0; t < GB; ++t)
            if (t * 4 / GB == part || (GB < 4 && t == part)) glds16(bp + off_b[t], base + A_BYTES + (wave * GB + t) * 1024);
    };
    auto stage = [&](int buf, int kt) {
#pragma unroll
        for (int part = 0; part < 4; ++part) stage_part(buf, kt, part);
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int jn = 0; jn < NI; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jn][r] = 0.0f;

    if (kt_begin < kt_end) {
        const int lr = lane & 31, sw = (lr >> 1) & 7, hi = lane >> 5;
        stage(0, kt_begin);
        __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0) expcnt(0) lgkmcnt(0)
        __syncthreads();
        int buf = 0;
        for (int kt = kt_begin; kt < kt_end; ++kt) {
            const bool more = kt + 1 < kt_end;
            const unsigned char* sa = sm + buf * STAGE_BYTES + (wm * (MI * 32) + lr) * 128;
            const unsigned char* sb = sm + buf * STAGE_BYTES + A_BYTES + (wn * (NI * 32) + lr) * 128;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                if (more) stage_part(buf ^ 1, kt + 1, kk);
                const int co = ((kk * 2 + hi) ^ sw) * 16;
                bf16x8 bv[NI];
#pragma unroll
                for (int jn = 0; jn < NI; ++jn) bv[jn] = *reinterpret_cast<const bf16x8*>(sb + jn * 32 * 128 + co);
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const bf16x8 av = *reinterpret_cast<const bf16x8*>(sa + i * 32 * 128 + co);
#pragma unroll
                    for (int jn = 0; jn < NI; ++jn) acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv[jn], acc[i][jn], 0, 0, 0);
                }
            }
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
            buf ^= 1;
        }
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int jn = 0; jn < NI; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * (MI * 32) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int n = n0 + wn * (NI * 32) + jn * 32 + (lane & 31);
                if (m < a.M && n < a.N) atomicAdd(a.C + static_cast<int64_t>(m) * a.ldc + n, a.alpha * acc[i][jn][r]);
            }
}

// A ring-pipelined variant (half k-tiles through 4 / 5 LDS stages, counted vmcnt, one raw barrier per half step: 96 / 128 KB
// in flight instead of 64) was measured at the SAME rate on every layer shape (profiles/README.md, round 2): the kernel is
// not waiting for its DMA requests, so the two-stage form stays.

// ------------------------------------------------------------------------------------------------
// Rotation GEMM: C[m, n] = alpha * sum_k A[m, k] B[n, k] (+ row_add[n]), bf16 row-major operands (K contiguous) and a bf16
// row-major result -- the eigenbasis rotations X Q of the Lambda stage and of the preconditioner (tracker/factor.py:218-226,
// tracker/precondition.py:102-123), where M = samples x positions is huge and K = N = I' (or O) is a few thousand at most.
// Same 256 x 256 tile / 8 waves / two 64 KB LDS-DMA stages as the score GEMM; no split-K (K is short), the whole result tile
// goes bf16 through the 128 KB of LDS (XOR-swizzled 16-byte chunks) and out in 16-byte row-major stores.
// ------------------------------------------------------------------------------------------------
struct RotateArgs {
    uint16_t* C; int64_t ldc;
    const uint16_t* A; int64_t lda;
    const uint16_t* B; int64_t ldb;
    int M, N, KT;                 // KT = K / 64
    int tiles_m, tiles_n;
    float alpha;
    const float* row_add; int row_add_n;
};

__global__ __launch_bounds__(SV2_THREADS) void rotate_gemm_v2_kernel(RotateArgs a) {
    constexpr int MI = 4, NI = 2, A_BYTES = 256 * 128, STAGE_BYTES = 2 * A_BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 2, wn = wave & 3;
    // XCD-aware order: m-major, n-minor items in 8 contiguous runs -- the n-tiles of an m-tile (same A rows) run on one XCD;
    // B (the eigenvector matrix, a few MB) is L2-resident everywhere
    const int64_t items = static_cast<int64_t>(a.tiles_m) * a.tiles_n, per_xcd = (items + 7) / 8;
    const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
    const int64_t item = static_cast<int64_t>(xcd) * per_xcd + j;
    if (j >= per_xcd || item >= items) return;
    const int m0 = static_cast<int>(item / a.tiles_n) * 256, n0 = static_cast<int>(item % a.tiles_n) * 256;

    const uint16_t* src_a[4];
    const uint16_t* src_b[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int row = (wave * 4 + t) * 8 + (lane >> 3), oct = (lane & 7) ^ lds_swz(row);
        src_a[t] = a.A + static_cast<int64_t>(min(m0 + row, a.M - 1)) * a.lda + oct * 8;
        src_b[t] = a.B + static_cast<int64_t>(min(n0 + row, a.N - 1)) * a.ldb + oct * 8;
    }
    auto stage_part = [&](int buf, int kt, int part) {  // one of the four A and one of the four B requests of this wave
        unsigned char* base = sm + buf * STAGE_BYTES;
        glds16(src_a[part] + kt * 64, base + (wave * 4 + part) * 1024);
        glds16(src_b[part] + kt * 64, base + A_BYTES + (wave * 4 + part) * 1024);
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int jn = 0; jn < NI; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jn][r] = 0.0f;
    {
        const int lr = lane & 31, sw = (lr >> 1) & 7, hi = lane >> 5;
#pragma unroll
        for (int part = 0; part < 4; ++part) stage_part(0, 0, part);
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        int buf = 0;
        for (int kt = 0; kt < a.KT; ++kt) {
            const bool more = kt + 1 < a.KT;
            const unsigned char* sa = sm + buf * STAGE_BYTES + (wm * 128 + lr) * 128;
            const unsigned char* sb = sm + buf * STAGE_BYTES + A_BYTES + (wn * 64 + lr) * 128;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                if (more) stage_part(buf ^ 1, kt + 1, kk);
                const int co = ((kk * 2 + hi) ^ sw) * 16;
                bf16x8 bv[NI];
#pragma unroll
                for (int jn = 0; jn < NI; ++jn) bv[jn] = *reinterpret_cast<const bf16x8*>(sb + jn * 32 * 128 + co);
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const bf16x8 av = *reinterpret_cast<const bf16x8*>(sa + i * 32 * 128 + co);
#pragma unroll
                    for (int jn = 0; jn < NI; ++jn) acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv[jn], acc[i][jn], 0, 0, 0);
                }
            }
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();  // also: every wave is done with the stage buffers before the epilogue reuses them
            buf ^= 1;
        }
    }
    // epilogue: element (ml, nl) of the tile as bf16 at ml * 512 + (((nl >> 3) ^ (ml & 31)) << 4) + (nl & 7) * 2
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int jn = 0; jn < NI; ++jn) {
            const int nl = wn * 64 + jn * 32 + (lane & 31), n = n0 + nl;
            const float add = (a.row_add && n < a.row_add_n) ? a.row_add[n] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ml = wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                uint32_t u = __float_as_uint(a.alpha * acc[i][jn][r] + add);
                if ((u & 0x7fffffffu) > 0x7f800000u) u |= 0x00400000u;
                else u += 0x7fffu + ((u >> 16) & 1u);
                *reinterpret_cast<uint16_t*>(sm + ml * 512 + (((nl >> 3) ^ (ml & 31)) << 4) + (nl & 7) * 2) = static_cast<uint16_t>(u >> 16);
            }
        }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int id = tid + SV2_THREADS * it, ml = id >> 5, ch = id & 31;
        const int m = m0 + ml, n = n0 + ch * 8;
        if (m < a.M && n < a.N)  // N % 8 == 0: a chunk is entirely in or out
            *reinterpret_cast<u32x4*>(a.C + static_cast<int64_t>(m) * a.ldc + n) =
                *reinterpret_cast<const u32x4*>(sm + ml * 512 + ((ch ^ (ml & 31)) << 4));
    }
}

// ------------------------------------------------------------------------------------------------
// Round 3: the same two 256 x 256 kernels on the wave-role-split main loop of kf_pingpong.h (two waves per SIMD half a
// phase apart, counted vmcnt, raw barriers).  Work decomposition, operand layouts and epilogues are those of the v2 kernels
// above, which stay as the 256 x 128 / 128 x 256 shapes and as the KF_ENGINE=2 fallback.
// ------------------------------------------------------------------------------------------------
template <int ISSUE>
__global__ __launch_bounds__(pp::THREADS) void score_gemm_v3_kernel(ScoreV2Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 2, wn = wave & 3;
    const int tiles = a.tiles_m * a.tiles_n;
    const int64_t items = static_cast<int64_t>(a.ksplit) * tiles, per_xcd = (items + 7) / 8;
    const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
    const int64_t item = static_cast<int64_t>(xcd) * per_xcd + j;
    if (j >= per_xcd || item >= items) return;
    const int chunk = static_cast<int>(item / tiles), tile = static_cast<int>(item % tiles);
    const int m0 = (tile / a.tiles_n) * 256, n0 = (tile % a.tiles_n) * 256;
    const int kt_begin = chunk * a.kchunk, kt_end = min(a.KT, kt_begin + a.kchunk);
    if (kt_begin >= kt_end) return;

    pp::Sources src;
    const int64_t kt_a = static_cast<int64_t>(a.a_rows) * 64, kt_b = static_cast<int64_t>(a.N) * 64;
    const uint16_t* abase = a.A + kt_begin * kt_a;
    const uint16_t* bbase = a.B + kt_begin * kt_b;
    pp::make_sources(src, wave, lane,
                     [&](int row) { return abase + static_cast<int64_t>(min(m0 + row, a.M - 1)) * 64; },
                     [&](int row) { return bbase + static_cast<int64_t>(min(n0 + row, a.N - 1)) * 64; });
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jn][r] = 0.0f;
    pp::mainloop<ISSUE>(acc, sm, src, kt_end - kt_begin, wave, lane, [&](int t) { return t * kt_a; }, [&](int t) { return t * kt_b; });
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int n = n0 + wn * 64 + jn * 32 + (lane & 31);
                if (m < a.M && n < a.N) atomicAdd(a.C + static_cast<int64_t>(m) * a.ldc + n, a.alpha * acc[i][jn][r]);
            }
}

constexpr int PP64_SMEM = pp64::Geo<256, 128>::SMEM_BYTES;   // == Geo<128, 256>::SMEM_BYTES: three stages of 48 KB
// Round 4: the 256 x 128 / 128 x 256 shapes of the score GEMM on the wave-role-split loop for waves of 64 x 64
// (kf_pingpong64.h): GPT-2's train batches of 128 sequences, query counts just above a multiple of 256.  Same work items and
// operand layouts as above.
template <int TA, int TB>
__global__ __launch_bounds__(pp64::THREADS) void score_gemm_v4_kernel(ScoreV2Args a) {
    using G = pp64::Geo<TA, TB>;
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave / G::WB, wn = wave % G::WB;
    const int tiles = a.tiles_m * a.tiles_n;
    const int64_t items = static_cast<int64_t>(a.ksplit) * tiles, per_xcd = (items + 7) / 8;
    const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
    const int64_t item = static_cast<int64_t>(xcd) * per_xcd + j;
    if (j >= per_xcd || item >= items) return;
    const int chunk = static_cast<int>(item / tiles), tile = static_cast<int>(item % tiles);
    const int m0 = (tile / a.tiles_n) * TA, n0 = (tile % a.tiles_n) * TB;
    const int kt_begin = chunk * a.kchunk, kt_end = min(a.KT, kt_begin + a.kchunk);
    if (kt_begin >= kt_end) return;

    pp64::Sources<TA, TB> src;
    const int64_t kt_a = static_cast<int64_t>(a.a_rows) * 64, kt_b = static_cast<int64_t>(a.N) * 64;
    const uint16_t* abase = a.A + kt_begin * kt_a;
    const uint16_t* bbase = a.B + kt_begin * kt_b;
    pp64::make_sources<TA, TB>(src, wave, lane,
                               [&](int row) { return abase + static_cast<int64_t>(min(m0 + row, a.M - 1)) * 64; },
                               [&](int row) { return bbase + static_cast<int64_t>(min(n0 + row, a.N - 1)) * 64; });
    f32x16 acc[2][2];
    zero_acc(acc);
    pp64::PlainCtl ctl;
    pp64::mainloop<TA, TB, 6>(acc, sm, src, kt_end - kt_begin, wave, lane, [&](int t) { return t * kt_a; }, [&](int t) { return t * kt_b; }, ctl);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int n = n0 + wn * 64 + jn * 32 + (lane & 31);
                if (m < a.M && n < a.N) atomicAdd(a.C + static_cast<int64_t>(m) * a.ldc + n, a.alpha * acc[i][jn][r]);
            }
}

// Round 4: 512 x 128 (4 x 2 waves of 128 x 64) and 128 x 512 (1 x 8) shapes on the generic wave grid of kf_pingpong.h (ppw):
// the two-phase loop of score_gemm_v3_kernel with 10 DMA requests per wave and k-tile instead of 8, all 160 KB of LDS.
template <int WM, int WN>
__global__ __launch_bounds__(pp::THREADS) void score_gemm_v5_kernel(ScoreV2Args a) {
    using G = ppw::Geo<WM, WN>;
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave / WN, wn = wave % WN;
    const int tiles = a.tiles_m * a.tiles_n;
    const int64_t items = static_cast<int64_t>(a.ksplit) * tiles, per_xcd = (items + 7) / 8;
    const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
    const int64_t item = static_cast<int64_t>(xcd) * per_xcd + j;
    if (j >= per_xcd || item >= items) return;
    const int chunk = static_cast<int>(item / tiles), tile = static_cast<int>(item % tiles);
    const int m0 = (tile / a.tiles_n) * G::TA, n0 = (tile % a.tiles_n) * G::TB;
    const int kt_begin = chunk * a.kchunk, kt_end = min(a.KT, kt_begin + a.kchunk);
    if (kt_begin >= kt_end) return;

    ppw::Sources<WM, WN> src;
    const int64_t kt_a = static_cast<int64_t>(a.a_rows) * 64, kt_b = static_cast<int64_t>(a.N) * 64;
    const uint16_t* abase = a.A + kt_begin * kt_a;
    const uint16_t* bbase = a.B + kt_begin * kt_b;
    ppw::make_sources<WM, WN>(src, wave, lane,
                              [&](int row) { return abase + static_cast<int64_t>(min(m0 + row, a.M - 1)) * 64; },
                              [&](int row) { return bbase + static_cast<int64_t>(min(n0 + row, a.N - 1)) * 64; });
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jn][r] = 0.0f;
    ppw::mainloop<WM, WN>(acc, sm, src, kt_end - kt_begin, wave, lane, [&](int t) { return t * kt_a; }, [&](int t) { return t * kt_b; });
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int n = n0 + wn * 64 + jn * 32 + (lane & 31);
                if (m < a.M && n < a.N) atomicAdd(a.C + static_cast<int64_t>(m) * a.ldc + n, a.alpha * acc[i][jn][r]);
            }
}
constexpr int PPW_SMEM = 2 * (512 + 128) * 128;   // == ppw::Geo<4, 2>::SMEM_BYTES == ppw::Geo<1, 8>::SMEM_BYTES: all of the CU's LDS

// ------------------------------------------------------------------------------------------------
// Lambda of a Linear layer on sequences (reference module/linear.py:112-122 + module/tracker/factor.py:218-226):
//     Lambda[o, i] += scale^2 * sum_n ( sum_r GtT[n][o][r] AtT[n][i][r] )^2
// over K-contiguous rotated factors GtT[n][O][R], AtT[n][W][R] (kf_rotate_rows_transposed_bf16 writes them).  One work item =
// (sample range, 256 x 128 tile of Lambda): the k-tiles of all its samples run through the 64 x 64-wave loop WITHOUT a break in
// the DMA pipeline; at the end of a sample every lane folds its 64 accumulators, squared, into 64 running sums (the first
// MFMAs of the next sample take C = 0), and the item ends with ONE coalesced fp32 atomic per element.  The round-2 kernel
// (lambda_bf16_kernel: 128 x 128 register-staged TN tiles) re-read the factors 3.6x and ran at 12 % of the MFMA peak.
// ------------------------------------------------------------------------------------------------
struct LambdaRowsArgs {
    float* L; int64_t ldl;
    const uint16_t* G; const uint16_t* A;   // [batch][O][R], [batch][W][R]
    int O, W, Ip, KS;                       // KS = R / 64
    int batch, tiles_m, tiles_n, zchunk, zblocks;
    float scale2;
};

struct LambdaFold {
    f32x16 (&sum)[2][2];
    int ks, KS;
    __device__ __forceinline__ bool first(int) const { return ks == 0; }
    __device__ __forceinline__ void done(int, f32x16 (&acc)[2][2]) {
        if (++ks == KS) {   // (wave-uniform) end of a sample
            ks = 0;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int jn = 0; jn < 2; ++jn)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sum[i][jn][r] = fmaf(acc[i][jn][r], acc[i][jn][r], sum[i][jn][r]);
        }
    }
};

template <int LREQ>
__global__ __launch_bounds__(pp64::THREADS) void lambda_rows_kernel(LambdaRowsArgs a) {
    using G = pp64::Geo<256, 128>;
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave / G::WB, wn = wave % G::WB;
    // XCD-aware order as the covariance kernels: sample range major, so the tiles of a sample range (which share its rotated
    // factors) are consecutive items of one XCD
    const int tiles = a.tiles_m * a.tiles_n;
    const int64_t items = static_cast<int64_t>(a.zblocks) * tiles, per_xcd = (items + 7) / 8;
    const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
    const int64_t item = static_cast<int64_t>(xcd) * per_xcd + j;
    if (j >= per_xcd || item >= items) return;
    const int zb = static_cast<int>(item / tiles), tile = static_cast<int>(item % tiles);
    const int m0 = (tile / a.tiles_n) * 256, n0 = (tile % a.tiles_n) * 128;
    const int z_begin = zb * a.zchunk, z_end = min(a.batch, z_begin + a.zchunk);
    if (z_begin >= z_end) return;

    const int64_t R = static_cast<int64_t>(a.KS) * 64, stride_g = a.O * R, stride_a = a.W * R;
    pp64::Sources<256, 128> src;
    const uint16_t* gbase = a.G + z_begin * stride_g;
    const uint16_t* abase = a.A + z_begin * stride_a;
    pp64::make_sources<256, 128>(src, wave, lane,
                                 [&](int row) { return gbase + static_cast<int64_t>(min(m0 + row, a.O - 1)) * R; },
                                 [&](int row) { return abase + static_cast<int64_t>(min(n0 + row, a.W - 1)) * R; });
    const int kshift = (a.KS & (a.KS - 1)) == 0 ? __builtin_ctz(a.KS) : -1;
    auto split = [&](int t, int& zq, int& ks) { zq = kshift >= 0 ? t >> kshift : t / a.KS; ks = t - zq * a.KS; };
    f32x16 acc[2][2], sum[2][2];
    zero_acc(acc);
    zero_acc(sum);
    LambdaFold ctl{sum, 0, a.KS};
    pp64::mainloop<256, 128, LREQ>(acc, sm, src, (z_end - z_begin) * a.KS, wave, lane,
                             [&](int t) { int zq, ks; split(t, zq, ks); return zq * stride_g + ks * 64; },
                             [&](int t) { int zq, ks; split(t, zq, ks); return zq * stride_a + ks * 64; }, ctl);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int c = n0 + wn * 64 + jn * 32 + (lane & 31);
                if (o < a.O && c < a.Ip) atomicAdd(a.L + static_cast<int64_t>(o) * a.ldl + c, a.scale2 * sum[i][jn][r]);
            }
}

// RotateArgs with two additions: col_add (per output ROW m: the bias row when the roles of the operands are swapped to get a
// transposed result) and a blocked result layout -- element (m, n) at (n / c_inner) * c_outer + m * c_inner + n % c_inner
// (c_inner == 0: plain row-major with ldc) -- which is how "Gt^T[n][o][r]" (K-contiguous per sample) is written.
struct RotateV3Args {
    RotateArgs r;
    const float* col_add; int col_add_m;
    int64_t c_inner, c_outer;
    int n_major;   // work order: consecutive items share the B (n) tile instead of the A (m) tile
    // xcd_m = gm in {2, 4, 8} (0: the linear order above): the 8 XCDs form a gm x (8 / gm) grid over (m-tiles, n-tiles) -- an XCD owns
    // ceil(tiles_m / gm) m-tiles, whose A rows (an eigenvector slab) then stay in ITS 4 MB L2 across all of its n-tiles, at the price
    // of every B tile being read by gm XCDs.  See rotate_xcd_m().
    int xcd_m;
    // EPI == 1 ("sum of squares over row groups"): rows are ordered (group, member) with `group_rows` members per group;
    // sumsq[group * ld_sumsq + n] += alpha^2 * sum_member C[(group, member), n]^2 and C itself is never stored.
    float* sumsq; int64_t ld_sumsq; int group_rows;
};

// EPI 0: bf16 result through LDS.  EPI 1: the Lambda reduction of the dense (per-sample-gradient) form -- see
// kf_lambda_conv2d_accum: rows = (o, sample), so a 256-row tile spans at most two values of o when group_rows >= 256.
template <int EPI, int ISSUE>
__global__ __launch_bounds__(pp::THREADS) void rotate_gemm_v3_kernel(RotateV3Args v) {
    const RotateArgs& a = v.r;
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 2, wn = wave & 3;
    const int64_t items = static_cast<int64_t>(a.tiles_m) * a.tiles_n, per_xcd = (items + 7) / 8;
    const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
    int tm, tn;
    if (v.xcd_m > 0) {
        const int gm = v.xcd_m, gn = 8 / gm;
        const int mper = (a.tiles_m + gm - 1) / gm, nper = (a.tiles_n + gn - 1) / gn;
        if (j >= mper * nper) return;
        tm = (xcd % gm) * mper + j % mper;   // consecutive workgroups of an XCD: the same B tile against the XCD's few A tiles
        tn = (xcd / gm) * nper + j / mper;
        if (tm >= a.tiles_m || tn >= a.tiles_n) return;
    } else {
        const int64_t item = static_cast<int64_t>(xcd) * per_xcd + j;
        if (j >= per_xcd || item >= items) return;
        tm = v.n_major ? static_cast<int>(item % a.tiles_m) : static_cast<int>(item / a.tiles_n);
        tn = v.n_major ? static_cast<int>(item / a.tiles_m) : static_cast<int>(item % a.tiles_n);
    }
    const int m0 = tm * 256, n0 = tn * 256;

    pp::Sources src;
    pp::make_sources(src, wave, lane,
                     [&](int row) { return a.A + static_cast<int64_t>(min(m0 + row, a.M - 1)) * a.lda; },
                     [&](int row) { return a.B + static_cast<int64_t>(min(n0 + row, a.N - 1)) * a.ldb; });
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jn][r] = 0.0f;
    pp::mainloop<ISSUE>(acc, sm, src, a.KT, wave, lane, [](int t) { return t * 64; }, [](int t) { return t * 64; });
    if constexpr (EPI == 1) {
        const int first = m0 / v.group_rows;                          // group of the tile's first row
        const int limit = min(256, a.M - m0);                         // rows of the tile that exist
        const int boundary = min(limit, (first + 1) * v.group_rows - m0);   // in-tile row where the next group starts
        const float a2 = a.alpha * a.alpha;
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) {
            float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ml = wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const float sq = acc[i][jn][r] * acc[i][jn][r];
                    s0 += ml < boundary ? sq : 0.0f;
                    s1 += (ml >= boundary && ml < limit) ? sq : 0.0f;
                }
            s0 += __shfl_xor(s0, 32);   // the other 16 rows of every 32 x 32 block sit in lane ^ 32
            s1 += __shfl_xor(s1, 32);
            const int n = n0 + wn * 64 + jn * 32 + (lane & 31);
            if (lane < 32 && n < a.N) {
                atomicAdd(v.sumsq + static_cast<int64_t>(first) * v.ld_sumsq + n, a2 * s0);
                if (boundary < limit) atomicAdd(v.sumsq + static_cast<int64_t>(first + 1) * v.ld_sumsq + n, a2 * s1);
            }
        }
        return;
    }
    __syncthreads();  // every wave is done with the stage buffers: the epilogue reuses them
    // epilogue: element (ml, nl) of the tile as bf16 at ml * 512 + (((nl >> 3) ^ (ml & 31)) << 4) + (nl & 7) * 2
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) {
            const int nl = wn * 64 + jn * 32 + (lane & 31), n = n0 + nl;
            const float add = (a.row_add && n < a.row_add_n) ? a.row_add[n] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ml = wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const float cadd = (v.col_add && m0 + ml < v.col_add_m) ? v.col_add[m0 + ml] : 0.0f;
                uint32_t u = __float_as_uint(a.alpha * acc[i][jn][r] + add + cadd);
                if ((u & 0x7fffffffu) > 0x7f800000u) u |= 0x00400000u;
                else u += 0x7fffu + ((u >> 16) & 1u);
                *reinterpret_cast<uint16_t*>(sm + ml * 512 + (((nl >> 3) ^ (ml & 31)) << 4) + (nl & 7) * 2) = static_cast<uint16_t>(u >> 16);
            }
        }
    __syncthreads();
    // 512 threads = 16 rows x 32 chunks per pass: a thread's chunk column is the same in all 16 passes
    const int ch = tid & 31, n = n0 + ch * 8;
    if (n < a.N) {  // N % 8 == 0 (and c_inner % 8 == 0): a chunk is entirely in or out, never across blocks
        const int inner = static_cast<int>(v.c_inner);
        const int64_t col = inner ? static_cast<int64_t>(n / inner) * v.c_outer + n % inner : n;
        const int64_t row_stride = inner ? inner : a.ldc;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int ml = (tid >> 5) + 16 * it, m = m0 + ml;
            if (m < a.M)
                *reinterpret_cast<u32x4*>(a.C + col + m * row_stride) = *reinterpret_cast<const u32x4*>(sm + ml * 512 + ((ch ^ (ml & 31)) << 4));
        }
    }
}

// 256 x 128 / 128 x 256 score shapes: 4 = wave-role-split loop of kf_pingpong64.h (round 4), 2 = the round-2 lock-step loop
// (KF_ENGINE=2 or KF_HALF_TILE_ENGINE=2: A/B measurements, fallback)
inline int engine_generation();
inline int half_tile_engine() {
    const char* e = getenv("KF_HALF_TILE_ENGINE");
    if (e && atoi(e) == 2) return 2;
    return engine_generation() == 3 ? 4 : 2;
}

// 512 x 128 / 128 x 512 tiles (score_gemm_v5_kernel): measured within 4 % of the 256 x 128 loop on the shapes that take them
// (GPT-2: 614 vs 588 us per launch, profiles/r04_ab_kernel_stats.csv) -- both sit on the HBM stream of P there -- so they are
// opt-in (KF_WIDE_TILE=1: measurements, tests)
inline bool wide_tile_enabled() { const char* e = getenv("KF_WIDE_TILE"); return e && atoi(e) == 1; }

// Where the 256 x 256 loop issues its LDS-DMA requests (kf_pingpong.h, ISSUE): 0 = in the L segments (round 3), 1 = between the
// MFMA groups of the M segments, 2 = as 1 with A0 left in the short L segment.  KF_PP_ISSUE overrides (read per call: A/B
// measurements in one process, race screens on every schedule).
constexpr int PP_ISSUE_DEFAULT = 1;
inline int pp_issue(int preferred) {
    const char* e = getenv("KF_PP_ISSUE");
    return (e && e[0] >= '0' && e[0] <= '2' && e[1] == 0) ? e[0] - '0' : preferred;
}
// `preferred`: the schedule measured fastest for the calling kernel (profiles/r04_issue_ab_kernel_stats.csv: 2 for the covariance
// kernel, whose L segments also carry the offset-table read; 1 for the others)
template <class F>
inline void with_pp_issue(F&& launch, int preferred = PP_ISSUE_DEFAULT) {
    switch (pp_issue(preferred)) {
    case 0: launch(std::integral_constant<int, 0>{}); break;
    case 2: launch(std::integral_constant<int, 2>{}); break;
    default: launch(std::integral_constant<int, 1>{}); break;
    }
}

inline int engine_generation() {  // KF_ENGINE=2 forces the round-2 main loop (A/B measurements, fallback)
    const char* e = getenv("KF_ENGINE");
    return (e && atoi(e) == 2) ? 2 : 3;
}

// Round 5: sequence layers on the K-major loop (kf_pingpong_tn.h) -- no transposed copies of the hooked tensors.  KF_TN=0 switches
// back to transpose_rows + the K-contiguous kernels (A/B measurements, fallback); KF_TN_IMG = 0 / 1 / 2 picks the LDS image
// (kf_tn_map.h; read per call).
constexpr int TN_IMAGE_DEFAULT = 0;   // measured (profiles/r05_tn_ab.log): 256-byte global rows per request win on the large shapes
inline bool tn_enabled() {
    const char* e = getenv("KF_TN");
    return engine_generation() == 3 && !(e && e[0] == '0' && e[1] == 0);
}
template <class F>
inline void with_tn_image(F&& launch) {
    const char* e = getenv("KF_TN_IMG");
    const int image = (e && e[0] >= '0' && e[0] <= '2' && e[1] == 0) ? e[0] - '0' : TN_IMAGE_DEFAULT;
    switch (image) {
    case 1: launch(std::integral_constant<int, 1>{}); break;
    case 2: launch(std::integral_constant<int, 2>{}); break;
    default: launch(std::integral_constant<int, 0>{}); break;
    }
}

// ------------------------------------------------------------------------------------------------
// Per-sample gradient: out[n][m, i] = sum_k A[n][m, k] B[n][i, k], bf16, written k-tile-major over d = m * N + i.
// 256 threads = 4 wave64 (2 x 2), 128 x 128 tile, k-step 64, two LDS stages of 32 KB -> two workgroups per CU.
// B rows are plain ([n][i][k]) or implicit-im2col rows of a zero-padded, column-phase-split copy of the input.
// ------------------------------------------------------------------------------------------------
struct PsgV2Args {
    uint16_t* out; int64_t out_tile_stride;     // element (n, d) at (d >> 6) * out_tile_stride + n * 64 + (d & 63)
    int out_rows;                               // != 0: plain rows ordered (m, sample) instead: element at ((m * batch + n) * N + i)
    const uint16_t* A; int64_t a_sample_stride; // [batch][M][K]
    const uint16_t* B; int64_t b_sample_stride; // plain: [batch][N][K]; conv: [phase][batch][C][Hp][Wq]
    int M, N, K;                                // K % 64 == 0
    int batch, tiles_m, tiles_n;
    int n_begin;                                // columns [n_begin, N) are produced (n_begin % 128 == 0; the 256 x 256 kernel owns the rest)
    // out_rows == 2 ("plain"): element (n, m, i) at (n * M + m) * N + i, times mul[m * ld_mul + i] for i < mul_n and 0 beyond --
    // the eigenbasis gradient of a query times 1 / (lambda + damping) (kf_precondition)
    const float* mul; int ld_mul, mul_n;
    // implicit im2col (conv != 0): row i = shift * C + c, shift = ky * k2 + kx;  k = oy * O2 + ox (O2 % 8 == 0);
    // element = Xs[(kx * d2) % s2][n][c][s1 * oy + ky * d1][ox + (kx * d2) / s2]
    int conv, C, k2, O2, s1, d1, s2, d2, Wq, plane /* Hp * Wq */;
    int64_t phase_stride;                       // batch * C * plane
};

constexpr int PV2_OPERAND_BYTES = 128 * 128;
constexpr int PV2_STAGE_BYTES = 2 * PV2_OPERAND_BYTES;
constexpr int PV2_KTAB_STEPS = 64;                       // k-steps (of 64 positions) the implicit-im2col offset table holds
constexpr int PV2_SMEM = 2 * PV2_STAGE_BYTES + PV2_KTAB_STEPS * 8 * 4;

// Implicit-im2col offset of k-octet `oc` of k-step `ks` -- oy * s1 * Wq + ox for position p0 = 64 ks + 8 oc -- for every (ks, oc)
// of a sample, once per workgroup, into the LDS tail: the division by the output width used to be paid per DMA request.
template <class Args>
__device__ __forceinline__ int* conv_offset_table(const Args& a, unsigned char* sm, int K, int nthreads) {
    int* tab = reinterpret_cast<int*>(sm + 2 * PV2_STAGE_BYTES);
    const int entries = min(K >> 6, PV2_KTAB_STEPS) * 8;
    for (int e = threadIdx.x; e < entries; e += nthreads) {
        const int p0 = (e >> 3) * 64 + (e & 7) * 8, oy = p0 / a.O2, ox = p0 - oy * a.O2;
        tab[e] = oy * a.s1 * a.Wq + ox;
    }
    return tab;
}

__global__ __launch_bounds__(NTHREADS) void psg_gemm_v2_kernel(PsgV2Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    // Work order (per XCD a contiguous range of items): blocks of PSG_ZB consecutive samples; inside a block tile-major,
    // sample-minor.  The workgroups running together on an XCD then cover a few samples x all their tiles -- the samples'
    // inputs (G, the padded input copy: ~0.2 MB each) stay in that XCD's L2 across their tiles -- and the PSG_ZB samples of
    // one tile write ADJACENT 128-byte pieces of the k-tile-major gradient buffer (element (n, d) sits at
    // (d / 64) * b * 64 + n * 64 + d % 64), i.e. 1 KB runs instead of isolated lines 128 KB apart.
    constexpr int PSG_ZB = 8;
    const int tiles = a.tiles_m * a.tiles_n;
    const int64_t zblocks = (a.batch + PSG_ZB - 1) / PSG_ZB;
    const int64_t items = zblocks * PSG_ZB * tiles, per_xcd = (items + 7) / 8;
    const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
    const int64_t item = static_cast<int64_t>(xcd) * per_xcd + j;
    if (j >= per_xcd || item >= items) return;
    const int64_t zb = item / (static_cast<int64_t>(tiles) * PSG_ZB);
    const int rem = static_cast<int>(item - zb * tiles * PSG_ZB);
    const int tile = rem / PSG_ZB, z = static_cast<int>(zb) * PSG_ZB + rem % PSG_ZB;
    if (z >= a.batch) return;
    const int m0 = (tile / a.tiles_n) * 128, n0 = a.n_begin + (tile % a.tiles_n) * 128;

    // per-lane DMA sources: 4 row groups of each operand per wave; the k-octet this lane fetches is chunk_src
    const uint16_t* src_a[4];
    const uint16_t* src_b[4];
    int oct[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int row = (wave * 4 + t) * 8 + (lane >> 3);
        oct[t] = (lane & 7) ^ lds_swz(row);
        const int m = min(m0 + row, a.M - 1);
        src_a[t] = a.A + static_cast<int64_t>(z) * a.a_sample_stride + static_cast<int64_t>(m) * a.K + oct[t] * 8;
        const int i = min(n0 + row, a.N - 1);
        if (a.conv) {
            const int shift = i / a.C, c = i - shift * a.C;
            const int ky = shift / a.k2, kx = shift - ky * a.k2;
            const int col = kx * a.d2, phase = col % a.s2, coff = col / a.s2;
            src_b[t] = a.B + phase * a.phase_stride + static_cast<int64_t>(z) * a.b_sample_stride +
                       static_cast<int64_t>(c) * a.plane + ky * a.d1 * a.Wq + coff;
        } else {
            src_b[t] = a.B + static_cast<int64_t>(z) * a.b_sample_stride + static_cast<int64_t>(i) * a.K + oct[t] * 8;
        }
    }
    const int* ktab = nullptr;
    if (a.conv && (a.K >> 6) <= PV2_KTAB_STEPS) {
        ktab = conv_offset_table(a, sm, a.K, NTHREADS);
        __syncthreads();
    }
    auto stage = [&](int buf, int k0) {
        unsigned char* base = sm + buf * PV2_STAGE_BYTES + wave * 4096;
#pragma unroll
        for (int t = 0; t < 4; ++t) glds16(src_a[t] + k0, base + t * 1024);
        if (a.conv) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                int off;
                if (ktab) off = ktab[(k0 >> 6) * 8 + oct[t]];
                else { const int p0 = k0 + oct[t] * 8, oy = p0 / a.O2, ox = p0 - oy * a.O2; off = oy * a.s1 * a.Wq + ox; }
                glds16(src_b[t] + off, base + PV2_OPERAND_BYTES + t * 1024);
            }
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) glds16(src_b[t] + k0, base + PV2_OPERAND_BYTES + t * 1024);
        }
    };

    f32x16 acc[2][2];
    zero_acc(acc);
    {
        // Two stages of 32 KB -> two workgroups per CU, whose load and MFMA phases interleave.  (A 4-stage ring with
        // counted vmcnt and one workgroup per CU was measured 10-15 % SLOWER on every layer shape: with K of only 1-4
        // k-steps the second resident workgroup hides more latency than a deeper ring does.)
        const int lr = lane & 31, sw = (lr >> 1) & 7, hi = lane >> 5;
        stage(0, 0);
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        int buf = 0;
        for (int k0 = 0; k0 < a.K; k0 += 64) {
            if (k0 + 64 < a.K) stage(buf ^ 1, k0 + 64);
            const unsigned char* sa = sm + buf * PV2_STAGE_BYTES + (wm * 64 + lr) * 128;
            const unsigned char* sb = sm + buf * PV2_STAGE_BYTES + PV2_OPERAND_BYTES + (wn * 64 + lr) * 128;
            wave_kstep_64x64(acc, sa, sb, hi, sw);
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
            buf ^= 1;
        }
    }
    // epilogue: bf16 through LDS (pitch 272 B), then 16 bytes per lane into the k-tile-major gradient buffer
    constexpr int OP = 272;
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ml = acc_row(wm, ti, r, lane), nl = acc_col(wn, tj, lane);
                uint32_t u = __float_as_uint(acc[ti][tj][r]);
                if ((u & 0x7fffffffu) > 0x7f800000u) u |= 0x00400000u;
                else u += 0x7fffu + ((u >> 16) & 1u);
                *reinterpret_cast<uint16_t*>(sm + ml * OP + nl * 2) = static_cast<uint16_t>(u >> 16);
            }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int id = tid + 256 * it, ml = id >> 4, ch = id & 15;
        const int m = m0 + ml, n = n0 + ch * 8;
        if (m < a.M && n < a.N) {  // N % 8 == 0: a chunk is entirely in or out
            const int64_t d = static_cast<int64_t>(m) * a.N + n;
            const int64_t idx = a.out_rows ? (static_cast<int64_t>(m) * a.batch + z) * a.N + n
                                           : (d >> 6) * a.out_tile_stride + static_cast<int64_t>(z) * 64 + (d & 63);
            *reinterpret_cast<u32x4*>(a.out + idx) = *reinterpret_cast<const u32x4*>(sm + ml * OP + ch * 16);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Round 3: the same per-sample-gradient tiles from PERSISTENT workgroups.  A workgroup walks a strided list of (sample,
// tile) items of its XCD's range and keeps the two-stage LDS-DMA pipeline running ACROSS item boundaries: while the last
// k-step of an item is being multiplied, the first k-step of the next item is already on its way.  In the v2 kernel every item (1-16 k-steps: K = R = 64 .. 1024) paid a full DMA round trip before its first
// MFMA and there was one workgroup launch per item.  The epilogue stages its bf16 tile in the stage buffer that was consumed
// last (128 rows x 256 B = 32 KB exactly), so two workgroups share a CU as before.
//
// An item is only 16-64 MFMAs per wave, so what the wave executes AROUND them decides the rate (the first persistent version
// issued ~1 800 vector instructions per item -- per-row divisions of the implicit-im2col decode, 64 two-byte LDS stores,
// 64-bit index arithmetic -- and its matrix pipe was busy 11 % of the time).  Now:
//   * everything about a DMA source that does not depend on the sample lives in an LDS table built once per workgroup
//     (`rowtab`: implicit-im2col row -> element offset into the phase copies), next to the per-k-step position table; an item's
//     decode is two scalar divisions, 8 table lookups and 8 adds, its sample base is wave-uniform (SGPR pair);
//   * the accumulators are TRANSPOSED (MFMA rows run along i, the contiguous direction of the result), so a lane owns four
//     consecutive bf16 of a result row: 16 packed 8-byte LDS stores per lane instead of 64 two-byte ones;
//   * the first k-step accumulates onto an inline zero instead of 64 cleared registers, and the 16-byte global stores of an
//     item step by one 64-bit stride.
// ------------------------------------------------------------------------------------------------
constexpr int PV2_ROWTAB_MAX = 3072;   // implicit-im2col rows the row table holds: 2 x (66 + 12) KB of LDS = two workgroups per CU
constexpr int PV3_SMEM_MAX = PV2_SMEM + PV2_ROWTAB_MAX * 4;

// sample-independent part of the DMA source of implicit-im2col row i = shift * C + c: element offset into the phase copies
__device__ __forceinline__ int conv_row_offset(const PsgV2Args& a, int i) {
    const int shift = i / a.C, c = i - shift * a.C;
    const int ky = shift / a.k2, kx = shift - ky * a.k2;
    const int col = kx * a.d2, phase = col % a.s2, coff = col / a.s2;
    return static_cast<int>(phase * a.phase_stride) + c * a.plane + ky * a.d1 * a.Wq + coff;
}

struct PsgItem {   // (wave-uniform)
    const uint16_t* pa;   // sample bases
    const uint16_t* pb;
    int m0, n0, z;
};
struct PsgLaneOffsets {   // element offsets of this lane's 4 + 4 DMA requests of an item, k-step 0
    int a[4], b[4];
};

// One 64-deep k-step with the operands swapped: accumulator rows run along the B image (i), columns along the A image (m).
// ZERO: the first k-step of an item accumulates onto an inline 0.
template <bool ZERO>
__device__ __forceinline__ void wave_kstep_transposed(f32x16 (&acc)[2][2], const unsigned char* sa, const unsigned char* sb, int hi, int sw) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int co = ((kk * 2 + hi) ^ sw) * 16;
        const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(sa + co);
        const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(sa + 32 * 128 + co);
        const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(sb + co);
        const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(sb + 32 * 128 + co);
        if (ZERO && kk == 0) {
            const f32x16 zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, a0, zero, 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, a1, zero, 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a0, zero, 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a1, zero, 0, 0, 0);
        } else {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, a0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, a1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a1, acc[1][1], 0, 0, 0);
        }
    }
}

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const bf16x2 v = __builtin_convertvector(f32x2{lo, hi}, bf16x2);
    return __builtin_bit_cast(uint32_t, v);
}

// ROWS: the result is written as plain rows ordered (m, sample) -- the A operand of the dense-form Lambda GEMM
// (kf_lambda_conv2d_accum) -- instead of k-tile-major for the score GEMM.  Its own instantiation so that the two uses have
// their own kernel names in a profile.
//
// Work order: groups of (block of 8 consecutive samples, tile), block-major; XCD x owns a contiguous range of groups, its
// workgroup j serves sample j % 8 of the groups j / 8, j / 8 + gridDim / 64, ...  The workgroups running together on an XCD
// then cover a few samples x all their tiles -- the samples' inputs stay in that XCD's L2 across their tiles -- and the 8
// samples of one tile write ADJACENT 128-byte pieces of the k-tile-major gradient buffer.
// (Two latency experiments on this kernel, both measured and neither adopted -- profiles/r03_psg_wave_roles_negative.log,
// r03_psg_deferred_wait_negative.log: waves specialised into DMA issuers and storers so that no wave waits for its own stores
// at the next "DMA landed" (6-11 % slower), and the DMA wait moved to the top of the consuming k-step with a counted vmcnt
// over the copy-out stores, so that the next item's first k-step stays in flight through the epilogue (no change).  The
// kernel is bound by memory-system throughput -- its 32 KB result tile per 16-64 MFMAs and the gather-like implicit-im2col
// requests -- not by the latency of the item boundary.)
// (Round 6: the tile stored straight from the accumulators instead -- 8-byte quads, or 16-byte octets after a v_permlane32_swap --
// was measured 41 % / 14 % slower: a store instruction then touches 32 result rows x 16-32 bytes instead of 8 rows x 128,
// profiles/r06_psg_register_stores_negative.log.)
template <int OUT>   // 0: k-tile-major (score GEMM operand), 1: rows ordered (m, sample), 2: plain per sample, times `mul`
__global__ __launch_bounds__(NTHREADS) void psg_gemm_v3_kernel(PsgV2Args a) {
    constexpr bool ROWS = OUT == 1, PLAIN = OUT == 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 1, wn = wave & 1;
    constexpr int PSG_ZB = 8;
    const int tiles = a.tiles_m * a.tiles_n;
    const int groups = ((a.batch + PSG_ZB - 1) / PSG_ZB) * tiles, per_xcd = (groups + 7) / 8;
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3, zsub = j & 7, gstride = gridDim.x >> 6;
    const int last = min(groups, (xcd + 1) * per_xcd);
    int group = xcd * per_xcd + (j >> 3);

    int* ktab = reinterpret_cast<int*>(sm + 2 * PV2_STAGE_BYTES);
    int* rowtab = ktab + PV2_KTAB_STEPS * 8;
    const bool k_table = a.conv && (a.K >> 6) <= PV2_KTAB_STEPS, row_table = a.conv && a.N <= PV2_ROWTAB_MAX;
    if (a.conv) {
        if (k_table) conv_offset_table(a, sm, a.K, NTHREADS);
        if (row_table)
            for (int i = tid; i < a.N; i += NTHREADS) rowtab[i] = conv_row_offset(a, i);
        __syncthreads();
    }
    // this thread's four DMA rows of an operand tile and the k-octet it fetches of each (item independent)
    int row[4], oct8[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        row[t] = (wave * 4 + t) * 8 + (lane >> 3);
        oct8[t] = ((lane & 7) ^ lds_swz(row[t])) * 8;
    }
    auto decode = [&](int g, PsgItem& it) -> bool {   // (wave-uniform result)
        if (g >= last) return false;
        const int zb = g / tiles, tile = g - zb * tiles;
        it.z = zb * PSG_ZB + zsub;
        if (it.z >= a.batch) return false;   // padding sample of the last block of 8: nothing follows for this workgroup
        const int tm = tile / a.tiles_n;
        it.m0 = tm * 128;
        it.n0 = a.n_begin + (tile - tm * a.tiles_n) * 128;
        it.pa = a.A + static_cast<int64_t>(it.z) * a.a_sample_stride;
        it.pb = a.B + static_cast<int64_t>(it.z) * a.b_sample_stride;
        return true;
    };
    PsgLaneOffsets off;   // of the item whose DMA is being issued (the current one, from its last k-step on the next one)
    auto lane_offsets = [&](const PsgItem& it) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            off.a[t] = min(it.m0 + row[t], a.M - 1) * a.K + oct8[t];
            const int i = min(it.n0 + row[t], a.N - 1);
            off.b[t] = !a.conv ? i * a.K + oct8[t] : row_table ? rowtab[i] : conv_row_offset(a, i);
        }
    };
    auto stage = [&](const PsgItem& it, int buf, int k0) {
        unsigned char* base = sm + buf * PV2_STAGE_BYTES + wave * 4096;
#pragma unroll
        for (int t = 0; t < 4; ++t) glds16(it.pa + (off.a[t] + k0), base + t * 1024);
        if (a.conv) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                int o;
                if (k_table) o = ktab[(k0 >> 3) + (oct8[t] >> 3)];
                else { const int p0 = k0 + oct8[t], oy = p0 / a.O2, ox = p0 - oy * a.O2; o = oy * a.s1 * a.Wq + ox; }
                glds16(it.pb + (off.b[t] + o), base + PV2_OPERAND_BYTES + t * 1024);
            }
        } else {
#pragma unroll
            for (int t = 0; t < 4; ++t) glds16(it.pb + (off.b[t] + k0), base + PV2_OPERAND_BYTES + t * 1024);
        }
    };

    PsgItem cur, nxt;
    bool have = decode(group, cur);
    if (!have) return;
    const int lr = lane & 31, sw = (lr >> 1) & 7, hi = lane >> 5;
    // epilogue constants: the lane's LDS positions (transposed accumulators) and its 16-byte piece of a result row
    const int er = tid >> 4, ech = tid & 15;   // row (+ 16 per pass) and 16-byte chunk this thread copies out
    const int64_t out_step = ROWS ? static_cast<int64_t>(16) * a.batch * a.N : PLAIN ? static_cast<int64_t>(16) * a.N
                                                                                      : static_cast<int64_t>(a.N >> 2) * a.out_tile_stride;
    lane_offsets(cur);
    stage(cur, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    int buf = 0;
    while (have) {
        group += gstride;
        const bool have_next = decode(group, nxt);
        f32x16 acc[2][2];
        for (int k0 = 0; k0 < a.K; k0 += 64) {
            if (k0 + 64 < a.K) stage(cur, buf ^ 1, k0 + 64);
            else if (have_next) { lane_offsets(nxt); stage(nxt, buf ^ 1, 0); }   // the next item's first k-step rides behind this one
            const unsigned char* sa = sm + buf * PV2_STAGE_BYTES + (wm * 64 + lr) * 128;
            const unsigned char* sb = sm + buf * PV2_STAGE_BYTES + PV2_OPERAND_BYTES + (wn * 64 + lr) * 128;
            if (k0 == 0) wave_kstep_transposed<true>(acc, sa, sb, hi, sw);
            else wave_kstep_transposed<false>(acc, sa, sb, hi, sw);
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
            buf ^= 1;
        }
        // `buf` now holds the next item's first k-step (landed); `buf ^ 1` was consumed last and stages the bf16 tile:
        // row m (256 B), 16-byte chunk c of it at chunk position c ^ (m & 15)
        unsigned char* ep = sm + (buf ^ 1) * PV2_STAGE_BYTES;
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj) {
                const int ml = wm * 64 + tj * 32 + lr;
                unsigned char* dst = ep + ml * 256 + hi * 8;
#pragma unroll
                for (int q = 0; q < 4; ++q) {   // accumulator registers 4 q .. 4 q + 3: columns i = 8 q + 4 hi + (0 .. 3) of the block
                    const int c = wn * 8 + ti * 4 + q;
                    const uint32_t w0 = pack_bf16x2(acc[ti][tj][4 * q], acc[ti][tj][4 * q + 1]);
                    const uint32_t w1 = pack_bf16x2(acc[ti][tj][4 * q + 2], acc[ti][tj][4 * q + 3]);
                    *reinterpret_cast<uint2*>(dst + ((c ^ (ml & 15)) << 4)) = uint2{w0, w1};
                }
            }
        __syncthreads();
        {
            const int m = cur.m0 + er, n = cur.n0 + ech * 8;
            const int64_t d = static_cast<int64_t>(m) * a.N + n;
            // 16 rows further: d grows by 16 N, a multiple of 64 -> the same place in a k-tile, N / 4 k-tiles on
            uint16_t* dst = a.out + (ROWS ? (static_cast<int64_t>(m) * a.batch + cur.z) * a.N + n
                                    : PLAIN ? (static_cast<int64_t>(cur.z) * a.M + m) * a.N + n
                                          : (d >> 6) * a.out_tile_stride + static_cast<int64_t>(cur.z) * 64 + (d & 63));
            const unsigned char* src = ep + er * 256 + ((ech ^ (er & 15)) << 4);
#pragma unroll
            for (int it = 0; it < 8; ++it)
                if (m + 16 * it < a.M && n < a.N) {  // N % 8 == 0: a chunk is entirely in or out
                    u32x4 w = *reinterpret_cast<const u32x4*>(src + it * 4096);
                    if constexpr (PLAIN) {   // eight products in fp32, rounded to bf16 once more
                        const float* mrow = a.mul + static_cast<int64_t>(m + 16 * it) * a.ld_mul + n;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float f0 = (n + 2 * e < a.mul_n) ? mrow[2 * e] : 0.0f, f1 = (n + 2 * e + 1 < a.mul_n) ? mrow[2 * e + 1] : 0.0f;
                            w[e] = pack_bf16x2(__uint_as_float(w[e] << 16) * f0, __uint_as_float(w[e] & 0xffff0000u) * f1);
                        }
                    }
                    *reinterpret_cast<u32x4*>(dst + it * out_step) = w;
                }
        }
        __syncthreads();   // the staging buffer is free again before the next k-step's DMA is issued into it
        cur = nxt;
        have = have_next;
    }
}

// ------------------------------------------------------------------------------------------------
// Per-sample gradients with a LONG contraction (sequences: K = T >= 256) on the wave-role-split 256 x 256 loop: one workgroup
// per (sample, 256 x 256 tile), 4 .. 16 k-tiles each -- the 128 x 128 kernel above spends a DMA round trip per k-step on these
// (0.3 PFLOP/s at T = 512).  The operands are swapped (tile rows = i, the contiguous direction of the result; tile columns = m),
// so a lane owns four consecutive bf16 of a result row; the whole 256 x 256 bf16 tile is transposed through the 128 KB of LDS
// (row m: 512 B, 16-byte chunk c at position c ^ (m & 31)) and leaves in 16-byte pieces of the k-tile-major layout.
// Covers columns [0, n_end) (n_end % 256 == 0 or n_end == N); M % 256 == 0.  A sample's tiles are consecutive items of one
// XCD (its two operands, < 2 MB, stay in that L2).
// ------------------------------------------------------------------------------------------------
struct PsgPpArgs {
    uint16_t* out; int64_t out_tile_stride;
    const uint16_t* A; int64_t a_sample_stride;   // [batch][M][K]
    const uint16_t* B; int64_t b_sample_stride;   // [batch][N][K]
    int M, N, KT, batch, tiles_m, tiles_n, n_end;
};

template <int ISSUE>
__global__ __launch_bounds__(pp::THREADS) void psg_gemm_pp_kernel(PsgPpArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 2, wn = wave & 3;
    const int tiles = a.tiles_m * a.tiles_n;
    const int64_t items = static_cast<int64_t>(a.batch) * tiles, per_xcd = (items + 7) / 8;
    const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
    const int64_t item = static_cast<int64_t>(xcd) * per_xcd + j;
    if (j >= per_xcd || item >= items) return;
    const int z = static_cast<int>(item / tiles), tile = static_cast<int>(item - static_cast<int64_t>(z) * tiles);
    const int tn = tile / a.tiles_m, tm = tile - tn * a.tiles_m;
    const int i0 = tn * 256, m0 = tm * 256;   // tile rows: i0 .. (B operand of the gradient), tile columns: m0 .. (A operand)
    const int K = a.KT * 64;
    const uint16_t* rows_i = a.B + static_cast<int64_t>(z) * a.b_sample_stride;
    const uint16_t* rows_m = a.A + static_cast<int64_t>(z) * a.a_sample_stride;

    pp::Sources src;
    pp::make_sources(src, wave, lane,
                     [&](int row) { return rows_i + static_cast<int64_t>(min(i0 + row, a.N - 1)) * K; },
                     [&](int row) { return rows_m + static_cast<int64_t>(m0 + row) * K; });
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jn][r] = 0.0f;
    pp::mainloop<ISSUE>(acc, sm, src, a.KT, wave, lane, [](int t) { return t * 64; }, [](int t) { return t * 64; });
    __syncthreads();   // every wave is done with the stage buffers: the epilogue reuses them
    const int hi = lane >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) {
            const int ml = wn * 64 + jn * 32 + (lane & 31);
            unsigned char* dst = sm + ml * 512 + hi * 8;
#pragma unroll
            for (int q = 0; q < 4; ++q) {   // registers 4 q .. 4 q + 3: tile rows i = wm * 128 + i * 32 + 8 q + 4 hi + (0 .. 3)
                const int c = wm * 16 + i * 4 + q;
                uint2 w;
                w.x = pack_bf16x2(acc[i][jn][4 * q], acc[i][jn][4 * q + 1]);
                w.y = pack_bf16x2(acc[i][jn][4 * q + 2], acc[i][jn][4 * q + 3]);
                *reinterpret_cast<uint2*>(dst + ((c ^ (ml & 31)) << 4)) = w;
            }
        }
    __syncthreads();
    // 512 threads = 16 result rows x 32 chunks per pass; 16 rows further d = m N + n grows by 16 N, a multiple of 64
    const int er = tid >> 5, ech = tid & 31, n = i0 + ech * 8;
    if (n < a.n_end) {  // N % 8 == 0: a chunk is entirely in or out
        const int64_t d = static_cast<int64_t>(m0 + er) * a.N + n;
        uint16_t* dst = a.out + (d >> 6) * a.out_tile_stride + static_cast<int64_t>(z) * 64 + (d & 63);
        const int64_t step = static_cast<int64_t>(a.N >> 2) * a.out_tile_stride;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int ml = er + 16 * it;
            *reinterpret_cast<u32x4*>(dst + it * step) = *reinterpret_cast<const u32x4*>(sm + ml * 512 + ((ech ^ (ml & 31)) << 4));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Round 5: the same per-sample gradients straight from the HOOKED tensors.  G[n] is [T][O] and A[n] is [T][I] in memory -- the
// contraction index t is the slow axis of both -- so the K-contiguous kernels above needed two transposed copies per call
// (transpose_rows_kernel: 22 % of a GPT-2 score call).  Here the 256 x 256 loop runs on K-major operands (kf_pingpong_tn.h:
// [t][feature] LDS images, ds_read_b64_tr_b16 fragments) and the two train micro-batches of a pair are two base pointers, not a
// copy.  Covers the real input columns [0, I) (I, O % 256 == 0).  The bias column (column I of the augmented axis: the ones column
// of A', i.e. sum_t G[n][t][o]) is summed from the G fragments the waves of the tiles tn == 0 hold anyway (v_dot2c_f32_bf16 beside
// the MFMAs) -- no second pass over G -- and written with the 7 zeros that pad the axis to Ip = I + 8.  Work items, operand roles
// (tile rows = i, tile columns = o) and the epilogue are those of psg_gemm_pp_kernel.  (A persistent, item-pipelined form was
// measured and removed: profiles/r05_psg_persistent_negative.log; so was, in round 6, a stream of items per persistent workgroup
// with the tile stored straight from the accumulators -- no LDS pass, no barrier, correct, slower: the stores share the in-order
// vmcnt queue with the LDS-DMA requests, profiles/r06_psg_register_stores_negative.log.)
// ------------------------------------------------------------------------------------------------
struct PsgTnArgs {
    uint16_t* out; int64_t out_tile_stride;
    const uint16_t* G[2]; const uint16_t* A[2];   // segment s: samples [s ? b0 : 0, ...), G[s]: [.][T][O], A[s]: [.][T][I]
    int b0;
    int O, I, Ip, KT, batch, tiles_m, tiles_n;    // tiles_m = O / 256, tiles_n = I / 256
    int ones;                                     // the bias column is wanted: Ip == I + 8 (else Ip == I)
};

template <int IMG>
__global__ __launch_bounds__(pptn::THREADS) void psg_gemm_tn_kernel(PsgTnArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 2, wn = wave & 3;
    const int tiles = a.tiles_m * a.tiles_n;
    const int64_t items = static_cast<int64_t>(a.batch) * tiles, per_xcd = (items + 7) / 8;
    const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
    const int64_t item = static_cast<int64_t>(xcd) * per_xcd + j;
    if (j >= per_xcd || item >= items) return;
    const int z = static_cast<int>(item / tiles), tile = static_cast<int>(item - static_cast<int64_t>(z) * tiles);
    const int tn = tile / a.tiles_m, tm = tile - tn * a.tiles_m;
    const int i0 = tn * 256, m0 = tm * 256;   // tile rows: i0 .. (activation columns), tile columns: m0 .. (output-gradient columns)
    const int seg = z >= a.b0, zs = z - (seg ? a.b0 : 0);
    const int64_t T = static_cast<int64_t>(a.KT) * 64;
    const uint16_t* rows_i = a.A[seg] + zs * T * a.I + i0;
    const uint16_t* rows_m = a.G[seg] + zs * T * a.O + m0;

    pptn::Sources src;
    pptn::make_sources<IMG>(src, wave, lane, [&](int f) { return rows_i + f; }, static_cast<int64_t>(a.I),
                            [&](int f) { return rows_m + f; }, static_cast<int64_t>(a.O));
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jn][r] = 0.0f;
    const bool bias = a.ones && tn == 0;   // (wave-uniform)
    const float cs = pptn::mainloop<IMG>(acc, sm, src, a.KT, wave, lane, static_cast<int64_t>(a.I) * 64, static_cast<int64_t>(a.O) * 64, bias);
    __syncthreads();   // every wave is done with the stage buffers: the epilogue reuses them
    const int hi = lane >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn) {
            const int ml = wn * 64 + jn * 32 + (lane & 31);
            unsigned char* dst = sm + ml * 512 + hi * 8;
#pragma unroll
            for (int q = 0; q < 4; ++q) {   // registers 4 q .. 4 q + 3: tile rows i = wm * 128 + i * 32 + 8 q + 4 hi + (0 .. 3)
                const int c = wm * 16 + i * 4 + q;
                uint2 w;
                w.x = pack_bf16x2(acc[i][jn][4 * q], acc[i][jn][4 * q + 1]);
                w.y = pack_bf16x2(acc[i][jn][4 * q + 2], acc[i][jn][4 * q + 3]);
                *reinterpret_cast<uint2*>(dst + ((c ^ (ml & 31)) << 4)) = w;
            }
        }
    __syncthreads();
    // 512 threads = 16 result rows x 32 chunks per pass; 16 rows further d = o Ip + i grows by 16 Ip, a multiple of 64
    const int er = tid >> 5, ech = tid & 31, n = i0 + ech * 8;
    const int64_t d = static_cast<int64_t>(m0 + er) * a.Ip + n;
    uint16_t* dst = a.out + (d >> 6) * a.out_tile_stride + static_cast<int64_t>(z) * 64 + (d & 63);
    const int64_t step = static_cast<int64_t>(a.Ip >> 2) * a.out_tile_stride;
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int ml = er + 16 * it;
        *reinterpret_cast<u32x4*>(dst + it * step) = *reinterpret_cast<const u32x4*>(sm + ml * 512 + ((ech ^ (ml & 31)) << 4));
    }
    if (bias) {   // column I: wave (wm, wn) summed G over t for output columns m0 + wn * 64 + wm * 32 + (lane & 31); k-octets in lanes l, l ^ 32
        const float total = cs + __shfl_xor(cs, 32);
        if (hi == 0) {
            const int64_t db = static_cast<int64_t>(m0 + wn * 64 + wm * 32 + (lane & 31)) * a.Ip + a.I;
            *reinterpret_cast<u32x4*>(a.out + (db >> 6) * a.out_tile_stride + static_cast<int64_t>(z) * 64 + (db & 63)) =
                u32x4{pack_bf16x2(total, 0.0f), 0u, 0u, 0u};
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Zero-padded, column-phase-split copy of a convolution input for the implicit im2col above:
//   Xs[phase][n][c][y][j] = x_pad[n][c][y][s2 * j + phase],  x_pad = x with p1 / p2 zero borders,
//   y < Hp, j < Wq.  One 16-byte store (8 columns) per thread.
// ------------------------------------------------------------------------------------------------
struct PadArgs {
    uint16_t* out; const uint16_t* x;
    int64_t planes;  // batch * Cp
    int C, Cp;       // real / padded channels (planes c >= C are zero)
    int H, W, Hp, Wq, p1, p2, s2;
};

__global__ __launch_bounds__(256) void conv_pad_phases_kernel(PadArgs a) {
    const int wq8 = a.Wq >> 3;
    const int64_t per_phase = a.planes * a.Hp * wq8, total = per_phase * a.s2;
    for (int64_t e = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; e < total;
         e += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const int phase = static_cast<int>(e / per_phase);
        int64_t rest = e - phase * per_phase;
        const int jv = static_cast<int>(rest % wq8);
        rest /= wq8;
        const int y = static_cast<int>(rest % a.Hp);
        const int64_t plane = rest / a.Hp;
        const int iy = y - a.p1;
        const int64_t n = plane / a.Cp;
        const int c = static_cast<int>(plane - n * a.Cp);
        const bool row_ok = iy >= 0 && iy < a.H && c < a.C;
        const uint16_t* src = a.x + ((n * a.C + (c < a.C ? c : 0)) * a.H + (row_ok ? iy : 0)) * a.W;
        uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int ix = a.s2 * (jv * 8 + t) + phase - a.p2;
            const bool ok = row_ok && ix >= 0 && ix < a.W;
            const uint32_t v = src[min(max(ix, 0), a.W - 1)];
            w[t >> 1] |= (ok ? v : 0u) << ((t & 1) * 16);
        }
        reinterpret_cast<u32x4*>(a.out)[e] = u32x4{w[0], w[1], w[2], w[3]};
    }
}

// ------------------------------------------------------------------------------------------------
// [batch][T][C] -> [batch][Cp][T] (bf16), rows C .. Cp-1: the ones row of a bias (ones != 0) then zeros.
// 64 x 64 tiles through LDS; T % 64 == 0, C % 8 == 0.
// ------------------------------------------------------------------------------------------------
struct TransposeArgs {
    uint16_t* out; const uint16_t* x;
    int T, C, Cp, ones;
    const void* mask; int mask_dtype;  // nullable [batch * T] integer mask (KF_I64 / KF_I32 / KF_U8): every row (incl. its one) times its mask value
};

__device__ __forceinline__ float mask_value(const void* mask, int dtype, int64_t idx) {
    if (dtype == I64) return static_cast<float>(reinterpret_cast<const int64_t*>(mask)[idx]);
    if (dtype == I32) return static_cast<float>(reinterpret_cast<const int32_t*>(mask)[idx]);
    return static_cast<float>(reinterpret_cast<const uint8_t*>(mask)[idx]);
}

__device__ __forceinline__ uint32_t scale_bf16_pair(uint32_t w, float m) {  // both bf16 halves of w times m, rounded to bf16
    const __bf16 lo = static_cast<__bf16>(__uint_as_float(w << 16) * m), hi = static_cast<__bf16>(__uint_as_float(w & 0xffff0000u) * m);
    return static_cast<uint32_t>(__builtin_bit_cast(uint16_t, lo)) | (static_cast<uint32_t>(__builtin_bit_cast(uint16_t, hi)) << 16);
}

__global__ __launch_bounds__(256) void transpose_rows_kernel(TransposeArgs a) {
    __shared__ uint16_t tile[64][72];
    const int t0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
    const int64_t z = blockIdx.z;
    const uint16_t* x = a.x + z * static_cast<int64_t>(a.T) * a.C;
    uint16_t* out = a.out + z * static_cast<int64_t>(a.Cp) * a.T;
    // load: 64 rows (t) x 8 chunks of 8 columns
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int id = threadIdx.x + 256 * it, r = id >> 3, ch = id & 7;
        const int c = c0 + ch * 8;
        u32x4 v = {0u, 0u, 0u, 0u};
        // the row times its mask value, bias one included (module/linear.py:39-43 multiplies both): 0 -> zeros, 1 -> a copy,
        // anything else (a weighted integer mask) -> the product rounded to bf16, as the reference's in-place ``mul_`` does
        const float mv = a.mask ? mask_value(a.mask, a.mask_dtype, z * a.T + t0 + r) : 1.0f;
        if (mv == 0.0f) { /* masked token: whole row zero */ }
        else if (c < a.C) {
            v = *reinterpret_cast<const u32x4*>(x + static_cast<int64_t>(t0 + r) * a.C + c);
            if (mv != 1.0f) v = u32x4{scale_bf16_pair(v[0], mv), scale_bf16_pair(v[1], mv), scale_bf16_pair(v[2], mv), scale_bf16_pair(v[3], mv)};
        } else if (a.ones && c == a.C) {
            v[0] = static_cast<uint32_t>(__builtin_bit_cast(uint16_t, static_cast<__bf16>(mv)));  // column C: the (masked) one, zeros after it
        }
        uint16_t* d = &tile[r][ch * 8];
        d[0] = static_cast<uint16_t>(v[0]); d[1] = static_cast<uint16_t>(v[0] >> 16);
        d[2] = static_cast<uint16_t>(v[1]); d[3] = static_cast<uint16_t>(v[1] >> 16);
        d[4] = static_cast<uint16_t>(v[2]); d[5] = static_cast<uint16_t>(v[2] >> 16);
        d[6] = static_cast<uint16_t>(v[3]); d[7] = static_cast<uint16_t>(v[3] >> 16);
    }
    __syncthreads();
    // store: 64 rows (c) x 8 chunks of 8 t's
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int id = threadIdx.x + 256 * it, r = id >> 3, ch = id & 7;
        const int c = c0 + r;
        if (c < a.Cp) {
            uint32_t w[4];
#pragma unroll
            for (int t = 0; t < 4; ++t)
                w[t] = static_cast<uint32_t>(tile[ch * 8 + 2 * t][r]) | (static_cast<uint32_t>(tile[ch * 8 + 2 * t + 1][r]) << 16);
            *reinterpret_cast<u32x4*>(out + static_cast<int64_t>(c) * a.T + t0 + ch * 8) = u32x4{w[0], w[1], w[2], w[3]};
        }
    }
}

// [planes][O1][O2] -> [planes][O1p][O2p], zero filled: the output-gradient grid of a convolution rounded up so that a row is
// a whole number of 16-byte chunks and the positions a whole number of 64-wide k-steps (padded positions carry zero
// gradient, so they contribute nothing to the per-sample gradient whatever the input holds there).
__global__ __launch_bounds__(256) void pad_grid_kernel(uint16_t* out, const uint16_t* g, int64_t planes, int O1, int O2, int O1p, int O2p) {
    const int64_t total = planes * O1p * O2p;
    for (int64_t e = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; e < total;
         e += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const int ox = static_cast<int>(e % O2p);
        const int64_t rest = e / O2p;
        const int oy = static_cast<int>(rest % O1p);
        const int64_t plane = rest / O1p;
        out[e] = (oy < O1 && ox < O2) ? g[(plane * O1 + oy) * O2 + ox] : static_cast<uint16_t>(0);
    }
}

// ------------------------------------------------------------------------------------------------
// Activation covariance on the LDS-DMA engine: C[i, j] += alpha * sum_{n, k} X[n][i, k] X[n][j, k] over K-contiguous rows
// X[n] = A'[n]^T -- the transposed (masked, bias-augmented) activations of a Linear layer on sequences, or the IMPLICIT
// im2col rows of a convolution (same addressing as the gradient kernel above: no patch tensor; kf_conv2d_cov_accum of
// SURVEY.md section 8b).  128 x 128 upper-triangular tile pairs, the contraction runs over (sample, k-step) without a
// break in the DMA double buffering; split over sample ranges.  The tiles land (coalesced fp32 atomics, or plain stores when
// there is one sample range) in a staging matrix held in the kernel's own row order; cov_finalize_kernel adds it to C.
// ------------------------------------------------------------------------------------------------
struct CovV2Args {
    float* stage; int np;                  // [np x np] fp32 staging matrix in OPERAND row order, np = tiles * 128, upper tile pairs
    const uint16_t* X; int64_t sample_stride;
    int N, K, batch, tiles, zchunk, zblocks, plain_store;
    // conv addressing as PsgV2Args; operand rows are (shift, c) with c < Cp
    int conv, Cp, k2, O2, s1, d1, s2, d2, Wq, plane;
    int64_t phase_stride;
};

__global__ __launch_bounds__(NTHREADS) void cov_gemm_v2_kernel(CovV2Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    // XCD-aware work order (workgroup L runs on XCD L % 8): items are (sample range, tile pair), sample range major, cut into 8
    // contiguous runs -- the tile pairs of a sample range share its rows, so every XCD streams its own samples through its own
    // L2 (with pair-major blocks every XCD read every sample: 1.65 GB of L2 misses per launch against 0.1 GB of operands)
    const int pairs = a.tiles * (a.tiles + 1) / 2;
    const int64_t items = static_cast<int64_t>(a.zblocks) * pairs, per_xcd = (items + 7) / 8;
    const int L = blockIdx.x, xcd = L & 7, jx = L >> 3;
    const int64_t item = static_cast<int64_t>(xcd) * per_xcd + jx;
    if (jx >= per_xcd || item >= items) return;
    const int zb = static_cast<int>(item / pairs);
    int t = static_cast<int>(item % pairs), ti = 0;
    while (t >= a.tiles - ti) { t -= a.tiles - ti; ++ti; }
    const int tj = ti + t;
    const int m0 = ti * 128, n0 = tj * 128;
    const int z_begin = zb * a.zchunk, z_end = min(a.batch, z_begin + a.zchunk);
    if (z_begin >= z_end) return;

    const uint16_t* src_a[4];
    const uint16_t* src_b[4];
    int oct[4];
    auto row_source = [&](int i) -> const uint16_t* {
        if (a.conv) {
            const int shift = i / a.Cp, c = i - shift * a.Cp;
            const int ky = shift / a.k2, kx = shift - ky * a.k2;
            const int col = kx * a.d2, phase = col % a.s2, coff = col / a.s2;
            return a.X + phase * a.phase_stride + static_cast<int64_t>(c) * a.plane + ky * a.d1 * a.Wq + coff;
        }
        return a.X + static_cast<int64_t>(i) * a.K;
    };
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int row = (wave * 4 + g) * 8 + (lane >> 3);
        oct[g] = (lane & 7) ^ lds_swz(row);
        src_a[g] = row_source(min(m0 + row, a.N - 1));
        src_b[g] = row_source(min(n0 + row, a.N - 1));
    }
    const int ksteps = a.K >> 6;
    const int* ktab = nullptr;   // implicit-im2col offsets per (k-step, octet): one division per table entry, not per request
    if (a.conv && ksteps <= PV2_KTAB_STEPS) {
        ktab = conv_offset_table(a, sm, a.K, NTHREADS);
        __syncthreads();
    }
    auto stage = [&](int buf, int step) {  // step = (z - z_begin) * ksteps + kstep
        const int zq = step / ksteps, ks = step - zq * ksteps, k0 = ks * 64;
        const int64_t zoff = static_cast<int64_t>(z_begin + zq) * a.sample_stride;
        unsigned char* base = sm + buf * PV2_STAGE_BYTES + wave * 4096;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            int koff = k0 + oct[g] * 8;
            if (ktab) koff = ktab[ks * 8 + oct[g]];
            else if (a.conv) { const int oy = koff / a.O2, ox = koff - oy * a.O2; koff = oy * a.s1 * a.Wq + ox; }
            glds16(src_a[g] + zoff + koff, base + g * 1024);
            glds16(src_b[g] + zoff + koff, base + PV2_OPERAND_BYTES + g * 1024);
        }
    };
    f32x16 acc[2][2];
    zero_acc(acc);
    {
        const int lr = lane & 31, sw = (lr >> 1) & 7, hi = lane >> 5;
        const int steps = (z_end - z_begin) * ksteps;
        stage(0, 0);
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        int buf = 0;
        for (int step = 0; step < steps; ++step) {
            if (step + 1 < steps) stage(buf ^ 1, step + 1);
            const unsigned char* sa = sm + buf * PV2_STAGE_BYTES + (wm * 64 + lr) * 128;
            const unsigned char* sb = sm + buf * PV2_STAGE_BYTES + PV2_OPERAND_BYTES + (wn * 64 + lr) * 128;
            wave_kstep_64x64(acc, sa, sb, hi, sw);
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
            buf ^= 1;
        }
    }
    // epilogue into the staging matrix (operand row order, no bounds: it is padded to whole tiles): lanes run along a row, so
    // the atomics of a wave are two 128-byte segments.  Writing the covariance itself from here -- index map c * taps + shift,
    // mirrored element -- was measured at MORE than the whole k-loop (scattered fp32 atomics: 1.54 ms vs 0.70 ms without them
    // on the 128 -> 128 3 x 3 layer); cov_finalize_kernel does the permutation and the mirror once per call instead.
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float* dst = a.stage + static_cast<int64_t>(m0 + acc_row(wm, i, r, lane)) * a.np + n0 + acc_col(wn, jn, lane);
                if (a.plain_store) *dst = acc[i][jn][r];
                else atomicAdd(dst, acc[i][jn][r]);
            }
}

// Round 3: the covariance contraction on the wave-role-split 256 x 256 main loop (kf_pingpong.h).  Same work items as the
// v2 kernel -- (sample range, upper-triangular tile pair), sample range major per XCD -- with 256-row tiles: one k-tile of the
// loop is one (sample, k-step), walked without a break; the 8 DMA requests of a lane share one k-octet, so the implicit-im2col
// offset (one division by the output width) is computed once per k-tile and lane.  A 128 x 128 / 4-wave tile issues twice
// the LDS-DMA requests per MFMA of this one, which is what bound the v2 kernel (485 TFLOP/s with its epilogue switched off).
constexpr int COV_KTAB_STEPS = 128;                       // k-steps per sample the offset table holds (4 KB)
constexpr int COV_V3_SMEM = pp::SMEM_BYTES + COV_KTAB_STEPS * 8 * 4;

template <int ISSUE>
__global__ __launch_bounds__(pp::THREADS) void cov_gemm_v3_kernel(CovV2Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 2, wn = wave & 3;
    const int pairs = a.tiles * (a.tiles + 1) / 2;
    const int64_t items = static_cast<int64_t>(a.zblocks) * pairs, per_xcd = (items + 7) / 8;
    const int L = blockIdx.x, xcd = L & 7, jx = L >> 3;
    const int64_t item = static_cast<int64_t>(xcd) * per_xcd + jx;
    if (jx >= per_xcd || item >= items) return;
    const int zb = static_cast<int>(item / pairs);
    int t = static_cast<int>(item % pairs), ti = 0;
    while (t >= a.tiles - ti) { t -= a.tiles - ti; ++ti; }
    const int tj = ti + t;
    const int m0 = ti * 256, n0 = tj * 256;
    const int z_begin = zb * a.zchunk, z_end = min(a.batch, z_begin + a.zchunk);
    if (z_begin >= z_end) return;

    const int oct = pp::lane_octet(wave, lane);
    auto row_source = [&](int i) -> const uint16_t* {
        if (a.conv) {
            const int shift = i / a.Cp, c = i - shift * a.Cp;
            const int ky = shift / a.k2, kx = shift - ky * a.k2;
            const int col = kx * a.d2, phase = col % a.s2, coff = col / a.s2;
            return a.X + phase * a.phase_stride + static_cast<int64_t>(c) * a.plane + ky * a.d1 * a.Wq + coff;
        }
        return a.X + static_cast<int64_t>(i) * a.K + oct * 8;
    };
    pp::Sources src;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int row = pp::request_row0(r, wave) + (lane >> 3);
        src.p[r] = row_source(min((r < 4 ? m0 : n0) + row, a.N - 1));
    }
    const int ksteps = a.K >> 6;
    const int64_t base = static_cast<int64_t>(z_begin) * a.sample_stride;
    // per-(k-step, octet) offsets inside a sample, computed ONCE per workgroup into the LDS tail (the implicit-im2col offset
    // needs a division by the output width; done per request it made the L segments longer than the M segments)
    int* ktab = reinterpret_cast<int*>(sm + pp::SMEM_BYTES);   // ksteps <= COV_KTAB_STEPS (the launcher's condition for this kernel)
    for (int e = tid; e < ksteps * 8; e += pp::THREADS) {
        const int p0 = (e >> 3) * 64 + (e & 7) * 8;
        int koff = (e >> 3) * 64;   // plain rows: the octet is already in the source pointer
        if (a.conv) { const int oy = p0 / a.O2, ox = p0 - oy * a.O2; koff = oy * a.s1 * a.Wq + ox; }
        ktab[e] = koff;
    }
    __syncthreads();
    const int kshift = (ksteps & (ksteps - 1)) == 0 ? __builtin_ctz(ksteps) : -1;
    auto walk = [&](int kt) -> int64_t {   // k-tile kt of this item = sample z_begin + kt / ksteps, k-step kt % ksteps
        const int zq = kshift >= 0 ? kt >> kshift : kt / ksteps, ks = kt - zq * ksteps;
        return base + static_cast<int64_t>(zq) * a.sample_stride + ktab[ks * 8 + oct];
    };
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jn][r] = 0.0f;
    pp::mainloop<ISSUE>(acc, sm, src, (z_end - z_begin) * ksteps, wave, lane, walk, walk);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ml = wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), nl = wn * 64 + jn * 32 + (lane & 31);
                float* dst = a.stage + static_cast<int64_t>(m0 + ml) * a.np + n0 + nl;
                if (a.plain_store) *dst = acc[i][jn][r];
                else atomicAdd(dst, acc[i][jn][r]);
            }
}

// Round 5: the covariance of UNMASKED sequence rows straight from the hooked tensor.  X is [rows][d] in memory (rows = all tokens
// of the batch, the contraction index), i.e. K-major: C += X^T X on the K-major loop of kf_pingpong_tn.h -- no transposed copy
// (kf_syrk_rows_bf16 used to write and re-read one per call).  Work items = (k-tile range, upper-triangular 256-row tile pair),
// range major per XCD as above; same staging matrix, same finalize pass.  The bias row / column of an activation covariance
// (the ones column of A') is a column sum: the tile pairs (0, tj) sum their B operand over k beside the MFMAs.
struct CovTnArgs {
    float* stage; int np;
    const uint16_t* X; int64_t ld;   // [KT * 64][ld], columns [0, N) are used
    int N, KT, tiles, kchunk, kblocks, plain_store;
    float* ones_out; int64_t ldc; float alpha;   // non-null: the covariance itself, whose row / column N (the bias one) gets alpha * column sums
    // split == 1 (fp32 rows as three bf16 planes, kf_syrk_rows_f32): X holds the planes H, M, L one behind the other (plane_stride
    // elements apart) and the work items run over SIX products  H^T H, H^T M, H^T L, M^T H, M^T M, L^T H  (kblocks per product)
    int split; int64_t plane_stride;
};

template <int IMG>
__global__ __launch_bounds__(pptn::THREADS) void cov_gemm_tn_kernel(CovTnArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 2, wn = wave & 3;
    const int pairs = a.tiles * (a.tiles + 1) / 2;
    const int64_t items = static_cast<int64_t>(a.kblocks) * pairs, per_xcd = (items + 7) / 8;
    const int L = blockIdx.x, xcd = L & 7, jx = L >> 3;
    const int64_t item = static_cast<int64_t>(xcd) * per_xcd + jx;
    if (jx >= per_xcd || item >= items) return;
    int kb = static_cast<int>(item / pairs);
    int t = static_cast<int>(item % pairs), ti = 0;
    while (t >= a.tiles - ti) { t -= a.tiles - ti; ++ti; }
    const int tj = ti + t;
    const int m0 = ti * 256, n0 = tj * 256;
    int plane_a = 0, plane_b = 0;
    if (a.split) {   // item = (product, k-tile range of it, tile pair); the sum of the six products is symmetric: upper pairs suffice
        const int per = a.kblocks / 6, product = kb / per;
        kb -= product * per;
        plane_a = product < 3 ? 0 : product < 5 ? 1 : 2;                      // H H H M M L
        plane_b = product < 3 ? product : product == 3 ? 0 : product == 4;     // H M L H M H
    }
    const int kt_begin = kb * a.kchunk, kt_end = min(a.KT, kt_begin + a.kchunk);
    if (kt_begin >= kt_end) return;

    const uint16_t* base_a = a.X + plane_a * a.plane_stride + static_cast<int64_t>(kt_begin) * 64 * a.ld;
    const uint16_t* base_b = a.X + plane_b * a.plane_stride + static_cast<int64_t>(kt_begin) * 64 * a.ld;
    pptn::Sources src;
    pptn::make_sources<IMG>(src, wave, lane, [&](int f) { return base_a + min(m0 + f, a.N - 8); }, a.ld,
                            [&](int f) { return base_b + min(n0 + f, a.N - 8); }, a.ld);
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jn][r] = 0.0f;
    // the bias row / column (ones_out != null): the column sums of the tile pairs (0, tj) cover every column once per k-tile range
    const bool colsum = a.ones_out != nullptr && ti == 0;   // (wave-uniform)
    const float cs = pptn::mainloop<IMG>(acc, sm, src, kt_end - kt_begin, wave, lane, a.ld * 64, a.ld * 64, colsum);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ml = wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), nl = wn * 64 + jn * 32 + (lane & 31);
                float* dst = a.stage + static_cast<int64_t>(m0 + ml) * a.np + n0 + nl;
                if (a.plain_store) *dst = acc[i][jn][r];
                else atomicAdd(dst, acc[i][jn][r]);
            }
    if (colsum) {   // C[N][j] += alpha sum_k X[k][j], C[j][N] += the same, for this item's k-tiles; the corner counts the rows
        const float total = a.alpha * (cs + __shfl_xor(cs, 32));
        const int col = n0 + wn * 64 + wm * 32 + (lane & 31);
        if (lane < 32 && col < a.N) {
            atomicAdd(a.ones_out + static_cast<int64_t>(a.N) * a.ldc + col, total);
            atomicAdd(a.ones_out + static_cast<int64_t>(col) * a.ldc + a.N, total);
        }
        if (tj == 0 && threadIdx.x == 0)
            atomicAdd(a.ones_out + static_cast<int64_t>(a.N) * a.ldc + a.N, a.alpha * 64.0f * static_cast<float>(kt_end - kt_begin));
    }
}

// fp32 rows -> three bf16 planes (round 5, kf_syrk_rows_f32): x = h + m + l EXACTLY (h = bf16(x), m = bf16(x - h), l = bf16(x - h - m):
// 8 + 8 + 8 significand bits), rows times their mask value first (module/linear.py:39-43, in fp32 as the reference's mul_ on an
// fp32 activation).  The covariance of the rows is then six bf16 products on the MFMA engine -- H^T H + H^T M + M^T H + M^T M + H^T L +
// L^T H; the three dropped ones are below 2^-23 of the result, the size of one fp32 rounding -- at ~20x the rate of the exact-fp32
// MFMA instruction (LayerNorm outputs under autocast with fp32 factors: BERT's query / key / value / intermediate inputs ran at
// 30 TFLOP/s).  The bias row / column of the covariance -- C[d][j] += alpha sum_r m_r (m_r x_rj), C[d][d] += alpha sum_r m_r^2 -- is
// summed here in fp32 from the values the kernel reads anyway.  planes: [3][rows_pad][d]; rows >= n are zero.
struct SplitArgs {
    uint16_t* planes; int64_t plane_stride;
    const float* x; int64_t n, rows_pad; int d;
    const void* mask; int mask_dtype;   // nullable [n]: KF_I64 / KF_I32 / KF_U8 / KF_F32
    float* C; int64_t ldc; int ones; float alpha;
};

__global__ __launch_bounds__(256) void split_rows_f32_kernel(SplitArgs a) {
    // a block: 64 rows x 1024 columns (4 per thread); column sums of the masked rows stay in registers
    const int c0 = blockIdx.x * 1024 + threadIdx.x * 4;
    const int64_t r0 = static_cast<int64_t>(blockIdx.y) * 64;
    if (c0 >= a.d) return;   // d % 8 == 0: four columns are entirely in or out
    float sum[4] = {0.0f, 0.0f, 0.0f, 0.0f}, count = 0.0f;
    for (int64_t r = r0; r < min(a.rows_pad, r0 + 64); ++r) {
        float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (r < a.n) {
            float mv = 1.0f;
            if (a.mask) mv = a.mask_dtype == F32 ? reinterpret_cast<const float*>(a.mask)[r] : mask_value(a.mask, a.mask_dtype, r);
            if (mv != 0.0f) {
                const float4 w = *reinterpret_cast<const float4*>(a.x + r * a.d + c0);
                v[0] = w.x * mv; v[1] = w.y * mv; v[2] = w.z * mv; v[3] = w.w * mv;
#pragma unroll
                for (int e = 0; e < 4; ++e) sum[e] = fmaf(mv, v[e], sum[e]);
                count = fmaf(mv, mv, count);
            }
        }
        uint32_t h[2], m[2], l[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const uint32_t hh = pack_bf16x2(v[2 * e], v[2 * e + 1]);
            const float r1a = v[2 * e] - __uint_as_float(hh << 16), r1b = v[2 * e + 1] - __uint_as_float(hh & 0xffff0000u);
            const uint32_t mm = pack_bf16x2(r1a, r1b);
            const float r2a = r1a - __uint_as_float(mm << 16), r2b = r1b - __uint_as_float(mm & 0xffff0000u);
            h[e] = hh; m[e] = mm; l[e] = pack_bf16x2(r2a, r2b);
        }
        uint16_t* dst = a.planes + r * a.d + c0;
        *reinterpret_cast<uint2*>(dst) = uint2{h[0], h[1]};
        *reinterpret_cast<uint2*>(dst + a.plane_stride) = uint2{m[0], m[1]};
        *reinterpret_cast<uint2*>(dst + 2 * a.plane_stride) = uint2{l[0], l[1]};
    }
    if (a.ones) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            atomicAdd(a.C + static_cast<int64_t>(a.d) * a.ldc + c0 + e, a.alpha * sum[e]);
            atomicAdd(a.C + static_cast<int64_t>(c0 + e) * a.ldc + a.d, a.alpha * sum[e]);
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(a.C + static_cast<int64_t>(a.d) * a.ldc + a.d, a.alpha * count);
    }
}

// covariance[i][j] += alpha * stage[p(i)][p(j)] (or its transpose: only tile pairs ti <= tj are computed); p = operand row of
// covariance index i: identity for plain rows, (i % taps) * Cp + i / taps for the (c, ky, kx) patch order of a convolution.
struct CovFinalizeArgs {
    float* out; int64_t ldc; const float* stage; int np, d, conv, Cp, taps; float alpha;
    int tile_shift;   // log2 of the tile size the staging matrix was filled with (7 or 8)
};

__global__ __launch_bounds__(256) void cov_finalize_kernel(CovFinalizeArgs a) {
    const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (j >= a.d) return;
    int pi = i, pj = j;
    if (a.conv) { pi = (i % a.taps) * a.Cp + i / a.taps; pj = (j % a.taps) * a.Cp + j / a.taps; }
    if ((pi >> a.tile_shift) > (pj >> a.tile_shift)) { const int t = pi; pi = pj; pj = t; }
    a.out[static_cast<int64_t>(i) * a.ldc + j] += a.alpha * a.stage[static_cast<int64_t>(pi) * a.np + pj];
}

inline int64_t align256(int64_t x) { return (x + 255) & ~static_cast<int64_t>(255); }
inline int64_t cov_stage_bytes(int64_t n_operand_rows) {
    const int64_t np = cdiv(n_operand_rows, 256) * 256;   // whole 256-row tiles (the v3 kernel; a superset of the v2 need)
    return align256(np * np * 4);
}

inline int cov_engine(int64_t n_rows, int64_t steps) {  // 2 = 128-row tiles / 4 waves, 3 = 256-row tiles on the wave-role-split loop
    if (const char* e = getenv("KF_COV_ENGINE")) return atoi(e) == 2 ? 2 : 3;
    if (engine_generation() == 2) return 2;
    const int64_t t2 = cdiv(n_rows, 128), t3 = cdiv(n_rows, 256);
    // MFMA work in 128 x 128 units: the 256-row tiling pads more and computes whole diagonal tiles.  Measured
    // (profiles/r03_cov_bench.log): its loop runs 1.3-1.5x the rate of the 4-wave kernel when the contraction is long (1600
    // rows, 4000 k-steps: 1.56 -> 1.20 ms; 1152 rows: 0.70 -> 0.65 ms; 2304 rows, 1000 k-steps: 0.61 -> 0.52 ms), but loses when
    // the contraction is short (transformer batches of 128 k-steps: the 64 K staging atomics of a 256 x 256 tile weigh more
    // than its k-loop) or the extra MFMA work exceeds a third
    // (round 4: also for very wide factors whatever the contraction length -- 14 336 rows, 64 k-steps: 2.02 -> 1.79 ms)
    return ((t3 * (t3 + 1) / 2) * 4 * 100 <= (t2 * (t2 + 1) / 2) * 134 && (steps >= 512 || t3 >= 48)) ? 3 : 2;
}

int launch_cov_v3(CovV2Args& c, CovFinalizeArgs& f, hipStream_t st) {
    c.tiles = static_cast<int>(cdiv(c.N, 256));
    c.np = c.tiles * 256;
    const int64_t pairs = static_cast<int64_t>(c.tiles) * (c.tiles + 1) / 2, steps = static_cast<int64_t>(c.batch) * (c.K >> 6);
    // One workgroup per CU (128 KB of LDS), so the items run in rounds of 256: split the samples so that the LAST round is
    // (nearly) full too -- 45 tile pairs x 12 sample blocks = 540 items ran as three rounds, the third 11 % full; 45 x 11 = 495
    // are two.  Among 1-4 rounds take the cheapest by (k-tiles per item + a prologue / staging-atomics charge of ~12 k-tiles).
    int64_t zblocks = 1, best = INT64_MAX;
    for (int rounds = 1; rounds <= 4; ++rounds) {
        const int64_t want = std::max<int64_t>(1, std::min<int64_t>({static_cast<int64_t>(c.batch), rounds * 256 / pairs, steps / 16}));
        const int64_t chunk = cdiv(c.batch, want), blocks = cdiv(c.batch, chunk);
        const int64_t cost = cdiv(blocks * pairs, 256) * (chunk * (c.K >> 6) + 12);
        if (cost < best) { best = cost; zblocks = blocks; c.zchunk = static_cast<int>(chunk); }
    }
    c.zblocks = static_cast<int>(zblocks);
    c.plain_store = zblocks == 1;
    const dim3 grid(static_cast<unsigned>(8 * cdiv(zblocks * pairs, 8)));
    if (!c.plain_store && hipMemsetAsync(c.stage, 0, static_cast<size_t>(c.np) * c.np * 4, st) != hipSuccess) return KF_ERR_LAUNCH_FAILED;
    with_pp_issue([&](auto iss) { hipLaunchKernelGGL((cov_gemm_v3_kernel<decltype(iss)::value>), grid, dim3(pp::THREADS), COV_V3_SMEM, st, c); }, 2);
    f.stage = c.stage; f.np = c.np; f.tile_shift = 8;
    hipLaunchKernelGGL(cov_finalize_kernel, dim3(static_cast<unsigned>(cdiv(f.d, 256)), static_cast<unsigned>(f.d)), dim3(256), 0, st, f);
    return launch_status();
}

int launch_cov_v2(CovV2Args& c, CovFinalizeArgs& f, hipStream_t st) {
    if ((c.K >> 6) <= COV_KTAB_STEPS && cov_engine(c.N, static_cast<int64_t>(c.batch) * (c.K >> 6)) == 3) return launch_cov_v3(c, f, st);
    c.tiles = static_cast<int>(cdiv(c.N, 128));
    c.np = c.tiles * 128;
    const int64_t pairs = static_cast<int64_t>(c.tiles) * (c.tiles + 1) / 2, steps = static_cast<int64_t>(c.batch) * (c.K >> 6);
    // work items = tile pairs x sample ranges.  Measured (profiles/r02_cov_bench.log): long contractions (>= 512 k-steps per
    // tile pair: the convolutions) want ~2000 items -- 0.70 -> 0.64 ms (3x3 128 -> 128), 0.72 -> 0.60 ms (256 -> 256 on 8 x 8)
    // against ~1000; short ones want ~500.  Round 4 (profiles/r04_cov_items_ab.log): an item must also be LONG enough for its 64 KB
    // staging epilogue -- at d = 768 (28 tile pairs: every BERT / GPT-2 attention-side factor) the best split is 16 sample
    // ranges whatever the batch: 32 k-steps per item at 512 steps per pair (140 -> 85 us), 64 at 1 024 (188 -> 128 us), 16 at
    // 128 (56 -> 49 us); the old floor of 8 k-steps per item cut those batches into 1 792 items of 8 k-steps.
    int64_t target = steps >= 512 ? 2048 : 512;
    if (const char* e = getenv("KF_COV_ITEMS")) target = std::max<int64_t>(1, atoll(e));   // measurements only
    int64_t min_ksteps = std::max<int64_t>(16, std::min<int64_t>(64, steps / 16));
    if (const char* e = getenv("KF_COV_MIN_KSTEPS")) min_ksteps = std::max<int64_t>(1, atoll(e));   // measurements only
    const int64_t zsplit = std::max<int64_t>(1, std::min<int64_t>({static_cast<int64_t>(c.batch), cdiv(target, pairs), steps / min_ksteps}));
    c.zchunk = static_cast<int>(cdiv(c.batch, zsplit));
    const int64_t zblocks = cdiv(c.batch, c.zchunk);
    c.zblocks = static_cast<int>(zblocks);
    c.plain_store = zblocks == 1;
    const dim3 grid(static_cast<unsigned>(8 * cdiv(zblocks * pairs, 8)));
    if (!c.plain_store && hipMemsetAsync(c.stage, 0, static_cast<size_t>(c.np) * c.np * 4, st) != hipSuccess) return KF_ERR_LAUNCH_FAILED;
    hipLaunchKernelGGL(cov_gemm_v2_kernel, grid, dim3(NTHREADS), PV2_SMEM, st, c);
    f.stage = c.stage; f.np = c.np; f.tile_shift = 7;
    hipLaunchKernelGGL(cov_finalize_kernel, dim3(static_cast<unsigned>(cdiv(f.d, 256)), static_cast<unsigned>(f.d)), dim3(256), 0, st, f);
    return launch_status();
}

// covariance of unmasked K-major rows X[rows][ld] (columns [0, N)) on the K-major loop: C[0..N)[0..N) += alpha X^T X
int launch_cov_tn(float* stage, const uint16_t* X, int64_t ld, int64_t rows, int64_t N, CovFinalizeArgs& f, bool ones, hipStream_t st,
                  int64_t split_plane_stride = 0) {
    CovTnArgs c{};
    c.stage = stage; c.X = X; c.ld = ld; c.N = static_cast<int>(N); c.KT = static_cast<int>(rows / 64);
    c.ones_out = ones ? f.out : nullptr; c.ldc = f.ldc; c.alpha = f.alpha;
    c.split = split_plane_stride != 0; c.plane_stride = split_plane_stride;
    const int64_t products = c.split ? 6 : 1;
    c.tiles = static_cast<int>(cdiv(N, 256));
    c.np = c.tiles * 256;
    const int64_t pairs = static_cast<int64_t>(c.tiles) * (c.tiles + 1) / 2, steps = c.KT;
    // one workgroup per CU: rounds of 256 items; among 1-4 rounds the cheapest by (k-tiles per item + ~12 for the prologue and
    // the 64 K staging atomics), as launch_cov_v3 -- the split is over k-tiles here, not samples
    int64_t kblocks = 1, best = INT64_MAX;
    c.kchunk = c.KT;
    for (int rounds = 1; rounds <= 4; ++rounds) {
        const int64_t want = std::max<int64_t>(1, std::min<int64_t>({steps, rounds * 256 / (pairs * products), steps / 16}));
        const int64_t chunk = cdiv(steps, want), blocks = cdiv(steps, chunk);
        const int64_t cost = cdiv(blocks * pairs * products, 256) * (chunk + 12);
        if (cost < best) { best = cost; kblocks = blocks; c.kchunk = static_cast<int>(chunk); }
    }
    kblocks *= products;   // (split: every product has its own k-tile ranges)
    c.kblocks = static_cast<int>(kblocks);
    c.plain_store = kblocks == 1;
    const dim3 grid(static_cast<unsigned>(8 * cdiv(kblocks * pairs, 8)));
    if (!c.plain_store && hipMemsetAsync(c.stage, 0, static_cast<size_t>(c.np) * c.np * 4, st) != hipSuccess) return KF_ERR_LAUNCH_FAILED;
    with_tn_image([&](auto img) { hipLaunchKernelGGL((cov_gemm_tn_kernel<decltype(img)::value>), grid, dim3(pptn::THREADS), pptn::SMEM_BYTES, st, c); });
    f.stage = c.stage; f.np = c.np; f.tile_shift = 8;
    hipLaunchKernelGGL(cov_finalize_kernel, dim3(static_cast<unsigned>(cdiv(f.d, 256)), static_cast<unsigned>(f.d)), dim3(256), 0, st, f);
    return launch_status();
}

int configure_once() {
    static std::once_flag flag;
    static int status = KF_OK;
    std::call_once(flag, [] {
        const bool ok =
            hipFuncSetAttribute(reinterpret_cast<const void*>(score_gemm_v2_kernel<256, 256, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 512 * 128) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(score_gemm_v2_kernel<256, 128, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 384 * 128) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(score_gemm_v2_kernel<128, 256, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 384 * 128) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(rotate_gemm_v2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 512 * 128) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(score_gemm_v3_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, pp::SMEM_BYTES) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(score_gemm_v3_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, pp::SMEM_BYTES) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(score_gemm_v3_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, pp::SMEM_BYTES) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(score_gemm_v4_kernel<256, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, PP64_SMEM) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(score_gemm_v4_kernel<128, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, PP64_SMEM) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(lambda_rows_kernel<6>), hipFuncAttributeMaxDynamicSharedMemorySize, PP64_SMEM) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(lambda_rows_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, PP64_SMEM) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(lambda_rows_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, PP64_SMEM) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(lambda_rows_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, PP64_SMEM) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(lambda_rows_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, PP64_SMEM) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(score_gemm_v5_kernel<4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, PPW_SMEM) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(score_gemm_v5_kernel<1, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, PPW_SMEM) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>((rotate_gemm_v3_kernel<0, 0>)), hipFuncAttributeMaxDynamicSharedMemorySize, pp::SMEM_BYTES) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>((rotate_gemm_v3_kernel<0, 1>)), hipFuncAttributeMaxDynamicSharedMemorySize, pp::SMEM_BYTES) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>((rotate_gemm_v3_kernel<0, 2>)), hipFuncAttributeMaxDynamicSharedMemorySize, pp::SMEM_BYTES) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>((rotate_gemm_v3_kernel<1, 0>)), hipFuncAttributeMaxDynamicSharedMemorySize, pp::SMEM_BYTES) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>((rotate_gemm_v3_kernel<1, 1>)), hipFuncAttributeMaxDynamicSharedMemorySize, pp::SMEM_BYTES) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>((rotate_gemm_v3_kernel<1, 2>)), hipFuncAttributeMaxDynamicSharedMemorySize, pp::SMEM_BYTES) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(psg_gemm_v2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, PV2_SMEM) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(psg_gemm_pp_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, pp::SMEM_BYTES) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(psg_gemm_pp_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, pp::SMEM_BYTES) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(psg_gemm_pp_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, pp::SMEM_BYTES) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(psg_gemm_v3_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, PV3_SMEM_MAX) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(psg_gemm_v3_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, PV3_SMEM_MAX) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(psg_gemm_v3_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, PV3_SMEM_MAX) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(cov_gemm_v2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, PV2_SMEM) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(cov_gemm_v3_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, COV_V3_SMEM) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(cov_gemm_v3_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, COV_V3_SMEM) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(cov_gemm_v3_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, COV_V3_SMEM) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(psg_gemm_tn_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, pptn::SMEM_BYTES) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(psg_gemm_tn_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, pptn::SMEM_BYTES) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(psg_gemm_tn_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, pptn::SMEM_BYTES) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(cov_gemm_tn_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, pptn::SMEM_BYTES) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(cov_gemm_tn_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, pptn::SMEM_BYTES) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(cov_gemm_tn_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, pptn::SMEM_BYTES) == hipSuccess;
        if (!ok) status = KF_ERR_LAUNCH_FAILED;
    });
    return status;
}

// one launch over the query rows [q_begin, q_begin + q_rows) of P (k-tile-major, `Q` rows per k-tile) on tile shape `shape`
// (-1: chosen by padded area)
void launch_score_rows(float* scores, int64_t ld, const uint16_t* P, const uint16_t* psg, int64_t Q, int64_t q_begin, int64_t q_rows,
                       int64_t b, int64_t D, float scale, int shape, hipStream_t st) {
    ScoreV2Args s;
    s.C = scores + q_begin * ld; s.ldc = ld; s.A = P + q_begin * 64; s.B = psg;
    s.M = static_cast<int>(q_rows); s.a_rows = static_cast<int>(Q); s.N = static_cast<int>(b); s.KT = static_cast<int>(D / 64);
    if (shape < 0) {
        // tile shape: least padded area, the 64 x 64-per-wave shapes charged 15 % for their extra LDS traffic per flop
        const int64_t area[3] = {cdiv(q_rows, 256) * 256 * cdiv(b, 256) * 256, cdiv(q_rows, 256) * 256 * cdiv(b, 128) * 128,
                                 cdiv(q_rows, 128) * 128 * cdiv(b, 256) * 256};
        shape = 0;
        if (area[1] * 115 < area[shape] * 100) shape = 1;
        if (area[2] * 115 < (shape == 0 ? area[0] * 100 : area[1] * 115)) shape = 2;
        // round 4: when a half tile wins, the 512 x 128 / 128 x 512 shapes (two-phase loop, 128 x 64 waves) cover the same narrow
        // side with fewer DMA requests per MFMA -- taken when they pad no more than the 256-row shape does
        if (half_tile_engine() == 4 && wide_tile_enabled()) {
            if (shape == 1 && cdiv(q_rows, 512) * 512 == cdiv(q_rows, 256) * 256) shape = 3;
            else if (shape == 2 && cdiv(b, 512) * 512 == cdiv(b, 256) * 256) shape = 4;
        }
        if (const char* e = getenv("KF_SCORE_SHAPE")) shape = std::min(4, std::max(0, atoi(e)));   // measurements only
    }
    const int tm = shape == 3 ? 512 : (shape == 2 || shape == 4) ? 128 : 256, tn = shape == 4 ? 512 : (shape == 1 || shape == 3) ? 128 : 256;
    s.tiles_m = static_cast<int>(cdiv(q_rows, tm)); s.tiles_n = static_cast<int>(cdiv(b, tn));
    const int64_t tiles = static_cast<int64_t>(s.tiles_m) * s.tiles_n;
    // one workgroup per CU: ONE round of work items over the 256 CUs (measured 4-6 % faster than two rounds of half the
    // length: fewer atomic epilogues and pipeline fills), at least 16 k-steps per item
    static const int64_t target = [] { const char* e = getenv("KF_SCORE_ITEMS"); return e ? std::max<int64_t>(1, atoll(e)) : 256; }();
    // ... ONE round: never more items than the target (cdiv(256, 6 tiles) = 43 chunks made 258 items -- two stragglers in a second
    // round doubled the launch: BERT's 768-row part ran 1 232 us instead of ~740, profiles/r05_bert_base_n2048_kernel_stats.csv)
    int64_t ksplit = std::max<int64_t>(1, std::min<int64_t>(tiles <= target ? target / tiles : 1, s.KT / 16));
    const int64_t kchunk = cdiv(s.KT, ksplit);
    ksplit = cdiv(s.KT, kchunk);
    s.ksplit = static_cast<int>(ksplit); s.kchunk = static_cast<int>(kchunk); s.alpha = scale;
    const dim3 grid(static_cast<unsigned>(8 * cdiv(ksplit * tiles, 8)));
    if (shape == 0 && engine_generation() == 3)
        with_pp_issue([&](auto iss) { hipLaunchKernelGGL((score_gemm_v3_kernel<decltype(iss)::value>), grid, dim3(pp::THREADS), pp::SMEM_BYTES, st, s); });
    else if (shape == 0) hipLaunchKernelGGL((score_gemm_v2_kernel<256, 256, 2>), grid, dim3(SV2_THREADS), 2 * 512 * 128, st, s);
    else if (shape == 3) hipLaunchKernelGGL((score_gemm_v5_kernel<4, 2>), grid, dim3(pp::THREADS), PPW_SMEM, st, s);
    else if (shape == 4) hipLaunchKernelGGL((score_gemm_v5_kernel<1, 8>), grid, dim3(pp::THREADS), PPW_SMEM, st, s);
    else if (shape == 1 && half_tile_engine() == 4)
        hipLaunchKernelGGL((score_gemm_v4_kernel<256, 128>), grid, dim3(pp64::THREADS), PP64_SMEM, st, s);
    else if (half_tile_engine() == 4)
        hipLaunchKernelGGL((score_gemm_v4_kernel<128, 256>), grid, dim3(pp64::THREADS), PP64_SMEM, st, s);
    else if (shape == 1) hipLaunchKernelGGL((score_gemm_v2_kernel<256, 128, 4>), grid, dim3(SV2_THREADS), 2 * 384 * 128, st, s);
    else hipLaunchKernelGGL((score_gemm_v2_kernel<128, 256, 2>), grid, dim3(SV2_THREADS), 2 * 384 * 128, st, s);
}

// whether the mixed row tiling below is taken without being asked for (round 6: opt-in until measured)
constexpr bool MIXED_ROWS_DEFAULT = false;
inline bool mixed_rows_default() { return MIXED_ROWS_DEFAULT; }

int launch_score_v2(float* scores, int64_t ld, const uint16_t* P, const uint16_t* psg, int64_t Q, int64_t b, int64_t D,
                    float scale, hipStream_t st) {
    // Mixed row tiling: Q = 256 a + r with 0 < r <= 128 and a wide train side (BERT: 872 queries against 512 sequences) would pad
    // its last 256-row tile more than half -- `a` row tiles on the 256 x 256 loop and ONE more launch for the last r rows: 896
    // instead of 1 024 padded rows.  Round 5 ran that remainder on 128 x 256 tiles (64 x 64 waves, kf_pingpong64.h): 253 us for an
    // eighth of the work -- 740 + 253 us against 970 us for one launch over 1 024 padded rows, no gain.  Round 6: when the train
    // side is whole 512-column tiles the remainder takes the 128 x 512 wave grid (score_gemm_v5_kernel<1, 8>: waves of 128 x 64
    // like the main loop, one tile across the whole batch, split over k on all CUs); KF_SCORE_MIXED = 0 / 1 forces it off / on
    // (1 without 512-column tiles: the 128 x 256 shape).
    const int64_t rest = Q % 256;
    const char* mixed_env = getenv("KF_SCORE_MIXED");
    const bool wide_rest = cdiv(b, 512) * 512 == cdiv(b, 256) * 256;
    const bool eligible = Q > 256 && rest > 0 && rest <= 128 && b > 128 && engine_generation() == 3 && half_tile_engine() == 4 &&
                          !getenv("KF_SCORE_SHAPE");
    const bool mixed = eligible && (mixed_env ? atoi(mixed_env) == 1 : (wide_rest && mixed_rows_default()));
    if (mixed) {
        launch_score_rows(scores, ld, P, psg, Q, 0, Q - rest, b, D, scale, 0, st);
        launch_score_rows(scores, ld, P, psg, Q, Q - rest, rest, b, D, scale, wide_rest ? 4 : 2, st);
    } else {
        launch_score_rows(scores, ld, P, psg, Q, 0, Q, b, D, scale, -1, st);
    }
    return launch_status();
}

int launch_psg_v2(PsgV2Args& p, hipStream_t st) {
    p.n_begin = 0;
    if (engine_generation() == 3 && !p.conv && !p.out_rows && p.K >= 256 && p.M % 256 == 0 && p.N >= 256 && !getenv("KF_PSG_PP_OFF")) {
        // long contraction: whole 256-column tiles on the wave-role-split loop; a remainder of at most 128 columns (the bias
        // column and the padding of an odd I') stays with the 128 x 128 kernel below
        const int rest = p.N % 256, n_end = rest > 128 ? p.N : p.N - rest;
        PsgPpArgs g;
        g.out = p.out; g.out_tile_stride = p.out_tile_stride; g.A = p.A; g.a_sample_stride = p.a_sample_stride;
        g.B = p.B; g.b_sample_stride = p.b_sample_stride; g.M = p.M; g.N = p.N; g.KT = p.K >> 6; g.batch = p.batch;
        g.tiles_m = p.M / 256; g.tiles_n = static_cast<int>(cdiv(n_end, 256)); g.n_end = n_end;
        const int64_t items = static_cast<int64_t>(g.batch) * g.tiles_m * g.tiles_n;
        if (items >= (1LL << 31)) return KF_ERR_INVALID_ARGUMENT;
        with_pp_issue([&](auto iss) {
            hipLaunchKernelGGL((psg_gemm_pp_kernel<decltype(iss)::value>), dim3(static_cast<unsigned>(8 * cdiv(items, 8))), dim3(pp::THREADS), pp::SMEM_BYTES, st, g);
        });
        if (n_end == p.N) return launch_status();
        p.n_begin = n_end;
    }
    p.tiles_m = static_cast<int>(cdiv(p.M, 128)); p.tiles_n = static_cast<int>(cdiv(p.N - p.n_begin, 128));
    const int64_t blocks = 8 * cdiv(cdiv(p.batch, 8) * 8 * p.tiles_m * p.tiles_n, 8);  // PSG_ZB = 8 samples per block
    if (blocks >= (1LL << 31)) return KF_ERR_INVALID_ARGUMENT;
    // the persistent kernel keeps per-lane source offsets in 32 bits (the sample base is a 64-bit scalar)
    const int64_t span_a = static_cast<int64_t>(p.M) * p.K + 64;
    const int64_t span_b = p.conv ? static_cast<int64_t>(p.s2) * p.phase_stride + p.plane : static_cast<int64_t>(p.N) * p.K + 64;
    if (engine_generation() == 3 && span_a < (1LL << 31) && span_b < (1LL << 31)) {
        // persistent: two resident workgroups per CU (64 KB of LDS each + tables), 512 in all = 64 per XCD, each walking its
        // XCD's range of (8-sample block, tile) groups; fewer when there are fewer than 8 groups per XCD
        const int64_t groups = cdiv(p.batch, 8) * p.tiles_m * p.tiles_n;
        const unsigned grid = 64u * static_cast<unsigned>(std::min<int64_t>(8, cdiv(groups, 8)));
        const size_t smem = PV2_SMEM + ((p.conv && p.N <= PV2_ROWTAB_MAX) ? static_cast<size_t>((p.N + 3) / 4 * 16) : 0);
        if (p.out_rows == 2) hipLaunchKernelGGL(psg_gemm_v3_kernel<2>, dim3(grid), dim3(NTHREADS), smem, st, p);
        else if (p.out_rows) hipLaunchKernelGGL(psg_gemm_v3_kernel<1>, dim3(grid), dim3(NTHREADS), smem, st, p);
        else hipLaunchKernelGGL(psg_gemm_v3_kernel<0>, dim3(grid), dim3(NTHREADS), smem, st, p);
        return launch_status();
    }
    if (p.out_rows == 2) return KF_ERR_INVALID_ARGUMENT;   // the plain / scaled result exists on the persistent kernel only
    hipLaunchKernelGGL(psg_gemm_v2_kernel, dim3(static_cast<unsigned>(blocks)), dim3(NTHREADS), PV2_SMEM, st, p);
    return launch_status();
}

// columns of one phase copy: the last octet of an output row starts at O2 - 8 + ((k2 - 1) d2) / s2
inline int64_t conv_wq(int64_t O2, int k2, int d2, int s2) { return (O2 + ((k2 - 1) * d2) / s2 + 7) / 8 * 8; }

}  // namespace

namespace kf {
// used by kf_pairwise_score's k-tile-major path (kf_kernels.hip): the score GEMM of a layer whose per-sample gradients were
// formed by the v1 (transposing) kernel
int score_gemm_tiled(float* scores, int64_t ld, const void* P, const void* psg, int64_t Q, int64_t b, int64_t D, float scale, void* stream) {
    if (configure_once() != KF_OK) return KF_ERR_LAUNCH_FAILED;
    return launch_score_v2(scores, ld, reinterpret_cast<const uint16_t*>(P), reinterpret_cast<const uint16_t*>(psg), Q, b, D, scale,
                           as_stream(stream));
}

// used by the bf16 GEMM entry points (kf_kernels.hip) for large row-major NT products with a bf16 result: see RotateArgs
int rotate_gemm_v2(void* C, int64_t ldc, const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N, int64_t K,
                   float alpha, const float* row_add, int row_add_n, void* stream) {
    if (configure_once() != KF_OK) return KF_ERR_LAUNCH_FAILED;
    RotateArgs r;
    r.C = reinterpret_cast<uint16_t*>(C); r.ldc = ldc; r.A = reinterpret_cast<const uint16_t*>(A); r.lda = lda;
    r.B = reinterpret_cast<const uint16_t*>(B); r.ldb = ldb;
    r.M = static_cast<int>(M); r.N = static_cast<int>(N); r.KT = static_cast<int>(K / 64);
    r.tiles_m = static_cast<int>(cdiv(M, 256)); r.tiles_n = static_cast<int>(cdiv(N, 256));
    r.alpha = alpha; r.row_add = row_add; r.row_add_n = row_add_n;
    const int64_t blocks = 8 * cdiv(static_cast<int64_t>(r.tiles_m) * r.tiles_n, 8);
    if (engine_generation() == 3) {
        RotateV3Args v{};
        v.r = r;
        with_pp_issue([&](auto iss) {
            hipLaunchKernelGGL((rotate_gemm_v3_kernel<0, decltype(iss)::value>), dim3(static_cast<unsigned>(blocks)), dim3(pp::THREADS), pp::SMEM_BYTES, as_stream(stream), v);
        });
        return launch_status();
    }
    hipLaunchKernelGGL(rotate_gemm_v2_kernel, dim3(static_cast<unsigned>(blocks)), dim3(SV2_THREADS), 2 * 512 * 128, as_stream(stream), r);
    return launch_status();
}

// ------------------------------------------------------------------------------------------------
// EK-FAC preconditioning of a batch of query gradients, bf16 preset, on the round-3 engines (kf_precondition's low-precision path
// when O, I and R are multiples of 64): four eigenbasis rotations on the 256 x 256 wave-role-split loop and the per-query
// gradient on the persistent 128 x 128 kernel, instead of five products on the register-staged engine (its batched TN
// products ran at 0.02-0.07 PFLOP/s: 35 % of the device time of a BERT step at 2 048 train sequences).
//   1  GtT[q][o'][r] = sum_o  QgT[o'][o] G[(q r)][o]                      rotation, result K-contiguous per query
//   2  AtT[q][i'][r] = sum_i  QaT[i'][i] A[(q r)][i]  (+ Qa[I][i'])       same; rows i' >= I' stay zero
//   3  rot[q][o][i'] = (sum_r GtT[q][o][r] AtT[q][i'][r]) / (lambda + damping)[o][i']     (psg_gemm_v3_kernel<2>)
//   4  Tt[q][j][o]   = sum_i' Qa[j][i'] rot[(q o)][i']                    contraction padded to W64 = W rounded up to 64
//   5  P[q][m][n]    = scale sum_o Qg[m][o] Tt[(q n)][o]
// Rounding points are those of the register-staged path (bf16 after every product, fp32 accumulation) plus one: the gradient of
// step 3 is rounded before AND after the multiplication.
// ------------------------------------------------------------------------------------------------
namespace {
__global__ void pad_rows_bf16_kernel(uint16_t* dst, const uint16_t* src, int rows, int cols, int ld_dst) {   // dst[r][c] = src[r][c] or 0
    const int64_t total = static_cast<int64_t>(rows) * ld_dst;
    for (int64_t e = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; e < total; e += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const int r = static_cast<int>(e / ld_dst), c = static_cast<int>(e - static_cast<int64_t>(r) * ld_dst);
        dst[e] = c < cols ? src[static_cast<int64_t>(r) * cols + c] : static_cast<uint16_t>(0);
    }
}
__global__ void cast_f32_bf16_kernel(uint16_t* dst, const float* src, int64_t n) {
    for (int64_t e = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; e < n; e += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const uint32_t w = pack_bf16x2(src[e], 0.0f);
        dst[e] = static_cast<uint16_t>(w & 0xffffu);
    }
}
// XCD grid of a blocked rotation (RotateV3Args::xcd_m).  A = the eigenvector matrix (M x K, M = K = O or W: a few 256-row slabs of
// K * 512 bytes), B = the rows being rotated (N = samples x rows: hundreds of tiles).  In the linear n-major order every XCD
// streams ALL of A for each of its n-tiles: once A outgrows the 4 MB L2 (O >= 2048) that is tiles_n x |A| of Infinity-Cache
// traffic per launch (GPT-2 c_fc, 128 sequences: 4.9 GB counted against 0.4 GB of B) -- the "8-10x algorithmic" of VERDICT r05
// item 4.  With a gm x gn XCD grid an XCD works through ceil(tiles_m / gm) slabs only and B is read by gm XCDs instead.
// Measured (profiles/r06_rotate_xcd_grid.log, two passes): d = 3072 (12 slabs) gm = 4: 1.27 -> 1.10 ms (GPT-2, 128 x 512 rows;
// FETCH_SIZE 4.94 -> 3.38 GB), 1.12 -> 1.03 ms (BERT, 512 x 128); d = 14336 gm = 8: 1.40 -> 1.35 ms; d = 4096: 0.115 -> 0.112 ms
// at any gm; d = 2304 (9 slabs: no gm divides them -- an XCD row would idle) and d = 768 (A fits L2) are fastest in the linear
// order.  Hence: A beyond ~3 MB -> the largest gm in {8, 4, 2} that divides the m-tiles, else linear.  The kernel stays bound by
// its MFMA / LDS schedule (0.98 -> 1.13 PFLOP/s), not by this traffic: the gain is the 5-13 % above.
// KF_ROT_XCD_M = 0 / 2 / 4 / 8 overrides (A/B measurements).
inline int rotate_xcd_m(int tiles_m, int tiles_n, int64_t K) {
    if (const char* e = getenv("KF_ROT_XCD_M")) {
        const int g = atoi(e);
        return (g == 2 || g == 4 || g == 8) ? g : 0;
    }
    const int64_t slab = 256 * K * 2;                                  // bytes of one A tile
    if (static_cast<int64_t>(tiles_m) * slab <= (3 << 20)) return 0;   // all of A is L2 resident anyway
    for (int gm : {8, 4, 2})
        if (tiles_m % gm == 0 && tiles_n >= 2 * (8 / gm)) return gm;
    return 0;
}
// C[m, n] = alpha sum_k A[m, k] B[n, k] (+ col_add[m]) on the 256 x 256 loop, element (m, n) at (n / c_inner) * c_outer + m * c_inner + n % c_inner
int launch_rotate_blocked(uint16_t* C, const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, int64_t M, int64_t N, int64_t K,
                          float alpha, const float* col_add, int col_add_m, int64_t c_inner, int64_t c_outer, hipStream_t st) {
    if (K % 64 != 0 || N % 8 != 0 || c_inner % 8 != 0 || M >= (1LL << 31) - 256 || N >= (1LL << 31) - 256) return KF_ERR_INVALID_ARGUMENT;
    RotateV3Args v{};
    v.r.C = C; v.r.ldc = 0; v.r.A = A; v.r.lda = lda; v.r.B = B; v.r.ldb = ldb;
    v.r.M = static_cast<int>(M); v.r.N = static_cast<int>(N); v.r.KT = static_cast<int>(K / 64);
    v.r.tiles_m = static_cast<int>(cdiv(M, 256)); v.r.tiles_n = static_cast<int>(cdiv(N, 256));
    v.r.alpha = alpha; v.r.row_add = nullptr; v.r.row_add_n = 0;
    v.col_add = col_add; v.col_add_m = col_add_m; v.c_inner = c_inner; v.c_outer = c_outer;
    v.n_major = 1;   // the few m-tiles (eigenvectors: L2 resident) of one n-tile run back to back: the big operand is read once
    v.xcd_m = rotate_xcd_m(v.r.tiles_m, v.r.tiles_n, K);
    int64_t blocks = 8 * cdiv(static_cast<int64_t>(v.r.tiles_m) * v.r.tiles_n, 8);
    if (v.xcd_m > 0) blocks = 8 * cdiv(v.r.tiles_m, v.xcd_m) * cdiv(v.r.tiles_n, 8 / v.xcd_m);
    with_pp_issue([&](auto iss) { hipLaunchKernelGGL((rotate_gemm_v3_kernel<0, decltype(iss)::value>), dim3(static_cast<unsigned>(blocks)), dim3(pp::THREADS), pp::SMEM_BYTES, st, v); });
    return launch_status();
}
struct PrecondPlan { int64_t W64, gt, at, rot, qa, tt, qg, total; };
PrecondPlan precond_plan(int64_t q, int64_t R, int64_t O, int64_t W) {
    PrecondPlan p;
    p.W64 = (W + 63) / 64 * 64;
    p.gt = 0;
    p.at = p.gt + align256(2 * q * O * R);
    p.rot = p.at + align256(2 * q * p.W64 * R);
    p.qa = p.rot + align256(2 * q * O * p.W64);
    p.tt = p.qa + align256(2 * W * p.W64);
    p.qg = p.tt + align256(2 * q * W * O);
    p.total = p.qg + align256(2 * O * O);
    return p;
}
}  // namespace

int64_t precondition_v3_workspace_bytes(int64_t q, int64_t R, int64_t O, int64_t W) { return precond_plan(q, R, O, W).total; }

bool precondition_v3_eligible(int64_t q, int64_t R, int64_t O, int64_t I, int64_t W) {
    return engine_generation() == 3 && !getenv("KF_PRECOND_V3_OFF") && O % 64 == 0 && I % 64 == 0 && R % 64 == 0 && W % 8 == 0 && W >= I &&
           q * R < (1LL << 31) - 256 && q * std::max(O, W) < (1LL << 31) - 256 && O * R + 64 < (1LL << 31) &&
           ((W + 63) / 64 * 64) * R + 64 < (1LL << 31) && q <= 65535;
}

// Qg: fp32 [O, O], cast to bf16 per call -- or Qg_bf16: the same matrix already in bf16 (then Qg is not read);
// bias_row: row I of Qa (fp32, Ip entries) when the ones column is appended, else null
int precondition_v3(void* Pout, const void* G, const void* A, int64_t q, int64_t R, int64_t O, int64_t I, int append_ones, const float* Qg,
                    const void* Qg_bf16, const float* bias_row, int64_t Ip, const float* inv_lambda, float scale, const void* Qa_bf16,
                    const void* QgT_bf16, const void* QaT_bf16, int64_t W, void* workspace, void* stream) {
    if (configure_once() != KF_OK) return KF_ERR_LAUNCH_FAILED;
    hipStream_t st = as_stream(stream);
    const PrecondPlan p = precond_plan(q, R, O, W);
    char* ws = reinterpret_cast<char*>(workspace);
    uint16_t* gt = reinterpret_cast<uint16_t*>(ws + p.gt);
    uint16_t* at = reinterpret_cast<uint16_t*>(ws + p.at);
    uint16_t* rot = reinterpret_cast<uint16_t*>(ws + p.rot);
    uint16_t* qa = reinterpret_cast<uint16_t*>(ws + p.qa);
    uint16_t* tt = reinterpret_cast<uint16_t*>(ws + p.tt);
    uint16_t* qg = reinterpret_cast<uint16_t*>(ws + p.qg);
    const int64_t W64 = p.W64;
    // rows i' in [W, W64) of AtT are never written by step 2: zero (they are operand rows of step 3)
    if (hipMemsetAsync(at, 0, static_cast<size_t>(2 * q * W64 * R), st) != hipSuccess) return KF_ERR_LAUNCH_FAILED;
    hipLaunchKernelGGL(pad_rows_bf16_kernel, dim3(static_cast<unsigned>(std::min<int64_t>(cdiv(W * W64, 256), 2048))), dim3(256), 0, st, qa,
                       reinterpret_cast<const uint16_t*>(Qa_bf16), static_cast<int>(W), static_cast<int>(W), static_cast<int>(W64));
    const uint16_t* qg_used = reinterpret_cast<const uint16_t*>(Qg_bf16);
    if (!qg_used) {
        hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(static_cast<unsigned>(std::min<int64_t>(cdiv(O * O, 256), 2048))), dim3(256), 0, st, qg, Qg, O * O);
        qg_used = qg;
    }
    int rc = launch_rotate_blocked(gt, reinterpret_cast<const uint16_t*>(QgT_bf16), O, reinterpret_cast<const uint16_t*>(G), O, O, q * R, O, 1.0f,
                                   nullptr, 0, R, O * R, st);
    if (rc != KF_OK) return rc;
    rc = launch_rotate_blocked(at, reinterpret_cast<const uint16_t*>(QaT_bf16), W, reinterpret_cast<const uint16_t*>(A), I, W, q * R, I, 1.0f,
                               append_ones ? bias_row : nullptr, append_ones ? static_cast<int>(Ip) : 0, R, W64 * R, st);
    if (rc != KF_OK) return rc;
    PsgV2Args g{};
    g.out = rot; g.out_tile_stride = 0; g.out_rows = 2;
    g.A = gt; g.a_sample_stride = O * R; g.B = at; g.b_sample_stride = W64 * R;
    g.M = static_cast<int>(O); g.N = static_cast<int>(W64); g.K = static_cast<int>(R); g.batch = static_cast<int>(q);
    g.conv = 0; g.C = 1; g.k2 = 1; g.O2 = 8; g.s1 = 1; g.d1 = 1; g.s2 = 1; g.d2 = 1; g.Wq = 8; g.plane = 0; g.phase_stride = 0;
    g.mul = inv_lambda; g.ld_mul = static_cast<int>(Ip); g.mul_n = static_cast<int>(Ip);
    rc = launch_psg_v2(g, st);
    if (rc != KF_OK) return rc;
    rc = launch_rotate_blocked(tt, qa, W64, rot, W64, W, q * O, W64, 1.0f, nullptr, 0, O, W * O, st);
    if (rc != KF_OK) return rc;
    return launch_rotate_blocked(reinterpret_cast<uint16_t*>(Pout), qg_used, O, tt, O, O, q * W, O, scale, nullptr, 0, W, O * W, st);
}
}  // namespace kf

extern "C" {

namespace {
struct ConvPlan {
    int64_t O1, O2, O1p, O2p, Cp, Hp, Wq, Pp, Ipp, D;
    int64_t copies_bytes, grid_bytes, psg_bytes;
    bool ok;
};
// cp_force != 0: that many (zero-padded) channels instead of C rounded up to 8, without the bound on the padding's extra work --
// the dense Lambda form of a layer with very few input channels (lambda_conv_channels below)
ConvPlan conv_plan(int64_t b, int64_t C, int64_t H, int64_t W, int64_t O, int k1, int k2, int s1, int s2, int p1, int p2, int d1, int d2,
                   int64_t cp_force = 0) {
    ConvPlan c{};
    c.O1 = (H + 2 * p1 - d1 * (k1 - 1) - 1) / s1 + 1;
    c.O2 = (W + 2 * p2 - d2 * (k2 - 1) - 1) / s2 + 1;
    c.ok = c.O1 > 0 && c.O2 > 0;
    if (!c.ok) return c;
    c.O2p = (c.O2 + 7) / 8 * 8;
    c.O1p = c.O1;
    while ((c.O1p * c.O2p) % 64 != 0) ++c.O1p;
    c.Cp = cp_force ? cp_force : (C + 7) / 8 * 8;
    c.Pp = c.O1p * c.O2p;
    c.Ipp = c.Cp * k1 * k2;
    c.D = O * c.Ipp;
    c.Hp = std::max<int64_t>(H + 2 * p1, s1 * (c.O1p - 1) + static_cast<int64_t>(d1) * (k1 - 1) + 1);
    c.Wq = conv_wq(c.O2p, k2, d2, s2);
    // padding the output grid and the channels may at most double the contraction work of the gradient kernel
    c.ok = c.Pp <= 2 * c.O1 * c.O2 && (cp_force != 0 || c.Cp * c.Pp <= 3 * C * c.O1 * c.O2) && c.D % 64 == 0 && O >= 8;
    c.copies_bytes = align256(2 * s2 * b * c.Cp * c.Hp * c.Wq + 64);
    c.grid_bytes = (c.O1p != c.O1 || c.O2p != c.O2) ? align256(2 * b * O * c.Pp) : 0;
    c.psg_bytes = align256(2 * b * c.D);
    return c;
}
}  // namespace

int64_t kf_pairwise_conv2d_workspace_bytes(int64_t b, int64_t C, int64_t H, int64_t W, int64_t O, int k1, int k2, int s1, int s2,
                                           int p1, int p2, int d1, int d2) {
    const ConvPlan c = conv_plan(b, C, H, W, O, k1, k2, s1, s2, p1, p2, d1, d2);
    if (!c.ok) return -1;
    return c.copies_bytes + c.grid_bytes + c.psg_bytes;
}

int kf_pairwise_score_conv2d(float* scores, int64_t ld_scores, const void* P_tiled, int64_t Q, const void* G_nchw, const void* x,
                             int64_t b, int64_t C, int64_t H, int64_t W, int64_t O, int k1, int k2, int s1, int s2, int p1, int p2,
                             int d1, int d2, float scale, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!scores || !P_tiled || !G_nchw || !x || Q < 0 || b < 0 || C <= 0 || O <= 0) return KF_ERR_INVALID_ARGUMENT;
    const ConvPlan c = conv_plan(b, C, H, W, O, k1, k2, s1, s2, p1, p2, d1, d2);
    // eligibility (the host checks the same conditions and uses the materialised-patch path otherwise)
    if (!c.ok) return KF_ERR_INVALID_ARGUMENT;
    if (((reinterpret_cast<uintptr_t>(P_tiled) | reinterpret_cast<uintptr_t>(G_nchw) | reinterpret_cast<uintptr_t>(x)) & 15) != 0)
        return KF_ERR_INVALID_ARGUMENT;
    if (!workspace || workspace_bytes < c.copies_bytes + c.grid_bytes + c.psg_bytes) return KF_ERR_WORKSPACE_TOO_SMALL;
    if (Q == 0 || b == 0) return KF_OK;
    if (configure_once() != KF_OK) return KF_ERR_LAUNCH_FAILED;
    hipStream_t st = as_stream(stream);
    uint16_t* copies = reinterpret_cast<uint16_t*>(workspace);
    uint16_t* grid = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(workspace) + c.copies_bytes);
    uint16_t* psg = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(workspace) + c.copies_bytes + c.grid_bytes);
    PadArgs pa;
    pa.out = copies; pa.x = reinterpret_cast<const uint16_t*>(x); pa.planes = b * c.Cp; pa.C = static_cast<int>(C); pa.Cp = static_cast<int>(c.Cp);
    pa.H = static_cast<int>(H); pa.W = static_cast<int>(W); pa.Hp = static_cast<int>(c.Hp); pa.Wq = static_cast<int>(c.Wq);
    pa.p1 = p1; pa.p2 = p2; pa.s2 = s2;
    const int64_t chunks = s2 * b * c.Cp * c.Hp * (c.Wq / 8);
    hipLaunchKernelGGL(conv_pad_phases_kernel, dim3(static_cast<unsigned>(std::min<int64_t>(cdiv(chunks, 256), 1 << 20))), dim3(256), 0,
                       st, pa);
    const uint16_t* gsrc = reinterpret_cast<const uint16_t*>(G_nchw);
    if (c.grid_bytes) {
        hipLaunchKernelGGL(pad_grid_kernel, dim3(static_cast<unsigned>(std::min<int64_t>(cdiv(b * O * c.Pp, 256), 1 << 20))), dim3(256), 0, st,
                           grid, gsrc, b * O, static_cast<int>(c.O1), static_cast<int>(c.O2), static_cast<int>(c.O1p), static_cast<int>(c.O2p));
        gsrc = grid;
    }
    PsgV2Args g{};
    g.out = psg; g.out_tile_stride = b * 64; g.out_rows = 0;
    g.A = gsrc; g.a_sample_stride = O * c.Pp;
    g.B = copies; g.b_sample_stride = c.Cp * c.Hp * c.Wq;
    g.M = static_cast<int>(O); g.N = static_cast<int>(c.Ipp); g.K = static_cast<int>(c.Pp); g.batch = static_cast<int>(b);
    g.conv = 1; g.C = static_cast<int>(c.Cp); g.k2 = k2; g.O2 = static_cast<int>(c.O2p); g.s1 = s1; g.d1 = d1; g.s2 = s2; g.d2 = d2;
    g.Wq = static_cast<int>(c.Wq); g.plane = static_cast<int>(c.Hp * c.Wq); g.phase_stride = b * c.Cp * c.Hp * c.Wq;
    // KF_CONV_CHUNKS = n (measurements only: profiles/r05_conv_chunk_pipeline_negative.log): the gradient -> score hand-over in n
    // chunks of output channels through ONE chunk-sized region of the workspace, so that a chunk written by the gradient kernel is
    // still in the 256 MB Infinity Cache when the score GEMM reads it.  Chunk boundaries are whole k-tiles (o_lo * Ipp % 64 == 0).
    int chunks_env = 1;
    if (const char* e = getenv("KF_CONV_CHUNKS")) chunks_env = std::max(1, atoi(e));
    if (chunks_env > 1) {
        int64_t align_o = 1;
        while ((align_o * c.Ipp) % 64 != 0) align_o *= 2;
        const int64_t per = cdiv(cdiv(O, chunks_env), align_o) * align_o;
        for (int64_t o_lo = 0; o_lo < O; o_lo += per) {
            const int64_t o_n = std::min<int64_t>(per, O - o_lo), kt0 = o_lo * c.Ipp / 64;
            PsgV2Args gc = g;
            gc.A = gsrc + o_lo * c.Pp; gc.M = static_cast<int>(o_n); gc.out = psg;   // every chunk through the same region
            const int rc = launch_psg_v2(gc, st);
            if (rc != KF_OK) return rc;
            const int rs = launch_score_v2(scores, ld_scores, reinterpret_cast<const uint16_t*>(P_tiled) + kt0 * Q * 64, psg, Q, b, o_n * c.Ipp, scale, st);
            if (rs != KF_OK) return rs;
        }
        return KF_OK;
    }
    int rc = launch_psg_v2(g, st);
    if (rc != KF_OK) return rc;
    return launch_score_v2(scores, ld_scores, reinterpret_cast<const uint16_t*>(P_tiled), psg, Q, b, c.D, scale, st);
}

namespace {
// Channels (incl. zero padding) the dense Lambda form carries the patch axis with.  Normally C rounded up to 8; when that does not
// give whole 64-deep k-steps of the Q_A product (ResNet-9's first layer: 8 * 9 = 72) the next multiple of 8 that does (64 * 9 = 576)
// is taken -- zero channels cost contraction work only, and such a layer is tiny: accepted up to 2e11 flop of per-sample-gradient
// GEMM per call (1 000 images of 32 x 32: 7.5e10).  Before round 6 that layer fell back to materialised fp32 patches + three fp32
// GEMMs: 1.35 ms per batch for 3.9 GFLOP (tools/r06_layer_times.py).  0: the default plan applies.
int64_t lambda_conv_cp_force(int64_t b, int64_t C, int64_t H, int64_t W, int64_t O, int k1, int k2, int s1, int s2, int p1, int p2, int d1, int d2) {
    const int64_t cp = (C + 7) / 8 * 8;
    if ((cp * k1 * k2) % 64 == 0) return 0;
    const ConvPlan base = conv_plan(b, C, H, W, O, k1, k2, s1, s2, p1, p2, d1, d2, cp);
    if (base.O1 <= 0 || base.O2 <= 0) return 0;
    for (int64_t wide = cp + 8; wide <= 64; wide += 8)
        if ((wide * k1 * k2) % 64 == 0)
            return 2.0 * static_cast<double>(b) * O * wide * k1 * k2 * base.Pp <= 2e11 ? wide : 0;
    return 0;
}
}  // namespace

int64_t kf_lambda_conv2d_channels(int64_t b, int64_t C, int64_t H, int64_t W, int64_t O, int k1, int k2, int s1, int s2, int p1, int p2,
                                  int d1, int d2) {
    if (kf_lambda_conv2d_workspace_bytes(b, C, H, W, O, k1, k2, s1, s2, p1, p2, d1, d2) < 0) return -1;
    const int64_t force = lambda_conv_cp_force(b, C, H, W, O, k1, k2, s1, s2, p1, p2, d1, d2);
    return force ? force : (C + 7) / 8 * 8;
}

int64_t kf_lambda_conv2d_workspace_bytes(int64_t b, int64_t C, int64_t H, int64_t W, int64_t O, int k1, int k2, int s1, int s2,
                                         int p1, int p2, int d1, int d2) {
    const ConvPlan c = conv_plan(b, C, H, W, O, k1, k2, s1, s2, p1, p2, d1, d2,
                                 lambda_conv_cp_force(b, C, H, W, O, k1, k2, s1, s2, p1, p2, d1, d2));
    // the dense Lambda form contracts over the (padded) patch axis on the 64-deep LDS-DMA engine and folds rows (o, sample)
    // of a 256-row tile into at most two rows of Lambda: whole k-steps, real output grid, >= 256 samples
    if (!c.ok || c.grid_bytes != 0 || c.Ipp % 64 != 0 || b < 256 || O * b >= (1LL << 31) - 256) return -1;
    return c.copies_bytes + align256(2 * b * O * c.Ipp);
}

int kf_lambda_conv2d_accum(float* Lambda, int64_t ld_lambda, const void* Gt_nchw, const void* x, int64_t b, int64_t C, int64_t H,
                           int64_t W, int64_t O, int k1, int k2, int s1, int s2, int p1, int p2, int d1, int d2, const void* QaT_perm,
                           int64_t n_out, int64_t ldq, float scale, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!Lambda || !Gt_nchw || !x || !QaT_perm || b < 0 || C <= 0 || O <= 0 || n_out <= 0) return KF_ERR_INVALID_ARGUMENT;
    const int64_t need = kf_lambda_conv2d_workspace_bytes(b, C, H, W, O, k1, k2, s1, s2, p1, p2, d1, d2);
    if (need < 0) return KF_ERR_INVALID_ARGUMENT;
    const ConvPlan c = conv_plan(b, C, H, W, O, k1, k2, s1, s2, p1, p2, d1, d2,
                                 lambda_conv_cp_force(b, C, H, W, O, k1, k2, s1, s2, p1, p2, d1, d2));
    if (ldq < c.Ipp || ldq % 8 != 0 || n_out > ld_lambda ||
        ((reinterpret_cast<uintptr_t>(Gt_nchw) | reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(QaT_perm)) & 15) != 0)
        return KF_ERR_INVALID_ARGUMENT;
    if (!workspace || workspace_bytes < need) return KF_ERR_WORKSPACE_TOO_SMALL;
    if (configure_once() != KF_OK) return KF_ERR_LAUNCH_FAILED;
    hipStream_t st = as_stream(stream);
    uint16_t* copies = reinterpret_cast<uint16_t*>(workspace);
    uint16_t* psg = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(workspace) + c.copies_bytes);
    PadArgs pa;
    pa.out = copies; pa.x = reinterpret_cast<const uint16_t*>(x); pa.planes = b * c.Cp; pa.C = static_cast<int>(C); pa.Cp = static_cast<int>(c.Cp);
    pa.H = static_cast<int>(H); pa.W = static_cast<int>(W); pa.Hp = static_cast<int>(c.Hp); pa.Wq = static_cast<int>(c.Wq);
    pa.p1 = p1; pa.p2 = p2; pa.s2 = s2;
    const int64_t chunks = s2 * b * c.Cp * c.Hp * (c.Wq / 8);
    hipLaunchKernelGGL(conv_pad_phases_kernel, dim3(static_cast<unsigned>(std::min<int64_t>(cdiv(chunks, 256), 1 << 20))), dim3(256), 0,
                       st, pa);
    // per-sample gradients in the gradient eigenbasis, rows ordered (o, sample), patch axis (ky, kx, c)
    PsgV2Args g{};
    g.out = psg; g.out_tile_stride = 0; g.out_rows = 1;
    g.A = reinterpret_cast<const uint16_t*>(Gt_nchw); g.a_sample_stride = O * c.Pp;
    g.B = copies; g.b_sample_stride = c.Cp * c.Hp * c.Wq;
    g.M = static_cast<int>(O); g.N = static_cast<int>(c.Ipp); g.K = static_cast<int>(c.Pp); g.batch = static_cast<int>(b);
    g.conv = 1; g.C = static_cast<int>(c.Cp); g.k2 = k2; g.O2 = static_cast<int>(c.O2p); g.s1 = s1; g.d1 = d1; g.s2 = s2; g.d2 = d2;
    g.Wq = static_cast<int>(c.Wq); g.plane = static_cast<int>(c.Hp * c.Wq); g.phase_stride = b * c.Cp * c.Hp * c.Wq;
    int rc = launch_psg_v2(g, st);
    if (rc != KF_OK) return rc;
    // Lambda[o, i'] += scale^2 * sum_n ( sum_j psg[(o, n), j] Qa[j, i'] )^2
    RotateV3Args v{};
    v.r.C = nullptr; v.r.ldc = 0; v.r.A = psg; v.r.lda = c.Ipp; v.r.B = reinterpret_cast<const uint16_t*>(QaT_perm); v.r.ldb = ldq;
    v.r.M = static_cast<int>(O * b); v.r.N = static_cast<int>(n_out); v.r.KT = static_cast<int>(c.Ipp / 64);
    v.r.tiles_m = static_cast<int>(cdiv(O * b, 256)); v.r.tiles_n = static_cast<int>(cdiv(n_out, 256));
    v.r.alpha = scale; v.r.row_add = nullptr; v.r.row_add_n = 0;
    v.sumsq = Lambda; v.ld_sumsq = ld_lambda; v.group_rows = static_cast<int>(b);
    const int64_t blocks = 8 * cdiv(static_cast<int64_t>(v.r.tiles_m) * v.r.tiles_n, 8);
    with_pp_issue([&](auto iss) { hipLaunchKernelGGL((rotate_gemm_v3_kernel<1, decltype(iss)::value>), dim3(static_cast<unsigned>(blocks)), dim3(pp::THREADS), pp::SMEM_BYTES, st, v); });
    return launch_status();
}

int64_t kf_pairwise_rows_workspace_bytes(int64_t b, int64_t R, int64_t O, int64_t Ip) {
    return align256(2 * b * O * R) + align256(2 * b * Ip * R) + align256(2 * b * O * Ip);
}

int kf_pairwise_score_rows(float* scores, int64_t ld_scores, const void* P_tiled, int64_t Q, const void* G, const void* A, int64_t b,
                           int64_t R, int64_t O, int64_t I, int64_t Ip, int append_ones, float scale, void* workspace,
                           int64_t workspace_bytes, void* stream) {
    return kf_pairwise_score_rows2(scores, ld_scores, P_tiled, Q, G, A, b, nullptr, nullptr, 0, R, O, I, Ip, append_ones, scale, workspace,
                                   workspace_bytes, stream);
}

int kf_pairwise_score_rows2(float* scores, int64_t ld_scores, const void* P_tiled, int64_t Q, const void* G, const void* A, int64_t b0,
                            const void* G1, const void* A1, int64_t b1, int64_t R, int64_t O, int64_t I, int64_t Ip, int append_ones,
                            float scale, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!scores || !P_tiled || !G || !A || Q < 0 || b0 < 0 || b1 < 0 || R <= 0 || O <= 0 || I <= 0) return KF_ERR_INVALID_ARGUMENT;
    if (b1 > 0 && (!G1 || !A1)) return KF_ERR_INVALID_ARGUMENT;
    if (R % 64 != 0 || O % 8 != 0 || I % 8 != 0 || Ip % 8 != 0 || Ip < I + (append_ones ? 1 : 0) || (O * Ip) % 64 != 0)
        return KF_ERR_INVALID_ARGUMENT;
    if (((reinterpret_cast<uintptr_t>(P_tiled) | reinterpret_cast<uintptr_t>(G) | reinterpret_cast<uintptr_t>(A) |
          reinterpret_cast<uintptr_t>(G1) | reinterpret_cast<uintptr_t>(A1)) & 15) != 0)
        return KF_ERR_INVALID_ARGUMENT;
    const int64_t b = b0 + b1;
    if (!workspace || workspace_bytes < kf_pairwise_rows_workspace_bytes(b, R, O, Ip)) return KF_ERR_WORKSPACE_TOO_SMALL;
    if (Q == 0 || b == 0) return KF_OK;
    if (configure_once() != KF_OK) return KF_ERR_LAUNCH_FAILED;
    if (b > 65535) return KF_ERR_INVALID_ARGUMENT;
    hipStream_t st = as_stream(stream);
    uint16_t* gt = reinterpret_cast<uint16_t*>(workspace);
    uint16_t* at = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(workspace) + align256(2 * b * O * R));
    uint16_t* psg = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(at) + align256(2 * b * Ip * R));
    int64_t tn_min_r = 128;   // T = 64 stays on the persistent 128 x 128 kernel; KF_TN_MIN_R: measurements, tests
    if (const char* e = getenv("KF_TN_MIN_R")) tn_min_r = std::max<int64_t>(64, atoll(e));
    if (tn_enabled() && R >= tn_min_r && O % 256 == 0 && I % 256 == 0 && R * std::max(O, I) < (1LL << 31) &&
        (Ip == I || (append_ones && Ip == I + 8))) {
        // K-major path: the hooked [t][feature] tensors are the operands; the two segments are two base pointers
        PsgTnArgs g{};
        g.out = psg; g.out_tile_stride = b * 64;
        g.G[0] = reinterpret_cast<const uint16_t*>(G); g.A[0] = reinterpret_cast<const uint16_t*>(A);
        g.G[1] = reinterpret_cast<const uint16_t*>(G1); g.A[1] = reinterpret_cast<const uint16_t*>(A1);
        g.b0 = static_cast<int>(b0);
        g.O = static_cast<int>(O); g.I = static_cast<int>(I); g.Ip = static_cast<int>(Ip); g.KT = static_cast<int>(R / 64); g.batch = static_cast<int>(b);
        g.tiles_m = static_cast<int>(O / 256); g.tiles_n = static_cast<int>(I / 256); g.ones = append_ones ? 1 : 0;
        const int64_t items = b * g.tiles_m * g.tiles_n;
        if (items >= (1LL << 31)) return KF_ERR_INVALID_ARGUMENT;
        with_tn_image([&](auto img) {
            hipLaunchKernelGGL((psg_gemm_tn_kernel<decltype(img)::value>), dim3(static_cast<unsigned>(8 * cdiv(items, 8))), dim3(pptn::THREADS), pptn::SMEM_BYTES, st, g);
        });
        if (launch_status() != KF_OK) return KF_ERR_LAUNCH_FAILED;
        return launch_score_v2(scores, ld_scores, reinterpret_cast<const uint16_t*>(P_tiled), psg, Q, b, O * Ip, scale, st);
    }
    // the two segments (train micro-batches) land one behind the other in the transposed copies: from there on ONE batch of b0 + b1
    const void* seg_g[2] = {G, G1};
    const void* seg_a[2] = {A, A1};
    const int64_t seg_b[2] = {b0, b1};
    int64_t done = 0;
    for (int seg = 0; seg < 2; ++seg) {
        if (seg_b[seg] == 0) continue;
        TransposeArgs t;
        t.out = gt + done * O * R; t.x = reinterpret_cast<const uint16_t*>(seg_g[seg]); t.T = static_cast<int>(R); t.C = static_cast<int>(O);
        t.Cp = static_cast<int>(O); t.ones = 0; t.mask = nullptr; t.mask_dtype = 0;
        hipLaunchKernelGGL(transpose_rows_kernel, dim3(static_cast<unsigned>(R / 64), static_cast<unsigned>(cdiv(O, 64)), static_cast<unsigned>(seg_b[seg])),
                           dim3(256), 0, st, t);
        t.out = at + done * Ip * R; t.x = reinterpret_cast<const uint16_t*>(seg_a[seg]); t.C = static_cast<int>(I); t.Cp = static_cast<int>(Ip);
        t.ones = append_ones ? 1 : 0;
        hipLaunchKernelGGL(transpose_rows_kernel, dim3(static_cast<unsigned>(R / 64), static_cast<unsigned>(cdiv(Ip, 64)), static_cast<unsigned>(seg_b[seg])),
                           dim3(256), 0, st, t);
        done += seg_b[seg];
    }
    PsgV2Args g{};
    g.out = psg; g.out_tile_stride = b * 64; g.out_rows = 0;
    g.A = gt; g.a_sample_stride = O * R; g.B = at; g.b_sample_stride = Ip * R;
    g.M = static_cast<int>(O); g.N = static_cast<int>(Ip); g.K = static_cast<int>(R); g.batch = static_cast<int>(b);
    g.conv = 0; g.C = 1; g.k2 = 1; g.O2 = 8; g.s1 = 1; g.d1 = 1; g.s2 = 1; g.d2 = 1; g.Wq = 8; g.plane = 0; g.phase_stride = 0;
    int rc = launch_psg_v2(g, st);
    if (rc != KF_OK) return rc;
    return launch_score_v2(scores, ld_scores, reinterpret_cast<const uint16_t*>(P_tiled), psg, Q, b, O * Ip, scale, st);
}

int kf_rotate_rows_transposed_bf16(void* out, const void* X, int64_t n, int64_t R, int64_t d, const void* QT, int64_t ldq, int64_t m,
                                   const float* bias, int64_t bias_n, void* stream) {
    if (!out || !X || !QT || n < 0 || R <= 0 || d <= 0 || m <= 0 || ldq < d || bias_n < 0 || bias_n > m || (bias_n > 0 && !bias))
        return KF_ERR_INVALID_ARGUMENT;
    if (d % 64 != 0 || R % 8 != 0 || ldq % 8 != 0 || m * R + 64 >= (1LL << 31) ||
        ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(QT)) & 15) != 0)
        return KF_ERR_INVALID_ARGUMENT;
    if (n == 0) return KF_OK;
    if (configure_once() != KF_OK) return KF_ERR_LAUNCH_FAILED;
    // roles swapped (tile rows = rows of Q^T, tile columns = the n R rows of X): the blocked epilogue writes element
    // (m', (sample, r)) at sample * m R + m' R + r, i.e. K-contiguous per sample
    return launch_rotate_blocked(reinterpret_cast<uint16_t*>(out), reinterpret_cast<const uint16_t*>(QT), ldq,
                                 reinterpret_cast<const uint16_t*>(X), d, m, n * R, d, 1.0f, bias_n > 0 ? bias : nullptr,
                                 static_cast<int>(bias_n), R, m * R, as_stream(stream));
}

int kf_lambda_rows_accum(float* Lambda, int64_t ld_lambda, const void* GtT, const void* AtT, int64_t b, int64_t R, int64_t O, int64_t W,
                         int64_t Ip, float scale, void* stream) {
    if (!Lambda || !GtT || !AtT || b < 0 || R <= 0 || O <= 0 || W <= 0 || Ip <= 0 || Ip > W || ld_lambda < Ip) return KF_ERR_INVALID_ARGUMENT;
    if (R % 64 != 0 || ((reinterpret_cast<uintptr_t>(GtT) | reinterpret_cast<uintptr_t>(AtT)) & 15) != 0 || O >= (1LL << 24) ||
        W >= (1LL << 24) || b * (R / 64) >= (1LL << 30))
        return KF_ERR_INVALID_ARGUMENT;
    if (b == 0) return KF_OK;
    if (configure_once() != KF_OK) return KF_ERR_LAUNCH_FAILED;
    LambdaRowsArgs a;
    a.L = Lambda; a.ldl = ld_lambda; a.G = reinterpret_cast<const uint16_t*>(GtT); a.A = reinterpret_cast<const uint16_t*>(AtT);
    a.O = static_cast<int>(O); a.W = static_cast<int>(W); a.Ip = static_cast<int>(Ip); a.KS = static_cast<int>(R / 64);
    a.batch = static_cast<int>(b); a.tiles_m = static_cast<int>(cdiv(O, 256)); a.tiles_n = static_cast<int>(cdiv(Ip, 128));
    a.scale2 = scale * scale;
    // one workgroup per CU (144 KB of LDS): items run in rounds of 256.  Split the samples so that the last round is nearly full;
    // among 1-4 rounds take the cheapest by (k-tiles per item + ~6 k-tiles for the prologue and the 32 K atomics of an item).
    const int64_t tiles = static_cast<int64_t>(a.tiles_m) * a.tiles_n;
    int64_t zblocks = 1, best = INT64_MAX;
    a.zchunk = a.batch;
    for (int rounds = 1; rounds <= 4; ++rounds) {
        const int64_t want = std::max<int64_t>(1, std::min<int64_t>(b, rounds * 256 / tiles));
        const int64_t chunk = cdiv(b, want), blocks = cdiv(b, chunk);
        const int64_t cost = cdiv(blocks * tiles, 256) * (chunk * a.KS + 6);
        if (cost < best) { best = cost; zblocks = blocks; a.zchunk = static_cast<int>(chunk); }
    }
    a.zblocks = static_cast<int>(zblocks);
    const int64_t blocks = 8 * cdiv(zblocks * tiles, 8);
    if (blocks >= (1LL << 31)) return KF_ERR_INVALID_ARGUMENT;
    // LDS-DMA requests of a k-tile issued in the L segment, the rest between the MFMA groups of the M segment.  Measured
    // (profiles/r04_lreq_ab.log): ALL of them in M is the fastest on every shape -- 768 x 3073 809 -> 929 TFLOP/s, 3072 x 769
    // 802 -> 854, Llama 4096^2 995 -> 1 041 -- the L segment (16 fragment reads + 6 requests at 100-185 cycles each) was
    // longer than the M segment it alternates with.  KF_PP64_LREQ = 6 / 4 / 3 / 2: measurements.
    int lreq = 0;
    if (const char* e = getenv("KF_PP64_LREQ")) lreq = atoi(e);
    const dim3 grid(static_cast<unsigned>(blocks)), block(pp64::THREADS);
    if (lreq == 4) hipLaunchKernelGGL(lambda_rows_kernel<4>, grid, block, PP64_SMEM, as_stream(stream), a);
    else if (lreq == 3) hipLaunchKernelGGL(lambda_rows_kernel<3>, grid, block, PP64_SMEM, as_stream(stream), a);
    else if (lreq == 2) hipLaunchKernelGGL(lambda_rows_kernel<2>, grid, block, PP64_SMEM, as_stream(stream), a);
    else if (lreq == 0) hipLaunchKernelGGL(lambda_rows_kernel<0>, grid, block, PP64_SMEM, as_stream(stream), a);
    else hipLaunchKernelGGL(lambda_rows_kernel<6>, grid, block, PP64_SMEM, as_stream(stream), a);
    return launch_status();
}

int64_t kf_syrk_rows_workspace_bytes(int64_t b, int64_t T, int64_t d_in, int append_ones) {
    const int64_t W = (d_in + (append_ones ? 1 : 0) + 7) / 8 * 8;
    return align256(2 * b * W * T) + cov_stage_bytes(W);
}

int kf_syrk_rows_bf16(float* C, int64_t ldc, const void* X, int64_t b, int64_t T, int64_t d_in, const void* mask, int mask_dtype,
                      int append_ones, float alpha, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!C || !X || b < 0 || T <= 0 || d_in <= 0) return KF_ERR_INVALID_ARGUMENT;
    if (T % 64 != 0 || d_in % 8 != 0 || (reinterpret_cast<uintptr_t>(X) & 15) != 0 || b > 65535 || d_in >= 32768) return KF_ERR_INVALID_ARGUMENT;
    if (mask && mask_dtype != KF_I64 && mask_dtype != KF_I32 && mask_dtype != KF_U8) return KF_ERR_UNSUPPORTED_DTYPE;
    if (!workspace || workspace_bytes < kf_syrk_rows_workspace_bytes(b, T, d_in, append_ones)) return KF_ERR_WORKSPACE_TOO_SMALL;
    if (b == 0) return KF_OK;
    if (configure_once() != KF_OK) return KF_ERR_LAUNCH_FAILED;
    hipStream_t st = as_stream(stream);
    const int64_t d = d_in + (append_ones ? 1 : 0), W = (d + 7) / 8 * 8;
    if (!mask && tn_enabled() && d_in >= 256 && b * T < (1LL << 30)) {
        // unmasked rows: X^T X on the K-major loop straight from the hooked tensor + the bias row / column as a column sum
        CovFinalizeArgs f{};
        f.out = C; f.ldc = ldc; f.d = static_cast<int>(d_in); f.conv = 0; f.alpha = alpha;
        float* stage = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(workspace) + align256(2 * b * W * T));
        return launch_cov_tn(stage, reinterpret_cast<const uint16_t*>(X), d_in, b * T, d_in, f, append_ones != 0, st);
    }
    uint16_t* xt = reinterpret_cast<uint16_t*>(workspace);
    TransposeArgs t;
    t.out = xt; t.x = reinterpret_cast<const uint16_t*>(X); t.T = static_cast<int>(T); t.C = static_cast<int>(d_in); t.Cp = static_cast<int>(W);
    t.ones = append_ones ? 1 : 0; t.mask = mask; t.mask_dtype = mask_dtype;
    hipLaunchKernelGGL(transpose_rows_kernel, dim3(static_cast<unsigned>(T / 64), static_cast<unsigned>(cdiv(W, 64)), static_cast<unsigned>(b)),
                       dim3(256), 0, st, t);
    CovV2Args c{};
    c.stage = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(workspace) + align256(2 * b * W * T));
    c.X = xt; c.sample_stride = W * T;
    c.N = static_cast<int>(W); c.K = static_cast<int>(T); c.batch = static_cast<int>(b); c.conv = 0;
    CovFinalizeArgs f{};
    f.out = C; f.ldc = ldc; f.d = static_cast<int>(d); f.conv = 0; f.alpha = alpha;
    return launch_cov_v2(c, f, st);
}

int64_t kf_syrk_rows_f32_workspace_bytes(int64_t n, int64_t d_in) {
    const int64_t rows_pad = cdiv(n, 64) * 64;
    return align256(3 * 2 * rows_pad * d_in) + cov_stage_bytes(d_in);
}

int kf_syrk_rows_f32(float* C, int64_t ldc, const void* X, int64_t n, int64_t d_in, const void* mask, int mask_dtype, int append_ones,
                     float alpha, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!C || !X || n < 0 || d_in <= 0) return KF_ERR_INVALID_ARGUMENT;
    if (d_in % 8 != 0 || d_in < 256 || d_in >= 32768 || (reinterpret_cast<uintptr_t>(X) & 15) != 0 || n > 65535LL * 64) return KF_ERR_INVALID_ARGUMENT;   // (one grid y-block per 64 rows)
    if (mask && mask_dtype != KF_I64 && mask_dtype != KF_I32 && mask_dtype != KF_U8 && mask_dtype != KF_F32) return KF_ERR_UNSUPPORTED_DTYPE;
    if (!workspace || workspace_bytes < kf_syrk_rows_f32_workspace_bytes(n, d_in)) return KF_ERR_WORKSPACE_TOO_SMALL;
    if (n == 0) return KF_OK;
    if (configure_once() != KF_OK) return KF_ERR_LAUNCH_FAILED;
    hipStream_t st = as_stream(stream);
    const int64_t rows_pad = cdiv(n, 64) * 64, plane = rows_pad * d_in;
    SplitArgs sp{};
    sp.planes = reinterpret_cast<uint16_t*>(workspace); sp.plane_stride = plane;
    sp.x = reinterpret_cast<const float*>(X); sp.n = n; sp.rows_pad = rows_pad; sp.d = static_cast<int>(d_in);
    sp.mask = mask; sp.mask_dtype = mask_dtype; sp.C = C; sp.ldc = ldc; sp.ones = append_ones ? 1 : 0; sp.alpha = alpha;
    hipLaunchKernelGGL(split_rows_f32_kernel, dim3(static_cast<unsigned>(cdiv(d_in, 1024)), static_cast<unsigned>(rows_pad / 64)), dim3(256), 0, st, sp);
    CovFinalizeArgs f{};
    f.out = C; f.ldc = ldc; f.d = static_cast<int>(d_in); f.conv = 0; f.alpha = alpha;
    float* stage = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(workspace) + align256(3 * 2 * plane));
    return launch_cov_tn(stage, sp.planes, d_in, rows_pad, d_in, f, false, st, plane);
}

int64_t kf_syrk_planes_workspace_bytes(int64_t d) { return cov_stage_bytes(d); }

int kf_syrk_planes_bf16(float* C, int64_t ldc, const void* X, int64_t b, int64_t d, int64_t K, float alpha, void* workspace,
                        int64_t workspace_bytes, void* stream) {
    if (!C || !X || b < 0 || d <= 0 || K <= 0) return KF_ERR_INVALID_ARGUMENT;
    if (K % 64 != 0 || (reinterpret_cast<uintptr_t>(X) & 15) != 0 || b > 65535 || d >= 32768) return KF_ERR_INVALID_ARGUMENT;
    if (!workspace || workspace_bytes < cov_stage_bytes(d)) return KF_ERR_WORKSPACE_TOO_SMALL;
    if (b == 0) return KF_OK;
    if (configure_once() != KF_OK) return KF_ERR_LAUNCH_FAILED;
    CovV2Args c{};
    c.stage = reinterpret_cast<float*>(workspace);
    c.X = reinterpret_cast<const uint16_t*>(X); c.sample_stride = d * K;
    c.N = static_cast<int>(d); c.K = static_cast<int>(K); c.batch = static_cast<int>(b); c.conv = 0;
    CovFinalizeArgs f{};
    f.out = C; f.ldc = ldc; f.d = static_cast<int>(d); f.conv = 0; f.alpha = alpha;
    return launch_cov_v2(c, f, as_stream(stream));
}

int64_t kf_conv2d_cov_workspace_bytes(int64_t b, int64_t C, int64_t H, int64_t W, int k1, int k2, int s1, int s2, int p1, int p2, int d1,
                                      int d2) {
    const ConvPlan c = conv_plan(b, C, H, W, 64, k1, k2, s1, s2, p1, p2, d1, d2);
    // the covariance sums over REAL output positions only: the grid must already be whole 16-byte chunks / k-steps
    if (c.O1 <= 0 || c.O2 <= 0 || c.O1p != c.O1 || c.O2p != c.O2 || c.Cp > 2 * C + 8) return -1;
    return align256(c.copies_bytes) + cov_stage_bytes(c.Ipp);
}

int kf_conv2d_cov_accum(float* Cov, int64_t ldc, const void* x, int64_t b, int64_t C, int64_t H, int64_t W, int k1, int k2, int s1, int s2,
                        int p1, int p2, int d1, int d2, float alpha, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!Cov || !x || b < 0 || C <= 0) return KF_ERR_INVALID_ARGUMENT;
    const int64_t need = kf_conv2d_cov_workspace_bytes(b, C, H, W, k1, k2, s1, s2, p1, p2, d1, d2);
    if (need < 0 || (reinterpret_cast<uintptr_t>(x) & 15) != 0 || b > 65535) return KF_ERR_INVALID_ARGUMENT;
    if (!workspace || workspace_bytes < need) return KF_ERR_WORKSPACE_TOO_SMALL;
    if (b == 0) return KF_OK;
    if (configure_once() != KF_OK) return KF_ERR_LAUNCH_FAILED;
    hipStream_t st = as_stream(stream);
    const ConvPlan p = conv_plan(b, C, H, W, 64, k1, k2, s1, s2, p1, p2, d1, d2);
    uint16_t* copies = reinterpret_cast<uint16_t*>(workspace);
    PadArgs pa;
    pa.out = copies; pa.x = reinterpret_cast<const uint16_t*>(x); pa.planes = b * p.Cp; pa.C = static_cast<int>(C); pa.Cp = static_cast<int>(p.Cp);
    pa.H = static_cast<int>(H); pa.W = static_cast<int>(W); pa.Hp = static_cast<int>(p.Hp); pa.Wq = static_cast<int>(p.Wq);
    pa.p1 = p1; pa.p2 = p2; pa.s2 = s2;
    const int64_t chunks = s2 * b * p.Cp * p.Hp * (p.Wq / 8);
    hipLaunchKernelGGL(conv_pad_phases_kernel, dim3(static_cast<unsigned>(std::min<int64_t>(cdiv(chunks, 256), 1 << 20))), dim3(256), 0,
                       st, pa);
    CovV2Args c{};
    c.stage = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(workspace) + align256(p.copies_bytes));
    c.X = copies; c.sample_stride = p.Cp * p.Hp * p.Wq;
    c.N = static_cast<int>(p.Ipp); c.K = static_cast<int>(p.Pp); c.batch = static_cast<int>(b);
    c.conv = 1; c.Cp = static_cast<int>(p.Cp); c.k2 = k2; c.O2 = static_cast<int>(p.O2p);
    c.s1 = s1; c.d1 = d1; c.s2 = s2; c.d2 = d2; c.Wq = static_cast<int>(p.Wq); c.plane = static_cast<int>(p.Hp * p.Wq);
    c.phase_stride = b * p.Cp * p.Hp * p.Wq;
    CovFinalizeArgs f{};
    f.out = Cov; f.ldc = ldc; f.d = static_cast<int>(C * k1 * k2); f.conv = 1; f.Cp = static_cast<int>(p.Cp); f.taps = k1 * k2; f.alpha = alpha;
    return launch_cov_v2(c, f, st);
}

}  // extern "C"
