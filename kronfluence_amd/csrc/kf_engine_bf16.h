// kf_engine_bf16.h -- bf16 MFMA tile engine (fp32 accumulation) for the two operand layouts of the
// EK-FAC hot path:
//   TRANS = false  ("NT"): both operands K-contiguous, C[m,n] += sum_k A[m,k] B[n,k]
//       the pairwise-score contraction: A = preconditioned query gradients [Q, O*I'],
//       B = per-sample train gradients [b, O*I'] (reference module/conv2d.py:199-209,
//       module/tracker/pairwise_score.py:41-45);
//   TRANS = true   ("TN"): both operands K-strided (rows contiguous), batched,
//       C[z][m,n] = sum_k A[z][k,m] B[z][k,n]
//       the per-sample gradient g[b] = G[b]^T A'[b] of module/linear.py:72 / module/conv2d.py:176
//       (k = token / output position).
//
// 256 threads = 4 wave64 (2x2), 128x128 output tile, k-step 64; each wave owns 64x64 as 2x2
// v_mfma_f32_32x32x16_bf16 accumulators.  Both layouts are staged global -> VGPR -> LDS into the same
// image: [row][64 k] bf16 with a 144-B row pitch, so the ds_read_b128 fragment reads (lane l: row
// l&31, k-octet l>>5) are bank-conflict free.  NT rows are copied with 16-B loads (8 lanes cover one
// 128-B row segment).  TN tiles are transposed in registers: a thread loads an 8(k) x 8(row) block
// with eight 16-B loads (8 lanes x 16 B = 128 contiguous bytes of one k-row), transposes it with
// 32 byte-permutes and writes eight 16-B k-octets.  The next k-tile's loads are issued before the
// MFMA block and written to the other LDS buffer after it (one barrier per k-step).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kf_engine.h"

namespace kf {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int HBK = 64;                 // k-step in bf16 elements (128 B per row)
constexpr int HPITCH = 144;             // LDS row pitch in bytes (128 + 16)
constexpr int HTILE_BYTES = 128 * HPITCH;
constexpr int HSMEM_BYTES = 2 * 2 * HTILE_BYTES;  // double-buffered A and B tiles: 73,728 B

struct HalfOperand {
    const uint16_t* p;
    int64_t batch_stride;   // elements
    int64_t ld;             // NT: row stride; TN: k stride (the other stride is 1); multiple of 8
    int64_t kt_stride;      // NT only: elements between consecutive 64-wide k-tiles of one row (64 = plain
                            // row-major; rows*64 = k-tile-major "[K/64][rows][64]" layout)
    int rows, depth;        // NT: depth % 8 == 0; TN: rows % 8 == 0
};

struct HalfGemmArgs {
    void* C; int c_dtype;   // F32 (store / accumulate / atomic) or BF16 (plain store)
    int64_t ldc, c_batch_stride;
    HalfOperand A, B;
    int M, N, K;
    int ksplit, kchunk;     // kchunk multiple of HBK
    float alpha, beta;
    int atomic;
    int tiles_m, tiles_n, chunks;   // XCD-aware 1-D grid: chunks = batch * ksplit
    int symmetric;                  // TN only, A == B: upper-triangular tile pairs, both triangles written (SYRK)
    int64_t c_tile_stride;          // != 0: C[z] is one ROW (index z) of a k-tile-major matrix whose k index is
                                    // d = m*ldc + n:  element at (d/64)*c_tile_stride + z*64 + d%64
    // epilogue extras (all nullable): v = alpha * acc + row_add[n] (n < row_add_n: the bias row of an eigenvector
    // matrix, "[A, 1] Q = A Q[:I] + Q[I]"), then v *= mul[m * ld_mul + n] for n < mul_n and v = 0 for n >= mul_n
    // (the EK-FAC Lambda^-1 on a column-padded output)
    const float* row_add; int row_add_n;
    const float* mul; int64_t ld_mul; int mul_n;
};

__device__ __forceinline__ float bf16_epilogue(const HalfGemmArgs& a, float acc, int m, int n) {
    float v = a.alpha * acc;
    if (m >= a.M || n >= a.N) return v;  // tile padding: never stored, and the side inputs must not be read there
    if (a.row_add && n < a.row_add_n) v += a.row_add[n];
    if (a.mul) v = n < a.mul_n ? v * a.mul[static_cast<int64_t>(m) * a.ld_mul + n] : 0.0f;
    return v;
}

// Native 4-dword vector (not HIP's uint4 struct): keeps the staging registers in VGPRs -- with the
// struct type hipcc demoted the arrays below to scratch memory (measured: 8 % of the bf16 peak).
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u32x4 ld16(const uint16_t* p) { return *reinterpret_cast<const u32x4*>(p); }

__device__ __forceinline__ void bf16_store(void* p, int64_t idx, float v) {
    uint32_t u = __float_as_uint(v);
    if ((u & 0x7fffffffu) > 0x7f800000u) u |= 0x00400000u;
    else u += 0x7fffu + ((u >> 16) & 1u);
    reinterpret_cast<uint16_t*>(p)[idx] = static_cast<uint16_t>(u >> 16);
}

// rows r[0..7] = 8 k-rows x 8 halves  ->  r[i] = 8 k-values of column i (in place)
#define KF_T8_WORD(i, w) (((i) & 1) ? ((r##w##lo[(i) >> 1] >> 16) | (r##w##hi[(i) >> 1] & 0xffff0000u)) \
                                   : ((r##w##lo[(i) >> 1] & 0xffffu) | (r##w##hi[(i) >> 1] << 16)))
#define KF_T8_ROW(i) u32x4{KF_T8_WORD(i, 0), KF_T8_WORD(i, 1), KF_T8_WORD(i, 2), KF_T8_WORD(i, 3)}
__device__ __forceinline__ void transpose8x8_b16(u32x4& r0, u32x4& r1, u32x4& r2, u32x4& r3, u32x4& r4, u32x4& r5,
                                                 u32x4& r6, u32x4& r7) {
    const u32x4 r0lo = r0, r0hi = r1, r1lo = r2, r1hi = r3, r2lo = r4, r2hi = r5, r3lo = r6, r3hi = r7;
    r0 = KF_T8_ROW(0); r1 = KF_T8_ROW(1); r2 = KF_T8_ROW(2); r3 = KF_T8_ROW(3);
    r4 = KF_T8_ROW(4); r5 = KF_T8_ROW(5); r6 = KF_T8_ROW(6); r7 = KF_T8_ROW(7);
}
#undef KF_T8_ROW
#undef KF_T8_WORD

// acc += (tile m0,n0 of) A B^T over k in [k_begin, k_end); all 256 threads; hsm: HSMEM_BYTES of LDS.
template <bool TRANS>
__device__ __forceinline__ void bf16_tile_mainloop(const HalfGemmArgs& a, const uint16_t* Ab, const uint16_t* Bb, int m0, int n0,
                                                   int k_begin, int k_end, f32x16 (&acc)[2][2], unsigned char* hsm) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const u32x4 zero = {0u, 0u, 0u, 0u};

    // Eight named staging registers (no arrays, no lambdas: everything stays in VGPRs).
    //   NT: q0..q3 = A rows r0+32j, q4..q7 = B rows r0+32j (octet `oct` of the k-tile)
    //   TN: q0..q7 = k-rows ko*8+j of this thread's 8-row-wide column block of A (waves 0,1) or B (2,3)
    u32x4 q0, q1, q2, q3, q4, q5, q6, q7;
    const int oct = tid & 7, r0 = tid >> 3;
    const int half = tid >> 7, t7 = tid & 127, ko = t7 & 7, mo = t7 >> 3;
    const HalfOperand& op = half ? a.B : a.A;
    const uint16_t* opb = half ? Bb : Ab;
    const int row_base = (half ? n0 : m0) + mo * 8;
    const bool rows_ok = row_base < op.rows;   // rows % 8 == 0: an octet is entirely in or out
    const int rb8 = rows_ok ? row_base : 0;

#define KF_NT_LOAD(q, base, row, lim, op_) \
    q = ld16((base) + static_cast<int64_t>((row) < (lim) ? (row) : (lim) - 1) * (op_).ld + static_cast<int64_t>(kc >> 6) * (op_).kt_stride + (kc & 63))
#define KF_TN_LOAD(q, j) q = ld16(opb + static_cast<int64_t>(min(kt_ + ko * 8 + (j), k_end - 1)) * op.ld + rb8)
#define KF_FETCH(KT)                                                                                        \
    do {                                                                                                    \
        const int kt_ = (KT);                                                                               \
        if constexpr (TRANS) {                                                                              \
            KF_TN_LOAD(q0, 0); KF_TN_LOAD(q1, 1); KF_TN_LOAD(q2, 2); KF_TN_LOAD(q3, 3);                     \
            KF_TN_LOAD(q4, 4); KF_TN_LOAD(q5, 5); KF_TN_LOAD(q6, 6); KF_TN_LOAD(q7, 7);                     \
        } else {                                                                                            \
            const int k_ = kt_ + oct * 8;                                                                   \
            const int kc = k_ < k_end ? k_ : k_end - 8; /* in bounds; value discarded in the stash */       \
            KF_NT_LOAD(q0, Ab, m0 + r0, a.M, a.A); KF_NT_LOAD(q1, Ab, m0 + r0 + 32, a.M, a.A);              \
            KF_NT_LOAD(q2, Ab, m0 + r0 + 64, a.M, a.A); KF_NT_LOAD(q3, Ab, m0 + r0 + 96, a.M, a.A);         \
            KF_NT_LOAD(q4, Bb, n0 + r0, a.N, a.B); KF_NT_LOAD(q5, Bb, n0 + r0 + 32, a.N, a.B);              \
            KF_NT_LOAD(q6, Bb, n0 + r0 + 64, a.N, a.B); KF_NT_LOAD(q7, Bb, n0 + r0 + 96, a.N, a.B);         \
        }                                                                                                   \
    } while (0)
#define KF_TN_ZERO(q, j) q = (rows_ok && kt_ + ko * 8 + (j) < k_end) ? q : zero
#define KF_ST(ptr, v) *reinterpret_cast<u32x4*>(ptr) = (v)
#define KF_STASH(BUF, KT)                                                                                   \
    do {                                                                                                    \
        const int kt_ = (KT);                                                                               \
        unsigned char* sa_ = hsm + (BUF) * 2 * HTILE_BYTES;                                                 \
        unsigned char* sb_ = sa_ + HTILE_BYTES;                                                             \
        if constexpr (TRANS) {                                                                              \
            KF_TN_ZERO(q0, 0); KF_TN_ZERO(q1, 1); KF_TN_ZERO(q2, 2); KF_TN_ZERO(q3, 3);                     \
            KF_TN_ZERO(q4, 4); KF_TN_ZERO(q5, 5); KF_TN_ZERO(q6, 6); KF_TN_ZERO(q7, 7);                     \
            transpose8x8_b16(q0, q1, q2, q3, q4, q5, q6, q7);                                               \
            unsigned char* d_ = (half ? sb_ : sa_) + (mo * 8) * HPITCH + ko * 16;                           \
            KF_ST(d_, q0); KF_ST(d_ + HPITCH, q1); KF_ST(d_ + 2 * HPITCH, q2); KF_ST(d_ + 3 * HPITCH, q3);  \
            KF_ST(d_ + 4 * HPITCH, q4); KF_ST(d_ + 5 * HPITCH, q5); KF_ST(d_ + 6 * HPITCH, q6);             \
            KF_ST(d_ + 7 * HPITCH, q7);                                                                     \
        } else {                                                                                            \
            const bool kok_ = kt_ + oct * 8 < k_end;                                                        \
            const int o_ = r0 * HPITCH + oct * 16;                                                          \
            KF_ST(sa_ + o_, (kok_ && m0 + r0 < a.M) ? q0 : zero);                                           \
            KF_ST(sa_ + o_ + 32 * HPITCH, (kok_ && m0 + r0 + 32 < a.M) ? q1 : zero);                        \
            KF_ST(sa_ + o_ + 64 * HPITCH, (kok_ && m0 + r0 + 64 < a.M) ? q2 : zero);                        \
            KF_ST(sa_ + o_ + 96 * HPITCH, (kok_ && m0 + r0 + 96 < a.M) ? q3 : zero);                        \
            KF_ST(sb_ + o_, (kok_ && n0 + r0 < a.N) ? q4 : zero);                                           \
            KF_ST(sb_ + o_ + 32 * HPITCH, (kok_ && n0 + r0 + 32 < a.N) ? q5 : zero);                        \
            KF_ST(sb_ + o_ + 64 * HPITCH, (kok_ && n0 + r0 + 64 < a.N) ? q6 : zero);                        \
            KF_ST(sb_ + o_ + 96 * HPITCH, (kok_ && n0 + r0 + 96 < a.N) ? q7 : zero);                        \
        }                                                                                                   \
    } while (0)

    if (k_begin < k_end) {
        KF_FETCH(k_begin);
        KF_STASH(0, k_begin);
        __syncthreads();
        int buf = 0;
        for (int kt = k_begin; kt < k_end; kt += HBK) {
            const bool more = kt + HBK < k_end;
            if (more) KF_FETCH(kt + HBK);
            const unsigned char* sa = hsm + buf * 2 * HTILE_BYTES + (wm * 64 + (lane & 31)) * HPITCH + (lane >> 5) * 16;
            const unsigned char* sb = hsm + buf * 2 * HTILE_BYTES + HTILE_BYTES + (wn * 64 + (lane & 31)) * HPITCH + (lane >> 5) * 16;
#pragma unroll
            for (int kk = 0; kk < HBK / 16; ++kk) {
                const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(sa + kk * 32);
                const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(sa + 32 * HPITCH + kk * 32);
                const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(sb + kk * 32);
                const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(sb + 32 * HPITCH + kk * 32);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
            }
            if (more) KF_STASH(buf ^ 1, kt + HBK);
            __syncthreads();
            buf ^= 1;
        }
    }
#undef KF_STASH
#undef KF_ST
#undef KF_TN_ZERO
#undef KF_FETCH
#undef KF_TN_LOAD
#undef KF_NT_LOAD
}

template <bool TRANS, bool SYM = false>  // SYM: the symmetric (SYRK) instantiation -- its mirror epilogue costs registers
__global__ __launch_bounds__(NTHREADS) void gemm_bf16_kernel(HalfGemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char hsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    // XCD-aware block -> work mapping.  Workgroup L is dispatched to XCD L % 8 (observed, used for speed
    // only): all output tiles of one (batch, k-chunk) slab go to the SAME XCD in consecutive order, so
    // the 8 m-tiles x 8 n-tiles that re-read the same A / B k-range hit in that XCD's 4 MB L2 instead
    // of each XCD pulling its own copy over the fabric.
    const int tiles = SYM ? a.tiles_m * (a.tiles_m + 1) / 2 : a.tiles_m * a.tiles_n;
    // work items (chunk-major, tile-minor) are cut into 8 contiguous ranges, one per XCD: whole k-chunks
    // when there are many, runs of neighbouring tiles (same A rows) when there are few.
    const int64_t items = static_cast<int64_t>(a.chunks) * tiles, per_xcd = (items + 7) / 8;
    const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
    const int64_t item = static_cast<int64_t>(xcd) * per_xcd + j;
    if (j >= per_xcd || item >= items) return;
    const int chunk = static_cast<int>(item / tiles);
    const int tile = static_cast<int>(item % tiles);
    int tile_i = tile / a.tiles_n, tile_j = tile % a.tiles_n;
    if (SYM) {  // tile -> (tile_i <= tile_j) of the upper triangle
        int t = tile;
        tile_i = 0;
        while (t >= a.tiles_m - tile_i) { t -= a.tiles_m - tile_i; ++tile_i; }
        tile_j = tile_i + t;
    }
    const int n0 = tile_j * 128, m0 = tile_i * 128;
    const bool mirror = SYM && tile_i != tile_j;
    const int z = chunk / a.ksplit, ks = chunk % a.ksplit;
    const int k_begin = ks * a.kchunk;
    const int k_end = min(a.K, k_begin + a.kchunk);
    const uint16_t* Ab = a.A.p + static_cast<int64_t>(z) * a.A.batch_stride;
    const uint16_t* Bb = a.B.p + static_cast<int64_t>(z) * a.B.batch_stride;
    f32x16 acc[2][2];
    zero_acc(acc);
    bf16_tile_mainloop<TRANS>(a, Ab, Bb, m0, n0, k_begin, k_end, acc, hsm);
    const int64_t cz = static_cast<int64_t>(z) * a.c_batch_stride;
    if (a.c_dtype == BF16) {
        // bf16 output: convert in registers, transpose through LDS (pitch 272 B) and write 16 B per lane --
        // the direct fragment layout would issue 64 two-byte stores per lane (measured store-bound).
        constexpr int OP = 272;
        __syncthreads();  // the mainloop's LDS buffers are free
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ml = acc_row(wm, ti, r, lane), nl = acc_col(wn, tj, lane);
                    uint32_t u = __float_as_uint(bf16_epilogue(a, acc[ti][tj][r], m0 + ml, n0 + nl));
                    if ((u & 0x7fffffffu) > 0x7f800000u) u |= 0x00400000u;
                    else u += 0x7fffu + ((u >> 16) & 1u);
                    *reinterpret_cast<uint16_t*>(hsm + ml * OP + nl * 2) = static_cast<uint16_t>(u >> 16);
                }
        __syncthreads();
        uint16_t* Cb = reinterpret_cast<uint16_t*>(a.C);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int id = tid + 256 * it, ml = id >> 4, ch = id & 15;
            const int m = m0 + ml, n = n0 + ch * 8;
            if (m < a.M && n < a.N) {
                const int64_t d = static_cast<int64_t>(m) * a.ldc + n;
                const int64_t idx = a.c_tile_stride ? (d >> 6) * a.c_tile_stride + static_cast<int64_t>(z) * 64 + (d & 63) : cz + d;
                const unsigned char* src = hsm + ml * OP + ch * 16;
                if (n + 8 <= a.N && (idx & 7) == 0) {
                    *reinterpret_cast<u32x4*>(Cb + idx) = *reinterpret_cast<const u32x4*>(src);
                } else {
                    for (int e = 0; e < 8 && n + e < a.N; ++e) {
                        const int64_t de = d + e;
                        const int64_t ie = a.c_tile_stride ? (de >> 6) * a.c_tile_stride + static_cast<int64_t>(z) * 64 + (de & 63) : cz + de;
                        Cb[ie] = reinterpret_cast<const uint16_t*>(src)[e];
                    }
                }
            }
        }
        return;
    }
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + acc_row(wm, ti, r, lane), n = n0 + acc_col(wn, tj, lane);
                if (m < a.M && n < a.N) {
                    const int64_t idx = cz + static_cast<int64_t>(m) * a.ldc + n;
                    const float v = bf16_epilogue(a, acc[ti][tj][r], m, n);
                    float* dst = reinterpret_cast<float*>(a.C) + idx;
                    if (a.atomic) atomicAdd(dst, v);
                    else *dst = (a.beta == 0.0f) ? v : v + a.beta * *dst;
                }
            }
    if constexpr (SYM) if (mirror)  // off-diagonal tile pair (uniform per workgroup): the lower triangle, coalesced
        mirror_through_lds(
            reinterpret_cast<float*>(hsm), wm, wn, lane, wave,
            [&](int ti, int tj, int r) { return bf16_epilogue(a, acc[ti][tj][r], m0 + acc_row(wm, ti, r, lane), n0 + acc_col(wn, tj, lane)); },
            [&](int nl, int ml, float v) {
                const int n = n0 + nl, m = m0 + ml;
                if (m < a.M && n < a.N) {
                    float* lo = reinterpret_cast<float*>(a.C) + cz + static_cast<int64_t>(n) * a.ldc + m;
                    if (a.atomic) atomicAdd(lo, v);
                    else *lo = (a.beta == 0.0f) ? v : v + a.beta * *lo;
                }
            });
}

// Lambda += scale2 * sum_z ( sum_k A[z][k,m] B[z][k,n] )^2 -- the EK-FAC corrected eigenvalues from bf16
// rotated gradient factors (module/tracker/factor.py:218-226): TN tiles per sample, squared and summed in
// registers over this workgroup's sample range, one fp32 atomic per output element at the end.
struct HalfLambdaArgs {
    HalfGemmArgs g;      // operands / shapes (C = Lambda fp32, ldc); batch handled by the z loop below
    int batch, zchunk, zblocks;
    float scale2;
};

__global__ __launch_bounds__(NTHREADS) void lambda_bf16_kernel(HalfLambdaArgs la) {
    extern __shared__ __attribute__((aligned(16))) unsigned char hsm[];
    const HalfGemmArgs& a = la.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    // XCD-aware 1-D grid (workgroup L runs on XCD L % 8): items are (sample range, tile), sample range major, cut into 8
    // contiguous runs -- the tiles of a sample range re-read the same rotated factors Gt[z] / At[z], so they run on ONE XCD and
    // share its L2 (with the round-2 3-D grid consecutive tiles went to different XCDs and every XCD fetched every sample:
    // 2.4x the algorithmic bytes, profiles/r02_pmc_resnet9.json)
    const int tiles = a.tiles_m * a.tiles_n;
    const int64_t items = static_cast<int64_t>(la.zblocks) * tiles, per_xcd = (items + 7) / 8;
    const int block = blockIdx.x, xcd = block & 7, jx = block >> 3;
    const int64_t item = static_cast<int64_t>(xcd) * per_xcd + jx;
    if (jx >= per_xcd || item >= items) return;
    const int zb = static_cast<int>(item / tiles), tile = static_cast<int>(item % tiles);
    const int n0 = (tile % a.tiles_n) * 128, m0 = (tile / a.tiles_n) * 128;
    const int z_begin = zb * la.zchunk, z_end = min(la.batch, z_begin + la.zchunk);
    f32x16 sq[2][2];
    zero_acc(sq);
    for (int z = z_begin; z < z_end; ++z) {
        f32x16 acc[2][2];
        zero_acc(acc);
        bf16_tile_mainloop<true>(a, a.A.p + static_cast<int64_t>(z) * a.A.batch_stride, a.B.p + static_cast<int64_t>(z) * a.B.batch_stride,
                                 m0, n0, 0, a.K, acc, hsm);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) sq[i][j][r] = fmaf(acc[i][j][r], acc[i][j][r], sq[i][j][r]);
    }
    float* L = reinterpret_cast<float*>(a.C);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + acc_row(wm, i, r, lane), n = n0 + acc_col(wn, j, lane);
                if (m < a.M && n < a.N) atomicAdd(L + static_cast<int64_t>(m) * a.ldc + n, la.scale2 * sq[i][j][r]);
            }
}

}  // namespace kf
