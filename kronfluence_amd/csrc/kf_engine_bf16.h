// kf_engine_bf16.h -- bf16 MFMA tile engine for "NT" contractions with both operands K-contiguous:
//   C[m, n] (+)= alpha * sum_k A[m, k] * B[n, k],   A: [M, K] bf16, B: [N, K] bf16, fp32 accumulate.
// This is the shape of the pairwise-score GEMM for sequence / conv layers: A = preconditioned query
// gradients [Q, O*I'], B = per-sample train gradients [b, O*I'] (reference module/conv2d.py:199-209,
// module/tracker/pairwise_score.py:41-45).
//
// 256 threads = 4 wave64 (2x2), 128x128 output tile, k-step 64; each wave owns 64x64 as 2x2
// v_mfma_f32_32x32x16_bf16 accumulators.  Rows are staged global -> VGPR (16-B loads, 8 lanes cover
// one 128-B row segment) -> LDS with a 144-B row pitch, so the ds_read_b128 fragment reads (lane l:
// row l&31, k-octet l>>5) are bank-conflict free; the next k-tile's loads are issued before the MFMA
// block and written to the other LDS buffer after it (one barrier per k-step).
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

struct HalfGemmArgs {
    float* C; int64_t ldc;
    const uint16_t* A; const uint16_t* B;
    int64_t lda, ldb;       // row strides in elements (multiples of 8)
    int M, N, K;            // K multiple of 8
    int kchunk;             // multiple of HBK
    float alpha;
    int atomic;             // 1: atomicAdd into C; 0: C = alpha*acc + beta*C
    float beta;
};

__device__ __forceinline__ uint4 ld16(const uint16_t* p) { return *reinterpret_cast<const uint4*>(p); }

__global__ __launch_bounds__(NTHREADS) void gemm_nt_bf16_kernel(HalfGemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char hsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int n0 = blockIdx.x * 128, m0 = blockIdx.y * 128;
    const int k_begin = blockIdx.z * a.kchunk;
    const int k_end = min(a.K, k_begin + a.kchunk);

    // staging map: 8 consecutive threads cover one row's 128-B k-segment; 4 row groups per operand
    const int oct = tid & 7, r0 = tid >> 3;  // r0 in [0,32)
    const uint16_t* ap[4];
    const uint16_t* bp[4];
    bool aok[4], bok[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ra = m0 + r0 + 32 * j, rb = n0 + r0 + 32 * j;
        aok[j] = ra < a.M; bok[j] = rb < a.N;
        ap[j] = a.A + static_cast<int64_t>(aok[j] ? ra : a.M - 1) * a.lda + oct * 8;
        bp[j] = a.B + static_cast<int64_t>(bok[j] ? rb : a.N - 1) * a.ldb + oct * 8;
    }
    uint4 ra[4], rb[4];
    const uint4 zero = make_uint4(0, 0, 0, 0);
    auto fetch = [&](int kt) {
        const int k = kt + oct * 8;
        const bool kok = k < k_end;           // K % 8 == 0: an octet is entirely in or out
        const int kc = kok ? kt : k_end - 8 - oct * 8;  // row base + (k_end - 8): in bounds, value discarded
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ra[j] = ld16(ap[j] + kc);
            rb[j] = ld16(bp[j] + kc);
        }
        (void)kok;
    };
    auto stash = [&](int buf, int kt) {
        const bool kok = kt + oct * 8 < k_end;
        unsigned char* sa = hsm + buf * 2 * HTILE_BYTES;
        unsigned char* sb = sa + HTILE_BYTES;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int off = (r0 + 32 * j) * HPITCH + oct * 16;
            *reinterpret_cast<uint4*>(sa + off) = (aok[j] && kok) ? ra[j] : zero;
            *reinterpret_cast<uint4*>(sb + off) = (bok[j] && kok) ? rb[j] : zero;
        }
    };

    f32x16 acc[2][2];
    zero_acc(acc);
    if (k_begin < k_end) {
        fetch(k_begin);
        stash(0, k_begin);
        __syncthreads();
        int buf = 0;
        for (int kt = k_begin; kt < k_end; kt += HBK) {
            const bool more = kt + HBK < k_end;
            if (more) fetch(kt + HBK);
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
            if (more) stash(buf ^ 1, kt + HBK);
            __syncthreads();
            buf ^= 1;
        }
    }
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + acc_row(wm, ti, r, lane), n = n0 + acc_col(wn, tj, lane);
                if (m < a.M && n < a.N) {
                    float* dst = a.C + static_cast<int64_t>(m) * a.ldc + n;
                    const float v = a.alpha * acc[ti][tj][r];
                    if (a.atomic) atomicAdd(dst, v);
                    else *dst = (a.beta == 0.0f) ? v : v + a.beta * *dst;
                }
            }
}

}  // namespace kf
