// kf_engine.h -- fp32 MFMA tile engine shared by every dense kernel of the EK-FAC hot path.
//
// One workgroup = 256 threads = 4 wave64s laid out 2x2; it owns a 128x128 output tile and each
// wave a 64x64 quadrant built from 2x2 v_mfma_f32_32x32x2_f32 accumulators (exact fp32, 64
// FLOP/clk/SIMD -- cdna_hip_programming.md section 3).  Operands are pulled from HBM through
// small "loader" functors (strided views with fused bias column / mask / square / dtype
// conversion) with lane-contiguous addresses, staged k-major in LDS ([BK][128+4] floats, double
// buffered, one barrier per k-step) and read back as one conflict-free ds_read_b32 per MFMA
// operand: lane l reads row (l>>5) of the k-pair, column (l&31).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kf {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128;        // tile rows (A operand rows)
constexpr int BN = 128;        // tile cols (B operand rows)
constexpr int BK = 16;         // k-step
constexpr int LDT = BM + 4;    // LDS row length (floats); +4 keeps 16-B alignment, breaks bank stride
constexpr int NTHREADS = 256;
constexpr int FETCH = BM * BK / NTHREADS;  // 8 elements per thread per operand per k-step
constexpr int SMEM_FLOATS = 2 * 2 * BK * LDT;

enum DType { F32 = 0, BF16 = 1, F16 = 2, F64 = 3, I64 = 4, I32 = 5, U8 = 6 };

__device__ __forceinline__ float load_f32(const void* p, int dtype, int64_t idx) {
    switch (dtype) {
        case F32: return reinterpret_cast<const float*>(p)[idx];
        case BF16: return __uint_as_float(static_cast<uint32_t>(reinterpret_cast<const uint16_t*>(p)[idx]) << 16);
        case F16: return static_cast<float>(reinterpret_cast<const _Float16*>(p)[idx]);
        case F64: return static_cast<float>(reinterpret_cast<const double*>(p)[idx]);
        case I64: return static_cast<float>(reinterpret_cast<const int64_t*>(p)[idx]);
        case I32: return static_cast<float>(reinterpret_cast<const int32_t*>(p)[idx]);
        default: return static_cast<float>(reinterpret_cast<const uint8_t*>(p)[idx]);
    }
}

// Strided 2-D operand view for one batch index: element (r, k).
struct StridedLoader {
    const void* p;
    int dtype;
    int64_t row_stride, k_stride;
    int rows, depth;       // real extents; indices beyond (plus the optional ones) read 0
    int ones_row, ones_k;  // virtual index rows / depth reads 1.0 (un-masked bias column)
    int square;
    int contig_k;          // memory is contiguous along k (else along rows): picks the lane mapping

    __device__ __forceinline__ float get(int r, int k) const {
        const bool r_real = r < rows, k_real = k < depth;
        if (r_real && k_real) {
            const float x = load_f32(p, dtype, static_cast<int64_t>(r) * row_stride + static_cast<int64_t>(k) * k_stride);
            return square ? x * x : x;
        }
        const bool r_one = ones_row && r == rows, k_one = ones_k && k == depth;
        if ((r_one && (k_real || k_one)) || (k_one && r_real)) return 1.0f;
        return 0.0f;
    }
};

// Accumulator coordinates inside the 128x128 tile (cdna_hip_programming.md section 3, C/D map of
// the 32x32 shapes: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)).
__device__ __forceinline__ int acc_row(int wm, int ti, int reg, int lane) {
    return wm * 64 + ti * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
}
__device__ __forceinline__ int acc_col(int wn, int tj, int lane) { return wn * 64 + tj * 32 + (lane & 31); }

__device__ __forceinline__ void zero_acc(f32x16 (&acc)[2][2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
}

// acc[ti][tj] += sum_{k in [k_begin, k_end)} A(m, k) * B(n, k) for this wave's quadrant of the tile.
// Loader rows are tile-relative (the caller bakes m0/n0 and the batch offset into the loader).
// Must be called by all 256 threads; `smem` holds SMEM_FLOATS floats.
template <class LA, class LB>
__device__ __forceinline__ void mainloop(const LA& la, const LB& lb, int k_begin, int k_end,
                                         f32x16 (&acc)[2][2], float* smem) {
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    float* sA = smem;                  // [2][BK][LDT]
    float* sB = smem + 2 * BK * LDT;   // [2][BK][LDT]

    // lane -> (row, k) assignment of the FETCH elements this thread stages per operand.
    int ar[FETCH], ak[FETCH], br[FETCH], bk[FETCH];
#pragma unroll
    for (int j = 0; j < FETCH; ++j) {
        if (la.contig_k) { ak[j] = tid % BK; ar[j] = tid / BK + (NTHREADS / BK) * j; }
        else             { ar[j] = tid % BM; ak[j] = tid / BM + (NTHREADS / BM) * j; }
        if (lb.contig_k) { bk[j] = tid % BK; br[j] = tid / BK + (NTHREADS / BK) * j; }
        else             { br[j] = tid % BN; bk[j] = tid / BN + (NTHREADS / BN) * j; }
    }
    float ra[FETCH], rb[FETCH];
    auto fetch = [&](int kt) {
#pragma unroll
        for (int j = 0; j < FETCH; ++j) {
            const int ka = kt + ak[j], kb = kt + bk[j];
            ra[j] = ka < k_end ? la.get(ar[j], ka) : 0.0f;
            rb[j] = kb < k_end ? lb.get(br[j], kb) : 0.0f;
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int j = 0; j < FETCH; ++j) {
            sA[(buf * BK + ak[j]) * LDT + ar[j]] = ra[j];
            sB[(buf * BK + bk[j]) * LDT + br[j]] = rb[j];
        }
    };

    if (k_begin >= k_end) return;
    fetch(k_begin);
    stash(0);
    __syncthreads();
    int buf = 0;
    for (int kt = k_begin; kt < k_end; kt += BK) {
        const bool more = kt + BK < k_end;
        if (more) fetch(kt + BK);
        const float* a_base = sA + buf * BK * LDT + (lane >> 5) * LDT + wm * 64 + (lane & 31);
        const float* b_base = sB + buf * BK * LDT + (lane >> 5) * LDT + wn * 64 + (lane & 31);
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const float a0 = a_base[kk * 2 * LDT], a1 = a_base[kk * 2 * LDT + 32];
            const float b0 = b_base[kk * 2 * LDT], b1 = b_base[kk * 2 * LDT + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (more) stash(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
}

}  // namespace kf
