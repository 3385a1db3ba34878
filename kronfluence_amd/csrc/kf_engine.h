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

// Compile-time dtype loads: no branch in the instruction stream, so the FETCH loads of a k-step
// stay in one basic block and are issued back to back under a single s_waitcnt.
template <int DT> __device__ __forceinline__ float load_t(const void* p, int64_t idx);
template <> __device__ __forceinline__ float load_t<F32>(const void* p, int64_t idx) {
    return reinterpret_cast<const float*>(p)[idx];
}
template <> __device__ __forceinline__ float load_t<BF16>(const void* p, int64_t idx) {
    return __uint_as_float(static_cast<uint32_t>(reinterpret_cast<const uint16_t*>(p)[idx]) << 16);
}
template <> __device__ __forceinline__ float load_t<F16>(const void* p, int64_t idx) {
    return static_cast<float>(reinterpret_cast<const _Float16*>(p)[idx]);
}
template <> __device__ __forceinline__ float load_t<I64>(const void* p, int64_t idx) {
    return static_cast<float>(reinterpret_cast<const int64_t*>(p)[idx]);
}
template <> __device__ __forceinline__ float load_t<U8>(const void* p, int64_t idx) {
    return static_cast<float>(reinterpret_cast<const uint8_t*>(p)[idx]);
}

// Raw (unconverted) loads: the mainloop issues them at the top of a k-step and converts / masks
// them only when staging into LDS, after the MFMA block, so HBM/L2 latency hides under the MFMAs.
template <int DT> struct RawOf { typedef uint32_t type; };
template <> struct RawOf<BF16> { typedef uint16_t type; };
template <> struct RawOf<F16> { typedef uint16_t type; };
template <> struct RawOf<I64> { typedef int64_t type; };
template <> struct RawOf<U8> { typedef uint8_t type; };
template <int DT> __device__ __forceinline__ typename RawOf<DT>::type load_raw(const void* p, int64_t idx) {
    return reinterpret_cast<const typename RawOf<DT>::type*>(p)[idx];
}
template <int DT> __device__ __forceinline__ float raw_to_f32(typename RawOf<DT>::type v);
template <> __device__ __forceinline__ float raw_to_f32<F32>(uint32_t v) { return __uint_as_float(v); }
template <> __device__ __forceinline__ float raw_to_f32<BF16>(uint16_t v) { return __uint_as_float(static_cast<uint32_t>(v) << 16); }
template <> __device__ __forceinline__ float raw_to_f32<F16>(uint16_t v) {
    _Float16 h;
    __builtin_memcpy(&h, &v, 2);
    return static_cast<float>(h);
}
template <> __device__ __forceinline__ float raw_to_f32<I64>(int64_t v) { return static_cast<float>(v); }
template <> __device__ __forceinline__ float raw_to_f32<U8>(uint8_t v) { return static_cast<float>(v); }

// Strided 2-D operand view for one batch index: element (r, k).  Out-of-range indices are read
// from a clamped (always valid) address and replaced by a select -- never by a branch.
// Requires rows_total >= 1 and depth >= 1 (checked on the host); `rows` is tile-relative and may
// be 0 for the tile that holds only the virtual ones row (the clamp then reads row -1, which is
// the last real row of the matrix).
template <int DT>
struct StridedLoader {
    const void* p;
    int64_t row_stride, k_stride;
    int rows, depth;       // real extents; indices beyond (plus the optional ones) read 0
    int ones_row, ones_k;  // virtual index rows / depth reads 1.0 (un-masked bias column)
    int square;
    int contig_k;          // memory is contiguous along k (else along rows): picks the lane mapping

    typedef typename RawOf<DT>::type Raw;

    __device__ __forceinline__ Raw fetch(int r, int k) const {
        const int rc = r < rows ? r : rows - 1, kc = k < depth ? k : depth - 1;
        return load_raw<DT>(p, static_cast<int64_t>(rc) * row_stride + static_cast<int64_t>(kc) * k_stride);
    }
    __device__ __forceinline__ float value(Raw raw, int r, int k) const {
        const bool r_real = r < rows, k_real = k < depth;
        float x = raw_to_f32<DT>(raw);
        x = square ? x * x : x;
        const bool r_one = ones_row && r == rows, k_one = ones_k && k == depth;
        const bool one = (r_one && (k_real || k_one)) || (k_one && r_real);
        return (r_real && k_real) ? x : (one ? 1.0f : 0.0f);
    }
};

// Accumulator coordinates inside the 128x128 tile (cdna_hip_programming.md section 3, C/D map of
// the 32x32 shapes: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)).
__device__ __forceinline__ int acc_row(int wm, int ti, int reg, int lane) {
    return wm * 64 + ti * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
}
__device__ __forceinline__ int acc_col(int wn, int tj, int lane) { return wn * 64 + tj * 32 + (lane & 31); }

// Lower-triangle mirror of a symmetric product: the wave's 2 x 2 accumulator blocks of the (ti < tj) tile go to
// C[n0 + n][m0 + m] TRANSPOSED THROUGH LDS, so that lanes run along the contiguous direction of C (two 128-byte segments per
// wave instruction).  Mirroring straight from the fragment layout puts the 64 lanes of an instruction on 64 different rows:
// scattered fp32 atomics that were measured at 2-3x the cost of the whole k-loop of a covariance tile.
// All 256 threads call it; `lds` holds 4 x 32 x 33 floats (free after the main loop); v(i, j, r) = finished value of acc[i][j][r].
template <typename Value, typename Store>
__device__ __forceinline__ void mirror_through_lds(float* lds, int wm, int wn, int lane, int wave, Value value, Store store) {
    float* mine = lds + wave * (32 * 33);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            __syncthreads();  // the region is free (main loop / previous block done)
#pragma unroll
            for (int r = 0; r < 16; ++r) mine[(lane & 31) * 33 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)] = value(i, j, r);
            __syncthreads();
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int nl = rr * 2 + (lane >> 5), ml = lane & 31;
                store(wn * 64 + j * 32 + nl, wm * 64 + i * 32 + ml, mine[nl * 33 + ml]);  // (n in tile, m in tile, value)
            }
        }
}

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
    typename LA::Raw ra[FETCH];
    typename LB::Raw rb[FETCH];
    // fetch: raw loads only (clamped addresses, no branch, no consumer) -- issued back to back
    auto fetch = [&](int kt) {
#pragma unroll
        for (int j = 0; j < FETCH; ++j) {
            ra[j] = la.fetch(ar[j], min(kt + ak[j], k_end - 1));
            rb[j] = lb.fetch(br[j], min(kt + bk[j], k_end - 1));
        }
    };
    // stash: convert / mask / zero-fill the raw values of k-tile `kt` and stage them in LDS
    auto stash = [&](int buf, int kt) {
#pragma unroll
        for (int j = 0; j < FETCH; ++j) {
            const int ka = kt + ak[j], kb = kt + bk[j];
            const float va = la.value(ra[j], ar[j], min(ka, k_end - 1));
            const float vb = lb.value(rb[j], br[j], min(kb, k_end - 1));
            sA[(buf * BK + ak[j]) * LDT + ar[j]] = ka < k_end ? va : 0.0f;
            sB[(buf * BK + bk[j]) * LDT + br[j]] = kb < k_end ? vb : 0.0f;
        }
    };

    if (k_begin >= k_end) return;
    fetch(k_begin);
    stash(0, k_begin);
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
        if (more) stash(buf ^ 1, kt + BK);
        __syncthreads();
        buf ^= 1;
    }
}

}  // namespace kf
