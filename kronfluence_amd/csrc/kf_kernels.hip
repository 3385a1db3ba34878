// kf_kernels.hip -- gfx950 kernels and C ABI of the EK-FAC hot path (see include/kronfluence_hip.h).
//
// Dense contractions (covariance SYRK, eigenbasis rotations, Lambda, preconditioner, per-sample
// gradient, pairwise-score GEMM) run on the fp32 MFMA tile engine of kf_engine.h; elementwise and
// reduction stages (im2col, 1/(Lambda/n+damping), casts) are coalesced streaming kernels.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <algorithm>
#include <mutex>

#include "../../include/kronfluence_hip.h"
#include "kf_engine.h"
#include "kf_engine_bf16.h"

using namespace kf;

namespace kf {
int score_gemm_tiled(float* scores, int64_t ld, const void* P, const void* psg, int64_t Q, int64_t b, int64_t D, float scale, void* stream);
int rotate_gemm_v2(void* C, int64_t ldc, const void* A, int64_t lda, const void* B, int64_t ldb, int64_t M, int64_t N, int64_t K,
                   float alpha, const float* row_add, int row_add_n, void* stream);
// kf_precondition's bf16 path on the round-3 engines (kf_score_v2.hip)
int64_t precondition_v3_workspace_bytes(int64_t q, int64_t R, int64_t O, int64_t W);
bool precondition_v3_eligible(int64_t q, int64_t R, int64_t O, int64_t I, int64_t W);
int precondition_v3(void* Pout, const void* G, const void* A, int64_t q, int64_t R, int64_t O, int64_t I, int append_ones, const float* Qg,
                    const void* Qg_bf16, const float* bias_row, int64_t Ip, const float* inv_lambda, float scale, const void* Qa_bf16,
                    const void* QgT_bf16, const void* QaT_bf16, int64_t W, void* workspace, void* stream);
}

namespace {

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline int launch_status() { return hipGetLastError() == hipSuccess ? KF_OK : KF_ERR_LAUNCH_FAILED; }
inline bool float_dtype(int d) { return d == KF_F32 || d == KF_BF16 || d == KF_F16 || d == KF_F64; }
inline int64_t dtype_size(int d) {
    switch (d) {
        case KF_F32: case KF_I32: return 4;
        case KF_BF16: case KF_F16: return 2;
        case KF_F64: case KF_I64: return 8;
        default: return 1;
    }
}

__device__ __forceinline__ void store_as(void* p, int dtype, int64_t idx, float v) {
    switch (dtype) {
        case F32: reinterpret_cast<float*>(p)[idx] = v; break;
        case BF16: {  // round-to-nearest-even
            uint32_t u = __float_as_uint(v);
            if ((u & 0x7fffffffu) > 0x7f800000u) { u |= 0x00400000u; }
            else { u += 0x7fffu + ((u >> 16) & 1u); }
            reinterpret_cast<uint16_t*>(p)[idx] = static_cast<uint16_t>(u >> 16);
            break;
        }
        case F16: reinterpret_cast<_Float16*>(p)[idx] = static_cast<_Float16>(v); break;
        default: reinterpret_cast<double*>(p)[idx] = static_cast<double>(v); break;
    }
}

// ------------------------------------------------------------------------------------------------
// Generic strided batched GEMM
// ------------------------------------------------------------------------------------------------
struct GemmArgs {
    void* C;
    int c_dtype;        // F32, or BF16 for plain (non-atomic, beta == 0) stores
    int64_t ldc, c_batch_stride;
    kf_view A, B;
    int M, N, K;        // extents including the virtual ones row / k
    int ksplit, kchunk;
    float alpha, beta;
    const float* mul;
    int64_t ld_mul;
    int atomic;         // accumulate with atomicAdd (split-K or batch-summing); beta pre-applied
    int64_t c_tile_stride;  // see HalfGemmArgs::c_tile_stride
};

template <int DT>
__device__ __forceinline__ StridedLoader<DT> make_loader(const kf_view& v, int64_t z, int row0, int contig_k) {
    StridedLoader<DT> l;
    l.row_stride = v.row_stride;
    l.k_stride = v.k_stride;
    const int64_t off = z * v.batch_stride + static_cast<int64_t>(row0) * v.row_stride;
    l.p = reinterpret_cast<const char*>(v.p) + off * (DT == F32 ? 4 : 2);
    l.rows = static_cast<int>(v.rows) - row0;  // >= 0: row0 <= rows whenever a tile exists
    l.depth = static_cast<int>(v.depth);
    l.ones_row = v.ones_row;
    l.ones_k = v.ones_k;
    l.square = v.square;
    l.contig_k = contig_k;
    return l;
}

template <int DTA, int DTB>
__global__ __launch_bounds__(NTHREADS) void gemm_kernel(GemmArgs a) {
    __shared__ float smem[SMEM_FLOATS];
    const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
    const int z = blockIdx.z / a.ksplit, ks = blockIdx.z % a.ksplit;
    const int k_begin = ks * a.kchunk;
    const int k_end = min(a.K, k_begin + a.kchunk);
    StridedLoader<DTA> la = make_loader<DTA>(a.A, z, m0, a.A.k_stride == 1);
    StridedLoader<DTB> lb = make_loader<DTB>(a.B, z, n0, a.B.k_stride == 1);
    f32x16 acc[2][2];
    zero_acc(acc);
    mainloop(la, lb, k_begin, k_end, acc, smem);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
    const int64_t cz = static_cast<int64_t>(z) * a.c_batch_stride;
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + acc_row(wm, ti, r, lane), n = n0 + acc_col(wn, tj, lane);
                if (m < a.M && n < a.N) {
                    float v = a.alpha * acc[ti][tj][r];
                    if (a.mul) v *= a.mul[static_cast<int64_t>(m) * a.ld_mul + n];
                    int64_t idx = cz + static_cast<int64_t>(m) * a.ldc + n;
                    if (a.c_tile_stride) {
                        const int64_t d = static_cast<int64_t>(m) * a.ldc + n;
                        idx = (d >> 6) * a.c_tile_stride + static_cast<int64_t>(z) * 64 + (d & 63);
                    }
                    if (a.c_dtype == BF16) {
                        store_as(a.C, BF16, idx, v);
                    } else {
                        float* dst = reinterpret_cast<float*>(a.C) + idx;
                        if (a.atomic) atomicAdd(dst, v);
                        else *dst = (a.beta == 0.0f) ? v : v + a.beta * *dst;
                    }
                }
            }
}

__global__ void scale_matrix_kernel(float* C, int64_t ldc, int64_t batch_stride, int M, int N, float beta) {
    const int64_t total = static_cast<int64_t>(M) * N;
    float* base = C + blockIdx.y * batch_stride;
    for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < total;
         i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        float* p = base + (i / N) * ldc + (i % N);
        *p = beta == 0.0f ? 0.0f : beta * *p;
    }
}

// bf16 MFMA engine eligibility: 0 = no, 1 = NT (both K-contiguous), 2 = TN (both K-strided, rows contiguous)
int bf16_engine_mode(const kf_view& A, const kf_view& B, const float* mul) {
    auto plain = [](const kf_view& v) {
        return v.dtype == KF_BF16 && !v.ones_row && !v.ones_k && !v.square && v.batch_stride % 8 == 0 &&
               (reinterpret_cast<uintptr_t>(v.p) & 15) == 0;
    };
    if (mul || !plain(A) || !plain(B)) return 0;
    auto nt = [](const kf_view& v) {
        return v.k_stride == 1 && v.row_stride % 8 == 0 && v.depth % 8 == 0 &&
               (v.k_tile_stride == 0 || (v.depth % 64 == 0 && v.k_tile_stride % 8 == 0));
    };
    auto tn = [](const kf_view& v) { return v.row_stride == 1 && v.k_stride % 8 == 0 && v.rows % 8 == 0 && v.k_tile_stride == 0; };
    if (nt(A) && nt(B) && A.depth >= HBK) return 1;
    if (tn(A) && tn(B) && A.rows > 1 && B.rows > 1) return 2;
    return 0;
}

// epilogue side inputs of the bf16 engine (see HalfGemmArgs)
struct HalfExtras {
    const float* row_add = nullptr; int row_add_n = 0;
    const float* mul = nullptr; int64_t ld_mul = 0; int mul_n = 0;
};

int configure_kernels();  // defined after the kernels it configures

int launch_gemm_bf16(int mode, void* C, int c_dtype, int64_t ldc, int64_t c_batch_stride, const kf_view& A, const kf_view& B,
                     int64_t batch, float alpha, float beta, hipStream_t st, int64_t c_tile_stride, bool symmetric = false,
                     const HalfExtras* extras = nullptr) {
    if (configure_kernels() != KF_OK) return KF_ERR_LAUNCH_FAILED;
    const int64_t M = A.rows, N = B.rows, K = A.depth;
    // tall row-major NT product with a bf16 result (the eigenbasis rotations of Lambda / preconditioning): 256 x 256-tile
    // LDS-DMA kernel (csrc/kf_score_v2.hip) -- 0.55-0.62 PFLOP/s on this engine at those shapes
    if (mode == 1 && batch == 1 && c_dtype == KF_BF16 && beta == 0.0f && !symmetric && c_tile_stride == 0 && !(extras && extras->mul) &&
        !A.k_tile_stride && !B.k_tile_stride && K % 64 == 0 && N % 8 == 0 && ldc % 8 == 0 && M < (1LL << 31) - 256 &&
        (reinterpret_cast<uintptr_t>(C) & 15) == 0 &&
        cdiv(M, 256) * cdiv(N, 256) >= 512 /* two rounds of one workgroup per CU */ && cdiv(N, 256) * 256 * 100 <= N * 115)
        return rotate_gemm_v2(C, ldc, A.p, A.row_stride, B.p, B.row_stride, M, N, K, alpha, extras ? extras->row_add : nullptr,
                              extras ? extras->row_add_n : 0, st);
    const bool batch_sum = (c_batch_stride == 0 && batch > 1 && c_tile_stride == 0);
    const int64_t tm = cdiv(M, 128), tn = cdiv(N, 128);
    const int64_t tiles = (symmetric ? tm * (tm + 1) / 2 : tm * tn) * batch, ksteps = cdiv(K, HBK);
    int64_t ksplit = 1;
    if (c_dtype == KF_F32 && tiles < 1024 && ksteps >= 8) ksplit = std::max<int64_t>(1, std::min<int64_t>(cdiv(1024, tiles), ksteps / 4));
    const int64_t kchunk = cdiv(ksteps, ksplit) * HBK;
    ksplit = cdiv(K, kchunk);
    const bool atomic = ksplit > 1 || batch_sum;
    if (c_dtype == KF_BF16 && (atomic || beta != 0.0f)) return KF_ERR_INVALID_ARGUMENT;
    if (atomic && beta != 1.0f)
        hipLaunchKernelGGL(scale_matrix_kernel, dim3(static_cast<unsigned>(std::min<int64_t>(cdiv(M * N, 256), 4096)), batch_sum ? 1 : static_cast<unsigned>(batch)),
                           dim3(256), 0, st, reinterpret_cast<float*>(C), ldc, c_batch_stride, static_cast<int>(M), static_cast<int>(N), beta);
    HalfGemmArgs h;
    h.C = C; h.c_dtype = c_dtype; h.ldc = ldc; h.c_batch_stride = c_batch_stride;
    h.A.p = reinterpret_cast<const uint16_t*>(A.p); h.A.batch_stride = A.batch_stride; h.A.rows = static_cast<int>(A.rows); h.A.depth = static_cast<int>(A.depth);
    h.B.p = reinterpret_cast<const uint16_t*>(B.p); h.B.batch_stride = B.batch_stride; h.B.rows = static_cast<int>(B.rows); h.B.depth = static_cast<int>(B.depth);
    h.A.ld = mode == 1 ? A.row_stride : A.k_stride;
    h.B.ld = mode == 1 ? B.row_stride : B.k_stride;
    h.A.kt_stride = A.k_tile_stride ? A.k_tile_stride : 64;
    h.B.kt_stride = B.k_tile_stride ? B.k_tile_stride : 64;
    h.c_tile_stride = c_tile_stride;
    h.M = static_cast<int>(M); h.N = static_cast<int>(N); h.K = static_cast<int>(K);
    h.ksplit = static_cast<int>(ksplit); h.kchunk = static_cast<int>(kchunk); h.alpha = alpha; h.beta = beta; h.atomic = atomic ? 1 : 0;
    h.tiles_m = static_cast<int>(tm); h.tiles_n = static_cast<int>(tn); h.chunks = static_cast<int>(batch * ksplit);
    h.symmetric = symmetric ? 1 : 0;
    const HalfExtras none;
    const HalfExtras& ex = extras ? *extras : none;
    h.row_add = ex.row_add; h.row_add_n = ex.row_add_n; h.mul = ex.mul; h.ld_mul = ex.ld_mul; h.mul_n = ex.mul_n;
    if ((ex.row_add || ex.mul) && (atomic || beta != 0.0f)) return KF_ERR_INVALID_ARGUMENT;  // applied once, on the full sum
    const int64_t nblocks = 8 * cdiv(batch * ksplit * (symmetric ? tm * (tm + 1) / 2 : tm * tn), 8);
    if (nblocks >= (1LL << 31)) return KF_ERR_INVALID_ARGUMENT;
    const dim3 grid(static_cast<unsigned>(nblocks));
    if (mode == 1) hipLaunchKernelGGL((gemm_bf16_kernel<false>), grid, dim3(NTHREADS), HSMEM_BYTES, st, h);
    else if (symmetric) hipLaunchKernelGGL((gemm_bf16_kernel<true, true>), grid, dim3(NTHREADS), HSMEM_BYTES, st, h);
    else hipLaunchKernelGGL((gemm_bf16_kernel<true>), grid, dim3(NTHREADS), HSMEM_BYTES, st, h);
    return launch_status();
}

int launch_gemm(void* Cv, int64_t ldc, int64_t c_batch_stride, const kf_view& A, const kf_view& B,
                int64_t batch, float alpha, float beta, const float* mul, int64_t ld_mul, hipStream_t st,
                int c_dtype = KF_F32, int64_t c_tile_stride = 0) {
    float* C = reinterpret_cast<float*>(Cv);
    if (!C || !A.p || !B.p || batch < 0) return KF_ERR_INVALID_ARGUMENT;
    if (c_dtype != KF_F32 && c_dtype != KF_BF16) return KF_ERR_UNSUPPORTED_DTYPE;
    // supported operand dtype pairs: (f32|bf16|f16, f32) and (x, x)
    const bool a_ok = A.dtype == KF_F32 || A.dtype == KF_BF16 || A.dtype == KF_F16;
    if (!a_ok || !(B.dtype == KF_F32 || B.dtype == A.dtype)) return KF_ERR_UNSUPPORTED_DTYPE;
    if (A.depth + A.ones_k != B.depth + B.ones_k) return KF_ERR_INVALID_ARGUMENT;
    if (A.rows < 1 || B.rows < 1 || A.depth < 1 || B.depth < 1) {
        // degenerate (empty real extent): the clamped loads of the engine need >= 1 real row / column
        return (A.rows + A.ones_row <= 0 || B.rows + B.ones_row <= 0 || batch == 0) ? KF_OK : KF_ERR_INVALID_ARGUMENT;
    }
    const int64_t M = A.rows + A.ones_row, N = B.rows + B.ones_row, K = A.depth + A.ones_k;
    if (M <= 0 || N <= 0 || batch == 0) return KF_OK;
    if (const int mode = bf16_engine_mode(A, B, mul)) {
        const bool accumulates = (c_batch_stride == 0 && batch > 1 && c_tile_stride == 0) || beta != 0.0f;
        if (!(c_dtype == KF_BF16 && accumulates))
            return launch_gemm_bf16(mode, Cv, c_dtype, ldc, c_batch_stride, A, B, batch, alpha, beta, st, c_tile_stride);
    }
    if (A.k_tile_stride || B.k_tile_stride) return KF_ERR_INVALID_ARGUMENT;  // tiled operands: bf16 NT engine only
    if (M >= (1LL << 30) || N >= (1LL << 30) || K >= (1LL << 30)) return KF_ERR_INVALID_ARGUMENT;
    const int64_t tiles = cdiv(M, BM) * cdiv(N, BN);
    const bool batch_sum = (c_batch_stride == 0 && batch > 1 && c_tile_stride == 0);
    // split-K so that small-output / deep-K contractions still fill 256 CUs
    int64_t ksplit = 1;
    const int64_t ksteps = cdiv(K, BK);
    const int64_t want = 1024;
    if (tiles * batch < want && ksteps >= 8) {
        ksplit = std::min<int64_t>(cdiv(want, tiles * batch), ksteps / 4);
        if (ksplit < 1) ksplit = 1;
    }
    int64_t kchunk = cdiv(ksteps, ksplit) * BK;
    ksplit = cdiv(K, kchunk);
    if (K == 0) { ksplit = 1; kchunk = BK; }
    if (c_dtype == KF_BF16) {  // low-precision outputs are plain stores: no split-K, no accumulation
        if (batch_sum || beta != 0.0f) return KF_ERR_INVALID_ARGUMENT;
        ksplit = 1; kchunk = cdiv(ksteps, 1) * BK;
    }
    const bool atomic = batch_sum || ksplit > 1;
    if (batch * ksplit > 65535) return KF_ERR_INVALID_ARGUMENT;
    if (atomic && beta != 1.0f) {
        dim3 g(static_cast<unsigned>(std::min<int64_t>(cdiv(M * N, 256), 4096)), batch_sum ? 1 : static_cast<unsigned>(batch));
        hipLaunchKernelGGL(scale_matrix_kernel, g, dim3(256), 0, st, C, ldc, c_batch_stride, static_cast<int>(M), static_cast<int>(N), beta);
    }
    GemmArgs a;
    a.C = C; a.c_dtype = c_dtype; a.ldc = ldc; a.c_batch_stride = c_batch_stride; a.A = A; a.B = B;
    a.M = static_cast<int>(M); a.N = static_cast<int>(N); a.K = static_cast<int>(K);
    a.ksplit = static_cast<int>(ksplit); a.kchunk = static_cast<int>(kchunk);
    a.alpha = alpha; a.beta = beta; a.mul = mul; a.ld_mul = ld_mul; a.atomic = atomic ? 1 : 0;
    a.c_tile_stride = c_tile_stride;
    dim3 grid(static_cast<unsigned>(cdiv(N, BN)), static_cast<unsigned>(cdiv(M, BM)), static_cast<unsigned>(batch * ksplit));
    if (A.dtype == KF_F32 && B.dtype == KF_F32) hipLaunchKernelGGL((gemm_kernel<F32, F32>), grid, dim3(NTHREADS), 0, st, a);
    else if (A.dtype == KF_BF16 && B.dtype == KF_F32) hipLaunchKernelGGL((gemm_kernel<BF16, F32>), grid, dim3(NTHREADS), 0, st, a);
    else if (A.dtype == KF_F16 && B.dtype == KF_F32) hipLaunchKernelGGL((gemm_kernel<F16, F32>), grid, dim3(NTHREADS), 0, st, a);
    else if (A.dtype == KF_BF16) hipLaunchKernelGGL((gemm_kernel<BF16, BF16>), grid, dim3(NTHREADS), 0, st, a);
    else hipLaunchKernelGGL((gemm_kernel<F16, F16>), grid, dim3(NTHREADS), 0, st, a);
    return launch_status();
}

// ------------------------------------------------------------------------------------------------
// Stage 1: covariance SYRK with fused flatten / mask / ones column
// ------------------------------------------------------------------------------------------------
template <int DT, int MDT>  // MDT: mask dtype, -1 = no mask
struct SyrkLoader {
    const void* p;
    int64_t n_rows, rows_inner, outer_stride, row_stride, col_stride;
    const void* mask;
    int d_in, d, col0;   // features [col0, col0+128) of X'
    int contig_k;
    // "row" r = feature index (tile-relative), k = sample row.  Branch-free: clamped loads + selects.
    struct Raw { typename RawOf<DT>::type x; typename RawOf<(MDT >= 0 ? MDT : F32)>::type m; };

    __device__ __forceinline__ Raw fetch(int r, int k) const {
        const int c = col0 + r;
        const uint32_t nc = static_cast<uint32_t>(k < n_rows ? k : static_cast<int>(n_rows) - 1);
        const int cc = c < d_in ? c : d_in - 1;
        Raw raw;
        raw.m = 0;
        if constexpr (MDT >= 0) raw.m = load_raw<MDT>(mask, nc);
        int64_t off;
        if (rows_inner >= n_rows) {
            off = static_cast<int64_t>(nc) * row_stride;
        } else {
            const uint32_t inner = static_cast<uint32_t>(rows_inner);
            const uint32_t hi = nc / inner, lo = nc - hi * inner;
            off = static_cast<int64_t>(hi) * outer_stride + static_cast<int64_t>(lo) * row_stride;
        }
        raw.x = load_raw<DT>(p, off + static_cast<int64_t>(cc) * col_stride);
        return raw;
    }
    __device__ __forceinline__ float value(const Raw& raw, int r, int k) const {
        const int c = col0 + r;
        const bool ok = c < d && k < n_rows;
        float mk = 1.0f;
        if constexpr (MDT >= 0) mk = raw_to_f32<MDT>(raw.m);
        const float x = raw_to_f32<DT>(raw.x);
        const float v = (c == d_in) ? mk : mk * x;  // c == d_in: the ones column (d == d_in + 1)
        return ok ? v : 0.0f;
    }
};

struct SyrkBase {
    const void* p; int dtype;
    int64_t n_rows, rows_inner, outer_stride, row_stride, col_stride;
    const void* mask; int mask_dtype;
    int d_in, d, contig_k;
};

struct SyrkArgs {
    float* C; int64_t ldc;
    SyrkBase base;
    int tiles, ksplit; int64_t kchunk;
    float alpha; int atomic;
};

template <int DT, int MDT>
__global__ __launch_bounds__(NTHREADS) void syrk_kernel(SyrkArgs a) {
    __shared__ float smem[SMEM_FLOATS];
    // upper-triangular tile pair (ti <= tj) from the linear block index
    int t = blockIdx.x, ti = 0;
    while (t >= a.tiles - ti) { t -= a.tiles - ti; ++ti; }
    const int tj = ti + t;
    SyrkLoader<DT, MDT> la;
    la.p = a.base.p; la.n_rows = a.base.n_rows; la.rows_inner = a.base.rows_inner; la.outer_stride = a.base.outer_stride;
    la.row_stride = a.base.row_stride; la.col_stride = a.base.col_stride; la.mask = a.base.mask;
    la.d_in = a.base.d_in; la.d = a.base.d; la.contig_k = a.base.contig_k;
    SyrkLoader<DT, MDT> lb = la;
    la.col0 = ti * BM;
    lb.col0 = tj * BN;
    const int64_t k_begin = static_cast<int64_t>(blockIdx.y) * a.kchunk;
    const int64_t k_end = min(a.base.n_rows, k_begin + a.kchunk);
    f32x16 acc[2][2];
    zero_acc(acc);
    mainloop(la, lb, static_cast<int>(k_begin), static_cast<int>(k_end), acc, smem);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
    const int d = a.base.d;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = ti * BM + acc_row(wm, i, r, lane), n = tj * BN + acc_col(wn, j, lane);
                if (m < d && n < d) {
                    const float v = a.alpha * acc[i][j][r];
                    float* up = a.C + static_cast<int64_t>(m) * a.ldc + n;
                    if (a.atomic) atomicAdd(up, v);
                    else *up += v;
                }
            }
    if (ti != tj)  // uniform per workgroup
        mirror_through_lds(
            smem, wm, wn, lane, wave, [&](int i, int j, int r) { return a.alpha * acc[i][j][r]; },
            [&](int nl, int ml, float v) {
                const int n = tj * BN + nl, m = ti * BM + ml;
                if (m < d && n < d) {
                    float* lo = a.C + static_cast<int64_t>(n) * a.ldc + m;
                    if (a.atomic) atomicAdd(lo, v);
                    else *lo += v;
                }
            });
}

__global__ void count_kernel(int64_t* count, const void* mask, int mask_dtype, int64_t n) {
    // one block: count += sum(mask)  (exact in int64 for 0/1 masks)
    __shared__ long long part[256];
    long long s = 0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        switch (mask_dtype) {
            case I64: s += reinterpret_cast<const int64_t*>(mask)[i]; break;
            case I32: s += reinterpret_cast<const int32_t*>(mask)[i]; break;
            case U8: s += reinterpret_cast<const uint8_t*>(mask)[i]; break;
            default: s += static_cast<long long>(llrintf(load_f32(mask, mask_dtype, i))); break;
        }
    }
    part[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) *count += part[0];
}

__global__ void add_count_kernel(int64_t* count, int64_t n) { *count += n; }

// ------------------------------------------------------------------------------------------------
// Activation covariance of a convolution with a SMALL patch width (round 6): d' = C k1 k2 (+ 1) <= 32 -- the first layer of an
// image model (ResNet-9: 3 x 3 x 3 = 27 columns, 1 024 positions x 1 000 images = a million rows per batch).  The general path
// materialised the fp32 patch matrix (110 MB per batch; im2col_kernel) and read it back through the 128 x 128 fp32 engine for a
// 27 x 27 result: 0.8 ms per batch for 0.8 GFLOP.  Here ONE v_mfma_f32_32x32x2_f32 covers the whole result: lane l of a wave
// holds the patch value P[pos + (l >> 5)][j = l & 31] gathered straight from the NCHW input (a lane's (c, ky, kx) never changes:
// its input offset is one multiply-add per position; the image is L1 / L2 resident after its first touch), and that ONE register
// is both MFMA operands -- A[i][k] = P[pos_k][i], B[k][j] = P[pos_k][j] -- so C += P^T P two positions per instruction, exact fp32
// products, fp32 accumulation (the arithmetic of kf_syrk_accum on fp32 rows).  Reference: module/conv2d.py:15-64 (patch order
// (c, ky, kx), zero padding), :106-128 (ones column), tracker/factor.py:58.
// ------------------------------------------------------------------------------------------------
struct ConvCovSmallArgs {
    float* C; int64_t ldc;
    const void* x;
    int64_t b, Cin, H, W, O1, O2, npos;
    int k1, k2, s1, s2, p1, p2, d1, d2, D, ones;
    float alpha;
};

template <int DT>
__global__ __launch_bounds__(256) void conv_cov_small_kernel(ConvCovSmallArgs a) {
    __shared__ float tile[32][33];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, half = lane >> 5;
    for (int e = tid; e < 32 * 33; e += 256) (&tile[0][0])[e] = 0.0f;
    __syncthreads();
    // this lane's patch column: (c, ky, kx) -> input offset relative to the top-left input pixel of a position
    const int kk = a.k1 * a.k2;
    const bool real = j < a.D;
    const int c = real ? j / kk : 0, ky = real ? (j % kk) / a.k2 : 0, kx = real ? j % a.k2 : 0;
    const int dy = ky * a.d1 - a.p1, dx = kx * a.d2 - a.p2;
    const int64_t plane = a.H * a.W, coff = static_cast<int64_t>(c) * plane, image = a.Cin * plane;
    const bool one = a.ones && j == a.D;
    // waves take contiguous ranges of position PAIRS
    const int64_t pairs = (a.npos + 1) / 2, waves = static_cast<int64_t>(gridDim.x) * 4, w = static_cast<int64_t>(blockIdx.x) * 4 + wave;
    const int64_t per = (pairs + waves - 1) / waves, first = w * per, last = min(pairs, first + per);
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
    if (first < last) {
        int64_t pos = 2 * first + half;
        const int64_t P = a.O1 * a.O2;
        int64_t n = pos / P;
        int64_t rem = pos - n * P;
        int oy = static_cast<int>(rem / a.O2), ox = static_cast<int>(rem - static_cast<int64_t>(oy) * a.O2);
        auto value = [&]() -> float {
            float v = 0.0f;
            if (pos < a.npos) {
                const int iy = oy * a.s1 + dy, ix = ox * a.s2 + dx;
                if (real && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) v = load_t<DT>(a.x, n * image + coff + static_cast<int64_t>(iy) * a.W + ix);
                else if (one) v = 1.0f;
            }
            return v;
        };
        auto advance = [&]() {   // two positions further (row-major over (n, oy, ox))
            pos += 2;
            ox += 2;
            while (ox >= a.O2) { ox -= static_cast<int>(a.O2); if (++oy >= a.O1) { oy = 0; ++n; } }
        };
        int64_t it = first;
        for (; it + 8 <= last; it += 8) {   // eight gathers in flight, then eight MFMAs on two accumulator chains
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { v[u] = value(); advance(); }
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v[u], v[u], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v[u + 1], v[u + 1], acc1, 0, 0, 0);
            }
        }
        for (; it < last; ++it) {
            const float v = value();
            advance();
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v, v, acc0, 0, 0, 0);
        }
    }
    // the four waves' tiles -> one 32 x 32 tile in LDS -> coalesced fp32 atomics (alpha folded in)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        atomicAdd(&tile[row][j], acc0[r] + acc1[r]);
    }
    __syncthreads();
    const int dp = a.D + a.ones;
    for (int e = tid; e < dp * dp; e += 256) {
        const int row = e / dp, col = e - row * dp;
        const float v = tile[row][col];
        if (v != 0.0f) atomicAdd(a.C + static_cast<int64_t>(row) * a.ldc + col, a.alpha * v);
    }
}

// ------------------------------------------------------------------------------------------------
// im2col (conv2d.py:15-64): out[b, p, i] with i = (c, ky, kx); group mean folded in
// ------------------------------------------------------------------------------------------------
struct Im2colArgs {
    void* out; int out_dtype; const void* x; int in_dtype;
    int64_t b, C, H, W; int k1, k2, s1, s2, p1, p2, d1, d2, groups, append_ones;
    int64_t O1, O2, Cg, Ip;
};

__global__ void im2col_kernel(Im2colArgs a) {
    const int64_t P = a.O1 * a.O2;
    const int64_t total = a.b * P * a.Ip;
    const float inv_g = 1.0f / static_cast<float>(a.groups);
    for (int64_t e = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; e < total;
         e += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const int64_t i = e % a.Ip, bp = e / a.Ip, p = bp % P, n = bp / P;
        float v;
        if (a.append_ones && i == a.Ip - 1) {
            v = 1.0f;
        } else {
            const int64_t kx = i % a.k2, ky = (i / a.k2) % a.k1, c = i / (a.k1 * a.k2);
            const int64_t oy = p / a.O2, ox = p % a.O2;
            const int64_t iy = oy * a.s1 - a.p1 + ky * a.d1, ix = ox * a.s2 - a.p2 + kx * a.d2;
            v = 0.0f;
            if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) {
                for (int g = 0; g < a.groups; ++g)
                    v += load_f32(a.x, a.in_dtype, ((n * a.C + g * a.Cg + c) * a.H + iy) * a.W + ix);
                if (a.groups > 1) v *= inv_g;
            }
        }
        store_as(a.out, a.out_dtype, e, v);
    }
}

// 8 consecutive patch elements per thread, one 16-B store (2-byte output dtypes, I' % 8 == 0, groups == 1):
// the (c, ky, kx) decomposition is done once per octet and advanced incrementally.
__global__ void im2col_vec8_kernel(Im2colArgs a) {
    const int P = static_cast<int>(a.O1 * a.O2), Ip = static_cast<int>(a.Ip), octs = Ip >> 3;
    const int64_t total = a.b * P * octs;
    const int H = static_cast<int>(a.H), W = static_cast<int>(a.W);
    const bool same_dtype = a.in_dtype == a.out_dtype;
    for (int64_t e = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; e < total;
         e += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const int iv = static_cast<int>(e % octs);
        const int64_t bp = e / octs;
        const int p = static_cast<int>(bp % P);
        const int64_t n = bp / P;
        const int oy = p / static_cast<int>(a.O2), ox = p - oy * static_cast<int>(a.O2);
        const int by = oy * a.s1 - a.p1, bx = ox * a.s2 - a.p2;
        int i0 = iv * 8;
        int kx = i0 % a.k2, ky = (i0 / a.k2) % a.k1, c = i0 / (a.k1 * a.k2);
        const int64_t img = n * a.C * H * W;
        // Addresses are clamped into the image and the loads issued unconditionally (zero selected afterwards):
        // a branch around a load makes the compiler serialise the eight gathers behind one another.
        uint32_t raw[8];
        bool inside[8];
        // (round 6: the ones column of a biased layer -- element I' - 1 of the LAST octet when C k1 k2 = 7 mod 8 -- used to be gathered
        // like a patch element, from channel C: one element past the image and not a one; found by kf_conv2d_cov_small's tests)
        const int real = Ip - a.append_ones;
        const uint32_t one_bits = a.out_dtype == KF_BF16 ? 0x3F80u : 0x3C00u;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int iy = by + ky * a.d1, ix = bx + kx * a.d2;
            inside[j] = iy >= 0 && iy < H && ix >= 0 && ix < W && i0 + j < real;
            const int cy = min(max(iy, 0), H - 1), cx = min(max(ix, 0), W - 1);
            const int64_t src = img + (static_cast<int64_t>(min(c, static_cast<int>(a.C) - 1)) * H + cy) * W + cx;
            if (same_dtype) {
                raw[j] = reinterpret_cast<const uint16_t*>(a.x)[src];
            } else {
                uint16_t tmp;
                store_as(&tmp, a.out_dtype, 0, load_f32(a.x, a.in_dtype, src));
                raw[j] = tmp;
            }
            if (++kx == a.k2) { kx = 0; if (++ky == a.k1) { ky = 0; ++c; } }
        }
        uint32_t w[4];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t bits = inside[j] ? raw[j] : (i0 + j == real ? one_bits : 0u);   // (i0 + j == real only with append_ones)
            if (j & 1) w[j >> 1] |= bits << 16; else w[j >> 1] = bits;
        }
        typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
        reinterpret_cast<u32x4_t*>(a.out)[e] = u32x4_t{w[0], w[1], w[2], w[3]};
    }
}

// LDS-staged variant for 2-byte dtypes (same in / out dtype, groups == 1, I' % 8 == 0, W % 8 == 0): one workgroup
// owns one image and a band of RB output rows.  The input rows the band touches are copied ONCE into LDS with
// coalesced 16-byte loads ([C][NR][W + 2 p2], zero-filled borders, so no bounds checks afterwards); every
// 16-byte chunk of the patch matrix is then assembled from eight ds_read_u16.  The scattered 2-byte global
// gathers of the kernel above kept it at 0.6 TB/s of writes (texture-address bound); this one streams.
struct Im2colLdsArgs { Im2colArgs a; int RB, NR, Wp; };

__global__ __launch_bounds__(256) void im2col_lds_kernel(Im2colLdsArgs p) {
    extern __shared__ uint16_t tile[];
    const Im2colArgs& a = p.a;
    const int C = static_cast<int>(a.C), H = static_cast<int>(a.H), W = static_cast<int>(a.W);
    const int O1 = static_cast<int>(a.O1), O2 = static_cast<int>(a.O2), Ip = static_cast<int>(a.Ip);
    const int bands = (O1 + p.RB - 1) / p.RB;
    const int64_t n = blockIdx.x / bands;
    const int oy0 = static_cast<int>(blockIdx.x % bands) * p.RB;
    const int rows_here = min(p.RB, O1 - oy0);
    const int iy0 = oy0 * a.s1 - a.p1;
    const uint16_t* x = reinterpret_cast<const uint16_t*>(a.x) + n * C * H * W;
    // ---- stage: [C][NR][Wp]; 8 elements (16 B) per thread-iteration along W
    const int w8 = W >> 3, chunks = C * p.NR * w8;
    for (int t = threadIdx.x; t < chunks; t += 256) {
        const int wv = t % w8, r = (t / w8) % p.NR, c = t / (w8 * p.NR);
        const int iy = iy0 + r;
        typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
        u32x4_t v = {0u, 0u, 0u, 0u};
        if (iy >= 0 && iy < H) v = *reinterpret_cast<const u32x4_t*>(x + (static_cast<int64_t>(c) * H + iy) * W + wv * 8);
        uint16_t* dst = tile + (c * p.NR + r) * p.Wp + a.p2 + wv * 8;
        dst[0] = static_cast<uint16_t>(v[0]); dst[1] = static_cast<uint16_t>(v[0] >> 16);
        dst[2] = static_cast<uint16_t>(v[1]); dst[3] = static_cast<uint16_t>(v[1] >> 16);
        dst[4] = static_cast<uint16_t>(v[2]); dst[5] = static_cast<uint16_t>(v[2] >> 16);
        dst[6] = static_cast<uint16_t>(v[3]); dst[7] = static_cast<uint16_t>(v[3] >> 16);
    }
    if (a.p2 > 0)
        for (int t = threadIdx.x; t < C * p.NR * 2 * a.p2; t += 256) {
            const int j = t % (2 * a.p2), row = t / (2 * a.p2);
            tile[row * p.Wp + (j < a.p2 ? j : W + j)] = 0;
        }
    __syncthreads();
    // ---- emit: octets of the [rows_here * O2, I'] block, consecutive threads -> consecutive 16-byte chunks
    const int octs = Ip >> 3, total = rows_here * O2 * octs, kk = a.k1 * a.k2;
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    u32x4_t* out = reinterpret_cast<u32x4_t*>(a.out) + (n * O1 * O2 + static_cast<int64_t>(oy0) * O2) * octs;
    for (int e = threadIdx.x; e < total; e += 256) {
        const int iv = e % octs, pos = e / octs;
        const int oyl = pos / O2, ox = pos - oyl * O2;
        const int i0 = iv * 8;
        int c = i0 / kk, rem = i0 - c * kk;
        int ky = rem / a.k2, kx = rem - ky * a.k2;
        const int base_r = oyl * a.s1, base_x = ox * a.s2;
        uint32_t w[4];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t bits = tile[(c * p.NR + base_r + ky * a.d1) * p.Wp + base_x + kx * a.d2];
            if (j & 1) w[j >> 1] |= bits << 16; else w[j >> 1] = bits;
            if (++kx == a.k2) { kx = 0; if (++ky == a.k1) { ky = 0; ++c; } }
        }
        out[e] = u32x4_t{w[0], w[1], w[2], w[3]};
    }
}

// ------------------------------------------------------------------------------------------------
// Stage 2: Lambda for R > 1 -- per sample z: M_z = Gt_z^T At_z (K = R), Lambda += M_z^2
// ------------------------------------------------------------------------------------------------
struct LambdaArgs {
    float* L; int64_t ldl; const float* Gt; const float* At;
    int b, R, O, Ip, zchunk; float scale2;
};

__global__ __launch_bounds__(NTHREADS) void lambda_kernel(LambdaArgs a) {
    __shared__ float smem[SMEM_FLOATS];
    const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
    const int z_begin = blockIdx.z * a.zchunk, z_end = min(a.b, z_begin + a.zchunk);
    f32x16 sq[2][2];
    zero_acc(sq);
    for (int z = z_begin; z < z_end; ++z) {
        StridedLoader<F32> la, lb;
        la.p = a.Gt + (static_cast<int64_t>(z) * a.R) * a.O + m0;
        la.row_stride = 1; la.k_stride = a.O; la.rows = max(a.O - m0, 0); la.depth = a.R;
        la.ones_row = la.ones_k = la.square = 0; la.contig_k = 0;
        lb = la;
        lb.p = a.At + (static_cast<int64_t>(z) * a.R) * a.Ip + n0; lb.k_stride = a.Ip; lb.rows = max(a.Ip - n0, 0);
        f32x16 acc[2][2];
        zero_acc(acc);
        mainloop(la, lb, 0, a.R, acc, smem);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) sq[i][j][r] += acc[i][j][r] * acc[i][j][r];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
    const bool atomic = gridDim.z > 1;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + acc_row(wm, i, r, lane), n = n0 + acc_col(wn, j, lane);
                if (m < a.O && n < a.Ip) {
                    float* dst = a.L + static_cast<int64_t>(m) * a.ldl + n;
                    const float v = a.scale2 * sq[i][j][r];
                    if (atomic) atomicAdd(dst, v); else *dst += v;
                }
            }
}

// ------------------------------------------------------------------------------------------------
// Stage 3a: inverse Lambda in fp64
// ------------------------------------------------------------------------------------------------
__global__ void sum_f64_kernel(double* out, const float* x, int64_t n) {
    __shared__ double part[256];
    double s = 0.0;
    for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n;
         i += static_cast<int64_t>(gridDim.x) * blockDim.x)
        s += static_cast<double>(x[i]);
    part[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(out, part[0]);
}

__global__ void inv_lambda_kernel(float* out, const float* L, int64_t n, double n_lambda, double damping,
                                  const double* sum) {
    double damp = damping;
    if (damping < 0.0) damp = 0.1 * (*sum / n_lambda) / static_cast<double>(n);
    for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n;
         i += static_cast<int64_t>(gridDim.x) * blockDim.x)
        out[i] = static_cast<float>(1.0 / (static_cast<double>(L[i]) / n_lambda + damp));
}

// ------------------------------------------------------------------------------------------------
// Stage 3b: pairwise score, R == 1:  scores[q,n] += scale * sum_o G[n,o] * (sum_i P[q,o,i] A'[n,i])
// One block = (q, 128 rows of o) x 128 train samples; K loop over i on the MFMA engine, then the
// G-weighted reduction over the tile's o rows in registers / LDS, one atomicAdd per (q, n).
// ------------------------------------------------------------------------------------------------
struct ScoreArgs {
    float* scores; int64_t ld_scores; const void* P; int p_dtype; const void* G; const void* A; int in_dtype;
    int Q, b, O, I, Ip, append_ones, tiles_per_q; float scale;
};

template <int DTP, int DT>
__global__ __launch_bounds__(NTHREADS) void score_r1_kernel(ScoreArgs a) {
    __shared__ float smem[SMEM_FLOATS];
    const int n0 = blockIdx.x * BN;
    const int q = blockIdx.y / a.tiles_per_q, o0 = (blockIdx.y % a.tiles_per_q) * BM;
    StridedLoader<DTP> la;
    StridedLoader<DT> lb;
    la.p = reinterpret_cast<const char*>(a.P) + (static_cast<int64_t>(q) * a.O + o0) * a.Ip * (DTP == F32 ? 4 : 2);
    la.row_stride = a.Ip; la.k_stride = 1; la.rows = a.O - o0; la.depth = a.Ip;
    la.ones_row = la.ones_k = la.square = 0; la.contig_k = 1;
    lb.p = reinterpret_cast<const char*>(a.A) + static_cast<int64_t>(n0) * a.I * (DT == F32 ? 4 : 2);
    lb.row_stride = a.I; lb.k_stride = 1; lb.rows = max(a.b - n0, 0); lb.depth = a.I;
    lb.ones_row = 0; lb.ones_k = a.append_ones; lb.square = 0; lb.contig_k = 1;
    f32x16 acc[2][2];
    zero_acc(acc);
    mainloop(la, lb, 0, a.Ip, acc, smem);

    // epilogue: this lane owns column n (two of them, tj = 0,1) and 32 rows per column
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
    float colsum[2] = {0.0f, 0.0f};
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
        const int n = n0 + acc_col(wn, tj, lane);
        const int nc = n < a.b ? n : a.b - 1;  // clamped loads + selects keep the 32 G loads in one batch
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = o0 + acc_row(wm, ti, r, lane);
                const int oc = o < a.O ? o : a.O - 1;
                // multiply by a 0/1 weight (not a select) so the load is unconditional and all 32 batch up
                const float gv = load_t<DT>(a.G, static_cast<int64_t>(nc) * a.O + oc);
                const float w = (n < a.b && o < a.O) ? 1.0f : 0.0f;
                colsum[tj] = fmaf(acc[ti][tj][r] * w, gv, colsum[tj]);
            }
    }
    // lanes l and l+32 hold the two row halves of the same column
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) colsum[tj] += __shfl_xor(colsum[tj], 32);
    // combine the two waves stacked along m (wm = 0,1) through LDS, then one atomic per column
    __syncthreads();  // smem no longer read by the mainloop
    if (wm == 1 && lane < 32) { smem[wn * 64 + lane] = colsum[0]; smem[wn * 64 + 32 + lane] = colsum[1]; }
    __syncthreads();
    if (wm == 0 && lane < 32) {
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
            const int n = n0 + wn * 64 + tj * 32 + lane;
            if (n < a.b)
                atomicAdd(a.scores + static_cast<int64_t>(q) * a.ld_scores + n,
                          a.scale * (colsum[tj] + smem[wn * 64 + tj * 32 + lane]));
        }
    }
}

void launch_score_r1(const ScoreArgs& a, dim3 grid, hipStream_t st) {
#define KF_SCORE_CASE(DTP)                                                                                         \
    do {                                                                                                           \
        if (a.in_dtype == KF_F32) hipLaunchKernelGGL((score_r1_kernel<DTP, F32>), grid, dim3(NTHREADS), 0, st, a);   \
        else if (a.in_dtype == KF_BF16) hipLaunchKernelGGL((score_r1_kernel<DTP, BF16>), grid, dim3(NTHREADS), 0, st, a); \
        else hipLaunchKernelGGL((score_r1_kernel<DTP, F16>), grid, dim3(NTHREADS), 0, st, a);                       \
    } while (0)
    if (a.p_dtype == KF_BF16) KF_SCORE_CASE(BF16);
    else KF_SCORE_CASE(F32);
#undef KF_SCORE_CASE
}

// ------------------------------------------------------------------------------------------------
// Row-wise weighted dot products (self-influence) and the broadcast product of the
// diagonal strategy.  Both are pure HBM streams: one read of each operand, 16-byte loads when the
// row length allows, fp32 accumulation, one atomicAdd per (row, split).
// ------------------------------------------------------------------------------------------------
template <int DT> struct Vec4;
template <> struct Vec4<F32> {
    static __device__ __forceinline__ void load(const void* p, int64_t idx, float (&v)[4]) {
        const float4 r = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p) + idx);
        v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w;
    }
};
template <> struct Vec4<BF16> {
    static __device__ __forceinline__ void load(const void* p, int64_t idx, float (&v)[4]) {
        const uint2 r = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(p) + idx);
        v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
        v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
    }
};

template <int DTX, int DTY>
__global__ __launch_bounds__(256) void rowwise_dot_kernel(float* out, const void* X, const void* Y, const float* W,
                                                          int64_t D, float scale, int vec) {
    __shared__ float part[4];
    const int64_t row = blockIdx.x;
    const int64_t base = row * D;
    const int64_t splits = gridDim.y, split = blockIdx.y;
    float s = 0.f;
    if (vec) {  // D % 4 == 0 and 16-byte aligned bases: groups of four elements
        const int64_t groups = D >> 2;
        for (int64_t j = split * 256 + threadIdx.x; j < groups; j += splits * 256) {
            float x[4], y[4];
            Vec4<DTX>::load(X, base + 4 * j, x);
            Vec4<DTY>::load(Y, base + 4 * j, y);
            if (W) {
                const float4 w = *reinterpret_cast<const float4*>(W + 4 * j);
                s += x[0] * y[0] * w.x + x[1] * y[1] * w.y + x[2] * y[2] * w.z + x[3] * y[3] * w.w;
            } else {
                s += x[0] * y[0] + x[1] * y[1] + x[2] * y[2] + x[3] * y[3];
            }
        }
    } else {
        for (int64_t i = split * 256 + threadIdx.x; i < D; i += splits * 256) {
            const float t = load_t<DTX>(X, base + i) * load_t<DTY>(Y, base + i);
            s += W ? t * W[i] : t;
        }
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out + row, scale * (part[0] + part[1] + part[2] + part[3]));
}

// Factored score of low-rank queries against SEQUENCE samples (module/linear.py:83-99, "qik,qko,b...i,b...o->qb" contracted as
// (G L_q) . (A' R_q^T)): scores[q, n] += scale * sum_{r < R} sum_{k < K} U[(n R + r), q K + k] V[(n R + r), q K + k] for bf16
// U, V = [b R, Q K] row-major (two tall GEMMs made them).  A pure HBM stream: a thread owns 8 consecutive columns (16 bytes) of
// one sample's rows, a workgroup 2048 columns; partial sums of the K / 8 lanes of a query are folded with shuffles when K / 8 is
// a power of two <= 64, else every lane adds its own partial.
struct SegDotArgs {
    float* scores; int64_t ld_scores;
    const uint16_t* U; const uint16_t* V;
    int64_t ld;       // Q * K
    int R, K, rsplit; float scale;
};

__global__ __launch_bounds__(256) void lowrank_rows_dot_kernel(SegDotArgs a) {
    const int64_t col = (static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x) * 8;
    const int n = blockIdx.y, part = blockIdx.z;
    const bool live = col < a.ld;
    const int r_per = (a.R + a.rsplit - 1) / a.rsplit, r_begin = part * r_per, r_end = min(a.R, r_begin + r_per);
    float s = 0.0f;
    if (live) {
        const uint16_t* u = a.U + (static_cast<int64_t>(n) * a.R + r_begin) * a.ld + col;
        const uint16_t* v = a.V + (static_cast<int64_t>(n) * a.R + r_begin) * a.ld + col;
        for (int r = r_begin; r < r_end; ++r, u += a.ld, v += a.ld) {
            const uint4 x = *reinterpret_cast<const uint4*>(u), y = *reinterpret_cast<const uint4*>(v);
            const uint32_t xs[4] = {x.x, x.y, x.z, x.w}, ys[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s = fmaf(__uint_as_float(xs[e] << 16), __uint_as_float(ys[e] << 16), s);
                s = fmaf(__uint_as_float(xs[e] & 0xffff0000u), __uint_as_float(ys[e] & 0xffff0000u), s);
            }
        }
    }
    const int lanes = a.K >> 3;   // lanes per query
    const bool fold = lanes <= 64 && (lanes & (lanes - 1)) == 0;
    if (fold) {
        for (int off = 1; off < lanes; off <<= 1) s += __shfl_xor(s, off, 64);   // (all 64 lanes take part: dead lanes carry 0)
        if (live && (threadIdx.x & (lanes - 1)) == 0) atomicAdd(a.scores + (col / a.K) * a.ld_scores + n, a.scale * s);
    } else if (live) {
        atomicAdd(a.scores + (col / a.K) * a.ld_scores + n, a.scale * s);
    }
}

template <int DTX>
__global__ __launch_bounds__(256) void mul_bcast_kernel(float* out, const void* X, const float* M, int64_t rows, int64_t D,
                                                        float scale) {
    const int64_t total = rows * D;
    for (int64_t e = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; e < total;
         e += static_cast<int64_t>(gridDim.x) * blockDim.x)
        out[e] = scale * load_t<DTX>(X, e) * M[e % D];
}

// ------------------------------------------------------------------------------------------------
// cast
// ------------------------------------------------------------------------------------------------
__global__ void cast_kernel(void* dst, int dd, const void* src, int sd, int64_t n) {
    for (int64_t i = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; i < n;
         i += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        if (sd == F64 && dd == F64) reinterpret_cast<double*>(dst)[i] = reinterpret_cast<const double*>(src)[i];
        else if (sd == F64) store_as(dst, dd, i, static_cast<float>(reinterpret_cast<const double*>(src)[i]));
        else if (dd == F64) reinterpret_cast<double*>(dst)[i] = static_cast<double>(load_f32(src, sd, i));
        else store_as(dst, dd, i, load_f32(src, sd, i));
    }
}

// One-time kernel attribute set-up, thread-safe (the header promises re-entrancy).
int configure_kernels() {
    static std::once_flag flag;
    static int status = KF_OK;
    std::call_once(flag, [] {
        const bool ok =
            hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, HSMEM_BYTES) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, HSMEM_BYTES) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, HSMEM_BYTES) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(lambda_bf16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, HSMEM_BYTES) == hipSuccess &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(im2col_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024) == hipSuccess;
        if (!ok) status = KF_ERR_LAUNCH_FAILED;
    });
    return status;
}

inline unsigned stream_grid(int64_t n) { return static_cast<unsigned>(std::max<int64_t>(1, std::min<int64_t>(cdiv(n, 256), 2048))); }

kf_view make_view(const void* p, int dtype, int64_t bs, int64_t rs, int64_t ks, int64_t rows, int64_t depth,
                  int ones_row = 0, int ones_k = 0, int square = 0) {
    kf_view v;
    v.p = p; v.dtype = dtype; v.batch_stride = bs; v.row_stride = rs; v.k_stride = ks;
    v.rows = rows; v.depth = depth; v.ones_row = ones_row; v.ones_k = ones_k; v.square = square;
    v.k_tile_stride = 0;
    return v;
}

}  // namespace

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

int kf_abi_version(void) { return 14; }

const char* kf_status_string(int s) {
    switch (s) {
        case KF_OK: return "ok";
        case KF_ERR_INVALID_ARGUMENT: return "invalid argument";
        case KF_ERR_UNSUPPORTED_DTYPE: return "unsupported dtype";
        case KF_ERR_LAUNCH_FAILED: return "kernel launch failed";
        case KF_ERR_WORKSPACE_TOO_SMALL: return "workspace too small";
        case KF_ERR_NOT_CONVERGED: return "eigensolver did not converge";
        case KF_ERR_NO_DEVICE: return "no HIP device";
        default: return "unknown status";
    }
}

int kf_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return KF_ERR_NO_DEVICE; }
    return n;
}

int kf_syrk_accum(float* C, int64_t ldc, const void* X, int in_dtype, int64_t n_rows, int64_t d_in,
                  int64_t rows_inner, int64_t outer_stride, int64_t row_stride, int64_t col_stride,
                  const void* mask, int mask_dtype, int append_ones, float alpha, int64_t* count, void* stream) {
    if (!C || !X || n_rows < 0 || d_in <= 0 || rows_inner <= 0) return KF_ERR_INVALID_ARGUMENT;
    if (in_dtype != KF_F32 && in_dtype != KF_BF16 && in_dtype != KF_F16) return KF_ERR_UNSUPPORTED_DTYPE;
    if (mask && mask_dtype != KF_F32 && mask_dtype != KF_I64 && mask_dtype != KF_U8) return KF_ERR_UNSUPPORTED_DTYPE;
    if (n_rows >= (1LL << 31) - BK) return KF_ERR_INVALID_ARGUMENT;
    hipStream_t st = as_stream(stream);
    const int64_t d = d_in + (append_ones ? 1 : 0);
    if (count) {
        if (mask) hipLaunchKernelGGL(count_kernel, dim3(1), dim3(256), 0, st, count, mask, mask_dtype, n_rows);
        else hipLaunchKernelGGL(add_count_kernel, dim3(1), dim3(1), 0, st, count, n_rows);
    }
    if (n_rows == 0) return launch_status();
    // bf16 rows without mask / bias column: the TN bf16 MFMA engine in symmetric (upper-triangle) mode
    if (in_dtype == KF_BF16 && !mask && !append_ones && rows_inner >= n_rows && col_stride == 1 && row_stride % 8 == 0 &&
        d_in % 8 == 0 && d_in > 1 && (reinterpret_cast<uintptr_t>(X) & 15) == 0) {
        kf_view v = make_view(X, KF_BF16, 0, 1, row_stride, d_in, n_rows);
        return launch_gemm_bf16(2, C, KF_F32, ldc, 0, v, v, 1, alpha, 1.0f, st, 0, true);
    }
    SyrkArgs a;
    a.C = C; a.ldc = ldc;
    a.base.p = X; a.base.n_rows = n_rows; a.base.rows_inner = rows_inner;
    a.base.outer_stride = outer_stride; a.base.row_stride = row_stride; a.base.col_stride = col_stride;
    a.base.mask = mask; a.base.mask_dtype = mask_dtype; a.base.d_in = static_cast<int>(d_in); a.base.dtype = in_dtype;
    a.base.d = static_cast<int>(d);
    a.base.contig_k = (col_stride != 1);  // lanes walk the contiguous memory direction
    a.tiles = static_cast<int>(cdiv(d, BM));
    const int64_t pairs = static_cast<int64_t>(a.tiles) * (a.tiles + 1) / 2;
    const int64_t ksteps = cdiv(n_rows, BK);
    int64_t ksplit = 1;
    if (pairs < 1024 && ksteps >= 8) ksplit = std::max<int64_t>(1, std::min<int64_t>(cdiv(1024, pairs), ksteps / 4));
    a.kchunk = cdiv(ksteps, ksplit) * BK;
    ksplit = cdiv(n_rows, a.kchunk);
    a.ksplit = static_cast<int>(ksplit);
    a.alpha = alpha; a.atomic = ksplit > 1;
    if (ksplit > 65535) return KF_ERR_INVALID_ARGUMENT;
    const dim3 grid(static_cast<unsigned>(pairs), static_cast<unsigned>(ksplit));
    const int md = mask ? mask_dtype : -1;
#define KF_SYRK_CASE(DT, MDT) hipLaunchKernelGGL((syrk_kernel<DT, MDT>), grid, dim3(NTHREADS), 0, st, a)
#define KF_SYRK_MASKS(DT)                                   \
    do {                                                    \
        if (md == -1) KF_SYRK_CASE(DT, -1);                 \
        else if (md == KF_F32) KF_SYRK_CASE(DT, F32);       \
        else if (md == KF_I64) KF_SYRK_CASE(DT, I64);       \
        else KF_SYRK_CASE(DT, U8);                          \
    } while (0)
    if (in_dtype == KF_F32) KF_SYRK_MASKS(F32);
    else if (in_dtype == KF_BF16) KF_SYRK_MASKS(BF16);
    else KF_SYRK_MASKS(F16);
#undef KF_SYRK_MASKS
#undef KF_SYRK_CASE
    return launch_status();
}

int kf_conv2d_cov_small(float* Cov, int64_t ldc, const void* x, int x_dtype, int64_t b, int64_t C, int64_t H, int64_t W, int k1, int k2,
                        int s1, int s2, int p1, int p2, int d1, int d2, int append_ones, float alpha, void* stream) {
    if (!Cov || !x || b < 0 || C <= 0 || H <= 0 || W <= 0 || k1 <= 0 || k2 <= 0 || s1 <= 0 || s2 <= 0 || d1 <= 0 || d2 <= 0 || p1 < 0 || p2 < 0)
        return KF_ERR_INVALID_ARGUMENT;
    if (x_dtype != KF_F32 && x_dtype != KF_BF16 && x_dtype != KF_F16) return KF_ERR_UNSUPPORTED_DTYPE;
    ConvCovSmallArgs a;
    a.C = Cov; a.ldc = ldc; a.x = x; a.b = b; a.Cin = C; a.H = H; a.W = W;
    a.k1 = k1; a.k2 = k2; a.s1 = s1; a.s2 = s2; a.p1 = p1; a.p2 = p2; a.d1 = d1; a.d2 = d2;
    a.O1 = (H + 2 * p1 - d1 * (k1 - 1) - 1) / s1 + 1;
    a.O2 = (W + 2 * p2 - d2 * (k2 - 1) - 1) / s2 + 1;
    if (a.O1 <= 0 || a.O2 <= 0) return KF_ERR_INVALID_ARGUMENT;
    const int64_t D = C * k1 * k2;
    a.ones = append_ones ? 1 : 0;
    if (D + a.ones > 32 || ldc < D + a.ones) return KF_ERR_INVALID_ARGUMENT;   // wider patches: kf_conv2d_cov_accum / kf_im2col + kf_syrk_accum
    a.D = static_cast<int>(D);
    a.npos = b * a.O1 * a.O2;
    if (a.npos >= (1LL << 40)) return KF_ERR_INVALID_ARGUMENT;
    if (a.npos == 0) return KF_OK;
    a.alpha = alpha;
    // one workgroup per CU at most; at least 64 position pairs per wave
    const unsigned grid = static_cast<unsigned>(std::max<int64_t>(1, std::min<int64_t>(256, cdiv(a.npos, 2 * 64 * 4))));
    hipStream_t st = as_stream(stream);
    if (x_dtype == KF_F32) hipLaunchKernelGGL(conv_cov_small_kernel<F32>, dim3(grid), dim3(256), 0, st, a);
    else if (x_dtype == KF_BF16) hipLaunchKernelGGL(conv_cov_small_kernel<BF16>, dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(conv_cov_small_kernel<F16>, dim3(grid), dim3(256), 0, st, a);
    return launch_status();
}

int kf_im2col(void* out, int out_dtype, const void* x, int in_dtype, int64_t b, int64_t C, int64_t H, int64_t W,
              int k1, int k2, int s1, int s2, int p1, int p2, int d1, int d2, int groups, int append_ones,
              void* stream) {
    if (!out || !x || b < 0 || C <= 0 || groups <= 0 || C % groups != 0) return KF_ERR_INVALID_ARGUMENT;
    if (!float_dtype(in_dtype) || !float_dtype(out_dtype)) return KF_ERR_UNSUPPORTED_DTYPE;
    Im2colArgs a;
    a.out = out; a.out_dtype = out_dtype; a.x = x; a.in_dtype = in_dtype; a.b = b; a.C = C; a.H = H; a.W = W;
    a.k1 = k1; a.k2 = k2; a.s1 = s1; a.s2 = s2; a.p1 = p1; a.p2 = p2; a.d1 = d1; a.d2 = d2;
    a.groups = groups; a.append_ones = append_ones ? 1 : 0;
    a.O1 = (H + 2 * p1 - d1 * (k1 - 1) - 1) / s1 + 1;
    a.O2 = (W + 2 * p2 - d2 * (k2 - 1) - 1) / s2 + 1;
    if (a.O1 <= 0 || a.O2 <= 0) return KF_ERR_INVALID_ARGUMENT;
    a.Cg = C / groups;
    a.Ip = a.Cg * k1 * k2 + a.append_ones;
    const int64_t total = b * a.O1 * a.O2 * a.Ip;
    if (total == 0) return KF_OK;
    const bool two_byte = out_dtype == KF_BF16 || out_dtype == KF_F16;
    if (two_byte && groups == 1 && a.Ip % 8 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 && total < (1LL << 40)) {
        if (in_dtype == out_dtype && !append_ones && W % 8 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0) {
            // largest band of output rows whose input footprint fits 64 KB of LDS (two workgroups per CU)
            Im2colLdsArgs la;
            la.a = a;
            la.Wp = static_cast<int>(W) + 2 * p2;
            const int64_t row_bytes = C * la.Wp * 2;
            int rb = 0;
            for (int cand = static_cast<int>(a.O1); cand >= 1; --cand) {
                const int64_t nr = static_cast<int64_t>(cand - 1) * s1 + static_cast<int64_t>(k1 - 1) * d1 + 1;
                if (nr * row_bytes <= 64 * 1024) { rb = cand; break; }
            }
            if (rb > 0) {
                // at least ~2 workgroups per CU: split the image further when the batch is small
                while (rb > 1 && b * ((a.O1 + rb - 1) / rb) < 512) rb = (rb + 1) / 2;
                la.RB = rb;
                la.NR = (rb - 1) * s1 + (k1 - 1) * d1 + 1;
                const size_t lds = static_cast<size_t>(la.NR) * row_bytes;
                if (configure_kernels() != KF_OK) return KF_ERR_LAUNCH_FAILED;
                const int64_t blocks = b * ((a.O1 + rb - 1) / rb);
                hipLaunchKernelGGL(im2col_lds_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), lds, as_stream(stream), la);
                return launch_status();
            }
        }
        hipLaunchKernelGGL(im2col_vec8_kernel, dim3(stream_grid(total / 8) * 4), dim3(256), 0, as_stream(stream), a);
        return launch_status();
    }
    hipLaunchKernelGGL(im2col_kernel, dim3(stream_grid(total)), dim3(256), 0, as_stream(stream), a);
    return launch_status();
}

int kf_gemm(float* C, int64_t ldc, int64_t c_batch_stride, const kf_view* A, const kf_view* B, int64_t batch,
            float alpha, float beta, const float* mul, int64_t ld_mul, void* stream) {
    if (!A || !B) return KF_ERR_INVALID_ARGUMENT;
    return launch_gemm(C, ldc, c_batch_stride, *A, *B, batch, alpha, beta, mul, ld_mul, as_stream(stream));
}

int kf_gemm_out(void* C, int c_dtype, int64_t ldc, int64_t c_batch_stride, const kf_view* A, const kf_view* B, int64_t batch,
                float alpha, void* stream) {
    if (!A || !B) return KF_ERR_INVALID_ARGUMENT;
    if (c_batch_stride == 0 && batch > 1) return KF_ERR_INVALID_ARGUMENT;
    return launch_gemm(C, ldc, c_batch_stride, *A, *B, batch, alpha, 0.0f, nullptr, 0, as_stream(stream), c_dtype);
}

int kf_gemm_bias_out(void* C, int64_t ldc, const kf_view* A, const kf_view* B, const float* bias, int64_t bias_n, void* stream) {
    if (!C || !A || !B || !bias || bias_n < 0 || bias_n > B->rows) return KF_ERR_INVALID_ARGUMENT;
    if (A->rows <= 0 || B->rows <= 0) return KF_OK;
    if (A->depth != B->depth || bf16_engine_mode(*A, *B, nullptr) != 1 || (reinterpret_cast<uintptr_t>(C) & 15) != 0 || ldc % 8 != 0)
        return KF_ERR_INVALID_ARGUMENT;
    HalfExtras ex;
    ex.row_add = bias; ex.row_add_n = static_cast<int>(bias_n);
    return launch_gemm_bf16(1, C, KF_BF16, ldc, 0, *A, *B, 1, 1.0f, 0.0f, as_stream(stream), 0, false, &ex);
}

int kf_lambda_accum(float* Lambda, int64_t ld_lambda, const void* Gtv, const void* Atv, int64_t ld_at, int dtype, int64_t b, int64_t R,
                    int64_t O, int64_t Ip, float scale, void* stream) {
    if (!Lambda || !Gtv || !Atv || b < 0 || R <= 0 || O <= 0 || Ip <= 0 || ld_at < Ip) return KF_ERR_INVALID_ARGUMENT;
    if (dtype != KF_BF16 && ld_at != Ip) return KF_ERR_INVALID_ARGUMENT;
    if (dtype != KF_F32 && dtype != KF_BF16) return KF_ERR_UNSUPPORTED_DTYPE;
    if (b == 0) return KF_OK;
    hipStream_t st = as_stream(stream);
    if (dtype == KF_BF16) {
        // bf16 rotated factors: TN tiles on the bf16 MFMA engine, squared + summed over samples in registers
        // rows of At are ld_at wide (I' zero-padded to a multiple of 8 by the rotation that produced them)
        if (R == 1 || O % 8 != 0 || ld_at % 8 != 0 || ((reinterpret_cast<uintptr_t>(Gtv) | reinterpret_cast<uintptr_t>(Atv)) & 15) != 0)
            return KF_ERR_INVALID_ARGUMENT;
        if (configure_kernels() != KF_OK) return KF_ERR_LAUNCH_FAILED;
        HalfLambdaArgs la;
        HalfGemmArgs& h = la.g;
        h.C = Lambda; h.c_dtype = KF_F32; h.ldc = ld_lambda; h.c_batch_stride = 0;
        h.A.p = reinterpret_cast<const uint16_t*>(Gtv); h.A.batch_stride = R * O; h.A.ld = O; h.A.kt_stride = 64;
        h.A.rows = static_cast<int>(O); h.A.depth = static_cast<int>(R);
        h.B.p = reinterpret_cast<const uint16_t*>(Atv); h.B.batch_stride = R * ld_at; h.B.ld = ld_at; h.B.kt_stride = 64;
        h.B.rows = static_cast<int>(ld_at); h.B.depth = static_cast<int>(R);
        h.M = static_cast<int>(O); h.N = static_cast<int>(Ip); h.K = static_cast<int>(R);
        h.ksplit = 1; h.kchunk = static_cast<int>(cdiv(R, HBK) * HBK); h.alpha = 1.0f; h.beta = 0.0f; h.atomic = 1;
        h.tiles_m = static_cast<int>(cdiv(O, 128)); h.tiles_n = static_cast<int>(cdiv(Ip, 128)); h.chunks = 1;
        h.symmetric = 0; h.c_tile_stride = 0;
        h.row_add = nullptr; h.row_add_n = 0; h.mul = nullptr; h.ld_mul = 0; h.mul_n = 0;
        const int64_t tiles = static_cast<int64_t>(h.tiles_m) * h.tiles_n;
        int64_t zsplit = std::max<int64_t>(1, std::min<int64_t>(b, cdiv(1024, tiles)));
        la.zchunk = static_cast<int>(cdiv(b, zsplit));
        zsplit = cdiv(b, la.zchunk);
        la.zblocks = static_cast<int>(zsplit);
        la.batch = static_cast<int>(b); la.scale2 = scale * scale;
        const int64_t blocks = 8 * cdiv(zsplit * tiles, 8);
        if (blocks >= (1LL << 31)) return KF_ERR_INVALID_ARGUMENT;
        hipLaunchKernelGGL(lambda_bf16_kernel, dim3(static_cast<unsigned>(blocks)), dim3(NTHREADS), HSMEM_BYTES, st, la);
        return launch_status();
    }
    const float* Gt = reinterpret_cast<const float*>(Gtv);
    const float* At = reinterpret_cast<const float*>(Atv);
    if (R == 1) {
        // Lambda += scale^2 * (Gt o Gt)^T (At o At): one GEMM over the batch dimension
        kf_view A = make_view(Gt, KF_F32, 0, 1, O, O, b, 0, 0, 1);
        kf_view B = make_view(At, KF_F32, 0, 1, Ip, Ip, b, 0, 0, 1);
        return launch_gemm(Lambda, ld_lambda, 0, A, B, 1, scale * scale, 1.0f, nullptr, 0, st);
    }
    LambdaArgs a;
    a.L = Lambda; a.ldl = ld_lambda; a.Gt = Gt; a.At = At;
    a.b = static_cast<int>(b); a.R = static_cast<int>(R); a.O = static_cast<int>(O); a.Ip = static_cast<int>(Ip);
    a.scale2 = scale * scale;
    const int64_t tiles = cdiv(O, BM) * cdiv(Ip, BN);
    int64_t zsplit = std::max<int64_t>(1, std::min<int64_t>(b, cdiv(1024, tiles)));
    a.zchunk = static_cast<int>(cdiv(b, zsplit));
    zsplit = cdiv(b, a.zchunk);
    if (zsplit > 65535) return KF_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(lambda_kernel, dim3(static_cast<unsigned>(cdiv(Ip, BN)), static_cast<unsigned>(cdiv(O, BM)), static_cast<unsigned>(zsplit)),
                       dim3(NTHREADS), 0, st, a);
    return launch_status();
}

int kf_inv_lambda(float* out, const float* Lambda, int64_t numel, double n_lambda, double damping, void* workspace,
                  void* stream) {
    if (!out || !Lambda || numel < 0 || n_lambda <= 0.0 || !workspace) return KF_ERR_INVALID_ARGUMENT;
    if (numel == 0) return KF_OK;
    hipStream_t st = as_stream(stream);
    double* sum = reinterpret_cast<double*>(workspace);
    if (damping < 0.0) {
        if (hipMemsetAsync(sum, 0, sizeof(double), st) != hipSuccess) return KF_ERR_LAUNCH_FAILED;
        hipLaunchKernelGGL(sum_f64_kernel, dim3(stream_grid(numel)), dim3(256), 0, st, sum, Lambda, numel);
    }
    hipLaunchKernelGGL(inv_lambda_kernel, dim3(stream_grid(numel)), dim3(256), 0, st, out, Lambda, numel, n_lambda, damping, sum);
    return launch_status();
}

namespace {
// The bf16 form of the preconditioner (ScoreArguments.precondition_dtype = bf16): the augmented axis is carried at width
// ldq = I' rounded up to a multiple of 8 -- the copies Qa_bf16 / QaT_bf16 are [ldq, ldq] with zero padding, P has row stride
// ldp == ldq and zero padding columns -- so layers with an odd I' (every Linear with bias on sequences: BERT, GPT-2) stay on the
// bf16 engine; the bias column "[A, 1] Qa = A Qa[:I] + Qa[I]" is a row (bias_row, fp32) added in the epilogue.
bool precondition_low_eligible(const void* Pout, const void* G, const void* A, int out_dtype, int in_dtype, int64_t R, int64_t O, int64_t I,
                               int64_t Ip, int64_t ldp, const void* Qa_bf16, const void* QgT_bf16, const void* QaT_bf16, int64_t ldq) {
    return Qa_bf16 && QgT_bf16 && QaT_bf16 && out_dtype == KF_BF16 && in_dtype == KF_BF16 && R > 1 && O % 8 == 0 && I % 8 == 0 &&
           I >= HBK && O >= HBK && ldq % 8 == 0 && ldq >= Ip && ldp == ldq &&
           ((reinterpret_cast<uintptr_t>(Qa_bf16) | reinterpret_cast<uintptr_t>(QgT_bf16) | reinterpret_cast<uintptr_t>(QaT_bf16) |
             reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(G) | reinterpret_cast<uintptr_t>(Pout)) & 15) == 0;
}

// Qg (fp32) or Qg_bf16 (the same matrix in bf16; preferred when given): the back rotation of the round-3 form needs Qg itself
int precondition_low(void* Pout, int64_t ldp, const void* G, const void* A, int64_t q, int64_t R, int64_t O, int64_t I, int append_ones,
                     const float* Qg, const void* Qg_bf16, const float* bias_row, const float* inv_lambda, float scale, const void* Qa_bf16,
                     const void* QgT_bf16, const void* QaT_bf16, int64_t ldq, void* workspace, void* stream) {
    hipStream_t st = as_stream(stream);
    const int64_t Ip = I + (append_ones ? 1 : 0);
    int rc;
    if (ldq == (Ip + 7) / 8 * 8 && precondition_v3_eligible(q, R, O, I, ldq))
        return precondition_v3(Pout, G, A, q, R, O, I, append_ones, Qg, Qg_bf16, bias_row, Ip, inv_lambda, scale, Qa_bf16, QgT_bf16, QaT_bf16, ldq,
                               workspace, stream);
    {
        const int64_t W = ldq;
        uint16_t* Gt16 = reinterpret_cast<uint16_t*>(workspace);   // [q R, O]
        uint16_t* At16 = Gt16 + ((q * R * O + 127) & ~127LL);        // [q R, W]
        uint16_t* rot16 = At16 + ((q * R * W + 127) & ~127LL);       // [q, O, W]
        uint16_t* T16 = rot16 + ((q * O * W + 127) & ~127LL);        // [q, O, W]
        // Gt[(q r), o'] = sum_o G[(q r), o] Qg[o, o']                (NT: B[n, k] = QgT[n, k])
        rc = launch_gemm_bf16(1, Gt16, KF_BF16, O, 0, make_view(G, KF_BF16, 0, O, 1, q * R, O), make_view(QgT_bf16, KF_BF16, 0, O, 1, O, O), 1,
                              1.0f, 0.0f, st, 0);
        if (rc != KF_OK) return rc;
        // At[(q r), i'] = sum_i A[(q r), i] Qa[i, i'] (+ Qa[I, i'])  (NT over the I real columns; pad columns come out zero)
        HalfExtras bias;
        if (append_ones) { bias.row_add = bias_row; bias.row_add_n = static_cast<int>(Ip); }
        rc = launch_gemm_bf16(1, At16, KF_BF16, W, 0, make_view(A, KF_BF16, 0, I, 1, q * R, I), make_view(QaT_bf16, KF_BF16, 0, W, 1, W, I), 1,
                              1.0f, 0.0f, st, 0, false, &bias);
        if (rc != KF_OK) return rc;
        // rot[q][o, i] = (sum_r Gt[q, r, o] At[q, r, i]) * inv_lambda[o, i]   (TN, batched over q; zero in the pad columns)
        HalfExtras lam;
        lam.mul = inv_lambda; lam.ld_mul = Ip; lam.mul_n = static_cast<int>(Ip);
        rc = launch_gemm_bf16(2, rot16, KF_BF16, W, O * W, make_view(Gt16, KF_BF16, R * O, 1, O, O, R), make_view(At16, KF_BF16, R * W, 1, W, W, R), q,
                              1.0f, 0.0f, st, 0, false, &lam);
        if (rc != KF_OK) return rc;
        // T[(q o), j] = sum_i rot[(q o), i] Qa[j, i]                 (NT)
        rc = launch_gemm_bf16(1, T16, KF_BF16, W, 0, make_view(rot16, KF_BF16, 0, W, 1, q * O, W), make_view(Qa_bf16, KF_BF16, 0, W, 1, W, W), 1,
                              1.0f, 0.0f, st, 0);
        if (rc != KF_OK) return rc;
        // P[q][m, n] = scale * sum_o QgT[o, m] T[q][o, n]            (TN, batched over q)
        return launch_gemm_bf16(2, Pout, KF_BF16, ldp, O * ldp, make_view(QgT_bf16, KF_BF16, 0, 1, O, O, O), make_view(T16, KF_BF16, O * W, 1, W, W, O), q,
                                scale, 0.0f, st, 0);
    }
}
}  // namespace

int64_t kf_precondition_workspace_bytes(int64_t q, int64_t R, int64_t O, int64_t Ip) {
    // Gt, At, T and (for low-precision outputs) the fp32 staging copy of the rotated gradient; I' rounded up to the
    // padded width the bf16 path may use
    const int64_t Ipp = (Ip + 7) / 8 * 8;
    const int64_t staged = static_cast<int64_t>(sizeof(float)) * (q * R * O + q * R * Ipp + 2 * q * O * Ipp) + 1024;
    return std::max(staged, precondition_v3_workspace_bytes(q, R, O, Ipp));
}

int kf_precondition(void* Pout, int out_dtype, int64_t ldp, const void* G, const void* A, int in_dtype, int64_t q, int64_t R, int64_t O,
                    int64_t I, int append_ones, const float* Qg, const float* Qa, const float* inv_lambda, float scale,
                    const void* Qa_bf16, const void* QgT_bf16, const void* QaT_bf16, int64_t ldq, void* workspace,
                    int64_t workspace_bytes, void* stream) {
    if (!Pout || !G || !A || !Qg || !Qa || !inv_lambda || q < 0 || R <= 0 || O <= 0 || I <= 0) return KF_ERR_INVALID_ARGUMENT;
    if (!float_dtype(in_dtype) || (out_dtype != KF_F32 && out_dtype != KF_BF16)) return KF_ERR_UNSUPPORTED_DTYPE;
    const int64_t Ip = I + (append_ones ? 1 : 0);
    if (ldp < Ip) return KF_ERR_INVALID_ARGUMENT;
    if (!workspace || workspace_bytes < kf_precondition_workspace_bytes(q, R, O, Ip)) return KF_ERR_WORKSPACE_TOO_SMALL;
    if (q == 0) return KF_OK;
    hipStream_t st = as_stream(stream);
    // precondition_dtype = bf16 (reference low-precision preset): every O(q ..) contraction runs on the bf16 MFMA engine
    // from bf16 copies of the eigenvectors (fp32 accumulation): precondition_low below.
    if (precondition_low_eligible(Pout, G, A, out_dtype, in_dtype, R, O, I, Ip, ldp, Qa_bf16, QgT_bf16, QaT_bf16, ldq))
        return precondition_low(Pout, ldp, G, A, q, R, O, I, append_ones, Qg, nullptr, append_ones ? Qa + I * Ip : nullptr, inv_lambda, scale,
                                Qa_bf16, QgT_bf16, QaT_bf16, ldq, workspace, stream);
    int rc;
    if (ldp != Ip) return KF_ERR_INVALID_ARGUMENT;  // the fp32 path writes compact rows
    float* Gt = reinterpret_cast<float*>(workspace);
    float* At = Gt + q * R * O;
    float* T = At + q * R * Ip;
    // fp32 staging of the rotated gradient: the caller's buffer when it is fp32, else workspace
    float* P = out_dtype == KF_F32 ? reinterpret_cast<float*>(Pout) : T + q * O * Ip;
    // Gt[(q r), o'] = sum_o G[(q r), o] Qg[o, o']
    rc = launch_gemm(Gt, O, 0, make_view(G, in_dtype, 0, O, 1, q * R, O), make_view(Qg, KF_F32, 0, 1, O, O, O), 1, 1.0f, 0.0f, nullptr, 0, st);
    if (rc != KF_OK) return rc;
    // At[(q r), i'] = sum_i [A,1][(q r), i] Qa[i, i']
    rc = launch_gemm(At, Ip, 0, make_view(A, in_dtype, 0, I, 1, q * R, I, 0, append_ones ? 1 : 0),
                     make_view(Qa, KF_F32, 0, 1, Ip, Ip, Ip), 1, 1.0f, 0.0f, nullptr, 0, st);
    if (rc != KF_OK) return rc;
    // rot[q][o,i] = (sum_r Gt[q,r,o] At[q,r,i]) * inv_lambda[o,i]   (stored in the caller's P buffer)
    rc = launch_gemm(P, Ip, O * Ip, make_view(Gt, KF_F32, R * O, 1, O, O, R), make_view(At, KF_F32, R * Ip, 1, Ip, Ip, R), q, 1.0f, 0.0f, inv_lambda, Ip, st);
    if (rc != KF_OK) return rc;
    // T[(q o), j] = sum_i rot[(q o), i] Qa[j, i]
    rc = launch_gemm(T, Ip, 0, make_view(P, KF_F32, 0, Ip, 1, q * O, Ip), make_view(Qa, KF_F32, 0, Ip, 1, Ip, Ip), 1, 1.0f, 0.0f, nullptr, 0, st);
    if (rc != KF_OK) return rc;
    // P[q][m, n] = scale * sum_o Qg[m, o] T[q][o, n]
    rc = launch_gemm(Pout, Ip, O * Ip, make_view(Qg, KF_F32, 0, O, 1, O, O), make_view(T, KF_F32, O * Ip, 1, Ip, Ip, O), q, scale, 0.0f,
                     nullptr, 0, st, out_dtype);
    return rc;
}

int kf_precondition_bf16(void* Pout, int64_t ldp, const void* G, const void* A, int64_t q, int64_t R, int64_t O, int64_t I, int append_ones,
                         const void* Qg_bf16, const void* QgT_bf16, const void* Qa_bf16, const void* QaT_bf16, int64_t ldq,
                         const float* bias_row, const float* inv_lambda, float scale, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!Pout || !G || !A || !Qg_bf16 || !inv_lambda || q < 0 || R <= 0 || O <= 0 || I <= 0 || (append_ones && !bias_row)) return KF_ERR_INVALID_ARGUMENT;
    const int64_t Ip = I + (append_ones ? 1 : 0);
    if ((reinterpret_cast<uintptr_t>(Qg_bf16) & 15) != 0 ||
        !precondition_low_eligible(Pout, G, A, KF_BF16, KF_BF16, R, O, I, Ip, ldp, Qa_bf16, QgT_bf16, QaT_bf16, ldq))
        return KF_ERR_INVALID_ARGUMENT;
    if (!workspace || workspace_bytes < kf_precondition_workspace_bytes(q, R, O, Ip)) return KF_ERR_WORKSPACE_TOO_SMALL;
    if (q == 0) return KF_OK;
    return precondition_low(Pout, ldp, G, A, q, R, O, I, append_ones, nullptr, Qg_bf16, bias_row, inv_lambda, scale, Qa_bf16, QgT_bf16,
                            QaT_bf16, ldq, workspace, stream);
}

int64_t kf_pairwise_workspace_bytes(int64_t b, int64_t R, int64_t O, int64_t Ip) {
    if (R <= 1 && O >= 32) return 16;
    return static_cast<int64_t>(sizeof(float)) * b * O * Ip;
}

int kf_pairwise_score(float* scores, int64_t ld_scores, const void* P, int p_dtype, int64_t p_k_tile_stride, int64_t Q,
                      const void* G, const void* A, int in_dtype, int64_t b, int64_t R, int64_t O, int64_t I, int append_ones,
                      float scale, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!scores || !P || !G || !A || Q < 0 || b < 0 || R <= 0 || O <= 0 || I <= 0) return KF_ERR_INVALID_ARGUMENT;
    if (in_dtype != KF_F32 && in_dtype != KF_BF16 && in_dtype != KF_F16) return KF_ERR_UNSUPPORTED_DTYPE;
    if (p_dtype != KF_F32 && p_dtype != KF_BF16) return KF_ERR_UNSUPPORTED_DTYPE;
    if (Q == 0 || b == 0) return KF_OK;
    hipStream_t st = as_stream(stream);
    const int64_t Ip = I + (append_ones ? 1 : 0);
    if (p_k_tile_stride != 0 && (R == 1 || p_dtype != KF_BF16 || (O * Ip) % 64 != 0 || p_k_tile_stride != Q * 64))
        return KF_ERR_INVALID_ARGUMENT;
    // R == 1: the fused kernel tiles (q, 128 rows of o) x 128 samples; with a handful of output rows (a classifier
    // head) almost the whole tile is padding and one workgroup per query is launched for nothing -- such layers go
    // through the generic "materialise the rank-one gradients, one GEMM over D" path below instead.
    if (R == 1 && O >= 32) {
        ScoreArgs a;
        a.scores = scores; a.ld_scores = ld_scores; a.P = P; a.p_dtype = p_dtype; a.G = G; a.A = A; a.in_dtype = in_dtype;
        a.Q = static_cast<int>(Q); a.b = static_cast<int>(b); a.O = static_cast<int>(O); a.I = static_cast<int>(I);
        a.Ip = static_cast<int>(Ip); a.append_ones = append_ones ? 1 : 0;
        a.tiles_per_q = static_cast<int>(cdiv(O, BM)); a.scale = scale;
        const int64_t gy = Q * a.tiles_per_q;
        if (gy > 65535) {
            // y-dimension limit: fall through in slabs of queries
            const int64_t qs = 65535 / a.tiles_per_q;
            for (int64_t q0 = 0; q0 < Q; q0 += qs) {
                ScoreArgs s = a;
                s.Q = static_cast<int>(std::min<int64_t>(qs, Q - q0));
                s.P = reinterpret_cast<const char*>(P) + q0 * O * Ip * dtype_size(p_dtype); s.scores = scores + q0 * ld_scores;
                launch_score_r1(s, dim3(static_cast<unsigned>(cdiv(b, BN)), static_cast<unsigned>(s.Q * a.tiles_per_q)), st);
            }
            return launch_status();
        }
        launch_score_r1(a, dim3(static_cast<unsigned>(cdiv(b, BN)), static_cast<unsigned>(gy)), st);
        return launch_status();
    }
    if (!workspace || workspace_bytes < kf_pairwise_workspace_bytes(b, R, O, Ip)) return KF_ERR_WORKSPACE_TOO_SMALL;
    // psg[n][o, i] = sum_r G[n,r,o] A'[n,r,i], stored in P's dtype: with bf16 P the contraction below runs on
    // the bf16 MFMA engine (fp32 accumulation), otherwise on the fp32 one.
    void* psg = workspace;
    const int64_t D = O * Ip;
    const bool tiled = p_k_tile_stride != 0;  // both big-K operands k-tile-major: [D/64][rows][64]
    int rc = launch_gemm(psg, Ip, tiled ? 0 : D, make_view(G, in_dtype, R * O, 1, O, O, R),
                         make_view(A, in_dtype, R * I, 1, I, I, R, append_ones ? 1 : 0, 0), b, 1.0f, 0.0f, nullptr, 0, st, p_dtype,
                         tiled ? b * 64 : 0);
    if (rc != KF_OK) return rc;
    // scores[q, n] += scale * sum_d P[q, d] psg[n, d]
    if (tiled) return score_gemm_tiled(scores, ld_scores, P, psg, Q, b, D, scale, stream);  // 256 x 256-tile LDS-DMA kernel
    kf_view vp = make_view(P, p_dtype, 0, tiled ? 64 : D, 1, Q, D);
    kf_view vg = make_view(psg, p_dtype, 0, tiled ? 64 : D, 1, b, D);
    if (tiled) { vp.k_tile_stride = Q * 64; vg.k_tile_stride = b * 64; }
    return launch_gemm(scores, ld_scores, 0, vp, vg, 1, scale, 1.0f, nullptr, 0, st);
}

int kf_rowwise_dot(float* out, const void* X, int x_dtype, const void* Y, int y_dtype, const float* W, int64_t rows,
                   int64_t D, float scale, int accumulate, void* stream) {
    if (!out || !X || !Y || rows < 0 || D < 0) return KF_ERR_INVALID_ARGUMENT;
    if ((x_dtype != KF_F32 && x_dtype != KF_BF16) || (y_dtype != KF_F32 && y_dtype != KF_BF16)) return KF_ERR_UNSUPPORTED_DTYPE;
    if (rows == 0) return KF_OK;
    hipStream_t st = as_stream(stream);
    if (!accumulate && hipMemsetAsync(out, 0, sizeof(float) * rows, st) != hipSuccess) return KF_ERR_LAUNCH_FAILED;
    if (D == 0) return KF_OK;
    const auto aligned = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    const int vec = (D % 8 == 0) && aligned(X) && aligned(Y) && (!W || aligned(W));
    // enough blocks to fill 256 CUs even for a handful of rows, without splitting short rows
    int64_t splits = std::max<int64_t>(1, std::min<int64_t>(cdiv(2048, rows), cdiv(D, 256 * 16)));
    const dim3 grid(static_cast<unsigned>(rows), static_cast<unsigned>(splits));
    if (x_dtype == KF_F32 && y_dtype == KF_F32)
        hipLaunchKernelGGL((rowwise_dot_kernel<F32, F32>), grid, dim3(256), 0, st, out, X, Y, W, D, scale, vec);
    else if (x_dtype == KF_BF16 && y_dtype == KF_F32)
        hipLaunchKernelGGL((rowwise_dot_kernel<BF16, F32>), grid, dim3(256), 0, st, out, X, Y, W, D, scale, vec);
    else if (x_dtype == KF_F32 && y_dtype == KF_BF16)
        hipLaunchKernelGGL((rowwise_dot_kernel<F32, BF16>), grid, dim3(256), 0, st, out, X, Y, W, D, scale, vec);
    else
        hipLaunchKernelGGL((rowwise_dot_kernel<BF16, BF16>), grid, dim3(256), 0, st, out, X, Y, W, D, scale, vec);
    return launch_status();
}

int kf_lowrank_rows_dot(float* scores, int64_t ld_scores, const void* U, const void* V, int64_t b, int64_t R, int64_t Q, int64_t K,
                        float scale, void* stream) {
    if (!scores || !U || !V || b < 0 || R <= 0 || Q < 0 || K <= 0) return KF_ERR_INVALID_ARGUMENT;
    if (K % 8 != 0 || ((reinterpret_cast<uintptr_t>(U) | reinterpret_cast<uintptr_t>(V)) & 15) != 0 || b > 65535 || R >= (1LL << 31))
        return KF_ERR_INVALID_ARGUMENT;
    if (b == 0 || Q == 0) return KF_OK;
    SegDotArgs a;
    a.scores = scores; a.ld_scores = ld_scores; a.U = reinterpret_cast<const uint16_t*>(U); a.V = reinterpret_cast<const uint16_t*>(V);
    a.ld = Q * K; a.R = static_cast<int>(R); a.K = static_cast<int>(K); a.scale = scale;
    const int64_t col_blocks = cdiv(a.ld, 2048);
    // >= ~4 workgroups per CU: split the rows of a sample when (column blocks x samples) alone would not fill the chip
    a.rsplit = static_cast<int>(std::max<int64_t>(1, std::min<int64_t>({cdiv(1024, col_blocks * b), R / 8, static_cast<int64_t>(64)})));
    if (col_blocks >= (1LL << 31)) return KF_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(lowrank_rows_dot_kernel, dim3(static_cast<unsigned>(col_blocks), static_cast<unsigned>(b), static_cast<unsigned>(a.rsplit)),
                       dim3(256), 0, as_stream(stream), a);
    return launch_status();
}

int kf_mul_bcast(float* out, const void* X, int x_dtype, const float* M, int64_t rows, int64_t D, float scale, void* stream) {
    if (!out || !X || !M || rows < 0 || D <= 0) return KF_ERR_INVALID_ARGUMENT;
    if (x_dtype != KF_F32 && x_dtype != KF_BF16) return KF_ERR_UNSUPPORTED_DTYPE;
    if (rows == 0) return KF_OK;
    hipStream_t st = as_stream(stream);
    if (x_dtype == KF_F32)
        hipLaunchKernelGGL((mul_bcast_kernel<F32>), dim3(stream_grid(rows * D)), dim3(256), 0, st, out, X, M, rows, D, scale);
    else
        hipLaunchKernelGGL((mul_bcast_kernel<BF16>), dim3(stream_grid(rows * D)), dim3(256), 0, st, out, X, M, rows, D, scale);
    return launch_status();
}

int kf_cast(void* dst, int dst_dtype, const void* src, int src_dtype, int64_t numel, void* stream) {
    if (!dst || !src || numel < 0) return KF_ERR_INVALID_ARGUMENT;
    if (!float_dtype(dst_dtype) || !float_dtype(src_dtype)) return KF_ERR_UNSUPPORTED_DTYPE;
    if (numel == 0) return KF_OK;
    hipLaunchKernelGGL(cast_kernel, dim3(stream_grid(numel)), dim3(256), 0, as_stream(stream), dst, dst_dtype, src, src_dtype, numel);
    return launch_status();
}

}  // extern "C"
