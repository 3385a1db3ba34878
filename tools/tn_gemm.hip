// tn_gemm.hip -- stand-alone harness of the K-major main loop (kronfluence_amd/csrc/kf_pingpong_tn.h), not part of the library.
// C[m][n] (fp32, += ) = sum_k A[k][m] B[k][n] for K-major bf16 operands, one workgroup per 256 x 256 tile, for each of the three
// candidate LDS images (kf_tn_map.h): validation of the TN staging / transposing fragment reads and the A/B that picked the image
// the per-sample-gradient and covariance kernels use.  Driver: tn_gemm_test.py (against torch on the same bf16 values).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -I kronfluence_amd/csrc tools/tn_gemm.hip -o tools/libtn_gemm.so
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kf_pingpong_tn.h"

using namespace kf;

struct TnArgs {
    float* C; int64_t ldc;
    const uint16_t* A; const uint16_t* B;   // A[K][M], B[K][N]
    int M, N, KT;                           // KT = K / 64
};

template <int IMG>
__global__ __launch_bounds__(pptn::THREADS) void tn_gemm_kernel(TnArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave >> 2, wn = wave & 3;
    const int tiles_n = (a.N + 255) / 256;
    const int m0 = (blockIdx.x / tiles_n) * 256, n0 = (blockIdx.x % tiles_n) * 256;
    pptn::Sources src;
    pptn::make_sources<IMG>(src, wave, lane,
                       [&](int f) { return a.A + min(m0 + f, a.M - 8); }, static_cast<int64_t>(a.M),
                       [&](int f) { return a.B + min(n0 + f, a.N - 8); }, static_cast<int64_t>(a.N));
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][jn][r] = 0.0f;
    pptn::mainloop<IMG>(acc, sm, src, a.KT, wave, lane, static_cast<int64_t>(a.M) * 64, static_cast<int64_t>(a.N) * 64);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jn = 0; jn < 2; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const int n = n0 + wn * 64 + jn * 32 + (lane & 31);
                if (m < a.M && n < a.N) a.C[static_cast<int64_t>(m) * a.ldc + n] += acc[i][jn][r];
            }
}

template <int IMG>
static int launch(const TnArgs& a, int64_t tiles, hipStream_t st) {
    static bool configured = hipFuncSetAttribute(reinterpret_cast<const void*>(tn_gemm_kernel<IMG>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                 pptn::SMEM_BYTES) == hipSuccess;
    if (!configured) return -2;
    hipLaunchKernelGGL(tn_gemm_kernel<IMG>, dim3(static_cast<unsigned>(tiles)), dim3(pptn::THREADS), pptn::SMEM_BYTES, st, a);
    return static_cast<int>(hipGetLastError());
}

extern "C" int tn_gemm(int image, float* C, int64_t ldc, const void* A, const void* B, int64_t M, int64_t N, int64_t K, void* stream) {
    if (!C || !A || !B || M < 8 || N < 8 || K <= 0 || M % 8 != 0 || N % 8 != 0 || K % 64 != 0) return -1;
    TnArgs a{C, ldc, static_cast<const uint16_t*>(A), static_cast<const uint16_t*>(B), static_cast<int>(M), static_cast<int>(N),
             static_cast<int>(K / 64)};
    const int64_t tiles = ((M + 255) / 256) * ((N + 255) / 256);
    hipStream_t st = static_cast<hipStream_t>(stream);
    return image == 0 ? launch<0>(a, tiles, st) : image == 1 ? launch<1>(a, tiles, st) : image == 2 ? launch<2>(a, tiles, st) : -3;
}
