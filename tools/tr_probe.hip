// tr_probe.hip -- stand-alone probe, not part of the library.
//
// Probes of gfx950's transposing LDS read (ds_read_b64_tr_b16, __builtin_amdgcn_ds_read_tr16_b64_*): what a TN operand path
// for the per-sample-gradient and covariance kernels needs to know before it is written (DESIGN.md section 8, item 1 -- the hooked
// [t][feature] rows are K-major; today two transposed copies per call make them K-contiguous for ds_read_b128 fragments).
//
//   tr_probe_semantics   LDS holds lds[i] = i (16-bit); every lane reads 64 bits at ITS byte address with the transposing read and
//                        reports the four 16-bit values it got: which lane's which element ends up where.
//   tr_probe_cycles      8 waves issue `iters` x 8 reads each at per-lane addresses taken from a table (one row of 64 byte
//                        addresses per variant; the k-th read of the unrolled body adds k * step bytes); s_memtime around the
//                        loop of wave 0.  Modes: 0 ds_read_b64_tr_b16, 1 ds_read_b64, 2 ds_read_b128 -- LDS-array cycles per
//                        read for candidate LDS images of an MFMA operand (bank conflicts of the transposing read are "hardware-
//                        transpose-specific", cdna_hip_programming.md T10: measured, not derived).
//
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/tr_probe.hip -o tools/libtr_probe.so ; driver: tr_probe.py
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

constexpr int LDS_BYTES = 65536;

__global__ __launch_bounds__(64) void semantics_kernel(const int* addr_bytes, uint16_t* out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[LDS_BYTES / 2];
    for (int i = threadIdx.x; i < LDS_BYTES / 2; i += 64) lds[i] = static_cast<uint16_t>(i);
    __syncthreads();
    const int a = addr_bytes[threadIdx.x];
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, reinterpret_cast<unsigned char*>(lds) + a));
#pragma unroll
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = static_cast<uint16_t>(v[j]);
}

template <int MODE>
__global__ __launch_bounds__(512) void cycles_kernel(const int* addr_bytes, int variant, int iters, int step, long long* cycles, int* sink) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[LDS_BYTES / 2];
    for (int i = threadIdx.x; i < LDS_BYTES / 2; i += 512) lds[i] = static_cast<uint16_t>(i);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    unsigned char* base = reinterpret_cast<unsigned char*>(lds) + addr_bytes[variant * 64 + lane];
    int acc = 0;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            unsigned char* p = base + k * step;
            if constexpr (MODE == 0) {
                s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4, p));
                acc += v[0] + v[3];
            } else if constexpr (MODE == 1) {
                i32x2 v = *LDS_PTR(i32x2, p);
                acc += v[0] + v[1];
            } else {
                i32x4 v = *LDS_PTR(i32x4, p);
                acc += v[0] + v[3];
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cycles[variant] = t1 - t0;
    if (acc == 0x7fffffff) sink[0] = acc;   // keeps the reads alive
}

extern "C" int tr_probe_semantics(const int* addr_bytes, uint16_t* out, void* stream) {
    hipLaunchKernelGGL(semantics_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), addr_bytes, out);
    return static_cast<int>(hipGetLastError());
}

extern "C" int tr_probe_cycles(const int* addr_bytes, int variants, int iters, int step, int mode, long long* cycles, int* sink, void* stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int v = 0; v < variants; ++v) {
        if (mode == 0) hipLaunchKernelGGL(cycles_kernel<0>, dim3(1), dim3(512), 0, st, addr_bytes, v, iters, step, cycles, sink);
        else if (mode == 1) hipLaunchKernelGGL(cycles_kernel<1>, dim3(1), dim3(512), 0, st, addr_bytes, v, iters, step, cycles, sink);
        else hipLaunchKernelGGL(cycles_kernel<2>, dim3(1), dim3(512), 0, st, addr_bytes, v, iters, step, cycles, sink);
    }
    return static_cast<int>(hipGetLastError());
}
