/*
 * kronfluence_hip.h -- C ABI of the MI355X (gfx950) EK-FAC hot-path library, libkronfluence_hip.so.
 *
 * This is the drop-in boundary one level below kronfluence's Python plugin interfaces
 * (TrackedModule operator API, Tracker hook API, FactorConfig strategy API).
 * Every entry point replaces the dense-math torch call sites of one reference routine; the
 * reference file:line each one stands in for is cited at its declaration (paths relative to the
 * reference checkout).
 *
 * Conventions
 *   - plain pointers and sizes only; no torch / hip types in signatures ("stream" is a hipStream_t
 *     passed as void*; NULL = the default stream);
 *   - all pointers are DEVICE pointers owned by the caller (PyTorch's allocator in the shipped host
 *     code); the library never allocates, frees or retains caller memory;
 *   - every call is asynchronous on `stream`, never synchronises it (kf_eigh_f64 is the one
 *     documented exception) and is re-entrant;
 *   - return value: KF_OK (0) or a negative kf_status; never throws;
 *   - accumulators ("+=" outputs) are fp32 regardless of the input dtype; the host casts on export;
 *   - matrices are row-major; "ld*" = elements between consecutive rows.
 */
#ifndef KRONFLUENCE_HIP_H
#define KRONFLUENCE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum kf_status {
    KF_OK = 0,
    KF_ERR_INVALID_ARGUMENT = -1,
    KF_ERR_UNSUPPORTED_DTYPE = -2,
    KF_ERR_LAUNCH_FAILED = -3,
    KF_ERR_WORKSPACE_TOO_SMALL = -4,
    KF_ERR_NOT_CONVERGED = -5,
    KF_ERR_NO_DEVICE = -6
} kf_status;

typedef enum kf_dtype {
    KF_F32 = 0,
    KF_BF16 = 1,
    KF_F16 = 2,
    KF_F64 = 3,
    KF_I64 = 4, /* mask dtypes only */
    KF_I32 = 5,
    KF_U8 = 6 /* torch.bool / torch.uint8 */
} kf_dtype;

/* ABI version: bumped on any signature change. */
int kf_abi_version(void);
const char* kf_status_string(int status);
/* Number of visible HIP devices, or a negative kf_status.  The only call that is legal without a GPU. */
int kf_device_count(void);

/* ---------------------------------------------------------------------------------------------
 * Stage 1 -- covariance accumulation
 * ------------------------------------------------------------------------------------------- */

/*
 * C[d,d] += alpha * X'^T X'        (d = d_in + append_ones)
 *
 * X' is the "flattened activation" of module/linear.py:30-46 built on the fly from the hooked
 * tensor: row n of X' is mask[n] * [X[n, 0..d_in), 1] (the ones column only if append_ones; the
 * mask multiplies the ones column too, exactly as linear.py:39-43).  With mask == NULL and
 * append_ones == 0 it is the gradient update of module/tracker/factor.py:93 (alpha =
 * gradient_scale^2, factor.py:90-92); the reference never masks gradients (linear.py:48-54).
 * Replaces: module/tracker/factor.py:58 (addmm_), :93 (addmm_), module/linear.py:33-43 (mask,
 * ones column, cat).
 *
 * Row n lives at X + (n / rows_inner) * outer_stride + (n % rows_inner) * row_stride, element c of
 * it col_stride further on (strides in elements).  A [n,d] matrix is rows_inner = n_rows,
 * row_stride = ld, col_stride = 1; an NCHW output gradient [b,O,P] seen as rows (b,p) x cols o
 * (module/conv2d.py:130-132) is rows_inner = P, outer_stride = O*P, row_stride = 1, col_stride = P.
 *
 * count (nullable, device int64[1]) += sum(mask) if mask else n_rows  (linear.py:45,
 * factor.py:57).  Both triangles of C are updated, C stays symmetric.
 */
int kf_syrk_accum(float* C, int64_t ldc, const void* X, int in_dtype, int64_t n_rows, int64_t d_in,
                  int64_t rows_inner, int64_t outer_stride, int64_t row_stride, int64_t col_stride,
                  const void* mask, int mask_dtype, int append_ones, float alpha, int64_t* count,
                  void* stream);

/*
 * Covariances on the LDS-DMA engine (ABI 8 / 9; bf16 inputs, exact products, fp32 accumulation, both triangles).
 *
 * kf_syrk_rows_bf16: C[d,d] += alpha * X'^T X' for the hooked input X [b, T, d_in] (bf16 contiguous) of a Linear layer on
 * sequences -- the same mathematics as kf_syrk_accum (module/linear.py:30-46 + tracker/factor.py:58) with the rows first
 * transposed to [b, d', T] in the workspace, where every row and its bias one are multiplied by the integer attention mask
 * (KF_I64 / KF_I32 / KF_U8, nullable, [b*T]; linear.py:39-43: 0 zeroes the row, 1 keeps it, other weights round the product to
 * bf16 as the reference's in-place mul_ does) and the ones row of the bias column is generated.  Needs T % 64 == 0,
 * d_in % 8 == 0, b <= 65535; the row counter is the caller's business.
 *
 * kf_conv2d_cov_accum: C[I',I'] += alpha * sum_{n,p} patches[n,p,:]^T patches[n,p,:] with IMPLICIT im2col (replaces
 * module/conv2d.py:15-64 + :106-128 + tracker/factor.py:58: no [b, P, I'] patch tensor): x [b, C, H, W] bf16 contiguous; C is
 * indexed in the reference's patch order (c, ky, kx).  Needs groups == 1, no bias, O2 % 8 == 0 and O1*O2 % 64 == 0
 * (kf_conv2d_cov_workspace_bytes returns -1 otherwise: use kf_im2col + kf_syrk_accum).
 *
 * kf_syrk_planes_bf16: C[d,d] += alpha * sum_z X_z X_z^T for X [b, d, K] bf16 contiguous (every sample d rows of K contiguous
 * values): the gradient covariance of a convolution straight from the NCHW output gradient, d = C_out, K = O1*O2 (replaces the
 * "b c o1 o2 -> (b o1 o2) c" rearrange of module/conv2d.py:130-132 + the addmm_ of tracker/factor.py:93).  Needs K % 64 == 0.
 *
 * All three accumulate the call's contribution in a [d_pad, d_pad] fp32 staging matrix inside the workspace (coalesced
 * atomics in the kernel's own row order) and add it to C -- permuted to the reference's index order and mirrored -- in one
 * finalize pass.
 */
int64_t kf_syrk_rows_workspace_bytes(int64_t b, int64_t T, int64_t d_in, int append_ones);
int kf_syrk_rows_bf16(float* C, int64_t ldc, const void* X, int64_t b, int64_t T, int64_t d_in, const void* mask, int mask_dtype,
                      int append_ones, float alpha, void* workspace, int64_t workspace_bytes, void* stream);
/*
 * kf_syrk_rows_f32 (ABI 12): C[d,d] += alpha * X'^T X' for FP32 rows X [n, d_in] (contiguous) -- the same mathematics as
 * kf_syrk_accum on fp32 input (module/linear.py:30-46 + tracker/factor.py:58: rows and their bias one times the mask value, in
 * fp32) -- on the bf16 MFMA engine: every value is split exactly into three bf16 terms (8 + 8 + 8 significand bits) and the
 * covariance is the sum of six bf16 products with fp32 accumulation (the three dropped products are below 2^-23 of the result).
 * LayerNorm outputs under autocast with fp32 factors (BERT) reach the covariance stage in fp32; the exact-fp32 MFMA instruction
 * behind kf_syrk_accum runs them at 1/16 of the bf16 rate.  mask: nullable [n], KF_I64 / KF_I32 / KF_U8 / KF_F32.  Needs
 * d_in % 8 == 0, 256 <= d_in < 32768, n <= 65535 * 64 (KF_ERR_INVALID_ARGUMENT beyond: callers take kf_syrk_accum); the row
 * counter is the caller's business.  Rows holding Inf become NaN here (x - bf16(x) = Inf - Inf), on kf_syrk_accum they stay Inf.
 */
int64_t kf_syrk_rows_f32_workspace_bytes(int64_t n, int64_t d_in);
int kf_syrk_rows_f32(float* C, int64_t ldc, const void* X, int64_t n, int64_t d_in, const void* mask, int mask_dtype, int append_ones,
                     float alpha, void* workspace, int64_t workspace_bytes, void* stream);
int64_t kf_syrk_planes_workspace_bytes(int64_t d);
int kf_syrk_planes_bf16(float* C, int64_t ldc, const void* X, int64_t b, int64_t d, int64_t K, float alpha, void* workspace,
                        int64_t workspace_bytes, void* stream);
int64_t kf_conv2d_cov_workspace_bytes(int64_t b, int64_t C, int64_t H, int64_t W, int k1, int k2, int s1, int s2, int p1,
                                      int p2, int d1, int d2);
int kf_conv2d_cov_accum(float* Cov, int64_t ldc, const void* x, int64_t b, int64_t C, int64_t H, int64_t W, int k1, int k2,
                        int s1, int s2, int p1, int p2, int d1, int d2, float alpha, void* workspace, int64_t workspace_bytes,
                        void* stream);

/*
 * kf_conv2d_cov_small (ABI 13): Cov[I', I'] += alpha * sum_{n,p} patches'[n,p,:]^T patches'[n,p,:] for a convolution whose patch
 * width I' = C*k1*k2 (+ 1 with append_ones: the bias column of module/conv2d.py:120-127) is AT MOST 32 -- the first layer of an
 * image model -- straight from x [b, C, H, W] (NCHW contiguous; KF_F32 / KF_BF16 / KF_F16), groups == 1: no patch tensor, one
 * fp32 MFMA (32x32x2) per two output positions, exact fp32 products and fp32 accumulation as kf_syrk_accum on fp32 rows.
 * Replaces module/conv2d.py:15-64 (extract_patches, patch order (c, ky, kx)) + :106-128 + tracker/factor.py:58 for such layers
 * (before: kf_im2col + kf_syrk_accum, a 110 MB fp32 patch matrix per ResNet-9 batch for a 27 x 27 result).  Both triangles are
 * written.  KF_ERR_INVALID_ARGUMENT when I' > 32; the row counter (b * O1 * O2) is the caller's business.
 */
int kf_conv2d_cov_small(float* Cov, int64_t ldc, const void* x, int x_dtype, int64_t b, int64_t C, int64_t H, int64_t W, int k1,
                        int k2, int s1, int s2, int p1, int p2, int d1, int d2, int append_ones, float alpha, void* stream);

/*
 * out[b, P, I'] = unfold(group_mean(x))  (+ ones column), I' = C/groups*k1*k2 + append_ones,
 * P = O1*O2.  Replaces module/conv2d.py:15-64 (extract_patches: rearrange, reduce "mean",
 * F.unfold, transpose) and :120-127 (ones column).  x is NCHW contiguous.
 */
int kf_im2col(void* out, int out_dtype, const void* x, int in_dtype, int64_t b, int64_t C, int64_t H,
              int64_t W, int k1, int k2, int s1, int s2, int p1, int p2, int d1, int d2, int groups,
              int append_ones, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Generic strided batched GEMM on the MFMA engine (building block of stages 2 and 3)
 * ------------------------------------------------------------------------------------------- */

/*
 * Operand view: element (z, r, k) = p[z*batch_stride + r*row_stride + k*k_stride] for r < rows,
 * k < depth; if ones_row (resp. ones_k) the index r == rows (resp. k == depth) exists and reads
 * 1.0 -- the un-masked bias column of module/linear.py:56-61 without a torch.cat.
 */
typedef struct kf_view {
    const void* p;
    int dtype;
    int64_t batch_stride, row_stride, k_stride;
    int64_t rows, depth;
    int ones_row, ones_k;
    int square; /* read x*x instead of x */
    /* 0 = plain strides.  Otherwise the operand is stored k-tile-major, "[depth/64][rows][64]":
     * element (r, k) = p[(k / 64) * k_tile_stride + r * row_stride + (k % 64)] with row_stride 64 and
     * k_tile_stride = rows * 64.  One k-step of a 128-row tile is then ONE contiguous 16 KB read
     * instead of 128 segments 2*depth bytes apart (3.6x faster on MI355X: TLB / DRAM-page locality).
     * Supported by the bf16 engine for K-contiguous operands (depth % 64 == 0) only. */
    int64_t k_tile_stride;
} kf_view;

/*
 * C[z, m, n] = alpha * sum_k A(z,m,k) * B(z,n,k) * (mul ? mul[m*ld_mul + n] : 1) + beta * C[z,m,n]
 * for z < batch.  C is fp32, z-th matrix at C + z*c_batch_stride.  c_batch_stride == 0 with
 * batch > 1 sums the batch into one matrix (atomically).  Replaces torch.matmul at
 * module/tracker/factor.py:205-226 and factor/config.py:350-352, torch.einsum at
 * module/linear.py:72 and module/conv2d.py:176.
 */
int kf_gemm(float* C, int64_t ldc, int64_t c_batch_stride, const kf_view* A, const kf_view* B,
            int64_t batch, float alpha, float beta, const float* mul, int64_t ld_mul, void* stream);

/* As kf_gemm without accumulation, with the output stored in c_dtype (KF_F32 or KF_BF16):
 * C[z,m,n] = alpha * sum_k A(z,m,k) B(z,n,k).  bf16 operands that are both K-contiguous (or both
 * K-strided) with extents / strides multiples of 8 run on the bf16 MFMA engine. */
int kf_gemm_out(void* C, int c_dtype, int64_t ldc, int64_t c_batch_stride, const kf_view* A,
                const kf_view* B, int64_t batch, float alpha, void* stream);

/* C[m, n] = sum_k A(m, k) B(n, k) + (n < bias_n ? bias[n] : 0), bf16 [A.rows, B.rows] with row stride ldc; bf16 operands,
 * both K-contiguous (NT bf16 MFMA engine).  With B = the first I columns of Q^T (zero rows for padding) and bias = Q[I, :]
 * this is "[X, 1] Q" -- the rotation of bias-augmented activations (module/linear.py:56-61 + tracker/factor.py:218-226)
 * without materialising the ones column, at a padded output width. */
int kf_gemm_bias_out(void* C, int64_t ldc, const kf_view* A, const kf_view* B, const float* bias, int64_t bias_n, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Stage 2 -- eigendecomposition and Lambda
 * ------------------------------------------------------------------------------------------- */

/*
 * Symmetric eigendecomposition in fp64 of  S = 0.5*(cov + cov^T) / count.
 * Replaces factor/eigen.py:193-205 (to(fp64), div_, symmetrise, torch.linalg.eigh -> LAPACK
 * syevd).  evals[d] ascending, evecs[d,d] row-major with eigenvectors in COLUMNS (torch
 * convention), both fp64; the host casts back to the covariance dtype (eigen.py:214-219).
 * cov is fp32 or fp64 [d,d]; count is a host value.  workspace: device, at least
 * kf_eigh_workspace_bytes(d) bytes.  max_sweeps <= 0 selects the default (100).
 * One-sided (Hestenes) Jacobi.  d >= 256: pairs of 32-column blocks on the fp64 matrix cores (Gram matrix, in-LDS
 * cyclic sweep, rotation applied with v_mfma_f64_16x16x4_f64); convergence -- a sweep without a rotation -- is decided
 * on the device and the kernels of sweeps enqueued past it return at once; the call enqueues sweeps in batches (8, then
 * 4) and synchronises `stream` once per batch to read three ints back.  d < 256: scalar rounds, one read-back per sweep.
 * d >= 256, default: FACTOR FIRST -- Cholesky of the diagonally sorted matrix plus shift * I, blocked Jacobi on the factor
 * without V (kf_eigh.hip); the shift is max(4 sqrt(d) eps, noise_rel) * ||S||_F, where noise_rel (ABI 11) is the relative
 * rounding noise of the covariance AS STORED (0: 2^-20 for an fp32 matrix, none for fp64; pass 2^-8 for a factor that was
 * exported in bf16); a non-positive pivot retries at 32x the shift twice, then falls back to the solver that carries V.
 * A covariance with NaN / Inf entries returns KF_ERR_NOT_CONVERGED.
 */
int64_t kf_eigh_workspace_bytes(int64_t d);
int kf_eigh_f64(const void* cov, int cov_dtype, double count, double noise_rel, int64_t d, double* evals, double* evecs,
                void* workspace, int64_t workspace_bytes, int max_sweeps, int* sweeps_done, void* stream);
/* Process-wide counters of the d >= 256 path taken by kf_eigh_f64 since the last reset: factor-first solves, fall-backs to the
 * solver that carries V, and Cholesky retries at a larger shift (ABI 11).  Any pointer may be NULL. */
void kf_eigh_stats(int64_t* factor_first, int64_t* fallback, int64_t* retries, int reset);

/*
 * Batched eigendecomposition of small symmetric matrices (l <= 96), one workgroup per matrix, all in LDS:
 * G[batch,l,l] fp32 -> evals[batch,l] DESCENDING, evecs[batch,l,l] row-major with eigenvectors in columns in
 * the same order.  inv_sqrt != 0 scales column j by 1/sqrt(max(lambda_j, floor_rel * lambda_max)): with G = Y^T Y
 * that makes Y * evecs an orthonormal basis of range(Y) (symmetric orthogonalisation).  Building block of the
 * low-rank query factorisation (module/tracker/precondition.py:19-75: torch.linalg.svd / torch.svd_lowrank).
 * max_sweeps <= 0 selects the default (60).  Does not synchronise.
 */
int kf_eigh_small_batched(const float* G, int64_t batch, int l, float* evals, float* evecs, int inv_sqrt,
                          float floor_rel, int max_sweeps, void* stream);

/*
 * Lambda[O,I'] += sum_b ( sum_r Gt[b,r,o] * At[b,r,i] )^2
 * where Gt = G Qg and At = [A,1] Qa are the factors of the per-sample gradient already rotated
 * into the eigenbasis (two kf_gemm calls).  Identical mathematics to
 * module/tracker/factor.py:218-226 (Qg^T (g_b Qa), square_, sum(0)) because
 * Qg^T (sum_r g_r a_r^T) Qa = (G Qg)^T (A' Qa); costs 2 R (I'^2 + O^2 + O I') instead of
 * 2 O I' (I' + O + R) flops per sample.  Gt: [b,R,O], At: [b,R,ld_at] contiguous (ld_at >= I': rows of At may carry
 * zero padding), dtype KF_F32 (ld_at == I'), or KF_BF16 (FactorArguments.lambda_dtype = bf16 of the reference's
 * low-precision preset: bf16 MFMA engine, fp32 accumulation; needs R > 1, O % 8 == 0, ld_at % 8 == 0 -- an odd I' is
 * carried zero-padded).
 * scale multiplies the per-sample gradient (gradient_scale, factor.py:269-270).
 */
int kf_lambda_accum(float* Lambda, int64_t ld_lambda, const void* Gt, const void* At, int64_t ld_at, int dtype,
                    int64_t b, int64_t R, int64_t O, int64_t Ip, float scale, void* stream);

/*
 * Lambda of a Linear layer on sequences, LDS-DMA engine (ABI 11; bf16 lambda_dtype, R % 64 == 0).  The same mathematics as
 * kf_lambda_accum (module/tracker/factor.py:218-226 on the factors of module/linear.py:112-122), with the rotated factors
 * K-CONTIGUOUS PER SAMPLE so that both operands of the per-sample product stream through global_load_lds:
 *
 *   kf_rotate_rows_transposed_bf16   out[s][j][r] = sum_k QT[j][k] X[s * R + r][k] (+ bias[j] for j < bias_n)
 *       X: bf16 [n * R, d] rows (the hooked output gradient or activation of n samples), QT: bf16 [m, ldq] = the transposed
 *       eigenvector matrix (row j = eigenvector j; only its first d columns are read -- for [A, 1] Qa the ones column is
 *       the fp32 row bias = Qa[I, :]), out: bf16 [n][m][R].  d % 64 == 0, R % 8 == 0, ldq % 8 == 0.  No workspace.
 *   kf_lambda_rows_accum             Lambda[o, i] += scale^2 * sum_s ( sum_r GtT[s][o][r] AtT[s][i][r] )^2,  i < Ip
 *       GtT: bf16 [b][O][R], AtT: bf16 [b][W][R] (W >= Ip rows per sample; rows >= Ip are not used).  One work item is a
 *       (sample range, 256 x 128 tile); per sample the 64 x 64 wave tiles are squared and summed in registers, one fp32
 *       atomic per element and item.  No workspace, no synchronisation.
 */
int kf_rotate_rows_transposed_bf16(void* out, const void* X, int64_t n, int64_t R, int64_t d, const void* QT, int64_t ldq, int64_t m,
                                   const float* bias, int64_t bias_n, void* stream);
int kf_lambda_rows_accum(float* Lambda, int64_t ld_lambda, const void* GtT, const void* AtT, int64_t b, int64_t R, int64_t O,
                         int64_t W, int64_t Ip, float scale, void* stream);

/*
 * Lambda of a Conv2d layer in the DENSE form (ABI 10): Lambda[o', i'] += scale^2 * sum_n ( Qg^T g_n Qa )[o', i']^2 with the
 * per-sample gradient g_n formed once (module/conv2d.py:164-177) and rotated as a whole (tracker/factor.py:218-226), instead of
 * rotating its factors: 2 R O I' + 2 O I'^2 flops per sample against 2 R (I'^2 + O^2 + O I') -- cheaper whenever the layer has
 * more output positions R than output channels O, and, with IMPLICIT im2col, free of the [b, R, I'] patch tensor and of its
 * [b R, I'] x [I', I'] rotation.  Three kernels: the padded, phase-split input copy (as kf_pairwise_score_conv2d); the
 * per-sample-gradient kernel fed by LDS-DMA, whose A operand is Gt_nchw = the hooked output gradient ALREADY rotated along
 * its channel axis, Gt[n, o', p] = sum_o Qg[o, o'] G[n, o, p] (one kf_gemm_out call), writing rows ordered (o', n) with the
 * patch axis (ky, kx, c) zero-padded to multiples of 8 channels; and one tall GEMM of those O*b rows with QaT_perm whose
 * epilogue squares and sums the rows of every o' in registers (no result matrix is stored).
 * QaT_perm: bf16 [n_out_padded, ldq], row i' = column i' of Qa with its rows permuted to the kernel's patch order (zero for
 * the padding channels); n_out = I' columns of Lambda are written.
 * Needs groups == 1, no bias, O2 % 8 == 0, O1*O2 % 64 == 0, (C rounded up to 8) * k1 * k2 % 64 == 0, b >= 256;
 * kf_lambda_conv2d_workspace_bytes returns -1 otherwise (use kf_im2col + kf_gemm + kf_lambda_accum).
 */
/* kf_lambda_conv2d_channels (ABI 13): the channel count Cp (C zero-padded) QaT_perm must be laid out with -- C rounded up to 8, or,
 * for a layer with very few input channels whose (C rounded up to 8) * k1 * k2 is not a multiple of 64 (a first convolution: 3
 * channels, 3 x 3 taps), the next multiple of 8 that makes it one (64), accepted while the padded per-sample-gradient GEMM stays
 * under 2e11 flop per call; -1 when the dense form does not apply (as kf_lambda_conv2d_workspace_bytes). */
int64_t kf_lambda_conv2d_channels(int64_t b, int64_t C, int64_t H, int64_t W, int64_t O, int k1, int k2, int s1, int s2, int p1, int p2,
                                  int d1, int d2);
int64_t kf_lambda_conv2d_workspace_bytes(int64_t b, int64_t C, int64_t H, int64_t W, int64_t O, int k1, int k2, int s1, int s2,
                                         int p1, int p2, int d1, int d2);
int kf_lambda_conv2d_accum(float* Lambda, int64_t ld_lambda, const void* Gt_nchw, const void* x, int64_t b, int64_t C, int64_t H,
                           int64_t W, int64_t O, int k1, int k2, int s1, int s2, int p1, int p2, int d1, int d2,
                           const void* QaT_perm, int64_t n_out, int64_t ldq, float scale, void* workspace,
                           int64_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Stage 3 -- preconditioning and pairwise scores
 * ------------------------------------------------------------------------------------------- */

/*
 * out[i] = 1 / (Lambda[i] / n_lambda + damping), computed in fp64 (utils/constants.py:82),
 * damping < 0 selects the heuristic 0.1 * mean(Lambda / n_lambda) (factor/config.py:331-338,
 * utils/constants.py:22).  Lambda fp32 [numel]; out fp32 [numel]; workspace: >= 16 bytes, device.
 */
int kf_inv_lambda(float* out, const float* Lambda, int64_t numel, double n_lambda, double damping,
                  void* workspace, void* stream);

/*
 * P[q,O,I'] = scale * Qg ( (Qg^T g_q Qa) o inv_lambda ) Qa^T,  g_q = sum_r G[q,r,:]^T [A[q,r,:],1]
 * Replaces module/tracker/precondition.py:102-123 (per-sample gradient, precondition, scale) and
 * factor/config.py:341-353 (four matmuls + mul_).  The forward rotation is done on the factors
 * (G Qg, A' Qa) instead of on the [O,I'] gradient.
 * G: [q,R,O], A: [q,R,I] (in_dtype, contiguous), Qg [O,O], Qa [I',I'], inv_lambda [O,I'] fp32.
 * P: [q,O,I'] in out_dtype (KF_F32, or KF_BF16 = ScoreArguments.score_dtype of the reference's
 * low-precision presets; all arithmetic and the staging of intermediate results stay fp32).
 * ldp: elements between consecutive rows (o) of P; I' for a compact block.
 * Qa_bf16, QaT_bf16 = Qa^T (bf16 [ldq, ldq], zero-padded from [I', I']; ldq = I' rounded up to a multiple of 8) and
 * QgT_bf16 = Qg^T (bf16 [O, O]); all three nullable.  When given with out_dtype KF_BF16, bf16 inputs, R > 1 and
 * ldp == ldq (ScoreArguments.precondition_dtype = bf16), all five contractions run on the bf16 MFMA engine at width
 * ldq: P comes out with ldq - I' zero columns per row, which is how the score kernels want an odd I' (a Linear with bias
 * on sequences, I' = I + 1) -- the bias column is the row Qa[I, :] added in an epilogue, not a torch.cat.
 * workspace (device): kf_precondition_workspace_bytes(q,R,O,I') bytes.
 */
int64_t kf_precondition_workspace_bytes(int64_t q, int64_t R, int64_t O, int64_t Ip);
int kf_precondition(void* P, int out_dtype, int64_t ldp, const void* G, const void* A, int in_dtype, int64_t q,
                    int64_t R, int64_t O, int64_t I, int append_ones, const float* Qg, const float* Qa,
                    const float* inv_lambda, float scale, const void* Qa_bf16, const void* QgT_bf16, const void* QaT_bf16,
                    int64_t ldq, void* workspace, int64_t workspace_bytes, void* stream);

/*
 * kf_precondition_bf16 (ABI 14): the bf16 form of kf_precondition WITHOUT the fp32 eigenvector matrices -- same result, same
 * call chain (module/tracker/precondition.py:102-123 + factor/config.py:341-353 with precondition_dtype = score_dtype = bf16,
 * factor/config.py:323-328: the reference casts the eigenvectors to that dtype in `prepare`).  kf_precondition reads Qg (for the
 * back rotation) and row I of Qa (the bias row) in fp32; a model whose eigenvectors are STORED in bf16 (the reference's
 * all-low-precision factors) would have to keep fp32 copies alive only for that: 99 GB at Llama-3-8B's full depth.
 * G: [q,R,O], A: [q,R,I] bf16; Qg_bf16, QgT_bf16 = Qg^T: [O,O]; Qa_bf16, QaT_bf16 = Qa^T: [ldq,ldq] zero-padded from [I',I'],
 * ldq = I' rounded up to a multiple of 8; bias_row: fp32, I' entries = Qa[I, :] (append_ones) or null; inv_lambda [O,I'] fp32.
 * P: [q,O,ldq] bf16 (ldp == ldq; ldq - I' zero columns per row).  KF_ERR_INVALID_ARGUMENT when the shape is not one the bf16
 * form takes (R == 1, O or I not multiples of 8 or below 64): the caller uses kf_precondition with fp32 eigenvectors then.
 * workspace (device): kf_precondition_workspace_bytes(q,R,O,I') bytes.
 */
int kf_precondition_bf16(void* P, int64_t ldp, const void* G, const void* A, int64_t q, int64_t R, int64_t O, int64_t I,
                         int append_ones, const void* Qg_bf16, const void* QgT_bf16, const void* Qa_bf16, const void* QaT_bf16,
                         int64_t ldq, const float* bias_row, const float* inv_lambda, float scale, void* workspace,
                         int64_t workspace_bytes, void* stream);

/*
 * scores[q, n] += scale * sum_{o,i} P[q,o,i] * ( sum_r G[n,r,o] * A'[n,r,i] )   for n < b
 * Replaces module/linear.py:112-122 and module/conv2d.py:199-209 (three-operand einsum),
 * module/tracker/pairwise_score.py:41-45 and the per-layer add_ of score/dot_product.py:105-117:
 * every layer accumulates into the same [Q, ld_scores] device buffer, one D2H per shard.
 * P: [Q,O,I'] contiguous, p_dtype KF_F32 or KF_BF16.  G: [b,R,O], A: [b,R,I] (in_dtype,
 * contiguous); A' = [A,1] if append_ones.  R == 1 never materialises the per-sample gradient;
 * R > 1 forms it in `workspace` (kf_pairwise_workspace_bytes) in P's dtype and contracts it with
 * P on the MFMA engine: v_mfma_f32_32x32x16_bf16 (fp32 accumulate) for bf16 P when O*I' is a
 * multiple of 8, v_mfma_f32_32x32x2_f32 otherwise.  p_k_tile_stride: 0 if P is plain [Q, O*I'];
 * Q*64 if the caller re-laid it out k-tile-major ([O*I'/64][Q][64], see kf_view) -- bf16, R > 1 and
 * O*I' % 64 == 0 only; the per-sample gradients are then written k-tile-major too.
 */
int64_t kf_pairwise_workspace_bytes(int64_t b, int64_t R, int64_t O, int64_t Ip);
int kf_pairwise_score(float* scores, int64_t ld_scores, const void* P, int p_dtype,
                      int64_t p_k_tile_stride, int64_t Q, const void* G, const void* A, int in_dtype,
                      int64_t b, int64_t R, int64_t O, int64_t I, int append_ones, float scale,
                      void* workspace, int64_t workspace_bytes, void* stream);

/*
 * Second-generation bf16 score path (ABI 8): the same contraction as kf_pairwise_score for R > 1, with P handed over
 * k-tile-major ([D/64][Q][64] bf16) and both kernels fed by LDS-DMA (global_load_lds_dwordx4), which needs K-contiguous
 * operands.  The per-sample gradients are formed k-tile-major in `workspace` and contracted with P by a 256 x 256-tile
 * MFMA kernel (8 waves, split-K, fp32 atomics into the shared [Q, ld_scores] block).
 *
 * kf_pairwise_score_conv2d -- IMPLICIT im2col: replaces module/conv2d.py:15-64 (extract_patches / F.unfold),
 * :134-177 (per-sample gradient einsum) and :179-209 (score einsum) without ever materialising the [b, P, I'] patch
 * tensor.  G_nchw is the hooked output gradient as autograd delivers it, [b, O, O1, O2] bf16 contiguous (its
 * (o, p) rows are already K-contiguous -- no `b c h w -> b (h w) c` copy, conv2d.py:130-132); x is the hooked layer
 * input [b, C, H, W] bf16 contiguous.  A zero-padded, column-phase-split copy of x (s2 copies, ~1.3x the input instead of
 * the k1 k2 / (s1 s2)-fold patch tensor) is written to the workspace and the gradient kernel fetches row
 * i = (ky, kx, c) of A'^T straight from it.  THE PATCH AXIS OF P IS ORDERED (ky, kx, c), d = o * I' + (ky * k2 + kx) * C + c
 * (the reference's is (c, ky, kx); the host permutes once when it builds the k-tile-major P).
 * Needs groups == 1, no bias, O2 % 8 == 0, O1 * O2 % 64 == 0, C % 8 == 0, O * I' % 64 == 0, 16-byte aligned pointers;
 * returns KF_ERR_INVALID_ARGUMENT otherwise (the host then uses kf_im2col + kf_pairwise_score).
 *
 * kf_pairwise_score_rows -- Linear layers on [b, R, .] activations (module/linear.py:68-77, :112-122): G [b, R, O] and
 * A [b, R, I] bf16 contiguous are transposed to [b, O, R] / [b, I'p, R] in the workspace (the ones row of the bias column,
 * linear.py:56-61, and the zero rows that pad I' to I'p, a multiple of 8, are generated there -- no torch.cat); P is
 * [O * I'p / 64][Q][64] with the same padding.  Needs R % 64 == 0, O % 8 == 0, I % 8 == 0, I'p % 8 == 0,
 * I'p >= I + append_ones, O * I'p % 64 == 0, b <= 65535.
 */
int64_t kf_pairwise_conv2d_workspace_bytes(int64_t b, int64_t C, int64_t H, int64_t W, int64_t O, int k1, int k2,
                                           int s1, int s2, int p1, int p2, int d1, int d2);
int kf_pairwise_score_conv2d(float* scores, int64_t ld_scores, const void* P_tiled, int64_t Q, const void* G_nchw,
                             const void* x, int64_t b, int64_t C, int64_t H, int64_t W, int64_t O, int k1, int k2,
                             int s1, int s2, int p1, int p2, int d1, int d2, float scale, void* workspace,
                             int64_t workspace_bytes, void* stream);
int64_t kf_pairwise_rows_workspace_bytes(int64_t b, int64_t R, int64_t O, int64_t Ip);
int kf_pairwise_score_rows(float* scores, int64_t ld_scores, const void* P_tiled, int64_t Q, const void* G, const void* A,
                           int64_t b, int64_t R, int64_t O, int64_t I, int64_t Ip, int append_ones, float scale,
                           void* workspace, int64_t workspace_bytes, void* stream);
/* The same with the train batch handed over in TWO segments (ABI 11): rows of (G, A) first (b0 samples -> score columns 0 .. b0),
 * then (G1, A1) (b1 samples -> columns b0 .. b0 + b1); b1 == 0 ignores the second segment.  With b0 + b1 = 256 the score GEMM
 * reads P once for two micro-batches of 128 -- at that batch size it is bound by the HBM stream of P (intensity = b flop/byte),
 * not by the matrix cores.  Workspace: kf_pairwise_rows_workspace_bytes(b0 + b1, ...). */
int kf_pairwise_score_rows2(float* scores, int64_t ld_scores, const void* P_tiled, int64_t Q, const void* G, const void* A, int64_t b0,
                            const void* G1, const void* A1, int64_t b1, int64_t R, int64_t O, int64_t I, int64_t Ip, int append_ones,
                            float scale, void* workspace, int64_t workspace_bytes, void* stream);

/*
 * out[r] (+)= scale * sum_i X[r,i] * Y[r,i] * (W ? W[i] : 1)      r < rows, i < D
 * The reduction of self-influence scores: module/tracker/self_score.py:61-62
 * (`preconditioned.mul_(gradient).sum(dim=(1,2))`) with X = Y = the rotated per-sample gradient and
 * W = Lambda^-1 for EK-FAC / K-FAC, X = Y = g for the identity strategy, W = Lambda^-1 for the diagonal
 * one; and self_score.py:164 / linear.py:141-146 (self-measurement scores) with X = the preconditioned
 * measurement gradient, Y = the loss gradient.  X, Y: [rows, D] contiguous, KF_F32 or KF_BF16
 * independently; W: fp32 [D] or NULL; accumulate == 0 overwrites out.  fp32 accumulation.
 */
int kf_rowwise_dot(float* out, const void* X, int x_dtype, const void* Y, int y_dtype, const float* W,
                   int64_t rows, int64_t D, float scale, int accumulate, void* stream);

/*
 * Factored score of LOW-RANK query gradients against samples with R rows each (ABI 11): the reference contracts
 * "qik,qko,b...i,b...o->qb" (module/linear.py:83-99) along whichever path opt_einsum finds cheapest; when the train batch is
 * small and the layer wide (Llama-3-8B projections, k = 64: expanding P_q = L_q R_q costs more bytes per pair than the factored
 * form costs flops) that is  scores[q, n] += scale * sum_{r, k} (G_n L_q)[r, k] (A'_n R_q^T)[r, k].  The two products are tall
 * NT GEMMs on the LDS-DMA engine (kf_gemm_out / kf_gemm_bias_out with the queries' factors stacked to [Q k, O] and [Q k, I']);
 * this entry reduces their bf16 results U, V: [b R, Q K] row-major over (r, k).  K % 8 == 0; fp32 accumulation, one atomic per
 * (query, sample, row split).
 */
int kf_lowrank_rows_dot(float* scores, int64_t ld_scores, const void* U, const void* V, int64_t b, int64_t R, int64_t Q, int64_t K,
                        float scale, void* stream);

/* out[r,i] = scale * X[r,i] * M[i]: the diagonal strategy's preconditioner (factor/config.py:215-222). */
int kf_mul_bcast(float* out, const void* X, int x_dtype, const float* M, int64_t rows, int64_t D,
                 float scale, void* stream);

/* Elementwise helper: dst[i] = (out_dtype) src[i] -- export of fp32 accumulators in the factor dtype. */
int kf_cast(void* dst, int dst_dtype, const void* src, int src_dtype, int64_t numel, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* KRONFLUENCE_HIP_H */
