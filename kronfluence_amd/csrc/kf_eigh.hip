// kf_eigh.hip -- fp64 symmetric eigensolver for the EK-FAC covariance factors (gfx950).
//
// Replaces torch.linalg.eigh (LAPACK syevd) at factor/eigen.py:205.  Algorithm: one-sided
// (Hestenes) Jacobi.  With S symmetric, W := S and V := I; every rotation J is applied to the
// columns of both (W <- W J, V <- V J) so that W = S V throughout; at convergence the columns of W
// are mutually orthogonal, hence V^T S^2 V is diagonal and V holds the eigenvectors of S.  The
// eigenvalue of column j is the Rayleigh quotient v_j . w_j (sign included).  Rotations of one
// round act on disjoint column pairs (round-robin tournament schedule), one workgroup per pair;
// both matrices are stored transposed (a "column" is a contiguous row) so every access is a
// coalesced stream of doubles.  All arithmetic is fp64 VALU (the matrices are L2/MALL resident).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <mutex>

#include "../../include/kronfluence_hip.h"

namespace {

constexpr int EB = 256;  // threads per pair-workgroup

__device__ __forceinline__ double block_sum(double v, double* scratch) {
    // wave64 shuffle reduction, then across the 4 waves through LDS
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    return scratch[0] + scratch[1] + scratch[2] + scratch[3];
}

// Wt = 0.5 * (cov + cov^T) / count   (symmetric, so Wt == W),  Vt = I
__global__ void eigh_init_kernel(double* Wt, double* Vt, const void* cov, int is_f64, double count, int64_t d) {
    const int64_t total = d * d;
    for (int64_t e = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; e < total;
         e += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const int64_t i = e / d, j = e % d;
        double a, b;
        if (is_f64) { a = reinterpret_cast<const double*>(cov)[i * d + j]; b = reinterpret_cast<const double*>(cov)[j * d + i]; }
        else { a = reinterpret_cast<const float*>(cov)[i * d + j]; b = reinterpret_cast<const float*>(cov)[j * d + i]; }
        // mirror the reference's op order: divide, then add the transpose, then halve (eigen.py:198-203)
        Wt[e] = 0.5 * (a / count + b / count);
        Vt[e] = (i == j) ? 1.0 : 0.0;
    }
}

// One round of the tournament: block k rotates the pair (p, q) of round `round`.
// npl = number of "players" (d rounded up to even); player npl-1 may be a bye when d is odd.
__global__ __launch_bounds__(EB) void jacobi_round_kernel(double* Wt, double* Vt, int64_t d, int npl, int round,
                                                          double tol, double null2, int* rotated) {
    __shared__ double scratch[4];
    const int k = blockIdx.x;
    int p, q;
    const int m = npl - 1;
    if (k == 0) { p = round % m; q = m; }
    else { p = (round + k) % m; q = (round - k + m) % m; }
    if (p >= d || q >= d) return;
    double* wp = Wt + static_cast<int64_t>(p) * d;
    double* wq = Wt + static_cast<int64_t>(q) * d;
    double alpha = 0.0, beta = 0.0, gamma = 0.0;
    for (int64_t i = threadIdx.x; i < d; i += EB) {
        const double x = wp[i], y = wq[i];
        alpha += x * x; beta += y * y; gamma += x * y;
    }
    alpha = block_sum(alpha, scratch);
    beta = block_sum(beta, scratch);
    gamma = block_sum(gamma, scratch);
    const double lim = tol * sqrt(alpha) * sqrt(beta);
    if (!(fabs(gamma) > lim)) return;  // uniform across the block (also catches NaN and zero columns)
    // A column with |S v| <= eps-level * ||S||_F already IS a null vector of S to working precision
    // (V stays orthonormal whatever we do): rotating noise against anything only burns sweeps.
    if (alpha <= null2 || beta <= null2) return;
    const double zeta = (beta - alpha) / (2.0 * gamma);
    const double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
    const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
    double* vp = Vt + static_cast<int64_t>(p) * d;
    double* vq = Vt + static_cast<int64_t>(q) * d;
    // (No de Rijk column swap: combined with the parallel round-robin order it stalls -- measured.)
    for (int64_t i = threadIdx.x; i < d; i += EB) {
        const double x = wp[i], y = wq[i];
        wp[i] = c * x - s * y; wq[i] = s * x + c * y;
        const double u = vp[i], v = vq[i];
        vp[i] = c * u - s * v; vq[i] = s * u + c * v;
    }
    if (threadIdx.x == 0) atomicAdd(rotated, 1);
}

// ------------------------------------------------------------------------------------------------
// Block variant of the round kernel for large d.  The scalar kernel streams all of W and V once per
// round and there are d - 1 rounds per sweep: at d = 2304 that is 390 GB per sweep, so the solver is
// bound by L2 / Infinity-Cache bandwidth.  Here a workgroup owns a PAIR OF BLOCKS of BS columns: it
// forms the 2BS x 2BS Gram matrix of its columns (one read of W), diagonalises it with a cyclic
// two-sided Jacobi sweep held in LDS by one wave (the rotation angles are exactly those of the
// one-sided method applied to the columns), and applies the accumulated 2BS x 2BS rotation U to its
// columns of W and V (one read + one write of each).  Rounds per sweep drop from d - 1 to d/BS - 1.
// The Gram matrix is rebuilt from the actual columns every round, so rounding in the small problem
// cannot accumulate; convergence is still "a whole sweep without a rotation" on direct dot products.
// ------------------------------------------------------------------------------------------------
constexpr int BS = 4, NC = 2 * BS, NG = NC * (NC + 1) / 2;

__global__ __launch_bounds__(EB) void jacobi_block_round_kernel(double* Wt, double* Vt, int64_t d, int nblocks, int players,
                                                                int round, double tol, double null2, int* rotated) {
    __shared__ double red[4][NG];
    __shared__ double Gs[NC][NC];
    __shared__ double Us[NC][NC];
    __shared__ int any_rotation;
    const int k = blockIdx.x, m = players - 1;
    int P, Q;
    if (k == 0) { P = round % m; Q = m; }
    else { P = (round + k) % m; Q = (round - k + m) % m; }
    if (P >= nblocks && Q >= nblocks) return;
    int64_t col[NC];
    bool valid[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int64_t j = static_cast<int64_t>(c < BS ? P : Q) * BS + (c % BS);
        valid[c] = (c < BS ? P : Q) < nblocks && j < d;
        col[c] = valid[c] ? j : 0;
    }
    // ---- pass 1: Gram matrix of the NC columns
    double acc[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) acc[g] = 0.0;
    for (int64_t i = threadIdx.x; i < d; i += EB) {
        double x[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) { const double v = Wt[col[c] * d + i]; x[c] = valid[c] ? v : 0.0; }
        int g = 0;
#pragma unroll
        for (int a = 0; a < NC; ++a)
#pragma unroll
            for (int b = a; b < NC; ++b) acc[g++] += x[a] * x[b];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        double v = acc[g];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if (lane == 0) red[wave][g] = v;
    }
    __syncthreads();
    if (threadIdx.x < NC * NC) {
        const int r = threadIdx.x / NC, c = threadIdx.x % NC;
        const int a = r < c ? r : c, b = r < c ? c : r;
        const int g = a * NC - a * (a - 1) / 2 + (b - a);
        Gs[r][c] = red[0][g] + red[1][g] + red[2][g] + red[3][g];
        Us[r][c] = r == c ? 1.0 : 0.0;
    }
    if (threadIdx.x == 0) any_rotation = 0;
    __syncthreads();
    // ---- the small problem: one cyclic sweep over the NC (NC - 1) / 2 pairs, wave 0, lane = (r, c)
    if (threadIdx.x < NC * NC) {
        const int r = threadIdx.x / NC, c = threadIdx.x % NC;
        int did = 0;
        for (int p = 0; p < NC - 1; ++p)
            for (int q = p + 1; q < NC; ++q) {
                const double a = Gs[p][p], b = Gs[q][q], g = Gs[p][q];
                if (!(fabs(g) > tol * sqrt(a) * sqrt(b)) || !(a > null2) || !(b > null2)) continue;  // uniform
                const double zeta = (b - a) / (2.0 * g);
                const double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
                // columns: G <- G J, U <- U J
                const double gp = Gs[r][p], gq = Gs[r][q], up = Us[r][p], uq = Us[r][q];
                __builtin_amdgcn_wave_barrier();
                if (c == p) { Gs[r][c] = cs * gp - sn * gq; Us[r][c] = cs * up - sn * uq; }
                if (c == q) { Gs[r][c] = sn * gp + cs * gq; Us[r][c] = sn * up + cs * uq; }
                __builtin_amdgcn_wave_barrier();
                // rows: G <- J^T G
                const double tp = Gs[p][c], tq = Gs[q][c];
                __builtin_amdgcn_wave_barrier();
                if (r == p) Gs[r][c] = cs * tp - sn * tq;
                if (r == q) Gs[r][c] = sn * tp + cs * tq;
                __builtin_amdgcn_wave_barrier();
                did = 1;
            }
        if (did && threadIdx.x == 0) { any_rotation = 1; atomicAdd(rotated, 1); }
    }
    __syncthreads();
    if (!any_rotation) return;
    // ---- pass 2: W <- W U, V <- V U on the owned columns
    double u[NC][NC];
#pragma unroll
    for (int a = 0; a < NC; ++a)
#pragma unroll
        for (int b = 0; b < NC; ++b) u[a][b] = Us[a][b];
#pragma unroll 1
    for (int which = 0; which < 2; ++which) {
        double* M = which == 0 ? Wt : Vt;
        for (int64_t i = threadIdx.x; i < d; i += EB) {
            double x[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) { const double v = M[col[c] * d + i]; x[c] = valid[c] ? v : 0.0; }
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                double y = 0.0;
#pragma unroll
                for (int c = 0; c < NC; ++c) y += x[c] * u[c][j];
                if (valid[j]) M[col[j] * d + i] = y;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Blocked rounds for d >= 256: pairs of 32-column blocks on the fp64 matrix cores.
//
// A "player" is a block of KB = 32 columns; a round pairs the players up (round-robin tournament) and for every
// pair (P, Q) -- 64 columns, stored as 64 contiguous rows of W^T and V^T -- runs three kernels:
//   gram    G = Wp Wp^T (64 x 64, K = d split over `gsplit` workgroups; v_mfma_f64_16x16x4_f64, rows staged in LDS)
//   solve   one cyclic two-sided Jacobi sweep over the 64 x 64 Gram matrix held in LDS (63 parallel rounds of 32
//           disjoint rotations; the angles are exactly those of the one-sided method applied to the columns) ->
//           the accumulated rotation U (64 x 64) and a "this pair rotated" flag
//   update  Wp <- U^T Wp, Vp <- U^T Vp (row form of W <- W U; MFMA again, K = 64), skipped for pairs that did not rotate
// Rounds per sweep: d / 32 - 1 instead of d / 4 - 1 for the 8-column VALU kernel above, each round still streaming W
// and V once -- the solver is bound by that stream (L2 / Infinity Cache / HBM), so the sweep time drops ~8x.
// The Gram matrix is rebuilt from the actual columns every round (no drift); convergence is "a whole sweep without a
// rotation", the rotation test is the relative one of the scalar kernel.
// ------------------------------------------------------------------------------------------------
typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int KB = 32, KP = 2 * KB;          // columns per block / per pair
constexpr int GK = 32, GPITCH = GK + 2;      // gram: k-tile (doubles) and LDS pitch (bank-conflict free operand reads)
constexpr int UT = 64, UPITCH = 80;          // update: tile width and LDS pitch (pitch % 32 == 16)

__device__ __forceinline__ void pair_of_round(int k, int round, int players, int& P, int& Q) {
    const int m = players - 1;
    if (k == 0) { P = round % m; Q = m; }
    else { P = (round + k) % m; Q = (round - k + m) % m; }
}

// row r (0..63) of the pair -> row of W^T / V^T, or -1 (bye block / beyond d)
__device__ __forceinline__ int64_t pair_row(int r, int P, int Q, int nblocks, int64_t d) {
    const int blk = r < KB ? P : Q;
    const int64_t j = static_cast<int64_t>(blk) * KB + (r & (KB - 1));
    return (blk < nblocks && j < d) ? j : -1;
}

// partial[pair][split][64][64] = sum over this split's i-range of Wp[a][i] Wp[b][i]
__global__ __launch_bounds__(256) void eigh_gram_kernel(const double* __restrict__ Wt, double* __restrict__ partial, int64_t d,
                                                        int nblocks, int players, int round, int gsplit, int64_t chunk,
                                                        const int* __restrict__ done) {
    __shared__ double tile[KP * GPITCH];
    if (*done) return;  // converged in an earlier sweep of this batch of launches (device-side convergence flag)
    const int pair = blockIdx.x, split = blockIdx.y;
    int P, Q;
    pair_of_round(pair, round, players, P, Q);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // loader (round 6): 32 lanes along a row -- a wave instruction reads 2 rows x 256 contiguous bytes (it used to read one
    // double from each of 64 rows) -- thread (lr0, lcol) fetches column lcol of rows lr0 + 8 p
    const int lcol = tid & 31, lr0 = tid >> 5;
    int64_t grow[8];
#pragma unroll
    for (int pp = 0; pp < 8; ++pp) grow[pp] = pair_row(lr0 + 8 * pp, P, Q, nblocks, d);
    const int64_t i_begin = split * chunk, i_end = min(d, i_begin + chunk);
    f64x4 acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[b] = f64x4{0.0, 0.0, 0.0, 0.0};
    // (round 6: the next k-tile's global loads are issued before the MFMAs of the current one -- the workgroup always has a tile in
    //  flight instead of alternating between waiting for one and consuming it; profiles/r06_eigh_prefetch_coalesced.log)
    double v[8];
    auto fetch = [&](int64_t i0) {
        const int64_t i = i0 + lcol;
#pragma unroll
        for (int pp = 0; pp < 8; ++pp) v[pp] = (grow[pp] >= 0 && i < i_end) ? Wt[grow[pp] * d + i] : 0.0;
    };
    if (i_begin < i_end) fetch(i_begin);
    for (int64_t i0 = i_begin; i0 < i_end; i0 += GK) {
        __syncthreads();  // previous tile fully consumed
#pragma unroll
        for (int pp = 0; pp < 8; ++pp) tile[(lr0 + 8 * pp) * GPITCH + lcol] = v[pp];
        __syncthreads();
        if (i0 + GK < i_end) fetch(i0 + GK);
#pragma unroll
        for (int ks = 0; ks < GK / 4; ++ks) {
            const int k = ks * 4 + (lane >> 4);
            const double a = tile[(wave * 16 + (lane & 15)) * GPITCH + k];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const double bv = tile[(b * 16 + (lane & 15)) * GPITCH + k];
                acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv, acc[b], 0, 0, 0);
            }
        }
    }
    double* out = partial + (static_cast<int64_t>(pair) * gsplit + split) * (KP * KP);
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            out[(wave * 16 + (lane >> 4) + 4 * r) * KP + b * 16 + (lane & 15)] = acc[b][r];
}

// Rotations of one pair from its 64 x 64 Gram matrix, held in LDS: U and the pair's "rotated" flag.
//   cross != 0: the 32 x 32 pairs (p in block P, q in block Q), as 32 parallel rounds of 32 disjoint rotations
//               (round t: p = i, q = 32 + (i + t) % 32) -- run for every pair of every tournament round;
//   cross == 0: the pairs INSIDE each of the two blocks (two 32-player tournaments side by side, 31 rounds of 16 + 16
//               rotations) -- run once per sweep on one perfect matching of the blocks.
// Together every column pair of the matrix is visited exactly once per sweep (a cyclic Jacobi ordering by blocks).
// (Driving the pair's 64 x 64 subproblem further before touching the columns -- 2, 3 or 5 alternating cross / in-block passes
// per block pair -- was measured: the sweep count stays at 20-21 for d = 3073 and the solve only gets longer.)
// The Gram matrix is rotated two-sidedly, G <- J^T G J, so the angles are exactly those of the one-sided method applied
// to the columns.  Two barriers per round: the column pass reads G and writes a second buffer (every column belongs to
// exactly one rotation, so all of it is rewritten), the row pass writes back.
__global__ __launch_bounds__(256) void eigh_solve_kernel(const double* __restrict__ partial, double* __restrict__ Ubuf, int* __restrict__ pair_flag,
                                                         int gsplit, int cross, double tol, const double* __restrict__ frob2,
                                                         double null_scale, int* rotated, const int* __restrict__ done) {
    constexpr int LP = KP + 1;
    if (*done) return;
    __shared__ double G[KP * LP];
    __shared__ double H[KP * LP];
    __shared__ double U[KP * LP];
    __shared__ double cs[KB], sn[KB];
    __shared__ int pp[KB], qq[KB];
    __shared__ int any;
    const int pair = blockIdx.x, tid = threadIdx.x;
    const double null2 = frob2[0] * null_scale;
    const double* src = partial + static_cast<int64_t>(pair) * gsplit * (KP * KP);
    for (int e = tid; e < KP * KP; e += 256) {
        double s = 0.0;
        for (int g = 0; g < gsplit; ++g) s += src[static_cast<int64_t>(g) * (KP * KP) + e];
        const int r = e / KP, c = e % KP;
        H[r * LP + c] = s;
        U[r * LP + c] = r == c ? 1.0 : 0.0;
    }
    if (tid == 0) any = 0;
    __syncthreads();
    // symmetrise (the two triangles were accumulated in different orders)
    for (int e = tid; e < KP * KP; e += 256) {
        const int r = e / KP, c = e % KP;
        G[r * LP + c] = 0.5 * (H[r * LP + c] + H[c * LP + r]);
    }
    __syncthreads();
    // Early out (round 4): if no pair of THIS pass's set -- cross-block pairs, or the pairs inside the two blocks -- exceeds the
    // rotation threshold on the Gram matrix as it arrived, no round below would rotate anything (G does not change without a
    // rotation), so the 32 rounds x 2 barriers are skipped: exact, and it is the common case in the quadratic phase of the
    // convergence and in the whole confirmation sweep.  (The in-LDS solve is latency bound -- 78 us per launch, 56 % of the
    // eigensolver's kernel time at d = 3073, profiles/r04_bert_base_n2048_kernel_stats.csv.)
    {
        int need = 0;
        for (int e = tid; e < KP * KP; e += 256) {
            const int r = e / KP, c = e % KP;
            const bool in_set = cross ? (r < KB && c >= KB) : (r < c && (r / KB) == (c / KB));
            if (in_set) {
                const double a = G[r * LP + r], b = G[c * LP + c], g = G[r * LP + c];
                need |= (g * g > tol * tol * a * b && a > null2 && b > null2) ? 1 : 0;
            }
        }
        if (!__syncthreads_or(need)) {
            if (tid == 0) pair_flag[pair] = 0;
            return;
        }
    }
    const int k = tid & 31, rg = tid >> 5;  // column pass: rotation k, rows rg + 8 j
    const int rounds = cross ? KB : KB - 1;
    int did = 0;
    for (int round = 0; round < rounds; ++round) {
        int p, q;
        if (cross) {
            p = k; q = KB + ((k + round) & (KB - 1));
        } else {
            const int kk = k & 15, base = (k >> 4) * KB, m = KB - 1;
            if (kk == 0) { p = base + round % m; q = base + m; }
            else { p = base + (round + kk) % m; q = base + (round - kk + m) % m; }
        }
        const double a = G[p * LP + p], b = G[q * LP + q], g = G[p * LP + q];
        double c = 1.0, s = 0.0;
        // |g| > tol sqrt(a b), tested without square roots; the rotation with two square roots and two divisions instead
        // of four and three (the fp64 sqrt / div sequences are the critical path of a round):
        //   zeta = (b - a) / 2g,  t = sgn(zeta) / (|zeta| + sqrt(1 + zeta^2)) = sgn(b - a) 2g / (|b - a| + sqrt((b - a)^2 + 4 g^2))
        if (g * g > tol * tol * a * b && a > null2 && b > null2) {
            const double w = b - a, h = 2.0 * g;
            const double t = copysign(1.0, w) * h / (fabs(w) + sqrt(w * w + h * h));
            c = 1.0 / sqrt(1.0 + t * t);
            s = c * t;
            did = 1;
        }
        if (rg == 0) { cs[k] = c; sn[k] = s; pp[k] = p; qq[k] = q; }
        // columns: H <- G J, U <- U J
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = rg + 8 * j;
            const double gp = G[r * LP + p], gq = G[r * LP + q];
            H[r * LP + p] = c * gp - s * gq; H[r * LP + q] = s * gp + c * gq;
            if (s != 0.0) {
                const double up = U[r * LP + p], uq = U[r * LP + q];
                U[r * LP + p] = c * up - s * uq; U[r * LP + q] = s * up + c * uq;
            }
        }
        __syncthreads();
        // rows: G <- J^T H   (column tid & 63, rotations (tid >> 6) + 4 j)
        {
            const int col = tid & 63;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int kr = (tid >> 6) + 4 * j;
                const double cr = cs[kr], sr = sn[kr];
                const int pr = pp[kr], qr = qq[kr];
                const double tp = H[pr * LP + col], tq = H[qr * LP + col];
                G[pr * LP + col] = cr * tp - sr * tq; G[qr * LP + col] = sr * tp + cr * tq;
            }
        }
        __syncthreads();
    }
    if (did) any = 1;
    __syncthreads();
    if (tid == 0) {
        pair_flag[pair] = any;
        if (any) atomicAdd(rotated, 1);
    }
    if (any) {
        double* out = Ubuf + static_cast<int64_t>(pair) * (KP * KP);
        for (int e = tid; e < KP * KP; e += 256) out[e] = U[(e / KP) * LP + (e % KP)];
    }
}

// rows of the pair: new[j][i] = sum_c U[c][j] old[c][i], for W^T and V^T, over this workgroup's i-range
__global__ __launch_bounds__(256) void eigh_update_kernel(double* __restrict__ Wt, double* __restrict__ Vt, const double* __restrict__ Ubuf,
                                                          const int* __restrict__ pair_flag, int64_t d, int nblocks, int players, int round,
                                                          int64_t chunk, const int* __restrict__ done, int with_v) {
    extern __shared__ double lds[];
    if (*done) return;
    double* Ul = lds;                 // [64 c][UPITCH]  (j contiguous)
    double* Tl = lds + KP * UPITCH;   // [64 c][UPITCH]  (i contiguous)
    const int pair = blockIdx.x;
    if (!pair_flag[pair]) return;
    int P, Q;
    pair_of_round(pair, round, players, P, Q);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double* usrc = Ubuf + static_cast<int64_t>(pair) * (KP * KP);
    for (int e = tid; e < KP * KP; e += 256) Ul[(e / KP) * UPITCH + (e % KP)] = usrc[e];
    // loader (round 6): 16 lanes along a row -- a wave instruction reads 4 rows x 128 contiguous bytes -- thread (lr0, lcol) fetches
    // columns lcol + 16 c of rows lr0 + 16 p
    const int lcol = tid & 15, lr0 = tid >> 4;
    int64_t grow[4];
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) grow[pp] = pair_row(lr0 + 16 * pp, P, Q, nblocks, d);
    // output rows of this wave: j = wave * 16 + (lane >> 4) + 4 r
    int64_t orow[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) orow[r] = pair_row(wave * 16 + (lane >> 4) + 4 * r, P, Q, nblocks, d);
    const int64_t i_begin = blockIdx.y * chunk, i_end = min(d, i_begin + chunk);
#pragma unroll 1
    for (int which = 0; which < 1 + with_v; ++which) {   // with_v == 0: the factor-first solver carries no V
        double* Mx = which == 0 ? Wt : Vt;
        double v[16];
        auto fetch = [&](int64_t i0) {   // (round 6: the next tile's loads fly under this tile's MFMAs and stores, see eigh_gram_kernel)
#pragma unroll
            for (int pp = 0; pp < 4; ++pp)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int64_t i = i0 + lcol + 16 * c;
                    v[pp * 4 + c] = (grow[pp] >= 0 && i < i_end) ? Mx[grow[pp] * d + i] : 0.0;
                }
        };
        if (i_begin < i_end) fetch(i_begin);
        for (int64_t i0 = i_begin; i0 < i_end; i0 += UT) {
            __syncthreads();  // previous tile consumed (and U staged, first time round)
#pragma unroll
            for (int pp = 0; pp < 4; ++pp)
#pragma unroll
                for (int c = 0; c < 4; ++c) Tl[(lr0 + 16 * pp) * UPITCH + lcol + 16 * c] = v[pp * 4 + c];
            __syncthreads();
            if (i0 + UT < i_end) fetch(i0 + UT);
            f64x4 acc[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[b] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < KP / 4; ++ks) {
                const int c = ks * 4 + (lane >> 4);
                const double a = Ul[c * UPITCH + wave * 16 + (lane & 15)];  // A[j][k = c] = U[c][j]
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const double bv = Tl[c * UPITCH + b * 16 + (lane & 15)];  // B[k = c][i]
                    acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv, acc[b], 0, 0, 0);
                }
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int64_t i = i0 + b * 16 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (orow[r] >= 0 && i < i_end) Mx[orow[r] * d + i] = acc[b][r];
            }
        }
    }
}

// End of a sweep, on the device: a sweep without a rotation is convergence.  state = {rotated, done, sweeps}.
__global__ void eigh_sweep_end_kernel(int* state) {
    if (state[1]) return;
    state[2] += 1;
    if (state[0] == 0) state[1] = 1;
    state[0] = 0;
}

// out[0] += ||W||_F^2 over this block's slice (out zeroed by the caller)
__global__ __launch_bounds__(EB) void frob2_kernel(double* out, const double* W, int64_t total) {
    __shared__ double scratch[4];
    double s = 0.0;
    for (int64_t i = blockIdx.x * static_cast<int64_t>(EB) + threadIdx.x; i < total; i += static_cast<int64_t>(gridDim.x) * EB)
        s += W[i] * W[i];
    s = block_sum(s, scratch);
    if (threadIdx.x == 0) atomicAdd(out, s);
}

// lambda_j = v_j . w_j
__global__ __launch_bounds__(EB) void rayleigh_kernel(double* lam, const double* Wt, const double* Vt, int64_t d) {
    __shared__ double scratch[4];
    const int64_t j = blockIdx.x;
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < d; i += EB) s += Wt[j * d + i] * Vt[j * d + i];
    s = block_sum(s, scratch);
    if (threadIdx.x == 0) lam[j] = s;
}

// rank_j = #{k : lam_k < lam_j or (lam_k == lam_j and k < j)}; evals[rank_j] = lam_j
__global__ void rank_kernel(int* rank, double* evals, const double* lam, int64_t d) {
    const int64_t j = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
    if (j >= d) return;
    const double x = lam[j];
    int r = 0;
    for (int64_t k = 0; k < d; ++k) {
        const double y = lam[k];
        r += (y < x || (y == x && k < j)) ? 1 : 0;
    }
    rank[j] = r;
    evals[r] = x;
}

// evecs[i, rank_j] = Vt[j, i]  (eigenvectors in columns, ascending eigenvalue order)
__global__ void scatter_vectors_kernel(double* evecs, const double* Vt, const int* rank, int64_t d) {
    __shared__ double tile[32][33];
    const int64_t j0 = blockIdx.y * 32, i0 = blockIdx.x * 32;
    const int tx = threadIdx.x, ty = threadIdx.y;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int64_t j = j0 + r, i = i0 + tx;
        tile[r][tx] = (j < d && i < d) ? Vt[j * d + i] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int64_t i = i0 + r, j = j0 + tx;
        if (i < d && j < d) evecs[i * d + rank[j]] = tile[tx][r];
    }
}

// ------------------------------------------------------------------------------------------------
// Factor-first solver (Veselic-Hari), the default for d >= 256 (KF_EIGH_CHOLESKY=0 disables it).  Designed on the CPU prototype
// tools/eigh_jacobi_proto.py (7-13 sweeps where the solver above needs 17-25, 40 % fewer bytes per round); on the MI355X
// d = 3073: 10 sweeps / 200 ms against 20 / 479 ms, 12 problems on 2 lanes 1.54 s against 4.5 s
// (profiles/r03_eigh_factor_first.log):   S' = P (S + shift I) P^T = R^T R  (columns sorted by decreasing diagonal, upper Cholesky, right-looking, 64 rows per
// step);  then the SAME blocked one-sided Jacobi on the rows of R (= the columns of L = R^T) WITHOUT V:  L V = U Sigma, so
// S' = U Sigma^2 U^T -- eigenvectors = normalised rows of the rotated Wt (scattered back through P), eigenvalues = squared row
// norms - shift.  A non-positive pivot (shift too small for this matrix) sets a flag and the caller falls back to the solver
// above.  Wt holds the working matrix: upper triangle of S' at first, R when the factorisation is done, zeros below the diagonal.
// ------------------------------------------------------------------------------------------------
constexpr int CB = 64;   // rows per factorisation step (= KP, so the panel transform is the update kernel's product)

__device__ __forceinline__ double cov_entry(const void* cov, int is_f64, double count, int64_t d, int64_t i, int64_t j) {
    double a, b;
    if (is_f64) { a = reinterpret_cast<const double*>(cov)[i * d + j]; b = reinterpret_cast<const double*>(cov)[j * d + i]; }
    else { a = reinterpret_cast<const float*>(cov)[i * d + j]; b = reinterpret_cast<const float*>(cov)[j * d + i]; }
    return 0.5 * (a / count + b / count);   // the reference's op order (eigen.py:198-203), as eigh_init_kernel
}

// diag[j] = S[j][j];  out[0] += ||S||_F^2 (out zeroed by the caller)
__global__ __launch_bounds__(EB) void chol_scan_kernel(double* diag, double* frob2, const void* cov, int is_f64, double count, int64_t d) {
    __shared__ double scratch[4];
    double s = 0.0;
    const int64_t total = d * d;
    for (int64_t e = blockIdx.x * static_cast<int64_t>(EB) + threadIdx.x; e < total; e += static_cast<int64_t>(gridDim.x) * EB) {
        const int64_t i = e / d, j = e % d;
        const double x = cov_entry(cov, is_f64, count, d, i, j);
        s += x * x;
        if (i == j) diag[i] = x;
    }
    s = block_sum(s, scratch);
    if (threadIdx.x == 0) atomicAdd(frob2, s);
}

// order[r] = j for the r-th LARGEST diagonal entry (ties by index)
__global__ void chol_order_kernel(int* order, const double* diag, int64_t d) {
    const int64_t j = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x;
    if (j >= d) return;
    const double x = diag[j];
    int r = 0;
    for (int64_t k = 0; k < d; ++k) {
        const double y = diag[k];
        r += (y > x || (y == x && k < j)) ? 1 : 0;
    }
    order[r] = static_cast<int>(j);
}

// Wt[a][b] = S[order[a]][order[b]] (+ shift on the diagonal) for b >= a, 0 below the diagonal;  shift = shift_scale * ||S||_F
__global__ void chol_init_kernel(double* Wt, const void* cov, int is_f64, double count, int64_t d, const int* order, const double* frob2,
                                 double shift_scale) {
    const int64_t total = d * d;
    const double shift = shift_scale * sqrt(frob2[0]);
    for (int64_t e = blockIdx.x * static_cast<int64_t>(blockDim.x) + threadIdx.x; e < total; e += static_cast<int64_t>(gridDim.x) * blockDim.x) {
        const int64_t a = e / d, b = e % d;
        double x = 0.0;
        if (b >= a) x = cov_entry(cov, is_f64, count, d, order[a], order[b]) + (a == b ? shift : 0.0);
        Wt[e] = x;
    }
}

// Step k, diagonal block: D = Wt[r0 .. r0+63][r0 .. r0+63] (upper part; rows / columns >= d count as identity) -> R_kk (upper
// Cholesky factor, written back with zeros below the diagonal) and U = R_kk^-1 (64 x 64, row-major [c][j]) for the panel
// transform.  One workgroup.  fail[0] = 1 when a pivot is not positive.
constexpr int CHOL_BLOCK_LDS = 2 * CB * (CB + 1) * static_cast<int>(sizeof(double));
__global__ __launch_bounds__(256) void chol_block_kernel(double* __restrict__ Wt, int64_t d, int k, double* __restrict__ U, int* __restrict__ fail) {
    constexpr int LP = CB + 1;
    extern __shared__ double lds[];   // CHOL_BLOCK_LDS bytes
    double* D = lds;
    double* X = lds + CB * LP;
    __shared__ int bad;
    const int tid = threadIdx.x;
    const int64_t r0 = static_cast<int64_t>(k) * CB;
    const int n = static_cast<int>(min<int64_t>(CB, d - r0));
    if (tid == 0) bad = 0;
    for (int e = tid; e < CB * CB; e += 256) {
        const int r = e / CB, c = e % CB;
        double x = (r == c) ? 1.0 : 0.0;
        if (r < n && c < n) x = Wt[(r0 + min(r, c)) * d + r0 + max(r, c)];   // the upper part, mirrored
        D[r * LP + c] = x;
        X[r * LP + c] = 0.0;
    }
    __syncthreads();
    // right-looking inside the block: row j becomes row j of R, the trailing part loses its outer product
    for (int j = 0; j < CB; ++j) {
        const double pivot = D[j * LP + j];
        if (!(pivot > 0.0)) { if (tid == 0) bad = 1; }
        const double root = sqrt(pivot > 0.0 ? pivot : 1.0);
        __syncthreads();
        if (tid > j && tid < CB) D[j * LP + tid] /= root;
        if (tid == 0) D[j * LP + j] = root;
        __syncthreads();
        for (int e = tid; e < CB * CB; e += 256) {
            const int r = e / CB, c = e % CB;
            if (r > j && c >= r) D[r * LP + c] -= D[j * LP + r] * D[j * LP + c];
        }
        __syncthreads();
    }
    // X = R^-1 (upper triangular), one thread per column, back substitution
    if (tid < CB) {
        const int j = tid;
        X[j * LP + j] = 1.0 / D[j * LP + j];
        for (int i = j - 1; i >= 0; --i) {
            double acc = 0.0;
            for (int m = i + 1; m <= j; ++m) acc += D[i * LP + m] * X[m * LP + j];
            X[i * LP + j] = -acc / D[i * LP + i];
        }
    }
    __syncthreads();
    for (int e = tid; e < CB * CB; e += 256) {
        const int r = e / CB, c = e % CB;
        U[e] = X[r * LP + c];
        if (r < n && c < n) Wt[(r0 + r) * d + r0 + c] = c >= r ? D[r * LP + c] : 0.0;
    }
    if (tid == 0 && bad) fail[0] = 1;
}

// Step k, panel row: Wt[r0 + j][i] <- sum_c U[c][j] Wt[r0 + c][i] for i in [i_lo, d) (= R_kk^-T A[k, i]); the product of
// eigh_update_kernel on 64 consecutive rows.  grid.x = column chunks of `chunk` (multiple of 64).
__global__ __launch_bounds__(256) void chol_panel_kernel(double* __restrict__ Wt, const double* __restrict__ U, int64_t d, int k, int64_t chunk) {
    extern __shared__ double lds[];
    double* Ul = lds;                 // [64 c][UPITCH]  (j contiguous)
    double* Tl = lds + CB * UPITCH;   // [64 c][UPITCH]  (i contiguous)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t r0 = static_cast<int64_t>(k) * CB, i_lo = r0 + CB;
    const int n = static_cast<int>(min<int64_t>(CB, d - r0));
    for (int e = tid; e < CB * CB; e += 256) Ul[(e / CB) * UPITCH + (e % CB)] = U[e];
    const int lcol = tid & 15, lr0 = tid >> 4;   // loader as eigh_update_kernel's: 16 lanes along a row, rows lr0 + 16 p, columns lcol + 16 c
    const int64_t i_begin = i_lo + blockIdx.x * chunk, i_end = min(d, i_begin + chunk);
    for (int64_t i0 = i_begin; i0 < i_end; i0 += UT) {
        double v[16];
#pragma unroll
        for (int pp = 0; pp < 4; ++pp)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int64_t i = i0 + lcol + 16 * c;
                const int row = lr0 + 16 * pp;
                v[pp * 4 + c] = (row < n && i < i_end) ? Wt[(r0 + row) * d + i] : 0.0;
            }
        __syncthreads();  // previous tile consumed (and U staged, first time round)
#pragma unroll
        for (int pp = 0; pp < 4; ++pp)
#pragma unroll
            for (int c = 0; c < 4; ++c) Tl[(lr0 + 16 * pp) * UPITCH + lcol + 16 * c] = v[pp * 4 + c];
        __syncthreads();
        f64x4 acc[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[b] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < CB / 4; ++ks) {
            const int c = ks * 4 + (lane >> 4);
            const double a = Ul[c * UPITCH + wave * 16 + (lane & 15)];  // A[j][k = c] = U[c][j]
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const double bv = Tl[c * UPITCH + b * 16 + (lane & 15)];  // B[k = c][i]
                acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv, acc[b], 0, 0, 0);
            }
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int64_t i = i0 + b * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = wave * 16 + (lane >> 4) + 4 * r;
                if (j < n && i < i_end) Wt[(r0 + j) * d + i] = acc[b][r];
            }
        }
    }
}

// Step k, trailing update: Wt[i][j] -= sum_c Wt[r0 + c][i] Wt[r0 + c][j] for the 64 x 64 tiles (ti <= tj) of the part behind the
// step (a diagonal tile is updated whole: what lands below the diagonal inside it is overwritten with zeros when that block is
// factored).  grid = (tiles, tiles); tiles with tj < ti return.
__global__ __launch_bounds__(256) void chol_trailing_kernel(double* __restrict__ Wt, int64_t d, int k) {
    extern __shared__ double lds[];   // UPDATE_LDS bytes
    double* Al = lds;                 // [c][i]
    double* Bl = lds + CB * UPITCH;   // [c][j]
    if (blockIdx.y < blockIdx.x) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t r0 = static_cast<int64_t>(k) * CB, c0 = r0 + CB;
    const int n = static_cast<int>(min<int64_t>(CB, d - r0));
    const int64_t i0 = c0 + static_cast<int64_t>(blockIdx.x) * CB, j0 = c0 + static_cast<int64_t>(blockIdx.y) * CB;
    const int lcol = tid & 15, lr0 = tid >> 4;   // 16 lanes along a row (coalesced: 4 rows x 128 bytes per wave instruction)
#pragma unroll
    for (int pp = 0; pp < 4; ++pp)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int row = lr0 + 16 * pp, col = lcol + 16 * c;
            const int64_t i = i0 + col, j = j0 + col;
            Al[row * UPITCH + col] = (row < n && i < d) ? Wt[(r0 + row) * d + i] : 0.0;
            Bl[row * UPITCH + col] = (row < n && j < d) ? Wt[(r0 + row) * d + j] : 0.0;
        }
    __syncthreads();
    f64x4 acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[b] = f64x4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < CB / 4; ++ks) {
        const int c = ks * 4 + (lane >> 4);
        const double a = Al[c * UPITCH + wave * 16 + (lane & 15)];   // A[i][k = c]
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const double bv = Bl[c * UPITCH + b * 16 + (lane & 15)];   // B[k = c][j]
            acc[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv, acc[b], 0, 0, 0);
        }
    }
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int64_t j = j0 + b * 16 + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t i = i0 + wave * 16 + (lane >> 4) + 4 * r;
            if (i < d && j < d) Wt[i * d + j] -= acc[b][r];
        }
    }
}

// sigma2[j] = ||row j of Wt||^2;  lam[j] = sigma2[j] - shift
__global__ __launch_bounds__(EB) void chol_norms_kernel(double* lam, double* sigma2, const double* Wt, int64_t d, const double* frob2,
                                                        double shift_scale) {
    __shared__ double scratch[4];
    const int64_t j = blockIdx.x;
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < d; i += EB) s += Wt[j * d + i] * Wt[j * d + i];
    s = block_sum(s, scratch);
    if (threadIdx.x == 0) { sigma2[j] = s; lam[j] = s - shift_scale * sqrt(frob2[0]); }
}

// evecs[order[i], rank_j] = Wt[j, i] / sqrt(sigma2[j])   (eigenvectors in columns, ascending eigenvalues, original row order)
__global__ void chol_scatter_kernel(double* evecs, const double* Wt, const int* rank, const int* order, const double* sigma2, int64_t d) {
    __shared__ double tile[32][33];
    const int64_t j0 = blockIdx.y * 32, i0 = blockIdx.x * 32;
    const int tx = threadIdx.x, ty = threadIdx.y;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int64_t j = j0 + r, i = i0 + tx;
        tile[r][tx] = (j < d && i < d) ? Wt[j * d + i] / sqrt(sigma2[j]) : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int64_t i = i0 + r, j = j0 + tx;
        if (i < d && j < d) evecs[static_cast<int64_t>(order[i]) * d + rank[j]] = tile[tx][r];
    }
}

// ------------------------------------------------------------------------------------------------
// Batched small symmetric eigensolver, one workgroup per matrix, everything in LDS (l <= 96).
// Used by the low-rank query factorisation (module/tracker/precondition.py:19-75 of the reference,
// torch.linalg.svd / svd_lowrank there): the l x l Gram matrices of the randomised range finder.
// Same one-sided Jacobi as above -- W = G V kept in fp64, tournament rounds, 8 lanes per column
// pair -- but a round is a __syncthreads() instead of a kernel launch.
// ------------------------------------------------------------------------------------------------
constexpr int SMALL_MAX = 96;

__global__ __launch_bounds__(256) void eigh_small_kernel(const float* G, int l, float* evals, float* evecs, int inv_sqrt,
                                                         float floor_rel, int max_sweeps) {
    extern __shared__ double lds[];
    const int ld = l | 1;
    double* W = lds;
    double* V = W + l * ld;
    double* lam = V + l * ld;
    int* rank = reinterpret_cast<int*>(lam + l);
    __shared__ double red[4];
    __shared__ double lam_max;
    const int tid = threadIdx.x, grp = tid >> 3, sub = tid & 7;
    const float* g = G + static_cast<int64_t>(blockIdx.x) * l * l;
    double f2 = 0.0;
    for (int e = tid; e < l * l; e += 256) {
        const int j = e / l, i = e % l;
        const double x = 0.5 * (static_cast<double>(g[j * l + i]) + static_cast<double>(g[i * l + j]));
        W[j * ld + i] = x;
        V[j * ld + i] = (i == j) ? 1.0 : 0.0;
        f2 += x * x;
    }
    f2 = block_sum(f2, red);
    const double eps = 2.220446049250313e-16;
    const double null2 = f2 * eps * eps * l, tol = 4.0 * eps * sqrt(static_cast<double>(l));
    const int npl = l + (l & 1), pairs = npl / 2, m = npl - 1;
    __syncthreads();
    for (int sweep = 0; sweep < max_sweeps && l > 1; ++sweep) {
        int rotated = 0;
        for (int round = 0; round < m; ++round) {
            for (int k = grp; k < pairs; k += 32) {
                int p, q;
                if (k == 0) { p = round % m; q = m; }
                else { p = (round + k) % m; q = (round - k + m) % m; }
                const bool live = p < l && q < l;  // the bye of an odd l
                double a = 0.0, b = 0.0, c = 0.0;
                if (live)
                    for (int i = sub; i < l; i += 8) {
                        const double x = W[p * ld + i], y = W[q * ld + i];
                        a += x * x; b += y * y; c += x * y;
                    }
                for (int off = 1; off < 8; off <<= 1) {
                    a += __shfl_xor(a, off); b += __shfl_xor(b, off); c += __shfl_xor(c, off);
                }
                if (live && fabs(c) > tol * sqrt(a) * sqrt(b) && a > null2 && b > null2) {
                    const double zeta = (b - a) / (2.0 * c);
                    const double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                    const double cs = 1.0 / sqrt(1.0 + t * t), sn = cs * t;
                    for (int i = sub; i < l; i += 8) {
                        const double x = W[p * ld + i], y = W[q * ld + i];
                        W[p * ld + i] = cs * x - sn * y; W[q * ld + i] = sn * x + cs * y;
                        const double u = V[p * ld + i], v = V[q * ld + i];
                        V[p * ld + i] = cs * u - sn * v; V[q * ld + i] = sn * u + cs * v;
                    }
                    rotated = 1;
                }
            }
            __syncthreads();
        }
        if (!__syncthreads_or(rotated)) break;
    }
    for (int j = grp; j < l; j += 32) {  // Rayleigh quotients
        double r = 0.0;
        for (int i = sub; i < l; i += 8) r += V[j * ld + i] * W[j * ld + i];
        for (int off = 1; off < 8; off <<= 1) r += __shfl_xor(r, off);
        if (sub == 0) lam[j] = r;
    }
    __syncthreads();
    if (tid < l) {  // descending order
        const double x = lam[tid];
        int r = 0;
        for (int k = 0; k < l; ++k) r += (lam[k] > x || (lam[k] == x && k < tid)) ? 1 : 0;
        rank[tid] = r;
        if (r == 0) lam_max = x;
        evals[static_cast<int64_t>(blockIdx.x) * l + r] = static_cast<float>(x);
    }
    __syncthreads();
    float* out = evecs + static_cast<int64_t>(blockIdx.x) * l * l;
    const double floor_value = static_cast<double>(floor_rel) * fmax(lam_max, 0.0);
    for (int e = tid; e < l * l; e += 256) {
        const int j = e / l, i = e % l;
        double scale = 1.0;
        if (inv_sqrt) {
            const double x = fmax(lam[j], floor_value);
            scale = x > 0.0 ? 1.0 / sqrt(x) : 0.0;
        }
        out[i * l + rank[j]] = static_cast<float>(V[j * ld + i] * scale);
    }
}

}  // namespace

extern "C" {

namespace {
struct BlockPlan { int nblocks, players, pairs, gsplit, usplit; int64_t gchunk, uchunk; };
BlockPlan block_plan(int64_t d) {
    BlockPlan p;
    p.nblocks = static_cast<int>((d + KB - 1) / KB);
    p.players = p.nblocks + (p.nblocks & 1);
    p.pairs = p.players / 2;
    // ~2 workgroups per CU per kernel: the pairs of a round alone (d / 64) would leave most of the 256 CUs idle
    static const int64_t target = [] { const char* e = getenv("KF_EIGH_WGS"); return e ? std::max<int64_t>(64, atoll(e)) : 512; }();   // (measurements)
    const int64_t want = std::max<int64_t>(1, (target + p.pairs - 1) / p.pairs);
    int64_t gs = std::max<int64_t>(1, std::min<int64_t>(want, (d + 255) / 256));
    p.gchunk = ((d + gs - 1) / gs + GK - 1) / GK * GK;
    p.gsplit = static_cast<int>((d + p.gchunk - 1) / p.gchunk);
    int64_t us = std::max<int64_t>(1, std::min<int64_t>(want, (d + 255) / 256));
    p.uchunk = ((d + us - 1) / us + UT - 1) / UT * UT;
    p.usplit = static_cast<int>((d + p.uchunk - 1) / p.uchunk);
    return p;
}
constexpr int UPDATE_LDS = 2 * KP * UPITCH * static_cast<int>(sizeof(double));
constexpr int64_t BLOCKED_MIN_D = 256;

int configure_eigh() {
    static std::once_flag flag;
    static int status = KF_OK;
    std::call_once(flag, [] {
        const int ld_max = SMALL_MAX | 1;
        const size_t small_bytes = sizeof(double) * (2 * static_cast<size_t>(SMALL_MAX) * ld_max + SMALL_MAX) + sizeof(int) * SMALL_MAX + 16;
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(eigh_small_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                static_cast<int>(small_bytes)) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(eigh_update_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                UPDATE_LDS) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(chol_panel_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                UPDATE_LDS) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(chol_trailing_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                UPDATE_LDS) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(chol_block_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                CHOL_BLOCK_LDS) != hipSuccess)
            status = KF_ERR_LAUNCH_FAILED;
    });
    return status;
}
}  // namespace

int64_t kf_eigh_workspace_bytes(int64_t d) {
    // Wt, Vt (d*d doubles each), lam (d doubles), rank (d ints), flags; for the blocked solver the per-pair Gram
    // partials, rotations and flags of one round; padded
    int64_t bytes = static_cast<int64_t>(sizeof(double)) * (2 * d * d + d) + static_cast<int64_t>(sizeof(int)) * (d + 16) + 512;
    if (d >= BLOCKED_MIN_D) {
        const BlockPlan p = block_plan(d);
        bytes += static_cast<int64_t>(sizeof(double)) * KP * KP * p.pairs * (p.gsplit + 1) + static_cast<int64_t>(sizeof(int)) * (p.pairs + 16) + 512;
    }
    return bytes;
}

namespace {
std::atomic<int64_t> g_factor_first{0}, g_fallback{0}, g_retries{0};
}

void kf_eigh_stats(int64_t* factor_first, int64_t* fallback, int64_t* retries, int reset) {
    if (factor_first) *factor_first = g_factor_first.load();
    if (fallback) *fallback = g_fallback.load();
    if (retries) *retries = g_retries.load();
    if (reset) { g_factor_first = 0; g_fallback = 0; g_retries = 0; }
}

int kf_eigh_f64(const void* cov, int cov_dtype, double count, double noise_rel, int64_t d, double* evals, double* evecs, void* workspace,
                int64_t workspace_bytes, int max_sweeps, int* sweeps_done, void* stream) {
    if (!cov || !evals || !evecs || !workspace || d <= 0 || !(count > 0.0)) return KF_ERR_INVALID_ARGUMENT;
    if (cov_dtype != KF_F32 && cov_dtype != KF_F64) return KF_ERR_UNSUPPORTED_DTYPE;
    if (workspace_bytes < kf_eigh_workspace_bytes(d)) return KF_ERR_WORKSPACE_TOO_SMALL;
    if (d >= (1 << 24)) return KF_ERR_INVALID_ARGUMENT;
    if (configure_eigh() != KF_OK) return KF_ERR_LAUNCH_FAILED;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (max_sweeps <= 0) max_sweeps = 100;
    double* Wt = reinterpret_cast<double*>(workspace);
    double* Vt = Wt + d * d;
    double* lam = Vt + d * d;
    int* rank = reinterpret_cast<int*>(lam + d);
    int* flag = rank + d + (d & 1);
    // 64-byte aligned tail: ||S||_F^2 on the device, then the blocked solver's per-round buffers
    char* tail = reinterpret_cast<char*>((reinterpret_cast<uintptr_t>(flag + 16) + 63) & ~static_cast<uintptr_t>(63));
    double* frob2_dev = reinterpret_cast<double*>(tail);

    const unsigned g = static_cast<unsigned>(std::max<int64_t>(1, std::min<int64_t>((d * d + 255) / 256, 4096)));
    const int is_f64 = cov_dtype == KF_F64 ? 1 : 0;
    // W = S, V = I and ||S||_F^2 (the threshold below which a column of W = S V counts as numerically null) for the solvers that
    // carry V; the factor-first solver below sets up its own working matrix and only comes here when it gives up
    auto init_with_v = [&]() -> bool {
        hipLaunchKernelGGL(eigh_init_kernel, dim3(g), dim3(256), 0, st, Wt, Vt, cov, is_f64, count, d);
        if (hipMemsetAsync(frob2_dev, 0, sizeof(double), st) != hipSuccess) return false;
        hipLaunchKernelGGL(frob2_kernel, dim3(static_cast<unsigned>(std::max<int64_t>(1, std::min<int64_t>(d * d / (EB * 16), 1024)))), dim3(EB), 0,
                           st, frob2_dev, Wt, d * d);
        return true;
    };
    bool initialised = false;
    const double eps = 2.220446049250313e-16;
    const double null_scale = (eps * eps) * static_cast<double>(d);
    const double tol = 4.0 * eps * sqrt(static_cast<double>(d));
    int sweeps = 0, status = KF_ERR_NOT_CONVERGED;
    const bool verbose = getenv("KF_EIGH_VERBOSE") != nullptr;
    if (d == 1) { status = KF_OK; }

    const char* factor_first_env = getenv("KF_EIGH_CHOLESKY");   // "0": the solver that carries V, from the start
    if (d >= BLOCKED_MIN_D && !(factor_first_env && atoi(factor_first_env) == 0) && getenv("KF_EIGH_SCALAR") == nullptr &&
        getenv("KF_EIGH_BLOCK8") == nullptr) {
        // ---- factor-first solver (see chol_* kernels): Cholesky of the diagonally sorted, shifted matrix, blocked Jacobi on the
        //      rows of R without V.  Scratch in the (unused) V area: diag / sigma2 (d doubles each), order (d ints).
        const BlockPlan p = block_plan(d);
        double* partial = frob2_dev + 8;
        double* Ubuf = partial + static_cast<int64_t>(KP) * KP * p.pairs * p.gsplit;
        int* pair_flag = reinterpret_cast<int*>(Ubuf + static_cast<int64_t>(KP) * KP * p.pairs);
        double* diag = Vt;
        double* sigma2 = Vt + d;
        int* order = reinterpret_cast<int*>(Vt + 2 * d);
        int* state = flag;   // {rotated, done, sweeps, cholesky failed}
        // The shift must exceed the most negative eigenvalue of the matrix AS STORED.  An exact fp64 covariance needs only the
        // factorisation's own rounding, 4 sqrt(d) eps ||S||_F; a rank-deficient covariance accumulated / stored in fp32 has noise
        // eigenvalues of ~ -1e-8 .. -1e-6 ||S||, one exported in bf16 of ~ -2^-9 ||S||: `noise_rel` (0: derived from the dtype)
        // says which.  The shift is added and subtracted in fp64 and the eigenvectors of S + shift I are those of S, so its
        // size costs no accuracy in the LAPACK sense (absolute, eps ||S|| / gap: tools/eigh_jacobi_proto.py) -- it
        // makes the factor BETTER conditioned.  A non-positive pivot retries with 32x the shift (a factorisation is a few
        // per cent of the solve), twice, before the solver that carries V takes over.
        const double noise = noise_rel > 0.0 ? noise_rel : (is_f64 ? 0.0 : 9.5367431640625e-07 /* 2^-20 */);
        const double shift_floor = 4.0 * sqrt(static_cast<double>(d)) * eps;
        if (hipMemsetAsync(frob2_dev, 0, sizeof(double), st) != hipSuccess || hipMemsetAsync(order, 0, d * sizeof(int), st) != hipSuccess)
            return KF_ERR_LAUNCH_FAILED;
        hipLaunchKernelGGL(chol_scan_kernel, dim3(static_cast<unsigned>(std::max<int64_t>(1, std::min<int64_t>(d * d / (EB * 16), 1024)))), dim3(EB), 0,
                           st, diag, frob2_dev, cov, is_f64, count, d);
        // (a NaN diagonal entry has no rank: `order` was zeroed above so that every slot holds a valid index whatever happens)
        hipLaunchKernelGGL(chol_order_kernel, dim3(static_cast<unsigned>((d + 255) / 256)), dim3(256), 0, st, order, diag, d);
        int host_state[4] = {0, 0, 0, 1};
        double shift_scale = std::max(shift_floor, noise);
        for (int attempt = 0; attempt < 3 && host_state[3]; ++attempt, shift_scale = std::min(shift_scale * 32.0, 0.05)) {
            if (hipMemsetAsync(state, 0, 4 * sizeof(int), st) != hipSuccess) return KF_ERR_LAUNCH_FAILED;
            hipLaunchKernelGGL(chol_init_kernel, dim3(g), dim3(256), 0, st, Wt, cov, is_f64, count, d, order, frob2_dev, shift_scale);
            const int steps = static_cast<int>((d + CB - 1) / CB);
            for (int k = 0; k < steps; ++k) {
                hipLaunchKernelGGL(chol_block_kernel, dim3(1), dim3(256), CHOL_BLOCK_LDS, st, Wt, d, k, Ubuf, state + 3);
                const int64_t rest = d - static_cast<int64_t>(k + 1) * CB;
                if (rest <= 0) break;
                const int64_t chunk = 256;   // columns per panel workgroup (4 tiles of 64)
                hipLaunchKernelGGL(chol_panel_kernel, dim3(static_cast<unsigned>((rest + chunk - 1) / chunk)), dim3(256), UPDATE_LDS, st, Wt, Ubuf, d, k, chunk);
                const unsigned tiles = static_cast<unsigned>((rest + CB - 1) / CB);
                hipLaunchKernelGGL(chol_trailing_kernel, dim3(tiles, tiles), dim3(256), UPDATE_LDS, st, Wt, d, k);
            }
            double frob2_host = 0.0;
            if (hipMemcpyAsync(host_state, state, 4 * sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess ||
                hipMemcpyAsync(&frob2_host, frob2_dev, sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess)
                return KF_ERR_LAUNCH_FAILED;
            if (hipStreamSynchronize(st) != hipSuccess) return KF_ERR_LAUNCH_FAILED;
            if (verbose)
                fprintf(stderr, "[kf_eigh] d=%lld factor-first: shift %.2e ||S||_F: cholesky %s\n", static_cast<long long>(d), shift_scale,
                        host_state[3] ? "FAILED" : "ok");
            if (!std::isfinite(frob2_host)) return KF_ERR_NOT_CONVERGED;   // NaN / Inf in the covariance (the reference's eigh raises)
            if (!host_state[3]) break;
            ++g_retries;
        }
        if (!host_state[3]) {
            ++g_factor_first;
            const int* done = state + 1;
            int enqueued = 0;
            while (enqueued < max_sweeps && !host_state[1]) {
                const int batch = 4;
                for (int b = 0; b < batch && enqueued < max_sweeps; ++b, ++enqueued) {
                    for (int r = -1; r < p.players - 1; ++r) {
                        const int pairing = r < 0 ? 0 : r;
                        hipLaunchKernelGGL(eigh_gram_kernel, dim3(p.pairs, p.gsplit), dim3(256), 0, st, Wt, partial, d, p.nblocks, p.players,
                                           pairing, p.gsplit, p.gchunk, done);
                        // null_scale 0: every column of R has norm >= sqrt(shift) -- no column is "null" here
                        hipLaunchKernelGGL(eigh_solve_kernel, dim3(p.pairs), dim3(256), 0, st, partial, Ubuf, pair_flag, p.gsplit, r < 0 ? 0 : 1,
                                           tol, frob2_dev, 0.0, state, done);
                        hipLaunchKernelGGL(eigh_update_kernel, dim3(p.pairs, p.usplit), dim3(256), UPDATE_LDS, st, Wt, Vt, Ubuf, pair_flag, d,
                                           p.nblocks, p.players, pairing, p.uchunk, done, 0);
                    }
                    hipLaunchKernelGGL(eigh_sweep_end_kernel, dim3(1), dim3(1), 0, st, state);
                }
                if (hipMemcpyAsync(host_state, state, 3 * sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess) return KF_ERR_LAUNCH_FAILED;
                if (hipStreamSynchronize(st) != hipSuccess) return KF_ERR_LAUNCH_FAILED;
                if (verbose) fprintf(stderr, "[kf_eigh] d=%lld factor-first: %d sweeps run, done=%d\n", static_cast<long long>(d), host_state[2], host_state[1]);
            }
            if (sweeps_done) *sweeps_done = host_state[2];
            hipLaunchKernelGGL(chol_norms_kernel, dim3(static_cast<unsigned>(d)), dim3(EB), 0, st, lam, sigma2, Wt, d, frob2_dev, shift_scale);
            hipLaunchKernelGGL(rank_kernel, dim3(static_cast<unsigned>((d + 255) / 256)), dim3(256), 0, st, rank, evals, lam, d);
            const unsigned tt = static_cast<unsigned>((d + 31) / 32);
            hipLaunchKernelGGL(chol_scatter_kernel, dim3(tt, tt), dim3(32, 8), 0, st, evecs, Wt, rank, order, sigma2, d);
            if (hipGetLastError() != hipSuccess) return KF_ERR_LAUNCH_FAILED;
            return host_state[1] ? KF_OK : KF_ERR_NOT_CONVERGED;
        }
        // three non-positive pivots: start over with the solver that carries V
        ++g_fallback;
    }
    if (!initialised && !init_with_v()) return KF_ERR_LAUNCH_FAILED;
    initialised = true;
    if (d >= BLOCKED_MIN_D && getenv("KF_EIGH_SCALAR") == nullptr && getenv("KF_EIGH_BLOCK8") == nullptr) {
        // ---- blocked solver on the fp64 matrix cores; ||S||_F^2 stays on the device (no host read-back up front)
        const BlockPlan p = block_plan(d);
        double* partial = frob2_dev + 8;
        double* Ubuf = partial + static_cast<int64_t>(KP) * KP * p.pairs * p.gsplit;
        int* pair_flag = reinterpret_cast<int*>(Ubuf + static_cast<int64_t>(KP) * KP * p.pairs);
        // Convergence is decided ON THE DEVICE (eigh_sweep_end_kernel); the host enqueues sweeps in batches and reads the
        // state back once per batch -- the kernels of sweeps enqueued past convergence exit at their first instruction.
        int* state = flag;  // {rotated, done, sweeps}: three of the 16 spare ints behind `rank`
        if (hipMemsetAsync(state, 0, 3 * sizeof(int), st) != hipSuccess) return KF_ERR_LAUNCH_FAILED;
        const int* done = state + 1;
        int enqueued = 0, host_state[3] = {0, 0, 0};
        while (enqueued < max_sweeps && !host_state[1]) {
            const int batch = enqueued < 8 ? 8 : 4;  // nothing converges in under 8 sweeps at these sizes
            for (int b = 0; b < batch && enqueued < max_sweeps; ++b, ++enqueued) {
                // pass -1: the column pairs inside every block (pairing of round 0); then the tournament of block pairs
                for (int r = -1; r < p.players - 1; ++r) {
                    const int pairing = r < 0 ? 0 : r;
                    hipLaunchKernelGGL(eigh_gram_kernel, dim3(p.pairs, p.gsplit), dim3(256), 0, st, Wt, partial, d, p.nblocks, p.players,
                                       pairing, p.gsplit, p.gchunk, done);
                    hipLaunchKernelGGL(eigh_solve_kernel, dim3(p.pairs), dim3(256), 0, st, partial, Ubuf, pair_flag, p.gsplit, r < 0 ? 0 : 1,
                                       tol, frob2_dev, null_scale, state, done);
                    hipLaunchKernelGGL(eigh_update_kernel, dim3(p.pairs, p.usplit), dim3(256), UPDATE_LDS, st, Wt, Vt, Ubuf, pair_flag, d,
                                       p.nblocks, p.players, pairing, p.uchunk, done, 1);
                }
                hipLaunchKernelGGL(eigh_sweep_end_kernel, dim3(1), dim3(1), 0, st, state);
            }
            if (hipMemcpyAsync(host_state, state, 3 * sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess) return KF_ERR_LAUNCH_FAILED;
            if (hipStreamSynchronize(st) != hipSuccess) return KF_ERR_LAUNCH_FAILED;
            if (verbose) fprintf(stderr, "[kf_eigh] d=%lld blocked: %d sweeps run, done=%d\n", static_cast<long long>(d), host_state[2], host_state[1]);
        }
        sweeps = host_state[2];
        if (host_state[1]) status = KF_OK;
    } else if (d > 1) {
        double frob2 = 0.0;
        if (hipMemcpyAsync(&frob2, frob2_dev, sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess) return KF_ERR_LAUNCH_FAILED;
        if (hipStreamSynchronize(st) != hipSuccess) return KF_ERR_LAUNCH_FAILED;
        const double null2 = frob2 * null_scale;
        const int npl = static_cast<int>(d + (d & 1));
        const int pairs = npl / 2, rounds = npl - 1;
        // block rounds of 8 columns (jacobi_block_round_kernel) from d = 512; KF_EIGH_SCALAR=1 forces the scalar kernel
        const int nblocks = static_cast<int>((d + BS - 1) / BS);
        const int block_players = nblocks + (nblocks & 1);
        const bool use_blocks = d >= 512 && getenv("KF_EIGH_SCALAR") == nullptr;
        for (; sweeps < max_sweeps; ++sweeps) {
            if (hipMemsetAsync(flag, 0, sizeof(int), st) != hipSuccess) return KF_ERR_LAUNCH_FAILED;
            if (use_blocks) {
                for (int r = 0; r < block_players - 1; ++r)
                    hipLaunchKernelGGL(jacobi_block_round_kernel, dim3(block_players / 2), dim3(EB), 0, st, Wt, Vt, d, nblocks,
                                       block_players, r, tol, null2, flag);
            } else {
                for (int r = 0; r < rounds; ++r)
                    hipLaunchKernelGGL(jacobi_round_kernel, dim3(pairs), dim3(EB), 0, st, Wt, Vt, d, npl, r, tol, null2, flag);
            }
            int host_flag = 1;
            if (hipMemcpyAsync(&host_flag, flag, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess) return KF_ERR_LAUNCH_FAILED;
            if (hipStreamSynchronize(st) != hipSuccess) return KF_ERR_LAUNCH_FAILED;
            if (verbose) fprintf(stderr, "[kf_eigh] d=%lld sweep %d: %d rotations\n", static_cast<long long>(d), sweeps, host_flag);
            if (host_flag == 0) { status = KF_OK; ++sweeps; break; }
        }
    }
    if (sweeps_done) *sweeps_done = sweeps;
    hipLaunchKernelGGL(rayleigh_kernel, dim3(static_cast<unsigned>(d)), dim3(EB), 0, st, lam, Wt, Vt, d);
    hipLaunchKernelGGL(rank_kernel, dim3(static_cast<unsigned>((d + 255) / 256)), dim3(256), 0, st, rank, evals, lam, d);
    const unsigned t = static_cast<unsigned>((d + 31) / 32);
    hipLaunchKernelGGL(scatter_vectors_kernel, dim3(t, t), dim3(32, 8), 0, st, evecs, Vt, rank, d);
    if (hipGetLastError() != hipSuccess) return KF_ERR_LAUNCH_FAILED;
    return status;
}

int kf_eigh_small_batched(const float* G, int64_t batch, int l, float* evals, float* evecs, int inv_sqrt, float floor_rel,
                          int max_sweeps, void* stream) {
    if (!G || !evals || !evecs || batch < 0 || l <= 0) return KF_ERR_INVALID_ARGUMENT;
    if (l > SMALL_MAX) return KF_ERR_INVALID_ARGUMENT;
    if (batch == 0) return KF_OK;
    if (max_sweeps <= 0) max_sweeps = 60;
    const int ld = l | 1;
    const size_t bytes = sizeof(double) * (2 * static_cast<size_t>(l) * ld + l) + sizeof(int) * l + 16;
    if (configure_eigh() != KF_OK) return KF_ERR_LAUNCH_FAILED;
    hipLaunchKernelGGL(eigh_small_kernel, dim3(static_cast<unsigned>(batch)), dim3(256), bytes,
                       reinterpret_cast<hipStream_t>(stream), G, l, evals, evecs, inv_sqrt, floor_rel, max_sweeps);
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) {
        fprintf(stderr, "[kf_eigh_small] launch (l=%d, lds=%zu): %s\n", l, bytes, hipGetErrorString(err));
        return KF_ERR_LAUNCH_FAILED;
    }
    return KF_OK;
}

}  // extern "C"
