"""CPU prototype (numpy, fp64, no GPU) of the eigensolver's ALGORITHM -- one-sided block Jacobi exactly as csrc/kf_eigh.hip runs it
(32-column blocks, round-robin tournament of block pairs, per pair the 64 x 64 Gram matrix and one cyclic pass of two-sided
rotations, in-block pairs once per sweep, relative rotation test 4 eps sqrt(d), null-column test) -- to count SWEEPS under
algorithmic variants.  The GPU solver's time is (sweeps) x (rounds per sweep) x (bytes per round), so a sweep count measured here
transfers; what was found (profiles/r03_eigh_cholesky_jacobi_prototype.log):

  * the dense, graded spectra of covariance matrices keep the method in its LINEAR phase for most of its 16-22 sweeps -- the
    relative off-diagonal of the small columns stays at 0.2-0.3 until the absolute off-norm is already at 1e-11;
  * sorting the columns by diagonal once at the start saves ~2 sweeps on unsorted inputs, two inner passes per pair ~3;
  * factoring first -- S + shift I = L L^T (Cholesky of the diagonally sorted matrix), then the SAME block Jacobi on the columns
    of L WITHOUT accumulating V (Veselic-Hari: L V = U Sigma, S = U Sigma^2 U^T, eigenvectors = normalised columns, eigenvalues
    = squared column norms - shift) -- converges quadratically after 4-5 sweeps: 8-9 sweeps instead of 17-22 on graded spectra,
    13 instead of 21 on a power-law spectrum with dense eigenvectors, with BETTER orthogonality (2e-14 vs 4e-13) and 40 % fewer
    bytes per round (no V).  Large exact null spaces (n < d samples) still take ~16 sweeps.

    python tools/eigh_jacobi_proto.py table [d]        the comparison table (d = 512: ~6 minutes on one core)
    python tools/eigh_jacobi_proto.py one d variant    variant in {base, sort, inner2, chol, chol_inner2}, sweep-by-sweep trace
"""
import sys
import time

import numpy as np

KB = 32
EPS = 2.220446049250313e-16


def rotation(a, b, g):
    """(c, s) of the Jacobi rotation that annihilates g in [[a, g], [g, b]] -- the formula of eigh_solve_kernel."""
    w, h = b - a, 2.0 * g
    t = np.where(w >= 0, 1.0, -1.0) * h / (np.abs(w) + np.sqrt(w * w + h * h))
    c = 1.0 / np.sqrt(1.0 + t * t)
    return c, c * t


def solve_pair(gram, cross, tol, null2, passes=1):
    """Cyclic two-sided Jacobi on the Gram matrix of a block pair, disjoint rotations of a round applied at once
    (cross: p in the first block, q in the second, 32 rounds; else the pairs inside each block, 31 rounds).  -> (U, rotated)."""
    n = gram.shape[0]
    h = n // 2
    u = np.eye(n)
    did = False
    idx = np.arange(h)
    for _ in range(passes):
        for r in range(h if cross else h - 1):
            if cross:
                p, q = idx, h + (idx + r) % h
            else:
                m, kk = h - 1, np.arange(h // 2)
                pp = np.where(kk == 0, r % m, (r + kk) % m)
                qq = np.where(kk == 0, m, (r - kk + m) % m)
                p, q = np.concatenate([pp, h + pp]), np.concatenate([qq, h + qq])
            a, b, g = gram[p, p], gram[q, q], gram[p, q]
            act = (g * g > tol * tol * a * b) & (a > null2) & (b > null2)
            if not act.any():
                continue
            did = True
            c, s = np.ones_like(a), np.zeros_like(a)
            c[act], s[act] = rotation(a[act], b[act], g[act])
            for mat in (gram, u):   # columns
                mp, mq = mat[:, p].copy(), mat[:, q].copy()
                mat[:, p], mat[:, q] = c * mp - s * mq, s * mp + c * mq
            gp, gq = gram[p, :].copy(), gram[q, :].copy()   # rows
            gram[p, :], gram[q, :] = c[:, None] * gp - s[:, None] * gq, s[:, None] * gp + c[:, None] * gq
    return u, did


def pairing(k, r, players):
    m = players - 1
    return (r % m, m) if k == 0 else ((r + k) % m, (r - k + m) % m)


def block_jacobi(w, v, tol, null2, passes, max_sweeps, trace):
    """The sweeps: columns of w (and of v when given) are rotated until a whole sweep passes without a rotation."""
    d = w.shape[1]
    nblocks = d // KB
    players = nblocks + (nblocks & 1)
    for sweep in range(max_sweeps):
        rotated = 0
        for r in range(-1, players - 1):
            for k in range(players // 2):
                pb, qb = pairing(k, max(r, 0), players)
                if pb >= nblocks or qb >= nblocks:
                    continue
                cols = np.concatenate([np.arange(pb * KB, (pb + 1) * KB), np.arange(qb * KB, (qb + 1) * KB)])
                wp = w[:, cols]
                u, did = solve_pair(wp.T @ wp, r >= 0, tol, null2, passes)
                if did:
                    rotated += 1
                    w[:, cols] = wp @ u
                    if v is not None:
                        v[:, cols] = v[:, cols] @ u
        if trace:
            g = w.T @ w
            dg = np.sqrt(np.abs(np.diag(g)))
            off = g - np.diag(np.diag(g))
            live = np.diag(g) > null2
            rel = (np.abs(off) / np.maximum(np.outer(dg, dg), 1e-300))[np.ix_(live, live)].max()
            print(f"    sweep {sweep + 1:2d}: block pairs rotated {rotated:5d}   off-norm / norm {np.linalg.norm(off) / np.linalg.norm(g):.1e}   "
                  f"max relative off-diagonal {rel:.1e}", flush=True)
        if rotated == 0:
            return sweep + 1
    return max_sweeps


def quality(s, lam, vecs):
    want = np.linalg.eigvalsh(s)
    return (np.linalg.norm((vecs * lam) @ vecs.T - s) / np.linalg.norm(s), np.abs(vecs.T @ vecs - np.eye(len(lam))).max(),
            np.abs(np.sort(lam) - want).max() / np.abs(want).max())


def solve_on_s(s, sort=False, passes=1, max_sweeps=60, trace=False):
    """Today's method: W = S V kept explicitly (kf_eigh.hip); `sort`: columns ordered by decreasing diagonal first."""
    d = s.shape[0]
    perm = np.argsort(-np.diag(s)) if sort else np.arange(d)
    w, v = s[:, perm].copy(), np.eye(d)[:, perm].copy()
    sweeps = block_jacobi(w, v, 4.0 * EPS * np.sqrt(d), (s * s).sum() * EPS * EPS * d, passes, max_sweeps, trace)
    return (sweeps,) + quality(s, (v * w).sum(0), v)


def solve_on_cholesky_factor(s, passes=1, max_sweeps=60, trace=False):
    """Veselic-Hari: Cholesky of the diagonally sorted, slightly shifted matrix, block Jacobi on the columns of L, no V."""
    d = s.shape[0]
    perm = np.argsort(-np.diag(s))
    shift = 4.0 * np.sqrt(d) * EPS * np.sqrt((s * s).sum())   # > the rounding of the factorisation: S may be singular
    w = np.linalg.cholesky(s[np.ix_(perm, perm)] + shift * np.eye(d))
    sweeps = block_jacobi(w, None, 4.0 * EPS * np.sqrt(d), 0.0, passes, max_sweeps, trace)
    sigma2 = (w * w).sum(0)
    vecs = np.empty_like(w)
    vecs[perm, :] = w / np.sqrt(sigma2)
    return (sweeps,) + quality(s, sigma2 - shift, vecs)


def spectra(d, rng):
    """Test matrices: (name, S)."""
    def gram(n, lo, dependent=0):
        x = rng.standard_normal((n, d)) * np.logspace(0, lo, d)
        if dependent:
            x[:, -dependent:] = x[:, :dependent] @ rng.standard_normal((dependent, dependent))
        x = x[:, rng.permutation(d)]
        return x.T @ x / n
    q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    yield "graded columns, eigenvalues 1 .. 1e-6 (tools/eigh_bench.py), shuffled", gram(2 * d, -3.0)
    yield "graded columns, eigenvalues 1 .. 1e-12, shuffled", gram(2 * d, -6.0)
    yield "the same with 3 exactly dependent columns", gram(2 * d, -3.0, dependent=3)
    yield "power law k^-2, dense eigenvectors", (q * (1.0 / np.arange(1, d + 1) ** 2.0)) @ q.T
    yield "16 spikes over a flat bulk (5 % spread), dense eigenvectors", \
        (q * np.concatenate([np.logspace(2, 0, 16), 1.0 + 0.05 * rng.standard_normal(d - 16)])) @ q.T
    yield "half the eigenvalues exactly zero (n = d / 2 samples)", gram(d // 2, -3.0)


VARIANTS = {
    "base": lambda s, trace=False: solve_on_s(s, trace=trace),
    "sort": lambda s, trace=False: solve_on_s(s, sort=True, trace=trace),
    "inner2": lambda s, trace=False: solve_on_s(s, sort=True, passes=2, trace=trace),
    "chol": lambda s, trace=False: solve_on_cholesky_factor(s, trace=trace),
    "chol_inner2": lambda s, trace=False: solve_on_cholesky_factor(s, passes=2, trace=trace),
}


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "table"
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    assert d % (2 * KB) == 0, "the prototype wants whole block pairs"
    rng = np.random.default_rng(1)
    if mode == "one":
        name, s = next(spectra(d, rng))
        s = 0.5 * (s + s.T)
        print(f"d = {d}, {name}, variant {sys.argv[3]}")
        print("  sweeps %d  reconstruction %.1e  orthogonality %.1e  eigenvalues %.1e" % VARIANTS[sys.argv[3]](s, trace=True))
        return
    print(f"d = {d}; sweeps (reconstruction error, orthogonality error) per variant")
    for name, s in spectra(d, rng):
        s = 0.5 * (s + s.T)
        print(f"{name}:")
        for variant in ("base", "sort", "inner2", "chol", "chol_inner2"):
            t0 = time.time()
            sweeps, recon, ortho, evals = VARIANTS[variant](s)
            print(f"    {variant:12s} {sweeps:3d} sweeps   ({recon:.1e}, {ortho:.1e}; eigenvalues {evals:.1e})   [{time.time() - t0:.0f} s]", flush=True)


if __name__ == "__main__":
    main()
