"""GPU parity of every C-ABI entry point against the CPU oracle on identical seeded inputs.

Tolerances (stated per the north star): fp32 factors/scores ``rel_F <= 2e-5`` against the oracle run
in fp64 on the same (fp32-representable) inputs; bf16/fp16 inputs are converted exactly to fp32
by the kernels, so the same bound applies against the oracle fed the up-cast inputs.
"""

import math
import os

import pytest
import torch
from torch import nn

from oracle import ekfac_ref as ref

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
TOL = 2e-5


def rel(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp(min=1e-300))


@pytest.fixture(scope="module")
def ops():
    from kronfluence_amd import ops as _ops

    return _ops


def _rand(*shape, dtype=torch.float32, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g).to(dtype)


# ---- stage 1 -------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,d,bias", [(1, 1, False), (37, 5, True), (300, 129, True), (1000, 785, True), (513, 256, False)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_linear_activation_cov(ops, n, d, bias, dtype):
    x = _rand(n, d, dtype=dtype)
    want = torch.zeros(d + bias, d + bias, dtype=torch.float64)
    flat, count = ref.linear_flat_activation(x.double(), None, bias)
    ref.covariance_update(want, flat)
    cov = torch.zeros(d + bias, d + bias, device=DEV)
    cnt = torch.zeros(1, dtype=torch.int64, device=DEV)
    for _ in range(2):  # accumulate twice: checks "+="
        ops.linear_activation_cov(cov, cnt, x.to(DEV), None, bias)
    assert rel(cov, 2 * want) <= TOL
    assert int(cnt) == 2 * count
    assert rel(cov, cov.t()) <= 1e-6  # split-K atomics: symmetric up to fp32 summation order


@pytest.mark.parametrize("mask_dtype", [torch.int64, torch.float32, torch.bool])
def test_linear_activation_cov_masked_sequence(ops, mask_dtype):
    b, t, d = 7, 19, 33
    x = _rand(b, t, d)
    lengths = torch.randint(1, t + 1, (b,), generator=torch.Generator().manual_seed(3))
    mask = (torch.arange(t)[None] < lengths[:, None]).to(mask_dtype)
    flat, count = ref.linear_flat_activation(x.double(), mask.double(), True)
    want = torch.zeros(d + 1, d + 1, dtype=torch.float64)
    ref.covariance_update(want, flat)
    cov = torch.zeros(d + 1, d + 1, device=DEV)
    cnt = torch.zeros(1, dtype=torch.int64, device=DEV)
    ops.linear_activation_cov(cov, cnt, x.to(DEV), mask.to(DEV), True)
    assert rel(cov, want) <= TOL
    assert int(cnt) == int(count)
    # gradient side: rows un-masked, count masked (linear.py:48-54)
    g = _rand(b, t, 11, seed=5)
    gcov = torch.zeros(11, 11, device=DEV)
    gcnt = torch.zeros(1, dtype=torch.int64, device=DEV)
    ops.linear_gradient_cov(gcov, gcnt, g.to(DEV), mask.to(DEV), alpha=4.0)
    gflat, gcount = ref.linear_flat_gradient(g.double(), mask.double())
    gwant = torch.zeros(11, 11, dtype=torch.float64)
    ref.covariance_update(gwant, gflat, alpha=4.0)
    assert rel(gcov, gwant) <= TOL and int(gcnt) == int(gcount)


CONVS = [
    dict(cin=3, cout=4, k=3, stride=1, padding=1, dilation=1, groups=1, bias=False, hw=(8, 8)),
    dict(cin=4, cout=8, k=5, stride=2, padding=2, dilation=1, groups=1, bias=True, hw=(9, 7)),
    dict(cin=8, cout=6, k=3, stride=1, padding=1, dilation=2, groups=2, bias=True, hw=(8, 8)),
    dict(cin=6, cout=6, k=(3, 2), stride=(2, 1), padding=(0, 1), dilation=1, groups=1, bias=False, hw=(10, 6)),
    dict(cin=4, cout=4, k=3, stride=1, padding="same", dilation=1, groups=1, bias=True, hw=(6, 6)),
]


def _conv(c):
    return nn.Conv2d(c["cin"], c["cout"], c["k"], stride=c["stride"], padding=c["padding"], dilation=c["dilation"],
                     groups=c["groups"], bias=c["bias"])


@pytest.mark.parametrize("c", CONVS)
def test_conv_covariances(ops, c):
    conv = _conv(c).double()
    b = 5
    x = _rand(b, c["cin"], *c["hw"])
    flat, count = ref.conv_flat_activation(x.double(), conv)
    d = flat.shape[1]
    want = torch.zeros(d, d, dtype=torch.float64)
    ref.covariance_update(want, flat)
    patches = ops.im2col(x.to(DEV), conv, conv.bias is not None)
    assert rel(patches.reshape(-1, d), flat) <= 1e-6
    cov = torch.zeros(d, d, device=DEV)
    cnt = torch.zeros(1, dtype=torch.int64, device=DEV)
    ops.conv_activation_cov(cov, cnt, x.to(DEV), conv)
    assert rel(cov, want) <= TOL and int(cnt) == count

    out = conv(x.double())
    g = _rand(*out.shape, seed=9)
    gflat, gcount = ref.conv_flat_gradient(g.double())
    gwant = torch.zeros(c["cout"], c["cout"], dtype=torch.float64)
    ref.covariance_update(gwant, gflat)
    gcov = torch.zeros(c["cout"], c["cout"], device=DEV)
    gcnt = torch.zeros(1, dtype=torch.int64, device=DEV)
    ops.conv_gradient_cov(gcov, gcnt, g.to(DEV))
    assert rel(gcov, gwant) <= TOL and int(gcnt) == gcount


@pytest.mark.parametrize("c", [
    dict(b=7, cin=3, k=3, stride=1, padding=1, dilation=1, hw=(32, 32), bias=False),           # ResNet-9's first layer: 27 columns
    dict(b=3, cin=3, k=3, stride=1, padding=1, dilation=1, hw=(16, 16), bias=True),            # + the bias column: 28
    dict(b=1, cin=1, k=5, stride=2, padding=0, dilation=1, hw=(13, 13), bias=True),            # 26 columns, 25 positions: an odd count
    dict(b=5, cin=2, k=(2, 3), stride=(1, 2), padding=(1, 0), dilation=(2, 1), hw=(9, 11), bias=False),
    dict(b=4, cin=8, k=2, stride=2, padding=0, dilation=1, hw=(8, 8), bias=False),             # exactly 32 columns
    dict(b=2, cin=31, k=1, stride=1, padding=0, dilation=1, hw=(5, 7), bias=True),             # 1 x 1 kernel, 32 with the bias column
])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_conv_activation_cov_small_patches(ops, c, dtype, monkeypatch):
    """kf_conv2d_cov_small: patch width C k1 k2 (+ 1) <= 32 straight from the NCHW input on one fp32 MFMA per two positions, in the
    reference's (c, ky, kx) order with its zero padding and ones column, against conv2d.py:15-64,106-128 + factor.py:58 in fp64 --
    and against the materialised path (kf_im2col + kf_syrk_accum) it replaces."""
    conv = nn.Conv2d(c["cin"], 4, c["k"], stride=c["stride"], padding=c["padding"], dilation=c["dilation"], bias=c["bias"])
    x = _rand(c["b"], c["cin"], *c["hw"], dtype=dtype)
    flat, count = ref.conv_flat_activation(x.double(), conv.double())
    d = flat.shape[1]
    assert d <= 32
    want = torch.zeros(d, d, dtype=torch.float64)
    ref.covariance_update(want, flat)
    xd = x.to(DEV)
    cov = torch.zeros(d, d, device=DEV)
    cnt = torch.zeros(1, dtype=torch.int64, device=DEV)
    if dtype == torch.bfloat16 and ops.conv2d_cov_geometry(xd, conv) is not None:
        pytest.skip("bf16 layer taken by the LDS-DMA implicit-im2col kernel")
    assert ops.conv2d_cov_small(cov, cnt, xd, conv)
    ops.conv_activation_cov(cov, cnt, xd, conv)                      # the dispatcher takes the same kernel: accumulates
    assert rel(cov, 2 * want) <= TOL, rel(cov, 2 * want)
    assert int(cnt) == 2 * count and rel(cov, cov.t()) <= 1e-6       # both triangles written (fp32 atomics: equal up to their order)
    monkeypatch.setenv("KF_CONV_COV_SMALL", "0")
    old = torch.zeros(d, d, device=DEV)
    ocnt = torch.zeros(1, dtype=torch.int64, device=DEV)
    assert not ops.conv2d_cov_small(old, ocnt, xd, conv)
    ops.conv_activation_cov(old, ocnt, xd, conv)
    assert rel(cov, 2 * old) <= TOL and int(ocnt) == count


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("cin,k", [(31, 1), (7, 3), (5, (3, 1))])
def test_im2col_ones_column_in_the_last_octet(ops, dtype, cin, k):
    """A biased layer whose augmented patch width C k1 k2 + 1 is a multiple of 8, 2-byte patches (the 16-byte-store kernel): the
    ones column is element I' - 1 of the last octet (module/conv2d.py:120-127).  Round 6 found it gathered from channel C -- one
    element past the image -- instead; exact against the reference's patches now."""
    conv = nn.Conv2d(cin, 4, k, padding=1, bias=True)
    x = _rand(3, cin, 6, 9, dtype=dtype)
    flat, _ = ref.conv_flat_activation(x.double(), conv.double())
    assert flat.shape[1] % 8 == 0
    patches = ops.im2col(x.to(DEV), conv, True, dtype)
    assert torch.equal(patches.reshape(-1, flat.shape[1]).double().cpu(), flat)          # a gather: bit exact
    assert bool((patches[..., -1] == 1).all())


def test_conv_activation_cov_small_declines_wide_patches(ops):
    conv = nn.Conv2d(4, 4, 3, padding=1, bias=False)                 # 36 columns
    x = _rand(2, 4, 8, 8).to(DEV)
    cov = torch.zeros(36, 36, device=DEV)
    cnt = torch.zeros(1, dtype=torch.int64, device=DEV)
    assert not ops.conv2d_cov_small(cov, cnt, x, conv) and int(cnt) == 0
    grouped = nn.Conv2d(4, 4, 3, padding=1, groups=2, bias=False)    # groups: the reference averages them (conv2d.py:55-56)
    assert not ops.conv2d_cov_small(torch.zeros(18, 18, device=DEV), cnt, x, grouped)


# ---- GEMM building block -------------------------------------------------------------------------
@pytest.mark.parametrize("n,d,m,ones", [(5, 7, 3, False), (300, 129, 130, True), (1000, 1024, 1025, True)])
def test_matmul_nn_asymmetric(ops, n, d, m, ones):
    x, w = _rand(n, d), _rand(d + ones, m, seed=1)
    xx = torch.cat([x, torch.ones(n, 1)], 1) if ones else x
    got = ops.matmul_nn(x.to(DEV), w.to(DEV), append_ones=ones)
    assert rel(got, xx.double() @ w.double()) <= TOL


@pytest.mark.parametrize("b,r,o,i,bias", [(3, 1, 5, 7, True), (4, 9, 33, 17, True), (2, 200, 130, 64, False)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_per_sample_gradient(ops, b, r, o, i, bias, dtype):
    g, a = _rand(b, r, o, dtype=dtype), _rand(b, r, i, dtype=dtype, seed=1)
    want = ref.linear_per_sample_gradient(a.double(), g.double(), bias)
    got = ops.per_sample_gradient(g.to(DEV), a.to(DEV), bias)
    assert got.shape == want.shape and rel(got, want) <= TOL


@pytest.mark.parametrize("b,t,d,bias", [(3, 64, 64, True), (5, 128, 768, True), (2, 64, 3072, True), (4, 192, 136, False)])
@pytest.mark.parametrize("mask_dtype", [None, torch.int64, torch.bool])
def test_linear_activation_cov_bf16_sequence_rows(ops, b, t, d, bias, mask_dtype):
    """kf_syrk_rows_bf16 (LDS-DMA covariance kernel): bf16 [b, T, d] rows with a 0/1 padding mask and the bias column --
    module/linear.py:30-46 + tracker/factor.py:58 -- against the oracle on the same bf16 values (exact products)."""
    x = _rand(b, t, d, dtype=torch.bfloat16)
    mask = None
    if mask_dtype is not None:
        lengths = torch.randint(1, t + 1, (b,), generator=torch.Generator().manual_seed(3))
        mask = (torch.arange(t)[None] < lengths[:, None]).to(mask_dtype)
    flat, count = ref.linear_flat_activation(x.double(), None if mask is None else mask.double(), bias)
    want = torch.zeros(d + bias, d + bias, dtype=torch.float64)
    ref.covariance_update(want, flat)
    cov = torch.zeros(d + bias, d + bias, device=DEV)
    cnt = torch.zeros(1, dtype=torch.int64, device=DEV)
    for _ in range(2):
        ops.linear_activation_cov(cov, cnt, x.to(DEV), None if mask is None else mask.to(DEV), bias)
    assert rel(cov, 2 * want) <= TOL, rel(cov, 2 * want)
    assert int(cnt) == 2 * int(count) and rel(cov, cov.t()) <= 1e-6


def test_linear_activation_cov_weighted_integer_mask(ops):
    """A non-binary integer mask weights the rows: the reference multiplies activations AND the bias one by the mask values in
    the activation dtype (module/linear.py:39-43) and counts ``mask.sum()``; the bf16 sequence kernel does the same product
    (rounded to bf16 once) instead of treating the mask as a row select."""
    b, t, d = 3, 64, 72
    x = _rand(b, t, d, dtype=torch.bfloat16)
    mask = torch.randint(0, 4, (b, t), generator=torch.Generator().manual_seed(5))
    cov, cnt = torch.zeros(d + 1, d + 1, device=DEV), torch.zeros(1, dtype=torch.int64, device=DEV)
    ops.linear_activation_cov(cov, cnt, x.to(DEV), mask.to(DEV), True)
    rows = (x.float() * mask[..., None].float()).to(torch.bfloat16).double().flatten(0, 1)
    rows = torch.cat([rows, mask.double().reshape(-1, 1)], dim=-1)
    assert rel(cov, rows.t() @ rows) <= TOL, rel(cov, rows.t() @ rows)
    assert int(cnt) == int(mask.sum())


@pytest.mark.parametrize("c", [
    dict(b=7, cin=32, k=3, padding=0, hw=(8, 8), bias=False),     # ResNet-9's unpadded layer in small: a 6 x 6 grid, 252 rows -> 256
    dict(b=3, cin=64, k=3, padding=0, hw=(7, 9), bias=False),     # 5 x 7 grid, 105 rows -> 128
    dict(b=2, cin=47, k=(3, 2), padding=0, hw=(6, 6), bias=True), # I' = 282 + 1 is odd: not eligible, the generic engine takes it
])
def test_conv_activation_cov_patch_rows_on_the_k_major_kernel(ops, c):
    """ops.conv_patch_rows_cov: a bf16 conv layer whose output grid is not whole k-steps (the implicit-im2col covariance declines it)
    accumulates its materialised patch rows as ONE K-major operand, zero rows up to a whole k-tile (kf_syrk_rows_bf16) -- against
    conv2d.py:106-128 + factor.py:58 in fp64; the count is the number of REAL rows."""
    conv = nn.Conv2d(c["cin"], 8, c["k"], padding=c["padding"], bias=c["bias"])
    x = _rand(c["b"], c["cin"], *c["hw"], dtype=torch.bfloat16)
    flat, count = ref.conv_flat_activation(x.double(), conv)
    d = flat.shape[1]
    want = torch.zeros(d, d, dtype=torch.float64)
    ref.covariance_update(want, flat)
    assert ops.conv2d_cov_geometry(x.to(DEV), conv) is None
    cov = torch.zeros(d, d, device=DEV)
    cnt = torch.zeros(1, dtype=torch.int64, device=DEV)
    taken = ops.conv_patch_rows_cov(cov, cnt, x.to(DEV), conv)
    assert taken == (d % 8 == 0 and d >= 256)
    if taken:
        assert rel(cov, want) <= TOL and int(cnt) == count
    ops.conv_activation_cov(cov, cnt, x.to(DEV), conv)   # the public entry takes the same route (or the generic engine)
    assert rel(cov, (2 if taken else 1) * want) <= TOL, rel(cov, want)
    assert int(cnt) == (2 if taken else 1) * count and rel(cov, cov.t()) <= 1e-6


@pytest.mark.parametrize("c", [
    dict(b=5, cin=16, k=3, stride=1, padding=1, dilation=1, hw=(16, 16)),
    dict(b=3, cin=8, k=5, stride=2, padding=2, dilation=1, hw=(16, 16)),
    dict(b=4, cin=128, k=3, stride=1, padding=1, dilation=1, hw=(8, 8)),
    dict(b=6, cin=24, k=(1, 3), stride=1, padding=(0, 1), dilation=1, hw=(8, 8)),
])
def test_conv_activation_cov_implicit_im2col(ops, c):
    """kf_conv2d_cov_accum: patches^T patches straight from the bf16 NCHW input (no patch tensor), in the reference's
    (c, ky, kx) patch order, against conv2d.py:106-128 + factor.py:58 in fp64 -- and equal to the materialised path."""
    conv = nn.Conv2d(c["cin"], 8, c["k"], stride=c["stride"], padding=c["padding"], dilation=c["dilation"], bias=False)
    x = _rand(c["b"], c["cin"], *c["hw"], dtype=torch.bfloat16)
    flat, count = ref.conv_flat_activation(x.double(), conv)
    d = flat.shape[1]
    want = torch.zeros(d, d, dtype=torch.float64)
    ref.covariance_update(want, flat)
    assert ops.conv2d_cov_geometry(x.to(DEV), conv) is not None
    cov = torch.zeros(d, d, device=DEV)
    cnt = torch.zeros(1, dtype=torch.int64, device=DEV)
    for _ in range(2):
        ops.conv_activation_cov(cov, cnt, x.to(DEV), conv)
    assert rel(cov, 2 * want) <= TOL, rel(cov, 2 * want)
    assert int(cnt) == 2 * count and rel(cov, cov.t()) <= 1e-6


# ---- stage 2 -------------------------------------------------------------------------------------
@pytest.mark.parametrize("d,n", [(1, 5), (2, 9), (17, 100), (64, 40), (129, 1000), (255, 300), (256, 1000), (300, 150), (513, 2000),
                                 (770, 300), (1030, 5000), (1601, 2500)])
# d >= 256: blocked rounds on the fp64 matrix cores (pairs of 32-column blocks), incl. d % 32 != 0, an odd number of
# blocks (bye) and rank-deficient covariances
def test_eigh_invariants_and_values(ops, d, n):
    x = _rand(n, d).double()
    cov = (x.t() @ x).float()
    evals, evecs, sweeps = ops.eigh(cov.to(DEV), float(n))
    count = torch.tensor([n])
    inv = ref.eigh_invariants(cov, count, evals.cpu(), evecs.cpu())
    # blocked solver: a column takes ~2000 64 x 64 block rotations over the 15 - 30 sweeps; their round-off accumulates to a
    # few 1e-12 (LAPACK: d * eps ~ 1e-13 .. 1e-12 at these sizes; the factors are stored in fp32, eps 6e-8)
    bound = 2e-12 if d < 256 else 1e-11
    assert inv["orthogonality"] < bound and inv["reconstruction"] < bound and inv["ascending"] == 0.0, (inv, sweeps)
    want, _ = ref.eigendecompose(cov.double(), count)
    scale = want.abs().max()
    assert float((evals.cpu() - want).abs().max() / scale) < 1e-10


@pytest.mark.parametrize("d,n", [(256, 600), (512, 200), (770, 3000), (1030, 5000), (1601, 2500)])
def test_eigh_factor_first(ops, d, n, monkeypatch):
    """The factor-first solver (default for d >= 256): Cholesky of the diagonally sorted, shifted covariance, then the blocked
    Jacobi on the factor without V (kf_eigh.hip, chol_* kernels; the algorithm of tools/eigh_jacobi_proto.py).  Same invariants
    and eigenvalue bound as the solver that carries V (KF_EIGH_CHOLESKY=0), no more sweeps; a rank-deficient fp32 covariance
    (n < d: the factorisation meets a non-positive pivot and falls back) and ragged sizes (d % 64 != 0) included."""
    monkeypatch.delenv("KF_EIGH_CHOLESKY", raising=False)
    x = _rand(n, d).double() * torch.logspace(0, -3, d, dtype=torch.float64)[torch.randperm(d, generator=torch.Generator().manual_seed(d))]
    cov = (x.t() @ x).float()
    evals, evecs, sweeps = ops.eigh(cov.to(DEV), float(n))
    monkeypatch.setenv("KF_EIGH_CHOLESKY", "0")
    _, _, sweeps_default = ops.eigh(cov.to(DEV), float(n))
    count = torch.tensor([n])
    inv = ref.eigh_invariants(cov, count, evals.cpu(), evecs.cpu())
    assert inv["orthogonality"] < 1e-11 and inv["reconstruction"] < 1e-11 and inv["ascending"] == 0.0, (inv, sweeps)
    want, _ = ref.eigendecompose(cov.double(), count)
    assert float((evals.cpu() - want).abs().max() / want.abs().max()) < 1e-10
    print(f"factor-first d={d} n={n}: {sweeps} sweeps (V-carrying solver {sweeps_default})")
    assert sweeps <= sweeps_default


def test_eigh_fp64_input_and_asymmetric_noise(ops):
    x = _rand(50, 20).double()
    cov = x.t() @ x
    cov[3, 7] += 1e-3  # the reference symmetrises (eigen.py:201-203)
    evals, evecs, _ = ops.eigh(cov.to(DEV), 50.0)
    want, _ = ref.eigendecompose(cov, torch.tensor([50]))
    assert float((evals.cpu() - want).abs().max() / want.abs().max()) < 1e-12


@pytest.mark.parametrize("b,r,o,i", [(6, 1, 16, 13), (5, 7, 33, 130), (3, 40, 129, 65), (300, 1, 200, 257)])
def test_lambda_accum(ops, b, r, o, i):
    g, a = _rand(b, r, o), _rand(b, r, i, seed=1)
    q_g = torch.linalg.qr(_rand(o, o, seed=2).double())[0]
    q_a = torch.linalg.qr(_rand(i, i, seed=3).double())[0]
    psg = ref.linear_per_sample_gradient(a.double(), g.double(), False) * 0.5
    want = torch.zeros(o, i, dtype=torch.float64)
    ref.lambda_update(want, psg, q_a, q_g)
    gt = ops.matmul_nn(g.reshape(b * r, o).to(DEV), q_g.float().to(DEV))
    at = ops.matmul_nn(a.reshape(b * r, i).to(DEV), q_a.float().to(DEV))
    lam = torch.zeros(o, i, device=DEV)
    ops.lambda_accum(lam, gt, at, b, r, scale=0.5)
    assert rel(lam, want) <= 5e-5


# ---- stage 3 -------------------------------------------------------------------------------------
@pytest.mark.parametrize("damping", [1e-8, None, 1e-3])
def test_inv_lambda(ops, damping):
    lam = _rand(37, 53).abs() * 100
    want = ref.ekfac_inverse_lambda(lam, torch.tensor([40]), damping, torch.float32)
    got = ops.inv_lambda(lam.to(DEV), 40.0, damping)
    assert rel(got, want) <= 1e-6


@pytest.mark.parametrize("q,r,o,i,bias", [(3, 1, 16, 12, True), (4, 6, 33, 65, True), (2, 50, 129, 40, False)])
def test_precondition(ops, q, r, o, i, bias):
    ip = i + bias
    g, a = _rand(q, r, o), _rand(q, r, i, seed=1)
    q_g = torch.linalg.qr(_rand(o, o, seed=2).double())[0]
    q_a = torch.linalg.qr(_rand(ip, ip, seed=3).double())[0]
    lam_inv = _rand(o, ip, seed=4).abs().double() + 0.1
    psg = ref.linear_per_sample_gradient(a.double(), g.double(), bias)
    want = ref.ekfac_precondition(psg, q_a, q_g, lam_inv) * 2.0
    got = ops.precondition(g.to(DEV), a.to(DEV), bias, q_g.float().to(DEV), q_a.float().to(DEV), lam_inv.float().to(DEV), scale=2.0)
    assert rel(got, want) <= TOL


@pytest.mark.parametrize("q,b,r,o,i,bias", [(3, 5, 1, 16, 12, True), (100, 250, 1, 130, 257, True), (7, 9, 6, 33, 20, True),
                                             (5, 3, 40, 65, 129, False)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_pairwise_score(ops, q, b, r, o, i, bias, dtype):
    p = _rand(q, o, i + bias, seed=7)
    g, a = _rand(b, r, o, dtype=dtype), _rand(b, r, i, dtype=dtype, seed=1)
    if r == 1:
        want = ref.linear_pairwise_score(p.double(), a[:, 0].double(), g[:, 0].double(), bias)
    else:
        want = ref.linear_pairwise_score(p.double(), a.double(), g.double(), bias)
    scores = torch.zeros(q, b + 4, device=DEV)
    ops.pairwise_score(scores, 2, p.to(DEV), g.to(DEV), a.to(DEV), bias, scale=1.0)
    ops.pairwise_score(scores, 2, p.to(DEV), g.to(DEV), a.to(DEV), bias, scale=0.5)  # "+=" across layers
    assert rel(scores[:, 2:2 + b], 1.5 * want) <= TOL
    assert float(scores[:, :2].abs().max()) == 0.0 and float(scores[:, 2 + b:].abs().max()) == 0.0


def test_cpu_tensors_fail_loudly(ops):
    from kronfluence_amd._native import KfError

    with pytest.raises(KfError):
        ops.eigh(torch.eye(3), 1.0)
    # the round-4 entry points refuse host tensors and malformed operands just as loudly
    g, a = torch.zeros(2, 64, 128, dtype=torch.bfloat16), torch.zeros(2, 64, 64, dtype=torch.bfloat16)
    with pytest.raises(KfError):
        ops.rotate_rows_transposed(g, torch.zeros(128, 128, dtype=torch.bfloat16))
    with pytest.raises(KfError):
        ops.lambda_rows_accum(torch.zeros(128, 64), g.transpose(1, 2).contiguous(), a.transpose(1, 2).contiguous())
    with pytest.raises(KfError):
        ops.lowrank_rows_dot(torch.zeros(4, 2), 0, torch.zeros(128, 32, dtype=torch.bfloat16), torch.zeros(128, 32, dtype=torch.bfloat16), 2, 64, 4, 8)
    with pytest.raises(KfError):   # on the device, but the rows are not whole 64-deep k-tiles
        ops.lambda_rows_accum(torch.zeros(128, 64, device=DEV), torch.zeros(2, 128, 40, dtype=torch.bfloat16, device=DEV),
                              torch.zeros(2, 64, 40, dtype=torch.bfloat16, device=DEV))
    with pytest.raises(KfError):   # contraction width of the rotation not a multiple of 64
        ops.rotate_rows_transposed(torch.zeros(2, 64, 72, dtype=torch.bfloat16, device=DEV), torch.zeros(72, 72, dtype=torch.bfloat16, device=DEV))


# ---- bf16 MFMA engine ------------------------------------------------------------------------------
@pytest.mark.parametrize("m,n,k", [(100, 250, 1728), (128, 128, 64), (257, 129, 20000), (1000, 999, 2304 * 16)])
def test_gemm_nt_bf16_engine(ops, m, n, k):
    a, b = _rand(m, k, dtype=torch.bfloat16), _rand(n, k, dtype=torch.bfloat16, seed=1)
    want = a.double() @ b.double().t()
    c = torch.full((m, n), 3.0, device=DEV)
    ad, bd = a.to(DEV), b.to(DEV)
    ops.gemm(c, n, 0, ops.view(ad, 0, k, 1, m, k), ops.view(bd, 0, k, 1, n, k), alpha=0.5, beta=1.0)
    assert rel(c, 0.5 * want + 3.0) <= 2e-5  # exact bf16 products, fp32 accumulation (split-K atomics)


@pytest.mark.parametrize("q,b,r,o,i,bias", [(100, 250, 1, 130, 257, True), (7, 9, 6, 32, 20, False), (40, 33, 50, 64, 127, True),
                                             (5, 3, 40, 65, 129, False)])
def test_pairwise_score_bf16_queries(ops, q, b, r, o, i, bias):
    """bf16 preconditioned gradients (reference score_dtype=bf16): R == 1 reads P in bf16; R > 1 stores
    the per-sample gradient in bf16 (one extra rounding, 2^-9 relative) and uses the bf16 MFMA engine."""
    p = _rand(q, o, i + bias, seed=7).to(torch.bfloat16)
    g, a = _rand(b, r, o, dtype=torch.bfloat16), _rand(b, r, i, dtype=torch.bfloat16, seed=1)
    if r == 1:
        want = ref.linear_pairwise_score(p.double(), a[:, 0].double(), g[:, 0].double(), bias)
    else:
        want = ref.linear_pairwise_score(p.double(), a.double(), g.double(), bias)
    scores = torch.zeros(q, b, device=DEV)
    ops.pairwise_score(scores, 0, p.to(DEV), g.to(DEV), a.to(DEV), bias)
    assert rel(scores, want) <= (2e-5 if r == 1 else 4e-3)


def test_precondition_bf16_output(ops):
    q, r, o, i = 4, 6, 33, 64
    g, a = _rand(q, r, o), _rand(q, r, i, seed=1)
    q_g = torch.linalg.qr(_rand(o, o, seed=2).double())[0].contiguous()
    q_a = torch.linalg.qr(_rand(i + 1, i + 1, seed=3).double())[0].contiguous()
    lam_inv = _rand(o, i + 1, seed=4).abs().double() + 0.1
    want = ref.ekfac_precondition(ref.linear_per_sample_gradient(a.double(), g.double(), True), q_a, q_g, lam_inv)
    got = ops.precondition(g.to(DEV), a.to(DEV), True, q_g.float().to(DEV), q_a.float().to(DEV), lam_inv.float().to(DEV),
                           out_dtype=torch.bfloat16)
    assert got.dtype == torch.bfloat16 and rel(got, want) <= 4e-3


@pytest.mark.parametrize("b,r,o,i", [(3, 64, 64, 128), (5, 36, 128, 2304), (2, 1024, 64, 1600), (4, 7, 256, 1152), (3, 100, 8, 16)])
def test_per_sample_gradient_bf16_tn_engine(ops, b, r, o, i):
    """TN layout (k = position, strided): register-transposed staging of the bf16 engine; fp32 output."""
    g, a = _rand(b, r, o, dtype=torch.bfloat16), _rand(b, r, i, dtype=torch.bfloat16, seed=1)
    want = ref.linear_per_sample_gradient(a.double(), g.double(), False)
    got = ops.per_sample_gradient(g.to(DEV), a.to(DEV), False)
    assert got.shape == want.shape and rel(got, want) <= TOL


@pytest.mark.parametrize("q,b,r,o,i", [(40, 33, 50, 64, 128), (100, 130, 36, 128, 2304), (9, 5, 7, 8, 27 * 8)])
def test_pairwise_score_k_tile_major_layout(ops, q, b, r, o, i):
    """bf16 P handed over k-tile-major ([D/64][Q][64]); per-sample gradients are written tiled too."""
    p = _rand(q, o, i, seed=7).to(torch.bfloat16)
    g, a = _rand(b, r, o, dtype=torch.bfloat16), _rand(b, r, i, dtype=torch.bfloat16, seed=1)
    want = ref.linear_pairwise_score(p.double(), a.double(), g.double(), False)
    from kronfluence_amd.module.tracker.pairwise_score import TiledQueries

    pd = p.to(DEV)
    tiled = TiledQueries(pd, 0)
    assert tiled.tiled.shape == (o * i // 64, q, 64) and torch.equal(tiled.tiled, ops.k_tile_major(pd))
    assert torch.equal(tiled.dense(), pd)
    scores = torch.zeros(q, b, device=DEV)
    ops.pairwise_score(scores, 0, tiled, g.to(DEV), a.to(DEV), False)
    plain = torch.zeros(q, b, device=DEV)
    ops.pairwise_score(plain, 0, pd, g.to(DEV), a.to(DEV), False)
    assert rel(scores, want) <= 4e-3 and rel(scores, plain) <= 2e-5


# ---- second-generation score path (csrc/kf_score_v2.hip): LDS-DMA kernels, implicit im2col ----------------------
V2_CONVS = [
    dict(b=5, cin=16, cout=32, k=3, stride=1, padding=1, dilation=1, hw=(16, 16)),     # 3x3 "same": O2 = 16, P = 256
    dict(b=3, cin=8, cout=16, k=5, stride=2, padding=2, dilation=1, hw=(16, 16)),      # strided 5x5: two column phases
    dict(b=4, cin=8, cout=24, k=3, stride=1, padding=2, dilation=2, hw=(8, 16)),       # dilation 2, non-square image
    dict(b=2, cin=128, cout=130, k=3, stride=1, padding=0, dilation=1, hw=(10, 18)),   # no padding, ragged O tile, 2 k-steps
    dict(b=7, cin=8, cout=8, k=(1, 3), stride=1, padding=(0, 1), dilation=1, hw=(8, 8)),  # 1x3 kernel, P = 64
    dict(b=6, cin=3, cout=64, k=3, stride=1, padding=1, dilation=1, hw=(16, 16)),      # 3 channels -> padded to 8 (a first layer)
    dict(b=5, cin=16, cout=24, k=3, stride=1, padding=0, dilation=1, hw=(8, 8)),       # 6 x 6 output grid -> padded to 8 x 8
]


@pytest.mark.parametrize("c", V2_CONVS)
@pytest.mark.parametrize("q", [3, 300])
def test_pairwise_score_conv2d_implicit_im2col(ops, c, q):
    """kf_pairwise_score_conv2d (no patch tensor, NCHW gradient consumed in place) against the oracle's
    ``"qio,bti,bto->qb"`` (module/conv2d.py:199-209) on the same bf16 inputs; P is held in the reference's (c, ky, kx)
    patch order and re-ordered by ``TiledQueries``.  Bound: the per-sample gradient is rounded to bf16 once (2^-9)."""
    from kronfluence_amd.module.tracker.pairwise_score import TiledQueries

    conv = nn.Conv2d(c["cin"], c["cout"], c["k"], stride=c["stride"], padding=c["padding"], dilation=c["dilation"], bias=False)
    b = c["b"]
    x = _rand(b, c["cin"], *c["hw"], dtype=torch.bfloat16)
    out = conv(x.float())
    g = _rand(*out.shape, dtype=torch.bfloat16, seed=1)
    ip = c["cin"] * conv.kernel_size[0] * conv.kernel_size[1]
    p = _rand(q, c["cout"], ip, seed=7).to(torch.bfloat16)
    assert ops.conv2d_score_geometry(x.shape, c["cout"], conv) is not None
    want = ref.conv_pairwise_score(p.double(), x.double(), g.double(), conv.double())
    tiled = TiledQueries(p.to(DEV), 0, conv_channels=c["cin"])
    assert torch.equal(tiled.dense(), p.to(DEV))  # the permutation round-trips
    scores = torch.zeros(q, b + 3, device=DEV)
    ops.pairwise_score_conv2d(scores, 1, tiled, g.to(DEV), x.to(DEV), conv, scale=1.0)
    ops.pairwise_score_conv2d(scores, 1, tiled, g.to(DEV), x.to(DEV), conv, scale=0.5)  # "+=" across layers
    assert rel(scores[:, 1:1 + b], 1.5 * want) <= 4e-3, rel(scores[:, 1:1 + b], 1.5 * want)
    assert float(scores[:, :1].abs().max()) == 0.0 and float(scores[:, 1 + b:].abs().max()) == 0.0
    # and identical (up to fp32 summation order) to the materialised-patch v1 path on the same data
    v1 = torch.zeros(q, b, device=DEV)
    patches = ops.im2col(x.to(DEV), conv, False, torch.bfloat16)
    ops.pairwise_score(v1, 0, p.to(DEV), g.to(DEV).flatten(2).transpose(1, 2).contiguous(), patches, False)
    assert rel(scores[:, 1:1 + b], 1.5 * v1) <= 2e-3


DENSE_LAMBDA_CONVS = [  # b >= 256 (rows (o, sample) of a 256-row tile span at most two o); Cp * k1 * k2 % 64 == 0
    dict(b=256, cin=64, cout=16, k=3, stride=1, padding=1, dilation=1, hw=(8, 8)),      # I' = 576 (9 k-steps), exact groups
    dict(b=300, cin=64, cout=24, k=1, stride=1, padding=0, dilation=1, hw=(8, 8)),      # 1x1: I' = 64, ragged groups of 300
    dict(b=333, cin=8, cout=40, k=(2, 4), stride=1, padding=0, dilation=1, hw=(9, 11)),  # I' = 64, O2 = 8, ragged O
    dict(b=260, cin=32, cout=8, k=(1, 2), stride=(1, 2), padding=0, dilation=1, hw=(8, 16)),  # strided columns: two phases
    dict(b=257, cin=20, cout=16, k=(2, 4), stride=1, padding=(1, 2), dilation=1, hw=(7, 15)),  # 20 channels -> 24: Ipp = 192
    # round 6: very few input channels whose rounding to 8 does not give whole k-steps -- the channel axis is padded wider
    dict(b=256, cin=3, cout=64, k=3, stride=1, padding=1, dilation=1, hw=(16, 16), wide=64),   # ResNet-9's first layer: 3 -> 64, I' = 27
    dict(b=300, cin=1, cout=16, k=3, stride=1, padding=1, dilation=1, hw=(8, 8), wide=64),     # 1 channel -> 64
    dict(b=256, cin=3, cout=8, k=(2, 3), stride=1, padding=0, dilation=1, hw=(9, 10), wide=32),  # 8 * 6 = 48 -> 32 * 6 = 192
]


@pytest.mark.parametrize("c", DENSE_LAMBDA_CONVS)
def test_lambda_conv2d_dense_form(ops, c):
    """kf_lambda_conv2d_accum (gradient rotated along channels -> implicit-im2col per-sample gradient -> tall GEMM with the
    sum-of-squares epilogue) against the reference's ``Lambda += sum_b (Qg^T g_b Qa)^2`` (module/tracker/factor.py:218-226 on
    module/conv2d.py:164-177 gradients) in fp64 on the same bf16 inputs and bf16-rounded eigenvectors.  Bound 2e-2: the
    rotated gradient factor and the per-sample gradient are each rounded to bf16 once."""
    conv = nn.Conv2d(c["cin"], c["cout"], c["k"], stride=c["stride"], padding=c["padding"], dilation=c["dilation"], bias=False)
    b, o = c["b"], c["cout"]
    x = _rand(b, c["cin"], *c["hw"], dtype=torch.bfloat16)
    out = conv(x.float())
    g = _rand(*out.shape, dtype=torch.bfloat16, seed=1)
    k1, k2 = conv.kernel_size
    ip = c["cin"] * k1 * k2
    q_a = torch.linalg.qr(_rand(ip, ip, seed=3).double())[0].float()
    q_g = torch.linalg.qr(_rand(o, o, seed=4).double())[0].float()
    geometry = ops.lambda_conv2d_geometry(tuple(x.shape), o, conv)
    assert geometry is not None
    padded = ops.lambda_conv2d_channels(geometry)
    assert padded == c.get("wide", c["cin"] + (-c["cin"]) % 8)
    qa_t_perm = ops.conv_patch_order_eigenvectors(q_a.to(DEV), c["cin"], k1 * k2, padded)
    assert qa_t_perm.shape[1] % 64 == 0
    lam = torch.zeros(o, ip, device=DEV)
    for scale in (1.0, 0.5):  # "+=" and gradient_scale
        gt = ops.rotate_channels(g.to(DEV), q_g.to(DEV).to(torch.bfloat16))
        ops.lambda_conv2d_accum(lam, gt, x.to(DEV), geometry, qa_t_perm, scale=scale)
    want = torch.zeros(o, ip, dtype=torch.float64)
    psg = ref.conv_per_sample_gradient(x.double(), g.double(), conv.double())
    ref.lambda_update(want, psg, q_a.to(torch.bfloat16).double(), q_g.to(torch.bfloat16).double())
    assert rel(lam, 1.25 * want) <= 2e-2, rel(lam, 1.25 * want)
    # the channel rotation on its own: out[n, o', p] = sum_o q[o, o'] g[n, o, p]
    gt = ops.rotate_channels(g.to(DEV), q_g.to(DEV).to(torch.bfloat16))
    want_gt = torch.einsum("ok,nohw->nkhw", q_g.to(torch.bfloat16).double(), g.double())
    assert rel(gt, want_gt) <= 4e-3


@pytest.mark.parametrize("q,b,r,o,i,bias", [(5, 6, 64, 64, 128, True), (300, 9, 128, 136, 72, True), (40, 3, 512, 72, 768, False),
                                             (7, 4, 64, 8, 8, True),
                                             # long contraction + O % 256 == 0: per-sample gradients on the 256 x 256 wave-role-split
                                             # loop -- whole tiles only / a masked 144-column tile / a 16-column remainder (bias +
                                             # padding) left to the 128 x 128 kernel; 4, 5 and 8 k-tiles
                                             (20, 3, 256, 256, 512, False), (12, 2, 320, 512, 392, True), (9, 3, 512, 256, 264, True)])
def test_pairwise_score_rows_v2(ops, q, b, r, o, i, bias):
    """kf_pairwise_score_rows: Linear layer on [b, R, .] rows; the bias column of ones and the zero padding of I' to a
    multiple of 8 are generated inside the call (no torch.cat); against ``"qio,b...i,b...o->qb"`` (linear.py:112-122)."""
    from kronfluence_amd.module.tracker.pairwise_score import TiledQueries

    width = i + int(bias)
    pad = (-width) % 8
    p = _rand(q, o, width, seed=7).to(torch.bfloat16)
    g, a = _rand(b, r, o, dtype=torch.bfloat16), _rand(b, r, i, dtype=torch.bfloat16, seed=1)
    want = ref.linear_pairwise_score(p.double(), a.double(), g.double(), bias)
    tiled = TiledQueries(p.to(DEV), pad)
    assert tiled.shape == (q, o, width + pad) and torch.equal(tiled.dense(), p.to(DEV))
    scores = torch.zeros(q, b, device=DEV)
    ops.pairwise_score_rows(scores, 0, tiled, g.to(DEV), a.to(DEV), bias)
    assert rel(scores, want) <= 4e-3, rel(scores, want)


@pytest.mark.parametrize("q,b0,b1,r,o,i,bias", [(300, 5, 4, 64, 64, 128, True), (40, 128, 128, 64, 128, 72, True), (9, 3, 1, 512, 256, 264, False)])
def test_pairwise_score_rows_two_segments(ops, q, b0, b1, r, o, i, bias):
    """kf_pairwise_score_rows2: two train micro-batches through ONE call land in adjacent score columns and equal two separate
    calls (same bf16 per-sample gradients; the split-K atomics order differs)."""
    from kronfluence_amd.module.tracker.pairwise_score import TiledQueries

    width = i + int(bias)
    pad = (-width) % 8
    tiled = TiledQueries(_rand(q, o, width, seed=7).to(torch.bfloat16).to(DEV), pad)
    g0, a0 = _rand(b0, r, o, dtype=torch.bfloat16).to(DEV), _rand(b0, r, i, dtype=torch.bfloat16, seed=1).to(DEV)
    g1, a1 = _rand(b1, r, o, dtype=torch.bfloat16, seed=2).to(DEV), _rand(b1, r, i, dtype=torch.bfloat16, seed=3).to(DEV)
    apart, together = torch.zeros(q, b0 + b1 + 3, device=DEV), torch.zeros(q, b0 + b1 + 3, device=DEV)
    ops.pairwise_score_rows(apart, 1, tiled, g0, a0, bias, scale=0.5)
    ops.pairwise_score_rows(apart, 1 + b0, tiled, g1, a1, bias, scale=0.5)
    ops.pairwise_score_rows(together, 1, tiled, g0, a0, bias, scale=0.5, second=(g1, a1))
    assert rel(together, apart) <= 1e-5, rel(together, apart)
    assert float(together[:, 0].abs().max()) == 0.0 and float(together[:, 1 + b0 + b1:].abs().max()) == 0.0


@pytest.mark.parametrize("q,b", [(520, 700), (1000, 1000), (130, 1000), (1000, 100)])
def test_score_gemm_long_k_loops(ops, q, b):
    """The score GEMM at ResNet-9 scale (D = 128 x 1152, ragged tiles, all three tile shapes, split-K chunks of 100+
    k-steps: the steady state of its LDS ring) against torch on the SAME bf16 per-sample gradients."""
    from kronfluence_amd.module.tracker.pairwise_score import TiledQueries

    r, o, i = 16, 128, 1152
    p = _rand(q, o, i, seed=7).to(torch.bfloat16).to(DEV)
    g, a = _rand(b, r, o, dtype=torch.bfloat16).to(DEV), _rand(b, r, i, dtype=torch.bfloat16, seed=1).to(DEV)
    psg = torch.einsum("bro,bri->boi", g.float(), a.float()).to(torch.bfloat16)  # the kernel's own rounding point
    want = p.float().flatten(1) @ psg.float().flatten(1).t()
    for _ in range(3):  # repeated launches: a stale-buffer race would show up as run-to-run differences
        scores = torch.zeros(q, b, device=DEV)
        ops.pairwise_score(scores, 0, TiledQueries(p, 0), g, a, False)
        assert rel(scores, want) <= 1e-5, rel(scores, want)


@pytest.fixture
def pp_issue():
    """Sets ``KF_PP_ISSUE`` (request schedule of the 256 x 256 loop, read per call) for a test and restores it."""
    before = os.environ.get("KF_PP_ISSUE")

    def choose(value):
        if value is None:
            os.environ.pop("KF_PP_ISSUE", None)
        else:
            os.environ["KF_PP_ISSUE"] = str(value)

    yield choose
    choose(before)


@pytest.mark.parametrize("issue", [0, 1, 2])
def test_wave_role_split_loop_race_screen(ops, pp_issue, issue):
    """The round-3 main loop (csrc/kf_pingpong.h) orders its LDS-DMA requests against fragment reads with counted ``vmcnt`` and raw
    barriers only: a mistake there shows up as RARE wrong tiles.  Screen, for every request schedule (``ISSUE`` 0: requests in the
    L segments, 1 / 2: between the MFMA groups): the 256 x 256 score GEMM, the bf16-output rotation and
    the 256-row covariance kernel launched 60 times each on the same operands while another stream keeps HBM busy; every
    result must match the first launch to fp32 summation-order noise (a stale 64-deep k-tile is >= 1e-4 of the result)."""
    from kronfluence_amd.module.tracker.pairwise_score import TiledQueries

    pp_issue(issue)
    q, b, r, o, i = 1000, 1000, 16, 128, 1152
    p = TiledQueries(_rand(q, o, i, seed=7).to(torch.bfloat16).to(DEV), 0)
    g, a = _rand(b, r, o, dtype=torch.bfloat16).to(DEV), _rand(b, r, i, dtype=torch.bfloat16, seed=1).to(DEV)
    x = _rand(60000, 1152, dtype=torch.bfloat16, seed=3).to(DEV)
    qt = _rand(1152, 1152, seed=4).to(torch.bfloat16).to(DEV)
    conv = nn.Conv2d(64, 32, 5, stride=2, padding=2, bias=False)   # I' = 1600: the 256-row covariance kernel
    xc = _rand(300, 64, 32, 32, dtype=torch.bfloat16, seed=5).to(DEV)
    geometry = ops.conv2d_cov_geometry(xc, conv)
    assert geometry is not None
    noise = torch.empty(1 << 28, dtype=torch.uint8, device=DEV)
    side = torch.cuda.Stream()
    first = {}
    for launch in range(60):
        with torch.cuda.stream(side):  # uneven memory load next to the kernels under test
            if launch % 3 != 2:
                noise[: (launch % 5 + 1) << 25].add_(1)
        scores = torch.zeros(q, b, device=DEV)
        ops.pairwise_score(scores, 0, p, g, a, False)
        rotated = ops.rotate_bf16(x, qt).float()
        cov, cnt = torch.zeros(1600, 1600, device=DEV), torch.zeros(1, dtype=torch.int64, device=DEV)
        ops.conv2d_cov_accum(cov, cnt, xc, conv, geometry)
        for key, value in (("score", scores), ("rotate", rotated), ("cov", cov)):
            if launch == 0:
                first[key] = value.clone()
            else:
                worst = float((value - first[key]).abs().max() / first[key].abs().max())
                assert worst <= (0.0 if key == "rotate" else 1e-5), (key, launch, worst)
    torch.cuda.synchronize()


@pytest.mark.parametrize("issue", [1, 2])
def test_request_schedules_of_the_256_loop_agree(ops, pp_issue, issue):
    """``KF_PP_ISSUE`` only moves LDS-DMA requests inside the main loop of csrc/kf_pingpong.h: every kernel on it must give what the
    round-3 schedule gives -- BIT FOR BIT where no atomics are involved (rotations: bf16 output, one workgroup per tile; per-sample
    gradients of a sequence layer) and to fp32 atomic-order noise elsewhere -- for loops of 1, 2, 3, 5 and many k-tiles (prologue
    and the two tail forms of the loop), ragged last tiles, the covariance kernel's LDS offset table and the dense-form Lambda."""
    from kronfluence_amd.module.tracker.pairwise_score import TiledQueries

    def run():
        out = {}
        for d in (64, 128, 192, 320, 1152):                       # rotations: 1, 2, 3, 5, 18 k-tiles; 547 x 2 tiles, both ragged
            x = _rand(140000, d, dtype=torch.bfloat16, seed=d).to(DEV)
            q_t = (_rand(264, d, seed=d + 1) / d ** 0.5).to(torch.bfloat16).to(DEV)
            out[f"rotate{d}"] = ops.rotate_bf16(x, q_t, _rand(264, seed=2).to(DEV))
        for o, i in ((8, 64), (16, 192), (128, 1152)):              # score GEMM: 8, 48 and 2 304 k-tiles before the split
            p = TiledQueries(_rand(700, o, i, seed=7).to(torch.bfloat16).to(DEV), 0)
            g, a = _rand(600, 4, o, dtype=torch.bfloat16).to(DEV), _rand(600, 4, i, dtype=torch.bfloat16, seed=1).to(DEV)
            scores = torch.zeros(700, 600, device=DEV)
            ops.pairwise_score(scores, 0, p, g, a, False)
            out[f"score{o}x{i}"] = scores
        for t_len in (256, 512):                                     # sequence layer: gradients on the 256 x 256 loop (4 / 8 k-tiles)
            p = TiledQueries(_rand(300, 256, 264, seed=9).to(torch.bfloat16).to(DEV), 0)
            g = _rand(256, t_len, 256, dtype=torch.bfloat16, seed=3).to(DEV)
            a = _rand(256, t_len, 256, dtype=torch.bfloat16, seed=4).to(DEV)   # I' = 257, padded to 264 columns of P
            scores = torch.zeros(300, 256, device=DEV)
            ops.pairwise_score_rows(scores, 0, p, g, a, True)
            out[f"rows{t_len}"] = scores
        conv = nn.Conv2d(64, 32, 5, stride=2, padding=2, bias=False)   # I' = 1600: the 256-row covariance kernel, offset table
        xc = _rand(300, 64, 32, 32, dtype=torch.bfloat16, seed=5).to(DEV)
        cov, cnt = torch.zeros(1600, 1600, device=DEV), torch.zeros(1, dtype=torch.int64, device=DEV)
        ops.conv2d_cov_accum(cov, cnt, xc, conv, ops.conv2d_cov_geometry(xc, conv))
        out["cov"] = cov
        rows = _rand(64, 512, 1536, dtype=torch.bfloat16, seed=6).to(DEV)   # rows of a sequence layer wide enough for the 256-row kernel
        cov_rows, cnt = torch.zeros(1536, 1536, device=DEV), torch.zeros(1, dtype=torch.int64, device=DEV)
        ops.linear_activation_cov(cov_rows, cnt, rows, None, False)
        out["cov_rows"] = cov_rows
        torch.cuda.synchronize()
        return out

    pp_issue(0)
    want = run()
    pp_issue(issue)
    got = run()
    for key, value in got.items():
        if key.startswith("rotate"):
            assert torch.equal(value, want[key]), key
        else:
            scale = float(want[key].abs().max())
            assert scale > 0, key
            assert float((value - want[key]).abs().max()) <= 1e-5 * scale, key


@pytest.mark.parametrize("n,d,m,bias", [(40000, 1152, 1152, False), (33000, 1600, 1608, True), (70001, 256, 2304, False)])
def test_rotate_bf16_tall_products(ops, n, d, m, bias):
    """The rotations ``X Q`` of the Lambda stage at sizes that take the 256 x 256-tile LDS-DMA kernel (>= 512 tiles): ragged last
    row tile, result width not a multiple of 256, bias row in the epilogue, zero padding rows of ``q_t``; against torch on the
    same bf16 inputs (one bf16 rounding of the result: 2^-9)."""
    x = _rand(n, d, dtype=torch.bfloat16).to(DEV)
    q_t = (_rand(m, d, seed=3) / d ** 0.5).to(torch.bfloat16).to(DEV)
    if m > d:
        q_t[d:] = 0  # padding rows -> zero output columns
    row = _rand(m, seed=5).to(DEV) if bias else None
    want = x.float() @ q_t.float().t() + (row if bias else 0.0)
    for _ in range(2):
        got = ops.rotate_bf16(x, q_t, row)
        assert got.shape == (n, m) and got.dtype == torch.bfloat16
        assert rel(got, want) <= 4e-3, rel(got, want)
        assert rel(got[-300:], want[-300:]) <= 4e-3 and rel(got[:, -40:], want[:, -40:]) <= 4e-3  # the ragged edges


def test_contiguous_but_misaligned_views(ops):
    """A contiguous view that starts 2 bytes into an allocation (``data_ptr() % 16 == 2``) is legal input: ops copies it to an
    aligned buffer instead of handing the vector loads / LDS-DMA requests an address they cannot take."""
    from kronfluence_amd.module.tracker.pairwise_score import TiledQueries

    def shifted(t):
        flat = torch.zeros(t.numel() + 1, dtype=t.dtype, device=DEV)
        flat[1:] = t.to(DEV).flatten()
        view = flat[1:].view(t.shape)
        assert view.is_contiguous() and view.data_ptr() % 16 != 0
        return view

    b, t, d, o = 3, 64, 72, 64
    x, g = _rand(b, t, d, dtype=torch.bfloat16), _rand(b, t, o, dtype=torch.bfloat16, seed=1)
    cov, cnt = torch.zeros(d, d, device=DEV), torch.zeros(1, dtype=torch.int64, device=DEV)
    ops.linear_activation_cov(cov, cnt, shifted(x), None, False)
    assert rel(cov, x.double().flatten(0, 1).t() @ x.double().flatten(0, 1)) <= TOL
    p = _rand(5, o, d, seed=7).to(torch.bfloat16)
    scores = torch.zeros(5, b, device=DEV)
    ops.pairwise_score_rows(scores, 0, TiledQueries(p.to(DEV), 0), shifted(g), shifted(x), False)
    assert rel(scores, ref.linear_pairwise_score(p.double(), x.double(), g.double(), False)) <= 4e-3
    gc = _rand(2, 64, 8, 8, dtype=torch.bfloat16, seed=2)
    gflat, gcount = ref.conv_flat_gradient(gc.double())
    want = torch.zeros(64, 64, dtype=torch.float64)
    ref.covariance_update(want, gflat)
    cov2, cnt2 = torch.zeros(64, 64, device=DEV), torch.zeros(1, dtype=torch.int64, device=DEV)
    ops.conv_gradient_cov(cov2, cnt2, shifted(gc))
    assert rel(cov2, want) <= TOL and int(cnt2) == gcount


@pytest.mark.parametrize("n,d", [(300, 128), (5000, 1152), (1030, 264), (64, 16)])
def test_syrk_bf16_symmetric_engine(ops, n, d):
    """bf16 rows, no mask / bias column: upper-triangular tile pairs on the bf16 TN engine."""
    x = _rand(n, d, dtype=torch.bfloat16)
    want = x.double().t() @ x.double()
    cov = torch.zeros(d, d, device=DEV)
    cnt = torch.zeros(1, dtype=torch.int64, device=DEV)
    for _ in range(2):
        ops.linear_activation_cov(cov, cnt, x.to(DEV), None, False)
    assert rel(cov, 2 * want) <= TOL and int(cnt) == 2 * n
    assert rel(cov, cov.t()) <= 1e-6


def test_conv_gradient_cov_bf16(ops):
    g = _rand(6, 64, 5, 7, dtype=torch.bfloat16)
    gflat, gcount = ref.conv_flat_gradient(g.double())
    want = torch.zeros(64, 64, dtype=torch.float64)
    ref.covariance_update(want, gflat)
    cov = torch.zeros(64, 64, device=DEV)
    cnt = torch.zeros(1, dtype=torch.int64, device=DEV)
    ops.conv_gradient_cov(cov, cnt, g.to(DEV))
    assert rel(cov, want) <= TOL and int(cnt) == gcount


@pytest.mark.parametrize("b,o,h,w,alpha", [(6, 64, 8, 8, 1.0), (3, 130, 16, 16, 0.25), (70, 256, 8, 16, 1.0), (2, 5, 8, 8, 4.0)])
def test_conv_gradient_cov_bf16_nchw_planes(ops, b, o, h, w, alpha):
    """kf_syrk_planes_bf16: O1*O2 % 64 == 0 -> the NCHW gradient is consumed in place (no "(b o1 o2) c" copy), staged in the
    kernel's row order and added / mirrored by the finalize pass; twice, to check the "+=" and that the staging matrix is
    re-zeroed; against module/conv2d.py:130-132 + tracker/factor.py:93 in fp64."""
    g = _rand(b, o, h, w, dtype=torch.bfloat16)
    gflat, gcount = ref.conv_flat_gradient(g.double())
    want = torch.zeros(o, o, dtype=torch.float64)
    ref.covariance_update(want, gflat)
    cov = torch.zeros(o, o, device=DEV)
    cnt = torch.zeros(1, dtype=torch.int64, device=DEV)
    ops.conv_gradient_cov(cov, cnt, g.to(DEV), alpha)
    ops.conv_gradient_cov(cov, cnt, g.to(DEV), alpha)
    assert rel(cov, 2 * alpha * want) <= TOL and int(cnt) == 2 * gcount
    assert rel(cov, cov.t()) <= 1e-6


@pytest.mark.parametrize("b,t,d,alpha", [(5, 64, 64, 1.0), (3, 128, 776, 0.5), (2, 512, 136, 1.0)])
def test_linear_gradient_cov_bf16_sequence_rows(ops, b, t, d, alpha):
    """Gradient rows of a sequence layer on kf_syrk_rows_bf16 (never masked, only the counter is: linear.py:48-54)."""
    g = _rand(b, t, d, dtype=torch.bfloat16)
    mask = (torch.rand(b, t, generator=torch.Generator().manual_seed(3)) < 0.7).to(torch.int64)
    want = alpha * g.double().flatten(0, 1).t() @ g.double().flatten(0, 1)
    cov = torch.zeros(d, d, device=DEV)
    cnt = torch.zeros(1, dtype=torch.int64, device=DEV)
    ops.linear_gradient_cov(cov, cnt, g.to(DEV), mask.to(DEV), alpha)
    assert rel(cov, want) <= TOL and int(cnt) == int(mask.sum())
    assert rel(cov, cov.t()) <= 1e-6


@pytest.mark.parametrize("c", [dict(cin=8, cout=16, k=3, stride=1, padding=1, dilation=1, groups=1, bias=False, hw=(9, 7)),
                               dict(cin=16, cout=8, k=5, stride=2, padding=2, dilation=1, groups=1, bias=False, hw=(12, 12)),
                               dict(cin=8, cout=8, k=(3, 3), stride=1, padding=0, dilation=2, groups=1, bias=False, hw=(10, 10))])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_im2col_vectorised_path(ops, c, dtype):
    """I' % 8 == 0 and a 2-byte output dtype take the 16-B-store kernel; must equal unfold exactly."""
    conv = _conv(c)
    x = _rand(3, c["cin"], *c["hw"], dtype=dtype)
    want = ref.conv_patches(x.float(), conv)
    got = ops.im2col(x.to(DEV), conv, False, torch.bfloat16)
    assert got.shape == want.shape
    assert torch.equal(got.cpu().float(), want.to(torch.bfloat16).float())


@pytest.mark.parametrize("b,r,o,i", [(5, 7, 64, 128), (3, 40, 128, 72), (6, 256, 64, 1152)])
def test_lambda_accum_bf16_engine(ops, b, r, o, i):
    """bf16 rotations (Q^T in bf16) + squared TN product on the bf16 engine vs the fp64 oracle fed the
    same bf16-rounded eigenvectors and inputs.  The rotated factors are rounded to bf16 once (2^-9)."""
    g, a = _rand(b, r, o, dtype=torch.bfloat16), _rand(b, r, i, dtype=torch.bfloat16, seed=1)
    q_g = torch.linalg.qr(_rand(o, o, seed=2).double())[0].to(torch.bfloat16)
    q_a = torch.linalg.qr(_rand(i, i, seed=3).double())[0].to(torch.bfloat16)
    psg = ref.linear_per_sample_gradient(a.double(), g.double(), False) * 0.5
    want = torch.zeros(o, i, dtype=torch.float64)
    ref.lambda_update(want, psg, q_a.double(), q_g.double())
    gt = ops.rotate_bf16(g.reshape(b * r, o).to(DEV), q_g.t().contiguous().to(DEV))
    at = ops.rotate_bf16(a.reshape(b * r, i).to(DEV), q_a.t().contiguous().to(DEV))
    assert gt.dtype == torch.bfloat16 and rel(gt, g.reshape(b * r, o).double() @ q_g.double()) <= 4e-3
    lam = torch.zeros(o, i, device=DEV)
    ops.lambda_accum(lam, gt, at, b, r, scale=0.5)
    assert rel(lam, want) <= 1e-2


@pytest.mark.parametrize("q,r,o,i", [(4, 6, 64, 128), (3, 50, 128, 72), (5, 1, 16, 64)])
def test_precondition_bf16_back_rotation(ops, q, r, o, i):
    """bf16 output from fp32 factors: without bf16 inputs and all three bf16 eigenvector copies the arithmetic stays on
    the fp32 engine (only the stored result is bf16)."""
    g, a = _rand(q, r, o), _rand(q, r, i, seed=1)
    q_g = torch.linalg.qr(_rand(o, o, seed=2).double())[0].contiguous()
    q_a = torch.linalg.qr(_rand(i, i, seed=3).double())[0].contiguous()
    lam_inv = _rand(o, i, seed=4).abs().double() + 0.1
    want = ref.ekfac_precondition(ref.linear_per_sample_gradient(a.double(), g.double(), False), q_a, q_g, lam_inv)
    qa_d, qg_d = q_a.float().to(DEV), q_g.float().to(DEV)
    got = ops.precondition(g.to(DEV), a.to(DEV), False, qg_d, qa_d, lam_inv.float().to(DEV), out_dtype=torch.bfloat16,
                           q_a_bf16=qa_d.to(torch.bfloat16).contiguous(), q_g_t_bf16=qg_d.t().contiguous().to(torch.bfloat16))
    assert got.dtype == torch.bfloat16 and rel(got, want) <= 1.5e-2


@pytest.mark.parametrize("q,r,o,i", [(4, 6, 64, 128), (3, 50, 128, 72), (7, 256, 128, 1152)])
def test_precondition_bf16_forward_and_back_rotation(ops, q, r, o, i):
    """bf16 inputs + ``Q_A^T`` in bf16: the forward rotation ``A Q_A`` runs on the bf16 engine too (ABI 7)."""
    g, a = _rand(q, r, o, dtype=torch.bfloat16), _rand(q, r, i, dtype=torch.bfloat16, seed=1)
    q_g = torch.linalg.qr(_rand(o, o, seed=2).double())[0].contiguous()
    q_a = torch.linalg.qr(_rand(i, i, seed=3).double())[0].contiguous()
    lam_inv = _rand(o, i, seed=4).abs().double() + 0.1
    want = ref.ekfac_precondition(ref.linear_per_sample_gradient(a.double(), g.double(), False), q_a, q_g, lam_inv)
    qa_d, qg_d = q_a.float().to(DEV), q_g.float().to(DEV)
    got = ops.precondition(g.to(DEV), a.to(DEV), False, qg_d, qa_d, lam_inv.float().to(DEV), out_dtype=torch.bfloat16,
                           q_a_bf16=qa_d.to(torch.bfloat16).contiguous(), q_g_t_bf16=qg_d.t().contiguous().to(torch.bfloat16),
                           q_a_t_bf16=qa_d.t().contiguous().to(torch.bfloat16))
    assert got.dtype == torch.bfloat16 and rel(got, want) <= 1.5e-2, rel(got, want)


@pytest.mark.parametrize("q,r,o,i", [(3, 64, 64, 128), (2, 128, 768, 768), (2, 128, 96, 3072)])
def test_precondition_bf16_odd_augmented_axis(ops, q, r, o, i):
    """Linear WITH bias on sequences (I' = I + 1 odd; BERT / GPT-2 shapes): the bf16 engine carries the augmented axis at
    W = I' rounded up to 8 -- bias column = the row Q_A[I] added in the epilogue, P comes back [q, O, W] with zero
    padding columns -- against the fp64 oracle's (K7 + K11) on the same bf16 inputs."""
    ip = i + 1
    w = ip + (-ip) % 8
    g, a = _rand(q, r, o, dtype=torch.bfloat16), _rand(q, r, i, dtype=torch.bfloat16, seed=1)
    q_g = torch.linalg.qr(_rand(o, o, seed=2).double())[0].contiguous()
    q_a = torch.linalg.qr(_rand(ip, ip, seed=3).double())[0].contiguous()
    lam_inv = _rand(o, ip, seed=4).abs().double() + 0.1
    want = ref.ekfac_precondition(ref.linear_per_sample_gradient(a.double(), g.double(), True), q_a, q_g, lam_inv) * 0.5
    qa_d, qg_d = q_a.float().to(DEV), q_g.float().to(DEV)
    padded = torch.nn.functional.pad(qa_d, (0, w - ip, 0, w - ip))
    got = ops.precondition(g.to(DEV), a.to(DEV), True, qg_d, qa_d, lam_inv.float().to(DEV), scale=0.5, out_dtype=torch.bfloat16,
                           q_a_bf16=padded.to(torch.bfloat16).contiguous(), q_g_t_bf16=qg_d.t().contiguous().to(torch.bfloat16),
                           q_a_t_bf16=padded.t().contiguous().to(torch.bfloat16))
    assert got.shape == (q, o, w) and got.dtype == torch.bfloat16
    assert float(got[..., ip:].float().abs().max()) == 0.0  # padding columns are exact zeros
    assert rel(got[..., :ip], want) <= 1.5e-2, rel(got[..., :ip], want)


@pytest.mark.parametrize("q,r,o,i,bias", [(3, 64, 64, 128, True), (2, 128, 768, 768, True), (4, 6, 72, 136, False), (2, 128, 128, 256, False),
                                          (3, 50, 128, 72, True)])
def test_precondition_bf16_eigenvectors_only(ops, q, r, o, i, bias):
    """kf_precondition_bf16 (ABI 14): the bf16 preconditioner from bf16 eigenvector matrices ALONE -- ``Q_G`` itself in bf16 for the
    back rotation, the bias row ``Q_A[I]`` in fp32 -- gives what ``kf_precondition`` gives when it is handed the fp32 matrices
    those were rounded from (it casts ``Q_G`` per call and reads row ``I`` of ``Q_A``): bit for bit, on the round-3 call chain
    (whole 64-deep tiles) and on the round-1 one (ragged shapes); and against the fp64 oracle (K7 + K11)."""
    ip = i + int(bias)
    w = ip + (-ip) % 8
    g, a = _rand(q, r, o, dtype=torch.bfloat16), _rand(q, r, i, dtype=torch.bfloat16, seed=1)
    q_g = torch.linalg.qr(_rand(o, o, seed=2).double())[0].to(torch.bfloat16)      # the factors as STORED: bf16
    q_a = torch.linalg.qr(_rand(ip, ip, seed=3).double())[0].to(torch.bfloat16)
    lam_inv = _rand(o, ip, seed=4).abs().double() + 0.1
    want = ref.ekfac_precondition(ref.linear_per_sample_gradient(a.double(), g.double(), bias), q_a.double(), q_g.double(), lam_inv) * 0.5
    qa_d, qg_d = q_a.to(DEV), q_g.to(DEV)
    padded = torch.nn.functional.pad(qa_d, (0, w - ip, 0, w - ip)).contiguous()
    padded_t = torch.nn.functional.pad(qa_d.t(), (0, w - ip, 0, w - ip)).contiguous()
    bias_row = qa_d[-1].float().contiguous() if bias else None
    assert ops.precondition_bf16_eligible(g.to(DEV), a.to(DEV))
    got = ops.precondition_bf16(g.to(DEV), a.to(DEV), bias, qg_d.contiguous(), qg_d.t().contiguous(), padded, padded_t, bias_row,
                                lam_inv.float().to(DEV), scale=0.5)
    same = ops.precondition(g.to(DEV), a.to(DEV), bias, qg_d.float(), qa_d.float().contiguous(), lam_inv.float().to(DEV), scale=0.5,
                            out_dtype=torch.bfloat16, q_a_bf16=padded, q_g_t_bf16=qg_d.t().contiguous(), q_a_t_bf16=padded_t)
    assert got.shape == same.shape == (q, o, w) and got.dtype == torch.bfloat16
    assert torch.equal(got, same)
    assert float(got[..., ip:].float().abs().max()) == 0.0 if w > ip else True
    assert rel(got[..., :ip], want) <= 1.5e-2, rel(got[..., :ip], want)


def test_precondition_bf16_declines_what_the_bf16_engine_does_not_take(ops):
    """One row per sample (R == 1) or narrow factors are not the bf16 call chain's: ``precondition_bf16_eligible`` says so and the
    entry point answers KF_ERR_INVALID_ARGUMENT -- the tracker then converts the eigenvectors to fp32 on first use."""
    from kronfluence_amd import _native

    g, a = _rand(3, 1, 64, dtype=torch.bfloat16).to(DEV), _rand(3, 1, 64, dtype=torch.bfloat16, seed=1).to(DEV)
    assert not ops.precondition_bf16_eligible(g, a)
    eye = torch.eye(64, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(_native.KfError):
        ops.precondition_bf16(g, a, False, eye, eye, eye, eye, None, torch.ones(64, 64, device=DEV))
    out = torch.empty(3, 64, 64, dtype=torch.bfloat16, device=DEV)
    ws = torch.empty(1 << 20, dtype=torch.uint8, device=DEV)
    rc = _native.lib().kf_precondition_bf16(out.data_ptr(), 64, g.data_ptr(), a.data_ptr(), 3, 1, 64, 64, 0, eye.data_ptr(), eye.data_ptr(),
                                            eye.data_ptr(), eye.data_ptr(), 64, None, torch.ones(64, 64, device=DEV).data_ptr(), 1.0,
                                            ws.data_ptr(), ws.numel(), None)
    assert rc != 0


@pytest.mark.parametrize("b,r,o,i,bias", [(4, 64, 64, 128, True), (2, 128, 768, 768, True), (3, 128, 64, 3072, True), (3, 64, 72, 136, False)])
def test_lambda_bf16_odd_augmented_axis(ops, b, r, o, i, bias):
    """bf16 Lambda rotations with the bias row added in the epilogue and the augmented axis zero-padded to a multiple of 8
    (``rotate_bf16(.., bias_row)`` + ``lambda_accum`` with wide rows) against tracker/factor.py:218-226 in fp64."""
    ip = i + int(bias)
    w = ip + (-ip) % 8
    g, a = _rand(b, r, o, dtype=torch.bfloat16), _rand(b, r, i, dtype=torch.bfloat16, seed=1)
    q_g = torch.linalg.qr(_rand(o, o, seed=2).double())[0]
    q_a = torch.linalg.qr(_rand(ip, ip, seed=3).double())[0]
    want = torch.zeros(o, ip, dtype=torch.float64)
    ref.lambda_update(want, ref.linear_per_sample_gradient(a.double(), g.double(), bias) * 0.5, q_a, q_g)
    qa_t = torch.nn.functional.pad(q_a.float().t(), (0, w - ip, 0, w - ip)).to(torch.bfloat16).contiguous().to(DEV)
    at = ops.rotate_bf16(a.reshape(b * r, i).to(DEV), qa_t, q_a[i].float().contiguous().to(DEV) if bias else None)
    assert at.shape == (b * r, w)
    exact = torch.cat([a.double(), a.new_ones(b, r, 1).double()], -1) @ q_a if bias else a.double() @ q_a
    assert rel(at[:, :ip], exact.reshape(b * r, ip)) <= 6e-3
    assert w == ip or float(at[:, ip:].float().abs().max()) == 0.0
    gt = ops.rotate_bf16(g.reshape(b * r, o).to(DEV), q_g.float().t().contiguous().to(torch.bfloat16).to(DEV))
    lam = torch.zeros(o, ip, device=DEV)
    ops.lambda_accum(lam, gt, at, b, r, scale=0.5)
    assert rel(lam, want) <= 2e-2, rel(lam, want)


@pytest.mark.parametrize("b,r,o,i,bias", [(4, 64, 128, 128, True), (5, 128, 768, 768, True), (3, 128, 192, 3072, True), (70, 64, 128, 192, False),
                                          (3, 192, 320, 256, True), (9, 512, 256, 64, True), (1, 64, 128, 64, False)])
def test_lambda_rows_engine(ops, b, r, o, i, bias):
    """Round-4 Lambda of a sequence layer: both rotations written K-contiguous per sample (``rotate_rows_transposed``, bias row in
    the epilogue, augmented axis zero-padded to a multiple of 8) + ``lambda_rows_accum`` (per-sample 256 x 128 tiles squared and
    summed in registers; ragged tiles in both directions, one / several sample ranges, 1-8 k-tiles per sample, k-tile counts that
    are not powers of two) against tracker/factor.py:218-226 in fp64 -- and against the round-2 kernel on the same rotations."""
    ip = i + int(bias)
    w = ip + (-ip) % 8
    g, a = _rand(b, r, o, dtype=torch.bfloat16), _rand(b, r, i, dtype=torch.bfloat16, seed=1)
    q_g = torch.linalg.qr(_rand(o, o, seed=2).double())[0]
    q_a = torch.linalg.qr(_rand(ip, ip, seed=3).double())[0]
    want = torch.zeros(o, ip, dtype=torch.float64)
    ref.lambda_update(want, ref.linear_per_sample_gradient(a.double(), g.double(), bias) * 0.5, q_a, q_g)
    qa_t = torch.nn.functional.pad(q_a.float().t(), (0, w - ip, 0, w - ip)).to(torch.bfloat16).contiguous().to(DEV)
    qg_t = q_g.float().t().contiguous().to(torch.bfloat16).to(DEV)
    bias_row = q_a[i].float().contiguous().to(DEV) if bias else None
    assert ops.lambda_rows_eligible(o, i, r)
    at_t = ops.rotate_rows_transposed(a.to(DEV), qa_t, bias_row)
    gt_t = ops.rotate_rows_transposed(g.to(DEV), qg_t)
    assert at_t.shape == (b, w, r) and gt_t.shape == (b, o, r)
    # the same rotations, row-major, from the round-2 entry point: identical bf16 values, transposed per sample
    at = ops.rotate_bf16(a.reshape(b * r, i).to(DEV), qa_t, bias_row).reshape(b, r, w)
    gt = ops.rotate_bf16(g.reshape(b * r, o).to(DEV), qg_t).reshape(b, r, o)
    assert torch.equal(at_t, at.transpose(1, 2)) and torch.equal(gt_t, gt.transpose(1, 2))
    lam = torch.full((o, ip), 3.0, device=DEV)   # accumulates INTO Lambda
    for _ in range(2):
        ops.lambda_rows_accum(lam, gt_t, at_t, scale=0.5)
    assert rel((lam - 3.0) / 2, want) <= 2e-2, rel((lam - 3.0) / 2, want)
    old = torch.zeros(o, ip, device=DEV)
    ops.lambda_accum(old, gt.contiguous(), at.contiguous(), b, r, scale=0.5)
    assert rel((lam - 3.0) / 2, old) <= 1e-5, rel((lam - 3.0) / 2, old)   # same bf16 factors: fp32 summation order only


@pytest.mark.parametrize("q,b", [(1024, 128), (300, 128), (128, 1000), (100, 520), (640, 100)])
@pytest.mark.parametrize("engine", ["wide", "half", "round2"])
def test_score_gemm_half_tile_shapes(ops, q, b, engine, monkeypatch):
    """Score GEMMs whose train batch or query count is half a 256-row tile (GPT-2's train batches of 128 sequences; few queries
    against many samples) on the three engines that take them: "half" = the default, 256 x 128 / 128 x 256 tiles on the loop for
    64 x 64 wave tiles (csrc/kf_pingpong64.h); "wide" = KF_WIDE_TILE=1, 512 x 128 / 128 x 512 tiles on the two-phase
    wave-role-split loop where they pad no more (csrc/kf_pingpong.h, ppw); "round2" = the lock-step loop.  Against torch on the SAME bf16 per-sample gradients; long split-K chunks, ragged tiles, repeated launches, and
    1 / 2 / 3 / 5 k-tiles per item (prologue and tail paths)."""
    from kronfluence_amd.module.tracker.pairwise_score import TiledQueries

    if engine == "wide":
        monkeypatch.setenv("KF_WIDE_TILE", "1")
    elif engine == "round2":
        monkeypatch.setenv("KF_HALF_TILE_ENGINE", "2")
    r, o, i = 16, 128, 1152
    p = _rand(q, o, i, seed=7).to(torch.bfloat16).to(DEV)
    g, a = _rand(b, r, o, dtype=torch.bfloat16).to(DEV), _rand(b, r, i, dtype=torch.bfloat16, seed=1).to(DEV)
    psg = torch.einsum("bro,bri->boi", g.float(), a.float()).to(torch.bfloat16)
    want = p.float().flatten(1) @ psg.float().flatten(1).t()
    tiled = TiledQueries(p, 0)
    for _ in range(3):
        scores = torch.zeros(q, b, device=DEV)
        ops.pairwise_score(scores, 0, tiled, g, a, False)
        assert rel(scores, want) <= 1e-5, rel(scores, want)
    for short in (8, 16, 24, 40):   # D = 8 * short: 1, 2, 3 and 5 k-tiles, no split-K -- the prologue / tail paths of the loop
        ps = _rand(q, 8, short, seed=9).to(torch.bfloat16).to(DEV)
        gs, as_ = _rand(b, r, 8, dtype=torch.bfloat16, seed=2).to(DEV), _rand(b, r, short, dtype=torch.bfloat16, seed=3).to(DEV)
        psg = torch.einsum("bro,bri->boi", gs.float(), as_.float()).to(torch.bfloat16)
        want_s = ps.float().flatten(1) @ psg.float().flatten(1).t()
        scores = torch.zeros(q, b, device=DEV)
        ops.pairwise_score(scores, 0, TiledQueries(ps, 0), gs, as_, False)
        assert rel(scores, want_s) <= 1e-5, (short, rel(scores, want_s))


def test_wave_role_split_64_loop_race_screen(ops):
    """Race screen of the round-4 loops (csrc/kf_pingpong64.h: three LDS stages, counted ``vmcnt``, raw barriers; the 4 x 2 wave grid
    of csrc/kf_pingpong.h with its re-counted waits): the 512 x 128 and 256 x 128 score GEMMs and the Lambda kernel launched 60 times on the same operands beside a stream that keeps HBM unevenly busy; every
    result must equal the first launch up to fp32 atomic-order noise (a stale 64-deep k-tile is >= 1e-4 of the result)."""
    from kronfluence_amd.module.tracker.pairwise_score import TiledQueries

    q, b, r, o, i = 1024, 128, 16, 128, 1152
    p = TiledQueries(_rand(q, o, i, seed=7).to(torch.bfloat16).to(DEV), 0)
    p640 = TiledQueries(_rand(640, o, i, seed=8).to(torch.bfloat16).to(DEV), 0)   # 640 queries: the 256 x 128 tile (512 would pad more)
    g, a = _rand(b, r, o, dtype=torch.bfloat16).to(DEV), _rand(b, r, i, dtype=torch.bfloat16, seed=1).to(DEV)
    gt_t = _rand(48, 768, 128, dtype=torch.bfloat16, seed=3).to(DEV)
    at_t = _rand(48, 776, 128, dtype=torch.bfloat16, seed=4).to(DEV)
    noise = torch.empty(1 << 28, dtype=torch.uint8, device=DEV)
    side = torch.cuda.Stream()
    first = {}
    for launch in range(60):
        with torch.cuda.stream(side):
            if launch % 3 != 2:
                noise[: (launch % 5 + 1) << 25].add_(1)
        scores = torch.zeros(q, b, device=DEV)
        os.environ["KF_WIDE_TILE"] = "1"
        try:
            ops.pairwise_score(scores, 0, p, g, a, False)        # 512 x 128 tiles (ppw loop; opt-in)
        finally:
            os.environ.pop("KF_WIDE_TILE", None)
        scores640 = torch.zeros(640, b, device=DEV)
        ops.pairwise_score(scores640, 0, p640, g, a, False)      # 256 x 128 tiles (pp64 loop)
        lam = torch.zeros(768, 769, device=DEV)
        ops.lambda_rows_accum(lam, gt_t, at_t)
        for key, value in (("score", scores), ("score640", scores640), ("lambda", lam)):
            if launch == 0:
                first[key] = value.clone()
            else:
                worst = float((value - first[key]).abs().max() / first[key].abs().max())
                assert worst <= 1e-5, (key, launch, worst)
    torch.cuda.synchronize()
    want = torch.einsum("sor,sir->soi", gt_t.float(), at_t.float()[:, :769]).square().sum(0)
    assert rel(first["lambda"], want) <= 1e-5


@pytest.mark.parametrize("b,r,q,k", [(3, 64, 5, 64), (16, 128, 40, 64), (2, 512, 130, 32), (5, 72, 7, 24), (1, 8, 1, 8), (4, 64, 300, 128)])
def test_lowrank_rows_dot(ops, b, r, q, k):
    """kf_lowrank_rows_dot: ``scores[j, n] += scale * sum_{t, c} U[n r + t, j k + c] V[n r + t, j k + c]`` -- the reduction of the
    factored low-rank score ("qik,qko,b...i,b...o->qb", module/linear.py:83-99) -- against torch in fp64 on the same bf16 values;
    K / 8 a power of two (shuffle fold) and not (24: per-lane atomics), ragged column blocks, row splits."""
    u, v = _rand(b * r, q * k, dtype=torch.bfloat16), _rand(b * r, q * k, dtype=torch.bfloat16, seed=1)
    want = 0.5 * (u.double() * v.double()).reshape(b, r, q, k).sum(dim=(1, 3)).t()
    scores = torch.full((q + 2, b + 3), 1.0, device=DEV)
    ops.lowrank_rows_dot(scores, 2, u.to(DEV), v.to(DEV), b, r, q, k, scale=0.5)
    assert rel(scores[:q, 2:2 + b] - 1.0, want) <= 1e-5
    assert float((scores[q:] - 1.0).abs().max()) == 0.0 and float((scores[:, :2] - 1.0).abs().max()) == 0.0


# ---- SURVEY.md 8(f) kernels: row-wise weighted dots, broadcast product, squared-operand GEMM ---------------------
@pytest.mark.parametrize("rows,d", [(1, 1), (5, 37), (48, 16 * 17), (3, 1 << 20), (1000, 1024 * 8), (7, 4096 + 8)])
@pytest.mark.parametrize("xdt,ydt", [(torch.float32, torch.float32), (torch.bfloat16, torch.float32),
                                     (torch.float32, torch.bfloat16), (torch.bfloat16, torch.bfloat16)])
@pytest.mark.parametrize("weighted", [False, True])
def test_rowwise_dot(ops, rows, d, xdt, ydt, weighted):
    x, y = _rand(rows, d, dtype=xdt, seed=1), _rand(rows, d, dtype=ydt, seed=2)
    w = _rand(d, seed=3).abs() if weighted else None
    prod = x.double() * y.double() * (w.double() if weighted else 1.0)
    want = 0.5 * prod.sum(1)
    bound = 0.5 * prod.abs().sum(1)  # fp32 accumulation: error relative to the sum of magnitudes
    out = torch.full((rows,), 3.0, device=DEV)
    ops.rowwise_dot(out, x.to(DEV), y.to(DEV), None if w is None else w.to(DEV), scale=0.5, accumulate=True)
    assert float(((out.double().cpu() - 3.0 - want).abs() / bound.clamp(min=1e-30)).max()) <= 2e-6
    ops.rowwise_dot(out, x.to(DEV), y.to(DEV), None if w is None else w.to(DEV), scale=0.5, accumulate=False)
    assert float(((out.double().cpu() - want).abs() / bound.clamp(min=1e-30)).max()) <= 2e-6


def test_rowwise_dot_unaligned_views(ops):
    """Row slices that start off a 16-byte boundary take the scalar path."""
    base_x, base_y = _rand(4 * 64 + 1, seed=4).to(DEV), _rand(4 * 64 + 1, seed=5).to(DEV)
    x, y = base_x[1:].view(4, 64), base_y[1:].view(4, 64)
    out = torch.zeros(4, device=DEV)
    ops.rowwise_dot(out, x, y, None, accumulate=False)
    assert rel(out, (x.double() * y.double()).sum(1)) <= 1e-6


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_mul_bcast(ops, dtype):
    x, m = _rand(9, 6, 11, dtype=dtype, seed=1), _rand(6, 11, seed=2)
    got = ops.mul_bcast(x.to(DEV), m.to(DEV), scale=0.25)
    assert got.dtype == torch.float32 and rel(got, 0.25 * x.double() * m.double()) <= 1e-6


@pytest.mark.parametrize("b,n", [(1, 5), (13, 16 * 17), (300, 1025 * 3)])
def test_gemm_squared_operand_batch_reduction(ops, b, n):
    """``C[0, n] = beta C + alpha sum_b x[b, n]^2``: the Lambda update on a materialised gradient."""
    x = _rand(b, n, seed=7)
    c0 = _rand(1, n, seed=8)
    c = c0.clone().to(DEV)
    ones = torch.ones(b, device=DEV)
    xd = x.to(DEV)
    ops.gemm(c, n, 0, ops.view(ones, 0, 0, 1, 1, b), ops.view(xd, 0, 1, n, n, b, square=True), alpha=0.5, beta=1.0)
    assert rel(c, c0.double() + 0.5 * (x.double() ** 2).sum(0, keepdim=True)) <= TOL


def test_summed_gradient_gemm(ops):
    """``total += scale * G^T [A, 1]`` over all b*R rows (GradientTracker)."""
    from kronfluence_amd.module.tracked_module import TrackedModule

    b, r, o, i = 5, 7, 12, 9
    g, a = _rand(b, r, o, seed=1), _rand(b, r, i, seed=2)
    total = torch.zeros(1, o, i + 1, device=DEV)
    for _ in range(2):
        TrackedModule.accumulate_summed_gradient(total, g.to(DEV), a.to(DEV), True, 0.5)
    a1 = torch.cat([a, torch.ones(b, r, 1)], dim=-1).double()
    want = torch.einsum("bro,bri->oi", g.double(), a1)
    assert rel(total[0], want) <= TOL


# ---- batched in-LDS eigensolver and the low-rank query factorisation --------------------------
@pytest.mark.parametrize("l", [1, 2, 3, 12, 17, 40, 64, 72, 95, 96])
def test_eigh_small_batched(ops, l):
    batch = 37
    y = _rand(batch, l + 5, l, seed=l)
    if l > 4:
        y[:, :, -2:] = y[:, :, :2] * 0.5  # rank deficient Gram matrices
    g = (y.transpose(1, 2) @ y).contiguous()
    evals, evecs = ops.eigh_small(g.to(DEV))
    evals, evecs = evals.double().cpu(), evecs.double().cpu()
    want = torch.linalg.eigvalsh(g.double()).flip(-1)
    assert float((evals - want).abs().max() / want.abs().max()) <= 2e-6
    assert bool((evals[:, :-1] >= evals[:, 1:] - 1e-6 * want.abs().max()).all())  # descending
    eye = torch.eye(l, dtype=torch.float64)
    assert float((evecs.transpose(1, 2) @ evecs - eye).abs().max()) <= 5e-6
    recon = evecs @ torch.diag_embed(evals) @ evecs.transpose(1, 2)
    assert rel(recon, g) <= 5e-6
    # inv_sqrt: Y V S^-1 is orthonormal on the numerical range
    _, basis = ops.eigh_small(g.to(DEV), inv_sqrt=True, floor_rel=1e-10)
    qmat = y.double() @ basis.double().cpu()
    gram = qmat.transpose(1, 2) @ qmat
    keep = want > 1e-5 * want[:, :1]
    diag = torch.diagonal(gram, dim1=1, dim2=2)
    assert float((diag[keep] - 1.0).abs().max()) <= 1e-3
    off = gram - torch.diag_embed(diag)
    assert float(off.abs().max()) <= 1e-3


@pytest.mark.parametrize("q,o,ip,k", [(3, 16, 13, 4), (2, 40, 120, 8), (5, 256, 300, 32), (2, 1024, 785, 64), (2, 512, 400, 128)])
def test_low_rank_factors_are_near_optimal(ops, q, o, ip, k):
    g = torch.Generator().manual_seed(q * 1000 + k)
    r = min(o, ip)
    u = torch.linalg.qr(torch.randn(q, o, r, generator=g))[0]
    v = torch.linalg.qr(torch.randn(q, ip, r, generator=g))[0]
    sv = torch.logspace(0, -3, r)
    p = ((u * sv) @ v.transpose(1, 2)).float().contiguous()
    left, right = ops.low_rank_factors(p.to(DEV), k)
    assert left.shape == (q, o, k) and right.shape == (q, k, ip)
    approx = ops.low_rank_product(left, right).double().cpu()
    best_err = float(sv[k:].norm() / sv.norm())  # Eckart-Young: error of the exact truncated SVD
    err = float((approx - p.double()).norm() / p.double().norm())
    assert err <= best_err * 1.02 + 1e-5, (err, best_err)


# ---- round 5: sequence layers on the K-major loop (csrc/kf_pingpong_tn.h) --------------------------------------------------------
@pytest.fixture
def tn_env():
    """Sets ``KF_TN`` / ``KF_TN_IMG`` (read per call by the library) for a test and restores them."""
    before = {k: os.environ.get(k) for k in ("KF_TN", "KF_TN_IMG")}

    def choose(tn=None, image=None):
        for key, value in (("KF_TN", tn), ("KF_TN_IMG", image)):
            if value is None:
                os.environ.pop(key, None)
            else:
                os.environ[key] = str(value)

    yield choose
    for key, value in before.items():
        if value is None:
            os.environ.pop(key, None)
        else:
            os.environ[key] = value


@pytest.mark.parametrize("image", [0, 1, 2])
@pytest.mark.parametrize("q,b0,b1,r,o,i,bias", [(20, 3, 0, 256, 256, 512, True), (33, 2, 3, 512, 768, 768, True), (9, 2, 0, 320, 256, 256, False),
                                                (16, 1, 0, 1024, 512, 256, True), (300, 5, 4, 64, 256, 256, True),
                                                # many (sample, tile) items: 2 / 3 / 4 / 8 k-tiles per item, two segments, O != I tilings,
                                                # with and without the bias column (summed from the G fragments of the tiles tn == 0)
                                                (33, 40, 30, 128, 768, 768, True), (17, 70, 0, 192, 768, 768, True), (16, 300, 0, 256, 512, 256, True),
                                                (9, 60, 40, 128, 512, 768, False), (12, 20, 13, 512, 1024, 1024, True), (8, 515, 0, 128, 256, 256, True)])
def test_pairwise_score_rows_k_major(ops, tn_env, image, q, b0, b1, r, o, i, bias, monkeypatch):
    """kf_pairwise_score_rows2 on the K-major loop: the hooked ``[b, T, O]`` / ``[b, T, I]`` tensors are the operands of the
    per-sample-gradient kernel (no transposed copies), the bias column is a column sum of ``G`` taken from the fragments beside the
    MFMAs (module/linear.py:68-77, 112-122).
    Every LDS image against the fp64 oracle and against the K-contiguous path (``KF_TN=0``) on the same inputs: same bf16
    per-sample gradients -> the scores agree to the order of the split-K atomics."""
    from kronfluence_amd.module.tracker.pairwise_score import TiledQueries

    monkeypatch.setenv("KF_TN_MIN_R", "64")   # the 64-deep case too (the default keeps T < 256 on the 128 x 128 kernel)
    width = i + int(bias)
    pad = (-width) % 8
    p = _rand(q, o, width, seed=7).to(torch.bfloat16)
    tiled = TiledQueries(p.to(DEV), pad)
    b = b0 + b1
    g, a = _rand(b, r, o, dtype=torch.bfloat16), _rand(b, r, i, dtype=torch.bfloat16, seed=1)
    want = ref.linear_pairwise_score(p.double(), a.double(), g.double(), bias)
    gd, ad = g.to(DEV), a.to(DEV)

    def run():
        scores = torch.zeros(q, b + 2, device=DEV)
        second = (gd[b0:].contiguous(), ad[b0:].contiguous()) if b1 else None
        ops.pairwise_score_rows(scores, 1, tiled, gd[:b0].contiguous(), ad[:b0].contiguous(), bias, second=second)
        assert float(scores[:, 0].abs().max()) == 0.0 and float(scores[:, -1].abs().max()) == 0.0
        return scores[:, 1:-1]

    tn_env(tn=1, image=image)
    got = run()
    assert rel(got, want) <= 4e-3, rel(got, want)
    tn_env(tn=0)
    old = run()
    assert rel(got, old) <= 2e-5, rel(got, old)


@pytest.mark.parametrize("image", [0, 1, 2])
@pytest.mark.parametrize("b,t,d,bias,alpha", [(5, 128, 768, True, 1.0), (3, 64, 256, False, 0.25), (2, 512, 776, False, 1.0), (4, 256, 1032, True, 1.0),
                                              (1, 64, 3072, True, 1.0), (70, 64, 264, True, 1.0)])
def test_sequence_covariance_k_major(ops, tn_env, image, b, t, d, bias, alpha):
    """kf_syrk_rows_bf16 on the K-major loop (unmasked rows): ``X^T X`` straight from the hooked ``[b, T, d]`` tensor, the bias row /
    column as a column sum (module/linear.py:30-54, tracker/factor.py:58, 93) -- ragged 256-row tiles (776, 1032, 264), one and
    many k-tile ranges; every LDS image against the fp64 oracle and the K-contiguous path."""
    x = _rand(b, t, d, dtype=torch.bfloat16)
    rows = x.double().flatten(0, 1)
    if bias:
        rows = torch.cat([rows, torch.ones(rows.shape[0], 1, dtype=torch.float64)], dim=1)
    want = alpha * rows.t() @ rows
    xd = x.to(DEV)

    def run():
        cov = torch.zeros(d + bias, d + bias, device=DEV)
        cnt = torch.zeros(1, dtype=torch.int64, device=DEV)
        if bias:
            assert alpha == 1.0
            ops.linear_activation_cov(cov, cnt, xd, None, True)
        else:
            ops.linear_gradient_cov(cov, cnt, xd, None, alpha)
        assert int(cnt) == b * t
        return cov

    tn_env(tn=1, image=image)
    got = run()
    assert rel(got, want) <= TOL, rel(got, want)
    assert rel(got, got.t()) <= 1e-6
    tn_env(tn=0)
    assert rel(got, run()) <= 1e-5


@pytest.mark.parametrize("image", [0, 1, 2])
def test_k_major_loop_race_screen(ops, tn_env, image):
    """The K-major loop keeps the counted-``vmcnt`` / raw-barrier ordering of csrc/kf_pingpong.h and reads its fragments with inline
    asm the compiler does not count: 40 launches of the per-sample-gradient + score call and of the covariance call on the same
    operands beside a stream that keeps HBM unevenly busy; every result must match the first to fp32 atomic-order noise."""
    from kronfluence_amd.module.tracker.pairwise_score import TiledQueries

    tn_env(tn=1, image=image)
    q, b, r, o, i = 512, 72, 512, 768, 768
    p = TiledQueries(_rand(q, o, i + 1, seed=7).to(torch.bfloat16).to(DEV), 7)
    g, a = _rand(b, r, o, dtype=torch.bfloat16).to(DEV), _rand(b, r, i, dtype=torch.bfloat16, seed=1).to(DEV)
    noise = torch.empty(1 << 28, dtype=torch.uint8, device=DEV)
    side = torch.cuda.Stream()
    first = {}
    for launch in range(40):
        with torch.cuda.stream(side):
            if launch % 3 != 2:
                noise[: (launch % 5 + 1) << 25].add_(1)
        scores = torch.zeros(q, b, device=DEV)
        ops.pairwise_score_rows(scores, 0, p, g, a, True)
        cov, cnt = torch.zeros(o, o, device=DEV), torch.zeros(1, dtype=torch.int64, device=DEV)
        ops.linear_gradient_cov(cov, cnt, g, None, 1.0)
        for key, value in (("score", scores), ("cov", cov)):
            if launch == 0:
                first[key] = value.clone()
            else:
                worst = float((value - first[key]).abs().max() / first[key].abs().max())
                assert worst <= 1e-5, (key, launch, worst)
    torch.cuda.synchronize()


@pytest.mark.parametrize("q,b", [(872, 512), (600, 256), (385, 300), (1100, 129)])
@pytest.mark.parametrize("mixed", [True, False])
def test_score_gemm_mixed_row_tiling(ops, q, b, mixed, monkeypatch):
    """Round 5 (opt-in, ``KF_SCORE_MIXED=1``): a query count of 256 a + r, 0 < r <= 128, against a wide train side (BERT: 872 queries x
    512 sequences) runs ``a`` row tiles on the 256 x 256 loop and ONE launch of 128 x 256 tiles for the last ``r`` rows (896 instead of
    1 024 padded rows); both launches add into the same score block.  Against torch on the same bf16 per-sample gradients, with the split switched off
    (``KF_SCORE_MIXED=0``) as the control; long split-K chunks and 1 / 2 / 3 / 5 k-tiles per item."""
    from kronfluence_amd.module.tracker.pairwise_score import TiledQueries

    monkeypatch.setenv("KF_SCORE_MIXED", "1" if mixed else "0")   # opt-in: measured no faster than one launch over the padded rows
    r, o, i = 16, 128, 1152
    p = _rand(q, o, i, seed=7).to(torch.bfloat16).to(DEV)
    g, a = _rand(b, r, o, dtype=torch.bfloat16).to(DEV), _rand(b, r, i, dtype=torch.bfloat16, seed=1).to(DEV)
    psg = torch.einsum("bro,bri->boi", g.float(), a.float()).to(torch.bfloat16)
    want = p.float().flatten(1) @ psg.float().flatten(1).t()
    tiled = TiledQueries(p, 0)
    for _ in range(2):
        scores = torch.zeros(q, b + 2, device=DEV)
        ops.pairwise_score(scores, 1, tiled, g, a, False)
        assert rel(scores[:, 1:-1], want) <= 1e-5, rel(scores[:, 1:-1], want)
        assert float(scores[:, 0].abs().max()) == 0.0 and float(scores[:, -1].abs().max()) == 0.0
    for short in (8, 16, 24, 40):
        ps = _rand(q, 8, short, seed=9).to(torch.bfloat16).to(DEV)
        gs, as_ = _rand(b, r, 8, dtype=torch.bfloat16, seed=2).to(DEV), _rand(b, r, short, dtype=torch.bfloat16, seed=3).to(DEV)
        psg = torch.einsum("bro,bri->boi", gs.float(), as_.float()).to(torch.bfloat16)
        want_s = ps.float().flatten(1) @ psg.float().flatten(1).t()
        scores = torch.zeros(q, b, device=DEV)
        ops.pairwise_score(scores, 0, TiledQueries(ps, 0), gs, as_, False)
        assert rel(scores, want_s) <= 1e-5, (short, rel(scores, want_s))


@pytest.mark.parametrize("b,t,d,bias,mask_kind", [(32, 128, 768, True, None), (20, 250, 264, False, "int64"), (3, 700, 1024, True, "float32"),
                                                 (16, 512, 3072, True, "bool"), (9, 256, 256, True, "int64")])
def test_fp32_rows_covariance_three_term_split(ops, b, t, d, bias, mask_kind, monkeypatch):
    """kf_syrk_rows_f32: fp32 rows (LayerNorm outputs under autocast with fp32 factors: BERT) through an EXACT split into three bf16
    terms and six bf16 MFMA products (module/linear.py:30-46 + tracker/factor.py:58).  Against the fp64 oracle at the fp32 factor
    tolerance -- including a constant column whose low significand bits would add up coherently if a term were lost, 0/1 and
    weighted masks (rows AND their bias one times the mask value, in fp32), row counts that are no multiple of 64 -- and against the
    exact-fp32 MFMA engine (``KF_COV_F32_SPLIT=0``)."""
    x = _rand(b, t, d)
    x[..., 0] = 1.2345678
    x[..., 1] = x[..., 1] * 1e-3 + 0.3333333
    mask = None
    if mask_kind is not None:
        gen = torch.Generator().manual_seed(3)
        if mask_kind == "float32":
            mask = torch.rand(b, t, generator=gen) * (torch.rand(b, t, generator=gen) < 0.8)
        else:
            lengths = torch.randint(1, t + 1, (b,), generator=gen)
            mask = (torch.arange(t)[None] < lengths[:, None]).to(getattr(torch, mask_kind))
    flat, count = ref.linear_flat_activation(x.double(), None if mask is None else mask.double(), bias)
    want = torch.zeros(d + bias, d + bias, dtype=torch.float64)
    ref.covariance_update(want, flat)
    xd, md = x.to(DEV), None if mask is None else mask.to(DEV)

    def run():
        cov, cnt = torch.zeros(d + bias, d + bias, device=DEV), torch.zeros(1, dtype=torch.int64, device=DEV)
        if bias:
            ops.linear_activation_cov(cov, cnt, xd, md, True)
        else:
            ops.linear_gradient_cov(cov, cnt, xd, md, 1.0)
        return cov, cnt

    cov, cnt = run()
    if not bias:   # gradient rows are never masked, only counted (linear.py:48-54)
        want = x.double().flatten(0, 1).t() @ x.double().flatten(0, 1)
    assert rel(cov, want) <= 2e-6, rel(cov, want)   # well inside the fp32 factor tolerance (2e-5)
    assert rel(cov, cov.t()) <= 1e-6
    if mask is not None and mask.dtype != torch.float32:
        assert int(cnt) == int(mask.sum())
    monkeypatch.setenv("KF_COV_F32_SPLIT", "0")
    exact, _ = run()
    assert rel(cov, exact) <= 2e-6, rel(cov, exact)
