"""CPU oracle: a plain torch-CPU restatement of kronfluence's EK-FAC hot path.

TEST INFRASTRUCTURE ONLY -- imported by ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``; never by the product package.

Every function restates one reference routine with the same torch op sequence (``addmm_``,
``linalg.eigh`` in fp64, two-``matmul`` eigenbasis rotations, ``einsum`` contractions) and cites
the reference ``file:line`` it follows (paths relative to the reference checkout,
``kronfluence/...``).  Parity is pinned: ``tests/test_oracle_golden.py`` checks this module against
golden tensors captured from the real reference (``tests/golden/make_golden.py``, run in the build
container where the reference is importable).

The eigendecomposition itself lives in a third-party dependency (``torch.linalg.eigh`` ->
LAPACK ``syevd``, torch 2.10.0); its eigenvalues are pinned by the goldens, its eigenvectors only
by the invariants in ``eigh_invariants`` (the reference's own tests assert existence only,
``tests/factors/test_eigendecompositions.py:60-66``).
"""

from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Callable, Dict, Iterable, List, Optional, Sequence, Tuple, Union

import torch
import torch.nn.functional as F
from torch import nn

HEURISTIC_DAMPING_SCALE = 0.1  # utils/constants.py:22
LAMBDA_DTYPE = torch.float64  # utils/constants.py:82

Count = Union[int, torch.Tensor]

# --------------------------------------------------------------------------------------------
# Per-module operator restatements
# --------------------------------------------------------------------------------------------


def linear_flat_activation(
    x: torch.Tensor, mask: Optional[torch.Tensor], has_bias: bool
) -> Tuple[torch.Tensor, Count]:
    """module/linear.py:30-46 -- rows = every leading index, masked rows zeroed (bias column too)."""
    flat = x.reshape(-1, x.shape[-1])
    flat_mask = None
    if mask is not None and flat.shape[0] == mask.numel():
        flat_mask = mask.reshape(-1, 1)
        flat = flat * flat_mask  # reference multiplies in place on its private copy
    if has_bias:
        ones = flat.new_ones((flat.shape[0], 1))
        if flat_mask is not None:
            ones = ones * flat_mask
        flat = torch.cat([flat, ones], dim=-1)
    count = flat.shape[0] if flat_mask is None else flat_mask.sum()
    return flat, count


def linear_flat_gradient(g: torch.Tensor, mask: Optional[torch.Tensor]) -> Tuple[torch.Tensor, Count]:
    """module/linear.py:48-54 -- the gradient is NOT masked; only the count uses the mask."""
    flat = g.reshape(-1, g.shape[-1])
    if mask is not None and flat.shape[0] == mask.numel():
        return flat, mask.sum()
    return flat, flat.shape[0]


def conv_patches(x: torch.Tensor, conv: nn.Conv2d) -> torch.Tensor:
    """module/conv2d.py:15-64 -- group-mean then unfold; result ``[b, P, C_in/groups*k1*k2]``."""
    padding = conv.padding
    if isinstance(padding, str):
        resolved = []
        for k, s, d in zip(conv.kernel_size, conv.stride, conv.dilation):
            if padding == "valid":
                left = right = 0
            else:
                total = d * (k - 1)
                left, right = total // 2, total - total // 2
            if left != right:
                raise ValueError("Unequal padding not supported in unfold.")
            resolved.append(left)
        padding = tuple(resolved)
    b, c, h, w = x.shape
    g = conv.groups
    x = x.reshape(b, g, c // g, h, w).mean(dim=1)
    unfolded = F.unfold(x, kernel_size=conv.kernel_size, dilation=conv.dilation, padding=padding, stride=conv.stride)
    return unfolded.transpose(1, 2)


def conv_flat_activation(x: torch.Tensor, conv: nn.Conv2d) -> Tuple[torch.Tensor, int]:
    """module/conv2d.py:106-128."""
    patches = conv_patches(x, conv)
    flat = patches.reshape(-1, patches.shape[-1])
    if conv.bias is not None:
        flat = torch.cat([flat, flat.new_ones((flat.shape[0], 1))], dim=-1)
    return flat, flat.shape[0]


def conv_flat_gradient(g: torch.Tensor) -> Tuple[torch.Tensor, int]:
    """module/conv2d.py:130-132 -- ``b c o1 o2 -> (b o1 o2) c``."""
    flat = g.permute(0, 2, 3, 1).reshape(-1, g.shape[1])
    return flat, flat.shape[0]


def covariance_update(cov: torch.Tensor, rows: torch.Tensor, alpha: float = 1.0) -> None:
    """module/tracker/factor.py:58 and :93 -- ``cov.addmm_(rows.t(), rows, alpha=alpha)``."""
    cov.addmm_(rows.t(), rows, alpha=alpha)


def eigendecompose(
    cov: torch.Tensor, count: torch.Tensor, eig_dtype: torch.dtype = torch.float64
) -> Tuple[torch.Tensor, torch.Tensor]:
    """factor/eigen.py:193-219 -- normalise, symmetrise, ``eigh`` in ``eig_dtype``, cast back."""
    original = cov.dtype
    c = cov.to(dtype=eig_dtype)
    c = c / count.to(dtype=eig_dtype)
    c = c + c.t()
    c.mul_(0.5)
    evals, evecs = torch.linalg.eigh(c)
    return evals.contiguous().to(dtype=original), evecs.contiguous().to(dtype=original)


def eigh_invariants(cov: torch.Tensor, count: torch.Tensor, evals: torch.Tensor, evecs: torch.Tensor) -> Dict[str, float]:
    """the acceptance test for an eigensolver (fp64 arithmetic)."""
    c = cov.double() / count.double()
    c = 0.5 * (c + c.t())
    q, lam = evecs.double(), evals.double()
    n = c.shape[0]
    scale = max(float(c.norm()), 1e-300)
    return {
        "orthogonality": float((q.t() @ q - torch.eye(n, dtype=torch.float64)).norm()) / math.sqrt(n),
        "reconstruction": float((q @ torch.diag(lam) @ q.t() - c).norm()) / scale,
        "ascending": float((lam[1:] - lam[:-1]).clamp(max=0).abs().max()) / scale if n > 1 else 0.0,
    }


def _with_bias_column(a: torch.Tensor, has_bias: bool) -> torch.Tensor:
    """module/linear.py:56-61 / module/conv2d.py:134-157 -- un-masked ones column."""
    if not has_bias:
        return a
    return torch.cat([a, a.new_ones(list(a.shape[:-1]) + [1])], dim=-1)


def linear_per_sample_gradient(a: torch.Tensor, g: torch.Tensor, has_bias: bool) -> torch.Tensor:
    """module/linear.py:68-77 -- ``einsum("b...i,b...o->bio", g, a')`` -> ``[b, O, I']``."""
    a = _with_bias_column(a, has_bias)
    return torch.einsum("b...i,b...o->bio", g, a)


def _conv_operands(a: torch.Tensor, g: torch.Tensor, conv: nn.Conv2d) -> Tuple[torch.Tensor, torch.Tensor]:
    patches = conv_patches(a, conv)
    patches = _with_bias_column(patches, conv.bias is not None)
    grads = g.flatten(2).transpose(1, 2)  # b o i1 i2 -> b (i1 i2) o
    return patches, grads


def conv_per_sample_gradient(a: torch.Tensor, g: torch.Tensor, conv: nn.Conv2d) -> torch.Tensor:
    """module/conv2d.py:164-177."""
    patches, grads = _conv_operands(a, g, conv)
    return torch.einsum("bci,bco->bio", grads, patches)


def lambda_update(lam: torch.Tensor, psg: torch.Tensor, q_a: torch.Tensor, q_g: torch.Tensor) -> None:
    """module/tracker/factor.py:218-226 -- ``lam += sum_b (Qg^T (g_b Qa))^2`` (two matmuls)."""
    rotated = torch.matmul(q_g.t(), torch.matmul(psg, q_a))
    lam.add_(rotated.square_().sum(dim=0))


def ekfac_inverse_lambda(
    lam: torch.Tensor, num_processed: torch.Tensor, damping: Optional[float], out_dtype: torch.dtype
) -> torch.Tensor:
    """factor/config.py:331-338 -- fp64: ``1 / (lam / n + damping)``; ``None`` -> 0.1 * mean."""
    work = lam.to(dtype=LAMBDA_DTYPE).clone()
    work.div_(num_processed.to(dtype=LAMBDA_DTYPE))
    if damping is None:
        damping = HEURISTIC_DAMPING_SCALE * torch.mean(work)
    work.add_(damping)
    work.reciprocal_()
    return work.to(dtype=out_dtype).contiguous()


def ekfac_precondition(g: torch.Tensor, q_a: torch.Tensor, q_g: torch.Tensor, lam_inv: torch.Tensor) -> torch.Tensor:
    """factor/config.py:350-352 -- four matmuls and one elementwise product."""
    rotated = torch.matmul(q_g.t(), torch.matmul(g, q_a))
    rotated.mul_(lam_inv)
    return torch.matmul(q_g, torch.matmul(rotated, q_a.t()))


def kfac_inverse_lambda(evals_a: torch.Tensor, evals_g: torch.Tensor, damping: Optional[float],
                        out_dtype: torch.dtype) -> torch.Tensor:
    """factor/config.py:264-276 (K-FAC ``prepare``) -- ``1 / (lambda_G (x) lambda_A + damping)``, fp64."""
    work = torch.kron(evals_a.to(dtype=LAMBDA_DTYPE).unsqueeze(0), evals_g.to(dtype=LAMBDA_DTYPE).unsqueeze(-1))
    if damping is None:
        damping = HEURISTIC_DAMPING_SCALE * torch.mean(work)
    work.add_(damping)
    work.reciprocal_()
    return work.to(dtype=out_dtype).contiguous()


def diagonal_lambda_update(lam: torch.Tensor, psg: torch.Tensor) -> None:
    """module/tracker/factor.py:227-229 -- without an eigenbasis Lambda accumulates the squared gradient itself."""
    lam.add_(psg.square().sum(dim=0))


def precondition_with_strategy(strategy: str, g: torch.Tensor, q_a: Optional[torch.Tensor], q_g: Optional[torch.Tensor],
                               lam_inv: Optional[torch.Tensor]) -> torch.Tensor:
    """``FactorConfig.precondition_gradient`` of the four strategies (factor/config.py:128-353)."""
    if strategy == "identity":
        return g
    if strategy == "diagonal":
        return g * lam_inv
    return ekfac_precondition(g, q_a, q_g, lam_inv)  # ekfac and kfac differ only in lam_inv


def self_score(preconditioned: torch.Tensor, psg: torch.Tensor) -> torch.Tensor:
    """module/tracker/self_score.py:61-62 (and :164 for the measurement variant) -- ``sum(P o g)`` per sample."""
    return (preconditioned * psg).sum(dim=(1, 2))


def linear_pairwise_score(p: torch.Tensor, a: torch.Tensor, g: torch.Tensor, has_bias: bool) -> torch.Tensor:
    """module/linear.py:112-122 -- ``"qio,b...i,b...o->qb"``.

    The reference asks opt_einsum (un-vendored, >=3.3.0) for a flop-optimal order; for a
    2-D activation that contracts ``p`` with one operand first, for sequences it forms the
    per-sample gradient first.  Both orders are written out here.
    """
    a = _with_bias_column(a, has_bias)
    if a.dim() == 2:
        partial = torch.einsum("qio,bo->qib", p, a)  # label "o" = input dim, as in the reference
        return torch.einsum("qib,bi->qb", partial, g)
    psg = torch.einsum("b...i,b...o->bio", g, a)
    return torch.einsum("qio,bio->qb", p, psg)


def linear_pairwise_score_low_rank(left: torch.Tensor, right: torch.Tensor, a: torch.Tensor, g: torch.Tensor,
                                   has_bias: bool) -> torch.Tensor:
    """module/linear.py:83-99 -- ``"qik,qko,b...i,b...o->qb"`` for a preconditioned query gradient held as the factor pair
    ``[left [Q, O, k], right [Q, k, I']]`` of module/tracker/precondition.py:19-75 (the reference's labels: "i" = output dim,
    "o" = input dim).  Written in the order that never forms an ``[O, I']`` block -- ``U = G L_q``, ``V = A' R_q^T``, then the sum
    over (row, k) -- which equals the dense contraction with ``P_q = L_q R_q`` exactly (tests/test_oracle_golden.py)."""
    a = _with_bias_column(a, has_bias)
    u = torch.einsum("b...i,qik->qb...k", g, left)
    v = torch.einsum("b...o,qko->qb...k", a, right)
    return (u * v).flatten(2).sum(dim=2)


def conv_pairwise_score(p: torch.Tensor, a: torch.Tensor, g: torch.Tensor, conv: nn.Conv2d) -> torch.Tensor:
    """module/conv2d.py:199-209 -- ``"qio,bti,bto->qb"``."""
    patches, grads = _conv_operands(a, g, conv)
    psg = torch.einsum("bti,bto->bio", grads, patches)
    return torch.einsum("qio,bio->qb", p, psg)


# --------------------------------------------------------------------------------------------
# Stage loops on a plain nn.Module (factor/covariance.py, factor/eigen.py, score/pairwise.py,
# score/dot_product.py), single process, CPU.
# --------------------------------------------------------------------------------------------


@dataclass
class OracleDtypes:
    """The dtype knobs of FactorArguments / ScoreArguments that touch arithmetic (arguments.py)."""

    activation_covariance: torch.dtype = torch.float32
    gradient_covariance: torch.dtype = torch.float32
    eigendecomposition: torch.dtype = torch.float64
    per_sample_gradient: torch.dtype = torch.float32
    lambda_: torch.dtype = torch.float32
    precondition: torch.dtype = torch.float32
    score: torch.dtype = torch.float32

    @classmethod
    def all(cls, dtype: torch.dtype) -> "OracleDtypes":
        return cls(dtype, dtype, torch.float64, dtype, dtype, dtype, dtype)


Factors = Dict[str, Dict[str, torch.Tensor]]
LossFn = Callable[[nn.Module, object], torch.Tensor]
MaskFn = Optional[Callable[[object], Optional[torch.Tensor]]]


class _Probe(nn.Module):
    """Adds a zero that requires grad so a frozen layer's output still gets a tensor hook
    (module/tracked_module.py:97-103,165-168)."""

    def __init__(self, dtype: torch.dtype) -> None:
        super().__init__()
        self.zero = nn.Parameter(torch.zeros(1, dtype=dtype))


class OracleEngine:
    """Runs the three EK-FAC stages on a model's ``nn.Linear`` / ``nn.Conv2d`` leaves."""

    def __init__(self, model: nn.Module, module_names: Optional[Sequence[str]] = None,
                 dtypes: Optional[OracleDtypes] = None) -> None:
        self.model = model.eval()
        for p in self.model.parameters():
            p.requires_grad_(False)  # analyzer.py:37-41
        self.dtypes = dtypes or OracleDtypes()
        self.layers: Dict[str, nn.Module] = {}
        for name, module in self.model.named_modules():
            if len(list(module.children())) > 0:
                continue
            if module_names is not None and name not in module_names:
                continue
            if isinstance(module, (nn.Linear, nn.Conv2d)):
                self.layers[name] = module
        if not self.layers:
            raise ValueError("no nn.Linear / nn.Conv2d leaf found")
        self._probes = {name: _Probe(m.weight.dtype) for name, m in self.layers.items()}
        self._handles: List[torch.utils.hooks.RemovableHandle] = []
        self.mask: Optional[torch.Tensor] = None

    # ---- hook plumbing ---------------------------------------------------------------------
    def _install(self, on_forward: Callable[[str, nn.Module, torch.Tensor], Callable[[torch.Tensor], None]]) -> None:
        self._remove()
        for name, module in self.layers.items():
            probe = self._probes[name]

            def fwd(mod, inputs, output, name=name, probe=probe):
                out = output if output.requires_grad else output + probe.zero
                with torch.no_grad():
                    on_backward = on_forward(name, mod, inputs[0].detach())
                handle_box = []

                def bwd(grad, on_backward=on_backward):
                    handle_box[0].remove()
                    with torch.no_grad():
                        on_backward(grad.detach())

                handle_box.append(out.register_hook(bwd))
                return out

            self._handles.append(module.register_forward_hook(fwd))

    def _remove(self) -> None:
        for h in self._handles:
            h.remove()
        self._handles = []

    def _backward(self, value: torch.Tensor) -> None:
        self.model.zero_grad(set_to_none=True)
        value.backward()

    # ---- stage 1: covariance (factor/covariance.py:153-266, tracker/factor.py:25-149) --------
    def fit_covariance(self, batches: Iterable[object], loss_fn: LossFn, mask_fn: MaskFn = None) -> Factors:
        d = self.dtypes
        out: Factors = {k: {} for k in ("activation_covariance", "gradient_covariance",
                                        "num_activation_covariance_processed",
                                        "num_gradient_covariance_processed")}

        def on_forward(name, mod, x):
            x = x.to(dtype=d.activation_covariance)
            if isinstance(mod, nn.Linear):
                flat, count = linear_flat_activation(x, self.mask, mod.bias is not None)
            else:
                flat, count = conv_flat_activation(x, mod)
            if name not in out["activation_covariance"]:
                out["activation_covariance"][name] = torch.zeros(flat.shape[1], flat.shape[1], dtype=flat.dtype)
                out["num_activation_covariance_processed"][name] = torch.zeros(1, dtype=torch.int64)
            out["num_activation_covariance_processed"][name].add_(count)
            covariance_update(out["activation_covariance"][name], flat)

            def on_backward(g):
                g = g.to(dtype=d.gradient_covariance)
                if isinstance(mod, nn.Linear):
                    gflat, gcount = linear_flat_gradient(g, self.mask)
                else:
                    gflat, gcount = conv_flat_gradient(g)
                if name not in out["gradient_covariance"]:
                    out["gradient_covariance"][name] = torch.zeros(gflat.shape[1], gflat.shape[1], dtype=gflat.dtype)
                    out["num_gradient_covariance_processed"][name] = torch.zeros(1, dtype=torch.int64)
                out["num_gradient_covariance_processed"][name].add_(gcount)
                covariance_update(out["gradient_covariance"][name], gflat)

            return on_backward

        self._install(on_forward)
        for batch in batches:
            self.mask = mask_fn(batch) if mask_fn is not None else None
            self._backward(loss_fn(self.model, batch))
        self.mask = None
        self._remove()
        return out

    # ---- stage 2a: eigendecomposition (factor/eigen.py:140-224) -----------------------------
    def eigendecomposition(self, cov: Factors) -> Factors:
        out: Factors = {k: {} for k in ("activation_eigenvectors", "activation_eigenvalues",
                                        "gradient_eigenvectors", "gradient_eigenvalues")}
        for name in self.layers:
            for side in ("activation", "gradient"):
                evals, evecs = eigendecompose(cov[f"{side}_covariance"][name],
                                              cov[f"num_{side}_covariance_processed"][name],
                                              self.dtypes.eigendecomposition)
                out[f"{side}_eigenvalues"][name] = evals
                out[f"{side}_eigenvectors"][name] = evecs
        return out

    def _per_sample_gradient(self, mod: nn.Module, a: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
        if isinstance(mod, nn.Linear):
            return linear_per_sample_gradient(a, g, mod.bias is not None)
        return conv_per_sample_gradient(a, g, mod)

    # ---- stage 2b: Lambda (factor/eigen.py:345-462, tracker/factor.py:152-327) ---------------
    def fit_lambda(self, batches: Iterable[object], loss_fn: LossFn, eig: Factors) -> Factors:
        d = self.dtypes
        out: Factors = {"lambda_matrix": {}, "num_lambda_processed": {}}
        q_a = {n: eig["activation_eigenvectors"][n].to(dtype=d.lambda_) for n in self.layers}
        q_g = {n: eig["gradient_eigenvectors"][n].to(dtype=d.lambda_) for n in self.layers}

        def on_forward(name, mod, x):
            cached = x.to(dtype=d.per_sample_gradient, copy=True)

            def on_backward(g):
                g = g.to(dtype=d.per_sample_gradient)
                psg = self._per_sample_gradient(mod, cached, g).to(dtype=d.lambda_)
                if name not in out["lambda_matrix"]:
                    out["lambda_matrix"][name] = torch.zeros(psg.shape[1], psg.shape[2], dtype=psg.dtype)
                    out["num_lambda_processed"][name] = torch.zeros(1, dtype=torch.int64)
                out["num_lambda_processed"][name].add_(psg.shape[0])
                lambda_update(out["lambda_matrix"][name], psg, q_a[name], q_g[name])

            return on_backward

        self._install(on_forward)
        for batch in batches:
            self._backward(loss_fn(self.model, batch))
        self._remove()
        return out

    # ---- other strategies, self-influence ---------------------------------------------
    def fit_diagonal_lambda(self, batches: Iterable[object], loss_fn: LossFn) -> Factors:
        """``strategy="diagonal"``: Lambda in parameter space (factor/eigen.py:345-462 with
        ``requires_eigendecomposition_for_lambda = False``)."""
        d = self.dtypes
        out: Factors = {"lambda_matrix": {}, "num_lambda_processed": {}}

        def on_forward(name, mod, x):
            cached = x.to(dtype=d.per_sample_gradient, copy=True)

            def on_backward(g):
                psg = self._per_sample_gradient(mod, cached, g.to(dtype=d.per_sample_gradient)).to(dtype=d.lambda_)
                if name not in out["lambda_matrix"]:
                    out["lambda_matrix"][name] = torch.zeros(psg.shape[1], psg.shape[2], dtype=psg.dtype)
                    out["num_lambda_processed"][name] = torch.zeros(1, dtype=torch.int64)
                out["num_lambda_processed"][name].add_(psg.shape[0])
                diagonal_lambda_update(out["lambda_matrix"][name], psg)

            return on_backward

        self._install(on_forward)
        for batch in batches:
            self._backward(loss_fn(self.model, batch))
        self._remove()
        return out

    def _strategy_state(self, strategy: str, eig: Optional[Factors], lam: Optional[Factors], damping: Optional[float]):
        """Per-layer ``(Q_A, Q_G, Lambda^-1)`` as ``FactorConfig.prepare`` leaves them (factor/config.py)."""
        d = self.dtypes
        state = {}
        for n in self.layers:
            q_a = q_g = lam_inv = None
            if strategy in ("ekfac", "kfac"):
                q_a = eig["activation_eigenvectors"][n].to(dtype=d.precondition)
                q_g = eig["gradient_eigenvectors"][n].to(dtype=d.precondition)
            if strategy == "ekfac":
                lam_inv = ekfac_inverse_lambda(lam["lambda_matrix"][n], lam["num_lambda_processed"][n], damping, d.precondition)
            elif strategy == "kfac":
                lam_inv = kfac_inverse_lambda(eig["activation_eigenvalues"][n], eig["gradient_eigenvalues"][n], damping,
                                              d.precondition)
            elif strategy == "diagonal":
                lam_inv = ekfac_inverse_lambda(lam["lambda_matrix"][n], lam["num_lambda_processed"][n], damping, d.precondition)
            state[n] = (q_a, q_g, lam_inv)
        return state

    def self_scores(self, train_batches: Iterable[object], loss_fn: LossFn, eig: Optional[Factors], lam: Optional[Factors],
                    damping: Optional[float] = 1e-8, strategy: str = "ekfac", measure_fn: LossFn = None) -> torch.Tensor:
        """score/self.py:135-290; with ``measure_fn`` the measurement variant of :293-443 (two backward passes)."""
        d = self.dtypes
        state = self._strategy_state(strategy, eig, lam, damping)
        held: Dict[str, torch.Tensor] = {}
        per_layer: Dict[str, torch.Tensor] = {}

        def gradient_of(name, mod, x, g):
            return self._per_sample_gradient(mod, x, g.to(dtype=d.per_sample_gradient)).to(dtype=d.precondition)

        def measure_forward(name, mod, x):
            cached = x.to(dtype=d.per_sample_gradient, copy=True)

            def on_backward(g):
                held[name] = precondition_with_strategy(strategy, gradient_of(name, mod, cached, g), *state[name]).to(dtype=d.score)

            return on_backward

        def loss_forward(name, mod, x):
            cached = x.to(dtype=d.per_sample_gradient, copy=True)

            def on_backward(g):
                psg = gradient_of(name, mod, cached, g)
                pre = held[name] if measure_fn is not None else precondition_with_strategy(strategy, psg, *state[name])
                per_layer[name] = self_score(pre.to(dtype=d.score), psg.to(dtype=d.score))

            return on_backward

        chunks: List[torch.Tensor] = []
        for batch in train_batches:
            if measure_fn is not None:
                self._install(measure_forward)
                self._backward(measure_fn(self.model, batch))
            self._install(loss_forward)
            self._backward(loss_fn(self.model, batch))
            chunks.append(sum(per_layer[n] for n in self.layers))
            per_layer.clear()
            held.clear()
        self._remove()
        return torch.cat(chunks, dim=0)

    # ---- stage 3: pairwise scores (score/pairwise.py:133-293, score/dot_product.py:39-153) ---
    def precondition_queries(self, query_batches: Iterable[object], measure_fn: LossFn, eig: Optional[Factors],
                             lam: Optional[Factors], damping: Optional[float], strategy: str = "ekfac") -> Dict[str, torch.Tensor]:
        d = self.dtypes
        state = self._strategy_state(strategy, eig, lam, damping)
        held: Dict[str, List[torch.Tensor]] = {n: [] for n in self.layers}

        def on_forward(name, mod, x):
            cached = x.to(dtype=d.per_sample_gradient, copy=True)

            def on_backward(g):  # tracker/precondition.py:102-123
                g = g.to(dtype=d.per_sample_gradient)
                psg = self._per_sample_gradient(mod, cached, g).to(dtype=d.precondition)
                held[name].append(precondition_with_strategy(strategy, psg, *state[name]).to(dtype=d.score))

            return on_backward

        self._install(on_forward)
        for batch in query_batches:
            self._backward(measure_fn(self.model, batch))
        self._remove()
        return {n: torch.cat(v, dim=0).contiguous() for n, v in held.items()}

    def pairwise_scores(self, query_batches: Iterable[object], train_batches: Iterable[object],
                        measure_fn: LossFn, loss_fn: LossFn, eig: Optional[Factors], lam: Optional[Factors],
                        damping: Optional[float] = 1e-8, strategy: str = "ekfac") -> torch.Tensor:
        d = self.dtypes
        precond = self.precondition_queries(query_batches, measure_fn, eig, lam, damping, strategy)
        per_layer: Dict[str, torch.Tensor] = {}

        def on_forward(name, mod, x):
            cached = x.to(dtype=d.score, copy=True)

            def on_backward(g):  # tracker/pairwise_score.py:73-103
                g = g.to(dtype=d.score)
                if isinstance(mod, nn.Linear):
                    per_layer[name] = linear_pairwise_score(precond[name], cached, g, mod.bias is not None)
                else:
                    per_layer[name] = conv_pairwise_score(precond[name], cached, g, mod)

            return on_backward

        self._install(on_forward)
        chunks: List[torch.Tensor] = []
        for batch in train_batches:
            self._backward(loss_fn(self.model, batch))
            total = None
            for name in self.layers:  # score/dot_product.py:105-117
                total = per_layer[name].clone() if total is None else total.add_(per_layer[name])
            chunks.append(total)
            per_layer.clear()
        self._remove()
        return torch.cat(chunks, dim=1)
