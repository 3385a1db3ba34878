"""Covariance and Lambda trackers (reference ``module/tracker/factor.py:25-327``).

MI355X-first differences, results identical within fp tolerance:

* accumulators are fp32 tensors resident in HBM whatever the factor dtype (the reference
  accumulates in the factor dtype, e.g. bf16); they are cast once, on export;
* the covariance hooks feed the hooked tensors straight to ``kf_syrk_accum`` -- flatten, mask,
  ones column and ``addmm_`` are one kernel;
* Lambda never materialises the ``[b, O, I']`` per-sample gradient: it rotates the gradient's
  factors (``G Qg``, ``[A,1] Qa``) and squares their batched product (``kf_lambda_accum``), which is
  the same mathematics at 2R(I'^2+O^2+OI') instead of 2OI'(I'+O+R) flops per sample.
"""

from __future__ import annotations

import weakref
from typing import Tuple

import torch
import torch.distributed as dist
from torch import nn

from kronfluence_amd import ops
from kronfluence_amd.factor.config import FactorConfig
from kronfluence_amd.module.tracker.base import BaseTracker
from kronfluence_amd.utils.constants import (
    ACTIVATION_COVARIANCE_MATRIX_NAME,
    ACTIVATION_EIGENVECTORS_NAME,
    COVARIANCE_FACTOR_NAMES,
    EIGENDECOMPOSITION_FACTOR_NAMES,
    GRADIENT_COVARIANCE_MATRIX_NAME,
    GRADIENT_EIGENVECTORS_NAME,
    LAMBDA_FACTOR_NAMES,
    LAMBDA_MATRIX_NAME,
    NUM_ACTIVATION_COVARIANCE_PROCESSED,
    NUM_GRADIENT_COVARIANCE_PROCESSED,
    NUM_LAMBDA_PROCESSED,
)
from kronfluence_amd.utils.exceptions import FactorsNotFoundError


def _all_reduce_sum(tensors) -> None:
    """SUM over ranks (RCCL when the tensors are on GPU; gloo in the CPU tests)."""
    for t in tensors:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)


def _to_covariance_dtype(t: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """The reference casts the hooked tensor to ``activation_covariance_dtype`` / ``gradient_covariance_dtype`` ahead of the
    update (tracker/factor.py:101-107, 118).  Here the accumulator is fp32 whatever that dtype and the kernels take bf16 /
    fp16 / fp32 rows with exact products, so only a NARROWING cast changes the result -- and it moves e.g. the fp32 LayerNorm
    outputs of an autocast model onto the bf16 LDS-DMA covariance kernel under the reference's low-precision presets."""
    if dtype in (torch.bfloat16, torch.float16) and t.is_floating_point() and t.dtype != dtype:
        return t.to(dtype)
    return t


class _SharedInput:
    """The activation-covariance increment of the tensor the PREVIOUS tracked Linear consumed (see CovarianceTracker)."""
    ref = None          # weakref to that tensor object
    mask_ref = None     # weakref to the attention mask the increment was formed under (None: no mask)
    version = -1        # its in-place version counter when it was consumed
    key = None          # everything else the increment depends on
    delta = None        # the increment X'^T X' in a buffer of its own (None: its owner accumulated in place)
    count = None        # the matching row-count increment
    owner = None        # the tracker that computed it


class CovarianceTracker(BaseTracker):
    # Several tracked Linear layers often consume ONE tensor -- the query / key / value projections of an attention block, the gate /
    # up projections of a SwiGLU MLP -- so their activation-covariance increments are the same matrix.  The reference computes each
    # (tracker/factor.py:98-112: three identical addmm_).  Here the first layer of such a group ("leader", learnt during the first
    # batch: a layer whose successor was handed the very same tensor object, unmodified) forms its increment in a buffer of its own
    # and the followers ADD that buffer instead of running the covariance kernel again: BERT 2 of 6, Llama 3 of 7 activation
    # covariances per block.  Same values as separate accumulations up to the order of fp32 additions.
    SHARE_INPUT_INCREMENTS = True
    _shared = _SharedInput()
    _leads = False   # this tracker's increments are wanted by the layers that follow it

    def register_hooks(self) -> None:
        module = self.module
        storage = module.storage

        def shared_key(source: torch.Tensor):
            mask = module.attention_mask
            return (tuple(source.shape), source.dtype, module.factor_args.activation_covariance_dtype, module.has_bias,
                    None if mask is None else mask._version)

        def same_mask() -> bool:   # identity, through a weak reference: a recycled id() of a freed mask can never match
            mask, held = module.attention_mask, CovarianceTracker._shared.mask_ref
            return (mask is None and held is None) or (mask is not None and held is not None and held() is mask)

        def remember(source: torch.Tensor) -> None:
            shared = CovarianceTracker._shared
            mask = module.attention_mask
            shared.ref, shared.version, shared.key = weakref.ref(source), source._version, shared_key(source)
            shared.mask_ref = None if mask is None else weakref.ref(mask)

        @torch.no_grad()
        def forward_hook(mod: nn.Module, inputs: Tuple[torch.Tensor], outputs: torch.Tensor) -> None:
            del mod
            source = inputs[0]
            hooked = source.detach()
            shared = CovarianceTracker._shared
            # (not beside the pass on a second stream: a leader's increment published from the side stream could be read by a follower
            # on the main stream before it is complete -- ADVICE r05; the side stream is an opt-in experiment, sharing is the default)
            sharing = (self.SHARE_INPUT_INCREMENTS and type(module).__name__ == "TrackedLinear" and not module.factor_args.has_shared_parameters
                       and self._side_stream(hooked.device) is None)
            hit = (sharing and shared.ref is not None and shared.ref() is source and shared.version == source._version
                   and shared.key == shared_key(source) and same_mask() and shared.owner is not self)

            def update() -> None:
                cov, count = storage[ACTIVATION_COVARIANCE_MATRIX_NAME], storage[NUM_ACTIVATION_COVARIANCE_PROCESSED]
                follower = hit and shared.delta is not None
                # (a follower never touches the rows: no narrowing cast of the activation for nothing)
                rows = None if follower else _to_covariance_dtype(hooked, module.factor_args.activation_covariance_dtype)
                if follower:   # the leader's increment is this layer's increment
                    if cov is None:
                        cov, count = torch.zeros_like(shared.delta), torch.zeros_like(shared.count)
                    cov.add_(shared.delta)
                    count.add_(shared.count)
                elif sharing and self._leads:          # leader: the increment in a buffer of its own, then added
                    delta, rows_seen = module.accumulate_activation_covariance(None, None, rows)
                    if cov is None:
                        cov, count = torch.zeros_like(delta), torch.zeros_like(rows_seen)
                    cov.add_(delta)
                    count.add_(rows_seen)
                    remember(source)
                    shared.delta, shared.count, shared.owner = delta, rows_seen, self
                else:
                    if hit:   # first sighting of the group: from the next batch on the previous layer publishes its increments
                        shared.owner._leads = True
                    cov, count = module.accumulate_activation_covariance(cov, count, rows)
                    if sharing and not hit:
                        remember(source)
                        shared.delta, shared.count, shared.owner = None, None, self
                storage[ACTIVATION_COVARIANCE_MATRIX_NAME] = cov
                storage[NUM_ACTIVATION_COVARIANCE_PROCESSED] = count

            # beside the rest of the forward pass when the stage loop allows it and memory is plentiful (BaseTracker._run_beside);
            # the batch's attention mask is read by the same kernels and freed by the stage loop: kept alive for them too
            mask = module.attention_mask
            self._run_beside(hooked.device, (hooked,) if mask is None else (hooked, mask), update)
            self.cached_hooks.append(outputs.register_hook(backward_hook))

        @torch.no_grad()
        def backward_hook(output_gradient: torch.Tensor) -> None:
            self.cached_hooks.pop().remove()
            alpha = module.gradient_scale**2.0 if module.gradient_scale != 1.0 else 1.0  # factor.py:90-92
            hooked = output_gradient.detach()

            def update() -> None:
                cov, count = module.accumulate_gradient_covariance(
                    storage[GRADIENT_COVARIANCE_MATRIX_NAME], storage[NUM_GRADIENT_COVARIANCE_PROCESSED],
                    _to_covariance_dtype(hooked, module.factor_args.gradient_covariance_dtype), alpha)
                storage[GRADIENT_COVARIANCE_MATRIX_NAME] = cov
                storage[NUM_GRADIENT_COVARIANCE_PROCESSED] = count

            mask = module.attention_mask
            self._run_beside(hooked.device, (hooked,) if mask is None else (hooked, mask), update)

        self.registered_hooks.append(module.register_forward_hook(forward_hook))

    def exist(self) -> bool:
        return all(self.module.storage[name] is not None for name in COVARIANCE_FACTOR_NAMES)

    def synchronize(self, num_processes: int) -> None:
        """Stand-alone form of C1 (reference ``factor.py:132-142``); the stage loop normally uses the
        bucketed ``module.utils.synchronize_factors`` instead (one flat all-reduce for all layers)."""
        del num_processes
        if dist.is_initialized() and self.exist():
            _all_reduce_sum([self.module.storage[name] for name in COVARIANCE_FACTOR_NAMES])

    def release_memory(self) -> None:
        for name in COVARIANCE_FACTOR_NAMES:
            self.module.storage[name] = None
        self._leads = False
        shared = CovarianceTracker._shared
        if shared.owner is self:
            shared.ref = shared.mask_ref = shared.delta = shared.count = shared.owner = None


class LambdaTracker(BaseTracker):
    def _eigenvectors(self, device: torch.device) -> Tuple[torch.Tensor, torch.Tensor]:
        storage = self.module.storage
        if storage[ACTIVATION_EIGENVECTORS_NAME] is None or storage[GRADIENT_EIGENVECTORS_NAME] is None:
            raise FactorsNotFoundError(
                f"The strategy {self.module.factor_args.strategy} requires eigendecomposition "
                f"results for Lambda computations, but they are not found."
            )
        # bf16 lambda_dtype + bf16-stored eigenvectors: the bf16 rotations below build their operands from the stored matrices, no
        # fp32 copy is made (Llama-3-8B at full depth: 99 GB); the fp32 engine converts on first use (``_eigenvectors32``)
        keep = self.module.factor_args.lambda_dtype == torch.bfloat16
        for name in (ACTIVATION_EIGENVECTORS_NAME, GRADIENT_EIGENVECTORS_NAME):
            q = storage[name]
            want = q.dtype if (keep and q.dtype == torch.bfloat16) else torch.float32
            if q.device != device or q.dtype != want or not q.is_contiguous():
                storage[name] = q.to(device=device, dtype=want).contiguous()  # once (factor.py:191-201)
        return storage[ACTIVATION_EIGENVECTORS_NAME], storage[GRADIENT_EIGENVECTORS_NAME]

    _bf16_eigenvectors = None  # (Q_A^T, Q_G^T) in bf16, for lambda_dtype == bf16
    _conv_dense_eigenvectors = None  # (Q_A^T in the implicit-im2col patch order, Q_G) in bf16

    # Conv2d layers with more output positions than output channels (R > O: the early ResNet-9 stages) fit Lambda in the
    # DENSE form -- per-sample gradient first (implicit im2col, gradient side already rotated), then ONE tall GEMM with
    # Q_A whose epilogue squares and sums -- instead of rotating the [b R, I'] patch matrix: fewer flops (SURVEY.md 8d:
    # F_lambda is the cheaper of the two exact forms) and no patch tensor.  ``kf_lambda_conv2d_accum``.
    CONV_DENSE = True
    CONV_DENSE_FLOP_RATIO = 1.3
    # Linear layers on sequences with whole 64-deep k-tiles (O, I, R multiples of 64: every transformer config): rotations
    # written K-contiguous per sample + ``kf_lambda_rows_accum`` (round 4).  False selects the round-2 kernel (A/B, tests).
    ROWS_ENGINE = True

    def _offload_activations(self) -> bool:
        return bool(self.module.factor_args.offload_activations_to_cpu)   # reference tracker/factor.py:239

    @staticmethod
    def algorithmic_flops(r: int, o: int, ip: int) -> float:
        """F_lambda per sample: the cheaper of the dense and the factored exact formulations."""
        dense = 2.0 * o * ip * (ip + o) + (2.0 * r * o * ip if r > 1 else 0.0)
        return min(dense, 2.0 * r * (ip * ip + o * o + o * ip))

    def _update_conv_dense(self, activation: torch.Tensor, output_gradient: torch.Tensor) -> bool:
        module, storage, conv = self.module, self.module.storage, self.module.original_module
        if not (self.CONV_DENSE and isinstance(conv, nn.Conv2d) and activation.dim() == 4 and output_gradient.dim() == 4
                and module.factor_args.lambda_dtype == torch.bfloat16 and output_gradient.dtype == torch.bfloat16
                and output_gradient.is_cuda and self._rotates()):
            return False
        b, c = activation.shape[0], activation.shape[1]
        o, r = output_gradient.shape[1], output_gradient.shape[2] * output_gradient.shape[3]
        taps = conv.kernel_size[0] * conv.kernel_size[1]
        ip = c * taps
        dense = 2.0 * r * o * ip + 2.0 * o * ip * ip + 2.0 * r * o * o   # gradient + Q_A GEMM + channel rotation
        # the dense form runs on the implicit-im2col kernels (no patch tensor, ~0.8 PFLOP/s), the factored one materialises the
        # patches and rotates them (~0.5): dense is taken up to ``CONV_DENSE_FLOP_RATIO`` x the factored flops (ResNet-9's
        # (256, 1152) layer with R = O = 256: equal flops, 1.69 -> 1.0 ms per batch of 1 000; round 6, tools/r06_layer_times.py)
        if dense >= self.CONV_DENSE_FLOP_RATIO * 2.0 * r * (ip * ip + o * o + o * ip) or o % 8 != 0:
            return False
        geometry = ops.lambda_conv2d_geometry(tuple(activation.shape), o, conv)
        if geometry is None:
            return False
        q_a, q_g = self._eigenvectors(output_gradient.device)
        if storage[LAMBDA_MATRIX_NAME] is None:
            storage[LAMBDA_MATRIX_NAME] = torch.zeros((o, ip), dtype=torch.float32, device=output_gradient.device)
            storage[NUM_LAMBDA_PROCESSED] = torch.zeros(1, dtype=torch.int64)
        storage[NUM_LAMBDA_PROCESSED].add_(b)
        padded = ops.lambda_conv2d_channels(geometry)   # (may differ between batch sizes: a first layer's wide padding is budgeted by flops)
        if self._conv_dense_eigenvectors is None or self._conv_dense_eigenvectors[0] != padded:
            self._conv_dense_eigenvectors = (padded, ops.conv_patch_order_eigenvectors(q_a, c, taps, padded),
                                             q_g.to(torch.bfloat16).contiguous())
        _, qa_t_perm, qg16 = self._conv_dense_eigenvectors
        x = activation if activation.dtype == torch.bfloat16 else activation.to(torch.bfloat16)
        gt = ops.rotate_channels(output_gradient, qg16)
        ops.lambda_conv2d_accum(storage[LAMBDA_MATRIX_NAME], gt, x, geometry, qa_t_perm, scale=module.gradient_scale)
        return True

    def _rotates(self) -> bool:
        """EK-FAC fits Lambda in the Kronecker eigenbasis; the diagonal strategy in parameter space
        (``requires_eigendecomposition_for_lambda``, reference factor.py:184-201)."""
        return FactorConfig.CONFIGS[self.module.factor_args.strategy].requires_eigendecomposition_for_lambda

    def _update_from_factors(self, g: torch.Tensor, a: torch.Tensor, append_ones: bool) -> None:
        module, storage = self.module, self.module.storage
        b, r, o = g.shape
        if not self._rotates():
            ip = a.shape[-1] + int(append_ones)
            if storage[LAMBDA_MATRIX_NAME] is None:
                storage[LAMBDA_MATRIX_NAME] = torch.zeros((o, ip), dtype=torch.float32, device=g.device)
                storage[NUM_LAMBDA_PROCESSED] = torch.zeros(1, dtype=torch.int64)
            storage[NUM_LAMBDA_PROCESSED].add_(b)
            gt, at = ops.cast(g, torch.float32), ops.cast(a, torch.float32)
            if append_ones:
                at = torch.cat([at, at.new_ones(at.shape[:-1] + (1,))], dim=-1).contiguous()
            ops.lambda_accum(storage[LAMBDA_MATRIX_NAME], gt, at, b, r, scale=module.gradient_scale)
            return
        q_a, q_g = self._eigenvectors(g.device)
        ip = q_a.shape[0]
        if storage[LAMBDA_MATRIX_NAME] is None:
            storage[LAMBDA_MATRIX_NAME] = torch.zeros((o, ip), dtype=torch.float32, device=g.device)
            storage[NUM_LAMBDA_PROCESSED] = torch.zeros(1, dtype=torch.int64)
        storage[NUM_LAMBDA_PROCESSED].add_(b)  # samples, not tokens (factor.py:203)
        i = a.shape[-1]
        if (module.factor_args.lambda_dtype == torch.bfloat16 and g.dtype == torch.bfloat16 and a.dtype == torch.bfloat16
                and r > 1 and o % 8 == 0 and i % 8 == 0 and o >= 64 and i >= 64):
            # The reference's bf16 lambda_dtype casts eigenvectors and gradients to bf16 (factor.py:191-201);
            # here the rotations and the squared product run on the bf16 MFMA engine with fp32 accumulation.  The
            # augmented axis is carried at width W = I' rounded up to a multiple of 8 (zero columns), and the bias column
            # of ones is the row Q_A[I] added in the rotation's epilogue -- an odd I' (Linear with bias on sequences:
            # BERT, GPT-2) therefore stays on the bf16 engine without a torch.cat.
            if self._bf16_eigenvectors is None:
                pad = (-ip) % 8
                qa_t = torch.nn.functional.pad(q_a.t(), (0, pad, 0, pad)).to(torch.bfloat16).contiguous()
                self._bf16_eigenvectors = (qa_t, q_g.t().contiguous().to(torch.bfloat16),
                                           q_a[i].to(torch.float32).contiguous() if append_ones else None)
            qa_t, qg_t, bias_row = self._bf16_eigenvectors
            if self.ROWS_ENGINE and ops.lambda_rows_eligible(o, i, r):
                # round 4: both rotations written K-contiguous per sample, the per-sample product + square + sum over
                # samples on the LDS-DMA engine (kf_lambda_rows_accum) -- the factors are read once
                gt_t = ops.rotate_rows_transposed(g, qg_t)
                at_t = ops.rotate_rows_transposed(a, qa_t, bias_row)
                ops.lambda_rows_accum(storage[LAMBDA_MATRIX_NAME], gt_t, at_t, scale=module.gradient_scale)
                return
            gt = ops.rotate_bf16(g.reshape(b * r, o), qg_t)
            at = ops.rotate_bf16(a.reshape(b * r, i), qa_t, bias_row)
            ops.lambda_accum(storage[LAMBDA_MATRIX_NAME], gt, at, b, r, scale=module.gradient_scale)
            return
        q_a, q_g = self._eigenvectors32(ACTIVATION_EIGENVECTORS_NAME), self._eigenvectors32(GRADIENT_EIGENVECTORS_NAME)
        gt = ops.matmul_nn(g.reshape(b * r, o), q_g)
        at = ops.matmul_nn(a.reshape(b * r, a.shape[-1]), q_a, append_ones=append_ones)
        ops.lambda_accum(storage[LAMBDA_MATRIX_NAME], gt, at, b, r, scale=module.gradient_scale)

    def _update_from_gradient(self, per_sample_gradient: torch.Tensor) -> None:
        """Materialised-gradient form (post-processed or shared-parameter gradients):
        ``Lambda += sum_b (Qg^T g_b Qa)^2`` with both rotations on the MFMA engine."""
        storage = self.module.storage
        if self.module.factor_args.use_iterative_lambda_aggregation and per_sample_gradient.shape[0] > 1:
            # reference tracker/factor.py:203-213: sample by sample, so that the rotated [b, O, I'] fp32 copies of this path never
            # exist for the whole batch (the factored paths of ``_update_from_factors`` do not materialise them in the first place)
            for sample in range(per_sample_gradient.shape[0]):
                self._update_from_gradient(per_sample_gradient[sample:sample + 1])
            return
        g = per_sample_gradient.to(torch.float32).contiguous()
        b, o, ip = g.shape
        if storage[LAMBDA_MATRIX_NAME] is None:
            storage[LAMBDA_MATRIX_NAME] = torch.zeros((o, ip), dtype=torch.float32, device=g.device)
            storage[NUM_LAMBDA_PROCESSED] = torch.zeros(1, dtype=torch.int64)
        storage[NUM_LAMBDA_PROCESSED].add_(b)
        if self._rotates():
            self._eigenvectors(g.device)
            q_a, q_g = self._eigenvectors32(ACTIVATION_EIGENVECTORS_NAME), self._eigenvectors32(GRADIENT_EIGENVECTORS_NAME)
            t1 = torch.empty((b * o, ip), dtype=torch.float32, device=g.device)
            ops.gemm(t1, ip, 0, ops.view(g, 0, ip, 1, b * o, ip), ops.view(q_a, 0, 1, ip, ip, ip))
            rotated = torch.empty((b, o, ip), dtype=torch.float32, device=g.device)
            ops.gemm(rotated, ip, o * ip, ops.view(q_g, 0, 1, o, o, o), ops.view(t1, o * ip, 1, ip, ip, o), batch=b)
        else:
            rotated = g
        # Lambda[n] += scale^2 * sum_b rotated[b, n]^2: a 1 x (O I') GEMM over the batch axis, squares fused in the loader
        scale = self.module.gradient_scale
        ones = torch.ones(b, dtype=torch.float32, device=g.device)
        ops.gemm(storage[LAMBDA_MATRIX_NAME], o * ip, 0, ops.view(ones, 0, 0, 1, 1, b),
                 ops.view(rotated, 0, 1, o * ip, o * ip, b, square=True), alpha=scale * scale, beta=1.0)

    def register_hooks(self) -> None:
        module = self.module

        @torch.no_grad()
        def forward_hook(mod: nn.Module, inputs: Tuple[torch.Tensor], outputs: torch.Tensor) -> None:
            del mod
            self._cache_activation(inputs[0].detach())
            self.cached_hooks.append(
                outputs.register_hook(shared_backward_hook if module.factor_args.has_shared_parameters else backward_hook))

        @torch.no_grad()
        def backward_hook(output_gradient: torch.Tensor) -> None:
            activation = self._take_activation()
            self.cached_hooks.pop().remove()
            if module.per_sample_gradient_process_fnc is None:
                weight = module.original_module.weight
                o, ip = weight.shape[0], weight[0].numel() + int(module.has_bias)
                rows = output_gradient.numel() // (output_gradient.shape[0] * o)
                hooked = output_gradient.detach()

                def update() -> None:
                    # "lambda_update": the whole Lambda update of this hook (rotations included) against F_lambda
                    with ops._Timed("lambda_update", hooked.device, hooked.shape[0] * self.algorithmic_flops(rows, o, ip),
                                    float(hooked.numel() + activation.numel()) * hooked.element_size()):
                        if not self._update_conv_dense(activation, hooked):
                            g, a, ones = module.gradient_factors(activation, hooked)
                            self._update_from_factors(g, a, ones)

                self._run_beside(hooked.device, (activation, hooked), update)
            else:
                self._update_from_gradient(module.compute_per_sample_gradient(activation, output_gradient.detach()))

        @torch.no_grad()
        def shared_backward_hook(output_gradient: torch.Tensor) -> None:
            activation = self._take_activation()
            self.cached_hooks.pop().remove()
            psg = module.compute_per_sample_gradient(activation, output_gradient.detach())
            if self.cached_per_sample_gradient is None:
                self.cached_per_sample_gradient = torch.zeros_like(psg)
            self.cached_per_sample_gradient.add_(psg)

        self.registered_hooks.append(module.register_forward_hook(forward_hook))

    @torch.no_grad()
    def finalize_iteration(self) -> None:
        if self.module.factor_args.has_shared_parameters and self.cached_per_sample_gradient is not None:
            self._update_from_gradient(self.cached_per_sample_gradient)
        self.clear_all_cache()

    def exist(self) -> bool:
        return all(self.module.storage[name] is not None for name in LAMBDA_FACTOR_NAMES)

    def synchronize(self, num_processes: int) -> None:
        del num_processes
        if dist.is_initialized() and self.exist():
            storage = self.module.storage
            count = storage[NUM_LAMBDA_PROCESSED].to(storage[LAMBDA_MATRIX_NAME].device)
            _all_reduce_sum([storage[LAMBDA_MATRIX_NAME], count])
            storage[NUM_LAMBDA_PROCESSED] = count.cpu()

    def release_memory(self) -> None:
        self.clear_all_cache()
        self._bf16_eigenvectors = None
        self._conv_dense_eigenvectors = None
        for name in LAMBDA_FACTOR_NAMES:
            self.module.storage[name] = None
