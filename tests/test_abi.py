"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol that
include/kronfluence_hip.h declares, and the product refuses to compute without an MI355X."""

import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "kronfluence_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(kf_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    names = _declared_symbols()
    for required in ("kf_syrk_accum", "kf_syrk_rows_bf16", "kf_syrk_planes_bf16", "kf_conv2d_cov_accum", "kf_im2col", "kf_gemm",
                     "kf_eigh_f64", "kf_lambda_accum", "kf_inv_lambda", "kf_precondition", "kf_pairwise_score",
                     "kf_pairwise_score_conv2d", "kf_pairwise_score_rows", "kf_cast", "kf_abi_version"):
        assert required in names


def test_library_exports_every_declared_symbol():
    from kronfluence_amd import _native

    path = _native.library_path()
    assert os.path.exists(path), "run `python -c 'import __graft_entry__ as g; g.build()'` first"
    raw = ctypes.CDLL(path)
    for name in _declared_symbols():
        assert hasattr(raw, name), f"{name} declared in the header but not exported"
    assert set(_native.SIGNATURES) == set(_declared_symbols())
    assert _native.lib().kf_abi_version() == _native.ABI_VERSION
    assert _native.lib().kf_status_string(0) == b"ok"


def test_no_cpu_fallback():
    from kronfluence_amd import _native, ops

    if torch.cuda.is_available():
        pytest.skip("GPU box: covered by the gpu-marked tests")
    assert _native.lib().kf_device_count() <= 0
    with pytest.raises(_native.KfError):
        ops.eigh(torch.eye(4), 1.0)
    with pytest.raises(_native.KfError):
        ops.linear_activation_cov(torch.zeros(3, 3), torch.zeros(1, dtype=torch.int64), torch.randn(5, 2), None, True)
    with pytest.raises(_native.KfError):
        ops.pairwise_score(torch.zeros(2, 3), 0, torch.zeros(2, 4, 5), torch.zeros(3, 1, 4), torch.zeros(3, 1, 5), False)


def test_analyzer_refuses_cpu(tmp_path):
    import fixtures as fx
    from kronfluence_amd import Analyzer, Task, prepare_model

    class T(Task):
        def compute_train_loss(self, batch, model, sample=False):
            return model(batch[0]).sum()

        def compute_measurement(self, batch, model):
            return model(batch[0]).sum()

    model = prepare_model(fx.make_model("mlp"), T())
    with pytest.raises(RuntimeError):
        Analyzer("a", model, T(), cpu=True, output_dir=str(tmp_path))
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            Analyzer("a", model, T(), output_dir=str(tmp_path))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "kronfluence_amd")
    for base, _dirs, files in os.walk(pkg):
        for name in files:
            if name.endswith(".py"):
                text = open(os.path.join(base, name)).read()
                assert "import oracle" not in text and "from oracle" not in text, os.path.join(base, name)


def test_operands_are_made_contiguous_and_16_byte_aligned():
    """``ops._contig`` (every input operand passes through it): strided views are compacted, and a CONTIGUOUS view that starts
    off a 16-byte boundary is copied -- the vector loads / LDS-DMA requests of the kernels need aligned bases."""
    from kronfluence_amd import ops

    base = torch.arange(4 * 6 + 1, dtype=torch.float32)
    aligned = base[:24].view(4, 6)
    assert ops._contig(aligned) is aligned
    shifted = base[1:].view(4, 6)  # contiguous, 4 bytes into the allocation
    assert shifted.is_contiguous() and shifted.data_ptr() % 16 != 0
    fixed = ops._contig(shifted)
    assert fixed is not shifted and fixed.data_ptr() % 16 == 0 and torch.equal(fixed, shifted)
    strided = aligned.t()
    fixed = ops._contig(strided)
    assert fixed.is_contiguous() and fixed.data_ptr() % 16 == 0 and torch.equal(fixed, strided)
