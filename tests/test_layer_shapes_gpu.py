"""GPU parity at the REAL layer shapes of the BERT-base and GPT-2-small configs (C3, C4):
one tracked ``nn.Linear`` with bias on ``[b, T, d]`` activations -- (O, I') = (768, 769), (3072, 769), (768, 3073) at
T = 128 with random-length padding masks (reference mask semantics: ``kronfluence/module/linear.py:30-54``), all four
GPT-2-small shapes -- (2304, 769), (768, 769), (3072, 769), (768, 3073) -- at T = 512, plus the Llama-3-8B MLP projections at 1/8 width (no bias, T = 512) -- against the CPU oracle run in
fp64 on the same seeded inputs.

Every stage is compared on its own, so an error cannot hide behind (or be blamed on) an earlier stage:
  covariance   product stage vs oracle                                   rel_F <= 2e-5, counters exact
  Lambda       product stage fed the ORACLE's eigenvectors vs oracle     rel_F <= 2e-4 (fp32) / 5e-2 (bf16 lambda_dtype)
  scores       product stage fed the ORACLE's eigenvectors and Lambda    rel_F <= 1e-4 (fp32, heuristic damping)
               (query preconditioning + train pass)                      rel_F <= 4e-2 (bf16 autocast model, bf16 gradients / P / scores)
"""

import pytest
import torch
import torch.nn.functional as F
from torch import nn

from oracle import ekfac_ref as ref

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp(min=1e-300))


class SeqLayer(nn.Module):
    def __init__(self, i: int, o: int, bias: bool = True) -> None:
        super().__init__()
        self.lin = nn.Linear(i, o, bias=bias)

    def forward(self, x):
        return self.lin(torch.tanh(x))


def seq_loss(model, batch):
    x, mask, labels = batch
    logits = model(x.to(next(model.parameters()).dtype))
    return F.cross_entropy(logits.reshape(-1, logits.shape[-1]).float() if logits.dtype != torch.float64 else
                           logits.reshape(-1, logits.shape[-1]), labels.reshape(-1), reduction="sum", ignore_index=-100)


def seq_measure(model, batch):
    x, mask, labels = batch
    logits = model(x.to(next(model.parameters()).dtype))
    logits = logits if logits.dtype == torch.float64 else logits.float()
    picked = logits.gather(-1, labels.clamp(min=0)[..., None])[..., 0]
    return ((picked - 0.5 * torch.logsumexp(logits, dim=-1)) * mask.to(logits.dtype)).sum()


def make_data(n, t, i, o, seed):
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(n, t, i, generator=gen)
    lengths = torch.randint(max(2, t // 16), t + 1, (n,), generator=gen)
    mask = (torch.arange(t)[None, :] < lengths[:, None]).to(torch.int64)
    labels = torch.randint(0, o, (n, t), generator=gen)
    labels = torch.where(mask.bool(), labels, torch.full_like(labels, -100))
    return (x, mask, labels)


def chunks(data, size):
    return [tuple(t[k:k + size] for t in data) for k in range(0, data[0].shape[0], size)]


SHAPES = [  # (O, I, T, n_train, bias): rows = n_train * T real tokens exceed I' so the activation covariance has full rank
    pytest.param(768, 768, 128, 16, True, id="bert-768x769"),
    pytest.param(3072, 768, 128, 16, True, id="bert-3072x769"),
    pytest.param(768, 3072, 128, 48, True, id="bert-768x3073"),
    pytest.param(2304, 768, 512, 4, True, id="gpt2-2304x769-T512"),
    pytest.param(768, 768, 512, 4, True, id="gpt2-768x769-T512"),
    pytest.param(3072, 768, 512, 4, True, id="gpt2-3072x769-T512"),
    pytest.param(768, 3072, 512, 10, True, id="gpt2-768x3073-T512"),
    # Llama-3-8B MLP projections (14336 x 4096 / 4096 x 14336, no bias, T = 512) at 1/8 width: the C5 parity slice
    pytest.param(1792, 512, 512, 3, False, id="llama-up-1/8-width"),
    pytest.param(512, 1792, 512, 5, False, id="llama-down-1/8-width"),
]


@pytest.mark.parametrize("o,i,t,n_train,bias", SHAPES)
def test_layer_shape_stages_match_oracle(o, i, t, n_train, bias):
    from kronfluence_amd import FactorArguments, ScoreArguments, Task, prepare_model
    from kronfluence_amd.factor.covariance import fit_covariance_matrices_with_loader
    from kronfluence_amd.factor.eigen import fit_lambda_matrices_with_loader
    from kronfluence_amd.score.pairwise import compute_pairwise_scores_with_loaders
    from kronfluence_amd.utils.dataset import ResidentLoader
    from kronfluence_amd.utils.state import State

    class LayerTask(Task):
        def compute_train_loss(self, batch, model, sample=False):
            return seq_loss(model, tuple(batch))

        def compute_measurement(self, batch, model):
            return seq_measure(model, tuple(batch))

        def get_attention_mask(self, batch):
            return batch[1]

    torch.manual_seed(0)
    raw, twin = SeqLayer(i, o, bias), SeqLayer(i, o, bias)
    twin.load_state_dict(raw.state_dict())
    n_query, fb, tb = 3, max(1, n_train // 2), max(1, n_train // 2)
    train, query = make_data(n_train, t, i, o, 1), make_data(n_query, t, i, o, 2)

    # ---- oracle, fp64 -----------------------------------------------------------------------------------
    engine = ref.OracleEngine(twin.double(), dtypes=ref.OracleDtypes.all(torch.float64))
    ocov = engine.fit_covariance(chunks(train, fb), seq_loss, lambda batch: batch[1])
    oeig = engine.eigendecomposition(ocov)
    olam = engine.fit_lambda(chunks(train, fb), seq_loss, oeig)
    want = engine.pairwise_scores(chunks(query, n_query), chunks(train, tb), seq_measure, seq_loss, oeig, olam, None)

    # ---- product ------------------------------------------------------------------------------------------
    state = State()
    dev = state.device
    task = LayerTask()
    model = prepare_model(raw, task).to(dev)
    train_d, query_d = tuple(v.to(dev) for v in train), tuple(v.to(dev) for v in query)
    fargs = FactorArguments(use_empirical_fisher=True)
    _, cov = fit_covariance_matrices_with_loader(model, state, task, ResidentLoader(train_d, fb), fargs)
    tokens = int(train[1].sum())
    for name in ("activation_covariance", "gradient_covariance"):
        assert rel(cov[name]["lin"], ocov[name]["lin"]) <= 2e-5, (name, rel(cov[name]["lin"], ocov[name]["lin"]))
    assert int(cov["num_activation_covariance_processed"]["lin"]) == tokens == int(ocov["num_activation_covariance_processed"]["lin"])
    assert int(cov["num_gradient_covariance_processed"]["lin"]) == tokens
    assert cov["activation_covariance"]["lin"].shape == (i + int(bias), i + int(bias))

    eig32 = {k: {n: v.float() for n, v in d.items()} for k, d in oeig.items()}
    _, lam = fit_lambda_matrices_with_loader(model, state, task, ResidentLoader(train_d, fb), fargs, eig32)
    assert rel(lam["lambda_matrix"]["lin"], olam["lambda_matrix"]["lin"]) <= 2e-4, rel(lam["lambda_matrix"]["lin"], olam["lambda_matrix"]["lin"])
    assert int(lam["num_lambda_processed"]["lin"]) == n_train  # samples, not tokens (factor.py:203)

    lam32 = {k: {n: (v.float() if v.is_floating_point() else v) for n, v in d.items()} for k, d in olam.items()}
    factors = {**eig32, **lam32}
    got = compute_pairwise_scores_with_loaders(factors, model, state, task, ResidentLoader(query_d, n_query), n_query,
                                               ResidentLoader(train_d, tb), ScoreArguments(damping_factor=None), fargs,
                                               None)["all_modules"]
    assert got.shape == want.shape == (n_query, n_train)
    assert rel(got, want) <= 1e-4, rel(got, want)

    # ---- bf16 gradients / bf16 Lambda rotations / bf16 P (the BERT / GPT-2 configs' dtypes), same fp32 factors ------
    low = FactorArguments(use_empirical_fisher=True, amp_dtype=torch.bfloat16, per_sample_gradient_dtype=torch.bfloat16,
                          lambda_dtype=torch.bfloat16)
    _, lam16 = fit_lambda_matrices_with_loader(model, state, task, ResidentLoader(train_d, fb), low, eig32)
    assert rel(lam16["lambda_matrix"]["lin"], olam["lambda_matrix"]["lin"]) <= 5e-2, rel(lam16["lambda_matrix"]["lin"], olam["lambda_matrix"]["lin"])
    sargs = ScoreArguments(damping_factor=None, amp_dtype=torch.bfloat16, score_dtype=torch.bfloat16,
                           precondition_dtype=torch.bfloat16)
    got16 = compute_pairwise_scores_with_loaders(factors, model, state, task, ResidentLoader(query_d, n_query), n_query,
                                                 ResidentLoader(train_d, tb), sargs, low, None)["all_modules"]
    # bf16 end to end: autocast forward / backward (the hooked tensors themselves are bf16 roundings of the oracle's), bf16
    # preconditioner intermediates, bf16 P and per-sample gradients, scores returned in bf16 (score_dtype, as the reference)
    assert rel(got16, want) <= 4e-2, rel(got16, want)
