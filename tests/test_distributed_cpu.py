"""World-size-2 / 3 / 8 (gloo, CPU) tests of the N > 1 exchange steps of the hot path:
bucketed factor all-reduce (C1-C3), query-gradient all-gather + interleave (C4), score-block
gather (C5).  The arithmetic on the shards needs a GPU, so shard-local results are synthesised
here; what is tested is that the exchange reproduces the single-process result."""

import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import fixtures as fx


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank: int, world: int, port: int, fn_name: str, out_dir: str) -> None:
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        globals()[fn_name](rank, world, out_dir)
    finally:
        dist.destroy_process_group()


def _run(fn_name: str, tmp_path, world: int = 2) -> None:
    mp.spawn(_worker, args=(world, _free_port(), fn_name, str(tmp_path)), nprocs=world, join=True)


class _T:
    def compute_train_loss(self, batch, model, sample=False): ...
    def compute_measurement(self, batch, model): ...


def _model():
    from kronfluence_amd import Task, prepare_model

    class T(Task):
        def compute_train_loss(self, batch, model, sample=False):
            return model(batch[0]).sum()

        def compute_measurement(self, batch, model):
            return model(batch[0]).sum()

    return prepare_model(fx.make_model("mlp"), T())


# ---- C1-C3 -------------------------------------------------------------------------------------
def _factor_allreduce(rank, world, out_dir):
    from kronfluence_amd.module.tracked_module import TrackedModule
    from kronfluence_amd.module.utils import get_tracked_module_names, synchronize_factors
    from kronfluence_amd.utils.constants import COVARIANCE_FACTOR_NAMES, LAMBDA_FACTOR_NAMES

    model = _model()
    mods = [m for m in model.modules() if isinstance(m, TrackedModule)]
    gen = torch.Generator().manual_seed(100 + rank)
    for i, m in enumerate(mods):
        m.storage["activation_covariance"] = torch.randn(4 + i, 4 + i, generator=gen)
        m.storage["gradient_covariance"] = torch.randn(3 + i, 3 + i, generator=gen)
        m.storage["num_activation_covariance_processed"] = torch.tensor([10 * (rank + 1) + i])
        m.storage["num_gradient_covariance_processed"] = torch.tensor([7 * (rank + 1) + i])
        m.storage["lambda_matrix"] = torch.randn(3 + i, 4 + i, generator=gen)
        m.storage["num_lambda_processed"] = torch.tensor([5 + rank])
    seen = torch.tensor([11 + rank])
    names = get_tracked_module_names(model)
    # a 64-byte bucket cap forces several buckets AND the in-place path for the factors above the cap
    cap = int(os.environ.get("KF_TEST_BUCKET_BYTES", "0")) or None
    synchronize_factors(model, COVARIANCE_FACTOR_NAMES, names, torch.device("cpu"), extra=[seen], bucket_bytes=cap)
    synchronize_factors(model, LAMBDA_FACTOR_NAMES, names, torch.device("cpu"), bucket_bytes=cap)
    state = {f"{m.name}/{k}": m.storage[k] for m in mods for k in COVARIANCE_FACTOR_NAMES + LAMBDA_FACTOR_NAMES}
    state["seen"] = seen
    torch.save(state, os.path.join(out_dir, f"rank{rank}.pt"))


@pytest.mark.parametrize("cap", [0, 64])
def test_bucketed_factor_allreduce_matches_sum(tmp_path, cap, monkeypatch):
    monkeypatch.setenv("KF_TEST_BUCKET_BYTES", str(cap))
    _run("_factor_allreduce", tmp_path)
    got = [torch.load(os.path.join(tmp_path, f"rank{r}.pt")) for r in range(2)]
    # every rank holds the SUM (all-reduce, not reduce-to-0), and it equals the sum of the two locals
    for key in got[0]:
        assert torch.equal(got[0][key], got[1][key]), key
    want = {}
    for rank in range(2):
        gen = torch.Generator().manual_seed(100 + rank)
        for i, name in enumerate(["0", "2", "4"]):
            for key, shape in (("activation_covariance", (4 + i, 4 + i)), ("gradient_covariance", (3 + i, 3 + i))):
                want[f"{name}/{key}"] = want.get(f"{name}/{key}", 0) + torch.randn(*shape, generator=gen)
            want[f"{name}/lambda_matrix"] = want.get(f"{name}/lambda_matrix", 0) + torch.randn(3 + i, 4 + i, generator=gen)
    for key, tensor in want.items():
        assert torch.allclose(got[0][key], tensor, atol=1e-6), key
    assert int(got[0]["0/num_activation_covariance_processed"]) == 10 + 20
    assert int(got[0]["4/num_gradient_covariance_processed"]) == 7 + 14 + 4
    assert int(got[0]["2/num_lambda_processed"]) == 5 + 6
    assert int(got[0]["seen"]) == 11 + 12


# ---- C4 ----------------------------------------------------------------------------------------
def _query_allgather(rank, world, out_dir):
    from kronfluence_amd.module.tracked_module import ModuleMode, TrackedModule
    from torch.utils.data import DistributedSampler

    model = _model()
    m = [x for x in model.modules() if isinstance(x, TrackedModule)][0]
    m.current_mode = ModuleMode.PRECONDITION_GRADIENT
    n_query, per_rank = 5, 3
    idx = list(DistributedSampler(range(n_query), world, rank, shuffle=False, drop_last=False))[:per_rank]
    # the "preconditioned gradient" of query i is a [2,3] matrix filled with i
    m.storage["preconditioned_gradient"] = torch.stack([torch.full((2, 3), float(i)) for i in idx])
    m.synchronize(num_processes=world)
    m.truncate(keep_size=n_query % (per_rank * world) or per_rank * world)
    m.accumulate_iterations()
    held = m.storage["accumulated_preconditioned_gradient"]  # QueryBlocks: the per-batch blocks, never concatenated
    assert held.shape == (n_query, 2, 3) and len(held.blocks) == 1
    torch.save(held.dense(), os.path.join(out_dir, f"rank{rank}.pt"))


def test_query_allgather_restores_dataset_order(tmp_path):
    _run("_query_allgather", tmp_path)
    for rank in range(2):
        got = torch.load(os.path.join(tmp_path, f"rank{rank}.pt"))
        assert got.shape == (5, 2, 3)
        assert got[:, 0, 0].tolist() == [0.0, 1.0, 2.0, 3.0, 4.0]


def _query_allgather_async(rank, world, out_dir):
    """The path the backward hook takes: ``_store(..., from_hook=True)`` issues the all-gather asynchronously, ``synchronize``
    waits and interleaves; the exchange log counts one call and the gathered bytes."""
    from kronfluence_amd.module.tracked_module import ModuleMode, TrackedModule
    from kronfluence_amd.utils import comm
    from torch.utils.data import DistributedSampler

    model = _model()
    m = [x for x in model.modules() if isinstance(x, TrackedModule)][0]
    m.current_mode = ModuleMode.PRECONDITION_GRADIENT
    tracker = m._trackers[ModuleMode.PRECONDITION_GRADIENT]
    n_query, per_rank = 5, 3
    idx = list(DistributedSampler(range(n_query), world, rank, shuffle=False, drop_last=False))[:per_rank]
    comm.EXCHANGE_LOG = {}
    tracker._store(torch.stack([torch.full((2, 3), float(i)) for i in idx]), from_hook=True)
    assert tracker._pending is None   # opt-in: nothing is exchanged unless the pairwise query loop has asked for it
    m.async_query_gather = True
    tracker._store(torch.stack([torch.full((2, 3), float(i)) for i in idx]), from_hook=True)
    assert tracker._pending is not None
    m.synchronize(num_processes=world)
    assert tracker._pending is None
    log, comm.EXCHANGE_LOG = comm.summary(comm.EXCHANGE_LOG), None
    assert log["query_all_gather"]["calls"] == 1 and log["query_all_gather"]["bytes"] == world * per_rank * 6 * 4
    m.truncate(keep_size=n_query % (per_rank * world) or per_rank * world)
    m.accumulate_iterations()
    torch.save(m.storage["accumulated_preconditioned_gradient"].dense(), os.path.join(out_dir, f"rank{rank}.pt"))


def test_async_query_allgather_from_the_hook(tmp_path):
    _run("_query_allgather_async", tmp_path)
    for rank in range(2):
        got = torch.load(os.path.join(tmp_path, f"rank{rank}.pt"))
        assert got[:, 0, 0].tolist() == [0.0, 1.0, 2.0, 3.0, 4.0]


# ---- C5 ----------------------------------------------------------------------------------------
def _score_gather(rank, world, out_dir):
    from kronfluence_amd.score.dot_product import gather_score_blocks
    from kronfluence_amd.utils.dataset import DistributedSamplerWithStack
    from kronfluence_amd.utils.state import State

    State._reset_state()
    state = State(cpu=True)
    n_train, q = 7, 3
    idx = list(DistributedSamplerWithStack(range(n_train), world, rank))
    block = torch.tensor([[100.0 * qi + t for t in idx] for qi in range(q)])
    total = gather_score_blocks(block, state, n_train)
    torch.save(total, os.path.join(out_dir, f"rank{rank}.pt"))


def test_score_block_gather_concatenates_in_dataset_order(tmp_path):
    _run("_score_gather", tmp_path)
    got = torch.load(os.path.join(tmp_path, "rank0.pt"))
    want = torch.tensor([[100.0 * qi + t for t in range(7)] for qi in range(3)])
    assert torch.equal(got, want)


# ---- whole stages on 2 ranks (host logic; HIP leaf operators replaced by tests/cpu_engine.py) -------------------
def _pipeline_two_ranks(rank, world, out_dir):
    import cpu_engine
    from torch.utils import data

    from kronfluence_amd import Analyzer, FactorArguments, ScoreArguments
    from test_pipeline_gpu import make_task
    from kronfluence_amd import prepare_model

    cpu_engine.install_in_worker()
    kind = "seq"
    spec = fx.FIXTURES[kind]
    task = make_task(kind)
    model = prepare_model(fx.make_model(kind), task)
    analyzer = Analyzer("t", model, task, output_dir=out_dir, disable_tqdm=True)
    assert analyzer.state.num_processes == world and analyzer.state.process_index == rank
    train = data.TensorDataset(*fx.make_data(kind, spec.n_train - 1, seed=1))  # odd size: uneven shards, padding
    query = data.TensorDataset(*fx.make_data(kind, spec.n_query - 1, seed=2))
    analyzer.fit_all_factors("f", train, per_device_batch_size=7, factor_args=FactorArguments(use_empirical_fisher=True))
    common = dict(per_device_query_batch_size=2, per_device_train_batch_size=5)
    results = {}
    for name, kw in (("plain", {}), ("aggq", dict(aggregate_query_gradients=True)),
                     ("aggt", dict(aggregate_train_gradients=True)), ("tok", dict(compute_per_token_scores=True)),
                     ("parts", dict(data_partitions=2, module_partitions=2)),
                     ("lowrank", dict(query_gradient_low_rank=4, query_gradient_accumulation_steps=2))):
        out = analyzer.compute_pairwise_scores(name, "f", query, train, score_args=ScoreArguments(damping_factor=None, **kw),
                                               **common)
        if rank == 0:
            results[name] = out["all_modules"]
        else:
            assert out is None
    for name, kw in (("self", {}), ("selfm", dict(use_measurement_for_self_influence=True))):
        out = analyzer.compute_self_scores(name, "f", train, per_device_train_batch_size=5,
                                           score_args=ScoreArguments(damping_factor=None, **kw))
        if rank == 0:
            results[name] = out["all_modules"]
    if rank == 0:
        torch.save(results, os.path.join(out_dir, "two_ranks.pt"))


def test_whole_stages_on_two_ranks_match_single_process(tmp_path, cpu_engine):
    """Sharded factor fit + all-reduce, query all-gather / interleave / truncate, score-block gather, summed-gradient
    all-reduce and self-score gather reproduce the single-process results (uneven shard sizes)."""
    from torch.utils import data

    from kronfluence_amd import Analyzer, FactorArguments, ScoreArguments, prepare_model
    from test_pipeline_gpu import make_task

    _run("_pipeline_two_ranks", tmp_path / "two")
    got = torch.load(tmp_path / "two" / "two_ranks.pt")
    kind = "seq"
    spec = fx.FIXTURES[kind]
    task = make_task(kind)
    analyzer = Analyzer("t", prepare_model(fx.make_model(kind), task), task, output_dir=str(tmp_path / "one"), disable_tqdm=True)
    train = data.TensorDataset(*fx.make_data(kind, spec.n_train - 1, seed=1))
    query = data.TensorDataset(*fx.make_data(kind, spec.n_query - 1, seed=2))
    analyzer.fit_all_factors("f", train, per_device_batch_size=7, factor_args=FactorArguments(use_empirical_fisher=True))
    common = dict(per_device_query_batch_size=2, per_device_train_batch_size=5)

    def close(a, b, tol=2e-5):
        a, b = a.double(), b.double()
        return a.shape == b.shape and float((a - b).abs().max() / b.abs().max()) <= tol

    plain = analyzer.compute_pairwise_scores("plain", "f", query, train, score_args=ScoreArguments(damping_factor=None), **common)["all_modules"]
    assert close(got["plain"], plain) and close(got["parts"], plain)
    assert plain.shape == (spec.n_query - 1, spec.n_train - 1)
    for name, kw in (("aggq", dict(aggregate_query_gradients=True)), ("aggt", dict(aggregate_train_gradients=True)),
                     ("tok", dict(compute_per_token_scores=True)),
                     ("lowrank", dict(query_gradient_low_rank=4))):  # factor pairs gathered + interleaved across ranks
        want = analyzer.compute_pairwise_scores(name, "f", query, train, score_args=ScoreArguments(damping_factor=None, **kw),
                                                **common)["all_modules"]
        assert close(got[name], want, 1e-4), name
    for name, kw in (("self", {}), ("selfm", dict(use_measurement_for_self_influence=True))):
        want = analyzer.compute_self_scores(name, "f", train, per_device_train_batch_size=5,
                                            score_args=ScoreArguments(damping_factor=None, **kw))["all_modules"]
        assert close(got[name], want), name


# ---- odd world sizes: wrap-around padding of the train shards together with a truncated last query round --------------------
_MANY = dict(kind="mlp", n_train=45, n_query=11, q=2, b=5, fit_b=7)   # N % P != 0 and Q % (q P) != 0 for P = 3 and P = 8


def _pipeline_many_ranks(rank, world, out_dir):
    import cpu_engine
    from torch.utils import data

    from kronfluence_amd import Analyzer, FactorArguments, ScoreArguments, prepare_model
    from kronfluence_amd.utils import comm
    from test_pipeline_gpu import make_task

    cpu_engine.install_in_worker()
    c = _MANY
    task = make_task(c["kind"])
    model = prepare_model(fx.make_model(c["kind"]), task)
    analyzer = Analyzer("t", model, task, output_dir=out_dir, disable_tqdm=True)
    assert analyzer.state.num_processes == world and analyzer.state.process_index == rank
    train = data.TensorDataset(*fx.make_data(c["kind"], c["n_train"], seed=1))
    query = data.TensorDataset(*fx.make_data(c["kind"], c["n_query"], seed=2))
    comm.EXCHANGE_LOG = {}
    analyzer.fit_all_factors("f", train, per_device_batch_size=c["fit_b"], factor_args=FactorArguments(use_empirical_fisher=True))
    fit_log, comm.EXCHANGE_LOG = comm.summary(comm.EXCHANGE_LOG), {}
    out = analyzer.compute_pairwise_scores("s", "f", query, train, per_device_query_batch_size=c["q"], per_device_train_batch_size=c["b"],
                                           score_args=ScoreArguments(damping_factor=None))
    score_log, comm.EXCHANGE_LOG = comm.summary(comm.EXCHANGE_LOG), None
    low = analyzer.compute_pairwise_scores("low", "f", query, train, per_device_query_batch_size=c["q"], per_device_train_batch_size=c["b"],
                                           score_args=ScoreArguments(damping_factor=None, query_gradient_low_rank=3,
                                                                     query_gradient_accumulation_steps=2))
    own = analyzer.compute_self_scores("self", "f", train, per_device_train_batch_size=c["b"], score_args=ScoreArguments(damping_factor=None))
    if rank == 0:
        torch.save({"scores": out["all_modules"], "low": low["all_modules"], "self": own["all_modules"], "fit_log": fit_log,
                    "score_log": score_log}, os.path.join(out_dir, "many_ranks.pt"))
    else:
        assert out is None and low is None
        torch.save({"fit_log": fit_log, "score_log": score_log}, os.path.join(out_dir, f"log_rank{rank}.pt"))


@pytest.mark.parametrize("mode", ["gather", "replicate"])
@pytest.mark.parametrize("world", [1, 3, 8])
def test_whole_stages_on_odd_world_sizes(tmp_path, cpu_engine, world, mode, monkeypatch):
    """(P = 1: ONE rank with ``KF_DIST_FORCE=1`` -- every exchange still issued, each an identity; the form in which RCCL runs this
    path on the one-GPU test box, tests/test_distributed_gpu.py.)  P = 3 and P = 8 with N = 45 train samples (N % P != 0: the contiguous train shards are wrap-padded to ceil(N / P),
    utils/dataset.py:181-196, and the gathered score blocks cut back with ``cat[:, :N]``), Q = 11 queries at 2 per rank (Q % (q P) !=
    0: the strided query sampler pads with duplicates and the last round is truncated, score/pairwise.py:239-246), the 2 L = 6
    eigenproblems of the 3-layer fixture dealt over 8 ranks (two ranks own none), the factor fit strided without padding -- all
    against the single-process run; and the bytes every exchange moved against the section-8(e) volumes BY FORMULA.
    ``mode``: the query side of the pairwise stage (score/query_exchange.py) -- ``gather`` is the reference's strided query shard +
    per-layer all-gather, ``replicate`` has every rank precondition all queries itself: same scores, same train passes and score
    gathers, and NOT ONE query byte exchanged."""
    from torch.utils import data

    from kronfluence_amd import Analyzer, FactorArguments, ScoreArguments, prepare_model
    from kronfluence_amd.module.tracked_module import TrackedModule
    from test_pipeline_gpu import make_task

    c = _MANY
    (tmp_path / "many").mkdir()
    if world == 1:
        monkeypatch.setenv("KF_DIST_FORCE", "1")
    monkeypatch.setenv("KF_QUERY_EXCHANGE", mode)
    _run("_pipeline_many_ranks", tmp_path / "many", world=world)
    monkeypatch.delenv("KF_QUERY_EXCHANGE")
    got = torch.load(tmp_path / "many" / "many_ranks.pt")
    task = make_task(c["kind"])
    model = prepare_model(fx.make_model(c["kind"]), task)
    analyzer = Analyzer("t", model, task, output_dir=str(tmp_path / "one"), disable_tqdm=True)
    train = data.TensorDataset(*fx.make_data(c["kind"], c["n_train"], seed=1))
    query = data.TensorDataset(*fx.make_data(c["kind"], c["n_query"], seed=2))
    analyzer.fit_all_factors("f", train, per_device_batch_size=c["fit_b"], factor_args=FactorArguments(use_empirical_fisher=True))
    common = dict(per_device_query_batch_size=c["q"], per_device_train_batch_size=c["b"])
    want = analyzer.compute_pairwise_scores("s", "f", query, train, score_args=ScoreArguments(damping_factor=None), **common)["all_modules"]
    low = analyzer.compute_pairwise_scores("low", "f", query, train, score_args=ScoreArguments(damping_factor=None, query_gradient_low_rank=3),
                                           **common)["all_modules"]
    own = analyzer.compute_self_scores("self", "f", train, per_device_train_batch_size=c["b"], score_args=ScoreArguments(damping_factor=None))["all_modules"]

    def close(a, b, tol):
        a, b = a.double(), b.double()
        return a.shape == b.shape and float((a - b).abs().max() / b.abs().max()) <= tol

    assert want.shape == (c["n_query"], c["n_train"])
    assert close(got["scores"], want, 2e-5) and close(got["low"], low, 1e-4) and close(got["self"], own, 2e-5)

    # exchange volumes, rank 0 (every rank logs the same collectives): SURVEY.md section 8(e)
    shapes = [(m.original_module.weight.shape[0], m.original_module.weight.shape[1] + int(m.original_module.bias is not None))
              for m in model.modules() if isinstance(m, TrackedModule)]
    cov_floats = sum(ip * ip + o * o for o, ip in shapes)
    lam_floats = sum(o * ip for o, ip in shapes)
    layers = len(shapes)
    fit = got["fit_log"]
    # covariances: one fp32 bucket + one int64 bucket (2 counters per layer + the stage's sample counter); Lambda likewise
    assert fit["factor_all_reduce"]["calls"] == 4
    assert fit["factor_all_reduce"]["bytes"] == 4 * (cov_floats + lam_floats) + 8 * (2 * layers + 1) + 8 * (layers + 1)
    # 2 L eigenproblems, each broadcast from its owner: eigenvalues + eigenvectors in the factor dtype
    assert fit["eigen_broadcast"]["calls"] == 2 * layers
    assert fit["eigen_broadcast"]["bytes"] == 4 * sum(ip * ip + ip + o * o + o for o, ip in shapes)
    # queries: ceil(Q / (q P)) rounds, each an all-gather of P q gradients per layer
    rounds = -(-c["n_query"] // (c["q"] * world))
    score = got["score_log"]
    if mode == "gather":
        assert score["query_all_gather"]["calls"] == rounds * layers
        # (every rank holds ceil(Q / P) queries after the sampler's wrap-around padding; the last round may be a partial batch)
        assert score["query_all_gather"]["bytes"] == world * (-(-c["n_query"] // world)) * lam_floats * 4
    else:
        assert "query_all_gather" not in score   # replicated query side: zero calls, zero bytes
    # one train pass (and one gather) per query round; the rounds' blocks add up to [Q, ceil(N / P)] (the last round truncated)
    assert score["score_gather"]["calls"] == rounds
    assert score["score_gather"]["bytes"] == c["n_query"] * (-(-c["n_train"] // world)) * 4
    for rank in range(1, world):
        other = torch.load(tmp_path / "many" / f"log_rank{rank}.pt")
        assert other["fit_log"] == {k: {x: v[x] for x in ("calls", "bytes")} | {"seconds": other["fit_log"][k]["seconds"]} for k, v in fit.items()}
        assert {k: (v["calls"], v["bytes"]) for k, v in other["score_log"].items()} == {k: (v["calls"], v["bytes"]) for k, v in score.items()}


# ---- layers that share an input, on two ranks: one eigendecomposition per distinct covariance, dealt over the ranks ------------
class _Attn(torch.nn.Module):
    def __init__(self):
        super().__init__()
        lin = torch.nn.Linear
        self.q, self.k, self.v, self.o = lin(6, 5), lin(6, 5), lin(6, 5), lin(5, 3)

    def forward(self, x):
        return self.o(torch.tanh(self.q(x)) * torch.tanh(self.k(x)) + self.v(x))


def _attn_task():
    from kronfluence_amd import Task

    class T(Task):
        def compute_train_loss(self, batch, model, sample=False):
            return (model(batch[0]) - batch[1]).square().sum()

        def compute_measurement(self, batch, model):
            return self.compute_train_loss(batch, model)

    return T()


def _attn_data():
    from torch.utils import data

    gen = torch.Generator().manual_seed(1)
    train = data.TensorDataset(torch.randn(41, 6, generator=gen), torch.randn(41, 3, generator=gen))
    query = data.TensorDataset(torch.randn(5, 6, generator=gen), torch.randn(5, 3, generator=gen))
    return train, query


def _attn_run(out_dir, solved=None):
    import cpu_engine
    from kronfluence_amd import Analyzer, FactorArguments, ScoreArguments, ops, prepare_model

    if solved is not None:
        cpu_engine.install_in_worker()
        real = ops.eigh
        ops.eigh = lambda cov, *a, **k: (solved.append(tuple(cov.shape)), real(cov, *a, **k))[1]
    torch.manual_seed(0)
    task = _attn_task()
    analyzer = Analyzer("t", prepare_model(_Attn(), task), task, output_dir=out_dir, disable_tqdm=True)
    train, query = _attn_data()
    analyzer.fit_all_factors("f", train, per_device_batch_size=8, factor_args=FactorArguments(use_empirical_fisher=True))
    scores = analyzer.compute_pairwise_scores("s", "f", query, train, per_device_query_batch_size=2, per_device_train_batch_size=7,
                                              score_args=ScoreArguments(damping_factor=None))
    return analyzer, scores


def _attn_two_ranks(rank, world, out_dir):
    solved = []
    analyzer, scores = _attn_run(out_dir, solved)
    torch.save({"solved": solved, "scores": None if scores is None else scores["all_modules"],
                "eig": analyzer.load_eigendecomposition("f") if rank == 0 else None}, os.path.join(out_dir, f"attn_rank{rank}.pt"))


def test_shared_input_layers_on_two_ranks(tmp_path, cpu_engine):
    """q / k / v consume one tensor: their covariance increments are shared on every rank (tracker/factor.py), the all-reduced
    activation covariances are recognised as one matrix by every rank alike, the 6 distinct eigenproblems (of 8) are dealt 3 + 3 over
    the two ranks, the aliases take their owner's broadcast result -- and the scores equal the single-process run's."""
    (tmp_path / "two").mkdir()
    _run("_attn_two_ranks", tmp_path / "two", world=2)
    r0, r1 = torch.load(tmp_path / "two" / "attn_rank0.pt"), torch.load(tmp_path / "two" / "attn_rank1.pt")
    assert len(r0["solved"]) == 3 and len(r1["solved"]) == 3
    assert (r0["solved"] + r1["solved"]).count((7, 7)) == 1          # q = k = v: solved once, on one rank
    eig = r0["eig"]
    for name in ("k", "v"):
        assert torch.equal(eig["activation_eigenvectors"][name], eig["activation_eigenvectors"]["q"])
        assert torch.equal(eig["activation_eigenvalues"][name], eig["activation_eigenvalues"]["q"])
    assert r1["scores"] is None
    _, want = _attn_run(str(tmp_path / "one"))
    got, want = r0["scores"].double(), want["all_modules"].double()
    assert got.shape == want.shape == (5, 41)
    assert float((got - want).abs().max() / want.abs().max()) <= 2e-5


# ---- the reference's launch idiom: prepare_model -> apply_ddp -> Analyzer ---------------------------------------------------
def _worker_without_group(rank: int, world: int, port: int, out_dir: str) -> None:
    import cpu_engine
    from torch.nn.parallel import DistributedDataParallel
    from torch.utils import data

    from kronfluence_amd import Analyzer, FactorArguments, ScoreArguments, prepare_model
    from kronfluence_amd.utils.model import apply_ddp
    from test_pipeline_gpu import make_task

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    cpu_engine.install_in_worker()
    kind = "mlp"
    spec = fx.FIXTURES[kind]
    task = make_task(kind)
    model = apply_ddp(prepare_model(fx.make_model(kind), task), local_rank=rank, rank=rank, world_size=world)   # creates the group
    try:
        assert isinstance(model, DistributedDataParallel) and dist.get_world_size() == world
        analyzer = Analyzer("t", model, task, output_dir=out_dir, disable_tqdm=True)
        assert analyzer.state.num_processes == world and analyzer.state.process_index == rank
        assert analyzer.state.is_last_process == (rank == world - 1)
        train = data.TensorDataset(*fx.make_data(kind, spec.n_train - 1, seed=1))
        query = data.TensorDataset(*fx.make_data(kind, spec.n_query, seed=2))
        analyzer.fit_all_factors("f", train, per_device_batch_size=7, factor_args=FactorArguments(use_empirical_fisher=True))
        out = analyzer.compute_pairwise_scores("s", "f", query, train, per_device_query_batch_size=2, per_device_train_batch_size=5,
                                               score_args=ScoreArguments(damping_factor=None))
        if rank == 0:
            torch.save(out["all_modules"], os.path.join(out_dir, "ddp_scores.pt"))
    finally:
        dist.destroy_process_group()


def test_apply_ddp_then_analyzer_on_two_ranks(tmp_path, cpu_engine):
    """``utils.model.apply_ddp`` (reference utils/model.py:17-55) creates the process group and returns the replica container the
    reference's multi-GPU scripts hand to the Analyzer; factor keys and scores are those of the unwrapped single-process run."""
    from torch.utils import data

    from kronfluence_amd import Analyzer, FactorArguments, ScoreArguments, prepare_model
    from kronfluence_amd.utils.model import apply_fsdp
    from test_pipeline_gpu import make_task

    (tmp_path / "two").mkdir()
    mp.spawn(_worker_without_group, args=(2, _free_port(), str(tmp_path / "two")), nprocs=2, join=True)
    got = torch.load(tmp_path / "two" / "ddp_scores.pt")
    kind = "mlp"
    spec = fx.FIXTURES[kind]
    task = make_task(kind)
    analyzer = Analyzer("t", prepare_model(fx.make_model(kind), task), task, output_dir=str(tmp_path / "one"), disable_tqdm=True)
    train = data.TensorDataset(*fx.make_data(kind, spec.n_train - 1, seed=1))
    query = data.TensorDataset(*fx.make_data(kind, spec.n_query, seed=2))
    analyzer.fit_all_factors("f", train, per_device_batch_size=7, factor_args=FactorArguments(use_empirical_fisher=True))
    want = analyzer.compute_pairwise_scores("s", "f", query, train, per_device_query_batch_size=2, per_device_train_batch_size=5,
                                            score_args=ScoreArguments(damping_factor=None))["all_modules"]
    assert got.shape == want.shape and float((got.double() - want.double()).abs().max() / want.abs().max()) <= 2e-5
    one = set(os.listdir(analyzer.factors_output_dir("f")))
    two = set(os.listdir(tmp_path / "two" / "t" / "factors_f"))
    assert one == two   # same files; module names carry no "module." prefix
    with pytest.raises(NotImplementedError, match="apply_ddp"):
        apply_fsdp(None, 0, 0, 1)


# ---- the query-exchange plan: bytes over the interconnect against redundant flops (score/query_exchange.py) -----------------
_GPT2 = [(2304, 769), (768, 769), (3072, 769), (768, 3073)] * 12


def test_query_exchange_plan_prices_bytes_against_flops(monkeypatch):
    from kronfluence_amd.score import query_exchange as qx

    monkeypatch.delenv("KF_QUERY_EXCHANGE", raising=False)
    monkeypatch.delenv("KF_XGMI_GBPS", raising=False)
    rows = [512] * len(_GPT2)
    bf16 = dict(score_dtype=torch.bfloat16, precondition_dtype=torch.bfloat16)
    plan = qx.plan_query_exchange(_GPT2, rows, 1024, 8, backend="nccl", **bf16)
    d = sum(o * i for o, i in _GPT2)
    assert d == 85_017_600
    assert plan.inbound_bytes == pytest.approx(1024 * d * 2 * 7 / 8)              # 152 GB inbound per rank (170 MB / query)
    assert plan.gather_exchange_seconds == pytest.approx(plan.inbound_bytes / 350e9)   # ~0.44 s over xGMI -- seconds, not minutes
    assert 0.3 < plan.gather_exchange_seconds < 0.6
    assert plan.mode == "gather" and plan.replicate_seconds > 2 * plan.gather_seconds
    assert 1.8e-3 < plan.replicate_seconds / 1024 < 2.8e-3                        # 2.23 ms per GPT-2 query (measured r06, --phase-split)
    # the same job over a host transport (gloo): the exchange would take minutes -> every rank preconditions all queries itself
    slow = qx.plan_query_exchange(_GPT2, rows, 1024, 8, backend="gloo", **bf16)
    assert slow.mode == "replicate" and slow.gather_exchange_seconds > 60
    # forced modes are taken as they are; one rank never exchanges
    assert qx.plan_query_exchange(_GPT2, rows, 1024, 8, backend="nccl", mode="replicate", **bf16).mode == "replicate"
    assert qx.plan_query_exchange(_GPT2, rows, 1024, 1, backend="gloo", **bf16).mode == "gather"
    monkeypatch.setenv("KF_XGMI_GBPS", "10")
    assert qx.plan_query_exchange(_GPT2, rows, 1024, 8, backend="nccl", **bf16).mode == "replicate"
    monkeypatch.setenv("KF_QUERY_EXCHANGE", "bogus")
    with pytest.raises(ValueError):
        qx.requested_mode()
    # rank-64 factor pairs (C5): k (O + I') elements per layer and query
    llama = [(14336, 4096), (14336, 4096), (4096, 14336)]
    assert qx.held_elements_per_query(llama, 64) == 3 * 64 * (14336 + 4096)
    assert qx.held_elements_per_query([(10, 1025)], 64) == 10 * 1025             # min(O, I') <= k: kept dense


def test_a_sharded_loader_is_refused_in_replicated_mode(cpu_engine, monkeypatch):
    """The stage loop checks the loader against the mode it was marked with (a rank-sharded loader marked replicated would
    silently score a subset of the queries)."""
    from kronfluence_amd.score.query_exchange import is_replicated, mark_replicated, probe_rows, layer_shapes

    model = _model()
    assert layer_shapes(model) == [(16, 11), (16, 17), (1, 17)] or len(layer_shapes(model)) == 3
    x = fx.make_data("mlp", 4, seed=3)[0]
    assert probe_rows(model, lambda: model(x)) == [1, 1, 1]
    loader = [1, 2, 3]

    class L(list):
        pass
    loader = mark_replicated(L(loader))
    assert is_replicated(loader) and not is_replicated(L())
