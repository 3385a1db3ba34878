"""Two ranks run the complete sharded pipeline -- strided factor-fit shards + bucketed all-reduce (every rank keeps the
sums in HBM: no pickled hand-off), round-robin eigendecomposition + tensor broadcasts, strided query shards +
all-gather/interleave/truncate, contiguous train chunks + score-block gather -- and must reproduce the single-process
factors and scores.  Once with both ranks sharing the one GPU of the test box over gloo, and -- when
at least two GPUs are visible -- one rank per GPU over RCCL (backend "nccl"), the configuration bench.py --gpus N runs."""

import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

import fixtures as fx

pytestmark = pytest.mark.gpu
N_TRAIN, N_QUERY = 45, 7  # deliberately not divisible by the world size


def _pipeline(world, rank, out_path):
    import torch.distributed as dist
    from torch.utils.data import DistributedSampler

    from kronfluence_amd import FactorArguments, ScoreArguments, Task, prepare_model
    from kronfluence_amd.factor.covariance import fit_covariance_matrices_with_loader
    from kronfluence_amd.factor.eigen import fit_lambda_matrices_with_loader, perform_eigendecomposition
    from kronfluence_amd.score.pairwise import compute_pairwise_scores_with_loaders
    from kronfluence_amd.utils.dataset import DistributedEvalSampler, DistributedSamplerWithStack, ResidentLoader
    from kronfluence_amd.utils.state import State

    kind = "seq"
    loss, measure, mask = fx.train_loss(kind), fx.measurement(kind), fx.attention_mask(kind)

    class T(Task):
        def compute_train_loss(self, batch, model, sample=False):
            return loss(model, tuple(batch))

        def compute_measurement(self, batch, model):
            return measure(model, tuple(batch))

        def get_attention_mask(self, batch):
            return mask(tuple(batch))

    State._reset_state()
    state = State()
    dev = state.device
    task = T()
    model = prepare_model(fx.make_model(kind), task).to(dev)
    train = tuple(t.to(dev) for t in fx.make_data(kind, N_TRAIN, seed=1))
    query = tuple(t.to(dev) for t in fx.make_data(kind, N_QUERY, seed=2))
    fargs, sargs = FactorArguments(use_empirical_fisher=True), ScoreArguments(damping_factor=None)

    def shard(sampler_cls, n):
        return list(sampler_cls(range(n), world, rank)) if world > 1 else None

    # every rank receives the all-reduced factors, device resident (what bench.py does for N > 1)
    _, cov = fit_covariance_matrices_with_loader(model, state, task, ResidentLoader(train, 8, shard(DistributedEvalSampler, N_TRAIN)), fargs,
                                                 all_ranks=True, cpu=False)
    assert all(t.is_cuda for t in cov["activation_covariance"].values())
    eig = perform_eigendecomposition(cov, model, state, fargs, cpu=False)
    _, lam = fit_lambda_matrices_with_loader(model, state, task, ResidentLoader(train, 8, shard(DistributedEvalSampler, N_TRAIN)), fargs, eig,
                                             all_ranks=True, cpu=False)
    q_idx = list(DistributedSampler(range(N_QUERY), world, rank, shuffle=False, drop_last=False)) if world > 1 else None
    scores = compute_pairwise_scores_with_loaders({**eig, **lam}, model, state, task, ResidentLoader(query, 2, q_idx), 2,
                                                  ResidentLoader(train, 10, shard(DistributedSamplerWithStack, N_TRAIN)),
                                                  sargs, fargs, None)
    cpu = lambda d: {k: {n: v.cpu() for n, v in m.items()} for k, m in d.items()}  # noqa: E731
    if rank == 0:
        torch.save({"cov": cpu(cov), "lam": cpu(lam), "scores": scores["all_modules"]}, out_path)
    elif world > 1:  # the other rank holds the same sums
        torch.save({"cov": cpu(cov), "lam": cpu(lam)}, out_path + f".rank{rank}")


def _worker(rank, world, port, out_path, backend="gloo"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank if backend == "nccl" else 0), KF_DIST_BACKEND=backend,
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist

    try:
        _pipeline(world, rank, out_path)
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_two_rank_pipeline_matches_single_process(tmp_path, backend):
    if backend == "nccl" and torch.cuda.device_count() < 2:
        pytest.skip("RCCL variant needs two visible GPUs (the 1-GPU test box runs the gloo variant)")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    single, double = str(tmp_path / "w1.pt"), str(tmp_path / "w2.pt")
    mp.spawn(_worker, args=(1, port, single, backend), nprocs=1, join=True)
    mp.spawn(_worker, args=(2, port + 1, double, backend), nprocs=2, join=True)
    one, two = torch.load(single), torch.load(double)
    other = torch.load(double + ".rank1")
    for name, per_module in two["cov"].items():  # rank 1 received the same all-reduced factors as rank 0
        for module, tensor in per_module.items():
            assert torch.equal(other["cov"][name][module], tensor), (name, module)
    assert two["scores"].shape == (N_QUERY, N_TRAIN)
    for name in ("activation_covariance", "gradient_covariance"):
        for module, want in one["cov"][name].items():
            assert rel(two["cov"][name][module], want) <= 1e-6, (name, module)
    for name in ("num_activation_covariance_processed", "num_gradient_covariance_processed"):
        for module, want in one["cov"][name].items():
            assert torch.equal(two["cov"][name][module], want)
    for module, want in one["lam"]["lambda_matrix"].items():
        assert rel(two["lam"]["lambda_matrix"][module], want) <= 1e-4, module
        assert torch.equal(two["lam"]["num_lambda_processed"][module], one["lam"]["num_lambda_processed"][module])
    assert rel(two["scores"], one["scores"]) <= 1e-4, rel(two["scores"], one["scores"])
