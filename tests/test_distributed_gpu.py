"""Two ranks (sharing the one GPU of the test box, gloo transport) run the complete sharded pipeline --
strided factor-fit shards + bucketed all-reduce, round-robin eigendecomposition + broadcast, strided
query shards + all-gather/interleave/truncate, contiguous train chunks + score-block gather -- and
must reproduce the single-process factors and scores.  On the 8-GPU node the same
code path runs over RCCL (backend "nccl")."""

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

    _, cov = fit_covariance_matrices_with_loader(model, state, task, ResidentLoader(train, 8, shard(DistributedEvalSampler, N_TRAIN)), fargs)
    if world > 1:
        box = [cov]
        dist.broadcast_object_list(box, src=0)
        cov = box[0]
    eig = perform_eigendecomposition(cov, model, state, fargs)
    _, lam = fit_lambda_matrices_with_loader(model, state, task, ResidentLoader(train, 8, shard(DistributedEvalSampler, N_TRAIN)), fargs, eig)
    if world > 1:
        box = [lam]
        dist.broadcast_object_list(box, src=0)
        lam = box[0]
    q_idx = list(DistributedSampler(range(N_QUERY), world, rank, shuffle=False, drop_last=False)) if world > 1 else None
    scores = compute_pairwise_scores_with_loaders({**eig, **lam}, model, state, task, ResidentLoader(query, 2, q_idx), 2,
                                                  ResidentLoader(train, 10, shard(DistributedSamplerWithStack, N_TRAIN)),
                                                  sargs, fargs, None)
    if rank == 0:
        torch.save({"cov": cov, "lam": lam, "scores": scores["all_modules"]}, out_path)


def _worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), KF_DIST_BACKEND="gloo")
    import torch.distributed as dist

    try:
        _pipeline(world, rank, out_path)
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def test_two_rank_pipeline_matches_single_process(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    single, double = str(tmp_path / "w1.pt"), str(tmp_path / "w2.pt")
    mp.spawn(_worker, args=(1, port, single), nprocs=1, join=True)
    mp.spawn(_worker, args=(2, port + 1, double), nprocs=2, join=True)
    one, two = torch.load(single), torch.load(double)
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
