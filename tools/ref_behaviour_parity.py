"""Side-by-side behaviour check against the real reference -- a development tool for the build container only (it imports
/root/reference through tools/ref_diff_fuzz.py; nothing in tests/, bench.py or the product uses it).

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tools/ref_behaviour_parity.py

Part 1: invalid or premature calls (scores before factors, Lambda before the eigendecomposition, too many partitions, unknown
names, invalid arguments ...): does each side return or raise, and which exception class.  Part 2: multi-call flows (refit with
other arguments, overwrite, another dataset, ``load_from_factors_name``, partial partitions, partitioned self scores): outcome AND
the set of files each side leaves in its output directory.  Part 3: interchange -- this engine scores from factor directories the
reference wrote and the reference from directories this engine wrote (partitioned fits, three strategies, three fixtures), the
argument JSON files are compared key by key.  A line ends in ``<--`` where the two differ.  Return-value
differences of ``compute_*`` (the reference returns None and stores; this engine also returns what it stored) are expected.
"""
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_diff_fuzz as f  # noqa: E402

import torch  # noqa: E402,F401
from torch.utils import data  # noqa: E402

fx = f.fx


def setup_task(pkg, ours, d, kind="mlp", **task_kw):
    task=f.make_task(pkg,kind,**task_kw)
    model=pkg.prepare_model(fx.make_model(kind).double(),task)
    kw=dict(disable_tqdm=True,output_dir=d)
    if not ours: kw["cpu"]=True
    else:
        from kronfluence_amd.utils.state import State
        State._reset_state()
    an=pkg.Analyzer("e",model,task,**kw)
    train=data.TensorDataset(*fx.make_data(kind,20,seed=1)); query=data.TensorDataset(*fx.make_data(kind,4,seed=2))
    return an,train,query
def scenarios(pkg):
    FA,SA=pkg.FactorArguments,pkg.ScoreArguments
    return {
     "scores before factors": lambda an,t,q: an.compute_pairwise_scores("s","nofactors",q,t,per_device_query_batch_size=2,per_device_train_batch_size=4),
     "lambda before eigen": lambda an,t,q: an.fit_lambda_matrices("f",t,per_device_batch_size=4),
     "eigen before covariance": lambda an,t,q: an.perform_eigendecomposition("f"),
     "too many module partitions": lambda an,t,q: an.fit_covariance_matrices("f",t,per_device_batch_size=4,factor_args=FA(covariance_module_partitions=9)),
     "too many data partitions": lambda an,t,q: an.fit_covariance_matrices("f",t,per_device_batch_size=4,factor_args=FA(covariance_data_partitions=50)),
     "target partition out of range": lambda an,t,q: an.fit_covariance_matrices("f",t,per_device_batch_size=4,factor_args=FA(covariance_data_partitions=2),target_data_partitions=[5]),
     "target partitions without partitioning": lambda an,t,q: an.fit_covariance_matrices("f",t,per_device_batch_size=4,target_data_partitions=[0]),
     "unknown strategy": lambda an,t,q: an.fit_all_factors("f",t,per_device_batch_size=4,factor_args=FA(strategy="nope")),
     "negative damping": lambda an,t,q: SA(damping_factor=-1.0),
     "zero accumulation": lambda an,t,q: SA(query_gradient_accumulation_steps=0),
     "zero partitions": lambda an,t,q: FA(lambda_data_partitions=0),
     "load missing scores": lambda an,t,q: an.load_pairwise_scores("missing"),
     "load missing factors": lambda an,t,q: an.load_all_factors("missing"),
     "aggregate missing": lambda an,t,q: an.aggregate_pairwise_scores("missing"),
     "aggregate self missing": lambda an,t,q: an.aggregate_self_scores("missing"),
     "aggregate covariance missing": lambda an,t,q: an.aggregate_covariance_matrices("missing"),
     "aggregate lambda missing": lambda an,t,q: an.aggregate_lambda_matrices("missing"),
     "load missing covariance": lambda an,t,q: an.load_covariance_matrices("missing"),
     "load missing eigen": lambda an,t,q: an.load_eigendecomposition("missing"),
     "load missing lambda": lambda an,t,q: an.load_lambda_matrices("missing"),
     "load missing self scores": lambda an,t,q: an.load_self_scores("missing"),
     "load factor args missing": lambda an,t,q: an.load_factor_args("missing"),
     "load score args missing": lambda an,t,q: an.load_score_args("missing"),
     "self scores before factors": lambda an,t,q: an.compute_self_scores("s","nofactors",t,per_device_train_batch_size=4),
     "per-token on a model without a token axis": lambda an,t,q: (an.fit_all_factors("f",t,per_device_batch_size=4,factor_args=FA(strategy="identity")), an.compute_pairwise_scores("s","f",q,t,per_device_query_batch_size=2,per_device_train_batch_size=4,score_args=SA(compute_per_token_scores=True)))[1],
     "query batch larger than the query set": lambda an,t,q: (an.fit_all_factors("f",t,per_device_batch_size=4,factor_args=FA(strategy="identity")), an.compute_pairwise_scores("s","f",q,t,per_device_query_batch_size=64,per_device_train_batch_size=4))[1],
     "score data partitions beyond the train set": lambda an,t,q: (an.fit_all_factors("f",t,per_device_batch_size=4,factor_args=FA(strategy="identity")), an.compute_pairwise_scores("s","f",q,t,per_device_query_batch_size=2,per_device_train_batch_size=4,score_args=SA(data_partitions=50)))[1],
     "low rank zero": lambda an,t,q: SA(query_gradient_low_rank=0),
     "max examples zero": lambda an,t,q: FA(covariance_max_examples=0),
    }
def outcome(pkg, ours, name):
    with tempfile.TemporaryDirectory() as d:
        an,t,q=setup_task(pkg,ours,d)
        try:
            r=scenarios(pkg)[name](an,t,q)
            return f"returns {type(r).__name__}"
        except Exception as e:
            return f"raises {type(e).__name__}"

def part_errors():
    for name in scenarios(f.ref_pkg):
        a=outcome(f.ref_pkg,False,name); b=outcome(f.our_pkg,True,name)
        print(f"{name:42s} reference {a:36s} ours {b:36s} {'' if a==b else '<-- DIFFERENT'}")
    # tracked-module errors
    for label,kw in (("unknown tracked module", dict(modules=["nope"])),):
        outs=[]
        for pkg,ours in ((f.ref_pkg,False),(f.our_pkg,True)):
            try:
                task=f.make_task(pkg,"mlp",**kw); pkg.prepare_model(fx.make_model("mlp"),task); outs.append("returns")
            except Exception as e: outs.append(f"raises {type(e).__name__}")
        print(f"{label:42s} reference {outs[0]:36s} ours {outs[1]:36s} {'' if outs[0]==outs[1] else '<-- DIFFERENT'}")



def setup_flow(pkg, ours, d, kind="mlp"):
    task=f.make_task(pkg,kind)
    model=pkg.prepare_model(fx.make_model(kind).double(),task)
    kw=dict(disable_tqdm=True,output_dir=d)
    if not ours: kw["cpu"]=True
    else:
        from kronfluence_amd.utils.state import State
        State._reset_state()
    an=pkg.Analyzer("e",model,task,**kw)
    return an, data.TensorDataset(*fx.make_data(kind,20,seed=1)), data.TensorDataset(*fx.make_data(kind,4,seed=2))
def files(d):
    out=[]
    for root,_,fs in os.walk(d):
        for x in fs: out.append(os.path.relpath(os.path.join(root,x),d))
    return sorted(out)
def flows(pkg):
    FA,SA=pkg.FactorArguments,pkg.ScoreArguments
    def refit_other_args(an,t,q):
        an.fit_covariance_matrices("f",t,per_device_batch_size=4,factor_args=FA(use_empirical_fisher=True))
        an.fit_covariance_matrices("f",t,per_device_batch_size=4,factor_args=FA(use_empirical_fisher=True,covariance_max_examples=10))
    def refit_overwrite(an,t,q):
        an.fit_covariance_matrices("f",t,per_device_batch_size=4,factor_args=FA(use_empirical_fisher=True))
        an.fit_covariance_matrices("f",t,per_device_batch_size=4,factor_args=FA(use_empirical_fisher=True,covariance_max_examples=10),overwrite_output_dir=True)
    def other_dataset(an,t,q):
        an.fit_covariance_matrices("f",t,per_device_batch_size=4,factor_args=FA(use_empirical_fisher=True))
        an.fit_covariance_matrices("f",q,per_device_batch_size=4,factor_args=FA(use_empirical_fisher=True))
    def rescore_other_args(an,t,q):
        an.fit_all_factors("f",t,per_device_batch_size=4,factor_args=FA(strategy="identity"))
        an.compute_pairwise_scores("s","f",q,t,per_device_query_batch_size=2,per_device_train_batch_size=4)
        an.compute_pairwise_scores("s","f",q,t,per_device_query_batch_size=2,per_device_train_batch_size=4,score_args=SA(damping_factor=1.0))
    def load_from(an,t,q):
        an.fit_all_factors("f",t,per_device_batch_size=4,factor_args=FA(use_empirical_fisher=True))
        an.perform_eigendecomposition("g",factor_args=FA(use_empirical_fisher=True),load_from_factors_name="f")
        an.fit_lambda_matrices("g",t,per_device_batch_size=4,factor_args=FA(use_empirical_fisher=True))
    def load_from_missing(an,t,q):
        an.perform_eigendecomposition("g",factor_args=FA(use_empirical_fisher=True),load_from_factors_name="nope")
    def partial_targets(an,t,q):
        an.fit_all_factors("f",t,per_device_batch_size=4,factor_args=FA(strategy="identity"))
        an.compute_pairwise_scores("s","f",q,t,per_device_query_batch_size=2,per_device_train_batch_size=4,score_args=SA(data_partitions=2),target_data_partitions=[0])
    def self_scores_flow(an,t,q):
        an.fit_all_factors("f",t,per_device_batch_size=4,factor_args=FA(strategy="diagonal",use_empirical_fisher=True))
        an.compute_self_scores("s","f",t,per_device_train_batch_size=4,score_args=SA(data_partitions=2,module_partitions=2))
    return dict(refit_other_args=refit_other_args, refit_overwrite=refit_overwrite, other_dataset=other_dataset, rescore_other_args=rescore_other_args,
                load_from=load_from, load_from_missing=load_from_missing, partial_targets=partial_targets, self_scores_flow=self_scores_flow)

def part_flows():
    for name in flows(f.ref_pkg):
        res=[]
        for pkg,ours in ((f.ref_pkg,False),(f.our_pkg,True)):
            with tempfile.TemporaryDirectory() as d:
                an,t,q=setup_flow(pkg,ours,d)
                try:
                    flows(pkg)[name](an,t,q); out="ok"
                except Exception as e:
                    out=f"raises {type(e).__name__}"
                res.append((out, files(d)))
        same = res[0][0]==res[1][0] and res[0][1]==res[1][1]
        print(f"{name:22s} reference {res[0][0]:28s} ours {res[1][0]:28s} files {'same' if res[0][1]==res[1][1] else 'DIFFERENT'} {'' if same else '<--'}")
        if res[0][1]!=res[1][1]:
            print("    only reference:", sorted(set(res[0][1])-set(res[1][1])))
            print("    only ours     :", sorted(set(res[1][1])-set(res[0][1])))



def analyzer(pkg, ours, d, kind):
    task=f.make_task(pkg,kind)
    model=pkg.prepare_model(fx.make_model(kind).double(),task)
    kw=dict(disable_tqdm=True,output_dir=d)
    if not ours: kw["cpu"]=True
    else:
        from kronfluence_amd.utils.state import State
        State._reset_state()
    return pkg.Analyzer("x",model,task,**kw)

def part_interchange():
    import json
    import shutil
    for kind in ("mlp","conv","seq"):
      for strategy in ("ekfac","kfac","diagonal"):
        train=data.TensorDataset(*fx.make_data(kind,20,seed=1)); query=data.TensorDataset(*fx.make_data(kind,4,seed=2))
        f64=torch.float64
        def fa(pkg): return pkg.FactorArguments(strategy=strategy,use_empirical_fisher=True,covariance_data_partitions=2,lambda_module_partitions=2,activation_covariance_dtype=f64,gradient_covariance_dtype=f64,per_sample_gradient_dtype=f64,lambda_dtype=f64)
        def sa(pkg): return pkg.ScoreArguments(damping_factor=1e-3,per_sample_gradient_dtype=f64,precondition_dtype=f64,score_dtype=f64)
        with tempfile.TemporaryDirectory() as dr, tempfile.TemporaryDirectory() as do:
            ar=analyzer(f.ref_pkg,False,dr,kind); ar.fit_all_factors("f",train,per_device_batch_size=4,factor_args=fa(f.ref_pkg))
            ao=analyzer(f.our_pkg,True,do,kind); ao.fit_all_factors("f",train,per_device_batch_size=4,factor_args=fa(f.our_pkg))
            # JSON contents
            for name in sorted(os.listdir(os.path.join(dr,"x","factors_f"))):
                if name.endswith(".json"):
                    a=json.load(open(os.path.join(dr,"x","factors_f",name))); b=json.load(open(os.path.join(do,"x","factors_f",name)))
                    if a!=b:
                        diff={k:(a.get(k),b.get(k)) for k in set(a)|set(b) if a.get(k)!=b.get(k)}
                        print(f"  {kind} {strategy} {name}: JSON differs: {diff}")
            # reference scores from reference factors
            ar.compute_pairwise_scores("s","f",query,train,per_device_query_batch_size=2,per_device_train_batch_size=4,score_args=sa(f.ref_pkg))
            want=ar.load_pairwise_scores("s")["all_modules"].double()
            # cross: our engine on the reference's factor files, the reference on ours
            shutil.copytree(os.path.join(dr,"x","factors_f"), os.path.join(do,"x","factors_fromref"))
            shutil.copytree(os.path.join(do,"x","factors_f"), os.path.join(dr,"x","factors_fromours"))
            try:
                ao.compute_pairwise_scores("s2","fromref",query,train,per_device_query_batch_size=2,per_device_train_batch_size=4,score_args=sa(f.our_pkg))
                e1=float((ao.load_pairwise_scores("s2")["all_modules"].double()-want).norm()/want.norm())
            except Exception as ex: e1=f"{type(ex).__name__}: {str(ex)[:120]}"
            try:
                ar.compute_pairwise_scores("s3","fromours",query,train,per_device_query_batch_size=2,per_device_train_batch_size=4,score_args=sa(f.ref_pkg))
                e2=float((ar.load_pairwise_scores("s3")["all_modules"].double()-want).norm()/want.norm())
            except Exception as ex: e2=f"{type(ex).__name__}: {str(ex)[:120]}"
            # score file metadata + reading each other's score files
            try:
                so=f.our_pkg.Analyzer.load_file(os.path.join(dr,"x","scores_s","pairwise_scores.safetensors"))
                e3=float((so["all_modules"].double()-want).norm()/want.norm())
            except Exception as ex: e3=f"{type(ex).__name__}: {str(ex)[:120]}"
            print(f"{kind:4s} {strategy:8s} ours on the reference's factors: {e1} | the reference on ours: {e2} | our load_file on its scores: {e3}")



if __name__ == "__main__":
    f.cpu_engine.install(f._Patch())
    print("== part 1: invalid or premature calls")
    part_errors()
    print("== part 2: flows and the files they leave")
    part_flows()
    print("== part 3: each engine on the other's factor files (partitioned fits), argument JSON contents, score files")
    part_interchange()
