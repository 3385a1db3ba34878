"""Prints the numbers of a bench.py JSON line that a human compares between runs (headline, stage times, rooflines of every
config in the line).  ``python tools/bench_digest.py <file with the JSON line>``"""
import json
import sys


def show(name, r, indent=""):
    if "error" in r:
        print(f"{indent}{name}: ERROR {r['error']}")
        return
    cfg = r.get("config", {})
    print(f"{indent}{name}: {r.get('value', 0):.4g} {r.get('unit', '')}  {r.get('ms_per_step', 0):.1f} ms/step  "
          f"[{cfg.get('n_train')} x {cfg.get('n_query')}]  peak {r.get('peak_hbm_gib')} GiB")
    fit = r.get("factor_fit")
    if fit:
        print(f"{indent}  factor_fit: " + ", ".join(f"{k} {v:.2f}s" for k, v in fit["seconds"].items())
              + f"  n_fit {fit['n_fit']}  eigh_paths {fit.get('eigh_paths')}")
    busy = r.get("device_busy")
    if busy:
        if "error" in busy:
            print(f"{indent}  device_busy: ERROR {busy['error']}")
        else:
            print(f"{indent}  device_busy: busy {busy['device_busy_frac']:.3f} (kf kernels {busy['kf_kernel_frac']:.3f}, model kernels "
                  f"{busy['model_kernel_frac']:.3f}) idle {busy['idle_frac']:.3f} of a {busy['wall_s']:.2f} s step over {busy['n_train']} "
                  f"train x {busy.get('n_query')} query samples; launches kf {busy['kf_kernel_launches']} model {busy['model_kernel_launches']}")
            for g in busy.get("largest_idle_after", [])[:4]:
                print(f"{indent}    idle {g['seconds']:.3f} s in {g['gaps']} gaps after {g['after']}")
    if r.get("parity"):
        print(f"{indent}  parity: {r['parity']}")
    for key in ("roofline", "roofline_cov", "roofline_cov_f32", "roofline_lambda", "roofline_lambda_update"):
        v = r.get(key)
        if v:
            extra = {k: (round(v[k], 3) if isinstance(v[k], (int, float)) else v[k])
                     for k in ("kernel_share_of_region", "model_share_of_region", "precondition_share_of_region",
                               "hbm_frac_of_8TBps", "mfma_util") if v.get(k) is not None}
            print(f"{indent}  {key}: {v['achieved']:.0f} {v['unit']} = {v['frac']:.3f} of peak, {v['launches']} launches x "
                  f"{v['avg_launch_ms']:.3f} ms, traffic {v.get('traffic')}  {extra}")


def main():
    line = [l for l in open(sys.argv[1]) if l.startswith("{")][-1]
    r = json.loads(line)
    show(r["config"]["workload"], r)
    cpu = r.get("cpu_baseline")
    if cpu:
        print(f"  cpu_baseline: {cpu['value']:.1f} {cpu['unit']} on {cpu['cores']} threads ({cpu['sample'][:90]}...)")
    for name, t in (r.get("targets") or {}).items():
        print(f"  target {name}: ratio {t['ratio']:.0f}x (>= {t['target_ratio']}), rel_F {t['scores_rel_F_vs_cpu_oracle']:.2e} "
              f"(<= {t['target_rel']}), {t['ms_per_step']:.1f} ms/step")
    for name, other in (r.get("other_configs") or {}).items():
        show(name, other, indent="  ")
    if r.get("exchanges"):
        print("  exchanges:", json.dumps(r["exchanges"])[:600])


if __name__ == "__main__":
    main()
