set -x
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out/r02x
mkdir -p $R
(timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "misaligned or score_gemm_long" 2>&1 | tail -5) > $R/pytest.log 2>&1
for prio in 0 1 2; do
(KF_SCORE_PRIO=$prio timeout 200 python tools/kernel_bench.py resnet9) > $R/kb_prio$prio.log 2>&1
done
(KF_SCORE_PRIO=1 timeout 300 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "score" 2>&1 | tail -3) > $R/pytest_prio1.log 2>&1
for z in 512 2048; do
(KF_COV_ZTARGET=$z timeout 200 python tools/cov_bench.py) > $R/cov_z$z.log 2>&1
done
ls -la $R
