set -x
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out/r02p
mkdir -p $R
for np in 1 2 3 5; do
(KF_EIGH_PASSES=$np timeout 300 python tools/eigh_bench.py 769 2304 3073) > $R/eigh_p$np.log 2>&1
(KF_EIGH_PASSES=$np timeout 300 python tools/eigh_bench.py multi 3073 16 8) > $R/eigh_multi_p$np.log 2>&1
done
(KF_EIGH_PASSES=3 timeout 300 python tools/eigh_bench.py multi 769 48 8) > $R/eigh_multi769_p3.log 2>&1
(KF_EIGH_PASSES=1 timeout 300 python tools/eigh_bench.py multi 769 48 8) > $R/eigh_multi769_p1.log 2>&1
ls -la $R
