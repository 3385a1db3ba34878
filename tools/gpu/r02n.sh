set -x
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out/r02n
mkdir -p $R
export KF_COV_TILE=128
cd /tmp && export TMPDIR=/tmp
(timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_cov -- python $GRAFT_REPO_ROOT/tools/cov_bench.py) > $R/cov_plain.log 2>&1
(KF_COV_PROBE=1 timeout 300 python $GRAFT_REPO_ROOT/tools/cov_bench.py) > $R/cov_probe1.log 2>&1
(KF_COV_PROBE=2 timeout 300 python $GRAFT_REPO_ROOT/tools/cov_bench.py) > $R/cov_probe2.log 2>&1
(KF_COV_ZTARGET=256 timeout 300 python $GRAFT_REPO_ROOT/tools/cov_bench.py) > $R/cov_z256.log 2>&1
(timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_kb -- python $GRAFT_REPO_ROOT/tools/kernel_bench.py resnet9) > $R/kb.log 2>&1
cd $R && find . -name "*kernel_stats.csv" | head; for f in $(find . -name "*kernel_stats.csv"); do echo "== $f"; head -12 $f | cut -c1-220; done > $R/stats_head.txt
find $R -name "*kernel_trace.csv" -delete; find $R -name "*agent_info.csv" -delete
ls -la $R
