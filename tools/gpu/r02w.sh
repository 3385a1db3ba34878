set -x
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out/r02w
mkdir -p $R
export TMPDIR=/tmp
BENCH="python $GRAFT_REPO_ROOT/bench.py --n-train 4000 --steps 1 --warmup 1 --no-extras --no-cpu-baseline --factor-reps 1"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_stats -- $BENCH > $R/bench_n4000_stats.json 2> $R/bench_n4000_stats.err)
find $R/prof_stats -name "*kernel_stats.csv" -exec cp {} $R/r02_resnet9_n4000_kernel_stats.csv \;
rm -rf $R/prof_stats
for pass in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $pass | cut -d' ' -f1)
  (cd /tmp && timeout 600 rocprofv3 --pmc $pass --output-format csv -d $R/pmc_$tag -- $BENCH > $R/pmc_$tag.json 2> $R/pmc_$tag.err)
done
python tools/pmc_summary.py resnet9 $R/r02_pmc_resnet9.json $R/pmc_FETCH_SIZE $R/pmc_WRITE_SIZE $R/pmc_SQ_VALU_MFMA_BUSY_CYCLES > $R/pmc_summary.log 2>&1
rm -rf $R/pmc_FETCH_SIZE $R/pmc_WRITE_SIZE $R/pmc_SQ_VALU_MFMA_BUSY_CYCLES
[ -s $R/r02_pmc_resnet9.json ] && cp $R/r02_pmc_resnet9.json profiles/r02_pmc_resnet9.json
(time timeout 1200 python bench.py --steps 5 --warmup 2) > $R/bench_default.json 2> $R/bench_default.err
(time timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8) > $R/pytest_gpu.log 2>&1
(time timeout 300 python -c "import __graft_entry__ as g; g.smoke()") > $R/smoke.log 2>&1
ls -la $R
