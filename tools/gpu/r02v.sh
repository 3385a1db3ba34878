set -x
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out/r02v
mkdir -p $R
export TMPDIR=/tmp
(timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "score or cov or syrk" 2>&1 | tail -6) > $R/pytest.log 2>&1
(timeout 300 python tools/cov_bench.py) > $R/cov_bench.log 2>&1
(time timeout 600 python bench.py --steps 3 --warmup 2 --no-extras --no-cpu-baseline) > $R/bench_resnet9.json 2> $R/bench_resnet9.err
KB="python $GRAFT_REPO_ROOT/tools/kernel_bench.py resnet9"
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $R/pmc_lds -- $KB > $R/pmc_lds.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $R/pmc_wait -- $KB > $R/pmc_wait.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $R/pmc_act -- $KB > $R/pmc_act.log 2>&1)
python tools/pmc_dump.py $R/pmc_lds $R/pmc_wait $R/pmc_act > $R/pmc_dump.txt 2>&1
du -sh $R/pmc_*; rm -rf $R/pmc_lds $R/pmc_wait $R/pmc_act
ls -la $R
