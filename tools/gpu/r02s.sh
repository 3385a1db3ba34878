set -x
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out/r02s
mkdir -p $R
for lanes in 8 12 16 24; do
(timeout 300 python tools/eigh_bench.py multi 3073 24 $lanes) > $R/eigh_3073_l$lanes.log 2>&1
done
for lanes in 8 16 32; do
(timeout 300 python tools/eigh_bench.py multi 769 96 $lanes) > $R/eigh_769_l$lanes.log 2>&1
done
ls -la $R
