# rocprofv3 kernel stats of the DEFAULT bench command (python bench.py), the command BENCH_rNN.json is produced with
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/profd; mkdir -p $R/gpurun_out/profd
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/profd -- python $R/bench.py > $R/gpurun_out/bench_default_under_rocprof.json 2> $R/gpurun_out/profd.err
find $R/gpurun_out/profd -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/bench_default_kernel_stats.csv \;
head -4 $R/gpurun_out/bench_default_kernel_stats.csv | cut -c1-220
python - <<PY
import json
d=json.loads(open("$R/gpurun_out/bench_default_under_rocprof.json").read().strip().splitlines()[-1])
print("json ms_per_launch", d["roofline"]["ms_per_launch"], "value", d["value"])
PY
