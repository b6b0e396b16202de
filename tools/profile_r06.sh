# round-6 evidence set -> gpurun_out/r06/: the default bench command plain and under rocprofv3 --kernel-trace --stats, the headline loop
# alone under rocprofv3 (per-workload kernel averages), the PMC passes on the bench workload (tools/pmc_r06.sh)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out/r06
(time python bench.py) > gpurun_out/r06/bench.json 2> gpurun_out/r06/bench.err
grep "\[bench\]" gpurun_out/r06/bench.err | cut -c1-400
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_head; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_head -- python $R/bench.py --headline-only > $R/gpurun_out/r06/bench_headline_under_rocprof.json 2> /tmp/prof_head.err
cp $(find /tmp/prof_head -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r06/bench_headline_kernel_stats.csv
rm -rf /tmp/prof_def; timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_def -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/r06/bench_under_rocprof.json 2> /tmp/prof_def.err
cp $(find /tmp/prof_def -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r06/bench_kernel_stats.csv
cd $R; head -4 gpurun_out/r06/bench_headline_kernel_stats.csv | cut -c1-220
bash tools/pmc_r06.sh 64 > gpurun_out/r06/pmc_r06.log 2>&1; cp gpurun_out/pmc_roofline.json gpurun_out/r06/pmc_roofline.json; head -c 1500 gpurun_out/r06/pmc_roofline.json
