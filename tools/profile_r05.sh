# round-5 evidence set -> gpurun_out/r05_*: self-controlled reverse step over batch sizes, kernel stats of the small-batch and Y-law
# gradients, the U law through its table at every table resolution, the default bench command plain and under rocprofv3 --kernel-trace --stats
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
python tools/rsc_probe.py > gpurun_out/r05_rsc_probe.txt 2>&1
python tools/rsc_probe.py alpine:4 cap:512:8 cap:1024:8 gridded >> gpurun_out/r05_rsc_probe.txt 2>&1
bash tools/profile_rsc.sh alpine:4 r05a4 > gpurun_out/r05_alpine4_kernel_trace.txt 2>&1
bash tools/profile_y.sh 512 8 r05y8 ODINN_X=1 > /dev/null 2>&1; cp gpurun_out/ykstats_r05y8.txt gpurun_out/r05_ylaw_8x512_kstats.txt
bash tools/profile_y.sh 1024 64 r05y64 ODINN_X=1 > /dev/null 2>&1; cp gpurun_out/ykstats_r05y64.txt gpurun_out/r05_ylaw_64x1024_kstats.txt
bash tools/utab_levels.sh 512 8 > /dev/null 2>&1; cp gpurun_out/r05/utab_levels_512_8.txt gpurun_out/r05_utab_levels_8x512.txt
(time python bench.py) > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err
bash tools/profile_bench_default.sh > gpurun_out/r05_bench_default_profile.txt 2>&1
cp gpurun_out/bench_default_kernel_stats.csv gpurun_out/r05_bench_kernel_stats.csv
cp gpurun_out/bench_default_under_rocprof.json gpurun_out/r05_bench_under_rocprof.json
tail -3 gpurun_out/r05_rsc_probe.txt; grep "\[bench\]" gpurun_out/r05_bench.err | cut -c1-400; head -5 gpurun_out/r05_bench_kernel_stats.csv | cut -c1-200
