# U law through its table at the four table resolutions (ODINN_UTAB_LEVEL = 0 .. 3: 16 x 8 ... 128 x 64 bi-quintic patches) and with the
# automatic choice, under rocprofv3 --kernel-trace --stats: tools/utab_levels.sh [n G]  -> gpurun_out/r05/utab_levels.txt
R=$GRAFT_REPO_ROOT
N=${1:-512}; G=${2:-8}
O=$R/gpurun_out/r05; mkdir -p $O
OUT=$O/utab_levels_${N}_${G}.txt
: > $OUT
cd /tmp && export TMPDIR=/tmp
for lev in auto 0 1 2 3; do
  rm -rf /tmp/p_ut
  if [ $lev = auto ]; then e=""; else e="ODINN_UTAB_LEVEL=$lev"; fi
  env $e ODINN_LAW_TABLE_VERBOSE=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_ut -- python $R/tools/workflow_probe.py U $N $G scaled > /tmp/p_ut.log 2>&1
  { echo; echo "=== $e python tools/workflow_probe.py U $N $G scaled"; grep -E "solve ms|LossH|odinn utab" /tmp/p_ut.log | sort | uniq -c | cut -c1-160; python $R/tools/kstats.py /tmp/p_ut 12; } >> $OUT
done
cat $OUT
