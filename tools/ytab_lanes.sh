# lanes / node-array sets of the overlapped :Linear contraction (ODINN_INTERP_ASYNC=lanes, ODINN_INTERP_SETS=sets)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
N=${1:-512}; G=${2:-8}
for e in "ODINN_INTERP_ASYNC=0" "ODINN_INTERP_ASYNC=1" "ODINN_INTERP_ASYNC=1 ODINN_INTERP_SETS=3" "ODINN_INTERP_ASYNC=2" "ODINN_INTERP_ASYNC=3" "ODINN_LAW_TABLE=0 ODINN_INTERP_ASYNC=1" "ODINN_LAW_TABLE=0 ODINN_INTERP_ASYNC=2" ; do
  echo "== $e"; env $e timeout 900 python tools/workflow_probe.py Y $N $G 2>&1 | grep -E "solve ms|LossH"
done > gpurun_out/ytab_lanes.txt 2>&1
cat gpurun_out/ytab_lanes.txt
