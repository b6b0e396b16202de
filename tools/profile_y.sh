# kernel stats of the Y-law workflow probe: bash tools/profile_y.sh <n> <G> <tag> [ENV=.. ...]
R=$GRAFT_REPO_ROOT; N=${1:-512}; G=${2:-8}; TAG=${3:-y}; shift 3
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_$TAG
env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$TAG -- python $R/tools/workflow_probe.py ${LAWK:-Y} $N $G ${EXTRA:-} > /tmp/p_$TAG.log 2>/dev/null
{ grep -v "^W20\|^E20\|rocprof" /tmp/p_$TAG.log | cut -c1-150; python $R/tools/kstats.py /tmp/p_$TAG 26; } > $R/gpurun_out/ykstats_$TAG.txt
cat $R/gpurun_out/ykstats_$TAG.txt
