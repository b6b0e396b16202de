# scratch: the command of the last ad-hoc gpurun cycle (tools/gpu_cycle.sh, profile_r04.sh, ytab_cycle.sh are the kept ones)
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -n 6 2>&1 | tail -3
