cd $GRAFT_REPO_ROOT
for e in "ODINN_INTERP_ASYNC=1" "ODINN_INTERP_ASYNC=2" "ODINN_INTERP_ASYNC=3" "ODINN_INTERP_ASYNC=4"; do echo "== $e"; env $e python tools/workflow_probe.py Y 512 8 2>&1 | grep -E "LossH"; done
