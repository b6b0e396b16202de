#!/bin/bash
# the fuzz tests against a build of the library whose host code carries libstdc++ assertions (-D_GLIBCXX_ASSERTIONS on odinn_hip.hip;
# odinn.jl_amd/csrc/libodinn_hip_chk.so from `make -C odinn.jl_amd/csrc chk`), glibc's fatal messages on stderr (CHK_MALLOC=1: MALLOC_CHECK_=3, MALLOC_PERTURB_): tools/chk_fuzz.sh a:b [pytest args]
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=gpurun_out/chk; mkdir -p $O
export ODINN_LIB=$R/odinn.jl_amd/csrc/libodinn_hip_chk.so
export LIBC_FATAL_STDERR_=1; [ -n "$CHK_MALLOC" ] && export MALLOC_CHECK_=3 MALLOC_PERTURB_=165
S=${1:-32300:32600}; shift
ODINN_FUZZ_BIG=1 ODINN_FUZZ_SEEDS=$S timeout 1400 python -X faulthandler -m pytest ${@:-tests/test_gpu_fuzz.py::test_random_batch_time_aggregated_terms_match_the_oracle} -m gpu -q --timeout 300 -p no:cacheprovider > $O/pytest_$S.txt 2>&1
grep -E "passed|failed" $O/pytest_$S.txt | tail -1
grep -n -i "assert\|corrupt\|invalid pointer\|free()\|malloc\|Aborted\|terminate\|crashed" $O/pytest_$S.txt | head -20 | cut -c1-250
