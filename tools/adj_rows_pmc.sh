# fused reverse step, 7 vs 4 rows per thread on 8 x 1024^2: launch time and HBM traffic (does a smaller resident set per XCD pay?)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/adjrows; rm -rf $O; mkdir -p $O
for rows in 7 4; do
  for i in 1 2; do ODINN_ADJ_ROWS=$rows python $R/tools/run_kernel.py adj_fused_step 8 1024 20 const 2>/dev/null | tail -1 | sed "s/^/rows=$rows /"; done
  for c in FETCH_SIZE WRITE_SIZE; do
    ODINN_ADJ_ROWS=$rows timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/r${rows}_$c -- python $R/tools/run_kernel.py adj_fused_step 8 1024 6 const > /dev/null 2>&1
    python $R/tools/pmc_one.py $O/r${rows}_$c k_adj_fused_strip $c $rows
  done
done
