# PMC passes behind bench.py's roofline block (profiles/r03/pmc_roofline.json): fp64 instruction counts and HBM traffic
# of the fused step kernel (gridded-NN and constant A, 8 x 1024^2) and HBM traffic of the HBM-bound kernels on
# 32 x 1024^2 (past the Infinity Cache).  Separate --pmc passes, kernel-trace only, as MI355X_MICROARCH.md prescribes.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/pmcr; rm -rf $O; mkdir -p $O
run() { # tag counters kernel G law
  timeout 300 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $O/$1 -- python $R/tools/run_kernel.py $3 $4 1024 6 $5 > $O/$1.log 2>&1
}
for law in nnA const; do
  run fused_${law}_f64 "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_WAVES" fused_step 8 $law
  run fused_${law}_busy "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" fused_step 8 $law
  run fused_${law}_fetch FETCH_SIZE fused_step 8 $law
  run fused_${law}_write WRITE_SIZE fused_step 8 $law
done
run dhdt32_nnA_fetch FETCH_SIZE dhdt 32 nnA
run dhdt32_nnA_write WRITE_SIZE dhdt 32 nnA
run stage32_fetch FETCH_SIZE rk_stage2 32 const
run stage32_write WRITE_SIZE rk_stage2 32 const
run vjpH32_fetch FETCH_SIZE vjp_H 32 const
run vjpH32_write WRITE_SIZE vjp_H 32 const
run vjpth32_fetch FETCH_SIZE vjp_theta 32 const
run vjpth32_write WRITE_SIZE vjp_theta 32 const
run adjf8_fetch FETCH_SIZE adj_fused_step 8 const
run adjf8_write WRITE_SIZE adj_fused_step 8 const
run adjf8_busy "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE" adj_fused_step 8 const
cd $R && python tools/pmc_roofline.py $O > $R/gpurun_out/pmc_roofline.json && cat $R/gpurun_out/pmc_roofline.json
