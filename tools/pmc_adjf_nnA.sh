# PMC passes of the fused reverse step with the gridded law (A field read in every stage): tools/pmc_adjf_nnA.sh
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/pmcA; rm -rf $O; mkdir -p $O
F64="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_WAVES"
BUSY="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"
run() { timeout 300 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $O/$1 -- python $R/tools/run_kernel.py $3 $4 $5 6 $6 > $O/$1.log 2>&1; }
for law in nnA const; do
  python $R/tools/run_kernel.py adj_fused_step 64 1024 10 $law
  run adjf_${law}_f64 "$F64" adj_fused_step 64 1024 $law
  run adjf_${law}_busy "$BUSY" adj_fused_step 64 1024 $law
  run adjf_${law}_fetch FETCH_SIZE adj_fused_step 64 1024 $law
  run adjf_${law}_write WRITE_SIZE adj_fused_step 64 1024 $law
done
python - <<P
import csv, glob, os
O="$O"
for tag in sorted(os.listdir(O)):
    if not os.path.isdir(os.path.join(O, tag)): continue
    for f in glob.glob(os.path.join(O, tag, "**", "*counter_collection.csv"), recursive=True):
        acc = {}
        n = {}
        for r in csv.DictReader(open(f)):
            if "k_adj_fused_strip" not in r["Kernel_Name"]: continue
            acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            n[r["Counter_Name"]] = n.get(r["Counter_Name"], 0) + 1
        print(tag, {k: acc[k] / n[k] for k in acc}, "launches", max(n.values()) if n else 0)
P
