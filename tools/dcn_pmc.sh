# counters of the deformable-conv forward kernels at the (8,24,256,256) shapes:  tools/dcn_pmc.sh [tile-variant]   (on the GPU box)
out=$GRAFT_REPO_ROOT/gpurun_out/dcn_pmc; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
export CFEN_TUNE=dcn.tile=${1:-1}
i=0
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_WAVES SQ_ACTIVE_INST_VMEM" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out/p$i -- python3 $GRAFT_REPO_ROOT/tools/bench_dcn.py --channels 24 --reps 2 > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/p$i.log; }
done
python3 $GRAFT_REPO_ROOT/tools/pmc_fold.py $(find $out -name "*counter_collection.csv") --match dcn > $out/fold_${1:-1}.txt 2>&1
find $out -name "*.csv" -delete; find $out -name "*.db" -delete
grep -v "dcn_prep" $out/fold_${1:-1}.txt
