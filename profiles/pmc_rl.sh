cd /tmp && export TMPDIR=/tmp
for PASS in "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VMEM SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((${i:-0}+1))
  timeout 300 rocprofv3 --pmc $PASS --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_rl/pass$i -o pmc -- python $GRAFT_REPO_ROOT/profiles/bench_rl.py 100 4000 50 > /dev/null 2>&1
done
cd $GRAFT_REPO_ROOT; python profiles/pmc_summarize.py gpurun_out/pmc_rl gpurun_out/pmc_rl.csv | grep -E "rl_front" 
find gpurun_out/pmc_rl -name "*.csv" -size +1M -delete
