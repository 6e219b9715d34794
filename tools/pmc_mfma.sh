# MFMA-utilisation counter passes for the forward bench (GPU box; run from the repo root through gpurun):
#   bash tools/pmc_mfma.sh            -> gpurun_out/pmc_mfma/{a,b}_counter_collection.csv, then tools/pmc_summary.py
# rocprofv3 gets the python program directly after `--` (no env / shell hop); counters only, no trace domains.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
A="SQ_INSTS_VALU_MFMA_F16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_VALU_MFMA_COEXEC_CYCLES"
B="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
i=0
for set in "$A" "$B"; do
  i=$((i+1)); n=$( [ $i = 1 ] && echo a || echo b )
  timeout -k 10 300 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_mfma -o $n -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-batch-split > $R/gpurun_out/pmc_mfma_$n.log 2>&1 || echo "pass $n failed"
done
find $R/gpurun_out/pmc_mfma -name "*counter_collection.csv" | head
