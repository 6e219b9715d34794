cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_VALU" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "MemUnitStalled MeanOccupancyPerCU" "TCC_TAG_STALL_sum TCC_BUBBLE_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_x -o s$i -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-batch-split > $R/gpurun_out/pmc_x_$i.log 2>&1 || echo "set $i failed"
done
ls $R/gpurun_out/pmc_x
