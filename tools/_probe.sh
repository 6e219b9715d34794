cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/probe -o p -- python3 $GRAFT_REPO_ROOT/tools/attn_probe.py > /dev/null 2>&1
grep -E "row_can|attn_core" $GRAFT_REPO_ROOT/gpurun_out/probe/p_kernel_stats.csv | cut -d, -f1,2,4
