# Round-3 profile set of one configuration (GPU box, through gpurun from the repo root):
#   bash tools/prof_r03.sh <tag> <bench args...>      e.g.  bash tools/prof_r03.sh cfg2 --repeats 4 --batch 32
# -> gpurun_out/r03/<tag>_bench.json, <tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats of the same command),
#    <tag>_pmc/{a,b,fetch,write}_counter_collection.csv (four separate --pmc passes, counters only), <tag>_pmc_mfma.md and
#    <tag>_traffic.md (tools/pmc_summary.py, tools/pmc_traffic.py).  rocprofv3 gets python3 directly behind `--`.
set -e
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r03
mkdir -p $OUT/${TAG}_pmc
ARGS="$@ --no-cpu-baseline --no-batch-split"
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py $ARGS --steps 20 --warmup 3 > $OUT/${TAG}_bench.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$TAG -o x -- python3 $R/bench.py $ARGS --steps 10 --warmup 2 > /dev/null 2>/tmp/p_$TAG.err
cp $(find /tmp/p_$TAG -name 'x_kernel_stats.csv' | head -1) $OUT/${TAG}_kernel_stats.csv
A="SQ_INSTS_VALU_MFMA_F16 SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_VALU_MFMA_COEXEC_CYCLES"
B="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
for n in a b fetch write; do
  case $n in a) set_="$A";; b) set_="$B";; fetch) set_="FETCH_SIZE";; write) set_="WRITE_SIZE";; esac
  timeout -k 10 300 rocprofv3 --pmc $set_ --output-format csv -d $OUT/${TAG}_pmc -o $n -- python3 $R/bench.py $ARGS --steps 2 --warmup 1 > $OUT/${TAG}_pmc/$n.log 2>&1 || echo "pass $n failed"
done
cd $R
python3 tools/pmc_summary.py $OUT/${TAG}_pmc $OUT/${TAG}_kernel_stats.csv > $OUT/${TAG}_pmc_mfma.md
python3 tools/pmc_traffic.py $OUT/${TAG}_pmc $OUT/${TAG}_kernel_stats.csv > $OUT/${TAG}_traffic.md
tail -1 $OUT/${TAG}_bench.json | cut -c1-160
