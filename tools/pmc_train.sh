set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3/train_pmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
A="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY"
timeout -k 10 300 rocprofv3 --pmc $A --output-format csv -d $OUT -o a -- python3 $R/bench.py --train --batch 16 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/a.log 2>&1 || echo failed
for n in FETCH_SIZE WRITE_SIZE; do timeout -k 10 300 rocprofv3 --pmc $n --output-format csv -d $OUT -o $n -- python3 $R/bench.py --train --batch 16 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/$n.log 2>&1 || echo failed; done
python3 - <<PY
import csv, collections, glob
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for f in glob.glob('$OUT/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:40]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
        if r['Counter_Name']=='GRBM_GUI_ACTIVE': cnt[k]+=1
for k,v in sorted(agg.items(), key=lambda kv:-kv[1].get('GRBM_GUI_ACTIVE',0))[:14]:
    g=v.get('GRBM_GUI_ACTIVE',1)/8  # kernel cycles summed over launches
    simd=g*1024
    print(f"{k:40s} n={cnt[k]:4d} cyc={g:.3g} mfma_busy={v.get('SQ_VALU_MFMA_BUSY_CYCLES',0)*4/simd:.2f} valu_busy={v.get('SQ_ACTIVE_INST_VALU',0)*4/simd:.2f} fetchMB/launch={v.get('FETCH_SIZE',0)/max(cnt[k],1)/1024:.0f} writeMB/launch={v.get('WRITE_SIZE',0)/max(cnt[k],1)/1024:.0f}")
PY
