# Same-box A/B of environment settings (temporary switches while an experiment lasts) or of whole library variants, on the GPU box through gpurun:
#   bash tools/kernel_ab.sh env  "A=1,B=2" "A=0,B=2" ...      each argument = comma-separated VAR=value pairs exported for that run
#   bash tools/kernel_ab.sh lib  base th32 base ...             each argument = variants/lib_<name>.so copied over rtfs-net_amd/librtfs_amd.so
# (variants/ is git-ignored but travels with the gpurun snapshot; put the base build there as lib_base.so and run it first AND last: the first
# run of a call is often slower.)  Every run prints ms per forward + sweep roofline of `bench.py --repeats 4 --batch 32`, then the per-launch
# averages of the kernels whose names contain one of $KERNELS (comma-separated, default dw1p) and the sum over the main chain, from a
# rocprofv3 --kernel-trace --stats pass of the same command.  Per-kernel numbers are only comparable within one call.
set -e
MODE=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
KERNELS=${KERNELS:-dw1p}
cd /tmp && export TMPDIR=/tmp
i=0
for v in "$@"; do i=$((i+1))
  if [ "$MODE" = lib ]; then cp $R/variants/lib_$v.so $R/rtfs-net_amd/librtfs_amd.so; else for kv in $(echo $v | tr "," " "); do export $kv; done; fi
  python3 $R/bench.py --repeats 4 --batch 32 --no-cpu-baseline --no-batch-split --steps 20 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['roofline']['frac'])"
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kab_$i -o x -- python3 $R/bench.py --repeats 4 --batch 32 --no-cpu-baseline --no-batch-split --steps 8 --warmup 2 > /dev/null 2>/tmp/kab.err
  python3 - <<PY
import csv, glob
keys = "$KERNELS".split(",")
tot = 0
for r in csv.DictReader(open(glob.glob('/tmp/kab_$i/**/x_kernel_stats.csv', recursive=True)[0])):
    n = r['Name']
    if any(k in n for k in keys): print('   ', n[:60].ljust(60), round(float(r['AverageNs']) / 1e3, 1))
    if not ('vp_block' in n or 'caf_video' in n): tot += int(r['TotalDurationNs']) / 10 / 1e3
print('    chain sum', round(tot, 1))
PY
done
if [ "$MODE" = lib ]; then cp $R/variants/lib_base.so $R/rtfs-net_amd/librtfs_amd.so; fi
