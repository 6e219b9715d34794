set -e
ROOT=$(pwd); mkdir -p $ROOT/gpurun_out/r3; export TMPDIR=/tmp
python3 bench.py --train --batch 16 --steps 5 --warmup 2 --no-cpu-baseline > $ROOT/gpurun_out/r3/train_bench.json 2>/tmp/train.err || (tail -20 /tmp/train.err; exit 1)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_train -o x -- python3 $ROOT/bench.py --train --batch 16 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>/tmp/prof_train.err
cp $(find /tmp/prof_train -name 'x_kernel_stats.csv' | head -1) $ROOT/gpurun_out/r3/train_kernel_stats.csv
tail -1 $ROOT/gpurun_out/r3/train_bench.json | cut -c1-300
