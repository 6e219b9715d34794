set -e
ROOT=$(pwd); mkdir -p $ROOT/gpurun_out/r3; export TMPDIR=/tmp
python3 bench.py --batch 1 --steps 50 --warmup 5 --no-cpu-baseline --no-batch-split > $ROOT/gpurun_out/r3/b1_bench.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b1 -o x -- python3 bench.py --batch 1 --steps 20 --warmup 2 --no-cpu-baseline --no-batch-split > /dev/null 2>/tmp/prof_b1.err
cp $(find /tmp/prof_b1 -name 'x_kernel_stats.csv' | head -1) $ROOT/gpurun_out/r3/b1_kernel_stats.csv
cp $(find /tmp/prof_b1 -name 'x_kernel_trace.csv' | head -1) $ROOT/gpurun_out/r3/b1_kernel_trace.csv
