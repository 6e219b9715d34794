# usage: bash tools/_prof3.sh <tag> [dir]   (kernel stats of 10 forwards -> gpurun_out/r3/<tag>_kernel_stats.csv)
set -e
TAG=$1; DIR=${2:-.}
ROOT=$(pwd)
mkdir -p $ROOT/gpurun_out/r3
export TMPDIR=/tmp
cd $DIR
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o x -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-batch-split > $ROOT/gpurun_out/r3/${TAG}_bench.json 2>/tmp/prof_$TAG.err
cp $(find /tmp/prof_$TAG -name 'x_kernel_stats.csv' | head -1) $ROOT/gpurun_out/r3/${TAG}_kernel_stats.csv
