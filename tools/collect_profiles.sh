#!/bin/bash
# Collects the artefacts profiles/README.md describes, on the GPU box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh <tag>      -> gpurun_out/<tag>_*   (copy what should be judged into profiles/)
set -e -o pipefail
TAG=${1:-rXX}
OUT=$PWD/gpurun_out
mkdir -p $OUT
python3 bench.py > $OUT/${TAG}_bench_default.json
python3 bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 --dump-ops $OUT/${TAG}_bench_fp32x3_per_op_hip_events.json > $OUT/${TAG}_bench_fp32x3_line.json
python3 bench.py --train --precision fp16 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/${TAG}_bench_train_fp16.json
python3 bench.py --train --precision fp16 --arch A1 --size 512 --batch 2 --in-ch 13 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/${TAG}_bench_train_config5_bs2_fp16.json
python3 bench.py --arch A1 --batch 8 --no-cpu-baseline --steps 10 --warmup 3 > $OUT/${TAG}_bench_A1_256_bs8.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_ks -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/${TAG}_ks.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_ks16 -- python3 $GRAFT_REPO_ROOT/bench.py --precision fp16 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/${TAG}_ks16.log 2>&1
rocprofv3 -i $GRAFT_REPO_ROOT/tools/pmc_traffic.txt --output-format csv -d $OUT/${TAG}_pmc -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary --no-op-timing > $OUT/${TAG}_pmc.log 2>&1
rocprofv3 -i $GRAFT_REPO_ROOT/tools/pmc_mfma.txt --output-format csv -d $OUT/${TAG}_mfma -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary --no-op-timing > $OUT/${TAG}_mfma.log 2>&1
cd $GRAFT_REPO_ROOT
find $OUT/${TAG}_ks $OUT/${TAG}_ks16 -name "*kernel_stats.csv" | head
echo collected
