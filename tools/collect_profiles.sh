#!/bin/bash
# Collects, in ONE invocation on the GPU box (through gpurun, from the repo root), every artefact profiles/README.md describes for a round:
#   gpurun -- "EOD_TREE=$(git rev-parse --short HEAD) bash tools/collect_profiles.sh r04 [A|B]"  -> gpurun_out/r04_*
#   (then, in the build container: python tools/collect_profiles_summarise.py r04 copies / condenses what is judged into profiles/)
# EOD_TREE stamps the tree id into every header (the GPU box has no .git; expand it in the build container's shell).
# Every rocprofv3 command has the program itself after `--` (python3 ...), counters in passes of their own (no trace domains with --pmc).
set -e -o pipefail
TAG=${1:-rXX}
PART=${2:-all}   # A = bench lines + end-to-end calls + kernel stats, B = PMC passes + condense (one gpurun call each: 20-minute limit)
export EOD_TREE=${EOD_TREE:-unstamped}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
if [ "$PART" != "B" ]; then
echo "[1] bench lines"; date
python3 bench.py > $OUT/${TAG}_bench_default.json
python3 bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 --dump-ops $OUT/${TAG}_bench_fp32x3_per_op_hip_events.json > $OUT/${TAG}_bench_fp32x3_line.json
python3 bench.py --arch A1 --batch 8 --no-cpu-baseline --steps 10 --warmup 3 > $OUT/${TAG}_bench_A1_256_bs8.json
python3 bench.py --arch A1 --batch 8 --no-cpu-baseline --no-secondary --steps 10 --warmup 3 --dump-ops $OUT/${TAG}_bench_A1_256_bs8_fp32x3_per_op_hip_events.json > /dev/null
python3 bench.py --size 64 --no-cpu-baseline --steps 200 --warmup 10 --no-op-timing > $OUT/${TAG}_bench_A0_64_bs16.json
python3 bench.py --size 64 --no-cpu-baseline --no-secondary --steps 100 --warmup 10 --dump-ops $OUT/${TAG}_bench_A0_64_bs16_fp32x3_per_op_hip_events.json > /dev/null
python3 bench.py --train --precision fp16 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/${TAG}_bench_train_fp16.json
python3 bench.py --train --precision fp16 --arch A1 --size 512 --batch 2 --in-ch 13 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/${TAG}_bench_train_config5_bs2_fp16.json
echo "[2] end-to-end calls"; date
{ python3 tools/full_sampling.py --precision fp32x3; python3 tools/full_sampling.py --precision fp16; python3 tools/full_sampling.py --precision fp32x3 --size 64; } 2>/dev/null > $OUT/${TAG}_full_1000step_sampling.txt
{ python3 tools/full_ddim_repaint.py --precision fp32x3; python3 tools/full_ddim_repaint.py --precision fp16; } 2>/dev/null > $OUT/${TAG}_full_ddim250_repaint_config3.txt
echo "[3] rocprofv3 kernel stats (three precision modes, A1, training)"; date
cd /tmp && export TMPDIR=/tmp
for MODE in fp32x3 fp16 fp32; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_ks_$MODE -- python3 $ROOT/bench.py --precision $MODE --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/${TAG}_ks_$MODE.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_ks_A1 -- python3 $ROOT/bench.py --arch A1 --batch 8 --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/${TAG}_ks_A1.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_ks_train -- python3 $ROOT/bench.py --train --precision fp16 --steps 10 --warmup 3 --no-cpu-baseline > $OUT/${TAG}_ks_train.log 2>&1
fi
if [ "$PART" != "A" ]; then
cd /tmp && export TMPDIR=/tmp
echo "[4] PMC passes: HBM traffic (fp32x3, fp16), MFMA busy / clock, wave-state split, attention"; date
rocprofv3 -i $ROOT/tools/pmc_traffic.txt --output-format csv -d $OUT/${TAG}_pmc_traffic_fp32x3 -- python3 $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary --no-op-timing > $OUT/${TAG}_pmc_traffic_fp32x3.log 2>&1
rocprofv3 -i $ROOT/tools/pmc_traffic.txt --output-format csv -d $OUT/${TAG}_pmc_traffic_fp16 -- python3 $ROOT/bench.py --precision fp16 --steps 4 --warmup 2 --no-cpu-baseline --no-secondary --no-op-timing > $OUT/${TAG}_pmc_traffic_fp16.log 2>&1
rocprofv3 -i $ROOT/tools/pmc_mfma.txt --output-format csv -d $OUT/${TAG}_pmc_mfma -- python3 $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-secondary --no-op-timing > $OUT/${TAG}_pmc_mfma.log 2>&1
rocprofv3 -i $ROOT/tools/pmc_stall.txt --output-format csv -d $OUT/${TAG}_pmc_stall -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-op-timing > $OUT/${TAG}_pmc_stall.log 2>&1
rocprofv3 -i $ROOT/tools/pmc_attn.txt --output-format csv -d $OUT/${TAG}_pmc_attn -- python3 $ROOT/bench.py --arch A1 --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-op-timing > $OUT/${TAG}_pmc_attn.log 2>&1
echo "[5] condense"; date
cd $ROOT
python3 tools/collect_profiles_summarise.py $TAG --out $OUT/${TAG}_condensed > $OUT/${TAG}_summary.log 2>&1 || { tail -20 $OUT/${TAG}_summary.log; exit 1; }
# the raw counter directories are large: keep the condensed files only
rm -rf $OUT/${TAG}_pmc_traffic_fp32x3 $OUT/${TAG}_pmc_traffic_fp16 $OUT/${TAG}_pmc_mfma $OUT/${TAG}_pmc_stall $OUT/${TAG}_pmc_attn
fi
for D in $OUT/${TAG}_ks_fp32x3 $OUT/${TAG}_ks_fp16 $OUT/${TAG}_ks_fp32 $OUT/${TAG}_ks_A1 $OUT/${TAG}_ks_train; do [ -d $D ] && find $D -type f ! -name "*kernel_stats.csv" -delete; done
echo collected $PART; date
