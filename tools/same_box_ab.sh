#!/bin/bash
# Same-box A/B of the working tree against another build of the library (tools/build_lib_at.sh <commit> <name> first), interleaved on ONE
# GPU box:  gpurun -- "EOD_TREE=$(git rev-parse --short HEAD) bash tools/same_box_ab.sh r3"   -> gpurun_out/r04_same_box_ab.txt
# (boxes of the pool differ by about +-2 %: only numbers taken on the same box, alternating, compare)
ALT=${1:-r3}
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
export EOD_ABI_ANY=1
OUT=gpurun_out/r04_same_box_ab.txt
echo "# same box, interleaved: bench.py (A0 @ 256x256, batch 16, fp32x3, 30 steps, 5 warmup) with this tree's library and with the round-3 library (EOD_LIBRARY); tree ${EOD_TREE:-unstamped}" > $OUT
for R in 1 2 3; do
  for L in tree $ALT; do
    if [ $L = tree ]; then unset EOD_LIBRARY; else export EOD_LIBRARY=$PWD/scratch/altlib/libeodiff_$ALT.so; fi
    python3 bench.py --no-cpu-baseline --no-secondary --steps 30 --warmup 5 --no-op-timing 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('A0@256 bs16 fp32x3  library=$L  run $R  %.3f ms/step  %.2f steps/s' % (d['ms_per_step'], d['value']))" >> $OUT
  done
done
for R in 1 2; do
  for L in tree $ALT; do
    if [ $L = tree ]; then unset EOD_LIBRARY; else export EOD_LIBRARY=$PWD/scratch/altlib/libeodiff_$ALT.so; fi
    EOD_ATTN=$([ $L != tree ] && echo gemm || echo nat) python3 bench.py --size 64 --no-cpu-baseline --no-secondary --steps 200 --warmup 10 --no-op-timing 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('A0@64 bs16 fp32x3   library=$L  run $R  %.3f ms/step' % d['ms_per_step'])" >> $OUT
    python3 bench.py --arch A1 --batch 8 --no-cpu-baseline --no-secondary --steps 20 --warmup 5 --no-op-timing 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('A1@256 bs8 fp32x3   library=$L  run $R  %.3f ms/step' % d['ms_per_step'])" >> $OUT
    python3 bench.py --precision fp16 --no-cpu-baseline --no-secondary --steps 30 --warmup 5 --no-op-timing 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('A0@256 bs16 fp16    library=$L  run $R  %.3f ms/step' % d['ms_per_step'])" >> $OUT
    python3 bench.py --train --precision fp16 --no-cpu-baseline --no-secondary --steps 10 --warmup 3 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('A0@256 bs16 fp16 TRAINING step  library=$L  run $R  %.3f ms/step' % d['ms_per_step'])" >> $OUT
  done
done
cat $OUT
