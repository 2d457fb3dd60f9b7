#!/bin/bash
# Per-op same-box A/B: bench.py --dump-ops with the working tree's library and with another build (tools/build_lib_at.sh), twice,
# alternating; prints per kernel family the summed HIP-event time of both and every op that got slower.  This is how round 4 found that
# the asm P split lost 3.4 % inside the attention softmax and that the run-time K-slice mode cost the fp16 convs 2-6 % (DESIGN_HISTORY 10.7, 10.13).
#   gpurun -- "bash tools/per_op_ab.sh r3 [bench.py args, e.g. --arch A1 --batch 8 | --precision fp16]"
ALT=${1:-r3}; shift
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
export EOD_ABI_ANY=1
for L in tree $ALT tree $ALT; do
  if [ $L = tree ]; then unset EOD_LIBRARY; else export EOD_LIBRARY=$PWD/scratch/altlib/libeodiff_$ALT.so; fi
  python3 bench.py "$@" --no-cpu-baseline --no-secondary --steps 30 --warmup 5 --dump-ops gpurun_out/per_op_ab_$L.json > /dev/null 2>&1
done
python3 - $ALT <<'PY'
import json, sys
alt = sys.argv[1]
a = json.load(open('gpurun_out/per_op_ab_tree.json')); b = json.load(open(f'gpurun_out/per_op_ab_{alt}.json'))
print('ops', len(a), len(b), 'sum ms', round(sum(o['ms'] for o in a), 3), round(sum(o['ms'] for o in b), 3))
if len(a) == len(b):
    agg = {}
    for x, y in zip(a, b):
        s = agg.setdefault((x['kind'], x.get('kernel')), [0.0, 0.0, 0])
        s[0] += x['ms']; s[1] += y['ms']; s[2] += 1
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        print('  ', k, v[2], 'tree', round(v[0], 3), alt, round(v[1], 3), 'diff', round(v[0] - v[1], 3))
    for x, y in zip(a, b):
        if x['ms'] - y['ms'] > 0.004:
            print('   SLOWER', x['kind'], x.get('label'), round(x['ms'], 3), round(y['ms'], 3))
else:
    print('the two programs differ in their op lists: compare the totals only')
PY
