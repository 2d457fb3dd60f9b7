R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3_ks_c2 -- python3 $R/bench.py --size 64 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-op-timing > $R/gpurun_out/r3_ks_c2.log 2>&1
cd $R
find gpurun_out/r3_ks_c2 -type f ! -name "*kernel_stats.csv" -delete
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r3_ks_c2/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:30]:
    print(r['Name'][:100], r['Calls'], round(float(r['TotalDurationNs'])/25/1e3,1),'us/step', round(float(r['AverageNs'])/1e3,1))
print(tot/25/1e6)
PY
