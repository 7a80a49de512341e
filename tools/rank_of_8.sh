cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r8; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $R/tools/wall.py 8 > $OUT/wall8.log 2>&1
cp $OUT/t/*/*kernel_stats.csv $OUT/kernel_stats_rank0of8.csv; cut -c1-60,200-400 $OUT/kernel_stats_rank0of8.csv | head -5
python3 - <<'PY'
import csv,os
f=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/r8/kernel_stats_rank0of8.csv'
tot=0
for r in csv.DictReader(open(f)):
    n=r['Name']; import re
    m=re.search(r'(k_\w+)(<[^>]*>)?',n); k=(m.group(1)+(m.group(2) or '')) if m else n[:40]
    print(f"{k[:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us total {float(r['TotalDurationNs'])/1e6:8.2f} ms")
    tot+=float(r['TotalDurationNs'])
print('sum of kernel durations', tot/1e6,'ms over 65 frames =', tot/1e6/65, 'ms/frame')
PY
cat $OUT/wall8.log | tail -1


