#!/bin/bash
# Build container side of a profiling session: turns gpurun_out/prof_$TAG/ (written by scripts/profile_session.sh on the
# GPU box) into profiles/$TAG_*.   bash scripts/collect_profile.sh r04
set -e
TAG=${1:-r04}
OUT=gpurun_out/prof_$TAG
python scripts/make_profile_summary.py $TAG $OUT/kernel_stats.csv $OUT/bench.json $OUT/pmc_fetch.csv $OUT/pmc_write.csv \
    $OUT/cls_bf16_kernel_stats.csv $OUT/cls_bf16.json --mfma $OUT/pmc_mfma.csv --overlapped $OUT/kernel_stats_overlapped.csv > /dev/null
tail -1 $OUT/bench.json > profiles/${TAG}_bench.json
for n in retina fcos cls_bf16 mae; do cp $OUT/classes_$n.json profiles/${TAG}_classes_$n.json; done
for n in fcos s2 cls_f32 mae; do cp $OUT/$n.json profiles/${TAG}_$n.json; done
# one overlapped step (between the last two optimizer launches) as a time-ordered table
python - "$OUT/kernel_trace_overlapped.csv" "profiles/${TAG}_step_timeline.csv" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
ad = [i for i, r in enumerate(rows) if 'adamw' in r['Kernel_Name']]
step = rows[ad[-2] + 1:ad[-1] + 1]
t0 = step[0]['s']
qs = {}
with open(sys.argv[2], 'w') as f:
    f.write('queue,start_us,duration_us,grid,workgroup,kernel\n')
    for r in step:
        q = qs.setdefault(r['Queue_Id'], len(qs) + 1)
        name = re.sub(r'\(.*', '', re.sub(r'^void ', '', r['Kernel_Name']).replace('(anonymous namespace)::', ''))
        f.write(f'{q},{(r["s"] - t0) / 1e3},{(r["e"] - r["s"]) / 1e3},{r["Grid_Size_X"]},{r["Workgroup_Size_X"]},"{name}"\n')
PY
echo "profiles/${TAG}_* written"
