#!/bin/bash
# Kernel-trace timeline of the headline step: per-dispatch start/end -> gpurun_out/trace/kernel_trace.csv (the last steps)
set -e -o pipefail
OUT=gpurun_out/trace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timer > $OUT/kt.log 2>&1
f=$(find $OUT/kt -name "*kernel_trace.csv" | head -1)
python3 - "$f" $OUT <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# keep the last ~1700 dispatches (two steps) in a compact form
keep = rows[-1800:]
with open(sys.argv[2] + '/timeline.csv', 'w') as f:
    f.write('name,queue,start_ns,end_ns,grid,wg,lds,vgpr\n')
    t0 = int(keep[0]['Start_Timestamp'])
    for r in keep:
        f.write('%s,%s,%d,%d,%s,%s,%s,%s\n' % (r['Kernel_Name'].split('(')[0][:70].replace(',', ';'), r.get('Queue_Id', ''),
                int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0, r.get('Grid_Size_X', r.get('Grid_Size', '')),
                r.get('Workgroup_Size_X', r.get('Workgroup_Size', '')), r.get('LDS_Block_Size', ''), r.get('VGPR_Count', '')))
print('rows', len(rows), 'cols', list(rows[0].keys()))
PY
rm -rf $OUT/kt
ls -la $OUT
