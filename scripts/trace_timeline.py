"""Timeline of one training step from a rocprofv3 kernel trace (csv): per queue busy time, gaps, per-kernel totals.
    python scripts/trace_timeline.py gpurun_out/r4_base/kernel_trace_overlapped.csv [--list]"""
import csv, sys, collections, re
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
def short(n):
    n = re.sub(r'^void ', '', n)
    n = n.replace('(anonymous namespace)::', '')
    n = re.sub(r'\(.*', '', n)
    return n[:60]
ad = [i for i, r in enumerate(rows) if 'adamw' in r['Kernel_Name']]
a, b = ad[-2], ad[-1]
step = rows[a + 1:b + 1]
t0, t1 = step[0]['s'], step[-1]['e']
print(f'step span {(t1 - t0) / 1e6:.3f} ms, {len(step)} dispatches')
queues = collections.defaultdict(list)
for r in step:
    queues[r['Queue_Id']].append(r)
for q, rs in queues.items():
    busy = sum(r['e'] - r['s'] for r in rs)
    gaps = [rs[i + 1]['s'] - rs[i]['e'] for i in range(len(rs) - 1)]
    pos = [g for g in gaps if g > 0]
    print(f'queue {q}: {len(rs)} kernels, busy {busy / 1e6:.3f} ms, first {(rs[0]["s"] - t0) / 1e6:.3f} last end {(rs[-1]["e"] - t0) / 1e6:.3f}; '
          f'sum positive gaps {sum(pos) / 1e6:.3f} ms (n={len(pos)}), gaps>5us: {sum(1 for g in pos if g > 5000)} totalling {sum(g for g in pos if g > 5000) / 1e6:.3f}')
    agg = collections.defaultdict(lambda: [0, 0])
    for r in rs:
        k = short(r['Kernel_Name']); agg[k][0] += 1; agg[k][1] += r['e'] - r['s']
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f'   {t / 1e6:8.3f} ms {n:4d}  {k}')
if '--list' in sys.argv:
    for r in step:
        print(f'{(r["s"] - t0) / 1e3:10.1f} {(r["e"] - r["s"]) / 1e3:8.1f} q{r["Queue_Id"]} g{r["Grid_Size_X"]:>8} wg{r["Workgroup_Size_X"]:>5} v{r["VGPR_Count"]:>4} {short(r["Kernel_Name"])}')
