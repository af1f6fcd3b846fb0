"""Builds profiles/rNN_* from the raw outputs of one profiling session on the GPU box (see profiles/README in the
summary it writes for the exact commands):

    python scripts/make_profile_summary.py r01 <kernel_stats.csv> <bench.json> <pmc_fetch.csv> <pmc_write.csv> \
           [<cls_kernel_stats.csv> <cls_bench.json>]

kernel_stats.csv: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 5 --warmup 2
--no-cpu-baseline` (7 steps recorded); pmc_*.csv: the counter_collection.csv of `rocprofv3 --pmc FETCH_SIZE` and
`--pmc WRITE_SIZE` (separate passes) over `bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer`.
"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def read_stats(path):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((r['Name'].replace('(anonymous namespace)::', ''), int(r['Calls']), float(r['TotalDurationNs']), float(r['AverageNs']), float(r['Percentage'])))
    return rows


# kernels of the forward + data-gradient convolution family (bench.py's conv_nt_* timer records): the direct
# implicit-GEMM kernel with its split-tile fix-up, and the Winograd path (transforms + GEMM).  A "launch" of the family
# is one convolution call = one dispatch of a main kernel.
NT_MAIN = ('conv_nt_kernel', 'wino_gemm_kernel', 'stem_conv_kernel')
NT_AUX = ('conv_fixup_kernel', 'wino_input_kernel', 'wino_weight_kernel')
TN_FAMILY = ('conv_tn_kernel', 'conv_tn3_kernel', 'wgrad_reduce_kernel', 'wino_wgrad_kernel', 'wino_wgrad3_kernel', 'wino_dy_kernel',
             'wino_wgrad_reduce_kernel', 'stem_wgrad_kernel', 'wgrad_reduce_stem_kernel')


def pmc_per_launch(path, counter, mains, aux=()):
    """(sum of the counter over the family's dispatches, number of main-kernel dispatches)."""
    tot, n = 0.0, 0
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        name = r['Kernel_Name']
        if any(m in name for m in mains):
            tot += float(r['Counter_Value'])
            n += 1
        elif any(m in name for m in aux):
            tot += float(r['Counter_Value'])
    return tot, n


def resnet50_bn_layers(B, H, W):
    """(elements, relu, residual) of every BatchNorm output of the ResNet50 trunk."""
    o = lambda s, k, st, p: (s + 2 * p - k) // st + 1
    h, w = o(H, 7, 2, 3), o(W, 7, 2, 3)
    layers = [(B * h * w * 64, True, False)]
    h, w = o(h, 3, 2, 1), o(w, 3, 2, 1)
    cin = 64
    for li, (n, pl) in enumerate(zip([3, 4, 6, 3], [64, 128, 256, 512])):
        for bi in range(n):
            st = 2 if (bi == 0 and li > 0) else 1
            ho, wo = o(h, 3, st, 1), o(w, 3, st, 1)
            layers.append((B * h * w * pl, True, False))        # conv1 at the input resolution
            layers.append((B * ho * wo * pl, True, False))      # conv2
            layers.append((B * ho * wo * pl * 4, True, True))   # conv3 + residual
            if st != 1 or cin != pl * 4:
                layers.append((B * ho * wo * pl * 4, False, False))  # projection shortcut
            cin, h, w = pl * 4, ho, wo
    return layers


def hbm_table(stats_by_name, steps, B, H, W, extra, elem_bytes=4):
    L = resnet50_bn_layers(B, H, W)
    gb = lambda f: sum(f(e, relu, res) for e, relu, res in L) * elem_bytes / 1e9
    rows = [
        ('bn_apply_kernel', gb(lambda e, r, s: e * (3 if s else 2)), 'read y, write z (+ the residual input of the 16 block outputs)'),
        # (round 4: bn_bwd_partial_kernel runs for 12 of the trunk's 53 layers only -- the others' first stage is an epilogue
        #  option of the producing data gradient -- so it has no row of its own any more)
        ('bn_bwd_apply_kernel', gb(lambda e, r, s: e * (5 if s else 3)), 'read dz, y (+ z), write dy (+ the masked gradient copy for the residual branch)'),
    ] + extra
    out = []
    for name, g, what in rows:
        hit = [v for k, v in stats_by_name.items() if name in k]  # templated kernels: 'void bn_apply_kernel<float>(...'
        if not hit:
            continue
        ms = sum(h[1] for h in hit) / steps / 1e6
        out.append(f'| `{name}` | {g:.2f} | {ms:.3f} | {g / ms:.2f} | {g / ms / 8.0:.2f} | {what} |')
    return out


def mfma_utilisation(path, tag):
    """SQ_VALU_MFMA_BUSY_CYCLES per kernel family / (duration x 2.4 GHz x 1024 SIMDs); wave-cycle breakdown."""
    import collections
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    dur, n, seen = collections.defaultdict(float), collections.Counter(), set()
    for r in csv.DictReader(open(path)):
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Dispatch_Id'] not in seen:
            seen.add(r['Dispatch_Id'])
            n[k] += 1
            dur[k] += float(r['End_Timestamp']) - float(r['Start_Timestamp'])
    out, md = {}, ['', '## Matrix-pipe utilisation (PMC pass, `SQ_VALU_MFMA_BUSY_CYCLES`; kernels run slower under the profiler)', '',
                   '`rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY` '
                   'over `bench.py --steps 2 --warmup 1`; utilisation = busy cycles / (kernel duration x 2.4 GHz x 1024 SIMDs); it counts '
                   'the zero-padded part of edge tiles too, which the algorithmic TFLOP/s above do not.', '',
                   '| kernel | launches | MFMA busy | waves parked (s_waitcnt/barrier) | issue stalls | issuing |', '|---|---|---|---|---|---|']
    for k in sorted(agg, key=lambda k: -dur[k]):
        a = agg[k]
        if a.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) <= 0:
            continue
        u = a['SQ_VALU_MFMA_BUSY_CYCLES'] / (dur[k] * 2.4 * 1024)
        w = a['SQ_WAVE_CYCLES']
        out[k] = dict(launches=n[k], mfma_busy_frac=round(u, 4), wait_any=round(a['SQ_WAIT_ANY'] / w, 3),
                      wait_inst_any=round(a['SQ_WAIT_INST_ANY'] / w, 3), active_inst=round(a['SQ_ACTIVE_INST_ANY'] / w, 3))
        md.append(f'| `{k}` | {n[k]} | {u:.2f} | {a["SQ_WAIT_ANY"] / w:.2f} | {a["SQ_WAIT_INST_ANY"] / w:.2f} | {a["SQ_ACTIVE_INST_ANY"] / w:.2f} |')
    json.dump(out, open(os.path.join(ROOT, 'profiles', f'{tag}_pmc_mfma.json'), 'w'), indent=1)
    return md


def main():
    mfma_md = []
    if '--mfma' in sys.argv:
        i = sys.argv.index('--mfma')
        mfma_md = mfma_utilisation(sys.argv[i + 1], sys.argv[1])
        del sys.argv[i:i + 2]
    overlapped = None
    if '--overlapped' in sys.argv:
        i = sys.argv.index('--overlapped')
        overlapped = sys.argv[i + 1]
        del sys.argv[i:i + 2]
    tag, stats_p, bench_p, pf, pw = sys.argv[1:6]
    steps = 7
    prof = os.path.join(ROOT, 'profiles')
    rows = read_stats(stats_p)
    by = {r[0]: (r[1], r[2]) for r in rows}
    with open(os.path.join(prof, f'{tag}_bench_kernel_stats.csv'), 'w') as f:
        f.write(open(stats_p).read())
    bench = json.loads([l for l in open(bench_p) if l.startswith('{')][-1])
    # PMC traffic of the conv_nt family
    fs, n1 = pmc_per_launch(pf, 'FETCH_SIZE', NT_MAIN, NT_AUX)
    ws, n2 = pmc_per_launch(pw, 'WRITE_SIZE', NT_MAIN, NT_AUX)
    traffic = dict(kernel='conv_nt_kernel<128,*,*> + conv_fixup_kernel, wino_gemm_kernel + wino_input_kernel + '
                          'wino_weight_kernel, stem_conv_kernel (per convolution call)', launches=n1, fetch_size_kb_per_launch=fs / n1,
                   write_size_kb_per_launch=ws / n2, hbm_bytes_per_launch=(2 * fs / n1 + ws / n2) * 1024,
                   note='rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over bench.py --steps 2 '
                        '--warmup 1; gfx950 correction: FETCH_SIZE counts 64 B per 128-B request, so reads are doubled '
                        '(MI355X_MICROARCH.md, HBM section); KB -> bytes x1024')
    json.dump(traffic, open(os.path.join(prof, f'{tag}_pmc_traffic.json'), 'w'), indent=1)
    bench['roofline']['traffic'] = round(traffic['hbm_bytes_per_launch'])
    if bench['roofline'].get('algorithmic_bytes_per_launch'):
        bench['roofline']['traffic_ratio'] = round(bench['roofline']['traffic'] / bench['roofline']['algorithmic_bytes_per_launch'], 2)
    json.dump(bench, open(os.path.join(prof, f'{tag}_bench.json'), 'w'))
    total_ms = sum(r[2] for r in rows) / steps / 1e6
    md = [f'# Round {tag[1:]} profile summary (MI355X, 1 GPU)', '',
          'Command: `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --engine-set side_stream=False` (`scripts/profile_session.sh`)',
          f'({steps} steps recorded; raw table: `{tag}_bench_kernel_stats.csv`; un-profiled bench line with CPU baseline: `{tag}_bench.json`;',
          f'PMC HBM-traffic passes for the conv kernel family: `{tag}_pmc_traffic.json`; generated by `scripts/make_profile_summary.py`).', '',
          f'Sum of kernel time per step: {total_ms:.2f} ms; bench ms_per_step {bench["ms_per_step"]} (un-profiled): the step is GPU-bound.', '',
          '| kernel | calls/step | ms/step | avg us | % |', '|---|---|---|---|---|']
    for name, calls, tot, avg, pct in rows[:30]:
        md.append(f'| `{name[:60]}` | {calls / steps:.1f} | {tot / steps / 1e6:.3f} | {avg / 1e3:.1f} | {pct:.2f} |')
    r = bench['roofline']
    fam = lambda names: sum(v[1] for k, v in by.items() if any(n in k for n in names)) / steps / 1e6
    nt_ms = fam(NT_MAIN + NT_AUX)
    nt_calls = sum(v[0] for k, v in by.items() if any(n in k for n in NT_MAIN)) / steps
    tn_ms = fam(TN_FAMILY)
    wino_ms, direct_ms = fam(('wino_gemm_kernel', 'wino_input_kernel', 'wino_weight_kernel')), fam(('conv_nt_kernel', 'conv_fixup_kernel'))
    md += ['', '## Roofline per kernel class', '',
           f'* MFMA-bound: convolution forward + data gradient (`conv_nt_kernel` direct, `wino_gemm_kernel` Winograd F(2x2,3x3) for the '
           f'head 3x3 convolutions; fp32 operands as three bf16 terms, six products per fp32 product on the {r.get("pipe", "fp32")} '
           f'matrix pipe) -- bench.py HIP events: {r["achieved"]} TFLOP/s EXECUTED on the matrix cores of {r["peak"]} '
           f'({r.get("pipe", "fp32")} dense) = {r["frac"]} (Winograd calls priced at their flops / 2.25, split-operand calls at 6 bf16 '
           f'flops per fp32 flop); the ALGORITHMIC (direct-convolution, fp32) rate of the same launches is '
           f'{r.get("effective_achieved")} TFLOP/s = {r.get("effective_frac")} of that pipe\'s ceiling for fp32 products '
           f'({r.get("effective_peak")} = peak / 6) and {r.get("vs_fp32_matrix_peak")} x the fp32 matrix pipe\'s 157.3; avg call '
           f'{r["avg_launch_ms"]} ms, '
           f'{r["algorithmic_gflop_per_launch"]} algorithmic GFLOP/call, PMC traffic / algorithmic bytes = {r.get("traffic_ratio")} '
           f'({r["launches_per_step"]} convolution calls/step). rocprofv3 over the same kernels (incl. fix-up and Winograd transforms): '
           f'{nt_calls:.0f} main-kernel launches/step, {nt_ms:.2f} ms/step (avg {nt_ms / nt_calls:.4f} ms): direct {direct_ms:.2f} ms, '
           f'Winograd {wino_ms:.2f} ms. HBM traffic {traffic["hbm_bytes_per_launch"] / 1e6:.0f} MB/call (PMC).',
           f'* MFMA-bound: weight gradient (`conv_tn3_kernel` direct, `wino_wgrad3_kernel` Winograd; with their reductions / transforms) -- '
           f'the same algorithmic flops as the forward convolutions '
           f'({r["algorithmic_gflop_per_launch"] * r["launches_per_step"] / 2 / 1e3:.3f} TFLOP/step) in {tn_ms:.2f} ms = '
           f'{r["algorithmic_gflop_per_launch"] * r["launches_per_step"] / 2 / tn_ms:.1f} TFLOP/s.',
           '* HBM-bound classes (algorithmic bytes / rocprof time, peak 8 TB/s spec):', '',
           '| kernel | algorithmic GB/step | ms/step | achieved TB/s | frac of 8 TB/s | bytes counted |', '|---|---|---|---|---|---|']
    extra = [('adamw_gated_kernel', 37968692 * 4 * 7 / 1e9, 'read p,g,m,v write p,m,v'),
             ('im2col_stem_kernel', (2 * 3 * 800 * 1333 + 2 * 400 * 667 * 160) * 4 / 1e9, 'read image, write im2col matrix'),
             ('wino_input_kernel', 0.0, ''), ('wino_dy_kernel', 0.0, '')]
    extra = [e for e in extra if e[1] > 0]
    md += hbm_table(by, steps, 2, 800, 1333, extra)
    md += mfma_md
    if overlapped:
        orow = read_stats(overlapped)
        with open(os.path.join(prof, f'{tag}_bench_kernel_stats_overlapped.csv'), 'w') as f:
            f.write(open(overlapped).read())
        ototal = sum(r[2] for r in orow) / steps / 1e6
        md += ['', '## The step as it runs: weight gradients on the side stream', '',
               'The tables above come from a run with `--engine-set side_stream=False` (every kernel in stream order, alone on the chip: the '
               "durations `roofline` is computed from; bench.py's instrumented steps run the same way).  In normal steps the weight "
               'gradients, their reductions, the bias-gradient column sums and the shortcut branch of the forward pass run on a second '
               f'HIP stream.  `rocprofv3 --kernel-trace --stats` of such a run (`{tag}_bench_kernel_stats_overlapped.csv`): the kernel '
               f'durations sum to {ototal:.2f} ms per step (two queues; overlapped kernels share the chip and each takes longer) while '
               f'the step takes {bench.get("ms_per_step_median", bench["ms_per_step"])} ms (median of the un-instrumented steps).', '',
               '| kernel | calls/step | ms/step (overlapped) | avg us |', '|---|---|---|---|']
        for name, calls, tot, avg, pct in orow[:14]:
            md.append(f'| `{name[:60]}` | {calls / steps:.1f} | {tot / steps / 1e6:.3f} | {avg / 1e3:.1f} |')
    if 'cpu_baseline' in bench:
        md += ['', f'CPU baseline (same run, {bench["cpu_baseline"]["cores"]} host threads): {bench["cpu_baseline"]}']
    if len(sys.argv) > 7:
        crow = read_stats(sys.argv[6])
        cb = json.loads([l for l in open(sys.argv[7]) if l.startswith('{')][-1])
        cby = {r[0]: (r[1], r[2]) for r in crow}
        half = cb.get('dtype') == 'bf16'
        with open(os.path.join(prof, f'{tag}_cls_kernel_stats.csv'), 'w') as f:
            f.write(open(sys.argv[6]).read())
        json.dump(cb, open(os.path.join(prof, f'{tag}_cls_bench.json'), 'w'))
        ctot = sum(r[2] for r in crow) / steps / 1e6
        rf = cb['roofline']
        md += ['', f'## Secondary workload: ResNet50 ImageNet classification, bs 256, 224x224, {cb.get("dtype", "f32")} (SURVEY 8f-1)', '',
               f'`bench.py --model resnet50_cls{" --dtype bf16" if half else ""}`: {cb["value"]} images/s, {cb["ms_per_step"]} ms/step '
               f'(un-profiled; `{tag}_cls_bench.json`); kernel time {ctot:.2f} ms/step (`{tag}_cls_kernel_stats.csv`). '
               f'conv_nt family: {rf["achieved"]} {rf["unit"]} = {rf["frac"]} of the {rf["bound"]} peak {rf["peak"]}'
               + (f' (matrix cores at {rf.get("matrix_tflops")} TFLOP/s; layer1 3x3: {rf.get("layer1_3x3")})' if half else '') + '.', '',
               '| kernel | calls/step | ms/step | avg us | % |', '|---|---|---|---|---|']
        for name, calls, tot, avg, pct in crow[:16]:
            md.append(f'| `{name[:60]}` | {calls / steps:.1f} | {tot / steps / 1e6:.3f} | {avg / 1e3:.1f} | {pct:.2f} |')
        md += ['', '| kernel | algorithmic GB/step | ms/step | achieved TB/s | frac of 8 TB/s | bytes counted |', '|---|---|---|---|---|---|']
        eb = 2 if half else 4
        kp = 192 if half else 160
        md += hbm_table(cby, steps, 256, 224, 224, [
            ('im2col_stem', (256 * 3 * 224 * 224 * 4 + 256 * 112 * 112 * kp * eb) / 1e9, 'read image (fp32), write im2col matrix')], eb)
    open(os.path.join(prof, f'{tag}_summary.md'), 'w').write('\n'.join(md) + '\n')
    print('\n'.join(md[:20]))


if __name__ == '__main__':
    main()
