#!/usr/bin/env python3
"""Headline benchmark: images/sec of the RetinaNet-ResNet50 COCO training step (forward, RetinaLoss,
backward, gradient all-reduce, AdamW) at 1333x800, batch 2 per GPU, fp32, synthetic data.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One process per GPU (RCCL).  Prints ONE JSON line on rank 0 (contract in the repo README / DESIGN.md).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H, W, BATCH = 800, 1333, 2
PEAK_FP32_MFMA_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, fp32 matrix
PEAK_HBM_GBS = 8000.0          # same guide: HBM3E 8 TB/s spec (6.3 TB/s measured with a float4 copy)
WINO_SAVING = 2.25             # F(2x2,3x3): 36 direct MACs per 16 Winograd-domain MACs
PEAK_BF16_MFMA_TFLOPS = 2516.6  # same guide: 256 CUs x 4 SIMDs x 1024 flop/clk (v_mfma_f32_32x32x16_bf16: 32 cycles) x 2.4 GHz
SPLIT_PRODUCTS = 6             # an fp32 product on the bf16 pipe: six bf16 products (operands as three bf16 terms)


def pipe_of(family):
    """('bf16' | 'fp32', executed matrix-core flops per algorithmic flop, dense peak of that pipe) of a timer family.
    Families ending in 3 (conv_nt3_*, conv_tn3, conv_nt_wino3, conv_tn_wino3) run fp32 GEMMs on the bf16 pipe with the
    operands split into three bf16 terms: 6 executed bf16 flops per fp32 flop; the Winograd families execute 1 / 2.25 of
    the direct-convolution flops; *_bf16 families are plain bf16 (autocast)."""
    fam = family.split('|')[0]
    ratio = 1.0 / WINO_SAVING if 'wino' in fam else 1.0
    if fam.endswith('_bf16'):
        return 'bf16', ratio, PEAK_BF16_MFMA_TFLOPS
    if fam in ('conv_tn3', 'conv_nt_wino3', 'conv_tn_wino3') or fam.startswith('conv_nt3'):
        return 'bf16', ratio * SPLIT_PRODUCTS, PEAK_BF16_MFMA_TFLOPS
    return 'fp32', ratio, PEAK_FP32_MFMA_TFLOPS


def family_roofline(agg, prefix):
    """Launches, time, algorithmic flops / bytes of the timer families starting with `prefix`, and their matrix-core
    accounting: executed flops on each pipe, the time they would take at that pipe's dense peak (`peak_ms`), hence the
    utilisation frac = peak_ms / ms whatever the mix of pipes.  `achieved` / `peak` are quoted on the pipe that does most
    of the work (launches of the other pipe enter at the ratio of the two peaks, so that achieved / peak = frac)."""
    k = dict(launches=0, flops=0.0, ms=0.0, bytes=0.0, peak_ms=0.0, exec_bf16=0.0, exec_fp32=0.0)
    for name, v in agg.items():
        if not name.startswith(prefix):
            continue
        pipe, ratio, peak = pipe_of(name)
        for f_ in ('launches', 'flops', 'ms'):
            k[f_] += v[f_]
        k['bytes'] += v.get('bytes', 0.0)
        k['peak_ms'] += v['flops'] * ratio / (peak * 1e12) * 1e3
        k['exec_' + pipe] += v['flops'] * ratio
    if not k['launches']:
        return None
    k['pipe'] = 'bf16' if k['exec_bf16'] / PEAK_BF16_MFMA_TFLOPS >= k['exec_fp32'] / PEAK_FP32_MFMA_TFLOPS else 'fp32'
    k['peak'] = PEAK_BF16_MFMA_TFLOPS if k['pipe'] == 'bf16' else PEAK_FP32_MFMA_TFLOPS
    k['frac'] = k['peak_ms'] / k['ms']
    k['achieved'] = k['frac'] * k['peak']
    k['effective'] = k['flops'] / (k['ms'] * 1e-3) / 1e12
    # ceiling of the ALGORITHMIC rate on the pipe in use: the fp32 matrix peak, or the bf16 peak / 6 products
    k['effective_peak'] = PEAK_BF16_MFMA_TFLOPS / SPLIT_PRODUCTS if k['pipe'] == 'bf16' else PEAK_FP32_MFMA_TFLOPS
    return k


def cpu_baseline(warmup=3, steps=5):
    """The CPU oracle's training step (torch-CPU fp32 restatement of the reference step, pinned to the reference by
    tests/golden) timed on this host's cores: `warmup` + `steps` steps of the SAME workload (B=2, 800x1333), median,
    with the forward / loss / backward / optimizer split."""
    import statistics
    import torch
    from oracle import losses as olosses, model as omodel, synth
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    # a 1-GPU box is a 16-core SHARE of a large host: the affinity mask shows every host core, but threads beyond the
    # share only time-slice against the other tenants (192 threads here: >50 s per step instead of ~6)
    cores = min(16, cores)
    torch.set_num_threads(cores)
    sd = omodel.init_retinanet_state(0)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()
              if v.dtype.is_floating_point and 'running_' not in k}
    state = {k: v.clone() for k, v in sd.items()}
    state.update(params)
    opt = torch.optim.AdamW([{'params': [p for p in params.values() if p.ndim == 1], 'weight_decay': 0.0},
                             {'params': [p for p in params.values() if p.ndim != 1], 'weight_decay': 1e-3}],
                            lr=1e-4)
    crit = olosses.RetinaLossOracle()
    images = torch.tensor(synth.images(0, BATCH, H, W))
    annots = torch.tensor(synth.annotations(0, BATCH, H, W, boxes_per_image=8))
    rows = []
    for it in range(warmup + steps):
        t0 = time.perf_counter()
        outs = omodel.retinanet_forward(state, images, training=True)
        t1 = time.perf_counter()
        ld = crit(outs, annots)
        t2 = time.perf_counter()
        opt.zero_grad()
        sum(ld.values()).backward()
        t3 = time.perf_counter()
        opt.step()
        t4 = time.perf_counter()
        rows.append((t4 - t0, t1 - t0, t2 - t1, t3 - t2, t4 - t3))
    timed = rows[warmup:]
    med = [statistics.median(r[i] for r in timed) for i in range(5)]
    return dict(value=BATCH / med[0], unit='images/sec', cores=torch.get_num_threads(), kind='port',
                seconds_per_step=round(med[0], 3),
                split_seconds=dict(forward=round(med[1], 3), loss=round(med[2], 3), backward=round(med[3], 3),
                                   optimizer=round(med[4], 3)),
                sample=f'median of {steps} timed steps (after {warmup} warm-up) of the same workload: B=2, 800x1333, '
                       f'RetinaNet-R50 fwd+RetinaLoss+bwd+AdamW, torch-CPU fp32 oracle, {med[0]:.2f} s/step')


def class_table(agg, steps):
    """Per conv class (kernel family, mode, kernel size, stride, channels): launches per step, time, matrix-core
    TFLOP/s and its fraction of the dtype's dense MFMA peak, algorithmic GB/s (input + output + weights once) and its
    fraction of the 8 TB/s HBM peak -- the bound that applies is the larger fraction's."""
    rows = []
    for key, v in sorted(agg.items(), key=lambda kv: -kv[1]['ms']):
        # `flops` of a record are ALGORITHMIC (direct-convolution, fp32) flops.  `tflops` / `mfma_frac` price what the
        # matrix cores EXECUTE on the pipe the class runs on (pipe_of: Winograd classes 1 / 2.25 of the direct flops,
        # split-operand classes 6 bf16 flops per fp32 flop) against that pipe's dense peak -- a utilisation, never above 1;
        # `effective_tflops` keeps the algorithmic rate
        pipe, ratio, peak = pipe_of(key)
        tf = v['flops'] * ratio / (v['ms'] * 1e-3) / 1e12 if v['ms'] else 0.0
        etf = v['flops'] / (v['ms'] * 1e-3) / 1e12 if v['ms'] else 0.0
        gbs = v.get('bytes', 0.0) / (v['ms'] * 1e-3) / 1e9 if v['ms'] else 0.0
        rows.append(dict(cls=key, launches_per_step=v['launches'] / steps, ms_per_step=round(v['ms'] / steps, 4),
                         pipe=pipe, tflops=round(tf, 1), mfma_frac=round(tf / peak, 4), effective_tflops=round(etf, 1),
                         gbs=round(gbs, 1), hbm_frac=round(gbs / PEAK_HBM_GBS, 4)))
    return rows


def launch_ranks(n):
    """`python bench.py --gpus N` from a plain shell: start N rank processes with torch.distributed.run (the same
    command line the driver uses) as a child, stream its output, return its exit code.  Mirrors the reference's
    launch line `torchrun --nproc_per_node=N tools/train_detection_model.py` (train.sh of each recipe;
    tools/train_detection_model.py:50-59 reads LOCAL_RANK and calls init_process_group('nccl'))."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')  # dmabuf IPC (RCCL across processes on this driver)
    env.setdefault('OMP_NUM_THREADS', '4')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    rc = subprocess.call(cmd, env=env)
    if rc != 0:
        print(f'bench.py: the {n}-rank run failed with exit code {rc}', file=sys.stderr, flush=True)
    sys.exit(rc)


def apply_launch_plan(args, ops):
    """--launch-plan NAME=VALUE[,NAME=VALUE...]: fields of the library's launch plan (ops.launch_share) for A/B runs."""
    if getattr(args, 'launch_plan', None):
        ops.launch_share(**{kv.split('=')[0]: int(kv.split('=')[1]) for kv in args.launch_plan.split(',')})


def apply_engine_settings(args, net):
    """--engine-set NAME=VALUE: attributes of the model's executor (A/B runs, the profile session's stream-ordered trace)."""
    import ast
    for kv in getattr(args, 'engine_set', []):
        name, value = kv.split('=', 1)
        if not hasattr(net.engine, name):
            raise SystemExit(f'--engine-set: the executor has no attribute {name!r}')
        setattr(net.engine, name, ast.literal_eval(value))


def bench_classification(args, rank, world, group, ops, synth):
    """Secondary workload: 0.classification_training/imagenet/resnet50/train_config.py --
    batch 256, 224x224, CELoss, SGD(0.1, 0.9, wd 1e-4) -- in fp32 (the reference recipe runs fp16 autocast)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from sacv_amd.classification import backbones, losses
    from sacv_amd.train import ClassificationTrainer
    CB, R = 256, 224
    amp = args.dtype == 'bf16'
    net = backbones.__dict__['resnet50'](num_classes=1000).cuda().train()
    trainer = ClassificationTrainer(net, losses.__dict__['CELoss'](), lr=0.1, momentum=0.9, weight_decay=1e-4,
                                    process_group=group, use_amp=amp)
    apply_engine_settings(args, net)
    images = torch.cat([torch.tensor(synth.images(rank * 64 + i, CB // 8, R, R)) for i in range(8)]).cuda()
    labels = torch.tensor((synth.uniform(rank + 99, (CB,)) * 1000).astype(np.int64) % 1000).cuda()

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = trainer.step(images, labels)
    timer = None
    if not args.no_kernel_timer:
        timer = ops.KernelTimer()
    sync()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        ops.TIMER = timer if (timer is not None and i % args.timer_every == 0) else None
        out = trainer.step(images, labels)
        marks[i + 1].record()
    sync()
    elapsed = time.perf_counter() - t0
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    ms_median = per_step[len(per_step) // 2]
    ops.TIMER = None
    n_inst = len(range(0, args.steps, args.timer_every))  # timed steps that carried per-launch events
    loss = trainer.finish_step(out)
    if world > 1:
        t = torch.tensor([elapsed], device='cuda', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    roofline = None
    if timer is not None:
        agg = timer.summary()
        k = family_roofline(agg, 'conv_nt') or dict(launches=0)
        if k['launches'] and not amp:
            roofline = dict(kernel='conv_nt_kernel / wino_gemm_kernel (conv forward + data gradient; fp32 operands as three '
                                   'bf16 terms, six products per fp32 product on the bf16 matrix pipe, fp32 accumulators); '
                                   'achieved = executed matrix-core flops / time, effective_achieved = algorithmic fp32 flops / '
                                   'time, effective_peak = that pipe\'s peak / 6',
                            bound='mfma', pipe=k['pipe'], achieved=round(k['achieved'], 2), peak=k['peak'], unit='TFLOP/s',
                            frac=round(k['frac'], 4), effective_achieved=round(k['effective'], 2),
                            effective_peak=round(k['effective_peak'], 1),
                            effective_frac=round(k['effective'] / k['effective_peak'], 4),
                            vs_fp32_matrix_peak=round(k['effective'] / PEAK_FP32_MFMA_TFLOPS, 3), traffic=None,
                            launches_per_step=k['launches'] // n_inst,
                            share_of_step_ms=round(k['ms'] / n_inst, 2))
        elif k['launches']:
            # bf16: these convolutions sit at or below the HBM ridge -> algorithmic bytes (input +
            # output + weights, each once) / HIP-event time against 8 TB/s; the matrix-core rate rides along
            gbs = k['bytes'] / (k['ms'] * 1e-3) / 1e9
            cls3 = dict(launches=0, flops=0.0, ms=0.0, bytes=0.0)
            for name, v in agg.items():  # layer1 3x3 (64 -> 64 @ 56x56), forward + data gradient
                if name.startswith('conv_nt_bf16|') and ' k3 s1 cs64 n64 ' in name:
                    for f_ in cls3:
                        cls3[f_] += v[f_]
            roofline = dict(kernel='conv_nt_bf16_kernel (implicit-GEMM conv forward + data gradient, bf16 MFMA 32x32x16)',
                            bound='hbm', achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit='GB/s',
                            frac=round(gbs / PEAK_HBM_GBS, 4), traffic=None,
                            launches_per_step=k['launches'] // n_inst,
                            share_of_step_ms=round(k['ms'] / n_inst, 2),
                            matrix_tflops=round(k['flops'] / (k['ms'] * 1e-3) / 1e12, 1))
            if cls3['launches']:
                g3 = cls3['bytes'] / (cls3['ms'] * 1e-3) / 1e9
                roofline['layer1_3x3'] = dict(achieved=round(g3, 1), frac=round(g3 / PEAK_HBM_GBS, 4), unit='GB/s',
                                              avg_launch_us=round(cls3['ms'] / cls3['launches'] * 1e3, 1),
                                              matrix_tflops=round(cls3['flops'] / (cls3['ms'] * 1e-3) / 1e12, 1))
        if os.environ.get('SACV_BENCH_DUMP') and rank == 0:
            recs = timer.records[-(len(timer.records) // n_inst):]
            with open(os.environ['SACV_BENCH_DUMP'], 'w') as f:
                for key, flops, e0, e1 in recs:
                    ms = e0.elapsed_time(e1)
                    f.write(f'{key} {flops/1e9:.3f} GFLOP {ms*1e3:.1f} us {flops/(ms*1e-3)/1e12:.1f} TF\n')
        if args.class_table and rank == 0:
            with open(args.class_table, 'w') as f:
                json.dump(dict(model='resnet50_cls', dtype=args.dtype, batch=CB, steps=args.steps,
                               peak_tflops_bf16_mfma=PEAK_BF16_MFMA_TFLOPS, peak_tflops_fp32_mfma=PEAK_FP32_MFMA_TFLOPS,
                               peak_hbm_gbs=PEAK_HBM_GBS, classes=class_table(agg, n_inst)), f, indent=1)
    if rank == 0:
        print(json.dumps({
            'metric': 'images/sec/node ResNet50 ImageNet 224x224 bs=256/GPU (secondary, SURVEY 8f-1)',
            'value': round(world * CB * args.steps / elapsed, 2), 'unit': 'images/sec', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 3),
            'ms_per_step_median': round(ms_median, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': 'ResNet50 ImageNet training step (fwd + CELoss + bwd + grad all-reduce + SGD '
                                   'momentum), images [256,3,224,224]/GPU, random init'
                                   + (', bf16 activations / fp32 accumulate+master weights (autocast)' if amp else ''),
                       'global_batch': world * CB, 'parallelism': f'dp{world}'},
            'loss': loss, 'roofline': roofline}), flush=True)
    trainer.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def bench_mae(args, rank, world, group, ops, synth):
    """Secondary workload (BASELINE config #5): ViT-B/16 MAE pre-training step of
    2.masked_image_modeling_training/imagenet/mae_vit_base_patch16_224/train_config.py -- total batch 1024 over 8 GPUs =
    128 images/GPU, 224x224, mask ratio 0.75, MSELoss on per-patch-normalised pixels, AdamW(6e-4, (0.9, 0.95), wd 0.05),
    autocast: bf16 matrix-core GEMMs, fp32 residual stream / statistics / master weights."""
    import torch
    import torch.distributed as dist
    from sacv_amd.masked_image_modeling import models, losses
    from sacv_amd.train import MAETrainer
    MB, R = 128, 224
    net = models.__dict__['vit_base_patch16_224_mae_pretrain_model']().cuda().train()
    trainer = MAETrainer(net, losses.__dict__['MSELoss'](), lr=6e-4, betas=(0.9, 0.95), weight_decay=5e-2,
                         process_group=group)
    apply_engine_settings(args, net)
    images = (torch.cat([torch.tensor(synth.images(rank * 64 + i, MB // 8, R, R)) for i in range(8)]) * 4 - 2).cuda()

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = trainer.step(images)
    timer = None
    if not args.no_kernel_timer:
        timer = ops.KernelTimer()
    sync()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        ops.TIMER = timer if (timer is not None and i % args.timer_every == 0) else None
        out = trainer.step(images)
        marks[i + 1].record()
    sync()
    elapsed = time.perf_counter() - t0
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    ms_median = per_step[len(per_step) // 2]
    ops.TIMER = None
    n_inst = len(range(0, args.steps, args.timer_every))  # timed steps that carried per-launch events
    loss = trainer.finish_step(out)
    if world > 1:
        t = torch.tensor([elapsed], device='cuda', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    roofline = None
    if timer is not None:
        agg = timer.summary()
        k = dict(launches=0, flops=0.0, ms=0.0, bytes=0.0)
        for name, v in agg.items():
            if name.startswith('conv_nt_bf16') or name.startswith('conv_tn_bf16'):
                for f_ in k:
                    k[f_] += v[f_]
        if k['launches']:
            tf = k['flops'] / (k['ms'] * 1e-3) / 1e12
            roofline = dict(kernel='conv_nt_bf16 / conv_tn_bf16 as Linear GEMMs (bf16 MFMA 32x32x16: forward, data and '
                                   'weight gradients of patch-embed / qkv / proj / fc1 / fc2 / pred)',
                            bound='mfma', achieved=round(tf, 1), peak=2500.0, unit='TFLOP/s', frac=round(tf / 2500.0, 4),
                            traffic=None, launches_per_step=k['launches'] // n_inst,
                            share_of_step_ms=round(k['ms'] / n_inst, 2),
                            algorithmic_gbs=round(k['bytes'] / (k['ms'] * 1e-3) / 1e9, 1))
        if args.class_table and rank == 0:
            with open(args.class_table, 'w') as f:
                json.dump(dict(model='vit_mae', dtype='bf16', batch=MB, steps=args.steps, peak_tflops_bf16_mfma=PEAK_BF16_MFMA_TFLOPS,
                               peak_hbm_gbs=PEAK_HBM_GBS, classes=class_table(agg, n_inst)), f, indent=1)
    if rank == 0:
        print(json.dumps({
            'metric': 'images/sec/node ViT-B/16 MAE pre-training 224x224 bs=128/GPU (secondary, SURVEY 8f-4)',
            'value': round(world * MB * args.steps / elapsed, 2), 'unit': 'images/sec', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 3),
            'ms_per_step_median': round(ms_median, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': 'ViT-B/16 MAE pre-training step (random masking 0.75, encoder 12x768 on 50 tokens, '
                                   'decoder 8x512 on 197 tokens, masked MSE on normalised patches, bwd, grad all-reduce, '
                                   'AdamW), images [128,3,224,224]/GPU, random init, autocast dtypes',
                       'global_batch': world * MB, 'parallelism': f'dp{world}'},
            'loss': loss, 'roofline': roofline}), flush=True)
    trainer.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def bench_decode(args, rank, world, ops, synth):
    """Secondary metric: decode + NMS latency per image.  One "step" = RetinaDecoder / FCOSDecoder on the
    head outputs of one evaluation batch [2, 800x1333] resident in HBM -> [B,100] scores / classes / boxes on the device
    (sacv_decode_topk_nms / sacv_decode_topk_soft_nms: score scan + argmax, exact top-1000, box decode, NMS).  The ranks
    decode independent batches (no collective on this path)."""
    import torch
    import torch.distributed as dist
    from sacv_amd.detection import decode
    fcos = args.model == 'fcos_decode'
    host = synth.head_outputs(100 + rank, BATCH, H, W, fcos=fcos)
    preds = [[torch.tensor(x).cuda() for x in grp] for grp in host]
    name = 'FCOSDecoder' if fcos else 'RetinaDecoder'
    kw = {}
    if os.environ.get('SACV_DECODE_MIN_SCORE'):  # stress case: 0 = every prediction is a candidate (dense lists)
        kw['min_score_threshold'] = float(os.environ['SACV_DECODE_MIN_SCORE'])
    dec = decode.__dict__[name](nms_type=args.nms_type, nms_threshold=0.6 if fcos else 0.5, **kw)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = dec.device_call(preds)
    sync()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        out = dec.device_call(preds)
        marks[i + 1].record()
    sync()
    elapsed = time.perf_counter() - t0
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    ms_median = per_step[len(per_step) // 2]
    if world > 1:
        t = torch.tensor([elapsed], device='cuda', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    kept = int((out[1] >= 0).sum())
    # algorithmic bytes of the call: every class score and box regression (and centerness) read once (the [B,100] results
    # are noise beside them).  The call is a chain of launches (scan, radix select + sort, NMS): the rate below is over the
    # WHOLE call on the launch stream, so it prices the latency-bound selection and NMS stages against the HBM peak too;
    # the per-kernel split is in profiles/ (rocprofv3 --kernel-trace of this command).
    nbytes = float(sum(x.numel() * 4 for grp in preds for x in grp))
    gbs = nbytes / (ms_median * 1e-3) / 1e9
    roofline = dict(kernel='decode_kernel + select_sort_kernel + ' + ('soft_nms_kernel' if 'soft' in args.nms_type else 'nms_kernel')
                           + ' (one C-ABI call; HIP events around the call on the launch stream, median)',
                    bound='hbm', achieved=round(gbs, 1), peak=PEAK_HBM_GBS, unit='GB/s', frac=round(gbs / PEAK_HBM_GBS, 4),
                    traffic=None, algorithmic_mb_per_launch=round(nbytes / 1e6, 1), avg_launch_ms=round(ms_median, 4))
    line = {
        'metric': f'decode+NMS latency per image, {name} {args.nms_type}, {H}x{W} bs={BATCH}/GPU (secondary, SURVEY 8d)',
        'value': round(elapsed / args.steps * 1e3 / BATCH, 4), 'unit': 'ms/image', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 4),
        'ms_per_step_median': round(ms_median, 4), 'higher_is_better': False, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': f'{name}(nms_type={args.nms_type}) on the head outputs of one batch [{BATCH},3,{H},{W}]: '
                               f'{sum(x[0].numel() // x.shape[-1] for x in preds[0])} predictions x 80 classes per '
                               'image, 1 in 29 above the 0.05 threshold (6 9xx candidates per image), top-1000 -> NMS -> '
                               f'100; outputs stay on the device; {kept} detections kept in the batch',
                   'global_batch': world * BATCH, 'parallelism': f'replicas{world}'},
        'roofline': roofline}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import statistics
        from oracle import decode as odecode  # the checker, timed as the CPU baseline
        fn = odecode.fcos_decode if fcos else odecode.retina_decode
        times = []
        for it in range(4):
            c0 = time.perf_counter()
            ref = fn(*host, nms_type=args.nms_type, nms_threshold=0.6 if fcos else 0.5, **kw)
            times.append(time.perf_counter() - c0)
        med = statistics.median(times[1:])
        same = all(bool((a.cpu().numpy() == b).all()) for a, b in zip(out, ref))
        line['cpu_baseline'] = dict(value=round(med * 1e3 / BATCH, 2), unit='ms/image', cores=1, kind='port',
                                    sample=f'median of 3 runs (after 1 warm-up) of oracle/decode.py on the same batch '
                                           f'(NumPy, one thread); GPU result identical to it: {same}')
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def bench_infer(args, rank, world, ops, synth):
    """Secondary workload: evaluation / inference throughput -- model.eval() forward of a batch [2,3,H,W] under no_grad
    (BatchNorm folded into the convolution epilogues, im2col-free stem) + the decoder on the device; what one iteration of
    the reference's evaluate_coco_detection does per batch (tools/scripts.py:725-760) minus the host post-steps."""
    import torch
    import torch.distributed as dist
    from sacv_amd.detection import models, decode
    fcos = args.model == 'fcos_infer'
    net = models.__dict__['resnet50_fcos' if fcos else 'resnet50_retinanet']().cuda().eval()
    dec = decode.__dict__['FCOSDecoder' if fcos else 'RetinaDecoder'](nms_type=args.nms_type)
    images = torch.tensor(synth.images(rank, BATCH, H, W)).cuda()

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for _ in range(args.warmup):
            out = dec.device_call(net(images))
        sync()
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        t0 = time.perf_counter()
        marks[0].record()
        for i in range(args.steps):
            out = dec.device_call(net(images))
            marks[i + 1].record()
        sync()
    elapsed = time.perf_counter() - t0
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    if world > 1:
        t = torch.tensor([elapsed], device='cuda', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    if rank == 0:
        print(json.dumps({
            'metric': f'images/sec/node {"FCOS" if fcos else "RetinaNet"}-R50 evaluation forward + decode/NMS {W}x{H} '
                      f'bs={BATCH}/GPU (secondary)',
            'value': round(world * BATCH * args.steps / elapsed, 2), 'unit': 'images/sec', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 3),
            'ms_per_step_median': round(per_step[len(per_step) // 2], 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': f'model.eval() forward + {type(dec).__name__}({args.nms_type}) on the device, images '
                                   f'[{BATCH},3,{H},{W}]/GPU, random init (scores sit at the 0.01 prior: the decoder '
                                   'scans every prediction and selects none; its loaded latency is the *_decode workload)',
                       'global_batch': world * BATCH, 'parallelism': f'replicas{world}'},
            'roofline': None}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--model', default='retinanet', choices=['retinanet', 'fcos', 'resnet50_cls', 'vit_mae', 'retinanet_decode', 'fcos_decode', 'retinanet_infer',
                             'fcos_infer'],
                    help='retinanet = the headline metric; fcos = secondary workload (BASELINE config 4); '
                         'resnet50_cls = ResNet50 ImageNet classification, bs 256/GPU, 224x224 (SURVEY.md 8f-1)')
    ap.add_argument('--nms-type', default='python_nms',
                    help='*_decode workloads: python_nms | diou_python_nms | torch_nms | soft_nms | soft_nms_gaussian')
    ap.add_argument('--no-kernel-timer', action='store_true')
    ap.add_argument('--timer-every', type=int, default=20,
                    help='per-launch HIP events (the roofline measurement) on every Nth timed step.  An instrumented '
                         'step costs ~3 ms more than a plain one: 440 event records (~2.7 us each) and, since round 3, '
                         'the weight-gradient side stream is folded back into the launch stream for that step so that '
                         'every timed kernel has the chip to itself (its own duration, not a shared one)')
    ap.add_argument('--image-size', default='800x1333',
                    help='HxW of the synthetic batch: 800x1333 = the headline shape S1, 1333x1333 = the collater\'s '
                         'padded square S2 (SURVEY.md 8d)')
    ap.add_argument('--class-table', default='',
                    help='write the per-conv-class roofline table (MFMA and HBM fractions) to this JSON file')
    ap.add_argument('--emulate-rccl', default='', metavar='WORKGROUPS[:LDS_BYTES[:GBPS]]',
                    help='one-GPU measurement of the CU reserve (DESIGN.md 5.2): every exchange stage launches a stand-in '
                         'for its all-reduce kernels -- that many resident workgroups for the duration of an 8-rank ring '
                         'all-reduce at GBPS GB/s (default 200) -- on its own stream, chained like the real collective')
    ap.add_argument('--launch-plan', default=None, metavar='NAME=VALUE[,...]',
                    help='fields of the library launch plan (sacv_launch_share) for A/B runs, e.g. split_policy=0')
    ap.add_argument('--engine-set', action='append', default=[], metavar='NAME=VALUE',
                    help='A/B runs: set an attribute of the model\'s executor before the first step, e.g. '
                         '--engine-set fuse_bn_bwd=False (value: a Python literal)')
    ap.add_argument('--dtype', default='f32', choices=['f32', 'bf16'],
                    help='bf16: the autocast recipe of resnet50_cls (BASELINE config #2); the detection recipes run '
                         'fp32 (use_amp=False in their train_config.py)')
    args = ap.parse_args()
    # stdout carries the ONE JSON line of the contract and nothing else: the model constructors mirror the reference's
    # `print('no backbone pretrained model!')` (detection/models/backbones/resnet.py:122), which goes to stderr here
    import builtins
    _print = builtins.print

    def _print_json_only(*a, **k):
        if k.get('file') is None and not (len(a) == 1 and isinstance(a[0], str) and a[0].startswith('{')):
            k['file'] = sys.stderr
        return _print(*a, **k)
    builtins.print = _print_json_only
    if args.model == 'vit_mae':
        args.dtype = 'bf16'  # the MAE recipe runs under autocast (use_amp=True); there is no fp32 engine for it
    if args.dtype == 'bf16' and args.model not in ('resnet50_cls', 'vit_mae'):
        raise SystemExit('--dtype bf16 is the resnet50_cls / vit_mae recipe; the detection recipes are fp32 '
                         '(use_amp=False)')

    global H, W
    H, W = (int(v) for v in args.image_size.lower().split('x'))
    # dmabuf IPC: what RCCL / cross-process device memory needs on this driver (already exported on the pool's boxes;
    # set here too so that a rank started by any launcher has it before the HIP runtime loads)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus > 1 and 'RANK' not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher.  Nothing has touched the GPU yet (no torch.cuda
        # call, no HIP library loaded), so starting the ranks as CHILD processes is safe; this process only
        # forwards their output and exit code.
        return launch_ranks(args.gpus)
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch one process per GPU '
                         f'(torch.distributed.run --nproc-per-node {args.gpus}) or run bench.py without a launcher')
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not os.environ.get('SACV_SHARE_GPU') and torch.cuda.device_count() < world:
        raise SystemExit(f'--gpus {world}: only {torch.cuda.device_count()} GPU(s) visible (one rank per GPU; '
                         'SACV_SHARE_GPU=1 SACV_DIST_BACKEND=gloo rehearses the multi-rank path on one GPU)')
    # SACV_DIST_BACKEND=gloo + SACV_SHARE_GPU=1: rehearsal of the multi-rank path on a one-GPU box (all ranks on
    # cuda:0, collectives through the host).  The real run uses RCCL ("nccl") with one GPU per rank.
    backend = os.environ.get('SACV_DIST_BACKEND', 'nccl')
    # one GPU per rank; if the launcher restricted each rank's visibility to a single device, that device is 0
    dev_index = 0 if os.environ.get('SACV_SHARE_GPU') else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    group = None
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        kw = dict(device_id=torch.device('cuda', dev_index)) if backend == 'nccl' else {}
        dist.init_process_group(backend, init_method='env://', rank=rank, world_size=world, **kw)
        group = dist.group.WORLD

    from sacv_amd import ops, synthetic as synth  # oracle/ is imported by the cpu_baseline leg only
    from sacv_amd.detection import models, losses
    from sacv_amd.train import DetectionTrainer

    torch.manual_seed(0)
    if args.model == 'resnet50_cls':
        return bench_classification(args, rank, world, group, ops, synth)
    if args.model == 'vit_mae':
        return bench_mae(args, rank, world, group, ops, synth)
    if args.model.endswith('_decode'):
        return bench_decode(args, rank, world, ops, synth)
    if args.model.endswith('_infer'):
        return bench_infer(args, rank, world, ops, synth)
    if args.model == 'fcos':
        net = models.__dict__['resnet50_fcos'](backbone_pretrained_path='', num_classes=80).cuda().train()
        crit = losses.__dict__['FCOSLoss']()
    else:
        net = models.__dict__['resnet50_retinanet'](backbone_pretrained_path='', num_classes=80).cuda().train()
        crit = losses.__dict__['RetinaLoss'](box_loss_type='SmoothL1')
    trainer = DetectionTrainer(net, crit, lr=1e-4, weight_decay=1e-3, process_group=group,
                               transport=os.environ.get('SACV_TRANSPORT', 'torch'))
    if args.emulate_rccl:
        if world != 1:
            raise SystemExit('--emulate-rccl stands in for RCCL on ONE GPU')
        f = args.emulate_rccl.split(':')
        trainer.exchanger.emulate = dict(workgroups=int(f[0]), lds_bytes=int(f[1]) if len(f) > 1 else 16384,
                                         gbps=float(f[2]) if len(f) > 2 else 200.0)
    apply_engine_settings(args, net)
    apply_launch_plan(args, ops)
    images = torch.tensor(synth.images(rank, BATCH, H, W)).cuda()
    annots = torch.tensor(synth.annotations(rank, BATCH, H, W, boxes_per_image=8)).cuda()

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = trainer.step(images, annots)
    timer = None
    if not args.no_kernel_timer:
        timer = ops.KernelTimer()
    sync()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        ops.TIMER = timer if (timer is not None and i % args.timer_every == 0) else None
        out = trainer.step(images, annots)
        marks[i + 1].record()  # an event on the launch stream: no synchronisation inside the timed region
    sync()
    elapsed = time.perf_counter() - t0
    ops.TIMER = None
    n_inst = len(range(0, args.steps, args.timer_every))  # timed steps that carried per-launch events
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    ms_median = per_step[len(per_step) // 2] if len(per_step) % 2 else 0.5 * (per_step[len(per_step) // 2 - 1] +
                                                                                per_step[len(per_step) // 2])
    losses_host = trainer.finish_step(out)
    if world > 1:
        t = torch.tensor([elapsed], device='cuda', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    ms_per_step = elapsed / args.steps * 1e3
    value = world * BATCH * args.steps / elapsed

    roofline = None
    if timer is not None:
        agg = timer.summary()
        if os.environ.get('SACV_BENCH_DUMP') and rank == 0:  # per-launch table of the last step (tuning aid)
            recs = timer.records[-(len(timer.records) // n_inst):]
            with open(os.environ['SACV_BENCH_DUMP'], 'w') as f:
                for key, flops, e0, e1 in recs:
                    ms = e0.elapsed_time(e1)
                    f.write(f'{key} {flops/1e9:.3f} GFLOP {ms*1e3:.1f} us {flops/(ms*1e-3)/1e12:.1f} TF\n')
        k = family_roofline(agg, 'conv_nt')  # keys carry a per-shape suffix after '|'
        if k:
            per_launch_flops = k['flops'] / k['launches']
            per_launch_ms = k['ms'] / k['launches']
            # HBM traffic per launch of the same kernel family from the PMC passes committed under profiles/
            # (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, gfx950 read-doubling applied there)
            traffic = None
            for tname in ('r04_pmc_traffic.json', 'r03_pmc_traffic.json', 'r02_pmc_traffic.json', 'r01_pmc_traffic.json'):
                tpath = os.path.join(ROOT, 'profiles', tname)
                if os.path.exists(tpath) and args.model == 'retinanet' and (H, W) == (800, 1333):
                    traffic = round(json.load(open(tpath))['hbm_bytes_per_launch'])
                    break
            tn = family_roofline(agg, 'conv_tn')
            secondary = None
            if tn:
                secondary = dict(kernel='weight gradient: conv_tn3_kernel + wgrad_reduce (direct), wino_wgrad3_kernel + transforms '
                                        '(Winograd: head 3x3 convolutions and the trunk\'s >= 128-channel ones); achieved = executed '
                                        'matrix-core flops / time, effective_achieved = algorithmic fp32 flops / time', bound='mfma',
                                 pipe=tn['pipe'], achieved=round(tn['achieved'], 2), peak=tn['peak'], unit='TFLOP/s',
                                 frac=round(tn['frac'], 4), effective_achieved=round(tn['effective'], 2),
                                 effective_peak=round(tn['effective_peak'], 1),
                                 effective_frac=round(tn['effective'] / tn['effective_peak'], 4),
                                 launches_per_step=tn['launches'] // n_inst,
                                 share_of_step_ms=round(tn['ms'] / n_inst, 2))
            wino = family_roofline(agg, 'conv_nt_wino')
            paths = None
            if wino:
                d_fl, d_ms = k['flops'] - wino['flops'], k['ms'] - wino['ms']
                paths = dict(
                    winograd=dict(calls_per_step=wino['launches'] // n_inst, pipe=wino['pipe'],
                                  algorithmic_tflops=round(wino['effective'], 2),
                                  matrix_core_tflops=round(wino['achieved'], 2), matrix_core_frac=round(wino['frac'], 4),
                                  share_of_step_ms=round(wino['ms'] / n_inst, 2),
                                  note='F(2x2,3x3): 2.25x fewer matrix-core flops than the direct sum; time includes '
                                       'the input transform'),
                    direct=dict(calls_per_step=(k['launches'] - wino['launches']) // n_inst,
                                algorithmic_tflops=round(d_fl / (d_ms * 1e-3) / 1e12, 2),
                                matrix_core_frac=round((k['peak_ms'] - wino['peak_ms']) / d_ms, 4),
                                share_of_step_ms=round(d_ms / n_inst, 2)))
            # `achieved` / `frac` price the flops the matrix cores EXECUTE on the pipe they run on (family_roofline):
            # matrix-core utilisation.  The algorithmic (direct-convolution, fp32) rate of the same launches -- what the
            # step's throughput follows -- is reported beside it as effective_*, against the ceiling of that pipe for
            # fp32 products (bf16 peak / 6) and, for orientation, against the fp32 matrix pipe's peak.
            alg_bytes = k['bytes'] / k['launches']
            roofline = dict(kernel='convolution forward + data gradient: conv_nt_kernel<128,64,mode,...,SP> (implicit GEMM) and, '
                                   'for the stride-1 3x3 head / P3 convolutions, wino_gemm_kernel<{128x128|64x64},split> '
                                   '(Winograd F(2x2,3x3), 16 chained GEMMs + output transform); fp32 operands as three bf16 '
                                   'terms, six v_mfma_f32_32x32x16_bf16 products per fp32 product, fp32 accumulators; '
                                   '`achieved` = flops EXECUTED on the matrix cores / time (utilisation); effective_achieved '
                                   '= algorithmic (direct-convolution fp32) flops / time',
                            bound='mfma', pipe=k['pipe'], achieved=round(k['achieved'], 2), peak=k['peak'], unit='TFLOP/s',
                            frac=round(k['frac'], 4),
                            effective_achieved=round(k['effective'], 2), effective_peak=round(k['effective_peak'], 1),
                            effective_frac=round(k['effective'] / k['effective_peak'], 4),
                            vs_fp32_matrix_peak=round(k['effective'] / PEAK_FP32_MFMA_TFLOPS, 3),
                            traffic=traffic,
                            algorithmic_bytes_per_launch=round(alg_bytes),
                            traffic_ratio=round(traffic / alg_bytes, 2) if (traffic and alg_bytes) else None,
                            launches_per_step=k['launches'] // n_inst,
                            avg_launch_ms=round(per_launch_ms, 4), algorithmic_gflop_per_launch=round(
                                per_launch_flops / 1e9, 2),
                            share_of_step_ms=round(k['ms'] / n_inst, 2))
            roofline['instrumented_steps'] = (f'{n_inst} of the {args.steps} timed steps (every {args.timer_every}th), run '
                                              'in stream order: the other steps overlap the weight gradients with the '
                                              'data-gradient chain on a side stream (ms_per_step_median = such a step)')
            if paths:
                roofline['paths'] = paths
            if secondary:
                roofline['second_kernel'] = secondary
        if args.class_table and rank == 0:
            with open(args.class_table, 'w') as f:
                json.dump(dict(model=args.model, image_size=[H, W], batch=BATCH, steps=args.steps,
                               peak_tflops_fp32_mfma=PEAK_FP32_MFMA_TFLOPS, peak_hbm_gbs=PEAK_HBM_GBS,
                               classes=class_table(agg, n_inst)), f, indent=1)
    if rank == 0:
        line = {
            'metric': 'images/sec/node RetinaNet-R50 COCO 1333x800 bs=2/GPU',
            'value': round(value, 3), 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 3), 'ms_per_step_median': round(ms_median, 3),
            'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'RetinaNet-ResNet50 COCO training step (fwd + RetinaLoss SmoothL1 + bwd + '
                                   f'grad all-reduce + AdamW), images [2,3,{H},{W}]/GPU, 8 GT boxes/image, '
                                   'random init', 'global_batch': world * BATCH, 'parallelism': f'dp{world}',
                       'matrix_arithmetic': ('fp32 tensors, fp32 accumulators; every matrix product of the convolutions is '
                                             'formed on the bf16 matrix pipe from operands split EXACTLY into three bf16 terms '
                                             '(x = hi + mid + lo), six partial products per fp32 product -- the error against '
                                             'fp64 is that of the fp32 matrix pipe (tests/test_gpu_kernels.py, '
                                             'test_gpu_winograd.py; scripts/diag_conv_split.py)')
                       if getattr(net.engine, 'conv_split', False) else 'fp32 matrix pipe'},
            'loss': losses_host,
            'roofline': roofline,
        }
        headline = args.model == 'retinanet' and (H, W) == (800, 1333)
        if not headline:
            name = 'FCOS-R50' if args.model == 'fcos' else 'RetinaNet-R50'
            line['metric'] = f'images/sec/node {name} COCO {W}x{H} bs=2/GPU (secondary)'
        if args.model == 'fcos':
            line['config']['workload'] = line['config']['workload'].replace('RetinaNet-ResNet50', 'FCOS-ResNet50') \
                .replace('RetinaLoss SmoothL1', 'FCOSLoss GIoU')
        if world == 1 and not args.no_cpu_baseline and args.model == 'retinanet' and (H, W) == (800, 1333):
            line['cpu_baseline'] = cpu_baseline()
        print(json.dumps(line), flush=True)
    trainer.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
