"""Step driver against the reference's own training loop: tests/golden/trainer.json holds what
tools/scripts.py::train_detection's body does for a few iterations with the reference model, RetinaLoss,
build_optimizer's AdamW and Scheduler (oracle/make_golden.py::sec_trainer) -- plain, with gradient accumulation +
clip_grad_value_/clip_grad_norm_, and with a batch the inf/nan input guard must skip.  Plus: checkpoint state in
torch.optim's layout, the guards' kernels, and the stale-arena check."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import model as omodel, synth
from tests import golden_inputs as gi


class _Cfg:
    pass


def _module_of(name):
    parts = name.split('.')
    return '.'.join(parts[:2]) if parts[0] == 'backbone' else parts[0]


def _build(H, W, accumulation_steps=1, clip_grad_value=0.0, clip_max_norm=0.0,
           sched=('MultiStepLR', {'warm_up_epochs': 0, 'gamma': 0.1, 'milestones': [1, 2]}), epochs=3):
    from sacv_amd.detection import models, losses
    from sacv_amd.utils import Scheduler, build_trainer
    net = models.__dict__['resnet50_retinanet'](num_classes=80)
    net.load_state_dict(omodel.init_retinanet_state(0))
    net = net.cuda().train()
    cfg = _Cfg()
    cfg.optimizer = ('AdamW', {'lr': 1e-4, 'global_weight_decay': False, 'weight_decay': 1e-3,
                               'no_weight_decay_layer_name_list': []})
    cfg.scheduler = sched
    cfg.epochs = epochs
    cfg.accumulation_steps = accumulation_steps
    cfg.clip_grad_value = clip_grad_value
    cfg.clip_max_norm = clip_max_norm
    trainer = build_trainer(cfg, net, losses.__dict__['RetinaLoss']())
    return net, trainer, Scheduler(cfg, trainer)


def _run(rec, H, W, n_iters, iters_per_epoch=2, bad_iter=None, loss_tol=(1e-4, 3e-3), **kw):
    """Drive trainer + Scheduler the way train_detection does and compare with the reference record."""
    net, trainer, scheduler = _build(H, W, **kw)
    assert [len(g['param_names']) for g in trainer.param_groups] == [g['n'] for g in rec['groups']]
    assert [g['weight_decay'] for g in trainer.param_groups] == [g['weight_decay'] for g in rec['groups']]
    A = trainer.accumulation_steps
    iter_index, epoch = 1, 1
    norms = []
    for data_i in range(n_iters):
        images = torch.tensor(synth.images(40 + data_i, 2, H, W)).cuda()
        annots = torch.tensor(synth.annotations(60 + data_i, 2, H, W, boxes_per_image=6))
        if bad_iter is not None and data_i == bad_iter:
            annots[0, 0, 0] = float('inf')
        want = rec['iters'][data_i]
        assert want['iter_index'] == iter_index
        fin = trainer.finish_step(trainer.step(images, annots.cuda()))
        assert fin['skipped'] == want['skipped'], (data_i, fin)
        if want['skipped']:
            continue  # the reference `continue`s before iter_index += 1 and before scheduler.step
        # iteration 0 starts from identical weights; later ones from weights that already differ by rounding-level
        # Adam noise (elements whose gradient is noise move by +-lr in both implementations)
        tol = loss_tol[0] if data_i == 0 else loss_tol[1]
        for k, v in want['loss'].items():
            assert abs(fin[k] - v) <= tol * max(1.0, abs(v)), (data_i, k, fin[k], v)
        if iter_index % A == 0:
            if 'grad_norm' in want:
                norms.append((float(trainer.state[5]), want['grad_norm']))
            scheduler.step(trainer, iter_index / iters_per_epoch + (epoch - 1))
            assert [g['lr'] for g in trainer.param_groups] == want['lr_after'], (data_i, want['lr_after'])
        if iter_index == iters_per_epoch:
            iter_index, epoch = 0, epoch + 1
            trainer.start_epoch()
        iter_index += 1
    fin = trainer.finish_step({'cls_loss': torch.zeros((), device='cuda'), 'reg_loss': torch.zeros((), device='cuda')})
    assert [float(fin['optimizer_steps'])] == rec['opt_steps']
    mods, dmods = {}, {}
    init = omodel.init_retinanet_state(0)
    for k, p in net.named_parameters():
        pc = p.detach().double().cpu()
        mods[_module_of(k)] = mods.get(_module_of(k), 0.0) + float(pc.pow(2).sum())
        dmods[_module_of(k)] = dmods.get(_module_of(k), 0.0) + float((pc - init[k].double()).pow(2).sum())
    for k, v in rec['module_param_norm'].items():
        assert abs(np.sqrt(mods[k]) - v) <= 1e-5 * v, (k, np.sqrt(mods[k]), v)
    for k, v in rec['module_delta_norm'].items():  # how far the optimizer moved the module: lr / clip / skip sensitive
        assert abs(np.sqrt(dmods[k]) - v) <= 3e-2 * v, (k, np.sqrt(dmods[k]), v)
    return net, trainer, norms


@pytest.mark.parametrize('case,shape', [('plain3', 'TINY'), ('plain3_MID', 'MID')])
def test_three_reference_iterations_with_the_lr_schedule(golden_dir, case, shape):
    rec = json.load(open(os.path.join(golden_dir, 'trainer.json')))[case]
    H, W = gi.SHAPES[shape]
    net, trainer, _ = _run(rec, H, W, 3)
    if shape == 'MID':  # Adam moments vs the reference optimizer's state (gradient-level agreement)
        rel = []
        for n in trainer.param_groups[0]['param_names'] + trainer.param_groups[1]['param_names']:
            o, p = net.arena.offsets[n], net.arena.module_params[n]
            m = float(trainer.exp_avg[o:o + p.numel()].double().norm())
            want = rec['opt_state_norm'][n][0]
            rel.append(abs(m - want) / (want + 1e-20))
        assert float(np.median(rel)) <= 3e-2 and max(rel) <= 0.3, (float(np.median(rel)), max(rel))


def test_gradient_accumulation_and_clipping_like_the_reference(golden_dir):
    """accumulation_steps 2 (the first micro-batch of a window is not exchanged and not stepped: DDP.no_sync,
    tools/scripts.py:886-949), clip_grad_value_ 0.05 then clip_grad_norm_ 0.5 (:978-990), CosineLR with warm-up."""
    rec = json.load(open(os.path.join(golden_dir, 'trainer.json')))['accum2_clip']
    H, W = gi.TINY
    _, trainer, norms = _run(rec, H, W, 4, iters_per_epoch=4, accumulation_steps=2, clip_grad_value=0.05,
                             clip_max_norm=0.5, sched=('CosineLR', {'warm_up_epochs': 1, 'min_lr': 1e-6}), epochs=3)
    assert len(norms) == 2
    for got, want in norms:  # total norm of the value-clipped accumulated gradient (what clip_grad_norm_ returns)
        assert abs(got - want) <= 3e-2 * want, (got, want)
    assert float(trainer.state[4]) < 1.0  # the norm clip was active


def test_inf_in_the_annotations_skips_the_batch(golden_dir):
    """tools/scripts.py:878-884: an inf/nan anywhere in images or targets skips the batch even when the loss is
    finite (here the inf box never reaches the loss)."""
    rec = json.load(open(os.path.join(golden_dir, 'trainer.json')))['skip_inf_annot']
    H, W = gi.TINY
    _run(rec, H, W, 3, bad_iter=1)


def test_checkpoint_state_roundtrip_and_torch_optim_layout():
    """state_dict() -> load_state_dict() into a fresh trainer -> the next step is bit-identical; the dict is a valid
    torch.optim.AdamW state_dict (the reference resumes with optimizer.load_state_dict,
    tools/train_detection_model.py:252-267) and torch's own step from it matches the fused kernel."""
    H, W = gi.TINY
    net, tr, sch = _build(H, W)
    batches = [(torch.tensor(synth.images(70 + i, 2, H, W)).cuda(),
                torch.tensor(synth.annotations(80 + i, 2, H, W, boxes_per_image=6)).cuda()) for i in range(3)]
    for i in range(2):
        tr.step(*batches[i])
        sch.step(tr, (i + 1) / 2)
    sd = tr.state_dict()
    model_sd = {k: v.clone() for k, v in net.state_dict().items()}
    assert len(sd['state']) == 195 and [len(g['params']) for g in sd['param_groups']] == [124, 71]
    assert float(sd['state'][0]['step']) == 2.0 and sd['param_groups'][0]['lr'] == tr.param_groups[0]['lr'] == 1e-5
    w = 'backbone.layer1.0.conv2.layer.0.weight'
    idx = 124 + tr.param_groups[1]['param_names'].index(w)
    assert tuple(sd['state'][idx]['exp_avg'].shape) == (64, 64, 3, 3) and sd['state'][idx]['exp_avg'].is_contiguous()
    # (1) torch.optim.AdamW accepts it and continues identically (same gradient fed to both)
    cpu_params = [[net.arena.module_params[n].detach().cpu().clone().contiguous().requires_grad_(True)
                   for n in g['param_names']] for g in tr.param_groups]
    opt = torch.optim.AdamW([{'params': cpu_params[0], 'weight_decay': 0.0},
                             {'params': cpu_params[1], 'weight_decay': 1e-3}], lr=1e-4)
    import copy
    opt.load_state_dict(copy.deepcopy({'state': sd['state'], 'param_groups': sd['param_groups']}))  # torch steps 'step' in place
    assert opt.param_groups[0]['lr'] == 1e-5
    tr.step(*batches[2])
    for plist, g in zip(cpu_params, tr.param_groups):
        for p, n in zip(plist, g['param_names']):
            p.grad = net.arena.grad_views[n].detach().cpu().clone().contiguous()
    opt.step()
    worst = 0.0
    for plist, g in zip(cpu_params, tr.param_groups):
        for p, n in zip(plist, g['param_names']):
            worst = max(worst, float((p.detach() - net.arena.module_params[n].detach().cpu()).abs().max()))
    assert worst <= 1.5e-7, worst  # 1 ulp of a weight of magnitude ~1 (fused multiply-add contraction differs)
    after3 = net.arena.params.clone()
    # (2) a fresh model + trainer resumed from the checkpoint takes a bit-identical third step
    net2, tr2, sch2 = _build(H, W)
    net2.load_state_dict(model_sd)
    tr2.load_state_dict(sd)
    assert tr2.param_groups[1]['lr'] == 1e-5 and float(tr2.state[0]) == 2.0
    tr2.step(*batches[2])
    assert torch.equal(net2.arena.params, after3)
    assert torch.equal(tr2.exp_avg, tr.exp_avg) and torch.equal(tr2.exp_avg_sq, tr.exp_avg_sq)


def test_trainer_refuses_a_stale_arena():
    """a model that re-allocates its arena (deepcopy for EMA, .to()) must not silently train on buffers the
    trainer no longer owns."""
    H, W = gi.TINY
    net, tr, _ = _build(H, W)
    net.prepare()  # what a re-allocated parameter triggers lazily
    with pytest.raises(RuntimeError, match='re-allocated'):
        tr.step(torch.tensor(synth.images(1, 2, H, W)).cuda(),
                torch.tensor(synth.annotations(2, 2, H, W, boxes_per_image=4)).cuda())


def test_guard_and_clip_kernels_vs_torch():
    from sacv_amd import ops
    g = torch.Generator().manual_seed(0)
    for n in (4, 1000, 1 << 20):
        x = torch.randn(n, generator=g).cuda()
        flag = torch.zeros(1, device='cuda')
        assert float(ops.nonfinite_flag(x, flag)) == 0.0
        for bad, nan_only, want in ((float('inf'), False, 1.0), (float('nan'), False, 1.0), (float('-inf'), True, 0.0),
                                    (float('nan'), True, 1.0)):
            y = x.clone()
            y[n - 1 if n % 2 else n // 3] = bad
            flag.zero_()
            assert float(ops.nonfinite_flag(y, flag, nan_only=nan_only)) == want, (n, bad, nan_only)
    cl = torch.randn(2, 3, 16, 12, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    cl[1, 2, 15, 11] = float('nan')
    flag = torch.zeros(1, device='cuda')
    assert float(ops.nonfinite_flag(cl, flag)) == 1.0
    # clip coefficient = torch's clip_grad_value_ followed by clip_grad_norm_
    n = 1 << 18
    gr = (torch.randn(n, generator=g) * 0.3)
    state = torch.zeros(ops.STATE_FLOATS, device='cuda')
    for cv, mn, scale in ((0.0, 0.0, 1.0), (0.2, 0.0, 1.0), (0.0, 1.5, 0.5), (0.25, 3.0, 0.5)):
        ref = (gr * scale).clone()
        if cv > 0:
            ref.clamp_(-cv, cv)
        norm = float(ref.double().norm())
        coef = min(1.0, mn / (norm + 1e-6)) if mn > 0 else 1.0
        ops.grad_clip_coef(gr.cuda(), state, scale, cv, mn)
        assert abs(float(state[5]) - norm) <= 1e-5 * norm and abs(float(state[4]) - coef) <= 1e-6, (cv, mn)
        # the gated AdamW applies clamp * coefficient
        p = torch.randn(n, generator=g).cuda()
        p0 = p.clone()
        m = torch.zeros(n, device='cuda')
        v = torch.zeros(n, device='cuda')
        st = state.clone()
        st[0:4] = torch.tensor([1.0, 0.0, 0.1, 0.001**0.5])
        ops.adamw_step_gated(p, gr.cuda(), m, v, 1024, 1e-3, 0.9, 0.999, 1e-8, 0.0, 0.01, st, grad_scale=scale,
                             clip_value=cv, lr_hi=2e-3)
        geff = (ref * coef).cuda()
        torch.testing.assert_close(m, 0.1 * geff, rtol=1e-6, atol=1e-9)
        lr = torch.full((n,), 2e-3, device='cuda')
        lr[:1024] = 1e-3
        wd = torch.full((n,), 0.01, device='cuda')
        wd[:1024] = 0.0
        want = p0 * (1 - lr * wd) - lr / 0.1 * (0.1 * geff) / ((0.001 * geff * geff).sqrt() / 0.001**0.5 + 1e-8)
        torch.testing.assert_close(p, want, rtol=1e-5, atol=1e-7)


def _small_models():
    from sacv_amd.detection import models
    from sacv_amd.classification import backbones
    from sacv_amd.masked_image_modeling import models as mm
    yield 'resnet18_fcos', lambda: models.__dict__['resnet18_fcos']()
    yield 'resnet18cifar', lambda: backbones.__dict__['resnet18cifar'](num_classes=10)
    yield 'mae_tiny', lambda: mm.VITMAEPretrainModel(patch_size=16, image_size=64, encoder_embedding_planes=128,
                                                     encoder_block_nums=2, encoder_head_nums=4,
                                                     decoder_embedding_planes=64, decoder_block_nums=1,
                                                     decoder_head_nums=4)


@pytest.mark.parametrize('name', ['resnet18_fcos', 'resnet18cifar', 'mae_tiny'])
def test_ema_model_like_the_reference(name):
    """tools/utils.py:136-163 (EmaModel) + its call site tools/scripts.py:1013-1015: `copy.deepcopy(model)` works on the
    HIP models (the copy owns fresh arenas and a fresh executor, the original is untouched), and `update` -- two launches
    over the flat blocks -- leaves EVERY state_dict entry (parameters, BatchNorm statistics, the integer
    num_batches_tracked, frozen tables) bit-identical to the reference's per-tensor statement evaluated by torch on the
    CPU, over three updates with moving model weights."""
    import copy
    from sacv_amd.utils import EmaModel
    make = dict(_small_models())[name]
    torch.manual_seed(3)
    net = make().cuda().train()
    ema = EmaModel(net, decay=0.9999)
    assert not ema.ema_model.training and ema.ema_model is not net
    a, b = net.state_dict(), ema.ema_model.state_dict()
    assert list(a) == list(b) and all(torch.equal(a[k], b[k]) and a[k].data_ptr() != b[k].data_ptr() for k in a)
    ref = {k: v.detach().cpu().clone() for k, v in b.items()}
    g = torch.Generator().manual_seed(7)
    for it in range(3):
        with torch.no_grad():  # the optimizer and the BatchNorm layers move the model
            for k, v in net.state_dict().items():
                if v.dtype.is_floating_point:
                    v.add_(torch.randn(v.shape, generator=g).to(v.device) * 0.05)
                else:
                    v.add_(3 + it)
        ema.update(net)
        for k, mv in net.state_dict().items():
            e, m = ref[k], mv.detach().cpu()
            ref[k] = e.copy_(0.9999 * e + (1. - 0.9999) * m)  # the reference's update_fn + copy_
    got = ema.ema_model.state_dict()
    for k in ref:
        assert torch.equal(got[k].cpu(), ref[k]), k
    # the copy is a working model of its own
    if name == 'resnet18cifar':
        with torch.no_grad():
            y = ema.ema_model(torch.rand(2, 3, 32, 32).cuda())
        assert tuple(y.shape) == (2, 10) and bool(torch.isfinite(y).all())
    c2 = copy.deepcopy(ema.ema_model)
    assert all(torch.equal(u, v) for u, v in zip(c2.state_dict().values(), got.values()))


def test_trainer_updates_the_ema_model_and_respects_the_skip_gate():
    """build_trainer with config.use_ema_model (tools/utils.py:172-177): the EMA model follows every optimizer step
    (tools/scripts.py:1013-1015) and stays put on a skipped batch (zero loss: the reference `continue`s at :968-974)."""
    from sacv_amd.detection import models, losses
    from sacv_amd import utils as U

    class Cfg:
        optimizer = ('AdamW', {'lr': 1e-3, 'weight_decay': 1e-3, 'global_weight_decay': False})
        use_ema_model, ema_model_decay = True, 0.99
    H, W = gi.TINY
    net = models.__dict__['resnet18_retinanet']().cuda().train()
    cfg = Cfg()
    tr = U.build_trainer(cfg, net, losses.__dict__['RetinaLoss']())
    assert tr.ema_model is cfg.ema_model and isinstance(cfg.ema_model, U.EmaModel)
    images = torch.tensor(synth.images(1, 2, H, W)).cuda()
    boxes = torch.tensor(synth.annotations(2, 2, H, W, boxes_per_image=4)).cuda()
    empty = torch.tensor(synth.annotations(2, 2, H, W, boxes_per_image=0)).cuda()
    e0 = {k: v.clone() for k, v in cfg.ema_model.state_dict().items()}
    p0 = {k: v.clone() for k, v in net.state_dict().items()}
    fin = tr.finish_step(tr.step(images, boxes))
    assert not fin['skipped']
    e1 = {k: v.clone() for k, v in cfg.ema_model.state_dict().items()}
    for k, v in net.state_dict().items():  # ema_1 = d e_0 + (1 - d) p_1, the reference's statement on the CPU
        want = 0.99 * e0[k].cpu() + (1. - 0.99) * v.cpu()
        assert torch.equal(e1[k].cpu(), e0[k].cpu().copy_(want)), k
    assert any(not torch.equal(e1[k], e0[k]) for k in e1) and any(not torch.equal(p0[k], v) for k, v in net.state_dict().items())
    fin = tr.finish_step(tr.step(images, empty))  # no positive anchor: zero loss -> skipped
    assert fin['skipped']
    for k, v in cfg.ema_model.state_dict().items():
        assert torch.equal(v, e1[k]), k


def test_models_survive_pickle():
    """torch.save(model) / pickle: the runtime state (arenas, executor, HIP streams) stays behind and is rebuilt."""
    import io
    for name, make in _small_models():
        net = make().cuda().eval()
        buf = io.BytesIO()
        torch.save(net, buf)
        buf.seek(0)
        back = torch.load(buf, weights_only=False)  # (a file this test wrote itself)
        a, b = net.state_dict(), back.state_dict()
        assert list(a) == list(b) and all(torch.equal(a[k], b[k]) for k in a), name
        if name == 'resnet18cifar':
            x = torch.rand(2, 3, 32, 32).cuda()
            with torch.no_grad():
                assert torch.equal(net(x), back(x))


def test_macs_params_probe_and_sync_bn_refusal():
    """compute_macs_and_params (tools/utils.py:116-133) counted at the launch sites: RetinaNet-R50 at 800 x 1333 is the
    249.5 GMAC / 37.97 M parameters SURVEY.md 8d measured with the reference's hooks (conv MACs); sync_bn is refused."""
    from sacv_amd.detection import models
    from sacv_amd import utils as U

    class Cfg:
        input_image_size = [800, 1333]
    net = models.__dict__['resnet50_retinanet']().cuda().train()
    macs, params = U.compute_macs_and_params(Cfg(), net)
    assert net.training and macs.endswith('G') and params.endswith('M')
    assert abs(float(macs[:-1]) - 249.5) < 1.0 and abs(float(params[:-1]) - 37.969) < 0.01, (macs, params)
    conv = torch.nn.SyncBatchNorm.convert_sync_batchnorm(models.__dict__['resnet18_retinanet']())
    with pytest.raises(NotImplementedError, match='sync_bn'):
        conv.cuda()


def test_torch_compile_wrapper_is_harmless():
    """config.use_compile = True (tools/train_detection_model.py:176-178 wraps the model in torch.compile): the forward
    pass is exempt from tracing, the wrapper returns what the model returns and backward works through it."""
    from sacv_amd.detection import models, losses
    H, W = gi.TINY
    net = models.__dict__['resnet18_retinanet']().cuda().train()
    x = torch.tensor(synth.images(1, 2, H, W)).cuda()
    ann = torch.tensor(synth.annotations(2, 2, H, W, boxes_per_image=4)).cuda()
    ref = [[o.detach().clone() for o in grp] for grp in net(x)]
    comp = torch.compile(net)
    outs = comp(x)
    assert all(torch.equal(a, b) for ga, gb in zip(outs, ref) for a, b in zip(ga, gb))
    sum(losses.__dict__['RetinaLoss']()(outs, ann).values()).backward()
    assert float(net.arena.grads.abs().sum()) > 0
