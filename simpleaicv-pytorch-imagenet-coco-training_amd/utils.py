"""Host-side pieces of the training loop that sit next to the step driver (reference: tools/utils.py).

`Scheduler` mirrors tools/utils.py:188-272: a stateless function of the (fractional) epoch that is called every
iteration with `iter_index / iters + (epoch - 1)` (tools/scripts.py:1031-1033) and rewrites `param_groups[i]['lr']`
of whatever it is given -- a torch optimizer in the reference, a sacv_amd trainer here (same attribute).  It is
pure host arithmetic, written so that the floating-point operation ORDER equals the reference's and the LR sequence
is bit-identical (tests/golden/scheduler.json).
"""
import math

__all__ = ['Scheduler', 'build_trainer', 'EmaModel', 'compute_macs_and_params']


class Scheduler:

    def __init__(self, config, optimizer):
        self.scheduler_name = config.scheduler[0]
        self.scheduler_parameters = config.scheduler[1]
        self.warm_up_epochs = self.scheduler_parameters['warm_up_epochs']
        self.epochs = config.epochs
        self.optimizer_parameters = config.optimizer[1]
        self.lr = self.optimizer_parameters['lr']
        self.current_lr = self.lr
        self.init_param_groups_lr = [g['lr'] for g in optimizer.param_groups]
        assert self.scheduler_name in ['MultiStepLR', 'CosineLR', 'PolyLR'], 'Unsupported scheduler!'
        assert self.warm_up_epochs >= 0, 'Illegal warm_up_epochs!'
        assert self.epochs > 0, 'Illegal epochs!'

    def _lr_at(self, base_lr, epoch):
        """Learning rate of a group whose initial rate is base_lr at fractional epoch `epoch`."""
        sp, warm = self.scheduler_parameters, self.warm_up_epochs
        if epoch < warm:  # linear warm-up from 0
            return epoch / warm * base_lr
        if self.scheduler_name == 'MultiStepLR':
            passed = len([m for m in sp['milestones'] if m <= epoch])
            return sp['gamma']**passed * base_lr
        min_lr = sp['min_lr'] if 'min_lr' in sp else 0.
        progress = (epoch - warm) / (self.epochs - warm)
        if self.scheduler_name == 'CosineLR':
            return 0.5 * (math.cos(progress * math.pi) + 1) * (base_lr - min_lr) + min_lr
        return ((1 - progress)**sp['power']) * (base_lr - min_lr) + min_lr  # PolyLR

    def step(self, optimizer, epoch):
        assert len(self.init_param_groups_lr) == len(optimizer.param_groups)
        for group, base_lr in zip(optimizer.param_groups, self.init_param_groups_lr):
            group['lr'] = self._lr_at(base_lr, epoch)
        self.current_lr = self._lr_at(self.lr, epoch)

    def state_dict(self):
        return dict(self.__dict__)

    def load_state_dict(self, state_dict):
        self.__dict__.update(state_dict)


def compute_macs_and_params(config, model):
    """tools/utils.py:116-133 for the HIP models.  The reference profiles a CPU forward pass with thop's nn.Module hooks
    (tools/test_detection_model.py:79); these models have no torch.nn forward to hook and no CPU path, so the
    multiply-accumulates are counted where the work is issued: one evaluation pass of a [1,3,H,W] input on the GPU with a
    counter in the place of bench.py's kernel timer (every convolution / Linear launch reports its algorithmic flops;
    MACs = flops / 2 -- normalisation and activation layers, which thop also tallies, are not counted).  Returns the two
    strings of thop.clever_format('%.3f')."""
    import torch
    from . import ops
    size = config.input_image_size
    assert isinstance(size, (int, list)), 'Illegal input_image_size type!'
    h, w = (size, size) if isinstance(size, int) else (size[0], size[1])

    class _Counter:
        flops = 0.0

        def time(self, key, flops, launch, nbytes=0.0):
            self.flops += flops
            return launch()

    dev = next(model.parameters()).device
    was_training, counter, saved = model.training, _Counter(), ops.TIMER
    model.eval()
    ops.TIMER = counter
    try:
        with torch.no_grad():
            model(torch.randn(1, 3, h, w, device=dev))
    finally:
        ops.TIMER = saved
        model.train(was_training)
    params = sum(p.numel() for p in model.parameters())

    def fmt(v):  # thop.clever_format
        for unit, div in (('T', 1e12), ('G', 1e9), ('M', 1e6), ('K', 1e3)):
            if v > div:
                return '%.3f' % (v / div) + unit
        return '%.3f' % v + 'B'
    return fmt(counter.flops / 2.0), fmt(float(params))


class EmaModel:
    """tools/utils.py:136-163: an exponential moving average of the model's whole state_dict,
    ema = decay * ema + (1 - decay) * model after every optimizer step (tools/scripts.py:1013-1015).

    `ema_model` is a deep copy of the model in eval mode, as in the reference (and what its evaluation loop runs,
    scripts.py:839-840).  `update` is TWO launches over the flat blocks the models keep their state in -- the parameter
    arena and the BatchNorm statistics block -- instead of a Python loop over 354 tensors; the integer buffers
    (num_batches_tracked) and any tensor outside the blocks (frozen position tables) take the reference's own statement.
    Bit-identical to the reference's loop."""

    def __init__(self, model, decay=0.9999):
        import copy
        self.ema_model = copy.deepcopy(model)
        self.ema_model.eval()
        self.decay = decay
        self.update_fn = lambda e, m: self.decay * e + (1. - self.decay) * m

    def update(self, model, gate=None):
        """gate: a trainer's device-side step state -- when the step it describes was skipped (inf / nan / zero loss,
        the cross-rank vote) nothing moves, like the reference's `continue` ahead of its update (scripts.py:968-974)."""
        import torch
        from . import ops
        model = getattr(model, 'module', model)  # a DistributedDataParallel wrapper (tools/utils.py:178-180)
        ea, ma = self.ema_model.arena, model.arena
        with torch.no_grad():
            ops.ema_update(ea.params, ma.params, self.decay, gate)
            if ea.buffers.numel():
                ops.ema_update(ea.buffers, ma.buffers, self.decay, gate)
            esd, msd = self.ema_model.state_dict(keep_vars=True), model.state_dict(keep_vars=True)
            arena_names = {n for n, _ in ea.order} | {n for n, _ in ea.buffer_order}
            for (k, ev), mv in zip(esd.items(), msd.values()):
                assert ev.shape == mv.shape, 'wrong ema model!'
                if k not in arena_names:
                    new = self.update_fn(ev.data, mv.data)
                    if gate is not None:
                        new = torch.where(gate[1] > 0, ev.data.to(new.dtype), new)
                    ev.data.copy_(new)

    def state_dict(self):
        return self.ema_model.state_dict()

    def load_state_dict(self, sd):
        """Accepts the reference's checkpoint layout too: there ema_model.ema_model is wrapped in DistributedDataParallel
        (tools/utils.py:172-177), so the keys of latest.pth['ema_model_state_dict'] carry a 'module.' prefix."""
        if sd and all(k.startswith('module.') for k in sd):
            sd = {k[len('module.'):]: v for k, v in sd.items()}
        return self.ema_model.load_state_dict(sd)

    def reference_state_dict(self):
        """state_dict() with the 'module.' prefix of the reference's DDP-wrapped EMA model (what its train script saves)."""
        return {'module.' + k: v for k, v in self.ema_model.state_dict().items()}


def build_trainer(config, model, criterion, process_group=None, transport='torch'):
    """What build_optimizer (tools/utils.py:275-582) + build_training_mode (:166-185) set up for the step, for the
    configurations this path supports: optimizer ('AdamW' for detection, 'SGD' for classification), two weight-decay
    groups (global_weight_decay False), accumulation_steps, clip_grad_value / clip_max_norm, skip_inf_nan_grad."""
    from .train import DetectionTrainer, ClassificationTrainer, MAETrainer
    from .masked_image_modeling.models.vit_mae import VITMAEPretrainModel
    name, prm = config.optimizer
    assert name in ['SGD', 'AdamW'], 'Unsupported optimizer!'
    if prm.get('global_weight_decay', False):
        raise NotImplementedError('global_weight_decay=True (weight decay on 1-D tensors) is not on this path')
    for k in ('no_weight_decay_layer_name_list', 'sub_layer_weight_decay', 'sub_layer_lr', 'lr_layer_decay'):
        if prm.get(k):
            raise NotImplementedError(f"optimizer option '{k}' is not on this path")
    common = dict(lr=prm['lr'], weight_decay=prm['weight_decay'],
                  accumulation_steps=getattr(config, 'accumulation_steps', 1), process_group=process_group,
                  clip_grad_value=getattr(config, 'clip_grad_value', 0.0) or 0.0,
                  clip_max_norm=getattr(config, 'clip_max_norm', 0.0) or 0.0,
                  skip_inf_nan_grad=getattr(config, 'skip_inf_nan_grad', False), transport=transport)
    def with_ema(trainer):  # build_training_mode's EmaModel (tools/utils.py:172-177): config.ema_model, updated by the step
        if getattr(config, 'use_ema_model', False):
            config.ema_model = EmaModel(model, decay=config.ema_model_decay)
            trainer.ema_model = config.ema_model
        return trainer

    # the trainer follows the MODEL (which step signature and loss call apply), the optimizer follows the config
    if isinstance(model, VITMAEPretrainModel):
        if name != 'AdamW':
            raise NotImplementedError('the MAE pre-training step is built on the fused AdamW kernel')
        # the MAE recipe runs autocast (use_amp=True); this engine's Linear layers are bf16 GEMMs in either case
        return with_ema(MAETrainer(model, criterion, betas=(prm.get('beta1', 0.9), prm.get('beta2', 0.999)),
                                   eps=prm.get('eps', 1e-8), **common))
    if hasattr(model, 'fpn'):  # RetinaNet / FCOS
        if name != 'AdamW':
            raise NotImplementedError('the detection step is built on the fused AdamW kernel')
        if getattr(config, 'use_amp', False):
            raise NotImplementedError('use_amp on the detection path (the graded recipes run fp32, '
                                      '3.detection_training/coco/res50_retinanet_retinaresize800/train_config.py:151)')
        return with_ema(DetectionTrainer(model, criterion, betas=(prm.get('beta1', 0.9), prm.get('beta2', 0.999)),
                                         eps=prm.get('eps', 1e-8), **common))
    if name != 'SGD':
        raise NotImplementedError('the classification step is built on the fused SGD kernel')
    return with_ema(ClassificationTrainer(model, criterion, momentum=prm['momentum'],
                                          nesterov=prm.get('nesterov', False),
                                          use_amp=getattr(config, 'use_amp', None), **common))
