"""Forward/backward executor of the RetinaNet-/FCOS-ResNet50 detector on the HIP kernels.

The network is a fixed graph, so instead of a per-op autograd tape the executor runs a
hand-ordered forward that keeps exactly the activations the backward needs and a
hand-ordered backward that writes parameter gradients straight into the flat gradient
arena (see detection/models/_arena.py).  Everything that executes on the GPU is a kernel
from libsacv_hip.so; tensors are NHWC.

Reference graph: ResNetBackbone.forward (simpleAICV/detection/models/backbones/resnet.py:
96-113), Bottleneck.forward (simpleAICV/classification/backbones/resnet.py:141-155),
RetinaFPN.forward (simpleAICV/detection/models/fpn.py:60-87), RetinaClsHead/RetinaRegHead
(head.py:46-52, 81-85), RetinaNet.forward (retinanet.py:61-94).
"""
import torch

from . import ops
from ._lib import lib, check
from .ops import CF_RELU, CF_SIGMOID, CF_ACCUM, CF_MASK, PACK_FWD, PACK_DGRAD, PACK_STEM

RESNET50_LAYERS = (3, 4, 6, 3)
BN_EPS = 1e-5
BN_MOMENTUM = 0.1


def _out_hw(h, w, k, s, p):
    return (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1


class _CBA:
    """Saved state of one conv + BatchNorm (+ residual) (+ ReLU) layer."""
    __slots__ = ('prefix', 'x', 'y', 'mean', 'invstd', 'z', 'k', 'stride', 'pad', 'relu', 'residual', 'wino_v')


class DetectorEngine:
    """Executes one detector (RetinaNet when fcos=False).

    p: dict name -> parameter/buffer tensor (reference state_dict key names)
    g: dict name -> gradient tensor of the same shape (views into the flat gradient arena)
    """

    def __init__(self, p, g, num_classes=80, num_anchors=9, planes=256, fcos=False, block='Bottleneck',
                 layers=RESNET50_LAYERS, stem='imagenet'):
        self.p, self.g = p, g
        # trunk: Bottleneck / BasicBlock residual stages (classification/backbones/resnet.py:51-155) behind the
        # ImageNet stem (7x7/s2 conv + 3x3/s2 max pool, :176-188) or the CIFAR stem (3x3/s1 conv, no pool,
        # resnetforcifar.py:38-45)
        assert block in ('Bottleneck', 'BasicBlock') and stem in ('imagenet', 'cifar')
        self.block, self.layers, self.stem = block, tuple(layers), stem
        # autocast mode: bf16 activations and weight panels on v_mfma_f32_32x32x16_bf16, fp32
        # accumulation / statistics / master weights / gradients.  Needs the flat arena (set by the model).
        self.amp = False
        self.arena = None
        self._hpanels = None
        self.num_classes, self.num_anchors, self.planes, self.fcos = num_classes, num_anchors, planes, fcos
        self.ctx = None
        self.bp = 'backbone.'  # key prefix of the trunk in the state_dict ('' for the bare classifier)
        # (Weight gradients on a side HIP stream next to the BatchNorm / data-gradient chain were measured in round 1:
        # no gain, 45.9 vs 44.8 ms/step -- the persistent conv kernels already hold every CU's LDS, so two streams'
        # workgroups cannot co-reside.  The switch was removed; weight gradients run in stream order.)
        self._panels = None  # data-gradient weight panels, refreshed by one launch per training step
        # Winograd F(2x2,3x3) for the stride-1 3x3 head convolutions (csrc/winograd.hip): layer i of both towers in
        # one launch.  False selects the direct implicit-GEMM kernel for every conv (A/B measurements, tests).
        self.winograd = True
        # The 7x7 stem without the im2col matrix (csrc/stem_conv.hip): the forward pass reads the image itself (124 us
        # instead of 143 + 155 for im2col + GEMM) and so does the weight gradient (121 us instead of 207 for conv_tn on the
        # matrix + its reduction) -- the last kernel of the backward pass, alone on the chip while the optimizer waits.
        # Round 3 had the forward kernel only and kept the matrix in training (its weight gradient needed it anyway);
        # with both, the 341 MB matrix is gone from the step.  False: im2col + GEMM everywhere (tests, A/B); 'eval': the
        # round-3 arrangement (image-reading forward only in passes that keep nothing for a backward pass).
        self.fused_stem = True
        # Evaluation passes fold every BatchNorm into its convolution's epilogue (one batched launch computes the
        # scales / shifts from the running statistics per forward pass): 53 bn_apply passes = 1.28 of the 8.9 ms of an
        # eval forward + decode at 2 x 800x1333.  False: conv + bn_apply as in training (tests, A/B).
        self.fold_bn = True
        self._fold = None
        self._wino_cache = {}
        self._side, self._side_busy = None, False  # the weight gradients' stream (see _on_side)
        self._side_evi, self._side_keep = 0, []
        self.side_stream = self.SIDE_STREAM and bool(p) and next(iter(p.values())).is_cuda
        # Output convolutions whose channel count is not a multiple of 4 (Objects365: 9 x 365 = 3285 class channels,
        # FCOS 365): the kernels move channels as 16-byte vectors, so the engine computes with zero-padded SHADOW copies
        # of the weight / bias (refreshed from the parameters at every forward pass) and shadow gradient buffers (their
        # first rows are copied into the gradient arena when the head gradients are final); the padded output channels
        # are stripped before the tensors reach the caller.  self.p / self.g hold the shadows; nothing else changes.
        self._shadows = []
        if p and next(iter(p.values())).is_cuda:
            name = 'clsregcnt_head.cls_out' if fcos else 'cls_head.cls_out'
            if name + '.weight' in p and p[name + '.weight'].shape[0] % 4 != 0:
                self._add_shadow(name)
        # (one switch, engine.side_stream = False, turns the second stream off: A/B runs, the profile session's serial trace)
        self._bias_on_side = self._ds_on_side = True

    SIDE_STREAM = True  # default of `side_stream` for executors built from now on (tests build stream-ordered twins)

    @property
    def side_stream(self):
        """Weight gradients / shortcut branches on a second HIP stream (_on_side).  Assign False for a stream-ordered
        executor (bit-identical results; `bench.py --engine-set side_stream=False`)."""
        return self._side is not None

    @side_stream.setter
    def side_stream(self, on):
        if not on:
            self._side_join()
            self._side = None
        elif self._side is None:
            # (the launch stream at high priority: 27.82 -> 27.62 ms; not adopted)
            dev = next(iter(self.p.values())).device if self.p else None  # the PARAMETERS' device, not the current one
            with torch.cuda.device(dev):
                self._side = torch.cuda.Stream(device=dev, priority=self.SIDE_PRIORITY)
                self._side_events = [torch.cuda.Event() for _ in range(64)]

    SIDE_PRIORITY = 0  # HIP priority of the weight gradients' stream (lower number = served first)

    def _add_shadow(self, name):
        if not self._shadows:  # the dicts are shared with the model / arena: the engine edits copies of its own
            self.p, self.g = dict(self.p), dict(self.g)
        w, b = self.p[name + '.weight'], self.p[name + '.bias']
        n = w.shape[0]
        npad = -(-n // 4) * 4
        mk = lambda: torch.zeros((npad, w.shape[2], w.shape[3], w.shape[1]), device=w.device,
                                 dtype=torch.float32).permute(0, 3, 1, 2)  # channels_last storage, like the arena's
        sw, sgw = mk(), mk()
        sb = torch.zeros(npad, device=w.device, dtype=torch.float32)
        sgb = torch.zeros(npad, device=w.device, dtype=torch.float32)
        self._shadows.append(dict(name=name, n=n, w=w, b=b, gw=self.g[name + '.weight'], gb=self.g[name + '.bias'],
                                  sw=sw, sb=sb, sgw=sgw, sgb=sgb))
        self.p[name + '.weight'], self.p[name + '.bias'] = sw, sb
        self.g[name + '.weight'], self.g[name + '.bias'] = sgw, sgb

    def _shadow_refresh(self):
        for sh in self._shadows:
            sh['sw'][:sh['n']].copy_(sh['w'])
            sh['sb'][:sh['n']].copy_(sh['b'])

    def _shadow_flush(self):
        """Shadow gradients -> gradient arena (call after the side stream has been joined)."""
        for sh in self._shadows:
            sh['gw'].copy_(sh['sgw'][:sh['n']])
            sh['gb'].copy_(sh['sgb'][:sh['n']])

    def _strip(self, levels, n):
        """[B,h,w,npad] -> contiguous [B,h,w,n] when an output conv ran with padded channels."""
        if not levels or levels[0].shape[3] == n:
            return levels
        return [t[..., :n].contiguous() for t in levels]

    @staticmethod
    def _pad_channels(levels, npad):
        if not levels or levels[0].shape[3] == npad:
            return levels
        out = []
        for t in levels:
            z = torch.zeros(t.shape[:3] + (npad,), device=t.device, dtype=t.dtype)
            z[..., :t.shape[3]].copy_(t)
            out.append(z)
        return out

    def _wgrad(self, dys, xs, dw, cin, kh, kw, stride, pad, sync=True, **kw_args):
        self._on_side(lambda: ops.conv_wgrad(dys, xs, dw, cin, kh, kw, stride, pad, **kw_args), list(dys) + list(xs),
                      sync=sync)

    # -- weight gradients on a side HIP stream ----------------------------------------------------------------------
    # A layer's weight gradient depends only on its dy and the saved input and nothing in the backward pass reads it
    # before the stage's exchange / the optimizer.  Enqueued on a second stream (behind an event of the launch
    # stream), its start-up, tail and fixed-order slab reduction run under the data-gradient chain's kernels and the
    # HBM-bound BatchNorm backward kernels run under its MFMA phase, instead of each kernel draining the chip before
    # the next one starts (measured, same box and session: RetinaNet step 29.42 -> 28.53 ms, bf16 classifier
    # 28.21 -> 27.35 ms with the BatchNorm layers' weight gradients alone).  All weight-gradient launches use the
    # side stream in program order, so their workspaces ('wgrad', 'wino_wgrad') never meet the launch stream's.
    # Contract for callers: `dys` / `xs` are not written by the launch stream after this call (every in-place
    # accumulation into a gradient tensor precedes its use as a dy; checked call site by call site).
    def _on_side(self, launch, tensors, sync=True):
        """sync=False: everything `launch` reads was already enqueued on the launch stream when the PREVIOUS _on_side
        call recorded its event (several launches of one dependency group, e.g. a layer's weight gradient and its bias
        gradient): no new event.  An event record is a barrier packet in the launch queue -- the next kernel of the
        launch stream starts 6-7 us after its predecessor instead of ~2 (kernel trace, round 4)."""
        s = self._side
        if s is None or ops.TIMER is not None:
            # no side stream, or a step instrumented with per-launch HIP events (bench.py's roofline measurement): a
            # kernel timed while another stream's kernels share the chip would not report its OWN duration
            return launch()
        if sync or not self._side_busy:
            self._side_wait_launch()
        # scratch buffers of their own: nothing the launch stream's kernels are using.  (A module-level switch: the
        # engines enqueue from ONE host thread, like the reference's training loop.)
        prev_suffix, ops.WS_SUFFIX = ops.WS_SUFFIX, '@side'  # (restored, not reset: a side-stream branch may nest)
        try:
            with torch.cuda.stream(s):
                launch()
        finally:
            ops.WS_SUFFIX = prev_suffix
        # The tensors the side stream reads stay referenced until the launch stream has joined the side stream
        # (_side_join): they are then freed in the launch stream's order, behind the join, so the caching allocator can
        # recycle them for launch-stream work without any cross-stream bookkeeping (no Tensor.record_stream).
        self._side_keep.extend(tensors)
        self._side_busy = True

    def _side_wait_launch(self):
        """The side stream waits for everything enqueued on the launch stream so far."""
        # events come from a small ring (creating / destroying ~150 HIP events per step showed up as multi-second host
        # stalls on a loaded box); a ring slot is reused only long after its wait has been enqueued
        if self.device_scope_events:
            check(lib.sacv_stream_wait_stream(self._side.cuda_stream, torch.cuda.current_stream().cuda_stream),
                  'stream_wait_stream')
            return
        ev = self._side_events[self._side_evi]
        self._side_evi = (self._side_evi + 1) % len(self._side_events)
        ev.record(torch.cuda.current_stream())
        self._side.wait_event(ev)

    # Cross-stream ordering through the library's events (sacv_stream_wait_stream: no system-scope fence per record)
    # instead of torch.cuda.Event / Stream.wait_stream.  False: torch's (A/B).
    device_scope_events = True
    # The launch stream's joins with the side stream through a ticket in device memory (sacv_stream_wait_stream_ticket)
    # instead of an event: built to remove the ~105 us the launch stream idles at the final join; measured 23.18 / 23.12 vs
    # 23.00 / 23.09 ms (two pairs) -- the idle time is not the event's.  Off.
    ticket_joins = False

    def _stage(self, on_stage, name, pre=None):
        """The gradients of exchange stage `name` are all enqueued (weight gradients on the side stream, the rest on the
        launch stream): run `pre` (optional) and hand the stage to the caller's exchange -- ON THE SIDE STREAM, behind an
        event of the launch stream, so that the collective is chained to the last weight gradient while the launch
        stream goes on with the data-gradient chain.  (Until round 4 the launch stream joined the side stream here: the
        kernel trace showed it idle for 954 us after the head towers' backward pass, 34 / 96 us at the other two stages
        -- the side stream's queue of weight gradients runs that far behind.)"""
        if on_stage is None and pre is None:
            return
        s = self._side
        if s is None or not self._side_busy or ops.TIMER is not None:
            self._side_join()
            if pre is not None:
                pre()
            if on_stage is not None:
                on_stage(name)
            return
        self._side_wait_launch()
        with torch.cuda.stream(s):
            if pre is not None:
                pre()
            if on_stage is not None:
                on_stage(name)

    def _side_join(self):
        if self._side is not None and self._side_busy:
            if self.ticket_joins:
                check(lib.sacv_stream_wait_stream_ticket(torch.cuda.current_stream().cuda_stream, self._side.cuda_stream),
                      'stream_wait_stream_ticket')
            elif self.device_scope_events:
                check(lib.sacv_stream_wait_stream(torch.cuda.current_stream().cuda_stream, self._side.cuda_stream),
                      'stream_wait_stream')
            else:
                torch.cuda.current_stream().wait_stream(self._side)
            self._side_busy = False
            self._side_keep.clear()

    def set_amp(self, enabled, arena=None):
        """Switch the bf16 matrix-core path on/off (the reference's config.use_amp / autocast)."""
        if enabled and (arena or self.arena) is None:
            raise ValueError('amp needs the flat parameter arena')
        self.arena = arena or self.arena
        self.amp = bool(enabled)
        return self

    def _fwd_panel(self, w):
        """Forward GEMM panel [Cout][taps*Cin] of a conv / fc weight in the compute dtype."""
        if self.amp:
            return self._hpanels.fwd(w)
        return ops.fwd_panel(w if w.ndim == 4 else w.view(w.shape[0], w.shape[1], 1, 1))

    def _dgrad_panel(self, w):
        """[Cin][KH*KW*Cout] panel of a weight: from the batched per-step refresh when the weight is one of the
        model's parameters, packed on the spot otherwise (composite weights, unit tests)."""
        if self.amp:
            pnl = self._hpanels.dgrad(w)
            if pnl is None:
                raise ValueError('amp: no bf16 data-gradient panel for this weight')
            return pnl
        if self._panels is not None:
            pnl = self._panels.get(w)
            if pnl is not None:
                return pnl
        return ops.pack_weight(w, PACK_DGRAD)

    def _refresh_dgrad_panels(self):
        if self.amp:
            if self._hpanels is None or not self._hpanels.is_current():
                stem = self.bp + 'conv1.layer.0.weight'
                ws = [t for k, t in self.p.items() if k.endswith('.weight') and t.ndim in (2, 4) and k != stem
                      and t.dtype == torch.float32]
                self._hpanels = ops.HalfPanels(self.arena, ws, stem_weight=self.p[stem])
            self._hpanels.refresh()
            return
        if self._panels is None or not self._panels.is_current():
            stem = self.bp + 'conv1.layer.0.weight'
            ws = [t for k, t in self.p.items() if k.endswith('.weight') and t.ndim == 4 and k != stem
                  and t.dtype == torch.float32]
            self._panels = ops.DgradPanels(ws) if ws else None
        if self._panels is not None:
            self._panels.refresh()

    # ------------------------------------------------------------------ building blocks
    def _cba_fwd(self, x, prefix, k, stride, pad, relu, training, residual=None, keep=True):
        p = self.p
        w = p[prefix + '.layer.0.weight']
        cout = w.shape[0]
        wp = self._fwd_panel(w)
        hw = _out_hw(x.shape[1], x.shape[2], k, stride, pad)
        if self._wino_trunk_ok(x, w, k, stride, pad):
            # stride-1 3x3 conv with enough work items for the Winograd GEMM (the bs-256 classifier's layer2 / layer3)
            v = self._wino_input([x], keep and training)
            r = ops.conv3x3_wino([[x]], [self._wino_u(w)], [cout], vs=[v], stats=training)
            (y,), st = (r[0][0], r[1][0]) if training else (r[0], None)
            z, c = self._bn_fwd(x, y, prefix, k, stride, pad, relu, training, residual, keep, st)
            if c is not None:
                c.wino_v = v if training else None
            return z, c
        if training:  # batch statistics come out of the conv epilogue (no extra pass over y)
            (y,), st = ops.conv_gemm([x], wp, cout, k, k, stride, pad, 0, [hw], stats=True)
        elif self._fold is not None and not keep:
            # evaluation: BatchNorm with running statistics is a per-channel scale and shift -- applied in the conv's
            # epilogue (SACV_CF_AFFINE) together with the ReLU; a residual block's last conv accumulates into the buffer
            # that holds the shortcut (dead after this block in a pass that keeps nothing): no bn_apply pass at all
            flags = ops.CF_AFFINE | (CF_RELU if relu else 0)
            if residual is not None:
                return ops.conv_gemm([x], wp, cout, k, k, stride, pad, 0, [hw], bias=self._fold.get(prefix),
                                     flags=flags | CF_ACCUM, outs=[residual])[0], None
            return ops.conv_gemm([x], wp, cout, k, k, stride, pad, 0, [hw], bias=self._fold.get(prefix),
                                 flags=flags)[0], None
        else:
            y, st = ops.conv_gemm([x], wp, cout, k, k, stride, pad, 0, [hw])[0], None
        return self._bn_fwd(x, y, prefix, k, stride, pad, relu, training, residual, keep, st)

    WINO_TRUNK_MIN_ITEMS = 250  # 64-tile x 128-channel work items from which a trunk 3x3 conv takes the Winograd path

    def _wino_trunk_ok(self, x, w, k, stride, pad):
        if not (self._use_wino() and k == 3 and stride == 1 and pad == 1):
            return False
        cout, cin = w.shape[0], w.shape[1]
        if cin < 64 or cout < 64 or cin % 32 != 0:
            return False
        tiles = x.shape[0] * ((x.shape[1] + 1) // 2) * ((x.shape[2] + 1) // 2)
        mtiles = -(-tiles // 64)
        if cin >= 128 and cout >= 128:
            return mtiles * -(-cout // 128) >= self.WINO_TRUNK_MIN_ITEMS
        # 64-channel layers (K = 64: two K tiles per g, the transforms weigh more): only with thousands of items -- the
        # bs-256 classifier's layer1 (3136); at 2 images per GPU the detector's layer1 (522 items) is slower than direct
        return mtiles * -(-cout // 64) >= 8 * self.WINO_TRUNK_MIN_ITEMS

    def _bn_fwd(self, x, y, prefix, k, stride, pad, relu, training, residual, keep, stats=None):
        p = self.p
        bn = prefix + '.layer.1.'
        if training:
            if stats is not None:
                mean, invstd = ops.bn_finalize_stats(stats, y.numel() // y.shape[-1], BN_EPS, BN_MOMENTUM,
                                                     p[bn + 'running_mean'], p[bn + 'running_var'],
                                                     p[bn + 'num_batches_tracked'])
            else:
                mean, invstd = ops.bn_train_stats(y, BN_EPS, BN_MOMENTUM, p[bn + 'running_mean'],
                                                  p[bn + 'running_var'], p[bn + 'num_batches_tracked'])
            z = ops.bn_apply(y, mean, invstd, p[bn + 'weight'], p[bn + 'bias'], residual, relu)
        else:
            mean = invstd = None
            z = ops.bn_apply(y, p[bn + 'running_mean'], p[bn + 'running_var'], p[bn + 'weight'], p[bn + 'bias'],
                             residual, relu, stat_is_var=True, eps=BN_EPS)
        if not keep:
            return z, None
        c = _CBA()
        c.wino_v = None
        c.prefix, c.x, c.y, c.mean, c.invstd, c.z = prefix, x, y, mean, invstd, z
        c.k, c.stride, c.pad, c.relu, c.residual = k, stride, pad, relu, residual is not None
        return z, c

    # BatchNorm backward, first stage, in the epilogue of the data gradient that PRODUCES the layer's upstream gradient
    # (ops.conv_gemm_bnbwd): the per-channel sums of g and g * xhat come from values that are in registers there, the
    # bn_bwd_partial pass (one read of dz and y, one launch; 0.9 ms per RetinaNet step in stream order, 1.7 ms beside the
    # weight gradients) disappears for 41 of the trunk's 53 layers.  False: every layer runs its own first stage (A/B,
    # tests).
    fuse_bn_bwd = True
    # The same under bf16 autocast (sacv_conv2d_gemm_bf16_bnbwd; the 64-channel 3x3 data gradients stay on the
    # halo-resident kernel, which has no such epilogue).  Built, parity-tested (tests/test_gpu_bf16.py) and measured on
    # ResNet50 bs 256 (scripts/ab_cls_bnbwd.sh, stream order, ms per step): bn_bwd_partial 2.82 -> 1.20 (38 of 53 launches
    # gone), but gemm_nt_bf16_dma_kernel 3.79 -> 4.82, conv_nt_bf16<128,128> data gradients 2.19 -> 3.22, <128,64> 0.51 ->
    # 0.70: the step 26.6 -> 27.4 ms.  The bf16 epilogues are latency-bound (one 512-thread workgroup per CU in the GEMM
    # kernel: every tile waits once more for its rows of y), the stand-alone pass streams the same bytes at 0.6-0.7 of the
    # HBM roofline.  Off.
    fuse_bn_bwd_amp = False
    # Where a layer's weight gradient enters the side stream: right behind the BatchNorm backward that produces its dy
    # (False: it starts WITH the layer's data gradient -- two matrix-core kernels taking turns on the CUs) or behind the
    # data gradient's launch (True: it starts when the data gradient has finished, next to the NEXT layer's BatchNorm
    # backward -- an HBM-bound phase of the launch stream).  Measured (round 4, same session, three runs each): 26.7 ms
    # with False, 27.45 with True -- two matrix-core kernels that each reach ~0.55 of the peak alone fill each other's
    # stalls when they share the chip; serialising them loses more than the complementary pairing gains.  For scale: the
    # step without ANY weight gradient takes 21.56 ms, with all of them in stream order 28.98.
    wgrad_after_dgrad = False

    def _cba_bwd(self, c, dz, need_dx=True, want_g=False, dx_out=None, dx_accum=False, partial=None, fuse=None):
        """Backward of one layer.  Returns (dx, g, st): g is the ReLU-masked upstream gradient; st -- when `fuse` (the
        saved state of the layer whose BatchNorm receives dx as ITS upstream gradient) could be served -- that layer's
        first-stage sums, to be handed to its own _cba_bwd as `partial`."""
        p, g = self.p, self.g
        bn = c.prefix + '.layer.1.'
        wname = c.prefix + '.layer.0.weight'
        # ReLU mask: layers without a residual input recompute it from y (one read of z less in both passes)
        recompute = c.relu and not c.residual
        dy, gm = ops.bn_train_bwd(dz, c.z if (c.relu and not recompute) else None, c.y, c.mean, c.invstd,
                                  p[bn + 'weight'], g[bn + 'weight'], g[bn + 'bias'], want_g=want_g,
                                  relu_beta=p[bn + 'bias'] if recompute else None, partial=partial)
        w = p[wname]
        cin = w.shape[1]
        v = getattr(c, 'wino_v', None)
        if v is not None:  # the forward pass took the Winograd path: so do both gradients
            self._wino_wgrad([dy], [c.x], v, g[wname], cin)
            dx = None
            if need_dx:
                dx = ops.conv3x3_wino([[dy]], [self._wino_u(w, dgrad=True)], [cin],
                                      flags=[CF_ACCUM if dx_accum else 0],
                                      outs_list=[[dx_out]] if dx_out is not None else None, mode=1)[0][0]
            return dx, gm, None
        if not self.wgrad_after_dgrad:
            self._trunk_wgrad(c, dy, g[wname], cin)
        dx, st = None, None
        if need_dx:
            if (fuse is not None and self.fuse_bn_bwd and c.stride == 1 and fuse.relu
                    and tuple(fuse.y.shape) == tuple(c.x.shape)
                    and (not self.amp or (self.fuse_bn_bwd_amp and not (c.k == 3 and cin == 64)))):
                f_rc = not fuse.residual  # as above: mask recomputed from y unless a residual entered the ReLU
                fbn = fuse.prefix + '.layer.1.'
                dx, st = ops.conv_gemm_bnbwd(dy, self._dgrad_panel(w), cin, c.k, c.k, c.pad, c.x.shape[1:3], fuse.y,
                                             fuse.mean, fuse.invstd, bn_gamma=p[fbn + 'weight'] if f_rc else None,
                                             bn_beta=p[fbn + 'bias'] if f_rc else None,
                                             bn_z=None if f_rc else fuse.z, out=dx_out,
                                             flags=CF_ACCUM if dx_accum else 0)
            else:
                dx = self._dgrad(dy, w, c.x.shape, c.k, c.stride, c.pad, out=dx_out, accumulate=dx_accum)
        if self.wgrad_after_dgrad:
            self._trunk_wgrad(c, dy, g[wname], cin)
        return dx, gm, st

    # Weight gradient of the trunk's stride-1 3x3 convolutions with >= 128 channels on the Winograd path (2.25x fewer
    # matrix-core flops): the forward pass ran them on the direct kernel and kept no transformed input, so the side
    # stream transforms x itself (wino_input into its scratch) before the 16 split-K GEMMs.  At 2 x 800 x 1333 (us,
    # transform included, scripts/bench_wino_wgrad_trunk.py): 128 channels 98 vs 122 for conv_tn, 256: 77 vs 112, 512: 77
    # vs 125; 64 channels (layer1) 174 vs 138 -- the transforms of the 200 x 334 maps outweigh the saving: conv_tn
    # there.  Step: 26.79 -> 26.57 ms with all four widths, same-session A/B.  False: conv_tn for every trunk layer.
    wino_trunk_wgrad = True

    def _trunk_wgrad(self, c, dy, dw, cin):
        if (self.wino_trunk_wgrad and self._use_wino() and c.k == 3 and c.stride == 1 and c.pad == 1 and cin >= 128
                and ops.wino_wgrad_supported(cin) and dy.shape[3] >= 128):
            def launch():
                ops.wino_wgrad([dy], ops.wino_input([c.x], tag='wino_vx', split=self.wino_split and self.wino_wgrad_split),
                               dw, cin)
            return self._on_side(launch, [dy, c.x])
        self._wgrad([dy], [c.x], dw, cin, c.k, c.k, c.stride, c.pad)

    def _dgrad(self, dy, w, x_shape, k, stride, pad, out=None, accumulate=False, mask=None):
        cin = w.shape[1]
        wd = self._dgrad_panel(w)
        n, h, wdt, _ = x_shape
        flags = (CF_ACCUM if accumulate else 0) | (CF_MASK if mask is not None else 0)
        if k == 1 and stride > 1:
            # 1x1 strided conv: only every stride-th input pixel receives gradient -> GEMM over the
            # dy pixels scattered (accumulated) into dx
            if out is None:
                out = torch.zeros(x_shape, device=dy.device, dtype=dy.dtype)
                flags |= CF_ACCUM
            elif not accumulate:
                raise ValueError('strided 1x1 dgrad writes a sparse pattern: out must be accumulated into')
            ops.conv_gemm([dy], wd, cin, 1, 1, 1, 0, 0, [(dy.shape[1], dy.shape[2])], flags=flags,
                          outs=[out], remap=(stride, h, wdt))
            return out
        if stride == 2 and k > 1:
            return self._dgrad_parity(dy, wd, cin, x_shape, k, pad, out, flags, mask)
        outs = [out] if out is not None else None
        return ops.conv_gemm([dy], wd, cin, k, k, stride, pad, 1, [(h, wdt)], flags=flags,
                             masks=[mask] if mask is not None else None, outs=outs)[0]

    def _dgrad_parity(self, dy, wd, cin, x_shape, k, pad, out, flags, mask):
        """Data gradient of a stride-2 k x k conv as four stride-1 problems, one per parity class of the
        input pixel (h%2, w%2): class (ph, pw) is reached only by taps kh = (ph+pad)%2 (mod 2) and likewise
        for kw, so the four launches together execute exactly the forward's MACs instead of 4x."""
        n, h, w, _ = x_shape
        if out is None:
            out = torch.empty(x_shape, device=dy.device, dtype=dy.dtype)
        for ph in (0, 1):
            for pw in (0, 1):
                hc, wc = (h - ph + 1) // 2, (w - pw + 1) // 2
                if hc <= 0 or wc <= 0:
                    continue
                kh0, kw0 = (ph + pad) & 1, (pw + pad) & 1
                nkh, nkw = len(range(kh0, k, 2)), len(range(kw0, k, 2))
                if nkh == 0 or nkw == 0:
                    raise NotImplementedError('parity class without taps')
                ah, aw = (ph + pad - kh0) // 2, (pw + pad - kw0) // 2
                ext = [ah, aw, kh0, 2, kw0, 2, k, k, 0, ph, pw, 0]
                ops.conv_gemm([dy], wd, cin, nkh, nkw, 1, 0, 1, [(hc, wc)], flags=flags,
                              masks=[mask] if mask is not None else None, outs=[out], remap=(2, h, w), ext=ext)
        return out

    # ------------------------------------------------------------------ forward
    def forward(self, images, training=True, keep=True):
        """images: logical [B,3,H,W] fp32 (any strides).  Returns (cls_levels, reg_levels)."""
        ctx = {} if keep else None
        self._apply_plan()
        self._wino_cache = {}  # Winograd-domain weights are per forward (the optimizer step changes the weights)
        self._shadow_refresh()
        self._wino_prepared = False
        if keep:
            if self.amp or not self._ds_on_side:
                self._refresh_dgrad_panels()  # amp: the same launch pair also casts the FORWARD panels
            else:
                # fp32: the data-gradient panels are read by the backward pass only (which joins the side stream first);
                # the Winograd-domain weights of the head towers (forward and data-gradient panels, one batched launch)
                # depend on the weights alone: behind the panels on the side stream, under the trunk's forward pass
                def prep():
                    self._refresh_dgrad_panels()
                    if self._use_wino() and not self.fcos and self._side is not None:
                        self._wino_prepare(self._retina_wino_names(), True, joined=True)
                        self._wino_prepared = True
                self._on_side(prep, [])
        self._fold_prepare(training, keep)
        early = {}
        on_feat = None
        if self._side is not None and self.fpn_laterals_early and ops.TIMER is None:
            # the 1x1 laterals of C3 / C4 depend on their stage alone: on the side stream, under layer3 / layer4
            def on_feat(li, x):
                name = {2: 'fpn.P3_1', 3: 'fpn.P4_1'}.get(li)
                if name is not None:
                    def lateral():
                        early[name] = self._conv_bias(x, name, 1, 1, 0)
                    self._on_side(lateral, [x])
        feats = self._backbone_fwd(images, training, keep, ctx, on_feat)
        c3f, c4f, c5f = feats[1], feats[2], feats[3]
        fpn = self._fpn_fwd(c3f, c4f, c5f, ctx, early)
        outs = self._fcos_heads_fwd(fpn, ctx) if self.fcos else self._heads_fwd(fpn, ctx)
        if keep:
            self.ctx = ctx
        return outs

    def _fold_prepare(self, training, keep):
        """Scales / shifts of the eval-mode BatchNorm layers for this forward pass (None: not an evaluation pass)."""
        if training or keep or self.amp or not self.fold_bn:
            self._fold = None
            return
        f = getattr(self, '_fold_table', None)
        if f is None or not f.is_current():
            tail = '.layer.1.weight'
            layers = [(k[:-len(tail)], self.p[k], self.p[k[:-len(tail)] + '.layer.1.bias'],
                       self.p[k[:-len(tail)] + '.layer.1.running_mean'], self.p[k[:-len(tail)] + '.layer.1.running_var'])
                      for k in self.p if k.endswith(tail)]
            f = self._fold_table = ops.BnFold(layers, BN_EPS)
        f.refresh()
        self._fold = f

    def _backbone_fwd(self, images, training, keep, ctx, on_feat=None):
        """ResNet trunk (classification/backbones/resnet.py:158-245): returns [C2, C3, C4, C5].  on_feat(li, x): called
        when stage li's output exists (the FPN laterals of C3 / C4 start on the side stream under the next stage)."""
        p, bp = self.p, self.bp
        b, _, h, w = images.shape
        # stem: explicit im2col (Cin=3) + GEMM, BN, ReLU (+ 3x3/s2 max pool for the ImageNet stem)
        sk, ss, sp = (7, 2, 3) if self.stem == 'imagenet' else (3, 1, 1)
        ho, wo = _out_hw(h, w, sk, ss, sp)
        fused = not self.amp and self.stem == 'imagenet' and (self.fused_stem is True or
                                                               (self.fused_stem == 'eval' and not keep))
        if self.amp:
            col = ops.im2col_stem(images, ho, wo, sk, sk, ss, sp, dtype=ops.BF16)
            wst = self._hpanels.stem_panel
        else:
            wst = ops.pack_weight(p[bp + 'conv1.layer.0.weight'], PACK_STEM)
            # fp32 7x7/s2 stem: the convolution reads the image itself (csrc/stem_conv.hip); the im2col matrix the
            # weight gradient contracts with is rebuilt in the backward pass, on the side stream (_backbone_bwd)
            col = None if fused else ops.im2col_stem(images, ho, wo, sk, sk, ss, sp)
        if fused:
            y, st = ops.stem_conv(images, wst, ho, wo, stats=True) if training else (ops.stem_conv(images, wst, ho, wo), None)
        elif training:
            (y,), st = ops.conv_gemm([col], wst, 64, 1, 1, 1, 0, 0, [(ho, wo)], stats=True)
        else:
            y, st = ops.conv_gemm([col], wst, 64, 1, 1, 1, 0, 0, [(ho, wo)])[0], None
        z, c = self._bn_fwd(col, y, bp + 'conv1', sk, ss, sp, True, training, None, keep, st)
        if fused and keep:
            ctx['stem_images'] = images
        if self.stem == 'imagenet':
            x, pool_idx = ops.maxpool_fwd(z, 3, 2, 1, want_idx=keep)
        else:
            x, pool_idx = z, None
        if keep:
            ctx['stem'] = c
            ctx['pool_idx'] = pool_idx
            ctx['pool_in_shape'] = tuple(z.shape)
        feats = []
        blocks = []
        fwd = self._bottleneck_fwd if self.block == 'Bottleneck' else self._basicblock_fwd
        for li, n in enumerate(self.layers, start=1):
            for bi in range(n):
                pre = f'{bp}layer{li}.{bi}'
                stride = 2 if (bi == 0 and li > 1) else 1
                out, saved = fwd(x, pre, stride, training, keep)
                blocks.append((saved, li, bi))
                x = out
            feats.append(x)
            if on_feat is not None:
                on_feat(li, x)
        if keep:
            ctx['blocks'] = blocks
        return feats

    def _basicblock_fwd(self, x, pre, stride, training, keep=True):
        """BasicBlock.forward (classification/backbones/resnet.py:51-97): 3x3(stride) -> 3x3, shortcut (1x1 conv + BN
        when the shape changes), add + ReLU fused into the last BN pass."""
        has_ds = (pre + '.downsample_conv.layer.0.weight') in self.p
        z1, c1 = self._cba_fwd(x, pre + '.conv1', 3, stride, 1, True, training, keep=keep)
        if has_ds:
            idt, cd = self._cba_fwd(x, pre + '.downsample_conv', 1, stride, 0, False, training, keep=keep)
        else:
            idt, cd = x, None
        out, c2 = self._cba_fwd(z1, pre + '.conv2', 3, 1, 1, True, training, residual=idt, keep=keep)
        return out, (c1, c2, cd)

    def _basicblock_bwd(self, saved, dout, accum_target):
        c1, c2, cd = saved
        dz1, gshared, _ = self._cba_bwd(c2, dout, want_g=True)
        if cd is None:
            dx, _, _ = self._cba_bwd(c1, dz1, dx_out=gshared, dx_accum=True)
            return dx
        if accum_target is not None:
            dx, _, _ = self._cba_bwd(c1, dz1, dx_out=accum_target, dx_accum=True)
        else:
            dx, _, _ = self._cba_bwd(c1, dz1)
        self._cba_bwd(cd, gshared, dx_out=dx, dx_accum=True)
        return dx

    def _bottleneck_fwd(self, x, pre, stride, training, keep=True):
        """Bottleneck.forward (classification/backbones/resnet.py:141-155): 1x1 -> 3x3(stride) -> 1x1,
        shortcut (1x1 conv + BN when the shape changes), add + ReLU fused into the last BN pass."""
        has_ds = (pre + '.downsample_conv.layer.0.weight') in self.p
        ds_side = has_ds and self._side is not None and self._ds_on_side
        if ds_side:
            # the shortcut branch (1x1 conv + BN) only meets the main branch at conv3's BatchNorm pass: it runs on the
            # side stream under conv1 / conv2.  Its tensors are allocated under the side stream but live on (the
            # residual add, the backward pass) on the launch stream: record_stream tells the caching allocator so.
            box = {}

            def shortcut():
                box['r'] = self._cba_fwd(x, pre + '.downsample_conv', 1, stride, 0, False, training, keep=keep)
            self._on_side(shortcut, [x])
        z1, c1 = self._cba_fwd(x, pre + '.conv1', 1, 1, 0, True, training, keep=keep)
        z2, c2 = self._cba_fwd(z1, pre + '.conv2', 3, stride, 1, True, training, keep=keep)
        if ds_side:
            self._side_join()
            idt, cd = box['r']
            cur = torch.cuda.current_stream()
            for t in (idt,) + ((cd.y, cd.mean, cd.invstd) if cd is not None else ()):
                if t is not None:
                    t.record_stream(cur)
        elif has_ds:
            idt, cd = self._cba_fwd(x, pre + '.downsample_conv', 1, stride, 0, False, training, keep=keep)
        else:
            idt, cd = x, None
        out, c3 = self._cba_fwd(z2, pre + '.conv3', 1, 1, 0, True, training, residual=idt, keep=keep)
        return out, (c1, c2, c3, cd)

    def _conv_bias(self, x, name, k, stride, pad, flags=0):
        w = self.p[name + '.weight']
        wp = ops.fwd_panel(w)
        hw = _out_hw(x.shape[1], x.shape[2], k, stride, pad)
        return ops.conv_gemm([x], wp, w.shape[0], k, k, stride, pad, 0, [hw], bias=self.p[name + '.bias'],
                             flags=flags)[0]

    fpn_on_side = True  # P5_2 / P4_2 / P6 / P7 of the forward pass on the side stream, beside P3_2 (_fpn_fwd)
    # P3_1 / P4_1 on the side stream as soon as C3 / C4 exist (forward()): measured a tie (23.03 vs 23.01 ms, three
    # same-session pairs) -- beside layer3 / layer4's kernels a lateral takes from them what it saves later.  Off.
    fpn_laterals_early = False

    WINO_MIN_TILES = 6000  # a single 3x3 conv goes to the Winograd path from this many 2x2 output tiles (enough
    #                        64-tile work items to occupy the 256 CUs; below it the direct kernel is faster)

    def _conv3x3_bias(self, x, name, ctx):
        """3x3 / stride 1 / pad 1 convolution + bias of one level: Winograd path for large levels (P3), else direct."""
        tiles = x.shape[0] * ((x.shape[1] + 1) // 2) * ((x.shape[2] + 1) // 2)
        w = self.p[name + '.weight']
        if not self._use_wino() or tiles < self.WINO_MIN_TILES:
            return self._conv_bias(x, name, 3, 1, 1)
        keep = ctx is not None
        v = self._wino_input([x], keep)
        if keep:
            ctx['wino_v:' + name] = v
        return ops.conv3x3_wino([[x]], [self._wino_u(w)], [w.shape[0]], biases=[self.p[name + '.bias']], vs=[v])[0][0]

    def _conv3x3_bias_bwd(self, dy, x, name):
        """Backward of _conv3x3_bias: returns dx; fills the weight and bias gradients."""
        v = self.ctx.get('wino_v:' + name)
        if v is None:
            return self._conv_bias_bwd(dy, x, name, 3, 1, 1)
        w = self.p[name + '.weight']
        self._wino_wgrad([dy], [x], v, self.g[name + '.weight'], w.shape[1])
        self._bias_grad([dy], name, sync=False)
        return ops.conv3x3_wino([[dy]], [self._wino_u(w, dgrad=True)], [w.shape[1]], mode=1)[0][0]

    def _fpn_fwd(self, c3, c4, c5, ctx, early=None):
        p5 = self._conv_bias(c5, 'fpn.P5_1', 1, 1, 0)
        if early:  # laterals enqueued on the side stream when their stage finished (forward())
            self._side_join()
            cur = torch.cuda.current_stream()
            for t in early.values():
                t.record_stream(cur)
        p4l = early['fpn.P4_1'] if early else self._conv_bias(c4, 'fpn.P4_1', 1, 1, 0)
        p4 = ops.upsample_add_fwd(p5, p4l)
        p3l = early['fpn.P3_1'] if early else self._conv_bias(c3, 'fpn.P3_1', 1, 1, 0)
        p3 = ops.upsample_add_fwd(p4, p3l)
        box = {}

        def small_levels():
            p5o = self._conv_bias(p5, 'fpn.P5_2', 3, 1, 1)
            p4o = self._conv_bias(p4, 'fpn.P4_2', 3, 1, 1)
            p6 = self._conv_bias(p5o if self.fcos else c5, 'fpn.P6', 3, 2, 1)  # use_p5 (fpn.py:80)
            p6r = ops.eltwise(ops.EW_RELU, p6)
            box['r'] = (p5o, p4o, p6, p6r, self._conv_bias(p6r, 'fpn.P7.1', 3, 2, 1))
        if self._side is not None and self.fpn_on_side and ops.TIMER is None:
            # P3_2 is one Winograd launch of 524 work items for 512 slots (a second, nearly empty round) or, below the
            # Winograd threshold, a direct launch with a long tail; the four small output convolutions are independent
            # of it: they run on the side stream (idle during the forward pass) beside it.  Their outputs are allocated
            # under the side stream and live on on the launch stream: record_stream, as for the shortcut branch.
            self._on_side(small_levels, [p5, p4, c5])
            p3o = self._conv3x3_bias(p3, 'fpn.P3_2', ctx)
            self._side_join()
            cur = torch.cuda.current_stream()
            for t in box['r']:
                t.record_stream(cur)
        else:
            small_levels()
            p3o = self._conv3x3_bias(p3, 'fpn.P3_2', ctx)
        p5o, p4o, p6, p6r, p7 = box['r']
        if ctx is not None:
            ctx['fpn'] = dict(c3=c3, c4=c4, c5=c5, p5=p5, p4=p4, p3=p3, p6=p6, p6r=p6r, p5o=p5o)
        return [p3o, p4o, p5o, p6, p7]

    def _tower_fwd(self, feats, tower, out_name, n_out, out_flags, ctx):
        p = self.p
        xs = feats
        hw = [(f.shape[1], f.shape[2]) for f in feats]
        saved = []
        for i in (0, 2, 4, 6):
            name = f'{tower}.{i}'
            wp = ops.fwd_panel(p[name + '.weight'])
            saved.append(xs)
            xs = ops.conv_gemm(xs, wp, self.planes, 3, 3, 1, 1, 0, hw, bias=p[name + '.bias'], flags=CF_RELU)
        saved.append(xs)
        wp = ops.fwd_panel(p[out_name + '.weight'])
        outs = ops.conv_gemm(xs, wp, n_out, 3, 3, 1, 1, 0, hw, bias=p[out_name + '.bias'], flags=out_flags)
        if ctx is not None:
            ctx[tower] = saved
        return outs

    def _wino_u(self, w, dgrad=False):
        """Winograd-domain weights U = G w G^T of a 3x3 parameter for this step (forward, or the rotated transposed
        kernel of the data gradient).  Cached per forward(): the weights only change in the optimizer step."""
        key = (w.data_ptr(), dgrad)
        u = self._wino_cache.get(key)
        if u is None:
            co, ci = w.shape[0], w.shape[1]
            if dgrad:
                u = ops.wino_weights(self._dgrad_panel(w), ci, co, flip=True, split=self.wino_split)
            else:
                u = ops.wino_weights(ops.fwd_panel(w), co, ci, split=self.wino_split)
            self._wino_cache[key] = u
        return u

    def _use_wino(self):
        return self.winograd and not self.amp

    # Operand form of the Winograd GEMMs (forward and data gradient): False = fp32 operands on the fp32 matrix pipe;
    # True = every operand as three bf16 planes whose sum IS the fp32 value, six partial products on the bf16 pipe
    # into fp32 accumulators (csrc/winograd.hip header: what is dropped lies below fp32 rounding).  The weight gradient
    # rebuilds the kept V from its planes exactly, so it does not change.
    _wino_prepared = False  # forward() prepared this pass's Winograd-domain weights on the side stream
    wino_split = True
    # The same operand form for the direct convolutions (sacv_conv2d_gemm: the split happens when the gathered fp32
    # values are stored to LDS, nothing changes in memory).  It is part of the calling thread's launch plan
    # (ops.launch_share(nt_split=...)); forward() / backward() set it from this attribute.
    conv_split = True
    # Winograd weight gradients on the bf16 pipe too (False: the fp32-pipe kernel; a kept split V is rebuilt exactly)
    wino_wgrad_split = True

    def _apply_plan(self):
        # every launch goes to the CURRENT device's current stream (ops._stream): a model on another device than the
        # current one would enqueue on the wrong stream -- one process drives one GPU (torch.cuda.set_device first)
        if self.p:
            d = next(iter(self.p.values())).device
            if d.type == 'cuda' and d.index is not None and d.index != torch.cuda.current_device():
                raise RuntimeError(f'the model is on {d} but the current device is cuda:{torch.cuda.current_device()}: '
                                   'call torch.cuda.set_device() before building / running the model')
        if not self.amp and ops.current_share()[7] != int(self.conv_split):
            ops.launch_share(nt_split=int(self.conv_split))

    def _retina_wino_names(self):
        return ([f'{t}.{i}.weight' for t, _ in self.RETINA_TOWERS for i in (0, 2, 4, 6)] +
                [o + '.weight' for _, o in self.RETINA_TOWERS])

    def _wino_prepare(self, names, training, joined=False):
        """All Winograd-domain weights of the head convolutions in ONE launch per pass (forward panels, and the
        data-gradient panels when a backward pass follows) instead of one 5 us launch per weight and direction."""
        jobs, keys = [], []
        if training and not joined:
            self._side_join()  # the data-gradient panels read below are refreshed on the side stream (forward())
        for n in names:
            w = self.p[n]
            co, ci = w.shape[0], w.shape[1]
            jobs.append((ops.fwd_panel(w), co, ci, False))
            keys.append((w.data_ptr(), False))
            if training:
                jobs.append((self._dgrad_panel(w), ci, co, True))
                keys.append((w.data_ptr(), True))
        for key, u in zip(keys, ops.wino_weights_batched(jobs, split=self.wino_split)):
            self._wino_cache[key] = u

    def _heads_fwd(self, feats, ctx):
        a, nc = self.num_anchors, self.num_classes
        n_cls = self.p['cls_head.cls_out.weight'].shape[0]  # a * nc, or its padded shadow's
        if self._use_wino():
            cls, reg = self._towers_fwd_wino(feats, ctx, n_cls, a * 4)
        else:
            cls = self._tower_fwd(feats, 'cls_head.cls_head', 'cls_head.cls_out', n_cls, CF_SIGMOID, ctx)
            reg = self._tower_fwd(feats, 'reg_head.reg_head', 'reg_head.reg_out', a * 4, 0, ctx)
        cls = self._strip(cls, a * nc)
        if ctx is not None:
            ctx['cls_probs'] = cls
            ctx['feat_shapes'] = [tuple(f.shape) for f in feats]
        cls_levels = [t.view(t.shape[0], t.shape[1], t.shape[2], a, nc) for t in cls]
        reg_levels = [t.view(t.shape[0], t.shape[1], t.shape[2], a, 4) for t in reg]
        return cls_levels, reg_levels

    RETINA_TOWERS = (('cls_head.cls_head', 'cls_head.cls_out'), ('reg_head.reg_head', 'reg_head.reg_out'))

    def _towers_fwd_wino(self, feats, ctx, n_cls, n_reg):
        """RetinaClsHead / RetinaRegHead (head.py:20-36, 55-70) with layer i of BOTH towers in one Winograd launch
        (the towers are independent chains over the same pyramid: 2 x 352 work items fill the 512 workgroup slots
        better than 352).  Layer 0 of the two towers reads the same input: one input transform.  When the pass is
        kept for a backward, every layer's transformed input V stays alive for its weight gradient."""
        p = self.p
        keep = ctx is not None
        if self._wino_prepared:
            self._side_join()  # transformed on the side stream since the start of forward()
        else:
            self._wino_prepare(self._retina_wino_names(), keep)
        xs = [feats, feats]
        saved, vsaved = [[], []], [[], []]
        for li, i in enumerate((0, 2, 4, 6)):
            names = [f'{t}.{i}' for t, _ in self.RETINA_TOWERS]
            vs = self._wino_inputs(xs, keep, shared=li == 0)
            for t in range(2):
                saved[t].append(xs[t])
                vsaved[t].append(vs[t])
            xs = ops.conv3x3_wino(xs, [self._wino_u(p[n + '.weight']) for n in names], [self.planes] * 2,
                                  biases=[p[n + '.bias'] for n in names], flags=[CF_RELU] * 2, vs=vs)
        names = [o for _, o in self.RETINA_TOWERS]
        vs = self._wino_inputs(xs, keep)
        for t in range(2):
            saved[t].append(xs[t])
            vsaved[t].append(vs[t])
        outs = ops.conv3x3_wino(xs, [self._wino_u(p[n + '.weight']) for n in names], [n_cls, n_reg],
                                biases=[p[n + '.bias'] for n in names], flags=[CF_SIGMOID, 0], vs=vs)
        if keep:
            for t in range(2):
                ctx[self.RETINA_TOWERS[t][0]] = saved[t]
                ctx[self.RETINA_TOWERS[t][0] + '.wino_v'] = vsaved[t]
        return outs

    def _wino_input(self, xs, keep, tag='wino_v0'):
        """Transformed input V of a 3x3 convolution over the levels xs: a fresh tensor when it is kept for the weight
        gradient of the backward pass, a shared scratch buffer otherwise."""
        return ops.wino_input(xs, tag=tag, split=self.wino_split, keep=keep)

    def _wino_inputs(self, xs2, keep, shared=False):
        """Transformed inputs of the two towers' next layer (one transform when both read the same levels)."""
        v0 = self._wino_input(xs2[0], keep, 'wino_v0')
        return [v0, v0 if shared else self._wino_input(xs2[1], keep, 'wino_v1')]

    def _wino_wgrad(self, dys, xs, v, dw, cin, sync=True):
        """3x3 weight gradient: Winograd path from the kept V when the channel count allows it, else conv_tn."""
        if v is not None and ops.wino_wgrad_supported(cin) and dys[0].shape[3] >= 64:
            self._on_side(lambda: ops.wino_wgrad(dys, v, dw, cin, exact=not self.wino_wgrad_split), list(dys) + [v], sync=sync)
        else:
            self._wgrad(dys, xs, dw, cin, 3, 3, 1, 1, sync=sync)

    # ------------------------------------------------------------------ FCOS heads
    # FCOSClsRegCntHead (head.py:88-181): two towers of 4 x (3x3 conv no bias, GroupNorm(32), ReLU) shared by
    # the 5 levels; cls_out (sigmoid) on the cls tower; reg_out and center_out on the reg tower are run as ONE
    # 8-channel conv (4 + 1 + 3 zero rows) and split; FCOS.forward scales reg by exp(scales[l]) (fcos.py:76).
    def _gn_tower_fwd(self, feats, tower, ctx):
        p = self.p
        xs = feats
        hw = [(f.shape[1], f.shape[2]) for f in feats]
        saved = []
        for i in (0, 3, 6, 9):
            w = p[f'{tower}.{i}.weight']
            ys = ops.conv_gemm(xs, ops.fwd_panel(w), self.planes, 3, 3, 1, 1, 0, hw)
            zs, sts = [], []
            for y in ys:
                z, st = ops.groupnorm_fwd(y, p[f'{tower}.{i + 1}.weight'], p[f'{tower}.{i + 1}.bias'], 32, 1e-5, True)
                zs.append(z)
                sts.append(st)
            saved.append((xs, ys, sts, zs))
            xs = zs
        if ctx is not None:
            ctx[tower] = saved
        return xs

    def _fcos_w8(self):
        p = self.p
        w8 = torch.zeros((8, self.planes, 3, 3), device=p['scales'].device, dtype=torch.float32)
        b8 = torch.zeros(8, device=w8.device, dtype=torch.float32)
        w8[0:4].copy_(p['clsregcnt_head.reg_out.weight'])
        w8[4:5].copy_(p['clsregcnt_head.center_out.weight'])
        b8[0:4].copy_(p['clsregcnt_head.reg_out.bias'])
        b8[4:5].copy_(p['clsregcnt_head.center_out.bias'])
        return w8, b8

    FCOS_TOWERS = ('clsregcnt_head.cls_head', 'clsregcnt_head.reg_head')

    def _gn_towers_fwd_wino(self, feats, ctx):
        """Both FCOS towers in lockstep: layer i of the two towers is one Winograd launch (no bias: GroupNorm follows),
        then GroupNorm + ReLU per level as in _gn_tower_fwd."""
        p = self.p
        keep = ctx is not None
        xs = [feats, feats]
        saved, vsaved = [[], []], [[], []]
        for li, i in enumerate((0, 3, 6, 9)):
            ws = [p[f'{t}.{i}.weight'] for t in self.FCOS_TOWERS]
            vs = self._wino_inputs(xs, keep, shared=li == 0)
            vsaved[0].append(vs[0])
            vsaved[1].append(vs[1])
            yss = ops.conv3x3_wino(xs, [self._wino_u(w) for w in ws], [self.planes] * 2, vs=vs)
            nxt = []
            for t, tower in enumerate(self.FCOS_TOWERS):
                zs, sts = [], []
                for y in yss[t]:
                    z, st = ops.groupnorm_fwd(y, p[f'{tower}.{i + 1}.weight'], p[f'{tower}.{i + 1}.bias'], 32, 1e-5,
                                              True)
                    zs.append(z)
                    sts.append(st)
                saved[t].append((xs[t], yss[t], sts, zs))
                nxt.append(zs)
            xs = nxt
        if keep:
            for t in range(2):
                ctx[self.FCOS_TOWERS[t]] = saved[t]
                ctx[self.FCOS_TOWERS[t] + '.wino_v'] = vsaved[t]
        return xs

    def _fcos_heads_fwd(self, feats, ctx):
        p = self.p
        hw = [(f.shape[1], f.shape[2]) for f in feats]
        w = p['clsregcnt_head.cls_out.weight']
        w8, b8 = self._fcos_w8()
        if self._use_wino():
            cls_x, reg_x = self._gn_towers_fwd_wino(feats, ctx)
            vs = self._wino_inputs([cls_x, reg_x], ctx is not None)
            if ctx is not None:
                ctx['fcos_out_v'] = vs
            cls, y8 = ops.conv3x3_wino([cls_x, reg_x], [self._wino_u(w), self._wino_u(w8)], [w.shape[0], 8],
                                       biases=[p['clsregcnt_head.cls_out.bias'], b8], flags=[CF_SIGMOID, 0], vs=vs)
        else:
            cls_x = self._gn_tower_fwd(feats, 'clsregcnt_head.cls_head', ctx)
            reg_x = self._gn_tower_fwd(feats, 'clsregcnt_head.reg_head', ctx)
            cls = ops.conv_gemm(cls_x, ops.fwd_panel(w), w.shape[0], 3, 3, 1, 1, 0, hw,
                                bias=p['clsregcnt_head.cls_out.bias'], flags=CF_SIGMOID)
            y8 = ops.conv_gemm(reg_x, ops.fwd_panel(w8), 8, 3, 3, 1, 1, 0, hw, bias=b8)
        cls = self._strip(cls, self.num_classes)
        regs, ctrs = [], []
        for l, y in enumerate(y8):
            r, c = ops.fcos_split_fwd(y, p['scales'][l:l + 1])
            regs.append(r)
            ctrs.append(c)
        if ctx is not None:
            ctx['fcos_out'] = dict(cls=cls, reg=regs, ctr=ctrs, cls_x=cls_x, reg_x=reg_x, w8=w8)
        return cls, regs, ctrs

    def _gn_towers_bwd_wino(self, dzs2):
        """Backward of the two FCOS towers in lockstep (dzs2[t]: per-level grads w.r.t. tower t's output)."""
        p, g = self.p, self.g
        saved = [self.ctx[t] for t in self.FCOS_TOWERS]
        vsaved = [self.ctx[t + '.wino_v'] for t in self.FCOS_TOWERS]
        cur = dzs2
        for li, i in reversed(list(enumerate((0, 3, 6, 9)))):
            dys2, ws = [], []
            for t, tower in enumerate(self.FCOS_TOWERS):
                xs, ys, sts, zs = saved[t][li]
                gname, bname = f'{tower}.{i + 1}.weight', f'{tower}.{i + 1}.bias'
                dys = [ops.groupnorm_bwd(y, dz, z, st, p[gname], g[gname], g[bname], 32, accumulate=l > 0)
                       for l, (y, dz, z, st) in enumerate(zip(ys, cur[t], zs, sts))]
                w = p[f'{tower}.{i}.weight']
                self._wino_wgrad(dys, xs, vsaved[t][li], g[f'{tower}.{i}.weight'], w.shape[1])
                dys2.append(dys)
                ws.append(w)
            us = [self._wino_u(w, dgrad=True) for w in ws]
            if li > 0:
                cur = ops.conv3x3_wino(dys2, us, [self.planes] * 2, mode=1)
            else:
                dfeat = ops.conv3x3_wino([dys2[0]], [us[0]], [self.planes], mode=1)[0]
                ops.conv3x3_wino([dys2[1]], [us[1]], [self.planes], flags=[CF_ACCUM], outs_list=[dfeat], mode=1)
        return dfeat

    def _gn_tower_bwd(self, dzs, tower, accum_into=None):
        """dzs: per-level grads w.r.t. the tower output (after the last GN + ReLU)."""
        p, g = self.p, self.g
        saved = self.ctx[tower]
        cur = dzs
        for li, i in reversed(list(enumerate((0, 3, 6, 9)))):
            xs, ys, sts, zs = saved[li]
            gname, bname = f'{tower}.{i + 1}.weight', f'{tower}.{i + 1}.bias'
            dys = [ops.groupnorm_bwd(y, dz, z, st, p[gname], g[gname], g[bname], 32, accumulate=l > 0)
                   for l, (y, dz, z, st) in enumerate(zip(ys, cur, zs, sts))]
            w = p[f'{tower}.{i}.weight']
            self._wgrad(dys, xs, g[f'{tower}.{i}.weight'], w.shape[1], 3, 3, 1, 1)
            hw = [(x.shape[1], x.shape[2]) for x in xs]
            wd = self._dgrad_panel(w)
            if li == 0 and accum_into is not None:
                cur = ops.conv_gemm(dys, wd, w.shape[1], 3, 3, 1, 1, 1, hw, flags=CF_ACCUM, outs=accum_into)
            else:
                cur = ops.conv_gemm(dys, wd, w.shape[1], 3, 3, 1, 1, 1, hw)
        return cur

    def _fcos_heads_bwd(self, dcls_levels, dreg_levels, dctr_levels, grad_scale):
        p, g = self.p, self.g
        o = self.ctx['fcos_out']
        hw = [(x.shape[1], x.shape[2]) for x in o['cls_x']]
        # classification branch
        dz_cls = [ops.eltwise(ops.EW_SCALE_SIGMOID_BWD, d.reshape(pr.shape), pr, scalar=grad_scale)
                  for d, pr in zip(dcls_levels, o['cls'])]
        w = p['clsregcnt_head.cls_out.weight']
        dz_cls = self._pad_channels(dz_cls, w.shape[0])
        vcls = self.ctx['fcos_out_v'][0] if self._use_wino() else None
        self._wino_wgrad(dz_cls, o['cls_x'], vcls, g['clsregcnt_head.cls_out.weight'], w.shape[1])
        self._bias_grad(dz_cls, 'clsregcnt_head.cls_out', sync=False)
        # regression + centerness branch (8-channel combined output conv)
        dy8 = [ops.fcos_split_bwd(dr.contiguous(), dc.contiguous(), r, c, p['scales'][l:l + 1], g['scales'], l,
                                  gscale=grad_scale)
               for l, (dr, dc, r, c) in enumerate(zip(dreg_levels, dctr_levels, o['reg'], o['ctr']))]
        dw8 = torch.empty_like(o['w8'])

        def wgrad8():  # on the weight gradients' stream, like every user of the 'wgrad' workspace (see _on_side)
            ops.conv_wgrad(dy8, o['reg_x'], dw8, self.planes, 3, 3, 1, 1)
            g['clsregcnt_head.reg_out.weight'].copy_(dw8[0:4])
            g['clsregcnt_head.center_out.weight'].copy_(dw8[4:5])
        self._on_side(wgrad8, list(dy8) + list(o['reg_x']) + [dw8])
        db8 = torch.empty(8, device=dw8.device, dtype=torch.float32)
        for l, d in enumerate(dy8):
            ops.colsum(d, 8, db8, accumulate=l > 0)
        g['clsregcnt_head.reg_out.bias'].copy_(db8[0:4])
        g['clsregcnt_head.center_out.bias'].copy_(db8[4:5])
        if self._use_wino():  # separate launches: 80 vs 8 input channels (see _towers_bwd_wino)
            d_cls_x = ops.conv3x3_wino([dz_cls], [self._wino_u(w, dgrad=True)], [self.planes], mode=1)[0]
            d_reg_x = ops.conv3x3_wino([dy8], [self._wino_u(o['w8'], dgrad=True)], [self.planes], mode=1)[0]
            return self._gn_towers_bwd_wino([d_cls_x, d_reg_x])
        d_cls_x = ops.conv_gemm(dz_cls, self._dgrad_panel(w), w.shape[1], 3, 3, 1, 1, 1, hw)
        dfeat = self._gn_tower_bwd(d_cls_x, 'clsregcnt_head.cls_head')
        d_reg_x = ops.conv_gemm(dy8, self._dgrad_panel(o['w8']), self.planes, 3, 3, 1, 1, 1, hw)
        return self._gn_tower_bwd(d_reg_x, 'clsregcnt_head.reg_head', accum_into=dfeat)

    # ------------------------------------------------------------------ backward
    def _bias_grad(self, dys, name, sync=True):
        """Column sums of dy: like the weight gradients, nothing in the backward pass waits for them -> side stream.
        sync=False: right behind the same dys' weight gradient (see _on_side)."""
        gb = self.g[name + '.bias']
        c = gb.numel()

        def launch():
            for i, d in enumerate(dys):
                ops.colsum(d, c, gb, accumulate=i > 0)
        if self._bias_on_side:
            self._on_side(launch, list(dys), sync=sync)
        else:
            launch()

    def _bias_grad_from_stats(self, st, gb, sync=True):
        """Second stage of a bias gradient whose column sums came out of a data-gradient epilogue (CF_STATS)."""
        if self._bias_on_side:
            self._on_side(lambda: ops.colsum_finalize(st, gb), [st], sync=sync)
        else:
            ops.colsum_finalize(st, gb)

    def _tower_bwd(self, dys, tower, out_name, accum_into=None):
        """dys: per-level grads w.r.t. the tower's output conv result (pre-activation).

        The bias gradient of tower conv i is the column sum of its dy, which is exactly what the data-
        gradient kernel of conv i+1 stores: it is taken from that kernel's CF_STATS epilogue (all five
        levels in one launch) instead of a separate pass over dy."""
        p, g = self.p, self.g
        saved = self.ctx[tower]
        xs = saved[4]
        w = p[out_name + '.weight']
        self._wgrad(dys, xs, g[out_name + '.weight'], w.shape[1], 3, 3, 1, 1)
        self._bias_grad(dys, out_name, sync=False)
        hw = [(x.shape[1], x.shape[2]) for x in xs]
        cur, st = ops.conv_gemm(dys, self._dgrad_panel(w), w.shape[1], 3, 3, 1, 1, 1, hw, flags=CF_MASK,
                                masks=xs, stats=True)
        for li, i in reversed(list(enumerate((0, 2, 4, 6)))):
            name = f'{tower}.{i}'
            w = p[name + '.weight']
            xs = saved[li]
            self._wgrad(cur, xs, g[name + '.weight'], w.shape[1], 3, 3, 1, 1)
            self._bias_grad_from_stats(st, g[name + '.bias'], sync=False)
            wd = self._dgrad_panel(w)
            if li > 0:
                cur, st = ops.conv_gemm(cur, wd, w.shape[1], 3, 3, 1, 1, 1, hw, flags=CF_MASK, masks=xs, stats=True)
            elif accum_into is not None:
                cur = ops.conv_gemm(cur, wd, w.shape[1], 3, 3, 1, 1, 1, hw, flags=CF_ACCUM, outs=accum_into)
            else:
                cur = ops.conv_gemm(cur, wd, w.shape[1], 3, 3, 1, 1, 1, hw)
        return cur

    def _towers_bwd_wino(self, dz_cls, dz_reg):
        """Backward of the two Retina towers in lockstep: weight gradients and data gradients on the Winograd path
        (ReLU mask and bias-gradient column sums in the data-gradient epilogue, as _tower_bwd), layer i of both towers
        per data-gradient launch."""
        p, g = self.p, self.g
        towers = self.RETINA_TOWERS
        saved = [self.ctx[t] for t, _ in towers]
        vsaved = [self.ctx[t + '.wino_v'] for t, _ in towers]
        dys = [dz_cls, dz_reg]
        for t, (_, out_name) in enumerate(towers):
            w = p[out_name + '.weight']
            self._wino_wgrad(dys[t], saved[t][4], vsaved[t][4], g[out_name + '.weight'], w.shape[1], sync=t == 0)
            self._bias_grad(dys[t], out_name, sync=False)
        # the two output convs have very different depths (720 / 36 channels): in one launch the item lengths differ
        # 11x and the static item order leaves whole XCDs idle (1384 us against 922 + 104 us separately)
        cur, st = [], []
        for t, (_, o) in enumerate(towers):
            c, s_ = ops.conv3x3_wino([dys[t]], [self._wino_u(p[o + '.weight'], dgrad=True)], [self.planes],
                                     flags=[CF_MASK], masks_list=[saved[t][4]], stats=True, mode=1)
            cur.append(c[0])
            st.append(s_[0])
        for li, i in reversed(list(enumerate((0, 2, 4, 6)))):
            names = [f'{t}.{i}' for t, _ in towers]
            ws = [p[n + '.weight'] for n in names]
            for t in range(2):
                self._wino_wgrad(cur[t], saved[t][li], vsaved[t][li], g[names[t] + '.weight'], ws[t].shape[1],
                                 sync=t == 0)
                self._bias_grad_from_stats(st[t], g[names[t] + '.bias'], sync=False)
            us = [self._wino_u(w, dgrad=True) for w in ws]
            if li > 0:
                cur, st = ops.conv3x3_wino(cur, us, [self.planes] * 2, flags=[CF_MASK] * 2,
                                           masks_list=[saved[0][li], saved[1][li]], stats=True, mode=1)
            else:
                # both towers feed the same pyramid gradient: the second launch accumulates into the first's output
                dfeat = ops.conv3x3_wino([cur[0]], [us[0]], [self.planes], mode=1)[0]
                ops.conv3x3_wino([cur[1]], [us[1]], [self.planes], flags=[CF_ACCUM], outs_list=[dfeat], mode=1)
        return dfeat

    def _conv_bias_bwd(self, dy, x, name, k, stride, pad, need_dx=True, out=None, accumulate=False, mask=None):
        w = self.p[name + '.weight']
        self._wgrad([dy], [x], self.g[name + '.weight'], w.shape[1], k, k, stride, pad)
        self._bias_grad([dy], name, sync=False)
        if need_dx:
            return self._dgrad(dy, w, x.shape, k, stride, pad, out=out, accumulate=accumulate, mask=mask)
        return None

    def backward(self, dcls_levels, dreg_levels, dctr_levels=None, grad_scale=1.0, on_stage=None):
        """dcls: grads w.r.t. the class PROBABILITIES, dreg: w.r.t. the box regressions (per level,
        same shapes as the forward outputs), dctr (FCOS): w.r.t. the centerness probabilities.
        Fills every entry of self.g (overwrite semantics)."""
        ctx = self.ctx
        if ctx is None:
            raise RuntimeError('backward() without a preceding forward(keep=True)')
        self._side_join()  # the forward pass's side-stream work (data-gradient panels) is complete
        self._apply_plan()
        prev_share = self._share_begin()
        try:
            self._backward(dcls_levels, dreg_levels, dctr_levels, grad_scale, on_stage)
        finally:
            self._share_end(prev_share)

    # Two queues share the chip for the whole backward pass (the data-gradient chain on the launch stream, the weight
    # gradients on the side stream): each plans for ITS share of a CU's workgroup slots (ops.launch_share) -- together the
    # two kernels fit a CU's LDS and registers, every workgroup of a persistent grid is resident from the start and the
    # stream-K ranges are cut for the workgroups that really run side by side.
    # Measured (round 4, same session, 2 x 800x1333; `bench.py --engine-set BWD_SHARE=...`): planning as if alone 27.8 ms;
    # (2,1,2,1) 29.9; (2,2,2,1) 30.0; (3,2,2,1) 29.7; (2,1,3,1) 28.9; (2,1,2,2) 28.5; (3,2,3,1) 28.8 -- every static partition
    # loses to over-subscription: the two queues' workgroups take turns on a CU at workgroup granularity (a CU's LDS holds
    # two conv_tn or three conv_nt workgroups, not both sets) and a kernel that runs in bursts at full occupancy beats one
    # that shares the matrix pipe all the time.  The knob stays for the CU reserve of multi-GPU runs.
    BWD_SHARE = None  # (nt64, nt128, wino, wgrad[, cu_reserve]) slots per CU during the backward pass; None: as if alone

    def _share_begin(self):
        share = self.BWD_SHARE
        if share is None or self._side is None or ops.TIMER is not None:
            return None
        return ops.launch_share(*share)

    def _share_end(self, prev):
        if prev is not None:
            ops.launch_share(*prev)

    def _backward(self, dcls_levels, dreg_levels, dctr_levels, grad_scale, on_stage):
        ctx = self.ctx
        if self.fcos:
            dfeat = self._fcos_heads_bwd(dcls_levels, dreg_levels, dctr_levels, grad_scale)
            return self._trunk_bwd(dfeat, on_stage)
        a, nc = self.num_anchors, self.num_classes
        probs = ctx['cls_probs']
        # sigmoid backward (and the upstream scalar) fused in one pass per level
        dz_cls = []
        for dprob, pr in zip(dcls_levels, probs):
            d = dprob.reshape(pr.shape)
            dz_cls.append(ops.eltwise(ops.EW_SCALE_SIGMOID_BWD, d, pr, scalar=grad_scale))
        dz_cls = self._pad_channels(dz_cls, self.p['cls_head.cls_out.weight'].shape[0])
        if grad_scale != 1.0:
            dz_reg = [ops.eltwise(ops.EW_SCALE, d.reshape(d.shape[0], d.shape[1], d.shape[2], a * 4),
                                  scalar=grad_scale) for d in dreg_levels]
        else:
            dz_reg = [d.reshape(d.shape[0], d.shape[1], d.shape[2], a * 4) for d in dreg_levels]
        if self._use_wino():
            dfeat = self._towers_bwd_wino(dz_cls, dz_reg)
        else:
            dfeat = self._tower_bwd(dz_cls, 'cls_head.cls_head', 'cls_head.cls_out')
            dfeat = self._tower_bwd(dz_reg, 'reg_head.reg_head', 'reg_head.reg_out', accum_into=dfeat)
        self._trunk_bwd(dfeat, on_stage)

    def _trunk_bwd(self, dfeat, on_stage):
        ctx = self.ctx
        # head gradients are final: their all-reduce can start now (behind the side stream's weight gradients)
        self._stage(on_stage, 'heads', pre=self._shadow_flush if self._shadows else None)
        dc3, dc4, dc5 = self._fpn_bwd(dfeat)
        self._backbone_bwd(dc5, dc4, dc3, on_stage)
        self._side_join()
        if on_stage:
            on_stage('rest')
        self.ctx = None

    def _backbone_bwd(self, dout, dc4=None, dc3=None, on_stage=None):
        """Backward of the trunk, last block first; dc4 / dc3 are the FPN lateral gradients that join the
        data gradient at the inputs of layer4 / layer3 (None for the classifier)."""
        ctx, bp = self.ctx, self.bp
        blocks = ctx['blocks']
        st = None
        for idx in reversed(range(len(blocks))):
            saved, li, bi = blocks[idx]
            accum_target = None
            if bi == 0 and li == 4:
                accum_target = dc4
            elif bi == 0 and li == 3:
                accum_target = dc3
            if self.block == 'Bottleneck':
                # an identity block's dx is the previous block's upstream gradient, complete in conv1's data gradient
                prev_c3 = blocks[idx - 1][0][2] if (bi > 0 and saved[3] is None) else None
                dout, st = self._bottleneck_bwd(saved, dout, accum_target, partial=st, fuse_prev=prev_c3)
            else:
                dout = self._basicblock_bwd(saved, dout, accum_target)
            if on_stage and bi == 0 and li in (3, 4):
                self._stage(on_stage, 'neck' if li == 4 else 'mid')  # fpn + layer4 / layer3 gradients are final
        # max pool + stem
        if self.stem == 'imagenet':
            dpool = ops.maxpool_bwd(dout, ctx['pool_idx'], ctx['pool_in_shape'], 3, 2, 1)
        else:
            dpool = dout
        st = ctx['stem']
        bn = bp + 'conv1.layer.1.'
        dy, _ = ops.bn_train_bwd(dpool, None, st.y, st.mean, st.invstd, self.p[bn + 'weight'], self.g[bn + 'weight'],
                                 self.g[bn + 'bias'], relu_beta=self.p[bn + 'bias'])
        if st.x is None:  # fused stem forward: no im2col matrix was kept -- the weight gradient reads the image too
            # On the LAUNCH stream: it is the last kernel of the backward pass (nothing left to run beside it) and the
            # optimizer waits for it -- on the side stream the cross-queue hand-over that the final join then has to wait
            # for cost 80-90 us between its last kernel and the optimizer's first (kernel trace, round 4).
            ops.stem_conv_wgrad(ctx['stem_images'], dy, self.g[bp + 'conv1.layer.0.weight'])
        else:
            self._wgrad([dy], [st.x], self.g[bp + 'conv1.layer.0.weight'], 3, 1, 1, 1, 0,
                        cin_pad=st.x.shape[-1], dw_taps=st.k * st.k)

    def _bottleneck_bwd(self, saved, dout, accum_target, partial=None, fuse_prev=None):
        """partial: first-stage sums of this block's bn3 (from the data gradient that produced dout); fuse_prev: conv3
        of the PREVIOUS block, whose bn3 receives this block's dx -- returns (dx, its first-stage sums or None)."""
        c1, c2, c3, cd = saved
        # out = relu(bn3(conv3(z2)) + idt): g is the masked gradient shared by both branches
        dz2, gshared, st2 = self._cba_bwd(c3, dout, want_g=True, partial=partial, fuse=c2)
        dz1, _, st1 = self._cba_bwd(c2, dz2, partial=st2, fuse=c1)
        if cd is None:
            # identity shortcut: dx = g + dgrad(conv1) -- complete in that launch's epilogue
            return self._cba_bwd(c1, dz1, partial=st1, dx_out=gshared, dx_accum=True, fuse=fuse_prev)[::2]
        if accum_target is not None:
            dx, _, _ = self._cba_bwd(c1, dz1, partial=st1, dx_out=accum_target, dx_accum=True)
        else:
            dx, _, _ = self._cba_bwd(c1, dz1, partial=st1)
        self._cba_bwd(cd, gshared, dx_out=dx, dx_accum=True)
        return dx, None

    def _fpn_bwd(self, dfeat):
        f = self.ctx['fpn']
        dp3o, dp4o, dp5o, dp6, dp7 = dfeat
        # P7 = conv(relu(P6)); P6 also feeds the heads directly
        t = self._conv_bias_bwd(dp7, f['p6r'], 'fpn.P7.1', 3, 2, 1, mask=f['p6'])
        dp6 = ops.eltwise(ops.EW_ADD, dp6, t, out=dp6)
        if self.fcos:  # P6 = conv(P5 smoothed): its data gradient joins dP5o
            self._conv_bias_bwd(dp6, f['p5o'], 'fpn.P6', 3, 2, 1, out=dp5o, accumulate=True)
            dc5 = None
        else:
            dc5 = self._conv_bias_bwd(dp6, f['c5'], 'fpn.P6', 3, 2, 1)
        dp5 = self._conv_bias_bwd(dp5o, f['p5'], 'fpn.P5_2', 3, 1, 1)
        dp4 = self._conv_bias_bwd(dp4o, f['p4'], 'fpn.P4_2', 3, 1, 1)
        dp3 = self._conv3x3_bias_bwd(dp3o, f['p3'], 'fpn.P3_2')
        # P3 = up(P4) + P3_1(C3)
        ops.upsample_bwd(dp3, dp4, accumulate=True)
        dc3 = self._conv_bias_bwd(dp3, f['c3'], 'fpn.P3_1', 1, 1, 0)
        # P4 = up(P5) + P4_1(C4)
        ops.upsample_bwd(dp4, dp5, accumulate=True)
        dc4 = self._conv_bias_bwd(dp4, f['c4'], 'fpn.P4_1', 1, 1, 0)
        if dc5 is None:
            dc5 = self._conv_bias_bwd(dp5, f['c5'], 'fpn.P5_1', 1, 1, 0)
        else:
            self._conv_bias_bwd(dp5, f['c5'], 'fpn.P5_1', 1, 1, 0, out=dc5, accumulate=True)
        return dc3, dc4, dc5


class ClassifierEngine(DetectorEngine):
    """ResNet classifiers (ResNet50 ImageNet, classification/backbones/resnet.py:158-245; BASELINE
    config #1: ResNet18 CIFAR-100, resnetforcifar.py:27-108): the trunk of the detector engine + global average pool
    + FC (a 1x1 problem of the conv GEMM kernels)."""

    def __init__(self, p, g, num_classes=1000, block='Bottleneck', layers=RESNET50_LAYERS, stem='imagenet'):
        super().__init__(p, g, num_classes=num_classes, block=block, layers=layers, stem=stem)
        self.bp = ''

    def forward(self, images, training=True, keep=True):
        """images logical [B,3,H,W] -> logits [B, num_classes]"""
        ctx = {} if keep else None
        self._apply_plan()
        self._wino_cache = {}  # Winograd-domain weights are per forward (see DetectorEngine.forward)
        if keep:
            self._refresh_dgrad_panels()
        self._fold_prepare(training, keep)
        c5 = self._backbone_fwd(images, training, keep, ctx)[3]
        pooled = ops.avgpool_fwd(c5)
        w, bias = self._fc_padded()
        nout = self.p['fc.weight'].shape[0]
        if self.amp:
            wp = self._hpanels.fwd(w) if w is self.p['fc.weight'] else ops.cast_bf16(w)
            logits = ops.conv_gemm([pooled], wp, w.shape[0], 1, 1, 1, 0, 0, [(1, 1)], bias=bias,
                                   flags=ops.CFH_OUT_F32)[0]
        else:
            wp = ops.fwd_panel(w.view(w.shape[0], w.shape[1], 1, 1))
            logits = ops.conv_gemm([pooled], wp, w.shape[0], 1, 1, 1, 0, 0, [(1, 1)], bias=bias)[0]
        logits = logits.view(logits.shape[0], w.shape[0])
        if w.shape[0] != nout:
            logits = logits[:, :nout].contiguous()
        if keep:
            ctx['pooled'] = pooled
            ctx['c5_shape'] = tuple(c5.shape)
            ctx['fc_w'] = w
            self.ctx = ctx
        return logits

    def _fc_padded(self):
        """fc weight / bias with the output count padded to a multiple of 4 (the GEMM's N granule): zero rows for
        e.g. the 10 classes of CIFAR-10.  Plain views when num_classes % 4 == 0."""
        w, b = self.p['fc.weight'], self.p['fc.bias']
        nout = w.shape[0]
        gran = 8 if self.amp else 4
        npad = -(-nout // gran) * gran
        if npad == nout:
            return w, b
        wp = torch.zeros((npad, w.shape[1]), device=w.device, dtype=torch.float32)
        bp = torch.zeros(npad, device=w.device, dtype=torch.float32)
        wp[:nout].copy_(w)
        bp[:nout].copy_(b)
        return wp, bp

    def backward(self, dlogits, on_stage=None):
        ctx = self.ctx
        if ctx is None:
            raise RuntimeError('backward() without a preceding forward(keep=True)')
        self._side_join()
        self._apply_plan()
        w = ctx['fc_w']
        nout, nin = self.p['fc.weight'].shape
        npad = w.shape[0]
        dl = dlogits.contiguous()
        if npad != nout:
            dy = torch.zeros((dl.shape[0], npad), device=dl.device, dtype=torch.float32)
            dy[:, :nout].copy_(dl)
            dwp = torch.empty((npad, nin, 1, 1), device=dl.device, dtype=torch.float32)
            dbp = torch.empty(npad, device=dl.device, dtype=torch.float32)
        else:
            dy, dwp, dbp = dl, self.g['fc.weight'].view(nout, nin, 1, 1), self.g['fc.bias']
        ops.colsum(dy.view(-1, npad), npad, dbp)
        if self.amp:
            dy = ops.cast_bf16(dy)
        dy = dy.view(dy.shape[0], 1, 1, npad)
        ops.conv_wgrad([dy], [ctx['pooled']], dwp, nin, 1, 1, 1, 0)  # launch stream: dwp is copied right below
        if npad != nout:
            self.g['fc.weight'].copy_(dwp.view(npad, nin)[:nout])
            self.g['fc.bias'].copy_(dbp[:nout])
        if self.amp:
            wd = self._hpanels.dgrad(w) if w is self.p['fc.weight'] else \
                ops.cast_bf16(w.t().contiguous())  # [nin][npad]
            dpooled = ops.conv_gemm([dy], wd, nin, 1, 1, 1, 0, 1, [(1, 1)])[0]
        else:
            dpooled = self._dgrad(dy, w.view(npad, nin, 1, 1), ctx['pooled'].shape, 1, 1, 0)
        dc5 = ops.avgpool_bwd(dpooled, ctx['c5_shape'])
        self._stage(on_stage, 'heads')
        self._backbone_bwd(dc5, None, None, on_stage)
        self._side_join()
        if on_stage:
            on_stage('rest')
        self.ctx = None
