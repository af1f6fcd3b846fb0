"""Forward/backward executor of the ViT MAE pre-training model on the HIP kernels.

Hand-ordered like engine.py: the forward keeps exactly what the backward needs; gradients go straight into the flat
fp32 gradient arena.  Reference graph: VITMAEPretrainModel.forward (simpleAICV/masked_image_modeling/models/vit_mae.py:
411-416) = encoder (:160-185, random masking :187-215) -> encoder_to_decoder -> decoder (:343-372); blocks are
TransformerEncoderLayer (simpleAICV/classification/backbones/vit.py:139-160: pre-norm attention + MLP residuals),
dropout / drop-path 0 in every MAE recipe.  dtypes follow the autocast recipe (csrc/vit.hip header).
"""
import torch

from . import ops, vit_ops as V

LN_EPS = 1e-6


class _Block:
    __slots__ = ('x0', 'm1', 'r1', 'h', 'qkv', 'a', 'lse', 'x1', 'm2', 'r2', 'h2', 'f', 'g')


class MAEEngine:
    """p: name -> parameter tensor (reference state_dict keys), g: name -> fp32 gradient view, arena: the flat arena
    (bf16 weight panels are cast from it once per step)."""

    def __init__(self, p, g, arena, patch, image_size, enc_dim, enc_depth, enc_heads, dec_dim, dec_depth, dec_heads,
                 mask_ratio):
        self.p, self.g, self.arena = p, g, arena
        self.patch, self.image_size = patch, image_size
        self.E, self.Le, self.He = enc_dim, enc_depth, enc_heads
        self.D, self.Ld, self.Hd = dec_dim, dec_depth, dec_heads
        self.n = (image_size // patch)**2
        self.keep = int(self.n * (1 - mask_ratio))
        self.ctx = None
        self._panels = None
        # weight gradients on a side HIP stream, as engine.DetectorEngine._on_side (its comment has the contract): every
        # dy handed to a weight gradient here is a fresh bf16 tensor (cast_bf16 / gelu_bwd / attention_bwd outputs)
        self._side, self._side_busy = None, False
        self._side_evi, self._side_keep = 0, []
        self.side_stream = self.SIDE_STREAM and arena.params.is_cuda

    SIDE_STREAM = True  # default of `side_stream` (see DetectorEngine.side_stream)

    @property
    def side_stream(self):
        return self._side is not None

    @side_stream.setter
    def side_stream(self, on):
        if not on:
            self._side_join()
            self._side = None
        elif self._side is None:
            self._side = torch.cuda.Stream()
            self._side_events = [torch.cuda.Event() for _ in range(64)]

    def _on_side(self, launch, tensors):
        s = self._side
        if s is None or ops.TIMER is not None:  # instrumented steps run in stream order (DetectorEngine._on_side)
            return launch()
        self._side_wait_launch()
        prev_suffix, ops.WS_SUFFIX = ops.WS_SUFFIX, '@side'  # restored, not reset (DetectorEngine._on_side)
        try:
            with torch.cuda.stream(s):
                launch()
        finally:
            ops.WS_SUFFIX = prev_suffix
        self._side_keep.extend(tensors)
        self._side_busy = True

    def _side_wait_launch(self):
        ev = self._side_events[self._side_evi]  # event ring and keep-alive list: see DetectorEngine._on_side
        self._side_evi = (self._side_evi + 1) % len(self._side_events)
        ev.record(torch.cuda.current_stream())
        self._side.wait_event(ev)

    def _stage(self, on_stage, name):
        """Hand exchange stage `name` to the caller on the side stream (DetectorEngine._stage): the collective is chained
        behind the last weight gradient, the launch stream does not wait."""
        if on_stage is None:
            return
        if self._side is None or not self._side_busy or ops.TIMER is not None:
            self._side_join()
            on_stage(name)
            return
        self._side_wait_launch()
        with torch.cuda.stream(self._side):
            on_stage(name)

    def _side_join(self):
        if self._side is not None and self._side_busy:
            torch.cuda.current_stream().wait_stream(self._side)
            self._side_busy = False
            self._side_keep.clear()

    # ------------------------------------------------------------------ weight panels
    def _refresh_panels(self):
        if self._panels is None or not self._panels.is_current():
            ws = [t for k, t in self.p.items() if k.endswith('.weight') and t.ndim == 2]
            self._panels = ops.HalfPanels(self.arena, ws)
        self._panels.refresh()

    def _wf(self, name):
        return self._panels.fwd(self.p[name])

    def _wd(self, name):
        return self._panels.dgrad(self.p[name])

    # ------------------------------------------------------------------ transformer block
    def _block_fwd(self, x, pre, b, n, heads, keep):
        p = self.p
        s = _Block() if keep else None
        h, m1, r1 = V.layernorm_fwd(x, p[pre + '.norm1.weight'], p[pre + '.norm1.bias'], LN_EPS)
        qkv = V.linear_fwd(h, self._wf(pre + '.attn.qkv.weight'), p[pre + '.attn.qkv.bias'])
        a, lse = V.attention_fwd(qkv, b, n, heads)
        # x1 = x + proj(a): the residual is an epilogue input of the GEMM (no copy of x; x stays for the backward pass)
        x1 = V.linear_fwd(a, self._wf(pre + '.attn.proj.weight'), p[pre + '.attn.proj.bias'], out_f32=True,
                          residual=x)
        h2, m2, r2 = V.layernorm_fwd(x1, p[pre + '.norm2.weight'], p[pre + '.norm2.bias'], LN_EPS)
        f = V.linear_fwd(h2, self._wf(pre + '.mlp.fc1.weight'), p[pre + '.mlp.fc1.bias'])
        gl = V.gelu_fwd(f)
        x2 = V.linear_fwd(gl, self._wf(pre + '.mlp.fc2.weight'), p[pre + '.mlp.fc2.bias'], out_f32=True,
                          residual=x1)
        if keep:
            s.x0, s.m1, s.r1, s.h, s.qkv, s.a, s.lse = x, m1, r1, h, qkv, a, lse
            s.x1, s.m2, s.r2, s.h2, s.f, s.g = x1, m2, r2, h2, f, gl
        return x2, s

    def _block_bwd(self, dx, s, pre, b, n, heads):
        """dx fp32 [R,C]: gradient w.r.t. the block output; returns it w.r.t. the block input (in place).  The second
        stages of the block's 8 column sums (4 Linear biases, 2 x LayerNorm dgamma / dbeta) go out as ONE launch."""
        p, g = self.p, self.g
        c = dx.shape[1]
        q = V.ColsumQueue()
        dxb = ops.cast_bf16(dx)
        V.linear_wgrad(dxb, s.g, g[pre + '.mlp.fc2.weight'], g[pre + '.mlp.fc2.bias'], queue=q, side=self._on_side)
        dg = V.linear_dgrad(dxb, self._wd(pre + '.mlp.fc2.weight'), s.g.shape[1])
        df = V.gelu_bwd(dg, s.f)
        V.linear_wgrad(df, s.h2, g[pre + '.mlp.fc1.weight'], g[pre + '.mlp.fc1.bias'], queue=q, side=self._on_side)
        dh2 = V.linear_dgrad(df, self._wd(pre + '.mlp.fc1.weight'), c)
        V.layernorm_bwd(dh2, s.x1, s.m2, s.r2, p[pre + '.norm2.weight'], g[pre + '.norm2.weight'],
                        g[pre + '.norm2.bias'], dx=dx, queue=q)
        dxb = ops.cast_bf16(dx)
        V.linear_wgrad(dxb, s.a, g[pre + '.attn.proj.weight'], g[pre + '.attn.proj.bias'], queue=q, side=self._on_side)
        da = V.linear_dgrad(dxb, self._wd(pre + '.attn.proj.weight'), c)
        dqkv = V.attention_bwd(s.qkv, s.a, da, s.lse, b, n, heads)
        V.linear_wgrad(dqkv, s.h, g[pre + '.attn.qkv.weight'], g[pre + '.attn.qkv.bias'], queue=q, side=self._on_side)
        dh = V.linear_dgrad(dqkv, self._wd(pre + '.attn.qkv.weight'), c)
        V.layernorm_bwd(dh, s.x0, s.m1, s.r1, p[pre + '.norm1.weight'], g[pre + '.norm1.weight'],
                        g[pre + '.norm1.bias'], dx=dx, queue=q)
        q.flush()
        return dx

    # ------------------------------------------------------------------ forward
    def forward(self, images, noise=None, keep=True):
        """images logical [B,3,S,S] fp32 -> (pred bf16 [B*(n+1), P*P*3] incl. the cls row, mask [B,n])."""
        p = self.p
        b = images.shape[0]
        if keep:
            self._refresh_panels()
        elif self._panels is None:
            self._refresh_panels()
        if noise is None:
            noise = torch.rand((b, self.n), device=images.device, dtype=torch.float32)  # vit_mae.py:198
        shuffle, restore, keep_ids, mask = V.random_masking(noise.contiguous(), self.keep)
        rows = V.patchify(images, self.patch, keep_ids)
        wpe = self.p['encoder.patch_embed.proj.weight']
        emb = V.linear_fwd(rows, self._panels.fwd(wpe), p['encoder.patch_embed.proj.bias'])
        x = V.encoder_tokens(emb, p['encoder.cls_token'].reshape(-1), p['encoder.pos_embed'].reshape(-1, self.E),
                             keep_ids)
        ne = self.keep + 1
        enc = []
        for i in range(self.Le):
            x, s = self._block_fwd(x, f'encoder.blocks.{i}', b, ne, self.He, keep)
            enc.append(s)
        xn, me, re_ = V.layernorm_fwd(x, p['encoder.norm.weight'], p['encoder.norm.bias'], LN_EPS)
        y = V.linear_fwd(xn, self._wf('encoder_to_decoder.weight'), p['encoder_to_decoder.bias'])
        xd = V.decoder_tokens(y, p['decoder.mask_token'].reshape(-1), p['decoder.pos_embed'].reshape(-1, self.D),
                              restore, self.keep)
        nd = self.n + 1
        dec = []
        for i in range(self.Ld):
            xd, s = self._block_fwd(xd, f'decoder.blocks.{i}', b, nd, self.Hd, keep)
            dec.append(s)
        xdn, md, rd = V.layernorm_fwd(xd, p['decoder.norm.weight'], p['decoder.norm.bias'], LN_EPS)
        pred = V.linear_fwd(xdn, self._wf('decoder.fc.weight'), p['decoder.fc.bias'])
        if keep:
            self.ctx = dict(b=b, rows=rows, shuffle=shuffle, keep_ids=keep_ids, enc=enc, dec=dec, x_enc=x, me=me,
                            re=re_, xn=xn, x_dec=xd, md=md, rd=rd, xdn=xdn)
        return pred, mask

    # ------------------------------------------------------------------ backward
    def backward(self, dpred, on_stage=None):
        """dpred bf16 [B*(n+1), P*P*3]; fills every gradient view (overwrite semantics)."""
        ctx, p, g = self.ctx, self.p, self.g
        if ctx is None:
            raise RuntimeError('backward() without a preceding forward(keep=True)')
        b = ctx['b']
        V.linear_wgrad(dpred, ctx['xdn'], g['decoder.fc.weight'], g['decoder.fc.bias'], side=self._on_side)
        dxdn = V.linear_dgrad(dpred, self._wd('decoder.fc.weight'), self.D)
        dx = V.layernorm_bwd(dxdn, ctx['x_dec'], ctx['md'], ctx['rd'], p['decoder.norm.weight'],
                             g['decoder.norm.weight'], g['decoder.norm.bias'])
        for i in reversed(range(self.Ld)):
            dx = self._block_bwd(dx, ctx['dec'][i], f'decoder.blocks.{i}', b, self.n + 1, self.Hd)
        dy, gmask = V.decoder_tokens_bwd(dx, ctx['shuffle'], self.keep)
        ops.colsum(gmask, self.D, g['decoder.mask_token'].reshape(-1))
        V.linear_wgrad(dy, ctx['xn'], g['encoder_to_decoder.weight'], g['encoder_to_decoder.bias'],
                       side=self._on_side)
        self._stage(on_stage, 'heads')  # decoder + encoder_to_decoder weights (the tail of the arena) are final
        dxn = V.linear_dgrad(dy, self._wd('encoder_to_decoder.weight'), self.E)
        dx = V.layernorm_bwd(dxn, ctx['x_enc'], ctx['me'], ctx['re'], p['encoder.norm.weight'],
                             g['encoder.norm.weight'], g['encoder.norm.bias'])
        for i in reversed(range(self.Le)):
            dx = self._block_bwd(dx, ctx['enc'][i], f'encoder.blocks.{i}', b, self.keep + 1, self.He)
        dcls_rows = dx.view(b, self.keep + 1, self.E)[:, 0, :].contiguous()
        ops.colsum(dcls_rows, self.E, g['encoder.cls_token'].reshape(-1))
        demb = V.encoder_tokens_bwd(dx, b, self.keep)
        V.linear_wgrad(demb, ctx['rows'], g['encoder.patch_embed.proj.weight'], g['encoder.patch_embed.proj.bias'],
                       side=self._on_side)
        self._side_join()
        if on_stage:
            for st in ('neck', 'mid', 'rest'):
                on_stage(st)
        self.ctx = None
