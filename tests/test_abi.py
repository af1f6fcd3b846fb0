"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/sacv_hip.h declares, and the Python mirror honours the reference's class contract
(constructor kwargs, state_dict keys/shapes/order, parameter groups) captured from the reference
in tests/golden/contract.json.  No kernels are launched here."""
import ctypes
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'sacv_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(sacv_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from sacv_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = _declared_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), f'{s} declared in include/sacv_hip.h but not exported'
    assert sorted(_lib.SIGNATURES) == syms, 'ctypes binding and header disagree'
    assert _lib.lib.sacv_abi_version() == 5


def test_library_exports_nothing_else():
    """-fvisibility=hidden + csrc/exports.map: the dynamic symbol table holds the C ABI and nothing else (no C++
    device stubs, kernel handles or internal helpers)."""
    import subprocess
    from sacv_amd import _lib
    out = subprocess.run(['nm', '-D', '--defined-only', _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(line.split()[-1] for line in out.splitlines() if line.strip())
    assert exported == _declared_symbols()


def test_state_dict_contract(golden_dir):
    from sacv_amd.detection import models
    c = json.load(open(os.path.join(golden_dir, 'contract.json')))['resnet50_retinanet']
    m = models.__dict__['resnet50_retinanet'](backbone_pretrained_path='', num_classes=80)
    sd = m.state_dict()
    assert [(k, list(v.shape), str(v.dtype).replace('torch.', '')) for k, v in sd.items()] == \
        [tuple(x) if False else (x[0], x[1], x[2]) for x in c['keys']]
    assert [k for k, _ in m.named_parameters()] == [k for k, _ in c['params']]
    assert sum(p.numel() for p in m.parameters()) == c['n_params'] == 37968692
    one_d = [k for k, p in m.named_parameters() if p.ndim == 1]
    assert one_d == c['one_dim_params'] and len(one_d) == 124


def test_no_cpu_fallback():
    import torch
    from sacv_amd.detection import models, losses, decode
    m = models.__dict__['resnet50_retinanet']()
    with pytest.raises(RuntimeError, match='MI355X'):
        m(torch.zeros(1, 3, 64, 64))
    crit = losses.__dict__['RetinaLoss']()
    with pytest.raises(RuntimeError, match='MI355X'):
        crit([[torch.zeros(1, 2, 2, 9, 80)], [torch.zeros(1, 2, 2, 9, 4)]], torch.zeros(1, 3, 5))
    dec = decode.__dict__['RetinaDecoder']()
    with pytest.raises(RuntimeError, match='MI355X'):
        dec([[torch.zeros(1, 2, 2, 9, 80)], [torch.zeros(1, 2, 2, 9, 4)]])
    decode.__dict__['RetinaDecoder'](nms_type='torch_nms')  # torchvision.ops.nms semantics, no torchvision needed
    # the north-star's extra mode (not in the reference's DetNMSMethod): linear / gaussian Soft-NMS
    for t in ('soft_nms', 'soft_nms_linear', 'soft_nms_gaussian'):
        assert decode.__dict__['RetinaDecoder'](nms_type=t).soft is not None
        assert decode.__dict__['FCOSDecoder'](nms_type=t).soft is not None
    with pytest.raises(AssertionError):  # the reference's message: 'wrong nms type!'
        decode.__dict__['RetinaDecoder'](nms_type='cluster_nms')


def test_bench_generator_matches_fixture_generator():
    """bench.py draws its synthetic batches from the product package (it must not import oracle/ outside its
    cpu_baseline leg); the recipe is the one the fixtures were generated with."""
    import numpy as np
    from oracle import synth as osynth
    from sacv_amd import synthetic
    assert np.array_equal(osynth.images(3, 1, 17, 23), synthetic.images(3, 1, 17, 23))
    assert np.array_equal(osynth.annotations(5, 2, 800, 1333), synthetic.annotations(5, 2, 800, 1333))
    src = open(os.path.join(ROOT, 'bench.py')).read()
    head, _, tail = src.partition('def cpu_baseline')
    body_after = tail.split('\ndef main', 1)[1]
    assert 'oracle' not in head.replace('oracle/', '').split('import argparse')[1].split('def ')[0]
    assert 'from oracle' not in body_after and 'import oracle' not in body_after


def test_arena_channels_last_weights_keep_the_state_dict_contract():
    """k x k conv weights live channels_last in the flat arena (= the forward GEMM panel) but stay OIHW-shaped
    parameters: state_dict values, load_state_dict and p.grad are unchanged for the caller."""
    import torch
    from sacv_amd.classification import backbones
    from sacv_amd.detection.models._arena import ParamArena
    from oracle import model as omodel
    m = backbones.__dict__['resnet50'](num_classes=16)
    sd0 = omodel.init_resnet50_cls_state(0, 16)
    m.load_state_dict(sd0, strict=True)
    arena = ParamArena(m)
    w = m.layer1[0].conv2.layer[0].weight
    assert tuple(w.shape) == (64, 64, 3, 3) and not w.is_contiguous() and w.permute(0, 2, 3, 1).is_contiguous()
    assert m.conv1.layer[0].weight.is_contiguous()          # Cin = 3: stays OIHW (packed into the stem panel)
    assert m.layer1[0].conv1.layer[0].weight.is_contiguous()  # 1x1: both layouts coincide
    for k, v in m.state_dict().items():
        assert torch.equal(v, sd0[k]), k
    # the arena slice of a 3x3 weight is the [Cout][KH][KW][Cin] panel
    o = arena.offsets['layer1.0.conv2.layer.0.weight']
    assert torch.equal(arena.params[o:o + w.numel()].view(64, 3, 3, 64), sd0['layer1.0.conv2.layer.0.weight'].permute(0, 2, 3, 1))
    sd1 = {k: (v + 1 if v.dtype.is_floating_point else v) for k, v in sd0.items()}
    m.load_state_dict(sd1, strict=True)
    assert arena.is_current()
    assert torch.equal(w, sd1['layer1.0.conv2.layer.0.weight'])
    arena.attach_grads()
    assert w.grad.shape == w.shape and w.grad.stride() == w.stride()
