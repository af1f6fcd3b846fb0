"""torch.library registration of the detector (sacv_amd/custom_ops.py): schemas, FakeTensor implementations and the
autograd formula are checked WITHOUT a GPU -- fake tensors carry the geometry only.  The real run under torch.compile
is tests/test_gpu_trainer.py::test_torch_compile_captures_the_detector_as_one_operator."""
import pytest
import torch


def _model(kind):
    from sacv_amd.detection import models
    from sacv_amd import custom_ops
    net = models.__dict__['resnet50_fcos' if kind == 'fcos' else 'resnet50_retinanet'](num_classes=80)
    h = custom_ops.register_model(net)
    net._param_list = list(net.named_parameters())  # what prepare() records on the GPU (arena order there)
    return net, h, custom_ops


def test_operators_are_registered_with_the_expected_schemas():
    from sacv_amd import custom_ops  # noqa: F401
    fwd = torch.ops.sacv.detector_forward.default._schema
    bwd = torch.ops.sacv.detector_backward.default._schema
    assert str(fwd) == ('sacv::detector_forward(Tensor images, Tensor[] params, SymInt handle, bool training, '
                        'bool keep) -> Tensor[]')
    assert str(bwd) == 'sacv::detector_backward(Tensor[] grads, SymInt handle) -> Tensor[]'


def test_level_sizes_match_the_survey_shapes():
    from sacv_amd.custom_ops import level_sizes
    assert level_sizes(800, 1333) == [(100, 167), (50, 84), (25, 42), (13, 21), (7, 11)]
    assert level_sizes(1333, 1333) == [(167, 167), (84, 84), (42, 42), (21, 21), (11, 11)]   # S2


@pytest.mark.parametrize('kind', ['retina', 'fcos'])
def test_fake_forward_and_backward_shapes(kind):
    from torch._subclasses.fake_tensor import FakeTensorMode
    net, h, co = _model(kind)
    with FakeTensorMode(allow_non_fake_inputs=True):
        images = torch.empty((2, 3, 800, 1333), device='cuda')
        params = [torch.empty(p.shape, device='cuda', requires_grad=True) for _, p in net._param_list]
        outs = torch.ops.sacv.detector_forward(images, params, h, True, True)
        want = co.output_shapes(net, 2, 800, 1333)
        assert [tuple(o.shape) for o in outs] == want and len(outs) == (15 if kind == 'fcos' else 10)
        assert all(o.requires_grad for o in outs)  # the registered autograd formula is attached
        if kind == 'retina':
            assert tuple(outs[0].shape) == (2, 100, 167, 9, 80) and tuple(outs[9].shape) == (2, 7, 11, 9, 4)
        # the backward operator's fake implementation: one gradient per parameter, the parameter's shape (running the
        # autograd engine itself on fake cuda tensors needs a device context: that is the GPU test's part)
        gp = torch.ops.sacv.detector_backward([torch.empty_like(o) for o in outs], h)
        assert [tuple(g.shape) for g in gp] == [tuple(p.shape) for _, p in net._param_list]


def test_a_dead_handle_fails_loudly():
    from sacv_amd import custom_ops  # noqa: F401
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode(allow_non_fake_inputs=True):
        with pytest.raises(RuntimeError, match='not alive'):
            torch.ops.sacv.detector_forward(torch.empty((1, 3, 64, 64), device='cuda'), [], 10 ** 9, False, False)
