"""Split remainder tiles of the implicit-GEMM convolution (stream-K / uniform K split): partial slabs of a tile meet in
the workspace and are summed in a fixed order (csrc/conv_gemm.hip: conv_fixup_kernel).  These tests hammer the slab
path: alternating inputs through the same slab addresses (a stale slab would surface as the previous problem's values),
uneven load from a second stream, run-to-run bit stability.  (Written for round 4's in-kernel finish by the last-arriving
workgroup -- measured a tie with the fix-up launch and not adopted -- and kept for the launch that stayed.)"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    if not torch.cuda.is_available():
        pytest.skip('needs the MI355X')
    from sacv_amd import ops as o
    return o


@pytest.fixture(autouse=True, params=['fp32_pipe', 'bf16x3_pipe'])
def matrix_pipe(request, ops):
    """Every test of this module runs twice: sacv_conv2d_gemm on the fp32 matrix pipe, and with the operands split into
    three bf16 terms at the LDS store (six products on the bf16 pipe, fp32 accumulators; csrc/conv_gemm.hip SP) -- the
    same tolerances: the split form carries the fp32 pipe's error."""
    prev = ops.launch_share(nt_split=int(request.param == 'bf16x3_pipe'))
    yield request.param
    ops.launch_share(*prev)


def rnd(seed, *shape, std=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * std


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().cuda()


# shapes whose tile counts leave a remainder on the 512 / 768 persistent workgroups (layer3 / layer4 of the trunk at two
# images, a long-K 3x3 and a ragged one): (N, Cin, H, W, Cout, k)
SPLIT_CASES = [(2, 1024, 50, 84, 256, 1), (2, 256, 50, 84, 1024, 1), (2, 512, 25, 42, 512, 3), (2, 2048, 25, 42, 512, 1),
               (2, 256, 50, 84, 256, 3), (3, 96, 37, 41, 160, 3)]


@pytest.mark.parametrize('case', SPLIT_CASES)
def test_split_tiles_alternating_inputs_under_load(ops, case):
    n, cin, h, w, cout, k = case
    p = k // 2
    wt = rnd(2, cout, cin, k, k, std=(1.0 / (cin * k * k)) ** 0.5)
    panel = ops.pack_weight(wt.cuda(), ops.PACK_FWD)
    xs = [rnd(10 + i, n, cin, h, w) for i in range(2)]
    refs = [F.conv2d(x, wt, None, 1, p) for x in xs]
    xd = [nhwc(x) for x in xs]
    side = torch.cuda.Stream()
    # uneven load: a second queue keeps part of the chip busy with weight-gradient GEMMs of another problem
    ldy, lx = nhwc(rnd(3, 2, 128, 100, 167)), nhwc(rnd(4, 2, 128, 100, 167))
    ldw = torch.empty((128, 128, 3, 3), device='cuda')
    first = [None, None]
    for it in range(24):
        if it % 3 == 0:
            with torch.cuda.stream(side):
                ops.WS_SUFFIX = '@side'
                try:
                    for _ in range(4):
                        ops.conv_wgrad([ldy], [lx], ldw, 128, 3, 3, 1, 1)
                finally:
                    ops.WS_SUFFIX = ''
        j = it & 1
        (y,), st = ops.conv_gemm([xd[j]], panel, cout, k, k, 1, p, 0, [(h, w)], stats=True)
        if first[j] is None:
            first[j] = (y.clone(), st.clone())
            got = y.permute(0, 3, 1, 2).cpu()
            assert float((got - refs[j]).abs().max()) <= 2e-5 * float(refs[j].abs().max())
            rows = y.reshape(-1, cout).double()
            assert torch.allclose(st[:, 0].double().sum(0), rows.sum(0), rtol=1e-5, atol=1e-3)
            assert torch.allclose(st[:, 1].double().sum(0), (rows * rows).sum(0), rtol=1e-5, atol=1e-3)
        else:
            assert torch.equal(y, first[j][0]), f'iteration {it}: output differs from the first run of input {j}'
            assert torch.equal(st, first[j][1]), f'iteration {it}: column sums differ'
    torch.cuda.synchronize()


def test_split_tiles_data_gradient_epilogues(ops):
    """Accumulate + ReLU-mask + column sums on split tiles (the data gradient of a head / trunk layer)."""
    n, cin, h, w, cout = 2, 256, 50, 84, 256
    wt = rnd(2, cout, cin, 3, 3, std=(1.0 / (cin * 9)) ** 0.5)
    x = rnd(3, n, cin, h, w).requires_grad_(True)
    y = F.conv2d(x, wt, None, 1, 1)
    dy = rnd(4, *y.shape)
    y.backward(dy)
    acc, msk = rnd(5, n, cin, h, w), rnd(6, n, cin, h, w)
    want = (x.grad + acc) * (msk > 0)
    wd = ops.pack_weight(wt.cuda(), ops.PACK_DGRAD)
    for it in range(6):
        out = nhwc(acc)
        (dx,), st = ops.conv_gemm([nhwc(dy)], wd, cin, 3, 3, 1, 1, 1, [(h, w)], flags=ops.CF_ACCUM | ops.CF_MASK,
                                  masks=[nhwc(msk)], outs=[out], stats=True)
        got = dx.permute(0, 3, 1, 2).cpu()
        assert float((got - want).abs().max()) <= 3e-5 * float(want.abs().max())
        assert torch.allclose(st[:, 0].double().sum(0), dx.reshape(-1, cin).double().sum(0), rtol=1e-5, atol=1e-3)
