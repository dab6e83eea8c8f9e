"""GPU parity of every C-ABI entry point of libtvae_hip.so against plain torch fp64/fp32 math.
These call through the C ABI (tvae._lib.call -> ctypes) on cuda:0."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err
from oracle import tvae_oracle as O

pytestmark = pytest.mark.gpu
TOL = 2e-5
SLOPE = 0.01
# per-tensor relative tolerance of the fp32-MFMA entry points (the *_x6 / *_dft entry points are held to the same
# number in their own tests below); the hot-path parity gate is 1e-4 (BASELINE.json north_star)
GEMM_TOL = {'f32': 2e-5}


@pytest.fixture(params=['f32'])
def gemm_mode(request):
    return request.param


def dev():
    return torch.device('cuda:0')


def call(*a):
    from tvae._lib import call as c
    return c(*a)


def query(*a):
    from tvae._lib import query as q
    return q(*a)


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale)


def act_ref(x, act):
    return x if act == 0 else (F.leaky_relu(x, SLOPE) if act == 1 else torch.tanh(x))


def dact_ref(y, act):
    if act == 0:
        return torch.ones_like(y)
    return torch.where(y > 0, torch.ones_like(y), torch.full_like(y, SLOPE)) if act == 1 else 1 - y * y


@pytest.mark.parametrize('M,N,K,act,use_g,use_res', [
    (128, 8712, 128, 1, False, False), (7, 300, 128, 0, False, False), (512, 1000, 1024, 1, True, False),
    (64, 517, 64, 1, False, True), (130, 129, 17, 2, True, True), (103, 2000, 128, 0, False, False),
    (128, 1024, 128, 1, True, True), (256, 512, 64, 0, False, False), (512, 2048, 512, 1, True, False),
    (128, 1024, 128, 1, False, True), (256, 1152, 32, 2, False, True), (384, 128, 16, 1, False, False)])
def test_linear_fwd(M, N, K, act, use_g, use_res, gemm_mode):
    W, X, b = rnd(M, K, seed=1, scale=K ** -0.5), rnd(K, N, seed=2), rnd(M, seed=3)
    group = 100
    ng = (N + group - 1) // group
    gb = rnd(ng, M, seed=4) if use_g else None
    res = rnd(M, N, seed=5) if use_res else None
    ref = W.double() @ X.double() + b.double()[:, None]
    if use_g:
        ref = ref + gb.double()[torch.arange(N) // group].t()
    if use_res:
        ref = ref + res.double()
    ref = act_ref(ref, act)
    Y = torch.empty(M, N, device=dev())
    call('tvae_linear_fwd', W.to(dev()), X.to(dev()), b.to(dev()), gb.to(dev()) if use_g else None, group,
         res.to(dev()) if use_res else None, Y, M, N, K, N, N, act, SLOPE)
    assert rel_err(Y, ref) < GEMM_TOL[gemm_mode]


@pytest.mark.parametrize('M,N,K,mask,use_add', [(7, 3000, 128, 1, False), (128, 1111, 128, 1, False),
                                                (64, 300, 64, 1, True), (512, 700, 1024, 0, False),
                                                (33, 257, 65, 2, True), (128, 1024, 128, 1, True), (64, 512, 256, 1, False),
                                                (512, 1024, 512, 0, False), (128, 512, 256, 2, True), (16, 1280, 384, 1, False)])
def test_linear_dgrad(M, N, K, mask, use_add, gemm_mode):
    W, d = rnd(M, K, seed=1, scale=M ** -0.5), rnd(M, N, seed=2)
    aux = rnd(K, N, seed=3).clamp(-0.9, 0.9)
    add = rnd(K, N, seed=4) if use_add else None
    ref = W.double().t() @ d.double()
    if use_add:
        ref = ref + add.double()
    ref = ref * dact_ref(aux.double(), mask)
    dX = torch.empty(K, N, device=dev())
    call('tvae_linear_dgrad', W.to(dev()), d.to(dev()), add.to(dev()) if use_add else None,
         aux.to(dev()) if mask else None, dX, M, N, K, N, N, mask, SLOPE)
    assert rel_err(dX, ref) < GEMM_TOL[gemm_mode]


@pytest.mark.parametrize('M,N,K,acc', [(7, 50000, 128, 0), (128, 8712 * 3, 128, 0), (64, 1000, 2, 1),
                                       (512, 20000, 512, 0), (130, 777, 1024, 0), (128, 8192, 128, 0), (256, 4096, 512, 1)])
def test_linear_wgrad(M, N, K, acc, gemm_mode):
    d, X = rnd(M, N, seed=1), rnd(K, N, seed=2)
    init = rnd(M, K, seed=3)
    ref = d.double() @ X.double().t() + (init.double() if acc else 0)
    dW = init.clone().to(dev()) if acc else torch.empty(M, K, device=dev())
    ws = torch.empty(1 << 24, device=dev())
    call('tvae_linear_wgrad', d.to(dev()), X.to(dev()), dW, ws, ws.numel(), M, N, K, N, N, acc)
    assert rel_err(dW, ref) < GEMM_TOL[gemm_mode]


def test_linear_wgrad_no_workspace():
    M, N, K = 20, 5000, 30
    d, X = rnd(M, N, seed=1), rnd(K, N, seed=2)
    dW = torch.empty(M, K, device=dev())
    call('tvae_linear_wgrad', d.to(dev()), X.to(dev()), dW, None, 0, M, N, K, N, N, 0)
    assert rel_err(dW, d.double() @ X.double().t()) < TOL


@pytest.mark.parametrize('k,R,Cin,C', [(5, 4, 1, 3), (28, 8, 1, 4), (9, 16, 3, 2), (7, 8, 3, 11), (32, 8, 1, 128)])
def test_rotate_bank(k, R, Cin, C):
    from tvae import ops
    w = rnd(C, Cin, 1, k, k, seed=k)
    wr = w.clone().requires_grad_(True)
    ref = O.rotated_bank(wr, R)                      # (C,R,Cin,1,k,k)
    wg = w.to(dev()).requires_grad_(True)
    bank = ops.BankFn.apply(wg, R)
    assert rel_err(bank.view(C, R, Cin, 1, k, k), ref) < 1e-6
    g = rnd(*ref.shape, seed=9)
    (ref * g).sum().backward()
    (bank.view(C, R, Cin, 1, k, k) * g.to(dev())).sum().backward()
    assert rel_err(wg.grad, wr.grad) < 1e-5


@pytest.mark.parametrize('B,Cin,n,k,pad,C,R,act', [(2, 1, 28, 28, 8, 8, 8, 1), (3, 3, 12, 9, 3, 4, 4, 0),
                                                   (2, 1, 64, 64, 16, 4, 8, 1), (5, 2, 20, 7, 0, 3, 16, 0),
                                                   (3, 1, 64, 64, 16, 16, 8, 1), (2, 2, 40, 32, 6, 32, 4, 0),
                                                   (2, 1, 20, 16, 3, 8, 16, 1)])
def test_conv1_fwd_wgrad(B, Cin, n, k, pad, C, R, act, gemm_mode):
    y = torch.rand(B, Cin, n, n, generator=torch.Generator().manual_seed(1))
    bank = rnd(C * R, Cin * k * k, seed=2, scale=(Cin * k * k) ** -0.5)
    bias = rnd(C, seed=3, scale=0.1)
    Ho = n + 2 * pad - k + 1
    ref = F.conv2d(y.double(), bank.double().view(C * R, Cin, k, k), None, 1, pad).view(B, C, R, Ho, Ho) \
        + bias.double().view(1, C, 1, 1, 1)
    ref = act_ref(ref, act)
    out = torch.empty(C, B * R * Ho * Ho, device=dev())
    call('tvae_conv1_fwd', y.to(dev()), bank.to(dev()), bias.to(dev()), out, B, Cin, n, k, pad, C, R, act, SLOPE)
    got = out.view(C, B, R, Ho, Ho).permute(1, 0, 2, 3, 4)
    assert rel_err(got, ref) < GEMM_TOL[gemm_mode]
    # weight gradient
    g = rnd(B, C, R, Ho, Ho, seed=4)
    ref_g = torch.nn.grad.conv2d_weight(y.double(), (C * R, Cin, k, k), g.double().view(B, C * R, Ho, Ho), padding=pad)
    dpre = g.permute(1, 0, 2, 3, 4).contiguous().view(C, -1).to(dev())
    dbank = torch.empty(C * R, Cin * k * k, device=dev())
    ws = torch.empty(1 << 22, device=dev())
    call('tvae_conv1_wgrad', y.to(dev()), dpre, dbank, ws, ws.numel(), B, Cin, n, k, pad, C, R)
    assert rel_err(dbank.view(C * R, Cin, k, k), ref_g) < GEMM_TOL[gemm_mode]


@pytest.mark.parametrize('B,Cin,n,k,pad,C,R,act', [(2, 1, 28, 28, 8, 8, 8, 1), (3, 3, 12, 9, 3, 4, 4, 0),
                                                   (2, 1, 64, 64, 16, 4, 8, 1), (5, 2, 20, 7, 0, 3, 16, 0),
                                                   (3, 1, 64, 64, 16, 16, 8, 1), (2, 1, 40, 32, 6, 32, 4, 0),
                                                   (2, 1, 20, 16, 3, 8, 16, 1), (17, 1, 64, 64, 16, 48, 8, 1)])
def test_conv1_x6_matches_fp64_at_fp32_tolerance(B, Cin, n, k, pad, C, R, act):
    """3xbf16-split conv kernels: same tolerance as the exact-fp32 MFMA path (GEMM_TOL['f32'])."""
    from tvae._lib import query
    if not query('tvae_conv1_x6_supported', Cin, n, k, pad):
        pytest.skip('geometry does not fit the LDS-resident x6 kernels')
    y = torch.rand(B, Cin, n, n, generator=torch.Generator().manual_seed(1))
    bank = rnd(C * R, Cin * k * k, seed=2, scale=(Cin * k * k) ** -0.5)
    bias = rnd(C, seed=3, scale=0.1)
    Ho = n + 2 * pad - k + 1
    ref = F.conv2d(y.double(), bank.double().view(C * R, Cin, k, k), None, 1, pad).view(B, C, R, Ho, Ho) \
        + bias.double().view(1, C, 1, 1, 1)
    ref = act_ref(ref, act)
    a3 = torch.empty(query('tvae_conv1_x6_bank_bytes', C, R, Cin, k) // 4, device=dev())
    call('tvae_bank_split3', bank.to(dev()), a3, a3.numel() * 4, C, R, Cin, k)
    out = torch.empty(C, B * R * Ho * Ho, device=dev())
    call('tvae_conv1_fwd_x6', y.to(dev()), a3, bias.to(dev()), out, B, Cin, n, k, pad, C, R, act, SLOPE)
    got = out.view(C, B, R, Ho, Ho).permute(1, 0, 2, 3, 4)
    assert rel_err(got, ref) < GEMM_TOL['f32']
    g = rnd(B, C, R, Ho, Ho, seed=4)
    ref_g = torch.nn.grad.conv2d_weight(y.double(), (C * R, Cin, k, k), g.double().view(B, C * R, Ho, Ho), padding=pad)
    dpre = g.permute(1, 0, 2, 3, 4).contiguous().view(C, -1).to(dev())
    dbank = torch.empty(C * R, Cin * k * k, device=dev())
    ws = torch.empty(1 << 22, device=dev())
    d3 = torch.empty(query('tvae_conv1_x6_dy_bytes', B, C, R, n, k, pad) // 4, device=dev())
    call('tvae_dy_split3', dpre, d3, d3.numel() * 4, B, Cin, n, k, pad, C, R)
    call('tvae_conv1_wgrad_x6', y.to(dev()), d3, dbank, ws, ws.numel(), B, Cin, n, k, pad, C, R)
    assert rel_err(dbank.view(C * R, Cin, k, k), ref_g) < GEMM_TOL['f32']


@pytest.mark.parametrize('M,N,K,act,use_res', [(512, 2048, 512, 1, False), (128, 1024, 128, 1, True), (300, 384, 40, 2, False),
                                               (512, 128, 512, 0, True), (64, 256, 1024, 1, False)])
def test_linear_fwd_dgrad_x6(M, N, K, act, use_res):
    """Dense layers in the exact-split bf16 arithmetic: same tolerance as the fp32 MFMA path."""
    from tvae._lib import query
    W, X, b = rnd(M, K, seed=1, scale=K ** -0.5), rnd(K, N, seed=2), rnd(M, seed=3)
    res = rnd(M, N, seed=5) if use_res else None
    ref = W.double() @ X.double() + b.double()[:, None]
    if use_res:
        ref = ref + res.double()
    ref = act_ref(ref, act)
    w3 = torch.empty(query('tvae_dense_x6_bytes', M, K) // 4, device=dev())
    Wd = W.to(dev())
    call('tvae_dense_split3', Wd, K, w3, w3.numel() * 4, M, K, 0, None, None)
    Y = torch.empty(M, N, device=dev())
    cw, cb = rnd(M, seed=11), rnd(1, seed=12)
    cy = torch.empty(N, device=dev())
    fuse = M <= 512
    call('tvae_linear_fwd_x6', w3, X.to(dev()), b.to(dev()), res.to(dev()) if use_res else None, Y, M, N, K, N, N, act,
         SLOPE, cw.to(dev()) if fuse else None, cb.to(dev()) if fuse else None, cy if fuse else None, None, None, None, None, 0, None, 3)
    if fuse:
        assert rel_err(cy, cw.double() @ ref + cb.double()) < GEMM_TOL['f32']
    assert rel_err(Y, ref) < GEMM_TOL['f32']
    # data gradient: dX[k][n] = act'(aux) * (add + sum_m W[m][k] d[m][n])
    d = rnd(M, N, seed=6)
    aux = rnd(K, N, seed=7).clamp(-0.9, 0.9)
    add = rnd(K, N, seed=8) if use_res else None
    refg = W.double().t() @ d.double()
    if use_res:
        refg = refg + add.double()
    refg = refg * dact_ref(aux.double(), act)
    w3t = torch.empty(query('tvae_dense_x6_bytes', K, M) // 4, device=dev())
    call('tvae_dense_split3', Wd, K, w3t, w3t.numel() * 4, K, M, 1, None, None)
    dX = torch.empty(K, N, device=dev())
    call('tvae_linear_dgrad_x6', w3t, d.to(dev()), add.to(dev()) if use_res else None, aux.to(dev()) if act else None,
         dX, M, N, K, N, N, act, SLOPE, None, None, None, None, 0, None, None, None, None, None, 0, None, 0, None, None, None, None, 3)
    assert rel_err(dX, refg) < GEMM_TOL['f32']
    with pytest.raises(Exception):
        call('tvae_linear_fwd_x6', w3, X.to(dev()), b.to(dev()), None, Y, M, N - 1, K, N, N, act, SLOPE, None, None, None, None, None, None, None, 0, None, 3)


@pytest.mark.parametrize('M,N,K,acc', [(512, 20000 // 16 * 16, 512, 0), (128, 8192, 128, 0), (300, 1600, 70, 1), (512, 4096, 512, 1)])
def test_linear_wgrad_x6(M, N, K, acc):
    d, X = rnd(M, N, seed=1), rnd(K, N, seed=2)
    init = rnd(M, K, seed=3)
    ref = d.double() @ X.double().t() + (init.double() if acc else 0)
    dW = init.clone().to(dev()) if acc else torch.empty(M, K, device=dev())
    ws = torch.empty(1 << 26, device=dev())       # >= tvae_linear_wgrad_x6_ws_floats(M, N, K): slices x M x K
    from tvae._lib import query
    need = query('tvae_linear_wgrad_x6_ws_floats', M, N, K)
    assert 2 * M * K <= need <= ws.numel()
    call('tvae_linear_wgrad_x6', d.to(dev()), X.to(dev()), dW, ws, ws.numel(), M, N, K, N, N, acc, None, None, 0, SLOPE, None, None, None, None, 0, None, 3)
    assert rel_err(dW, ref) < GEMM_TOL['f32']
    dW2 = init.clone().to(dev()) if acc else torch.empty(M, K, device=dev())
    call('tvae_linear_wgrad_x6', d.to(dev()), X.to(dev()), dW2, ws, need, M, N, K, N, N, acc, None, None, 0, SLOPE, None, None, None, None, 0, None, 3)
    assert torch.equal(dW, dW2)          # the summation order does not depend on how much workspace is offered
    with pytest.raises(Exception):
        call('tvae_linear_wgrad_x6', d.to(dev()), X.to(dev()), dW2, ws, need - 1, M, N, K, N, N, acc, None, None, 0, SLOPE, None, None, None, None, 0, None, 3)
    with pytest.raises(Exception):
        call('tvae_linear_wgrad_x6', d.to(dev()), X.to(dev()), dW, ws, ws.numel(), M, N - 8, K, N, N, acc, None, None, 0, SLOPE, None, None, None, None, 0, None, 3)


@pytest.mark.parametrize('B,n,k,pad,C,R,act,Cin', [
    (2, 28, 28, 8, 32, 8, 1, 1), (3, 64, 64, 16, 32, 8, 1, 1), (5, 40, 32, 6, 64, 4, 0, 1), (17, 64, 64, 16, 64, 8, 1, 1),
    (4, 32, 32, 8, 16, 8, 1, 1), (3, 20, 9, 2, 5, 4, 0, 1),
    # the three frames of the ring (LDS-DMA) transforms along w, with several tiles per wave (steady-state vmcnt
    # bookkeeping: iterations 0, 1 and >= 2 differ), ragged last column tiles, tanh; and the real 50x50 MNIST-U geometry
    (40, 28, 28, 8, 32, 8, 1, 1), (24, 64, 64, 16, 16, 8, 1, 1), (12, 50, 28, 8, 32, 8, 1, 1), (2, 50, 28, 8, 8, 4, 2, 1),
    (3, 28, 28, 8, 16, 8, 2, 1),
    # several input channels (they join the reduction of the spectral GEMM)
    (3, 28, 28, 8, 16, 8, 1, 3), (2, 20, 9, 2, 5, 4, 0, 2),
    # frames beyond the specialised transforms along w (Lh > 64 or Ho > 64): spectra in frequency blocks, generic
    # transforms -- the galaxy configuration's shape class (192-wide frame, 3 channels) at a small size
    (2, 96, 32, 16, 4, 4, 1, 2), (2, 128, 64, 32, 2, 16, 1, 3),
    # round 4: the spectral GEMM kernels of their own shape at their edges -- one image (a single 128-column panel, two k-steps
    # per reduction slice), 2 M = 512 / 1 024 rows; and the weight gradient's 192-column tiles on a 1 152-column problem
    # (galaxy frame, three channels) with 2 M = 512 rows
    (1, 64, 64, 16, 32, 8, 1, 1), (2, 64, 64, 16, 32, 16, 1, 1), (3, 128, 64, 32, 16, 16, 1, 3),
    # round 5, the short circular frame L = max(n + pad, k) rounded up to 4 (csrc/abi_conv_dft.hip: dft_plan): a filter longer
    # than n + pad (frame = k), no padding at all (frame = n), odd sizes whose rounding stops at n + 2 pad, padding >= k
    (3, 20, 40, 12, 4, 4, 1, 1), (2, 24, 9, 0, 4, 4, 0, 1), (2, 21, 7, 3, 4, 4, 1, 1), (2, 19, 5, 1, 4, 4, 1, 2),
    (2, 16, 6, 8, 4, 4, 0, 1)])
@pytest.mark.parametrize('nparts', [3, 2])
def test_conv1_dft_matches_fp64(B, n, k, pad, C, R, act, Cin, nparts):
    """Frequency-domain lifting convolution (DFT + batched split-pipe GEMM): fp32-level agreement with fp64, in the exact
    bf16 split (3 parts, six products) and in the h3 arithmetic (2 fp16 parts, three products)."""
    from tvae._lib import query
    if not query('tvae_conv1_dft_supported', B, Cin, n, k, pad, C, R):
        pytest.skip('geometry not handled by the frequency-domain path')
    y = torch.rand(B, Cin, n, n, generator=torch.Generator().manual_seed(1))
    bank = rnd(C * R, Cin * k * k, seed=2, scale=(Cin * k * k) ** -0.5)
    bias = rnd(C, seed=3, scale=0.1)
    Ho = n + 2 * pad - k + 1
    ref = F.conv2d(y.double(), bank.double().view(C * R, Cin, k, k), None, 1, pad).view(B, C, R, Ho, Ho) \
        + bias.double().view(1, C, 1, 1, 1)
    ref = act_ref(ref, act)
    at = torch.zeros(query('tvae_conv1_dft_at_floats', B, Cin, n, k, pad, C, R), device=dev())
    ws = torch.empty(query('tvae_conv1_dft_ws_floats', B, Cin, n, k, pad, C, R), device=dev())
    out = torch.empty(C, B * R * Ho * Ho, device=dev())
    call('tvae_conv1_fwd_dft', y.to(dev()), bank.to(dev()), bias.to(dev()), out, at, ws, ws.numel(), B, Cin, n, k, pad,
         C, R, act, SLOPE, nparts)
    got = out.view(C, B, R, Ho, Ho).permute(1, 0, 2, 3, 4)
    assert rel_err(got, ref) < GEMM_TOL['f32']
    g = rnd(B, C, R, Ho, Ho, seed=4)
    ref_g = torch.nn.grad.conv2d_weight(y.double(), (C * R, Cin, k, k), g.double().view(B, C * R, Ho, Ho), padding=pad)
    dpre = g.permute(1, 0, 2, 3, 4).contiguous().view(C, -1).to(dev())
    dbank = torch.empty(C * R, Cin * k * k, device=dev())
    dbias = torch.empty(C, device=dev())
    call('tvae_conv1_wgrad_dft', dpre, at, dbank, dbias, ws, ws.numel(), B, Cin, n, k, pad, C, R, nparts)
    assert rel_err(dbias, g.double().sum(dim=(0, 2, 3, 4))) < TOL
    assert rel_err(dbank.view(C * R, Cin, k, k), ref_g) < GEMM_TOL['f32']


@pytest.mark.parametrize('B,n,k,pad,C,R,act', [
    (24, 64, 64, 16, 16, 8, 1), (40, 28, 28, 8, 32, 8, 1), (12, 50, 28, 8, 32, 8, 2),      # ring frames: bf16 STORAGE of T and S'
    (5, 40, 32, 6, 64, 4, 0), (3, 28, 28, 8, 16, 8, 1)])                                  # register-staged frame; ragged ring batch
def test_conv1_dft_bf16_mode(B, n, k, pad, C, R, act):
    """The one-part bf16 throughput mode of the frequency-domain convolution (parts = 1; never fp32-equivalent): operands
    rounded to bf16 and, on the ring frames, T and S' STORED as bf16 (round 4).  Own tolerance: 1.5e-2 relative Frobenius
    against fp64 for the output and the weight gradient (the exact modes hold 2e-5 on the same geometries)."""
    from tvae._lib import query
    Cin = 1
    y = torch.rand(B, Cin, n, n, generator=torch.Generator().manual_seed(1))
    bank = rnd(C * R, Cin * k * k, seed=2, scale=(Cin * k * k) ** -0.5)
    bias = rnd(C, seed=3, scale=0.1)
    Ho = n + 2 * pad - k + 1
    ref = act_ref(F.conv2d(y.double(), bank.double().view(C * R, Cin, k, k), None, 1, pad).view(B, C, R, Ho, Ho)
                  + bias.double().view(1, C, 1, 1, 1), act)
    at = torch.zeros(query('tvae_conv1_dft_at_floats', B, Cin, n, k, pad, C, R), device=dev())
    ws = torch.empty(query('tvae_conv1_dft_ws_floats', B, Cin, n, k, pad, C, R), device=dev())
    out = torch.full((C, B * R * Ho * Ho), float('nan'), device=dev())
    call('tvae_conv1_fwd_dft', y.to(dev()), bank.to(dev()), bias.to(dev()), out, at, ws, ws.numel(), B, Cin, n, k, pad, C, R, act,
         SLOPE, 1)
    got = out.view(C, B, R, Ho, Ho).permute(1, 0, 2, 3, 4)
    assert torch.isfinite(got).all()
    assert rel_err(got, ref) < 1.5e-2
    g = rnd(B, C, R, Ho, Ho, seed=4)
    ref_g = torch.nn.grad.conv2d_weight(y.double(), (C * R, Cin, k, k), g.double().view(B, C * R, Ho, Ho), padding=pad)
    dpre = g.permute(1, 0, 2, 3, 4).contiguous().view(C, -1).to(dev())
    dbank = torch.full((C * R, Cin * k * k), float('nan'), device=dev())
    dbias = torch.empty(C, device=dev())
    call('tvae_conv1_wgrad_dft', dpre, at, dbank, dbias, ws, ws.numel(), B, Cin, n, k, pad, C, R, 1)
    assert torch.isfinite(dbank).all()
    assert rel_err(dbank.view(C * R, Cin, k, k), ref_g) < 1.5e-2
    assert rel_err(dbias, g.double().sum(dim=(0, 2, 3, 4))) < 1.5e-2


@pytest.mark.parametrize('F_,B,Np,M,act', [(512, 3, 256, 512, 1), (300, 2, 384, 512, 1), (64, 2, 128, 200, 2)])
def test_linear_dgrad_x6_fused_first_layer(F_, B, Np, M, act):
    """Data gradient of the first hidden layer fused with the backward of the coordinate layer: dX is never stored."""
    from tvae._lib import query
    Nt = B * Np
    W, d = rnd(M, F_, seed=1, scale=M ** -0.5), rnd(M, Nt, seed=2)
    aux = rnd(F_, Nt, seed=3).clamp(-0.9, 0.9)
    xr, Wc = rnd(Nt, 2, seed=4), rnd(F_, 2, seed=5)
    d0 = (W.double().t() @ d.double()) * dact_ref(aux.double(), act)
    w3t = torch.empty(query('tvae_dense_x6_bytes', F_, M) // 4, device=dev())
    call('tvae_dense_split3', W.to(dev()), F_, w3t, w3t.numel() * 4, F_, M, 1, None, None)
    gxr = torch.empty(Nt, 2, device=dev())
    part = torch.empty((Nt // 128) * F_ * 3, device=dev())
    call('tvae_linear_dgrad_x6', w3t, d.to(dev()), None, aux.to(dev()), None, M, Nt, F_, Nt, Nt, act, SLOPE, xr.to(dev()),
         Wc.to(dev()), gxr, part, part.numel(), None, None, None, None, None, 0, None, 0, None, None, None, None, 3)
    Simg = torch.empty(B, F_, device=dev())
    dbc = torch.empty(F_, device=dev())
    dWc = torch.empty(F_, 2, device=dev())
    call('tvae_dec_in_total', part, B, Np // 128, F_, Simg, dbc, dWc)
    assert rel_err(gxr, d0.t() @ Wc.double()) < GEMM_TOL['f32']
    assert rel_err(Simg, d0.view(F_, B, Np).sum(2).t()) < GEMM_TOL['f32']
    assert rel_err(dbc, d0.sum(1)) < GEMM_TOL['f32']
    assert rel_err(dWc, d0 @ xr.double()) < GEMM_TOL['f32']


def test_linear_x6_implicit_gradient_operand():
    """dgrad / wgrad with the gradient operand formed on the fly from (Wo, gy, saved activation)."""
    from tvae._lib import query
    M, N, K = 512, 1024, 512
    W, H = rnd(M, K, seed=1, scale=M ** -0.5), rnd(M, N, seed=2).clamp(-0.9, 0.9)
    wo, gy = rnd(M, seed=3), rnd(N, seed=4)
    aux, X = rnd(K, N, seed=5).clamp(-0.9, 0.9), rnd(K, N, seed=6)
    d = wo.double()[:, None] * gy.double()[None, :] * dact_ref(H.double(), 1)
    w3t = torch.empty(query('tvae_dense_x6_bytes', K, M) // 4, device=dev())
    call('tvae_dense_split3', W.to(dev()), K, w3t, w3t.numel() * 4, K, M, 1, None, None)
    dX = torch.empty(K, N, device=dev())
    call('tvae_linear_dgrad_x6', w3t, H.to(dev()), None, aux.to(dev()), dX, M, N, K, N, N, 1, SLOPE, None, None, None,
         None, 0, wo.to(dev()), gy.to(dev()), None, None, None, 0, None, 0, None, None, None, None, 3)
    assert rel_err(dX, (W.double().t() @ d) * dact_ref(aux.double(), 1)) < GEMM_TOL['f32']
    # the same product in the two-valued form (LeakyReLU): weights scaled by wo before the split, 0 / 1 streamed operand
    csum = torch.empty(K, device=dev())
    call('tvae_dense_split3', W.to(dev()), K, w3t, w3t.numel() * 4, K, M, 1, wo.to(dev()), csum)
    assert rel_err(csum, (W.double() * wo.double()[:, None]).sum(0)) < TOL
    dX2 = torch.empty(K, N, device=dev())
    call('tvae_linear_dgrad_x6', w3t, H.to(dev()), None, aux.to(dev()), dX2, M, N, K, N, N, 1, SLOPE, None, None, None,
         None, 0, None, gy.to(dev()), csum, None, None, 0, None, 0, None, None, None, None, 3)
    assert rel_err(dX2, (W.double().t() @ d) * dact_ref(aux.double(), 1)) < GEMM_TOL['f32']
    # ... and with the row sums of the streamed activation against gy (ABI 3: what tvae_dec_out_bwd computes in a pass of its
    # own): the data gradient is bitwise the same launch result, db / dwo agree with fp64
    for Mr in (512, 384):
        Hr, wor = H[:Mr].contiguous(), wo[:Mr].contiguous()
        w3r = torch.empty(query('tvae_dense_x6_bytes', K, Mr) // 4, device=dev())
        csr = torch.empty(K, device=dev())
        call('tvae_dense_split3', W[:Mr].contiguous().to(dev()), K, w3r, w3r.numel() * 4, K, Mr, 1, wor.to(dev()), csr)
        dXa, dXb = torch.empty(K, N, device=dev()), torch.empty(K, N, device=dev())
        call('tvae_linear_dgrad_x6', w3r, Hr.to(dev()), None, aux.to(dev()), dXa, Mr, N, K, N, N, 1, SLOPE, None, None, None,
             None, 0, None, gy.to(dev()), csr, None, None, 0, None, 0, None, None, None, None, 3)
        rs_part = torch.full(((N // 128) * Mr * 2,), float('nan'), device=dev())
        gys = gy.sum().reshape(1).to(dev())
        rs_db, rs_dwo = torch.empty(Mr, device=dev()), torch.empty(Mr, device=dev())
        call('tvae_linear_dgrad_x6', w3r, Hr.to(dev()), None, aux.to(dev()), dXb, Mr, N, K, N, N, 1, SLOPE, None, None, None,
             None, 0, None, gy.to(dev()), csr, None, None, 0, rs_part, rs_part.numel(), wor.to(dev()), gys, rs_db, rs_dwo, 3)
        assert torch.equal(dXa, dXb)
        assert rel_err(rs_db, (wor.double()[:, None] * gy.double()[None, :] * dact_ref(Hr.double(), 1)).sum(1)) < TOL
        assert rel_err(rs_dwo, Hr.double() @ gy.double()) < TOL
    with pytest.raises(Exception):         # the row sums need the two-valued form (csum) and a workspace of N/128 * M * 2
        call('tvae_linear_dgrad_x6', w3t, H.to(dev()), None, aux.to(dev()), dX2, M, N, K, N, N, 1, SLOPE, None, None, None,
             None, 0, None, gy.to(dev()), csum, None, None, 0, rs_part, 16, wo.to(dev()), gys, rs_db, rs_dwo, 3)
    # packed sign bits of the activation (the weight gradient below can read its 0 / 1 operand from them)
    def pack_bits(t):                      # bit (n & 31) of word n / 32 = [t[m][n] > 0]
        b = (t > 0).to(torch.int64).view(t.shape[0], -1, 32)
        w = (b << torch.arange(32)).sum(-1)
        return (w - ((w >> 31) << 32)).to(torch.int32).contiguous()
    hb = pack_bits(H).to(dev())
    with pytest.raises(Exception):         # the two-valued form exists for LeakyReLU only
        call('tvae_linear_dgrad_x6', w3t, H.to(dev()), None, aux.to(dev()), dX2, M, N, K, N, N, 2, SLOPE, None, None, None,
             None, 0, None, gy.to(dev()), csum, None, None, 0, None, 0, None, None, None, None, 3)
    ws = torch.empty(1 << 26, device=dev())       # >= tvae_linear_wgrad_x6_ws_floats(M, N, K): slices x M x K
    for vact in (1, 2):                    # LeakyReLU: two-valued weight gradient; tanh: generic implicit operand
        dv = wo.double()[:, None] * gy.double()[None, :] * dact_ref(H.double(), vact)
        dW = torch.empty(M, K, device=dev())
        call('tvae_linear_wgrad_x6', H.to(dev()), X.to(dev()), dW, ws, ws.numel(), M, N, K, N, N, 0, wo.to(dev()),
             gy.to(dev()), vact, SLOPE, None, None, None, None, 0, None, 3)
        assert rel_err(dW, dv @ X.double().t()) < GEMM_TOL['f32'], vact
        if vact == 1:                      # the same from the packed sign bits: bitwise the same sums
            dWb = torch.empty(M, K, device=dev())
            call('tvae_linear_wgrad_x6', None, X.to(dev()), dWb, ws, ws.numel(), M, N, K, N, N, 0, wo.to(dev()),
                 gy.to(dev()), 1, SLOPE, None, None, None, None, 0, hb, 3)
            assert torch.equal(dWb, dW)
    # a LeakyReLU forward launch stores the sign bits of its output
    w3 = torch.empty(query('tvae_dense_x6_bytes', M, K) // 4, device=dev())
    call('tvae_dense_split3', W.to(dev()), K, w3, w3.numel() * 4, M, K, 0, None, None)
    Y = torch.empty(M, N, device=dev())
    yb = torch.empty(M, N // 32, dtype=torch.int32, device=dev())
    call('tvae_linear_fwd_x6', w3, X.to(dev()), None, None, Y, M, N, K, N, N, 1, SLOPE, None, None, None, None, None,
         None, None, 0, yb, 3)
    assert torch.equal(yb.cpu(), pack_bits(Y.cpu()))
    # dec_out_bwd without the gradient tensor: sums only
    F_ = M
    part = torch.empty(((N + 1023) // 1024) * F_ * 2, device=dev())
    tot = torch.empty(2, F_, device=dev())
    call('tvae_dec_out_bwd', gy.view(N, 1).to(dev()), 1, wo.view(1, F_).to(dev()), H.to(dev()), N, None, N, F_, N, 1,
         SLOPE, part, part.numel(), tot)
    assert rel_err(tot[0], d.sum(1)) < TOL and rel_err(tot[1], H.double() @ gy.double()) < TOL


@pytest.mark.parametrize('sw,sg,sx', [(1.0, 1.0, 1.0), (1e-3, 1e-6, 1.0), (30.0, 1e3, 50.0), (1.0, 0.0, 1.0), (1e-12, 1e-9, 1e-3)])
def test_h3_decoder_entry_points(sw, sg, sx):
    """The three decoder launches with an h3 instance (two fp16 parts per operand under a power-of-two tensor scale: three
    products per block, two against the exact 0 / 1 operand) against fp64 at the fp32 tolerance, and against the exact x6
    split of the same launch: forward with the recomputed first-layer operand (+ sign bits), two-valued data gradient (+ row
    sums), weight gradient from sign bits against gy x the recomputed operand.  Operand magnitudes from 1e-12 to 1e3 (the
    scales must absorb them: fp16 alone spans 6e-8 .. 6.5e4), a zero gradient (maximum 0)."""
    from tvae._lib import query
    F_, B, Np, M, act = 512, 2, 256, 512, 1
    Nt = B * Np
    xr, Wc, bc = (rnd(Nt, 2, seed=1) * sx).to(dev()), rnd(F_, 2, seed=2).to(dev()), rnd(F_, seed=3).to(dev())
    LB = rnd(B, F_, seed=4).to(dev())
    h0 = torch.empty(F_, Nt, device=dev())
    call('tvae_dec_l0_fwd', xr, Wc, bc, LB, h0, Nt, F_, Nt, Np, act, SLOPE)
    W, b = rnd(M, F_, seed=5, scale=sw * F_ ** -0.5), rnd(M, seed=6, scale=sw)
    wo, gy = rnd(M, seed=8), rnd(Nt, seed=9) * sg
    va = (xr, Wc, bc, LB, Np)
    H0 = h0.double().cpu()
    res = {}
    for nparts, split in ((3, 'tvae_dense_split3'), (2, 'tvae_dense_split2h')):
        w3 = torch.empty(query('tvae_dense_x6_bytes', M, F_) // 4, device=dev())
        call(split, W.to(dev()), F_, w3, w3.numel() * 4, M, F_, 0, None, None)
        Y = torch.empty(M, Nt, device=dev())
        bits = torch.empty(M, Nt // 32, dtype=torch.int32, device=dev())
        call('tvae_linear_fwd_x6', w3, None, b.to(dev()), None, Y, M, Nt, F_, Nt, Nt, act, SLOPE, None, None, None, *va, bits,
             nparts)
        # two-valued data gradient of THIS layer's input from (wo, gy, Y): weights W^T diag(wo), 0 / 1 operand [Y > 0]
        w3t = torch.empty(query('tvae_dense_x6_bytes', F_, M) // 4, device=dev())
        csum = torch.empty(F_, device=dev())
        call(split, W.to(dev()), F_, w3t, w3t.numel() * 4, F_, M, 1, wo.to(dev()), csum)
        dX = torch.empty(F_, Nt, device=dev())
        rs_part = torch.full(((Nt // 128) * M * 2,), float('nan'), device=dev())
        gys = gy.sum().reshape(1).to(dev())
        rs_db, rs_dwo = torch.empty(M, device=dev()), torch.empty(M, device=dev())
        call('tvae_linear_dgrad_x6', w3t, Y, None, h0, dX, M, Nt, F_, Nt, Nt, 1, SLOPE, None, None, None, None, 0, None,
             gy.to(dev()), csum, None, None, 0, rs_part, rs_part.numel(), wo.to(dev()), gys, rs_db, rs_dwo, nparts)
        ws = torch.empty(query('tvae_linear_wgrad_x6_ws_floats', M, Nt, F_), device=dev())
        dW = torch.empty(M, F_, device=dev())
        call('tvae_linear_wgrad_x6', None, None, dW, ws, ws.numel(), M, Nt, F_, Nt, Nt, 0, wo.to(dev()), gy.to(dev()), 1, SLOPE,
             *va, bits, nparts)
        res[nparts] = (Y, dX, dW, rs_db, rs_dwo, bits)
        assert torch.isfinite(Y).all() and torch.isfinite(dX).all() and torch.isfinite(dW).all()
    Yd = res[3][0].double().cpu()                     # masks from the x6 forward (the two forwards may differ in a sign at 0)
    ref_Y = act_ref(W.double() @ H0 + b.double()[:, None], act)
    d = wo.double()[:, None] * gy.double()[None, :] * dact_ref(Yd, 1)
    ref_dX = (W.double().t() @ d) * dact_ref(H0, 1)
    ref_dW = d @ H0.t()
    flips = int(((res[2][0] > 0) != (res[3][0] > 0)).sum())
    assert flips <= 2                                 # elements at zero to rounding
    for nparts in (3, 2):
        Y, dX, dW, rs_db, rs_dwo, bits = res[nparts]
        assert rel_err(Y, ref_Y) < GEMM_TOL['f32'], nparts
        if sg == 0.0:
            assert float(dX.abs().max()) == 0.0 and float(dW.abs().max()) == 0.0
            continue
        tol = GEMM_TOL['f32'] * (1 if (nparts == 3 or flips == 0) else 50)
        assert rel_err(dX, ref_dX) < tol, nparts
        assert rel_err(dW, ref_dW) < tol, nparts
    # h3 is not less accurate than the exact split by more than rounding (both are dominated by fp32 accumulation)
    e3, e2 = rel_err(res[3][0], ref_Y), rel_err(res[2][0], ref_Y)
    assert e2 < 2 * e3 + 1e-7, (e2, e3)
    # the recomputed operand with tanh (its bound |act(pre)| <= |pre| holds for every activation of the reference); unit scales
    # only: a saturated tanh of a difference of large terms is ill-conditioned in every arithmetic
    if (sw, sg, sx) != (1.0, 1.0, 1.0):
        return
    h0t = torch.empty(F_, Nt, device=dev())
    call('tvae_dec_l0_fwd', xr, Wc, bc, LB, h0t, Nt, F_, Nt, Np, 2, SLOPE)
    w3h = torch.empty(query('tvae_dense_x6_bytes', M, F_) // 4, device=dev())
    call('tvae_dense_split2h', W.to(dev()), F_, w3h, w3h.numel() * 4, M, F_, 0, None, None)
    Yt = torch.empty(M, Nt, device=dev())
    call('tvae_linear_fwd_x6', w3h, None, b.to(dev()), None, Yt, M, Nt, F_, Nt, Nt, 2, SLOPE, None, None, None, xr, Wc, bc, LB, Np,
         None, 2)
    assert rel_err(Yt, act_ref(W.double() @ h0t.double().cpu() + b.double()[:, None], 2)) < GEMM_TOL['f32']


@pytest.mark.parametrize('slack', [1.0, 256.0])
def test_linear_x6_h3_memory_operand_bounds(slack):
    """h3 (parts = 2) with the streamed operand READ FROM MEMORY under a caller-supplied bound (ABI 5: x_amax / a_amax): forward,
    plain data gradient and plain weight gradient against fp64 at the fp32 tolerance, with the exact maximum and with a bound
    2^8 too large (a loose bound costs head room, not correctness: include/tvae_hip.h)."""
    from tvae._lib import query
    M, N, K = 512, 1024, 384
    W, X, b = rnd(M, K, seed=1, scale=K ** -0.5), rnd(K, N, seed=2), rnd(M, seed=3)
    w3 = torch.empty(query('tvae_dense_x6_bytes', M, K) // 4, device=dev())
    call('tvae_dense_split2h', W.to(dev()), K, w3, w3.numel() * 4, M, K, 0, None, None)
    xmax = (X.abs().max() * slack).reshape(1).to(dev())
    Y = torch.full((M, N), float('nan'), device=dev())
    call('tvae_linear_fwd_x6', w3, X.to(dev()), b.to(dev()), None, Y, M, N, K, N, N, 1, SLOPE, None, None, None, None, None, None,
         None, 0, None, 2, xmax)
    assert rel_err(Y, act_ref(W.double() @ X.double() + b.double()[:, None], 1)) < GEMM_TOL['f32']
    with pytest.raises(Exception):                       # h3 with an operand from memory needs its bound
        call('tvae_linear_fwd_x6', w3, X.to(dev()), b.to(dev()), None, Y, M, N, K, N, N, 1, SLOPE, None, None, None, None, None,
             None, None, 0, None, 2, None)
    # plain data gradient dX = W^T d
    d = rnd(M, N, seed=4)
    w3t = torch.empty(query('tvae_dense_x6_bytes', K, M) // 4, device=dev())
    call('tvae_dense_split2h', W.to(dev()), K, w3t, w3t.numel() * 4, K, M, 1, None, None)
    dX = torch.full((K, N), float('nan'), device=dev())
    dmax = (d.abs().max() * slack).reshape(1).to(dev())
    call('tvae_linear_dgrad_x6', w3t, d.to(dev()), None, None, dX, M, N, K, N, N, 0, SLOPE, None, None, None, None, 0, None, None,
         None, None, None, 0, None, 0, None, None, None, None, 2, None, None, None, dmax)
    assert rel_err(dX, W.double().t() @ d.double()) < GEMM_TOL['f32']
    # plain weight gradient dW = d X^T
    ws = torch.empty(query('tvae_linear_wgrad_x6_ws_floats', M, N, K), device=dev())
    dW = torch.full((M, K), float('nan'), device=dev())
    call('tvae_linear_wgrad_x6', d.to(dev()), X.to(dev()), dW, ws, ws.numel(), M, N, K, N, N, 0, None, None, 0, SLOPE, None, None,
         None, None, 0, None, 2, None, 0, None, dmax, xmax)
    assert rel_err(dW, d.double() @ X.double().t()) < GEMM_TOL['f32']


def row_rel_err(got, ref, dim=0):
    """Largest PER-ROW relative error: max over the slices along `dim` of |got - ref|_2 / |ref|_2 (slices that are
    exactly zero in the reference must be exactly zero)."""
    g = got.detach().double().cpu().movedim(dim, 0).reshape(got.shape[dim], -1)
    r = ref.detach().double().cpu().movedim(dim, 0).reshape(ref.shape[dim], -1)
    num, den = (g - r).norm(dim=1), r.norm(dim=1)
    assert bool((num[den == 0] == 0).all())
    return float((num[den > 0] / den[den > 0]).max())


ROW_TOL = 1e-5          # per-ROW relative error against fp64, rows scaled by 2^-16 .. 2^-32 inside a tensor


@pytest.mark.parametrize('e', [16, 24, 32])
def test_h3_row_dynamic_range_decoder(e):
    """h3 (parts = 2) with ONE ROW of an operand 2^-e below the rest of its tensor: the output row / column that row feeds
    must be as accurate RELATIVE TO ITSELF as every other one (round 3 scaled per tensor: 4.8e-5 at e = 24, 1.3e-2 at e = 32).
    Decoder launches: forward (a row of W), two-valued data gradient (a column of W = a row of W^T), weight gradient from sign
    bits against the recomputed operand (a hidden unit of the coordinate layer = a row of the X operand)."""
    from tvae._lib import query
    s = 2.0 ** -e
    F_, B, Np, M, act = 512, 2, 256, 512, 1
    Nt = B * Np
    r0, k0, u0 = 37, 301, 100
    xr = rnd(Nt, 2, seed=1).to(dev())
    Wc, bc, LB = rnd(F_, 2, seed=2), rnd(F_, seed=3), rnd(B, F_, seed=4)
    Wc[u0] *= s; bc[u0] *= s; LB[:, u0] *= s             # hidden unit u0 of the coordinate layer: h0[u0][:] is 2^-e small
    Wc, bc, LB = Wc.to(dev()), bc.to(dev()), LB.to(dev())
    h0 = torch.empty(F_, Nt, device=dev())
    call('tvae_dec_l0_fwd', xr, Wc, bc, LB, h0, Nt, F_, Nt, Np, act, SLOPE)
    H0 = h0.double().cpu()
    W, b = rnd(M, F_, seed=5, scale=F_ ** -0.5), rnd(M, seed=6)
    W[r0] *= s; b[r0] *= s                                # output row r0 of the layer
    W[:, k0] *= s                                         # input feature k0: row k0 of W^T (the data gradient's operand)
    wo, gy = rnd(M, seed=8), rnd(Nt, seed=9)
    va = (xr, Wc, bc, LB, Np)
    w3 = torch.empty(query('tvae_dense_x6_bytes', M, F_) // 4, device=dev())
    call('tvae_dense_split2h', W.to(dev()), F_, w3, w3.numel() * 4, M, F_, 0, None, None)
    Y = torch.empty(M, Nt, device=dev())
    bits = torch.empty(M, Nt // 32, dtype=torch.int32, device=dev())
    call('tvae_linear_fwd_x6', w3, None, b.to(dev()), None, Y, M, Nt, F_, Nt, Nt, act, SLOPE, None, None, None, *va, bits, 2)
    ref_Y = act_ref(W.double() @ H0 + b.double()[:, None], act)
    assert row_rel_err(Y, ref_Y) < ROW_TOL
    assert float(ref_Y[r0].abs().max()) < 64 * s          # (the scaled row really is that small)
    Yd = Y.double().cpu()
    w3t = torch.empty(query('tvae_dense_x6_bytes', F_, M) // 4, device=dev())
    csum = torch.empty(F_, device=dev())
    call('tvae_dense_split2h', W.to(dev()), F_, w3t, w3t.numel() * 4, F_, M, 1, wo.to(dev()), csum)
    dX = torch.empty(F_, Nt, device=dev())
    rs_part = torch.empty((Nt // 128) * M * 2, device=dev())
    gys = gy.sum().reshape(1).to(dev())
    rs_db, rs_dwo = torch.empty(M, device=dev()), torch.empty(M, device=dev())
    call('tvae_linear_dgrad_x6', w3t, Y, None, h0, dX, M, Nt, F_, Nt, Nt, 1, SLOPE, None, None, None, None, 0, None,
         gy.to(dev()), csum, None, None, 0, rs_part, rs_part.numel(), wo.to(dev()), gys, rs_db, rs_dwo, 2)
    d = wo.double()[:, None] * gy.double()[None, :] * dact_ref(Yd, 1)
    ref_dX = (W.double().t() @ d) * dact_ref(H0, 1)
    assert row_rel_err(dX, ref_dX) < ROW_TOL
    ws = torch.empty(query('tvae_linear_wgrad_x6_ws_floats', M, Nt, F_), device=dev())
    dW = torch.empty(M, F_, device=dev())
    call('tvae_linear_wgrad_x6', None, None, dW, ws, ws.numel(), M, Nt, F_, Nt, Nt, 0, wo.to(dev()), gy.to(dev()), 1, SLOPE,
         *va, bits, 2)
    ref_dW = d @ H0.t()
    assert row_rel_err(dW, ref_dW, dim=1) < ROW_TOL      # per COLUMN of dW = per row of the X operand
    assert float(ref_dW[:, u0].abs().max()) < 1e4 * s
    # ADVICE r04 / ABI 6: the same weight gradient with its X operand read from MEMORY (Fourier decoders, stored first layers):
    # one bound per ROW of X -- (|w0| + |w1|) max |x'| + max_b |bc + lb|, times max |gy| -- keeps the small unit's column
    # exact to itself; with ONE bound for the tensor it loses the bits the unit lies below the others (measured here)
    rows = ((Wc.abs().sum(1) * xr.abs().max() + (bc[None, :] + LB).abs().amax(0)) * gy.abs().max()).contiguous()
    dWm = torch.empty(M, F_, device=dev())
    call('tvae_linear_wgrad_x6', None, h0, dWm, ws, ws.numel(), M, Nt, F_, Nt, Nt, 0, wo.to(dev()), gy.to(dev()), 1, SLOPE,
         None, None, None, None, 0, bits, 2, None, 0, None, None, rows, 1)
    assert row_rel_err(dWm, ref_dW, dim=1) < ROW_TOL
    if e >= 24:
        dW1 = torch.empty(M, F_, device=dev())
        call('tvae_linear_wgrad_x6', None, h0, dW1, ws, ws.numel(), M, Nt, F_, Nt, Nt, 0, wo.to(dev()), gy.to(dev()), 1, SLOPE,
             None, None, None, None, 0, bits, 2, None, 0, None, None, rows.amax().reshape(1), 0)
        err1 = float(((dW1.double().cpu() - ref_dW)[:, u0]).abs().max() / ref_dW[:, u0].abs().max())
        assert err1 > 10 * ROW_TOL, err1                  # (the per-tensor bound really is the weaker form: the reason for the rows)


@pytest.mark.parametrize('e', [16, 24, 32])
@pytest.mark.parametrize('B,n,k,pad,C,R', [(24, 64, 64, 16, 16, 8), (6, 28, 28, 8, 16, 8), (2, 96, 32, 16, 4, 4)])
def test_h3_row_dynamic_range_conv1(B, n, k, pad, C, R, e):
    """The frequency-domain lifting convolution in h3 with one FILTER (all its spectrum rows) and one IMAGE 2^-e below the
    rest: the outputs of that filter / that image, and that filter's weight gradient, to 1e-5 relative to themselves (ring,
    register-staged and generic transforms along w)."""
    from tvae._lib import query
    s = 2.0 ** -e
    Cin = 1
    c0, b0 = 3, 1
    y = torch.rand(B, Cin, n, n, generator=torch.Generator().manual_seed(1))
    y[b0] *= s
    bank = rnd(C * R, Cin * k * k, seed=2, scale=(Cin * k * k) ** -0.5)
    bank.view(C, R, -1)[c0] *= s
    Ho = n + 2 * pad - k + 1
    ref = F.conv2d(y.double(), bank.double().view(C * R, Cin, k, k), None, 1, pad).view(B, C, R, Ho, Ho)
    at = torch.zeros(query('tvae_conv1_dft_at_floats', B, Cin, n, k, pad, C, R), device=dev())
    ws = torch.empty(query('tvae_conv1_dft_ws_floats', B, Cin, n, k, pad, C, R), device=dev())
    out = torch.empty(C, B * R * Ho * Ho, device=dev())
    call('tvae_conv1_fwd_dft', y.to(dev()), bank.to(dev()), None, out, at, ws, ws.numel(), B, Cin, n, k, pad, C, R, 0, SLOPE, 2)
    got = out.view(C, B, R, Ho, Ho).permute(1, 0, 2, 3, 4)
    # per (image, filter) block: the dim image under the dim filter is 2^-2e below the rest
    g2 = got.reshape(B * C, -1)
    r2 = ref.reshape(B * C, -1)
    assert row_rel_err(g2, r2) < ROW_TOL
    # the per-channel maxima the output transform leaves for the encoder tail (last C floats behind A^T)
    assert torch.allclose(at[-C:].cpu(), got.abs().amax(dim=(0, 2, 3, 4)).cpu(), rtol=0, atol=0)
    g = rnd(B, C, R, Ho, Ho, seed=4)
    g[:, c0] *= s                                         # the gradient that reaches a dead filter
    ref_g = torch.nn.grad.conv2d_weight(y.double(), (C * R, Cin, k, k), g.double().view(B, C * R, Ho, Ho), padding=pad)
    dpre = g.permute(1, 0, 2, 3, 4).contiguous().view(C, -1).to(dev())
    dbank = torch.empty(C * R, Cin * k * k, device=dev())
    call('tvae_conv1_wgrad_dft', dpre, at, dbank, None, ws, ws.numel(), B, Cin, n, k, pad, C, R, 2)
    assert row_rel_err(dbank.view(C * R, -1), ref_g.reshape(C * R, -1)) < ROW_TOL


@pytest.mark.parametrize('e', [16, 24, 32])
def test_h3_row_dynamic_range_enc_tail(e):
    """The fused encoder tail in h3 with rows of its operands 2^-e below the rest: a row of W2 (forward), a column of W2
    (data gradient), a channel of A1 and a column of Wh (= a row of dH) in the weight gradient."""
    from tvae._lib import query
    from tvae.ops import _enc_tail_perm
    s = 2.0 ** -e
    C, nh, N = 128, 7, 4096
    r0, k0, a0, h0_ = 5, 77, 40, 19
    W2, b2 = rnd(C, C, seed=1, scale=C ** -0.5), rnd(C, seed=2)
    Wh, bh = rnd(nh, C, seed=3, scale=C ** -0.5), rnd(nh, seed=4)
    A1 = rnd(C, N, seed=5)
    W2[r0] *= s; b2[r0] *= s
    W2[:, k0] *= s
    A1[a0] *= s
    Wh[:, h0_] *= s
    a1max = A1.abs().amax(dim=1).contiguous().to(dev())
    w32 = torch.empty(query('tvae_dense_x6_bytes', C, C) // 4, device=dev())
    call('tvae_dense_split2h', W2.to(dev()), C, w32, w32.numel() * 4, C, C, 0, None, None)
    H = torch.empty(C, N, device=dev())
    heads = torch.empty(nh, N, device=dev())
    bits_h = torch.zeros(N, 4, dtype=torch.int32, device=dev())
    bits_a = torch.zeros(N, 4, dtype=torch.int32, device=dev())
    call('tvae_enc_tail_fwd_x6', w32, A1.to(dev()), N, b2.to(dev()), Wh.to(dev()), bh.to(dev()), nh, H, N, heads, N, bits_h, bits_a,
         C, N, 1, SLOPE, 2, a1max)
    Hr = act_ref(W2.double() @ A1.double() + b2.double()[:, None], 1)
    assert row_rel_err(H, Hr) < ROW_TOL
    Hs = H.double().cpu()
    dheads = rnd(nh, N, seed=6)
    w3p = torch.empty(query('tvae_dense_x6_bytes', C, C) // 4, device=dev())
    call('tvae_dense_split2h', W2.t()[:, _enc_tail_perm(dev()).cpu()].contiguous().to(dev()), C, w3p, w3p.numel() * 4, C, C, 0,
         None, None)
    wh3 = torch.empty(query('tvae_dense_x6_bytes', C, nh) // 4, device=dev())
    call('tvae_dense_split3', Wh.to(dev()), C, wh3, wh3.numel() * 4, C, nh, 1, None, None)
    dA1 = torch.empty(C, N, device=dev())
    call('tvae_enc_tail_dgrad_x6', w3p, wh3, dheads.to(dev()), N, nh, bits_h, bits_a, dA1, N, C, N, SLOPE, 2)
    dH = (Wh.double().t() @ dheads.double()) * dact_ref(Hs, 1)
    ref = (W2.double().t() @ dH) * dact_ref(A1.double(), 1)
    assert row_rel_err(dA1, ref) < ROW_TOL
    wsl = torch.empty(query('tvae_enc_tail_wgrad_x6_ws_floats', N), device=dev())
    dW2 = torch.empty(C, C, device=dev())
    call('tvae_enc_tail_wgrad_x6', A1.to(dev()), N, dheads.to(dev()), N, nh, bits_h, Wh.to(dev()), dW2, wsl, wsl.numel(), C, N,
         SLOPE, 2, a1max)
    ref_w = dH @ A1.double().t()
    assert row_rel_err(dW2, ref_w, dim=0) < ROW_TOL      # rows = rows of dH (column h0_ of Wh)
    assert row_rel_err(dW2, ref_w, dim=1) < ROW_TOL      # columns = channels of A1


@pytest.mark.parametrize('F_,B,Np,act,has_lb', [(512, 3, 256, 1, True), (300, 2, 384, 2, True), (256, 2, 128, 1, False)])
def test_linear_x6_recomputed_first_layer_operand(F_, B, Np, act, has_lb):
    """The output of the coordinate layer formed inside its three consumers (forward X, data-gradient mask, weight-gradient
    X) instead of being read: bitwise the same results as with the tensor written by tvae_dec_l0_fwd."""
    from tvae._lib import query
    Nt, M = B * Np, F_
    xr, Wc, bc = rnd(Nt, 2, seed=1).to(dev()), rnd(F_, 2, seed=2).to(dev()), rnd(F_, seed=3).to(dev())
    LB = rnd(B, F_, seed=4).to(dev()) if has_lb else None
    h0 = torch.empty(F_, Nt, device=dev())
    call('tvae_dec_l0_fwd', xr, Wc, bc, LB, h0, Nt, F_, Nt, Np, act, SLOPE)
    pre = (Wc.double().cpu() @ xr.double().cpu().t() + bc.double().cpu()[:, None] +
           (LB.double().cpu().t().repeat_interleave(Np, dim=1) if has_lb else 0.0))
    assert rel_err(h0, act_ref(pre, act)) < TOL
    W, b, d = rnd(M, F_, seed=5, scale=F_ ** -0.5), rnd(M, seed=6), rnd(M, Nt, seed=7).to(dev())
    w3 = torch.empty(query('tvae_dense_x6_bytes', M, F_) // 4, device=dev())
    w3t = torch.empty(query('tvae_dense_x6_bytes', F_, M) // 4, device=dev())
    call('tvae_dense_split3', W.to(dev()), F_, w3, w3.numel() * 4, M, F_, 0, None, None)
    call('tvae_dense_split3', W.to(dev()), F_, w3t, w3t.numel() * 4, F_, M, 1, None, None)
    va = (xr, Wc, bc, LB, Np)
    # forward
    Y = [torch.empty(M, Nt, device=dev()) for _ in range(2)]
    call('tvae_linear_fwd_x6', w3, h0, b.to(dev()), None, Y[0], M, Nt, F_, Nt, Nt, act, SLOPE, None, None, None,
         None, None, None, None, 0, None, 3)
    call('tvae_linear_fwd_x6', w3, None, b.to(dev()), None, Y[1], M, Nt, F_, Nt, Nt, act, SLOPE, None, None, None, *va, None, 3)
    assert torch.equal(Y[0], Y[1])
    assert rel_err(Y[0], act_ref(W.double() @ h0.double().cpu() + b.double()[:, None], act)) < GEMM_TOL['f32']
    # data gradient with the fused coordinate-layer backward
    outs = []
    for virt in (False, True):
        gxr = torch.empty(Nt, 2, device=dev())
        part = torch.empty((Nt // 128) * F_ * 3, device=dev())
        call('tvae_linear_dgrad_x6', w3t, d, None, None if virt else h0, None, M, Nt, F_, Nt, Nt, act, SLOPE, xr, Wc,
             gxr, part, part.numel(), None, None, None, bc if virt else None, LB if virt else None, Np if virt else 0, None, 0, None, None, None, None, 3)
        outs.append((gxr, part))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    d0 = (W.double().t() @ d.double().cpu()) * dact_ref(h0.double().cpu(), act)
    assert rel_err(outs[1][0], d0.t() @ Wc.double().cpu()) < GEMM_TOL['f32']
    # weight gradient
    ws = torch.empty(1 << 26, device=dev())       # >= tvae_linear_wgrad_x6_ws_floats(M, N, K): slices x M x K
    dW = [torch.empty(M, F_, device=dev()) for _ in range(2)]
    call('tvae_linear_wgrad_x6', d, h0, dW[0], ws, ws.numel(), M, Nt, F_, Nt, Nt, 0, None, None, act, SLOPE,
         None, None, None, None, 0, None, 3)
    call('tvae_linear_wgrad_x6', d, None, dW[1], ws, ws.numel(), M, Nt, F_, Nt, Nt, 0, None, None, act, SLOPE, *va, None, 3)
    assert torch.equal(dW[0], dW[1])
    assert rel_err(dW[1], d.double().cpu() @ h0.double().cpu().t()) < GEMM_TOL['f32']
    # images that are not whole column tiles are refused
    with pytest.raises(Exception):
        call('tvae_linear_fwd_x6', w3, None, b.to(dev()), None, Y[1], M, Nt, F_, Nt, Nt, act, SLOPE, None, None, None,
             xr, Wc, bc, LB, 96, None, 3)


def test_reductions():
    M, N = 37, 10007
    X = rnd(M, N, seed=1)
    V = rnd(N, 3, seed=2)
    sl = 1000
    nseg = (N + sl - 1) // sl
    out = torch.empty(nseg, M, 3, device=dev())
    call('tvae_rowdot_seg', X.to(dev()), N, V.to(dev()), 3, M, N, sl, out)
    ref = torch.stack([X[:, s * sl:(s + 1) * sl].double() @ V[s * sl:(s + 1) * sl].double() for s in range(nseg)])
    assert rel_err(out, ref) < TOL
    out1 = torch.empty(nseg, M, device=dev())
    call('tvae_rowdot_seg', X.to(dev()), N, None, 1, M, N, sl, out1)
    ref1 = torch.stack([X[:, s * sl:(s + 1) * sl].double().sum(1) for s in range(nseg)])
    assert rel_err(out1, ref1) < TOL
    tot = torch.full((M,), 2.0, device=dev())
    call('tvae_seg_sum', out1, nseg, M, tot, 0.5, 1)
    assert rel_err(tot, 2.0 + 0.5 * X.double().sum(1)) < TOL


@pytest.mark.parametrize('N,rows_d,ldpad', [(32 * 700, 103, 0), (32 * 3, 128, 0), (32 * 1031, 8, 64), (32, 1, 0)])
def test_enc_tail_wgrad_wide(N, rows_d, ldpad):
    """Round 6: dW[r][c] = sum_n D[r][n] A[c][n] for two stored operands (the encoder tail's weight gradients with many head rows:
    autograd of the 1x1x1 convolutions, reference src/models.py:347-358,390-392) against float64 -- fewer chunks than
    workgroups, one chunk, padded leading dimensions, rows of both operands 2^-20 below the rest (per-row relative error of dW's
    rows AND columns), loose bounds (8x the true maxima), repeatable bit for bit."""
    from tvae._lib import query
    C = 128
    s = 2.0 ** -20
    D, A = rnd(rows_d, N, seed=1), rnd(C, N, seed=2)
    A[7] *= s
    if rows_d > 2:
        D[1] *= 2.0 ** -10
    ld = N + ldpad
    Dp = torch.full((rows_d, ld), float('nan')); Dp[:, :N] = D
    Ap = torch.full((C, ld), float('nan')); Ap[:, :N] = A
    if ldpad:
        Dp[:, N:] = 1e30; Ap[:, N:] = 1e30               # (never read)
    amax_d = (8.0 * D.abs().max()).reshape(1).to(dev())
    amax_a = (8.0 * A.abs().amax(dim=1)).contiguous().to(dev())
    ws = torch.empty(query('tvae_enc_tail_wgrad_x6_ws_floats', N), device=dev())
    out = [torch.full((C, C), float('nan'), device=dev()) for _ in range(2)]
    for o in out:
        call('tvae_enc_tail_wgrad_wide', Dp.to(dev()), ld, rows_d, Ap.to(dev()), ld, o, ws, ws.numel(), C, N, amax_d, amax_a)
    assert torch.equal(out[0][:rows_d], out[1][:rows_d])
    ref = D.double() @ A.double().t()
    got = out[0][:rows_d].double().cpu()
    assert torch.isfinite(got).all()
    assert row_rel_err(got, ref) < ROW_TOL
    assert row_rel_err(got.t(), ref.t()) < ROW_TOL
    # argument checks: ragged N, too many rows, missing bounds
    for bad in ((N + 1, rows_d, amax_d), (N, 129, amax_d), (N, rows_d, None)):
        with pytest.raises(Exception):
            call('tvae_enc_tail_wgrad_wide', Dp.to(dev()), ld, bad[1], Ap.to(dev()), ld, out[0], ws, ws.numel(), C, bad[0], bad[2], amax_a)


def test_enc_tail_wgrad_wide_galaxy_size():
    """The same entry point at the galaxy configuration's column count (8 images x 16 rotations x 129^2 positions, 103 head
    rows; BASELINE.json configs[4]) against a float64 product formed on the GPU, bounds as the product path forms them
    (max |D| measured; per-row analytic bounds of A loose by the row-sum factor)."""
    from tvae._lib import query
    C, rows_d, N = 128, 103, 8 * 16 * 129 * 129
    g = torch.Generator(device=dev()).manual_seed(3)
    D = torch.randn(rows_d, N, device=dev(), generator=g) * torch.logspace(-6, 0, rows_d, device=dev())[:, None]
    A = torch.nn.functional.leaky_relu(torch.randn(C, N, device=dev(), generator=g), SLOPE)
    amax_d = D.abs().max().reshape(1)
    amax_a = (11.0 * A.abs().amax(dim=1)).contiguous()
    ws = torch.empty(query('tvae_enc_tail_wgrad_x6_ws_floats', N), device=dev())
    out = torch.full((C, C), float('nan'), device=dev())
    call('tvae_enc_tail_wgrad_wide', D, N, rows_d, A, N, out, ws, ws.numel(), C, N, amax_d, amax_a)
    ref = D.double() @ A.double().t()
    assert row_rel_err(out[:rows_d], ref) < ROW_TOL
    assert rel_err(out[:rows_d], ref) < 1e-5        # (fp32 accumulation over 8 320 columns per workgroup: ~2e-6 measured)


@pytest.mark.parametrize('S,L', [(256, 512), (100, 130), (67, 65), (64, 7), (5, 300), (256, 3)])
def test_seg_sum_instances(S, L):
    """tvae_seg_sum picks one of three kernels by shape (one thread per output / one wave per output / 64 outputs x 16 segment
    lanes): all agree with the fp64 sum, with scaling and accumulation, and repeat bit for bit."""
    X = rnd(S, L, seed=S + L).to(dev())
    out = torch.full((L,), 3.0, device=dev())
    call('tvae_seg_sum', X, S, L, out, 0.25, 1)
    assert rel_err(out, 3.0 + 0.25 * X.double().sum(0)) < TOL
    o1, o2 = torch.empty(L, device=dev()), torch.empty(L, device=dev())
    call('tvae_seg_sum', X, S, L, o1, 1.0, 0)
    call('tvae_seg_sum', X, S, L, o2, 1.0, 0)
    assert torch.equal(o1, o2) and rel_err(o1, X.double().sum(0)) < TOL


@pytest.mark.parametrize('no', [1, 2, 3])
def test_coldot_outer_mask(no):
    M, N = 64, 1500
    X, W, b = rnd(M, N, seed=1), rnd(no, M, seed=2), rnd(no, seed=3)
    out = torch.empty(N, no, device=dev())
    call('tvae_coldot', X.to(dev()), N, M, N, W.to(dev()), 1, M, b.to(dev()), no, out)
    assert rel_err(out, (W.double() @ X.double()).t() + b.double()) < TOL
    dy = rnd(N, no, seed=4)
    D = torch.empty(M, N, device=dev())
    call('tvae_outer_mask', dy.to(dev()), no, W.to(dev()), 1, M, X.to(dev()), N, D, N, M, N, 1, SLOPE)
    ref = (W.double().t() @ dy.double().t()) * dact_ref(X.double(), 1)
    assert rel_err(D, ref) < TOL
    pre = torch.empty(M, N, device=dev())
    call('tvae_act_bwd', D, X.to(dev()), pre, M * N, 1, SLOPE)
    assert rel_err(pre, ref * dact_ref(X.double(), 1)) < TOL


@pytest.mark.parametrize('F_,N,no,act', [(64, 3000, 1, 1), (37, 2051, 2, 1), (130, 1024, 3, 2), (16, 5000, 4, 0)])
def test_dec_out_bwd(F_, N, no, act):
    gy, Wo = rnd(N, no, seed=1), rnd(no, F_, seed=2)
    H = rnd(F_, N, seed=3).clamp(-0.9, 0.9)
    D = torch.empty(F_, N, device=dev())
    npan = (N + 1023) // 1024
    part = torch.empty(npan * F_ * (1 + no), device=dev())
    tot = torch.empty(1 + no, F_, device=dev())
    call('tvae_dec_out_bwd', gy.to(dev()), no, Wo.to(dev()), H.to(dev()), N, D, N, F_, N, act, SLOPE, part,
         part.numel(), tot)
    ref = (Wo.double().t() @ gy.double().t()) * dact_ref(H.double(), act)
    assert rel_err(D, ref) < TOL
    assert rel_err(tot[0], ref.sum(1)) < TOL
    assert rel_err(tot[1:], (H.double() @ gy.double()).t()) < TOL
    small = torch.empty(4, device=dev())
    with pytest.raises(Exception):
        call('tvae_dec_out_bwd', gy.to(dev()), no, Wo.to(dev()), H.to(dev()), N, D, N, F_, N, act, SLOPE, small, 4, tot)


@pytest.mark.parametrize('F_,B,Np', [(64, 3, 784), (33, 2, 1089), (128, 2, 4096), (8, 5, 2500)])
def test_dec_in_bwd(F_, B, Np):
    Nt = B * Np
    d, xr, Wc = rnd(F_, Nt, seed=1), rnd(Nt, 2, seed=2), rnd(F_, 2, seed=3)
    cpi = (Np + 1023) // 1024
    part = torch.empty(B * cpi * F_ * 3, device=dev())
    gxr = torch.empty(Nt, 2, device=dev())
    Simg = torch.empty(B, F_, device=dev())
    dbc = torch.empty(F_, device=dev())
    dWc = torch.empty(F_, 2, device=dev())
    call('tvae_dec_in_bwd', d.to(dev()), Nt, xr.to(dev()), Wc.to(dev()), F_, B, Np, gxr, Simg, dbc, dWc, part,
         part.numel())
    dd = d.double()
    assert rel_err(gxr, dd.t() @ Wc.double()) < TOL
    assert rel_err(Simg, dd.view(F_, B, Np).sum(2).t()) < TOL
    assert rel_err(dbc, dd.sum(1)) < TOL
    assert rel_err(dWc, dd @ xr.double()) < TOL


@pytest.mark.parametrize('nh,C,N,act', [(7, 128, 8712, 1), (3, 33, 1001, 1), (8, 64, 2048, 2), (1, 16, 515, 0),
                                        (5, 128, 4356, 1)])
def test_heads_fwd_bwd(nh, C, N, act):
    W, b = rnd(nh, C, seed=1, scale=C ** -0.5), rnd(nh, seed=2)
    X = rnd(C, N, seed=3).clamp(-0.9, 0.9)
    Y = torch.empty(nh, N, device=dev())
    call('tvae_heads_fwd', W.to(dev()), X.to(dev()), N, b.to(dev()), Y, N, nh, C, N)
    assert rel_err(Y, W.double() @ X.double() + b.double()[:, None]) < TOL
    dY = rnd(nh, N, seed=4)
    dX = torch.empty(C, N, device=dev())
    npan = (N + 511) // 512
    part = torch.empty(npan * C * (nh + 1), device=dev())
    tot = torch.empty(nh + 1, C, device=dev())
    call('tvae_heads_bwd', W.to(dev()), dY.to(dev()), N, X.to(dev()), N, dX, N, nh, C, N, act, SLOPE, part,
         part.numel(), tot)
    ref = (W.double().t() @ dY.double()) * dact_ref(X.double(), act)
    assert rel_err(dX, ref) < TOL
    assert rel_err(tot[:nh], dY.double() @ X.double().t()) < TOL
    assert rel_err(tot[nh], ref.sum(1)) < TOL
    with pytest.raises(Exception):
        call('tvae_heads_fwd', W.to(dev()), X.to(dev()), N, b.to(dev()), Y, N, 9, C, N)


def _split_w(W, rows, K, transpose):
    a3 = torch.empty(query('tvae_dense_x6_bytes', rows, K) // 4, device=dev())
    call('tvae_dense_split3', W.contiguous(), W.shape[1], a3, a3.numel() * 4, rows, K, transpose, None, None)
    return a3


def _enc_tail_perm():
    from tvae.ops import _enc_tail_perm as f
    return f(dev())


@pytest.mark.parametrize('N,nh,act,parts', [(64 * 40, 7, 1, 3), (64 * 2048 + 37, 7, 1, 3), (1000, 5, 2, 3), (31, 1, 0, 3),
                                            (64 * 300, 7, 1, 1), (64 * 1024 + 36, 3, 1, 3), (4, 7, 1, 3), (32 * 9001, 7, 1, 3),
                                            (32, 2, 1, 3)])
def test_enc_tail_x6(N, nh, act, parts):
    """Fused conv2 + head projection (reference models.py:356-358, 390-392) and its fused data gradient against fp64."""
    C = 128
    W2, b2 = rnd(C, C, seed=1, scale=C ** -0.5), rnd(C, seed=2)
    Wh, bh = rnd(nh, C, seed=3, scale=C ** -0.5), rnd(nh, seed=4)
    A1 = rnd(C, N, seed=5)
    H = torch.full((C, N), float('nan'), device=dev())
    heads = torch.full((nh, N), float('nan'), device=dev())
    w3 = _split_w(W2.to(dev()), C, C, 0)
    lrelu = act == 1
    bits_h = torch.zeros(N, 4, dtype=torch.int32, device=dev()) if lrelu else None
    bits_a = torch.zeros(N, 4, dtype=torch.int32, device=dev()) if lrelu else None
    call('tvae_enc_tail_fwd_x6', w3, A1.to(dev()), N, b2.to(dev()), Wh.to(dev()), bh.to(dev()), nh, H, N, heads, N, bits_h,
         bits_a, C, N, act, SLOPE, parts)
    Hr = act_ref(W2.double() @ A1.double() + b2.double()[:, None], act)
    hr = Wh.double() @ Hr + bh.double()[:, None]
    tol = TOL if parts == 3 else 2e-2
    assert rel_err(H, Hr) < tol
    assert rel_err(heads, hr) < tol
    if parts == 3:
        # the same launch in the h3 arithmetic: two fp16 parts under the scale of max |A1| (per-channel device words from A1's producer),
        # for activations of size 1, 1e-5 and 1e3 (a bound 64 x too large must not matter either)
        w32 = torch.empty(query('tvae_dense_x6_bytes', C, C) // 4, device=dev())
        call('tvae_dense_split2h', W2.to(dev()), C, w32, w32.numel() * 4, C, C, 0, None, None)
        # (tanh at 1e3 saturates: every arithmetic then differs where a pre-activation of size 1 is a difference of terms of size 1e3)
        for sc, slack in ((1.0, 1.0), (1e-5, 1.0)) + (((1e3, 64.0),) if act != 2 else ()):
            A1s = (A1 * sc).to(dev())
            amax = (A1s.abs().amax(dim=1) * slack).contiguous()      # one maximum per channel (ABI 5), from A1's producer
            H2, h2 = torch.full((C, N), float('nan'), device=dev()), torch.full((nh, N), float('nan'), device=dev())
            call('tvae_enc_tail_fwd_x6', w32, A1s, N, b2.to(dev()), Wh.to(dev()), bh.to(dev()), nh, H2, N, h2, N, None, None, C, N,
                 act, SLOPE, 2, amax)
            Hr2 = act_ref(W2.double() @ (A1.double() * sc) + b2.double()[:, None], act)
            assert rel_err(H2, Hr2) < TOL and rel_err(h2, Wh.double() @ Hr2 + bh.double()[:, None]) < TOL, sc
        with pytest.raises(Exception):                   # h3 without the operand maximum
            call('tvae_enc_tail_fwd_x6', w32, A1.to(dev()), N, b2.to(dev()), Wh.to(dev()), bh.to(dev()), nh, H, N, heads, N, None,
                 None, C, N, act, SLOPE, 2, None)
    with pytest.raises(Exception):                       # only the 128-channel layer is built
        call('tvae_enc_tail_fwd_x6', w3, A1.to(dev()), N, b2.to(dev()), Wh.to(dev()), bh.to(dev()), nh, H, N, heads, N, None,
             None, 64, N, act, SLOPE, parts)
    if not lrelu:
        with pytest.raises(Exception):                   # the sign words describe a LeakyReLU
            junk = torch.zeros(N, 4, dtype=torch.int32, device=dev())
            call('tvae_enc_tail_fwd_x6', w3, A1.to(dev()), N, b2.to(dev()), Wh.to(dev()), bh.to(dev()), nh, H, N, heads, N,
                 junk, junk, C, N, act, SLOPE, parts)
        return
    # sign words: bit (r & 31) of word r >> 5 of column n
    def unpack(b):
        w = b.cpu().numpy().astype(np.uint32)                              # [N][4]
        return torch.from_numpy(((w[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(N, 128).T.copy()).bool()
    assert torch.equal(unpack(bits_a), A1 > 0)
    assert torch.equal(unpack(bits_h), H.cpu() > 0)
    dheads = rnd(nh, N, seed=6)
    w3p = _split_w(W2.t()[:, _enc_tail_perm().cpu()].contiguous().to(dev()), C, C, 0)
    wh3 = torch.empty(query('tvae_dense_x6_bytes', C, nh) // 4, device=dev())
    call('tvae_dense_split3', Wh.to(dev()), C, wh3, wh3.numel() * 4, C, nh, 1, None, None)
    dA1 = torch.full((C, N), float('nan'), device=dev())
    call('tvae_enc_tail_dgrad_x6', w3p, wh3, dheads.to(dev()), N, nh, bits_h, bits_a, dA1, N, C, N, SLOPE, parts)
    Hs = H.cpu().double()                                # the mask of the kernel's own forward (kinks of fp64 H may differ)
    dH = (Wh.double().t() @ dheads.double()) * dact_ref(Hs, 1)
    ref = (W2.double().t() @ dH) * dact_ref(A1.double(), 1)
    assert rel_err(dA1, ref) < tol
    if parts == 3:
        # the same launch in the h3 arithmetic (parts = 2: the 128 x 128 GEMM on two fp16 parts with the exact scale of each
        # 32-column chunk; the skinny head-row GEMM stays in the three-part split), also for gradients of size 1e-7
        w3p2 = torch.empty(query('tvae_dense_x6_bytes', C, C) // 4, device=dev())
        call('tvae_dense_split2h', W2.t()[:, _enc_tail_perm().cpu()].contiguous().to(dev()), C, w3p2, w3p2.numel() * 4, C, C, 0,
             None, None)
        for sc in (1.0, 1e-7):
            dA2 = torch.full((C, N), float('nan'), device=dev())
            call('tvae_enc_tail_dgrad_x6', w3p2, wh3, (dheads * sc).to(dev()), N, nh, bits_h, bits_a, dA2, N, C, N, SLOPE, 2)
            assert rel_err(dA2, ref * sc) < TOL, sc
    # conv2's weight gradient in one pass (dH formed from the head gradients and the sign words, never stored), and the
    # sums-only form of tvae_heads_bwd (dX = NULL) that supplies dWh / db2 beside it
    if N % 32 == 0:
        wsl = torch.full((query('tvae_enc_tail_wgrad_x6_ws_floats', N),), float('nan'), device=dev())
        dW2 = torch.full((C, C), float('nan'), device=dev())
        call('tvae_enc_tail_wgrad_x6', A1.to(dev()), N, dheads.to(dev()), N, nh, bits_h, Wh.to(dev()), dW2, wsl, wsl.numel(), C, N,
             SLOPE, parts)
        assert rel_err(dW2, dH @ A1.double().t()) < tol
        if parts == 3:                       # the same in the h3 arithmetic, also for head gradients of size 1e-7
            for sc in (1.0, 1e-7):
                dW2h = torch.full((C, C), float('nan'), device=dev())
                call('tvae_enc_tail_wgrad_x6', A1.to(dev()), N, (dheads * sc).to(dev()), N, nh, bits_h, Wh.to(dev()), dW2h, wsl,
                     wsl.numel(), C, N, SLOPE, 2, A1.abs().amax(dim=1).contiguous().to(dev()))
                assert rel_err(dW2h, (dH @ A1.double().t()) * sc) < TOL, sc
        npan = (N + 511) // 512
        part = torch.empty(npan * C * (nh + 1), device=dev())
        tot = torch.empty(nh + 1, C, device=dev())
        call('tvae_heads_bwd', Wh.to(dev()), dheads.to(dev()), N, H, N, None, N, nh, C, N, 1, SLOPE, part, part.numel(), tot)
        assert rel_err(tot[:nh], dheads.double() @ Hs.t()) < TOL and rel_err(tot[nh], dH.sum(1)) < 1e-4
    else:
        with pytest.raises(Exception):                   # whole 32-column chunks only (the unfused path takes the rest)
            call('tvae_enc_tail_wgrad_x6', A1.to(dev()), N, dheads.to(dev()), N, nh, bits_h, Wh.to(dev()), dA1, dA1, dA1.numel(),
                 C, N, SLOPE, parts)


def test_enc_tail_x6_full_size_properties():
    """The fused encoder tail at the benchmark's full size (N = 256 * 8 * 33 * 33 columns): two runs are bitwise
    identical (persistent workgroups, no atomics), a random sample of columns matches fp64, and every sign word equals
    the sign of what was stored."""
    C, nh, N = 128, 7, 256 * 8 * 33 * 33
    g = torch.Generator(device=dev()).manual_seed(11)
    W2 = torch.randn(C, C, device=dev(), generator=g) * C ** -0.5
    b2 = torch.randn(C, device=dev(), generator=g)
    Wh = torch.randn(nh, C, device=dev(), generator=g) * C ** -0.5
    bh = torch.randn(nh, device=dev(), generator=g)
    A1 = torch.randn(C, N, device=dev(), generator=g)
    dheads = torch.randn(nh, N, device=dev(), generator=g)
    w3 = _split_w(W2, C, C, 0)
    w3p = _split_w(W2.t()[:, _enc_tail_perm()].contiguous(), C, C, 0)
    wh3 = torch.empty(query('tvae_dense_x6_bytes', C, nh) // 4, device=dev())
    call('tvae_dense_split3', Wh, C, wh3, wh3.numel() * 4, C, nh, 1, None, None)
    outs = []
    for _ in range(2):
        H = torch.empty(C, N, device=dev())
        heads = torch.empty(nh, N, device=dev())
        bits = torch.zeros(2, N, 4, dtype=torch.int32, device=dev())
        dA1 = torch.empty(C, N, device=dev())
        call('tvae_enc_tail_fwd_x6', w3, A1, N, b2, Wh, bh, nh, H, N, heads, N, bits[0], bits[1], C, N, 1, SLOPE, 3)
        call('tvae_enc_tail_dgrad_x6', w3p, wh3, dheads, N, nh, bits[0], bits[1], dA1, N, C, N, SLOPE, 3)
        outs.append((H, heads, bits, dA1))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    H, heads, bits, dA1 = outs[0]
    cols = torch.randint(0, N, (4096,), device=dev(), generator=g)
    cols[:64] = torch.arange(N - 64, N, device=dev())                      # the last chunks too
    Hr = act_ref(W2.double() @ A1[:, cols].double() + b2.double()[:, None], 1)
    assert rel_err(H[:, cols], Hr) < TOL
    assert rel_err(heads[:, cols], Wh.double() @ Hr + bh.double()[:, None]) < TOL
    dH = (Wh.double().t() @ dheads[:, cols].double()) * dact_ref(H[:, cols].double(), 1)
    assert rel_err(dA1[:, cols], (W2.double().t() @ dH) * dact_ref(A1[:, cols].double(), 1)) < TOL
    shifts = torch.arange(32, device=dev(), dtype=torch.int32)
    for t, w in ((H, bits[0]), (A1, bits[1])):
        got = ((w[cols][:, :, None] >> shifts) & 1).reshape(-1, 128).t().bool()
        assert torch.equal(got, t[:, cols] > 0)
    # the weight gradients of the same step at full size (every CU busy): dH, dWh, db2 from tvae_heads_bwd, dW2 from the
    # fp32-MFMA reduction over all 2.2 M columns, against fp64 sums
    dH = torch.empty(C, N, device=dev())
    npan = (N + 511) // 512
    part = torch.empty(npan * C * (nh + 1), device=dev())
    tot = torch.empty(nh + 1, C, device=dev())
    call('tvae_heads_bwd', Wh, dheads, N, H, N, dH, N, nh, C, N, 1, SLOPE, part, part.numel(), tot)
    dW2 = torch.empty(C, C, device=dev())
    ws = torch.empty(1 << 24, device=dev())
    call('tvae_linear_wgrad', dH, A1, dW2, ws, ws.numel(), C, N, C, N, N, 0)
    dHr = torch.zeros(C, C, dtype=torch.float64, device=dev())
    dWhr = torch.zeros(nh, C, dtype=torch.float64, device=dev())
    dbr = torch.zeros(C, dtype=torch.float64, device=dev())
    step = 1 << 18                                       # fp64 references in column slices (memory)
    for c0 in range(0, N, step):
        sl = slice(c0, min(N, c0 + step))
        Hd = H[:, sl].double()
        dHd = (Wh.double().t() @ dheads[:, sl].double()) * dact_ref(Hd, 1)
        dHr += dHd @ A1[:, sl].double().t()
        dWhr += dheads[:, sl].double() @ Hd.t()
        dbr += dHd.sum(1)
    assert rel_err(tot[:nh], dWhr) < TOL
    assert rel_err(tot[nh], dbr) < 1e-4
    assert rel_err(dW2, dHr) < TOL
    # the one-pass form (dH never stored): same sums, bitwise repeatable
    wsl = torch.empty(query('tvae_enc_tail_wgrad_x6_ws_floats', N), device=dev())
    dW2f, dW2g = torch.empty(C, C, device=dev()), torch.empty(C, C, device=dev())
    call('tvae_enc_tail_wgrad_x6', A1, N, dheads, N, nh, bits[0], Wh, dW2f, wsl, wsl.numel(), C, N, SLOPE, 3)
    call('tvae_enc_tail_wgrad_x6', A1, N, dheads, N, nh, bits[0], Wh, dW2g, wsl, wsl.numel(), C, N, SLOPE, 3)
    assert torch.equal(dW2f, dW2g) and rel_err(dW2f, dHr) < TOL
    tot2 = torch.empty(nh + 1, C, device=dev())
    call('tvae_heads_bwd', Wh, dheads, N, H, N, None, N, nh, C, N, 1, SLOPE, part, part.numel(), tot2)
    assert torch.equal(tot2, tot)


@pytest.mark.parametrize('C,B,R,Ho,act', [(8, 3, 4, 9, 1), (128, 5, 8, 29, 1), (16, 2, 16, 7, 2), (5, 1, 8, 3, 0)])
def test_rot_pool(C, B, R, Ho, act):
    """fc_r pooling over the rotation axis (reference models.py:303-305) and its backward through the activation."""
    P = Ho * Ho
    A1 = rnd(C, B, R, P, seed=1).clamp(-0.9, 0.9)
    fw, fb = rnd(R, seed=2), rnd(1, seed=3)
    X = torch.empty(C, B * P, device=dev())
    call('tvae_rot_pool_fwd', A1.to(dev()), fw.to(dev()), fb.to(dev()), X, C, B, R, P)
    ref = torch.einsum('cbrp,r->cbp', A1.double(), fw.double()) + fb.double()
    assert rel_err(X.view(C, B, P), ref) < TOL
    dX = rnd(C, B, P, seed=4)
    dA1 = torch.empty(C, B, R, P, device=dev())
    nb = min(1024, (C * B * P + 255) // 256)
    part = torch.empty(nb * (R + 1), device=dev())
    dtot = torch.empty(R + 1, device=dev())
    call('tvae_rot_pool_bwd', A1.to(dev()), dX.to(dev()), fw.to(dev()), dA1, part, part.numel(), dtot, C, B, R, P, act,
         SLOPE)
    dref = dX.double()[:, :, None, :] * fw.double()[None, None, :, None] * dact_ref(A1.double(), act)
    assert rel_err(dA1, dref) < TOL
    assert rel_err(dtot[:R], torch.einsum('cbrp,cbp->r', A1.double(), dX.double())) < TOL
    assert rel_err(dtot[R], dX.double().sum()) < 1e-4
    with pytest.raises(Exception):                       # short partial-sum workspace
        call('tvae_rot_pool_bwd', A1.to(dev()), dX.to(dev()), fw.to(dev()), dA1, part, R, dtot, C, B, R, P, act, SLOPE)


def test_coord():
    B, n = 3, 9
    xc = O.image_coords(n)
    dx = rnd(B, 2, seed=1, scale=0.2).requires_grad_(True)
    th = rnd(B, seed=2).requires_grad_(True)
    x = xc.expand(B, -1, -1) - dx.unsqueeze(1)
    rot = torch.stack([torch.stack([torch.cos(th), torch.sin(th)], 1),
                       torch.stack([-torch.sin(th), torch.cos(th)], 1)], 1)
    ref = torch.bmm(x, rot)
    g = rnd(B, n * n, 2, seed=3)
    (ref * g).sum().backward()
    from tvae import ops
    dxg = dx.detach().to(dev()).requires_grad_(True)
    thg = th.detach().to(dev()).requires_grad_(True)
    xr = ops.CoordFn.apply(xc.to(dev()), dxg, thg)
    assert rel_err(xr, ref) < 1e-6
    (xr * g.to(dev())).sum().backward()
    assert rel_err(dxg.grad, dx.grad) < TOL and rel_err(thg.grad, th.grad) < TOL


@pytest.mark.parametrize('kind,name', [(0, 'bce'), (1, 'gauss'), (2, 'gauss_var')])
def test_loglik(kind, name):
    B, L = 3, 500
    y = torch.rand(B, L, generator=torch.Generator().manual_seed(1))
    yh = rnd(B, 2 * L if kind == 2 else L, seed=2, scale=2.0).requires_grad_(True)
    w = rnd(B, seed=3)
    if kind == 0:
        per = -(F.binary_cross_entropy_with_logits(yh, y, reduction='none')).sum(1)
    elif kind == 1:
        per = -0.5 * ((yh - y) ** 2).sum(1)
    else:
        per = -0.5 * ((yh[:, :L] - y) ** 2 / torch.exp(yh[:, L:]) + yh[:, L:]).sum(1)
    (per * w).sum().backward()
    assert abs(float(per.mean()) - float(O.likelihood_logp(yh.detach(), y, name))) < 1e-3 * abs(float(per.mean()))
    from tvae import ops
    yg = yh.detach().to(dev()).requires_grad_(True)
    lp = ops.LogLikFn.apply(yg, y.to(dev()), kind)
    assert rel_err(lp, per) < TOL
    (lp * w.to(dev())).sum().backward()
    assert rel_err(yg.grad, yh.grad) < TOL


def test_adam_flat():
    n = 10001
    p0, steps = rnd(n, seed=1), 3
    p_ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([p_ref], lr=2e-4)
    p = p0.clone().to(dev())
    m = torch.zeros(n, device=dev())
    v = torch.zeros(n, device=dev())
    from tvae import ops
    for s in range(steps):
        g = rnd(n, seed=10 + s)
        p_ref.grad = g.clone()
        opt.step()
        ops.adam_flat(p, g.to(dev()), m, v, s + 1, 2e-4)
    assert rel_err(p, p_ref.detach()) < 1e-6


def _head_reference(hd, E, eps_z, eps_t, R, Ho, zd, refine, theta_prior, normal, spacing):
    """Torch restatement of the head through the oracle (encoder tail + posterior_pool_kl)."""
    B = E.shape[0]
    hv = hd.view(3 + 2 * zd, B, R, Ho, Ho)
    p_r = torch.from_numpy(O.rotation_log_prior(R, refine, theta_prior, normal)).view(R, 1, 1)
    attn = hv[0] + p_r
    q = F.log_softmax(attn.reshape(B, -1), 1).view(B, R, Ho, Ho)
    a = F.softmax(attn.reshape(B, -1) - torch.log(E), 1).view(B, R, Ho, Ho)
    theta = hv[1:3].permute(1, 0, 2, 3, 4)
    offs = torch.from_numpy(O.rotation_offsets(R)) if refine else torch.zeros(R)
    if refine:
        theta = torch.stack((theta[:, 0] + offs.view(1, R, 1, 1), theta[:, 1]), 1)
    zv = hv[3:].permute(1, 0, 2, 3, 4)
    x = torch.zeros(B, 4, 2)
    z, th, dx, _, kl = O.posterior_pool_kl(x, attn, q, p_r, a, offs, theta, zv, spacing, eps_z, eps_t, R, theta_prior)
    return attn, q, a, z, th, dx.view(B, 2), kl


@pytest.mark.parametrize('R,Ho,zd,refine,normal,scale', [(8, 17, 2, True, False, 1.0), (4, 9, 3, False, False, 4.0),
                                                         (16, 5, 2, True, True, 8.0), (8, 33, 2, True, False, 30.0),
                                                         # few images x many positions: the chunked (several workgroups
                                                         # per image) kernels, R*P = 17 424 / 34 848
                                                         (16, 33, 5, True, False, 6.0), (8, 66, 2, True, False, 20.0)])
def test_attn_head(R, Ho, zd, refine, normal, scale):
    from tvae import ops
    B = 3
    RP = R * Ho * Ho
    theta_prior = math.pi / 4 if normal else math.pi
    spacing = 2.0 / 27
    hd = rnd(3 + 2 * zd, B * RP, seed=1, scale=0.5)
    hd[0] *= scale                                       # peaked attention for large scale (forces exp(q)==0)
    g = torch.Generator().manual_seed(2)
    E = torch.empty(B, RP).exponential_(generator=g)
    eps_z, eps_t = rnd(B, zd, seed=3), rnd(B, seed=4)
    hr = hd.clone().requires_grad_(True)
    ref = _head_reference(hr, E, eps_z, eps_t, R, Ho, zd, refine, theta_prior, normal, spacing)
    tb = ops.HeadTables(R, Ho, spacing, refine, theta_prior, normal, dev())
    hg = hd.clone().to(dev()).requires_grad_(True)
    got = ops.HeadFn.apply(hg, E.to(dev()), eps_z.to(dev()), eps_t.to(dev()), tb, B, zd)
    names = ['attn', 'q', 'a', 'z', 'theta', 'dx', 'kl']
    for nm, r_, g_ in zip(names, ref, got):
        assert rel_err(g_.reshape(-1), r_.reshape(-1)) < 5e-5, nm
    ws = [rnd(*r_.shape, seed=20 + i) for i, r_ in enumerate(ref)]
    coef = [0.3, 0.2, 5.0, 1.0, 1.0, 1.0, 1.0]
    sum(c * (r_ * w.to(r_.dtype)).sum() for c, r_, w in zip(coef, ref, ws)).backward()
    sum(c * (g_ * w.to(dev()).view_as(g_)).sum() for c, g_, w in zip(coef, got, ws)).backward()
    assert rel_err(hg.grad, hr.grad) < 2e-4


@pytest.mark.parametrize('fourier,zd,F_', [(False, 2, 64), (True, 3, 32)])
def test_decoder_ends(fourier, zd, F_):
    """dec_l0 / latent / fourier kernels against torch."""
    B, Np = 2, 50
    Nt = B * Np
    xr = (torch.rand(B, Np, 2, generator=torch.Generator().manual_seed(1)) * 2 - 1)
    z = rnd(B, zd, seed=2)
    Wl = rnd(F_, zd, seed=3)
    LB = torch.empty(B, F_, device=dev())
    call('tvae_latent_bias', Wl.to(dev()), z.to(dev()), LB, B, F_, zd)
    assert rel_err(LB, z.double() @ Wl.double().t()) < TOL
    if not fourier:
        Wc, bc = rnd(F_, 2, seed=4), rnd(F_, seed=5)
        h = torch.empty(F_, Nt, device=dev())
        call('tvae_dec_l0_fwd', xr.to(dev()), Wc.to(dev()), bc.to(dev()), LB, h, Nt, F_, Nt, Np, 1, SLOPE)
        ref = F.leaky_relu(xr.view(Nt, 2).double() @ Wc.double().t() + bc.double() +
                           (z.double() @ Wl.double().t()).repeat_interleave(Np, 0), SLOPE).t()
        assert rel_err(h, ref) < TOL
    else:
        Ff, sigma = 48, 2.0 / 27
        Wf = rnd(Ff, 2, seed=6)
        bf = torch.rand(Ff, generator=torch.Generator().manual_seed(7)) * 2 * math.pi
        feat = torch.empty(Ff, Nt, device=dev())
        call('tvae_fourier_fwd', xr.to(dev()), Wf.to(dev()), bf.to(dev()), sigma, feat, Nt, Ff, Nt)
        x64 = xr.view(Nt, 2).double().requires_grad_(True)
        w32 = (Wf / torch.tensor(sigma, dtype=torch.float32)).double()
        ref = torch.cos(x64 @ w32.t() + bf.double())
        assert rel_err(feat, ref.t()) < 5e-5
        g = rnd(Ff, Nt, seed=8)
        (ref.t() * g.double()).sum().backward()
        gx = torch.empty(Nt, 2, device=dev())
        call('tvae_fourier_bwd', xr.to(dev()), Wf.to(dev()), bf.to(dev()), sigma, g.to(dev()), Nt, Ff, Nt, gx)
        assert rel_err(gx, x64.grad) < 5e-5
    S = rnd(B, F_, seed=9)
    dWl = torch.empty(F_, zd, device=dev())
    dz = torch.empty(B, zd, device=dev())
    call('tvae_latent_bwd', S.to(dev()), Wl.to(dev()), z.to(dev()), dWl, dz, B, F_, zd)
    assert rel_err(dWl, S.double().t() @ z.double()) < TOL
    assert rel_err(dz, S.double() @ Wl.double()) < TOL


def test_implicit_wgrad_and_multi_tile_dft_fp64():
    """Weight gradient with the implicit gradient operand over several reduction steps per slice, and the
    frequency-domain convolution on a batch whose (image, row) columns span several tiles: fp64 comparison."""
    from tvae._lib import query
    def rel(a, b):
        a, b = a.double().cpu(), b.double().cpu()
        return float((a - b).norm() / b.norm())
    g = torch.Generator().manual_seed(3)
    # weight gradient, several reduction steps per slice, implicit gradient operand
    M, N, K = 512, 20000 // 16 * 16, 384
    H = torch.randn(M, N, generator=g).clamp(-0.9, 0.9); X = torch.randn(K, N, generator=g)
    wo = torch.randn(M, generator=g); gy = torch.randn(N, generator=g)
    dW = torch.empty(M, K, device=dev()); ws = torch.empty(1 << 26, device=dev())       # >= tvae_linear_wgrad_x6_ws_floats(M, N, K): slices x M x K
    call('tvae_linear_wgrad_x6', H.to(dev()), X.to(dev()), dW, ws, ws.numel(), M, N, K, N, N, 0, wo.to(dev()), gy.to(dev()), 1, 0.01,
         None, None, None, None, 0, None, 3)
    d = wo.double()[:, None] * gy.double()[None, :] * torch.where(H.double() > 0, 1.0, 0.01)
    e1 = rel(dW, d @ X.double().t())
    # frequency-domain convolution, forward + weight gradient, a batch whose (image, row) columns span several tiles
    B, Cin, n, k, pad, C, R = 12, 1, 64, 64, 16, 32, 8
    Ho = n + 2 * pad - k + 1
    y = torch.rand(B, Cin, n, n, generator=g); bank = torch.randn(C * R, k * k, generator=g) * 0.02; bias = torch.randn(C, generator=g)
    at = torch.zeros(query('tvae_conv1_dft_at_floats', B, Cin, n, k, pad, C, R), device=dev())
    wsd = torch.empty(query('tvae_conv1_dft_ws_floats', B, Cin, n, k, pad, C, R), device=dev())
    out = torch.empty(C, B * R * Ho * Ho, device=dev())
    call('tvae_conv1_fwd_dft', y.to(dev()), bank.to(dev()), bias.to(dev()), out, at, wsd, wsd.numel(), B, Cin, n, k, pad, C, R, 0, 0.01, 3)
    ref = torch.nn.functional.conv2d(y.double(), bank.double().view(C * R, Cin, k, k), padding=pad).view(B, C, R, Ho, Ho) + bias.double().view(1, C, 1, 1, 1)
    e2 = rel(out.view(C, B, R, Ho, Ho).permute(1, 0, 2, 3, 4), ref)
    gg = torch.randn(B, C, R, Ho, Ho, generator=g)
    dbank = torch.empty(C * R, k * k, device=dev()); dbias = torch.empty(C, device=dev())
    call('tvae_conv1_wgrad_dft', gg.permute(1, 0, 2, 3, 4).contiguous().view(C, -1).to(dev()), at, dbank, dbias, wsd, wsd.numel(),
         B, Cin, n, k, pad, C, R, 3)
    refg = torch.nn.grad.conv2d_weight(y.double(), (C * R, Cin, k, k), gg.double().view(B, C * R, Ho, Ho), padding=pad)
    e3 = rel(dbank.view(C * R, Cin, k, k), refg)
    torch.cuda.synchronize()

    assert max(e1, e2, e3) < GEMM_TOL['f32'], (e1, e2, e3)


@pytest.mark.parametrize('N,nh,act,parts', [(4096, 103, 1, 2), (1000, 8, 1, 2), (2080, 23, 1, 2), (4128, 128, 1, 2), (999, 64, 2, 2),
                                            (1024, 65, 0, 2), (4096, 103, 1, 1)])
def test_enc_tail_wide(N, nh, act, parts):
    """Round 6: the encoder tail with 8 .. 128 head rows (galaxy: z_dim = 50 -> 103 rows; reference src/models.py:347-358,
    390-392) as two chained split-pipe GEMMs per direction -- tvae_enc_tail_fwd_wide / tvae_enc_tail_dgrad_wide against
    float64: H, the head rows, the sign words, dH and dA1; ragged N (not a multiple of the 32-column chunk), head-row counts
    that leave partial k-steps (nh % 16 != 0) and partial row tiles, a row of every operand 2^-20 below the rest (h3: per
    row relative error), and the bf16 throughput arithmetic at its own tolerance."""
    from tvae._lib import query
    C = 128
    s = 2.0 ** -20
    W2, b2 = rnd(C, C, seed=1, scale=C ** -0.5), rnd(C, seed=2)
    Wh, bh = rnd(nh, C, seed=3, scale=C ** -0.5), rnd(nh, seed=4)
    A1 = rnd(C, N, seed=5)
    W2[5] *= s; b2[5] *= s                                # a row of W2 (row of H), a head row, a column of Wh (row of dH)
    Wh[nh - 2] *= s; bh[nh - 2] *= s
    Wh[:, 19] *= s
    perm = _enc_tail_perm().cpu()
    split = 'tvae_dense_split2h' if parts == 2 else 'tvae_dense_split3'

    def cells(W, rows, K, transpose):
        a3 = torch.empty(query('tvae_dense_x6_bytes', rows, K) // 4, device=dev())
        call(split, W.contiguous().to(dev()), W.shape[1], a3, a3.numel() * 4, rows, K, transpose, None, None)
        return a3

    a1max = A1.abs().amax(dim=1).contiguous().to(dev()) if parts == 2 else None
    w3, whp = cells(W2, C, C, 0), cells(Wh[:, perm], nh, C, 0)
    H = torch.full((C, N), float('nan'), device=dev())
    heads = torch.full((nh, N), float('nan'), device=dev())
    bits = torch.zeros(2, N, 4, dtype=torch.int32, device=dev()) if act == 1 else None
    call('tvae_enc_tail_fwd_wide', w3, whp, A1.to(dev()), N, b2.to(dev()), bh.to(dev()), nh, H, N, heads, N,
         bits[0] if bits is not None else None, bits[1] if bits is not None else None, C, N, act, SLOPE, parts, a1max)
    Hr = act_ref(W2.double() @ A1.double() + b2.double()[:, None], act)
    hr = Wh.double() @ Hr + bh.double()[:, None]
    tol = ROW_TOL if parts == 2 else 2e-2
    assert row_rel_err(H, Hr) < tol
    assert row_rel_err(heads, hr) < (tol if parts == 2 else 3e-2)
    # inference form: no H, no sign words, the same head rows bit for bit
    h2 = torch.full((nh, N), float('nan'), device=dev())
    call('tvae_enc_tail_fwd_wide', w3, whp, A1.to(dev()), N, b2.to(dev()), bh.to(dev()), nh, None, N, h2, N, None, None, C, N,
         act, SLOPE, parts, a1max)
    assert torch.equal(h2, heads)
    if act != 1:
        return
    # sign words: bit (row & 31) of word (row >> 5) of column n (the layout tvae_enc_tail_fwd_x6 writes)
    Hs = H.double().cpu()
    for t, ref in ((bits[0], Hs), (bits[1], A1.double())):
        w = t.cpu().numpy().astype(np.uint32)             # [N][4]
        got = ((w[:, :, None] >> np.arange(32, dtype=np.uint32)[None, None, :]) & 1).reshape(N, 128).T
        assert np.array_equal(got.astype(bool), (ref > 0).numpy())
    dheads = rnd(nh, N, seed=6)
    dheads[3] *= s
    dmax = dheads.abs().max().reshape(1).to(dev()) if parts == 2 else None
    wht, w3p = cells(Wh, C, nh, 1), cells(W2.t()[:, perm], C, C, 0)
    dH = torch.full((C, N), float('nan'), device=dev())
    dA1 = torch.full((C, N), float('nan'), device=dev())
    call('tvae_enc_tail_dgrad_wide', wht, w3p, dheads.to(dev()), N, nh, bits[0], bits[1], dH, N, dA1, N, C, N, SLOPE, parts, dmax)
    dHr = (Wh.double().t() @ dheads.double()) * dact_ref(Hs, 1)
    dAr = (W2.double().t() @ dHr) * dact_ref(A1.double(), 1)
    assert row_rel_err(dH, dHr) < tol
    assert row_rel_err(dA1, dAr) < tol
    # without the dH output: the same dA1 bit for bit
    dA2 = torch.full((C, N), float('nan'), device=dev())
    call('tvae_enc_tail_dgrad_wide', wht, w3p, dheads.to(dev()), N, nh, bits[0], bits[1], None, N, dA2, N, C, N, SLOPE, parts, dmax)
    assert torch.equal(dA2, dA1)
    with pytest.raises(Exception):                       # more head rows than the kernel's 128
        call('tvae_enc_tail_fwd_wide', w3, whp, A1.to(dev()), N, b2.to(dev()), bh.to(dev()), 129, H, N, heads, N, None, None, C, N,
             act, SLOPE, parts, a1max)
    if parts == 2:
        with pytest.raises(Exception):                   # h3 without the operand bound
            call('tvae_enc_tail_dgrad_wide', wht, w3p, dheads.to(dev()), N, nh, bits[0], bits[1], dH, N, dA1, N, C, N, SLOPE, 2, None)


def test_rowdot_seg_amax():
    """tvae_rowdot_seg's optional by-product (ABI 7): max |X| into a zeroed word, next to the segment sums."""
    M, N = 9, 70001
    X = rnd(M, N, seed=3)
    X[4, 12345] = -77.5
    tmp = torch.empty(35, M, device=dev())
    amax = torch.zeros(1, device=dev())
    call('tvae_rowdot_seg', X.to(dev()), N, None, 1, M, N, 2048, tmp, amax)
    assert float(amax) == 77.5
    assert rel_err(tmp.sum(0), X.double().sum(1)) < 1e-5


@pytest.mark.parametrize('sigma,span,amin', [(2.0 / 127, 2.0, 300.0), (0.2 / 127, 2.0, 3000.0), (0.01, 1.5, 300.0), (2.0 / 27, 1.0, 30.0)])
def test_fourier_bwd_large_arguments(sigma, span, amin):
    """tvae_fourier_bwd evaluates sin with v_sin_f32 after its own range reduction.  Against float64 at the galaxy
    sigma with |coordinates| ~ 2 (|arg| up to ~500 rad), at a 10x smaller sigma (~5 000 rad: far outside the instruction's native
    +-256 revolutions), and at the dsprites default 0.01 (reference train_dsprites.py:492-494).  The float64 reference takes
    the fp32-ROUNDED argument (what any fp32 forward, the reference's included, evaluates cos of)."""
    Nt, Ff = 4096, 256
    xr = (torch.rand(Nt, 2, generator=torch.Generator().manual_seed(1)) * 2 - 1) * span
    Wf, bf = rnd(Ff, 2, seed=2), torch.rand(Ff, generator=torch.Generator().manual_seed(3)) * 2 * math.pi
    g = rnd(Ff, Nt, seed=4)
    gx = torch.empty(Nt, 2, device=dev())
    call('tvae_fourier_bwd', xr.to(dev()), Wf.to(dev()), bf.to(dev()), sigma, g.to(dev()), Nt, Ff, Nt, gx)
    w = (Wf / sigma)                                      # fp32, as the kernel forms it
    arg = (xr[:, 0:1] * w[:, 0][None, :] + xr[:, 1:2] * w[:, 1][None, :] + bf[None, :])      # [Nt][Ff] fp32
    assert float(arg.abs().max()) > amin
    t = -torch.sin(arg.double()) * g.double().t()
    ref = torch.stack([(t * w[:, 0].double()[None, :]).sum(1), (t * w[:, 1].double()[None, :]).sum(1)], 1)
    # per element: rounding of the fp32 argument sum (up to 3 ulp of |arg|) times the slope of sin, summed over Ff features
    tol = 4e-7 * float(arg.abs().max()) + 2e-6
    assert rel_err(gx, ref) < tol, (rel_err(gx, ref), tol)


@pytest.mark.timeout(900)
@pytest.mark.parametrize('var', ['TVAE_DFT_FULL_FRAME', 'TVAE_DFT_YSPECTRAL'])
def test_conv1_dft_alternative_formulations(var):
    """The two round-4 formulations of the frequency-domain convolution that the library still carries behind a switch -- the
    reference's full zero-padded frame n + 2 pad (its own ring instances 4-6) and the transform along BOTH axes -- run the whole
    fp64 comparison above in a process of their own (the switches are read once per process).  This test is what keeps them
    alive (every remaining switch names its test)."""
    import subprocess
    import sys
    env = dict(os.environ, **{var: '1'})
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.abspath(__file__), '-q', '-x', '-k', 'conv1_dft_matches_fp64',
                        '-p', 'no:cacheprovider'], env=env, capture_output=True, text=True, timeout=850)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert ' passed' in r.stdout and 'failed' not in r.stdout


@pytest.mark.parametrize('M,K', [(512, 384), (1024, 512)])
@pytest.mark.parametrize('parts', [3, 2, 1])
def test_linear_x6_lean_store_epilogue_and_measured_maximum(M, K, parts):
    """Round 6: plain hidden-layer launches that store their output take the lean store epilogue (dense_x6_kernel<0, NP, 3>: whole
    512-row tiles -- one and two row tiles here) and can leave max |stored value| in a zeroed word (y_amax, ABI 7): forward with
    bias + LeakyReLU and without activation, data gradient under the LeakyReLU mask of a saved activation, in all three split
    arithmetics against float64; the word must hold EXACTLY the maximum of what was stored."""
    from tvae._lib import query
    N = 640
    W, X, b = rnd(M, K, seed=1, scale=K ** -0.5), rnd(K, N, seed=2), rnd(M, seed=3)
    split = 'tvae_dense_split2h' if parts == 2 else 'tvae_dense_split3'
    tol = {3: GEMM_TOL['f32'], 2: GEMM_TOL['f32'], 1: 2e-2}[parts]

    def cells(Wm, rows, Kk, tr):
        a3 = torch.empty(query('tvae_dense_x6_bytes', rows, Kk) // 4, device=dev())
        call(split, Wm.to(dev()), Wm.shape[1], a3, a3.numel() * 4, rows, Kk, tr, None, None)
        return a3

    xmax = X.abs().max().reshape(1).to(dev()) if parts == 2 else None
    w3 = cells(W, M, K, 0)
    for act in (1, 0):
        Y = torch.full((M, N), float('nan'), device=dev())
        amax = torch.zeros(1, device=dev())
        call('tvae_linear_fwd_x6', w3, X.to(dev()), b.to(dev()), None, Y, M, N, K, N, N, act, SLOPE, None, None, None, None, None, None,
             None, 0, None, parts, xmax, amax)
        ref = act_ref(W.double() @ X.double() + b.double()[:, None], act)
        assert rel_err(Y, ref) < tol
        assert float(amax) == float(Y.abs().max())
    d = rnd(K, N, seed=6)                                 # gradient of this layer's OUTPUT rows ... transposed problem: rows = K
    aux = rnd(M, N, seed=7).clamp(-0.9, 0.9)
    # dX[m][n] = act'(aux[m][n]) * sum_k Wt[k][m] d[k][n] with Wt = W^T (K x M): reuse W as the (K = rows of d) x (M = outputs) weight
    Wt = rnd(K, M, seed=8, scale=K ** -0.5)
    w3t = cells(Wt, M, K, 1)
    dmax = d.abs().max().reshape(1).to(dev()) if parts == 2 else None
    dX = torch.full((M, N), float('nan'), device=dev())
    amax = torch.zeros(1, device=dev())
    call('tvae_linear_dgrad_x6', w3t, d.to(dev()), None, aux.to(dev()), dX, K, N, M, N, N, 1, SLOPE, None, None, None, None, 0, None,
         None, None, None, None, 0, None, 0, None, None, None, None, parts, None, None, None, dmax, amax)
    refg = (Wt.double().t() @ d.double()) * dact_ref(aux.double(), 1)
    assert rel_err(dX, refg) < tol
    assert float(amax) == float(dX.abs().max())
    with pytest.raises(Exception):                       # y_amax needs a stored output with nothing fused behind it
        call('tvae_linear_fwd_x6', w3, X.to(dev()), b.to(dev()), None, None, M, N, K, N, N, 1, SLOPE, None, None, None, None, None, None,
             None, 0, None, parts, xmax, amax)
