"""Micro-benchmark of the GEMM-shaped entry points at the cfg4 (64x64, P8, B=256) shapes: prints TFLOP/s
(algorithmic FLOPs / event time).  Used for the optimisation loop and under rocprofv3 --pmc."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'target-vae_amd')]
import torch
import tvae._lib as _L
if os.environ.get('TVAE_LIB'):
    _L.LIB_PATH = os.path.abspath(os.environ['TVAE_LIB'])
from tvae._lib import call, set_gemm_mode, query
set_gemm_mode(os.environ.get('MODE', 'f32'))
print('mode', os.environ.get('MODE', 'f32'))
dev = torch.device('cuda:0')
reps = int(os.environ.get('REPS', '5'))
PARTS = int(os.environ.get('PARTS', '3'))      # 3 = exact split, 1 = bf16 throughput mode
B = int(os.environ.get('B', '256'))
only = os.environ.get('ONLY', '')


def timeit(name, flops, fn):
    if only and not any(tok in name for tok in only.split(',')):
        return
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / reps
    print(f'{name:28s} {ms:9.3f} ms  {flops / ms / 1e9:8.1f} TFLOP/s', flush=True)


C, R, Cin, n, k, pad = 128, 8, 1, 64, 64, 16
Ho = n + 2 * pad - k + 1
P = Ho * Ho
N = B * R * P
y = torch.randn(B, Cin, n, n, device=dev)
bank = torch.randn(C * R, Cin * k * k, device=dev) * 0.02
bias = torch.randn(C, device=dev)
A1 = torch.empty(C, N, device=dev)
ws = torch.empty(1 << 26, device=dev)
fl_conv = 2.0 * C * R * Cin * k * k * P * B
timeit('conv1_fwd', fl_conv, lambda: call('tvae_conv1_fwd', y, bank, bias, A1, B, Cin, n, k, pad, C, R, 1, 0.01))
dA1 = torch.randn(C, N, device=dev)
dbank = torch.empty_like(bank)
timeit('conv1_wgrad', fl_conv, lambda: call('tvae_conv1_wgrad', y, dA1, dbank, ws, ws.numel(), B, Cin, n, k, pad, C, R))
if query('tvae_conv1_x6_supported', Cin, n, k, pad):
    a3 = torch.empty(query('tvae_conv1_x6_bank_bytes', C, R, Cin, k) // 4, device=dev)
    d3 = torch.empty(query('tvae_conv1_x6_dy_bytes', B, C, R, n, k, pad) // 4, device=dev)
    call('tvae_bank_split3', bank, a3, a3.numel() * 4, C, R, Cin, k)      # always: the x6 kernels below need real operands
    timeit('x6_bank_split', 0.0, lambda: call('tvae_bank_split3', bank, a3, a3.numel() * 4, C, R, Cin, k))
    A1x = torch.empty(C, N, device=dev)
    timeit('x6_conv1_fwd', fl_conv, lambda: call('tvae_conv1_fwd_x6', y, a3, bias, A1x, B, Cin, n, k, pad, C, R, 1, 0.01))
    dbx = torch.empty_like(bank)
    call('tvae_dy_split3', dA1, d3, d3.numel() * 4, B, Cin, n, k, pad, C, R)
    timeit('x6_dy_split', 0.0, lambda: call('tvae_dy_split3', dA1, d3, d3.numel() * 4, B, Cin, n, k, pad, C, R))
    timeit('x6_conv1_wgrad', fl_conv, lambda: call('tvae_conv1_wgrad_x6', y, d3, dbx, ws, ws.numel(), B, Cin, n, k, pad, C, R))
    if not only or 'x6' in only:
        call('tvae_conv1_fwd', y, bank, bias, A1, B, Cin, n, k, pad, C, R, 1, 0.01)
        call('tvae_conv1_wgrad', y, dA1, dbank, ws, ws.numel(), B, Cin, n, k, pad, C, R)
        ref = torch.nn.functional.conv2d(y.double(), bank.double().view(C * R, Cin, k, k), None, 1, pad).view(B, C, R, Ho, Ho) + bias.double().view(1, C, 1, 1, 1)
        ref = torch.nn.functional.leaky_relu(ref, 0.01).permute(1, 0, 2, 3, 4).reshape(C, -1)
        e = lambda a: float((a.double() - ref).norm() / ref.norm())
        print('fwd rel err vs fp64: f32 %.3e  x6 %.3e' % (e(A1), e(A1x)))
        print('wgrad x6 vs f32 rel diff %.3e' % float((dbx - dbank).norm() / dbank.norm()))
W2 = torch.randn(C, C, device=dev) * 0.1
b2 = torch.randn(C, device=dev)
H = torch.empty(C, N, device=dev)
timeit('conv2_fwd 128x128', 2.0 * C * C * N, lambda: call('tvae_linear_fwd', W2, A1, b2, None, 1, None, H, C, N, C, N, N, 1, 0.01))
timeit('conv2_dgrad', 2.0 * C * C * N, lambda: call('tvae_linear_dgrad', W2, dA1, None, A1, H, C, N, C, N, N, 1, 0.01))
dW2 = torch.empty(C, C, device=dev)
timeit('conv2_wgrad', 2.0 * C * C * N, lambda: call('tvae_linear_wgrad', dA1, A1, dW2, ws, ws.numel(), C, N, C, N, N, 0))
F_, Nt = 512, B * n * n
h1 = torch.randn(F_, Nt, device=dev)
h2 = torch.empty(F_, Nt, device=dev)
h3 = torch.randn(F_, Nt, device=dev)          # distinct operands: sharing one tensor would halve the HBM traffic
W = torch.randn(F_, F_, device=dev) * 0.05
bb = torch.randn(F_, device=dev)
timeit('dec_fwd 512x512', 2.0 * F_ * F_ * Nt, lambda: call('tvae_linear_fwd', W, h1, bb, None, 1, None, h2, F_, Nt, F_, Nt, Nt, 1, 0.01))
timeit('dec_dgrad', 2.0 * F_ * F_ * Nt, lambda: call('tvae_linear_dgrad', W, h1, None, h3, h2, F_, Nt, F_, Nt, Nt, 1, 0.01))
dW = torch.empty(F_, F_, device=dev)
timeit('dec_wgrad', 2.0 * F_ * F_ * Nt, lambda: call('tvae_linear_wgrad', h1, h3, dW, ws, ws.numel(), F_, Nt, F_, Nt, Nt, 0))
w3 = torch.empty(query('tvae_dense_x6_bytes', F_, F_) // 4, device=dev)
call('tvae_dense_split3', W, F_, w3, w3.numel() * 4, F_, F_, 0, None, None)
w3t = torch.empty_like(w3)
call('tvae_dense_split3', W, F_, w3t, w3t.numel() * 4, F_, F_, 1, None, None)
timeit('x6_dec_fwd', 2.0 * F_ * F_ * Nt, lambda: call('tvae_linear_fwd_x6', w3, h1, bb, None, h2, F_, Nt, F_, Nt, Nt, 1, 0.01, None, None, None, None, None, None, None, 0, None, PARTS))
timeit('x6_dec_dgrad', 2.0 * F_ * F_ * Nt, lambda: call('tvae_linear_dgrad_x6', w3t, h1, None, h3, h2, F_, Nt, F_, Nt, Nt, 1, 0.01, None, None, None, None, 0, None, None, None, None, None, 0, None, 0, None, None, None, None, PARTS))
timeit('x6_dec_dgrad_nomask', 2.0 * F_ * F_ * Nt, lambda: call('tvae_linear_dgrad_x6', w3t, h1, None, None, h2, F_, Nt, F_, Nt, Nt, 0, 0.01, None, None, None, None, 0, None, None, None, None, None, 0, None, 0, None, None, None, None, PARTS))
timeit('x6_dec_fwd_res', 2.0 * F_ * F_ * Nt, lambda: call('tvae_linear_fwd_x6', w3, h1, bb, h3, h2, F_, Nt, F_, Nt, Nt, 1, 0.01, None, None, None, None, None, None, None, 0, None, PARTS))
timeit('x6_dec_wgrad', 2.0 * F_ * F_ * Nt, lambda: call('tvae_linear_wgrad_x6', h1, h3, dW, ws, ws.numel(), F_, Nt, F_, Nt, Nt, 0, None, None, 0, 0.01, None, None, None, None, 0, None, PARTS))
if query('tvae_conv1_dft_supported', B, Cin, n, k, pad, C, R):
    at = torch.zeros(query('tvae_conv1_dft_at_floats', B, Cin, n, k, pad, C, R), device=dev)
    wsd = torch.empty(query('tvae_conv1_dft_ws_floats', B, Cin, n, k, pad, C, R), device=dev)
    A1d = torch.empty(C, N, device=dev)
    timeit('dft_conv1_fwd', fl_conv, lambda: call('tvae_conv1_fwd_dft', y, bank, bias, A1d, at, wsd, wsd.numel(), B, Cin, n, k, pad, C, R, 1, 0.01, PARTS))
    dbd = torch.empty_like(bank)
    timeit('dft_conv1_wgrad', fl_conv, lambda: call('tvae_conv1_wgrad_dft', dA1, at, dbd, None, wsd, wsd.numel(), B, Cin, n, k, pad, C, R, PARTS))
    if not only or 'dft' in only:
        call('tvae_conv1_fwd', y, bank, bias, A1, B, Cin, n, k, pad, C, R, 1, 0.01)
        call('tvae_conv1_wgrad', y, dA1, dbank, ws, ws.numel(), B, Cin, n, k, pad, C, R)
        print('dft fwd vs f32-MFMA rel diff %.3e   wgrad rel diff %.3e' % (float((A1d - A1).norm() / A1.norm()), float((dbd - dbank).norm() / dbank.norm())))
if only and 'zero' in only:
    w3z = torch.zeros_like(w3)
    hz = torch.zeros_like(h1)
    timeit('zero_w_x6_dec_fwd', 2.0 * F_ * F_ * Nt, lambda: call('tvae_linear_fwd_x6', w3z, h1, bb, None, h2, F_, Nt, F_, Nt, Nt, 1, 0.01, None, None, None, None, None, None, None, 0, None, PARTS))
    timeit('zero_wx_x6_dec_fwd', 2.0 * F_ * F_ * Nt, lambda: call('tvae_linear_fwd_x6', w3z, hz, bb, None, h2, F_, Nt, F_, Nt, Nt, 1, 0.01, None, None, None, None, None, None, None, 0, None, PARTS))
    timeit('zero_ref_x6_dec_fwd', 2.0 * F_ * F_ * Nt, lambda: call('tvae_linear_fwd_x6', w3, h1, bb, None, h2, F_, Nt, F_, Nt, Nt, 1, 0.01, None, None, None, None, None, None, None, 0, None, PARTS))
if only and 'tail' in only:
    xr2 = torch.randn(Nt, 2, device=dev); wc2 = torch.randn(F_, 2, device=dev)
    gxr = torch.empty(Nt, 2, device=dev); partf = torch.empty((Nt // 128) * F_ * 3, device=dev)
    wo1 = torch.randn(F_, device=dev); gy1 = torch.randn(Nt, device=dev)
    fl = 2.0 * F_ * F_ * Nt
    timeit('tail_dgrad_plain', fl, lambda: call('tvae_linear_dgrad_x6', w3t, h1, None, h3, h2, F_, Nt, F_, Nt, Nt, 1, 0.01, None, None, None, None, 0, None, None, None, None, None, 0, None, 0, None, None, None, None, PARTS))
    timeit('tail_dgrad_intail', fl, lambda: call('tvae_linear_dgrad_x6', w3t, h1, None, h3, None, F_, Nt, F_, Nt, Nt, 1, 0.01, xr2, wc2, gxr, partf, partf.numel(), None, None, None, None, None, 0, None, 0, None, None, None, None, PARTS))
    timeit('tail_dgrad_virt', fl, lambda: call('tvae_linear_dgrad_x6', w3t, h1, None, h3, h2, F_, Nt, F_, Nt, Nt, 1, 0.01, None, None, None, None, 0, wo1, gy1, None, None, None, 0, None, 0, None, None, None, None, PARTS))
    timeit('tail_dgrad_both', fl, lambda: call('tvae_linear_dgrad_x6', w3t, h1, None, h3, None, F_, Nt, F_, Nt, Nt, 1, 0.01, xr2, wc2, gxr, partf, partf.numel(), wo1, gy1, None, None, None, 0, None, 0, None, None, None, None, PARTS))
    timeit('tail_wgrad_plain', fl, lambda: call('tvae_linear_wgrad_x6', h1, h3, dW, ws, ws.numel(), F_, Nt, F_, Nt, Nt, 0, None, None, 0, 0.01, None, None, None, None, 0, None, PARTS))
    Np_ = n * n
    bc2 = torch.randn(F_, device=dev); lb2 = torch.randn(B, F_, device=dev)
    va = (xr2, wc2, bc2, lb2, Np_)
    # the step's actual launches: forward with the recomputed first layer (+ fused output column), data gradient with the
    # implicit gradient + fused first-layer backward + recomputed mask, weight gradient with both implicit operands
    cy = torch.empty(Nt, device=dev)
    timeit('tail_fwd_step', fl, lambda: call('tvae_linear_fwd_x6', w3, None, bb, None, h2, F_, Nt, F_, Nt, Nt, 1, 0.01, wo1, bb[:1].contiguous(), cy, *va, None, PARTS))
    sb = torch.zeros(F_, Nt // 32, dtype=torch.int32, device=dev)
    timeit('tail_fwd_step+bits', fl, lambda: call('tvae_linear_fwd_x6', w3, None, bb, None, h2, F_, Nt, F_, Nt, Nt, 1, 0.01, wo1, bb[:1].contiguous(), cy, *va, sb, PARTS))
    timeit('tail_fwd_step_nostore', fl, lambda: call('tvae_linear_fwd_x6', w3, None, bb, None, None, F_, Nt, F_, Nt, Nt, 1, 0.01, wo1, bb[:1].contiguous(), cy, *va, None, PARTS))
    # the step's data gradient: two-valued implicit gradient (weights scaled by wo before the split, 0 / 1 operand)
    csum2 = torch.empty(F_, device=dev)
    w3s = torch.empty_like(w3)
    call('tvae_dense_split3', W, F_, w3s, w3s.numel() * 4, F_, F_, 1, wo1, csum2)
    timeit('tail_dgrad_step', fl, lambda: call('tvae_linear_dgrad_x6', w3s, h1, None, None, None, F_, Nt, F_, Nt, Nt, 1, 0.01, xr2, wc2, gxr, partf, partf.numel(), None, gy1, csum2, bc2, lb2, Np_, None, 0, None, None, None, None, PARTS))
    timeit('tail_dgrad_generic', fl, lambda: call('tvae_linear_dgrad_x6', w3t, h1, None, None, None, F_, Nt, F_, Nt, Nt, 1, 0.01, xr2, wc2, gxr, partf, partf.numel(), wo1, gy1, None, bc2, lb2, Np_, None, 0, None, None, None, None, PARTS))
    hbits = torch.zeros(F_, Nt // 32, dtype=torch.int32, device=dev).random_(-2 ** 31, 2 ** 31 - 1)
    timeit('tail_wgrad_step', fl, lambda: call('tvae_linear_wgrad_x6', None, None, dW, ws, ws.numel(), F_, Nt, F_, Nt, Nt, 0, wo1, gy1, 1, 0.01, *va, hbits, PARTS))
    timeit('tail_wgrad_from_H', fl, lambda: call('tvae_linear_wgrad_x6', h1, None, dW, ws, ws.numel(), F_, Nt, F_, Nt, Nt, 0, wo1, gy1, 1, 0.01, *va, None, PARTS))
    timeit('tail_wgrad_virt', fl, lambda: call('tvae_linear_wgrad_x6', h1, h3, dW, ws, ws.numel(), F_, Nt, F_, Nt, Nt, 0, wo1, gy1, 1, 0.01, None, None, None, None, 0, None, PARTS))
if only and 'enc' in only:
    # encoder tail (conv2 1x1x1 + head projection), current separate launches vs the fused split-pipe kernels
    nh = 7
    Wh = torch.randn(nh, C, device=dev) * 0.1
    bh = torch.randn(nh, device=dev)
    heads = torch.empty(nh, N, device=dev)
    A1.normal_()
    fl2 = 2.0 * C * C * N
    timeit('enc_conv2_fwd_f32', fl2, lambda: call('tvae_linear_fwd', W2, A1, b2, None, 1, None, H, C, N, C, N, N, 1, 0.01))
    timeit('enc_heads_fwd', 2.0 * nh * C * N, lambda: call('tvae_heads_fwd', Wh, H, N, bh, heads, N, nh, C, N))
    w23 = torch.empty(query('tvae_dense_x6_bytes', C, C) // 4, device=dev)
    call('tvae_dense_split3', W2, C, w23, w23.numel() * 4, C, C, 0, None, None)
    H2 = torch.empty(C, N, device=dev)
    heads2 = torch.empty(nh, N, device=dev)
    timeit('enc_tail_fwd_x6', fl2, lambda: call('tvae_enc_tail_fwd_x6', w23, A1, N, b2, Wh, bh, nh, H2, N, heads2, N, None, None, C, N, 1, 0.01, PARTS))
    bits = torch.zeros(2, N, 4, dtype=torch.int32, device=dev)
    timeit('enc_tail_fwd_x6+bits', fl2, lambda: call('tvae_enc_tail_fwd_x6', w23, A1, N, b2, Wh, bh, nh, H2, N, heads2, N, bits[0], bits[1], C, N, 1, 0.01, PARTS))
    print('fused fwd vs separate: H %.3e heads %.3e' % (float((H2 - H).norm() / H.norm()), float((heads2 - heads).norm() / heads.norm())))
    # backward: separate launches (head projection backward, conv2 data gradient, conv2 weight gradient) vs fused
    dheads = torch.randn(nh, N, device=dev)
    dH = torch.empty(C, N, device=dev)
    npan = (N + 511) // 512
    part = torch.empty(npan * C * (nh + 1), device=dev)
    tot = torch.empty(nh + 1, C, device=dev)
    timeit('enc_heads_bwd', 2.0 * nh * C * N, lambda: call('tvae_heads_bwd', Wh, dheads, N, H, N, dH, N, nh, C, N, 1, 0.01, part, part.numel(), tot))
    dA1b = torch.empty(C, N, device=dev)
    timeit('enc_conv2_dgrad_f32', fl2, lambda: call('tvae_linear_dgrad', W2, dH, None, A1, dA1b, C, N, C, N, N, 1, 0.01))
    timeit('enc_conv2_wgrad_f32', fl2, lambda: call('tvae_linear_wgrad', dH, A1, dW2, ws, ws.numel(), C, N, C, N, N, 0))
    from tvae.ops import _enc_tail_perm
    w2p = torch.empty_like(w23)
    call('tvae_dense_split3', W2.t()[:, _enc_tail_perm(dev)].contiguous(), C, w2p, w2p.numel() * 4, C, C, 0, None, None)
    wh3 = torch.empty(query('tvae_dense_x6_bytes', C, nh) // 4, device=dev)
    call('tvae_dense_split3', Wh, C, wh3, wh3.numel() * 4, C, nh, 1, None, None)
    dA1c = torch.empty(C, N, device=dev)
    timeit('enc_tail_dgrad_x6', fl2, lambda: call('tvae_enc_tail_dgrad_x6', w2p, wh3, dheads, N, nh, bits[0], bits[1], dA1c, N, C, N, 0.01, PARTS))
    print('fused dgrad vs separate: %.3e' % float((dA1c - dA1b).norm() / dA1b.norm()))
