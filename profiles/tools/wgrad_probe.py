import os, sys, torch
ROOT='/root/repo'
sys.path[:0]=[ROOT, ROOT+'/target-vae_amd']
from tvae._lib import call, query
dev=torch.device('cuda:0')
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    s,e=torch.cuda.Event(True),torch.cuda.Event(True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/reps
for (M,N,K) in [(512,131072,512),(512,1048576,512),(512,524288,512),(512,200704,1026),(512,32768,512)]:
    d=torch.randn(M,N,device=dev); X=torch.randn(K,N,device=dev); dW=torch.empty(M,K,device=dev)
    ws=torch.empty(max(query('tvae_linear_wgrad_x6_ws_floats',M,N,K),1<<24),device=dev)
    am=d.abs().max().reshape(1); xm=X.abs().max().reshape(1)
    for p in (2,3):
        ms=t(lambda: call('tvae_linear_wgrad_x6', d, X, dW, ws, ws.numel(), M, N, K, N, N, 0, None, None, 0, 0.01, None,None,None,None,0,None,p,None,K,None, am if p==2 else None, xm if p==2 else None, 0))
        fl=2.0*M*N*K*(3 if p==2 else 6)
        print('M %d N %d K %d parts %d: %.3f ms  %.2f PF/s executed' % (M,N,K,p,ms,fl/ms/1e12))
# row stride a power of two (N = 2^17 columns of fp32 = 512 KB) against the same problem with padded rows
print('--- padded leading dimension (ld = N + 64)')
for (M,N,K) in [(512,131072,512),(512,1048576,512)]:
    ld = N + 64
    dbuf=torch.randn(M,ld,device=dev); Xbuf=torch.randn(K,ld,device=dev); dW=torch.empty(M,K,device=dev)
    ws=torch.empty(max(query('tvae_linear_wgrad_x6_ws_floats',M,N,K),1<<24),device=dev)
    am=dbuf.abs().max().reshape(1); xm=Xbuf.abs().max().reshape(1)
    for p in (2,3):
        ms=t(lambda: call('tvae_linear_wgrad_x6', dbuf, Xbuf, dW, ws, ws.numel(), M, N, K, ld, ld, 0, None, None, 0, 0.01, None,None,None,None,0,None,p,None,K,None, am if p==2 else None, xm if p==2 else None, 0))
        fl=2.0*M*N*K*(3 if p==2 else 6)
        print('M %d N %d K %d ld %d parts %d: %.3f ms  %.2f PF/s executed' % (M,N,K,ld,p,ms,fl/ms/1e12))
print('--- plain forward (bias + LeakyReLU, stored output): ld = N against ld = N + 64')
for (M,N,K) in [(512,131072,512),(512,229376,512),(512,200704,1026)]:
    for pad in (0, 64):
        ld = N + pad
        Xb=torch.randn(K,ld,device=dev); Yb=torch.empty(M,ld,device=dev)
        W=torch.randn(M,K,device=dev)*K**-0.5; b=torch.randn(M,device=dev)
        w3=torch.empty(query('tvae_dense_x6_bytes',M,K)//4,device=dev)
        call('tvae_dense_split2h', W, K, w3, w3.numel()*4, M, K, 0, None, None)
        xm=Xb.abs().max().reshape(1)
        ms=t(lambda: call('tvae_linear_fwd_x6', w3, Xb, b, None, Yb, M, N, K, ld, ld, 1, 0.01, None,None,None,None,None,None,None,0,None,2,xm,None))
        print('fwd M %d N %d K %d ld %d: %.3f ms  %.2f PF/s executed' % (M,N,K,ld,ms,2.0*M*N*K*3/ms/1e12))
