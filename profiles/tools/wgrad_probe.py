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
