import sys, time
sys.path.insert(0,'sed-net_amd'); sys.path.insert(0,'.')
import numpy as np, torch
from sednet_hip import synth, ops
B=int(sys.argv[1]) if len(sys.argv)>1 else 16
N,d=10000,128
X=np.stack([synth.clustered_embedding(N=N,d=d,n_clusters=12+b%8,sigma=0.01,seed=b)[0] for b in range(B)])
X=torch.from_numpy(X).cuda()
def t(fn,n=3):
    fn(); torch.cuda.synchronize()
    s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): r=fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/n, r
ms,bw=t(lambda: ops.ms_bandwidth(X,150)); print('bandwidth ms',ms,'per cloud',ms/B)
ms,nx=t(lambda: ops.ms_iterate(X,bw,50),2); fl=4*N*N*d*50*B; print('iterate ms',ms,'per cloud',ms/B,'TFLOP/s',fl/ms/1e9)
ms,r=t(lambda: ops.ms_nms(nx,X,bw)); print('nms ms',ms,'per cloud',ms/B, r[2][:4].tolist())
f=torch.randn(B,N,64,device='cuda')
ms,r=t(lambda: ops.knn_features(f,20)); print('knn64 k20 ms',ms,'per cloud',ms/B)
ms,r=t(lambda: ops.knn_features(f,64)); print('knn64 k64 ms',ms,'per cloud',ms/B)
