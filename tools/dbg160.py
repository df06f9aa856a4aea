import sys
sys.path[:0]=['/root/repo','/root/repo/sed-net_amd','/root/repo/tests']
import numpy as np, torch
from sednet_hip import ops, synth
from test_gpu_mean_shift import set_schedule, reset_schedule
N=int(sys.argv[1]) if len(sys.argv)>1 else 6003
Xs=np.stack([synth.clustered_embedding(N=N,d=140,n_clusters=8+c,sigma=0.02,seed=160+c)[0] for c in range(3)])
X=ops.pad_features(torch.from_numpy(Xs).cuda())
bw=ops.ms_bandwidth(X,90,0.003)
set_schedule("batched"); ref=ops.ms_iterate(X,bw,1)
for v in ("f16","f16c","f16/1"):
    set_schedule(v); got=ops.ms_iterate(X,bw,1)
    nan=torch.isnan(got)
    print(v,'nan rows per cloud',nan.any(2).sum(1).tolist(),'max err', (got-ref)[~nan].abs().max().item() if (~nan).any() else None)
    if nan.any():
        rows=torch.nonzero(nan.any(2)[0])[:10,0].tolist(); print(' first nan rows cloud0',rows, 'cols', torch.nonzero(nan[0,rows[0]])[:10,0].tolist() if rows else None)
reset_schedule()
