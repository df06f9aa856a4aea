import sys, numpy as np
sys.path.insert(0,'/root/repo/sed-net_amd')
from sednet_hip import synth
N=10000;k=20;W=1.0
x=synth.batch_clouds(1,N,seed0=int(sys.argv[1]) if len(sys.argv)>1 else 1234)[0][0].T.astype(np.float64)  # N,6
P=x[:,:3];Nn=x[:,3:]
def morton(Pm,bits=5):
    lo=Pm.min(0);hi=Pm.max(0);q=np.minimum(((Pm-lo)/(hi-lo+1e-12)*(1<<bits)).astype(np.int64),(1<<bits)-1)
    code=np.zeros(len(Pm),np.int64);D=Pm.shape[1]
    for b in range(bits):
        for d in range(D): code|=((q[:,d]>>b)&1)<<(b*D+d)
    return np.argsort(code,kind='stable')
def run(order,name):
    Ps=P[order];Ns=Nn[order];nt=(N+31)//32
    lo=np.array([Ps[32*t:32*t+32].min(0) for t in range(nt)]);hi=np.array([Ps[32*t:32*t+32].max(0) for t in range(nt)])
    vis_own=0;vis_true=0;tot=0;cands=[]
    for w in range(0,N,64*5):
        q=slice(w,min(w+64,N))
        Dp=((Ps[q,None,:]-Ps[None])**2).sum(2); Dn=2-2*Ns[q]@Ns.T; M=Dp*(1+W*Dn)
        Ttrue=np.sort(M,1)[:,k-1]
        b0=(w//256)*256; own=M[:,b0:b0+256]; Town=np.sort(own,1)[:,k-1]
        qlo=Ps[q].min(0);qhi=Ps[q].max(0)
        g=np.maximum(np.maximum(lo-qhi,qlo-hi),0); lb=(g**2).sum(1)
        vis_own+=(lb<=Town.max()).sum(); vis_true+=(lb<=Ttrue.max()).sum(); tot+=nt
        cands.append((M<=Town[:,None]).sum(1))
    c=np.concatenate(cands)
    print(f'{name}: tiles visited with own-block T {vis_own/tot:.3f}, with exact T {vis_true/tot:.3f}; candidates under own T: mean {c.mean():.0f} p99 {np.percentile(c,99):.0f} max {c.max()}')
run(morton(x,5),'morton 6-d (5 bits)')
run(morton(P,10),'morton xyz (10 bits)')
