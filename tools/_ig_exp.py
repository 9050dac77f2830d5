import os, sys, numpy as np, torch
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/bench.py') else os.environ.get('GRAFT_REPO_ROOT','.'))
import bench
from adt_amd import ops
DEV="cuda:0"; B,L,V=256,200,3416; T,V1=B*L,V+1
seq,dec,pos,neg=bench.synth_batches(1,B,L,V,7)[0]
d_ids=[torch.from_numpy(a.reshape(-1).copy()).to(DEV) for a in (seq,dec,pos,neg)]
rows=[torch.randn(T,64,device=DEV) for _ in range(3)]; rows.append(rows[2])
coef=[None,None,torch.randn(T,device=DEV),torch.randn(T,device=DEV)]
sd=torch.from_numpy(np.array([5],dtype=np.uint32).view(np.int32)).to(DEV)
dE=torch.zeros(V1,64,device=DEV)
work=ops.item_sort(d_ids,V1,rows,coef,[0,0,1,1],0)
def t(fn,n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e3
for name,mask,p in (("none",0,0.5),("seq p=.5",1,0.5),("seq p=0",1,0.0),("dec",2,0.5),("pos",4,0.5),("pos+neg",12,0.5),("A=dec+pos+neg",14,0.5),("all",15,0.5)):
    us=t(lambda: ops.item_segsum(work,4,T,V1,mask,[1,2,0,0],p,sd,8.0,dE))
    print("%-16s %.1f us (segsum + carry)"%(name,us))
import ctypes
if os.environ.get("ADT_IG_STAMPS"):
    ops.item_segsum(work,4,T,V1,14,rows,coef,[0,0,1,1],[1,2,0,0],0.5,sd,0,8.0,dE)
    buf=(ctypes.c_ulonglong*256)()
    lib=ops._lib.load(); print("rc", lib.adt_ig_stamps_read(buf))
    t=np.array(list(buf),dtype=np.int64).reshape(16,16)
    t0=t[:,0].min()
    for w in range(16): print(w, [int(x-t0) if x else -1 for x in t[w,:14]])
