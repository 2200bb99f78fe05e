import os, sys, ctypes as C
sys.path.insert(0,'/root/repo')
import numpy as np
from tstwo_amd import _lib as L
L.init(0)
n=int(sys.argv[1]) if len(sys.argv)>1 else 16
N=1<<n
rng=np.random.default_rng(0)
a=rng.integers(0,L.P,size=N,dtype=np.uint32)
half=1<<(31-(n+1))
tw=L.DeviceBuffer(2*N)
L.call("tstwo_twiddles_build", half, n-1, C.c_void_p(tw.ptr), C.c_void_p(0))
def run(mode):
    os.environ["TSTWO_CFFT_GENERIC"]=str(mode)
    b=L.DeviceBuffer(4*N); b.upload(a)
    L.call("tstwo_cfft_evaluate", L.ptr_array([b.ptr]), 1, n, half, C.c_void_p(tw.ptr), n-1)
    L.sync()
    return b.download(np.uint32, N)
good=run(4|2); bad=run(4)
d=np.nonzero(good!=bad)[0]
print("n",n,"pass A only: mismatches",d.size,"of",N)
P=L.P
otw=None
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc
otw,_=orc.precompute_twiddles(half,n-1,inverse=False); Lw=len(otw)
def layer(v,i):
    v=v.astype(np.uint64).copy()
    for h in range(1<<(n-1-i)):
        t=int(otw[Lw-(1<<(n-i))+h])
        x=np.arange(1<<i)+(h<<(i+1)); y=x+(1<<i)
        m=v[y]*t%P; a0=v[x].copy()
        v[x]=(a0+m)%P; v[y]=(a0+P-m)%P
    return v
v=a
for i in range(n-1,12,-1):
    v=layer(v,i)
    print("after layer",i,": equals fast?", (v==bad).all(), " equals generic?", (v==good).all())
print("input", a[:6]); print("good ", good[:6]); print("fast ", bad[:6])
v2=layer(layer(a,15),14)
# hypotheses for the last layer's twiddle
for name,tfun in [("t=0",lambda t:0),("t=1",lambda t:1),("2t",lambda t:2*t%P),("t/2",lambda t:t*pow(2,P-2,P)%P)]:
    v=v2.astype(np.uint64).copy(); i=13
    for h in range(1<<(n-1-i)):
        t=tfun(int(otw[Lw-(1<<(n-i))+h]))
        x=np.arange(1<<i)+(h<<(i+1)); y=x+(1<<i)
        m=v[y]*t%P; a0=v[x].copy(); v[x]=(a0+m)%P; v[y]=(a0+P-m)%P
    print(name, (v==bad).all(), int((v==bad).sum()))
print("fast==input?", (bad==a).sum(), " fast==after15,14?", (bad==v2).sum())
