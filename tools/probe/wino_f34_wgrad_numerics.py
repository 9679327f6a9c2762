"""Numerics of a Winograd F(3x3,4x4) weight gradient against the shipped F(3x3,2x2): one channel pair, 96 x 96 map (576 / 2304 tiles
summed in fp32), LeakyReLU-shaped input and a small output gradient, against the direct sum in float64.  CPU only.

    python3 tools/probe/wino_f34_wgrad_numerics.py"""
import numpy as np, sys
import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from wino_f43_numerics import cook_toom, scale_rows
rng=np.random.default_rng(0)
# wgrad: dW[a][b] = sum_{y,x} X[y+a][x+b] * dZ[y][x]  (3x3 output, m x m dZ tile)  == F(3, m) with "filter" = dZ tile (r=m), output m'=3
def mats(m):
    # F(out=3, r=m): A^T 3 x n, G n x m, B^T n x n ; n = m + 2
    n=m+2
    pts={4:[0,1,-1],6:[0,1,-1,2,-2]}[n]
    AT,G,BT=cook_toom(pts,3,m)
    return AT,G,BT
def wgrad_wino(X,dZ,m,dtype):
    # X [H+2,W+2] padded single channel pair; dZ [H,W]; returns 3x3
    AT,G,BT=[a.astype(dtype) for a in mats(m)]
    H,W=dZ.shape; n=m+2
    acc=np.zeros((n,n),dtype)
    for ty in range(0,H,m):
        for tx in range(0,W,m):
            d=X[ty:ty+n,tx:tx+n].astype(dtype); z=dZ[ty:ty+m,tx:tx+m].astype(dtype)
            V=(BT@d@BT.T).astype(dtype); Z=(G@z@G.T).astype(dtype)
            acc=(acc+(V*Z).astype(dtype)).astype(dtype)
    return (AT@acc@AT.T).astype(dtype)
H=W=96
errs={2:[],4:[]}
for trial in range(24):
    x=rng.standard_normal((H+2,W+2)); x=np.where(x>0,x,0.1*x); x[0,:]=x[-1,:]=0; x[:,0]=x[:,-1]=0
    dz=rng.standard_normal((H,W))*1e-3
    ref=np.zeros((3,3))
    for a in range(3):
        for b in range(3): ref[a,b]=(x[a:a+H,b:b+W]*dz).sum()
    for m in (2,4):
        r=wgrad_wino(x.astype(np.float32),dz.astype(np.float32),m,np.float32)
        errs[m].append(np.abs(r-ref).max()/np.abs(ref).max())
    # direct fp32 chain
for m in (2,4): print('F(3x3,%dx%d) fp32: max rel err over trials  mean %.2e  max %.2e'%(m,m,np.mean(errs[m]),np.max(errs[m])))
