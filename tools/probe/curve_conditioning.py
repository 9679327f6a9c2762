"""One-ulp conditioning of a darkcapsule training recipe on the REFERENCE (build container only: imports /root/reference):
20-step Adam curve from the default initialisation and the spread of runs whose inputs are moved by one ulp.
    python tools/probe/curve_conditioning.py H n_grid batch seed lr steps members"""
import sys, os, time
import numpy as np, torch
sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/reference'); sys.dont_write_bytecode=True
from helpers import make_params, synth_images, synth_gtsdb_labels
import models as RM, loss_fns as RL
torch.set_num_threads(8)
def perturb(x, seed, ulps=1):
    rng=np.random.default_rng(seed)
    s=rng.integers(0,2,x.size).astype(bool).reshape(x.shape)
    up=x.copy(); dn=x.copy()
    for _ in range(ulps):
        up=np.nextafter(up,np.float32(np.inf)); dn=np.nextafter(dn,np.float32(-np.inf))
    return np.where(s,up,dn).astype(np.float32)
def run(H,g,B,seed,lr,steps,x):
    pd=make_params(model='darkcapsule',n_grid=g,darknet_input=H,recon=False)
    y=torch.from_numpy(synth_gtsdb_labels(B,g,43,seed=seed+1))
    torch.manual_seed(1234)
    net=RM.DarkCapsuleNet(pd).train()
    opt=torch.optim.Adam([q for q in net.parameters() if q.requires_grad],lr=lr)
    c=[]
    for _ in range(steps):
        loss=RL.darkcapsule_loss(net(x),y,pd); opt.zero_grad(); loss.backward(); opt.step(); c.append(loss.item())
    return np.array(c)
H,g,B,seed=int(sys.argv[1]),int(sys.argv[2]),int(sys.argv[3]),int(sys.argv[4]); lr=float(sys.argv[5]); steps=int(sys.argv[6]); nens=int(sys.argv[7])
x0=synth_images(B,H,seed=seed)
t=time.time(); base=run(H,g,B,seed,lr,steps,torch.from_numpy(x0)); print('time/run',time.time()-t)
np.set_printoptions(precision=4,linewidth=220,suppress=True)
print(base); rng=base.max()-base.min()
ens=np.array([run(H,g,B,seed,lr,steps,torch.from_numpy(perturb(x0,100+i))) for i in range(nens)])
print('max|dev|/range % per step'); print(np.abs(ens-base).max(0)/rng*100)
print('sigma/range %'); print(ens.std(0)/rng*100)
