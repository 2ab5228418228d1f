#!/usr/bin/env python3
"""Diagnostic (CPU, numpy): sweeps / rotations / skip tests of the one-sided Jacobi SVD behind the 2-view DLT on 300 pairs of
a C3' scene's seed observations (minimum view id, last observation) - what one DLT stream of k3b_expand executes.
(DESIGN_LOG.md, round 6 item 8.)"""
import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from edgegraph3d_amd import host
s = host.Synth(3)
off, view, xy = s.seeds_np()
P = np.ascontiguousarray(s.scene_np()["cam_P"]).astype(np.float64).reshape(-1,4,4)
def dlt_count(P1,x1,y1,P2,x2,y2):
    A = np.zeros((6,4))
    for k in range(4):
        A[0,k]=x1*P1[2,k]-P1[0,k]; A[1,k]=y1*P1[2,k]-P1[1,k]; A[2,k]=x1*P1[1,k]-y1*P1[0,k]
        A[3,k]=x2*P2[2,k]-P2[0,k]; A[4,k]=y2*P2[2,k]-P2[1,k]; A[5,k]=x2*P2[1,k]-y2*P2[0,k]
    At = A.T.copy(); W = (At*At).sum(1); eps = 2.2204460492503131e-15
    sweeps=0; rots=0; tests=0
    for it in range(30):
        changed=False; sweeps+=1
        for i in range(3):
            for j in range(i+1,4):
                tests+=1
                a=W[i]; b=W[j]; p=(At[i]*At[j]).sum()
                if abs(p) <= eps*np.sqrt(a*b): continue
                p*=2; beta=a-b; gamma=np.sqrt(p*p+beta*beta)
                if beta<0:
                    delta=(gamma-beta)*0.5; sn=np.sqrt(delta/gamma); c=p/(gamma*sn*2)
                else:
                    c=np.sqrt((gamma+beta)/(gamma*2)); sn=p/(gamma*c*2)
                t0=c*At[i]+sn*At[j]; t1=c*At[j]-sn*At[i]; At[i]=t0; At[j]=t1; W[i]=(t0*t0).sum(); W[j]=(t1*t1).sum()
                rots+=1; changed=True
        if not changed: break
    return sweeps, rots, tests
rng=np.random.default_rng(0); res=[]
for i in rng.choice(s.n_seeds, 300, replace=False):
    a0,k=off[i],off[i+1]-off[i]
    if k<2: continue
    v=view[a0:a0+k]; mi=int(np.argmin(v)); la=k-1
    if mi==la: continue
    res.append(dlt_count(P[v[mi]],xy[a0+mi][0],xy[a0+mi][1],P[v[la]],xy[a0+la][0],xy[a0+la][1]))
r=np.array(res); print("n",len(r),"sweeps mean %.2f max %d; rotations mean %.1f max %d; tests mean %.1f"%(r[:,0].mean(),r[:,0].max(),r[:,1].mean(),r[:,1].max(),r[:,2].mean()))
print(np.bincount(r[:,0]))
