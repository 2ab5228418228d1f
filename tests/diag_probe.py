import sys, ctypes as C, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from edgegraph3d_amd import api, host, _cdefs as D
from oracle import binding as ob
L = api.lib()
L.eg3d_probe_arith.argtypes=[C.c_void_p,C.c_uint64,D.f64p,D.f64p,D.f64p,D.f64p,D.f32p,D.f32p,D.f32p,D.f32p]
L.eg3d_probe_triangulate.argtypes=[C.c_void_p,C.c_uint64,C.c_int,D.i32p,D.f32p,D.f32p,D.u8p,D.f64p]
s = host.Synth(1)
ctx = api.Context(s.scene)
rng = np.random.default_rng(1)
n = 200000
a = rng.standard_normal(n)*10**rng.uniform(-8,8,n); b = rng.standard_normal(n)*10**rng.uniform(-8,8,n); c = rng.standard_normal(n)*10**rng.uniform(-8,8,n)
fa = a.astype(np.float32); fb=b.astype(np.float32); fc=c.astype(np.float32)
od = np.zeros((5,n)); of = np.zeros((5,n),np.float32)
rc = L.eg3d_probe_arith(ctx._h, n, D.np_ptr(a,C.c_double), D.np_ptr(b,C.c_double), D.np_ptr(c,C.c_double), D.np_ptr(od,C.c_double), D.np_ptr(fa,C.c_float), D.np_ptr(fb,C.c_float), D.np_ptr(fc,C.c_float), D.np_ptr(of,C.c_float))
assert rc==0
ref_d = [a/b, np.sqrt(np.abs(a)), (a*b)+c, a.astype(np.float32).astype(np.float64), 1./np.sqrt(np.abs(b))]
names=["f64 div","f64 sqrt","f64 mul+add","f64->f32->f64","f64 1/sqrt"]
for i in range(5):
    bad = np.nonzero(od[i].view(np.uint64)!=ref_d[i].view(np.uint64))[0]
    print(names[i], "mismatches", len(bad), (od[i][bad[:3]], ref_d[i][bad[:3]]) if len(bad) else "")
dx=(fa-fc).astype(np.float64); dy=(fb-fa).astype(np.float64)
ref_f = [fa/fb, np.sqrt(np.abs(fa)), (fa*fb)+fc, (dx*dx+dy*dy).astype(np.float32), np.sqrt(np.abs(fa))]
names=["f32 div","f32 sqrt (EG3D_SQRTF)","f32 mul+add","dist2","f32 __builtin_sqrtf"]
for i in range(5):
    bad = np.nonzero(of[i].view(np.uint32)!=ref_f[i].view(np.uint32))[0]
    print(names[i], "mismatches", len(bad), (of[i][bad[:3]], ref_f[i][bad[:3]]) if len(bad) else "")
# triangulation probe: seeds' own observations (3 views) + degenerate variants
off, view, xy = s.seeds_np()
cases_v=[]; cases_xy=[]
for i in range(s.n_seeds):
    a0=off[i]; k=off[i+1]-a0
    if k>=3:
        cases_v.append(view[a0:a0+3]); cases_xy.append(xy[a0:a0+3])
        # degenerate: min view last
        cases_v.append(view[a0:a0+3][::-1].copy()); cases_xy.append(xy[a0:a0+3][::-1].copy())
cv=np.ascontiguousarray(np.array(cases_v,np.int32)); cxy=np.ascontiguousarray(np.array(cases_xy,np.float32))
m=len(cv)
X=np.zeros((m,3),np.float32); val=np.zeros(m,np.uint8); dlt=np.zeros((m,3))
rc=L.eg3d_probe_triangulate(ctx._h,m,3,D.np_ptr(cv,C.c_int32),D.np_ptr(cxy,C.c_float),D.np_ptr(X,C.c_float),D.np_ptr(val,C.c_uint8),D.np_ptr(dlt,C.c_double))
assert rc==0
P=s.scene_np()["cam_P"]
OL=ob.lib()
nbadX=nbadD=nbadV=0
for i in range(m):
    Xo=np.zeros(3,np.float32); deg=C.c_int(0)
    ids=(C.c_int*3)(*[int(v) for v in cv[i]])
    ok=OL.orc_triangulate(D.np_ptr(P,C.c_float),ids,D.np_ptr(cxy[i],C.c_float),3,D.np_ptr(Xo,C.c_float),C.byref(deg))
    d0=np.zeros(3)
    mi=int(np.argmin(cv[i]))
    OL.orc_dlt(D.np_ptr(P[cv[i][mi]],C.c_float),D.np_ptr(cxy[i][mi],C.c_float),D.np_ptr(P[cv[i][2]],C.c_float),D.np_ptr(cxy[i][2],C.c_float),D.np_ptr(d0,C.c_double))
    if not np.array_equal(d0.view(np.uint64),dlt[i].view(np.uint64)):
        nbadD+=1
        if nbadD<4: print("DLT diff case",i,"deg",deg.value,d0,dlt[i])
    if bool(ok)!=bool(val[i]): nbadV+=1
    elif ok and not np.array_equal(Xo.view(np.uint32),X[i].view(np.uint32)):
        nbadX+=1
        if nbadX<4: print("X diff case",i,"deg",deg.value,Xo,X[i])
print("tri cases",m,"dlt mismatches",nbadD,"valid mismatches",nbadV,"X mismatches",nbadX)
