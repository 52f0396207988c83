import sys, os
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, oracle as O, refraction_raytracing_dxr_amd as rr
from conftest import procedural_env
def xf(tx,ty,tz,s=1.0):
    m=np.eye(4,dtype=np.float32)[:3]*np.float32(s); m[:,3]=(tx,ty,tz); return m
meshes=[]
for n in ("shell.obj","cube.obj","ott.obj"):
    m=rr.Mesh(); m.load(O.asset(n)); meshes.append(m)
inst = rr.make_instances(transforms=[xf(0,0,0),xf(0,0,-4.0),xf(0,0,4.0)], meshes=[0,1,2])
s=O.Scene()
for m in meshes: s.add_mesh(m.verts,m.indices)
oi=np.zeros(3,O.INSTANCE_DTYPE); oi["transform"]=inst["transform"]; oi["id_mask"]=inst["instance_id_mask"]; oi["hitgroup_flags"]=inst["hitgroup_flags"]; oi["blas"]=inst["blas"]
env=procedural_env(256,128,seed=4)
s.set_instances(oi); s.set_envmap(env)
sc=rr.camera_orbit(0.01); M,cam=np.array(sc.proj_inv,np.float32),np.array(sc.camera_loc,np.float32)
g=rr.Renderer(0)
ids=[]
for m in meshes:
    mid=g.upload_mesh(m.verts,m.indices); g.build_blas(mid); ids.append(mid)
i2=inst.copy(); i2["blas"]=ids
g.build_tlas(i2); g.upload_envmap(env); g.set_camera(sc)
W,H=240,135
# primary rays of row 67
rays=np.zeros(W,rr.RAY_DTYPE)
for x in range(W):
    o,d=O.camera_ray(M,cam,x,67,W,H)
    rays["origin"][x]=o; rays["dir"][x]=d
rays["tmin"]=1e-4; rays["tmax"]=100; rays["flags"]=0x10
hits=g.trace_rays(rays)
bad=0
for x in range(W):
    h=s.trace(rays["origin"][x],rays["dir"][x],1e-4,100.0,0x10,use_bvh=0)
    if bool(hits["hit"][x])!=bool(h.hit) or (h.hit and (hits["prim"][x]!=h.prim or hits["inst"][x]!=h.inst or np.float32(hits["t"][x]).view(np.uint32)!=np.float32(h.t).view(np.uint32))):
        bad+=1; print("primary x",x,"gpu",hits[x],"oracle",h.hit,h.inst,h.prim,h.t, "dir", rays["dir"][x])
print("primary mismatches", bad)
# secondary: random in-plane rays (y=0 plane) from points around
rng=np.random.default_rng(0)
n=20000
r2=np.zeros(n,rr.RAY_DTYPE)
ang=rng.uniform(0,2*np.pi,n)
r2["origin"]=np.stack([rng.uniform(-2,2,n),np.zeros(n),rng.uniform(-6,6,n)],1).astype(np.float32)
r2["dir"]=np.stack([np.cos(ang),np.zeros(n),np.sin(ang)],1).astype(np.float32)
r2["tmin"]=1e-3; r2["tmax"]=1000; r2["flags"]=rng.choice([0x10,0x20],n)
h2=g.trace_rays(r2)
bad=0
for k in range(n):
    h=s.trace(r2["origin"][k],r2["dir"][k],1e-3,1000.0,int(r2["flags"][k]),use_bvh=0)
    if bool(h2["hit"][k])!=bool(h.hit) or (h.hit and (h2["prim"][k]!=h.prim or h2["inst"][k]!=h.inst)):
        bad+=1
        if bad<=8: print("inplane",k,"gpu",h2[k],"oracle",h.hit,h.inst,h.prim,h.t,"o",r2["origin"][k],"d",r2["dir"][k])
print("in-plane mismatches", bad, "of", n)
g.set_tile_partition(0,1)
g.dispatch_rays(W,H,rr.default_params(max_refract=8, flags=rr.DISPATCH_FLOAT_OUTPUT))
rgba,f32=g.read_frame(want_float=True)
print("gpu rays", g.stats().rays)
ref=s.render(M,cam,W,H,O.default_params(use_bvh=0,max_refract=8,accum_mode=1),want_rays=True)
print("oracle brute rays", ref["stats"].rays)
d=np.abs(f32[...,:3]-ref["rgb"]).max(axis=2)
ys,xs=np.nonzero(d>0)
print("differing pixels", len(ys), list(zip(xs.tolist(),ys.tolist()))[:20])
for x,y in list(zip(xs.tolist(),ys.tolist()))[:6]:
    print((x,y), "gpu", f32[y,x,:3], "oracle", ref["rgb"][y,x], "oracle rays", ref["rays"][y,x])

# ---- walk the ray tree of a differing pixel with both tracers -----------------------------------
f32t=np.float32
def fma(a,b,c): return f32t(np.float64(a)*np.float64(b)+np.float64(c))
def dot(a,b): return fma(a[2],b[2],fma(a[1],b[1],f32t(a[0]*b[0])))
def norm(v):
    inv=f32t(1.0)/np.sqrt(dot(v,v)).astype(f32t); return (v*inv).astype(f32t)
def gtrace(o,d,tmin,tmax,fl):
    r=np.zeros(1,rr.RAY_DTYPE); r["origin"][0]=o; r["dir"][0]=d; r["tmin"]=tmin; r["tmax"]=tmax; r["flags"]=fl
    return g.trace_rays(r)[0]
def nrm_of(inst_i, prim, u, v):
    m=meshes[int(inst["blas"][inst_i])]
    n=m.verts["norm"][m.indices[3*prim:3*prim+3]]
    A,B,C=n[0],n[1],n[2]
    return norm(np.array([fma(v,C[k]-A[k],fma(u,B[k]-A[k],A[k])) for k in range(3)],f32t))
def walk(o,d,tmin,tmax,outside,count,depth=0):
    fl=0x10 if outside else 0x20
    hg=gtrace(o,d,tmin,tmax,fl); ho=s.trace(o,d,float(tmin),float(tmax),fl,use_bvh=0)
    same = bool(hg["hit"])==bool(ho.hit) and (not ho.hit or (hg["prim"]==ho.prim and hg["inst"]==ho.inst and f32t(hg["t"]).view(np.uint32)==f32t(ho.t).view(np.uint32)))
    print("  "*depth+"ray count %d outside %d o %s d %s -> gpu(hit %d inst %d prim %d t %.9g) oracle(hit %d inst %d prim %d t %.9g) %s" % (count,outside,o,d,hg["hit"],hg["inst"],hg["prim"],hg["t"],ho.hit,ho.inst,ho.prim,ho.t,"" if same else "<<<<<< MISMATCH"))
    if not ho.hit or count>=8: return
    N=nrm_of(ho.inst,ho.prim,f32t(ho.u),f32t(ho.v))
    X=np.array([fma(f32t(ho.t),d[k],o[k]) for k in range(3)],f32t)
    Nf=N if outside else -N
    eta=f32t(1.0)/f32t(1.3) if outside else f32t(1.3)
    c=dot(Nf,d); k=f32t(1.0)-f32t(eta*eta)*f32t(f32t(1.0)-f32t(c*c))
    if k>=0:
        a=f32t(f32t(eta*c)+np.sqrt(k).astype(f32t))
        d1=norm(np.array([f32t(f32t(eta*d[i])-f32t(a*Nf[i])) for i in range(3)],f32t))
        walk(X,d1,f32t(1e-3),f32t(1000),not outside,count+1,depth+1)
    if count<2:
        kk=f32t(2.0)*dot(Nf,d)
        d2=norm(np.array([f32t(d[i]-f32t(kk*Nf[i])) for i in range(3)],f32t))
        walk(X,d2,f32t(1e-3),f32t(1000),outside,count+1,depth+1)
for (x,y) in [(59,67),(94,67)]:
    print("pixel",x,y)
    o,d=O.camera_ray(M,cam,x,y,W,H)
    walk(o,d,f32t(1e-4),f32t(100),True,0)
