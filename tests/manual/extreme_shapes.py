#!/usr/bin/env python3
"""Extreme shapes and value ranges against the oracle (development aid; second call of each shape is timed)."""
import sys, time
from pathlib import Path; ROOT = Path(__file__).resolve().parents[2]; sys.path[:0] = [str(ROOT), str(ROOT / 'mcmc-db_amd')]
import numpy as np
from mcmc_ref_hip import _ffi
from oracle import oracle as orc
ctx=_ffi.Context(0)
rng=np.random.default_rng(0)
def chk(x, what, min_chains=1, rel=1e-9):
    ctx.summarize(x,"pcn",min_chains=min_chains)
    t=time.perf_counter(); g=ctx.summarize(x,"pcn",min_chains=min_chains); tg=time.perf_counter()-t
    e=orc.summarize(x,"pcn",min_chains=min_chains)
    ok=True
    for k in ("mean","std","rhat","ess_bulk","ess_tail","median"):
        a,b=g[k],e[k]
        same=np.isnan(a)==np.isnan(b)
        f=np.isfinite(b)
        err=np.max(np.abs(a[f]-b[f])/np.maximum(np.abs(b[f]),1e-300)) if f.any() else 0
        if k=="mean": err=np.max(np.abs(a-b)/(np.abs(b)+e["std"]+1e-300))
        if not same.all() or err>rel: ok=False; print("  MISMATCH",k,err)
    if not (np.array_equal(g["lag_bulk"],e["lag_bulk"]) and np.array_equal(g["lag_tail"],e["lag_tail"]) and np.array_equal(g["q"],e["q"],equal_nan=True)): ok=False; print("  MISMATCH ints/q")
    print(what, x.shape, "ok" if ok else "FAIL", f"{tg*1e3:.1f} ms", flush=True)
chk(rng.normal(size=(70000,2,8)),"P=70000 tiny chains")
chk(rng.normal(size=(3,256,40)),"C=256")
chk(rng.normal(size=(5,4,2)),"N=2")
chk(rng.normal(size=(5,4,3)),"N=3")
chk(rng.normal(size=(2,1,5000)),"C=1")
chk(rng.normal(size=(1,4,500000)),"M=2M single param")
x=rng.normal(size=(2,4,300000)); x[1]=np.round(x[1],2)
chk(x,"M=1.2M ties")
chk(np.full((3,4,1000),7.25),"all constant")
x=rng.normal(size=(3,4,1000)); x[1,2]=x[1,2,0]
chk(x,"one constant chain")
x=np.cumsum(rng.normal(size=(2,4,20000)),axis=2)
chk(x,"random walks (slow mixing)")
x=rng.normal(size=(4,4,4096))*1e-300
chk(x,"denormal-scale")
x=rng.normal(size=(4,4,4096))*1e300
chk(x,"huge-scale")
print("done")
