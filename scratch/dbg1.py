import sys; sys.path.insert(0,'.')
import numpy as np
from mitsubaer_amd import capi, params as P
from oracle import orc
from tests import scenes
from tests.test_gpu_render import CASES
ctx=capi.Context(0)
for name in ["cfg2_straight_woodcock2","parity_verlet_bspline","cfg3_curved_verlet_trilinear"]:
    p=CASES[name]()
    sc,vols=ctx.upload_scene(p)
    a=ctx.render_paths(sc,0,seed=3); b=orc.render_paths(p,0,3)
    diff=np.abs(a-b).max(2)
    bad=diff>1e-4*np.maximum(1,np.abs(b).max(2))
    print(name,"bad frac",bad.mean(),"max diff",diff.max(),"mean a",a.mean(),"mean b",b.mean())
    ys,xs=np.nonzero(bad)
    for k in range(min(8,len(ys))):
        print("   ",ys[k],xs[k],a[ys[k],xs[k]],b[ys[k],xs[k]])
    print("   hist of diff among bad:",np.percentile(diff[bad],[10,50,90]) if bad.any() else None)
