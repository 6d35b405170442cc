import numpy as np, sys
sys.path.insert(0, '.')
from mitsubaer_amd import capi, params as P
from tests import scenes
ctx = capi.Context(0)
p = scenes.curved_scene(N=24, rif="radial", stepper=P.STEP_VERLET)
rng = np.random.RandomState(0)
n = 64
p1 = rng.uniform(-0.6, 0.6, (n, 3)).astype(np.float32); p2 = rng.uniform(-0.6, 0.6, (n, 3)).astype(np.float32)
for lay in (capi.LAYOUT_DENSE, capi.LAYOUT_CELL8):
    for bl in (1, 0):
        ctx.set_option("buffer_loads", bl)
        sc, vols = ctx.upload_scene(p, layout=lay)
        a = ctx.connect(sc, p1, p2, 1)
        print("layout", lay, "buffer_loads", bl, "ok frac", (a[:, 0] == 1).mean(), a[0])
cc = capi.Context(0, check=True)
sc, vols = cc.upload_scene(p)
a = cc.connect(sc, p1, p2, 1)
print("check build ok frac", (a[:, 0] == 1).mean(), cc.debug_bounds())
