import numpy as np, sys, os
sys.path.insert(0, '.')
from mitsubaer_amd import capi, params as P
from tests import scenes
from oracle import orc
p = scenes.curved_scene(N=24, rif="radial", stepper=P.STEP_VERLET)
rng = np.random.RandomState(0)
n = 64
p1 = rng.uniform(-0.6, 0.6, (n, 3)).astype(np.float32); p2 = rng.uniform(-0.6, 0.6, (n, 3)).astype(np.float32)
b = orc.connect(p, p1, p2, 1)
print("oracle ok frac", (b[:, 0] == 1).mean(), b[0, :10])
for name in ("libmer.so", "libmer_dbgA.so", "libmer_dbgB.so", "libmer_dbgC.so"):
    capi.LIB_PATH = os.path.join("mitsubaer_amd", name)
    ctx = capi.Context(0)
    sc, vols = ctx.upload_scene(p)
    a = ctx.connect(sc, p1, p2, 1)
    print(name, "ok frac", (a[:, 0] == 1).mean(), a[0, :10])
