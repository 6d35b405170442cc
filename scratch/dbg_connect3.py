import numpy as np, sys, os
sys.path.insert(0, '.')
from mitsubaer_amd import capi, params as P, synth
from tests import scenes
p = scenes.curved_scene(N=24, rif="radial", stepper=P.STEP_VERLET)
box = ([-1.2] * 3, [1.2] * 3)
p = p.copy(boundary=P.BOUNDARY_SDF, sdf=-synth.sphere_sdf(64, radius=0.9, aabb_min=box[0], aabb_max=box[1]), sdf_aabb=box)
rng = np.random.RandomState(0)
n = 64
p1 = rng.uniform(-0.45, 0.45, (n, 3)).astype(np.float32); p2 = rng.uniform(-0.45, 0.45, (n, 3)).astype(np.float32)
for libn in sys.argv[1:]:
    capi.LIB_PATH = "mitsubaer_amd/" + libn
    ctx = capi.Context(0)
    for lay in (capi.LAYOUT_DENSE, capi.LAYOUT_CELL8):
        sc, vols = ctx.upload_scene(p, layout=lay)
        a = ctx.connect(sc, p1, p2, 1)
        print(libn, "layout", lay, "ok frac", (a[:, 0] == 1).mean(), a[0, [0, 1, 2, 8, 10, 11, 5, 6, 7]], p1[0], p2[0])
