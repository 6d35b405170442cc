#!/usr/bin/env python3
"""Regenerates tests/golden/oracle_vectors.npz: seeded input / output vectors of the CPU oracle (oracle/libmer_oracle.so) for every
row of the hot path.  These are REGRESSION vectors of our own restatement -- the reference cannot be built here
-- and they pin the oracle against silent changes; the known answers that do come from the reference are in
reference_known_answers.json.  Usage (from the repo root):  python tests/golden/make_golden.py
"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mitsubaer_amd import params as P, synth      # noqa: E402
from oracle import orc                            # noqa: E402
from tests import scenes                          # noqa: E402


def scene_set():
    """name -> SceneParams; small enough for the oracle to finish in milliseconds"""
    pt = dict(env_radiance=[0, 0, 0], point_position=[0.2, 0.3, -0.1], point_intensity=[1.0, 0.8, 0.5])
    return {
        "straight_ratio": scenes.straight_scene(N=16, w=16, h=12),
        "straight_woodcock2_u8": scenes.straight_scene(N=16, w=16, h=12, tr_estimator=P.TR_WOODCOCK2,
                                                       density=(synth.density_field(16) * 255).astype(np.uint8)),
        "homogeneous_isotropic": scenes.homogeneous_scene(w=16, h=12),
        "curved_rk4_trilinear": scenes.curved_scene(N=16, w=16, h=12),
        "curved_verlet_bspline": scenes.bspline_scene(N=16, w=16, h=12),
        "curved_radial_sphere": scenes.curved_scene(N=16, w=16, h=12, rif="radial", boundary=P.BOUNDARY_SPHERE, sph_radius=0.9),
        "curved_point_emissive": scenes.curved_scene(N=16, w=16, h=12, emission=[0.2, 0.12, 0.06], **pt),
        "straight_point": scenes.straight_scene(N=16, w=16, h=12, **pt),
        # rows added at the end of round 1: signed-distance boundary (+ hdielectric, + aggressivetracing), analytic acoustic RIF
        "sdf_dielectric_verlet": scenes.curved_scene(N=16, w=16, h=12, rif="radial", stepper=P.STEP_VERLET, boundary=P.BOUNDARY_SDF,
                                                     sdf=-synth.sphere_sdf(32, radius=0.9, aabb_min=[-1.2] * 3, aabb_max=[1.2] * 3),
                                                     sdf_aabb=([-1.2] * 3, [1.2] * 3), boundary_bsdf=P.BSDF_HDIELECTRIC),
        "sdf_aggressive_rk4": scenes.curved_scene(N=16, w=16, h=12, rif="radial", boundary=P.BOUNDARY_SDF, aggressive_tracing=True,
                                                  sdf=-synth.sphere_sdf(32, radius=0.9, aabb_min=[-1.2] * 3, aabb_max=[1.2] * 3),
                                                  sdf_aabb=([-1.2] * 3, [1.2] * 3)),
        "acoustic_rk4_m1": scenes.straight_scene(N=16, w=16, h=12, rif_mode=P.RIF_ACOUSTIC, ac_n_o=1.33, ac_n_max=0.08, ac_k_r=4.0, ac_mode=1,
                                                 stepper=P.STEP_RK4, stepsize=0.5 * 2.0 / 15),
    }


def build():
    orc.build()
    out = {}
    for name, p in scene_set().items():
        out["paths/" + name] = np.stack([orc.render_paths(p, s, 7, nthreads=1) for s in (0, 1)])
    # transient film of one scene
    p = scenes.curved_scene(N=16, w=8, h=6, rfilter=P.FILTER_BOX, rfilter_param=0.5, decomposition=P.DECOMPOSITION_TRANSIENT,
                            min_bound=0.0, max_bound=12.0, bin_width=0.5)
    out["transient/curved_film"] = orc.render(p, 0, 4, 7, nthreads=1)[0]
    # leafs
    rng = np.random.RandomState(11)
    pc = scenes.curved_scene(N=16)
    pts = rng.uniform(-1.05, 1.05, (64, 3)).astype(np.float32)
    out["leaf/points"] = pts
    v, idx = orc.lookup_trilinear(pc.density, pc.density_aabb[0], pc.density_aabb[1], pts)
    out["leaf/lookup_value"] = v; out["leaf/lookup_index"] = idx
    inside = rng.uniform(-0.9, 0.9, (32, 3)).astype(np.float32)
    dirs = rng.normal(size=(32, 3)).astype(np.float32); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    out["leaf/ray_o"] = inside; out["leaf/ray_d"] = dirs
    dist = rng.uniform(0.05, 0.6, 32).astype(np.float32)
    out["leaf/trace_dist"] = dist
    tr = orc.er_trace(pc, inside, dirs, dist)
    out["leaf/trace_p"], out["leaf/trace_v"], out["leaf/trace_opt"], out["leaf/trace_ok"] = tr[0], tr[1], tr[3], tr[4]
    out["leaf/sample_distance"] = orc.sample_distance(pc, inside, dirs, np.full(32, np.inf, np.float32), 5)
    out["leaf/connect"] = orc.connect(pc, inside, np.tile(np.array([[0.2, 0.3, -0.1]], np.float32), (32, 1)), 5)[:, :10]
    out["leaf/rng"] = orc.rng_floats(42, 1234, 3, 16)
    return out


if __name__ == "__main__":
    g = build()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_vectors.npz")
    np.savez_compressed(path, **g)
    print("wrote", path, os.path.getsize(path), "bytes;", len(g), "arrays")
