import sys, os; sys.path.insert(0,'.')
import numpy as np
pix = 11*48+18
os.environ["MER_DEBUG_PIXEL"]=str(pix); os.environ["ORC_DEBUG_PIXEL"]=str(pix)
from mitsubaer_amd import capi, params as P
from oracle import orc
from tests.test_gpu_render import CASES
ctx=capi.Context(0)
p=CASES["cfg2_straight_woodcock2"]()
sc,vols=ctx.upload_scene(p)
a=ctx.render_paths(sc,0,seed=3)
sys.stdout.flush()
b=orc.render_paths(p,0,3,nthreads=1)
print(a[11,18],b[11,18])
