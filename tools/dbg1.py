import sys, numpy as np
sys.path.insert(0,'.')
import icicle_amd
from icicle_amd import msm as M, runtime
from oracle import pyref, ref
from tests.util import *
runtime.set_device(0)
for cname in ('bls12_381','bn254'):
    C = pyref.CURVES[cname]; refc = ref.RefCurve(cname)
    rng = np.random.default_rng(19); n=1000
    bases = points_to_array(C, cached_points(C, n)); sc = to_words(rand_scalars(rng, n, C.r), 8)
    base = refc.to_affine(refc.msm(sc, bases))
    for c in (13,14,15,16):
        cfg = icicle_amd.MSMConfig.default(); cfg.c = c
        r = M.msm(cname, sc, bases, cfg)
        print(cname, c, np.array_equal(refc.to_affine(r), base), r[0][:4])
