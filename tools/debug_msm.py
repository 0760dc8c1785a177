import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "oracle"))
import numpy as np
from oracle_lib import Oracle, golden, h2i, ints_to_limbs
import simpleworks_amd as swm
orc = Oracle(); ctx = swm.Context(0)
g = golden("msm.json")
pt = lambda p: None if p is None else (h2i(p[0]), h2i(p[1]))
srs = [pt(p) for p in g["srs_bases"]]
bm = orc.points_to_mont(srs)
bh = ctx.srs_upload(bm)
def aff(j):
    xy, inf = ctx.g1_normalize(j)
    return None if inf else orc.points_from_mont(xy.reshape(1,12))[0]
def check(name, scal):
    sc = ints_to_limbs(scal, 4)
    ref = orc.jac_to_affine_int(orc.msm(np.ascontiguousarray(bm[:len(scal)]), sc))
    got = aff(ctx.msm_g1(bh, sc))
    print(name, "OK" if got == ref else "FAIL")

c=[x for x in g["cases"] if x["name"]=="uniform_2"][0]
sc=[h2i(x) for x in c["scalars"]]
check("u2 both", sc)
check("u2 first only", [sc[0], 0])
check("u2 second only", [0, sc[1]])
for bits in (32, 64, 96, 128, 160, 192, 224, 250):
    m=(1<<bits)-1
    check("u2 low %d bits"%bits, [sc[0]&m, sc[1]&m])
for w in range(43):
    m=((1<<6)-1)<<(6*w)
    check("u2 window %d"%w, [sc[0]&m, sc[1]&m])
print(hex(sc[0]), hex(sc[1]))
