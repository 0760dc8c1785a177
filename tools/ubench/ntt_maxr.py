"""The 2^11 / 2^12-element LDS tiles of the transform (SWM_NTT_MAXR = 11 / 12: 2^22 and 2^24 become two passes) against the
default 2^10-element tiles: same output, time per transform.  usage: python tools/ubench/ntt_maxr.py [log_n ...]"""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "oracle"))
import numpy as np
import simpleworks_amd as swm
from pyref.prng import fr_array
ctx = swm.Context(0)
for lg in [int(a) for a in sys.argv[1:]] or [21, 22, 23, 24]:
    n = 1 << lg
    x = fr_array(min(n, 1 << 20), 5)
    x = np.tile(x, (n // x.shape[0], 1))
    ref = {}
    for maxr in (10, 11, 12):
        os.environ["SWM_NTT_MAXR"] = str(maxr)
        for inv, coset in ((0, 0), (1, 1)):
            d = ctx.to_device(x)
            ctx.ntt_fr_dev(d, lg, inv, coset)
            y = d.download(x.shape, x.dtype)
            if maxr == 10: ref[(inv, coset)] = y
            same = bool(np.array_equal(y, ref[(inv, coset)]))
            for _ in range(3): ctx.ntt_fr_dev(d, lg, inv, coset)
            ctx.synchronize()
            t0 = time.perf_counter()
            reps = 20
            for _ in range(reps): ctx.ntt_fr_dev(d, lg, inv, coset)
            ctx.synchronize()
            dt = (time.perf_counter() - t0) / reps
            print(json.dumps({"log_n": lg, "max_log_r": maxr, "inverse": inv, "coset": coset, "ms": round(dt * 1e3, 4),
                              "same_as_default": same}), flush=True)
            assert same
