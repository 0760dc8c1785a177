import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import simpleworks_amd as swm
ctx = swm.Context(0)
for which, name in ((3, "p28_add 1 site"), (4, "p28_add 4 sites")):
    for threads in (256*256, 256*256*2, 256*256*4, 256*256*12):
        iters = 64
        ms = ctx.selftest_mul_throughput(which, threads, iters)
        waves_per_simd = threads/64/1024
        print(name, "threads", threads, "waves/SIMD %.1f" % waves_per_simd, round(ms,3), "ms", "us per add per wave: %.1f" % (ms*1e3/iters), "Gadd/s %.2f" % (threads*iters/ms/1e6))
