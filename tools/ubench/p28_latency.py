import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import simpleworks_amd as swm
ctx = swm.Context(0)
for which,name,per in ((2,"mul28",1),(3,"p28_add 1 site",1),(4,"p28_add 4 sites",1)):
    for thr in (65536, 131072, 196608, 262144):
        it = 256 if which==2 else 64
        ms = ctx.selftest_mul_throughput(which, thr, it)
        print(f"{name:18s} threads={thr:7d} waves/SIMD={thr//65536} iters={it} {ms:8.3f} ms  -> {ms*1e3/it:7.2f} us per op  ({thr*it/ms/1e6:8.2f} G op/s)")
