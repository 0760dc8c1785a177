import sys, time, ctypes
sys.path.insert(0, '.')
from simpleworks_amd import marlin as M
from simpleworks_amd._lib import load_library
a = M.generate_rand(); lib = load_library(); n = 256 << 20; buf = (ctypes.c_uint8 * n)()
for _ in range(3):
    t = time.time(); lib.swm_rng_fill_bytes(a.h, buf, n); dt = time.time() - t
    print("fill 256MB (malloc'd): %.3fs -> %.2f GB/s" % (dt, n / dt / 1e9))
print(open('/proc/cpuinfo').read().count('avx2'), [l for l in open('/proc/cpuinfo') if 'MHz' in l][:2])
