import sys, os, random
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import simpleworks_amd as swm
Q = 0x01AE3A4617C510EAC63B05C06CA1493B1A22D9F300F5138F1EF3622FBA094800170B5D44300000008508C00000000001
ctx = swm.Context(0)
random.seed(5)
def limbs(vals):
    out = np.zeros((len(vals), 6), dtype=np.uint64)
    for i, v in enumerate(vals):
        for k in range(6): out[i, k] = (v >> (64*k)) & ((1<<64)-1)
    return out
def ints(arr): return [sum(int(arr[i,k]) << (64*k) for k in range(6)) for i in range(arr.shape[0])]
n = 20000
A = [random.randrange(2*Q) for _ in range(n)]; B = [random.randrange(2*Q) for _ in range(n)]
edge = [0, 1, Q-1, Q, Q+1, 2*Q-1, (1<<377)-1, (1<<378)-1]
for x in edge:
    for y in edge: A.append(x); B.append(y)
out = ints(ctx.selftest_mul(2, limbs(A), limbs(B)))
inv = pow(1 << 392, -1, Q)
bad = sum(1 for a,b,o in zip(A,B,out) if o != a*b*inv % Q)
print("mul28 mismatches:", bad, "of", len(A))
for which, name in ((0, "Fq 32-bit Comba"), (2, "Fq 28-bit lazy")):
    for threads in (256*256*4, 256*256*16):
        ms = ctx.selftest_mul_throughput(which, threads, 2000)
        print(name, threads, round(ms,2), "ms", round(threads*2000*2/ms/1e6,1), "Gmul/s")
# squarer on lazy operands (a + b limb-wise, each < 2^377 so limbs stay < 2^29)
A2 = [random.randrange(1 << 377) for _ in range(5000)]; B2 = [random.randrange(1 << 377) for _ in range(5000)]
out = ints(ctx.selftest_mul(5, limbs(A2), limbs(B2)))
bad = sum(1 for a,b,o in zip(A2,B2,out) if o != (a+b)*(a+b)*inv % Q)
print("sqr28 mismatches:", bad, "of", len(A2))
