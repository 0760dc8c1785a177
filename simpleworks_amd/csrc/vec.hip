// vec.hip — K4 (batch inversion, pointwise products) over BLS12-377 Fr, plus the device self-tests of the
// field / curve primitives.
//
// (K3, the R1CS mat-vec, lives in spmv.hip.)  K4 replaces ark_ff::batch_inversion and the cfg_iter pointwise loops of ark-marlin's prover rounds.
// All of it is HBM-bound integer work: one lane per row / element, 16-B vector loads, no LDS, no MFMA.
// Algorithmic bytes: vec_mul 96 B per element; batch inverse 64 B per element.
#include <string.h>
#include "context.h"
#include "fq28.cuh"
#include "fr29.cuh"
#include "frinv.cuh"
#include "g1.cuh"

namespace swm {

__global__ void __launch_bounds__(256) vec_mul_kernel(const Fr* __restrict__ a, const Fr* __restrict__ b,
                                                      Fr* __restrict__ out, size_t n) {
    SWM_LIGHT_KERNEL();
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = fp_mul(a[i], b[i]);
}

// Montgomery-trick inversion, two levels (zeros stay zero, like ark_ff::batch_inversion): a lane multiplies up its
// BINV_CHUNK elements, the workgroup combines the 256 lane products with a prefix and a suffix product scan in LDS,
// ONE lane pays the inversion (binary extended Euclid, fr_inv_single) for all 4096 elements, and every lane rebuilds the inverse
// of its own chunk product as total^-1 x prefix x suffix before walking its chunk backwards.
// (One inversion per lane made the inversion 89 % of the kernel's multiplications.)
// (fr_inv_single: frinv.cuh)
// Small inputs (a few workgroups' worth: the proofs of small circuits, where the kernel is a chain of dependent products on an
// idle chip) take 4 elements per lane instead of 16: 4 + 8 products on a lane's chain instead of 16 + 32.
static constexpr int BINV_CHUNK = 16, BINV_CHUNK_SMALL = 4;
static constexpr int BINV_THREADS = 256;
// (the 8 x 32-bit Comba form of this kernel, r01 - r04, was removed in r06; what follows is the one kernel)
// The same kernel on the transform's multiplier (r05): nine 29-bit lazy limbs, 197 instructions per product instead of ~330.
// fr29_mul(a, b) = a b 2^-261, five bits more than the memory format's 2^-256 — and NO correction is needed: call lambda = 2^-5;
// a product of k elements built by any tree of these multiplications carries lambda^(k-1), the exact inverse of such a product
// (fr_inv_single, which inverts the Montgomery form it is given) carries lambda^-(k-1), and multiplying the inverse of a k-group
// by a j-subgroup of it leaves the inverse of the (k-j)-group with lambda^-(k-j-1): every inverse of ONE element comes out with
// lambda^0.  The neutral 1 counts as an element of its group; skipped zeros simply are not in any.  Values stay < 2r
// (normalised limbs) between products and are stored as 8 words; inputs are canonical, outputs are made canonical.
template <int BINV_CHUNK>
__global__ void __launch_bounds__(BINV_THREADS) batch_inverse_kernel29(Fr* __restrict__ v, size_t n) {
    SWM_LIGHT_KERNEL();
    __shared__ Fr sp[BINV_THREADS], ss[BINV_THREADS];
    __shared__ Fr s_inv;
    const unsigned tid = threadIdx.x;
    size_t t = blockIdx.x * (size_t)blockDim.x + tid;
    size_t lo = t * BINV_CHUNK;
    size_t hi = lo + BINV_CHUNK < n ? lo + BINV_CHUNK : n;
    Fr pref[BINV_CHUNK];
    Fr29 acc = fr29_unpack(fp_one<Fr>());
#pragma unroll
    for (int i = 0; i < BINV_CHUNK; i++) {
        pref[i] = fr29_pack(acc);
        if (lo + i < hi) {
            Fr x = v[lo + i];
            if (!fp_is_zero(x)) acc = fr29_mul_fenced(acc, fr29_unpack(x));
        }
    }
    const Fr accp = fr29_pack(acc);
    sp[tid] = accp;
    ss[tid] = accp;
    __syncthreads();
    for (unsigned d = 1; d < BINV_THREADS; d <<= 1) {  // inclusive prefix products in sp, inclusive suffix products in ss
        const bool hp = tid >= d, hs = tid + d < BINV_THREADS;
        Fr a = hp ? sp[tid - d] : accp, b = hs ? ss[tid + d] : accp;
        Fr mp = sp[tid], ms = ss[tid];
        __syncthreads();
        if (hp) sp[tid] = fr29_pack(fr29_mul_fenced(fr29_unpack(mp), fr29_unpack(a)));
        if (hs) ss[tid] = fr29_pack(fr29_mul_fenced(fr29_unpack(ms), fr29_unpack(b)));
        __syncthreads();
    }
    if (tid == 0) s_inv = fr_inv_single(fr29_pack(fr29_canonical(fr29_unpack(ss[0]), true)));  // (never zero: zeros are skipped)
    __syncthreads();
    Fr29 inv = fr29_unpack(s_inv);
    if (tid > 0) inv = fr29_mul_fenced(inv, fr29_unpack(sp[tid - 1]));
    if (tid + 1 < BINV_THREADS) inv = fr29_mul_fenced(inv, fr29_unpack(ss[tid + 1]));
#pragma unroll
    for (int i = BINV_CHUNK - 1; i >= 0; i--) {
        if (lo + i < hi) {
            Fr x = v[lo + i];
            if (!fp_is_zero(x)) {
                v[lo + i] = fr29_pack(fr29_canonical(fr29_mul_fenced(inv, fr29_unpack(pref[i])), true));
                inv = fr29_mul_fenced(inv, fr29_unpack(x));
            }
        }
    }
}

int vec_mul_run(swm_ctx* ctx, const void* a, const void* b, void* out, size_t n) {
    if (n == 0) return SWM_OK;
    unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, 256 * 32);
    SWM_LAUNCH(ctx, "vec_mul", vec_mul_kernel, dim3(grid), dim3(256), 0, (const Fr*)a, (const Fr*)b, (Fr*)out, n);
    return SWM_OK;
}
int batch_inverse_run(swm_ctx* ctx, void* d, size_t n) {
    if (n == 0) return SWM_OK;
    // (r05: up to 2^20 elements — the inversions of proofs up to 2^18 constraints: a lane's chain is 29 instead of 65 products beside the
    // one inversion every workgroup waits for; 2^16 proofs 7.1 -> 6.9 ms, 2^14 4.15 -> 4.05, 2^18 unchanged; r04: 65 536)
    // (the kernel on the transform's 29-bit multiplier; the 8 x 32-bit Comba kernel of r01 - r04 is gone: r06)
    // r06: measured alone over n (tools/ubench/binv_time.py): 4-element chunks 109 / 142 / 248 us at n = 2^18 / 2^19 / 2^20, 16-element
    // chunks 129 / 132 / 136 us — the bound is 2^18 (r05 had 2^20: the inversions of a 2^18-constraint proof took 248 instead of 136 us)
    static constexpr size_t small_below = 262144;
    if (n <= small_below) {
        const size_t threads = (n + BINV_CHUNK_SMALL - 1) / BINV_CHUNK_SMALL;
        const dim3 grid((unsigned)((threads + BINV_THREADS - 1) / BINV_THREADS));
        SWM_LAUNCH(ctx, "batch_inverse", batch_inverse_kernel29<BINV_CHUNK_SMALL>, grid, dim3(BINV_THREADS), 0, (Fr*)d, n);
        return SWM_OK;
    }
    const size_t threads = (n + BINV_CHUNK - 1) / BINV_CHUNK;
    const dim3 grid((unsigned)((threads + BINV_THREADS - 1) / BINV_THREADS));
    SWM_LAUNCH(ctx, "batch_inverse", batch_inverse_kernel29<BINV_CHUNK>, grid, dim3(BINV_THREADS), 0, (Fr*)d, n);
    return SWM_OK;
}

// ---------------------------------------------------------------------------------------------- self-tests
template <class F>
__global__ void __launch_bounds__(256) selftest_mul_kernel(const F* a, const F* b, F* out, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) out[i] = fp_mul(a[i], b[i]);
}
template <class F>
__global__ void __launch_bounds__(256) selftest_mul_chain(F* out, int iters) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    F a = fp_one<F>(), b = fp_one<F>();
    a.v[0] ^= (uint32_t)i;
    b.v[1] ^= (uint32_t)(i * 2654435761u);
    for (int k = 0; k < iters; k++) {
        a = fp_mul(a, b);
        b = fp_mul(b, a);
    }
    out[i] = fp_add(a, b);
}
// 28-bit-limb multiplier: out = canonical(a * b * 2^-392 mod p), operands given as packed 384-bit integers
__global__ void __launch_bounds__(256) selftest_mul28_kernel(const Fq* a, const Fq* b, Fq* out, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) out[i] = fq28_pack(fq28_canonical(fq28_mul(fq28_unpack(a[i]), fq28_unpack(b[i]))));
}
// dedicated squarer on a lazy operand: out = canonical((a + b)^2 * 2^-392) with a + b formed limb-wise (no carry)
__global__ void __launch_bounds__(256) selftest_sqr28_kernel(const Fq* a, const Fq* b, Fq* out, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i < n) out[i] = fq28_pack(fq28_canonical(fq28_sqr(fq28_add(fq28_unpack(a[i]), fq28_unpack(b[i])))));
}
// fused two-product multiplier at its operand bounds: out = canonical((x y + c d) 2^-392) with
// x = a + 8p-spread, y = b + 32p-spread (limbs < 1.5 * 2^29), c = 8p-spread - a (limbs < 2^29), d = b
__global__ void __launch_bounds__(256) selftest_mul2_28_kernel(const Fq* a, const Fq* b, Fq* out, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fq28 ua = fq28_unpack(a[i]), ub = fq28_unpack(b[i]), x, y, c;
#pragma unroll
    for (int k = 0; k < 14; k++) {
        x.l[k] = ua.l[k] + Fq28Consts::SPREAD8[k];
        y.l[k] = ub.l[k] + Fq28Consts::SPREAD32[k];
        c.l[k] = Fq28Consts::SPREAD8[k] - ua.l[k];
    }
    out[i] = fq28_pack(fq28_canonical(fq28_mul2(x, y, c, ub)));
}
__global__ void __launch_bounds__(256) selftest_mul28_chain(Fq* out, int iters) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    Fq28 a = fq28_const(Fq28Consts::ONE), b = fq28_const(Fq28Consts::TO384);
    a.l[0] ^= (uint32_t)i & 0xffff;
    b.l[1] ^= (uint32_t)(i * 2654435761u) & 0xffff;
    for (int k = 0; k < iters; k++) {
        a = fq28_mul(a, b);
        b = fq28_mul(b, a);
    }
    out[i] = fq28_pack(fq28_canonical(fq28_mul(a, b)));
}
// throughput probe of the general 28-bit XYZZ addition: one call site in a loop (which = 3) or four unrolled call
// sites (which = 4) — same arithmetic, different code footprint
template <int SITES>
__global__ void __launch_bounds__(256) selftest_p28_add_chain(Fq* out, int iters) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    P28 a, b;
    a.x = fq28_const(Fq28Consts::ONE);
    a.x.l[0] ^= (uint32_t)i & 0xffff;
    a.y = fq28_const(Fq28Consts::TO384);
    a.zz = fq28_const(Fq28Consts::ONE);
    a.zzz = a.zz;
    b = a;
    b.x.l[1] ^= 0x1234;
    b.y.l[2] ^= (uint32_t)(i * 2654435761u) & 0xfff;
    for (int k = 0; k < iters; k++) {
        if (SITES == 1) {
            p28_add_fast<MulInline>(a, b);
        } else {
            p28_add_fast<MulInline>(a, b);
            p28_add_fast<MulInline>(b, a);
            p28_add_fast<MulInline>(a, b);
            p28_add_fast<MulInline>(b, a);
        }
    }
    out[i] = fq28_pack(fq28_canonical(fq28_mul(a.x, b.y)));
}
__global__ void __launch_bounds__(256) selftest_g1_add_kernel(const G1Affine* a, const G1Affine* b, G1Jac* out,
                                                              size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    // exercise both adders: mixed (acc from a, += b) and full XYZZ add of (a + b) + (a + b), then - (a + b)
    G1XYZZ acc = g1_from_affine(a[i]);
    g1_add_mixed(acc, b[i]);
    G1XYZZ twice = acc;
    g1_add(twice, acc);           // doubling branch of the full adder
    g1_add(twice, g1_neg(acc));   // back to a + b through the generic branch
    out[i] = g1_to_jacobian(twice);
}

int selftest_mul_run(swm_ctx* ctx, int which, const void* a, const void* b, void* out, size_t n) {
    unsigned grid = (unsigned)((n + 255) / 256);
    if (which == 6)
        SWM_LAUNCH(ctx, "selftest_mul2_28", selftest_mul2_28_kernel, dim3(grid), dim3(256), 0, (const Fq*)a, (const Fq*)b,
                   (Fq*)out, n);
    else if (which == 5)
        SWM_LAUNCH(ctx, "selftest_sqr28", selftest_sqr28_kernel, dim3(grid), dim3(256), 0, (const Fq*)a, (const Fq*)b,
                   (Fq*)out, n);
    else if (which == 2)
        SWM_LAUNCH(ctx, "selftest_mul28", selftest_mul28_kernel, dim3(grid), dim3(256), 0, (const Fq*)a, (const Fq*)b,
                   (Fq*)out, n);
    else if (which == 0)
        SWM_LAUNCH(ctx, "selftest_mul_fq", selftest_mul_kernel<Fq>, dim3(grid), dim3(256), 0, (const Fq*)a,
                   (const Fq*)b, (Fq*)out, n);
    else
        SWM_LAUNCH(ctx, "selftest_mul_fr", selftest_mul_kernel<Fr>, dim3(grid), dim3(256), 0, (const Fr*)a,
                   (const Fr*)b, (Fr*)out, n);
    return SWM_OK;
}
int selftest_chain_run(swm_ctx* ctx, int which, void* out, size_t threads, int iters) {
    unsigned grid = (unsigned)(threads / 256);
    if (which == 3)
        SWM_LAUNCH(ctx, "selftest_p28_1site", selftest_p28_add_chain<1>, dim3(grid), dim3(256), 0, (Fq*)out, iters);
    else if (which == 4)
        SWM_LAUNCH(ctx, "selftest_p28_4site", selftest_p28_add_chain<4>, dim3(grid), dim3(256), 0, (Fq*)out, iters / 4);
    else if (which == 2)
        SWM_LAUNCH(ctx, "selftest_chain28", selftest_mul28_chain, dim3(grid), dim3(256), 0, (Fq*)out, iters);
    else if (which == 0)
        SWM_LAUNCH(ctx, "selftest_chain_fq", selftest_mul_chain<Fq>, dim3(grid), dim3(256), 0, (Fq*)out, iters);
    else
        SWM_LAUNCH(ctx, "selftest_chain_fr", selftest_mul_chain<Fr>, dim3(grid), dim3(256), 0, (Fr*)out, iters);
    return SWM_OK;
}
int selftest_g1_add_run(swm_ctx* ctx, const void* a, const void* b, void* out, size_t n) {
    SWM_LAUNCH(ctx, "selftest_g1_add", selftest_g1_add_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
               (const G1Affine*)a, (const G1Affine*)b, (G1Jac*)out, n);
    return SWM_OK;
}

}  // namespace swm

// ---------------------------------------------------------------------------------------------- host self-test of the single-element inversion
extern "C" int swm_selftest_fr_inv(const uint64_t* a_mont, uint64_t* out_mont, size_t n, unsigned* fallbacks) {
    using namespace swm;
    if ((n && (!a_mont || !out_mont)) || !fallbacks) return SWM_ERR_INVALID_ARG;
    *fallbacks = 0;
    for (size_t i = 0; i < n; i++) {
        Fr a;
        memcpy(a.v, a_mont + 4 * i, 32);
        if (fp_is_zero(a)) return SWM_ERR_INVALID_ARG;
        bool ok = false;
        const Fr r = fr_inv_bingcd(a, &ok);
        if (!ok) (*fallbacks)++;
        memcpy(out_mont + 4 * i, r.v, 32);
    }
    return SWM_OK;
}
