// Micro-benchmark + bit-exactness check of the twisted Edwards mixed addition of msm_accumulate_te (fq28.cuh):
//   r03  packed 144-byte rows, compiler-scheduled multiplier (fq28_mul behind scheduling fences), sign on the accumulator
//   r04  192-byte limb rows, the one-statement asm multiplier (fq28_mul_asm), sign by load address
// Every lane runs a chain of CHAIN additions against rows gathered from a small (L2-resident) table, i.e. the loop of the
// accumulation kernel without its HBM gathers: what is measured is instruction issue.  The final accumulators of both
// variants must be identical limb for limb (same products, same limbs) — printed as "identical: yes".
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I simpleworks_amd/csrc -I include tools/ubench/te28_bench.hip -o tools/ubench/te28_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "fq28.cuh"
using namespace swm;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct alignas(16) RowPacked {  // the r03 row
    Fq ymx, ypx, kt;
};
// r03's addition, verbatim in structure: sign taken on the accumulator side, rows unpacked before each product
template <class M>
__device__ __forceinline__ void madd_r03(T28& a, const RowPacked* __restrict__ rp, bool neg) {
    Fq28 a1, b1;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        const uint32_t d = a.y.l[i] + Fq28Consts::SPREAD4[i] - a.x.l[i], s = a.y.l[i] + a.x.l[i];
        a1.l[i] = neg ? s : d;
        b1.l[i] = neg ? d : s;
    }
    Fq28 A = M::mul(a1, fq28_unpack(rp->ymx));
    Fq28 B = M::mul(b1, fq28_unpack(rp->ypx));
    Fq28 C = M::mul(a.t, fq28_unpack(rp->kt));
    Fq28 E, H, F, G;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        const uint32_t sp = Fq28Consts::SPREAD4[i];
        E.l[i] = neg ? A.l[i] + sp - B.l[i] : B.l[i] + sp - A.l[i];
        H.l[i] = A.l[i] + B.l[i];
        const uint32_t D = a.z.l[i] + a.z.l[i];
        const uint32_t dm = D + sp - C.l[i], dp = D + C.l[i];
        F.l[i] = neg ? dp : dm;
        G.l[i] = neg ? dm : dp;
    }
    a.x = M::mul(E, F);
    a.y = M::mul(G, H);
    a.t = M::mul(E, H);
    a.z = M::mul(F, G);
#pragma unroll
    for (int i = 0; i < 14; i++) {
        asm volatile("" : "+v"(a.x.l[i]));
        asm volatile("" : "+v"(a.y.l[i]));
        asm volatile("" : "+v"(a.t.l[i]));
        asm volatile("" : "+v"(a.z.l[i]));
    }
}

__device__ __forceinline__ void store_acc(G1XYZZ& o, const T28& acc) {
    o.x = fq28_pack(acc.x);
    o.y = fq28_pack(acc.y);
    o.zz = fq28_pack(acc.t);
    o.zzz = fq28_pack(acc.z);
}
template <int WAVES>
__global__ void __launch_bounds__(256, WAVES) k_r03(const RowPacked* __restrict__ rows, const uint32_t* __restrict__ idx, uint32_t chain,
                                                    G1XYZZ* __restrict__ out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t* my = idx + (size_t)t * chain;
    const RowPacked r0 = rows[my[0] & 0x7fffffffu];
    T28 acc = te28_from_row(fq28_unpack(r0.ymx), fq28_unpack(r0.ypx), fq28_unpack(r0.kt), (my[0] >> 31) != 0);
    for (uint32_t k = 1; k < chain; k++) madd_r03<MulFenced>(acc, rows + (my[k] & 0x7fffffffu), (my[k] >> 31) != 0);
    store_acc(out[t], acc);
}
template <int WAVES, class M>
__global__ void __launch_bounds__(256, WAVES) k_r04(const G1TE* __restrict__ rows, const uint32_t* __restrict__ idx, uint32_t chain,
                                                    G1XYZZ* __restrict__ out) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t* my = idx + (size_t)t * chain;
    const G1TE* r0 = rows + (my[0] & 0x7fffffffu);
    T28 acc = te28_from_row(te28_load_coord(r0->ymx), te28_load_coord(r0->ypx), te28_load_coord(r0->kt), (my[0] >> 31) != 0);
    for (uint32_t k = 1; k < chain; k++) te28_madd_row<M>(acc, rows + (my[k] & 0x7fffffffu), (my[k] >> 31) != 0);
    store_acc(out[t], acc);
}
// one product, both multipliers, on random lazy operands (limbs < 2^30): the limbs must agree
__global__ void k_mul_check(const uint32_t* __restrict__ in, uint32_t* __restrict__ bad) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    Fq28 a, b;
#pragma unroll
    for (int i = 0; i < 14; i++) {
        a.l[i] = in[(size_t)t * 28 + i];
        b.l[i] = in[(size_t)t * 28 + 14 + i];
    }
    const Fq28 r0 = fq28_mul(a, b), r1 = fq28_mul_asm(a, b), s0 = fq28_mul(a, a), s1 = fq28_mul_asm(a, a);
    uint32_t x = 0;
#pragma unroll
    for (int i = 0; i < 14; i++) x |= (r0.l[i] ^ r1.l[i]) | (s0.l[i] ^ s1.l[i]);
    if (x) atomicAdd(bad, 1u);
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() {
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 16);
}
// a random value below p as 14 limbs (top limb below p's: 0x1ae3) and packed into twelve words
static void rand_coord(uint32_t limbs[14], uint32_t words[12]) {
    for (int i = 0; i < 13; i++) limbs[i] = rnd() & 0xfffffffu;
    limbs[13] = rnd() % 0x1ae3u;
    memset(words, 0, 48);
    for (int i = 0; i < 14; i++) {
        const int bit = 28 * i, w = bit >> 5, off = bit & 31;
        words[w] |= limbs[i] << off;
        if (off > 4 && w + 1 < 12) words[w + 1] |= limbs[i] >> (32 - off);
    }
}

template <class F>
static int time_kernel(const char* name, F launch, uint32_t lanes, uint32_t chain) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) launch();
    CHECK(hipDeviceSynchronize());
    const int reps = 10;
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("{\"variant\": \"%s\", \"lanes\": %u, \"chain\": %u, \"ms\": %.4f, \"mixed_adds_per_s\": %.4g}\n", name, lanes, chain, ms,
           (double)lanes * (chain - 1) / (ms * 1e-3));
    return 0;
}

int main() {
    const uint32_t nrows = 1u << 14, chain = 48;
    const uint32_t lanes = 256 * 256 * 12;  // twelve workgroups per CU: whole waves of work at 3 and at 4 waves per SIMD
    std::vector<RowPacked> hp(nrows);
    std::vector<G1TE> hu(nrows);
    memset(hu.data(), 0, nrows * sizeof(G1TE));
    for (uint32_t r = 0; r < nrows; r++) {
        rand_coord(hu[r].ymx, hp[r].ymx.v);
        rand_coord(hu[r].ypx, hp[r].ypx.v);
        rand_coord(hu[r].kt, hp[r].kt.v);
    }
    std::vector<uint32_t> hidx((size_t)lanes * chain);
    for (auto& v : hidx) v = (rnd() % nrows) | ((rnd() & 1u) << 31);
    RowPacked* dp;
    G1TE* du;
    uint32_t *didx, *dbad, *dmul;
    G1XYZZ *o3, *o4, *o4f;
    CHECK(hipMalloc(&dp, nrows * sizeof(RowPacked)));
    CHECK(hipMalloc(&du, nrows * sizeof(G1TE)));
    CHECK(hipMalloc(&didx, hidx.size() * 4));
    CHECK(hipMalloc(&o3, (size_t)lanes * sizeof(G1XYZZ)));
    CHECK(hipMalloc(&o4, (size_t)lanes * sizeof(G1XYZZ)));
    CHECK(hipMalloc(&o4f, (size_t)lanes * sizeof(G1XYZZ)));
    CHECK(hipMalloc(&dbad, 4));
    CHECK(hipMemcpy(dp, hp.data(), nrows * sizeof(RowPacked), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(du, hu.data(), nrows * sizeof(G1TE), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(didx, hidx.data(), hidx.size() * 4, hipMemcpyHostToDevice));
    // the multiplier alone on lazy operands
    {
        const uint32_t n = 1u << 16;
        std::vector<uint32_t> hm((size_t)n * 28);
        for (size_t i = 0; i < hm.size(); i++) hm[i] = (i % 14 == 13) ? (rnd() & 0x3ffffu) : (rnd() & 0x3fffffffu);  // limbs < 2^30, value < 128 p
        CHECK(hipMalloc(&dmul, hm.size() * 4));
        CHECK(hipMemcpy(dmul, hm.data(), hm.size() * 4, hipMemcpyHostToDevice));
        CHECK(hipMemset(dbad, 0, 4));
        k_mul_check<<<n / 256, 256>>>(dmul, dbad);
        uint32_t bad = 1;
        CHECK(hipMemcpy(&bad, dbad, 4, hipMemcpyDeviceToHost));
        printf("{\"check\": \"fq28_mul_asm == fq28_mul on %u lazy operand pairs (and squares)\", \"mismatches\": %u}\n", n, bad);
        if (bad) return 2;
    }
    const dim3 grid(lanes / 256), block(256);
    if (time_kernel("r03 packed rows, compiler multiplier, 3 waves", [&] { k_r03<3><<<grid, block>>>(dp, didx, chain, o3); }, lanes, chain)) return 1;
    if (time_kernel("r04 limb rows, compiler multiplier, 3 waves", [&] { k_r04<3, MulFenced><<<grid, block>>>(du, didx, chain, o4f); }, lanes, chain)) return 1;
    if (time_kernel("r04 limb rows, asm multiplier, 3 waves", [&] { k_r04<3, MulAsm><<<grid, block>>>(du, didx, chain, o4); }, lanes, chain)) return 1;
    std::vector<G1XYZZ> h3(lanes), h4(lanes), h4f(lanes);
    CHECK(hipMemcpy(h3.data(), o3, (size_t)lanes * sizeof(G1XYZZ), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(h4.data(), o4, (size_t)lanes * sizeof(G1XYZZ), hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(h4f.data(), o4f, (size_t)lanes * sizeof(G1XYZZ), hipMemcpyDeviceToHost));
    const bool same = memcmp(h3.data(), h4.data(), (size_t)lanes * sizeof(G1XYZZ)) == 0 &&
                      memcmp(h3.data(), h4f.data(), (size_t)lanes * sizeof(G1XYZZ)) == 0;
    if (time_kernel("r04 limb rows, asm multiplier, 4 waves", [&] { k_r04<4, MulAsm><<<grid, block>>>(du, didx, chain, o4); }, lanes, chain)) return 1;
    CHECK(hipMemcpy(h4.data(), o4, (size_t)lanes * sizeof(G1XYZZ), hipMemcpyDeviceToHost));
    const bool same4 = memcmp(h3.data(), h4.data(), (size_t)lanes * sizeof(G1XYZZ)) == 0;
    printf("{\"identical\": \"%s\", \"lanes_compared\": %u}\n", same && same4 ? "yes" : "NO", lanes);
    return same && same4 ? 0 : 3;
}
