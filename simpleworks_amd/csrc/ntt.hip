// ntt.hip — K2: radix-2 NTT / iNTT over BLS12-377 Fr for gfx950 (MI355X).
//
// Replaces ark_poly::Radix2EvaluationDomain::{fft,ifft,coset_fft,coset_ifft}_in_place (ark-poly 0.3.0,
// SURVEY.md A.3), reached from /root/reference/src/marlin/mod.rs:75,92.  Semantics kept: natural order in and
// out, root = TWO_ADIC_ROOT^(2^(47-log n)), inverse scales by 1/n, coset variants scale coefficient i by 22^i
// (before a forward transform) or 22^-i (after an inverse one).
//
// MI355X design: Stockham autosort in ceil(log n / 10) passes.  A workgroup stages a tile of R = 2^r strided
// rows x 8 contiguous columns (256-B coalesced runs of 32-B elements, 64 KB of LDS at r = 8) and runs all r
// butterfly levels of that tile in LDS; the inter-pass twiddle w_n^(k t), the coset scaling and the 1/n factor
// are fused into the tile load / store, so each pass reads and writes every element exactly once
// (2 x 32 B x n HBM bytes per pass; algorithmic bytes 64 B per element per transform).
// Twiddles are not streamed from HBM: w^e is rebuilt from two small L2-resident tables (w^(e mod 1024),
// w^(1024 (e div 1024))) with one extra multiply.
#include <atomic>
#include "context.h"
#include "ff.cuh"
#include "fr29.cuh"
#include <string.h>
#include <algorithm>
#include <vector>

namespace swm {

// contiguous columns per tile: chosen per pass so that a workgroup holds ~1024 elements (32 KB of LDS: four to five
// workgroups per CU, every lane busy in every butterfly level) — 512 when the transform is too small to fill the chip
static constexpr int NTT_THREADS = 256;
static constexpr unsigned NTT_MAX_LOG_R = 10;  // 2^20 in two passes (10 + 10); 11 + 11 for 2^22 loses to 8 + 7 + 7 (r01 sweep)

struct NttPassArgs {
    const Fr* src;
    Fr* dst;
    unsigned log_n, log_r, log_ns;
    const Fr* tw_small;   // w_R^e, e < R/2
    const Fr* tw_lo;      // w_n^i, i < 1024
    const Fr* tw_hi;      // w_n^(1024 i)
    const Fr* cs_lo;      // g^i (or g^-i)
    const Fr* cs_hi;
    uint64_t src_len;     // elements of src that exist: the rest of the 2^log_n inputs are zero (first pass of a zero-extended transform)
    const Fr* pass_tw;    // lazy kernel, passes >= 2: the inter-pass twiddles of THIS pass as a table, [t * Ns + k] = w_n^((k t) << shift)
    int coset_in;         // multiply input i by g^i while loading (first pass of a forward coset transform)
    int scale_out;        // multiply output by n_inv (last pass of an inverse transform)
    int coset_out;        // ... and by g^-i
    Fr n_inv;
};

__device__ __forceinline__ Fr two_level_pow(const Fr* lo, const Fr* hi, uint64_t e) {
    Fr a = lo[e & 1023];
    uint64_t h = e >> 10;
    if (h) a = fp_mul(a, hi[h]);
    return a;
}

__device__ __forceinline__ unsigned bitrev_u(unsigned x, unsigned bits) { return bits ? (__brev(x) >> (32 - bits)) : 0u; }

template <int J>
__global__ void __launch_bounds__(NTT_THREADS) ntt_pass(NttPassArgs a) {
    SWM_LIGHT_KERNEL();
    extern __shared__ __align__(16) unsigned char smem_raw[];
    Fr* tile = reinterpret_cast<Fr*>(smem_raw);  // [R][J]
    const unsigned R = 1u << a.log_r;
    const uint64_t n = 1ull << a.log_n;
    const uint64_t stride = n >> a.log_r;  // n / R
    const uint64_t j0 = (uint64_t)blockIdx.x * J;
    const uint64_t ns_mask = (1ull << a.log_ns) - 1;
    const unsigned tw_shift = a.log_n - a.log_ns - a.log_r;  // w_{Ns R} = w_n^(2^tw_shift)
    // ---- load (+ inter-pass twiddle, + coset scaling).  A tile has at most 1024 elements (ntt_run): four per lane, all
    // four global loads issued before the first twiddle multiplication so that their latencies overlap.
    for (unsigned e0 = 0; e0 < R * J; e0 += 4 * NTT_THREADS) {
        Fr xs[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            unsigned e = e0 + threadIdx.x + u * NTT_THREADS;
            if (e < R * J) {
                const uint64_t at = j0 + e % J + (uint64_t)(e / J) * stride;
                xs[u] = at < a.src_len ? a.src[at] : fp_zero<Fr>();
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            unsigned e = e0 + threadIdx.x + u * NTT_THREADS;
            if (e >= R * J) break;
            unsigned jj = e % J, t = e / J;
            uint64_t idx = j0 + jj + (uint64_t)t * stride;
            Fr x = xs[u];
            if (a.coset_in) x = fp_mul(x, two_level_pow(a.cs_lo, a.cs_hi, idx));
            if (a.log_ns != 0 && t != 0) {
                uint64_t k = (j0 + jj) & ns_mask;
                if (k) x = fp_mul(x, two_level_pow(a.tw_lo, a.tw_hi, (k * t) << tw_shift));
            }
            tile[t * J + jj] = x;
        }
    }
    __syncthreads();
    // ---- r radix-2 DIF levels in LDS (natural in, bit-reversed rows out), TWO levels per LDS round trip: a lane takes
    // the four elements of a radix-4 group (i0, i0 + h/2, i0 + h, i0 + 3h/2), runs the level-h butterflies and then the
    // level-h/2 butterflies on them in registers (same four multiplications as two radix-2 levels — in a prime field the
    // fourth root of unity is not free — but half the LDS traffic and half the barriers).  An odd r starts with one
    // plain radix-2 level.
    unsigned h = R >> 1;
    if (a.log_r & 1) {
        for (unsigned bq = threadIdx.x; bq < (R >> 1) * J; bq += NTT_THREADS) {
            unsigned jj = bq % J, pos = bq / J;  // h = R/2: one block, pos = q
            unsigned i0 = pos, i1 = i0 + h;
            Fr u = tile[i0 * J + jj], v = tile[i1 * J + jj];
            tile[i0 * J + jj] = fp_add(u, v);
            Fr d = fp_sub(u, v);
            if (pos) d = fp_mul(d, a.tw_small[pos]);
            tile[i1 * J + jj] = d;
        }
        h >>= 1;
        __syncthreads();
    }
    for (; h >= 2; h >>= 2) {
        const unsigned hh = h >> 1, s1 = (R >> 1) / h, s2 = 2 * s1;
        for (unsigned q = threadIdx.x; q < (R >> 2) * J; q += NTT_THREADS) {
            unsigned jj = q % J, qq = q / J;
            unsigned p = qq & (hh - 1), blk = qq / hh;
            unsigned i0 = blk * 2 * h + p, i1 = i0 + hh, i2 = i0 + h, i3 = i2 + hh;
            Fr x0 = tile[i0 * J + jj], x1 = tile[i1 * J + jj], x2 = tile[i2 * J + jj], x3 = tile[i3 * J + jj];
            Fr a0 = fp_add(x0, x2), a2 = fp_sub(x0, x2);
            if (p) a2 = fp_mul(a2, a.tw_small[p * s1]);
            Fr a1 = fp_add(x1, x3), a3 = fp_mul(fp_sub(x1, x3), a.tw_small[(p + hh) * s1]);
            Fr y1 = fp_sub(a0, a1), y3 = fp_sub(a2, a3);
            if (p) {
                Fr w = a.tw_small[p * s2];
                y1 = fp_mul(y1, w);
                y3 = fp_mul(y3, w);
            }
            tile[i0 * J + jj] = fp_add(a0, a1);
            tile[i1 * J + jj] = y1;
            tile[i2 * J + jj] = fp_add(a2, a3);
            tile[i3 * J + jj] = y3;
        }
        __syncthreads();
    }
    // ---- store (Stockham index map; + 1/n, + coset unscaling)
    for (unsigned e = threadIdx.x; e < R * J; e += NTT_THREADS) {
        unsigned jj, u;
        if (a.log_ns == 0) {  // out[(j0+jj) R + u]: runs of R contiguous elements
            u = e % R;
            jj = e / R;
        } else {              // out[(j/Ns) Ns R + k + u Ns]: runs of J contiguous elements
            jj = e % J;
            u = e / J;
        }
        uint64_t j = j0 + jj;
        uint64_t k = j & ns_mask;
        uint64_t o = ((j - k) << a.log_r) + k + ((uint64_t)u << a.log_ns);
        Fr x = tile[bitrev_u(u, a.log_r) * J + jj];
        if (a.scale_out) {
            Fr s = a.n_inv;
            if (a.coset_out) s = fp_mul(s, two_level_pow(a.cs_lo, a.cs_hi, o));
            x = fp_mul(x, s);
        }
        a.dst[o] = x;
    }
}

// ---------------------------------------------------------------------------------------------- the same pass in lazy 29-bit limbs
// ntt_pass with the arithmetic of fr29.cuh: same tiles, same index maps, same pass plan.  Elements sit in LDS as 9 limbs
// (36 B: an odd word stride, conflict-free), twiddles come from tables in Montgomery form of radix 2^261 (so the data keep
// the factor they came with: no conversion), butterflies add and subtract without carries or comparisons, and every value is
// multiplied where the schedule says so — by its twiddle, or by one — so that the bounds below hold for every lane alike:
//   after the load      every element is a product (inter-pass twiddle, w^0 included; coset factor) or canonical input: < 2r
//   radix-2 level       u' = u + v (normalised on the store), d = (u - v + 2B r) w
//   radix-4 step, values < B r (B <= 64):   a0 = x0 + x2, a1 = x1 + x3 lazy;  a2 = (x0 - x2 + 2B r) w_a, a3 = (x1 - x3 + 2B r) w_b;
//                       y0 = a0 + a1 (< 4B r; multiplied by one when 4B > 64 or in the last step of a pass), y1 = (a0 - a1 + 4B r) w_c,
//                       y2 = a2 + a3 (< 4r), y3 = (a2 - a3 + 4r) w_c;  everything stored normalised
//   after the last step every value is < 4r: it fits the 256-bit memory format between passes; the last pass makes it canonical.
// The spreads (limbs of 2B r and 4B r with borrows) and the reduction flags are the host's plan of the pass (LazyPlan);
// tools/check_ntt29.py emulates plan + arithmetic bit by bit and compares whole transforms with a direct DFT.
struct LazyPlan {
    Spread29 r2;        // radix-2 level: 2 B0 r, one borrow
    Spread29 s1[6];     // per radix-4 step: 2 B r, one borrow (six steps: tiles of up to 2^12 elements, SWM_NTT_MAXR)
    Spread29 s2[6];     //                   4 B r, two borrows (the subtrahend is a lazy sum of two)
    Spread29 s4;        //                   4 r, one borrow (difference of two products)
    uint32_t reduce;    // bit s: y0 of step s is multiplied by one
    uint32_t out_below_2r;  // every value leaving the tile is < 2r (else < 4r)
};
struct NttLazyArgs {
    NttPassArgs a;      // tw_small / tw_lo / tw_hi / cs_lo / cs_hi / n_inv: the radix-2^261 forms
    LazyPlan plan;
    int last_pass;      // the output is the transform's result: canonical
};
__device__ __forceinline__ Fr29 two_level_pow29(const Fr* lo, const Fr* hi, uint64_t e) {
    Fr29 a = fr29_unpack(lo[e & 1023]);
    uint64_t h = e >> 10;
    if (h) a = fr29_mul_fenced(a, fr29_unpack(hi[h]));
    return a;
}
// LDS slot of tile element idx: 9 words per element (an odd stride) and one slot of padding per 32 elements — the late
// butterfly levels and the bit-reversed reads of the store touch elements 16, 32, ... apart, which without the padding fall on
// two banks (measured: the first pass of a 2^22 transform was slower than with the 32-bit-limb kernel)
__device__ __forceinline__ unsigned tile_slot(unsigned idx) { return (idx + (idx >> 5)) * 9; }
__device__ __forceinline__ void tile_put(uint32_t* tile, unsigned idx, const Fr29& x) {
    const unsigned o = tile_slot(idx);
#pragma unroll
    for (int i = 0; i < 9; i++) tile[o + i] = x.l[i];
}
__device__ __forceinline__ Fr29 tile_get(const uint32_t* tile, unsigned idx) {
    const unsigned o = tile_slot(idx);
    Fr29 x;
#pragma unroll
    for (int i = 0; i < 9; i++) x.l[i] = tile[o + i];
    return x;
}
template <int J>
__global__ void __launch_bounds__(NTT_THREADS, 4) ntt_pass_lazy(NttLazyArgs args) {
    SWM_LIGHT_KERNEL();
    extern __shared__ __align__(16) unsigned char smem_raw[];
    uint32_t* tile = reinterpret_cast<uint32_t*>(smem_raw);  // [R][J] elements of 9 words
    const NttPassArgs& a = args.a;
    const LazyPlan& P = args.plan;
    const unsigned R = 1u << a.log_r;
    const uint64_t n = 1ull << a.log_n;
    const uint64_t stride = n >> a.log_r;
    const uint64_t j0 = (uint64_t)blockIdx.x * J;
    const uint64_t ns_mask = (1ull << a.log_ns) - 1;
    const unsigned tw_shift = a.log_n - a.log_ns - a.log_r;
    // ---- load: unpack, coset factor, inter-pass twiddle (w^0 = one included: every loaded value of a later pass is a product)
    for (unsigned e0 = 0; e0 < R * J; e0 += 4 * NTT_THREADS) {
        Fr xs[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            unsigned e = e0 + threadIdx.x + u * NTT_THREADS;
            if (e < R * J) {
                const uint64_t at = j0 + e % J + (uint64_t)(e / J) * stride;
                xs[u] = at < a.src_len ? a.src[at] : fp_zero<Fr>();
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            unsigned e = e0 + threadIdx.x + u * NTT_THREADS;
            if (e >= R * J) break;
            unsigned jj = e % J, t = e / J;
            uint64_t idx = j0 + jj + (uint64_t)t * stride;
            Fr29 x = fr29_unpack(xs[u]);
            if (a.coset_in) x = fr29_mul_fenced(x, two_level_pow29(a.cs_lo, a.cs_hi, idx));
            if (a.log_ns != 0) {
                uint64_t k = (j0 + jj) & ns_mask;
                // per-pass table (one multiplication less per element; 32 B more traffic on a kernel at 5 % of HBM), else the
                // two-level product
                const Fr29 tw = a.pass_tw ? fr29_unpack(a.pass_tw[((uint64_t)t << a.log_ns) + k])
                                          : two_level_pow29(a.tw_lo, a.tw_hi, (k * t) << tw_shift);
                x = fr29_mul_fenced(x, tw);
            }
            tile_put(tile, t * J + jj, x);
        }
    }
    __syncthreads();
    unsigned h = R >> 1;
    if (a.log_r & 1) {
        for (unsigned bq = threadIdx.x; bq < (R >> 1) * J; bq += NTT_THREADS) {
            unsigned jj = bq % J, pos = bq / J;
            unsigned i0 = pos, i1 = i0 + h;
            Fr29 u = tile_get(tile, i0 * J + jj), v = tile_get(tile, i1 * J + jj);
            tile_put(tile, i0 * J + jj, fr29_normalize(fr29_add(u, v)));
            tile_put(tile, i1 * J + jj, fr29_mul_fenced(fr29_sub(u, v, P.r2), fr29_unpack(a.tw_small[pos])));
        }
        h >>= 1;
        __syncthreads();
    }
    unsigned step = 0;
    for (; h >= 2; h >>= 2, step++) {
        const unsigned hh = h >> 1, s1 = (R >> 1) / h, s2 = 2 * s1;
        const bool red = (P.reduce >> step) & 1u;
        const Spread29& S1 = P.s1[step];
        const Spread29& S2 = P.s2[step];
        for (unsigned q = threadIdx.x; q < (R >> 2) * J; q += NTT_THREADS) {
            unsigned jj = q % J, qq = q / J;
            unsigned p = qq & (hh - 1), blk = qq / hh;
            unsigned i0 = blk * 2 * h + p, i1 = i0 + hh, i2 = i0 + h, i3 = i2 + hh;
            Fr29 a0, a1, a2, a3;
            {
                Fr29 x0 = tile_get(tile, i0 * J + jj), x2 = tile_get(tile, i2 * J + jj);
                a0 = fr29_add(x0, x2);
                a2 = fr29_mul_fenced(fr29_sub(x0, x2, S1), fr29_unpack(a.tw_small[p * s1]));
            }
            {
                Fr29 x1 = tile_get(tile, i1 * J + jj), x3 = tile_get(tile, i3 * J + jj);
                a1 = fr29_add(x1, x3);
                a3 = fr29_mul_fenced(fr29_sub(x1, x3, S1), fr29_unpack(a.tw_small[(p + hh) * s1]));
            }
            const Fr29 w = fr29_unpack(a.tw_small[p * s2]);
            Fr29 y0 = fr29_add(a0, a1);
            y0 = red ? fr29_mul_fenced(y0, fr29_const(Fr29Consts::ONE)) : fr29_normalize(y0);
            tile_put(tile, i0 * J + jj, y0);
            tile_put(tile, i1 * J + jj, fr29_mul_fenced(fr29_sub(a0, a1, S2), w));
            tile_put(tile, i2 * J + jj, fr29_normalize(fr29_add(a2, a3)));
            tile_put(tile, i3 * J + jj, fr29_mul_fenced(fr29_sub(a2, a3, P.s4), w));
        }
        __syncthreads();
    }
    // ---- store (Stockham index map; 1/n and coset unscaling on the last pass of an inverse transform; canonical on any last pass)
    for (unsigned e = threadIdx.x; e < R * J; e += NTT_THREADS) {
        unsigned jj, u;
        if (a.log_ns == 0) {
            u = e % R;
            jj = e / R;
        } else {
            jj = e % J;
            u = e / J;
        }
        uint64_t j = j0 + jj;
        uint64_t k = j & ns_mask;
        uint64_t o = ((j - k) << a.log_r) + k + ((uint64_t)u << a.log_ns);
        Fr29 x = tile_get(tile, bitrev_u(u, a.log_r) * J + jj);
        if (a.scale_out) {
            Fr29 sc = fr29_unpack(a.n_inv);
            if (a.coset_out) sc = fr29_mul_fenced(two_level_pow29(a.cs_lo, a.cs_hi, o), sc);
            x = fr29_canonical(fr29_mul_fenced(x, sc), true);
        } else if (args.last_pass) {
            x = fr29_canonical(x, P.out_below_2r != 0);
        }
        a.dst[o] = fr29_pack(x);
    }
}

// ---------------------------------------------------------------------------------------------- tables (host built)
static Fr host_root_of_unity(unsigned log_n, bool inverse) {
    Fr w;
    const uint32_t root[8] = SWM_FR_ROOT47_MONT;
    for (int i = 0; i < 8; i++) w.v[i] = root[i];
    for (unsigned i = log_n; i < 47; i++) w = fp_sqr(w);
    if (inverse) w = fp_inv(w);
    return w;
}

// lazy29: every entry times 2^5, i.e. in Montgomery form of radix 2^261 (what fr29_mul takes as its second operand)
static int upload_powers(swm_ctx* ctx, const Fr& base, size_t count, void** out, bool lazy29 = false) {
    std::vector<Fr> h(count);
    Fr cur = lazy29 ? fp_from_u64<Fr>(32) : fp_one<Fr>();
    for (size_t i = 0; i < count; i++) {
        h[i] = cur;
        cur = fp_mul(cur, base);
    }
    SWM_HIP(ctx, hipMalloc(out, count * sizeof(Fr)));
    SWM_HIP(ctx, hipMemcpy(*out, h.data(), count * sizeof(Fr), hipMemcpyHostToDevice));
    return SWM_OK;
}

static int two_level_tables(swm_ctx* ctx, const Fr& base, size_t max_exp, NttTables* t, bool lazy29 = false) {
    SWM_TRY(upload_powers(ctx, base, 1024, &t->lo, lazy29));
    Fr b1024 = base;
    for (int i = 0; i < 10; i++) b1024 = fp_sqr(b1024);
    t->hi_len = (max_exp >> 10) + 1;
    SWM_TRY(upload_powers(ctx, b1024, t->hi_len, &t->hi, lazy29));
    return SWM_OK;
}
static constexpr uint64_t LAZY_KEY = 1ull << 50;  // table-cache keys of the radix-2^261 forms

static int get_root_tables_form(swm_ctx* ctx, unsigned log_n, int inverse, bool lazy29, NttTables** out) {
    uint64_t key = ((uint64_t)log_n << 1) | (inverse ? 1 : 0) | (lazy29 ? LAZY_KEY : 0);
    auto it = ctx->ntt_tables.find(key);
    if (it == ctx->ntt_tables.end()) {
        NttTables t;
        SWM_TRY(two_level_tables(ctx, host_root_of_unity(log_n, inverse), 1ull << log_n, &t, lazy29));
        it = ctx->ntt_tables.emplace(key, t).first;
    }
    *out = &it->second;
    return SWM_OK;
}

int get_root_tables(swm_ctx* ctx, unsigned log_n, int inverse, NttTables** out) {
    return get_root_tables_form(ctx, log_n, inverse, false, out);
}

static int get_coset_tables(swm_ctx* ctx, unsigned log_n, int inverse, NttTables** out, bool lazy29 = false) {
    uint64_t key = (1ull << 40) | (inverse ? 1 : 0) | (lazy29 ? LAZY_KEY : 0);
    auto it = ctx->ntt_tables.find(key);
    if (it != ctx->ntt_tables.end() && it->second.hi_len < ((1ull << log_n) >> 10) + 1) {
        (void)hipFree(it->second.lo);
        (void)hipFree(it->second.hi);
        ctx->ntt_tables.erase(it);
        it = ctx->ntt_tables.end();
    }
    if (it == ctx->ntt_tables.end()) {
        Fr g;
        const uint32_t gm[8] = SWM_FR_GEN_MONT, gi[8] = SWM_FR_GEN_INV_MONT;
        for (int i = 0; i < 8; i++) g.v[i] = inverse ? gi[i] : gm[i];
        NttTables t;
        unsigned cap = log_n < 20 ? 20 : log_n;  // build for at least 2^20 so that it is rarely rebuilt
        SWM_TRY(two_level_tables(ctx, g, 1ull << cap, &t, lazy29));
        it = ctx->ntt_tables.emplace(key, t).first;
    }
    *out = &it->second;
    return SWM_OK;
}

static int get_small_table(swm_ctx* ctx, unsigned log_r, int inverse, const Fr** out, bool lazy29 = false) {
    uint64_t key = ((uint64_t)log_r << 1) | (inverse ? 1 : 0) | (lazy29 ? LAZY_KEY : 0);
    auto it = ctx->ntt_small.find(key);
    if (it == ctx->ntt_small.end()) {
        void* d = nullptr;
        size_t cnt = log_r ? (1u << (log_r - 1)) : 1;
        SWM_TRY(upload_powers(ctx, host_root_of_unity(log_r, inverse), cnt, &d, lazy29));
        it = ctx->ntt_small.emplace(key, d).first;
    }
    *out = reinterpret_cast<const Fr*>(it->second);
    return SWM_OK;
}

// out[t Ns + k] = w_n^((k t) << shift) in radix-2^261 form, canonical: the inter-pass twiddles of one pass (built once per
// (size, direction, pass) and kept: 2^(log_ns + log_r) entries — n for the last pass of a transform)
__global__ void __launch_bounds__(256) ntt_build_pass_tw(const Fr* __restrict__ lo29, const Fr* __restrict__ hi29, unsigned log_ns,
                                                         unsigned log_r, unsigned shift, Fr* __restrict__ out) {
    const uint64_t x = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    if (x >> (log_ns + log_r)) return;
    const uint64_t k = x & (((uint64_t)1 << log_ns) - 1), t = x >> log_ns;
    out[x] = fr29_pack(fr29_canonical(two_level_pow29(lo29, hi29, (k * t) << shift), true));
}
static int get_pass_table(swm_ctx* ctx, unsigned log_n, int inverse, unsigned log_ns, unsigned log_r, const NttTables* rt29,
                          const Fr** out) {
    *out = nullptr;
    static const bool off = env_switch("SWM_NTT_PASS_TABLES", 1, 0, 1) == 0;  // 0: twiddles from the two-level tables (what a transform beyond the table budget runs)
    if (off) return SWM_OK;
    const uint64_t key = (1ull << 51) | ((uint64_t)log_n << 20) | ((uint64_t)log_ns << 10) | ((uint64_t)log_r << 1) | (inverse ? 1 : 0);
    auto it = ctx->ntt_small.find(key);
    if (it == ctx->ntt_small.end()) {
        const size_t count = (size_t)1 << (log_ns + log_r);
        // all pass tables of a context together stay below 4 GB (2^22: 2 x 134 MB + 2 x 1 MB per direction)
        if (ctx->ntt_pass_table_bytes + count * sizeof(Fr) > ((size_t)4 << 30)) return SWM_OK;
        void* d = nullptr;
        if (hipMalloc(&d, count * sizeof(Fr)) != hipSuccess) {
            (void)hipGetLastError();
            return SWM_OK;  // no room: the two-level product stays
        }
        hipLaunchKernelGGL(ntt_build_pass_tw, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, ctx->stream,
                           reinterpret_cast<const Fr*>(rt29->lo), reinterpret_cast<const Fr*>(rt29->hi), log_ns, log_r,
                           log_n - log_ns - log_r, reinterpret_cast<Fr*>(d));
        SWM_HIP(ctx, hipGetLastError());
        ctx->ntt_pass_table_bytes += count * sizeof(Fr);
        it = ctx->ntt_small.emplace(key, d).first;
    }
    *out = reinterpret_cast<const Fr*>(it->second);
    return SWM_OK;
}

// limbs of k r (k <= 256) with `borrows` x 2^29 moved into every limb below the top one (fr29_sub's SPREAD)
static Spread29 lazy_spread(unsigned k, unsigned borrows) {
    Spread29 sp;
    uint64_t carry = 0;
    for (int i = 0; i < 9; i++) {
        uint64_t t = (uint64_t)Fr29Consts::P[i] * k + carry;
        sp.l[i] = i < 8 ? (uint32_t)(t & M29) : (uint32_t)t;
        carry = t >> 29;
    }
    sp.l[0] += borrows << 29;
    for (int i = 1; i < 8; i++) sp.l[i] += (borrows << 29) - borrows;
    sp.l[8] -= borrows;
    return sp;
}
// the bounds of one pass (see ntt_pass_lazy): inputs < b_in r
static LazyPlan lazy_plan(unsigned log_r, unsigned b_in) {
    LazyPlan P;
    memset(&P, 0, sizeof(P));
    unsigned B = b_in;
    P.r2 = lazy_spread(2 * B, 1);
    if (log_r & 1) B = std::max(2 * B, 2u);
    const unsigned nsteps = log_r / 2;
    for (unsigned st = 0; st < nsteps && st < 6; st++) {
        P.s1[st] = lazy_spread(2 * B, 1);
        P.s2[st] = lazy_spread(4 * B, 2);
        const bool red = 4 * B > 64 || st + 1 == nsteps;
        if (red) P.reduce |= 1u << st;
        B = red ? 4 : 4 * B;
    }
    P.s4 = lazy_spread(4, 1);
    P.out_below_2r = B <= 2 ? 1u : 0u;
    return P;
}

// In-place (from the caller's view) transform of 2^log_n Montgomery Fr elements resident in HBM.
int ntt_run_from(swm_ctx* ctx, void* d_data, unsigned log_n, int inverse, int coset, const void* first_src, size_t src_len);
int ntt_run(swm_ctx* ctx, void* d_data, unsigned log_n, int inverse, int coset) {
    return ntt_run_from(ctx, d_data, log_n, inverse, coset, nullptr, 0);
}
// The transform of first_src[0 .. src_len) zero-extended to 2^log_n elements, written to d_data (which need not be
// initialised; first_src is left as it was and must not overlap d_data).  The first pass reads first_src and takes the
// missing inputs as zero: no padded copy in front of the transform.  first_src == nullptr: in place on d_data.
int ntt_run_from(swm_ctx* ctx, void* d_data, unsigned log_n, int inverse, int coset, const void* first_src, size_t src_len) {
    if (log_n > 30) return set_err(ctx, SWM_ERR_INVALID_ARG, "ntt: log_n > 30 unsupported");
    const uint64_t n = 1ull << log_n;
    if (first_src && src_len > n) src_len = n;
    ctx->stat_ntt_calls++;
    ctx->log_call('n', log_n);
    ctx->stat_ntt_elems += n;
    Fr* data = reinterpret_cast<Fr*>(d_data);
    // pass plan
    const unsigned maxr = NTT_MAX_LOG_R;
    // arithmetic: lazy 29-bit limbs (fr29.cuh, ntt_pass_lazy) unless SWM_NTT_LAZY=0 asks for the 32-bit-limb kernel of r01 / r02
    // (tiles of 2^11 / 2^12 elements — 76 / 152 KB of LDS — make 2^22 / 2^24 two passes: measured slower in r04, CHANGELOG.md)
    static const bool lazy = env_switch("SWM_NTT_LAZY", 1, 0, 1) != 0;
    NttTables *rt = nullptr, *ct = nullptr;
    SWM_TRY(get_root_tables_form(ctx, log_n, inverse, lazy, &rt));
    if (coset) SWM_TRY(get_coset_tables(ctx, log_n, inverse, &ct, lazy));
    Fr n_inv = fp_one<Fr>();
    if (inverse) n_inv = fp_inv(fp_from_u64<Fr>(n));
    if (lazy) n_inv = fp_mul(n_inv, fp_from_u64<Fr>(32));  // radix 2^261
    unsigned npass = log_n <= maxr ? 1 : (log_n + maxr - 1) / maxr;
    unsigned radices[8];
    {
        unsigned rem = log_n;
        for (unsigned p = 0; p < npass; p++) {
            unsigned r = (rem + (npass - p) - 1) / (npass - p);
            radices[p] = r;
            rem -= r;
        }
    }
    Fr *tmp = nullptr, *tmp2 = nullptr;
    SWM_TRY(scratch(ctx, "ntt.tmp", n * sizeof(Fr), (void**)&tmp));
    // The last pass has to land in `data`.  Even pass counts ping-pong data <-> tmp; an odd count of three or more goes
    // data -> tmp -> tmp2 -> ... -> data through a second scratch buffer (one more n-element buffer in HBM instead of a
    // full device-to-device copy in front of every such transform); a single pass (n <= 2^10) copies, it is tiny.
    if (npass % 2 == 1 && npass >= 3) SWM_TRY(scratch(ctx, "ntt.tmp2", n * sizeof(Fr), (void**)&tmp2));
    const Fr* src = first_src ? reinterpret_cast<const Fr*>(first_src) : data;
    if (npass == 1 && !first_src) {
        SWM_HIP(ctx, hipMemcpyAsync(tmp, data, n * sizeof(Fr), hipMemcpyDeviceToDevice, ctx->stream));
        src = tmp;
    }
    unsigned log_ns = 0;
    for (unsigned p = 0; p < npass; p++) {
        NttPassArgs a;
        a.src = src;
        if (p == npass - 1) a.dst = data;
        else if (tmp2) a.dst = (p % 2 == 0) ? tmp : tmp2;
        else a.dst = (src == tmp) ? data : tmp;
        a.src_len = (p == 0 && first_src) ? (uint64_t)src_len : n;
        a.log_n = log_n;
        a.log_r = radices[p];
        a.log_ns = log_ns;
        SWM_TRY(get_small_table(ctx, a.log_r, inverse, &a.tw_small, lazy));
        a.tw_lo = reinterpret_cast<const Fr*>(rt->lo);
        a.tw_hi = reinterpret_cast<const Fr*>(rt->hi);
        a.cs_lo = ct ? reinterpret_cast<const Fr*>(ct->lo) : nullptr;
        a.cs_hi = ct ? reinterpret_cast<const Fr*>(ct->hi) : nullptr;
        a.pass_tw = nullptr;
        if (lazy && log_ns != 0) SWM_TRY(get_pass_table(ctx, log_n, inverse, log_ns, a.log_r, rt, &a.pass_tw));
        a.coset_in = (coset && !inverse && p == 0) ? 1 : 0;
        a.scale_out = (inverse && p == npass - 1) ? 1 : 0;
        a.coset_out = (coset && inverse && p == npass - 1) ? 1 : 0;
        a.n_inv = n_inv;
        uint64_t cols = n >> a.log_r;
        if (lazy) {
            NttLazyArgs la;
            la.a = a;
            la.plan = lazy_plan(a.log_r, (p > 0 || a.coset_in) ? 2u : 1u);
            la.last_pass = p == npass - 1 ? 1 : 0;
            unsigned J = 1;
            if (npass > 1) {
                J = (log_n >= 19 ? 1024u : 512u) >> a.log_r;
                if (J < 1) J = 1;
                if (J > 16) J = 16;
                while (J > cols) J >>= 1;
            }
            const size_t elems = (size_t)J << a.log_r;
            size_t shmem = (elems + (elems >> 5) + 1) * 36;  // tile_slot: one slot of padding per 32 elements
            if (shmem < 64) shmem = 64;
            dim3 grid((unsigned)(cols / J)), block(NTT_THREADS);
            if (shmem > 64 * 1024) {  // only the experimental tile sizes: J = 1
                static std::atomic<size_t> granted[64];
                if (J != 1 || shmem > 160 * 1024) return set_err(ctx, SWM_ERR_INVALID_ARG, "ntt: tile does not fit LDS");
                if (granted[ctx->device & 63].load() < shmem) {
                    SWM_HIP(ctx, hipFuncSetAttribute((const void*)ntt_pass_lazy<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
                    granted[ctx->device & 63].store(shmem);
                }
            }
            switch (J) {
                case 1: SWM_LAUNCH(ctx, "ntt_pass", ntt_pass_lazy<1>, grid, block, shmem, la); break;
                case 2: SWM_LAUNCH(ctx, "ntt_pass", ntt_pass_lazy<2>, grid, block, shmem, la); break;
                case 4: SWM_LAUNCH(ctx, "ntt_pass", ntt_pass_lazy<4>, grid, block, shmem, la); break;
                case 8: SWM_LAUNCH(ctx, "ntt_pass", ntt_pass_lazy<8>, grid, block, shmem, la); break;
                default: SWM_LAUNCH(ctx, "ntt_pass", ntt_pass_lazy<16>, grid, block, shmem, la); break;
            }
        } else if (npass == 1) {
            size_t shmem = sizeof(Fr) << a.log_r;
            if (shmem < 64) shmem = 64;
            SWM_LAUNCH(ctx, "ntt_pass", ntt_pass<1>, dim3((unsigned)cols), dim3(NTT_THREADS), shmem, a);
        } else {
            unsigned J = (log_n >= 19 ? 1024u : 512u) >> a.log_r;
            if (J < 1) J = 1;
            if (J > 16) J = 16;
            while (J > cols) J >>= 1;
            size_t shmem = (sizeof(Fr) * J) << a.log_r;
            dim3 grid((unsigned)(cols / J)), block(NTT_THREADS);
            switch (J) {
                case 1: SWM_LAUNCH(ctx, "ntt_pass", ntt_pass<1>, grid, block, shmem, a); break;
                case 2: SWM_LAUNCH(ctx, "ntt_pass", ntt_pass<2>, grid, block, shmem, a); break;
                case 4: SWM_LAUNCH(ctx, "ntt_pass", ntt_pass<4>, grid, block, shmem, a); break;
                case 8: SWM_LAUNCH(ctx, "ntt_pass", ntt_pass<8>, grid, block, shmem, a); break;
                default: SWM_LAUNCH(ctx, "ntt_pass", ntt_pass<16>, grid, block, shmem, a); break;
            }
        }
        log_ns += a.log_r;
        src = a.dst;
    }
    return SWM_OK;
}


// ---------------------------------------------------------------------------------------------- one transform over G GPUs
// SURVEY.md §8e "NTT partitioning (ii)": the four-step split of ONE transform of n = 2^log_n elements over G = 2^g ranks
// (m = n / G elements per rank, blk = m / G) with a single all-to-all.  A transform with one exchange maps one layout to
// the other of
//     CYCLIC   local[j] = v[rank + G j]                              (j < m)
//     BLOCKS   local[k1 blk + t] = v[m k1 + rank blk + t]            (k1 < G, t < blk)
// (with i = i1 + G i2 and k = m k1 + k2:  w^(ik) = w_G^(i1 k1) w_n^(i1 k2) w_m^(i2 k2)):
//   CYCLIC -> BLOCKS   local length-m transform over i2 (the existing passes) | twiddle w_n^(rank k2) | all-to-all: chunk c of
//                      k2 to rank c | length-G transform over i1 of what arrived (G values per column, in registers)
//   BLOCKS -> CYCLIC   the same four steps backwards: length-G transform over the block index | twiddle w_n^(i_lo k1) |
//                      all-to-all: block k1 to rank k1 | local length-m transform
// Per pair of GPUs the exchange moves n * 32 / G^2 bytes (2^22, 8 GPUs: 2 MB over each direct xGMI link).  The prover keeps
// evaluations in BLOCKS and coefficients in CYCLIC layout: an inverse transform (BLOCKS -> CYCLIC) leaves every rank with
// the coefficients rank, rank + G, ... — which the sharded commitment MSM takes as they are (MsmTable::blk_log = 0,
// bstride = G): no gather between the transform and the commitment.
__global__ void __launch_bounds__(256) ntt_shard_twiddle(Fr* __restrict__ v, size_t count, const Fr* __restrict__ lo,
                                                         const Fr* __restrict__ hi, uint64_t e_mul, uint64_t e_add_mul, unsigned blk_log) {
    // v[x] *= w^(e(x)):  CYCLIC -> BLOCKS: e = rank * x (e_mul = rank, e_add_mul = 0);
    //                    BLOCKS -> CYCLIC: x = k1 blk + t, e = (rank blk + t) k1  (e_mul = 0 marks this form, e_add_mul = rank blk)
    size_t x = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (x >= count) return;
    uint64_t e;
    if (e_mul) e = e_mul * x;
    else e = (e_add_mul + (x & (((uint64_t)1 << blk_log) - 1))) * (x >> blk_log);
    if (e) v[x] = fp_mul(v[x], two_level_pow(lo, hi, e));
}
// dst[k1 blk + t] = scale * sum_{i1 < G} w_G^(i1 k1) src[i1 blk + t],  w_G^e = w_n^(e m) from the two-level tables
template <int G>
__global__ void __launch_bounds__(256) ntt_shard_cross(const Fr* __restrict__ src, Fr* __restrict__ dst, size_t blk, const Fr* __restrict__ lo,
                                                       const Fr* __restrict__ hi, uint64_t m, int scale, Fr scale_by) {
    size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (t >= blk) return;
    Fr x[G];
#pragma unroll
    for (int i = 0; i < G; i++) x[i] = src[(size_t)i * blk + t];
#pragma unroll 1
    for (int k = 0; k < G; k++) {
        Fr acc = x[0];
#pragma unroll
        for (int i = 1; i < G; i++) {
            const unsigned e = (unsigned)(i * k) & (G - 1);
            acc = fp_add(acc, e ? fp_mul(x[i], two_level_pow(lo, hi, (uint64_t)e * m)) : x[i]);
        }
        if (scale) acc = fp_mul(acc, scale_by);
        dst[(size_t)k * blk + t] = acc;
    }
}
int ntt_sharded_run(swm_ctx* ctx, void* d_local, unsigned log_n, int inverse, int blocks_in) {
    const unsigned G = ctx->shard_world, rank = ctx->shard_rank;
    unsigned log_g = 0;
    while ((1u << log_g) < G) log_g++;
    if ((1u << log_g) != G || G > 16) return set_err(ctx, SWM_ERR_INVALID_ARG, "sharded ntt: the number of ranks must be a power of two <= 16");
    if (log_n > 30 || log_n < 2 * log_g) return set_err(ctx, SWM_ERR_INVALID_ARG, "sharded ntt: needs at least G^2 elements");
    if (G == 1) return ntt_run(ctx, d_local, log_n, inverse, 0);
    const unsigned log_m = log_n - log_g, blk_log = log_m - log_g;
    const size_t m = (size_t)1 << log_m, blk = (size_t)1 << blk_log;
    Fr* local = reinterpret_cast<Fr*>(d_local);
    NttTables* rt = nullptr;
    SWM_TRY(get_root_tables(ctx, log_n, inverse, &rt));
    const Fr *lo = reinterpret_cast<const Fr*>(rt->lo), *hi = reinterpret_cast<const Fr*>(rt->hi);
    Fr* tmp = nullptr;
    SWM_TRY(scratch(ctx, "ntt.shard", m * sizeof(Fr), (void**)&tmp));
    Fr g_inv = fp_one<Fr>();
    if (inverse) g_inv = fp_inv(fp_from_u64<Fr>(G));  // the local transform scales by 1 / m, the cross step by 1 / G
    const dim3 grid_m((unsigned)((m + 255) / 256)), grid_b((unsigned)((blk + 255) / 256));
    auto cross = [&](const Fr* src, Fr* dst) -> int {
        switch (G) {
            case 2: SWM_LAUNCH(ctx, "ntt_shard_cross", ntt_shard_cross<2>, grid_b, dim3(256), 0, src, dst, blk, lo, hi, (uint64_t)m, inverse, g_inv); break;
            case 4: SWM_LAUNCH(ctx, "ntt_shard_cross", ntt_shard_cross<4>, grid_b, dim3(256), 0, src, dst, blk, lo, hi, (uint64_t)m, inverse, g_inv); break;
            case 8: SWM_LAUNCH(ctx, "ntt_shard_cross", ntt_shard_cross<8>, grid_b, dim3(256), 0, src, dst, blk, lo, hi, (uint64_t)m, inverse, g_inv); break;
            default: SWM_LAUNCH(ctx, "ntt_shard_cross", ntt_shard_cross<16>, grid_b, dim3(256), 0, src, dst, blk, lo, hi, (uint64_t)m, inverse, g_inv); break;
        }
        return SWM_OK;
    };
    if (!blocks_in) {  // CYCLIC -> BLOCKS
        SWM_TRY(ntt_run(ctx, local, log_m, inverse, 0));
        if (rank) SWM_LAUNCH(ctx, "ntt_shard_twiddle", ntt_shard_twiddle, grid_m, dim3(256), 0, local, m, lo, hi, (uint64_t)rank, (uint64_t)0, blk_log);
        SWM_TRY(shard_alltoall_dev(ctx, local, tmp, blk * sizeof(Fr)));
        SWM_TRY(cross(tmp, local));
    } else {           // BLOCKS -> CYCLIC
        SWM_TRY(cross(local, tmp));
        SWM_LAUNCH(ctx, "ntt_shard_twiddle", ntt_shard_twiddle, grid_m, dim3(256), 0, tmp, m, lo, hi, (uint64_t)0, (uint64_t)rank * blk, blk_log);
        SWM_TRY(shard_alltoall_dev(ctx, tmp, local, blk * sizeof(Fr)));
        SWM_TRY(ntt_run(ctx, local, log_m, inverse, 0));
    }
    return SWM_OK;
}

}  // namespace swm
