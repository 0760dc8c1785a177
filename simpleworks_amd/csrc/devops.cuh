// devops.cuh — device-resident Fr vectors and the polynomial plumbing of the Marlin prover (SURVEY.md §8f rank 1:
// "KZG open + polynomial plumbing on device"): pointwise kernels, Horner evaluation, division by (X^m - z) as a
// blocked linear recurrence, strided vanishing-polynomial division, and bulk ark_ff::UniformRand sampling from a
// ChaCha keystream.  Everything here is HBM-bound integer work: one lane per element (or per short chunk),
// 32-B elements, no LDS staging needed, no MFMA.
// Replaces the cfg_iter!/rayon loops of ark-marlin's ahp/prover.rs and ark-poly's DensePolynomial helpers that
// /root/reference/src/marlin/mod.rs:75 reaches (sources not vendored; behaviour from SURVEY.md A.3, A.5-A.7).
#pragma once
#include <chrono>
#include <exception>
#include <functional>
#include <utility>
#include <atomic>
#include "context.h"
#include "ff.cuh"
#include "fill.cuh"
#include "fr29.cuh"
#include "host/chacha.h"
#include "host/marlin_types.h"

namespace swm {

int ntt_run(swm_ctx* ctx, void* d_data, unsigned log_n, int inverse, int coset);
int ntt_run_from(swm_ctx* ctx, void* d_data, unsigned log_n, int inverse, int coset, const void* first_src, size_t src_len);
// one transform of 2^log_n elements over the ranks of the context's sharding (ntt.hip): in place on the rank's n / G
// elements, CYCLIC -> BLOCKS layout (blocks_in = 0) or BLOCKS -> CYCLIC (blocks_in = 1), one all-to-all
int ntt_sharded_run(swm_ctx* ctx, void* d_local, unsigned log_n, int inverse, int blocks_in);
// What is known about a CSR matrix once its row pointers have been seen on the host (spmv.hip): the longest row picks the
// schedule; rows longer than the direct kernel handles are pre-cut into chunks (row, start, length) with one
// (row, first chunk, chunk count) record per long row.
struct SpmvPlanHost {
    uint64_t nnz = 0, max_row = 0;
    std::vector<uint32_t> chunks, lrows;
};
struct SpmvPlan {
    uint64_t nnz = 0, max_row = 0;
    const uint32_t* d_chunks = nullptr;
    const uint32_t* d_lrows = nullptr;
    uint32_t n_chunks = 0, n_lrows = 0;
};
void spmv_plan_build(const uint32_t* rowptr, size_t rows, SpmvPlanHost* plan);
int spmv_run(swm_ctx* ctx, const void* d_rowptr, const void* d_col, const void* d_val, const void* d_z, void* d_out,
             size_t rows, const SpmvPlan* plan = nullptr);
int batch_inverse_run(swm_ctx* ctx, void* d, size_t n);

inline void hip_check(swm_ctx* ctx, hipError_t e, const char* what) {
    if (e != hipSuccess) {
        set_err(ctx, SWM_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
        throw MarlinError(e == hipErrorOutOfMemory ? SWM_ERR_OOM : SWM_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
    }
}
inline void rc_check(swm_ctx* ctx, int rc) {
    if (rc != SWM_OK) throw MarlinError(rc, ctx->err);
}

// ------------------------------------------------------------------------------------------------ device vectors
template <class T>
struct DBuf {
    swm_ctx* ctx = nullptr;
    T* p = nullptr;
    size_t n = 0, cap = 0;
    DBuf() {}
    DBuf(swm_ctx* c, size_t count) : ctx(c), n(count) {
        void* q = nullptr;
        rc_check(c, pool_alloc(c, (count ? count : 1) * sizeof(T), &q, &cap));
        p = (T*)q;
    }
    DBuf(const DBuf&) = delete;
    DBuf& operator=(const DBuf&) = delete;
    DBuf(DBuf&& o) noexcept { *this = std::move(o); }
    DBuf& operator=(DBuf&& o) noexcept {
        if (this != &o) {
            release();
            ctx = o.ctx; p = o.p; n = o.n; cap = o.cap;
            o.p = nullptr; o.n = 0; o.cap = 0;
        }
        return *this;
    }
    ~DBuf() { release(); }
    // Hands the block over from the context's pool to its holder: a detached buffer belongs to no context (a proving key is
    // resident per DEVICE and may outlive the context that built it, swm_pk) and goes back to the runtime when it is released.
    // Every pool block is a hipMalloc of its own, so nothing else changes.  The methods below that enqueue on ctx->stream are
    // not for detached buffers.
    void detach() { ctx = nullptr; }
    void release() {
        if (p) {
            if (!ctx) {
                (void)hipFree(p);
            } else {
                // unwinding after a failure: kernels queued on the main or the auxiliary MSM streams may still read this
                // block; wait for them before it returns to the pool (idle streams make this a no-op)
                if (std::uncaught_exceptions() > 0) drain_streams(ctx);
                pool_free(ctx, p, cap);
            }
        }
        p = nullptr;
    }
    void zero() { hip_check(ctx, zero_fill_async(p, n * sizeof(T), ctx->stream), "memset"); }
    void upload(const T* h, size_t count) {
        hip_check(ctx, hipMemcpyAsync(p, h, count * sizeof(T), hipMemcpyHostToDevice, ctx->stream), "h2d");
        hip_check(ctx, hipStreamSynchronize(ctx->stream), "sync");
    }
    std::vector<T> download(size_t off, size_t count) const {
        std::vector<T> h(count);
        hip_check(ctx, hipMemcpyAsync(h.data(), p + off, count * sizeof(T), hipMemcpyDeviceToHost, ctx->stream), "d2h");
        hip_check(ctx, hipStreamSynchronize(ctx->stream), "sync");
        return h;
    }
};
typedef DBuf<Fr> DVec;

inline DVec dv_zeros(swm_ctx* ctx, size_t n) {
    DVec v(ctx, n);
    v.zero();
    return v;
}
// copy of src[0..len) zero-extended to n elements
inline DVec dv_copy_padded(swm_ctx* ctx, const Fr* src, size_t len, size_t n) {
    DVec v(ctx, n);
    if (len > n) len = n;
    if (len) hip_check(ctx, hipMemcpyAsync(v.p, src, len * sizeof(Fr), hipMemcpyDeviceToDevice, ctx->stream), "d2d");
    if (n > len) hip_check(ctx, zero_fill_async(v.p + len, (n - len) * sizeof(Fr), ctx->stream), "memset");
    return v;
}

// first kernel of every proof when SWM_TRACE / SWM_PROOF_MARKS is set: the delimiter tools/trace_dump.py and trace_share.py cut a
// kernel trace by (a proof with a caller-owned generator has no bulk-sampling kernel to go by)
static __global__ void swm_proof_begin() {}

// ------------------------------------------------------------------------------------------------ pointwise launcher
template <class F>
static __global__ void __launch_bounds__(256) ew_kernel(size_t n, F f) {
    SWM_LIGHT_KERNEL();
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) f(i);
}
template <class F>
inline void ew(swm_ctx* ctx, const char* name, size_t n, F f) {
    if (n == 0) return;
    unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, 256 * 32);
    prof_begin(ctx, name);
    hipLaunchKernelGGL(ew_kernel<F>, dim3(grid), dim3(256), 0, ctx->stream, n, f);
    prof_end(ctx);
    hip_check(ctx, hipGetLastError(), name);
}

// w^e from the two-level tables (lo: e mod 1024, hi: e div 1024)
struct PowTable {
    const Fr* lo;
    const Fr* hi;
    __device__ __forceinline__ Fr at(uint64_t e) const {
        Fr a = lo[e & 1023];
        uint64_t h = e >> 10;
        if (h) a = fp_mul(a, hi[h]);
        return a;
    }
};
inline PowTable root_pow_table(swm_ctx* ctx, unsigned log_n, bool inverse = false) {
    NttTables* t = nullptr;
    rc_check(ctx, get_root_tables(ctx, log_n, inverse ? 1 : 0, &t));
    return PowTable{(const Fr*)t->lo, (const Fr*)t->hi};
}

inline void dv_ntt(swm_ctx* ctx, DVec& v, unsigned log_n, bool inverse, bool coset = false) {
    if (v.n != ((size_t)1 << log_n)) throw MarlinError(SWM_ERR_INTERNAL, "dv_ntt: size mismatch");
    rc_check(ctx, ntt_run(ctx, v.p, log_n, inverse ? 1 : 0, coset ? 1 : 0));
}

// the transform of src[0 .. len) zero-extended to 2^log_n elements, out of place (no padded copy in front: the first pass
// takes the missing inputs as zero); src is left as it was
inline DVec dv_ntt_from(swm_ctx* ctx, const Fr* src, size_t len, unsigned log_n, bool inverse, bool coset = false) {
    DVec v(ctx, (size_t)1 << log_n);
    rc_check(ctx, ntt_run_from(ctx, v.p, log_n, inverse ? 1 : 0, coset ? 1 : 0, src, len));
    return v;
}

// ------------------------------------------------------------------------------------------------ suffix recurrence
// In place: a[k] <- a[k] + z * a[k + m] (k descending), i.e. a[k] = sum_{i >= 0} z^i a_old[k + i m].
// m = 1, z = point: synthetic division (quotient of p / (X - z) = a[1..], p(z) = a[0]).
// z = 1, stride m:  quotient of p / (X^m - 1) = a[m..]  (DensePolynomial::divide_by_vanishing_poly).
static constexpr int REC_T = 16;  // rows per lane: short chains, many lanes (the kernels are latency-bound)

static __global__ void __launch_bounds__(256) rec_local(Fr* a, size_t n, size_t m, Fr z, Fr* head, size_t nblk) {
    SWM_LIGHT_KERNEL();
    // lane = (block of REC_T rows, column); rows = ceil(n / m)
    size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (t >= nblk * m) return;
    size_t blk = t / m, col = t % m;
    size_t rows = (n + m - 1) / m;
    size_t r_hi = (blk + 1) * REC_T < rows ? (blk + 1) * REC_T : rows;
    Fr acc = fp_zero<Fr>();
    for (size_t r = r_hi; r-- > blk * REC_T;) {
        size_t k = r * m + col;
        if (k >= n) continue;
        acc = fp_add(a[k], fp_mul(acc, z));
        a[k] = acc;
    }
    head[blk * m + col] = acc;
}
static __global__ void __launch_bounds__(256) rec_fix(Fr* a, size_t n, size_t m, Fr z, const Fr* head, size_t nblk) {
    SWM_LIGHT_KERNEL();
    size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (t >= nblk * m) return;
    size_t blk = t / m, col = t % m;
    if (blk + 1 >= nblk) return;  // last block has no incoming carry
    Fr carry = head[(blk + 1) * m + col];  // true value at the first row of the next block
    size_t rows = (n + m - 1) / m;
    size_t r_hi = (blk + 1) * REC_T < rows ? (blk + 1) * REC_T : rows;
    Fr pw = z;
    for (size_t r = r_hi; r-- > blk * REC_T;) {
        size_t k = r * m + col;
        if (k < n) a[k] = fp_add(a[k], fp_mul(pw, carry));
        pw = fp_mul(pw, z);
    }
}
static __global__ void rec_serial(Fr* a, size_t n, size_t m, Fr z) {
    SWM_LIGHT_KERNEL();
    size_t col = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (col >= m) return;
    size_t rows = (n + m - 1) / m;
    Fr acc = fp_zero<Fr>();
    for (size_t r = rows; r-- > 0;) {
        size_t k = r * m + col;
        if (k >= n) continue;
        acc = fp_add(a[k], fp_mul(acc, z));
        a[k] = acc;
    }
}
// ---- tiled form for the contiguous case (m = 1): three coalesced passes
//   1. rec_tile_total   T_b = sum_i a[2048 b + i] z^i                      (read n)
//   2. recursion        S_b = sum_{c >= b} T_c (z^2048)^(c - b) = true value at the first element of tile b
//   3. rec_tile_scan    full in-tile scan with carry-in S_{b+1}            (read n, write n)
// Lanes own 8 consecutive elements; tiles travel through a padded LDS image so that global traffic stays coalesced.
static constexpr int RT_PER = 8;
static constexpr int RT_TILE = 256 * RT_PER;
// out[i] = base^i, i < 256, built on the device from the eight squarings base^(2^b) (passed by value): a host-built
// table would have to be uploaded and waited for, i.e. a host synchronisation in the middle of a proof.
struct Pow256Args {
    Fr sq[8];
};
static __global__ void __launch_bounds__(256) pow256_kernel(Pow256Args a, Fr* __restrict__ out) {
    SWM_LIGHT_KERNEL();
    const unsigned i = threadIdx.x;
    Fr r = fp_one<Fr>();
#pragma unroll
    for (int b = 0; b < 8; b++)
        if ((i >> b) & 1) r = fp_mul(r, a.sq[b]);
    out[i] = r;
}
// returns base^256
inline Fr dv_pow256(swm_ctx* ctx, const Fr& base, Fr* d_out) {
    Pow256Args a;
    a.sq[0] = base;
    for (int b = 1; b < 8; b++) a.sq[b] = fp_sqr(a.sq[b - 1]);
    hipLaunchKernelGGL(pow256_kernel, dim3(1), dim3(256), 0, ctx->stream, a, d_out);
    hip_check(ctx, hipGetLastError(), "pow256");
    return fp_sqr(a.sq[7]);
}

struct RecConsts {
    Fr zpow[RT_PER + 1];  // z^0 .. z^8
    Fr zstep[9];          // (z^8)^(2^k), k = 0..8
    Fr z256;              // z^256 (pass 1 Horner step)
};
// (the 8 x 32-bit Comba forms of the two tiled passes, r02 - r04, were removed in r06: the passes below are the kernels)
__device__ __forceinline__ unsigned rt_pad(unsigned i) { return i + i / RT_PER; }  // one 32-B pad slot per lane chunk

// The two tiled passes on the transform's multiplier (r05; fr29.cuh: nine 29-bit lazy limbs, 197 instructions per product instead
// of ~330, additions without a comparison against the modulus).  `rc` holds the constants in Montgomery form of radix 2^261 (times
// 2^5: what fr29_mul takes as its second operand), so the data keep the memory format; values are < 2r between products, < 5r
// before they are made canonical for the store.  Same results bit for bit (every golden-bytes test runs through them).
__device__ __forceinline__ Fr29 fr29_zero() {
    Fr29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = 0;
    return r;
}
__device__ __forceinline__ Fr29 fr29_below_2r(const Fr29& lazy_below_4r) {  // lazy limbs, value < 4r -> normalised, < 2r
    return fr29_cond_sub(fr29_normalize(lazy_below_4r), Fr29Consts::P2);
}
static __global__ void __launch_bounds__(256) rec_tile_total29(const Fr* __restrict__ a, size_t n, RecConsts rc,
                                                               const Fr* __restrict__ zlow /* z^0..z^255, memory form */,
                                                               Fr* __restrict__ totals) {
    SWM_LIGHT_KERNEL();
    __shared__ Fr sm[256];
    const size_t base = (size_t)blockIdx.x * RT_TILE;
    const unsigned t = threadIdx.x;
    const Fr29 z256 = fr29_unpack(rc.z256);
    Fr29 acc = fr29_zero();
#pragma unroll 1
    for (int j = RT_PER - 1; j >= 0; j--) {
        size_t k = base + (size_t)j * 256 + t;
        acc = fr29_mul_fenced(acc, z256);  // (acc < 3r with limbs < 2^30)
        if (k < n) acc = fr29_add(acc, fr29_unpack(a[k]));
    }
    // zlow is in the memory format: this product comes out with 2^-5, put right on the tile's total (rc.zpow[0] = 2^261 = the
    // memory form of 2^5)
    sm[t] = fr29_pack(fr29_canonical(fr29_mul_fenced(acc, fr29_unpack(zlow[t])), true));
    __syncthreads();
    for (unsigned s = 128; s > 0; s >>= 1) {
        if (t < s) sm[t] = fp_add(sm[t], sm[t + s]);
        __syncthreads();
    }
    if (t == 0) totals[blockIdx.x] = fp_mul(sm[0], rc.zpow[0]);
}
static __global__ void __launch_bounds__(256) rec_tile_scan29(Fr* __restrict__ a, size_t n, RecConsts rc,
                                                              const Fr* __restrict__ tile_true /* S_b */, size_t ntiles) {
    SWM_LIGHT_KERNEL();
    extern __shared__ __align__(16) unsigned char smem_raw[];
    Fr* tile = reinterpret_cast<Fr*>(smem_raw);      // RT_TILE + 256 padded slots
    Fr* hs = tile + RT_TILE + 256;                   // 257 heads, each < 2r
    const size_t b = blockIdx.x, base = b * RT_TILE;
    const unsigned t = threadIdx.x;
#pragma unroll
    for (int j = 0; j < RT_PER; j++) {
        unsigned e = j * 256 + t;
        size_t k = base + e;
        tile[rt_pad(e)] = k < n ? a[k] : fp_zero<Fr>();
    }
    __syncthreads();
    // local scan of this lane's 8 consecutive elements (carry-in 0): element + product, < 3r with limbs < 2^30
    Fr29 loc[RT_PER];
    {
        const Fr29 z1 = fr29_unpack(rc.zpow[1]);
        Fr29 acc = fr29_zero();
#pragma unroll
        for (int i = RT_PER - 1; i >= 0; i--) {
            acc = fr29_add(fr29_unpack(tile[rt_pad(t * RT_PER + i)]), fr29_mul_fenced(acc, z1));
            loc[i] = acc;
        }
        hs[t] = fr29_pack(fr29_below_2r(acc));
    }
    if (t == 0) hs[256] = b + 1 < ntiles ? tile_true[b + 1] : fp_zero<Fr>();  // carry into the tile
    __syncthreads();
    // inclusive suffix scan of the heads with multiplier z^8 per lane step (Hillis-Steele, 257 entries)
#pragma unroll 1
    for (int k = 0; k < 9; k++) {
        unsigned d = 1u << k;
        Fr v = hs[t];
        bool has = t + d <= 256;
        Fr o = has ? hs[t + d] : fp_zero<Fr>();
        __syncthreads();
        if (has) hs[t] = fr29_pack(fr29_below_2r(fr29_add(fr29_unpack(v), fr29_mul_fenced(fr29_unpack(o), fr29_unpack(rc.zstep[k])))));
        __syncthreads();
    }
    const Fr29 carry = fr29_unpack(hs[t + 1]);  // true value (< 2r) at the first element of the next lane's chunk
#pragma unroll
    for (int i = 0; i < RT_PER; i++) {
        const Fr29 v = fr29_add(loc[i], fr29_mul_fenced(carry, fr29_unpack(rc.zpow[RT_PER - i])));  // < 5r
        tile[rt_pad(t * RT_PER + i)] = fr29_pack(fr29_cond_sub(fr29_cond_sub(fr29_cond_sub(fr29_normalize(v), Fr29Consts::P2), Fr29Consts::P2), Fr29Consts::P));
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RT_PER; j++) {
        unsigned e = j * 256 + t;
        size_t k = base + e;
        if (k < n) a[k] = tile[rt_pad(e)];
    }
}

inline void suffix_recurrence(swm_ctx* ctx, Fr* a, size_t n, size_t m, const Fr& z);
inline void suffix_recurrence_tiled(swm_ctx* ctx, Fr* a, size_t n, const Fr& z) {
    size_t ntiles = (n + RT_TILE - 1) / RT_TILE;
    RecConsts rc;
    rc.zpow[0] = fp_one<Fr>();
    for (int i = 1; i <= RT_PER; i++) rc.zpow[i] = fp_mul(rc.zpow[i - 1], z);
    rc.zstep[0] = rc.zpow[RT_PER];
    for (int k = 1; k < 9; k++) rc.zstep[k] = fp_sqr(rc.zstep[k - 1]);
    DVec zlow(ctx, 256), totals(ctx, ntiles);
    rc.z256 = dv_pow256(ctx, z, zlow.p);
    RecConsts rc29;  // the same constants in radix-2^261 form
    const Fr two5 = fp_from_u64<Fr>(32);
    for (int i = 0; i <= RT_PER; i++) rc29.zpow[i] = fp_mul(rc.zpow[i], two5);
    for (int k = 0; k < 9; k++) rc29.zstep[k] = fp_mul(rc.zstep[k], two5);
    rc29.z256 = fp_mul(rc.z256, two5);
    prof_begin(ctx, "rec_tile_total");
    hipLaunchKernelGGL(rec_tile_total29, dim3((unsigned)ntiles), dim3(256), 0, ctx->stream, (const Fr*)a, n, rc29, (const Fr*)zlow.p, totals.p);
    prof_end(ctx);
    hip_check(ctx, hipGetLastError(), "rec_tile_total");
    Fr ztile = rc.z256;
    for (int i = 0; i < 3; i++) ztile = fp_sqr(ztile);  // z^2048
    static_assert(RT_TILE == 2048, "tile exponent");
    suffix_recurrence(ctx, totals.p, ntiles, 1, ztile);  // totals[b] <- true value at the first element of tile b
    size_t lds = (size_t)(RT_TILE + 256 + 257) * sizeof(Fr);
    static std::atomic<bool> attr_set[64];  // per device: the attribute belongs to the device's copy of the kernel
    if (!attr_set[ctx->device & 63].load(std::memory_order_acquire)) {
        hip_check(ctx, hipFuncSetAttribute((const void*)rec_tile_scan29, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), "attr");
        attr_set[ctx->device & 63].store(true, std::memory_order_release);
    }
    prof_begin(ctx, "rec_tile_scan");
    hipLaunchKernelGGL(rec_tile_scan29, dim3((unsigned)ntiles), dim3(256), lds, ctx->stream, a, n, rc29, (const Fr*)totals.p, ntiles);
    prof_end(ctx);
    hip_check(ctx, hipGetLastError(), "rec_tile_scan");
}

inline void suffix_recurrence(swm_ctx* ctx, Fr* a, size_t n, size_t m, const Fr& z) {
    if (n == 0) return;
    if (m == 1 && n >= 4 * (size_t)RT_TILE) {
        suffix_recurrence_tiled(ctx, a, n, z);
        return;
    }
    size_t rows = (n + m - 1) / m;
    if (rows <= REC_T) {
        prof_begin(ctx, "rec_serial");
        hipLaunchKernelGGL(rec_serial, dim3((unsigned)((m + 63) / 64)), dim3(64), 0, ctx->stream, a, n, m, z);
        prof_end(ctx);
        hip_check(ctx, hipGetLastError(), "rec_serial");
        return;
    }
    size_t nblk = (rows + REC_T - 1) / REC_T;
    DVec head(ctx, nblk * m);
    unsigned grid = (unsigned)((nblk * m + 255) / 256);
    prof_begin(ctx, "rec_local");
    hipLaunchKernelGGL(rec_local, dim3(grid), dim3(256), 0, ctx->stream, a, n, m, z, head.p, nblk);
    prof_end(ctx);
    hip_check(ctx, hipGetLastError(), "rec_local");
    // heads obey the same recurrence with multiplier z^REC_T
    Fr zt = z;
    for (int i = 0; i < 4; i++) zt = fp_sqr(zt);  // z^REC_T
    static_assert(REC_T == 16, "zt exponent");
    suffix_recurrence(ctx, head.p, nblk * m, m, zt);
    prof_begin(ctx, "rec_fix");
    hipLaunchKernelGGL(rec_fix, dim3(grid), dim3(256), 0, ctx->stream, a, n, m, z, head.p, nblk);
    prof_end(ctx);
    hip_check(ctx, hipGetLastError(), "rec_fix");
}

// p(x) for a device polynomial: copy + recurrence, value at index 0.  (Also yields the quotient by (X - x).)
struct DivResult {
    DVec work;  // work[0] = p(x); work[1..n) = quotient of p / (X - x)
};
inline DivResult div_linear(swm_ctx* ctx, const Fr* p, size_t n, const Fr& x) {
    DivResult r;
    r.work = dv_copy_padded(ctx, p, n, n ? n : 1);
    if (n == 0) r.work.zero();
    suffix_recurrence(ctx, r.work.p, n, 1, x);
    return r;
}

// ------------------------------------------------------------------------------------------------ Horner evaluation
// One workgroup evaluates a contiguous tile of EVAL_TILE = 256 * EVAL_PER coefficients: lane t runs Horner in
// y = x^256 over the coefficients t, t + 256, t + 512, ... of the tile (adjacent lanes read adjacent 32-B elements:
// coalesced), multiplies by x^t, and the workgroup sums its 256 values in LDS.  out[tile] = sum_i c[tile*T + i] x^i,
// so the next level evaluates `out` at x^T.
static constexpr int EVAL_PER = 16;
static constexpr int EVAL_TILE = 256 * EVAL_PER;
static __global__ void __launch_bounds__(256) eval_chunks(const Fr* __restrict__ c, size_t n, Fr x256, const Fr* __restrict__ xpow,
                                                          Fr* __restrict__ out) {
    SWM_LIGHT_KERNEL();
    __shared__ Fr sm[256];
    const size_t base = (size_t)blockIdx.x * EVAL_TILE;
    const unsigned t = threadIdx.x;
    Fr acc = fp_zero<Fr>();
#pragma unroll 1
    for (int j = EVAL_PER - 1; j >= 0; j--) {
        size_t k = base + (size_t)j * 256 + t;
        acc = fp_mul(acc, x256);
        if (k < n) acc = fp_add(acc, c[k]);
    }
    sm[t] = fp_mul(acc, xpow[t]);
    __syncthreads();
    for (unsigned s = 128; s > 0; s >>= 1) {
        if (t < s) sm[t] = fp_add(sm[t], sm[t + s]);
        __syncthreads();
    }
    if (t == 0) out[blockIdx.x] = sm[0];
}
// Evaluation point prepared once: per level l (tile T = 4096), the table y_l^0 .. y_l^255 and y_l^256, y_0 = x,
// y_{l+1} = y_l^T.  Three levels cover 2^36 coefficients.
struct EvalPoint {
    DVec pw[3];
    Fr y256[3];
};
inline EvalPoint eval_point(swm_ctx* ctx, Fr x) {
    EvalPoint ep;
    for (int l = 0; l < 3; l++) {
        ep.pw[l] = DVec(ctx, 256);
        ep.y256[l] = dv_pow256(ctx, x, ep.pw[l].p);
        x = ep.y256[l];
        for (int i = 0; i < 4; i++) x = fp_sqr(x);  // y^(256 * 16)
        static_assert(EVAL_PER == 16, "x exponent");
    }
    return ep;
}
// p(x) for a device polynomial, written to the device word *d_result (no host synchronisation)
inline void poly_eval_async(swm_ctx* ctx, const Fr* p, size_t n, const EvalPoint& ep, Fr* d_result) {
    if (n == 0) {
        hip_check(ctx, hipMemsetAsync(d_result, 0, sizeof(Fr), ctx->stream), "memset");
        return;
    }
    DVec cur;
    const Fr* src = p;
    size_t len = n;
    for (int level = 0;; level++) {
        if (level >= 3) throw MarlinError(SWM_ERR_INTERNAL, "poly_eval: polynomial too long");
        size_t ntiles = (len + EVAL_TILE - 1) / EVAL_TILE;
        DVec next;
        Fr* dst = d_result;
        if (ntiles > 1) {
            next = DVec(ctx, ntiles);
            dst = next.p;
        }
        prof_begin(ctx, "poly_eval");
        hipLaunchKernelGGL(eval_chunks, dim3((unsigned)ntiles), dim3(256), 0, ctx->stream, src, len, ep.y256[level],
                           ep.pw[level].p, dst);
        prof_end(ctx);
        hip_check(ctx, hipGetLastError(), "poly_eval");
        if (ntiles == 1) break;
        cur = std::move(next);
        src = cur.p;
        len = ntiles;
    }
}
// Several polynomials at the SAME point in one launch per level (the evaluation phase of a proof reads ~20 polynomials
// at beta and ~6 at gamma: one launch each was 70 launches of ~25 us for a small proof).  Same tiles, same arithmetic
// and the same order of additions as eval_chunks, so the values are bit-identical to poly_eval_async's.
static constexpr int EVAL_MANY = 32;
struct EvalBatch {
    const Fr* c[EVAL_MANY];
    size_t n[EVAL_MANY];
    Fr* out[EVAL_MANY];
};
static __global__ void __launch_bounds__(256) eval_chunks_many(EvalBatch b, Fr x256, const Fr* __restrict__ xpow) {
    SWM_LIGHT_KERNEL();
    __shared__ Fr sm[256];
    const Fr* __restrict__ c = b.c[blockIdx.y];
    const size_t n = b.n[blockIdx.y];
    const size_t base = (size_t)blockIdx.x * EVAL_TILE;
    if (base >= n) return;  // this polynomial has fewer tiles than the longest of the batch (uniform per workgroup)
    const unsigned t = threadIdx.x;
    Fr acc = fp_zero<Fr>();
#pragma unroll 1
    for (int j = EVAL_PER - 1; j >= 0; j--) {
        size_t k = base + (size_t)j * 256 + t;
        acc = fp_mul(acc, x256);
        if (k < n) acc = fp_add(acc, c[k]);
    }
    sm[t] = fp_mul(acc, xpow[t]);
    __syncthreads();
    for (unsigned s = 128; s > 0; s >>= 1) {
        if (t < s) sm[t] = fp_add(sm[t], sm[t + s]);
        __syncthreads();
    }
    if (t == 0) b.out[blockIdx.y][blockIdx.x] = sm[0];
}
struct EvalItem {
    const Fr* p;
    size_t n;
    Fr* d_result;
};
inline void poly_eval_many(swm_ctx* ctx, const std::vector<EvalItem>& items, const EvalPoint& ep) {
    std::vector<EvalItem> cur;
    for (const EvalItem& it : items) {
        if (it.n == 0) hip_check(ctx, hipMemsetAsync(it.d_result, 0, sizeof(Fr), ctx->stream), "memset");
        else cur.push_back(it);
    }
    std::vector<DVec> keep;  // intermediate tile sums stay alive until the last level has been enqueued
    for (int level = 0; !cur.empty(); level++) {
        if (level >= 3) throw MarlinError(SWM_ERR_INTERNAL, "poly_eval: polynomial too long");
        size_t total_tiles = 0;
        for (const EvalItem& it : cur)
            if (it.n > (size_t)EVAL_TILE) total_tiles += (it.n + EVAL_TILE - 1) / EVAL_TILE;
        DVec tmp;
        if (total_tiles) tmp = DVec(ctx, total_tiles);
        size_t used = 0;
        std::vector<EvalItem> next;
        for (size_t lo = 0; lo < cur.size(); lo += EVAL_MANY) {
            const size_t k = std::min<size_t>(EVAL_MANY, cur.size() - lo);
            EvalBatch b;
            memset(&b, 0, sizeof(b));
            size_t max_tiles = 1;
            for (size_t j = 0; j < k; j++) {
                const EvalItem& it = cur[lo + j];
                const size_t ntiles = (it.n + EVAL_TILE - 1) / EVAL_TILE;
                b.c[j] = it.p;
                b.n[j] = it.n;
                if (ntiles > 1) {
                    b.out[j] = tmp.p + used;
                    next.push_back({tmp.p + used, ntiles, it.d_result});
                    used += ntiles;
                } else {
                    b.out[j] = it.d_result;
                }
                max_tiles = std::max(max_tiles, ntiles);
            }
            prof_begin(ctx, "poly_eval");
            hipLaunchKernelGGL(eval_chunks_many, dim3((unsigned)max_tiles, (unsigned)k), dim3(256), 0, ctx->stream, b,
                               ep.y256[level], ep.pw[level].p);
            prof_end(ctx);
            hip_check(ctx, hipGetLastError(), "poly_eval");
        }
        if (total_tiles) keep.push_back(std::move(tmp));
        cur.swap(next);
    }
}
// returns p(x) on the host (synchronises)
inline Fr poly_eval(swm_ctx* ctx, const Fr* p, size_t n, Fr x) {
    EvalPoint ep = eval_point(ctx, x);
    DVec r(ctx, 1);
    poly_eval_async(ctx, p, n, ep, r.p);
    return r.download(0, 1)[0];
}

// ------------------------------------------------------------------------------------------------ u32 exclusive scan
static constexpr int SC_BLOCK = 256, SC_ITEMS = 8, SC_TILE = SC_BLOCK * SC_ITEMS;
__device__ __forceinline__ uint32_t block_scan_incl(uint32_t v, uint32_t* sm) {
    uint32_t tid = threadIdx.x;
    sm[tid] = v;
    __syncthreads();
    for (uint32_t d = 1; d < SC_BLOCK; d <<= 1) {
        uint32_t t = tid >= d ? sm[tid - d] : 0;
        __syncthreads();
        sm[tid] += t;
        __syncthreads();
    }
    return sm[tid];
}
static __global__ void __launch_bounds__(SC_BLOCK) scan_totals(const uint32_t* in, size_t n, uint32_t* tot) {
    SWM_LIGHT_KERNEL();
    __shared__ uint32_t sm[SC_BLOCK];
    size_t lo = (size_t)blockIdx.x * SC_TILE + threadIdx.x * SC_ITEMS;
    uint32_t a = 0;
    for (size_t i = lo; i < lo + SC_ITEMS && i < n; i++) a += in[i];
    uint32_t inc = block_scan_incl(a, sm);
    if (threadIdx.x == SC_BLOCK - 1) tot[blockIdx.x] = inc;
}
static __global__ void __launch_bounds__(SC_BLOCK) scan_mid(uint32_t* tot, uint32_t ntiles) {
    SWM_LIGHT_KERNEL();
    __shared__ uint32_t sm[SC_BLOCK];
    uint32_t per = (ntiles + SC_BLOCK - 1) / SC_BLOCK;
    uint32_t lo = threadIdx.x * per, hi = min(lo + per, ntiles);
    uint32_t a = 0;
    for (uint32_t i = lo; i < hi; i++) a += tot[i];
    uint32_t inc = block_scan_incl(a, sm);
    uint32_t run = inc - a;
    for (uint32_t i = lo; i < hi; i++) {
        uint32_t c = tot[i];
        tot[i] = run;
        run += c;
    }
    if (threadIdx.x == SC_BLOCK - 1) tot[ntiles] = inc;
}
static __global__ void __launch_bounds__(SC_BLOCK) scan_final(const uint32_t* in, size_t n, const uint32_t* tot, uint32_t* out) {
    SWM_LIGHT_KERNEL();
    __shared__ uint32_t sm[SC_BLOCK];
    size_t lo = (size_t)blockIdx.x * SC_TILE + threadIdx.x * SC_ITEMS;
    uint32_t a = 0;
    for (size_t i = lo; i < lo + SC_ITEMS && i < n; i++) a += in[i];
    uint32_t inc = block_scan_incl(a, sm);
    uint32_t run = tot[blockIdx.x] + inc - a;
    for (size_t i = lo; i < lo + SC_ITEMS && i < n; i++) {
        out[i] = run;
        run += in[i];
    }
}
// out[i] = sum_{j < i} in[j]; *total (device word at tot[ntiles]) is returned through `total_out` on the host
inline uint32_t scan_exclusive_u32(swm_ctx* ctx, const uint32_t* in, uint32_t* out, size_t n) {
    unsigned ntiles = (unsigned)((n + SC_TILE - 1) / SC_TILE);
    DBuf<uint32_t> tot(ctx, ntiles + 1);
    hipLaunchKernelGGL(scan_totals, dim3(ntiles), dim3(SC_BLOCK), 0, ctx->stream, in, n, tot.p);
    hipLaunchKernelGGL(scan_mid, dim3(1), dim3(SC_BLOCK), 0, ctx->stream, tot.p, ntiles);
    hipLaunchKernelGGL(scan_final, dim3(ntiles), dim3(SC_BLOCK), 0, ctx->stream, in, n, tot.p, out);
    hip_check(ctx, hipGetLastError(), "scan");
    return tot.download(ntiles, 1)[0];
}

// ------------------------------------------------------------------------------------------------ bulk Fr sampling
// Candidate j = keystream words [pos + 8 j, pos + 8 j + 8) as 4 little-endian u64 limbs with the top 3 bits
// cleared; accepted when < r (ark_ff UniformRand for Fp256).  The accepted limbs are the Montgomery
// representation, so they are written to the coefficient vector as they are.
struct ChaChaKey {
    uint32_t k[8];
};
static __global__ void __launch_bounds__(256) sample_candidates(ChaChaKey key, int rounds, uint64_t pos, size_t m,
                                                         Fr* __restrict__ cand, uint32_t* __restrict__ flag) {
    SWM_LIGHT_KERNEL();
    size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (j >= m) return;
    uint64_t w0 = pos + 8 * j;
    uint32_t blk[16], blk2[16];
    chacha_block(key.k, w0 >> 4, rounds, blk);
    unsigned off = (unsigned)(w0 & 15);
    bool two = off + 8 > 16;
    if (two) chacha_block(key.k, (w0 >> 4) + 1, rounds, blk2);
    Fr r;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        unsigned idx = off + i;
        r.v[i] = idx < 16 ? blk[idx & 15] : blk2[idx & 15];
    }
    r.v[7] &= 0xffffffffu >> 3;
    bool lt = false, decided = false;
#pragma unroll
    for (int i = 7; i >= 0; i--) {
        if (!decided && r.v[i] != FrParams::P[i]) {
            lt = r.v[i] < FrParams::P[i];
            decided = true;
        }
    }
    cand[j] = r;
    flag[j] = lt ? 1u : 0u;
}
static __global__ void __launch_bounds__(256) sample_compact(const Fr* __restrict__ cand, const uint32_t* __restrict__ flag,
                                                      const uint32_t* __restrict__ rank, size_t m, size_t need,
                                                      Fr* __restrict__ out, uint32_t* __restrict__ last_idx) {
    SWM_LIGHT_KERNEL();
    size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (j >= m || !flag[j]) return;
    uint32_t r = rank[j];
    if (r < need) {
        out[r] = cand[j];
        if (r + 1 == need) *last_idx = (uint32_t)j;
    }
}
// candidates uploaded as raw bytes (caller-owned generator): clear the top bits in place, flag the ones below r
static __global__ void __launch_bounds__(256) sample_flag_raw(Fr* __restrict__ cand, size_t m, uint32_t* __restrict__ flag) {
    SWM_LIGHT_KERNEL();
    size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (j >= m) return;
    Fr r = cand[j];
    r.v[7] &= 0xffffffffu >> 3;
    bool lt = false, decided = false;
#pragma unroll
    for (int i = 7; i >= 0; i--) {
        if (!decided && r.v[i] != FrParams::P[i]) {
            lt = r.v[i] < FrParams::P[i];
            decided = true;
        }
    }
    cand[j] = r;
    flag[j] = lt ? 1u : 0u;
}
// compaction of a run of the caller-owned draw: the run's accepted candidates go behind the *base accepted before it (a device
// word: the host does not count the runs it knows cannot complete the draw), as long as they are among the first `need`
static __global__ void __launch_bounds__(256) sample_compact_base(const Fr* __restrict__ cand, const uint32_t* __restrict__ flag,
                                                           const uint32_t* __restrict__ rank, size_t m, size_t need,
                                                           Fr* __restrict__ out, const uint32_t* __restrict__ base) {
    SWM_LIGHT_KERNEL();
    size_t j = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (j >= m || !flag[j]) return;
    const size_t r = (size_t)*base + rank[j];
    if (r < need) out[r] = cand[j];
}
static __global__ void sample_base_add(uint32_t* __restrict__ base, const uint32_t* __restrict__ run_total) { *base += *run_total; }
// accepted candidates among n raw 32-byte draws (Fr::rand: the top three bits masked off, accepted when below the modulus)
inline size_t ext_count_accepted(const uint32_t* buf, size_t n) {
    size_t acc = 0;
    const uint32_t top_mask = 0xffffffffu >> 3, p7 = FrParams::P[7];
    for (size_t i = 0; i < n; i++) {  // little-endian host: 8 x u32 limbs, low first; decided by the top limb
        const uint32_t* r = buf + 8 * i;  // in all but 2^-29 of the cases
        const uint32_t t = r[7] & top_mask;
        if (t != p7) {
            acc += t < p7;
        } else {
            bool lt = false;
            for (int k = 6; k >= 0; k--) {
                if (r[k] < FrParams::P[k]) { lt = true; break; }
                if (r[k] > FrParams::P[k]) break;
            }
            acc += lt;
        }
    }
    return acc;
}
// Draws `need` field elements from rng's stream into out[0..need) (device), advancing rng exactly as `need`
// successive Fr::rand(rng) calls would.
// Caller-owned generator (rng.ext): the candidates come from its fill_bytes in runs of at most EXT_CHUNK and never more than
// the number still missing, so the stream stops right behind the candidate that completes the draw.  The host only COUNTS
// the accepted candidates of a run when the run could complete the draw (r05: as long as even a fully accepted run cannot, whole
// runs go up uncounted — the GPU's scan keeps the running total in a device word the compaction reads its offset from — and
// that word follows the host asynchronously, one pinned copy per run: the host's counting shrinks to the last few runs of a
// draw, ~2 ms of a 3 x 2^20 draw, and it never waits for the device); the raw run goes up from a ring of
// two host chunks on the context's copy stream and is flagged, scanned and compacted into place on the GPU (the kernels of the
// built-in path), while the callback produces the next run: ~170 MB through the callback for 3 * 2^20 elements, i.e. as
// fast as the caller's generator.  The call returns once the callback is done; the context's stream is made to wait for the
// last run.  `marked`: sample_fr_ext_mark was called when `out` was allocated — the transfers then wait for that point of
// the context's stream, not for what was enqueued since (the prover requests the mask after enqueueing the rest of round 1,
// so that the GPU works while the host draws).
static constexpr size_t EXT_CHUNK = (size_t)1 << 18;
// Host chunks in the ring.  Four, not two (r05): the copy of a run is a kernel of the runtime, and beside the accumulations of
// round 1 it is starved like every kernel without an issue priority — the first copy of a 3 x 2^20 draw completed 5 ms after
// the draw began and the host, two runs ahead, waited 2.1 ms for its chunk.  With four chunks the copies lag and catch up
// once the accumulations are through (SWM_EXT_RING=2: the ring of r02 - r04).
static constexpr int EXT_RING = 4;
static constexpr int EXT_TOTALS = 8;  // device totals in flight (pinned words + events), a power of two
inline void sample_fr_ext_mark(swm_ctx* ctx) {  // call right after allocating the destination of a later bulk draw
    if (!ctx->ext_event[EXT_RING]) hip_check(ctx, hipEventCreateWithFlags(&ctx->ext_event[EXT_RING], hipEventDisableTiming), "event");
    hip_check(ctx, hipEventRecord(ctx->ext_event[EXT_RING], ctx->stream), "record");
}
// `progress` (caller-owned generator only): called with the number of elements known to be in place whenever the host learns it
// exactly (a read-back of the device's running total, or a run it counted itself) — every kernel that writes those elements has
// been enqueued on ctx->copy_stream by then.  The prover commits to the mask polynomial piece by piece from it.
inline void sample_fr_bulk(swm_ctx* ctx, ChaChaRng& rng, Fr* out, size_t need, bool marked = false,
                           const std::function<void(size_t)>* progress = nullptr) {
    if (rng.ext) {
        if (!ctx->copy_stream) hip_check(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking), "copy stream");
        // (the chunks are ordinary cacheable host memory: a host ChaCha fills pinned memory at a fifth of the rate it fills
        // malloc'd memory on the bench host — 1.1 against 5.2 GB/s — and the runtime's staged copy of 8 MB is fast)
        // r05: ... and REGISTERED with the runtime (hipHostRegister pins the pages where they are: still cacheable for the host's
        // writes, but the copy engine reads them directly).  From unregistered memory hipMemcpyAsync is a synchronous staged copy:
        // ~0.35 ms per 8-MB run in which the host neither draws nor returns — 7 ms of a 3 x 2^20 draw that no counter showed
        // (round 1 took 50 ms with a 35-ms draw).  SWM_EXT_REGISTER=0: the unregistered ring.
        if (!ctx->ext_pinned) {
            void* ring = nullptr;
            if (posix_memalign(&ring, 4096, EXT_RING * EXT_CHUNK * sizeof(Fr)) != 0 || !ring) throw MarlinError(SWM_ERR_OOM, "sample: host ring");
            ctx->ext_pinned = ring;
            static constexpr bool reg = true;
            if (reg && hipHostRegister(ring, EXT_RING * EXT_CHUNK * sizeof(Fr), hipHostRegisterDefault) != hipSuccess) (void)hipGetLastError();  // (not fatal: staged copies)
            else ctx->ext_registered = reg;
        }
        for (int i = 0; i < EXT_RING; i++)
            if (!ctx->ext_event[i]) hip_check(ctx, hipEventCreateWithFlags(&ctx->ext_event[i], hipEventDisableTiming), "event");
        // device side of the ring: raw candidates, flags, ranks and the scan's tile totals, per slot
        const unsigned ntiles = (unsigned)((EXT_CHUNK + SC_TILE - 1) / SC_TILE);
        const size_t slot_bytes = (EXT_CHUNK * (sizeof(Fr) + 8) + ((size_t)ntiles + 2) * 4 + 255) & ~(size_t)255;
        char* dev = nullptr;
        if (scratch(ctx, "ext.ring", 2 * slot_bytes + 256, (void**)&dev) != SWM_OK) throw MarlinError(SWM_ERR_OOM, "sample: device ring");
        if (!marked) sample_fr_ext_mark(ctx);  // no earlier mark: everything enqueued so far may still use the buffer
        hipStream_t cs = ctx->copy_stream;
        hip_check(ctx, hipStreamWaitEvent(cs, ctx->ext_event[EXT_RING], 0), "wait");
        // have: accepted candidates the host knows of exactly; unc: candidates uploaded since whose acceptance only the device knows
        size_t have = 0, unc = 0;
        uint32_t* d_base = reinterpret_cast<uint32_t*>(dev + 2 * slot_bytes);
        hip_check(ctx, zero_fill_async(d_base, 4, cs), "clear");  // (the runtime's fill kernel waited 2 ms beside round 1's accumulations: fill.cuh)
        static const bool trace = env_flag("SWM_TRACE");
        static constexpr bool count_all = false;  // (true: the r02 - r04 behaviour, every run counted on the host: + 3.4 ms at 2^20, r05)
        double t_cb = 0, t_count = 0, t_wait = 0;
        int readbacks = 0;
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
            return std::chrono::duration<double, std::milli>(b - a).count();
        };
        // What the host knows of the accepted count (second half of r05): behind every run the device's running total is copied
        // into one of eight pinned words and an event recorded; before a run the host takes the NEWEST total that has arrived
        // (hipEventQuery: no waiting) and treats the runs behind it as uncounted.  Where a run could complete the draw it counts
        // the uncounted runs itself — the last two are still in the host ring, hot in its cache — and waits for the device only
        // for older ones (never seen: the device lags by less than a run).  The first half of the round read the total back with
        // a stream synchronisation where the bound got close: 3 - 4 round trips behind whatever the copy stream's kernels were
        // waiting for, 3.1 ms of a 3 x 2^20 draw (SWM_EXT_READBACK=1).  A helper thread counting beside the callback was tried
        // and dropped: the callback slowed from 33 to 48 ms (every line it writes is shared with the helper's core).
        static constexpr bool readback = false;
        if (!ctx->ext_totals) hip_check(ctx, hipHostMalloc((void**)&ctx->ext_totals, EXT_TOTALS * sizeof(uint32_t), hipHostMallocDefault), "totals");
        for (auto& e : ctx->ext_cnt_event)
            if (!e) hip_check(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming), "event");
        volatile uint32_t* totals = ctx->ext_totals;
        constexpr size_t UNKNOWN = ~(size_t)0;
        std::vector<float> ring_waits;  // (trace) per run: how long the host waited for its chunk of the ring
        long j = 0, dev_upto = -1;  // runs uploaded; the newest run whose device total the host has
        size_t dev_total = 0, told = 0;
        size_t cands4[EXT_TOTALS] = {}, accm4[EXT_TOTALS] = {};  // per run (index mod 8): candidates, accepted as counted here
        int buf4[EXT_TOTALS] = {};                                   // ... and the host chunk it was drawn into
        auto take_total = [&](long i, bool wait) {  // the device's total behind run i
            if (wait) hip_check(ctx, hipEventSynchronize(ctx->ext_cnt_event[i & (EXT_TOTALS - 1)]), "total");
            dev_upto = i;
            dev_total = totals[i & (EXT_TOTALS - 1)];
        };
        static constexpr long ring = EXT_RING;  // (two chunks instead of four: the host waited 2.1 ms for the first run's copy, r05)
        for (;;) {
            const int hb = (int)(j % ring), b = (int)(j & 1);  // host chunk and device slot of the next run
            if (!readback) {
                for (long i = j - 1; i > dev_upto && i + EXT_TOTALS >= j; i--)
                    if (hipEventQuery(ctx->ext_cnt_event[i & (EXT_TOTALS - 1)]) == hipSuccess) {
                        take_total(i, false);
                        break;
                    }
                have = dev_total;
                unc = 0;
                for (long k = dev_upto + 1; k < j; k++) {
                    if (accm4[k & (EXT_TOTALS - 1)] != UNKNOWN) have += accm4[k & (EXT_TOTALS - 1)];
                    else unc += cands4[k & (EXT_TOTALS - 1)];
                }
                if (progress && std::min(have, need) > told) (*progress)(told = std::min(have, need));
            }
            // a whole run cannot complete the draw even if every candidate of it (and of the uncounted runs before it) is accepted
            const bool blind = !count_all && have + unc + EXT_CHUNK <= need;
            if (!blind && unc) {  // close to the end with uncounted runs behind us
                auto t0 = now();
                if (readback) {  // ask the device how many it has accepted
                    uint32_t h = 0;
                    hip_check(ctx, hipMemcpyAsync(&h, d_base, 4, hipMemcpyDeviceToHost, cs), "d2h");
                    hip_check(ctx, hipStreamSynchronize(cs), "sync");
                    have = h;
                    unc = 0;
                    t_wait += ms(t0, now());
                    if (progress) (*progress)(told = std::min(have, need));
                } else {
                    // (the runs j - ring .. j - 1 are still in the host ring; older uncounted ones: the device's word)
                    if (dev_upto + 1 < j - ring) take_total(j - ring - 1, true);
                    auto t1 = now();
                    for (long k = std::max(dev_upto + 1, j - ring); k < j; k++)
                        if (accm4[k & (EXT_TOTALS - 1)] == UNKNOWN)
                            accm4[k & (EXT_TOTALS - 1)] = ext_count_accepted(reinterpret_cast<const uint32_t*>((char*)ctx->ext_pinned + (size_t)buf4[k & (EXT_TOTALS - 1)] * EXT_CHUNK * sizeof(Fr)),
                                                              cands4[k & (EXT_TOTALS - 1)]);
                    t_wait += ms(t0, t1);
                    t_count += ms(t1, now());
                }
                readbacks++;
                continue;  // (this trip drew nothing)
            }
            if (!blind && have >= need) break;
            const size_t want = blind ? EXT_CHUNK : std::min(need - have, EXT_CHUNK);
            uint32_t* buf = reinterpret_cast<uint32_t*>((char*)ctx->ext_pinned + (size_t)hb * EXT_CHUNK * sizeof(Fr));
            auto t0 = now();
            // (the first trips of a draw as well: with the ring registered the copies are asynchronous, and the last run of the
            // PREVIOUS draw — the piece before this one, or the proof before — may still be reading the chunk)
            hip_check(ctx, hipEventSynchronize(ctx->ext_event[hb]), "ring");
            if (!readback && j >= EXT_TOTALS && dev_upto < j - EXT_TOTALS) take_total(j - EXT_TOTALS, true);  // its word and event are about to be reused
            auto t1 = now();
            rng.ext(rng.ext_user, reinterpret_cast<uint8_t*>(buf), want * 32);
            auto t2 = now();
            t_wait += ms(t0, t1);
            t_cb += ms(t1, t2);
            if (trace) ring_waits.push_back((float)ms(t0, t1));
            size_t counted = UNKNOWN;
            if (blind) {
                unc += want;
            } else {
                counted = ext_count_accepted(buf, want);
                have += counted;
                t_count += ms(t2, now());
            }
            cands4[j & (EXT_TOTALS - 1)] = want;
            accm4[j & (EXT_TOTALS - 1)] = counted;
            buf4[j & (EXT_TOTALS - 1)] = hb;
            Fr* d_raw = reinterpret_cast<Fr*>(dev + (size_t)b * slot_bytes);
            uint32_t* d_flag = reinterpret_cast<uint32_t*>(d_raw + EXT_CHUNK);
            uint32_t* d_rank = d_flag + EXT_CHUNK;
            uint32_t* d_tot = d_rank + EXT_CHUNK;
            hip_check(ctx, hipMemcpyAsync(d_raw, buf, want * sizeof(Fr), hipMemcpyHostToDevice, cs), "h2d");
            hip_check(ctx, hipEventRecord(ctx->ext_event[hb], cs), "record");  // the host chunk may be refilled after the copy
            const unsigned grid = (unsigned)((want + 255) / 256), nt = (unsigned)((want + SC_TILE - 1) / SC_TILE);
            hipLaunchKernelGGL(sample_flag_raw, dim3(grid), dim3(256), 0, cs, d_raw, want, d_flag);
            hipLaunchKernelGGL(scan_totals, dim3(nt), dim3(SC_BLOCK), 0, cs, (const uint32_t*)d_flag, want, d_tot);
            hipLaunchKernelGGL(scan_mid, dim3(1), dim3(SC_BLOCK), 0, cs, d_tot, nt);
            hipLaunchKernelGGL(scan_final, dim3(nt), dim3(SC_BLOCK), 0, cs, (const uint32_t*)d_flag, want, d_tot, d_rank);
            hipLaunchKernelGGL(sample_compact_base, dim3(grid), dim3(256), 0, cs, (const Fr*)d_raw, (const uint32_t*)d_flag,
                               (const uint32_t*)d_rank, want, need, out, (const uint32_t*)d_base);
            hipLaunchKernelGGL(sample_base_add, dim3(1), dim3(1), 0, cs, d_base, (const uint32_t*)(d_tot + nt));  // the scan's total of the run
            hip_check(ctx, hipGetLastError(), "sample (caller-owned generator)");
            if (!readback) {
                hip_check(ctx, hipMemcpyAsync((void*)&totals[j & (EXT_TOTALS - 1)], d_base, 4, hipMemcpyDeviceToHost, cs), "total");
                hip_check(ctx, hipEventRecord(ctx->ext_cnt_event[j & (EXT_TOTALS - 1)], cs), "record");
            }
            j++;
            if (progress && !blind) (*progress)(told = std::max(told, std::min(have, need)));
        }
        hip_check(ctx, hipEventRecord(ctx->ext_event[EXT_RING], cs), "record");
        hip_check(ctx, hipStreamWaitEvent(ctx->stream, ctx->ext_event[EXT_RING], 0), "wait");
        if (trace)
            fprintf(stderr, "[swm trace]   bulk draw of %zu elements from the caller's generator: callback %.1f ms, counting %.1f ms, "
                            "waiting for the ring and %d read-backs %.1f ms\n", need, t_cb, t_count, readbacks, t_wait);
        if (trace) {
            fprintf(stderr, "[swm trace]   ring waits per run (ms):");
            for (float w : ring_waits) fprintf(stderr, " %.2f", w);
            fprintf(stderr, "\n");
        }
        // the device ring is reused by the next draw: its kernels are ordered behind these on the copy stream
        return;
    }
    size_t done = 0;
    while (done < need) {
        size_t want = need - done;
        size_t m = (size_t)((double)want / 0.58 * 1.02) + 2048;
        // test hook: only as many candidates as elements still missing, so that the retry branch below runs several times
        // (the acceptance rate is r / 2^253 = 0.58) — tests/test_gpu_marlin.py pins it against the sequential stream
        if (env_flag("SWM_SAMPLE_TIGHT")) m = want;
        DVec cand(ctx, m);
        DBuf<uint32_t> flag(ctx, m), rank(ctx, m), last(ctx, 1);
        ChaChaKey key;
        for (int i = 0; i < 8; i++) key.k[i] = rng.key[i];
        unsigned grid = (unsigned)((m + 255) / 256);
        prof_begin(ctx, "sample_fr");
        hipLaunchKernelGGL(sample_candidates, dim3(grid), dim3(256), 0, ctx->stream, key, rng.rounds, rng.pos, m, cand.p,
                           flag.p);
        prof_end(ctx);
        uint32_t accepted = scan_exclusive_u32(ctx, flag.p, rank.p, m);
        size_t take = accepted < want ? accepted : want;
        if (take) {
            hipLaunchKernelGGL(sample_compact, dim3(grid), dim3(256), 0, ctx->stream, cand.p, flag.p, rank.p, m, take,
                               out + done, last.p);
            hip_check(ctx, hipGetLastError(), "sample_compact");
        }
        if (accepted >= want) {
            uint32_t li = last.download(0, 1)[0];
            rng.pos += 8ull * ((uint64_t)li + 1);
        } else {
            rng.pos += 8ull * m;
        }
        rng.have = false;
        done += take;
    }
}

}  // namespace swm
