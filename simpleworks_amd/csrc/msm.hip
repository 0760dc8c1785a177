// msm.hip — K1: G1 multi-scalar multiplication on BLS12-377 for gfx950 (MI355X).
//
// Replaces ark_ec::msm::VariableBaseMSM::multi_scalar_mul (ark-ec 0.3.0), reached from
// /root/reference/src/marlin/mod.rs:75,92 through KZG10::commit/open.  arkworks runs an unsigned-window Pippenger
// with one rayon task per window; the MI355X design is different on purpose (the sum is a canonical group element,
// so only the result has to agree):
//
//   1. msm_count    one thread per scalar: Montgomery -> standard form if needed, signed-digit recoding into
//                   W = ceil(254/c) windows (digits in [-2^(c-1), 2^(c-1)] halve the bucket count), histogram
//                   with global atomics.                                                       HBM: n x 32 B read
//   2. msm_scan     single-workgroup exclusive scan of the W*2^(c-1) bucket counts -> bucket offsets, plus a
//                   balanced split of every bucket into segments of <= SEG points (huge buckets: ~sqrt(count)
//                   segments) -> segment offsets.  Keeps lanes of a wave on equal-length work and bounds the
//                   serial chain for structured scalars (all-ones witnesses).
//   3. msm_scatter  recompute digits, counting-sort (point index | sign) by bucket.              HBM: n*W*4 B write
//   4. msm_accumulate  one lane per segment: gather 96-B affine bases through L2 / Infinity Cache, XYZZ mixed
//                   additions (8M+2S, carry-chain integer VALU, no MFMA) -> one partial per segment.
//                   This is the dominant kernel: n*W mixed adds.
//   5. msm_bucket_sum  one lane per bucket: fold that bucket's segment partials.
//   6. msm_window_reduce  per window sum_b (b+1)*S_b: lanes take 8 consecutive buckets (local running sum +
//                   small scalar multiple for the chunk offset), LDS tree across the workgroup.
//   7. host         sums the per-workgroup partials and does the W-term Horner fold (c doublings per window):
//                   a 250-step serial dependency chain belongs on a CPU core, not on a 64-wide SIMD.
//
// Algorithmic bytes per point: 96 B base + 32 B scalar = 128 B.
#include "context.h"
#include "g1.cuh"

namespace swm {

// ---------------------------------------------------------------------------------------------- parameters
static constexpr int SEG = 32;        // target points per accumulation segment
static constexpr int RED_CHUNK = 8;   // buckets per lane in the window reduction
static constexpr int RED_BLOCK = 256;

struct MsmPlan {
    unsigned c;       // window bits
    unsigned nwin;    // number of windows
    uint32_t B;       // buckets per window = 2^(c-1)
    uint32_t NB;      // total buckets
};

MsmPlan msm_plan(size_t n) {
    // Window choice for the GPU schedule (NOT arkworks' ln-based rule): large enough that the n*W accumulate
    // adds dominate the fixed-latency bucket reduction, small enough that buckets stay populated.
    unsigned c;
    if (n <= (1u << 8)) c = 6;
    else if (n <= (1u << 11)) c = 8;
    else if (n <= (1u << 14)) c = 10;
    else if (n <= (1u << 16)) c = 12;
    else if (n <= (1u << 18)) c = 13;
    else if (n <= (1u << 20)) c = 14;
    else if (n <= (1u << 22)) c = 15;
    else c = 16;
    MsmPlan p;
    p.c = c;
    p.nwin = (254 + c - 1) / c;
    p.B = 1u << (c - 1);
    p.NB = p.nwin * p.B;
    return p;
}

// ---------------------------------------------------------------------------------------------- digits
// Signed-digit recoding of a 253-bit standard-form scalar; calls f(window, bucket_index, negative) for every
// non-zero digit.  Digit d in [-2^(c-1), 2^(c-1)]; bucket index |d| - 1.
template <class Fn>
__device__ __forceinline__ void for_each_digit(const Fr& s, unsigned c, unsigned nwin, Fn f) {
    uint32_t carry = 0;
    const uint32_t mask = (1u << c) - 1;
    const uint32_t half = 1u << (c - 1);
    for (unsigned w = 0; w < nwin; w++) {
        unsigned bit = w * c;
        unsigned limb = bit >> 5, off = bit & 31;
        uint32_t v = limb < 8 ? (s.v[limb] >> off) : 0;
        if (off + c > 32 && limb + 1 < 8) v |= s.v[limb + 1] << (32 - off);
        uint32_t d = (v & mask) + carry;
        if (d > half) {
            carry = 1;
            if (d != (1u << c)) f(w, (1u << c) - d - 1, true);  // digit d - 2^c < 0 (0 when d == 2^c), |digit| - 1
        } else {
            carry = 0;
            if (d != 0) f(w, d - 1, false);
        }
    }
}

__device__ __forceinline__ Fr load_scalar(const Fr* scalars, size_t i, int mont) {
    Fr s = scalars[i];
    if (mont) s = fp_to_std(s);
    return s;
}

__global__ void __launch_bounds__(256) msm_count(const Fr* __restrict__ scalars, size_t n, int mont, unsigned c,
                                                 unsigned nwin, uint32_t B, uint32_t* __restrict__ hist) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        Fr s = load_scalar(scalars, i, mont);
        if (fp_is_zero(s)) continue;
        for_each_digit(s, c, nwin, [&](unsigned w, uint32_t b, bool) { atomicAdd(&hist[w * B + b], 1u); });
    }
}

__global__ void __launch_bounds__(256) msm_scatter(const Fr* __restrict__ scalars, size_t n, int mont, unsigned c,
                                                   unsigned nwin, uint32_t B, const uint32_t* __restrict__ bucket_off,
                                                   uint32_t* __restrict__ cursor, uint32_t* __restrict__ sorted) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        Fr s = load_scalar(scalars, i, mont);
        if (fp_is_zero(s)) continue;
        for_each_digit(s, c, nwin, [&](unsigned w, uint32_t b, bool neg) {
            uint32_t g = w * B + b;
            uint32_t pos = bucket_off[g] + atomicAdd(&cursor[g], 1u);
            sorted[pos] = (uint32_t)i | (neg ? 0x80000000u : 0u);
        });
    }
}

__device__ __forceinline__ uint32_t nseg_of(uint32_t cnt) {
    if (cnt == 0) return 0;
    if (cnt <= (uint32_t)SEG * SEG) return (cnt + SEG - 1) / SEG;
    // ~sqrt(cnt) segments of ~sqrt(cnt) points: bounds the serial chain of a pathological bucket
    uint32_t r = (uint32_t)sqrtf((float)cnt);
    while ((uint64_t)r * r < cnt) r++;
    return r;
}

// Single workgroup (1024 lanes): exclusive scans of bucket counts and of per-bucket segment counts.
__global__ void __launch_bounds__(1024) msm_scan(const uint32_t* __restrict__ hist, uint32_t NB,
                                                 uint32_t* __restrict__ bucket_off, uint32_t* __restrict__ seg_off) {
    __shared__ uint32_t s_cnt[1024];
    __shared__ uint32_t s_seg[1024];
    uint32_t tid = threadIdx.x;
    uint32_t per = (NB + 1023) / 1024;
    uint32_t lo = tid * per, hi = min(lo + per, NB);
    uint32_t a = 0, b = 0;
    for (uint32_t i = lo; i < hi; i++) {
        a += hist[i];
        b += nseg_of(hist[i]);
    }
    s_cnt[tid] = a;
    s_seg[tid] = b;
    __syncthreads();
    // Hillis-Steele inclusive scan over 1024 partials
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        uint32_t va = tid >= d ? s_cnt[tid - d] : 0, vb = tid >= d ? s_seg[tid - d] : 0;
        __syncthreads();
        s_cnt[tid] += va;
        s_seg[tid] += vb;
        __syncthreads();
    }
    uint32_t ra = s_cnt[tid] - a, rb = s_seg[tid] - b;  // exclusive prefix of this lane's chunk
    for (uint32_t i = lo; i < hi; i++) {
        bucket_off[i] = ra;
        seg_off[i] = rb;
        ra += hist[i];
        rb += nseg_of(hist[i]);
    }
    if (tid == 1023) {
        bucket_off[NB] = s_cnt[1023];
        seg_off[NB] = s_seg[1023];
    }
}

__global__ void __launch_bounds__(256) msm_segdesc(const uint32_t* __restrict__ bucket_off,
                                                   const uint32_t* __restrict__ seg_off, uint32_t NB,
                                                   uint32_t* __restrict__ seg_start, uint32_t* __restrict__ seg_end) {
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= NB) return;
    uint32_t s0 = seg_off[b], ns = seg_off[b + 1] - s0;
    uint32_t o = bucket_off[b], cnt = bucket_off[b + 1] - o;
    for (uint32_t s = 0; s < ns; s++) {
        seg_start[s0 + s] = o + (uint32_t)(((uint64_t)cnt * s) / ns);
        seg_end[s0 + s] = o + (uint32_t)(((uint64_t)cnt * (s + 1)) / ns);
    }
}

// Dominant kernel: one lane per segment, XYZZ accumulator in registers, affine bases gathered from HBM/L2.
__global__ void __launch_bounds__(256) msm_accumulate(const G1Affine* __restrict__ bases,
                                                      const uint32_t* __restrict__ sorted,
                                                      const uint32_t* __restrict__ seg_start,
                                                      const uint32_t* __restrict__ seg_end, uint32_t nseg,
                                                      G1XYZZ* __restrict__ partial) {
    uint32_t seg = blockIdx.x * blockDim.x + threadIdx.x;
    if (seg >= nseg) return;
    uint32_t k = seg_start[seg], e = seg_end[seg];
    G1XYZZ acc = g1_xyzz_identity();
    if (k < e) {
        uint32_t ent = sorted[k];
        G1Affine p = bases[ent & 0x7fffffffu];
        if (ent >> 31) p.y = fp_neg(p.y);
        acc = g1_from_affine(p);
        k++;
    }
    for (; k < e; k++) {
        uint32_t ent = sorted[k];
        G1Affine p = bases[ent & 0x7fffffffu];
        if (ent >> 31) p.y = fp_neg(p.y);
        g1_add_mixed(acc, p);
    }
    partial[seg] = acc;
}

__global__ void __launch_bounds__(256) msm_bucket_sum(const G1XYZZ* __restrict__ partial,
                                                      const uint32_t* __restrict__ seg_off, uint32_t NB,
                                                      G1XYZZ* __restrict__ buckets) {
    uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= NB) return;
    uint32_t s = seg_off[b], e = seg_off[b + 1];
    G1XYZZ acc = g1_xyzz_identity();
    if (s < e) acc = partial[s++];
    for (; s < e; s++) g1_add(acc, partial[s]);
    buckets[b] = acc;
}

// grid = (ceil(B / RED_CHUNK / RED_BLOCK), nwin).  out[w * gridDim.x + blockIdx.x] = partial of sum_b (b+1) S_b.
__global__ void __launch_bounds__(RED_BLOCK) msm_window_reduce(const G1XYZZ* __restrict__ buckets, uint32_t B,
                                                               G1XYZZ* __restrict__ out) {
    extern __shared__ __align__(16) unsigned char smem_raw[];
    G1XYZZ* sm = reinterpret_cast<G1XYZZ*>(smem_raw);
    uint32_t w = blockIdx.y;
    uint32_t chunk = blockIdx.x * RED_BLOCK + threadIdx.x;
    uint32_t lo = chunk * RED_CHUNK;
    G1XYZZ acc = g1_xyzz_identity();
    if (lo < B) {
        const G1XYZZ* base = buckets + (size_t)w * B;
        uint32_t hi = min(lo + RED_CHUNK, B);
        G1XYZZ run = g1_xyzz_identity();
        for (uint32_t b = hi; b-- > lo;) {
            g1_add(run, base[b]);
            g1_add(acc, run);
        }
        // acc = sum (b - lo + 1) S_b ; add lo * sum S_b
        if (lo) {
            G1XYZZ t = g1_mul_small(run, lo);
            g1_add(acc, t);
        }
    }
    sm[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t stride = RED_BLOCK / 2; stride > 0; stride >>= 1) {
        if (threadIdx.x < stride) {
            G1XYZZ a = sm[threadIdx.x];
            g1_add(a, sm[threadIdx.x + stride]);
            sm[threadIdx.x] = a;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[(size_t)w * gridDim.x + blockIdx.x] = sm[0];
}

// ---------------------------------------------------------------------------------------------- host driver
// d_scalars: n Fr in HBM.  Result: XYZZ on the host.
int msm_run(swm_ctx* ctx, const G1Affine* d_bases, const void* d_scalars, size_t n, int mont, G1XYZZ* result) {
    *result = g1_xyzz_identity();
    if (n == 0) return SWM_OK;
    if (n >= (1ull << 31)) return set_err(ctx, SWM_ERR_INVALID_ARG, "msm: n must be < 2^31");
    MsmPlan pl = msm_plan(n);
    const size_t total = n * (size_t)pl.nwin;
    if (total >= (1ull << 32)) return set_err(ctx, SWM_ERR_INVALID_ARG, "msm: n * windows must be < 2^32");
    uint32_t *hist, *bucket_off, *seg_off, *cursor, *sorted, *seg_start, *seg_end;
    SWM_TRY(scratch(ctx, "msm.hist", (pl.NB + 1) * 4ull * 2, (void**)&hist));
    cursor = hist + pl.NB + 1;
    SWM_TRY(scratch(ctx, "msm.bucket_off", (pl.NB + 1) * 4ull, (void**)&bucket_off));
    SWM_TRY(scratch(ctx, "msm.seg_off", (pl.NB + 1) * 4ull, (void**)&seg_off));
    SWM_TRY(scratch(ctx, "msm.sorted", total * 4, (void**)&sorted));
    SWM_HIP(ctx, hipMemsetAsync(hist, 0, (pl.NB + 1) * 4ull * 2, ctx->stream));
    const Fr* sc = reinterpret_cast<const Fr*>(d_scalars);
    unsigned grid_n = (unsigned)std::min<size_t>((n + 255) / 256, 256 * 16);
    SWM_LAUNCH(ctx, "msm_count", msm_count, dim3(grid_n), dim3(256), 0, sc, n, mont, pl.c, pl.nwin, pl.B, hist);
    SWM_LAUNCH(ctx, "msm_scan", msm_scan, dim3(1), dim3(1024), 0, hist, pl.NB, bucket_off, seg_off);
    SWM_LAUNCH(ctx, "msm_scatter", msm_scatter, dim3(grid_n), dim3(256), 0, sc, n, mont, pl.c, pl.nwin, pl.B,
               bucket_off, cursor, sorted);
    uint32_t nseg = 0;
    SWM_HIP(ctx, hipMemcpyAsync(&nseg, seg_off + pl.NB, 4, hipMemcpyDeviceToHost, ctx->stream));
    SWM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    G1XYZZ *partial, *buckets, *wpart;
    SWM_TRY(scratch(ctx, "msm.seg_start", (size_t)(nseg + 1) * 8, (void**)&seg_start));
    seg_end = seg_start + nseg + 1;
    SWM_TRY(scratch(ctx, "msm.partial", (size_t)(nseg + 1) * sizeof(G1XYZZ), (void**)&partial));
    SWM_TRY(scratch(ctx, "msm.buckets", (size_t)pl.NB * sizeof(G1XYZZ), (void**)&buckets));
    unsigned grid_b = (pl.NB + 255) / 256;
    SWM_LAUNCH(ctx, "msm_segdesc", msm_segdesc, dim3(grid_b), dim3(256), 0, bucket_off, seg_off, pl.NB, seg_start,
               seg_end);
    if (nseg)
        SWM_LAUNCH(ctx, "msm_accumulate", msm_accumulate, dim3((nseg + 255) / 256), dim3(256), 0, d_bases, sorted,
                   seg_start, seg_end, nseg, partial);
    SWM_LAUNCH(ctx, "msm_bucket_sum", msm_bucket_sum, dim3(grid_b), dim3(256), 0, partial, seg_off, pl.NB, buckets);
    unsigned red_blocks = (pl.B + RED_CHUNK * RED_BLOCK - 1) / (RED_CHUNK * RED_BLOCK);
    SWM_TRY(scratch(ctx, "msm.wpart", (size_t)pl.nwin * red_blocks * sizeof(G1XYZZ), (void**)&wpart));
    SWM_LAUNCH(ctx, "msm_window_reduce", msm_window_reduce, dim3(red_blocks, pl.nwin), dim3(RED_BLOCK),
               RED_BLOCK * sizeof(G1XYZZ), buckets, pl.B, wpart);
    std::vector<G1XYZZ> h((size_t)pl.nwin * red_blocks);
    SWM_HIP(ctx, hipMemcpyAsync(h.data(), wpart, h.size() * sizeof(G1XYZZ), hipMemcpyDeviceToHost, ctx->stream));
    SWM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    // host: fold workgroup partials, then Horner over windows (high -> low, c doublings each)
    G1XYZZ total_pt = g1_xyzz_identity();
    for (unsigned w = pl.nwin; w-- > 0;) {
        for (unsigned k = 0; k < pl.c; k++) total_pt = g1_dbl(total_pt);
        G1XYZZ ws = g1_xyzz_identity();
        for (unsigned j = 0; j < red_blocks; j++) g1_add(ws, h[(size_t)w * red_blocks + j]);
        g1_add(total_pt, ws);
    }
    *result = total_pt;
    return SWM_OK;
}

}  // namespace swm
