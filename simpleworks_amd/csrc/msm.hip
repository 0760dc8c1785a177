// msm.hip — K1: G1 multi-scalar multiplication on BLS12-377 for gfx950 (MI355X).
//
// Replaces ark_ec::msm::VariableBaseMSM::multi_scalar_mul (ark-ec 0.3.0), reached from
// /root/reference/src/marlin/mod.rs:75,92 through KZG10::commit/open.  arkworks runs an unsigned-window Pippenger
// with one rayon task per window; the MI355X design is different on purpose (the sum is a canonical group element,
// so only the result has to agree):
//
//   1. msm_digits   one lane per scalar: Montgomery -> standard form if needed, signed-digit recoding into
//                   W = ceil(254/c) windows (digits in [-2^(c-1), 2^(c-1)] halve the bucket count), written
//                   window-major so that the sort streams them coalesced.           HBM: 32 B read + 4W B write / point
//   2. msm_hist / msm_scan / msm_scatter   counting sort of (point index | sign) by (window, bucket): 64K-digit
//                   tiles histogram in LDS (LDS atomics), one global atomic per non-empty LDS bin, single-workgroup
//                   exclusive scan, then each tile reserves its run per bucket and ranks with LDS atomics.
//                   Every bucket is split into balanced segments of <= 32 points.
//   3. msm_accumulate  one lane per segment: gather 96-B affine bases through L2 / Infinity Cache, XYZZ mixed
//                   additions (8M+2S, carry-chain integer VALU, no MFMA) -> one partial per segment.
//                   This is the dominant kernel: n*W mixed adds.
//   4. msm_bucket_sum / msm_big_bucket_sum   fold a bucket's segment partials: one lane per ordinary bucket, a
//                   whole workgroup (strided sums + LDS tree) per bucket with > 16 segments, so structured scalars
//                   (all-ones witnesses) and a short top window do not serialise on one lane.
//   5. msm_window_reduce  per window sum_b (b+1)*S_b: lanes take 8 consecutive buckets (local running sum +
//                   small scalar multiple for the chunk offset), LDS tree across the workgroup.
//   6. host         sums the per-workgroup partials and does the W-term Horner fold (c doublings per window):
//                   a 250-step serial dependency chain belongs on a CPU core, not on a 64-wide SIMD.
//
// Algorithmic bytes per point: 96 B base + 32 B scalar = 128 B.
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include "context.h"
#include "fill.cuh"
#include "fq28.cuh"
#include "g1.cuh"
#include "msm.h"
#include <atomic>
#include <chrono>

namespace swm {

// ---------------------------------------------------------------------------------------------- parameters
static constexpr int SEG_MAX = 128;     // points per accumulation segment: upper bound, chosen per MSM (msm_seg_bound)
static constexpr int BIG_NSEG = 16;     // buckets with more segments than this are folded by a whole workgroup
static constexpr int RED_BLOCK = 256;
static constexpr uint32_t SORT_TILE_MIN = 65536;  // digits per workgroup in the LDS-privatised counting sort (at least)
static constexpr int SORT_THREADS = 1024;

WinLayout msm_plan(size_t n) {
    // Target window size for the GPU schedule (NOT arkworks' ln-based rule): large enough that the n*W accumulate
    // adds dominate the fixed-latency bucket reduction, small enough that buckets stay populated.
    // measured on MI355X (sweep over c = 8..16, r01): 2^10 -> 10, 2^12..2^16 -> 12, 2^18 -> 14, >= 2^19 -> 16.
    // c is capped at 16 by the LDS histogram of the counting sort (2^(c-1) words).
    unsigned c;
    if (n <= (1u << 11)) c = 10;
    else if (n <= 98304) c = 12;
    else if (n <= 393216) c = 14;
    else c = 16;
    WinLayout L;
    L.nwin = (254 + c - 1) / c;
    unsigned base = 254 / L.nwin, extra = 254 % L.nwin;
    unsigned bit = 0;
    L.NB = 0;
    L.maxB = 0;
    for (unsigned w = 0; w < L.nwin; w++) {
        unsigned cw = base + (w < extra ? 1 : 0);
        L.c[w] = (uint8_t)cw;
        L.bit[w] = (uint16_t)bit;
        L.boff[w] = L.NB;
        bit += cw;
        uint32_t B = 1u << (cw - 1);
        L.NB += B;
        if (B > L.maxB) L.maxB = B;
    }
    L.boff[L.nwin] = L.NB;
    return L;
}

unsigned msm_table_windows(unsigned c) { return (254 + c - 1) / c; }
// Windows of the flat schedule for a table of width c: ceil(254 / c) windows whose widths differ by at most one (as in
// msm_plan: a short top window would put a quarter of all points into two buckets), the widest being c bits; all of them
// index ONE set of 2^(c-1) buckets.  c[0] is always the widest window.
WinLayout msm_table_layout(unsigned c) {
    WinLayout L;
    memset(&L, 0, sizeof(L));
    L.nwin = msm_table_windows(c);
    const unsigned base = 254 / L.nwin, extra = 254 % L.nwin;
    unsigned bit = 0;
    for (unsigned w = 0; w < L.nwin; w++) {
        const unsigned cw = base + (w < extra ? 1 : 0);
        L.c[w] = (uint8_t)cw;
        L.bit[w] = (uint16_t)bit;
        L.boff[w] = 0;
        bit += cw;
    }
    L.NB = L.maxB = 1u << (L.c[0] - 1);
    L.boff[L.nwin] = L.NB;
    return L;
}
unsigned msm_table_width(size_t n_bases) {
    // SWM_MSM_TABLE_C: the width for every base set of the process (8 .. 22; tests/test_gpu_switches.py proves with 16 and 18)
    static const long forced = env_switch("SWM_MSM_TABLE_C", 0, 8, 22);
    if (forced) return (unsigned)forced;
    if (n_bases < 512) return 0;  // tiny base sets keep the per-window schedule
    // measured r02 (prove() at 2^10 .. 2^20 constraints, base sets of 3 x that): two bits above the size of the base
    // set up to 2^15 points — 4 .. 10 points per bucket for the MSMs of a proof, which keeps the accumulation chains
    // short — one bit above it from there, up to 20 (21 and 22 lose at 2^20 and above: the bucket stage grows faster
    // than the accumulation shrinks)
    unsigned lg = 0;
    while (((size_t)2 << lg) <= n_bases) lg++;
    return lg <= 14 ? lg + 2 : std::min(20u, lg + 1);  // (lg = 15, the key of a 2^14-constraint circuit: 17 until r04; 16: 4.73 instead of 4.98 ms per proof)
}

// ---------------------------------------------------------------------------------------------- digits
// Signed-digit recoding of a 253-bit standard-form scalar; calls f(window, bucket_index, negative) for every
// non-zero digit.  Digit d in [-2^(c-1), 2^(c-1)]; bucket index |d| - 1.
// (r06: the scalar is consumed as a shift register — the window's bits are the low bits of word 0, then the eight words move down by
// the window's width with funnel shifts — instead of being indexed by the window's bit position: the dynamic index made the compiler
// keep the scalar in LDS (promoted alloca), and msm_digits spent 76 % of its wave-cycles waiting, 81 % of its LDS cycles in bank
// conflicts.  The windows are consecutive from bit 0 (msm_plan, msm_table_layout), 2 <= c <= 22.)
template <class Fn>
__device__ __forceinline__ void for_each_digit(const Fr& s, const WinLayout& L, Fn f) {
    uint32_t carry = 0;
    uint32_t r[8];
#pragma unroll
    for (int i = 0; i < 8; i++) r[i] = s.v[i];
    for (unsigned w = 0; w < L.nwin; w++) {
        const unsigned c = L.c[w];
        const uint32_t mask = (1u << c) - 1, half = 1u << (c - 1);
        const uint32_t v = r[0];
#pragma unroll
        for (int i = 0; i < 7; i++) r[i] = __funnelshift_r(r[i], r[i + 1], c);
        r[7] >>= c;
        uint32_t d = (v & mask) + carry;
        uint32_t code = 0;  // 0 = zero digit, else ((bucket << 1) | negative) + 1
        if (d > half) {
            carry = 1;
            if (d != (1u << c)) code = ((((1u << c) - d - 1) << 1) | 1u) + 1;  // digit d - 2^c < 0
        } else {
            carry = 0;
            if (d != 0) code = ((d - 1) << 1) + 1;
        }
        f(w, code);
    }
}

// digits[w * n + i] = code of window w of scalar i (coalesced over i).            HBM: 32 B read + 4 W B written / point
// A scalar that is not a canonical field element (>= r; arkworks' BigInteger256 scalars always are) raises *bad: the
// recoding only covers 254 bits, so such an input cannot be given a meaning.  Points flagged in the infinity mask get
// zero digits (they contribute nothing, as in VariableBaseMSM).  bad[1] counts the points that contribute nothing.
// Bucket-range split of one proof over several ranks (MsmTable::shard_world): a digit of a full-width window is kept when
// its bucket lies in [blo, bhi), a digit of a narrower window when its point lies in [plo, phi).
struct DigitShard {
    uint32_t on, cfull, blo, bhi;
    uint64_t plo, phi;
};
__global__ void __launch_bounds__(256) msm_digits(const Fr* __restrict__ scalars, size_t n, int mont, WinLayout L,
                                                  uint32_t* __restrict__ digits, const uint32_t* __restrict__ inf_mask,
                                                  size_t inf_first, uint32_t* __restrict__ bad, DigitShard sh, unsigned sblk_log,
                                                  size_t sbstride) {
    SWM_LIGHT_KERNEL();
    // scalar i sits at scalars[(i >> sblk_log) * sbstride + (i & (2^sblk_log - 1))]: contiguous by default (sblk_log = 31), the
    // blocks a rank of a sharded proof takes of a polynomial every rank holds otherwise (MsmTable::scalar_stride, blk_log, bstride)
    const size_t smask = ((size_t)1 << sblk_log) - 1;
    // (r06: two scalars per lane and trip, both loads issued before the first is recoded — 64.5 instead of 60.3 us per 2^20 points on
    // the box that measured it: not kept)
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        Fr s = scalars[(i >> sblk_log) * sbstride + (i & smask)];
        bool ge = true;  // s >= r ?
#pragma unroll
        for (int k = 7; k >= 0; k--) {
            if (s.v[k] != FrParams::P[k]) {
                ge = s.v[k] > FrParams::P[k];
                break;
            }
        }
        if (ge) atomicOr(bad, 1u);
        if (mont) s = fp_to_std(s);
        bool skip = false;
        if (inf_mask) {
            size_t b = inf_first + i;
            skip = (inf_mask[b >> 5] >> (b & 31)) & 1u;
        }
        {   // points that contribute nothing (zero scalar, or an identity base): counted per wave, one atomic per wave —
            // the measurement prices them at their 32-B scalar only (bench.py: roofline numerator)
            uint32_t z = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) z |= s.v[k];
            const unsigned long long m = __ballot(z == 0 || skip);
            if (m && (threadIdx.x & 63) == (unsigned)__ffsll((long long)m) - 1) atomicAdd(bad + 1, (uint32_t)__popcll(m));
        }
        for_each_digit(s, L, [&](unsigned w, uint32_t code) {
            if (sh.on && code) {
                const uint32_t bucket = (code - 1) >> 1;
                const bool keep = L.c[w] == sh.cfull ? (bucket >= sh.blo && bucket < sh.bhi) : (i >= sh.plo && i < sh.phi);
                if (!keep) code = 0;
            }
            digits[(size_t)w * n + i] = skip ? 0u : code;
        });
    }
}

// Counting sort, pass 1: per-(tile, window) histogram in LDS, one global atomic per non-empty LDS bin.
__global__ void __launch_bounds__(SORT_THREADS) msm_hist(const uint32_t* __restrict__ digits, size_t n, WinLayout L,
                                                         uint32_t SORT_TILE, uint32_t* __restrict__ hist) {
    extern __shared__ uint32_t lh[];
    const uint32_t w = blockIdx.y;
    const uint32_t B = 1u << (L.c[w] - 1), boff = L.boff[w];
    const size_t lo = (size_t)blockIdx.x * SORT_TILE, hi = min(lo + (size_t)SORT_TILE, n);
    for (uint32_t b = threadIdx.x; b < B; b += SORT_THREADS) lh[b] = 0;
    __syncthreads();
    const uint32_t* d = digits + (size_t)w * n;
    // four independent loads in flight per lane: the loop is bound by load latency, not by bandwidth
    for (size_t i = lo + threadIdx.x; i < hi; i += 4 * SORT_THREADS) {
        uint32_t c[4];
#pragma unroll
        for (int u = 0; u < 4; u++) c[u] = i + u * SORT_THREADS < hi ? d[i + u * SORT_THREADS] : 0u;
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (c[u]) atomicAdd(&lh[(c[u] - 1) >> 1], 1u);
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < B; b += SORT_THREADS) {
        uint32_t v = lh[b];
        if (v) atomicAdd(&hist[boff + b], v);
    }
}

// Counting sort, pass 2: the tile reserves a contiguous run in every bucket it touches (one returning global
// atomic per non-empty bin), then ranks its digits with LDS atomics and writes (point index | sign).
__global__ void __launch_bounds__(SORT_THREADS) msm_scatter(const uint32_t* __restrict__ digits, size_t n, WinLayout L,
                                                            uint32_t SORT_TILE, const uint32_t* __restrict__ bucket_off,
                                                            uint32_t* __restrict__ cursor,
                                                            uint32_t* __restrict__ sorted,
                                                            const uint32_t* __restrict__ two_level_bad /* null: always run */) {
    extern __shared__ uint32_t lh[];
    const uint32_t w = blockIdx.y;
    if (two_level_bad && two_level_bad[w] == 0) return;  // the two-level scatter below handles this window
    const uint32_t B = 1u << (L.c[w] - 1), boff = L.boff[w];
    const size_t lo = (size_t)blockIdx.x * SORT_TILE, hi = min(lo + (size_t)SORT_TILE, n);
    for (uint32_t b = threadIdx.x; b < B; b += SORT_THREADS) lh[b] = 0;
    __syncthreads();
    const uint32_t* d = digits + (size_t)w * n;
    for (size_t i = lo + threadIdx.x; i < hi; i += 4 * SORT_THREADS) {
        uint32_t c[4];
#pragma unroll
        for (int u = 0; u < 4; u++) c[u] = i + u * SORT_THREADS < hi ? d[i + u * SORT_THREADS] : 0u;
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (c[u]) atomicAdd(&lh[(c[u] - 1) >> 1], 1u);
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < B; b += 4 * SORT_THREADS) {  // four returning global atomics in flight per lane
        uint32_t v[4], r[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = b + u * SORT_THREADS < B ? lh[b + u * SORT_THREADS] : 0u;
#pragma unroll
        for (int u = 0; u < 4; u++) r[u] = v[u] ? atomicAdd(&cursor[boff + b + u * SORT_THREADS], v[u]) : 0u;
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (v[u]) lh[b + u * SORT_THREADS] = bucket_off[boff + b + u * SORT_THREADS] + r[u];
    }
    __syncthreads();
    for (size_t i = lo + threadIdx.x; i < hi; i += 4 * SORT_THREADS) {
        uint32_t c[4];
#pragma unroll
        for (int u = 0; u < 4; u++) c[u] = i + u * SORT_THREADS < hi ? d[i + u * SORT_THREADS] : 0u;
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (c[u]) {
                uint32_t pos = atomicAdd(&lh[(c[u] - 1) >> 1], 1u);
                sorted[pos] = (uint32_t)(i + u * SORT_THREADS) | (((c[u] - 1) & 1u) << 31);
            }
    }
}

// ---- two-level scatter for large MSMs
// The one-level scatter above writes every entry to its final position with a 4-byte store; at ~2 entries per
// (tile, bucket) those stores are scattered over the whole array and the kernel is bound by write sectors (73 G
// digits/s whatever the tile size).  Two levels keep every store in a run:
//   msm_partition : a workgroup groups the (entry, bucket) pairs of an 8192-digit tile by COARSE bin (bucket >> fb[w])
//                   in LDS, reserves one run per (tile, bin) with a global atomic and copies the tile out run by run
//                   (~32 pairs = 256 B each);
//   msm_bin_sort  : one workgroup per bin (<= BIN_CAP entries: the region of `sorted` between the offsets of its
//                   first and last bucket) places the entries by fine bucket in LDS — the per-bucket offsets are
//                   already known from the prefix sums — and copies the region out in order.
// Bins are sized per window (BinPlan): the top window only sees digits up to r >> bit, i.e. much denser buckets.
// A bin larger than BIN_CAP (skewed digits: many equal scalars) marks its WINDOW bad and that window takes the
// one-level path: both sets of kernels are launched and the unused one returns at once — no host read-back.
static constexpr uint32_t BIN_CAP = 24576;   // entries per bin (96 KB of LDS in msm_bin_sort)
static constexpr int BIN_THREADS = 1024;
static constexpr uint32_t PART_TILE = 8192;  // digits per workgroup in msm_partition (64 KB of LDS for the pairs)
static constexpr uint32_t PART_MAX_BINS = 1024;
struct BinPlan {
    uint8_t fb[MAX_WIN];      // fine bits of window w: bin = bucket >> fb[w], at most 128 buckets per bin
    uint16_t nbins[MAX_WIN];  // bins that can be non-empty in window w
    uint32_t max_nbins;
};
__global__ void __launch_bounds__(256) msm_bin_check(const uint32_t* __restrict__ bucket_off, WinLayout L, BinPlan P,
                                                     uint32_t* __restrict__ bad) {
    const uint32_t w = blockIdx.y, bin = blockIdx.x * blockDim.x + threadIdx.x;
    if (bin >= P.nbins[w]) return;
    const uint32_t B = 1u << (L.c[w] - 1), fb = P.fb[w], first = bin << fb, last = min(first + (1u << fb), B);
    uint32_t size = bucket_off[L.boff[w] + last] - bucket_off[L.boff[w] + first];
    if (bin + 1 == P.nbins[w]) size = bucket_off[L.boff[w] + B] - bucket_off[L.boff[w] + first];  // nothing may lie beyond
    if (size > BIN_CAP || (bin + 1 == P.nbins[w] && last < B && bucket_off[L.boff[w] + B] != bucket_off[L.boff[w] + last]))
        bad[w] = 1u;
}
__global__ void __launch_bounds__(1024) msm_partition(const uint32_t* __restrict__ digits, size_t n, WinLayout L, BinPlan P,
                                                      const uint32_t* __restrict__ bucket_off,
                                                      uint32_t* __restrict__ bin_cursor, const uint32_t* __restrict__ bad,
                                                      uint2* __restrict__ tmp) {
    __shared__ uint32_t cnt[PART_MAX_BINS], start[PART_MAX_BINS], gpos[PART_MAX_BINS];
    extern __shared__ uint2 stage[];  // PART_TILE pairs
    const uint32_t w = blockIdx.y;
    if (bad[w]) return;
    const uint32_t nb = P.nbins[w], fb = P.fb[w], boff = L.boff[w], t = threadIdx.x;
    const size_t lo = (size_t)blockIdx.x * PART_TILE;
    for (uint32_t b = t; b < nb; b += 1024) cnt[b] = 0;
    __syncthreads();
    const uint32_t* d = digits + (size_t)w * n;
    uint32_t c[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
        size_t i = lo + t + (size_t)u * 1024;
        c[u] = i < n ? d[i] : 0u;
    }
#pragma unroll
    for (int u = 0; u < 8; u++)
        if (c[u]) atomicAdd(&cnt[((c[u] - 1) >> 1) >> fb], 1u);
    __syncthreads();
    // exclusive scan of cnt over the (<= 1024) bins: one bin per lane, Hillis-Steele in `start`
    uint32_t mine = t < nb ? cnt[t] : 0u;
    start[t] = mine;
    __syncthreads();
    for (uint32_t dd = 1; dd < 1024; dd <<= 1) {
        uint32_t v = t >= dd ? start[t - dd] : 0u;
        __syncthreads();
        start[t] += v;
        __syncthreads();
    }
    const uint32_t excl = start[t] - mine;
    __syncthreads();
    start[t] = excl;
    if (t < nb) {
        cnt[t] = excl;  // running cursor of the bin inside the staged tile
        gpos[t] = mine ? bucket_off[boff + (t << fb)] + atomicAdd(&bin_cursor[w * PART_MAX_BINS + t], mine) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 8; u++)
        if (c[u]) {
            const uint32_t bucket = (c[u] - 1) >> 1;
            const uint32_t p = atomicAdd(&cnt[bucket >> fb], 1u);
            stage[p] = make_uint2((uint32_t)(lo + t + (size_t)u * 1024) | (((c[u] - 1) & 1u) << 31), bucket);
        }
    __syncthreads();
    const uint32_t total = start[nb - 1] + (cnt[nb - 1] - start[nb - 1]);  // pairs staged by this tile
    for (uint32_t i = t; i < total; i += 1024) {
        uint2 e = stage[i];
        uint32_t b = e.y >> fb;
        tmp[gpos[b] + (i - start[b])] = e;
    }
}
__global__ void __launch_bounds__(BIN_THREADS) msm_bin_sort(const uint2* __restrict__ tmp, WinLayout L, BinPlan P,
                                                            const uint32_t* __restrict__ bucket_off,
                                                            const uint32_t* __restrict__ bad, uint32_t* __restrict__ sorted) {
    extern __shared__ uint32_t stage32[];  // BIN_CAP entries
    __shared__ uint32_t fc[128];
    const uint32_t w = blockIdx.y, bin = blockIdx.x;
    if (bad[w] || bin >= P.nbins[w]) return;
    const uint32_t B = 1u << (L.c[w] - 1), fb = P.fb[w], first = bin << fb, nf = min(1u << fb, B - first);
    const uint32_t* off = bucket_off + L.boff[w] + first;
    const uint32_t lo = off[0], cnt = off[nf] - lo;
    for (uint32_t f = threadIdx.x; f < nf; f += BIN_THREADS) fc[f] = off[f] - lo;
    __syncthreads();
    const uint32_t fmask = (1u << fb) - 1;
    for (uint32_t i = threadIdx.x; i < cnt; i += 4 * BIN_THREADS) {
        uint2 e[4];
#pragma unroll
        for (int u = 0; u < 4; u++) e[u] = i + u * BIN_THREADS < cnt ? tmp[lo + i + u * BIN_THREADS] : make_uint2(0u, 0u);
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (i + u * BIN_THREADS < cnt) stage32[atomicAdd(&fc[e[u].y & fmask], 1u)] = e[u].x;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < cnt; i += BIN_THREADS) sorted[lo + i] = stage32[i];
}


// ---------------------------------------------------------------------------------------------- precomputed-window ("flat") schedule
// For a RESIDENT base set the multiples 2^(bit_w) P_i of every point are precomputed once (msm_table_build: W copies
// of the set, 96 B x n x W in HBM — what 288 GB are for).  A digit of window w then selects table row w instead of a
// bucket set of its own: ALL windows share ONE set of 2^(c-1) buckets, so
//   * the window width is no longer capped by "buckets per window x windows" (the bucket stage runs once): c = 20 gives
//     13 windows instead of 16, i.e. 19 % fewer mixed additions in msm_accumulate for the same result;
//   * there is no Horner chain over the windows at the end (the shifts are in the table).
// The counting sort over 2^19 buckets cannot histogram in LDS in one go; it is two-level from the start:
//   msm_flat_coarse_hist / msm_flat_scan_bins  entries per COARSE bin (bucket >> fb, <= 4096 bins) — LDS histogram per tile,
//   msm_flat_partition                         entries grouped by bin in LDS and written out in runs (as msm_partition),
//   msm_flat_bin_sort                          one workgroup per bin: fine counts (they ARE the bucket histogram the
//                                              scans below consume), placement in LDS, coalesced copy-out; a bin that
//                                              does not fit LDS (structured scalars) is placed directly in HBM.
static constexpr uint32_t FLAT_MAX_BINS = 4096;
static constexpr uint32_t FLAT_MAX_FINE = 2048;  // buckets per bin at most (2^fb): fine counts + offsets (16 KB) next to the 128-KB stage
static constexpr uint32_t FLAT_BIN_CAP = 32768;  // entries a bin may hold to be placed in LDS (128 KB)
// The per-bin cursors of msm_flat_partition sit one per 128-byte line: every workgroup of the launch reserves its runs with
// one returning atomic per bin, and packed (32 cursors to a line) those atomics queued up behind one another at the
// memory side — 135 of the kernel's 191 us at 2^20 points (measured by replacing the atomic with arithmetic).
static constexpr uint32_t FLAT_CUR_STRIDE = 32;   // = the most windows a table can have (msm_flat_applies): one cursor per (bin, window)
// grid (tiles over the points, windows): counts per (window, coarse bin), bin_count[w * nbins + b] (window-major: a workgroup's
// atomics land on consecutive words; bin-major lines made this kernel 23 instead of 11 us) — the partition places a
// bin's entries WINDOW BY WINDOW (see msm_flat_scan_bins), for which it needs the bin's count of every window
__global__ void __launch_bounds__(SORT_THREADS) msm_flat_coarse_hist(const uint32_t* __restrict__ digits, size_t n, unsigned fb,
                                                                     uint32_t nbins, uint32_t tile, uint32_t* __restrict__ bin_count) {
    SWM_LIGHT_KERNEL();
    __shared__ uint32_t lh[FLAT_MAX_BINS];
    for (uint32_t b = threadIdx.x; b < nbins; b += SORT_THREADS) lh[b] = 0;
    __syncthreads();
    const uint32_t w = blockIdx.y;
    const uint32_t* d = digits + (size_t)w * n;
    const size_t lo = (size_t)blockIdx.x * tile, hi = min(lo + (size_t)tile, n);
    for (size_t i = lo + threadIdx.x; i < hi; i += 4 * SORT_THREADS) {
        uint32_t c[4];
#pragma unroll
        for (int u = 0; u < 4; u++) c[u] = i + u * SORT_THREADS < hi ? d[i + u * SORT_THREADS] : 0u;
#pragma unroll
        for (int u = 0; u < 4; u++)
            if (c[u]) atomicAdd(&lh[((c[u] - 1) >> 1) >> fb], 1u);
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nbins; b += SORT_THREADS) {
        uint32_t v = lh[b];
        if (v) atomicAdd(&bin_count[(size_t)w * nbins + b], v);
    }
}
// Exclusive scan of <= 4096 bin totals (summed over the windows) by one workgroup: bin_off[0 .. nbins]; the offsets of every
// (bin, window) inside that, win_off[b * FLAT_CUR_STRIDE + w] — a bin's entries are laid out WINDOW BY WINDOW, so that a bucket's entries
// reach the accumulation window by window too: all lanes of the chip then gather from the table rows of the same one or two
// windows at a time (1 / 13 of a multi-GB table: TLB reach and Infinity Cache, r04 — DESIGN.md §3.1) instead of from all of it;
// and the bins' SEGMENT capacities 2^fb + count / SEG (an upper bound of sum over the bin's buckets of ceil(count_b / SEG)):
// bin_seg_off[0 .. nbins] — the segment indices a bin's workgroup of msm_flat_bin_sort hands out without knowing what the
// other bins need.
// BPT bins per lane: 1 up to 1024 bins (one lane per bin: its 2 x 32 loads are all the latency there is), 4 above
template <int BPT>
__global__ void __launch_bounds__(1024) msm_flat_scan_bins(const uint32_t* __restrict__ bin_count, uint32_t nbins, uint32_t nwin,
                                                           uint32_t* __restrict__ bin_off, uint32_t* __restrict__ win_off, unsigned fb,
                                                           uint32_t SEG, uint32_t* __restrict__ bin_seg_off) {
    SWM_LIGHT_KERNEL();
    __shared__ uint32_t sm[1024], sg[1024];
    const uint32_t t = threadIdx.x;
    uint32_t v[BPT], g[BPT], s = 0, q = 0;
    // a bin's counts of all windows.  No branches: every load is unconditional, from a clamped (window, bin), and masked afterwards;
    // the offsets leave as whole 128-byte lines.  (r04's first form looped over the windows with a load per trip, each waiting for
    // the one before; its second predicated every load and store: 392 branches and 35 us in a kernel every MSM waits for.)
#pragma unroll
    for (int u = 0; u < BPT; u++) {
        const bool in = BPT * t + u < nbins;
        const uint32_t bc = min(BPT * t + u, nbins - 1);
        uint32_t sum = 0;
#pragma unroll
        for (uint32_t w = 0; w < FLAT_CUR_STRIDE; w++) {
            const uint32_t x = bin_count[(size_t)min(w, nwin - 1) * nbins + bc];
            sum += w < nwin ? x : 0u;
        }
        v[u] = in ? sum : 0u;
        g[u] = in ? (1u << fb) + v[u] / SEG : 0u;
        s += v[u];
        q += g[u];
    }
    sm[t] = s;
    sg[t] = q;
    __syncthreads();
    const uint32_t T = blockDim.x;  // a power of two >= nbins / BPT (a single wave for the few bins of a small MSM)
    for (uint32_t d = 1; d < T; d <<= 1) {
        uint32_t x = t >= d ? sm[t - d] : 0u, y = t >= d ? sg[t - d] : 0u;
        __syncthreads();
        sm[t] += x;
        sg[t] += y;
        __syncthreads();
    }
    uint32_t run = sm[t] - s, rung = sg[t] - q;
#pragma unroll
    for (int u = 0; u < BPT; u++) {
        const uint32_t b = BPT * t + u;
        if (b < nbins) {
            bin_off[b] = run;
            bin_seg_off[b] = rung;
            uint4* o = reinterpret_cast<uint4*>(win_off + (size_t)b * FLAT_CUR_STRIDE);
            uint32_t at = run;
#pragma unroll
            for (uint32_t k = 0; k < FLAT_CUR_STRIDE / 4; k++) {  // exclusive prefix over the windows (the words beyond nwin: unused)
                uint4 c;
                c.x = bin_count[(size_t)min(4 * k + 0, nwin - 1) * nbins + b];
                c.y = bin_count[(size_t)min(4 * k + 1, nwin - 1) * nbins + b];
                c.z = bin_count[(size_t)min(4 * k + 2, nwin - 1) * nbins + b];
                c.w = bin_count[(size_t)min(4 * k + 3, nwin - 1) * nbins + b];
                uint4 w4;
                w4.x = at;
                w4.y = w4.x + c.x;
                w4.z = w4.y + c.y;
                w4.w = w4.z + c.z;
                at = w4.w + c.w;
                o[k] = w4;
            }
        }
        run += v[u];
        rung += g[u];
    }
    if (t == T - 1) {
        bin_off[nbins] = sm[T - 1];
        bin_seg_off[nbins] = sg[T - 1];
    }
}
// (entry, bucket) pairs grouped by coarse bin; grid (tiles over the points, windows).  entry = table row of the point:
// w * tstride + toff + i, with the sign of the digit in bit 31.
template <int FLAT_PART_U>
__global__ void __launch_bounds__(1024) msm_flat_partition(const uint32_t* __restrict__ digits, size_t n, uint32_t tstride,
                                                           uint32_t toff, unsigned blk_log, uint32_t bstride, unsigned fb, uint32_t nbins,
                                                           const uint32_t* __restrict__ win_off, uint32_t nwin,
                                                           uint32_t* __restrict__ bin_cursor, uint2* __restrict__ tmp) {
    SWM_LIGHT_KERNEL();
    // LDS: PART_TILE pairs | cnt, start, gpos (nbins words each, rounded up to a multiple of 4) | 1024 scan words: sized by
    // the bin count of the call (74 KB at 512 bins: two workgroups per CU; the fixed 4096-bin arrays allowed one)
    constexpr uint32_t FLAT_PART_TILE = FLAT_PART_U * 1024u;
    extern __shared__ uint2 stage[];  // FLAT_PART_TILE pairs
    const uint32_t nb4 = (nbins + 3) & ~3u;
    uint32_t* cnt = reinterpret_cast<uint32_t*>(stage + FLAT_PART_TILE);
    uint32_t* start = cnt + nb4;
    uint32_t* gpos = start + nb4;
    uint32_t* scan = gpos + nb4;
    const uint32_t w = blockIdx.y, t = threadIdx.x;
    const size_t lo = (size_t)blockIdx.x * FLAT_PART_TILE;
    for (uint32_t b = t; b < nbins; b += 1024) cnt[b] = 0;
    __syncthreads();
    const uint32_t* d = digits + (size_t)w * n;
    uint32_t c[FLAT_PART_U];
#pragma unroll
    for (int u = 0; u < FLAT_PART_U; u++) {
        size_t i = lo + t + (size_t)u * 1024;
        c[u] = i < n ? d[i] : 0u;
    }
#pragma unroll
    for (int u = 0; u < FLAT_PART_U; u++)
        if (c[u]) atomicAdd(&cnt[((c[u] - 1) >> 1) >> fb], 1u);
    __syncthreads();
    // exclusive scan of cnt over <= 4096 bins: four bins per lane
    uint32_t v[4], sum = 0;
#pragma unroll
    for (int u = 0; u < 4; u++) {
        v[u] = 4 * t + u < nbins ? cnt[4 * t + u] : 0u;
        sum += v[u];
    }
    // inclusive scan over the 1024 lanes: shuffles inside a wave, the 16 wave totals through LDS (two barriers instead of
    // the twenty of a Hillis-Steele scan in LDS)
    uint32_t inc = sum;
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) {
        uint32_t x = __shfl_up(inc, dd, 64);
        if ((t & 63) >= (uint32_t)dd) inc += x;
    }
    if ((t & 63) == 63) scan[t >> 6] = inc;
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (uint32_t wv = 0; wv < 16; wv++) {
        const uint32_t x = scan[wv];
        if (wv < (t >> 6)) before += x;
        total += x;
    }
    uint32_t run = before + inc - sum;
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const uint32_t b = 4 * t + u;
        if (b < nbins) {
            start[b] = run;
            cnt[b] = run;  // running cursor of the bin inside the staged tile
            // the run of this (tile, bin) inside the bin's range of window w; the cursors of a bin's windows share one line
            gpos[b] = v[u] ? win_off[(size_t)b * FLAT_CUR_STRIDE + w] + atomicAdd(&bin_cursor[(size_t)b * FLAT_CUR_STRIDE + w], v[u]) : 0u;
        }
        run += v[u];
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < FLAT_PART_U; u++)
        if (c[u]) {
            const uint32_t bucket = (c[u] - 1) >> 1;
            const uint32_t p = atomicAdd(&cnt[bucket >> fb], 1u);
            const uint32_t i = (uint32_t)(lo + t + (size_t)u * 1024);  // scalar i -> its base (MsmTable: blocks of a sharded layout)
            const uint32_t row = w * tstride + toff + (i >> blk_log) * bstride + (i & ((1u << blk_log) - 1u));
            stage[p] = make_uint2(row | (((c[u] - 1) & 1u) << 31), bucket);
        }
    __syncthreads();
    for (uint32_t i = t; i < total; i += 1024) {
        uint2 e = stage[i];
        uint32_t b = e.y >> fb;
        tmp[gpos[b] + (i - start[b])] = e;
    }
}
// One workgroup per bin: bucket histogram of the bin, placement by bucket — and everything the accumulation and the bucket
// stage need to know about the bin's buckets (r04; four launches — three scans over the histogram and a binary search per
// segment — did this before): a bin is a contiguous bucket range whose first entry (bin_off) and first segment index
// (bin_seg_off, a capacity: see msm_flat_scan_bins) are known, so with the bin's fine counts in LDS the workgroup writes
//   hist[b], bucket_off[b], seg_off[b]           count, first entry and first segment of bucket b
//   seg_start[s], seg_len[s]                     the balanced segments (<= SEG entries) of its buckets; the unused tail of the
//                                                bin's segment indices gets length 0 (msm_seg_order skips those)
//   len_hist                                     segments per length (for the ordering by length that follows)
//   big_list                                     buckets with more than big_nseg segments (folded ahead of the bucket stage)
struct FlatSegOut {
    uint32_t *hist, *bucket_off, *seg_off, *seg_start, *seg_len, *len_hist, *big_count, *big_list;
    uint32_t SEG, big_nseg;
    // the ns segments of a bucket with more than SEG entries take its entries ROUND-ROBIN (segment k: entries k, k + ns, ...;
    // seg_len = length | ns << 8) instead of as consecutive ranges: a bucket's entries lie window by window, and every lane
    // should walk the windows in the same order (the locality of the accumulation's gathers).  TE jobs only.
    uint32_t interleave;
};
#ifndef SWM_PLACE_U
#define SWM_PLACE_U 4
#endif
static constexpr int PLACE_U = SWM_PLACE_U;
static constexpr uint32_t LEN_STRIDE = 32;  // the SEG + 1 length counters sit one per 128-byte line (see FLAT_CUR_STRIDE)
__device__ __forceinline__ uint32_t nseg_of(uint32_t cnt, uint32_t seg) { return (cnt + seg - 1) / seg; }
// segment k of the ns balanced segments of a bucket with c entries starting at `ent` (segment indices from s0); lh: the
// workgroup's length histogram in LDS.  Interleaved segments pack their stride into seg_len (length | ns << 8): ns < 2^24,
// otherwise (only reachable with a one-entry segment bound on a degenerate input) the bucket keeps consecutive ranges.
static constexpr uint32_t FLAT_WIDE_NS = 16, FLAT_WIDE_Q = 32;  // segments above which a bucket is written by the workgroup; queue length
__device__ __forceinline__ void flat_seg_write(const FlatSegOut& o, uint32_t c, uint32_t ns, uint32_t ent, uint32_t s0, uint32_t k,
                                               uint32_t* lh) {
    if (o.interleave && ns > 1 && ns < (1u << 24)) {
        const uint32_t len = (c - k + ns - 1) / ns;
        o.seg_start[s0 + k] = ent + k;
        o.seg_len[s0 + k] = len | (ns << 8);
        atomicAdd(&lh[o.SEG - len], 1u);
    } else {
        const uint32_t ks = (uint32_t)(((uint64_t)c * k) / ns), ke = (uint32_t)(((uint64_t)c * (k + 1)) / ns);
        o.seg_start[s0 + k] = ent + ks;
        o.seg_len[s0 + k] = ke - ks;
        atomicAdd(&lh[o.SEG - (ke - ks)], 1u);
    }
}
// A twin's second sorted array (MsmTwin): the same entries with the rows of the follower's table — an entry's window is
// row / tstride (the rows of a window are tstride apart and offset + i < tstride), its row there row + w * dstride + doff
// (mod 2^31: the sign of the digit stays in bit 31).
struct TwinOut {
    uint32_t* sorted2;  // null: no follower
    uint32_t tstride, dstride, doff;
    __device__ __forceinline__ uint32_t map(uint32_t e) const {
        const uint32_t row = e & 0x7fffffffu, w = row / tstride;
        return (e & 0x80000000u) | ((row + w * dstride + doff) & 0x7fffffffu);
    }
};
template <int BT>
__global__ void __launch_bounds__(BT) msm_flat_bin_sort(const uint2* __restrict__ tmp, unsigned fb, uint32_t NB,
                                                                 const uint32_t* __restrict__ bin_off,
                                                                 const uint32_t* __restrict__ bin_seg_off, FlatSegOut o,
                                                                 uint32_t* __restrict__ sorted, TwinOut tw) {
    SWM_LIGHT_KERNEL();
    extern __shared__ uint32_t stage32[];  // fc[nf] | fo[nf] | FLAT_BIN_CAP entries
    __shared__ uint32_t wsum[2][BT / 64];
    __shared__ uint32_t lh[SEG_MAX + 1];
    __shared__ uint32_t wide[FLAT_WIDE_Q][3], nwide;  // buckets whose descriptors the whole workgroup writes: (count, first entry, first segment)
    uint32_t* fc = stage32;
    uint32_t* fo = fc + (1u << fb);
    uint32_t* stage = fo + (1u << fb);
    const uint32_t bin = blockIdx.x, nf = 1u << fb, first = bin << fb, fmask = nf - 1, t = threadIdx.x;
    const uint32_t lo = bin_off[bin], cnt = bin_off[bin + 1] - lo;
    const uint32_t seg_lo = bin_seg_off[bin], seg_hi = bin_seg_off[bin + 1];
    for (uint32_t f = t; f < nf; f += BT) fc[f] = 0;
    for (uint32_t i = t; i <= o.SEG; i += BT) lh[i] = 0;
    if (t == 0) nwide = 0;
    __syncthreads();
    for (uint32_t i = t; i < cnt; i += BT) atomicAdd(&fc[tmp[lo + i].y & fmask], 1u);
    __syncthreads();
    // exclusive prefixes of the fine counts and of the segments per bucket (nf <= 2048): a chunk per lane, shuffles inside
    // a wave, the 16 wave totals through LDS
    const uint32_t per = (nf + BT - 1) / BT, f0 = t * per, f1 = min(f0 + per, nf);
    uint32_t sum = 0, sums = 0;
    for (uint32_t f = f0; f < f1; f++) {
        sum += fc[f];
        sums += nseg_of(fc[f], o.SEG);
    }
    uint32_t inc = sum, incs = sums;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t x = __shfl_up(inc, d, 64), y = __shfl_up(incs, d, 64);
        if ((t & 63) >= (uint32_t)d) {
            inc += x;
            incs += y;
        }
    }
    if ((t & 63) == 63) {
        wsum[0][t >> 6] = inc;
        wsum[1][t >> 6] = incs;
    }
    __syncthreads();
    uint32_t run = inc - sum, runs = incs - sums, used = 0;
#pragma unroll
    for (uint32_t wv = 0; wv < BT / 64; wv++) {
        if (wv < (t >> 6)) {
            run += wsum[0][wv];
            runs += wsum[1][wv];
        }
        used += wsum[1][wv];
    }
    for (uint32_t f = f0; f < f1; f++) {
        const uint32_t c = fc[f], ns = nseg_of(c, o.SEG);
        fo[f] = run;
        if (first + f < NB) {
            const uint32_t b = first + f, ent = lo + run, s0 = seg_lo + runs;
            o.hist[b] = c;
            o.bucket_off[b] = ent;
            o.seg_off[b] = s0;
            // balanced split, as msm_seg_desc: one segment per bucket unless c > SEG.  A bucket with many segments (skewed
            // scalars: 2^20 equal ones are 8192 segments of one bucket) is queued for the WHOLE workgroup (below) instead of
            // being written by this lane alone — the kernel every flat MSM waits for must not serialise on one lane
            uint32_t slot = FLAT_WIDE_Q;
            if (ns > FLAT_WIDE_NS) slot = atomicAdd(&nwide, 1u);
            if (slot < FLAT_WIDE_Q) {
                wide[slot][0] = c;
                wide[slot][1] = ent;
                wide[slot][2] = s0;
            } else {
                for (uint32_t k = 0; k < ns; k++) flat_seg_write(o, c, ns, ent, s0, k, lh);
            }
            if (ns > o.big_nseg) o.big_list[atomicAdd(o.big_count, 1u)] = b;
        }
        run += c;
        runs += ns;
    }
    for (uint32_t k = seg_lo + used + t; k < seg_hi; k += BT) o.seg_len[k] = 0;  // indices the bin did not need
    __syncthreads();
    for (uint32_t q = 0, nq = min(nwide, FLAT_WIDE_Q); q < nq; q++) {  // the queued many-segment buckets: one segment per lane and trip
        const uint32_t c = wide[q][0], ns = nseg_of(c, o.SEG);
        for (uint32_t k = t; k < ns; k += BT) flat_seg_write(o, c, ns, wide[q][1], wide[q][2], k, lh);
    }
    __syncthreads();
    for (uint32_t i = t; i <= o.SEG; i += BT)
        if (lh[i]) atomicAdd(&o.len_hist[i * LEN_STRIDE], lh[i]);
    if (cnt <= FLAT_BIN_CAP) {
        // the bin's entries lie window by window (msm_flat_scan_bins); they are placed in chunks of PLACE_U x 1024 consecutive
        // entries, so a bucket's entries keep that order up to the chunk size — what the accumulation's locality rests on
        for (uint32_t i = t; i < cnt; i += PLACE_U * BT) {
            uint2 e[PLACE_U];
#pragma unroll
            for (int u = 0; u < PLACE_U; u++) e[u] = i + u * BT < cnt ? tmp[lo + i + u * BT] : make_uint2(0u, 0u);
#pragma unroll
            for (int u = 0; u < PLACE_U; u++)
                if (i + u * BT < cnt) stage[atomicAdd(&fo[e[u].y & fmask], 1u)] = e[u].x;
        }
        __syncthreads();
        for (uint32_t i = t; i < cnt; i += BT) sorted[lo + i] = stage[i];
        if (tw.sorted2)
            for (uint32_t i = t; i < cnt; i += BT) tw.sorted2[lo + i] = tw.map(stage[i]);
    } else {  // oversized bin (many equal digits): scattered 4-byte stores, correct for any size
        for (uint32_t i = t; i < cnt; i += BT) {
            uint2 e = tmp[lo + i];
            const uint32_t at = lo + atomicAdd(&fo[e.y & fmask], 1u);
            sorted[at] = e.x;
            if (tw.sorted2) tw.sorted2[at] = tw.map(e.x);
        }
    }
}
// The follower of a twin pair takes the lead's bucket and segment descriptors: up to eight word ranges, one launch.
struct TwinCopy {
    const uint32_t* src[8];
    uint32_t* dst[8];
    uint32_t words[8];
    int k;
};
__global__ void __launch_bounds__(256) msm_twin_copy(TwinCopy c) {
    SWM_LIGHT_KERNEL();
    const uint32_t gid = blockIdx.x * 256u + threadIdx.x, step = gridDim.x * 256u;
    for (int r = 0; r < c.k; r++)
        for (uint32_t i = gid; i < c.words[r]; i += step) c.dst[r][i] = c.src[r][i];
}

// ---- table construction: out[i] = 2^k in[i] as XYZZ (k doublings), then batch normalisation back to affine
__global__ void __launch_bounds__(256) msm_table_shift(const G1Affine* __restrict__ in, size_t n, unsigned k, G1XYZZ* __restrict__ out) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    G1Affine p = in[i];
    G1XYZZ acc = g1_is_inf(p) ? g1_xyzz_identity() : g1_dbl_affine(p);
    for (unsigned j = 1; j < k; j++) acc = g1_dbl(acc);
    out[i] = acc;
}
static constexpr int TAB_NORM_CHUNK = 16;
__global__ void __launch_bounds__(256) msm_table_normalize(const G1XYZZ* __restrict__ in, size_t n, Fq* __restrict__ pref,
                                                           G1Affine* __restrict__ out) {
    size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    size_t lo = t * TAB_NORM_CHUNK;
    if (lo >= n) return;
    size_t hi = lo + TAB_NORM_CHUNK < n ? lo + TAB_NORM_CHUNK : n;
    Fq acc = fp_one<Fq>();
    for (size_t i = lo; i < hi; i++) {
        pref[i] = acc;
        if (!fp_is_zero(in[i].zz)) acc = fp_mul(acc, fp_mul(in[i].zz, in[i].zzz));
    }
    Fq inv = fp_inv(acc);
    for (size_t i = hi; i-- > lo;) {
        G1XYZZ p = in[i];
        G1Affine a;
        if (fp_is_zero(p.zz)) {
            a.x = fp_zero<Fq>();
            a.y = fp_zero<Fq>();
        } else {
            Fq zi = fp_mul(inv, pref[i]);
            inv = fp_mul(inv, fp_mul(p.zz, p.zzz));
            a.x = fp_mul(p.x, fp_mul(zi, p.zzz));
            a.y = fp_mul(p.y, fp_mul(zi, p.zz));
        }
        out[i] = a;
    }
}

// affine points on y^2 = x^3 + 1 (plain form, radix 2^384) -> twisted Edwards table rows (y - x, y + x, 2 d x y), each
// coordinate x 2^8 like the XYZZ table.  x = f (x_w + 1)/y_w, y = (u - 1)/(u + 1), u = s (x_w + 1): one inversion per
// TAB_NORM_CHUNK points (of the product of their y_w (u + 1)).  The identity (0, 0) becomes the row (1, 1, 0) of (0, 1).
// A point with y_w (u + 1) = 0 (order 2, or mapped to infinity: never in the prime-order subgroup) raises *bad.
__global__ void __launch_bounds__(256) msm_te_convert(const G1Affine* __restrict__ in, size_t n, Fq* __restrict__ pref,
                                                      G1TE* __restrict__ out, uint32_t* __restrict__ bad) {
    size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    size_t lo = t * TAB_NORM_CHUNK;
    if (lo >= n) return;
    size_t hi = lo + TAB_NORM_CHUNK < n ? lo + TAB_NORM_CHUNK : n;
    const Fq one = fp_one<Fq>(), cs = fq_const(TeParams::S);
    Fq acc = one;
    for (size_t i = lo; i < hi; i++) {
        G1Affine p = in[i];
        pref[i] = acc;
        if (g1_is_inf(p)) continue;
        Fq den = fp_mul(p.y, fp_add(fp_mul(cs, fp_add(p.x, one)), one));
        if (fp_is_zero(den)) atomicOr(bad, 1u);
        else acc = fp_mul(acc, den);
    }
    Fq inv = fp_inv(acc);
    const uint32_t k256[12] = SWM_FQ_SCALE256_MONT;
    Fq c256, one256;
#pragma unroll
    for (int j = 0; j < 12; j++) c256.v[j] = k256[j];
    one256 = fp_mul(one, c256);
    for (size_t i = hi; i-- > lo;) {
        G1Affine p = in[i];
        Fq ymx, ypx, kt;
        Fq x1 = fp_add(p.x, one), u = fp_mul(cs, x1), up1 = fp_add(u, one), den = fp_mul(p.y, up1);
        if (g1_is_inf(p) || fp_is_zero(den)) {
            ymx = one256;
            ypx = one256;
            kt = fp_zero<Fq>();
        } else {
            Fq di = fp_mul(inv, pref[i]);
            inv = fp_mul(inv, den);
            Fq xt = fp_mul(fp_mul(fq_const(TeParams::F), x1), fp_mul(di, up1));  // f (x_w + 1) / y_w
            Fq yt = fp_mul(fp_sub(u, one), fp_mul(di, p.y));                       // (u - 1) / (u + 1)
            ymx = fp_mul(fp_sub(yt, xt), c256);
            ypx = fp_mul(fp_add(yt, xt), c256);
            kt = fp_mul(fp_mul(fq_const(TeParams::K2D), fp_mul(xt, yt)), c256);
        }
        // canonical values split into the 28-bit limbs the accumulation multiplies, one 64-byte sector per coordinate
        const Fq28 l0 = fq28_unpack(ymx), l1 = fq28_unpack(ypx), l2 = fq28_unpack(kt);
        uint4* o = reinterpret_cast<uint4*>(&out[i]);
        o[0] = make_uint4(l0.l[0], l0.l[1], l0.l[2], l0.l[3]);
        o[1] = make_uint4(l0.l[4], l0.l[5], l0.l[6], l0.l[7]);
        o[2] = make_uint4(l0.l[8], l0.l[9], l0.l[10], l0.l[11]);
        o[3] = make_uint4(l0.l[12], l0.l[13], 0u, 0u);
        o[4] = make_uint4(l1.l[0], l1.l[1], l1.l[2], l1.l[3]);
        o[5] = make_uint4(l1.l[4], l1.l[5], l1.l[6], l1.l[7]);
        o[6] = make_uint4(l1.l[8], l1.l[9], l1.l[10], l1.l[11]);
        o[7] = make_uint4(l1.l[12], l1.l[13], 0u, 0u);
        o[8] = make_uint4(l2.l[0], l2.l[1], l2.l[2], l2.l[3]);
        o[9] = make_uint4(l2.l[4], l2.l[5], l2.l[6], l2.l[7]);
        o[10] = make_uint4(l2.l[8], l2.l[9], l2.l[10], l2.l[11]);
        o[11] = make_uint4(l2.l[12], l2.l[13], 0u, 0u);
    }
}
// [r]P = O and P on the curve, for every point that is not the identity (0, 0): double-and-add over the 253 bits of r
__global__ void __launch_bounds__(256) msm_subgroup_kernel(const G1Affine* __restrict__ in, size_t n, uint32_t* __restrict__ bad) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    G1Affine p = in[i];
    if (g1_is_inf(p)) return;
    if (!g1_is_on_curve(p)) {
        atomicOr(bad, 1u);
        return;
    }
    G1XYZZ acc = g1_xyzz_identity();
    bool started = false;
#pragma unroll 1
    for (int b = 252; b >= 0; b--) {
        if (started) acc = g1_dbl(acc);
        if ((FrParams::P[b >> 5] >> (b & 31)) & 1) {
            g1_add_mixed(acc, p);
            started = true;
        }
    }
    if (!g1_is_inf(acc)) atomicOr(bad, 2u);
}

// Exclusive scans of the bucket counts and of the per-bucket segment counts, in three launches:
// per-workgroup totals, a single-workgroup scan of those totals, per-workgroup scan + offset.
static constexpr int SCAN_BLOCK = 256;
static constexpr int SCAN_ITEMS = 8;  // consecutive buckets per lane
static constexpr int SCAN_TILE = SCAN_BLOCK * SCAN_ITEMS;

__device__ __forceinline__ void block_inclusive_scan2(uint32_t& a, uint32_t& b, uint32_t* sa, uint32_t* sb) {
    // Hillis-Steele over SCAN_BLOCK lanes (values returned in place: inclusive prefix)
    uint32_t tid = threadIdx.x;
    sa[tid] = a;
    sb[tid] = b;
    __syncthreads();
    for (uint32_t d = 1; d < SCAN_BLOCK; d <<= 1) {
        uint32_t va = tid >= d ? sa[tid - d] : 0, vb = tid >= d ? sb[tid - d] : 0;
        __syncthreads();
        sa[tid] += va;
        sb[tid] += vb;
        __syncthreads();
    }
    a = sa[tid];
    b = sb[tid];
}

__global__ void __launch_bounds__(SCAN_BLOCK) msm_scan_totals(const uint32_t* __restrict__ hist, uint32_t NB, uint32_t SEG,
                                                             uint32_t* __restrict__ tot_cnt,
                                                             uint32_t* __restrict__ tot_seg) {
    __shared__ uint32_t sa[SCAN_BLOCK], sb[SCAN_BLOCK];
    uint32_t lo = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint32_t a = 0, b = 0;
    for (uint32_t i = lo; i < min(lo + SCAN_ITEMS, NB); i++) {
        a += hist[i];
        b += nseg_of(hist[i], SEG);
    }
    block_inclusive_scan2(a, b, sa, sb);
    if (threadIdx.x == SCAN_BLOCK - 1) {
        tot_cnt[blockIdx.x] = a;
        tot_seg[blockIdx.x] = b;
    }
}

// one workgroup: exclusive scan of the tile totals, in place; grand totals to tot[ntiles]
__global__ void __launch_bounds__(SCAN_BLOCK) msm_scan_mid(uint32_t* __restrict__ tot_cnt, uint32_t* __restrict__ tot_seg,
                                                          uint32_t ntiles) {
    __shared__ uint32_t sa[SCAN_BLOCK], sb[SCAN_BLOCK];
    uint32_t per = (ntiles + SCAN_BLOCK - 1) / SCAN_BLOCK;
    uint32_t lo = threadIdx.x * per, hi = min(lo + per, ntiles);
    uint32_t a = 0, b = 0;
    for (uint32_t i = lo; i < hi; i++) {
        a += tot_cnt[i];
        b += tot_seg[i];
    }
    uint32_t la = a, lb = b;
    block_inclusive_scan2(a, b, sa, sb);
    uint32_t ra = a - la, rb = b - lb;
    for (uint32_t i = lo; i < hi; i++) {
        uint32_t ca = tot_cnt[i], cb = tot_seg[i];
        tot_cnt[i] = ra;
        tot_seg[i] = rb;
        ra += ca;
        rb += cb;
    }
    if (threadIdx.x == SCAN_BLOCK - 1) {
        tot_cnt[ntiles] = a;
        tot_seg[ntiles] = b;
    }
}

// Also appends buckets with > big_nseg segments to the big list.
__global__ void __launch_bounds__(SCAN_BLOCK) msm_scan_final(const uint32_t* __restrict__ hist, uint32_t NB, uint32_t SEG,
                                                            const uint32_t* __restrict__ tot_cnt,
                                                            const uint32_t* __restrict__ tot_seg, uint32_t ntiles,
                                                            uint32_t* __restrict__ bucket_off,
                                                            uint32_t* __restrict__ seg_off,
                                                            uint32_t* __restrict__ big_count,
                                                            uint32_t* __restrict__ big_list, uint32_t big_nseg) {
    __shared__ uint32_t sa[SCAN_BLOCK], sb[SCAN_BLOCK];
    uint32_t lo = blockIdx.x * SCAN_TILE + threadIdx.x * SCAN_ITEMS;
    uint32_t hi = min(lo + SCAN_ITEMS, NB);
    uint32_t a = 0, b = 0;
    for (uint32_t i = lo; i < hi; i++) {
        a += hist[i];
        b += nseg_of(hist[i], SEG);
    }
    uint32_t la = a, lb = b;
    block_inclusive_scan2(a, b, sa, sb);
    uint32_t ra = tot_cnt[blockIdx.x] + a - la, rb = tot_seg[blockIdx.x] + b - lb;
    for (uint32_t i = lo; i < hi; i++) {
        uint32_t h = hist[i], ns = nseg_of(h, SEG);
        bucket_off[i] = ra;
        seg_off[i] = rb;
        ra += h;
        rb += ns;
        if (ns > big_nseg) big_list[atomicAdd(big_count, 1u)] = i;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        bucket_off[NB] = tot_cnt[ntiles];
        seg_off[NB] = tot_seg[ntiles];
    }
}

// bucket that owns segment `seg`: largest b with seg_off[b] <= seg (empty buckets share offsets; skip them)
__device__ __forceinline__ uint32_t bucket_of_segment(const uint32_t* __restrict__ seg_off, uint32_t NB, uint32_t seg) {
    uint32_t lo = 0, hi = NB;  // invariant: seg_off[lo] <= seg < seg_off[hi]
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (seg_off[mid] <= seg) lo = mid;
        else hi = mid;
    }
    return lo;
}

// Segment descriptors + ordering by length.  Lanes of a wave run in lock step, so a wave costs as much as its
// longest segment; bucket sizes are Poisson-spread (a wave of 64 unsorted segments idles ~25 % of its lane-cycles).
// Segments are therefore counting-sorted by length (longest first) and handed to lanes in that order.
static constexpr int ORD_THREADS = 256;
// (the SEG + 1 length counters / cursors sit one per 128-byte line, LEN_STRIDE: every workgroup of msm_seg_desc and
// msm_seg_order updates most of them; packed, those atomics queue up behind one another — see FLAT_CUR_STRIDE)
__global__ void __launch_bounds__(ORD_THREADS) msm_seg_desc(const uint32_t* __restrict__ bucket_off,
                                                            const uint32_t* __restrict__ seg_off, uint32_t NB,
                                                            uint32_t SEG, uint32_t* __restrict__ seg_start,
                                                            uint32_t* __restrict__ seg_len,
                                                            uint32_t* __restrict__ len_hist) {
    __shared__ uint32_t lh[SEG_MAX + 1];
    for (uint32_t i = threadIdx.x; i <= SEG; i += ORD_THREADS) lh[i] = 0;
    __syncthreads();
    uint32_t seg = blockIdx.x * ORD_THREADS + threadIdx.x;
    if (seg < seg_off[NB]) {
        uint32_t b = bucket_of_segment(seg_off, NB, seg);
        uint32_t s = seg - seg_off[b], ns = seg_off[b + 1] - seg_off[b];
        uint32_t o = bucket_off[b], cnt = bucket_off[b + 1] - o;
        uint32_t k = o + (uint32_t)(((uint64_t)cnt * s) / ns), e = o + (uint32_t)(((uint64_t)cnt * (s + 1)) / ns);
        seg_start[seg] = k;
        seg_len[seg] = e - k;
        atomicAdd(&lh[SEG - (e - k)], 1u);  // bin 0 = longest
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i <= SEG; i += ORD_THREADS)
        if (lh[i]) atomicAdd(&len_hist[i * LEN_STRIDE], lh[i]);
}
// (4096 segments per workgroup — r04; 256 before: every workgroup reserves a run per length class with a returning global atomic,
// and ~230 k of those on 129 addresses were the kernel's 24 us at 2^20 points)
static constexpr int ORD2_THREADS = 1024, ORD2_U = 4;
// r06: the exclusive scan of the SEG + 1 length counts — until r05 a single-wave kernel of its own between the sort and the
// accumulation of every MSM (5 us + a launch gap on the path of every job whose accumulation waits for its sort) — is done by
// every workgroup here for itself: three counters per lane of its first wave, a shuffle scan, the offsets in LDS.  The counts stay
// as the sort wrote them; the runs a workgroup reserves come from a second, zeroed array of cursors.  Workgroup 0 writes
// *nseg_live (the segments that hold entries: the lanes of work of the accumulation).
__global__ void __launch_bounds__(ORD2_THREADS) msm_seg_order(const uint32_t* __restrict__ seg_len,
                                                              const uint32_t* __restrict__ nseg_ptr, uint32_t SEG,
                                                              const uint32_t* __restrict__ len_hist /* SEG + 1 counts */,
                                                              uint32_t* __restrict__ len_cursor /* zeroed, advanced */,
                                                              uint32_t* __restrict__ nseg_live, uint32_t* __restrict__ order) {
    SWM_LIGHT_KERNEL();
    __shared__ uint32_t lh[SEG_MAX + 1], loff[SEG_MAX + 1];
    static_assert(SEG_MAX + 1 <= 3 * 64, "three counters per lane");
    for (uint32_t i = threadIdx.x; i <= SEG; i += ORD2_THREADS) lh[i] = 0;
    if (threadIdx.x < 64) {
        const uint32_t l = threadIdx.x;
        uint32_t c[3], sum = 0;
#pragma unroll
        for (int u = 0; u < 3; u++) {
            const uint32_t i = 3 * l + u;
            c[u] = i <= SEG ? len_hist[i * LEN_STRIDE] : 0u;
            sum += c[u];
        }
        uint32_t inc = sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t x = __shfl_up(inc, d, 64);
            if (l >= (uint32_t)d) inc += x;
        }
        uint32_t run = inc - sum;
#pragma unroll
        for (int u = 0; u < 3; u++) {
            const uint32_t i = 3 * l + u;
            if (i <= SEG) loff[i] = run;
            if (i == SEG && blockIdx.x == 0) *nseg_live = run;  // class SEG holds the empty segments: everything before it has entries
            run += c[u];
        }
    }
    __syncthreads();
    // *nseg_ptr bounds the segment INDICES in use; the flat schedule leaves indices without entries between the bins
    // (msm_flat_bin_sort): those are not handed to a lane
    const uint32_t nseg = *nseg_ptr;
    uint32_t bin[ORD2_U];
    bool live[ORD2_U];
#pragma unroll
    for (int u = 0; u < ORD2_U; u++) {
        const uint32_t seg = (blockIdx.x * ORD2_U + u) * ORD2_THREADS + threadIdx.x;
        const uint32_t len = seg < nseg ? seg_len[seg] & 0xffu : 0u;  // (bits 8 ..: the stride of an interleaved segment)
        live[u] = len != 0;
        bin[u] = SEG - len;
        if (live[u]) atomicAdd(&lh[bin[u]], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i <= SEG; i += ORD2_THREADS) {
        uint32_t v = lh[i];
        if (v) lh[i] = loff[i] + atomicAdd(&len_cursor[i * LEN_STRIDE], v);  // reserve a run; lh[i] = its start
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < ORD2_U; u++)
        if (live[u]) order[atomicAdd(&lh[bin[u]], 1u)] = (blockIdx.x * ORD2_U + u) * ORD2_THREADS + threadIdx.x;
}

// Dominant kernel: one lane per segment, XYZZ accumulator in registers, affine bases gathered from HBM/L2.
//
// The inner loop runs in the 28-bit lazy-carry representation of fq28.cuh (bases28 = the same points with both
// coordinates pre-multiplied by 2^8, i.e. in Montgomery radix 2^392).  Bounds, per EFD madd-2008-s step
// (N: normalised limbs, value < 2p;  X1: normalised, < 18p;  Y1: normalised, < 6p;  ZZ1, ZZZ1: N):
//   U2 = x2 ZZ1, S2 = y2 ZZZ1                              N          (y2 may be 4p - y2: limbs < 2^30, value < 4p)
//   P  = U2 + 32p - X1                                     limbs < 2^30, value in (14p, 34p)
//   R  = S2 +  8p - Y1                                     limbs < 2^30, value in ( 2p, 10p)
//   PP = P^2, PPP = P PP, Q = X1 PP, RR = R^2              N          (products of two lazy values: 14 * 2^60 < 2^64)
//   X3 = normalise(RR + 16p - PPP - 2Q)                    normalised, value in (10p, 18p)
//   V  = Q + 32p - X3                                      limbs < 2^30, value in (14p, 24p)
//   Y3 = (R V + (8p - Y1) PPP) 2^-392, ONE reduction        N          (fq28_mul2: limbs 1.5 2^29 x 1.5 2^29 + 2^29 x 2^28)
//   ZZ3 = ZZ1 PP, ZZZ3 = ZZZ1 PPP                          N
// P = 0 mod p (doubling or cancellation: the point equals +-accumulator) is detected on PP, which is an N value; the
// segment is then recomputed with the fully reducing 32-bit-limb adder (cold path).
struct Acc28 {
    Fq28 x, y, zz, zzz;
};
__device__ __forceinline__ bool madd28(Acc28& a, const Fq28& x2, const Fq28& y2) {
    Fq28 u2 = fq28_mul(x2, a.zz);
    Fq28 s2 = fq28_mul(y2, a.zzz);
    Fq28 p = FQ28_SUB(u2, a.x, SPREAD32);
    Fq28 r = FQ28_SUB(s2, a.y, SPREAD8);
    Fq28 pp = fq28_mul(p, p);
    if (fq28_is_zero_mod_p(pp)) return false;
    Fq28 ppp = fq28_mul(p, pp);
    Fq28 q = fq28_mul(a.x, pp);
    Fq28 rr = fq28_mul(r, r);
    Fq28 x3;
#pragma unroll
    for (int i = 0; i < 14; i++) x3.l[i] = rr.l[i] + Fq28Consts::SPREAD16_3[i] - ppp.l[i] - q.l[i] - q.l[i];
    x3 = fq28_normalize(x3);
    Fq28 v = FQ28_SUB(q, x3, SPREAD32);
    Fq28 ny;  // 8p - Y1 > 0, limbs < 2^29
#pragma unroll
    for (int i = 0; i < 14; i++) ny.l[i] = Fq28Consts::SPREAD8[i] - a.y.l[i];
    a.y = fq28_mul2(r, v, ny, ppp);  // R V - Y1 PPP with one reduction
    a.x = x3;
    a.zz = fq28_mul(a.zz, pp);
    a.zzz = fq28_mul(a.zzz, ppp);
    return true;
}
__global__ void __launch_bounds__(256, 3) msm_accumulate(const G1Affine* __restrict__ bases,
                                                      const G1Affine* __restrict__ bases28,
                                                      const uint32_t* __restrict__ sorted,
                                                      const uint32_t* __restrict__ seg_start,
                                                      const uint32_t* __restrict__ seg_len,
                                                      const uint32_t* __restrict__ order,
                                                      const uint32_t* __restrict__ nseg_ptr,
                                                      G1XYZZ* __restrict__ partial) {
    // grid-stride over the length-sorted segments: the host may launch fewer workgroups than segments (SWM_ACC_WGS) so that
    // the kernel leaves register-file room on every SIMD for the kernels that run beside it
    const uint32_t nseg_total = *nseg_ptr;
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < nseg_total; t += gridDim.x * blockDim.x) {
    uint32_t seg = order[t];
    const uint32_t k0 = seg_start[seg], e = k0 + seg_len[seg];
    if (k0 >= e) {
        p28_store(partial[seg], p28_identity());
        continue;
    }
    bool ok = true;
    Acc28 acc;
    {
        uint32_t ent = sorted[k0];
        G1Affine p = bases28[ent & 0x7fffffffu];
        acc.x = fq28_unpack(p.x);
        Fq28 y = fq28_unpack(p.y);
        if (ent >> 31) {  // -y = 4p - y, brought back to normalised limbs (value < 4p < 6p)
            Fq28 z;
#pragma unroll
            for (int i = 0; i < 14; i++) z.l[i] = Fq28Consts::SPREAD4[i] - y.l[i];
            y = fq28_normalize(z);
        }
        acc.y = y;
        acc.zz = fq28_const(Fq28Consts::ONE);
        acc.zzz = acc.zz;
    }
    for (uint32_t k = k0 + 1; k < e && ok; k++) {
        uint32_t ent = sorted[k];
        G1Affine p = bases28[ent & 0x7fffffffu];
        Fq28 x2 = fq28_unpack(p.x), y2 = fq28_unpack(p.y);
        if (ent >> 31) {
#pragma unroll
            for (int i = 0; i < 14; i++) y2.l[i] = Fq28Consts::SPREAD4[i] - y2.l[i];  // limbs < 2^29, value < 4p
        }
        ok = madd28(acc, x2, y2);
    }
    if (ok) {
        // partial sums stay in the 28-bit domain (radix 2^392, "point form" of fq28.cuh) for the bucket stage
        P28 out{acc.x, acc.y, acc.zz, acc.zzz};
        p28_store(partial[seg], out);
        continue;
    }
    // cold path: a point met +-(the running sum); redo this segment with the fully reducing adder, then move the
    // result into the 28-bit domain (coordinates x 2^8)
    G1XYZZ acc32 = g1_xyzz_identity();
    for (uint32_t k = k0; k < e; k++) {
        uint32_t ent = sorted[k];
        G1Affine p;
        if (bases) {
            p = bases[ent & 0x7fffffffu];
        } else {  // table rows exist only in the scaled form: x 2^-8 (a Montgomery product with the integer 2^376)
            p = bases28[ent & 0x7fffffffu];
            Fq d256 = fp_zero<Fq>();
            d256.v[11] = 0x01000000u;
            if (!g1_is_inf(p)) {
                p.x = fp_mul(p.x, d256);
                p.y = fp_mul(p.y, d256);
            }
        }
        if (ent >> 31) p.y = fp_neg(p.y);
        g1_add_mixed(acc32, p);
    }
    const uint32_t k256[12] = SWM_FQ_SCALE256_MONT;
    Fq c;
#pragma unroll
    for (int j = 0; j < 12; j++) c.v[j] = k256[j];
    acc32.x = fp_mul(acc32.x, c);
    acc32.y = fp_mul(acc32.y, c);
    acc32.zz = fp_mul(acc32.zz, c);   // identity (zz = 0) stays exactly zero
    acc32.zzz = fp_mul(acc32.zzz, c);
    partial[seg] = acc32;
    }
}

// The same kernel over a twisted Edwards table (msm_table_build_te): rows are the affine triples (y - x, y + x, 2 d x y), each
// coordinate as the fourteen 28-bit limbs the multiplier takes, in a 64-byte sector of its own (192 B per row, consumed as loaded);
// the accumulator is an extended point and every addition is the UNIFIED 7-product law of te28_madd_row — no doubling /
// cancellation test, no cold path: 3 223 VALU instructions per addition of which 2 646 are multiply-adds (SQ pass, r04:
// profiles/r04_pmc_sq_msm_accumulate.json; XYZZ: 4 817), 139 VGPRs, no scratch, three waves per SIMD.  Rows go HBM -> registers,
// not through LDS: the kernel is issue-bound, a row is read once by one lane.  It is the one kernel of a
// proof WITHOUT an issue priority (ff.cuh SWM_LIGHT_KERNEL): the filler everything else runs beside.  Partial sums leave as
// extended points (X, Y, T, Z in the four slots of a G1XYZZ).  Algorithmic bytes per point: 96 B base + 32 B scalar.
__global__ void __launch_bounds__(256, SWM_TE_EARLY_LOADS ? 3 : 4) msm_accumulate_te(const G1TE* __restrict__ rows,
                                                           const uint32_t* __restrict__ sorted,
                                                           const uint32_t* __restrict__ seg_start,
                                                           const uint32_t* __restrict__ seg_len,
                                                           const uint32_t* __restrict__ order,
                                                           const uint32_t* __restrict__ nseg_ptr,
                                                           G1XYZZ* __restrict__ partial) {
    const uint32_t nseg_total = *nseg_ptr;
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < nseg_total; t += gridDim.x * blockDim.x) {
        const uint32_t seg = order[t];
        // entries of the segment: sorted[k0 + j * stride], j < len (stride 1 unless the bucket's segments interleave)
        const uint32_t k0 = seg_start[seg], sl = seg_len[seg], len = sl & 0xffu, stride = max(1u, sl >> 8);
        if (len == 0) {
            te28_store_identity(partial[seg]);
            continue;
        }
        T28 acc;
        {
            const uint32_t ent = sorted[k0];
            const G1TE* row = rows + (ent & 0x7fffffffu);
            acc = te28_from_row(te28_load_coord(row->ymx), te28_load_coord(row->ypx), te28_load_coord(row->kt), (ent >> 31) != 0);
        }
        // the entry of the NEXT addition is read while this one computes (index clamped to the segment: no branch), so that an
        // addition waits for one memory round trip — its row — and not for two in a row
        uint32_t ent = sorted[k0 + min(1u, len - 1) * stride];
        for (uint32_t j = 1; j < len; j++) {
            const uint32_t next = sorted[k0 + min(j + 1, len - 1) * stride];
            te28_madd_row(acc, rows + (ent & 0x7fffffffu), (ent >> 31) != 0);
            ent = next;
        }
        partial[seg].x = fq28_pack(acc.x);
        partial[seg].y = fq28_pack(acc.y);
        partial[seg].zz = fq28_pack(acc.t);
        partial[seg].zzz = fq28_pack(acc.z);
    }
}

// The accumulation of SMALL MSMs (low-latency schedule, r04): four lanes per segment, the accumulator one coordinate per lane
// (te28_quad_madd_row): the chain of a segment of 8 entries is 16 products long instead of 50, on a chip that such a job
// cannot fill anyway.  Same partial sums, bit for bit.
__global__ void __launch_bounds__(256) msm_accumulate_te_quad(const G1TE* __restrict__ rows, const uint32_t* __restrict__ sorted,
                                                              const uint32_t* __restrict__ seg_start,
                                                              const uint32_t* __restrict__ seg_len,
                                                              const uint32_t* __restrict__ order,
                                                              const uint32_t* __restrict__ nseg_ptr, G1XYZZ* __restrict__ partial) {
    const uint32_t nseg_total = *nseg_ptr;
    const unsigned q = threadIdx.x & 3u;
    for (uint32_t t = (blockIdx.x * blockDim.x + threadIdx.x) >> 2; t < nseg_total; t += (gridDim.x * blockDim.x) >> 2) {
        const uint32_t seg = order[t];
        const uint32_t k0 = seg_start[seg], sl = seg_len[seg], len = sl & 0xffu, stride = max(1u, sl >> 8);
        if (len == 0) {
            te28_quad_store_identity(&partial[seg], q);
            continue;
        }
        uint32_t ent = sorted[k0];
        Fq28 own = te28_quad_from_row(rows + (ent & 0x7fffffffu), (ent >> 31) != 0, q);
        ent = sorted[k0 + min(1u, len - 1) * stride];
        for (uint32_t j = 1; j < len; j++) {
            const uint32_t next = sorted[k0 + min(j + 1, len - 1) * stride];
            te28_quad_madd_row(own, rows + (ent & 0x7fffffffu), (ent >> 31) != 0, q);
            ent = next;
        }
        reinterpret_cast<Fq*>(&partial[seg])[q] = fq28_pack(own);  // X, Y, T (slot zz), Z (slot zzz)
    }
}

// ---------------------------------------------------------------------------------------------- bucket stage (28-bit domain)
// Register budget of the bucket stage: a general addition keeps two points (2 x 56 VGPRs) plus ~8 temporaries
// (112 VGPRs) live; a third live point spills to scratch, i.e. to HBM-backed private memory with nothing to hide the
// latency at one wave per SIMD (measured: 5x slower).  So each lane parks the point it is not currently adding in LDS
// (packed 192-B slots) and at most two points are ever in registers.
// (the unfenced multiplier with a 512-register budget — amdgpu_waves_per_eu(1, 1) — was measured in r02: the bucket stage
// takes the same 0.85 ms either way)
__device__ __forceinline__ void p28_add_ool(P28& a, const P28& q) { p28_add<MulFenced>(a, q); }

// Streamed form of the general addition for the bucket stage:  *dst = *pa + *pq  (dst may be pa), operands in memory
// (LDS slots or HBM, packed 192-B points of the 28-bit domain).  Coordinates are loaded when they are needed and results
// stored as soon as they exist, so that about eight field elements are live at the peak instead of two whole points plus
// temporaries: the kernel then fits the 168-VGPR footprint of an msm_accumulate wave.  That matters more than the
// instruction count: with 248 VGPRs a bucket-stage wave could only be placed on a SIMD that holds at most ONE
// accumulation wave, i.e. never while an accumulation grid was in flight — the r02 timeline showed all 12.6 ms of
// bucket stage per 2^20 proof running with no accumulation beside it.  Same formulas and bounds as p28_add_fast.
// *dst = 2 * *pa, streamed (EFD dbl-2008-s-1, a = 0; the identity stays the identity: zz, zzz are only multiplied)
__device__ __forceinline__ void p28_slot_dbl(G1XYZZ* dst, const G1XYZZ* pa) {
    typedef MulFenced M;
    Fq28 y = fq28_unpack(pa->y);
    Fq28 u = fq28_add(y, y);                    // limbs < 2^29, value < 12p
    Fq28 v = M::sqr(u);
    Fq28 w = M::mul(u, v);
    Fq28 ny;                                    // 8p - Y1 > 0 (Y1 < 6p), limbs < 2^29
#pragma unroll
    for (int i = 0; i < 14; i++) ny.l[i] = Fq28Consts::SPREAD8[i] - y.l[i];
    Fq28 zz3 = M::mul(v, fq28_unpack(pa->zz));
    Fq28 zzz3 = M::mul(w, fq28_unpack(pa->zzz));
    Fq28 x = fq28_unpack(pa->x);
    Fq28 sv = M::mul(x, v);
    Fq28 xx = M::sqr(x);
    dst->zz = fq28_pack(zz3);
    dst->zzz = fq28_pack(zzz3);
    Fq28 m = fq28_add(fq28_add(xx, xx), xx);   // limbs < 3 * 2^28, value < 6p
    Fq28 mm = M::sqr(m);
    Fq28 x3;
#pragma unroll
    for (int i = 0; i < 14; i++) x3.l[i] = mm.l[i] + Fq28Consts::SPREAD16_3[i] - sv.l[i] - sv.l[i];
    x3 = fq28_normalize(x3);                   // (12p, 18p)
    dst->x = fq28_pack(x3);
    dst->y = fq28_pack(M::mul2(m, FQ28_SUB(sv, x3, SPREAD32), ny, w));  // M (S - X3) - W Y1, one reduction
}
__device__ __forceinline__ bool fq28_all_zero(const Fq28& a) {
    uint32_t z = 0;
#pragma unroll
    for (int i = 0; i < 14; i++) z |= a.l[i];
    return z == 0;
}
__device__ __forceinline__ void p28_slot_add(G1XYZZ* dst, const G1XYZZ* pa, const G1XYZZ* pq) {
    typedef MulFenced M;
    Fq28 azz = fq28_unpack(pa->zz), qzz = fq28_unpack(pq->zz);
    if (fq28_all_zero(qzz)) {  // q is the identity
        if (dst != pa) *dst = *pa;
        return;
    }
    if (fq28_all_zero(azz)) {  // a is the identity
        *dst = *pq;
        return;
    }
    Fq28 u1 = M::mul(fq28_unpack(pa->x), qzz);
    Fq28 p = M::mul(fq28_unpack(pq->x), azz);  // U2
    p = FQ28_SUB(p, u1, SPREAD4);               // P = U2 - U1, limbs < 2^30, value in (2p, 6p)
    Fq28 t = M::mul(azz, qzz);
    Fq28 pp = M::sqr(p);
    if (fq28_is_zero_mod_p(pp)) {  // q = +-a: doubling or cancellation, decided on (S2 - S1)^2 (cold)
        Fq28 c1 = M::mul(fq28_unpack(pa->y), fq28_unpack(pq->zzz)), c2 = M::mul(fq28_unpack(pq->y), fq28_unpack(pa->zzz));
        if (fq28_is_zero_mod_p(M::sqr(FQ28_SUB(c2, c1, SPREAD4)))) p28_slot_dbl(dst, pa);
        else p28_store(*dst, p28_identity());
        return;
    }
    // the x / zz side is finished first (its inputs die early): of a only y and zzz are still to be read, and those slots
    // are written last
    dst->zz = fq28_pack(M::mul(t, pp));
    Fq28 ppp = M::mul(p, pp);
    Fq28 qq = M::mul(u1, pp);
    Fq28 azzz = fq28_unpack(pa->zzz), qzzz = fq28_unpack(pq->zzz);
    Fq28 s1 = M::mul(fq28_unpack(pa->y), qzzz);
    Fq28 r = M::mul(fq28_unpack(pq->y), azzz);  // S2
    r = FQ28_SUB(r, s1, SPREAD4);                // R = S2 - S1
    Fq28 t2 = M::mul(azzz, qzzz);
    dst->zzz = fq28_pack(M::mul(t2, ppp));
    Fq28 rr = M::sqr(r);
    Fq28 x3;
#pragma unroll
    for (int i = 0; i < 14; i++) x3.l[i] = rr.l[i] + Fq28Consts::SPREAD16_3[i] - ppp.l[i] - qq.l[i] - qq.l[i];
    x3 = fq28_normalize(x3);  // (10p, 18p)
    dst->x = fq28_pack(x3);
    Fq28 ns1;  // 4p - S1 > 0 (S1 < 2p), limbs < 2^29
#pragma unroll
    for (int i = 0; i < 14; i++) ns1.l[i] = Fq28Consts::SPREAD4[i] - s1.l[i];
    dst->y = fq28_pack(M::mul2(r, FQ28_SUB(qq, x3, SPREAD32), ns1, ppp));  // R (Q - X3) - S1 PPP, one reduction
}

// The two point forms of the bucket stage: XYZZ on the Weierstrass curve (per-window schedule, XYZZ tables) and extended
// twisted Edwards (TE tables).  Each kernel below is instantiated once per form; a launch handles jobs of one form.
// LANES: hardware lanes that share one chain of the bucket stage (msm_bucket_reduce); `q` = the lane's index among them.
// One LDS slot of the bucket stage: a packed point PLUS 16 bytes (r06).  At 192 bytes the slots of consecutive lanes lie 48 dwords
// = 16 banks apart, so the 128-bit reads and writes of a wave fall on 8 of the 32 banks: 87 % of the kernel's LDS cycles were bank
// conflicts (SQ_LDS_BANK_CONFLICT 51.1 M of SQ_LDS_IDX_ACTIVE 58.7 M per launch) and the four lone waves of a workgroup queued up
// behind one another's operand loads — the "26 % of wave-cycles parked" of profiles/r05_pmc_sq_msm_bucket_reduce_and_sort.json.
// At 208 bytes (52 dwords: 20 banks) the 128-bit accesses of 8 consecutive lanes cover all 32 banks.  (3 x 256 + 1) x 208 B =
// 159 952 B: just inside the 160 KB of a CU.
struct alignas(16) RedSlot {
    G1XYZZ p;
    uint32_t pad[4];
};
static_assert(sizeof(RedSlot) == 208, "slot stride");
// ... and the slot of the QUAD forms, where a lane reads ONE coordinate (48 bytes) of its chain's slot: there the plain 192 bytes
// are the conflict-free stride (the sixteen 128-bit accesses of four chains fall on every 4-bank group exactly twice; 208 bytes
// puts four of them on one group)
struct alignas(16) PlainSlot {
    G1XYZZ p;
};
struct FormXYZZ {
    using Slot = RedSlot;
    static constexpr int LANES = 1;
    static constexpr bool QUAD_TREE = false;
    static __device__ __forceinline__ void slot_add(G1XYZZ* dst, const G1XYZZ* pa, const G1XYZZ* pq, unsigned = 0) { p28_slot_add(dst, pa, pq); }
    static __device__ __forceinline__ void slot_dbl(G1XYZZ* dst, const G1XYZZ* pa, unsigned = 0) { p28_slot_dbl(dst, pa); }
    static __device__ __forceinline__ void copy(G1XYZZ* dst, const G1XYZZ* src, unsigned = 0) { *dst = *src; }
    static __device__ __forceinline__ void store_identity(G1XYZZ& m, unsigned = 0) { p28_store(m, p28_identity()); }
    static __device__ __forceinline__ void store_384(G1XYZZ& m, const G1XYZZ& slot) { p28_store_384(m, p28_load(slot)); }
};
struct FormTE {
    using Slot = RedSlot;
    static constexpr int LANES = 1;
    static constexpr bool QUAD_TREE = true;  // the narrow steps of the final tree switch to four lanes per sum
    static __device__ __forceinline__ void slot_add(G1XYZZ* dst, const G1XYZZ* pa, const G1XYZZ* pq, unsigned = 0) { te28_slot_add(dst, pa, pq); }
    static __device__ __forceinline__ void slot_dbl(G1XYZZ* dst, const G1XYZZ* pa, unsigned = 0) { te28_slot_add(dst, pa, pa); }  // unified law
    static __device__ __forceinline__ void copy(G1XYZZ* dst, const G1XYZZ* src, unsigned = 0) { *dst = *src; }
    static __device__ __forceinline__ void store_identity(G1XYZZ& m, unsigned = 0) { te28_store_identity(m); }
    static __device__ __forceinline__ void store_384(G1XYZZ& m, const G1XYZZ& slot) { te28_store_384(m, slot); }
};
// Twisted Edwards with every chain worked by a quad of lanes (te28_quad_add: three products per lane and step instead of
// nine): the bucket stage of SMALL MSMs, where the chain's latency is all there is.
struct FormTEQuad {
    using Slot = PlainSlot;
    static constexpr int LANES = 4;
    static constexpr bool QUAD_TREE = false;
    static __device__ __forceinline__ void slot_add(G1XYZZ* dst, const G1XYZZ* pa, const G1XYZZ* pq, unsigned q) { te28_quad_add(dst, pa, pq, q); }
    static __device__ __forceinline__ void slot_dbl(G1XYZZ* dst, const G1XYZZ* pa, unsigned q) { te28_quad_add(dst, pa, pa, q); }
    static __device__ __forceinline__ void copy(G1XYZZ* dst, const G1XYZZ* src, unsigned q) { te28_quad_copy(dst, src, q); }
    static __device__ __forceinline__ void store_identity(G1XYZZ& m, unsigned q) { te28_quad_store_identity(&m, q); }
    static __device__ __forceinline__ void store_384(G1XYZZ& m, const G1XYZZ& slot) { te28_store_384(m, slot); }
};

// Oversized buckets (> BIG_NSEG segments: structured scalars) are folded first, one workgroup each: strided partial
// sums + LDS tree; the result replaces the bucket's first partial.
template <class Form>
__global__ void __launch_bounds__(RED_BLOCK) msm_big_bucket_sum(G1XYZZ* __restrict__ partial,
                                                                const uint32_t* __restrict__ seg_off,
                                                                const uint32_t* __restrict__ hist, uint32_t SEG,
                                                                const uint32_t* __restrict__ big_count,
                                                                const uint32_t* __restrict__ big_list, unsigned log_g) {
    SWM_TAIL_KERNEL();
    extern __shared__ __align__(16) unsigned char smem_raw[];
    G1XYZZ* sm = reinterpret_cast<G1XYZZ*>(smem_raw);
    const uint32_t nbig = *big_count;
    // 2^log_g lanes per listed bucket (a whole workgroup for the oversized buckets of structured scalars; 4 .. 64 lanes
    // in the low-latency schedule of small MSMs, where EVERY bucket with more than two segments is folded here)
    const uint32_t G = 1u << log_g, per = RED_BLOCK >> log_g, g = threadIdx.x & (G - 1);
    for (uint32_t j0 = blockIdx.x * per; j0 < nbig; j0 += gridDim.x * per) {
        const uint32_t j = j0 + (threadIdx.x >> log_g);
        uint32_t s = 0, e = 0;
        if (j < nbig) {
            uint32_t b = big_list[j];
            s = seg_off[b];
            e = s + nseg_of(hist[b], SEG);
        }
        Form::store_identity(sm[threadIdx.x]);
#pragma unroll 1
        for (uint32_t k = s + g; k < e; k += G) Form::slot_add(&sm[threadIdx.x], &sm[threadIdx.x], &partial[k]);
        __syncthreads();
#pragma unroll 1
        for (uint32_t stride = G / 2; stride > 0; stride >>= 1) {
            if (g < stride) Form::slot_add(&sm[threadIdx.x], &sm[threadIdx.x], &sm[threadIdx.x + stride]);
            __syncthreads();
        }
        if (g == 0 && j < nbig) partial[s] = sm[threadIdx.x];
        __syncthreads();
    }
}

// Per window: sum_b (b+1) S_b with S_b = sum of the bucket's partials, fused in one kernel.
// grid = (nblk, nwin), RED_BLOCK lanes, lane t owns the m = 2^log_m buckets [lo, lo+m), lo = (blk RED_BLOCK + t) m:
//     acc_t = sum_j (j+1) S_{lo+j},  run_t = sum_j S_{lo+j}                 (running sums, 2m adds)
//     sum over the workgroup of (lo - lo_blk) run_t = m sum_{t>=1} Suffix_t,  Suffix_t = sum_{u>=t} run_u   (LDS suffix scan)
//     A_blk = sum_t acc_t + m sum_{t>=1} Suffix_t  (one LDS tree),  R_blk = Suffix_0
// The host finishes with sum_blk [A_blk + blk (RED_BLOCK m) R_blk] — a handful of additions per window.
// No scalar multiplication by the bucket offset is needed: the serial chain is ~3m + 20 group operations.
//
// Code layout matters as much as the arithmetic here.  A general addition is ~6 k instructions (48 KB); with one
// inlined copy per use (running sums, scan, tree: five sites) the kernel was 480 KB of straight-line code executed
// once per step at one wave per SIMD, i.e. every instruction came from L2 through a 64 KB instruction cache shared by
// two CUs (measured ~32 us per addition).  So every step of every phase is expressed as the same micro-operation
//     slot[dst] <- slot[dst] + *src          (dst: an LDS slot; src: an LDS slot or a partial sum in HBM)
// issued from ONE loop around ONE copy of the adder; the phases only differ in how (dst, src, active) are chosen.
// LDS: two packed slots per lane (running sum, weighted sum) + one for R = 96 KB per workgroup.
//
// Batch form: the bucket stages of up to TAIL_MAX MSMs (the commitments of one prover round) run as ONE launch,
// blockIdx.z = job.  The stage is a ~25-deep chain of ~20 us group operations whatever the number of buckets, so the
// jobs of a round share one chain latency instead of paying it one after the other.
static constexpr int TAIL_MAX = 4;
struct TailJob {
    const G1XYZZ* partial;
    const uint32_t *seg_off, *hist;  // first segment and entry count of every bucket: its segments are seg_off[b] .. + ceil(hist[b] / seg)
    uint32_t seg;
    G1XYZZ* out;
    G1XYZZ* acc;  // msm_bucket_reduce_low: the weighted sums of the lanes, one slot per lane in HBM / L2 (red_blocks x RB per window)
    unsigned log_m, red_blocks, big_nseg;
    unsigned blk_lo, blk_hi, blk_low;   // workgroups of this rank's bucket share (MsmJob); the others only emit the identity
    const uint32_t *status, *entries;  // device status words of the job, forwarded to ...
    uint32_t* host_flags;              // ... the tail of its pinned result slot
    WinLayout L;
};
struct TailBatch {
    TailJob j[TAIL_MAX];
};
// (RB = chains per workgroup; a chain is one lane, or Form::LANES of them: RB x LANES threads)
template <int RB, class Form>
__global__ void __launch_bounds__(RB * Form::LANES) msm_bucket_reduce(TailBatch batch) {
    SWM_TAIL_KERNEL();
    using Slot = typename Form::Slot;
    const TailJob& job = batch.j[blockIdx.z];
    if (blockIdx.x >= job.red_blocks || blockIdx.y >= job.L.nwin) return;
    const G1XYZZ* __restrict__ partial = job.partial;
    const uint32_t* __restrict__ seg_off = job.seg_off;
    const WinLayout& L = job.L;
    const unsigned log_m = job.log_m;
    G1XYZZ* __restrict__ out = job.out;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    // LDS: three packed slots per lane — the running sum in two copies (the scan and the shift read a NEIGHBOUR's slot
    // while every lane rewrites its own, so those steps go from one copy to the other) and the weighted sum — plus one
    // slot for R_blk: (3 x 256 + 1) x 192 B = 144 KB per workgroup.
    Slot* sm_run = reinterpret_cast<Slot*>(smem_raw);
    Slot* sm_alt = sm_run + RB;
    Slot* sm_acc = sm_alt + RB;
    Slot* sm_r = sm_acc + RB;
    constexpr unsigned Q = Form::LANES;
    const uint32_t w = blockIdx.y, t = threadIdx.x / Q, q = threadIdx.x % Q;
    if (!((blockIdx.x >= job.blk_lo && blockIdx.x < job.blk_hi) || blockIdx.x < job.blk_low)) {
        // not a workgroup of this rank's bucket share: its buckets are empty here, the host fold sees the identity
        if (threadIdx.x < Q) Form::store_identity(sm_run[0].p, q);
        __syncthreads();
        if (threadIdx.x == 0) {
            size_t o = ((size_t)w * job.red_blocks + blockIdx.x) * 2;
            Form::store_384(out[o], sm_run[0].p);
            Form::store_384(out[o + 1], sm_run[0].p);
            if (blockIdx.x == 0 && w == 0 && job.host_flags) {
                job.host_flags[0] = job.status[0];
                job.host_flags[1] = *job.entries;
                job.host_flags[2] = job.status[1];
            }
        }
        return;
    }
    const uint32_t B = 1u << (L.c[w] - 1), m = 1u << log_m;
    const uint32_t lo = (blockIdx.x * RB + t) << log_m;
    const uint32_t base = L.boff[w];
    Form::store_identity(sm_run[t].p, q);
    Form::store_identity(sm_acc[t].p, q);
    // phase-1 sequencer of this lane: buckets b = hi-1 .. lo; per bucket "run += partial[s]" for its segments, then
    // "acc += run"
    uint32_t b = min(lo + m, B), s = 0, e = 0;
    bool walking = lo < B;
    if (walking) {
        b--;
        s = seg_off[base + b];
        e = s + nseg_of(job.hist[base + b], job.seg);
        if (e - s > job.big_nseg) e = s + 1;  // already folded into the first partial
    }
    enum { WALK = 0, SCAN = 1, SHIFT = 2, FOLD = 3, TREE = 4 };
    int phase = WALK;
    uint32_t d = 1;  // scan distance / tree stride
    __syncthreads();
#pragma unroll 1
    for (;;) {
        // one micro-operation per iteration: *dst = *pa + *pq (act), or *dst = *pa (copy: inactive lane of a scan step)
        G1XYZZ* dst = &sm_run[t].p;
        const G1XYZZ* pa = &sm_run[t].p;
        const G1XYZZ* pq = &sm_run[t].p;
        bool act = false, copy = false;
        if (phase == WALK) {
            // (r06) the walk has no meeting point but its end: a lane works on slots of its own, so every WAVE walks at its own
            // pace — its LDS traffic no longer in step with the other three waves' — and meets the workgroup once, when none of
            // its lanes has a bucket left (until r05: a workgroup-wide vote, three barriers, in front of every step)
            if (__ballot(walking) == 0) {
                __syncthreads();
                phase = SCAN;
                continue;
            }
            if (walking) {
                act = true;
                if (s < e) {
                    pq = &partial[s++];
                } else {
                    dst = &sm_acc[t].p;
                    pa = &sm_acc[t].p;
                    if (b == lo) {
                        walking = false;
                    } else {
                        b--;
                        s = seg_off[base + b];
                        e = s + nseg_of(job.hist[base + b], job.seg);
                        if (e - s > job.big_nseg) e = s + 1;
                    }
                }
            }
        } else if (phase == SCAN) {  // inclusive suffix scan of run over the workgroup (Hillis-Steele), copy to copy
            if (d >= RB) {
                phase = SHIFT;
                continue;
            }
            dst = &sm_alt[t].p;
            act = t + d < RB;
            copy = !act;
            if (act) pq = &sm_run[t + d].p;
            d <<= 1;
        } else if (phase == SHIFT) {  // run_t <- m Suffix_{t+1} (into the other copy); R_blk = Suffix_0 is parked
            if (t + 1 < RB) Form::copy(&sm_alt[t].p, &sm_run[t + 1].p, q);
            else Form::store_identity(sm_alt[t].p, q);
            if (t == 0) Form::copy(&sm_r[0].p, &sm_run[0].p, q);
#pragma unroll 1
            for (unsigned i = 0; i < log_m; i++) Form::slot_dbl(&sm_alt[t].p, &sm_alt[t].p, q);
            __syncthreads();
            Slot* x = sm_run;
            sm_run = sm_alt;
            sm_alt = x;
            phase = FOLD;
            continue;
        } else if (phase == FOLD) {  // acc_t += m Suffix_{t+1}; summed over t this is A_blk
            dst = &sm_acc[t].p;
            pa = &sm_acc[t].p;
            act = true;
            phase = TREE;
            d = RB / 2;
        } else {
            if (d == 0) break;
            if constexpr (Form::QUAD_TREE) {
                // a tree step with at most RB / 4 sums left: four lanes per sum (te28_quad_add), a third of the step's latency
                if (d <= RB / 4) {
                    const uint32_t c = threadIdx.x >> 2;
                    if (c < d) te28_quad_add(&sm_acc[c].p, &sm_acc[c].p, &sm_acc[c + d].p, threadIdx.x & 3u);
                    __syncthreads();
                    d >>= 1;
                    continue;
                }
            }
            dst = &sm_acc[t].p;
            pa = &sm_acc[t].p;
            act = t < d;
            if (act) pq = &sm_acc[t + d].p;  // the lanes t + d .. are idle in this step: nobody rewrites what is read
            d >>= 1;
        }
        if (act) Form::slot_add(dst, pa, pq, q);
        else if (copy) Form::copy(dst, pa, q);
        if (phase != WALK) __syncthreads();  // (a walk step touches the lane's own slots only)
        if (dst == &sm_alt[t].p) {  // a scan step went from one copy of the running sums to the other (uniform per step)
            Slot* x = sm_run;
            sm_run = sm_alt;
            sm_alt = x;
        }
    }
    if (threadIdx.x == 0) {
        // `out` is the job's PINNED host slot (zero-copy: 384 B per workgroup over the fabric instead of three
        // stream-ordered copies per job after the kernel — ~25 us per job between a round's last kernel and its challenge)
        size_t o = ((size_t)w * job.red_blocks + blockIdx.x) * 2;
        Form::store_384(out[o], sm_acc[0].p);
        Form::store_384(out[o + 1], sm_r[0].p);
        if (blockIdx.x == 0 && w == 0 && job.host_flags) {
            job.host_flags[0] = job.status[0];
            job.host_flags[1] = *job.entries;
            job.host_flags[2] = job.status[1];  // points with a zero scalar / identity base
        }
    }
}

// The bucket stage of twisted Edwards jobs with a THIRD of the LDS (r05).  msm_bucket_reduce keeps three slots per lane in LDS
// (running sum twice — the scan goes from one copy to the other — and the weighted sum): 144 KB per 256-lane workgroup, i.e.
// ONE workgroup per CU and one wave per SIMD, where a lone wave issues an instruction every ~5.5 cycles (SQ utilisation 0.62,
// profiles/r04_pmc_sq_msm_bucket_reduce_and_sort.json) and nothing else that wants LDS fits beside it.  Here
//   * the running sum is ONE slot per lane: the suffix scan works in place, every lane's loads ahead of a barrier and its
//     stores behind it (te28_slot_add_sync) — the unified law reads all eight input coordinates before it writes any;
//   * the weighted sum of a lane lives in HBM / L2 (job.acc: touched once per bucket, 384 B per step against ~4 400
//     instructions) and moves into the lane's LDS slot with the step that folds the suffix into it; the tree runs there;
// (256 + 1) x 192 B = 49 KB per workgroup, 171 VGPRs: two workgroups per CU by registers (three: 35 spills, slower), and a stage
// no longer owns its CU.  Same micro-operations in the same order per lane as msm_bucket_reduce<RB, FormTE>: the same limbs, the
// same workgroup results.
// Where it is used (measured, CHANGELOG r05): in the JOINT stage of a round's deferred jobs (up to 10^6 points each), one
// workgroup per CU — 0.1 - 0.3 ms per mid-size proof; NOT for the thin stages that run beside the accumulations of large jobs
// (two workgroups per CU there: each step 2.2x slower, +0.8 ms at 2^20).  SWM_MSM_LOW = 2 (default) / 1 (everywhere) / 0 (never).
#ifndef SWM_LOW_WAVES
#define SWM_LOW_WAVES 2  // waves per SIMD the register budget allows (3: 168 VGPRs, spills 37 of them)
#endif
template <int RB>
__global__ void __launch_bounds__(RB, SWM_LOW_WAVES) msm_bucket_reduce_low(TailBatch batch) {
    SWM_TAIL_KERNEL();
    const TailJob& job = batch.j[blockIdx.z];
    if (blockIdx.x >= job.red_blocks || blockIdx.y >= job.L.nwin) return;
    const G1XYZZ* __restrict__ partial = job.partial;
    const uint32_t* __restrict__ seg_off = job.seg_off;
    const WinLayout& L = job.L;
    const unsigned log_m = job.log_m;
    G1XYZZ* __restrict__ out = job.out;
    extern __shared__ __align__(16) unsigned char smem_raw[];
    RedSlot* sm_run = reinterpret_cast<RedSlot*>(smem_raw);  // RB + 1 padded slots (RedSlot: no bank conflicts)
    RedSlot* sm_tree = sm_run + 1;
    const uint32_t w = blockIdx.y, t = threadIdx.x;
    if (!((blockIdx.x >= job.blk_lo && blockIdx.x < job.blk_hi) || blockIdx.x < job.blk_low)) {
        if (t == 0) {
            te28_store_identity(sm_run[0].p);
            size_t o = ((size_t)w * job.red_blocks + blockIdx.x) * 2;
            te28_store_384(out[o], sm_run[0].p);
            te28_store_384(out[o + 1], sm_run[0].p);
            if (blockIdx.x == 0 && w == 0 && job.host_flags) {
                job.host_flags[0] = job.status[0];
                job.host_flags[1] = *job.entries;
                job.host_flags[2] = job.status[1];
            }
        }
        return;
    }
    G1XYZZ* __restrict__ my_acc = job.acc + ((size_t)w * job.red_blocks + blockIdx.x) * RB + t;
    const uint32_t B = 1u << (L.c[w] - 1), m = 1u << log_m;
    const uint32_t lo = (blockIdx.x * RB + t) << log_m;
    const uint32_t base = L.boff[w];
    te28_store_identity(sm_run[t].p);
    if (t == 0) te28_store_identity(sm_run[RB].p);
    te28_store_identity(*my_acc);
    uint32_t b = min(lo + m, B), s = 0, e = 0;
    bool walking = lo < B;
    if (walking) {
        b--;
        s = seg_off[base + b];
        e = s + nseg_of(job.hist[base + b], job.seg);
        if (e - s > job.big_nseg) e = s + 1;  // already folded into the first partial
    }
    enum { WALK = 0, SCAN = 1, SHIFT = 2, FOLD = 3, TREE = 4 };
    int phase = WALK;
    uint32_t d = 1;
    __syncthreads();
#pragma unroll 1
    for (;;) {
        G1XYZZ* dst = &sm_run[t].p;
        const G1XYZZ* pa = &sm_run[t].p;
        const G1XYZZ* pq = &sm_run[t].p;
        bool act = false;
        if (phase == WALK) {
            // (r06) the walk has no meeting point but its end: a lane works on slots of its own, so every WAVE walks at its own
            // pace — its LDS traffic no longer in step with the other three waves' — and meets the workgroup once, when none of
            // its lanes has a bucket left (until r05: a workgroup-wide vote, three barriers, in front of every step)
            if (__ballot(walking) == 0) {
                __syncthreads();
                phase = SCAN;
                continue;
            }
            if (walking) {
                act = true;
                if (s < e) {
                    pq = &partial[s++];
                } else {  // acc += run
                    dst = my_acc;
                    pa = my_acc;
                    if (b == lo) {
                        walking = false;
                    } else {
                        b--;
                        s = seg_off[base + b];
                        e = s + nseg_of(job.hist[base + b], job.seg);
                        if (e - s > job.big_nseg) e = s + 1;
                    }
                }
            }
        } else if (phase == SCAN) {  // inclusive suffix scan of run over the workgroup (Hillis-Steele), IN PLACE
            if (d >= RB) {
                phase = SHIFT;
                continue;
            }
            act = t + d < RB;
            if (act) pq = &sm_run[t + d].p;
            d <<= 1;
        } else if (phase == SHIFT) {  // slot_t <- m Suffix_t for t >= 1, in place; slot 0 keeps R_blk = Suffix_0 (m Suffix_0 is never needed)
#pragma unroll 1
            for (unsigned i = 0; i < log_m; i++) te28_slot_add_sync(&sm_run[t].p, &sm_run[t].p, &sm_run[t].p, t >= 1, false);
            __syncthreads();  // the fold reads the NEIGHBOUR's doubled slot
            phase = FOLD;
            continue;
        } else if (phase == FOLD) {
            // slot_{t+1} <- acc_t + m Suffix_{t+1}: lane t reads and rewrites slot t + 1 alone (slot RB holds the identity for the
            // last lane), the weighted sums move into LDS, slot 0 is left alone; summed over t this is A_blk
            dst = &sm_run[t + 1].p;
            pa = my_acc;
            pq = &sm_run[t + 1].p;
            act = true;
            phase = TREE;
            d = RB / 2;
        } else {  // tree over the slots 1 .. RB
            if (d == 0) break;
            if (d <= RB / 4) {  // narrow tree steps: four lanes per sum (te28_quad_add)
                const uint32_t c = threadIdx.x >> 2;
                if (c < d) te28_quad_add(&sm_tree[c].p, &sm_tree[c].p, &sm_tree[c + d].p, threadIdx.x & 3u);
                __syncthreads();
                d >>= 1;
                continue;
            }
            dst = &sm_tree[t].p;
            pa = &sm_tree[t].p;
            act = t < d;
            if (act) pq = &sm_tree[t + d].p;
            d >>= 1;
        }
        // barriers: the in-place scan needs its loads ahead of one and its stores behind it; a tree step and the fold are read by
        // other lanes in the NEXT step (trailing barrier); a walk step touches the lane's own slots only (the vote at the top
        // of the loop is the workgroup's only meeting point there)
        const bool walk = phase == WALK;
        te28_slot_add_sync(dst, pa, pq, act, phase == SCAN);
        if (!walk) __syncthreads();
    }
    if (threadIdx.x == 0) {
        size_t o = ((size_t)w * job.red_blocks + blockIdx.x) * 2;
        te28_store_384(out[o], sm_tree[0].p);
        te28_store_384(out[o + 1], sm_run[0].p);
        if (blockIdx.x == 0 && w == 0 && job.host_flags) {
            job.host_flags[0] = job.status[0];
            job.host_flags[1] = *job.entries;
            job.host_flags[2] = job.status[1];
        }
    }
}

// ---------------------------------------------------------------------------------------------- host driver
// Kernels that want more than 64 KB of dynamic LDS need the attribute raised once per device; `slot` names the kernel
// (0 hist, 1 scatter, 2 bucket_reduce, 3 partition, 4 bin_sort) in a small process-wide cache so that the runtime call is not repeated per MSM.
static int allow_big_lds(swm_ctx* ctx, int slot, const void* fn, size_t bytes) {
    static std::atomic<size_t> granted[64][12];
    if (bytes <= 64 * 1024) return SWM_OK;
    const int dev = ctx->device & 63;
    if (granted[dev][slot].load(std::memory_order_acquire) >= bytes) return SWM_OK;
    SWM_HIP(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    size_t cur = granted[dev][slot].load(std::memory_order_relaxed);
    while (cur < bytes && !granted[dev][slot].compare_exchange_weak(cur, bytes)) {}
    return SWM_OK;
}

// bases28[i] = (2^8 x, 2^8 y): the same points with coordinates in Montgomery radix 2^392 (infinity stays (0, 0))
__global__ void __launch_bounds__(256) msm_scale_bases(const G1Affine* __restrict__ in, size_t n, G1Affine* __restrict__ out,
                                                       uint32_t* __restrict__ inf_mask) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (inf_mask) {
        uint32_t z = 0;
        for (int j = 0; j < 12; j++) z |= in[i].x.v[j] | in[i].y.v[j];
        if (z == 0) atomicOr(&inf_mask[i >> 5], 1u << (i & 31));
    }
    const uint32_t k[12] = SWM_FQ_SCALE256_MONT;
    Fq c;
#pragma unroll
    for (int j = 0; j < 12; j++) c.v[j] = k[j];
    G1Affine p = in[i];
    p.x = fp_mul(p.x, c);
    p.y = fp_mul(p.y, c);
    out[i] = p;
}
int msm_scale_bases_run(swm_ctx* ctx, const G1Affine* d_in, size_t n, G1Affine* d_out, uint32_t* d_inf_mask) {
    if (n == 0) return SWM_OK;
    SWM_LAUNCH(ctx, "msm_scale_bases", msm_scale_bases, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, d_in, n, d_out,
               d_inf_mask);
    return SWM_OK;
}

int msm_table_build(swm_ctx* ctx, const G1Affine* d_points, size_t n, unsigned c, G1Affine** out) {
    *out = nullptr;
    if (n == 0 || c < 2) return set_err(ctx, SWM_ERR_INVALID_ARG, "msm table: bad arguments");
    const WinLayout L = msm_table_layout(c);
    const unsigned W = L.nwin;
    G1Affine* tab = nullptr;
    hipError_t e = hipMalloc((void**)&tab, (size_t)W * n * sizeof(G1Affine));
    if (e != hipSuccess) (void)hipGetLastError();  // not sticky: the caller falls back to a smaller form
    if (e != hipSuccess) return set_err(ctx, SWM_ERR_OOM, "msm table (%u windows x %zu points): %s", W, n, hipGetErrorString(e));
    G1XYZZ* x = nullptr;
    Fq* pref = nullptr;
    G1Affine* cur = nullptr;  // 2^(w c) P in the plain form, input of the next shift
    int rc = SWM_OK;
    do {
        if ((rc = scratch(ctx, "tab.xyzz", n * sizeof(G1XYZZ), (void**)&x)) != SWM_OK) break;
        if ((rc = scratch(ctx, "tab.pref", n * sizeof(Fq), (void**)&pref)) != SWM_OK) break;
        if ((rc = scratch(ctx, "tab.cur", n * sizeof(G1Affine), (void**)&cur)) != SWM_OK) break;
        if ((rc = msm_scale_bases_run(ctx, d_points, n, tab)) != SWM_OK) break;
        const G1Affine* src = d_points;
        const unsigned grid = (unsigned)((n + 255) / 256), gridn = (unsigned)(((n + TAB_NORM_CHUNK - 1) / TAB_NORM_CHUNK + 255) / 256);
        for (unsigned w = 1; w < W && rc == SWM_OK; w++) {
            hipLaunchKernelGGL(msm_table_shift, dim3(grid), dim3(256), 0, ctx->stream, src, n, (unsigned)L.c[w - 1], x);
            hipLaunchKernelGGL(msm_table_normalize, dim3(gridn), dim3(256), 0, ctx->stream, (const G1XYZZ*)x, n, pref, cur);
            if (hipGetLastError() != hipSuccess) rc = set_err(ctx, SWM_ERR_HIP, "msm table: launch failed");
            if (rc == SWM_OK) rc = msm_scale_bases_run(ctx, cur, n, tab + (size_t)w * n);
            src = cur;
        }
        if (rc == SWM_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = set_err(ctx, SWM_ERR_HIP, "msm table: sync failed");
    } while (0);
    scratch_release(ctx, "tab.xyzz");
    scratch_release(ctx, "tab.pref");
    scratch_release(ctx, "tab.cur");
    if (rc != SWM_OK) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(tab);
        return rc;
    }
    *out = tab;
    return SWM_OK;
}

bool msm_te_enabled() {
    static const bool on = env_switch("SWM_MSM_TE", 1, 0, 1) != 0;  // 0: XYZZ tables everywhere (the form of sets outside the subgroup)
    return on;
}
bool msm_table_fits(size_t bytes) {
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return true;  // cannot tell: let the allocation decide
    return bytes <= free_b - free_b / 4;
}
int msm_subgroup_check(swm_ctx* ctx, const G1Affine* d_points, size_t n, bool* ok) {
    *ok = true;
    if (n == 0) return SWM_OK;
    uint32_t* d_bad = nullptr;
    SWM_TRY(scratch(ctx, "tab.bad", 64, (void**)&d_bad));
    SWM_HIP(ctx, hipMemsetAsync(d_bad, 0, 4, ctx->stream));
    SWM_LAUNCH(ctx, "msm_subgroup_check", msm_subgroup_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, d_points, n, d_bad);
    uint32_t h = 0;
    SWM_HIP(ctx, hipMemcpyAsync(&h, d_bad, 4, hipMemcpyDeviceToHost, ctx->stream));
    SWM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *ok = h == 0;
    return SWM_OK;
}
int msm_table_build_te(swm_ctx* ctx, const G1Affine* d_points, size_t n, unsigned c, G1TE** out) {
    *out = nullptr;
    if (n == 0 || c < 2) return set_err(ctx, SWM_ERR_INVALID_ARG, "msm table: bad arguments");
    const WinLayout L = msm_table_layout(c);
    const unsigned W = L.nwin;
    G1TE* tab = nullptr;
    hipError_t e = hipMalloc((void**)&tab, (size_t)W * n * sizeof(G1TE));
    if (e != hipSuccess) (void)hipGetLastError();  // not sticky: the caller falls back to a smaller form
    if (e != hipSuccess) return set_err(ctx, SWM_ERR_OOM, "msm table (%u windows x %zu points): %s", W, n, hipGetErrorString(e));
    G1XYZZ* x = nullptr;
    Fq* pref = nullptr;
    G1Affine* cur = nullptr;  // 2^(bit_w) P in the plain form, input of the next shift
    uint32_t* d_bad = nullptr;
    uint32_t bad = 0;
    int rc = SWM_OK;
    do {
        if ((rc = scratch(ctx, "tab.xyzz", n * sizeof(G1XYZZ), (void**)&x)) != SWM_OK) break;
        if ((rc = scratch(ctx, "tab.pref", n * sizeof(Fq), (void**)&pref)) != SWM_OK) break;
        if ((rc = scratch(ctx, "tab.cur", n * sizeof(G1Affine), (void**)&cur)) != SWM_OK) break;
        if ((rc = scratch(ctx, "tab.bad", 64, (void**)&d_bad)) != SWM_OK) break;
        if (hipMemsetAsync(d_bad, 0, 4, ctx->stream) != hipSuccess) {
            rc = set_err(ctx, SWM_ERR_HIP, "msm table: memset failed");
            break;
        }
        const G1Affine* src = d_points;
        const unsigned grid = (unsigned)((n + 255) / 256), gridn = (unsigned)(((n + TAB_NORM_CHUNK - 1) / TAB_NORM_CHUNK + 255) / 256);
        for (unsigned w = 0; w < W && rc == SWM_OK; w++) {
            if (w > 0) {
                hipLaunchKernelGGL(msm_table_shift, dim3(grid), dim3(256), 0, ctx->stream, src, n, (unsigned)L.c[w - 1], x);
                hipLaunchKernelGGL(msm_table_normalize, dim3(gridn), dim3(256), 0, ctx->stream, (const G1XYZZ*)x, n, pref, cur);
                src = cur;
            }
            hipLaunchKernelGGL(msm_te_convert, dim3(gridn), dim3(256), 0, ctx->stream, src, n, pref, tab + (size_t)w * n, d_bad);
            if (hipGetLastError() != hipSuccess) rc = set_err(ctx, SWM_ERR_HIP, "msm table: launch failed");
        }
        if (rc == SWM_OK && (hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                             hipStreamSynchronize(ctx->stream) != hipSuccess))
            rc = set_err(ctx, SWM_ERR_HIP, "msm table: sync failed");
    } while (0);
    scratch_release(ctx, "tab.xyzz");
    scratch_release(ctx, "tab.pref");
    scratch_release(ctx, "tab.cur");
    if (rc != SWM_OK || bad) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(tab);
        return rc;  // bad: SWM_OK with *out == nullptr — a point without an image; the caller keeps the XYZZ form
    }
    *out = tab;
    return SWM_OK;
}

// Everything a resident base set needs for its MSMs, in one place (swm_srs_upload, the prover's committer keys):
//   *c    table width (0: the set is too small, its table would not fit, or n * windows >= 2^31 — per-window schedule only)
//   *te   twisted Edwards table when the set qualifies (subgroup established, SWM_MSM_TE != 0, it fits): the flat schedule
//         then runs in that form and *d28 is just the scaled copy of the set (n points) for the per-window schedule
//   *d28  otherwise the XYZZ table (row 0 = the scaled copy) when *c != 0, else the scaled copy
// A table that cannot be allocated is not an error: the next cheaper form is used (TE -> XYZZ table -> no table).
int msm_install_bases(swm_ctx* ctx, const G1Affine* d_points, size_t n, bool in_subgroup, G1Affine** d28, G1TE** te, unsigned* c,
                      uint32_t* d_inf_mask) {
    *d28 = nullptr;
    *te = nullptr;
    *c = msm_table_width(n);
    // A context that is one rank of a sharded proof (swm_set_msm_sharding before the key is built) accumulates n / G points per
    // MSM into the same bucket set, and the bucket stage — which every rank runs over all 2^(c-1) buckets — becomes the largest
    // replicated item: the table is as many bits narrower as the width rule gives for the rank's share of the set, at most
    // log2(G) - 1.  Measured per rank (tools/ubench/shard_emulate.py, same box): 2^20 constraints, G = 8: c = 20 / 18 / 17 ->
    // 23.7 / 20.9 / 20.9 ms; G = 4: flat; 2^22 constraints (1.5 M points per rank and more): 20 stays best (18: + 3 %).
    if (*c > 12 && ctx->shard_world >= 4 && !env_switch("SWM_MSM_TABLE_C", 0, 8, 22)) {
        unsigned lg = 0;
        while ((2u << lg) <= ctx->shard_world) lg++;
        const unsigned share = msm_table_width(std::max<size_t>(n / ctx->shard_world, 512));
        const unsigned delta = share && share < *c ? std::min(*c - share, lg - 1) : 0;
        *c = std::max(12u, *c - delta);
    }
    if (env_flag("SWM_TRACE")) fprintf(stderr, "[swm] base set of %zu points: table width %u (world %u)\n", n, *c, ctx->shard_world);
    if (*c && (uint64_t)n * msm_table_windows(*c) >= (1ull << 31)) *c = 0;  // the sort addresses table rows with 31 bits
    const unsigned W = *c ? msm_table_windows(*c) : 0;
    if (*c && msm_te_enabled()) {
        bool ok = in_subgroup;
        if (!ok) SWM_TRY(msm_subgroup_check(ctx, d_points, n, &ok));
        if (ok && msm_table_fits((size_t)W * n * sizeof(G1TE) + n * sizeof(G1Affine))) {
            int rc = msm_table_build_te(ctx, d_points, n, *c, te);
            if (rc != SWM_OK && rc != SWM_ERR_OOM) return rc;
        }
    }
    if (*c && !*te) {
        int rc = SWM_ERR_OOM;
        if (msm_table_fits((size_t)W * n * sizeof(G1Affine))) rc = msm_table_build(ctx, d_points, n, *c, d28);
        if (rc == SWM_ERR_OOM) *c = 0;  // no room for a table: the per-window schedule needs 1x the set
        else if (rc != SWM_OK) return rc;
    }
    if (!*d28) {
        hipError_t e = hipMalloc((void**)d28, std::max<size_t>(n, 1) * sizeof(G1Affine));
        if (e != hipSuccess) {
            if (*te) (void)hipFree(*te);
            *te = nullptr;
            return set_err(ctx, SWM_ERR_OOM, "msm bases (%zu points): %s", n, hipGetErrorString(e));
        }
    }
    // the scaled copy (row 0 of an XYZZ table is written again: same bytes) and, when asked for, the infinity mask
    int rc = msm_scale_bases_run(ctx, d_points, n, *d28, d_inf_mask);
    if (rc == SWM_OK && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = set_err(ctx, SWM_ERR_HIP, "msm bases: sync failed");
    if (rc != SWM_OK) {
        (void)hipFree(*d28);
        if (*te) (void)hipFree(*te);
        *d28 = nullptr;
        *te = nullptr;
    }
    return rc;
}

// ---------------------------------------------------------------------------------------------- hardware-queue probe
// Do two streams execute concurrently, i.e. sit on different hardware queues?  One single-lane kernel per stream that
// spins ~60 us on the constant-rate clock and records when it started and ended: on one queue the second starts after
// the first has ended.  ~0.15 ms per probe, used once per context while the stage streams are set up.
__global__ void msm_probe_spin(unsigned long long* out, long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    out[0] = (unsigned long long)t0;
    out[1] = (unsigned long long)wall_clock64();
}
static int streams_concurrent(swm_ctx* ctx, hipStream_t x, hipStream_t y, bool* concurrent) {
    unsigned long long* d = nullptr;
    SWM_TRY(scratch(ctx, "msm.probe", 64, (void**)&d));
    hipLaunchKernelGGL(msm_probe_spin, dim3(1), dim3(1), 0, x, d, (long long)6000);  // 100 MHz clock: 60 us
    hipLaunchKernelGGL(msm_probe_spin, dim3(1), dim3(1), 0, y, d + 2, (long long)6000);
    SWM_HIP(ctx, hipGetLastError());
    SWM_HIP(ctx, hipStreamSynchronize(x));
    SWM_HIP(ctx, hipStreamSynchronize(y));
    unsigned long long h[4];
    SWM_HIP(ctx, hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    *concurrent = h[2] < h[1];  // y started before x ended
    return SWM_OK;
}


// ---- asynchronous form -------------------------------------------------------------------------------------
// msm_enqueue launches every kernel of one MSM plus the download of its window sums WITHOUT host synchronisation;
// msm_finish waits for that download and does the host Horner fold.  `lane` selects the stream + device scratch set:
//   lane < 0 : the context's own stream (what the K1 ABI entry points use);
//   lane >= 0: one of MSM_LANES auxiliary streams; it first waits for everything enqueued on the context's stream so far.
// The prover alternates lanes, so that the latency-bound tail of one MSM (bucket fold, window reduction, download)
// and the sort of the next overlap with the VALU-bound accumulation.  Scratch is per lane (stream-ordered reuse);
// results land in per-job pinned host slots.
// Stream of one of the roles main / sort / accumulation / tail: non-blocking, default priority (stream priorities per role were
// measured in r04 / r05 — every assignment slower, CHANGELOG.md — and are gone; what orders the kernels on the chip is s_setprio).
hipError_t msm_create_stream(hipStream_t* out) { return hipStreamCreateWithFlags(out, hipStreamNonBlocking); }
bool msm_flat_applies(const MsmTable& tab, size_t n) {
    static const bool no_table = env_flag("SWM_MSM_NO_TABLE");  // the per-window schedule for every set (what table-less sets run)
    return tab.any() && !no_table && n >= ((size_t)1 << (tab.c > 8 ? tab.c - 8 : 0)) &&
           (uint64_t)tab.stride * msm_table_windows(tab.c) < (1ull << 31);
}
int msm_enqueue(swm_ctx* ctx, int lane, const G1Affine* d_bases, const G1Affine* d_bases28, const void* d_scalars, size_t n,
                int mont, MsmJob* job, MsmInfMask inf, bool defer_tail, MsmTable tab, MsmTwin twin) {
    job->active = false;
    job->tail_pending = false;
    job->n = n;
    job->twin = MsmTwinSrc();
    job->twin_copied = nullptr;
    job->twin_kept_lane = false;
    if (n == 0) return SWM_OK;
    if (n >= (1ull << 31)) return set_err(ctx, SWM_ERR_INVALID_ARG, "msm: n must be < 2^31");
    ctx->stat_msm_calls++;
    ctx->stat_msm_points += n;
    ctx->log_call('m', n);
    struct SinceWait {  // (counted when this call returns: the choices below see the jobs BEFORE this one)
        swm_ctx* c;
        ~SinceWait() { c->msm_since_wait++; }
    } since_wait{ctx};
    // the job before this one waited to learn whether anything follows it: something does, so its bucket stage takes the
    // thin shape that runs BESIDE this job's sort and accumulation (msm_tail_shape)
    SWM_TRY(msm_launch_lazy_tail(ctx, false));
    // flat schedule: the base set comes with its precomputed window multiples and the MSM is large enough to populate the
    // shared bucket set (below ~2^(c-4) points the per-window schedule with its small windows wins)
    const bool flat = msm_flat_applies(tab, n);
    const bool te = flat && tab.te != nullptr;  // twisted Edwards rows: accumulation and bucket stage run in that form
    job->te = te;
    // Twin jobs (msm.h): LEAD writes the follower's sorted array beside its own, FOLLOW copies the lead's descriptors.  Any
    // mismatch of the shapes leaves both on the ordinary schedule.  SWM_MSM_TWIN=0: never.
    static const bool twin_on = env_switch("SWM_MSM_TWIN", 1, 0, 1) != 0;
    // (from 2^15 points: below, a proof is a chain of launches and the follower's wait for the lead's sort costs more than its own
    // sort beside it — 2^14 constraints 3.88 -> 4.0 ms, 2^16 6.58 -> 6.49 ms, r06)
    static constexpr size_t twin_min = 32768;
    const bool twin_shape = twin_on && n >= twin_min && lane >= 0 && flat && te && tab.contiguous() && tab.scalar_stride == 1 && tab.shard_world <= 1 && !inf.mask;
    const bool lead = twin_shape && twin.role == MsmTwin::LEAD && twin.tab2.te && twin.tab2.c == tab.c && twin.tab2.contiguous() &&
                      twin.tab2.scalar_stride == 1 && twin.tab2.shard_world <= 1 && msm_flat_applies(twin.tab2, n) &&
                      twin.tab2.offset + n <= twin.tab2.stride && tab.offset + n <= tab.stride;
    MsmJob* const tsrc = twin_shape && twin.role == MsmTwin::FOLLOW ? twin.lead : nullptr;
    bool follow = tsrc && tsrc != job && tsrc->active && tsrc->twin.ready && tsrc->twin.scalars == d_scalars && tsrc->n == n &&
                  tsrc->twin.mont == mont && tsrc->twin.c == tab.c && tsrc->twin.te2 == tab.te && tsrc->twin.stride2 == tab.stride &&
                  tsrc->twin.offset2 == tab.offset;
    if (!tab.contiguous() && !flat)
        return set_err(ctx, SWM_ERR_INTERNAL, "msm: a strided base layout needs the precomputed-window schedule");
    // low-latency schedule (flat MSMs below ~2^18 points, where a proof is a chain of dependent additions rather than
    // a throughput problem): short segments (8 points), every bucket with more than two segments folded by a lane
    // group in msm_big_bucket_sum (a tree instead of the serial walk of the bucket stage), one bucket per lane in the
    // bucket stage
    static constexpr size_t lat_below = 262144;
    const bool lat = flat && n < lat_below;
    WinLayout pl = flat ? msm_table_layout(tab.c) : msm_plan(n);
    // rl: the layout the bucket stage and the host fold see — one window of 2^(c-1) buckets in the flat schedule
    WinLayout rl = pl;
    if (flat) {
        rl.nwin = 1;
        rl.boff[1] = rl.NB;
    }
    const size_t total = n * (size_t)pl.nwin;
    ctx->stat_msm_digits += total;
    if (total >= (1ull << 32)) return set_err(ctx, SWM_ERR_INVALID_ARG, "msm: n * windows must be < 2^32");
    // buckets per lane in the reduction: at most 16 workgroups per window (the host folds one (A, R) pair per workgroup)
    // small windows: one bucket per lane shortens the serial walk of a lone MSM (r01 sweep); inside a round's batch the
    // workgroups of all jobs have to be resident together (one per CU: 96 KB of LDS each), which four buckets per lane allow
    unsigned log_m = pl.maxB <= 2048 ? (defer_tail ? 2 : 0) : 2;
    if (lat) log_m = 0;
    // (A, R) pairs per window that fit a result slot and that the host folds: 16 per window, or 256 for the single
    // window of the flat schedule (one workgroup per CU either way)
    // workgroup width of the bucket stage: 256 lanes (single-wave workgroups of 64 lanes were measured in r02: the stage 1.00
    // instead of 0.72 ms, a 2^20 proof 82.7 instead of 77.4 ms).
    // Low-latency schedule on twisted Edwards rows: every chain of the bucket stage is worked by a quad of lanes (FormTEQuad;
    // SWM_MSM_QUAD=0: one lane per chain as for large jobs): 128 chains per workgroup (512 threads), at most 64 workgroups per
    // job, bucket sets up to 2^15 (r04 sweeps, CHANGELOG.md).
    static const bool quad_on = env_switch("SWM_MSM_QUAD", 1, 0, 1) != 0;
    static constexpr unsigned quad_rb = 128u, quad_blocks = 64u, quad_maxb = 32768u;
    const bool quad = lat && te && quad_on && pl.maxB <= quad_maxb;
    const unsigned rb = quad ? quad_rb : 256u;
    job->quad = quad;
    // One-lane twisted Edwards stages may take the low-LDS kernel (msm_bucket_reduce_low: 49 KB per workgroup);
    // SWM_MSM_LOW=0: always the 144-KB kernel of r02 - r04.  SWM_MSM_LOW_BLOCKS: workgroups a stage may be cut into (result slot:
    // up to 1024 (A, R) pairs; the host folds them in groups of 16).
    // r05, last collection: in the JOINT stage of deferred jobs (shaped by msm_joint_shape) it is worth 0.1 - 0.3 ms per mid-size
    // proof (2^16 7.2 -> 7.07, 2^18 17.2 -> 16.9, Merkle circuit 15.0 -> 14.7 ms); for the thin stages of large jobs it costs
    // 0.8 ms at 2^20 (two workgroups per CU).  SWM_MSM_LOW: 2 (default) = joint stages only, 1 = every one-lane stage, 0 = never.
    static const long low_mode = env_switch("SWM_MSM_LOW", 2, 0, 2);
    const bool low_on = low_mode == 1 || (low_mode == 2 && defer_tail);
    static constexpr unsigned low_blocks = 256u;
    // (joint stages of small jobs: SWM_MSM_JOINT_BLOCKS workgroups per job — with the low-LDS kernel twelve waves per CU are resident)
    // 64 per job: the four stages of a round's launch are resident together, one workgroup per CU — in the low-latency schedule
    // (r02) and for the larger jobs that join a round's launch since r05 (up to 10^6 points: 2^19 buckets -> 32 per lane).  With the
    // low-LDS kernel 128 per job (two workgroups per CU) measured SLOWER per launch: 1.43 against 1.06 ms for four 2^19-bucket jobs.
    static constexpr unsigned joint_blocks = 64u;
    const bool low = te && !quad && rb == 256 && low_on;
    job->low = low;
    // (low-latency schedule: the bucket stages of a round's four MSMs run in one launch and have to be resident together)
    const unsigned max_blocks = quad ? quad_blocks : (flat ? (defer_tail ? joint_blocks : (low ? low_blocks : 256u)) : 16u);
    job->max_blocks = low ? std::max(max_blocks, low_blocks) : std::max(max_blocks, 256u);
    while (((pl.maxB >> log_m) + rb - 1) / rb > max_blocks) log_m++;
    unsigned red_blocks = ((pl.maxB >> log_m) + rb - 1) / rb;
    if (red_blocks == 0) red_blocks = 1;
    // ---- stage set-up.  An asynchronous MSM (lane >= 0) is a three-stage pipeline over three auxiliary streams:
    //   S  digits + counting sort          (HBM / LDS-atomic bound)
    //   A  segment planning + accumulation (saturates the integer pipes)      waits for the job's sort
    //   T  bucket stage + download         (a latency-bound chain at one wave per SIMD)   waits for the job's accumulation
    // so that consecutive jobs keep stage A busy back to back while the tail of the previous job and the sort of the
    // next one run beside it.  (r02 timeline: with two lanes that each ran sort -> accumulate -> tail in order, both
    // lanes reached their tails together and ~10 ms of bucket stage per 2^20 proof ran with nothing beside it.)
    // `lane` only selects the scratch set of the per-lane buffers (two sets: a job's sort may overwrite a set once
    // the accumulation that last read it has finished).  lane < 0: everything on the context's stream (K1 ABI).
    hipStream_t main_stream = ctx->stream;
    hipStream_t st_sort = main_stream, st_acc = main_stream, st_tail = main_stream;
    // S | A0, A1 by lane | T (two accumulations may overlap: one alone leaves bubbles at the end of its length-sorted grid).
    // Small MSMs keep a single stream: the extra event hops cost more than they hide.
    // r05: a proof whose commitments fall on both sides of the bound ran them on streams that alias (a small job's lane stream is
    // a large job's accumulation or bucket-stage stream): at 2^16 constraints — 65 535-point and 196 608-point commitments — the
    // opening's second sort waited for the first accumulation.  The prover therefore sets ctx->msm_pipe_min per proof (every
    // commitment of a proof up to 2^19 constraints on one stream of its lane, four lanes: 2^16 7.05 -> 6.5 ms, 2^19 29.4 -> 28.6 ms;
    // from 2^20 on the three-stage pipeline is ahead).
    const size_t pipe_min = ctx->msm_pipe_min ? ctx->msm_pipe_min : (size_t)131072;
    const bool one_stream = n < pipe_min;
    if (lane >= 0) {
        // small MSMs: four single-stream lanes (scratch sets 0 .. 3 on the streams 1, 2, 3, 0), so that the launch chains of
        // all four commitments of a prover round proceed side by side; sets 0 and 1 are shared with the pipelined form,
        // whose accumulations run on the same streams 1 and 2 (stream order covers the reuse)
        static constexpr int small_lanes = 4;
        // scratch sets of the pipelined form: two — the sort of job k + 2 waits for the accumulation of job k (four sets, the
        // sorts running further ahead, measured nothing in r04; memory: ~110 B per digit and set)
        static constexpr int sets = 2;
        lane %= one_stream ? small_lanes : sets;
        if (!ctx->aux_stream[0]) {
            // Hardware-queue placement.  ROCm 7 hands its hardware queues (four by default) to streams in creation order,
            // bouncing: 1, 2, 3, 4, 4, 3, 2, 1, ... (rocprofv3 Queue_Id, r02).  With the context's stream and the null stream
            // first, four auxiliary streams created in a row put the sort and the tail stream on ONE queue: tail(k) and
            // sort(k+1) — the two phases that run between consecutive accumulations — executed one after the other
            // (2^20 proofs: 73.0 instead of 70.1 ms).  Wanted: the context's stream, the sort stream, the first accumulation
            // stream and the tail stream on four different queues (the second accumulation stream may share: accumulations
            // cannot overlap, each fills the register files).  A placeholder stream before the tail stream gives that in the
            // usual creation history; the result is PROBED (streams_concurrent) and a stream that shares a queue with an
            // earlier role is replaced by a fresh one, a few times at most — whatever streams the host application created
            // before.
            static constexpr bool steer = true;
            // The four streams are created and probed into locals and published to the context only when all of it
            // succeeded: a failure half way must not leave aux_stream[0] set with the later entries null (every later MSM
            // would skip this block and silently run its tail on the legacy default stream).
            hipStream_t aux[swm_ctx::MSM_LANES] = {nullptr, nullptr, nullptr, nullptr};
            auto setup = [&]() -> int {
                for (int i = 0; i < 3; i++) SWM_HIP(ctx, msm_create_stream(&aux[i]));
                if (steer) {
                    hipStream_t ph = nullptr;
                    SWM_HIP(ctx, hipStreamCreateWithFlags(&ph, hipStreamNonBlocking));
                    ctx->spare_streams.push_back(ph);
                }
                SWM_HIP(ctx, msm_create_stream(&aux[3]));
                if (steer) {
                    SWM_HIP(ctx, hipStreamSynchronize(main_stream));
                    const int roles[3] = {0, 1, 3};  // sort, accumulation 0, tail
                    std::vector<hipStream_t> fixed = {main_stream};
                    int budget = 8;  // replacement streams at most
                    for (int r : roles) {
                        for (;;) {
                            bool clash = false;
                            for (hipStream_t f : fixed) {
                                bool conc = true;
                                SWM_TRY(streams_concurrent(ctx, f, aux[r], &conc));
                                if (!conc) {
                                    clash = true;
                                    break;
                                }
                            }
                            if (!clash || budget-- <= 0) break;
                            ctx->spare_streams.push_back(aux[r]);  // kept alive: destroying it would free its slot
                            aux[r] = nullptr;
                            SWM_HIP(ctx, msm_create_stream(&aux[r]));
                        }
                        fixed.push_back(aux[r]);
                    }
                }
                return SWM_OK;
            };
            const int src = setup();
            if (src != SWM_OK) {
                for (hipStream_t a : aux)
                    if (a) ctx->spare_streams.push_back(a);  // destroyed with the context
                return src;
            }
            for (int i = 0; i < swm_ctx::MSM_LANES; i++) ctx->aux_stream[i] = aux[i];
        }
        if (one_stream) {
            // lanes 0, 1, 2 on three different hardware queues (accumulation 0, tail, sort); lane 3 shares the
            // second accumulation stream's queue
            static const int lane_stream[4] = {1, 3, 0, 2};
            st_sort = st_acc = st_tail = ctx->aux_stream[lane_stream[lane]];
        } else {
            st_sort = ctx->aux_stream[0];
            st_acc = ctx->aux_stream[1 + (lane & 1)];
            st_tail = ctx->aux_stream[3];  // (a second bucket-stage stream: 51.4 -> 57.4 ms at 2^20, r05)
        }
        if (!ctx->fork_event) SWM_HIP(ctx, hipEventCreateWithFlags(&ctx->fork_event, hipEventDisableTiming));
        SWM_HIP(ctx, hipEventRecord(ctx->fork_event, main_stream));
        SWM_HIP(ctx, hipStreamWaitEvent(st_sort, ctx->fork_event, 0));  // the scalars are ready
        // (the waits for the previous readers of this lane's scratch set: below, once it is known whether the job uses the set)
    }
    hipStream_t st = st_sort;
    const size_t slot_bytes = (size_t)MAX_WIN * 32 * sizeof(G1XYZZ) + 64;  // up to 1024 (A, R) pairs + status words
    const size_t flags_off = slot_bytes - 16;  // the last 16 bytes of a slot carry the status words of the job
    if ((size_t)rl.nwin * std::max(red_blocks, flat ? job->max_blocks : 0u) * 2 * sizeof(G1XYZZ) > flags_off) return set_err(ctx, SWM_ERR_INTERNAL, "msm: result slot too small");
    if (!ctx->pinned) {  // coherent + mapped: the bucket stage writes its results straight into the slots
        SWM_HIP(ctx, hipHostMalloc(&ctx->pinned, slot_bytes * swm_ctx::MSM_SLOTS, hipHostMallocCoherent | hipHostMallocMapped));
        SWM_HIP(ctx, hipHostGetDevicePointer(&ctx->pinned_dev, ctx->pinned, 0));
    }
    int slot = ctx->next_slot;
    if (ctx->slot_busy[slot])  // its previous job has not been collected: the download would overwrite live results
        return set_err(ctx, SWM_ERR_INTERNAL, "msm: more than %d jobs in flight", swm_ctx::MSM_SLOTS);
    ctx->next_slot = (ctx->next_slot + 1) % swm_ctx::MSM_SLOTS;
    if (!ctx->slot_event[slot]) SWM_HIP(ctx, hipEventCreateWithFlags(&ctx->slot_event[slot], hipEventDisableTiming));
    job->slot = slot;
    job->host = reinterpret_cast<G1XYZZ*>((char*)ctx->pinned + slot_bytes * slot);
    job->host_flags = reinterpret_cast<const uint32_t*>((char*)job->host + flags_off);
    job->host_dev = reinterpret_cast<G1XYZZ*>((char*)ctx->pinned_dev + slot_bytes * slot);
    job->host_flags_dev = reinterpret_cast<uint32_t*>((char*)job->host_dev + flags_off);
    // the bucket stage writes its results straight into the pinned slot (see msm_bucket_reduce; against stream-ordered copies,
    // r02 on one box: 2^16 proofs 11.4 vs 11.8 ms, 2^20 75.1 vs 75.7 ms)
    job->done = ctx->slot_event[slot];
    if (!ctx->acc_event[slot]) SWM_HIP(ctx, hipEventCreateWithFlags(&ctx->acc_event[slot], hipEventDisableTiming));
    job->acc_done = ctx->acc_event[slot];
    if (!ctx->sort_event[slot]) SWM_HIP(ctx, hipEventCreateWithFlags(&ctx->sort_event[slot], hipEventDisableTiming));
    job->stream = st_tail;
    job->pl = rl;
    job->red_blocks = red_blocks;
    job->blk_lo = 0;
    job->blk_hi = red_blocks;
    job->blk_low = 0;
    if (tab.scalar_stride != 1 && !flat) return set_err(ctx, SWM_ERR_INTERNAL, "msm: strided scalars need the table schedule");
    DigitShard dshard{};
    if (tab.shard_world > 1) {
        if (!flat || !tab.contiguous()) return set_err(ctx, SWM_ERR_INTERNAL, "msm: a bucket-range split needs the table schedule");
        const unsigned G = tab.shard_world, g = tab.shard_rank;
        const uint64_t span = (uint64_t)rb << log_m;  // buckets per bucket-stage workgroup
        job->blk_lo = (unsigned)((uint64_t)red_blocks * g / G);
        job->blk_hi = (unsigned)((uint64_t)red_blocks * (g + 1) / G);
        unsigned cfull = 0, cnarrow = 0;
        for (unsigned w = 0; w < pl.nwin; w++) cfull = std::max<unsigned>(cfull, pl.c[w]);
        for (unsigned w = 0; w < pl.nwin; w++)
            if (pl.c[w] < cfull) cnarrow = std::max<unsigned>(cnarrow, pl.c[w]);
        if (cnarrow) job->blk_low = (unsigned)((((uint64_t)1 << (cnarrow - 1)) + span - 1) / span);  // digits of a narrow window: buckets < 2^(c - 1)
        dshard.on = 1;
        dshard.cfull = cfull;
        dshard.blo = (uint32_t)std::min<uint64_t>(job->blk_lo * span, pl.NB);
        dshard.bhi = (uint32_t)std::min<uint64_t>(job->blk_hi * span, pl.NB);
        dshard.plo = (uint64_t)(((unsigned __int128)n * g) / G);
        dshard.phi = (uint64_t)(((unsigned __int128)n * (g + 1)) / G);
    }
    job->log_m = log_m;
    job->rb = rb;

    // everything below is enqueued on `st`: temporarily make it the context's stream so that launches, memsets,
    // scratch growth and the profiling events all refer to it
    struct StreamSwap {
        swm_ctx* c;
        hipStream_t old;
        ~StreamSwap() { c->stream = old; }
    } swap{ctx, main_stream};
    ctx->stream = st;
    char nm[10][32];
    const char* base[10] = {"hist", "segs", "bucket_off", "seg_off", "digits", "sorted", "scan_tot", "big_list", "points", "pairs"};
    // Per-slot buffers are sized by the LARGEST request any slot has seen, not by the job at hand (r05): the eight result slots
    // rotate through the jobs of successive proofs (15 jobs per Merkle-circuit proof: a slot meets a different commitment every
    // proof), and a slot that had only held 131 072-point jobs grew — all streams synchronised, hipFree, hipMalloc: 2.4 ms with
    // the GPU idle — when a 786 432-point job reached it, somewhere in each of the first dozen proofs of a key.
    auto slot_scratch = [&](int kind, size_t bytes) {
        if (bytes > ctx->msm_slot_bytes[kind]) ctx->msm_slot_bytes[kind] = bytes;
        return ctx->msm_slot_bytes[kind];
    };
    // what the deferred bucket stage of a job still reads (histogram block with the status words, bucket / segment offsets,
    // big-bucket list, partial sums) is kept per result SLOT; the rest is per lane (stream-ordered reuse)
    for (int i = 0; i < 10; i++) {
        const bool per_slot = i == 0 || i == 2 || i == 3 || i == 7 || i == 8;
        snprintf(nm[i], sizeof(nm[i]), per_slot ? "msmS%d.%s" : "msm%d.%s", per_slot ? slot : (lane < 0 ? 9 : lane), base[i]);
    }

    // Segment bound.  Long segments mean one partial sum per bucket (the bucket stage walks fewer partials) but fewer,
    // longer lanes in the accumulation; they pay once the buckets alone oversubscribe the chip (r01 sweep: 128 beats
    // 32 by 10 % at 2^20 and 2^22, loses at 2^16 where 45 k buckets cannot fill 196 k lane slots).
    uint32_t SEG = pl.NB >= 262144 ? SEG_MAX : 32;  // smaller bounds for small MSMs measured within run-to-run noise
    if (lat) SEG = 16;  // (8 below 2^16 points until r04: with four lanes per segment the chain per entry is 2 products, not 7)
    // (r05, prefix tables: a job of |H| points on a table one bit narrower than before has ~14 entries per bucket — with 16-point
    // segments a quarter of the buckets would split in two and the bucket stage walk 2.3 instead of 2 steps per bucket)
    if (lat && total / pl.NB >= 10) SEG = 32;
    // segments per bucket above which a bucket is folded ahead of the bucket stage, and the lanes that fold one
    uint32_t big_nseg = BIG_NSEG;
    unsigned log_g = 8;
    if (lat) {
        // every bucket with more than two segments below 2^16 points; from there (16-point segments, 1 - 2 per bucket) only
        // the rare long ones: listing ~2 % of the buckets cost a ~120-us launch per MSM for a two-step shorter walk
        // (r02: 2^16 proofs 11.5 -> 10.8 ms)
        big_nseg = n < 65536 ? 4 : 6;  // (2 with 8-point segments, until r04)
        const size_t per_bucket = total / ((size_t)pl.NB * SEG) + 1;  // expected segments per bucket
        log_g = 2;
        while (log_g < 6 && ((size_t)1 << log_g) < 2 * per_bucket) log_g++;
    }
    job->big_nseg = big_nseg;
    // every bucket adds at most one short segment; the flat schedule hands out segment indices per coarse bin with the bin's
    // capacity (msm_flat_scan_bins): at most one more per bucket slot of the (padded) bins
    const size_t nseg_max = total / SEG + pl.NB + 1 + (flat ? FLAT_MAX_FINE + FLAT_MAX_BINS : 0);
    uint32_t *hist, *cursor, *big_count, *len_hist, *bucket_off, *seg_off, *digits, *sorted, *big_list, *tot_cnt, *tot_seg;
    uint32_t *seg_start, *seg_len, *order;
    // two-level scatter for large MSMs (see msm_partition): bins of ~8 k entries, sized per window
    bool two_level = !flat && n >= 262144 && pl.maxB >= 8192;
    BinPlan bp;
    memset(&bp, 0, sizeof(bp));
    if (two_level) {
        uint32_t target = 64;  // bins per full window: next power of two >= n / 8192, within [64, PART_MAX_BINS]
        while (target < PART_MAX_BINS && (size_t)target * 8192 < n) target <<= 1;
        for (unsigned w = 0; w < pl.nwin; w++) {
            uint32_t B = 1u << (pl.c[w] - 1), beff = B;
            if (w + 1 == pl.nwin) {  // top window: digits only reach (r - 1) >> bit (+ 1 for the carry), never negative
                uint64_t top = 0;
                for (int i = 7; i >= 0; i--) {
                    int sh = 32 * i - (int)pl.bit[w];
                    if (sh >= 0 && sh < 64) top |= (uint64_t)FrParams::P[i] << sh;
                    else if (sh < 0 && sh > -32) top |= (uint64_t)FrParams::P[i] >> (-sh);
                }
                beff = (uint32_t)std::min<uint64_t>(B, top + 2);
            }
            unsigned lb = 0;
            while ((1u << lb) < beff) lb++;
            unsigned lt = 0;
            while ((1u << lt) < target) lt++;
            // a window with 2^lb possible buckets out of the nominal 2^(c-1) holds 2^(c-1-lb) times denser buckets:
            // give it proportionally more bins so that a bin still holds ~n / target entries
            unsigned fb = lb > lt ? lb - lt : 0;
            if (fb > 7) fb = 7;
            uint32_t nb = (beff + (1u << fb) - 1) >> fb;
            if (nb > PART_MAX_BINS) {
                two_level = false;
                break;
            }
            bp.fb[w] = (uint8_t)fb;
            bp.nbins[w] = (uint16_t)nb;
            bp.max_nbins = std::max(bp.max_nbins, nb);
        }
    }
    const uint32_t maxbins = two_level ? PART_MAX_BINS : 0;
    // flat schedule: coarse bins of the shared bucket set (<= 4096 bins, <= 512 buckets each, ~8 k entries per bin)
    unsigned flat_fb = 0;
    uint32_t flat_bins = 0;
    if (flat) {
        // as few bins as the LDS of msm_flat_bin_sort allows (FLAT_BIN_CAP entries): every (tile, bin) run of the partition
        // costs one global atomic, and with ~2 entries per run those atomics (27 M at 2^22 points) were the whole kernel
        static constexpr size_t bin_target = 28000;  // (14 000: 52.3 - 52.5 vs 51.4 ms at 2^20, r04)
        // (small MSMs: at least ~1024 bins as long as a bin keeps 1024 entries — 64 bins meant 64 workgroups in the bin
        // sort and 1024 lanes contending for 64 LDS counters in the coarse histogram)
        const size_t target = std::min(bin_target, std::max<size_t>(1024, total / 1024));
        uint32_t want = 64;
        while (want < FLAT_MAX_BINS && (size_t)want * target < total) want <<= 1;
        while ((pl.NB >> flat_fb) > want) flat_fb++;
        while ((1u << flat_fb) > FLAT_MAX_FINE) flat_fb--;  // keeps the fine-count arrays of msm_flat_bin_sort within LDS
        flat_bins = (pl.NB + (1u << flat_fb) - 1) >> flat_fb;
        if (flat_bins > FLAT_MAX_BINS) return set_err(ctx, SWM_ERR_INTERNAL, "msm: too many coarse bins");
    }
    size_t zero_words = 2 * (size_t)(pl.NB + 1) + 4 + 2 * (size_t)(SEG_MAX + 1) * LEN_STRIDE + MAX_WIN + (size_t)pl.nwin * maxbins +
                              (flat ? (2 + 3 * (size_t)FLAT_CUR_STRIDE) * FLAT_MAX_BINS + 4 + 32 : 0);
    zero_words = (zero_words + 63) & ~(size_t)63;  // whole 256-byte lines: the runtime then clears them with one kernel, not two
    // the follower's arrays must have the geometry of the lead's
    follow = follow && tsrc->twin.zero_words == zero_words && tsrc->twin.nseg_max == nseg_max && tsrc->twin.SEG == SEG &&
             tsrc->twin.big_nseg == big_nseg && tsrc->twin.flat_bins == flat_bins;
    if (follow) {
        // A follower touches nothing of its lane's scratch set — its sorted array and its segment descriptors live in the pair's own
        // buffers, free once the previous pair's follower has accumulated (the lead waited for that before its bin sort) — so the
        // NEXT job may take this lane's set: the caller gets the lane back (pipelined form: job->twin_kept_lane), and that job's sort
        // runs beside the lead's accumulation instead of waiting for it (two sets: the sort of job k + 2 waits for accumulation k).
        snprintf(nm[1], sizeof(nm[1]), "msmT.segs");
        job->twin_kept_lane = !one_stream;
    } else if (lane >= 0) {
        // the scratch set of this lane was last read by the accumulation of the job two back
        // (single-stream jobs too: their stream is not necessarily the one the set's previous reader ran on)
        if (ctx->set_acc_event[lane]) SWM_HIP(ctx, hipStreamWaitEvent(st_sort, ctx->set_acc_event[lane], 0));
        // ... and, when that job led a twin pair, copied by its follower
        if (ctx->lane_copy_event[lane]) SWM_HIP(ctx, hipStreamWaitEvent(st_sort, ctx->lane_copy_event[lane], 0));
    }
    SWM_TRY(scratch(ctx, nm[0], slot_scratch(0, zero_words * 4), (void**)&hist));
    cursor = hist + pl.NB + 1;
    big_count = cursor + pl.NB + 1;
    len_hist = big_count + 4;
    uint32_t* len_cursor = len_hist + (SEG_MAX + 1) * LEN_STRIDE;       // the run cursors of msm_seg_order, one per length class
    uint32_t* two_level_bad = len_cursor + (SEG_MAX + 1) * LEN_STRIDE;  // one flag per window, then the per-(window, bin) cursors
    uint32_t* bin_cursor = two_level_bad + MAX_WIN;
    // [windows x bins] counts (room for 32 windows) | [bins x 32] offsets | [bins x 32] cursors (one 128-byte line per bin each) |
    // [bins + 1] offsets | [bins + 1] first segment index
    // (the [bins x 32] blocks start on 128-byte lines: vector loads / stores of a bin's windows)
    uint32_t* flat_cnt = hist + ((((size_t)(bin_cursor - hist) + (size_t)pl.nwin * maxbins) + 31) & ~(size_t)31);
    uint32_t* flat_win_off = flat_cnt + (size_t)FLAT_CUR_STRIDE * FLAT_MAX_BINS;
    uint32_t* flat_cur = flat_win_off + (size_t)FLAT_CUR_STRIDE * FLAT_MAX_BINS;
    uint32_t* flat_off = flat_cur + (size_t)FLAT_CUR_STRIDE * FLAT_MAX_BINS;
    uint32_t* flat_seg_off = flat_off + FLAT_MAX_BINS + 2;  // [bins + 1] first segment index of every bin
    uint2* pairs = nullptr;
    if (two_level || flat) SWM_TRY(scratch(ctx, nm[9], total * sizeof(uint2), (void**)&pairs));
    SWM_TRY(scratch(ctx, nm[1], nseg_max * 12, (void**)&seg_start));
    seg_len = seg_start + nseg_max;
    order = seg_len + nseg_max;
    SWM_TRY(scratch(ctx, nm[2], slot_scratch(1, (pl.NB + 1) * 4ull), (void**)&bucket_off));
    SWM_TRY(scratch(ctx, nm[3], slot_scratch(2, (pl.NB + 1) * 4ull), (void**)&seg_off));
    SWM_TRY(scratch(ctx, nm[4], total * 4, (void**)&digits));
    SWM_TRY(scratch(ctx, nm[5], total * 4, (void**)&sorted));
    const unsigned scan_tiles = (pl.NB + SCAN_TILE - 1) / SCAN_TILE;
    SWM_TRY(scratch(ctx, nm[6], (size_t)(scan_tiles + 1) * 8, (void**)&tot_cnt));
    tot_seg = tot_cnt + scan_tiles + 1;
    SWM_TRY(scratch(ctx, nm[7], slot_scratch(3, (size_t)pl.NB * 4), (void**)&big_list));
    // XYZZ scratch: partial[nseg_max] | wpart[nwin * red_blocks * 2]
    G1XYZZ *partial, *wpart;
    const size_t wpart_n = (size_t)rl.nwin * std::max(red_blocks, flat ? job->max_blocks : 0u) * 2;
    SWM_TRY(scratch(ctx, nm[8], slot_scratch(4, (nseg_max + wpart_n) * sizeof(G1XYZZ)), (void**)&partial));
    wpart = partial + nseg_max;
    job->d_acc = nullptr;
    if (low) {  // the lanes' weighted sums of the low-LDS bucket stage: one point per lane, per result slot like `partial`
        char nma[32];
        snprintf(nma, sizeof(nma), "msmS%d.acc", slot);
        SWM_TRY(scratch(ctx, nma, slot_scratch(5, (size_t)rl.nwin * job->max_blocks * 256 * sizeof(G1XYZZ)), (void**)&job->d_acc));
    }

    SWM_HIP(ctx, zero_fill_async(hist, zero_words * 4, ctx->stream));  // (a kernel with issue priority, not the runtime's fill: fill.cuh)
    const Fr* sc = reinterpret_cast<const Fr*>(d_scalars);
    unsigned grid_n = (unsigned)std::min<size_t>((n + 255) / 256, 256 * 16);
    if (follow) {
        // the second sorted array the lead's bin sort wrote; the descriptors: counts, first segments, the list of oversized buckets,
        // status words, length histogram, the bins' entry and segment totals, the segments.  Not the cursors and the count of live
        // segments, which the lead's msm_seg_order may be advancing at this moment: this job's own stay zero for its own.
        SWM_TRY(scratch(ctx, "msmT.sorted", total * 4, (void**)&sorted));
        SWM_HIP(ctx, hipStreamWaitEvent(ctx->stream, tsrc->twin.sorted, 0));
        const uint32_t* a = tsrc->twin.block;
        TwinCopy tc;
        memset(&tc, 0, sizeof(tc));
        auto range = [&](const uint32_t* from, uint32_t* to, size_t words) {
            tc.src[tc.k] = from;
            tc.dst[tc.k] = to;
            tc.words[tc.k] = (uint32_t)words;
            tc.k++;
        };
        range(a, hist, pl.NB + 1);
        range(a + (big_count - hist), big_count, 3);
        range(a + (len_hist - hist), len_hist, (size_t)(SEG_MAX + 1) * LEN_STRIDE);
        range(a + (flat_off - hist), flat_off, 2 * (size_t)FLAT_MAX_BINS + 4);
        range(tsrc->twin.seg_off, seg_off, pl.NB + 1);
        range(tsrc->twin.big_list, big_list, pl.NB);
        range(tsrc->twin.seg_start, seg_start, 2 * nseg_max);  // seg_start | seg_len
        SWM_LAUNCH(ctx, "msm_twin_copy", msm_twin_copy, dim3(1024), dim3(256), 0, tc);
        if (!ctx->twin_copy_event[slot]) SWM_HIP(ctx, hipEventCreateWithFlags(&ctx->twin_copy_event[slot], hipEventDisableTiming));
        SWM_HIP(ctx, hipEventRecord(ctx->twin_copy_event[slot], ctx->stream));
        tsrc->twin_copied = ctx->twin_copy_event[slot];
        if (tsrc->twin.lane >= 0) ctx->lane_copy_event[tsrc->twin.lane] = ctx->twin_copy_event[slot];
        ctx->stat_msm_twins++;
    } else {
    SWM_LAUNCH(ctx, "msm_digits", msm_digits, dim3(grid_n), dim3(256), 0, sc, n, mont, pl, digits, inf.mask, inf.first,
               big_count + 1 /* zeroed with the histogram */, dshard, tab.scalar_stride != 1 ? tab.blk_log : 31u,
               tab.scalar_stride != 1 ? tab.bstride : (size_t)0);
    if (flat) {
        // two-level counting sort over the shared bucket set; the fine counts written by msm_flat_bin_sort are the
        // histogram the scans consume, and the bins are contiguous bucket ranges, so `sorted` is in bucket order
        uint32_t ctile = 65536;  // digits per workgroup of the coarse histogram: at least ~256 workgroups
        while (ctile > 4096 && (size_t)ctile * 256 > total) ctile >>= 1;
        if (pl.nwin > FLAT_CUR_STRIDE) return set_err(ctx, SWM_ERR_INTERNAL, "msm: too many windows for the flat sort");
        // digits per workgroup of the partition: 16 K (r04: half the (tile, bin) runs and reserving atomics of 8 K tiles, one
        // workgroup per CU instead of two — prove 2^20 51.6 -> 51.0 ms, and the transforms beside it run a quarter faster) as long
        // as the pairs and the three per-bin arrays fit the 160 KB of LDS (up to 2 048 bins); 8 K above
        const size_t lds_bins = (3 * (size_t)((flat_bins + 3) & ~3u) + 1024) * 4;
        const bool tile16 = 16384 * sizeof(uint2) + lds_bins <= 160 * 1024;
        const uint32_t part_tile = tile16 ? 16384u : 8192u;
        const size_t lds_part = (size_t)part_tile * sizeof(uint2) + lds_bins;
        SWM_TRY(allow_big_lds(ctx, tile16 ? 10 : 5, tile16 ? (const void*)msm_flat_partition<16> : (const void*)msm_flat_partition<8>, lds_part));
        const size_t lds_bin = ((size_t)FLAT_BIN_CAP + 2 * ((size_t)1 << flat_fb)) * 4;
        SWM_TRY(allow_big_lds(ctx, 6, (const void*)msm_flat_bin_sort<BIN_THREADS>, lds_bin));
        // (256-lane forms of the partition and the bin sort for sorts that run beside an accumulation: built and measured in r05 —
        // 51.15 vs 50.8 ms at 2^20 with the issue priorities in place — and removed in r06)
        SWM_LAUNCH(ctx, "msm_flat_hist", msm_flat_coarse_hist, dim3((unsigned)((n + ctile - 1) / ctile), pl.nwin), dim3(SORT_THREADS), 0,
                   digits, n, flat_fb, flat_bins, ctile, flat_cnt);
        unsigned scan_threads = 64;
        const bool one_bin_per_lane = flat_bins <= 1024;
        while (scan_threads * (one_bin_per_lane ? 1 : 4) < flat_bins) scan_threads <<= 1;
        if (one_bin_per_lane)
            SWM_LAUNCH(ctx, "msm_flat_hist", msm_flat_scan_bins<1>, dim3(1), dim3(scan_threads), 0, flat_cnt, flat_bins, pl.nwin, flat_off, flat_win_off,
                   flat_fb, SEG, flat_seg_off);
        else
            SWM_LAUNCH(ctx, "msm_flat_hist", msm_flat_scan_bins<4>, dim3(1), dim3(scan_threads), 0, flat_cnt, flat_bins, pl.nwin, flat_off, flat_win_off,
                   flat_fb, SEG, flat_seg_off);
        if (tile16)
            SWM_LAUNCH(ctx, "msm_flat_partition", msm_flat_partition<16>, dim3((unsigned)((n + part_tile - 1) / part_tile), pl.nwin),
                   dim3(1024), lds_part, digits, n, (uint32_t)tab.stride, (uint32_t)tab.offset, tab.blk_log, (uint32_t)tab.bstride, flat_fb,
                   flat_bins, flat_win_off, pl.nwin, flat_cur, pairs);
        else
            SWM_LAUNCH(ctx, "msm_flat_partition", msm_flat_partition<8>, dim3((unsigned)((n + part_tile - 1) / part_tile), pl.nwin),
                   dim3(1024), lds_part, digits, n, (uint32_t)tab.stride, (uint32_t)tab.offset, tab.blk_log, (uint32_t)tab.bstride, flat_fb,
                   flat_bins, flat_win_off, pl.nwin, flat_cur, pairs);
        // (bucket / segment offsets, segment descriptors, the length histogram and the list of oversized buckets come out of
        // the bin sort: no scans over the bucket histogram, no msm_seg_desc)
        const FlatSegOut fso{hist, bucket_off, seg_off, seg_start, seg_len, len_hist, big_count, big_list, SEG, big_nseg, te ? 1u : 0u};
        TwinOut two{nullptr, 1u, 0u, 0u};
        if (lead) {
            // (the array is read by the accumulation of the previous pair's follower)
            SWM_TRY(scratch(ctx, "msmT.sorted", total * 4, (void**)&two.sorted2));
            if (ctx->twin_sorted_event) SWM_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->twin_sorted_event, 0));
            two.tstride = (uint32_t)tab.stride;
            two.dstride = (uint32_t)twin.tab2.stride - (uint32_t)tab.stride;
            two.doff = (uint32_t)twin.tab2.offset - (uint32_t)tab.offset;
        }
        SWM_LAUNCH(ctx, "msm_flat_bin_sort", msm_flat_bin_sort<BIN_THREADS>, dim3(flat_bins), dim3(BIN_THREADS), lds_bin,
                   (const uint2*)pairs, flat_fb, pl.NB, flat_off, flat_seg_off, fso, sorted, two);
        if (lead) {
            MsmTwinSrc& t = job->twin;
            t.scalars = d_scalars;
            t.mont = mont;
            t.lane = lane;
            t.c = tab.c;
            t.te2 = twin.tab2.te;
            t.stride2 = twin.tab2.stride;
            t.offset2 = twin.tab2.offset;
            t.block = hist;
            t.seg_start = seg_start;
            t.seg_off = seg_off;
            t.big_list = big_list;
            t.zero_words = zero_words;
            t.nseg_max = nseg_max;
            t.SEG = SEG;
            t.big_nseg = big_nseg;
            t.flat_bins = flat_bins;
            t.sorted = ctx->sort_event[slot];
            SWM_HIP(ctx, hipEventRecord(t.sorted, ctx->stream));
            t.ready = true;
        }
    } else {
        // tile size: flat between 2^15 and 2^18 on MI355X (the scatter is bound by its 4-byte scattered writes: 73 G digits/s)
        const uint32_t SORT_TILE = SORT_TILE_MIN;
        unsigned tiles = (unsigned)((n + SORT_TILE - 1) / SORT_TILE);
        size_t lds_sort = (size_t)pl.maxB * 4;
        SWM_TRY(allow_big_lds(ctx, 0, (const void*)msm_hist, lds_sort));
        SWM_TRY(allow_big_lds(ctx, 1, (const void*)msm_scatter, lds_sort));
        SWM_LAUNCH(ctx, "msm_hist", msm_hist, dim3(tiles, pl.nwin), dim3(SORT_THREADS), lds_sort, digits, n, pl, SORT_TILE, hist);
        SWM_LAUNCH(ctx, "msm_scan", msm_scan_totals, dim3(scan_tiles), dim3(SCAN_BLOCK), 0, hist, pl.NB, SEG, tot_cnt, tot_seg);
        SWM_LAUNCH(ctx, "msm_scan", msm_scan_mid, dim3(1), dim3(SCAN_BLOCK), 0, tot_cnt, tot_seg, scan_tiles);
        SWM_LAUNCH(ctx, "msm_scan", msm_scan_final, dim3(scan_tiles), dim3(SCAN_BLOCK), 0, hist, pl.NB, SEG, tot_cnt, tot_seg,
                   scan_tiles, bucket_off, seg_off, big_count, big_list, big_nseg);
        if (two_level) {
            SWM_TRY(allow_big_lds(ctx, 3, (const void*)msm_partition, (size_t)PART_TILE * sizeof(uint2)));
            SWM_TRY(allow_big_lds(ctx, 4, (const void*)msm_bin_sort, (size_t)BIN_CAP * 4));
            SWM_LAUNCH(ctx, "msm_bin_check", msm_bin_check, dim3((bp.max_nbins + 255) / 256, pl.nwin), dim3(256), 0, bucket_off, pl, bp,
                       two_level_bad);
            SWM_LAUNCH(ctx, "msm_partition", msm_partition, dim3((unsigned)((n + PART_TILE - 1) / PART_TILE), pl.nwin), dim3(1024),
                       (size_t)PART_TILE * sizeof(uint2), digits, n, pl, bp, bucket_off, bin_cursor, two_level_bad, pairs);
            SWM_LAUNCH(ctx, "msm_bin_sort", msm_bin_sort, dim3(bp.max_nbins, pl.nwin), dim3(BIN_THREADS), (size_t)BIN_CAP * 4,
                       (const uint2*)pairs, pl, bp, bucket_off, two_level_bad, sorted);
        }
        SWM_LAUNCH(ctx, "msm_scatter", msm_scatter, dim3(tiles, pl.nwin), dim3(SORT_THREADS), lds_sort, digits, n, pl,
                   SORT_TILE, bucket_off, cursor, sorted, two_level ? (const uint32_t*)two_level_bad : (const uint32_t*)nullptr);
    }
    }  // (!follow)
    // ---- stage A
    if (st_acc != st_sort) {
        SWM_HIP(ctx, hipEventRecord(ctx->sort_event[slot], st_sort));
        SWM_HIP(ctx, hipStreamWaitEvent(st_acc, ctx->sort_event[slot], 0));
        ctx->stream = st_acc;
    }
    unsigned grid_s = (unsigned)((nseg_max + ORD_THREADS - 1) / ORD_THREADS);
    // segment indices in use (a device word): the flat schedule's bins hand them out by capacity, the per-window schedule densely
    const uint32_t* seg_space = flat ? flat_seg_off + flat_bins : seg_off + pl.NB;
    uint32_t* nseg_live = big_count + 3;  // segments that hold entries = lanes of work for the accumulation
    if (!flat)
        SWM_LAUNCH(ctx, "msm_seg_order", msm_seg_desc, dim3(grid_s), dim3(ORD_THREADS), 0, bucket_off, seg_off, pl.NB,
                   SEG, seg_start, seg_len, len_hist);
    SWM_LAUNCH(ctx, "msm_seg_order", msm_seg_order, dim3((unsigned)((nseg_max + ORD2_THREADS * ORD2_U - 1) / (ORD2_THREADS * ORD2_U))),
               dim3(ORD2_THREADS), 0, seg_len, seg_space, SEG, (const uint32_t*)len_hist, len_cursor, nseg_live, order);
    unsigned acc_grid = (unsigned)((nseg_max + 255) / 256);
    const dim3 big_grid(std::min<unsigned>((pl.NB + (RED_BLOCK >> log_g) - 1) / (RED_BLOCK >> log_g), lat ? 2048 : 512));
    if (te) {
        // small jobs (up to 2^20 digits: r04 sweep): four lanes per segment
        static constexpr size_t quad_acc = 1048576;
        if (lat && total <= quad_acc && quad_on)
            SWM_LAUNCH(ctx, "msm_accumulate", msm_accumulate_te_quad, dim3((unsigned)((nseg_max + 63) / 64)), dim3(256), 0, tab.te, sorted,
                       seg_start, seg_len, order, nseg_live, partial);
        else
            SWM_LAUNCH(ctx, "msm_accumulate", msm_accumulate_te, dim3(acc_grid), dim3(256), 0, tab.te, sorted, seg_start,
                       seg_len, order, nseg_live, partial);
        SWM_LAUNCH(ctx, "msm_big_bucket_sum", msm_big_bucket_sum<FormTE>, big_grid, dim3(RED_BLOCK), RED_BLOCK * sizeof(G1XYZZ),
                   partial, seg_off, hist, SEG, big_count, big_list, log_g);
    } else {
        SWM_LAUNCH(ctx, "msm_accumulate", msm_accumulate, dim3(acc_grid), dim3(256), 0,
                   flat ? (const G1Affine*)nullptr : d_bases, flat ? tab.t28 : d_bases28, sorted, seg_start, seg_len, order,
                   nseg_live, partial);
        SWM_LAUNCH(ctx, "msm_big_bucket_sum", msm_big_bucket_sum<FormXYZZ>, big_grid, dim3(RED_BLOCK), RED_BLOCK * sizeof(G1XYZZ),
                   partial, seg_off, hist, SEG, big_count, big_list, log_g);
    }
    job->needs_acc_wait = st_tail != ctx->stream || defer_tail;  // the tail runs on another stream (or later, with others)
    job->d_partial = partial;
    job->d_wpart = wpart;
    job->d_seg_off = seg_off;
    job->d_hist = hist;
    job->seg = SEG;
    job->d_status = big_count + 1;
    job->d_entries = flat ? flat_off + flat_bins : bucket_off + pl.NB;  // entries the sort placed
    job->active = true;
    job->tail_pending = true;
    ctx->slot_busy[slot] = true;
    if (lane >= 0 || defer_tail) {
        SWM_HIP(ctx, hipEventRecord(job->acc_done, ctx->stream));
        if (lane >= 0 && !follow) ctx->set_acc_event[lane] = job->acc_done;
        if (follow) ctx->twin_sorted_event = job->acc_done;
    }
    job->joint_tail = defer_tail;
    if (defer_tail) {  // the bucket stage runs with the other jobs of the round (msm_flush_tails)
        ctx->pending_tails.push_back(job);
        return SWM_OK;
    }
    if (flat && !lat && lane >= 0 && rb == 256 && tab.shard_world <= 1) {
        ctx->lazy_tail = job;  // shaped and launched by the next msm_enqueue (thin) or by the flush of the round (wide)
        return SWM_OK;
    }
    MsmJob* one[1] = {job};
    return msm_launch_tails(ctx, one, 1);
}

// Shape of the bucket stage of a large flat job (r04).  The stage is a dependent chain of group operations — 2 m for the walk
// over the m buckets of a lane plus ~20 across the workgroup — on (2^(c-1) / m) / 256 workgroups of 144 KB of LDS each.  With
// m = 8 that is one workgroup on EVERY CU for ~0.5 ms: nothing that needs LDS (the partition and the bin sort of the next
// job) runs beside it, and the r04 timeline showed exactly that serialisation between consecutive accumulations.  With
// m = 32 the chain is ~84 steps (~1 ms) on a quarter of the CUs: as CU-time half the cost, and the rest of the chip goes
// on with the next job.  So: WIDE (m = 8, shortest chain) when the stage is exposed — the last job before a flush, i.e. of
// a prover round —, THIN (m = 32) when another job follows.  The job's result slot takes up to 256 workgroup results either way.
static void msm_tail_shape(MsmJob* j, bool wide) {
    // log2 buckets per lane: thin 32 (16 / 8 per lane: 51.5 / 52.5 vs 52.0 ms, r05); wide 4 before the workgroup cap applies
    // (the cap is the job's max_blocks: 256, which makes it 8 per lane for 2^19 buckets)
    static constexpr unsigned thin_log_m = 5u, wide_log_m = 2u;
    unsigned log_m = wide ? wide_log_m : thin_log_m;
    while (((j->pl.maxB >> log_m) + j->rb - 1) / j->rb > j->max_blocks) log_m++;
    j->log_m = log_m;
    j->red_blocks = std::max(1u, ((j->pl.maxB >> log_m) + j->rb - 1) / j->rb);
    j->blk_hi = j->red_blocks;  // (no bucket-range share on this path: blk_lo = 0, blk_low = 0)
}
// Measured in r06 and not kept (CHANGELOG.md): (i) the WIDE stage with four lanes per chain (msm_bucket_reduce<256, FormTEQuad>, 1024
// threads per workgroup, four waves per SIMD on the same 144 KB): 554 instead of 462 us per 2^19-bucket job — the quad form's 12
// products per addition and its lane exchanges cost more than the idle lanes of the tree steps; (ii) the second operand of every step
// in registers, the partial sum of a lane's NEXT walk step requested from HBM while the current addition runs (the three flat-load
// waits per "run += partial" step gone, 240 VGPRs): 524 instead of 467 us — the lone wave's cycles without an issue are not its
// operand loads (with typed loads of the whole operand up front, one wait instead of three and nothing else changed: 462.8 vs 463.3 us).
// What those cycles WERE: LDS bank conflicts and barriers — see RedSlot and the walk's wave-level vote (msm_bucket_reduce).
int msm_launch_lazy_tail(swm_ctx* ctx, bool wide) {
    MsmJob* j = ctx->lazy_tail;
    if (!j) return SWM_OK;
    ctx->lazy_tail = nullptr;
    msm_tail_shape(j, wide);
    MsmJob* one[1] = {j};
    return msm_launch_tails(ctx, one, 1);
}

// One bucket-stage launch + the downloads for k <= TAIL_MAX jobs, on the stream of the last one.
int msm_launch_tails(swm_ctx* ctx, MsmJob** jobs, int k) {
    if (k <= 0) return SWM_OK;
    hipStream_t st = jobs[k - 1]->stream;
    struct StreamSwap {
        swm_ctx* c;
        hipStream_t old;
        ~StreamSwap() { c->stream = old; }
    } swap{ctx, ctx->stream};
    ctx->stream = st;
    TailBatch batch;
    memset(&batch, 0, sizeof(batch));
    unsigned max_red = 1, max_win = 1;
    for (int i = 0; i < k; i++) {
        MsmJob* j = jobs[i];
        if (j->acc_done && j->stream != nullptr && j->needs_acc_wait) SWM_HIP(ctx, hipStreamWaitEvent(st, j->acc_done, 0));
        batch.j[i].partial = j->d_partial;
        batch.j[i].seg_off = j->d_seg_off;
        batch.j[i].hist = j->d_hist;
        batch.j[i].seg = j->seg;
        batch.j[i].out = j->host_dev;
        batch.j[i].acc = j->d_acc;
        batch.j[i].status = j->d_status;
        batch.j[i].entries = j->d_entries;
        batch.j[i].host_flags = j->host_flags_dev;
        batch.j[i].log_m = j->log_m;
        batch.j[i].red_blocks = j->red_blocks;
        batch.j[i].blk_lo = j->blk_lo;
        batch.j[i].blk_hi = j->blk_hi;
        batch.j[i].blk_low = j->blk_low;
        batch.j[i].big_nseg = j->big_nseg;
        batch.j[i].L = j->pl;
        max_red = std::max(max_red, j->red_blocks);
        max_win = std::max(max_win, j->pl.nwin);
    }
    const bool te = jobs[0]->te;  // every job of a launch has the same point form (msm_flush_tails groups them)
    if (jobs[0]->quad) {
        const size_t lds = (3 * (size_t)jobs[0]->rb + 1) * sizeof(PlainSlot);
        const dim3 grid(max_red, max_win, (unsigned)k);
        if (jobs[0]->rb == 256) {
            SWM_TRY(allow_big_lds(ctx, 8, (const void*)msm_bucket_reduce<256, FormTEQuad>, lds));
            SWM_LAUNCH(ctx, "msm_bucket_reduce", (msm_bucket_reduce<256, FormTEQuad>), grid, dim3(1024), lds, batch);
        } else {
            SWM_TRY(allow_big_lds(ctx, 9, (const void*)msm_bucket_reduce<128, FormTEQuad>, lds));
            SWM_LAUNCH(ctx, "msm_bucket_reduce", (msm_bucket_reduce<128, FormTEQuad>), grid, dim3(512), lds, batch);
        }
    } else if (te && jobs[0]->low) {
        SWM_LAUNCH(ctx, "msm_bucket_reduce", (msm_bucket_reduce_low<256>), dim3(max_red, max_win, (unsigned)k), dim3(RED_BLOCK),
                   (RED_BLOCK + 1) * sizeof(RedSlot), batch);
    } else if (te) {
        SWM_TRY(allow_big_lds(ctx, 7, (const void*)msm_bucket_reduce<256, FormTE>, (3 * RED_BLOCK + 1) * sizeof(RedSlot)));
        SWM_LAUNCH(ctx, "msm_bucket_reduce", (msm_bucket_reduce<256, FormTE>), dim3(max_red, max_win, (unsigned)k), dim3(RED_BLOCK),
                   (3 * RED_BLOCK + 1) * sizeof(RedSlot), batch);
    } else {
        SWM_TRY(allow_big_lds(ctx, 2, (const void*)msm_bucket_reduce<256, FormXYZZ>, (3 * RED_BLOCK + 1) * sizeof(RedSlot)));
        SWM_LAUNCH(ctx, "msm_bucket_reduce", (msm_bucket_reduce<256, FormXYZZ>), dim3(max_red, max_win, (unsigned)k), dim3(RED_BLOCK),
                   (3 * RED_BLOCK + 1) * sizeof(RedSlot), batch);
    }
    for (int i = 0; i < k; i++) {
        SWM_HIP(ctx, hipEventRecord(jobs[i]->done, st));
        jobs[i]->tail_pending = false;
    }
    return SWM_OK;
}

// Shape of a JOINT stage (r05): the jobs of one launch share the chip, so they share its 256 CUs — the smallest common number
// of buckets per lane (a power of two from 4) with which the workgroups of all k jobs fit one per CU.  One job: 8 per lane on
// every CU (the wide shape); two 2^19-bucket jobs: 16 per lane, 128 workgroups each; four: 32 per lane; a round of three
// 2^17-bucket jobs and one of 2^19 (the Merkle circuit's round 1 with its prefix tables): 16 per lane, 32 + 32 + 32 + 128.
// (r02 - r04: 64 workgroups per job whatever the job.)
static void msm_joint_shape(MsmJob** jobs, size_t k) {
    for (size_t i = 0; i < k; i++)  // one-lane 256-chain stages over ONE bucket set, no bucket-range share
        if (jobs[i]->rb != 256 || jobs[i]->quad || jobs[i]->pl.nwin != 1 || jobs[i]->blk_lo != 0 || jobs[i]->blk_low != 0 ||
            jobs[i]->blk_hi != jobs[i]->red_blocks)
            return;
    for (unsigned log_m = 2; log_m <= 10; log_m++) {
        unsigned total = 0;
        bool ok = true;
        for (size_t i = 0; i < k; i++) {
            const unsigned nb = std::max(1u, ((jobs[i]->pl.maxB >> log_m) + 255u) / 256u);
            ok = ok && nb <= jobs[i]->max_blocks;
            total += nb;
        }
        if (!ok || total > 256) continue;
        for (size_t i = 0; i < k; i++) {
            jobs[i]->log_m = log_m;
            jobs[i]->red_blocks = std::max(1u, ((jobs[i]->pl.maxB >> log_m) + 255u) / 256u);
            jobs[i]->blk_hi = jobs[i]->red_blocks;
        }
        return;
    }
}

// Launches the deferred bucket stages (all jobs enqueued with defer_tail since the last flush), TAIL_MAX per launch.
int msm_flush_tails(swm_ctx* ctx) {
    SWM_TRY(msm_launch_lazy_tail(ctx, true));  // nothing follows it before the round's results are awaited
    std::vector<MsmJob*> jobs;
    jobs.swap(ctx->pending_tails);
    for (size_t i = 0; i < jobs.size();) {  // one launch per run of up to TAIL_MAX jobs of the same workgroup width
        size_t k = 1;
        while (i + k < jobs.size() && k < TAIL_MAX && jobs[i + k]->rb == jobs[i]->rb && jobs[i + k]->te == jobs[i]->te &&
               jobs[i + k]->quad == jobs[i]->quad && jobs[i + k]->low == jobs[i]->low)
            k++;
        msm_joint_shape(jobs.data() + i, k);
        SWM_TRY(msm_launch_tails(ctx, jobs.data() + i, (int)k));
        i += k;
    }
    return SWM_OK;
}

// Waits for a job's download, releases its slot and checks the status words.
static int msm_finish_wait(swm_ctx* ctx, MsmJob* job) {
    if (job->tail_pending) SWM_TRY(msm_flush_tails(ctx));  // awaited before its round was flushed
    // a twin's follower copies this job's descriptors out of the slot's arrays: the slot is not handed on before that
    if (job->twin_copied) (void)hipEventSynchronize(job->twin_copied);
    job->twin_copied = nullptr;
    job->twin.ready = false;
    const hipError_t werr = hipEventSynchronize(job->done);
    job->active = false;
    ctx->slot_busy[job->slot] = false;  // released whatever the wait returned: the slot must not stay blocked for good
    SWM_HIP(ctx, werr);
    ctx->stat_msm_adds += job->host_flags[1];  // entries the sort placed = non-zero digits
    ctx->stat_msm_zero_points += job->host_flags[2];
    if (job->host_flags[0])
        return set_err(ctx, SWM_ERR_INVALID_ARG, "msm: a scalar is not a canonical field element (>= r)");
    return SWM_OK;
}

// point form of a job's workgroup results on the host: XYZZ, or extended twisted Edwards (job->te; g1.cuh)
struct HostXYZZ {
    static G1XYZZ identity() { return g1_xyzz_identity(); }
    static void add(G1XYZZ& a, const G1XYZZ& q) { g1_add(a, q); }
    static G1XYZZ dbl(const G1XYZZ& p) { return g1_dbl(p); }
    static bool is_inf(const G1XYZZ& p) { return g1_is_inf(p); }
    static G1XYZZ to_xyzz(const G1XYZZ& p) { return p; }
};
struct HostTE {
    static G1XYZZ identity() { return g1te_identity(); }
    static void add(G1XYZZ& a, const G1XYZZ& q) { g1te_add(a, q); }
    static G1XYZZ dbl(const G1XYZZ& p) { return g1te_dbl(p); }
    static bool is_inf(const G1XYZZ& p) { return fp_is_zero(p.x) && fp_eq(p.y, p.zzz); }  // (0 : z : 0 : z)
    static G1XYZZ to_xyzz(const G1XYZZ& p) { return g1te_to_xyzz(p); }
};
// over the workgroups blk in [lo, hi): sum of A, sum of R, and sum of (blk - lo) R_blk by suffix sums
template <class HF>
static void fold_range(const G1XYZZ* h, unsigned lo, unsigned hi, G1XYZZ* sa_out, G1XYZZ* sr_out, G1XYZZ* wt_out) {
    G1XYZZ sa = HF::identity(), suffix = HF::identity(), wt = HF::identity();
    for (unsigned blk = hi; blk-- > lo;) {
        HF::add(sa, h[2 * blk]);
        HF::add(suffix, h[2 * blk + 1]);                // Suffix_blk = sum_{u >= blk} R_u
        if (blk > lo) HF::add(wt, suffix);              // sum_{blk > lo} Suffix_blk = sum (blk - lo) R_blk
    }
    *sa_out = sa;
    *sr_out = suffix;
    *wt_out = wt;
}
// flat schedule: one window, up to 256 workgroups, folded in groups of 16 consecutive workgroups (independent: host
// workers); with g0 = 16 g the first workgroup of group g:  sum blk R_blk = sum_g [W_g + 16 g SR_g], and sum_g g SR_g
// comes from suffix sums over the groups, times 16 by four doublings.
static constexpr unsigned FOLD_GROUP = 16;
template <class HF>
static void fold_groups_combine(const G1XYZZ* ga, const G1XYZZ* gr, const G1XYZZ* gw, unsigned G, G1XYZZ* sum_a, G1XYZZ* weighted) {
    G1XYZZ sa = HF::identity(), wsum = HF::identity(), suffix = HF::identity(), gsum = HF::identity();
    for (unsigned g = G; g-- > 0;) {
        HF::add(sa, ga[g]);
        HF::add(wsum, gw[g]);
        HF::add(suffix, gr[g]);
        if (g > 0) HF::add(gsum, suffix);  // sum_g g SR_g
    }
    for (int k = 0; k < 4; k++) gsum = HF::dbl(gsum);
    static_assert(FOLD_GROUP == 16, "four doublings");
    HF::add(wsum, gsum);
    *sum_a = sa;
    *weighted = wsum;
}

// Host fold of a job's downloaded workgroup results.  `pool`: the context's host workers for the independent parts of
// ONE job (null: serial — the caller is already folding several jobs side by side).  Returns false on an inconsistent
// window layout (internal error).
// `groups` (flat schedule, more than one group): the group sums (ga | gr | gw, G each) were already computed by the caller.
template <class HF>
static bool msm_fold_form(const MsmJob* job, HostPool* pool, G1XYZZ* result, const G1XYZZ* groups) {
    // host: per window  X_w = sum_blk A_blk + 2^shift * W_w,  W_w = sum_blk blk R_blk  (2^shift = RED_BLOCK * m buckets per
    // workgroup; W_w by suffix sums over the <= 16 workgroups).  The windows are independent: they are folded on the
    // context's host workers.  Then Horner over the windows (high -> low, c_w doublings each), with the 2^shift of W_w
    // riding on the window's own doublings:  T <- 2^shift (2^(c_w - shift) T + W_w) + sum_blk A_blk.
    const WinLayout& pl = job->pl;
    const unsigned nb = job->red_blocks;
    unsigned shift = job->log_m;
    for (unsigned v = job->rb; v > 1; v >>= 1) shift++;
    G1XYZZ sum_a[MAX_WIN], weighted[MAX_WIN];
    auto fold_window = [&](int w) {
        G1XYZZ sr;
        fold_range<HF>(job->host + (size_t)w * nb * 2, 0, nb, &sum_a[w], &sr, &weighted[w]);
    };
    auto run = [&](int n, const std::function<void(int)>& fn) {
        if (pool) pool->parallel_for(n, fn);
        else
            for (int i = 0; i < n; i++) fn(i);
    };
    if (pl.nwin == 1 && nb > FOLD_GROUP && (pool || groups)) {
        const unsigned G = (nb + FOLD_GROUP - 1) / FOLD_GROUP;
        std::vector<G1XYZZ> own;
        if (!groups) {
            own.resize(3 * (size_t)G);
            G1XYZZ* o = own.data();
            run((int)G, [&](int g) {
                fold_range<HF>(job->host, FOLD_GROUP * g, std::min(nb, FOLD_GROUP * (g + 1)), &o[g], &o[G + g], &o[2 * G + g]);
            });
            groups = o;
        }
        fold_groups_combine<HF>(groups, groups + G, groups + 2 * G, G, &sum_a[0], &weighted[0]);
    } else if (nb > 1 && pl.nwin > 1) {
        run((int)pl.nwin, fold_window);
    } else {
        for (unsigned w = 0; w < pl.nwin; w++) fold_window((int)w);
    }
    G1XYZZ total_pt = HF::identity();
    for (unsigned w = pl.nwin; w-- > 0;) {
        unsigned cw = pl.c[w];
        if (!HF::is_inf(weighted[w])) {
            if (cw >= shift) {
                for (unsigned k = shift; k < cw; k++) total_pt = HF::dbl(total_pt);
                HF::add(total_pt, weighted[w]);
                cw = shift;
            } else {  // a window narrower than one workgroup's span cannot have content beyond workgroup 0
                return false;
            }
        }
        for (unsigned k = 0; k < cw; k++) total_pt = HF::dbl(total_pt);
        HF::add(total_pt, sum_a[w]);
    }
    *result = HF::to_xyzz(total_pt);
    return true;
}
static bool msm_fold(const MsmJob* job, HostPool* pool, G1XYZZ* result, const G1XYZZ* groups = nullptr) {
    return job->te ? msm_fold_form<HostTE>(job, pool, result, groups) : msm_fold_form<HostXYZZ>(job, pool, result, groups);
}

// Host workers of a context: 15 (7 on hosts with fewer than 32 hardware threads), but never more than this process's share of
// the machine: hardware threads / ranks on the node (LOCAL_WORLD_SIZE of the launcher, or the context's shard world for ranks
// that are threads of one process) minus the proving thread — eight ranks with 15 polling workers each would be 120 spinning
// threads beside the provers, torch and the RCCL proxies.  SWM_POOL_WORKERS overrides.
static HostPool* host_pool_of(swm_ctx* ctx) {
    if (!ctx->host_pool) {
        unsigned hw = std::thread::hardware_concurrency();
        unsigned want = hw > 1 ? std::min(hw - 1, hw >= 32 ? 15u : 7u) : 0u;
        unsigned ranks = std::max(1u, ctx->shard_world);
        ranks = std::max(ranks, (unsigned)env_switch("LOCAL_WORLD_SIZE", 1, 1, 4096));  // (torchrun's: ranks that share this host's cores)
        if (ranks > 1 && hw) want = std::min(want, std::max(1u, hw / ranks) - 1);
        want = (unsigned)env_switch("SWM_POOL_WORKERS", (long)want, 0, 64);  // (diagnostic: the host fold's worker threads)
        ctx->host_pool = new HostPool(want);
    }
    return ctx->host_pool;
}
// how long the workers poll for the fold that follows a wait (microseconds; SWM_POOL_SPIN_US, 0 = they sleep)
static unsigned pool_spin_us() {
    static const unsigned us = (unsigned)env_switch("SWM_POOL_SPIN_US", 1500, 0, 100000);
    return us;
}

int msm_finish(swm_ctx* ctx, MsmJob* job, G1XYZZ* result) {
    *result = g1_xyzz_identity();
    ctx->msm_since_wait = 0;
    if (!job->active) return SWM_OK;
    static const bool trace = env_flag("SWM_TRACE");
    auto tw0 = std::chrono::steady_clock::now();
    if (pool_spin_us()) host_pool_of(ctx)->arm(pool_spin_us());  // the fold follows the wait at once: the workers poll for it meanwhile
    SWM_TRY(msm_finish_wait(ctx, job));
    auto tw1 = std::chrono::steady_clock::now();
    if (!msm_fold(job, host_pool_of(ctx), result)) return set_err(ctx, SWM_ERR_INTERNAL, "msm: inconsistent window fold");
    if (trace)
        fprintf(stderr, "[swm trace]   msm_finish n=%zu: waited %.3f ms, host fold %.3f ms\n", job->n,
                std::chrono::duration<double, std::milli>(tw1 - tw0).count(),
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw1).count());
    return SWM_OK;
}

// Several jobs at once (the commitments of a prover round): wait for all of them, then fold them side by side, one job
// per host worker (a fold is a serial chain of ~3 additions per workgroup result: four of them back to back were
// ~0.25 ms between the last kernel of a round and its Fiat-Shamir challenge).
int msm_finish_many(swm_ctx* ctx, MsmJob** jobs, int k, G1XYZZ* results) {
    static const bool trace = env_flag("SWM_TRACE");
    ctx->msm_since_wait = 0;
    auto tw0 = std::chrono::steady_clock::now();
    std::vector<int> live;
    for (int i = 0; i < k; i++) {
        results[i] = g1_xyzz_identity();
        if (jobs[i] && jobs[i]->active) live.push_back(i);
    }
    int rc = SWM_OK;
    if (!live.empty() && pool_spin_us()) host_pool_of(ctx)->arm(pool_spin_us());  // the folds follow the wait at once: the workers poll for them meanwhile
    // Jobs with their own tail finish one after the other (the tail stream runs them in order): each is folded as soon as
    // its results are there, while the GPU is still busy with the tails of the later ones — only the last fold is exposed.
    // Jobs of one joint tail launch finish together: those are folded side by side below.
    bool pipelined = false;
    for (int i : live) pipelined = pipelined || !jobs[i]->joint_tail;
    if (pipelined && live.size() > 1) {
        bool ok_all = true;
        for (int i : live) {
            int r = msm_finish_wait(ctx, jobs[i]);
            if (rc == SWM_OK) rc = r;
            if (r == SWM_OK && !msm_fold(jobs[i], host_pool_of(ctx), &results[i])) ok_all = false;
        }
        if (rc != SWM_OK) return rc;
        if (!ok_all) return set_err(ctx, SWM_ERR_INTERNAL, "msm: inconsistent window fold");
        if (trace)
            fprintf(stderr, "[swm trace]   msm_finish_many k=%zu: waits and folds interleaved, %.3f ms\n", live.size(),
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw0).count());
        return SWM_OK;
    }
    for (int i : live) {  // every job is waited for and released even if one reports an error
        int r = msm_finish_wait(ctx, jobs[i]);
        if (rc == SWM_OK) rc = r;
    }
    if (rc != SWM_OK) return rc;
    auto tw1 = std::chrono::steady_clock::now();
    // tasks: one per group of 16 workgroup results of a flat job, one per job otherwise
    struct Task {
        int job;
        int group;  // -1: the whole job
    };
    std::vector<Task> tasks;
    std::vector<std::vector<G1XYZZ>> groups(live.size());
    for (size_t t = 0; t < live.size(); t++) {
        const MsmJob* j = jobs[live[t]];
        if (j->pl.nwin == 1 && j->red_blocks > FOLD_GROUP) {
            const unsigned G = (j->red_blocks + FOLD_GROUP - 1) / FOLD_GROUP;
            groups[t].resize(3 * (size_t)G);
            for (unsigned g = 0; g < G; g++) tasks.push_back({(int)t, (int)g});
        } else {
            tasks.push_back({(int)t, -1});
        }
    }
    std::vector<char> ok(live.size(), 1);
    host_pool_of(ctx)->parallel_for((int)tasks.size(), [&](int i) {
        const Task& tk = tasks[i];
        const MsmJob* j = jobs[live[tk.job]];
        if (tk.group < 0) {
            ok[tk.job] = msm_fold(j, nullptr, &results[live[tk.job]]);
        } else {
            const unsigned G = (j->red_blocks + FOLD_GROUP - 1) / FOLD_GROUP, g = (unsigned)tk.group;
            G1XYZZ* o = groups[tk.job].data();
            if (j->te) fold_range<HostTE>(j->host, FOLD_GROUP * g, std::min(j->red_blocks, FOLD_GROUP * (g + 1)), &o[g], &o[G + g], &o[2 * G + g]);
            else fold_range<HostXYZZ>(j->host, FOLD_GROUP * g, std::min(j->red_blocks, FOLD_GROUP * (g + 1)), &o[g], &o[G + g], &o[2 * G + g]);
        }
    });
    host_pool_of(ctx)->parallel_for((int)live.size(), [&](int t) {  // per job: the combine of its group sums
        if (!groups[t].empty()) ok[t] = msm_fold(jobs[live[t]], nullptr, &results[live[t]], groups[t].data());
    });
    for (char c : ok)
        if (!c) return set_err(ctx, SWM_ERR_INTERNAL, "msm: inconsistent window fold");
    if (trace)
        fprintf(stderr, "[swm trace]   msm_finish_many k=%zu: waited %.3f ms, host folds %.3f ms\n", live.size(),
                std::chrono::duration<double, std::milli>(tw1 - tw0).count(),
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tw1).count());
    return SWM_OK;
}

// Synchronous form on the context's stream (K1 ABI).
int msm_run(swm_ctx* ctx, const G1Affine* d_bases, const G1Affine* d_bases28, const void* d_scalars, size_t n, int mont,
            G1XYZZ* result, MsmInfMask inf, MsmTable tab) {
    MsmJob job;
    SWM_TRY(msm_enqueue(ctx, -1, d_bases, d_bases28, d_scalars, n, mont, &job, inf, false, tab));
    return msm_finish(ctx, &job, result);
}

}  // namespace swm
