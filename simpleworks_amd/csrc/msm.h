// msm.h — internal interface of the G1 MSM (msm.hip): synchronous and enqueue/finish forms.
#pragma once
#include "context.h"
#include "g1.cuh"

namespace swm {

static constexpr int MAX_WIN = 64;
// Window layout: the 254 recoding bits (253 scalar bits + 1 for the carry) are spread over nwin windows whose
// widths differ by at most one, so that no window is degenerate (a short top window would put n/2 points in two
// buckets).  boff[w] = first bucket of window w in the flat bucket array; window w has 2^(c[w]-1) buckets.
struct WinLayout {
    uint32_t nwin;
    uint32_t NB;
    uint32_t maxB;
    uint8_t c[MAX_WIN];
    uint16_t bit[MAX_WIN];
    uint32_t boff[MAX_WIN + 1];
};

// Precomputed window multiples of a resident base set (msm_table_build): row w holds 2^(w c) P_i for every point of the
// set, coordinates in the scaled form of the accumulation kernel.  `offset` is the index (within the set) of the first
// point of the MSM at hand.
struct MsmTable {
    const G1Affine* t28 = nullptr;  // [windows][stride]
    size_t stride = 0;
    unsigned c = 0;
    size_t offset = 0;
    // the same multiples as twisted Edwards rows (msm_table_build_te; g1.cuh): when set, the flat schedule accumulates and
    // reduces in that form (7 multiplications per mixed addition instead of 8M + 2S) and t28 is not needed
    const G1TE* te = nullptr;
    // Which base scalar i of the MSM belongs to: offset + (i >> blk_log) * bstride + (i & (2^blk_log - 1)).  The default
    // (blk_log = 31: one block) is the contiguous range [offset, offset + n); a rank of a sharded transform holds its
    // coefficients cyclically (blk_log = 0, bstride = G, offset = rank) or in G blocks (blk_log = log2(n / G^2)), and
    // commits to them where they are (flat schedule only).
    unsigned blk_log = 31;
    size_t bstride = 0;
    // scalar_stride != 1: the SCALARS follow the block map of the bases too — scalar i of the MSM is
    // d_scalars[(i >> blk_log) * bstride + (i & (2^blk_log - 1))] (a rank of a sharded proof that takes the blocks g, g + G, ... of a
    // polynomial all ranks hold: bstride = G << blk_log, offset = g << blk_log, the scalar pointer advanced likewise; blk_log = 0 is
    // the plain cyclic split; flat schedule only)
    size_t scalar_stride = 1;
    // One proof over G ranks, split by BUCKET range (flat schedule, every rank holds all n scalars): the MSM takes all its
    // points but keeps only the digits whose bucket lies in the rank's share of the bucket-stage workgroups — windows
    // narrower than the table width (the top one) are split by point range instead, their few low buckets are reduced by
    // every rank.  Accumulation, sort AND bucket stage shrink with G; the per-rank sums add up to the MSM like those of a
    // point-range split.  shard_world <= 1: off.
    unsigned shard_rank = 0, shard_world = 0;
    bool any() const { return t28 != nullptr || te != nullptr; }
    bool contiguous() const { return blk_log >= 31; }
};
unsigned msm_table_windows(unsigned c);
WinLayout msm_table_layout(unsigned c);
// window width for the table of a base set of n points (0: no table, the set is too small to profit)
unsigned msm_table_width(size_t n_bases);
// builds the table of d_points[0 .. n) (affine, radix 2^384) for width c into *out (hipMalloc'd, windows * n points);
// row 0 is the plain scaled copy of the set (what msm_scale_bases_run produces)
int msm_table_build(swm_ctx* ctx, const G1Affine* d_points, size_t n, unsigned c, G1Affine** out);
// The twisted Edwards form of the same table (144-B rows).  The points must lie in the prime-order subgroup (the unified
// addition law has exceptional pairs among points of even order): the prover's committer keys do by construction or by
// their checked deserialisation; msm_subgroup_check establishes it for caller-supplied sets.  *out stays null (and the
// call returns SWM_OK) when a point has no image under the map — the caller then keeps the XYZZ table.
int msm_table_build_te(swm_ctx* ctx, const G1Affine* d_points, size_t n, unsigned c, G1TE** out);
// *ok = every point is the identity (0, 0) or lies on the curve and in the prime-order subgroup ([r]P = O)
int msm_subgroup_check(swm_ctx* ctx, const G1Affine* d_points, size_t n, bool* ok);
// scaled copy + the best table that fits for a resident base set (see msm.hip); frees nothing of the caller's
int msm_install_bases(swm_ctx* ctx, const G1Affine* d_points, size_t n, bool in_subgroup, G1Affine** d28, G1TE** te, unsigned* c,
                      uint32_t* d_inf_mask = nullptr);
// SWM_MSM_TE=0 keeps the XYZZ tables everywhere (A/B measurements)
bool msm_te_enabled();
// HBM left for a table of `bytes` bytes? (hipMemGetInfo, keeping a quarter of the free memory for the prover's temporaries)
bool msm_table_fits(size_t bytes);

// a non-blocking stream for one role of the prover (msm.hip)
hipError_t msm_create_stream(hipStream_t* out);
// does msm_enqueue run the precomputed-window ("flat") schedule for n points on this table?  (callers that hand over a
// strided layout have to know: only that schedule maps scalars to bases through MsmTable::blk_log / bstride)
bool msm_flat_applies(const MsmTable& tab, size_t n);

// Twin jobs (r06).  The plain and the degree-shifted commitment of one polynomial are MSMs of the SAME scalars over two base sets
// whose tables have the same width: digits, coarse histogram, partition and bin sort of the second job would repeat the first
// one's work entry for entry — only the table ROW of an entry differs (w * stride + offset + i of the other table).  The first
// job (LEAD) therefore writes a second sorted array with the rows of the second table from its bin sort, and the second job
// (FOLLOW) copies the bucket / segment descriptors (~12 MB) instead of sorting.  Either side falls back to the ordinary
// schedule when the shapes do not match (different width, a job below the table schedule, a sharded split, SWM_MSM_TWIN=0).
struct MsmJob;
struct MsmTwin {
    enum Role { NONE = 0, LEAD = 1, FOLLOW = 2 };
    Role role = NONE;
    MsmTable tab2;           // LEAD: the table (with the offset) of the job that follows
    MsmJob* lead = nullptr;  // FOLLOW: the job whose sort this one takes over
};
// what a LEAD leaves for its follower
struct MsmTwinSrc {
    bool ready = false;
    const void* scalars = nullptr;
    int mont = 0, lane = -1;
    unsigned c = 0;
    const G1TE* te2 = nullptr;
    size_t stride2 = 0, offset2 = 0;
    // the lead's device arrays and the geometry they were built with (the follower's must be the same)
    const uint32_t *block = nullptr, *seg_start = nullptr, *seg_off = nullptr, *big_list = nullptr;
    size_t zero_words = 0, nseg_max = 0;
    uint32_t SEG = 0, big_nseg = 0, flat_bins = 0;
    hipEvent_t sorted = nullptr;  // recorded after the lead's bin sort
};

struct MsmJob {
    bool active = false;
    MsmTwinSrc twin;                    // LEAD only
    bool twin_kept_lane = false;        // FOLLOW: the job did not use its lane's scratch set, the next job may take the same lane
    hipEvent_t twin_copied = nullptr;   // LEAD: the follower's copy of this job's descriptors (msm_finish waits for it before the slot is released)
    size_t n = 0;
    WinLayout pl;
    unsigned big_nseg = 16;  // buckets with more segments than this were folded into their first partial sum
    bool quad = false;  // bucket stage with four lanes per chain (small twisted Edwards jobs)
    bool low = false;   // bucket stage by msm_bucket_reduce_low (one LDS slot per lane, weighted sums in d_acc)
    unsigned max_blocks = 256;  // most workgroups per window a later re-shaping of the stage may ask for (result slot, d_acc)
    unsigned red_blocks = 0, log_m = 0, rb = 256;  // bucket stage: workgroups per window, log2 buckets per lane, lanes per workgroup
    unsigned blk_lo = 0, blk_hi = 0, blk_low = 0;  // workgroups [blk_lo, blk_hi) and [0, blk_low) hold buckets of this rank (bucket-range split); the others emit the identity
    bool te = false;         // partial sums and workgroup results are twisted Edwards points (the host fold converts the total)
    int slot = 0;            // index of the pinned result slot (ctx->slot_busy)
    // deferred bucket stage (msm_flush_tails): what it reads / writes, the stream the job ran on and its "partials ready" event
    bool tail_pending = false;
    bool needs_acc_wait = false;
    hipStream_t stream = nullptr;
    hipEvent_t acc_done = nullptr;
    G1XYZZ *d_partial = nullptr, *d_wpart = nullptr, *d_acc = nullptr;
    const uint32_t *d_seg_off = nullptr, *d_hist = nullptr, *d_status = nullptr, *d_entries = nullptr;
    uint32_t seg = 0;  // segment bound of the accumulation: bucket b owns the partial sums d_seg_off[b] .. + ceil(d_hist[b] / seg)
    G1XYZZ* host = nullptr;  // pinned slot receiving nwin * red_blocks (A, R) pairs
    G1XYZZ* host_dev = nullptr;          // the same slot as the device addresses it (the bucket stage writes there)
    uint32_t* host_flags_dev = nullptr;
    bool joint_tail = false;             // bucket stage launched together with the other jobs of its round
    const uint32_t* host_flags = nullptr;  // tail of the slot: [0] != 0 when a scalar was not a canonical field element
    hipEvent_t done = nullptr;
};

// d_bases28: the bases with both coordinates pre-multiplied by 2^8 (msm_scale_bases_run), consumed by the 28-bit-limb
// inner loop of msm_accumulate; d_bases: the plain Montgomery (radix 2^384) points, used by the cold path.
// d_inf_mask (optional, n bits, zeroed by the caller): bit i is set when point i is the point at infinity (x = y = 0).
int msm_scale_bases_run(swm_ctx* ctx, const G1Affine* d_in, size_t n, G1Affine* d_out, uint32_t* d_inf_mask = nullptr);
// Which of a base set's points are the identity: the accumulation kernel adds whatever it is given, so their digits are
// dropped in msm_digits.  mask == nullptr (the prover: SRS powers are never the identity) skips the test.
struct MsmInfMask {
    const uint32_t* mask = nullptr;  // bit (first + i) belongs to point i of this MSM
    size_t first = 0;
};
// defer_tail: stop after the bucket folds and leave the bucket stage (the latency-bound tail) to msm_flush_tails, which
// runs the tails of every job enqueued so far in one launch.  The MsmJob must stay at its address until msm_finish.
int msm_enqueue(swm_ctx* ctx, int lane, const G1Affine* d_bases, const G1Affine* d_bases28, const void* d_scalars, size_t n,
                int mont, MsmJob* job, MsmInfMask inf = MsmInfMask(), bool defer_tail = false, MsmTable tab = MsmTable(),
                MsmTwin twin = MsmTwin());
int msm_launch_tails(swm_ctx* ctx, MsmJob** jobs, int k);
int msm_launch_lazy_tail(swm_ctx* ctx, bool wide);  // the held-back bucket stage of ctx->lazy_tail, if any
int msm_flush_tails(swm_ctx* ctx);
int msm_finish(swm_ctx* ctx, MsmJob* job, G1XYZZ* result);
int msm_finish_many(swm_ctx* ctx, MsmJob** jobs, int k, G1XYZZ* results);  // a round's jobs: one wait, folds side by side
int msm_run(swm_ctx* ctx, const G1Affine* d_bases, const G1Affine* d_bases28, const void* d_scalars, size_t n, int mont,
            G1XYZZ* result, MsmInfMask inf = MsmInfMask(), MsmTable tab = MsmTable());

}  // namespace swm
