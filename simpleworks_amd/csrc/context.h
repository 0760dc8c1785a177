// context.h — swm_ctx: one GPU, one HIP stream, a growable HBM workspace, cached NTT twiddle tables and the
// per-kernel HIP-event log behind swm_profile_*.  Internal to libswmarlin.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <map>
#include <string>
#include <vector>
#include "../../include/swmarlin.h"
#include "host/pool.h"
#include "switches.h"

namespace swm {

struct MsmJob;
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
};

struct NttTables {
    // two-level powers of a base b: lo[i] = b^i (i < 1024), hi[i] = b^(1024 i)
    void* lo = nullptr;
    void* hi = nullptr;
    size_t hi_len = 0;
};

struct ProfAgg {
    int calls = 0;
    double ms = 0;
};
struct ProfPending {
    std::string name;
    hipEvent_t e0, e1;
};

}  // namespace swm

struct swm_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    char err[512] = {0};
    // named scratch buffers, grown on demand and reused between calls (kept resident: 288 GB of HBM)
    std::map<std::string, swm::DevBuf> scratch;
    // NTT root tables keyed by (log_n << 1 | inverse); coset tables keyed by inverse flag
    std::map<uint64_t, swm::NttTables> ntt_tables;
    std::map<uint64_t, void*> ntt_small;  // per-radix intra-tile twiddles keyed by (log_r << 1 | inverse); per-pass twiddle tables
    size_t ntt_pass_table_bytes = 0;      // HBM held by the per-pass twiddle tables of the lazy transform (capped, ntt.hip)
    // asynchronous MSM lanes: auxiliary streams (the prover alternates between two of them), a fork event, pinned result slots with their completion events
    static constexpr int MSM_SLOTS = 8;
    static constexpr int MSM_LANES = 4;  // sort | accumulation 0 | accumulation 1 | bucket stage
    hipStream_t aux_stream[MSM_LANES] = {nullptr, nullptr, nullptr, nullptr};
    std::vector<hipStream_t> spare_streams;  // never used: placeholders / rejected candidates of the hardware-queue placement (msm_enqueue)
    hipEvent_t fork_event = nullptr;
    void* pinned = nullptr;
    void* pinned_dev = nullptr;  // device address of `pinned`
    void* h2d_stage = nullptr;   // small pinned staging area for the witness upload of small proofs (marlin.hip, upload_small)
    size_t h2d_stage_used = 0;
    hipEvent_t slot_event[MSM_SLOTS] = {nullptr};
    bool slot_busy[MSM_SLOTS] = {false};  // enqueued and not yet collected by msm_finish
    hipEvent_t acc_event[MSM_SLOTS] = {nullptr};  // "partial sums ready" per slot (stage A -> stage T, deferred bucket stages)
    hipEvent_t sort_event[MSM_SLOTS] = {nullptr};  // "sorted" per slot (stage S -> stage A)
    hipEvent_t set_acc_event[4] = {nullptr, nullptr, nullptr, nullptr};  // accumulation that last read each per-lane scratch set (not owned)
    // twin jobs (msm.h): the follower's copy that last read a lane's segment descriptors, the accumulation that last read the
    // second sorted array (neither owned), and the "copied" events per result slot (owned)
    hipEvent_t lane_copy_event[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t twin_sorted_event = nullptr;
    hipEvent_t twin_copy_event[MSM_SLOTS] = {nullptr};
    std::vector<swm::MsmJob*> pending_tails;      // jobs whose bucket stage waits for msm_flush_tails
    swm::MsmJob* lazy_tail = nullptr;             // large job whose bucket stage is shaped by what follows it (msm.hip, msm_tail_shape)
    int next_slot = 0;
    std::multimap<size_t, void*> pool;  // freed device blocks by capacity (stream-ordered reuse)
    // work log since the last swm_profile_reset (the prove() byte count is the sum over logged calls)
    uint64_t stat_msm_digits = 0;  // points x windows (zero digits included)
    uint64_t stat_msm_twins = 0;   // jobs that took the sort of their twin (msm.h: MsmTwin)
    uint64_t stat_msm_adds = 0;    // NON-ZERO digits = bucket entries = the mixed additions msm_accumulate performs (counted by
                                   // the sort on the device; collected in msm_finish)
    uint64_t stat_msm_zero_points = 0;  // MSM points with a zero scalar or an identity base (nothing but their scalar is read)
    uint64_t stat_spmv_nnz = 0;
    // the K1-K3 calls themselves while profiling is on ("the log is mandatory"): ('m', points),
    // ('n', log2 size), ('s', rows) + ('z', non-zeros); bench.py replays this list on the CPU oracle
    std::vector<std::pair<char, uint64_t>> call_log;
    void log_call(char kind, uint64_t v) {
        if (profiling && call_log.size() < 8192) call_log.push_back({kind, v});
    }
    uint64_t stat_msm_calls = 0, stat_msm_points = 0, stat_ntt_calls = 0, stat_ntt_elems = 0, stat_spmv_calls = 0,
             stat_spmv_rows = 0;
    int profiling = 0;  // 0 off, 1 every launch, 2 the dominant kernel (msm_accumulate) only
    bool prof_open = false;
    std::map<std::string, swm::ProfAgg> prof;
    std::vector<swm::ProfPending> pending;
    std::vector<hipEvent_t> event_pool;
    // point-range sharding of the prover's MSMs over several contexts/GPUs (swm_set_msm_sharding)
    unsigned shard_rank = 0, shard_world = 1;
    swm_allgather_fn shard_allgather = nullptr;
    void* shard_user = nullptr;
    // the same exchange through RCCL inside the library (swm_rccl_init / swm_set_rccl_comm): ncclAllGather on the context's
    // stream between device staging buffers; librccl is resolved at run time (the copy already loaded in the process, else
    // librccl.so.1), so the library carries no link-time dependency on it
    void* rccl_comm = nullptr;
    bool rccl_own = false;
    void *rccl_send = nullptr, *rccl_recv = nullptr;
    size_t rccl_cap = 0;
    uint64_t stat_exchanges = 0, stat_exchange_bytes = 0;  // all-gathers issued / bytes contributed per rank
    swm::HostPool* host_pool = nullptr;  // created on first use (msm_finish), joined in swm_destroy
    // bulk draws from a caller-owned generator (sample_fr_bulk): a ring of two host chunks filled through the callback
    // and sent up on a copy stream of its own, so that the transfers run beside whatever the context's stream is doing
    hipStream_t copy_stream = nullptr;
    void* ext_pinned = nullptr;  // the ring's four host chunks (page-aligned host memory, registered with the runtime: see sample_fr_bulk)
    bool ext_registered = false;
    // MSMs below this many points run sort, accumulation and bucket stage on ONE stream of their lane (0: the library's default,
    // 131 072).  The prover sets it per proof: all of a small proof's commitments the same way (msm.hip, msm_enqueue)
    size_t msm_pipe_min = 0;
    // largest size requested so far for each kind of per-slot MSM scratch (hist, bucket_off, seg_off, big_list, points, acc): a
    // slot's buffer is always grown to that, not to what its current job needs (msm.hip, msm_enqueue)
    size_t msm_slot_bytes[6] = {0, 0, 0, 0, 0, 0};
    bool emulated_exchange = false;  // set by an EMULATED device exchange (measurement builds only, -DSWM_MEASURE_HOOKS: capi.hip)
    unsigned msm_since_wait = 0;  // MSM jobs enqueued since the last msm_finish*: 0 = nothing of this context is in flight
    uint32_t* ext_totals = nullptr;  // eight pinned words: the device's running total behind each of the last eight runs of a draw
    hipEvent_t ext_cnt_event[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // ... and when each has arrived
    hipEvent_t ext_event[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // [0 .. 3]: host chunk free again; [4]: destination may be written
};

struct swm_bases {
    void* d_points = nullptr;    // n x G1Affine (96 B, Montgomery radix 2^384)
    void* d_points28 = nullptr;  // same points, coordinates x 2^8 (radix 2^392) for the MSM inner loop
    uint32_t* d_inf_mask = nullptr;  // n bits, allocated only when some base is the point at infinity
    unsigned table_c = 0;            // != 0: the set has a table of window multiples — d_te when that is set, else d_points28 (row 0 = the scaled copy)
    void* d_te = nullptr;            // twisted Edwards form of the table (msm_table_build_te); d_points28 is then the n-point scaled copy
    size_t n = 0;
};

namespace swm {

int set_err(swm_ctx* ctx, int code, const char* fmt, ...);
// Waits for everything enqueued on the context's stream and on its auxiliary MSM streams, and forgets the jobs that
// were in flight.  Error paths call it before buffers that queued kernels may still read go back to the pool.
void drain_streams(swm_ctx* ctx);
// all-gather of `bytes` bytes per rank over the exchange configured on the context (RCCL communicator or caller's callback)
int shard_exchange(swm_ctx* ctx, const void* send, size_t bytes, void* recv);
// the same for DEVICE buffers, and the all-to-all of a sharded transform (chunk c of d_send -> rank c)
int shard_allgather_dev(swm_ctx* ctx, const void* d_send, size_t bytes, void* d_recv);
int shard_alltoall_dev(swm_ctx* ctx, const void* d_send, void* d_recv, size_t bytes_per_peer);

// Every extern "C" entry point that takes a context runs on that context's GPU, whatever device the calling thread had
// current (another context, torch, a thread that never called hipSetDevice); the caller's device is restored on exit.
struct DeviceGuard {
    int prev = -1;
    bool ok = true, switched = false;
    explicit DeviceGuard(const swm_ctx* ctx) {
        if (!ctx) return;
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != ctx->device) {
            ok = hipSetDevice(ctx->device) == hipSuccess;
            switched = ok && prev >= 0;
        }
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
};
#define SWM_ON_DEVICE(ctx)                                                                             \
    swm::DeviceGuard dev_guard__(ctx);                                                                 \
    if (!dev_guard__.ok) return swm::set_err(ctx, SWM_ERR_HIP, "hipSetDevice(%d) failed", (ctx)->device)
// returns device pointer of a scratch buffer with at least `bytes` capacity (contents undefined)
int scratch(swm_ctx* ctx, const char* name, size_t bytes, void** out);
void scratch_release(swm_ctx* ctx, const char* name);  // frees a named scratch buffer (one-off builders)
// pooled device allocations for the prover's polynomial temporaries
int pool_alloc(swm_ctx* ctx, size_t bytes, void** out, size_t* cap);
void pool_free(swm_ctx* ctx, void* p, size_t cap);
// two-level power tables of the 2^log_n-th root of unity (lo[i] = w^i, i < 1024; hi[i] = w^(1024 i))
int get_root_tables(swm_ctx* ctx, unsigned log_n, int inverse, NttTables** out);
void prof_begin(swm_ctx* ctx, const char* name);
void prof_end(swm_ctx* ctx);
void prof_flush(swm_ctx* ctx);

#define SWM_HIP(ctx, call)                                                                              \
    do {                                                                                                \
        hipError_t e__ = (call);                                                                        \
        if (e__ != hipSuccess)                                                                          \
            return swm::set_err(ctx, e__ == hipErrorOutOfMemory ? SWM_ERR_OOM : SWM_ERR_HIP, "%s: %s",  \
                                #call, hipGetErrorString(e__));                                         \
    } while (0)

#define SWM_TRY(expr)            \
    do {                         \
        int rc__ = (expr);       \
        if (rc__ != SWM_OK) return rc__; \
    } while (0)

// launch wrapper: bracket with events when profiling is on
#define SWM_LAUNCH(ctx, name, kernel, grid, block, shmem, ...)                      \
    do {                                                                            \
        swm::prof_begin(ctx, name);                                                 \
        hipLaunchKernelGGL(kernel, grid, block, shmem, (ctx)->stream, __VA_ARGS__); \
        swm::prof_end(ctx);                                                         \
        SWM_HIP(ctx, hipGetLastError());                                            \
    } while (0)

}  // namespace swm
