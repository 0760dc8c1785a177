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

struct MsmJob {
    bool active = false;
    size_t n = 0;
    WinLayout pl;
    unsigned red_blocks = 0, log_m = 0;
    G1XYZZ* host = nullptr;  // pinned slot receiving nwin * red_blocks (A, R) pairs
    hipEvent_t done = nullptr;
};

// d_bases28: the bases with both coordinates pre-multiplied by 2^8 (msm_scale_bases_run), consumed by the 28-bit-limb
// inner loop of msm_accumulate; d_bases: the plain Montgomery (radix 2^384) points, used by the cold path.
int msm_scale_bases_run(swm_ctx* ctx, const G1Affine* d_in, size_t n, G1Affine* d_out);
int msm_enqueue(swm_ctx* ctx, int lane, const G1Affine* d_bases, const G1Affine* d_bases28, const void* d_scalars, size_t n,
                int mont, MsmJob* job);
int msm_finish(swm_ctx* ctx, MsmJob* job, G1XYZZ* result);
int msm_run(swm_ctx* ctx, const G1Affine* d_bases, const G1Affine* d_bases28, const void* d_scalars, size_t n, int mont,
            G1XYZZ* result);

}  // namespace swm
