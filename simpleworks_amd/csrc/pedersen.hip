// pedersen.hip — native Pedersen CRH on ed-on-BLS12-377 and the Merkle tree built from it ("below the line":
// the tree BASELINE config #5 builds before it proves membership).
//
// What the reference does there (all through ark-crypto-primitives 0.3, on one CPU thread):
//   src/merkle_tree/simple_merkle_tree.rs:47-49   MerkleTree::<MerkleConfig>::new(&leaf_crh_params, &two_to_one_crh_params, leaves)
//   src/merkle_tree/common.rs:11-30               LeafHash / TwoToOneHash = PedersenCRHCompressor<EdwardsProjective, TECompressor, W>,
//                                                 W = 144 x 4 bits (leaves) and 128 x 4 bits (two digests)
//   src/hash/mod.rs:13-28                         the same Pedersen CRH called on its own
// pedersen::CRH::evaluate [U]: the input bytes, zero-padded to WINDOW_SIZE x NUM_WINDOWS bits, LSB-first inside a byte;
// bit j of window w adds generators[w][j] = 2^j g_w; the digest is the affine x coordinate of the sum (TECompressor),
// written as 32 little-endian bytes when it enters the next hash (to_bytes!).  A tree over n = 2^k leaves: n leaf digests,
// then n / 2, n / 4, ... 1 two-to-one digests of (left bytes || right bytes).
//
// On the GPU.  The curve's base field is BLS12-377 Fr (src/merkle_tree/common.rs:52), so this is ff.cuh's Fr arithmetic.
// Per window the 2^WINDOW_SIZE multiples v g_w are tabulated once per parameter set as (y - x, y + x, 2 d x y): a window
// costs ONE mixed addition in extended coordinates (7 Fr multiplications) instead of up to WINDOW_SIZE — the same group
// element, hence the same digest.  A hash is spread over L lanes (windows w = lane, lane + L, ...; the L partial sums meet
// in a shuffle tree of unified additions): L = 1 on the wide levels of a tree, 64 on the narrow ones near the root, where
// the chain of 128 dependent additions would otherwise be the whole latency.  One inversion per digest (binary Euclid, frinv.cuh:
// a Fermat inversion was 380 of the ~450 dependent multiplications of a narrow level).
// a = -1 is a square and d = 3021 a non-square in Fr: the unified law is complete, identity and doublings included.
#include <hip/hip_runtime.h>

#include <vector>

#include "context.h"
#include "ff.cuh"
#include "frinv.cuh"
#include "swmarlin.h"

struct swm_pedersen {
    void* d_table = nullptr;  // num_windows x 2^window_size rows (swm::EdRow)
    unsigned num_windows = 0, window_size = 0;
};

namespace swm {

struct EdExt {
    Fr x, y, t, z;
};
struct EdRow {  // an affine point as the mixed addition wants it; the identity is (1, 1, 0)
    Fr ymx, ypx, kt;
};
static constexpr uint64_t ED_D = 3021;

SWM_HD EdExt ed_identity() {
    EdExt p;
    p.x = fp_zero<Fr>();
    p.y = fp_one<Fr>();
    p.t = fp_zero<Fr>();
    p.z = fp_one<Fr>();
    return p;
}
// add-2008-hwcd-3 (a = -1), 8 multiplications + one by 2d
SWM_HD EdExt ed_add(const EdExt& p, const EdExt& q, const Fr& k2d) {
    Fr a = fp_mul(fp_sub(p.y, p.x), fp_sub(q.y, q.x));
    Fr b = fp_mul(fp_add(p.y, p.x), fp_add(q.y, q.x));
    Fr c = fp_mul(fp_mul(p.t, k2d), q.t);
    Fr d = fp_dbl(fp_mul(p.z, q.z));
    Fr e = fp_sub(b, a), f = fp_sub(d, c), g = fp_add(d, c), h = fp_add(b, a);
    EdExt r;
    r.x = fp_mul(e, f);
    r.y = fp_mul(g, h);
    r.t = fp_mul(e, h);
    r.z = fp_mul(f, g);
    return r;
}
// madd-2008-hwcd-3 against a tabulated affine point: 7 multiplications
SWM_HD void ed_madd(EdExt& p, const EdRow& q) {
    Fr a = fp_mul(fp_sub(p.y, p.x), q.ymx);
    Fr b = fp_mul(fp_add(p.y, p.x), q.ypx);
    Fr c = fp_mul(p.t, q.kt);
    Fr d = fp_dbl(p.z);
    Fr e = fp_sub(b, a), f = fp_sub(d, c), g = fp_add(d, c), h = fp_add(b, a);
    p.x = fp_mul(e, f);
    p.y = fp_mul(g, h);
    p.t = fp_mul(e, h);
    p.z = fp_mul(f, g);
}

__device__ __forceinline__ EdExt ed_shfl_xor(const EdExt& p, int mask) {
    EdExt r;
    const uint32_t* s = reinterpret_cast<const uint32_t*>(&p);
    uint32_t* d = reinterpret_cast<uint32_t*>(&r);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(EdExt) / 4); i++) d[i] = (uint32_t)__shfl_xor((int)s[i], mask, 64);
    return r;
}

// `count` hashes, `lanes` (a power of two <= 64) lanes each.  Input h = in[h * stride .. + len), digest h = 32 bytes at out[32 h].
__global__ void __launch_bounds__(256) pedersen_hash_kernel(const EdRow* __restrict__ table, unsigned num_windows, unsigned ws,
                                                            const uint8_t* __restrict__ in, size_t stride, size_t len, size_t count,
                                                            unsigned lanes, Fr k2d, uint8_t* __restrict__ out) {
    const size_t gid = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t h = gid / lanes;
    const unsigned lane = (unsigned)(gid % lanes);
    // (a wave never straddles the end raggedly: lanes divides 64, so the lanes of one hash are all in or all out)
    const bool live = h < count;
    EdExt acc = ed_identity();
    if (live) {
        const uint8_t* msg = in + h * stride;
        const size_t nbits = len * 8;
        const unsigned used = (unsigned)min((size_t)num_windows, (nbits + ws - 1) / ws);  // windows past the input are zero bits
        const unsigned mask = (1u << ws) - 1u;
#pragma unroll 1
        for (unsigned w = lane; w < used; w += lanes) {
            const size_t bit = (size_t)w * ws, byte = bit >> 3;
            unsigned v = msg[byte];
            if (byte + 1 < len) v |= (unsigned)msg[byte + 1] << 8;
            v = (v >> (bit & 7)) & mask;
            if (v) ed_madd(acc, table[((size_t)w << ws) + v]);
        }
    }
#pragma unroll 1
    for (unsigned s = lanes >> 1; s; s >>= 1) {
        EdExt other = ed_shfl_xor(acc, (int)s);
        acc = ed_add(acc, other, k2d);
    }
    if (live && lane == 0) {
        Fr x = fp_to_std(fp_mul(acc.x, fr_inv_single(acc.z)));  // Z != 0: the law is complete
        uint32_t* o = reinterpret_cast<uint32_t*>(out + 32 * h);  // device buffers of this library are 256-byte aligned
#pragma unroll
        for (int i = 0; i < 8; i++) o[i] = x.v[i];
    }
}

static unsigned lanes_for(size_t count) {
    // ~2^17 lanes keep the chip busy; fewer hashes than that are spread over more lanes each
    unsigned l = 1;
    while (l < 64 && (size_t)l * count < ((size_t)1 << 17)) l <<= 1;
    return l;
}

int pedersen_hash_run(swm_ctx* ctx, const swm_pedersen* p, const uint8_t* d_in, size_t stride, size_t len, size_t count,
                      uint8_t* d_out) {
    if (!count) return SWM_OK;
    if (len * 8 > (size_t)p->num_windows * p->window_size)
        return set_err(ctx, SWM_ERR_INVALID_ARG, "pedersen: %zu input bytes do not fit %u windows of %u bits", len, p->num_windows,
                       p->window_size);
    const unsigned lanes = lanes_for(count);
    const size_t threads = count * lanes;
    SWM_LAUNCH(ctx, "pedersen_hash", pedersen_hash_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
               reinterpret_cast<const EdRow*>(p->d_table), p->num_windows, p->window_size, d_in, stride, len, count, lanes,
               fp_from_u64<Fr>(2 * ED_D), d_out);
    return SWM_OK;
}

// nodes: n leaf digests | n / 2 | ... | root — (2 n - 1) x 32 bytes
int merkle_build_run(swm_ctx* ctx, const swm_pedersen* leaf, const swm_pedersen* inner, const uint8_t* d_leaves, size_t leaf_len,
                     size_t n, uint8_t* d_nodes) {
    SWM_TRY(pedersen_hash_run(ctx, leaf, d_leaves, leaf_len, leaf_len, n, d_nodes));
    size_t off = 0;
    for (size_t cnt = n; cnt > 1; cnt >>= 1) {
        SWM_TRY(pedersen_hash_run(ctx, inner, d_nodes + 32 * off, 64, 64, cnt >> 1, d_nodes + 32 * (off + cnt)));
        off += cnt;
    }
    return SWM_OK;
}

static bool fr_from_le_bytes(const uint8_t* b, Fr* out) {  // canonical (< r) or refused
    Fr s;
    for (int i = 0; i < 8; i++) s.v[i] = (uint32_t)b[4 * i] | (uint32_t)b[4 * i + 1] << 8 | (uint32_t)b[4 * i + 2] << 16 | (uint32_t)b[4 * i + 3] << 24;
    Fr r;
    for (int i = 0; i < 8; i++) r.v[i] = FrParams::P[i];
    if (fp_cmp_std(s, r) >= 0) return false;
    *out = fp_from_std(s);
    return true;
}
static bool ed_on_curve(const Fr& x, const Fr& y) {  // -x^2 + y^2 = 1 + d x^2 y^2
    Fr x2 = fp_sqr(x), y2 = fp_sqr(y);
    Fr lhs = fp_sub(y2, x2);
    Fr rhs = fp_add(fp_one<Fr>(), fp_mul(fp_from_u64<Fr>(ED_D), fp_mul(x2, y2)));
    return fp_eq(lhs, rhs);
}

}  // namespace swm

using namespace swm;

extern "C" {

int swm_pedersen_create(swm_ctx* ctx, const uint8_t* generators_xy, size_t num_windows, size_t window_size, swm_pedersen** out) {
    if (!ctx || !generators_xy || !out || !num_windows || window_size < 1 || window_size > 8 || num_windows > 4096)
        return set_err(ctx, SWM_ERR_INVALID_ARG, "pedersen_create: bad arguments");
    SWM_ON_DEVICE(ctx);
    const Fr k2d = fp_from_u64<Fr>(2 * ED_D);
    const size_t per = (size_t)1 << window_size;
    std::vector<EdRow> rows(num_windows * per);
    for (size_t w = 0; w < num_windows; w++) {
        // generators[w][j], j < window_size: on the curve, and each the double of the one before (pedersen::CRH::setup [U])
        EdExt g[8];
        for (size_t j = 0; j < window_size; j++) {
            const uint8_t* xy = generators_xy + 64 * (w * window_size + j);
            Fr x, y;
            if (!fr_from_le_bytes(xy, &x) || !fr_from_le_bytes(xy + 32, &y) || !ed_on_curve(x, y))
                return set_err(ctx, SWM_ERR_INVALID_ARG, "pedersen_create: generator [%zu][%zu] is not a point of ed-on-BLS12-377", w, j);
            g[j].x = x;
            g[j].y = y;
            g[j].t = fp_mul(x, y);
            g[j].z = fp_one<Fr>();
            if (j) {
                EdExt dbl = ed_add(g[j - 1], g[j - 1], k2d);  // compare projectively: X1 Z2 = X2 Z1, Y1 Z2 = Y2 Z1 (Z2 = 1)
                if (!fp_eq(dbl.x, fp_mul(x, dbl.z)) || !fp_eq(dbl.y, fp_mul(y, dbl.z)))
                    return set_err(ctx, SWM_ERR_INVALID_ARG, "pedersen_create: generator [%zu][%zu] is not twice its predecessor", w, j);
            }
        }
        EdExt acc = ed_identity();
        for (size_t v = 0; v < per; v++) {  // acc = v g_w
            Fr zi = fp_inv(acc.z);
            Fr x = fp_mul(acc.x, zi), y = fp_mul(acc.y, zi);
            EdRow& r = rows[w * per + v];
            r.ymx = fp_sub(y, x);
            r.ypx = fp_add(y, x);
            r.kt = fp_mul(k2d, fp_mul(x, y));
            acc = ed_add(acc, g[0], k2d);
        }
    }
    swm_pedersen* p = new swm_pedersen;
    p->num_windows = (unsigned)num_windows;
    p->window_size = (unsigned)window_size;
    hipError_t e = hipMalloc(&p->d_table, rows.size() * sizeof(EdRow));
    if (e == hipSuccess) e = hipMemcpyAsync(p->d_table, rows.data(), rows.size() * sizeof(EdRow), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // `rows` goes out of scope
    if (e != hipSuccess) {
        if (p->d_table) (void)hipFree(p->d_table);
        delete p;
        (void)hipGetLastError();
        return set_err(ctx, e == hipErrorOutOfMemory ? SWM_ERR_OOM : SWM_ERR_HIP, "pedersen_create: %s", hipGetErrorString(e));
    }
    *out = p;
    return SWM_OK;
}

void swm_pedersen_destroy(swm_ctx* ctx, swm_pedersen* p) {
    if (!p) return;
    DeviceGuard guard(ctx);
    if (ctx) drain_streams(ctx);
    if (p->d_table) (void)hipFree(p->d_table);
    delete p;
}

int swm_pedersen_hash_dev(swm_ctx* ctx, const swm_pedersen* p, const void* d_inputs, size_t input_len, size_t count, void* d_digests) {
    if (!ctx || !p || (count && (!d_inputs || !d_digests || !input_len)))
        return set_err(ctx, SWM_ERR_INVALID_ARG, "pedersen_hash: bad arguments");
    SWM_ON_DEVICE(ctx);
    return pedersen_hash_run(ctx, p, (const uint8_t*)d_inputs, input_len, input_len, count, (uint8_t*)d_digests);
}

int swm_pedersen_hash(swm_ctx* ctx, const swm_pedersen* p, const uint8_t* inputs, size_t input_len, size_t count, uint8_t* digests) {
    if (!ctx || !p || (count && (!inputs || !digests || !input_len)))
        return set_err(ctx, SWM_ERR_INVALID_ARG, "pedersen_hash: bad arguments");
    SWM_ON_DEVICE(ctx);
    if (!count) return SWM_OK;
    uint8_t *d_in = nullptr, *d_out = nullptr;
    SWM_TRY(scratch(ctx, "stage.a", count * input_len + 32, (void**)&d_in));
    SWM_TRY(scratch(ctx, "stage.b", count * 32, (void**)&d_out));
    SWM_HIP(ctx, hipMemcpyAsync(d_in, inputs, count * input_len, hipMemcpyHostToDevice, ctx->stream));
    SWM_TRY(pedersen_hash_run(ctx, p, d_in, input_len, input_len, count, d_out));
    SWM_HIP(ctx, hipMemcpyAsync(digests, d_out, count * 32, hipMemcpyDeviceToHost, ctx->stream));
    SWM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SWM_OK;
}

static int merkle_args(swm_ctx* ctx, const swm_pedersen* leaf, const swm_pedersen* inner, const void* leaves, size_t leaf_len, size_t n,
                       const void* nodes) {
    if (!ctx || !leaf || !inner || !leaves || !nodes || !leaf_len) return set_err(ctx, SWM_ERR_INVALID_ARG, "merkle_tree_build: bad arguments");
    if (n < 2 || (n & (n - 1)))  // ark-crypto-primitives MerkleTree::new: a power of two, at least one two-to-one level [U]
        return set_err(ctx, SWM_ERR_INVALID_ARG, "merkle_tree_build: %zu leaves (a power of two >= 2 is required)", n);
    if ((size_t)inner->num_windows * inner->window_size < 512)
        return set_err(ctx, SWM_ERR_INVALID_ARG, "merkle_tree_build: the two-to-one parameters hold fewer than 2 x 256 bits");
    return SWM_OK;
}

int swm_merkle_tree_build_dev(swm_ctx* ctx, const swm_pedersen* leaf, const swm_pedersen* inner, const void* d_leaves, size_t leaf_len,
                              size_t n_leaves, void* d_nodes) {
    SWM_TRY(merkle_args(ctx, leaf, inner, d_leaves, leaf_len, n_leaves, d_nodes));
    SWM_ON_DEVICE(ctx);
    return merkle_build_run(ctx, leaf, inner, (const uint8_t*)d_leaves, leaf_len, n_leaves, (uint8_t*)d_nodes);
}

int swm_merkle_tree_build(swm_ctx* ctx, const swm_pedersen* leaf, const swm_pedersen* inner, const uint8_t* leaves, size_t leaf_len,
                          size_t n_leaves, uint8_t* nodes) {
    SWM_TRY(merkle_args(ctx, leaf, inner, leaves, leaf_len, n_leaves, nodes));
    SWM_ON_DEVICE(ctx);
    uint8_t *d_in = nullptr, *d_nodes = nullptr;
    const size_t node_bytes = (2 * n_leaves - 1) * 32;
    SWM_TRY(scratch(ctx, "stage.a", n_leaves * leaf_len + 32, (void**)&d_in));
    SWM_TRY(scratch(ctx, "stage.b", node_bytes, (void**)&d_nodes));
    SWM_HIP(ctx, hipMemcpyAsync(d_in, leaves, n_leaves * leaf_len, hipMemcpyHostToDevice, ctx->stream));
    SWM_TRY(merkle_build_run(ctx, leaf, inner, d_in, leaf_len, n_leaves, d_nodes));
    SWM_HIP(ctx, hipMemcpyAsync(nodes, d_nodes, node_bytes, hipMemcpyDeviceToHost, ctx->stream));
    SWM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SWM_OK;
}

}  // extern "C"
