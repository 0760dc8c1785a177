// Semantics check of global_load_lds_dwordx4 on gfx950 with PER-LANE (gathered) global addresses:
// lane L of a wave writes its 16 bytes to LDS at (M0 base) + 16 * L.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void __launch_bounds__(256) k(const uint4* __restrict__ src, const unsigned* __restrict__ idx, uint4* __restrict__ dst) {
    extern __shared__ uint4 lds[];
    int t = threadIdx.x, wave = t >> 6;
    unsigned i = idx[blockIdx.x * 256 + t];
    for (int piece = 0; piece < 6; piece++) {
        const uint4* p = src + (size_t)i * 6 + piece;
        __builtin_amdgcn_global_load_lds((const void*)p, (__attribute__((address_space(3))) void*)(lds + piece * 256 + wave * 64), 16, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int piece = 0; piece < 6; piece++) dst[(size_t)(blockIdx.x * 256 + t) * 6 + piece] = lds[piece * 256 + t];
}
int main() {
    const int n = 1 << 16, blocks = 64;
    std::vector<uint4> h(n * 6);
    for (int i = 0; i < n * 6; i++) h[i] = make_uint4(i, i * 3 + 1, ~i, i ^ 0x5555);
    std::vector<unsigned> idx(blocks * 256);
    for (size_t i = 0; i < idx.size(); i++) idx[i] = (unsigned)((i * 2654435761u) % n);
    uint4 *ds, *dd; unsigned* di;
    CHECK(hipMalloc(&ds, h.size() * 16)); CHECK(hipMalloc(&dd, idx.size() * 6 * 16)); CHECK(hipMalloc(&di, idx.size() * 4));
    CHECK(hipMemcpy(ds, h.data(), h.size() * 16, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(di, idx.data(), idx.size() * 4, hipMemcpyHostToDevice));
    k<<<blocks, 256, 6 * 256 * 16>>>(ds, di, dd);
    CHECK(hipDeviceSynchronize());
    std::vector<uint4> out(idx.size() * 6);
    CHECK(hipMemcpy(out.data(), dd, out.size() * 16, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t t = 0; t < idx.size(); t++)
        for (int p = 0; p < 6; p++) {
            uint4 a = out[t * 6 + p], b = h[(size_t)idx[t] * 6 + p];
            if (a.x != b.x || a.y != b.y || a.z != b.z || a.w != b.w) bad++;
        }
    printf("global_load_lds_dwordx4 gather: %zu mismatches of %zu\n", bad, out.size());
    return bad != 0;
}
