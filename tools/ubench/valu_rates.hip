// Micro-benchmark: issue rate of the integer/FP64 VALU instructions the Montgomery kernels are built from.
// Prints cycles per wave-instruction per SIMD (s_memtime based) and chip-wide Gops/s.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define REP16(x) x x x x x x x x x x x x x x x x
#define ITERS 4096

template <int OP>
__global__ void __launch_bounds__(256) k(uint64_t* out, uint32_t seed, unsigned long long* cyc) {
    uint32_t a = seed + threadIdx.x, b = seed * 3 + threadIdx.x;
    uint64_t acc0 = a, acc1 = b, acc2 = a + b, acc3 = a ^ b;
    uint32_t c0 = a, c1 = b, c2 = a + 1, c3 = b + 1;
    double d0 = a, d1 = b, d2 = 1.5, d3 = 2.5;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(); unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < ITERS; i++) {
        if (OP == 0) { REP16(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3" : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : "v"(a), "v"(b) : "vcc");) }
        if (OP == 1) { REP16(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a));) }
        if (OP == 2) { REP16(asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a));) }
        if (OP == 3) { REP16(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(d0), "v"(d1));) }
        if (OP == 4) { REP16(asm volatile("v_addc_co_u32 %0, vcc, %0, %4, vcc\n v_addc_co_u32 %1, vcc, %1, %4, vcc\n v_addc_co_u32 %2, vcc, %2, %4, vcc\n v_addc_co_u32 %3, vcc, %3, %4, vcc" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a) : "vcc");) }
        if (OP == 5) { REP16(asm volatile("v_lshl_add_u64 %0, %0, 0, %4\n v_lshl_add_u64 %1, %1, 0, %4\n v_lshl_add_u64 %2, %2, 0, %4\n v_lshl_add_u64 %3, %3, 0, %4" : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : "v"(acc0));) }
        if (OP == 6) { REP16(asm volatile("v_mad_u32_u24 %0, %0, %4, %5\n v_mad_u32_u24 %1, %1, %4, %5\n v_mad_u32_u24 %2, %2, %4, %5\n v_mad_u32_u24 %3, %3, %4, %5" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b));) }
        if (OP == 7) { REP16(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a));) }
        if (OP == 8) { REP16(asm volatile("v_mul_f64 %0, %0, %4\n v_mul_f64 %1, %1, %4\n v_mul_f64 %2, %2, %4\n v_mul_f64 %3, %3, %4" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"(d1));) }
        if (OP == 9) { REP16(asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0\n v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc0) : "v"(a), "v"(b) : "vcc");) }
        if (OP == 10) { REP16(asm volatile("v_mad_i32_i24 %0, %0, %4, %5\n v_mul_u32_u24 %1, %1, %4\n v_mul_hi_u32_u24 %2, %2, %4\n v_mad_u32_u24 %3, %3, %4, %5" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b));) }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(); unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    out[tid] = acc0 + acc1 + acc2 + acc3 + c0 + c1 + c2 + c3 + (uint64_t)(d0 + d1 + d2 + d3);
    if (threadIdx.x == 0) { cyc[2*blockIdx.x] = t1 - t0; cyc[2*blockIdx.x+1] = r1 - r0; }
}

template <int OP>
int run(const char* name, int waves_per_simd) {
    int blocks = 256 * waves_per_simd;  // 256-thread blocks = 4 waves = 1 wave per SIMD per block
    uint64_t* out; unsigned long long* cyc;
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 8));
    CHECK(hipMalloc(&cyc, blocks * 16));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 20; w++) k<OP><<<blocks, 256>>>(out, 12345, cyc);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    k<OP><<<blocks, 256>>>(out, 12345, cyc);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(blocks * 2);
    CHECK(hipMemcpy(h.data(), cyc, blocks * 16, hipMemcpyDeviceToHost));
    double avg = 0, avgr = 0; for (int i = 0; i < blocks; i++) { avg += h[2*i]; avgr += h[2*i+1]; } avg /= blocks; avgr /= blocks;
    double clk_ghz = avg / avgr * 0.1;
    double ninstr = (double)ITERS * 16 * 4;
    double total_ops = ninstr * 64.0 * 4 * blocks;  // lane-ops
    // s_memtime ticks at 100MHz constant clock on some parts; report both
    printf("%-22s waves/SIMD=%d clk=%.2fGHz  cycles/instr/SIMD (in-kernel)=%.2f  wall=%.3f ms  lane-Gops/s=%.0f  cycles/instr/SIMD (wall@clk)=%.2f\n",
           name, waves_per_simd, clk_ghz, avg / ninstr / waves_per_simd, ms, total_ops / ms / 1e6, (ms * 1e-3 * clk_ghz * 1e9) / (ninstr * waves_per_simd));
    hipFree(out); hipFree(cyc);
    return 0;
}

int main() {
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_mad_u64_u32", w); run<9>("v_mad_u64_u32 (dep)", w); run<1>("v_mul_lo_u32", w); run<2>("v_mul_hi_u32", w);
        run<3>("v_fma_f64", w); run<8>("v_mul_f64", w); run<4>("v_addc_co_u32", w); run<5>("v_lshl_add_u64", w);
        run<6>("v_mad_u32_u24", w); run<10>("24-bit mix", w); run<7>("v_add_u32", w);
    }
    return 0;
}
