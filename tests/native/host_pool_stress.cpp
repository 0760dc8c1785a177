// Stress test of simpleworks_amd/csrc/host/pool.h (the host workers that fold a round's MSM results): every task of every job runs
// exactly once whether the workers are asleep, polling (arm) or arriving late; compiled and run by tests/test_host_pool.py, also
// under -fsanitize=thread.
#include "host/pool.h"
#include <cstdio>
#include <atomic>
#include <cstdint>
int main() {
    swm::HostPool pool(7);
    std::atomic<long> sum{0};
    long expect = 0;
    for (int it = 0; it < 20000; it++) {
        int n = 2 + (it * 7) % 61;
        if (it % 3 == 0) pool.arm(50);
        pool.parallel_for(n, [&](int i) { sum.fetch_add(i + 1); });
        expect += (long)n * (n + 1) / 2;
        if (it % 1000 == 0) std::this_thread::sleep_for(std::chrono::microseconds(300));
    }
    printf("%s %ld %ld\n", sum.load() == expect ? "OK" : "MISMATCH", sum.load(), expect);
    auto t0 = std::chrono::steady_clock::now();
    for (int it = 0; it < 2000; it++) { pool.arm(100); pool.parallel_for(16, [&](int i) { sum.fetch_add(i); }); }
    auto t1 = std::chrono::steady_clock::now();
    printf("armed parallel_for(16): %.2f us\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / 2000);
    t0 = std::chrono::steady_clock::now();
    for (int it = 0; it < 2000; it++) { std::this_thread::sleep_for(std::chrono::microseconds(200)); pool.parallel_for(16, [&](int i) { sum.fetch_add(i); }); }
    t1 = std::chrono::steady_clock::now();
    printf("cold parallel_for(16) incl 200us sleep: %.2f us\n", std::chrono::duration<double, std::micro>(t1 - t0).count() / 2000);
    return sum.load() == 0;
}
