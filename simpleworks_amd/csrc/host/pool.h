// pool.h — a handful of host worker threads for the serial-latency tails of the GPU path (today: the per-window fold
// of an MSM's block results, which sits between the last kernel of a round and the next Fiat-Shamir challenge).
// Not a general scheduler: one parallel_for at a time, indices handed out by an atomic counter, the caller works too.
//
// The tasks are microseconds long, so what counts is how fast a job reaches the workers (r04: with a mutex-protected job
// pointer every worker took the lock once per job, and 15 threads queueing on one futex were ~70 of the ~110 us a small
// proof's round spent folding):
//   * a job is ONE 64-bit word, generation | task count | next index: a worker's fetch_add hands it all three at once, so
//     claiming a task takes no lock, and a worker that arrives late can only ever claim a task of the job that is current
//     (the callable is read only after a task of that job has been claimed, i.e. while the job cannot complete);
//   * arm(): the caller is about to wait for the GPU and will publish a job right after: the workers poll the word instead
//     of sleeping, for a bounded time.
// Workers that are asleep are woken through the condition variable as before.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace swm {

class HostPool {
public:
    explicit HostPool(unsigned workers) {
        for (unsigned i = 0; i < workers; i++) th_.emplace_back([this] { worker(); });
    }
    ~HostPool() {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_.store(true, std::memory_order_release);
        }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    HostPool(const HostPool&) = delete;
    HostPool& operator=(const HostPool&) = delete;

    // The workers poll for the next job for at most `us` microseconds instead of sleeping (costs that much spinning per
    // worker when no job follows).
    void arm(unsigned us) {
        if (th_.empty()) return;
        spin_until_.store(now_ns() + (int64_t)us * 1000, std::memory_order_seq_cst);  // (store, then load: see worker())
        if (sleepers_.load(std::memory_order_seq_cst) == 0) return;
        {
            std::lock_guard<std::mutex> g(m_);
        }
        cv_.notify_all();
    }

    // runs fn(0) .. fn(n-1), returns when all have finished
    void parallel_for(int n, const std::function<void(int)>& fn) {
        if (n <= 0) return;
        if (th_.empty() || n == 1 || n >= (1 << IDX_BITS)) {
            for (int i = 0; i < n; i++) fn(i);
            return;
        }
        fn_.store(&fn, std::memory_order_relaxed);
        left_.store(n, std::memory_order_relaxed);
        gen_ = (gen_ + 1) & GEN_MASK;
        word_.store((gen_ << (2 * IDX_BITS)) | ((uint64_t)n << IDX_BITS), std::memory_order_seq_cst);  // publishes fn_, left_
        if (sleepers_.load(std::memory_order_seq_cst) != 0) {
            {
                std::lock_guard<std::mutex> g(m_);
            }
            cv_.notify_all();
        }
        drain();
        for (int spin = 0; spin < 40000 && left_.load(std::memory_order_acquire) != 0; spin++) cpu_relax();
        if (left_.load(std::memory_order_acquire) != 0) {
            std::unique_lock<std::mutex> g(m_);
            waiting_.store(true, std::memory_order_seq_cst);  // (store, then load of left_; the last task: the other way round)
            done_.wait(g, [&] { return left_.load(std::memory_order_seq_cst) == 0; });
            waiting_.store(false, std::memory_order_release);
        }
        // every task has run: nobody reads fn_ any more (a claim needs an index below n, and those are used up)
    }

private:
    static constexpr int IDX_BITS = 22;
    static constexpr uint64_t IDX_MASK = ((uint64_t)1 << IDX_BITS) - 1, GEN_MASK = ((uint64_t)1 << (64 - 2 * IDX_BITS)) - 1;
    void drain() {
        for (;;) {
            const uint64_t v = word_.fetch_add(1, std::memory_order_acq_rel);
            const uint64_t i = v & IDX_MASK, n = (v >> IDX_BITS) & IDX_MASK;
            if (i >= n) break;  // (the index field cannot reach its 4 M limit: a handful of failed claims per worker and job)
            (*fn_.load(std::memory_order_relaxed))((int)i);  // task i of the job the word held: current until it is counted below
            if (left_.fetch_sub(1, std::memory_order_seq_cst) == 1 && waiting_.load(std::memory_order_seq_cst)) {
                std::lock_guard<std::mutex> g(m_);  // pairs with the waiter's predicate check
                done_.notify_all();
            }
        }
    }
    bool has_work() const {
        const uint64_t v = word_.load(std::memory_order_seq_cst);
        return (v & IDX_MASK) < ((v >> IDX_BITS) & IDX_MASK);
    }
    void worker() {
        for (;;) {
            if (stop_.load(std::memory_order_acquire)) return;
            if (has_work()) {
                drain();
                continue;
            }
            if (now_ns() < spin_until_.load(std::memory_order_acquire)) {  // armed: poll
                for (unsigned i = 1; !has_work(); i++) {
                    cpu_relax();
                    if ((i & 63u) == 0 && (now_ns() >= spin_until_.load(std::memory_order_acquire) || stop_.load(std::memory_order_acquire)))
                        break;
                }
                continue;
            }
            std::unique_lock<std::mutex> g(m_);
            sleepers_.fetch_add(1, std::memory_order_seq_cst);
            // (a job or an arm() published between the checks above and here is seen by the predicate: the publisher reads
            // sleepers_ after its store, we read its store after our increment — sequentially consistent on both sides)
            cv_.wait(g, [&] { return stop_.load(std::memory_order_acquire) || has_work() || now_ns() < spin_until_.load(std::memory_order_seq_cst); });
            sleepers_.fetch_sub(1, std::memory_order_acq_rel);
        }
    }
    static int64_t now_ns() {
        return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
    }
    static void cpu_relax() {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    std::atomic<uint64_t> word_{0};  // generation | task count | next index
    std::atomic<const std::function<void(int)>*> fn_{nullptr};
    std::atomic<int> left_{0};
    std::atomic<int> sleepers_{0};
    std::atomic<bool> waiting_{false}, stop_{false};
    std::atomic<int64_t> spin_until_{0};  // workers poll instead of sleeping until this time (arm)
    uint64_t gen_ = 0;                    // caller side only
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_, done_;
};

}  // namespace swm
