// pool.h — a handful of host worker threads for the serial-latency tails of the GPU path (today: the per-window fold
// of an MSM's block results, which sits between the last kernel of a round and the next Fiat-Shamir challenge).
// Not a general scheduler: one parallel_for at a time, indices handed out by an atomic counter, the caller works too.
#pragma once
#include <atomic>
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
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : th_) t.join();
    }
    HostPool(const HostPool&) = delete;
    HostPool& operator=(const HostPool&) = delete;

    // runs fn(0) .. fn(n-1), returns when all have finished
    void parallel_for(int n, const std::function<void(int)>& fn) {
        if (n <= 0) return;
        if (th_.empty() || n == 1) {
            for (int i = 0; i < n; i++) fn(i);
            return;
        }
        {
            std::lock_guard<std::mutex> g(m_);
            fn_ = &fn;
            n_ = n;
            next_.store(0, std::memory_order_relaxed);
            left_.store(n, std::memory_order_relaxed);
            gen_++;
        }
        cv_.notify_all();
        drain();
        std::unique_lock<std::mutex> g(m_);
        done_.wait(g, [this] { return left_.load(std::memory_order_acquire) == 0 && busy_ == 0; });
        fn_ = nullptr;
    }

private:
    void drain() {
        for (;;) {
            int i = next_.fetch_add(1, std::memory_order_relaxed);
            if (i >= n_) break;
            (*fn_)(i);
            left_.fetch_sub(1, std::memory_order_release);
        }
    }
    void worker() {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> g(m_);
        for (;;) {
            cv_.wait(g, [&] { return stop_ || gen_ != seen; });
            if (stop_) return;
            seen = gen_;
            busy_++;
            g.unlock();
            drain();
            g.lock();
            busy_--;
            if (busy_ == 0 && left_.load(std::memory_order_acquire) == 0) done_.notify_all();
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void(int)>* fn_ = nullptr;
    int n_ = 0;
    std::atomic<int> next_{0}, left_{0};
    int busy_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false;
};

}  // namespace swm
