// pool.h — a handful of host worker threads for the serial-latency tails of the GPU path (today: the per-window fold
// of an MSM's block results, which sits between the last kernel of a round and the next Fiat-Shamir challenge).
// Not a general scheduler: one parallel_for at a time, indices handed out by an atomic counter, the caller works too.
#pragma once
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
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

    // runs fn(0) .. fn(n-1), returns when all have finished.  Each call publishes its own Job object; a worker only
    // ever touches the Job it copied (under the lock) when it woke up, so a worker that wakes late for an earlier
    // call finds that call's counter exhausted and cannot claim an index of the next one.
    void parallel_for(int n, const std::function<void(int)>& fn) {
        if (n <= 0) return;
        if (th_.empty() || n == 1) {
            for (int i = 0; i < n; i++) fn(i);
            return;
        }
        auto job = std::make_shared<Job>();
        job->fn = &fn;
        job->n = n;
        job->left.store(n, std::memory_order_relaxed);
        {
            std::lock_guard<std::mutex> g(m_);
            job_ = job;
            gen_++;
        }
        cv_.notify_all();
        drain(*job);
        std::unique_lock<std::mutex> g(m_);
        done_.wait(g, [&] { return job->left.load(std::memory_order_acquire) == 0; });
        job_.reset();  // fn dies with the caller's frame: nobody may start on it any more (late wakers see a null job)
    }

private:
    struct Job {
        const std::function<void(int)>* fn = nullptr;
        int n = 0;
        std::atomic<int> next{0}, left{0};
    };
    void drain(Job& j) {
        for (;;) {
            int i = j.next.fetch_add(1, std::memory_order_relaxed);
            if (i >= j.n) break;
            (*j.fn)(i);
            if (j.left.fetch_sub(1, std::memory_order_acq_rel) == 1) {
                std::lock_guard<std::mutex> g(m_);  // pairs with the waiter's predicate check
                done_.notify_all();
            }
        }
    }
    void worker() {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> g(m_);
        for (;;) {
            cv_.wait(g, [&] { return stop_ || gen_ != seen; });
            if (stop_) return;
            seen = gen_;
            std::shared_ptr<Job> job = job_;  // snapshot under the lock
            if (!job) continue;
            g.unlock();
            drain(*job);
            g.lock();
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    std::shared_ptr<Job> job_;
    uint64_t gen_ = 0;
    bool stop_ = false;
};

}  // namespace swm
