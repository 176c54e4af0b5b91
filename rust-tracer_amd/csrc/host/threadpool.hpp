// threadpool.hpp -- stand-in for the `threadpool` crate (1.3.2; ThreadPool::new(n), execute(FnOnce + Send)) and
// for std::sync::mpsc::sync_channel: no arithmetic lives in either, behaviour only.
#pragma once
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace rtrace {

class ThreadPool {
public:
    explicit ThreadPool(size_t n)
    {
        if (n == 0) n = 1;
        for (size_t i = 0; i < n; ++i) workers_.emplace_back([this] { run(); });
    }
    ~ThreadPool()
    {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
        cv_.notify_all();
        for (auto &t : workers_) t.join();
    }
    void execute(std::function<void()> job)
    {
        { std::lock_guard<std::mutex> lk(mu_); jobs_.push_back(std::move(job)); }
        cv_.notify_one();
    }
    size_t size() const { return workers_.size(); }

private:
    void run()
    {
        for (;;) {
            std::function<void()> job;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [this] { return stop_ || !jobs_.empty(); });
                if (jobs_.empty()) return;
                job = std::move(jobs_.front());
                jobs_.pop_front();
            }
            job();
        }
    }
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<std::function<void()>> jobs_;
    std::vector<std::thread> workers_;
    bool stop_ = false;
};

// sync_channel::<T>(bound): send blocks while `bound` items are queued (back-pressure, render.rs:271).
template <typename T> class SyncChannel {
public:
    explicit SyncChannel(size_t bound) : bound_(bound) {}
    void send(T v)
    {
        std::unique_lock<std::mutex> lk(mu_);
        not_full_.wait(lk, [this] { return q_.size() < bound_; });
        q_.push_back(std::move(v));
        not_empty_.notify_one();
    }
    T recv()
    {
        std::unique_lock<std::mutex> lk(mu_);
        not_empty_.wait(lk, [this] { return !q_.empty(); });
        T v = std::move(q_.front());
        q_.pop_front();
        not_full_.notify_one();
        return v;
    }

private:
    size_t bound_;
    std::mutex mu_;
    std::condition_variable not_full_, not_empty_;
    std::deque<T> q_;
};

}  // namespace rtrace
