// gd_async_io.hpp -- the output side of a batched driver, off the stepping thread.
//
// The reference writes one trajectory file per process and its CPU step is slow enough that deflate never shows.  A batched
// driver advances R replicas at tens of microseconds per step and owes R files a snapshot every sampling interval and a
// contact-map dump every thinning interval: on one thread the filter pipeline of the HDF5 library (shuffle + deflate 6, the
// reference's dataset properties, simulation_store.cc:318-345) costs more than the stepping in between.  So:
//   thread_pool    packs chunks (gd_h5util.hpp: pack_chunk -- no HDF5 call inside) on every core the process may use;
//   async_writer   one thread that runs queued output jobs in order -- the ONLY thread that touches the HDF5 library while jobs are
//                  pending (the library is not re-entrant); the stepping thread hands over copies of its buffers and goes on.
// drain() is the fence: before the stepping thread reads a file or changes a store's phase, and at the end of the run.
#pragma once
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <exception>
#include <cstdlib>
#include <fstream>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace gd {

class thread_pool {
public:
    explicit thread_pool(unsigned threads)
    {
        for (unsigned k = 1; k < threads; k++) _workers.emplace_back([this] { loop(); });      // (the caller of parallel_for is one of them)
    }
    ~thread_pool()
    {
        { std::lock_guard<std::mutex> lk(_m); _stop = true; }
        _cv_work.notify_all();
        for (auto &t : _workers) t.join();
    }
    thread_pool(thread_pool const &) = delete;
    thread_pool &operator=(thread_pool const &) = delete;

    // fn(0) ... fn(n - 1), each once, on the pool's threads and the caller; returns when all are done; rethrows the first exception
    void parallel_for(std::size_t n, std::function<void(std::size_t)> const &fn)
    {
        if (n == 0) return;
        {
            std::lock_guard<std::mutex> lk(_m);
            _fn = &fn; _n = n; _next = 0; _active = (unsigned)_workers.size(); _error = nullptr; _generation++;
        }
        _cv_work.notify_all();
        work();
        std::unique_lock<std::mutex> lk(_m);
        _cv_done.wait(lk, [&] { return _active == 0; });
        if (_error) std::rethrow_exception(_error);
    }

private:
    void work()
    {
        for (;;) {
            std::size_t const i = _next.fetch_add(1);
            if (i >= _n) return;
            try { (*_fn)(i); }
            catch (...) { std::lock_guard<std::mutex> lk(_m); if (!_error) _error = std::current_exception(); }
        }
    }
    void loop()
    {
        unsigned long seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lk(_m);
            _cv_work.wait(lk, [&] { return _stop || _generation != seen; });
            if (_stop) return;
            seen = _generation;
            lk.unlock();
            work();
            lk.lock();
            if (--_active == 0) _cv_done.notify_all();
        }
    }
    std::vector<std::thread> _workers;
    std::mutex _m;
    std::condition_variable _cv_work, _cv_done;
    std::function<void(std::size_t)> const *_fn = nullptr;
    std::size_t _n = 0;
    std::atomic<std::size_t> _next{0};
    unsigned _active = 0;
    unsigned long _generation = 0;
    bool _stop = false;
    std::exception_ptr _error;
};

class async_writer {
public:
    explicit async_writer(std::size_t max_pending = 2) : _max_pending(max_pending), _thread([this] { loop(); }) {}
    ~async_writer()
    {
        { std::lock_guard<std::mutex> lk(_m); _stop = true; }
        _cv_job.notify_all();
        _thread.join();
    }
    async_writer(async_writer const &) = delete;
    async_writer &operator=(async_writer const &) = delete;

    // queues a job behind the ones already there; waits while max_pending jobs are queued; rethrows what an earlier job threw
    void submit(std::function<void()> job)
    {
        std::unique_lock<std::mutex> lk(_m);
        _cv_room.wait(lk, [&] { return _error || _jobs.size() < _max_pending; });
        rethrow(lk);
        _jobs.push_back(std::move(job));
        _cv_job.notify_one();
    }
    // returns when every queued job has run; rethrows what a job threw
    void drain()
    {
        std::unique_lock<std::mutex> lk(_m);
        _cv_room.wait(lk, [&] { return _error || (_jobs.empty() && !_busy); });
        rethrow(lk);
    }

private:
    void rethrow(std::unique_lock<std::mutex> &)
    {
        if (!_error) return;
        std::exception_ptr e = _error;
        _error = nullptr; _jobs.clear();
        std::rethrow_exception(e);
    }
    void loop()
    {
        for (;;) {
            std::function<void()> job;
            {
                std::unique_lock<std::mutex> lk(_m);
                _cv_job.wait(lk, [&] { return _stop || !_jobs.empty(); });
                if (_jobs.empty()) return;             // (_stop: the queue has been worked off)
                job = std::move(_jobs.front());
                _jobs.pop_front();
                _busy = true;
            }
            std::exception_ptr err;
            try { job(); } catch (...) { err = std::current_exception(); }
            {
                std::lock_guard<std::mutex> lk(_m);
                _busy = false;
                if (err && !_error) _error = err;
            }
            _cv_room.notify_all();
        }
    }
    std::size_t _max_pending;
    std::mutex _m;
    std::condition_variable _cv_job, _cv_room;
    std::deque<std::function<void()>> _jobs;
    bool _busy = false, _stop = false;
    std::exception_ptr _error;
    std::thread _thread;      // (last: started when everything above exists)
};

// CPUs this process may use: the smaller of its affinity mask and its cgroup's CPU quota (a farm shares the host; a container's
// quota is usually far below the cores it can see, and threads beyond it only get the whole group throttled -- the stepping
// thread included, whose stream synchronisations then stall for tens of milliseconds).
inline unsigned usable_cpus()
{
    unsigned n = std::max(1u, std::thread::hardware_concurrency());
#ifdef __linux__
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) n = std::max(1, CPU_COUNT(&set));
    auto quota = [](char const *quota_file, char const *period_file) -> double {
        std::ifstream q(quota_file);
        std::string a, b;
        if (!(q >> a)) return 0;
        if (!period_file) { if (!(q >> b)) return 0; }                     // cgroup v2: "<quota|max> <period>"
        else { std::ifstream p(period_file); if (!(p >> b)) return 0; }    // cgroup v1: two files
        if (a == "max" || a == "-1") return 0;
        double const period = std::atof(b.c_str());
        return period > 0 ? std::atof(a.c_str()) / period : 0;
    };
    double c = quota("/sys/fs/cgroup/cpu.max", nullptr);
    if (!(c > 0)) c = quota("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us");
    if (c > 0) n = std::min(n, std::max(1u, (unsigned)(c + 0.5)));
#endif
    return n;
}

// threads for the packing pool: the usable CPUs less two (the stepping thread and the writer), at most `at_most`
inline unsigned usable_threads(unsigned at_most = 16)
{
    unsigned const n = usable_cpus();
    return std::max(1u, std::min(n > 2 ? n - 2 : 1u, at_most));
}

}  // namespace gd
