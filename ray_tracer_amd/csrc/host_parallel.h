// host_parallel.h -- the few lines of threading the host code shares.  A thread that the system refuses to start (a
// process or container at its thread limit) is not an error: its work runs on the calling thread instead.
#ifndef RT_HOST_PARALLEL_H
#define RT_HOST_PARALLEL_H

#include <atomic>
#include <future>
#include <system_error>
#include <utility>
#include <vector>

namespace rt {

// Starts fn on a thread of its own; false, and nothing started, when no thread could be had.
template <class R, class Fn>
bool try_async(std::future<R>& out, Fn&& fn) {
    try {
        out = std::async(std::launch::async, std::forward<Fn>(fn));
        return true;
    } catch (const std::system_error&) {
        return false;
    }
}

// fn(p) for every p in [0, parts), on the calling thread and up to `helpers` more; each takes the next part not yet begun.
template <class Fn>
void for_parts(int parts, int helpers, Fn fn) {
    std::atomic<int> next{0};
    auto work = [&]() { for (int p = next.fetch_add(1); p < parts; p = next.fetch_add(1)) fn(p); };
    std::vector<std::future<void>> jobs;
    for (int h = 0; h < helpers && h + 1 < parts; ++h) {
        std::future<void> f;
        if (!try_async(f, work)) break;
        jobs.push_back(std::move(f));
    }
    work();
    for (auto& j : jobs) j.get();
}

} // namespace rt
#endif
