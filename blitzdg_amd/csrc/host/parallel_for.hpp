// Element-parallel loops of the set-up path without an OpenMP runtime.
//
// The library used to be built with g++ -fopenmp and linked by hipcc against LLVM's libomp.so (its GOMP
// compatibility entry points), beside the libgomp copies other modules of a caller's process bring (the test
// suite's CPU checker, PyTorch): several OpenMP runtimes with their idle worker pools and fork handlers next to
// the ROCr threads. The set-up loops are a handful of statically scheduled element loops, so they now run on
// plain std::thread workers that exist only for the duration of the call: nothing of this library is
// running, sleeping or registered when a caller forks.
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdlib>
#include <exception>
#include <mutex>
#include <system_error>
#include <thread>
#include <vector>

namespace blitzdg {
namespace detail {

// BDG_NUM_THREADS, then OMP_NUM_THREADS (callers that sized the old build keep their setting), then the
// hardware's count; at most 64.
inline unsigned maxWorkers() {
    for (const char* name : {"BDG_NUM_THREADS", "OMP_NUM_THREADS"}) {
        if (const char* v = std::getenv(name)) {
            const long n = std::strtol(v, nullptr, 10);
            if (n >= 1) return static_cast<unsigned>(std::min<long>(n, 64));
        }
    }
    const unsigned hw = std::thread::hardware_concurrency();
    return std::max(1u, std::min(hw, 64u));
}

// body(begin, end) on contiguous chunks of [0, n) (the static schedule), one chunk per worker; a loop shorter than
// `grain` iterations per worker uses fewer workers, down to the calling thread alone. The first exception thrown
// by a chunk is rethrown in the caller after all workers have been joined.
template <class Index, class Body>
void parallelChunks(Index n, Body&& body, std::size_t grain = 256) {
    if (n <= 0) return;
    const std::size_t total = static_cast<std::size_t>(n);
    const std::size_t want = std::max<std::size_t>(1, total / std::max<std::size_t>(1, grain));
    const unsigned workers = static_cast<unsigned>(std::min<std::size_t>(maxWorkers(), want));
    if (workers <= 1) {
        body(static_cast<Index>(0), n);
        return;
    }
    std::exception_ptr failure;
    std::mutex failureLock;
    auto run = [&](unsigned t) {
        const Index begin = static_cast<Index>(total * t / workers), end = static_cast<Index>(total * (t + 1) / workers);
        try {
            if (begin < end) body(begin, end);
        } catch (...) {
            std::lock_guard<std::mutex> hold(failureLock);
            if (!failure) failure = std::current_exception();
        }
    };
    std::vector<std::thread> pool;
    pool.reserve(workers - 1);
    unsigned started = 1;
    try {
        for (; started < workers; ++started) pool.emplace_back(run, started);
    } catch (const std::system_error&) {
        // no more threads to be had: the calling thread takes the chunks that are left
    }
    run(0);
    for (unsigned t = started; t < workers; ++t) run(t);
    for (auto& th : pool) th.join();
    if (failure) std::rethrow_exception(failure);
}

// body(i) for every i of [0, n).
template <class Index, class Body>
void parallelFor(Index n, Body&& body, std::size_t grain = 256) {
    parallelChunks(n, [&](Index begin, Index end) { for (Index i = begin; i < end; ++i) body(i); }, grain);
}

} // namespace detail
} // namespace blitzdg
