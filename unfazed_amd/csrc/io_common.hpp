// io_common.hpp -- shared pieces of the native input decoders (host only).
#pragma once
#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>
#include <zlib.h>

#include "unfazed_io.h"

namespace uzio {

extern thread_local std::string last_error;

struct IoError {
    int code;
    std::string msg;
};

[[noreturn]] inline void fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    throw IoError{code, buf};
}

inline double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

inline int resolve_threads(int threads) {
    if (threads <= 0) threads = (int)std::thread::hardware_concurrency();
    if (threads <= 0) threads = 1;
    return threads > 256 ? 256 : threads;
}

// workers worth starting for n items when one worker should get at least `grain` of them
inline int workers_for(int64_t n, int threads, int64_t grain) {
    const int64_t by_work = (n + grain - 1) / grain;
    return (int)std::max<int64_t>(1, std::min<int64_t>(threads, by_work));
}

// fn(lo, hi, worker) over [0, n) cut into one contiguous slice per worker; the first exception wins
template <typename F>
void parallel_slices(int64_t n, int threads, F fn) {
    if (n <= 0) return;
    int w = (int)std::min<int64_t>(threads, n);
    if (w <= 1) { fn((int64_t)0, n, 0); return; }
    std::vector<std::thread> pool;
    std::vector<IoError> errs(w, IoError{0, ""});
    for (int k = 0; k < w; k++) {
        const int64_t lo = n * k / w, hi = n * (k + 1) / w;
        pool.emplace_back([&, lo, hi, k] {
            try { fn(lo, hi, k); } catch (const IoError &e) { errs[k] = e; } catch (const std::exception &e) { errs[k] = IoError{UZ_IO_E_FORMAT, e.what()}; }
        });
    }
    for (auto &t : pool) t.join();
    for (auto &e : errs) if (e.code) throw e;
}

inline uint16_t rd16(const uint8_t *p) { uint16_t v; memcpy(&v, p, 2); return v; }
inline uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
inline int32_t rdi32(const uint8_t *p) { int32_t v; memcpy(&v, p, 4); return v; }

// malloc-backed byte buffer: no zero fill of memory that is about to be overwritten
struct Bytes {
    uint8_t *p = nullptr;
    size_t n = 0;
    Bytes() = default;
    explicit Bytes(size_t count) { alloc(count); }
    Bytes(const Bytes &) = delete;
    Bytes &operator=(const Bytes &) = delete;
    Bytes(Bytes &&o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    Bytes &operator=(Bytes &&o) noexcept { if (this != &o) { free(p); p = o.p; n = o.n; o.p = nullptr; o.n = 0; } return *this; }
    ~Bytes() { free(p); }
    void alloc(size_t count) {
        free(p);
        p = (uint8_t *)malloc(count ? count : 1);
        n = count;
        if (!p) fail(UZ_IO_E_RANGE, "out of memory for %zu bytes", count);
    }
    void release() { free(p); p = nullptr; n = 0; }
    size_t size() const { return n; }
    const uint8_t *data() const { return p; }
    uint8_t *data() { return p; }
    const uint8_t &operator[](size_t i) const { return p[i]; }
};

// whole file -> memory
Bytes read_file(const char *path);
// gzip / BGZF stream -> bytes.  BGZF blocks (BC extra field) are inflated in parallel, any other
// gzip stream sequentially, member after member.  Not gzip at all -> the input itself is handed
// back (moved) with *was_gzip = false.
Bytes inflate_all(Bytes &file, int threads, bool *was_gzip);

} // namespace uzio
