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
#include <dlfcn.h>
#include <sched.h>
#include <pthread.h>
#include <unistd.h>
#include <new>
#include <atomic>
#include <condition_variable>
#include <mutex>

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

// CPUs' worth of run time the container's cgroup grants per period (cpu.max / cfs_quota), 0 = no limit.  A pod that shows 256
// processors can be held to 16 of them: threads beyond about twice the quota only get throttled (measured on the MI355X boxes:
// cpu.max = 1600000 100000, a spin loop scales to 16 threads and not one further).
inline int cgroup_cpu_quota() {
    static const int q = [] {
        long long quota = -1, period = 100000;
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char a[64] = {0};
            if (fscanf(f, "%63s %lld", a, &period) >= 1 && strcmp(a, "max") != 0) quota = atoll(a);
            fclose(f);
        } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
            if (fscanf(g, "%lld", &quota) != 1) quota = -1;
            fclose(g);
            if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(h, "%lld", &period) != 1) period = 100000; fclose(h); }
        }
        if (quota <= 0 || period <= 0) return 0;
        return (int)std::max<long long>(1, (quota + period - 1) / period);
    }();
    return q;
}

inline int resolve_threads(int threads) {
    if (threads <= 0) { // the CPUs this process may run on (an affinity mask narrower than the machine counts), not the machine's
        cpu_set_t set;
        CPU_ZERO(&set);
        if (sched_getaffinity(0, sizeof(set), &set) == 0) threads = CPU_COUNT(&set);
        if (threads <= 0) threads = (int)std::thread::hardware_concurrency();
        const int q = cgroup_cpu_quota();
        if (q > 0) threads = std::min(threads, 2 * q);
    }
    if (threads <= 0) threads = 1;
    return threads > 256 ? 256 : threads;
}

// workers worth starting for n items when one worker should get at least `grain` of them
inline int workers_for(int64_t n, int threads, int64_t grain) {
    const int64_t by_work = (n + grain - 1) / grain;
    return (int)std::max<int64_t>(1, std::min<int64_t>(threads, by_work));
}

// The library's worker threads: created once and parked between jobs (a staged batch runs a dozen short parallel passes; starting
// 256 threads for each costs more than the passes).  One job at a time (callers on other threads queue); a job started from inside a
// job runs inline.  The pool follows the caller's CPU affinity: when that changes it is rebuilt.
class ThreadPool {
  public:
    // One pool per process, on the heap and never destroyed: no join at exit (a destructor that ran in a forked child would join the
    // PARENT's thread handles), and a forked child gets a fresh pool -- the atfork handler builds one over the inherited bytes, whose
    // mutexes may have been copied in a locked state and whose workers do not exist on this side of the fork.
    static ThreadPool &get() {
        static ThreadPool *p = [] {
            ThreadPool *q = new ThreadPool();
            pthread_atfork(nullptr, nullptr, [] { new (&ThreadPool::get()) ThreadPool(); });
            return q;
        }();
        return *p;
    }
    // fn(worker) on `w` workers (the caller is worker 0); the first exception wins
    template <typename F>
    void run(int w, F &&fn) {
        if (w <= 1 || in_job()) { fn(0); return; }
        std::lock_guard<std::mutex> submit(submit_mu_);
        ensure(w - 1);
        w = std::min<int>(w, (int)th_.size() + 1);
        std::vector<IoError> errs((size_t)w, IoError{0, ""});
        std::function<void(int)> job = [&](int k) {
            try { fn(k); } catch (const IoError &e) { errs[(size_t)k] = e; } catch (const std::exception &e) { errs[(size_t)k] = IoError{UZ_IO_E_FORMAT, e.what()}; }
        };
        {
            std::lock_guard<std::mutex> g(mu_);
            job_ = &job; want_ = w - 1; active_ = w - 1; gen_++;
        }
        cv_job_.notify_all();
        in_job() = true;
        job(0);
        in_job() = false;
        {
            std::unique_lock<std::mutex> g(mu_);
            cv_done_.wait(g, [&] { return active_ == 0; });
            job_ = nullptr;
        }
        for (auto &e : errs) if (e.code) throw e;
    }

  private:
    ThreadPool() = default;
    static bool &in_job() { static thread_local bool f = false; return f; }
    void shutdown() {
        {
            std::lock_guard<std::mutex> g(mu_);
            stop_ = true; gen_++;
        }
        cv_job_.notify_all();
        for (auto &t : th_) t.join();
        th_.clear();
        stop_ = false;
    }
    void ensure(int n) {
        cpu_set_t now;
        CPU_ZERO(&now);
        sched_getaffinity(0, sizeof(now), &now);
        if (!th_.empty() && !CPU_EQUAL(&now, &mask_)) shutdown();
        mask_ = now;
        n = std::min(n, 1023);
        while ((int)th_.size() < n) {
            const int k = (int)th_.size() + 1;
            const int gen0 = gen_;
            th_.emplace_back([this, k, gen0] {
                in_job() = true;
                int seen = gen0;
                for (;;) {
                    std::function<void(int)> *job = nullptr;
                    {
                        std::unique_lock<std::mutex> g(mu_);
                        cv_job_.wait(g, [&] { return gen_ != seen; });
                        seen = gen_;
                        if (stop_) return;
                        if (k <= want_) job = job_;
                    }
                    if (job) {
                        (*job)(k);
                        std::lock_guard<std::mutex> g(mu_);
                        if (--active_ == 0) cv_done_.notify_all();
                    }
                }
            });
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_, submit_mu_;
    std::condition_variable cv_job_, cv_done_;
    std::function<void(int)> *job_ = nullptr;
    int gen_ = 0, want_ = 0, active_ = 0;
    bool stop_ = false;
    cpu_set_t mask_;
};

// fn(lo, hi, worker) over [0, n) cut into one contiguous slice per worker; the first exception wins
template <typename F>
void parallel_slices(int64_t n, int threads, F fn) {
    if (n <= 0) return;
    const int w = (int)std::min<int64_t>(threads, n);
    if (w <= 1) { fn((int64_t)0, n, 0); return; }
    ThreadPool::get().run(w, [&](int k) { fn(n * k / w, n * (k + 1) / w, k); });
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

// ------------------------------------------------------------------------------------------------ inflate
struct LibDeflate {
    void *(*alloc)() = nullptr;
    int (*decompress)(void *, const void *, size_t, void *, size_t, size_t *) = nullptr;
    void (*release)(void *) = nullptr;
    uint32_t (*crc)(uint32_t, const void *, size_t) = nullptr;
    bool ok = false;
};

inline const LibDeflate &libdeflate() {
    static LibDeflate L;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *e = getenv("UZ_INFLATE");
        if (e && strcmp(e, "zlib") == 0) return;
        void *h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("libdeflate.so", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        L.alloc = (void *(*)())dlsym(h, "libdeflate_alloc_decompressor");
        L.decompress = (int (*)(void *, const void *, size_t, void *, size_t, size_t *))dlsym(h, "libdeflate_deflate_decompress");
        L.release = (void (*)(void *))dlsym(h, "libdeflate_free_decompressor");
        L.crc = (uint32_t (*)(uint32_t, const void *, size_t))dlsym(h, "libdeflate_crc32");
        L.ok = L.alloc && L.decompress && L.release && L.crc;
    });
    return L;
}

struct Inflater {
    void *ld = nullptr;
    z_stream z;
    bool z_init = false;
    Inflater() {
        const LibDeflate &L = libdeflate();
        if (L.ok) ld = L.alloc();
        if (!ld) {
            memset(&z, 0, sizeof(z));
            if (inflateInit2(&z, -15) != Z_OK) fail(UZ_IO_E_FORMAT, "zlib init failed");
            z_init = true;
        }
    }
    ~Inflater() {
        if (ld) libdeflate().release(ld);
        if (z_init) inflateEnd(&z);
    }
    Inflater(const Inflater &) = delete;
    uint32_t crc_of(const uint8_t *p, size_t n) const { return ld ? libdeflate().crc(0, p, n) : (uint32_t)crc32(0L, p, (uInt)n); }
    void block(const uint8_t *c, size_t clen, uint8_t *dst, size_t isize, uint32_t crc, int64_t coff) {
        if (isize == 0) return;
        if (ld) {
            size_t got = 0;
            if (libdeflate().decompress(ld, c, clen, dst, isize, &got) != 0 || got != isize) fail(UZ_IO_E_FORMAT, "corrupt BGZF block at byte %lld", (long long)coff);
            if (libdeflate().crc(0, dst, isize) != crc) fail(UZ_IO_E_FORMAT, "CRC mismatch in the BGZF block at byte %lld", (long long)coff);
            return;
        }
        inflateReset(&z);
        z.next_in = const_cast<Bytef *>(c);
        z.avail_in = (uInt)clen;
        z.next_out = dst;
        z.avail_out = (uInt)isize;
        if (inflate(&z, Z_FINISH) != Z_STREAM_END || z.avail_out != 0) fail(UZ_IO_E_FORMAT, "corrupt BGZF block at byte %lld", (long long)coff);
        if ((uint32_t)crc32(0L, dst, (uInt)isize) != crc) fail(UZ_IO_E_FORMAT, "CRC mismatch in the BGZF block at byte %lld", (long long)coff);
    }
};

template <typename F>
void parallel_dynamic(int64_t n, int threads, F fn) { // fn(item, worker): items handed out one by one (uneven costs)
    if (n <= 0) return;
    const int w = (int)std::min<int64_t>(std::max(1, threads), n);
    if (w <= 1) { for (int64_t i = 0; i < n; i++) fn(i, 0); return; }
    std::atomic<int64_t> next{0};
    ThreadPool::get().run(w, [&](int k) {
        try {
            for (;;) {
                const int64_t i = next.fetch_add(1);
                if (i >= n) break;
                fn(i, k);
            }
        } catch (...) { next.store(n); throw; }
    });
}

// ---- the pair form of tlen / mate / name id (uz_reads_packed_view.pair_d8): what both packers ask of one output record
struct PairRec {
    int64_t mate;   // output index of the record mate() returns, -1 = none
    int32_t start, end, tlen;
    uint32_t qid;   // name id as the output numbers it (first appearance)
    bool is_new;    // the first record of the output carrying its name
};
// code of output record k; get(j) -> PairRec of output record j.  A pair gets the FIRST / SECOND codes only when every statement
// the device rebuilds it from holds: the two name each other, lie at most UZ_P8_MAX_DIST records apart, the earlier one brings the
// name, and the template lengths are +-t (t = the span of the pair, or the SECOND carries its own).
template <typename Get>
inline uint8_t pair8_code(int64_t k, Get &&get) {
    const PairRec x = get(k);
    if (x.mate >= 0 && x.mate != k) {
        const int64_t a = std::min(k, x.mate), b = std::max(k, x.mate);
        if (b - a <= UZ_P8_MAX_DIST) {
            const PairRec A = a == k ? x : get(a), B = b == k ? x : get(b);
            const int64_t span = (int64_t)std::max(A.end, B.end) - (int64_t)A.start;
            if (A.mate == b && B.mate == a && A.is_new && !B.is_new && B.qid == A.qid && (int64_t)B.tlen == -(int64_t)A.tlen && span < 0x7FFFFFFFLL)
                return a == k ? (uint8_t)(b - a) : ((int64_t)A.tlen == span ? (uint8_t)UZ_P8_SECOND : (uint8_t)UZ_P8_SECOND_TLEN);
        }
    }
    return x.is_new ? (uint8_t)UZ_P8_NEW : (uint8_t)UZ_P8_OLD;
}
// the escape-list entries a record of the pair form owns besides its start: e[1] tlen, e[2] mate, e[3] name id; returns how many (0 .. 3)
inline int pair8_escapes(uint8_t code, const PairRec &x, int32_t e[4], bool has[4]) {
    has[1] = has[2] = has[3] = false;
    if (code == UZ_P8_SECOND_TLEN) { has[1] = true; e[1] = x.tlen; return 1; }
    if (code != UZ_P8_NEW && code != UZ_P8_OLD) return 0;
    has[1] = has[2] = true; e[1] = x.tlen; e[2] = (int32_t)x.mate;
    if (code == UZ_P8_OLD) { has[3] = true; e[3] = (int32_t)x.qid; }
    return code == UZ_P8_OLD ? 3 : 2;
}

// whole file -> memory
Bytes read_file(const char *path);
// gzip / BGZF stream -> bytes.  BGZF blocks (BC extra field) are inflated in parallel, any other
// gzip stream sequentially, member after member.  Not gzip at all -> the input itself is handed
// back (moved) with *was_gzip = false.
Bytes inflate_all(Bytes &file, int threads, bool *was_gzip);

} // namespace uzio
