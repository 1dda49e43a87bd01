// abi.hip -- the extern "C" entry points of include/unfazed_hip.h: context, staging of
// the decoded columns into HBM, and the launch sequence of the stages.
#include "uz_ctx.hpp"

#include <chrono>
#include <algorithm>
#include <map>
#include <mutex>

void uz_fold_complex(uz_ctx *c, uint8_t *gt, const uint8_t *sflags, int64_t n);
bool uz_site_scan_fresh(const uz_ctx *c, const FamilyDev &f, bool need_cnv);

namespace {

// The message of a failed call.  uz_bgzf_inflate_to_host may run on a decoder's worker thread beside the owner's calls, so the
// context's message is written under a lock, and uz_last_error hands out a copy that belongs to the calling thread.
void set_error(uz_ctx *c, const std::string &msg) {
    std::lock_guard<std::mutex> lk(c->err_mu);
    c->err = msg;
}

template <typename F>
int guarded(uz_ctx *c, F &&fn) {
    if (!c) return UZ_E_ARG;
    try {
        (void)hipSetDevice(c->device);
        fn();
        return 0;
    } catch (const UzError &e) {
        set_error(c, e.msg);
        return e.code;
    } catch (const std::exception &e) {
        set_error(c, e.what());
        return UZ_E_ARG;
    } catch (...) {
        set_error(c, "unknown error");
        return UZ_E_ARG;
    }
}

template <typename T>
T *upload(uz_ctx *c, const T *host, size_t n) {
    T *d = nullptr;
    UZ_HIP(hipMalloc((void **)&d, (n ? n : 1) * sizeof(T) + 64)); // +64: vector tail reads stay in-bounds
    if (n) {
        UZ_REQUIRE(host != nullptr, UZ_E_ARG, "null column pointer");
        UZ_HIP(hipMemcpyAsync(d, host, n * sizeof(T), hipMemcpyHostToDevice, c->stream));
    }
    return d;
}

// One copy instead of one per column: when the host columns of a table sit back to back in one slab of (pinned) memory --
// a decoder that carves its output from one allocation -- the slab goes over the link as a whole into a mirror block and
// the columns' device addresses follow from their offsets in it.  Every separate copy costs the link ~8 us of set-up, and
// a staged table has about twenty columns, half of them tiny.  A staging routine runs its h2d() calls twice: a planning
// pass that only looks at the host addresses, then -- mirrored or not -- the pass that yields the device pointers.
// the page-locked blocks this library handed out (uz_pinned_alloc): the one-copy path ships a span of host memory as a whole, so the
// span must lie inside ONE block the caller really owns -- never a guess from address density (columns of separate allocations
// that happen to sit close together, with unmapped or foreign memory between them)
static std::mutex g_pinned_mu;
static std::map<uintptr_t, size_t> g_pinned;
static bool inside_one_pinned_block(const uint8_t *lo, const uint8_t *hi) {
    std::lock_guard<std::mutex> g(g_pinned_mu);
    auto it = g_pinned.upper_bound((uintptr_t)lo);
    if (it == g_pinned.begin()) return false;
    --it;
    return (uintptr_t)lo >= it->first && (uintptr_t)hi <= it->first + it->second;
}

struct SlabPlan {
    int mode = 1; // 1 planning, 2 mirrored
    const uint8_t *lo = nullptr, *hi = nullptr;
    size_t sum = 0;
    std::vector<const uint8_t *> at;
    uint8_t *dev = nullptr;
    void add(const void *p, size_t bytes) {
        const uint8_t *q = (const uint8_t *)p;
        if (!lo || q < lo) lo = q;
        if (!hi || q + bytes > hi) hi = q + bytes;
        sum += bytes;
        at.push_back(q);
    }
    size_t span() const { return (size_t)(hi - lo); }
    bool worth() const {
        if (at.size() < 2 || span() > sum + sum / 8 + 1024 * at.size()) return false;
        for (const uint8_t *q : at)
            if ((size_t)(q - lo) % 256) return false; // the kernels' vector loads want the carver's alignment
        return inside_one_pinned_block(lo, hi);
    }
};
static thread_local SlabPlan *g_slab = nullptr;
struct SlabScope {
    explicit SlabScope(SlabPlan *p) { g_slab = p; }
    ~SlabScope() { g_slab = nullptr; }
};
static const bool g_slab_off = getenv("UZ_NO_SLAB_COPY") != nullptr;

template <typename T>
const T *h2d(hipStream_t st, T *dst, const T *host, size_t n) {
    if (n) {
        UZ_REQUIRE(host != nullptr, UZ_E_ARG, "null column pointer");
        if (g_slab && g_slab->mode == 1) { g_slab->add(host, n * sizeof(T)); return dst; }
        if (g_slab && g_slab->mode == 2) return reinterpret_cast<const T *>(g_slab->dev + ((const uint8_t *)host - g_slab->lo));
        UZ_HIP(hipMemcpyAsync(dst, host, n * sizeof(T), hipMemcpyHostToDevice, st));
    }
    return dst;
}
// after the planning pass: ship the slab (mirror: the device block that receives it) or fall back to column copies
static void slab_commit(uz_ctx *c, hipStream_t st, SlabPlan &plan, DevBlock &mirror) {
    if (!g_slab_off && plan.worth()) {
        mirror = uz_block_get(c, plan.span() + 512);
        UZ_HIP(hipMemcpyAsync(mirror.p, plan.lo, plan.span(), hipMemcpyHostToDevice, st));
        plan.dev = mirror.p;
        plan.mode = 2;
    } else
        g_slab = nullptr;
}

template <typename T>
void stage(uz_ctx *c, DevBuf<T> &b, const T *host, size_t n) {
    b.ensure(n + 1);
    if (n) {
        UZ_REQUIRE(host != nullptr, UZ_E_ARG, "null DNM column pointer");
        UZ_HIP(hipMemcpyAsync(b.p, host, n * sizeof(T), hipMemcpyHostToDevice, c->stream));
    }
}

template <typename V>
int new_slot(std::vector<V> &v) {
    for (size_t i = 0; i < v.size(); i++)
        if (!v[i].live) return (int)i;
    v.emplace_back();
    return (int)v.size() - 1;
}

SitesDev &sites_of(uz_ctx *c, int id) {
    UZ_REQUIRE(id >= 0 && id < (int)c->sites.size() && c->sites[id].live, UZ_E_ARG, "unknown sites handle");
    SitesDev &s = c->sites[id];
    if (s.pending) { // queued by uz_sites_family_upload_async
        s.pending = false;
        UZ_HIP(hipStreamWaitEvent(c->stream, s.ready, 0));
    }
    return s;
}
FamilyDev &fam_of(uz_ctx *c, int id) {
    UZ_REQUIRE(id >= 0 && id < (int)c->fams.size() && c->fams[id].live, UZ_E_ARG, "unknown family handle");
    FamilyDev &f = c->fams[id];
    if (f.pending) { // queued by uz_sites_family_upload_async: the compute stream waits for the copies, then folds the complex flag
        f.pending = false;
        UZ_HIP(hipStreamWaitEvent(c->stream, f.ready, 0));
        SitesDev &s = sites_of(c, f.sites_id);
        uz_family_widen(c, f, s.n);
        uz_fold_complex(c, f.gt, s.sflags, s.n);
    }
    return f;
}
ReadsDev &reads_of(uz_ctx *c, int id) {
    UZ_REQUIRE(id >= 0 && id < (int)c->reads.size() && c->reads[id].live, UZ_E_ARG, "unknown reads handle");
    return c->reads[id];
}

void free_family(uz_ctx *c, FamilyDev &f) {
    if (f.ready) { (void)hipEventSynchronize(f.ready); (void)hipEventDestroy(f.ready); f.ready = nullptr; }
    f.pending = false;
    uz_block_put(c, f.block);
    uz_block_put(c, f.wide_block);
    f = FamilyDev();
}
void free_sites(uz_ctx *c, SitesDev &s) {
    if (s.ready) { (void)hipEventSynchronize(s.ready); (void)hipEventDestroy(s.ready); }
    uz_block_put(c, s.block);
    uz_block_put(c, s.mirror);
    s = SitesDev();
}
void free_reads(uz_ctx *c, ReadsDev &r) {
    if (r.built) { (void)hipEventSynchronize(r.built); (void)hipEventDestroy(r.built); } // (a table freed before its first use: its build may still run)
    if (r.ready) (void)hipEventDestroy(r.ready);
    uz_block_put(c, r.block);
    uz_block_put(c, r.mirror);
    r = ReadsDev();
}

// carve arrays out of a table's block
struct Carver {
    uint8_t *base;
    size_t off = 0;
    explicit Carver(uint8_t *b) : base(b) {}
    template <typename T>
    T *take(size_t n) {
        off = (off + 255) & ~(size_t)255;
        T *p = base ? reinterpret_cast<T *>(base + off) : nullptr;
        off += (n ? n : 1) * sizeof(T) + 64; // +64: vector tail reads stay in-bounds
        return p;
    }
};

} // namespace

// ---------------------------------------------------------------- copy kernel (see uz_ctx.hpp)
template <typename W>
__global__ __launch_bounds__(256) void k_copy(W *__restrict__ dst, const W *__restrict__ src, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
void uz_kcopy(uz_ctx *c, void *dst, const void *src, size_t bytes) {
    if (!bytes) return;
    const uintptr_t a = (uintptr_t)dst | (uintptr_t)src | (uintptr_t)bytes;
    auto grid = [](size_t n) { return dim3((unsigned)std::min<size_t>((n + 255) / 256, 1024)); };
    if (a % 16 == 0) hipLaunchKernelGGL(k_copy<uint4>, grid(bytes / 16), dim3(256), 0, c->stream, (uint4 *)dst, (const uint4 *)src, bytes / 16);
    else if (a % 4 == 0) hipLaunchKernelGGL(k_copy<uint32_t>, grid(bytes / 4), dim3(256), 0, c->stream, (uint32_t *)dst, (const uint32_t *)src, bytes / 4);
    else hipLaunchKernelGGL(k_copy<uint8_t>, grid(bytes), dim3(256), 0, c->stream, (uint8_t *)dst, (const uint8_t *)src, bytes);
    UZ_HIP(hipGetLastError());
}

// ---------------------------------------------------------------- device block pool
DevBlock uz_block_get(uz_ctx *c, size_t bytes) {
    int best = -1;
    for (size_t i = 0; i < c->block_pool.size(); i++)
        if (c->block_pool[i].cap >= bytes && (best < 0 || c->block_pool[i].cap < c->block_pool[best].cap)) best = (int)i;
    if (best >= 0 && c->block_pool[best].cap <= 2 * bytes + (1 << 20)) {
        DevBlock b = c->block_pool[best];
        c->block_pool.erase(c->block_pool.begin() + best);
        return b;
    }
    DevBlock b;
    b.cap = bytes + bytes / 16 + 4096;
    UZ_HIP(hipMalloc((void **)&b.p, b.cap));
    return b;
}
void uz_block_put(uz_ctx *c, DevBlock b) {
    if (!b.p) return;
    c->block_pool.push_back(b);
    // keep the parked memory bounded: beyond 64 blocks the smallest ones are returned to the driver
    while (c->block_pool.size() > 64) {
        size_t k = 0;
        for (size_t i = 1; i < c->block_pool.size(); i++)
            if (c->block_pool[i].cap < c->block_pool[k].cap) k = i;
        (void)hipFree(c->block_pool[k].p);
        c->block_pool.erase(c->block_pool.begin() + k);
    }
}

// ---------------------------------------------------------------- profiling
void uz_prof_begin(uz_ctx *c, int kernel, hipEvent_t *a, hipEvent_t *b) {
    *a = *b = nullptr;
    if (!(c->prof_mask >> kernel & 1u)) return;
    for (hipEvent_t *e : {a, b}) {
        if (!c->event_pool.empty()) { *e = c->event_pool.back(); c->event_pool.pop_back(); }
        else UZ_HIP(hipEventCreate(e));
    }
    UZ_HIP(hipEventRecord(*a, c->stream));
}
void uz_prof_end(uz_ctx *c, int kernel, hipEvent_t a, hipEvent_t b) {
    if (!a) return;
    (void)hipEventRecord(b, c->stream);
    c->prof_pending.push_back(ProfPending{kernel, a, b});
}
void uz_prof_drain(uz_ctx *c) {
    for (auto &p : c->prof_pending) {
        float ms = 0;
        if (hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            c->prof[p.kernel].total_ms += ms;
            c->prof[p.kernel].launches += 1;
        }
        c->event_pool.push_back(p.a);
        c->event_pool.push_back(p.b);
    }
    c->prof_pending.clear();
}

void uz_stage_dnms(uz_ctx *c, const uz_dnms_view *d) {
    UZ_REQUIRE(d != nullptr && d->n >= 0, UZ_E_ARG, "bad DNM view");
    const size_t n = (size_t)d->n;
    c->dn.n = d->n;
    UZ_REQUIRE(n == 0 || (d->contig && d->rcontig && d->start && d->end && d->vartype && d->dflags && d->mult && d->allele_off),
               UZ_E_ARG, "null DNM column pointer");
    const size_t nb = n ? (size_t)d->allele_off[2 * n] : 0;
    UZ_REQUIRE(nb == 0 || d->alleles != nullptr, UZ_E_ARG, "null DNM column pointer");
    // The nine small columns go through ONE pinned staging buffer: copies from pageable memory are staged
    // by the runtime one call at a time (~40 us each with the gaps between them, ~0.4 ms per batch).
    const size_t sizes[9] = {n * 4, n * 4, n * 4, n * 4, n, n, n, (2 * n + 1) * 4, nb};
    size_t off[10];
    off[0] = 0;
    for (int k = 0; k < 9; k++) off[k + 1] = (off[k] + sizes[k] + 63) & ~(size_t)63;
    // (an earlier batch's copies out of the buffer are usually long done -- every entry point but uz_phase_begin ends with a stream sync)
    if (c->dn_stage_done) UZ_HIP(hipEventSynchronize(c->dn_stage_done));
    if (c->dn_stage_cap < off[9]) {
        if (c->dn_stage) (void)hipHostFree(c->dn_stage);
        c->dn_stage = nullptr;
        c->dn_stage_cap = off[9] + off[9] / 4 + 4096;
        UZ_HIP(hipHostMalloc((void **)&c->dn_stage, c->dn_stage_cap, hipHostMallocDefault));
    }
    const void *src[9] = {d->contig, d->rcontig, d->start, d->end, d->vartype, d->dflags, d->mult, d->allele_off, d->alleles};
    for (int k = 0; k < 9; k++)
        if (sizes[k]) memcpy(c->dn_stage + off[k], src[k], sizes[k]);
    c->dn.block.ensure(off[9] + 64);
    uint8_t *const b = c->dn.block.p;
    c->dn.contig.p = (int32_t *)(b + off[0]); c->dn.rcontig.p = (int32_t *)(b + off[1]); c->dn.start.p = (int32_t *)(b + off[2]);
    c->dn.end.p = (int32_t *)(b + off[3]); c->dn.vartype.p = b + off[4]; c->dn.dflags.p = b + off[5]; c->dn.mult.p = b + off[6];
    c->dn.allele_off.p = (uint32_t *)(b + off[7]); c->dn.alleles.p = b + off[8];
    if (off[9]) uz_kcopy(c, b, c->dn_stage, off[9]); // out of the pinned staging buffer, by ONE kernel (the offsets are multiples of 64)
    if (!c->dn_stage_done) UZ_HIP(hipEventCreateWithFlags(&c->dn_stage_done, hipEventDisableTiming));
    UZ_HIP(hipEventRecord(c->dn_stage_done, c->stream));
    c->dn.cutoff = d->cutoff;
}

// ---------------------------------------------------------------- window lists of the last finds (uz_ctx.hpp: find_key, find_alt)
static FindKey find_key_of(const uz_ctx *c, int fam_id, int mode, const uz_dnms_view *d) {
    FindKey k;
    k.valid = d != nullptr && d->n >= 0;
    k.fam = fam_id; k.mode = mode; k.n = d ? d->n : -1; k.cohort = c->cohort_on;
    k.P = c->P;
    uint64_t h = 0x9E3779B97F4A7C15ULL;
    // (four chains side by side over 32-byte steps, folded at the end of every column: one multiply chain over the 1.4 MB of a 100 k-DNM batch was
    // 0.21 ms of every read-stage call with the device idle; this is 0.04)
    auto mix = [&](const void *p, size_t bytes) {
        if (!p) return;
        const uint8_t *b = (const uint8_t *)p;
        uint64_t a0 = h ^ 0x243F6A8885A308D3ULL, a1 = h ^ 0x13198A2E03707344ULL, a2 = h ^ 0xA4093822299F31D0ULL, a3 = h ^ 0x082EFA98EC4E6C89ULL;
        size_t i = 0;
        for (; i + 32 <= bytes; i += 32) {
            uint64_t w[4];
            memcpy(w, b + i, 32);
            a0 = (a0 ^ w[0]) * 0x100000001B3ULL; a0 ^= a0 >> 29;
            a1 = (a1 ^ w[1]) * 0x100000001B3ULL; a1 ^= a1 >> 29;
            a2 = (a2 ^ w[2]) * 0x100000001B3ULL; a2 ^= a2 >> 29;
            a3 = (a3 ^ w[3]) * 0x100000001B3ULL; a3 ^= a3 >> 29;
        }
        h = (((a0 * 0x9E3779B97F4A7C15ULL) ^ a1) * 0x9E3779B97F4A7C15ULL ^ a2) * 0x9E3779B97F4A7C15ULL ^ a3;
        for (; i + 8 <= bytes; i += 8) { uint64_t w; memcpy(&w, b + i, 8); h = (h ^ w) * 0x100000001B3ULL; h ^= h >> 29; }
        for (; i < bytes; i++) h = (h ^ b[i]) * 0x100000001B3ULL;
    };
    if (k.valid && d->n > 0) {
        const size_t n = (size_t)d->n;
        mix(d->contig, n * 4); mix(d->start, n * 4); mix(d->end, n * 4); mix(d->vartype, n); mix(d->mult, n);
    }
    k.hash = h;
    return k;
}
static bool find_key_eq(const FindKey &a, const FindKey &b) {
    return a.valid && b.valid && a.fam == b.fam && a.mode == b.mode && a.n == b.n && a.cohort == b.cohort && a.hash == b.hash &&
           memcmp(&a.P, &b.P, sizeof(uz_params)) == 0;
}
static void find_swap(uz_ctx *c, int which) { // a parked set becomes the context's, and the other way round (pointers only)
    FindSlot &a = c->find_alt[which];
    std::swap(c->cnt_c, a.cnt_c); std::swap(c->cnt_h, a.cnt_h); std::swap(c->win_range, a.win_range);
    std::swap(c->cand_off, a.cand_off); std::swap(c->het_off, a.het_off);
    std::swap(c->cand_idx, a.cand_idx); std::swap(c->het_idx, a.het_idx); std::swap(c->cand_flags, a.cand_flags);
    std::swap(c->n_cand, a.n_cand); std::swap(c->n_het, a.n_het);
    c->cand_off_h.swap(a.cand_off_h); c->het_off_h.swap(a.het_off_h);
    std::swap(c->find_key, a.key); std::swap(c->find_stamp, a.stamp);
}
// a new find overwrites the OLDEST set, or one that holds nothing (kernels still reading it are ahead of the new ones on the stream)
static void find_target(uz_ctx *c) {
    if (c->find_key.valid) {
        int best = -1;
        unsigned long long best_stamp = c->find_stamp;
        for (int k = 0; k < UZ_FIND_ALTS; k++) {
            const FindSlot &a = c->find_alt[k];
            if (!a.key.valid) { best = k; break; }
            if (a.stamp < best_stamp) { best = k; best_stamp = a.stamp; }
        }
        if (best >= 0) find_swap(c, best);
    }
    c->find_key.valid = false;
    c->find_valid = false;
}
static void find_done(uz_ctx *c, const FindKey &k) { c->find_key = k; c->find_stamp = ++c->find_counter; }
// the context's lists become those of batch `k` if one of the sets holds them; false: they must be computed
static bool find_recall(uz_ctx *c, const FindKey &k) {
    if (find_key_eq(c->find_key, k)) { c->find_valid = true; c->find_mode = k.mode; return true; }
    for (int a = 0; a < UZ_FIND_ALTS; a++)
        if (find_key_eq(c->find_alt[a].key, k)) { find_swap(c, a); c->find_valid = true; c->find_mode = k.mode; return true; }
    return false;
}
static void find_forget(uz_ctx *c, int fam_id /* -1: everything */) {
    if (fam_id < 0 || c->find_key.fam == fam_id) { c->find_key.valid = false; c->find_valid = false; }
    for (int a = 0; a < UZ_FIND_ALTS; a++)
        if (fam_id < 0 || c->find_alt[a].key.fam == fam_id) c->find_alt[a].key.valid = false;
}

extern "C" {

int uz_create(int device, uz_ctx **out) {
    if (!out) return UZ_E_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return UZ_E_NODEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return UZ_E_HIP;
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) return UZ_E_NODEVICE; // code objects are gfx950 only
    uz_ctx *c = new uz_ctx();
    c->device = device;
    memset(&c->P, 0, sizeof(c->P));
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess ||
        hipHostMalloc((void **)&c->hflags, 64, hipHostMallocMapped) != hipSuccess) {
        delete c;
        return UZ_E_HIP;
    }
    memset(c->hflags, 0, 64);
    *out = c;
    return 0;
}

void uz_destroy(uz_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    uz_prof_drain(c);
    uz_phase_state_free(c);
    for (auto &f : c->fams) if (f.live) free_family(c, f);
    for (auto &s : c->sites) if (s.live) free_sites(c, s);
    for (auto &r : c->reads) if (r.live) free_reads(c, r);
    for (auto &b : c->block_pool) (void)hipFree(b.p);
    if (c->hflags) (void)hipHostFree(c->hflags);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->build_stream) (void)hipStreamDestroy(c->build_stream);
    c->dn.block.release();
    if (c->dn_stage_done) (void)hipEventDestroy(c->dn_stage_done);
    if (c->dn_stage) (void)hipHostFree(c->dn_stage);
    if (c->find_pin) (void)hipHostFree(c->find_pin);
    if (c->nm_pin) (void)hipHostFree(c->nm_pin);
    c->nm_ids.release(); c->nm_len.release(); c->nm_off.release(); c->nm_out.release();
    c->ab_lut.release(); c->win_range.release();
    c->dn_fam.release(); c->dn_cutoff.release(); c->fam_cls.release();
    c->cnv_counts.release(); c->cnv_pos.release(); c->cnv_origin.release(); c->cnv_evidence.release(); c->cnv_etype.release(); c->cnv_rb.release();
    c->cnt_c.release(); c->cnt_h.release(); c->cand_off.release(); c->het_off.release();
    c->cand_idx.release(); c->het_idx.release(); c->cand_flags.release(); c->win_range.release();
    c->inf_comp.release(); c->inf_out.release(); c->inf_in.release(); c->inf_off.release(); c->inf_flags.release();
    if (c->inf_stream) (void)hipStreamDestroy(c->inf_stream);
    if (c->inf_stream2) (void)hipStreamDestroy(c->inf_stream2);
    if (c->inf_ready) (void)hipEventDestroy(c->inf_ready);
    for (auto &w : c->walk) {
        if (w.s0) (void)hipStreamDestroy(w.s0);
        if (w.s1) (void)hipStreamDestroy(w.s1);
        if (w.ev) (void)hipEventDestroy(w.ev);
        w.comp.release(); w.out.release(); w.in_off.release(); w.out_off.release(); w.blk_coff.release(); w.span.release(); w.count.release(); w.first.release();
        w.walked.release(); w.task.release(); w.reach.release(); w.fetch.release(); w.flags.release(); w.iflags.release(); w.blk_crc.release(); w.desc.release();
        w.desc_kept.release(); w.n_direct.release(); w.tab_first.release(); w.kcount.release(); w.kfirst.release(); w.tab.release();
        auto &J = w.join;
        J.jtask.release(); J.keep.release(); J.mate.release(); J.target.release(); J.h_flags.release(); J.jt_tid.release(); J.reach_a.release(); J.reach_host.release();
        J.cnt.release(); J.cspan.release(); J.look_tid.release(); J.reach_key.release(); J.totals.release(); J.hkey_in.release(); J.hkey.release(); J.fkey_in.release();
        J.fkey.release(); J.ccount.release(); J.hval_in.release(); J.hperm.release(); J.inv.release(); J.fval_in.release(); J.fidx.release(); J.front0.release();
        J.front1.release(); J.need.release(); J.first.release(); J.runid.release(); J.pos_of_k.release(); J.fo.release(); J.name_rec.release(); J.gidx.release();
        J.tmp.release(); J.aux.release(); J.s5_in.release(); J.s5_out.release(); J.need_rec.release(); J.kept.release();
    }
    for (auto &b : c->walk_park) (void)hipFree(b.first);
    for (FindSlot &a : c->find_alt) {
        a.cnt_c.release(); a.cnt_h.release(); a.win_range.release(); a.cand_off.release(); a.het_off.release();
        a.cand_idx.release(); a.het_idx.release(); a.cand_flags.release();
    }
    for (auto e : c->event_pool) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(c->stream);
    delete c;
}

const char *uz_last_error(const uz_ctx *c) {
    if (!c) return "null context";
    static thread_local std::string mine; // valid until this thread's next uz_last_error
    {
        std::lock_guard<std::mutex> lk(const_cast<uz_ctx *>(c)->err_mu);
        mine = c->err;
    }
    return mine.c_str();
}

int uz_sync(uz_ctx *c) {
    return guarded(c, [&] { UZ_HIP(hipStreamSynchronize(c->stream)); });
}

int uz_set_params(uz_ctx *c, const uz_params *p) {
    return guarded(c, [&] {
        UZ_REQUIRE(p != nullptr, UZ_E_ARG, "null params");
        c->P = *p;
    });
}

int uz_sites_upload(uz_ctx *c, const uz_sites_view *v, int *id) {
    return guarded(c, [&] {
        UZ_REQUIRE(v && id && v->n_sites >= 0 && v->n_contigs >= 0, UZ_E_ARG, "bad sites view");
        UZ_REQUIRE(v->n_sites < (int64_t)0x7FFFFFF0, UZ_E_RANGE, "more than 2^31 sites");
        const int k = new_slot(c->sites);
        SitesDev s;
        s.live = true; s.owned = true;
        s.n = v->n_sites; s.n_contigs = v->n_contigs;
        s.contig_off_h.assign(v->contig_off, v->contig_off + v->n_contigs + 1);
        const size_t n = (size_t)s.n;
        for (int pass = 0; pass < 2; pass++) { // one pooled block for the table: no hipMalloc / hipFree per staged pass
            Carver cv(pass ? s.block.p : nullptr);
            s.contig_off = cv.take<int64_t>((size_t)v->n_contigs + 1);
            s.pos = cv.take<int32_t>(n); s.sflags = cv.take<uint8_t>(n); s.ref_base = cv.take<uint8_t>(n); s.alt_base = cv.take<uint8_t>(n);
            if (!pass) s.block = uz_block_get(c, cv.off + 256);
        }
        try {
            h2d(c->stream, s.contig_off, v->contig_off, (size_t)v->n_contigs + 1);
            h2d(c->stream, s.pos, v->pos, n); h2d(c->stream, s.sflags, v->sflags, n);
            h2d(c->stream, s.ref_base, v->ref_base, n); h2d(c->stream, s.alt_base, v->alt_base, n);
            UZ_HIP(hipStreamSynchronize(c->stream));
        } catch (...) { uz_block_put(c, s.block); throw; }
        c->sites[k] = s;
        *id = k;
    });
}

int uz_sites_adopt_device(uz_ctx *c, const uz_sites_view *v, int *id) {
    return guarded(c, [&] {
        UZ_REQUIRE(v && id && v->n_sites >= 0 && v->n_contigs >= 0, UZ_E_ARG, "bad sites view");
        const int k = new_slot(c->sites);
        SitesDev s;
        s.live = true; s.owned = false;
        s.n = v->n_sites; s.n_contigs = v->n_contigs;
        s.contig_off_h.resize((size_t)v->n_contigs + 1);
        UZ_HIP(hipMemcpy(s.contig_off_h.data(), v->contig_off, ((size_t)v->n_contigs + 1) * sizeof(int64_t), hipMemcpyDeviceToHost));
        for (int pass = 0; pass < 2; pass++) {
            Carver cv(pass ? s.block.p : nullptr);
            s.contig_off = cv.take<int64_t>((size_t)v->n_contigs + 1);
            if (!pass) s.block = uz_block_get(c, cv.off + 256);
        }
        h2d(c->stream, s.contig_off, (const int64_t *)s.contig_off_h.data(), (size_t)v->n_contigs + 1);
        s.pos = const_cast<int32_t *>(v->pos);
        s.sflags = const_cast<uint8_t *>(v->sflags);
        s.ref_base = const_cast<uint8_t *>(v->ref_base);
        s.alt_base = const_cast<uint8_t *>(v->alt_base);
        UZ_HIP(hipStreamSynchronize(c->stream));
        c->sites[k] = s;
        *id = k;
    });
}

// which form the nine columns of a family view come in: true = eight bits (all nine of ref_depth8 / alt_depth8 / gq8 set, the 16-bit ones null)
static bool family_view_eight(const uz_family_view *v) {
    int n8 = 0, n16 = 0;
    for (int m = 0; m < 3; m++) {
        n8 += (v->ref_depth8[m] != nullptr) + (v->alt_depth8[m] != nullptr) + (v->gq8[m] != nullptr);
        n16 += (v->ref_depth[m] != nullptr) + (v->alt_depth[m] != nullptr) + (v->gq[m] != nullptr);
    }
    // (a table without sites may leave every pointer null)
    UZ_REQUIRE((n8 == 9 && n16 == 0) || (n8 == 0 && n16 == 9) || (n8 == 0 && n16 == 0), UZ_E_ARG,
               "a family view carries its nine columns in 16 bits OR in 8 bits (ref_depth8 / alt_depth8 / gq8), all nine");
    return n8 == 9;
}

// the too-deep sites of a family view (host pointers) -> the family's side table on the device
static void family_wide(uz_ctx *c, hipStream_t st, const uz_family_view *v, FamilyDev &f, int64_t n_sites) {
    f.n_wide = 0;
    if (!v || v->n_wide <= 0) return;
    UZ_REQUIRE(v->wide_site != nullptr, UZ_E_ARG, "n_wide set but wide_site is null");
    const size_t w = (size_t)v->n_wide;
    f.wide_host = std::make_shared<std::vector<int32_t>>(6 * w);
    std::vector<int32_t> &dep = *f.wide_host;
    for (int m = 0; m < 3; m++) {
        UZ_REQUIRE(v->wide_ref_depth[m] && v->wide_alt_depth[m], UZ_E_ARG, "null wide depth column");
        for (size_t k = 0; k < w; k++) {
            const int32_t r = v->wide_ref_depth[m][k], a = v->wide_alt_depth[m][k];
            UZ_REQUIRE(r >= -1 && a >= -1 && r <= (1 << 30) && a <= (1 << 30), UZ_E_RANGE, "wide depth outside [-1, 2^30]");
            dep[(size_t)m * w + k] = r; dep[(size_t)(3 + m) * w + k] = a;
        }
    }
    for (size_t k = 0; k < w; k++) {
        UZ_REQUIRE(v->wide_site[k] >= 0 && v->wide_site[k] < n_sites && (k == 0 || v->wide_site[k] > v->wide_site[k - 1]), UZ_E_ARG,
                   "wide_site must be ascending site indices of the table");
    }
    for (int pass = 0; pass < 2; pass++) {
        Carver cv(pass ? f.wide_block.p : nullptr);
        f.wide_site = cv.take<int64_t>(w);
        f.wide_depth = cv.take<int32_t>(6 * w);
        if (!pass) f.wide_block = uz_block_get(c, cv.off + 256);
    }
    UZ_HIP(hipMemcpyAsync(f.wide_site, v->wide_site, w * sizeof(int64_t), hipMemcpyHostToDevice, st));
    UZ_HIP(hipMemcpyAsync(f.wide_depth, dep.data(), 6 * w * sizeof(int32_t), hipMemcpyHostToDevice, st)); // (`dep` lives as long as the family)
    f.n_wide = (int64_t)w;
}

static void family_common(uz_ctx *c, int sites_id, FamilyDev &f) {
    SitesDev &s = sites_of(c, sites_id);
    f.live = true;
    f.sites_id = sites_id;
    uz_fold_complex(c, f.gt, s.sflags, s.n);
    UZ_HIP(hipStreamSynchronize(c->stream));
}

int uz_family_upload(uz_ctx *c, int sites_id, const uz_family_view *v, int *id) {
    return guarded(c, [&] {
        UZ_REQUIRE(v && id, UZ_E_ARG, "bad family view");
        SitesDev &s = sites_of(c, sites_id);
        FamilyDev f;
        f.owned = true;
        const size_t n = (size_t)s.n;
        const bool eight = family_view_eight(v);
        uint8_t *st8[9] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        for (int pass = 0; pass < 2; pass++) {
            Carver cv(pass ? f.block.p : nullptr);
            f.cls = cv.take<uint8_t>(n);
            f.gt = cv.take<uint8_t>(n);
            for (int m = 0; m < 3; m++) { f.rd[m] = cv.take<uint16_t>(n); f.ad[m] = cv.take<uint16_t>(n); f.gq[m] = cv.take<uint16_t>(n); }
            if (eight) for (int k = 0; k < 9; k++) st8[k] = cv.take<uint8_t>(n);
            if (!pass) f.block = uz_block_get(c, cv.off + 256);
        }
        try {
            h2d(c->stream, f.gt, v->gt, n);
            if (eight) {
                for (int m = 0; m < 3; m++) {
                    f.stage8[m] = h2d(c->stream, st8[m], v->ref_depth8[m], n); f.stage8[3 + m] = h2d(c->stream, st8[3 + m], v->alt_depth8[m], n);
                    f.stage8[6 + m] = h2d(c->stream, st8[6 + m], v->gq8[m], n);
                }
                f.widen_pending = true; f.gq_clamped = true;
                uz_family_widen(c, f, s.n);
            } else
                for (int m = 0; m < 3; m++) {
                    h2d(c->stream, f.rd[m], v->ref_depth[m], n); h2d(c->stream, f.ad[m], v->alt_depth[m], n); h2d(c->stream, f.gq[m], v->gq[m], n);
                }
            family_wide(c, c->stream, v, f, s.n);
            family_common(c, sites_id, f);
        } catch (...) { uz_block_put(c, f.block); uz_block_put(c, f.wide_block); throw; }
        const int k = new_slot(c->fams);
        c->fams[k] = f;
        *id = k;
    });
}

int uz_sites_family_upload_async(uz_ctx *c, const uz_sites_view *v, const uz_family_view *fv, int *sites_id, int *fam_id) {
    return guarded(c, [&] {
        UZ_REQUIRE(v && fv && sites_id && fam_id && v->n_sites >= 0 && v->n_contigs >= 0, UZ_E_ARG, "bad sites / family view");
        UZ_REQUIRE(v->n_sites < (int64_t)0x7FFFFFF0, UZ_E_RANGE, "more than 2^31 sites");
        SitesDev s;
        s.live = true; s.owned = true;
        s.n = v->n_sites; s.n_contigs = v->n_contigs;
        s.contig_off_h.assign(v->contig_off, v->contig_off + v->n_contigs + 1);
        const size_t n = (size_t)s.n;
        FamilyDev f;
        f.owned = true;
        const bool eight = family_view_eight(fv);
        uint8_t *st8[9] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        for (int pass = 0; pass < 2; pass++) {
            Carver cv(pass ? s.block.p : nullptr);
            s.contig_off = cv.take<int64_t>((size_t)v->n_contigs + 1);
            s.pos = cv.take<int32_t>(n); s.sflags = cv.take<uint8_t>(n); s.ref_base = cv.take<uint8_t>(n); s.alt_base = cv.take<uint8_t>(n);
            if (!pass) s.block = uz_block_get(c, cv.off + 256);
        }
        for (int pass = 0; pass < 2; pass++) {
            Carver cv(pass ? f.block.p : nullptr);
            f.cls = cv.take<uint8_t>(n);
            f.gt = cv.take<uint8_t>(n);
            for (int m = 0; m < 3; m++) { f.rd[m] = cv.take<uint16_t>(n); f.ad[m] = cv.take<uint16_t>(n); f.gq[m] = cv.take<uint16_t>(n); }
            if (eight) for (int k = 0; k < 9; k++) st8[k] = cv.take<uint8_t>(n);
            if (!pass) f.block = uz_block_get(c, cv.off + 256);
        }
        try {
            hipStream_t st = c->copy_stream;
            SlabPlan plan;
            SlabScope slab_scope(&plan);
            for (int stage_pass = 0; stage_pass < 2; stage_pass++) {
                if (stage_pass) slab_commit(c, st, plan, s.mirror);
                s.contig_off = const_cast<int64_t *>(h2d(st, s.contig_off, v->contig_off, (size_t)v->n_contigs + 1));
                s.pos = const_cast<int32_t *>(h2d(st, s.pos, v->pos, n)); s.sflags = const_cast<uint8_t *>(h2d(st, s.sflags, v->sflags, n));
                s.ref_base = const_cast<uint8_t *>(h2d(st, s.ref_base, v->ref_base, n)); s.alt_base = const_cast<uint8_t *>(h2d(st, s.alt_base, v->alt_base, n));
                f.gt = const_cast<uint8_t *>(h2d(st, f.gt, fv->gt, n)); // (bit 6 is written by the library: the mirror is the library's own memory)
                if (eight) { // (the staged bytes may end up in the mirror block: the 16-bit columns are always the family's own)
                    for (int m = 0; m < 3; m++) {
                        f.stage8[m] = h2d(st, st8[m], fv->ref_depth8[m], n); f.stage8[3 + m] = h2d(st, st8[3 + m], fv->alt_depth8[m], n);
                        f.stage8[6 + m] = h2d(st, st8[6 + m], fv->gq8[m], n);
                    }
                } else
                    for (int m = 0; m < 3; m++) {
                        f.rd[m] = const_cast<uint16_t *>(h2d(st, f.rd[m], fv->ref_depth[m], n));
                        f.ad[m] = const_cast<uint16_t *>(h2d(st, f.ad[m], fv->alt_depth[m], n));
                        f.gq[m] = const_cast<uint16_t *>(h2d(st, f.gq[m], fv->gq[m], n));
                    }
            }
            if (eight) { f.widen_pending = true; f.gq_clamped = true; }
            family_wide(c, st, fv, f, s.n);
            UZ_HIP(hipEventCreateWithFlags(&f.ready, hipEventDisableTiming));
            UZ_HIP(hipEventRecord(f.ready, st));
            UZ_HIP(hipEventCreateWithFlags(&s.ready, hipEventDisableTiming));
            UZ_HIP(hipEventRecord(s.ready, st));
            s.pending = true;
        } catch (...) { uz_block_put(c, s.block); uz_block_put(c, s.mirror); uz_block_put(c, f.block); uz_block_put(c, f.wide_block); throw; }
        const int ks = new_slot(c->sites);
        c->sites[ks] = s;
        f.live = true; f.sites_id = ks; f.pending = true;
        const int kf = new_slot(c->fams);
        c->fams[kf] = f;
        *sites_id = ks; *fam_id = kf;
    });
}

int uz_family_adopt_device(uz_ctx *c, int sites_id, const uz_family_view *v, int *id) {
    return guarded(c, [&] {
        UZ_REQUIRE(v && id, UZ_E_ARG, "bad family view");
        UZ_REQUIRE(!family_view_eight(v), UZ_E_ARG, "uz_family_adopt_device takes the 16-bit columns (the kernels read them in place)");
        FamilyDev f;
        f.owned = false;
        f.gt = const_cast<uint8_t *>(v->gt); // bit 6 is written by the library (complex flag)
        for (int m = 0; m < 3; m++) {
            f.rd[m] = const_cast<uint16_t *>(v->ref_depth[m]);
            f.ad[m] = const_cast<uint16_t *>(v->alt_depth[m]);
            f.gq[m] = const_cast<uint16_t *>(v->gq[m]);
            UZ_REQUIRE(((uintptr_t)f.rd[m] | (uintptr_t)f.ad[m] | (uintptr_t)f.gq[m]) % 16 == 0, UZ_E_ARG,
                       "device columns must be 16-byte aligned");
        }
        UZ_REQUIRE((uintptr_t)f.gt % 16 == 0, UZ_E_ARG, "device columns must be 16-byte aligned");
        {
            SitesDev &s = sites_of(c, sites_id);
            for (int pass = 0; pass < 2; pass++) {
                Carver cv(pass ? f.block.p : nullptr);
                f.cls = cv.take<uint8_t>((size_t)s.n);
                if (!pass) f.block = uz_block_get(c, cv.off + 256);
            }
        }
        try {
            family_wide(c, c->stream, v, f, sites_of(c, sites_id).n);
            family_common(c, sites_id, f);
        } catch (...) { uz_block_put(c, f.block); uz_block_put(c, f.wide_block); throw; }
        const int k = new_slot(c->fams);
        c->fams[k] = f;
        *id = k;
    });
}

} // extern "C"

// ---- alignment records -> HBM ---------------------------------------------------------------------------
// Layout of a table's block: [record headers | flag word | QC word | coarse + mid search index | contig tables] then, for
// uploads, [cigar | seq4 | qlow] (+ the full qualities of an ASCII upload) and the staged fixed-width columns
// the headers are built from.
static void carve_common(Carver &cv, ReadsDev &r) {
    const size_t n = (size_t)r.n;
    r.rec_a = cv.take<uint8_t>(n * 16);
    r.rec_b = cv.take<uint8_t>(n * 16);
    r.fm = cv.take<uint32_t>(n);
    r.qoff = cv.take<uint32_t>(n);
    r.nlow = cv.take<uint8_t>(n);
    r.umask = cv.take<uint16_t>(n);
    r.qs = cv.take<uint16_t>(n);
    r.coarse = cv.take<int32_t>((n >> 12) + 2);
    r.mid = cv.take<int32_t>((n >> 6) + 2);
    r.mid8 = cv.take<int32_t>((n >> 3) + 16); // (read eight entries at a time: uz_mid8_refine)
    r.contig_off = cv.take<int64_t>((size_t)r.n_contigs + 1);
    r.max_span = cv.take<int32_t>((size_t)r.n_contigs + 1);
}

static void check_packed_view(const uz_reads_packed_view *v) {
    UZ_REQUIRE(v->n_segs >= 0 && v->n_segs < (int64_t)0x7FFFFFF0, UZ_E_RANGE, "more than 2^31 alignment records");
    UZ_REQUIRE(v->n_contigs >= 0, UZ_E_ARG, "bad reads view");
    UZ_REQUIRE(v->n_cigar_total >= 0 && v->n_cigar_total < ((int64_t)1 << 32), UZ_E_RANGE, "more than 2^32 CIGAR operations");
    UZ_REQUIRE(v->n_row_units >= 0 && v->n_row_units < ((int64_t)1 << 32), UZ_E_RANGE, "more than 2^32 row units (2^37 bases)");
    UZ_REQUIRE(v->n_seq_units >= 0 && v->n_seq_units <= v->n_row_units, UZ_E_ARG, "n_seq_units must lie in [0, n_row_units]");
    UZ_REQUIRE(!(v->seq2 && v->seq4), UZ_E_ARG, "a packed table has four-bit (seq4) OR two-bit (seq2) base rows, not both");
    UZ_REQUIRE(v->n_seq_units == 0 || v->seq2 || v->seq4, UZ_E_ARG, "n_seq_units > 0 but neither seq4 nor seq2 is set");
    if (v->seq2) UZ_REQUIRE(v->n_exc >= 0 && (v->n_exc == 0 || (v->exc_rec && v->exc_pos && v->exc_code)), UZ_E_ARG, "bad exc_* columns");
    UZ_REQUIRE(!(v->qlow && v->n_low), UZ_E_ARG, "a packed table has the quality plane (qlow) OR its list form (n_low / qlow_pos), not both");
    const bool v_dict = v->tup != nullptr || v->tup8 != nullptr;
    const bool v_lists = v->n_low != nullptr || (v_dict && v->tup_n_low);
    UZ_REQUIRE(v->n_segs == 0 || v->qlow || v_lists, UZ_E_ARG, "neither qlow nor n_low is set");
    UZ_REQUIRE(!(v->qlow && v_lists), UZ_E_ARG, "a packed table has the quality plane (qlow) OR its list form, not both");
    if (v_lists) UZ_REQUIRE(v->n_qlow_pos >= 0 && (v->n_qlow_pos == 0 || v->qlow_pos), UZ_E_ARG, "bad qlow_pos column");
    UZ_REQUIRE(!(v->umask || v->tup_umask) || v_lists, UZ_E_ARG, "umask (staged units) needs the list form of the qualities (n_low / qlow_pos)");
    UZ_REQUIRE(!v->tup_umask || (v_dict && !v->umask), UZ_E_ARG, "tup_umask needs tup, and umask NULL");
    if (v->start_d || v->start_d8) {
        UZ_REQUIRE(!(v->start_d && v->start_d8), UZ_E_ARG, "start_d and start_d8 are both set");
        const bool n8 = v->mate_d8 != nullptr || v->qname_d8 != nullptr;
        UZ_REQUIRE(!n8 || (v->start_d8 && v->mate_d8 && v->qname_d8 && !v->mate_d && !v->qname_d), UZ_E_ARG, "mate_d8 and qname_d8 come together, with start_d8, instead of mate_d / qname_d");
        if (v->pair_d8)
            UZ_REQUIRE(v->start_d8 && !v->tlen_s && !v->mate_d && !v->qname_d && !n8, UZ_E_ARG, "pair_d8 comes with start_d8, instead of tlen_s / mate_d* / qname_d*");
        UZ_REQUIRE((v->pair_d8 || (v->tlen_s && (n8 || (v->mate_d && v->qname_d)))) && v->n_esc16 >= 0 && (v->n_esc16 == 0 || (v->esc16_key && v->esc16_val)), UZ_E_ARG, "bad 16-bit difference columns");
        UZ_REQUIRE(!v->start && !v->tlen && !v->mate && !v->qname, UZ_E_ARG, "start_d / start_d8 is set: start / tlen / mate / qname must be NULL");
    }
    if (v->tup8) { // the index in one byte: hot table, escape list, escapes in front of every span (ascending from 0 to the total)
        UZ_REQUIRE(!v->tup && v->tup_hot && v->tup_esc_off && v->n_tup_esc >= 0 && v->n_tup_esc <= v->n_segs && (v->n_tup_esc == 0 || v->tup_esc), UZ_E_ARG, "bad tup8 form");
        const int64_t nsp = (v->n_segs + UZ_TUP8_SPAN - 1) / UZ_TUP8_SPAN;
        UZ_REQUIRE(v->tup_esc_off[0] == 0 && (int64_t)v->tup_esc_off[nsp] == v->n_tup_esc, UZ_E_RANGE, "tup_esc_off does not run from 0 to n_tup_esc");
        for (int64_t b = 0; b < nsp; b++) UZ_REQUIRE(v->tup_esc_off[b] <= v->tup_esc_off[b + 1], UZ_E_RANGE, "tup_esc_off does not ascend");
    }
    if (v_dict) {
        UZ_REQUIRE(v->n_tup >= 1 && v->n_tup <= 65536 && v->tup_flag && v->tup_l_seq && v->tup_n_cigar && v->tup_mapq && v->tup_aux, UZ_E_ARG, "bad tup_* table");
        UZ_REQUIRE(!v->flag && !v->l_seq && !v->n_cigar && !v->mapq && !v->aux && !v->n_low, UZ_E_ARG, "tup is set: flag / l_seq / n_cigar / mapq / aux / n_low must be NULL");
    }
    if (v->bl_n || v->tup_n_bl) {
        UZ_REQUIRE(!(v->bl_n && v->tup_n_bl) && (!v->tup_n_bl || v_dict), UZ_E_ARG, "bl_n OR tup_n_bl (the latter with tup)");
        UZ_REQUIRE((v->umask || v->tup_umask) && v_lists && (v->seq2 || v->n_seq_units == 0) && !v->seq4, UZ_E_ARG,
                   "the list form of the bases (bl_*) needs unit masks, the list form of the qualities and two-bit rows");
        UZ_REQUIRE(v->n_bl >= 0 && v->n_bl_units >= 0 && v->n_bl_units <= v->n_bl && (v->n_bl == 0 || (v->bl_pos && v->bl_code)), UZ_E_ARG, "bad bl_* columns");
        UZ_REQUIRE(v->n_seq_units + v->n_bl_units <= v->n_row_units, UZ_E_ARG, "n_seq_units + n_bl_units must not exceed n_row_units");
    } else
        UZ_REQUIRE(v->n_bl == 0 && v->n_bl_units == 0, UZ_E_ARG, "n_bl / n_bl_units without bl_n / tup_n_bl");
    UZ_REQUIRE(v->cigar_compact ? v->n_cigar_omitted >= 0 && v->n_cigar_omitted <= v->n_segs : v->n_cigar_omitted == 0, UZ_E_ARG, "bad n_cigar_omitted");
    UZ_REQUIRE(v->n_cigar_total + v->n_cigar_omitted < ((int64_t)1 << 32), UZ_E_RANGE, "more than 2^32 CIGAR operations");
    if (v->pk_sums) { // the packer's span sums: rows that ascend from zero to the totals the view declares (the device holds every span against its row)
        const int shift = UZ_PK_SHIFT(v->n_segs);
        const int64_t nb = (v->n_segs + ((int64_t)1 << shift) - 1) >> shift;
        UZ_REQUIRE(v->n_pk_spans == nb, UZ_E_ARG, "pk_sums: n_pk_spans is not the number of spans of n_segs records (UZ_PK_SHIFT)");
        const uint64_t *S = v->pk_sums, *T = S + (size_t)nb * UZ_PK_SUMS;
        bool ok = true;
        for (int k = 0; k < UZ_PK_SUMS; k++) ok &= S[k] == 0;
        for (int64_t b = 0; b < nb && ok; b++)
            for (int k = 0; k < UZ_PK_SUMS; k++)
                if (k != 5 && k != 6) ok &= S[(size_t)(b + 1) * UZ_PK_SUMS + k] >= S[(size_t)b * UZ_PK_SUMS + k];
        const bool lists_f = v->n_low != nullptr || (v_dict && v->tup_n_low);
        ok = ok && T[0] == (uint64_t)(v->n_cigar_total + v->n_cigar_omitted) && T[1] == (uint64_t)v->n_row_units && T[2] == (uint64_t)v->n_seq_units &&
             T[3] == (uint64_t)(lists_f ? v->n_qlow_pos : 0) && (!v->cigar_compact || T[4] == (uint64_t)v->n_cigar_total) && T[7] == (uint64_t)v->n_bl_units &&
             T[8] == (uint64_t)v->n_bl && T[9] == T[10];
        UZ_REQUIRE(ok, UZ_E_RANGE, "pk_sums: the span sums do not ascend from zero to the totals the view declares");
    }
}

// packed columns in HOST memory -> one block; every command goes to stream `st`
static void reads_from_packed_host(uz_ctx *c, hipStream_t st, const uz_reads_packed_view *v, ReadsDev &r, bool defer_build) {
    check_packed_view(v);
    r.live = true;
    r.n = v->n_segs; r.n_contigs = v->n_contigs; r.n_qnames = v->n_qnames;
    r.n_cigar_total = v->n_cigar_total + v->n_cigar_omitted; // the device's store holds every record's words
    const bool blf = v->bl_n != nullptr || v->tup_n_bl != nullptr; // bases of some records as lists: their units exist on the device only, behind the rows that travelled
    r.n_bl = blf ? v->n_bl : 0; r.n_bl_units = blf ? v->n_bl_units : 0;
    r.n_row_units = v->n_row_units; r.n_seq_units = v->n_seq_units + r.n_bl_units;
    r.n_cigar_staged = v->n_cigar_total;
    const bool ccompact = v->cigar_compact != 0;
    const size_t n = (size_t)r.n, nc = (size_t)r.n_cigar_total, nu = (size_t)r.n_row_units, ns = (size_t)r.n_seq_units, ncs = (size_t)v->n_cigar_total;
    const size_t nsl = (size_t)v->n_seq_units; // units on the link
    const size_t nbl = (size_t)r.n_bl, nblp = nbl * (v->bl_wide ? 2 : 1), nblc = (nbl + 3) / 4;
    uint8_t *bl_n = nullptr, *t_nbl = nullptr, *bl_pos = nullptr, *bl_code = nullptr;
    uint32_t *cigar_staged = nullptr;
    const bool two_bit = v->seq2 != nullptr;
    const size_t ne = two_bit ? (size_t)v->n_exc : 0;
    const bool t8 = v->tup8 != nullptr; // the dictionary index in one byte: rebuilt into the 16-bit column's place by the header build's first kernel
    const bool tupf = v->tup != nullptr || t8;
    const bool lists = v->n_low != nullptr || (tupf && v->tup_n_low); // quality rows only for the records with bases (at their base-row position), written by the header build
    const size_t nt = tupf ? (size_t)v->n_tup : 0;
    const size_t nte = t8 ? (size_t)v->n_tup_esc : 0, ntsp = t8 ? ((size_t)v->n_segs + UZ_TUP8_SPAN - 1) / UZ_TUP8_SPAN + 1 : 0;
    uint8_t *d_tup8 = nullptr; uint16_t *t_hot = nullptr, *t_esc = nullptr; uint32_t *t_eoff = nullptr;
    uint16_t *tup = nullptr, *t_flag = nullptr, *t_ls = nullptr, *t_nc = nullptr, *t_um = nullptr; uint8_t *t_mq = nullptr, *t_ax = nullptr, *t_nl = nullptr;
    const size_t nql = lists ? (size_t)v->n_qlow_pos * (v->qlow_pos_wide ? 2 : 1) : 0;
    uint8_t *n_low = nullptr, *qpos = nullptr;
    uint16_t *umask_in = nullptr;
    r.n_qlow_pos = lists ? v->n_qlow_pos : 0;
    r.n_plane_units = lists ? r.n_seq_units : v->n_row_units;
    uint32_t *cigar = nullptr; uint8_t *seq4 = nullptr, *qlow = nullptr, *seq2 = nullptr;
    uint32_t *exc_rec = nullptr; uint16_t *exc_pos = nullptr; uint8_t *exc_code = nullptr;
    int32_t *start, *end, *tlen, *mate; uint32_t *qname; uint16_t *flag, *l_seq, *n_cigar; uint8_t *mapq, *aux;
    const bool d8 = v->start_d8 != nullptr;
    const bool n8 = v->mate_d8 != nullptr;
    const bool p8 = v->pair_d8 != nullptr;
    const bool d16 = v->start_d != nullptr || d8;
    const size_t nes = d16 ? (size_t)v->n_esc16 : 0;
    int16_t *d_start = nullptr, *d_tlen = nullptr, *d_mate = nullptr, *d_qname = nullptr;
    uint8_t *d_start8 = nullptr;
    int8_t *d_mate8 = nullptr, *d_qname8 = nullptr;
    uint8_t *d_pair8 = nullptr;
    unsigned long long *e_key = nullptr; int32_t *e_val = nullptr;
    unsigned long long *pk = nullptr;
    const size_t npk = v->pk_sums ? ((size_t)v->n_pk_spans + 1) * UZ_PK_SUMS : 0; // the packer's span sums (checked below)
    void *scratch = nullptr;
    for (int pass = 0; pass < 2; pass++) {
        Carver cv(pass ? r.block.p : nullptr);
        carve_common(cv, r);
        cigar = cv.take<uint32_t>(nc);
        if (ccompact) cigar_staged = cv.take<uint32_t>(ncs);
        if (d16) {
            d_start = cv.take<int16_t>(d8 ? 0 : n); d_start8 = cv.take<uint8_t>(d8 ? n : 0);
            d_tlen = cv.take<int16_t>(p8 ? 0 : n); d_mate = cv.take<int16_t>(n8 || p8 ? 0 : n); d_qname = cv.take<int16_t>(n8 || p8 ? 0 : n);
            d_mate8 = cv.take<int8_t>(n8 ? n : 0); d_qname8 = cv.take<int8_t>(n8 ? n : 0); d_pair8 = cv.take<uint8_t>(p8 ? n : 0);
            e_key = cv.take<unsigned long long>(nes); e_val = cv.take<int32_t>(nes);
        }
        seq4 = cv.take<uint8_t>(ns * UZ_SEQ4_UNIT_BYTES);
        qlow = cv.take<uint8_t>((lists ? ns : nu) * UZ_QLOW_UNIT_BYTES);
        if (lists) { n_low = cv.take<uint8_t>(tupf ? 0 : n); qpos = cv.take<uint8_t>(nql); }
        if (tupf) {
            tup = cv.take<uint16_t>(n); t_flag = cv.take<uint16_t>(nt); t_ls = cv.take<uint16_t>(nt); t_nc = cv.take<uint16_t>(nt);
            t_mq = cv.take<uint8_t>(nt); t_ax = cv.take<uint8_t>(nt); t_nl = cv.take<uint8_t>(nt); t_um = cv.take<uint16_t>(nt);
            if (t8) { d_tup8 = cv.take<uint8_t>(n); t_hot = cv.take<uint16_t>(256); t_esc = cv.take<uint16_t>(nte); t_eoff = cv.take<uint32_t>(ntsp); }
        }
        if (v->umask) umask_in = cv.take<uint16_t>(n);
        if (blf) { bl_n = cv.take<uint8_t>(v->bl_n ? n : 0); t_nbl = cv.take<uint8_t>(v->tup_n_bl ? nt : 0); bl_pos = cv.take<uint8_t>(nblp); bl_code = cv.take<uint8_t>(nblc + 4); }
        if (two_bit) {
            seq2 = cv.take<uint8_t>(nsl * UZ_SEQ2_UNIT_BYTES);
            exc_rec = cv.take<uint32_t>(ne); exc_pos = cv.take<uint16_t>(ne); exc_code = cv.take<uint8_t>(ne);
        }
        const size_t nw = d16 ? 0 : n; // the plain wide columns
        start = cv.take<int32_t>(nw); end = cv.take<int32_t>(n); tlen = cv.take<int32_t>(nw); mate = cv.take<int32_t>(nw);
        qname = cv.take<uint32_t>(nw);
        const size_t npl = tupf ? 0 : n; // the plain small columns
        flag = cv.take<uint16_t>(npl); l_seq = cv.take<uint16_t>(npl); n_cigar = cv.take<uint16_t>(npl);
        mapq = cv.take<uint8_t>(npl); aux = cv.take<uint8_t>(npl);
        if (npk) pk = cv.take<unsigned long long>(npk);
        scratch = cv.take<uint8_t>(uz_rec_scratch_bytes(r.n));
        if (!pass) r.block = uz_block_get(c, cv.off + 256);
    }
    RecColumns col;
    SlabPlan plan;
    SlabScope slab_scope(&plan);
    for (int stage_pass = 0; stage_pass < 2; stage_pass++) {
    if (stage_pass) slab_commit(c, st, plan, r.mirror);
    col = RecColumns();
    r.contig_off = const_cast<int64_t *>(h2d(st, r.contig_off, v->contig_off, (size_t)v->n_contigs + 1));
    r.max_span = const_cast<int32_t *>(h2d(st, r.max_span, v->max_span, (size_t)v->n_contigs));
    col.end = v->end ? h2d(st, end, v->end, n) : nullptr;
    if (npk) col.pk_sums = h2d(st, pk, (const unsigned long long *)v->pk_sums, npk);
    if (d16) { // 16-bit differences + the escape list instead of four 32-bit columns
        if (d8) col.start_d8 = h2d(st, d_start8, v->start_d8, n);
        else col.start_d = h2d(st, d_start, v->start_d, n);
        if (p8) col.pair_d8 = h2d(st, d_pair8, v->pair_d8, n);
        else {
            col.tlen_s = h2d(st, d_tlen, v->tlen_s, n);
            if (n8) { col.mate_d8 = h2d(st, d_mate8, v->mate_d8, n); col.qname_d8 = h2d(st, d_qname8, v->qname_d8, n); }
            else { col.mate_d = h2d(st, d_mate, v->mate_d, n); col.qname_d = h2d(st, d_qname, v->qname_d, n); }
        }
        col.esc16_key = h2d(st, e_key, (const unsigned long long *)v->esc16_key, nes); col.esc16_val = h2d(st, e_val, v->esc16_val, nes);
        col.n_esc16 = (int64_t)nes;
        col.start = nullptr; col.tlen = nullptr; col.mate = nullptr; col.qname = nullptr;
    } else {
        col.start = h2d(st, start, v->start, n); col.tlen = h2d(st, tlen, v->tlen, n);
        col.mate = h2d(st, mate, v->mate, n); col.qname = h2d(st, qname, v->qname, n);
    }
    if (tupf) { // a 16-bit index per record + the table of combinations instead of nine bytes of small columns
        if (t8) {
            col.tup8 = h2d(st, d_tup8, v->tup8, n); col.tup_hot = h2d(st, t_hot, v->tup_hot, (size_t)256);
            col.tup_esc = nte ? h2d(st, t_esc, v->tup_esc, nte) : t_esc; col.tup_esc_off = h2d(st, t_eoff, v->tup_esc_off, ntsp);
            col.n_tup_esc = (int64_t)nte;
            col.tup_out = tup; col.tup = tup; // (k_tup_expand writes it before anything reads it)
        } else
            col.tup = h2d(st, tup, v->tup, n);
        col.n_tup = (int64_t)nt;
        col.tup_flag = h2d(st, t_flag, v->tup_flag, nt); col.tup_l_seq = h2d(st, t_ls, v->tup_l_seq, nt); col.tup_n_cigar = h2d(st, t_nc, v->tup_n_cigar, nt);
        col.tup_mapq = h2d(st, t_mq, v->tup_mapq, nt); col.tup_aux = h2d(st, t_ax, v->tup_aux, nt);
        if (v->tup_n_low) col.tup_n_low = h2d(st, t_nl, v->tup_n_low, nt);
        if (v->tup_umask) col.tup_umask = h2d(st, t_um, v->tup_umask, nt);
        if (v->tup_n_bl) col.tup_n_bl = h2d(st, t_nbl, v->tup_n_bl, nt);
    } else {
        col.flag = h2d(st, flag, v->flag, n);
        col.l_seq = h2d(st, l_seq, v->l_seq, n); col.n_cigar = h2d(st, n_cigar, v->n_cigar, n);
        col.mapq = h2d(st, mapq, v->mapq, n); col.aux = h2d(st, aux, v->aux, n);
    }
    col.lists = lists ? 1 : 0;
    if (ccompact) { // the travelled words land in a staging area; the header build writes the store
        r.cigar = cigar;
        col.cigar_staged = h2d(st, cigar_staged, v->cigar, ncs);
        col.cigar_out = cigar;
    } else
        r.cigar = h2d(st, cigar, v->cigar, nc);
    col.cigar_in = r.cigar;
    if (two_bit) { // half the bytes over the link; the header build expands them into seq4
        r.seq4 = seq4;
        r.seq2_staged = h2d(st, seq2, v->seq2, nsl * UZ_SEQ2_UNIT_BYTES);
        r.n_exc = (int64_t)ne;
        r.exc_rec = h2d(st, exc_rec, v->exc_rec, ne); r.exc_pos = h2d(st, exc_pos, v->exc_pos, ne); r.exc_code = h2d(st, exc_code, v->exc_code, ne);
    } else
        r.seq4 = h2d(st, seq4, v->seq4, ns * UZ_SEQ4_UNIT_BYTES);
    if (blf) {
        if (v->bl_n) col.bl_n = h2d(st, bl_n, v->bl_n, n);
        col.bl_pos = h2d(st, bl_pos, v->bl_pos, nblp); col.bl_code = h2d(st, bl_code, v->bl_code, nblc);
        col.bl_wide = v->bl_wide;
        col.n_seq_link = (int64_t)nsl;
        col.seq4_out = reinterpret_cast<uint32_t *>(seq4);
        r.seq4 = seq4;
    }
    r.qlow = qlow;
    if (lists) {
        if (!tupf) col.n_low = h2d(st, n_low, v->n_low, n);
        col.qlow_pos = h2d(st, qpos, v->qlow_pos, nql);
        col.qpos_wide = v->qlow_pos_wide;
        if (v->umask) col.umask = h2d(st, umask_in, v->umask, n);
    } else {
        r.qlow = const_cast<uint8_t *>(h2d(st, qlow, v->qlow, nu * UZ_QLOW_UNIT_BYTES));
        col.plane_in = reinterpret_cast<const uint32_t *>(r.qlow);
    }
    } // stage_pass
    r.qlow_thr = v->min_base_qual;
    r.qlow_valid = true;
    r.col_q[0] = col.plane_in; r.col_q[1] = col.n_low; r.col_q[2] = col.qlow_pos; r.col_q[3] = col.cigar_in; r.col_q[4] = col.umask; r.col_q[5] = col.cigar_staged; r.col_q[6] = col.cigar_out; r.col_qwide = col.qpos_wide;
    {
        const void *t[8] = {col.tup, col.tup_flag, col.tup_l_seq, col.tup_n_cigar, col.tup_mapq, col.tup_aux, col.tup_n_low, col.tup_umask};
        for (int k = 0; k < 8; k++) r.col_t[k] = t[k];
        r.col_t8[0] = col.tup8; r.col_t8[1] = col.tup_hot; r.col_t8[2] = col.tup_esc; r.col_t8[3] = col.tup_esc_off; r.col_ntesc = col.n_tup_esc;
        r.col_lists = col.lists;
        r.col_ntup = col.n_tup;
        const void *dd[10] = {col.start_d, col.tlen_s, col.mate_d, col.qname_d, col.esc16_key, col.esc16_val, col.start_d8, col.mate_d8, col.qname_d8, col.pair_d8};
        for (int k = 0; k < 10; k++) r.col_d[k] = dd[k];
        r.col_nesc = col.n_esc16;
        const void *bb[4] = {col.bl_n, col.tup_n_bl, col.bl_pos, col.bl_code};
        for (int k = 0; k < 4; k++) r.col_b[k] = bb[k];
        r.col_bwide = col.bl_wide;
        r.col_pk = col.pk_sums;
    }
    if (defer_build) { // asynchronous upload: copies only on the copy stream, the header build at first use (uz_reads_make_ready)
        const void *p[10] = {col.start, col.end, col.tlen, col.mate, col.qname, col.flag, col.l_seq, col.n_cigar, col.mapq, col.aux};
        for (int k = 0; k < 10; k++) r.col_ptrs[k] = p[k];
        r.build_scratch = scratch;
        return;
    }
    uz_build_records(c, st, r, col, scratch);
}

// the header build of an asynchronously uploaded table, from the staged columns the upload left in its block
static void build_staged(uz_ctx *c, hipStream_t st, ReadsDev &r) {
    RecColumns col;
    col.start = (const int32_t *)r.col_ptrs[0]; col.end = (const int32_t *)r.col_ptrs[1]; col.tlen = (const int32_t *)r.col_ptrs[2];
    col.mate = (const int32_t *)r.col_ptrs[3]; col.qname = (const uint32_t *)r.col_ptrs[4]; col.flag = (const uint16_t *)r.col_ptrs[5];
    col.l_seq = (const uint16_t *)r.col_ptrs[6]; col.n_cigar = (const uint16_t *)r.col_ptrs[7]; col.mapq = (const uint8_t *)r.col_ptrs[8];
    col.aux = (const uint8_t *)r.col_ptrs[9];
    col.plane_in = (const uint32_t *)r.col_q[0]; col.n_low = (const uint8_t *)r.col_q[1]; col.qlow_pos = (const uint8_t *)r.col_q[2];
    col.cigar_in = (const uint32_t *)r.col_q[3]; col.umask = (const uint16_t *)r.col_q[4];
    col.cigar_staged = (const uint32_t *)r.col_q[5]; col.cigar_out = (uint32_t *)const_cast<void *>(r.col_q[6]);
    col.tup = (const uint16_t *)r.col_t[0]; col.tup_flag = (const uint16_t *)r.col_t[1]; col.tup_l_seq = (const uint16_t *)r.col_t[2];
    col.tup_n_cigar = (const uint16_t *)r.col_t[3]; col.tup_mapq = (const uint8_t *)r.col_t[4]; col.tup_aux = (const uint8_t *)r.col_t[5];
    col.tup_n_low = (const uint8_t *)r.col_t[6]; col.tup_umask = (const uint16_t *)r.col_t[7]; col.lists = r.col_lists; col.n_tup = r.col_ntup;
    col.tup8 = (const uint8_t *)r.col_t8[0]; col.tup_hot = (const uint16_t *)r.col_t8[1]; col.tup_esc = (const uint16_t *)r.col_t8[2];
    col.tup_esc_off = (const uint32_t *)r.col_t8[3]; col.n_tup_esc = r.col_ntesc;
    if (col.tup8) col.tup_out = const_cast<uint16_t *>(col.tup);
    col.start_d = (const int16_t *)r.col_d[0]; col.tlen_s = (const int16_t *)r.col_d[1]; col.mate_d = (const int16_t *)r.col_d[2];
    col.qname_d = (const int16_t *)r.col_d[3]; col.esc16_key = (const unsigned long long *)r.col_d[4]; col.esc16_val = (const int32_t *)r.col_d[5];
    col.start_d8 = (const uint8_t *)r.col_d[6];
    col.mate_d8 = (const int8_t *)r.col_d[7]; col.qname_d8 = (const int8_t *)r.col_d[8]; col.pair_d8 = (const uint8_t *)r.col_d[9];
    col.n_esc16 = r.col_nesc;
    col.qpos_wide = r.col_qwide;
    col.bl_n = (const uint8_t *)r.col_b[0]; col.tup_n_bl = (const uint8_t *)r.col_b[1]; col.bl_pos = (const uint8_t *)r.col_b[2]; col.bl_code = (const uint8_t *)r.col_b[3];
    col.bl_wide = r.col_bwide;
    col.pk_sums = (const unsigned long long *)r.col_pk;
    if (col.bl_form()) { col.n_seq_link = r.n_seq_units - r.n_bl_units; col.seq4_out = reinterpret_cast<uint32_t *>(const_cast<uint8_t *>(r.seq4)); }
    uz_build_records(c, st, r, col, r.build_scratch);
}
void uz_reads_make_ready(uz_ctx *c, ReadsDev &r) {
    if (!r.pending) return;
    if (r.built) UZ_HIP(hipStreamWaitEvent(c->stream, r.built, 0)); // built beside whatever the compute stream was doing
    else {
        UZ_HIP(hipStreamWaitEvent(c->stream, r.ready, 0));
        build_staged(c, c->stream, r);
    }
    r.pending = false;
}

int uz_reads_upload_impl(uz_ctx *c, const uz_reads_view *v, ReadsDev &r) {
    // the ASCII form (uz_reads_view): the fixed-width columns are staged as they are, the CIGAR words and
    // the bases are re-laid in the packed geometry on the device, the qualities are kept for the plane
    UZ_REQUIRE(v->n_segs >= 0 && v->n_segs < (int64_t)0x7FFFFFF0, UZ_E_RANGE, "more than 2^31 alignment records");
    hipStream_t st = c->stream;
    r.live = true;
    r.n = v->n_segs; r.n_contigs = v->n_contigs; r.n_qnames = v->n_qnames;
    const size_t n = (size_t)r.n;
    // totals of the packed geometry from the host columns
    int64_t tot_c = 0, tot_u = 0;
    for (size_t i = 0; i < n; i++) { tot_c += v->n_cigar[i]; tot_u += UZ_ROW_UNITS(v->l_seq[i]); }
    r.n_cigar_total = tot_c; r.n_row_units = tot_u; r.n_seq_units = tot_u; // the ASCII form carries every record's bases
    r.n_plane_units = tot_u;
    UZ_REQUIRE(tot_c < ((int64_t)1 << 32) && tot_u < ((int64_t)1 << 32), UZ_E_RANGE, "table exceeds the 32-bit CIGAR / row offsets");
    const size_t nc = (size_t)tot_c, nu = (size_t)tot_u, nci = (size_t)v->n_cigar_total, nsq = (size_t)v->n_sq_bytes;
    uint32_t *cigar = nullptr, *cigar_in = nullptr, *cigar_off_in = nullptr; uint8_t *seq4 = nullptr, *qlow = nullptr, *seq_in = nullptr;
    int32_t *start, *end, *tlen, *mate; uint32_t *qname; uint16_t *flag, *l_seq, *n_cigar; uint8_t *mapq, *aux;
    void *scratch = nullptr;
    for (int pass = 0; pass < 2; pass++) {
        Carver cv(pass ? r.block.p : nullptr);
        carve_common(cv, r);
        cigar = cv.take<uint32_t>(nc);
        seq4 = cv.take<uint8_t>(nu * UZ_SEQ4_UNIT_BYTES);
        qlow = cv.take<uint8_t>(nu * UZ_QLOW_UNIT_BYTES);
        r.qual8 = cv.take<uint8_t>(nsq);
        r.qual_off16 = cv.take<uint32_t>(n);
        start = cv.take<int32_t>(n); end = cv.take<int32_t>(n); tlen = cv.take<int32_t>(n); mate = cv.take<int32_t>(n);
        qname = cv.take<uint32_t>(n); flag = cv.take<uint16_t>(n); l_seq = cv.take<uint16_t>(n); n_cigar = cv.take<uint16_t>(n);
        mapq = cv.take<uint8_t>(n); aux = cv.take<uint8_t>(n);
        cigar_in = cv.take<uint32_t>(nci); cigar_off_in = cv.take<uint32_t>(n); seq_in = cv.take<uint8_t>(nsq);
        scratch = cv.take<uint8_t>(uz_rec_scratch_bytes(r.n));
        if (!pass) r.block = uz_block_get(c, cv.off + 256);
    }
    h2d(st, r.contig_off, v->contig_off, (size_t)v->n_contigs + 1);
    h2d(st, r.max_span, v->max_span, (size_t)v->n_contigs);
    RecColumns col;
    col.start = h2d(st, start, v->start, n); col.end = h2d(st, end, v->end, n); col.tlen = h2d(st, tlen, v->tlen, n);
    col.mate = h2d(st, mate, v->mate, n); col.qname = h2d(st, qname, v->qname, n); col.flag = h2d(st, flag, v->flag, n);
    col.l_seq = h2d(st, l_seq, v->l_seq, n); col.n_cigar = h2d(st, n_cigar, v->n_cigar, n);
    col.mapq = h2d(st, mapq, v->mapq, n); col.aux = h2d(st, aux, v->aux, n);
    h2d(st, cigar_in, v->cigar, nci); h2d(st, cigar_off_in, v->cigar_off, n);
    h2d(st, seq_in, v->seq, nsq); h2d(st, r.qual8, v->qual, nsq); h2d(st, r.qual_off16, v->sq_off16, n);
    r.cigar = cigar; r.seq4 = seq4; r.qlow = qlow;
    r.qlow_valid = false; // built for the threshold of the first uz_phase
    uz_build_records(c, st, r, col, scratch);
    uz_pack_ascii_rows(c, st, r, cigar_in, cigar_off_in, seq_in, r.qual_off16, cigar, seq4);
    UZ_HIP(hipStreamSynchronize(st));
    if (c->hflags[0]) {
        const int f = c->hflags[0];
        c->hflags[0] = 0;
        throw UzError{UZ_E_RANGE, f == 2 ? "SEQ holds a character outside BAM's 16-code alphabet" : "inconsistent reads view"};
    }
    return 0;
}

extern "C" {

int uz_reads_upload(uz_ctx *c, const uz_reads_view *v, int *id) {
    return guarded(c, [&] {
        UZ_REQUIRE(v && id, UZ_E_ARG, "bad reads view");
        ReadsDev r;
        try { uz_reads_upload_impl(c, v, r); } catch (...) { uz_block_put(c, r.block); throw; }
        const int k = new_slot(c->reads);
        c->reads[k] = r;
        *id = k;
    });
}

int uz_reads_upload_packed(uz_ctx *c, const uz_reads_packed_view *v, int *id) {
    return guarded(c, [&] {
        UZ_REQUIRE(v && id, UZ_E_ARG, "bad reads view");
        ReadsDev r;
        try {
            reads_from_packed_host(c, c->copy_stream, v, r, true);
            UZ_HIP(hipEventCreateWithFlags(&r.ready, hipEventDisableTiming));
            UZ_HIP(hipEventRecord(r.ready, c->copy_stream));
            r.pending = true;
            static const bool lazy = getenv("UZ_BUILD_LAZY") != nullptr; // development aid: the header build at first use, on the compute stream
            if (!lazy) {
                if (!c->build_stream) UZ_HIP(hipStreamCreateWithFlags(&c->build_stream, hipStreamNonBlocking));
                UZ_HIP(hipStreamWaitEvent(c->build_stream, r.ready, 0));
                build_staged(c, c->build_stream, r);
                UZ_HIP(hipEventCreateWithFlags(&r.built, hipEventDisableTiming));
                UZ_HIP(hipEventRecord(r.built, c->build_stream));
            }
        } catch (...) {
            if (r.built) { (void)hipEventSynchronize(r.built); (void)hipEventDestroy(r.built); }
            if (r.ready) { (void)hipEventSynchronize(r.ready); (void)hipEventDestroy(r.ready); }
            uz_block_put(c, r.block); uz_block_put(c, r.mirror); throw;
        }
        const int k = new_slot(c->reads);
        c->reads[k] = r;
        *id = k;
    });
}

int uz_reads_wait(uz_ctx *c, int reads_id) {
    return guarded(c, [&] {
        ReadsDev &r = reads_of(c, reads_id);
        uz_reads_make_ready(c, r);
        UZ_HIP(hipStreamSynchronize(c->stream));
        if (c->hflags[0]) {
            c->hflags[0] = 0;
            throw UzError{UZ_E_RANGE, "n_cigar_total / n_row_units of the reads view do not match its columns"};
        }
    });
}

int uz_reads_headers(uz_ctx *c, int reads_id, int32_t *start, int32_t *end, int32_t *tlen, int32_t *mate, uint32_t *qname) {
    return guarded(c, [&] {
        ReadsDev &r = reads_of(c, reads_id);
        uz_reads_make_ready(c, r);
        UZ_HIP(hipStreamSynchronize(c->stream));
        if (c->hflags[0]) {
            c->hflags[0] = 0;
            throw UzError{UZ_E_RANGE, "inconsistent reads view"};
        }
        struct HA { int32_t start, end; uint32_t cigar_off, sq_off; }; // the record headers (RecA / RecB of phase_body.hpp)
        struct HB { int32_t mate; uint32_t qname; uint16_t l_seq, n_cigar; int32_t tlen; };
        const size_t n = (size_t)r.n;
        std::vector<HA> a(n);
        std::vector<HB> b(n);
        if (n) {
            UZ_HIP(hipMemcpy(a.data(), r.rec_a, n * sizeof(HA), hipMemcpyDeviceToHost));
            UZ_HIP(hipMemcpy(b.data(), r.rec_b, n * sizeof(HB), hipMemcpyDeviceToHost));
        }
        for (size_t i = 0; i < n; i++) {
            if (start) start[i] = a[i].start;
            if (end) end[i] = a[i].end;
            if (tlen) tlen[i] = b[i].tlen;
            if (mate) mate[i] = b[i].mate;
            if (qname) qname[i] = b[i].qname;
        }
    });
}

int uz_bgzf_inflate(uz_ctx *c, const uint8_t *comp, int64_t comp_bytes, int64_t n_blocks, const int64_t *in_off, const int64_t *out_off, uint8_t *out,
                    int repeat, double *kernel_ms) {
    return guarded(c, [&] {
        UZ_REQUIRE(comp && in_off && out_off && out && n_blocks >= 0 && comp_bytes >= 0, UZ_E_ARG, "bad arguments");
        if (kernel_ms) *kernel_ms = 0;
        if (n_blocks == 0) return;
        const int64_t out_bytes = out_off[n_blocks];
        UZ_REQUIRE(out_off[0] >= 0, UZ_E_ARG, "bad block table");
        for (int64_t k = 0; k < n_blocks; k++)
            UZ_REQUIRE(in_off[k] >= 0 && in_off[k] < comp_bytes && out_off[k] <= out_off[k + 1] && out_off[k + 1] - out_off[k] <= 65536, UZ_E_ARG,
                       "bad block table (a BGZF block inflates to at most 64 KiB)");
        uint8_t *d_comp = nullptr, *d_out = nullptr;
        int64_t *d_in = nullptr, *d_off = nullptr;
        int32_t *d_flags = nullptr;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        auto cleanup = [&] {
            (void)hipFree(d_comp); (void)hipFree(d_out); (void)hipFree(d_in); (void)hipFree(d_off); (void)hipFree(d_flags);
            if (e0) (void)hipEventDestroy(e0);
            if (e1) (void)hipEventDestroy(e1);
        };
        try {
            UZ_HIP(hipMalloc((void **)&d_comp, (size_t)comp_bytes + 1024));
            UZ_HIP(hipMalloc((void **)&d_out, (size_t)out_bytes + 64));
            UZ_HIP(hipMalloc((void **)&d_in, (size_t)n_blocks * 8));
            UZ_HIP(hipMalloc((void **)&d_off, (size_t)(n_blocks + 1) * 8));
            UZ_HIP(hipMalloc((void **)&d_flags, 64));
            UZ_HIP(hipMemsetAsync(d_comp + comp_bytes, 0, 1024, c->stream));
            UZ_HIP(hipMemcpyAsync(d_comp, comp, (size_t)comp_bytes, hipMemcpyHostToDevice, c->stream));
            UZ_HIP(hipMemcpyAsync(d_in, in_off, (size_t)n_blocks * 8, hipMemcpyHostToDevice, c->stream));
            UZ_HIP(hipMemcpyAsync(d_off, out_off, (size_t)(n_blocks + 1) * 8, hipMemcpyHostToDevice, c->stream));
            UZ_HIP(hipEventCreate(&e0));
            UZ_HIP(hipEventCreate(&e1));
            uz_launch_inflate(c, c->stream, n_blocks, d_comp, (comp_bytes + 1024) & ~(int64_t)3, d_in, d_off, d_out, d_flags); // (warm-up and the run that is checked)
            UZ_HIP(hipEventRecord(e0, c->stream));
            for (int r = 0; r < std::max(repeat, 0); r++) uz_launch_inflate(c, c->stream, n_blocks, d_comp, (comp_bytes + 1024) & ~(int64_t)3, d_in, d_off, d_out, d_flags);
            UZ_HIP(hipEventRecord(e1, c->stream));
            int32_t flags[2] = {0, 0};
            UZ_HIP(hipMemcpyAsync(flags, d_flags, sizeof(flags), hipMemcpyDeviceToHost, c->stream));
            UZ_HIP(hipMemcpyAsync(out, d_out, (size_t)out_bytes, hipMemcpyDeviceToHost, c->stream));
            UZ_HIP(hipStreamSynchronize(c->stream));
            if (kernel_ms && repeat > 0) {
                float ms = 0;
                UZ_HIP(hipEventElapsedTime(&ms, e0, e1));
                *kernel_ms = (double)ms / repeat;
            }
            if (flags[1]) throw UzError{UZ_E_RANGE, "BGZF block " + std::to_string(flags[1] >> 4) + ": not a valid DEFLATE stream of the declared size (code " + std::to_string(flags[1] & 15) + ")"};
        } catch (...) { cleanup(); throw; }
        cleanup();
    });
}

int uz_bgzf_inflate_to_host(uz_ctx *c, const uint8_t *comp, int64_t comp_bytes, int64_t n_blocks, const int64_t *in_off, const int64_t *out_off,
                            uint8_t *out) {
    return guarded(c, [&] {
        UZ_REQUIRE(n_blocks >= 0 && comp_bytes >= 0 && (n_blocks == 0 || (comp && in_off && out_off && out)), UZ_E_ARG, "bad arguments");
        if (n_blocks == 0) return;
        const int64_t out_bytes = out_off[n_blocks];
        UZ_REQUIRE(out_bytes >= 0 && out_off[0] >= 0, UZ_E_ARG, "bad block table");
        for (int64_t k = 0; k < n_blocks; k++)
            UZ_REQUIRE(in_off[k] >= 0 && in_off[k] < comp_bytes && (k == 0 || in_off[k] > in_off[k - 1]) && out_off[k] <= out_off[k + 1] &&
                           out_off[k + 1] - out_off[k] <= 65536,
                       UZ_E_ARG, "bad block table (blocks in the order they lie in `comp`; a BGZF block inflates to at most 64 KiB)");
        UZ_HIP(hipSetDevice(c->device)); // (a decoder's worker thread calls this: the current device is per thread)
        if (!c->inf_stream) {
            UZ_HIP(hipStreamCreateWithFlags(&c->inf_stream, hipStreamNonBlocking));
            UZ_HIP(hipStreamCreateWithFlags(&c->inf_stream2, hipStreamNonBlocking));
            UZ_HIP(hipEventCreateWithFlags(&c->inf_ready, hipEventDisableTiming));
        }
        // The batch in slices of ~8 k blocks, alternating between two streams: the blocks of slice i + 1 go up and are inflated
        // while the bytes of slice i come down (the two directions of the link and the kernel overlap: the call takes about as long as
        // its slowest leg, the way down).  The blocks lie in `comp` in the order of the table, so a slice is a contiguous piece of it.
        std::vector<int64_t> cut{0};
        {
            const int64_t SLICE = 8192; // blocks: a wave is one block's decoder, and the chip holds 6 000 ... 8 000 waves -- a smaller launch leaves slots idle
            for (int64_t k = 1; k <= n_blocks; k++)
                if (k == n_blocks || (k - cut.back() >= SLICE && n_blocks - k >= SLICE / 2)) cut.push_back(k);
        }
        const size_t ns = cut.size() - 1;
        c->inf_comp.ensure((size_t)comp_bytes + 1024); c->inf_out.ensure((size_t)out_bytes + 64);
        c->inf_in.ensure((size_t)n_blocks); c->inf_off.ensure((size_t)n_blocks + 1); c->inf_flags.ensure(2 * ns + 2);
        hipStream_t st[2] = {c->inf_stream, c->inf_stream2};
        UZ_HIP(hipMemsetAsync(c->inf_comp.p + comp_bytes, 0, 1024, st[0]));
        UZ_HIP(hipMemcpyAsync(c->inf_in.p, in_off, (size_t)n_blocks * 8, hipMemcpyHostToDevice, st[0]));
        UZ_HIP(hipMemcpyAsync(c->inf_off.p, out_off, (size_t)(n_blocks + 1) * 8, hipMemcpyHostToDevice, st[0]));
        UZ_HIP(hipEventRecord(c->inf_ready, st[0]));
        UZ_HIP(hipStreamWaitEvent(st[1], c->inf_ready, 0));
        std::vector<int32_t> flags(2 * ns, 0);
        for (size_t i = 0; i < ns; i++) {
            hipStream_t s = st[i & 1];
            const int64_t b0 = cut[i], b1 = cut[i + 1];
            // (whole blocks travel: a slice's bytes run from the framing of its first block to the start of the next slice's)
            const int64_t c0 = i == 0 ? 0 : in_off[b0] - 18 < 0 ? 0 : in_off[b0] - 18, c1 = i + 1 == ns ? comp_bytes : std::max<int64_t>(in_off[b1] - 18, c0);
            UZ_HIP(hipMemcpyAsync(c->inf_comp.p + c0, comp + c0, (size_t)(c1 - c0), hipMemcpyHostToDevice, s));
            uz_launch_inflate(c, s, b1 - b0, c->inf_comp.p, (comp_bytes + 1024) & ~(int64_t)3, c->inf_in.p + b0, c->inf_off.p + b0, c->inf_out.p, c->inf_flags.p + 2 * i);
            UZ_HIP(hipMemcpyAsync(flags.data() + 2 * i, c->inf_flags.p + 2 * i, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
            UZ_HIP(hipMemcpyAsync(out + out_off[b0], c->inf_out.p + out_off[b0], (size_t)(out_off[b1] - out_off[b0]), hipMemcpyDeviceToHost, s));
        }
        UZ_HIP(hipStreamSynchronize(st[0]));
        UZ_HIP(hipStreamSynchronize(st[1]));
        for (size_t i = 0; i < ns; i++)
            if (flags[2 * i + 1])
                throw UzError{UZ_E_RANGE, "BGZF block " + std::to_string(cut[i] + (flags[2 * i + 1] >> 4)) + " of the batch: not a valid DEFLATE stream of the declared size (code " +
                                              std::to_string(flags[2 * i + 1] & 15) + ")"};
    });
}

int uz_reads_adopt_device(uz_ctx *c, const uz_reads_packed_view *v, int *id) {
    return guarded(c, [&] {
        UZ_REQUIRE(v && id, UZ_E_ARG, "bad reads view");
        check_packed_view(v);
        UZ_REQUIRE(!v->cigar_compact, UZ_E_ARG, "uz_reads_adopt_device takes every CIGAR word (cigar_compact = 0): the caller's column IS the device's store");
        UZ_REQUIRE(!v->bl_n && !v->tup_n_bl, UZ_E_ARG, "uz_reads_adopt_device takes base rows: the list form of the bases (bl_*) is a form of the host link");
        UZ_REQUIRE(!v->tup8, UZ_E_ARG, "uz_reads_adopt_device takes the 16-bit dictionary index: the one-byte form (tup8) is a form of the host link");
        UZ_REQUIRE(((uintptr_t)v->qlow | (uintptr_t)v->seq4 | (uintptr_t)v->seq2 | (uintptr_t)v->cigar) % 16 == 0, UZ_E_ARG, "device columns must be 16-byte aligned");
        ReadsDev r;
        r.live = true;
        r.n = v->n_segs; r.n_contigs = v->n_contigs; r.n_qnames = v->n_qnames;
        r.n_cigar_total = v->n_cigar_total; r.n_row_units = v->n_row_units; r.n_seq_units = v->n_seq_units;
        void *scratch = nullptr;
        uint8_t *seq4_own = nullptr; // two-bit rows are expanded into the library's own block
        uint8_t *qlow_own = nullptr; // the list form of the quality plane likewise
        const bool a_lists = v->n_low != nullptr || (v->tup && v->tup_n_low);
        r.n_qlow_pos = a_lists ? v->n_qlow_pos : 0;
        r.n_plane_units = a_lists ? v->n_seq_units : v->n_row_units;
        for (int pass = 0; pass < 2; pass++) {
            Carver cv(pass ? r.block.p : nullptr);
            carve_common(cv, r);
            if (v->seq2) seq4_own = cv.take<uint8_t>((size_t)r.n_seq_units * UZ_SEQ4_UNIT_BYTES);
            if (a_lists) qlow_own = cv.take<uint8_t>((size_t)r.n_seq_units * UZ_QLOW_UNIT_BYTES);
            scratch = cv.take<uint8_t>(uz_rec_scratch_bytes(r.n));
            if (!pass) r.block = uz_block_get(c, cv.off + 256);
        }
        try {
            hipStream_t st = c->stream;
            UZ_HIP(hipMemcpyAsync(r.contig_off, v->contig_off, ((size_t)v->n_contigs + 1) * sizeof(int64_t), hipMemcpyDeviceToDevice, st));
            if (v->n_contigs) UZ_HIP(hipMemcpyAsync(r.max_span, v->max_span, (size_t)v->n_contigs * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
            RecColumns col;
            col.start = v->start; col.end = v->end; col.tlen = v->tlen; col.mate = v->mate; col.qname = v->qname;
            if (v->start_d || v->start_d8) {
                col.start_d = v->start_d; col.start_d8 = v->start_d8; col.tlen_s = v->tlen_s; col.mate_d = v->mate_d; col.qname_d = v->qname_d;
                col.mate_d8 = v->mate_d8; col.qname_d8 = v->qname_d8; col.pair_d8 = v->pair_d8;
                col.esc16_key = (const unsigned long long *)v->esc16_key; col.esc16_val = v->esc16_val; col.n_esc16 = v->n_esc16;
            }
            col.flag = v->flag; col.l_seq = v->l_seq; col.n_cigar = v->n_cigar; col.mapq = v->mapq; col.aux = v->aux;
            r.cigar = v->cigar; r.seq4 = v->seq4; r.qlow = const_cast<uint8_t *>(v->qlow);
            col.cigar_in = v->cigar;
            if (v->seq2) {
                r.seq4 = seq4_own; r.seq2_staged = v->seq2;
                r.n_exc = v->n_exc; r.exc_rec = v->exc_rec; r.exc_pos = v->exc_pos; r.exc_code = v->exc_code;
            }
            if (v->tup) {
                col.tup = v->tup; col.tup_flag = v->tup_flag; col.tup_l_seq = v->tup_l_seq; col.tup_n_cigar = v->tup_n_cigar;
                col.tup_mapq = v->tup_mapq; col.tup_aux = v->tup_aux; col.tup_n_low = v->tup_n_low; col.tup_umask = v->tup_umask; col.n_tup = v->n_tup;
            }
            if (v->n_low || (v->tup && v->tup_n_low)) { r.qlow = qlow_own; col.lists = 1; col.n_low = v->n_low; col.qlow_pos = v->qlow_pos; col.qpos_wide = v->qlow_pos_wide; col.umask = v->umask; }
            else col.plane_in = reinterpret_cast<const uint32_t *>(v->qlow);
            r.qlow_thr = v->min_base_qual;
            r.qlow_valid = true;
            uz_build_records(c, st, r, col, scratch);
            UZ_HIP(hipStreamSynchronize(st));
            r.exc_rec = nullptr; r.exc_pos = nullptr; r.exc_code = nullptr; r.n_exc = 0; // (the caller's memory: not kept)
            if (c->hflags[0]) {
                const int f = c->hflags[0];
                c->hflags[0] = 0;
                throw UzError{UZ_E_RANGE, f == 5 ? "umask: a unit beyond the read's length, or a mask on a read longer than 480 bases"
                                          : f == 4 ? "qlow_pos: positions of a record are not ascending or lie beyond l_seq"
                                          : f == 3 ? "exc_* columns: an entry names a record without bases, a base beyond l_seq or a code above 15"
                                                 : "n_cigar_total / n_row_units of the reads view do not match its columns"};
            }
        } catch (...) { uz_block_put(c, r.block); uz_block_put(c, r.mirror); throw; }
        const int k = new_slot(c->reads);
        c->reads[k] = r;
        *id = k;
    });
}

int uz_pinned_alloc(size_t bytes, void **out) {
    if (!out) return UZ_E_ARG;
    *out = nullptr;
    if (hipHostMalloc(out, bytes ? bytes : 64, hipHostMallocDefault) != hipSuccess) return UZ_E_HIP;
    std::lock_guard<std::mutex> g(g_pinned_mu);
    g_pinned[(uintptr_t)*out] = bytes ? bytes : 64;
    return 0;
}
void uz_pinned_free(void *p) {
    if (!p) return;
    {
        std::lock_guard<std::mutex> g(g_pinned_mu);
        g_pinned.erase((uintptr_t)p);
    }
    (void)hipHostFree(p);
}

int uz_drop_derived(uz_ctx *c) {
    return guarded(c, [&] {
        for (auto &f : c->fams) f.cls_valid = false;
        find_forget(c, -1);
    });
}

int uz_sites_free(uz_ctx *c, int sites_id) {
    return guarded(c, [&] {
        SitesDev &s = sites_of(c, sites_id);
        UZ_HIP(hipStreamSynchronize(c->stream));
        for (size_t k = 0; k < c->fams.size(); k++)
            if (c->fams[k].live && c->fams[k].sites_id == sites_id) { find_forget(c, (int)k); free_family(c, c->fams[k]); }
        free_sites(c, s);
    });
}
int uz_reads_free(uz_ctx *c, int reads_id) {
    return guarded(c, [&] {
        ReadsDev &r = reads_of(c, reads_id);
        // the block goes back to the pool: whatever still reads or fills it must have finished
        if (r.ready) UZ_HIP(hipEventSynchronize(r.ready));
        if (r.built) UZ_HIP(hipEventSynchronize(r.built));
        UZ_HIP(hipStreamSynchronize(c->stream));
        // a merged cohort table built from this table (or the merged table itself) is forgotten with it
        const bool member = std::find(c->cohort_ids.begin(), c->cohort_ids.end(), reads_id) != c->cohort_ids.end();
        if (reads_id == c->cohort_reads) { c->cohort_reads = -1; c->cohort_ids.clear(); }
        else if (member && c->cohort_reads >= 0) {
            free_reads(c, c->reads[(size_t)c->cohort_reads]);
            c->cohort_reads = -1;
            c->cohort_ids.clear();
        }
        free_reads(c, r);
    });
}

int uz_site_scan(uz_ctx *c, int fam_id) {
    return guarded(c, [&] {
        FamilyDev &f = fam_of(c, fam_id);
        uz_launch_site_scan(c, f, sites_of(c, f.sites_id), true);
    });
}

int uz_site_scan_many(uz_ctx *c, const int32_t *fam_ids, int32_t n_fam) {
    return guarded(c, [&] {
        UZ_REQUIRE(n_fam >= 0 && (n_fam == 0 || fam_ids != nullptr), UZ_E_ARG, "bad family list");
        if (n_fam == 0) return;
        std::vector<FamilyDev *> fams;
        for (int32_t k = 0; k < n_fam; k++) {
            FamilyDev &f = fam_of(c, fam_ids[k]);
            UZ_REQUIRE(f.sites_id == fam_of(c, fam_ids[0]).sites_id, UZ_E_ARG, "the families of one cohort scan must share a sites table");
            for (int32_t j = 0; j < k; j++) UZ_REQUIRE(fam_ids[j] != fam_ids[k], UZ_E_ARG, "family listed twice");
            fams.push_back(&f);
        }
        uz_launch_site_scan_many(c, fams.data(), n_fam, sites_of(c, fams[0]->sites_id), true);
    });
}

int uz_site_classes(uz_ctx *c, int fam_id, uint8_t *out) {
    return guarded(c, [&] {
        FamilyDev &f = fam_of(c, fam_id);
        SitesDev &s = sites_of(c, f.sites_id);
        UZ_REQUIRE(out != nullptr, UZ_E_ARG, "null output");
        if (!uz_site_scan_fresh(c, f, true)) uz_launch_site_scan(c, f, s, true);
        if (s.n) UZ_HIP(hipMemcpyAsync(out, f.cls, (size_t)s.n, hipMemcpyDeviceToHost, c->stream));
        UZ_HIP(hipStreamSynchronize(c->stream));
    });
}

int uz_find(uz_ctx *c, int fam_id, const uz_dnms_view *d, int mode, int64_t *cand_off, int64_t *het_off) {
    return guarded(c, [&] {
        FamilyDev &f = fam_of(c, fam_id);
        SitesDev &s = sites_of(c, f.sites_id);
        UZ_REQUIRE(d != nullptr && d->n >= 0, UZ_E_ARG, "bad DNM view");
        c->cohort_on = false;
        c->phase_valid = false;
        const bool cnv = (mode & UZ_FIND_WHOLE_REGION) != 0;
        if (!uz_site_scan_fresh(c, f, cnv)) uz_launch_site_scan(c, f, s, cnv); // (reads nothing of the batch: queued before the batch is staged)
        uz_stage_dnms(c, d);
        const FindKey key = find_key_of(c, fam_id, mode, d); // (behind the site scan's launch: host work beside the device's)
        find_target(c);
        uz_launch_find(c, f, s, mode);
        find_done(c, key);
        c->find_fam = fam_id;
        if (cand_off) memcpy(cand_off, c->cand_off_h.data(), ((size_t)d->n + 1) * sizeof(int64_t));
        if (het_off) memcpy(het_off, c->het_off_h.data(), ((size_t)d->n + 1) * sizeof(int64_t));
    });
}

int uz_find_fetch(uz_ctx *c, int32_t *cand_idx, uint8_t *cand_flags, int32_t *het_idx) {
    return guarded(c, [&] {
        UZ_REQUIRE(c->find_valid, UZ_E_STATE, "uz_find_fetch before uz_find");
        // destinations in page-locked memory of this library (uz_pinned_alloc): by copy kernels, past the DMA engine's queue (uz_launch_find)
        const bool pinned = (!cand_idx || !c->n_cand || inside_one_pinned_block((const uint8_t *)cand_idx, (const uint8_t *)(cand_idx + c->n_cand))) &&
                            (!cand_flags || !c->n_cand || inside_one_pinned_block(cand_flags, cand_flags + c->n_cand)) &&
                            (!het_idx || !c->n_het || inside_one_pinned_block((const uint8_t *)het_idx, (const uint8_t *)(het_idx + c->n_het)));
        if (pinned) {
            if (cand_idx && c->n_cand) uz_kcopy(c, cand_idx, c->cand_idx.p, (size_t)c->n_cand * sizeof(int32_t));
            if (cand_flags && c->n_cand) uz_kcopy(c, cand_flags, c->cand_flags.p, (size_t)c->n_cand);
            if (het_idx && c->n_het) uz_kcopy(c, het_idx, c->het_idx.p, (size_t)c->n_het * sizeof(int32_t));
            UZ_HIP(hipStreamSynchronize(c->stream));
            return;
        }
        if (cand_idx && c->n_cand)
            UZ_HIP(hipMemcpyAsync(cand_idx, c->cand_idx.p, (size_t)c->n_cand * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        if (cand_flags && c->n_cand)
            UZ_HIP(hipMemcpyAsync(cand_flags, c->cand_flags.p, (size_t)c->n_cand, hipMemcpyDeviceToHost, c->stream));
        if (het_idx && c->n_het)
            UZ_HIP(hipMemcpyAsync(het_idx, c->het_idx.p, (size_t)c->n_het * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        UZ_HIP(hipStreamSynchronize(c->stream));
    });
}

// uz_phase in two halves (unfazed_hip.h)
static void phase_whole(uz_ctx *c, int fam_id, int reads_id, const uz_dnms_view *d, int find_mode, int32_t *status, int32_t *counts, int32_t *origin,
                        int32_t *evidence, bool defer) {
    FamilyDev &f = fam_of(c, fam_id);
    SitesDev &s = sites_of(c, f.sites_id);
    ReadsDev &r = reads_of(c, reads_id);
    UZ_REQUIRE(d != nullptr, UZ_E_ARG, "null DNM view");
    UZ_REQUIRE(!(find_mode & UZ_FIND_WHOLE_REGION), UZ_E_ARG, "the read stage runs on SNV / breakpoint windows");
    c->cohort_on = false;
    c->phase_valid = false;
    c->phase_qbase.clear();
    static const bool trace = getenv("UZ_PHASE_HOST_TRACE") != nullptr; // development aid: where the HOST's time of a read-stage call goes (us)
    const auto t0 = std::chrono::steady_clock::now();
    auto us = [&] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count(); };
    // the read stage consumes the window lists of this batch in SNV / breakpoint mode: those of the caller's own uz_find, if one of
    // the last two finds was over this batch; else computed here
    // (the batch goes up and the site scan is queued BEFORE the key of the batch is worked out: a hash over its columns, host work the device
    // need not wait for)
    // ... and the site scan, which reads nothing of the batch, before the batch's columns are copied into the staging buffer)
    if (!uz_site_scan_fresh(c, f, false)) uz_launch_site_scan(c, f, s, false);
    const double t_scan0 = us();
    uz_stage_dnms(c, d);
    const double t_stage0 = us() - t_scan0, t_scan = us();
    const FindKey key = find_key_of(c, fam_id, find_mode, d);
    const double t_key = us() - t_scan, t_stage = us();
    const bool have = find_recall(c, key);
    if (!have) find_target(c);
    if (!have) {
        uz_launch_find(c, f, s, find_mode, false);
        find_done(c, key);
    }
    const double t_find = us();
    c->find_fam = fam_id;
    uz_launch_phase(c, f, s, r, status, counts, origin, evidence, defer);
    if (trace) fprintf(stderr, "[uz] read-stage call, host us: site scan queued %.0f | stage dnms %.0f | key %.0f | find %.0f | phase (launches, waits, results) %.0f\n", t_scan0,
                       t_stage0, t_key, t_find - t_stage, us() - t_find);
}
int uz_phase(uz_ctx *c, int fam_id, int reads_id, const uz_dnms_view *d, int find_mode, int32_t *status,
             int32_t *counts, int32_t *origin, int32_t *evidence) {
    return guarded(c, [&] {
        UZ_REQUIRE(!c->phase_open, UZ_E_STATE, "uz_phase: a batch is still open (uz_phase_end)");
        phase_whole(c, fam_id, reads_id, d, find_mode, status, counts, origin, evidence, false);
    });
}
int uz_phase_begin(uz_ctx *c, int fam_id, int reads_id, const uz_dnms_view *d, int find_mode) {
    return guarded(c, [&] {
        UZ_REQUIRE(!c->phase_open, UZ_E_STATE, "uz_phase_begin: the batch before this one is still open (uz_phase_end)");
        phase_whole(c, fam_id, reads_id, d, find_mode, nullptr, nullptr, nullptr, nullptr, true);
        c->phase_open = true;
    });
}
int uz_phase_end(uz_ctx *c, int fam_id, int reads_id, const uz_dnms_view *d, int find_mode, int32_t *status, int32_t *counts, int32_t *origin,
                 int32_t *evidence) {
    return guarded(c, [&] {
        UZ_REQUIRE(c->phase_open, UZ_E_STATE, "uz_phase_end without uz_phase_begin");
        c->phase_open = false;
        if (uz_finish_phase(c, status, counts, origin, evidence)) return;
        // the batch outgrew the sizes it was run on: once more, whole, on its own sizes (the DNMs and the window lists of the
        // context may be another batch's by now)
        phase_whole(c, fam_id, reads_id, d, find_mode, status, counts, origin, evidence, false);
    });
}

int uz_phase_cohort(uz_ctx *c, const uz_cohort_group *groups, int32_t n_groups, const uz_dnms_view *d, int find_mode, int32_t *status,
                    int32_t *counts, int32_t *origin, int32_t *evidence) {
    return guarded(c, [&] {
        UZ_REQUIRE(groups && n_groups > 0 && d, UZ_E_ARG, "bad cohort batch");
        UZ_REQUIRE(!(find_mode & UZ_FIND_WHOLE_REGION), UZ_E_ARG, "the read stage runs on SNV / breakpoint windows");
        const int32_t n = d->n;
        FamilyDev &f0 = fam_of(c, groups[0].fam_id);
        SitesDev &s = sites_of(c, f0.sites_id);
        // ---- the kids' tables end to end as one table
        std::vector<int> ids;
        int64_t tot_n = 0, tot_c = 0, tot_u = 0, tot_s = 0, tot_p = 0, tot_contigs = 0;
        uint64_t tot_q = 0;
        for (int32_t g = 0; g < n_groups; g++) {
            UZ_REQUIRE(groups[g].dnm_first >= 0 && groups[g].dnm_count >= 0 && groups[g].dnm_first + groups[g].dnm_count <= n, UZ_E_ARG,
                       "cohort group outside the DNM batch");
            UZ_REQUIRE(fam_of(c, groups[g].fam_id).sites_id == f0.sites_id, UZ_E_ARG, "the families of a cohort batch must share a sites table");
            ReadsDev &r = reads_of(c, groups[g].reads_id);
            UZ_REQUIRE(groups[g].reads_id != c->cohort_reads, UZ_E_ARG, "the merged cohort table cannot be a member of a cohort");
            uz_reads_make_ready(c, r);
            if (!r.qlow_valid || r.qlow_thr != c->P.min_gt_qual) {
                UZ_REQUIRE(r.qual8 != nullptr, UZ_E_STATE, "a reads table of the cohort was packed for another --min-gt-qual");
                uz_build_qlow(c, c->stream, r, c->P.min_gt_qual);
            }
            ids.push_back(groups[g].reads_id);
            tot_n += r.n; tot_c += r.n_cigar_total; tot_u += r.n_row_units; tot_s += r.n_seq_units; tot_p += r.n_plane_units;
            tot_contigs += r.n_contigs; tot_q += r.n_qnames;
        }
        UZ_REQUIRE(tot_n < (int64_t)0x7FFFFFF0 && tot_c < ((int64_t)1 << 32) && tot_u < ((int64_t)1 << 32) && tot_q < ((uint64_t)1 << 32), UZ_E_RANGE,
                   "the cohort's alignment records exceed one table's index ranges");
        std::vector<int64_t> rec_base((size_t)n_groups), contig_base((size_t)n_groups);
        std::vector<uint32_t> q_base((size_t)n_groups);
        {
            int64_t rb = 0, cb = 0;
            uint32_t qb = 0;
            for (int32_t g = 0; g < n_groups; g++) {
                const ReadsDev &r = reads_of(c, groups[g].reads_id);
                rec_base[(size_t)g] = rb; contig_base[(size_t)g] = cb; q_base[(size_t)g] = qb;
                rb += r.n; cb += r.n_contigs; qb += r.n_qnames;
            }
        }
        if (c->cohort_reads < 0 || !c->reads[(size_t)c->cohort_reads].live || c->cohort_ids != ids) {
            if (c->cohort_reads >= 0 && c->reads[(size_t)c->cohort_reads].live) {
                UZ_HIP(hipStreamSynchronize(c->stream));
                free_reads(c, c->reads[(size_t)c->cohort_reads]);
            }
            ReadsDev m;
            m.live = true;
            m.n = tot_n; m.n_contigs = (int32_t)tot_contigs; m.n_qnames = (uint32_t)tot_q; m.n_cigar_total = tot_c; m.n_row_units = tot_u;
            m.n_seq_units = tot_s; m.n_plane_units = tot_p;
            uint32_t *cigar = nullptr; uint8_t *seq4 = nullptr, *qlow = nullptr;
            for (int pass = 0; pass < 2; pass++) {
                Carver cv(pass ? m.block.p : nullptr);
                carve_common(cv, m);
                cigar = cv.take<uint32_t>((size_t)tot_c);
                seq4 = cv.take<uint8_t>((size_t)tot_s * UZ_SEQ4_UNIT_BYTES);
                qlow = cv.take<uint8_t>((size_t)tot_p * UZ_QLOW_UNIT_BYTES);
                if (!pass) m.block = uz_block_get(c, cv.off + 256);
            }
            m.cigar = cigar; m.seq4 = seq4; m.qlow = qlow;
            m.qlow_thr = c->P.min_gt_qual; m.qlow_valid = true;
            try {
                std::vector<int64_t> co((size_t)tot_contigs + 1, 0);
                std::vector<int32_t> ms((size_t)tot_contigs + 1, 0);
                int64_t cg = 0, un = 0, sn = 0;
                for (int32_t g = 0; g < n_groups; g++) {
                    const ReadsDev &r = reads_of(c, groups[g].reads_id);
                    std::vector<int64_t> rco((size_t)r.n_contigs + 1);
                    std::vector<int32_t> rms((size_t)r.n_contigs + 1);
                    UZ_HIP(hipMemcpyAsync(rco.data(), r.contig_off, rco.size() * sizeof(int64_t), hipMemcpyDeviceToHost, c->stream));
                    if (r.n_contigs) UZ_HIP(hipMemcpyAsync(rms.data(), r.max_span, (size_t)r.n_contigs * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
                    UZ_HIP(hipStreamSynchronize(c->stream));
                    for (int32_t k = 0; k < r.n_contigs; k++) {
                        co[(size_t)contig_base[(size_t)g] + k] = rec_base[(size_t)g] + rco[(size_t)k];
                        ms[(size_t)contig_base[(size_t)g] + k] = rms[(size_t)k];
                    }
                    uz_concat_table(c, c->stream, m, r, rec_base[(size_t)g], cg, un, sn, q_base[(size_t)g]);
                    cg += r.n_cigar_total; un += r.n_plane_units; sn += r.n_seq_units;
                }
                co[(size_t)tot_contigs] = tot_n;
                UZ_HIP(hipMemcpyAsync(m.contig_off, co.data(), co.size() * sizeof(int64_t), hipMemcpyHostToDevice, c->stream));
                UZ_HIP(hipMemcpyAsync(m.max_span, ms.data(), ms.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
                uz_finish_table(c, c->stream, m);
                UZ_HIP(hipStreamSynchronize(c->stream)); // co / ms are stack-local
            } catch (...) { uz_block_put(c, m.block); throw; }
            const int k = new_slot(c->reads);
            c->reads[(size_t)k] = m;
            c->cohort_reads = k;
            c->cohort_ids = ids;
        }
        ReadsDev &R = c->reads[(size_t)c->cohort_reads];
        // ---- the batch: reads contigs renumbered into the merged table, family / cutoff / name base per DNM
        std::vector<int32_t> rc((size_t)n), fam_h((size_t)n, 0);
        std::vector<double> cut_h((size_t)n, 0.0);
        c->phase_qbase.assign((size_t)n, 0);
        for (int32_t k = 0; k < n; k++) rc[(size_t)k] = -1;
        std::vector<uint8_t *> cls_h((size_t)n_groups);
        for (int32_t g = 0; g < n_groups; g++) {
            FamilyDev &f = fam_of(c, groups[g].fam_id);
            if (!uz_site_scan_fresh(c, f, false)) uz_launch_site_scan(c, f, s, false);
            cls_h[(size_t)g] = f.cls;
            const int32_t nc_g = reads_of(c, groups[g].reads_id).n_contigs;
            for (int32_t k = groups[g].dnm_first; k < groups[g].dnm_first + groups[g].dnm_count; k++) {
                const int32_t t = d->rcontig[k];
                rc[(size_t)k] = (t >= 0 && t < nc_g) ? (int32_t)(contig_base[(size_t)g] + t) : -1;
                fam_h[(size_t)k] = g; cut_h[(size_t)k] = groups[g].cutoff; c->phase_qbase[(size_t)k] = q_base[(size_t)g];
            }
        }
        uz_dnms_view dv = *d;
        dv.rcontig = rc.data();
        find_target(c); c->phase_valid = false; // (the cohort's lists take the place of the older set, under no key)
        uz_stage_dnms(c, &dv);
        c->dn_fam.ensure((size_t)n + 1); c->dn_cutoff.ensure((size_t)n + 1); c->fam_cls.ensure((size_t)n_groups + 1);
        if (n) {
            UZ_HIP(hipMemcpyAsync(c->dn_fam.p, fam_h.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
            UZ_HIP(hipMemcpyAsync(c->dn_cutoff.p, cut_h.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        }
        UZ_HIP(hipMemcpyAsync(c->fam_cls.p, cls_h.data(), (size_t)n_groups * sizeof(uint8_t *), hipMemcpyHostToDevice, c->stream));
        UZ_HIP(hipStreamSynchronize(c->stream)); // the staging vectors above are locals
        c->cohort_on = true;
        try {
            uz_launch_find(c, f0, s, find_mode, false);
            c->find_fam = groups[0].fam_id;
            uz_launch_phase(c, f0, s, R, status, counts, origin, evidence);
        } catch (...) { c->cohort_on = false; throw; }
        c->cohort_on = false;
    });
}

int uz_phase_cnv(uz_ctx *c, int fam_id, const uz_dnms_view *d, const int32_t *rb_counts, int32_t *cnv_counts, int32_t *origin,
                 int32_t *evidence, int32_t *etype) {
    return guarded(c, [&] {
        FamilyDev &f = fam_of(c, fam_id);
        SitesDev &s = sites_of(c, f.sites_id);
        UZ_REQUIRE(d != nullptr, UZ_E_ARG, "null DNM view");
        find_target(c); c->phase_valid = false; c->cnv_valid = false;
        uz_stage_dnms(c, d);
        const size_t n = (size_t)d->n;
        if (!uz_site_scan_fresh(c, f, true)) uz_launch_site_scan(c, f, s, true);
        const int32_t sd = c->P.search_dist;
        c->P.search_dist = 0; // run_cnv_phasing calls find(..., search_dist=0, whole_region=True), sv_phaser.py:375-389
        try { uz_launch_find(c, f, s, UZ_FIND_WHOLE_REGION, false); } catch (...) { c->P.search_dist = sd; throw; }
        c->P.search_dist = sd;
        c->find_fam = fam_id;
        c->cnv_n = d->n;
        c->cnv_counts.ensure(2 * n + 2); c->cnv_pos.ensure((size_t)c->n_cand + 1); c->cnv_origin.ensure(n + 1);
        c->cnv_evidence.ensure(n + 1); c->cnv_etype.ensure(n + 1);
        const int32_t *rb = nullptr;
        if (rb_counts && n) {
            c->cnv_rb.ensure(4 * n);
            UZ_HIP(hipMemcpyAsync(c->cnv_rb.p, rb_counts, 4 * n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
            rb = c->cnv_rb.p;
        }
        uz_launch_cnv(c, s, rb, c->cnv_counts.p, c->cnv_pos.p, c->cnv_origin.p, c->cnv_evidence.p, c->cnv_etype.p);
        c->cnv_counts_h.resize(2 * n + 2);
        if (n) {
            UZ_HIP(hipMemcpyAsync(c->cnv_counts_h.data(), c->cnv_counts.p, 2 * n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
            if (origin) UZ_HIP(hipMemcpyAsync(origin, c->cnv_origin.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
            if (evidence) UZ_HIP(hipMemcpyAsync(evidence, c->cnv_evidence.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
            if (etype) UZ_HIP(hipMemcpyAsync(etype, c->cnv_etype.p, n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        }
        UZ_HIP(hipStreamSynchronize(c->stream));
        if (cnv_counts && n) memcpy(cnv_counts, c->cnv_counts_h.data(), 2 * n * sizeof(int32_t));
        c->cnv_valid = true;
    });
}

int uz_phase_cnv_sites(uz_ctx *c, int64_t *off, int32_t *pos) {
    return guarded(c, [&] {
        UZ_REQUIRE(c->cnv_valid && c->find_valid, UZ_E_STATE, "uz_phase_cnv_sites before uz_phase_cnv");
        UZ_REQUIRE(off != nullptr, UZ_E_ARG, "null output");
        const size_t n = (size_t)c->cnv_n;
        // the lists keep the layout of the device array: the DNM's slice of the candidate list, dad's sites then mom's
        std::vector<int64_t> co(n + 1);
        if (n) UZ_HIP(hipMemcpy(co.data(), c->cand_off.p, (n + 1) * sizeof(int64_t), hipMemcpyDeviceToHost));
        int64_t total = 0;
        for (size_t d = 0; d < n; d++) {
            off[2 * d] = total; total += c->cnv_counts_h[2 * d];
            off[2 * d + 1] = total; total += c->cnv_counts_h[2 * d + 1];
        }
        off[2 * n] = total;
        if (!pos || total == 0) return;
        std::vector<int32_t> all((size_t)c->n_cand + 1);
        UZ_HIP(hipMemcpy(all.data(), c->cnv_pos.p, (size_t)c->n_cand * sizeof(int32_t), hipMemcpyDeviceToHost));
        for (size_t d = 0; d < n; d++) {
            const int64_t len = off[2 * d + 2] - off[2 * d];
            if (len) memcpy(pos + off[2 * d], all.data() + co[d], (size_t)len * sizeof(int32_t));
        }
    });
}

int uz_phase_votes(uz_ctx *c, int64_t *vote_off, int32_t *vote_val) {
    int rc = 0;
    int g = guarded(c, [&] {
        UZ_REQUIRE(c->phase_valid, UZ_E_STATE, "uz_phase_votes before uz_phase");
        rc = uz_phase_votes_impl(c, vote_off, vote_val);
    });
    return g ? g : rc;
}
int uz_phase_groups(uz_ctx *c, int64_t *grp_off, int32_t *grp_q) {
    int rc = 0;
    int g = guarded(c, [&] {
        UZ_REQUIRE(c->phase_valid, UZ_E_STATE, "uz_phase_groups before uz_phase");
        rc = uz_phase_groups_impl(c, grp_off, grp_q);
    });
    return g ? g : rc;
}

int uz_prof_enable(uz_ctx *c, int on) {
    return guarded(c, [&] { c->prof_mask = on == 0 ? 0u : on == 1 ? ~0u : (uint32_t)on >> 1; });
}
int uz_prof_reset(uz_ctx *c) {
    return guarded(c, [&] {
        UZ_HIP(hipStreamSynchronize(c->stream));
        uz_prof_drain(c);
        for (auto &p : c->prof) p = ProfSlot();
    });
}
int uz_prof_get(uz_ctx *c, int kernel, double *total_ms, int64_t *launches) {
    return guarded(c, [&] {
        UZ_REQUIRE(kernel >= 0 && kernel < UZ_K_COUNT, UZ_E_ARG, "bad kernel id");
        uz_prof_drain(c);
        if (total_ms) *total_ms = c->prof[kernel].total_ms;
        if (launches) *launches = c->prof[kernel].launches;
    });
}
int uz_prof_units(uz_ctx *c, int kernel, int64_t *units) {
    return guarded(c, [&] {
        UZ_REQUIRE(kernel >= 0 && kernel < UZ_K_COUNT && units != nullptr, UZ_E_ARG, "bad kernel id");
        *units = c->prof[kernel].last_units;
    });
}

// ---- the record walk on the device (include/uz_bamwalk.h, csrc/k_bamwalk.hip)
int uz_bam_walk(uz_ctx *c, const uint8_t *comp, int64_t comp_bytes, int64_t n_blocks, const int64_t *in_off, const int64_t *out_off, const int64_t *blk_coff,
                const uint32_t *blk_crc, int32_t n_tasks, const int32_t *task, int64_t n_spans, const int64_t *span, int64_t n_reach, const int32_t *reach, int64_t n_fetch,
                const int32_t *fetch, int *walk_id, int64_t *n_desc) {
    return guarded(c, [&] {
        UZ_REQUIRE(walk_id && n_desc && n_blocks >= 0 && comp_bytes >= 0 && n_tasks >= 0 && n_spans >= 0 && n_reach >= 0 && n_fetch >= 0, UZ_E_ARG, "bad arguments");
        UZ_REQUIRE(n_blocks == 0 || (comp && in_off && out_off && blk_coff), UZ_E_ARG, "null block table");
        UZ_REQUIRE(n_tasks == 0 || (task && span && reach && fetch), UZ_E_ARG, "null walk plan");
        const int64_t out_bytes = n_blocks ? out_off[n_blocks] : 0;
        for (int64_t k = 0; k < n_blocks; k++)
            UZ_REQUIRE(in_off[k] >= 0 && in_off[k] < comp_bytes && (k == 0 || in_off[k] > in_off[k - 1]) && out_off[k] >= 0 && out_off[k] <= out_off[k + 1] &&
                           out_off[k + 1] - out_off[k] <= 65536,
                       UZ_E_ARG, "bad block table (blocks in the order they lie in `comp`; a BGZF block inflates to at most 64 KiB)");
        // the plan is the kernel's only guard: every index it names must lie inside the arrays it names
        for (int32_t t = 0; t < n_tasks; t++) {
            const int32_t *tc = task + UZ_WALK_TASK_COLS * (size_t)t;
            UZ_REQUIRE(tc[2] >= 0 && tc[2] <= tc[3] && tc[3] <= n_spans && tc[4] >= 0 && tc[4] <= tc[5] && tc[5] <= n_reach && tc[6] >= 0 && tc[6] <= tc[7] && tc[7] <= n_fetch,
                       UZ_E_ARG, "walk plan: a task names spans, reach intervals or fetches outside the arrays");
            // column 9: the stage task a walk task belongs to -- its sub-tasks adjacent and in order (the hash sets of k_tab_insert / k_desc_filter and
            // the joins find a stage task's first sub-task by walking back over equal values)
            UZ_REQUIRE(tc[9] >= 0 && (t == 0 ? true : (tc[9] == tc[9 - UZ_WALK_TASK_COLS] || tc[9] == tc[9 - UZ_WALK_TASK_COLS] + 1)), UZ_E_ARG,
                       "walk plan: column 9 (the stage task of a walk task) must start at or above 0 and go up by at most one from task to task");
        }
        for (int64_t k = 0; k < n_spans; k++) {
            const int64_t *sc = span + UZ_WALK_SPAN_COLS * (size_t)k;
            UZ_REQUIRE(sc[4] >= 0 && sc[4] <= sc[5] && sc[5] <= n_blocks && (sc[4] == sc[5] || (sc[2] >= out_off[sc[4]] && sc[2] <= sc[3] && sc[3] == out_off[sc[5]])),
                       UZ_E_ARG, "walk plan: a span names blocks or bytes outside the block table");
        }
        UZ_HIP(hipSetDevice(c->device));
        // one pass: every task writes its descriptors into a slice sized for the most records its bytes can hold (a record is at least 36 bytes)
        std::vector<int64_t> first((size_t)n_tasks + 1, 0);
        for (int32_t t = 0; t < n_tasks; t++) {
            const int32_t *tc = task + UZ_WALK_TASK_COLS * (size_t)t;
            int64_t cap = 0;
            for (int32_t sp = tc[2]; sp < tc[3]; sp++) cap += (span[UZ_WALK_SPAN_COLS * (size_t)sp + 3] - span[UZ_WALK_SPAN_COLS * (size_t)sp + 2]) / 36 + 1;
            first[(size_t)t + 1] = first[(size_t)t] + cap;
        }
        // The slot: a free one whose large buffers already hold this batch -- the smallest such (best fit) --, else one that has never been
        // used, else the smallest (it is grown to the largest sizes ANY batch of this context has asked for: a slot grows once).  Nothing is
        // freed while batches are in flight (hipFree waits for the whole device -- round 5: a process's second call, whose larger last chunk met
        // another slot than in the first call, stood still for 0.9 s): an outgrown block is parked (DevBuf::ensure_parked).
        const size_t need_out = (size_t)out_bytes + uz_bam_walk_pad(), need_comp = (size_t)comp_bytes + 1024, need_desc = (size_t)first.back() + 1;
        int k = -1;
        {
            std::lock_guard<std::mutex> lk(c->err_mu);
            int fit = -1, fresh = -1, small = -1, busy = 0;
            for (int i = 0; i < uz_ctx::WALK_SLOTS; i++) {
                const uz_ctx::WalkSlot &s = c->walk[i];
                if (s.busy) { busy++; continue; }
                if (s.out.cap >= need_out && s.comp.cap >= need_comp && s.desc.cap >= need_desc) { if (fit < 0 || s.out.cap < c->walk[fit].out.cap) fit = i; }
                else if (s.out.cap == 0) { if (fresh < 0) fresh = i; }
                else if (small < 0 || s.out.cap < c->walk[small].out.cap) small = i;
            }
            k = fit >= 0 ? fit : fresh >= 0 ? fresh : small;
            if (k >= 0) c->walk[k].busy = true;
            if (busy == 0 && k >= 0) { // nothing of an earlier batch is in flight: what was parked can go (only worth a device-wide wait when it is a lot)
                std::lock_guard<std::mutex> lk2(c->walk_mu);
                size_t parked = 0;
                for (auto &b : c->walk_park) parked += b.second;
                if (parked > ((size_t)24 << 30)) {
                    for (auto &b : c->walk_park) (void)hipFree(b.first);
                    c->walk_park.clear();
                }
            }
        }
        UZ_REQUIRE(k >= 0, UZ_E_STATE, "four walked batches are waiting for uz_reads_from_bam / uz_reads_from_walk / uz_bam_walk_release");
        uz_ctx::WalkSlot &w = c->walk[k];
        try {
            // streams of the slot's own: the blocks of the next batch go up and are inflated while this one is still walked (two calls may run at once)
            if (!w.s0) {
                UZ_HIP(hipStreamCreateWithFlags(&w.s0, hipStreamNonBlocking));
                UZ_HIP(hipStreamCreateWithFlags(&w.s1, hipStreamNonBlocking));
                UZ_HIP(hipEventCreateWithFlags(&w.ev, hipEventDisableTiming));
            }
            hipStream_t st = w.s0;
            w.n_blocks = n_blocks; w.out_bytes = out_bytes; w.n_tasks = n_tasks; w.n_desc = 0; w.n_reach = n_reach;
            w.max_host = n_tasks ? task[UZ_WALK_TASK_COLS * (size_t)(n_tasks - 1) + 9] : -1;
            w.join.started = false; w.join.done = false; w.join.n_need = 0; w.join.n_all = 0; w.join.n_dev = 0; w.join.aux_bytes = 0; w.join.n_look = 0;
            uz_walk_grow(c, w.comp, need_comp, 0); uz_walk_grow(c, w.out, need_out, 1);
            uz_walk_grow(c, w.in_off, (size_t)n_blocks + 1, 2); uz_walk_grow(c, w.out_off, (size_t)n_blocks + 1, 3); uz_walk_grow(c, w.blk_coff, (size_t)n_blocks + 1, 4);
            uz_walk_grow(c, w.task, (size_t)n_tasks * UZ_WALK_TASK_COLS + 1, 5); uz_walk_grow(c, w.span, (size_t)n_spans * UZ_WALK_SPAN_COLS + 1, 6);
            uz_walk_grow(c, w.reach, (size_t)n_reach * 2 + 1, 7); uz_walk_grow(c, w.fetch, (size_t)n_fetch * 3 + 1, 8);
            uz_walk_grow(c, w.count, (size_t)n_tasks + 1, 9); uz_walk_grow(c, w.first, (size_t)n_tasks + 2, 10); uz_walk_grow(c, w.walked, (size_t)n_tasks + 1, 11);
            uz_walk_grow(c, w.flags, (size_t)n_tasks + 1, 12);
            uz_walk_grow(c, w.n_direct, (size_t)n_tasks + 1, 13); uz_walk_grow(c, w.tab_first, (size_t)n_tasks + 2, 14); uz_walk_grow(c, w.kcount, (size_t)n_tasks + 1, 15);
            uz_walk_grow(c, w.kfirst, (size_t)n_tasks + 2, 16);
            // the blocks in slices of ~8 k on two streams, as uz_bgzf_inflate_to_host sends them: slice i + 1 goes up while slice i is inflated
            std::vector<int64_t> cut{0};
            for (int64_t b = 1; b <= n_blocks; b++)
                if (b == n_blocks || (b - cut.back() >= 8192 && n_blocks - b >= 4096)) cut.push_back(b);
            const size_t ns = cut.size() - 1;
            uz_walk_grow(c, w.iflags, 2 * ns + 8, 17); uz_walk_grow(c, w.blk_crc, (size_t)n_blocks + 1, 18);
            std::vector<int32_t> iflags(2 * ns + 4, 0);
            if (n_blocks) {
                hipStream_t s2[2] = {w.s0, w.s1};
                UZ_HIP(hipMemsetAsync(w.comp.p + comp_bytes, 0, 1024, st));
                UZ_HIP(hipMemsetAsync(w.out.p + out_bytes, 0, uz_bam_walk_pad(), st));
                UZ_HIP(hipMemcpyAsync(w.in_off.p, in_off, (size_t)n_blocks * 8, hipMemcpyHostToDevice, st));
                UZ_HIP(hipMemcpyAsync(w.out_off.p, out_off, (size_t)(n_blocks + 1) * 8, hipMemcpyHostToDevice, st));
                UZ_HIP(hipMemcpyAsync(w.blk_coff.p, blk_coff, (size_t)n_blocks * 8, hipMemcpyHostToDevice, st));
                UZ_HIP(hipEventRecord(w.ev, st));
                UZ_HIP(hipStreamWaitEvent(s2[1], w.ev, 0));
                for (size_t i = 0; i < ns; i++) {
                    const int64_t b0 = cut[i], b1 = cut[i + 1];
                    const int64_t c0 = i == 0 ? 0 : std::max<int64_t>(in_off[b0] - 18, 0), c1 = i + 1 == ns ? comp_bytes : std::max<int64_t>(in_off[b1] - 18, c0);
                    UZ_HIP(hipMemcpyAsync(w.comp.p + c0, comp + c0, (size_t)(c1 - c0), hipMemcpyHostToDevice, s2[i & 1]));
                    uz_launch_inflate(c, s2[i & 1], b1 - b0, w.comp.p, (comp_bytes + 1024) & ~(int64_t)3, w.in_off.p + b0, w.out_off.p + b0, w.out.p, w.iflags.p + 2 * i);
                }
                if (ns > 1) { // the walk (first stream) reads what both streams inflated
                    UZ_HIP(hipEventRecord(w.ev, s2[1]));
                    UZ_HIP(hipStreamWaitEvent(st, w.ev, 0));
                }
                if (blk_crc) { // every block against the CRC-32 of its footer, as htslib's reader (and the host's walk) holds it
                    UZ_HIP(hipMemcpyAsync(w.blk_crc.p, blk_crc, (size_t)n_blocks * 4, hipMemcpyHostToDevice, st));
                    uz_launch_crc32(c, st, n_blocks, w.out.p, w.out_off.p, w.blk_crc.p, w.iflags.p + 2 * ns);
                }
            }
            int64_t tab_total = 0, kept = 0;
            if (n_tasks) {
                w.n_desc_all = first.back();
                uz_walk_grow(c, w.desc, need_desc, 19);
                UZ_HIP(hipMemcpyAsync(w.first.p, first.data(), ((size_t)n_tasks + 1) * 8, hipMemcpyHostToDevice, st));
                UZ_HIP(hipMemcpyAsync(w.task.p, task, (size_t)n_tasks * UZ_WALK_TASK_COLS * 4, hipMemcpyHostToDevice, st));
                if (n_spans) UZ_HIP(hipMemcpyAsync(w.span.p, span, (size_t)n_spans * UZ_WALK_SPAN_COLS * 8, hipMemcpyHostToDevice, st));
                if (n_reach) UZ_HIP(hipMemcpyAsync(w.reach.p, reach, (size_t)n_reach * 8, hipMemcpyHostToDevice, st));
                if (n_fetch) UZ_HIP(hipMemcpyAsync(w.fetch.p, fetch, (size_t)n_fetch * 12, hipMemcpyHostToDevice, st));
                uz_launch_bam_walk(c, st, n_tasks, w.out.p, w.out_off.p, w.blk_coff.p, w.task.p, w.span.p, w.reach.p, w.fetch.p, w.count.p, w.first.p, w.walked.p,
                                   w.flags.p, w.desc.p, w.n_direct.p, w.tab_first.p);
                UZ_HIP(hipMemcpyAsync(&tab_total, w.tab_first.p + n_tasks, 8, hipMemcpyDeviceToHost, st));
                UZ_HIP(hipStreamSynchronize(st)); // (the pageable `first` has been read; the hash sets' size is known)
                // the descriptors the host's joins can need at all (direct, or sharing a name hash with a direct record of the task) are counted
                uz_walk_grow(c, w.tab, (size_t)tab_total + 1, 20);
                UZ_HIP(hipMemsetAsync(w.tab.p, 0, (size_t)tab_total * 8, st));
                uz_launch_desc_filter(c, st, false, n_tasks, w.desc.p, w.first.p, w.count.p, w.task.p, w.tab_first.p, w.tab.p, w.kcount.p, w.kfirst.p, nullptr);
                UZ_HIP(hipMemcpyAsync(&kept, w.kfirst.p + n_tasks, 8, hipMemcpyDeviceToHost, st));
            }
            if (n_blocks) UZ_HIP(hipMemcpyAsync(iflags.data(), w.iflags.p, (2 * ns + (blk_crc ? 1 : 0)) * sizeof(int32_t), hipMemcpyDeviceToHost, st));
            UZ_HIP(hipStreamSynchronize(st));
            for (size_t i = 0; i < ns; i++)
                if (iflags[2 * i + 1])
                    throw UzError{UZ_E_RANGE, "BGZF block " + std::to_string(cut[i] + (iflags[2 * i + 1] >> 4)) + " of the batch: not a valid DEFLATE stream of the declared size (code " +
                                                  std::to_string(iflags[2 * i + 1] & 15) + ")"};
            if (blk_crc && n_blocks && iflags[2 * ns])
                throw UzError{UZ_E_RANGE, "CRC mismatch in BGZF block " + std::to_string(iflags[2 * ns] - 1) + " of the batch (file offset " +
                                              std::to_string((long long)blk_coff[iflags[2 * ns] - 1]) + ")"};
            w.n_desc = kept;
            *n_desc = kept;
            *walk_id = k;
        } catch (...) {
            // (copies into this frame's locals and kernels on the slot's buffers may still be queued: nothing of the slot is handed on, and the frame
            // is not left, before both of its streams have drained)
            if (w.s0) (void)hipStreamSynchronize(w.s0);
            if (w.s1) (void)hipStreamSynchronize(w.s1);
            (void)hipGetLastError();
            std::lock_guard<std::mutex> lk(c->err_mu);
            w.busy = false;
            throw;
        }
    });
}

int uz_bam_walk_fetch(uz_ctx *c, int walk_id, uz_walk_desc *desc, int64_t *d_first, int32_t *d_flags, int64_t *d_walked) {
    return guarded(c, [&] {
        UZ_REQUIRE(walk_id >= 0 && walk_id < uz_ctx::WALK_SLOTS && c->walk[walk_id].busy, UZ_E_ARG, "bad walk id");
        uz_ctx::WalkSlot &w = c->walk[walk_id];
        UZ_REQUIRE(d_first && (w.n_desc == 0 || desc), UZ_E_ARG, "null output");
        UZ_HIP(hipSetDevice(c->device));
        hipStream_t st = w.s0;
        const int32_t nt = w.n_tasks;
        if (nt == 0) { d_first[0] = 0; return; }
        uz_walk_grow(c, w.desc_kept, (size_t)w.n_desc + 1, 41);
        uz_launch_desc_filter(c, st, true, nt, w.desc.p, w.first.p, w.count.p, w.task.p, w.tab_first.p, w.tab.p, w.kcount.p, w.kfirst.p, w.desc_kept.p);
        if (w.n_desc) UZ_HIP(hipMemcpyAsync(desc, w.desc_kept.p, (size_t)w.n_desc * sizeof(uz_walk_desc), hipMemcpyDeviceToHost, st));
        UZ_HIP(hipMemcpyAsync(d_first, w.kfirst.p, (size_t)(nt + 1) * 8, hipMemcpyDeviceToHost, st));
        if (d_flags) UZ_HIP(hipMemcpyAsync(d_flags, w.flags.p, (size_t)nt * 4, hipMemcpyDeviceToHost, st));
        if (d_walked) UZ_HIP(hipMemcpyAsync(d_walked, w.walked.p, (size_t)nt * 8, hipMemcpyDeviceToHost, st));
        UZ_HIP(hipStreamSynchronize(st));
    });
}

int uz_bam_walk_release(uz_ctx *c, int walk_id) {
    return guarded(c, [&] {
        UZ_REQUIRE(walk_id >= 0 && walk_id < uz_ctx::WALK_SLOTS, UZ_E_ARG, "bad walk id");
        std::lock_guard<std::mutex> lk(c->err_mu);
        c->walk[walk_id].busy = false;
    });
}

// ---- the batch-wide joins on the device (csrc/k_bamjoin.hip)
int uz_bam_walk_flags(uz_ctx *c, int walk_id, int32_t *d_flags, int64_t *d_walked) {
    return guarded(c, [&] {
        UZ_REQUIRE(walk_id >= 0 && walk_id < uz_ctx::WALK_SLOTS && c->walk[walk_id].busy, UZ_E_ARG, "bad walk id");
        uz_ctx::WalkSlot &w = c->walk[walk_id];
        if (w.n_tasks == 0) return;
        if (d_flags) UZ_HIP(hipMemcpyAsync(d_flags, w.flags.p, (size_t)w.n_tasks * 4, hipMemcpyDeviceToHost, w.s0));
        if (d_walked) UZ_HIP(hipMemcpyAsync(d_walked, w.walked.p, (size_t)w.n_tasks * 8, hipMemcpyDeviceToHost, w.s0));
        UZ_HIP(hipStreamSynchronize(w.s0));
    });
}

int uz_bam_join(uz_ctx *c, int walk_id, int32_t n_host, const int32_t *h_flags, int32_t n_ref, int all_bases, const uz_walk_desc *xdesc, int64_t n_x, const uint8_t *xaux,
                int64_t xaux_bytes, const int32_t *look_tid, int64_t n_look, const int32_t *need_jtask, int64_t *n_need, int64_t totals[8]) {
    return guarded(c, [&] {
        UZ_REQUIRE(walk_id >= 0 && walk_id < uz_ctx::WALK_SLOTS && c->walk[walk_id].busy, UZ_E_ARG, "bad walk id");
        UZ_REQUIRE(n_need != nullptr, UZ_E_ARG, "null output");
        uz_ctx::WalkSlot &w = c->walk[walk_id];
        UZ_REQUIRE(n_host > w.max_host, UZ_E_ARG, "uz_bam_join: the plan names more tasks of the stage (column 9) than n_host");
        for (int64_t k = 0; k < n_x; k++)
            UZ_REQUIRE((xdesc[k].task & UZ_WALK_TASK_JOIN) && (xdesc[k].src & UZ_WALK_SRC_AUX), UZ_E_ARG, "uz_bam_join: a descriptor of the host without its join task or outside the aux bytes");
        JoinPlanHost P;
        P.n_host = n_host; P.n_ref = n_ref; P.h_flags = h_flags; P.all_bases = all_bases != 0;
        uz_join_run(c, w, P, xdesc, n_x, xaux, xaux_bytes, look_tid, n_look, need_jtask);
        *n_need = w.join.n_need;
        if (totals) for (int k = 0; k < 8; k++) totals[k] = w.join.done ? w.join.tot_h[k] : 0;
    });
}

int uz_bam_join_needs(uz_ctx *c, int walk_id, uz_need_rec *need) {
    return guarded(c, [&] {
        UZ_REQUIRE(walk_id >= 0 && walk_id < uz_ctx::WALK_SLOTS && c->walk[walk_id].busy && need, UZ_E_ARG, "bad walk id");
        uz_join_needs(c, c->walk[walk_id], need);
    });
}

int uz_bam_join_fetch(uz_ctx *c, int walk_id, uint64_t *voff, uint32_t *qname, int32_t *mate, uint8_t *bases, uz_kept_rec *kept, int64_t *contig_off, int32_t *max_span) {
    return guarded(c, [&] {
        UZ_REQUIRE(walk_id >= 0 && walk_id < uz_ctx::WALK_SLOTS && c->walk[walk_id].busy, UZ_E_ARG, "bad walk id");
        uz_ctx::WalkSlot &w = c->walk[walk_id];
        uz_join_fetch(c, w, voff, qname, mate, bases, kept);
        if (contig_off) for (size_t k = 0; k < w.join.contig_off_h.size(); k++) contig_off[k] = w.join.contig_off_h[k];
        if (max_span) for (int32_t k = 0; k < w.join.n_ref; k++) max_span[k] = w.join.max_span_h[(size_t)k];
    });
}

// Every free slot grown to the largest sizes any batch of this context has asked for, NOW -- by a caller that has just finished a batch and knows
// that more are coming (a staged pipeline keeps up to four in flight): a slot's first use otherwise pays for gigabytes of hipMalloc inside some later
// batch's walk (a pass of bench.py's feed leg that met a fresh slot took 0.7 s instead of 0.17).  n_slots: how many slots the caller will have in use.
int uz_walk_reserve(uz_ctx *c, int n_slots) {
    return guarded(c, [&] {
        UZ_HIP(hipSetDevice(c->device));
        for (int i = 0; i < uz_ctx::WALK_SLOTS && i < n_slots; i++) {
            uz_ctx::WalkSlot &w = c->walk[i];
            {
                std::lock_guard<std::mutex> lk(c->err_mu);
                if (w.busy) continue;
                w.busy = true; // (nobody takes it while it grows)
            }
            try {
                auto hi = [&](int kind) { std::lock_guard<std::mutex> lk(c->walk_mu); return c->walk_hi[kind]; };
                auto &J = w.join;
#define UZ_RSV(buf, kind) do { const size_t h__ = hi(kind); if (h__) uz_walk_grow(c, buf, h__, kind); } while (0)
                UZ_RSV(w.comp, 0); UZ_RSV(w.out, 1); UZ_RSV(w.in_off, 2); UZ_RSV(w.out_off, 3); UZ_RSV(w.blk_coff, 4); UZ_RSV(w.task, 5); UZ_RSV(w.span, 6); UZ_RSV(w.reach, 7);
                UZ_RSV(w.fetch, 8); UZ_RSV(w.count, 9); UZ_RSV(w.first, 10); UZ_RSV(w.walked, 11); UZ_RSV(w.flags, 12); UZ_RSV(w.n_direct, 13); UZ_RSV(w.tab_first, 14);
                UZ_RSV(w.kcount, 15); UZ_RSV(w.kfirst, 16); UZ_RSV(w.iflags, 17); UZ_RSV(w.blk_crc, 18); UZ_RSV(w.desc, 19); UZ_RSV(w.tab, 20);
                UZ_RSV(J.tmp, 40); UZ_RSV(w.desc_kept, 41); UZ_RSV(J.jtask, 42); UZ_RSV(J.keep, 43); UZ_RSV(J.mate, 44); UZ_RSV(J.target, 45); UZ_RSV(J.hkey_in, 46);
                UZ_RSV(J.hval_in, 47); UZ_RSV(J.hkey, 48); UZ_RSV(J.hperm, 49); UZ_RSV(J.inv, 50); UZ_RSV(J.front0, 51); UZ_RSV(J.front1, 52); UZ_RSV(J.need, 53);
                UZ_RSV(J.cnt, 54); UZ_RSV(J.aux, 55); UZ_RSV(J.jt_tid, 56); UZ_RSV(J.reach_key, 57); UZ_RSV(J.reach_a, 58); UZ_RSV(J.reach_host, 59); UZ_RSV(J.h_flags, 60);
                UZ_RSV(J.fkey_in, 63); UZ_RSV(J.fkey, 64); UZ_RSV(J.fval_in, 65); UZ_RSV(J.fidx, 66); UZ_RSV(J.first, 67); UZ_RSV(J.runid, 68); UZ_RSV(J.pos_of_k, 69);
                UZ_RSV(J.fo, 70); UZ_RSV(J.gidx, 71); UZ_RSV(J.s5_in, 72); UZ_RSV(J.s5_out, 73); UZ_RSV(J.kept, 74); UZ_RSV(J.name_rec, 75); UZ_RSV(J.ccount, 76);
                UZ_RSV(J.cspan, 77); UZ_RSV(J.totals, 78);
#undef UZ_RSV
            } catch (...) {
                std::lock_guard<std::mutex> lk(c->err_mu);
                w.busy = false;
                throw;
            }
            std::lock_guard<std::mutex> lk(c->err_mu);
            w.busy = false;
        }
    });
}

int uz_walk_slot_stats(uz_ctx *c, int64_t out[8]) {
    return guarded(c, [&] {
        UZ_REQUIRE(out != nullptr, UZ_E_ARG, "null output");
        std::lock_guard<std::mutex> lk(c->walk_mu);
        size_t parked = 0;
        for (auto &b : c->walk_park) parked += b.second;
        out[0] = c->walk_allocs; out[1] = (int64_t)c->walk_park.size(); out[2] = (int64_t)parked; out[3] = 0;
        for (int i = 0; i < uz_ctx::WALK_SLOTS && i < 4; i++) out[4 + i] = (int64_t)c->walk[i].out.cap;
    });
}

// The table of a batch whose joins ran on the device: the kept list lies in the slot (uz_bam_join), the records are unpacked where they lie.
int uz_reads_from_walk(uz_ctx *c, int walk_id, int32_t min_base_qual, int want_names, int *reads_id, int64_t totals[8]) {
    return guarded(c, [&] {
        UZ_REQUIRE(walk_id >= 0 && walk_id < uz_ctx::WALK_SLOTS && c->walk[walk_id].busy && reads_id, UZ_E_ARG, "bad walk id");
        uz_ctx::WalkSlot &w = c->walk[walk_id];
        auto &J = w.join;
        UZ_REQUIRE(J.done, UZ_E_STATE, "uz_reads_from_walk: the joins of this batch are not finished (uz_bam_join until it needs nothing)");
        const int64_t n = J.tot_h[JT_N], n_cigar_total = J.tot_h[JT_CIGAR], n_row_units = J.tot_h[JT_UNITS], n_seq_units = J.tot_h[JT_SEQ_UNITS], names_bytes = J.tot_h[JT_NAME_BYTES];
        const int64_t n_qnames = J.tot_h[JT_QNAMES];
        const int32_t n_contigs = J.n_ref;
        UZ_REQUIRE(n < (int64_t)0x7FFFFFF0 && n_cigar_total < ((int64_t)1 << 32) && n_row_units < ((int64_t)1 << 32) && names_bytes < ((int64_t)1 << 32), UZ_E_RANGE,
                   "uz_reads_from_walk: the batch does not fit the table's 32-bit offsets");
        if (totals) for (int k = 0; k < 8; k++) totals[k] = J.tot_h[k];
        DevBlock blk;
        uz_kept_rec *d_kept = nullptr;
        uint8_t *mapq = nullptr, *aux_col = nullptr, *seq4 = nullptr, *d_names = nullptr;
        int64_t *d_coff = nullptr;
        int32_t *d_span = nullptr, *start = nullptr, *tlen = nullptr, *mate = nullptr, *err = nullptr;
        uint32_t *qname = nullptr, *cigar = nullptr, *plane = nullptr, *name_rec = nullptr;
        uint16_t *flag = nullptr, *l_seq = nullptr, *n_cigar = nullptr;
        for (int pass = 0; pass < 2; pass++) {
            Carver cv(pass ? blk.p : nullptr);
            d_coff = cv.take<int64_t>((size_t)n_contigs + 1); d_span = cv.take<int32_t>((size_t)n_contigs + 1); err = cv.take<int32_t>(4);
            start = cv.take<int32_t>((size_t)n); tlen = cv.take<int32_t>((size_t)n); mate = cv.take<int32_t>((size_t)n); qname = cv.take<uint32_t>((size_t)n);
            flag = cv.take<uint16_t>((size_t)n); l_seq = cv.take<uint16_t>((size_t)n); n_cigar = cv.take<uint16_t>((size_t)n);
            mapq = cv.take<uint8_t>((size_t)n); aux_col = cv.take<uint8_t>((size_t)n);
            cigar = cv.take<uint32_t>((size_t)n_cigar_total); seq4 = cv.take<uint8_t>((size_t)n_seq_units * UZ_SEQ4_UNIT_BYTES);
            plane = cv.take<uint32_t>((size_t)n_row_units);
            if (want_names) { d_kept = cv.take<uz_kept_rec>((size_t)n); name_rec = cv.take<uint32_t>((size_t)n_qnames); d_names = cv.take<uint8_t>((size_t)names_bytes); }
            if (!pass) blk = uz_block_get(c, cv.off + 256);
        }
        int id = -1;
        try {
            hipStream_t st = c->stream; // (the joins ran on the slot's stream and were waited for: uz_bam_join returns behind them)
            UZ_HIP(hipMemcpyAsync(d_coff, J.contig_off_h.data(), ((size_t)n_contigs + 1) * 8, hipMemcpyHostToDevice, st));
            if (n_contigs) UZ_HIP(hipMemcpyAsync(d_span, J.max_span_h.data(), (size_t)n_contigs * 4, hipMemcpyHostToDevice, st));
            UZ_HIP(hipMemsetAsync(err, 0, 16, st));
            uz_launch_bam_extract(c, st, n, w.out.p, w.out_bytes, J.aux.p, J.aux_bytes, J.kept.p, min_base_qual, start, tlen, mate, qname, flag, l_seq, n_cigar, mapq, aux_col,
                                  cigar, seq4, plane, err, want_names ? d_names : nullptr, n_cigar_total, n_row_units, n_seq_units, names_bytes);
            if (want_names && n) {
                UZ_HIP(hipMemcpyAsync(d_kept, J.kept.p, (size_t)n * sizeof(uz_kept_rec), hipMemcpyDeviceToDevice, st));
                if (n_qnames) UZ_HIP(hipMemcpyAsync(name_rec, J.name_rec.p, (size_t)n_qnames * 4, hipMemcpyDeviceToDevice, st));
            }
            int32_t e = 0;
            UZ_HIP(hipMemcpyAsync(&e, err, 4, hipMemcpyDeviceToHost, st));
            UZ_HIP(hipStreamSynchronize(st));
            UZ_REQUIRE(e != 2, UZ_E_RANGE, "uz_reads_from_walk: an offset of the kept list points beyond the stores its totals declare");
            UZ_REQUIRE(e == 0, UZ_E_RANGE, "uz_reads_from_walk: a kept record lies outside the walked bytes, or overruns its block_size");
            uz_reads_packed_view v;
            memset(&v, 0, sizeof(v));
            v.n_segs = n; v.n_contigs = n_contigs; v.contig_off = d_coff; v.max_span = d_span;
            v.start = start; v.tlen = tlen; v.mate = mate; v.qname = qname; v.flag = flag; v.l_seq = l_seq; v.n_cigar = n_cigar; v.mapq = mapq; v.aux = aux_col;
            v.n_cigar_total = n_cigar_total; v.cigar = cigar; v.n_row_units = n_row_units; v.n_seq_units = n_seq_units; v.seq4 = seq4;
            v.qlow = reinterpret_cast<const uint8_t *>(plane); v.min_base_qual = min_base_qual; v.n_qnames = (uint32_t)n_qnames;
            const int rc = uz_reads_adopt_device(c, &v, &id);
            if (rc) throw UzError{rc, c->err};
            ReadsDev &r = c->reads[(size_t)id];
            r.mirror = blk;
            if (want_names) { r.kept_list = d_kept; r.name_rec = name_rec; r.names = d_names; r.names_bytes = names_bytes; }
        } catch (...) { uz_block_put(c, blk); throw; }
        { std::lock_guard<std::mutex> lk(c->err_mu); w.busy = false; }
        *reads_id = id;
    });
}

// the read names of name ids of such a table: off [n + 1] (host) and the bytes back to back in page-locked memory of the CONTEXT (valid until the
// next call).  The context's own buffers, grown as needed, and copy kernels out of them: a first version made and freed four device blocks per call --
// hipFree waits for the whole device, the next chunk's read stage included: 8 ms per chunk on the calling thread, a tenth of the product's drop-in call.
int uz_reads_names(uz_ctx *c, int reads_id, const uint32_t *ids, int64_t n, int64_t *off, const uint8_t **bytes) {
    return guarded(c, [&] {
        UZ_REQUIRE(reads_id >= 0 && (size_t)reads_id < c->reads.size() && c->reads[(size_t)reads_id].live, UZ_E_ARG, "bad reads id");
        ReadsDev &r = c->reads[(size_t)reads_id];
        UZ_REQUIRE(r.kept_list && r.name_rec, UZ_E_STATE, "uz_reads_names: a table built by uz_reads_from_walk with names");
        UZ_REQUIRE(n >= 0 && off && bytes && (n == 0 || ids), UZ_E_ARG, "bad arguments");
        off[0] = 0;
        *bytes = nullptr;
        if (n == 0) return;
        for (int64_t k = 0; k < n; k++) UZ_REQUIRE(ids[k] < r.n_qnames, UZ_E_ARG, "uz_reads_names: a name id out of range");
        hipStream_t st = c->stream;
        auto pin = [&](size_t bytes_needed) {
            if (c->nm_pin_cap >= bytes_needed) return;
            if (c->nm_pin) (void)hipHostFree(c->nm_pin);
            c->nm_pin = nullptr; c->nm_pin_cap = 0;
            const size_t want = bytes_needed + bytes_needed / 4 + 4096;
            uint8_t *p = nullptr;
            UZ_HIP(hipHostMalloc((void **)&p, want, hipHostMallocDefault));
            c->nm_pin = p; c->nm_pin_cap = want;
        };
        c->nm_ids.ensure((size_t)n + 16); c->nm_len.ensure((size_t)n + 16); c->nm_off.ensure((size_t)n + 16);
        pin((size_t)n * 4 + 64);
        memcpy(c->nm_pin, ids, (size_t)n * 4);
        uz_kcopy(c, c->nm_ids.p, c->nm_pin, (size_t)n * 4);
        uz_launch_name_lens(c, st, n, c->nm_ids.p, r.name_rec, r.kept_list, r.n, r.names_bytes, c->nm_len.p);
        UZ_HIP(hipStreamSynchronize(st)); // (the ids have left the staging block)
        uz_kcopy(c, c->nm_pin, c->nm_len.p, (size_t)n * 4);
        UZ_HIP(hipStreamSynchronize(st));
        const uint32_t *len = reinterpret_cast<const uint32_t *>(c->nm_pin);
        for (int64_t k = 0; k < n; k++) off[k + 1] = off[k] + (int64_t)len[k];
        const int64_t total = off[n];
        UZ_REQUIRE(total < ((int64_t)1 << 32), UZ_E_RANGE, "uz_reads_names: more than 4 GiB of names");
        if (total == 0) { *bytes = c->nm_pin; return; }
        std::vector<uint32_t> o32((size_t)n);
        for (int64_t k = 0; k < n; k++) o32[(size_t)k] = (uint32_t)off[k];
        pin(std::max((size_t)n * 4, (size_t)total) + 64);
        memcpy(c->nm_pin, o32.data(), (size_t)n * 4);
        c->nm_out.ensure((size_t)total + 64);
        uz_kcopy(c, c->nm_off.p, c->nm_pin, (size_t)n * 4);
        uz_launch_name_gather(c, st, n, c->nm_ids.p, r.name_rec, r.kept_list, r.names, c->nm_len.p, c->nm_off.p, c->nm_out.p);
        UZ_HIP(hipStreamSynchronize(st)); // (the offsets have left the staging block before the names land in it)
        uz_kcopy(c, c->nm_pin, c->nm_out.p, ((size_t)total + 3) & ~(size_t)3);
        UZ_HIP(hipStreamSynchronize(st));
        *bytes = c->nm_pin;
    });
}

int uz_reads_from_bam(uz_ctx *c, int walk_id, const uz_kept_rec *kept, int64_t n, const uint8_t *aux, int64_t aux_bytes, const int64_t *contig_off,
                      const int32_t *max_span, int32_t n_contigs, int64_t n_cigar_total, int64_t n_row_units, int64_t n_seq_units, uint32_t n_qnames,
                      int32_t min_base_qual, uint8_t *names_out, int64_t names_bytes, int *reads_id) {
    return guarded(c, [&] {
        UZ_REQUIRE(walk_id >= 0 && walk_id < uz_ctx::WALK_SLOTS && c->walk[walk_id].busy, UZ_E_ARG, "bad walk id");
        UZ_REQUIRE(names_bytes >= 0 && (names_out == nullptr || names_bytes < ((int64_t)1 << 32)), UZ_E_ARG, "bad name store size");
        UZ_REQUIRE(reads_id && n >= 0 && n < (int64_t)0x7FFFFFF0 && (n == 0 || kept) && aux_bytes >= 0 && (aux_bytes == 0 || aux) && contig_off && max_span && n_contigs >= 0 &&
                       n_cigar_total >= 0 && n_cigar_total < ((int64_t)1 << 32) && n_row_units >= 0 && n_row_units < ((int64_t)1 << 32) && n_seq_units >= 0 && n_seq_units <= n_row_units,
                   UZ_E_ARG, "bad arguments");
        uz_ctx::WalkSlot &w = c->walk[walk_id];
        static const bool rfb_log = getenv("UZ_RFB_LOG") != nullptr; // development aid: where the call's time goes
        const auto t0 = std::chrono::steady_clock::now();
        auto since = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
        double t_blk = 0, t_enq = 0, t_sync = 0;
        // one block for the kept list, the aux bytes and the columns the records are unpacked into; the table adopts the columns in place
        // (uz_reads_adopt_device: cigar, seq4 and the quality plane ARE the device's stores) and keeps the block as its `mirror`
        DevBlock blk;
        uz_kept_rec *d_kept = nullptr;
        uint8_t *d_aux = nullptr, *mapq = nullptr, *aux_col = nullptr, *seq4 = nullptr, *d_names = nullptr;
        int64_t *d_coff = nullptr;
        int32_t *d_span = nullptr, *start = nullptr, *tlen = nullptr, *mate = nullptr, *err = nullptr;
        uint32_t *qname = nullptr, *cigar = nullptr, *plane = nullptr;
        uint16_t *flag = nullptr, *l_seq = nullptr, *n_cigar = nullptr;
        for (int pass = 0; pass < 2; pass++) {
            Carver cv(pass ? blk.p : nullptr);
            d_kept = cv.take<uz_kept_rec>((size_t)n); d_aux = cv.take<uint8_t>((size_t)aux_bytes + 64);
            d_coff = cv.take<int64_t>((size_t)n_contigs + 1); d_span = cv.take<int32_t>((size_t)n_contigs + 1); err = cv.take<int32_t>(4);
            start = cv.take<int32_t>((size_t)n); tlen = cv.take<int32_t>((size_t)n); mate = cv.take<int32_t>((size_t)n); qname = cv.take<uint32_t>((size_t)n);
            flag = cv.take<uint16_t>((size_t)n); l_seq = cv.take<uint16_t>((size_t)n); n_cigar = cv.take<uint16_t>((size_t)n);
            mapq = cv.take<uint8_t>((size_t)n); aux_col = cv.take<uint8_t>((size_t)n);
            cigar = cv.take<uint32_t>((size_t)n_cigar_total); seq4 = cv.take<uint8_t>((size_t)n_seq_units * UZ_SEQ4_UNIT_BYTES);
            plane = cv.take<uint32_t>((size_t)n_row_units);
            if (names_out) d_names = cv.take<uint8_t>((size_t)names_bytes);
            if (!pass) blk = uz_block_get(c, cv.off + 256);
        }
        int id = -1;
        t_blk = since();
        try {
            hipStream_t st = c->stream;
            if (n) UZ_HIP(hipMemcpyAsync(d_kept, kept, (size_t)n * sizeof(uz_kept_rec), hipMemcpyHostToDevice, st));
            if (aux_bytes) UZ_HIP(hipMemcpyAsync(d_aux, aux, (size_t)aux_bytes, hipMemcpyHostToDevice, st));
            UZ_HIP(hipMemcpyAsync(d_coff, contig_off, ((size_t)n_contigs + 1) * 8, hipMemcpyHostToDevice, st));
            if (n_contigs) UZ_HIP(hipMemcpyAsync(d_span, max_span, (size_t)n_contigs * 4, hipMemcpyHostToDevice, st));
            UZ_HIP(hipMemsetAsync(err, 0, 16, st));
            uz_launch_bam_extract(c, st, n, w.out.p, w.out_bytes, d_aux, aux_bytes, d_kept, min_base_qual, start, tlen, mate, qname, flag, l_seq, n_cigar, mapq, aux_col,
                                  cigar, seq4, plane, err, names_out ? d_names : nullptr, n_cigar_total, n_row_units, n_seq_units, names_bytes);
            if (names_out && names_bytes) UZ_HIP(hipMemcpyAsync(names_out, d_names, (size_t)names_bytes, hipMemcpyDeviceToHost, st));
            int32_t e = 0;
            UZ_HIP(hipMemcpyAsync(&e, err, 4, hipMemcpyDeviceToHost, st));
            t_enq = since();
            UZ_HIP(hipStreamSynchronize(st));
            t_sync = since();
            UZ_REQUIRE(e != 2, UZ_E_RANGE, "uz_reads_from_bam: an offset of the kept list points beyond the stores its totals declare");
            UZ_REQUIRE(e == 0, UZ_E_RANGE, "uz_reads_from_bam: a kept record lies outside the walked bytes, or overruns its block_size");
            uz_reads_packed_view v;
            memset(&v, 0, sizeof(v));
            v.n_segs = n; v.n_contigs = n_contigs; v.contig_off = d_coff; v.max_span = d_span;
            v.start = start; v.tlen = tlen; v.mate = mate; v.qname = qname; v.flag = flag; v.l_seq = l_seq; v.n_cigar = n_cigar; v.mapq = mapq; v.aux = aux_col;
            v.n_cigar_total = n_cigar_total; v.cigar = cigar; v.n_row_units = n_row_units; v.n_seq_units = n_seq_units; v.seq4 = seq4;
            v.qlow = reinterpret_cast<const uint8_t *>(plane); v.min_base_qual = min_base_qual; v.n_qnames = n_qnames;
            const int rc = uz_reads_adopt_device(c, &v, &id);
            if (rc) throw UzError{rc, c->err};
            c->reads[(size_t)id].mirror = blk;
        } catch (...) { uz_block_put(c, blk); throw; }
        { std::lock_guard<std::mutex> lk(c->err_mu); w.busy = false; }
        if (rfb_log) fprintf(stderr, "[uz] reads_from_bam n=%lld kept %.1f MB names %.1f MB: block %.2f, enqueue %.2f, sync %.2f, adopt %.2f ms\n", (long long)n, n * 32 / 1e6, names_bytes / 1e6, t_blk, t_enq - t_blk, t_sync - t_enq, since() - t_sync);
        *reads_id = id;
    });
}

int uz_crc32_blocks(uz_ctx *c, const uint8_t *data, int64_t n_blocks, const int64_t *off, const uint32_t *want, int64_t *first_bad) {
    return guarded(c, [&] {
        UZ_REQUIRE(n_blocks >= 0 && first_bad && (n_blocks == 0 || (data && off && want)), UZ_E_ARG, "bad arguments");
        *first_bad = -1;
        if (n_blocks == 0) return;
        for (int64_t k = 0; k < n_blocks; k++) UZ_REQUIRE(off[k] >= 0 && off[k] <= off[k + 1] && off[k + 1] - off[k] <= 65536, UZ_E_ARG, "bad block table");
        uint8_t *d = upload(c, data, (size_t)off[n_blocks]);
        int64_t *d_off = upload(c, off, (size_t)n_blocks + 1);
        uint32_t *d_want = upload(c, want, (size_t)n_blocks);
        int32_t *d_err = nullptr, e = 0;
        UZ_HIP(hipMalloc((void **)&d_err, 64));
        uz_launch_crc32(c, c->stream, n_blocks, d, d_off, d_want, d_err);
        UZ_HIP(hipMemcpyAsync(&e, d_err, 4, hipMemcpyDeviceToHost, c->stream));
        UZ_HIP(hipStreamSynchronize(c->stream));
        (void)hipFree(d); (void)hipFree(d_off); (void)hipFree(d_want); (void)hipFree(d_err);
        if (e) *first_bad = e - 1;
    });
}

} // extern "C"
