// abi.hip -- the extern "C" entry points of include/unfazed_hip.h: context, staging of
// the decoded columns into HBM, and the launch sequence of the stages.
#include "uz_ctx.hpp"

void uz_fold_complex(uz_ctx *c, uint8_t *gt, const uint8_t *sflags, int64_t n);
bool uz_site_scan_fresh(const uz_ctx *c, const FamilyDev &f, bool need_cnv);
void uz_build_coarse(uz_ctx *c, ReadsDev &r);

namespace {

template <typename F>
int guarded(uz_ctx *c, F &&fn) {
    if (!c) return UZ_E_ARG;
    try {
        (void)hipSetDevice(c->device);
        fn();
        return 0;
    } catch (const UzError &e) {
        c->err = e.msg;
        return e.code;
    } catch (const std::exception &e) {
        c->err = e.what();
        return UZ_E_ARG;
    } catch (...) {
        c->err = "unknown error";
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
    return c->sites[id];
}
FamilyDev &fam_of(uz_ctx *c, int id) {
    UZ_REQUIRE(id >= 0 && id < (int)c->fams.size() && c->fams[id].live, UZ_E_ARG, "unknown family handle");
    return c->fams[id];
}
ReadsDev &reads_of(uz_ctx *c, int id) {
    UZ_REQUIRE(id >= 0 && id < (int)c->reads.size() && c->reads[id].live, UZ_E_ARG, "unknown reads handle");
    return c->reads[id];
}

void free_family(FamilyDev &f) {
    if (f.owned) {
        (void)hipFree(f.gt);
        for (int m = 0; m < 3; m++) { (void)hipFree(f.rd[m]); (void)hipFree(f.ad[m]); (void)hipFree(f.gq[m]); }
    }
    (void)hipFree(f.cls);
    f = FamilyDev();
}
void free_sites(SitesDev &s) {
    if (s.owned) {
        (void)hipFree(s.pos); (void)hipFree(s.sflags); (void)hipFree(s.ref_base); (void)hipFree(s.alt_base);
    }
    (void)hipFree(s.contig_off);
    s = SitesDev();
}
void free_reads(ReadsDev &r) {
    if (r.owned) {
        (void)hipFree(r.start); (void)hipFree(r.end); (void)hipFree(r.flag); (void)hipFree(r.mapq); (void)hipFree(r.aux);
        (void)hipFree(r.tlen); (void)hipFree(r.qname); (void)hipFree(r.mate); (void)hipFree(r.cigar_off);
        (void)hipFree(r.n_cigar); (void)hipFree(r.cigar); (void)hipFree(r.l_seq); (void)hipFree(r.sq_off16);
        (void)hipFree(r.seq); (void)hipFree(r.qual);
    }
    (void)hipFree(r.contig_off); (void)hipFree(r.max_span); (void)hipFree(r.qc); (void)hipFree(r.need); (void)hipFree(r.coarse); (void)hipFree(r.rec_a); (void)hipFree(r.rec_b);
    r = ReadsDev();
}

} // namespace

// ---------------------------------------------------------------- profiling
void uz_prof_begin(uz_ctx *c, int kernel, hipEvent_t *a, hipEvent_t *b) {
    *a = *b = nullptr;
    if (!c->prof_on) return;
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
    if (c->dn_stage_cap < off[9]) {
        if (c->dn_stage) (void)hipHostFree(c->dn_stage);
        c->dn_stage = nullptr;
        c->dn_stage_cap = off[9] + off[9] / 4 + 4096;
        UZ_HIP(hipHostMalloc((void **)&c->dn_stage, c->dn_stage_cap, hipHostMallocDefault));
    }
    // an earlier batch's copies out of this buffer have completed: every entry point ends with a stream sync
    const void *src[9] = {d->contig, d->rcontig, d->start, d->end, d->vartype, d->dflags, d->mult, d->allele_off, d->alleles};
    for (int k = 0; k < 9; k++)
        if (sizes[k]) memcpy(c->dn_stage + off[k], src[k], sizes[k]);
    c->dn.contig.ensure(n + 1); c->dn.rcontig.ensure(n + 1); c->dn.start.ensure(n + 1); c->dn.end.ensure(n + 1);
    c->dn.vartype.ensure(n + 1); c->dn.dflags.ensure(n + 1); c->dn.mult.ensure(n + 1);
    c->dn.allele_off.ensure(2 * n + 2); c->dn.alleles.ensure(nb + 1);
    void *dst[9] = {c->dn.contig.p, c->dn.rcontig.p, c->dn.start.p, c->dn.end.p, c->dn.vartype.p, c->dn.dflags.p, c->dn.mult.p,
                    c->dn.allele_off.p, c->dn.alleles.p};
    for (int k = 0; k < 9; k++)
        if (sizes[k]) UZ_HIP(hipMemcpyAsync(dst[k], c->dn_stage + off[k], sizes[k], hipMemcpyHostToDevice, c->stream));
    c->dn.cutoff = d->cutoff;
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
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return UZ_E_HIP;
    }
    *out = c;
    return 0;
}

void uz_destroy(uz_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    uz_prof_drain(c);
    uz_phase_state_free(c);
    for (auto &f : c->fams) if (f.live) free_family(f);
    for (auto &s : c->sites) if (s.live) free_sites(s);
    for (auto &r : c->reads) if (r.live) free_reads(r);
    c->dn.contig.release(); c->dn.rcontig.release(); c->dn.start.release(); c->dn.end.release();
    c->dn.vartype.release(); c->dn.dflags.release(); c->dn.mult.release(); c->dn.allele_off.release();
    c->dn.alleles.release();
    if (c->dn_stage) (void)hipHostFree(c->dn_stage);
    c->ab_lut.release(); c->win_range.release();
    c->cnt_c.release(); c->cnt_h.release(); c->cand_off.release(); c->het_off.release();
    c->cand_idx.release(); c->het_idx.release(); c->cand_flags.release();
    for (auto e : c->event_pool) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(c->stream);
    delete c;
}

const char *uz_last_error(const uz_ctx *c) { return c ? c->err.c_str() : "null context"; }

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
        s.contig_off = upload(c, v->contig_off, (size_t)v->n_contigs + 1);
        s.pos = upload(c, v->pos, (size_t)s.n);
        s.sflags = upload(c, v->sflags, (size_t)s.n);
        s.ref_base = upload(c, v->ref_base, (size_t)s.n);
        s.alt_base = upload(c, v->alt_base, (size_t)s.n);
        UZ_HIP(hipStreamSynchronize(c->stream));
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
        s.contig_off = upload(c, s.contig_off_h.data(), (size_t)v->n_contigs + 1);
        s.pos = const_cast<int32_t *>(v->pos);
        s.sflags = const_cast<uint8_t *>(v->sflags);
        s.ref_base = const_cast<uint8_t *>(v->ref_base);
        s.alt_base = const_cast<uint8_t *>(v->alt_base);
        UZ_HIP(hipStreamSynchronize(c->stream));
        c->sites[k] = s;
        *id = k;
    });
}

static void family_common(uz_ctx *c, int sites_id, FamilyDev &f) {
    SitesDev &s = sites_of(c, sites_id);
    f.live = true;
    f.sites_id = sites_id;
    UZ_HIP(hipMalloc((void **)&f.cls, (size_t)(s.n ? s.n : 1) + 64));
    uz_fold_complex(c, f.gt, s.sflags, s.n);
    UZ_HIP(hipStreamSynchronize(c->stream));
}

int uz_family_upload(uz_ctx *c, int sites_id, const uz_family_view *v, int *id) {
    return guarded(c, [&] {
        UZ_REQUIRE(v && id, UZ_E_ARG, "bad family view");
        SitesDev &s = sites_of(c, sites_id);
        FamilyDev f;
        f.owned = true;
        f.gt = upload(c, v->gt, (size_t)s.n);
        for (int m = 0; m < 3; m++) {
            f.rd[m] = upload(c, v->ref_depth[m], (size_t)s.n);
            f.ad[m] = upload(c, v->alt_depth[m], (size_t)s.n);
            f.gq[m] = upload(c, v->gq[m], (size_t)s.n);
        }
        family_common(c, sites_id, f);
        const int k = new_slot(c->fams);
        c->fams[k] = f;
        *id = k;
    });
}

int uz_family_adopt_device(uz_ctx *c, int sites_id, const uz_family_view *v, int *id) {
    return guarded(c, [&] {
        UZ_REQUIRE(v && id, UZ_E_ARG, "bad family view");
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
        family_common(c, sites_id, f);
        const int k = new_slot(c->fams);
        c->fams[k] = f;
        *id = k;
    });
}

static void reads_fill(uz_ctx *c, const uz_reads_view *v, ReadsDev &r, bool copy) {
    UZ_REQUIRE(v->n_segs >= 0 && v->n_segs < (int64_t)0x7FFFFFF0, UZ_E_RANGE, "more than 2^31 alignment records");
    r.live = true; r.owned = copy;
    r.n = v->n_segs; r.n_contigs = v->n_contigs; r.n_qnames = v->n_qnames;
    r.n_cigar_total = v->n_cigar_total; r.n_sq_bytes = v->n_sq_bytes;
    const size_t n = (size_t)r.n;
    if (copy) {
        r.contig_off = upload(c, v->contig_off, (size_t)v->n_contigs + 1);
        r.max_span = upload(c, v->max_span, (size_t)v->n_contigs);
        r.start = upload(c, v->start, n); r.end = upload(c, v->end, n);
        r.flag = upload(c, v->flag, n); r.mapq = upload(c, v->mapq, n); r.aux = upload(c, v->aux, n);
        r.tlen = upload(c, v->tlen, n); r.qname = upload(c, v->qname, n); r.mate = upload(c, v->mate, n);
        r.cigar_off = upload(c, v->cigar_off, n); r.n_cigar = upload(c, v->n_cigar, n);
        r.cigar = upload(c, v->cigar, (size_t)v->n_cigar_total);
        r.l_seq = upload(c, v->l_seq, n); r.sq_off16 = upload(c, v->sq_off16, n);
        r.seq = upload(c, v->seq, (size_t)v->n_sq_bytes); r.qual = upload(c, v->qual, (size_t)v->n_sq_bytes);
    } else {
        std::vector<int64_t> co((size_t)v->n_contigs + 1);
        std::vector<int32_t> ms((size_t)v->n_contigs + 1);
        UZ_HIP(hipMemcpy(co.data(), v->contig_off, co.size() * sizeof(int64_t), hipMemcpyDeviceToHost));
        if (v->n_contigs) UZ_HIP(hipMemcpy(ms.data(), v->max_span, (size_t)v->n_contigs * sizeof(int32_t), hipMemcpyDeviceToHost));
        r.contig_off = upload(c, co.data(), co.size());
        r.max_span = upload(c, ms.data(), (size_t)v->n_contigs);
        r.start = const_cast<int32_t *>(v->start); r.end = const_cast<int32_t *>(v->end);
        r.flag = const_cast<uint16_t *>(v->flag); r.mapq = const_cast<uint8_t *>(v->mapq); r.aux = const_cast<uint8_t *>(v->aux);
        r.tlen = const_cast<int32_t *>(v->tlen); r.qname = const_cast<uint32_t *>(v->qname); r.mate = const_cast<int32_t *>(v->mate);
        r.cigar_off = const_cast<uint32_t *>(v->cigar_off); r.n_cigar = const_cast<uint16_t *>(v->n_cigar);
        r.cigar = const_cast<uint32_t *>(v->cigar); r.l_seq = const_cast<uint16_t *>(v->l_seq);
        r.sq_off16 = const_cast<uint32_t *>(v->sq_off16);
        r.seq = const_cast<uint8_t *>(v->seq); r.qual = const_cast<uint8_t *>(v->qual);
    }
    UZ_HIP(hipMalloc((void **)&r.qc, n + 64));
    UZ_HIP(hipMalloc((void **)&r.need, n + 64));
    UZ_HIP(hipMemsetAsync(r.qc, 0, n + 64, c->stream));
    uz_build_coarse(c, r);
    uz_build_rec_headers(c, r);
    UZ_HIP(hipStreamSynchronize(c->stream));
}

int uz_reads_upload(uz_ctx *c, const uz_reads_view *v, int *id) {
    return guarded(c, [&] {
        UZ_REQUIRE(v && id, UZ_E_ARG, "bad reads view");
        ReadsDev r;
        reads_fill(c, v, r, true);
        const int k = new_slot(c->reads);
        c->reads[k] = r;
        *id = k;
    });
}
int uz_reads_adopt_device(uz_ctx *c, const uz_reads_view *v, int *id) {
    return guarded(c, [&] {
        UZ_REQUIRE(v && id, UZ_E_ARG, "bad reads view");
        ReadsDev r;
        reads_fill(c, v, r, false);
        const int k = new_slot(c->reads);
        c->reads[k] = r;
        *id = k;
    });
}

int uz_drop_derived(uz_ctx *c) {
    return guarded(c, [&] {
        for (auto &f : c->fams) f.cls_valid = false;
        for (auto &r : c->reads) r.qc_valid = false;
    });
}

int uz_sites_free(uz_ctx *c, int sites_id) {
    return guarded(c, [&] {
        SitesDev &s = sites_of(c, sites_id);
        UZ_HIP(hipStreamSynchronize(c->stream));
        for (auto &f : c->fams)
            if (f.live && f.sites_id == sites_id) free_family(f);
        free_sites(s);
        c->find_valid = false;
    });
}
int uz_reads_free(uz_ctx *c, int reads_id) {
    return guarded(c, [&] {
        ReadsDev &r = reads_of(c, reads_id);
        UZ_HIP(hipStreamSynchronize(c->stream));
        free_reads(r);
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
        c->find_valid = false;
        c->phase_valid = false;
        uz_stage_dnms(c, d);
        const bool cnv = (mode & UZ_FIND_WHOLE_REGION) != 0;
        if (!uz_site_scan_fresh(c, f, cnv)) uz_launch_site_scan(c, f, s, cnv);
        uz_launch_find(c, f, s, mode);
        c->find_fam = fam_id;
        if (cand_off) memcpy(cand_off, c->cand_off_h.data(), ((size_t)d->n + 1) * sizeof(int64_t));
        if (het_off) memcpy(het_off, c->het_off_h.data(), ((size_t)d->n + 1) * sizeof(int64_t));
    });
}

int uz_find_fetch(uz_ctx *c, int32_t *cand_idx, uint8_t *cand_flags, int32_t *het_idx) {
    return guarded(c, [&] {
        UZ_REQUIRE(c->find_valid, UZ_E_STATE, "uz_find_fetch before uz_find");
        if (cand_idx && c->n_cand)
            UZ_HIP(hipMemcpyAsync(cand_idx, c->cand_idx.p, (size_t)c->n_cand * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        if (cand_flags && c->n_cand)
            UZ_HIP(hipMemcpyAsync(cand_flags, c->cand_flags.p, (size_t)c->n_cand, hipMemcpyDeviceToHost, c->stream));
        if (het_idx && c->n_het)
            UZ_HIP(hipMemcpyAsync(het_idx, c->het_idx.p, (size_t)c->n_het * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        UZ_HIP(hipStreamSynchronize(c->stream));
    });
}

int uz_phase(uz_ctx *c, int fam_id, int reads_id, const uz_dnms_view *d, int find_mode, int32_t *status,
             int32_t *counts, int32_t *origin, int32_t *evidence) {
    return guarded(c, [&] {
        FamilyDev &f = fam_of(c, fam_id);
        SitesDev &s = sites_of(c, f.sites_id);
        ReadsDev &r = reads_of(c, reads_id);
        UZ_REQUIRE(d != nullptr, UZ_E_ARG, "null DNM view");
        UZ_REQUIRE(!(find_mode & UZ_FIND_WHOLE_REGION), UZ_E_ARG, "the read stage runs on SNV / breakpoint windows");
        // the read stage consumes the lists of a find over the same batch in SNV / breakpoint mode
        c->find_valid = false;
        c->phase_valid = false;
        uz_stage_dnms(c, d);
        if (!uz_site_scan_fresh(c, f, false)) uz_launch_site_scan(c, f, s, false);
        uz_launch_find(c, f, s, find_mode, false);
        c->find_fam = fam_id;
        uz_launch_phase(c, f, s, r, status, counts, origin, evidence);
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
    return guarded(c, [&] { c->prof_on = on != 0; });
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

} // extern "C"
