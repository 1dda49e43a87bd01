/* unfazed_io.h -- C ABI of the native input decoders (host only, no GPU needed).
 *
 * "native decode throughput".  These entry points replace what the
 * reference obtains from pysam / cyvcf2 on the way INTO the hot path:
 *
 *   uz_bam_decode      pysam.AlignmentFile(bam) + the per-read accessors used in
 *                      unfazed/read_collector.py:11-25 (first tlen values), :372-392 (open, fetch),
 *                      :28-53 / :56-73 / :394-425 (flag, mapq, cigar, sequence, qualities, SA tag)
 *                      and bamfile.mate(read) (:402, :186, :509).  One multi-threaded pass turns the
 *                      whole BGZF file into the column table `uz_reads_upload` takes
 *                      (include/unfazed_hip.h), so region fetches and mate look-ups become index
 *                      arithmetic on the device.
 *   uz_vcf_decode      cyvcf2.VCF(sites) + Variant.start/end/REF/ALT/gt_types/gt_ref_depths/
 *                      gt_alt_depths/gt_quals (unfazed/informative_site_finder.py:213-339, :571-600):
 *                      text VCF (plain or gzip/bgzip) or BCF2 -> site columns for all samples.
 *
 * Error convention: 0 = success, negative = failure with a message in uz_io_last_error()
 * (thread local).  Nothing is written to stdout / stderr.  Handles own their memory; the views they
 * hand out stay valid until the handle is freed.
 */
#ifndef UNFAZED_IO_H
#define UNFAZED_IO_H
#include <stdint.h>
#include "uz_types.h"
#include "uz_bamwalk.h"

#ifdef __cplusplus
extern "C" {
#endif

#define UZ_IO_OK 0
#define UZ_IO_E_OPEN (-1)     /* file missing / unreadable */
#define UZ_IO_E_FORMAT (-2)   /* not BGZF / BAM / VCF, truncated or corrupt (CRC, sizes) */
#define UZ_IO_E_UNSORTED (-3) /* records not coordinate sorted / not grouped by contig */
#define UZ_IO_E_RANGE (-4)    /* a value does not fit the column types (l_seq > 65535, > 2^31 records ...) */
#define UZ_IO_E_ARG (-5)

const char *uz_io_last_error(void);

/* ------------------------------------------------------------------ BAM */
typedef struct uz_bam uz_bam;

/* Decode a whole BAM with `threads` worker threads (<= 0: all hardware threads). */
int uz_bam_decode(const char *path, int threads, uz_bam **out);
/* Region decode through the BAI index (`bai_path` NULL: NAME.bam.bai, then NAME.bai): only the BGZF blocks the index
 * names for the intervals are read and inflated.  The table holds the records `fetch(contig, lo, hi)` returns for the
 * intervals (start < hi and end > lo; read_collector.py:385, :167, :478-497) and, closed under it, the records `mate()`
 * returns for them (:400, :185: overlap of the mate position, same name) -- everything of the file the read stage can
 * look at -- in file order, with the same columns, name ids in order of first appearance and mate links as a whole-file
 * decode restricted to those records.  head_records: how many template lengths of the START of the file to keep for
 * uz_bam_tlen_head (estimate_concordant_insert_len, :11-25). */
int uz_bam_decode_regions(const char *path, const char *bai_path, int64_t n_iv, const int32_t *tid, const int32_t *lo, const int32_t *hi,
                          int64_t head_records, int threads, uz_bam **out);
/* what the last decode touched: [0] compressed file bytes read, [1] BGZF blocks inflated, [2] records walked, [3] records kept */
/* The same table from an UNCOMPRESSED BAM stream in memory (magic, header text, references, records) */
int uz_bam_decode_memory(const uint8_t *stream, int64_t n, int threads, uz_bam **out);
void uz_bam_io_stats(const uz_bam *h, int64_t out[4]);
void uz_bam_free(uz_bam *h);

int32_t uz_bam_n_contigs(const uz_bam *h);
const char *uz_bam_contig_name(const uz_bam *h, int32_t i);
int32_t uz_bam_contig_length(const uz_bam *h, int32_t i);
/* records in the file / records kept in the table (reference id >= 0) */
int64_t uz_bam_n_file_records(const uz_bam *h);
int64_t uz_bam_n_records(const uz_bam *h);
/* borrowed pointers into the handle: exactly the struct uz_reads_upload() takes */
int uz_bam_view(const uz_bam *h, uz_reads_view *out);
/* query name of an interned id (ids are assigned in order of first appearance in the file) */
const char *uz_bam_qname(const uz_bam *h, uint32_t id, int32_t *len);
/* template lengths of the first `cap` records of the FILE (estimate_concordant_insert_len reads the
 * head of the file, read_collector.py:11-25); returns the number written */
int64_t uz_bam_tlen_head(const uz_bam *h, int32_t *out, int64_t cap);
/* wall-clock seconds of the stages of the last decode: [0] read, [1] inflate, [2] columns, [3] names + mates */
void uz_bam_timing(const uz_bam *h, double out[4]);

/* (CRAM input is refused by the session with a clear message -- unfazed_amd/session.py -- since round 6: the decoder of rounds 2 - 5 could only be held
 * against the repo's own writer, never against htslib, and an alignment decoder that nothing pins is not something to phase variants through.) */

/* ------------------------------------------------------------------ VCF */
typedef struct uz_vcf uz_vcf;

typedef struct uz_vcf_view {
    int64_t n_sites;
    int32_t n_samples;
    int32_t n_contigs;
    const int64_t *contig_off; /* [n_contigs+1] */
    const int32_t *pos;        /* 0-based start */
    const int32_t *end;        /* INFO/END when present, else start + len(REF) */
    const uint8_t *sflags;     /* UZ_SF_COMPLEX */
    const uint8_t *ref_base;   /* 0 for complex records */
    const uint8_t *alt_base;
    const uint8_t *gt;         /* [n_samples][n_sites] cyvcf2 gt_types codes 0/1/2/3 */
    const int32_t *ref_depth;  /* [n_samples][n_sites], -1 = missing */
    const int32_t *alt_depth;
    const double *gq;          /* [n_samples][n_sites], -1 = missing */
} uz_vcf_view;

int uz_vcf_decode(const char *path, int threads, uz_vcf **out);
/* Region decode through the file's index (`tbi_path` NULL: NAME.tbi next to the file, else NAME.csi): only the BGZF blocks the
 * index names for the intervals are read and inflated.  The table holds the header and, in file order, the records that overlap an
 * interval [lo, hi) (0-based) of reference `ref[k]` (an index into uz_vcf_index_names) -- what `vcf(region)` hands the
 * reference per DNM (informative_site_finder.py:42, :399-420, :566, and :41-43, :213 for a .bcf); a record is kept once however many
 * intervals it overlaps.  A BGZF VCF goes through a TBI or a CSI (`tabix -C`: any min_shift / depth), a BCF through its CSI
 * (`bcftools index`): its references are the header's contigs by id, a record overlaps by POS and rlen. */
int uz_vcf_decode_regions(const char *path, const char *tbi_path, int64_t n_iv, const int32_t *ref, const int32_t *lo, const int32_t *hi,
                          int threads, uz_vcf **out);
/* sequence names of the index, each NUL-terminated, into buf -- a text file's in the order of their first record in the file (the
 * TBI's, or the CSI's aux block), a BCF's the header's contigs by id; returns the bytes needed (call with cap 0 first) or a negative UZ_IO_E_* */
int64_t uz_vcf_index_names(const char *path, const char *tbi_path, char *buf, int64_t cap);
/* What the index readers make of a BAI (kind 0) or TBI (kind 1) file, for checks against an independent reader: per reference six
 * numbers -- bins (the 37450 pseudo-bin left out), chunks in them, linear-index entries, sum of the chunks' begins, of their ends, of
 * the linear entries (each modulo 2^62).  Returns the number of references (fills at most cap_refs of them) or a negative UZ_IO_E_*. */
int64_t uz_index_summary(const char *path, int kind, int64_t *out, int64_t cap_refs);
/* what the last decode touched: [0] compressed file bytes read, [1] BGZF blocks inflated (region decode), [2] lines walked, [3] records kept */
void uz_vcf_io_stats(const uz_vcf *h, int64_t out[4]);
void uz_vcf_free(uz_vcf *h);
int uz_vcf_view_get(const uz_vcf *h, uz_vcf_view *out);
const char *uz_vcf_sample(const uz_vcf *h, int32_t i);
const char *uz_vcf_contig(const uz_vcf *h, int32_t i);
/* REF string and comma-joined ALT string of record i (get_refalt, snv_phaser.py:73-84) */
const char *uz_vcf_ref(const uz_vcf *h, int64_t i, int32_t *len);
const char *uz_vcf_alt(const uz_vcf *h, int64_t i, int32_t *len);
/* header lines (incl. #CHROM) joined by '\n' and the raw text line of record i (VCF writer) */
const char *uz_vcf_header(const uz_vcf *h, int64_t *len);
const char *uz_vcf_line(const uz_vcf *h, int64_t i, int32_t *len);
/* text value of INFO/<key> of record i (e.g. SVTYPE: `unfazed.py:67-83` reads it through
 * variant.INFO.get); NULL when the key is absent or a flag; for BCF only string-typed values are returned */
const char *uz_vcf_info(const uz_vcf *h, int64_t i, const char *key, int32_t *len);
/* 1 when the file was BCF (no text lines: uz_vcf_line is empty) */
int uz_vcf_is_bcf(const uz_vcf *h);

/* ------------------------------------------------------------------ staged (packed) records
 * uz_reads_packed_view (uz_types.h) is what crosses the host link.  The caller owns the output buffers (pinned
 * memory from uz_pinned_alloc for the upload): the `out` view arrives with every pointer set to a WRITABLE
 * buffer of the right size -- [n] for the per-record columns, n_cigar_total words, n_row_units * 16 / * 4 bytes
 * for seq4 / qlow (or n_row_units * 8 for seq2 and n_exc entries of exc_*), [n_contigs + 1] / [n_contigs] for the contig
 * tables -- and the scalar fields are filled in.  A selection keeps the base-row form of its source; its qualities are
 * lists when out->n_low is set (from either form of the source), the plane otherwise (plane sources only). */
/* sizes of the packed form of an ASCII table */
int uz_reads_pack_sizes(const uz_reads_view *in, int64_t *n_cigar_total, int64_t *n_row_units);
/* records whose CIGAR is one M / = / X operation spanning the read: with out->cigar_compact their words stay home (out->n_cigar_total
 * then is the plain total minus this, out->n_cigar_omitted this) */
int uz_reads_pack_cigar_omitted(const uz_reads_view *in, int threads, int64_t *n_omitted);
/* 1 when every record's `end` is what htslib's bam_endpos gives for its CIGAR (true of a BAM decoder's table): the packed form
 * may then leave the column out (out->end = NULL) and the device derives it */
int uz_reads_pack_end_derivable(const uz_reads_view *in, int threads, int32_t *yes);
/* the quality plane as lists (uz_types.h): entries of qlow_pos for the threshold, and whether a read is longer than 256 bases */
int uz_reads_pack_lists(const uz_reads_view *in, int min_base_qual, int threads, int64_t *n_qlow_pos, int32_t *wide);
/* number of bases that are not A/C/G/T: the length of the exc_* columns of the two-bit form */
int uz_reads_pack_exceptions(const uz_reads_view *in, int threads, int64_t *n_exc);
/* ASCII table -> packed columns for the base-quality threshold `min_base_qual` (= --min-gt-qual);
 * out->n_cigar_total / n_row_units must hold the sizes the buffers were made for.  Base rows: two-bit (out->seq2 set,
 * n_row_units * 8 bytes, plus exc_rec / exc_pos / exc_code of out->n_exc entries) or four-bit (out->seq2 null: out->seq4).
 * Qualities: lists (out->n_low set: [n], plus qlow_pos of out->n_qlow_pos entries, out->qlow_pos_wide as uz_reads_pack_lists
 * says) or the plane (out->n_low null: out->qlow, n_row_units * 4 bytes) */
int uz_reads_pack(const uz_reads_view *in, int min_base_qual, int threads, uz_reads_packed_view *out);

/* Fetch-reach selection: the records `bamfile.fetch(contig, lo, hi)` returns for a list of fetches (start <
 * hi and end > lo: read_collector.py:385, :167) plus the records `bamfile.mate()` returns for them (:400,
 * :185) -- everything of an alignment file the reference can ever look at for those fetches.  Record order and
 * query-name ids are kept, mate links are renumbered, max_span is recomputed. */
typedef struct uz_psrc uz_psrc;     /* a packed table opened as the source of selections (host pointers, borrowed) */
typedef struct uz_select uz_select; /* one selection */
int uz_reads_source_open(const uz_reads_packed_view *full, int threads, uz_psrc **out);
void uz_reads_source_close(uz_psrc *src);
/* all_bases = 0: a record a fetch returns keeps its bases, a record reachable only as a mate is staged WITHOUT them
 * (UZ_AUX_NO_SEQ: quality-plane row, no seq4 row) -- every base the extended read stage reads lies at a fetch point the
 * record overlaps.  all_bases = 1 (--no-extended: the join reads the mates of the DNM reads at candidate sites nobody
 * fetched): every kept record keeps its bases. */
/* unit_masks = 1 (ignored with all_bases): a record whose single CIGAR operation spans the read keeps only the 32-base units
 * of its rows that hold a fetched position -- position hi - 1 of a one- or two-base fetch, and extra[f] bases on (extra: NULL
 * = 0; for the fetch at a DNM the length of its longer allele) -- see uz_reads_packed_view.umask; the output then needs
 * the umask column and the list form of the qualities.
 * unit_masks = 3 (SV batches): as 1, and a fetch wider than two bases stages NO unit of the records it returns: collect_reads_sv
 * (read_collector.py:476-596) looks at flags, CIGARs and mates only; the bases of its records are read at the het sites they overlap
 * -- one-base fetches of the same batch -- and the only quality bits the read stage ever asks for are those of records that pass
 * goodread (:43-46) at such sites, so the list form serves SV batches too. */
/* tuples: 0, or 1 (build the dictionary of the small columns, uz_reads_packed_view.tup) | 2 (the output will set cigar_compact) | 4 (the
 * output will take the qualities as lists: n_low joins the combination); uz_select_n_tuples then gives the table length, or -1
 * when the selection holds more than 65536 combinations (the output keeps the plain columns) */
int uz_reads_select_plan(const uz_psrc *src, int64_t n_fetch, const int32_t *contig, const int32_t *lo, const int32_t *hi, int all_bases,
                         int unit_masks, const uint16_t *extra, int tuples, int threads, uz_select **out);
int64_t uz_select_n_tuples(const uz_select *s);
int64_t uz_select_n_esc16(const uz_select *s); /* entries of esc16_* when the output takes start / tlen / mate / qname as 16-bit differences */
int64_t uz_select_n_esc16_start8(const uz_select *s); /* ... when it takes the start differences in eight bits (start_d8) */
int64_t uz_select_n_esc16_narrow8(const uz_select *s); /* ... and the mate / name-id differences in eight bits too (mate_d8, qname_d8) */
/* The pair form (uz_reads_packed_view.pair_d8: tlen, mate and name id in one byte).  It numbers the names of the selection by first
 * appearance, which keeps the order of the source's ids only when those ascend by first appearance too (any decoder's table):
 * uz_select_pair8_ok says whether the plan found it so; n_esc16_pair8 = entries of esc16_* for that form (-1: not available);
 * n_new_names = the output's n_qnames; qname_map[output id] = source id. */
int uz_select_pair8_ok(const uz_select *s);
int64_t uz_select_n_esc16_pair8(const uz_select *s);
int64_t uz_select_n_new_names(const uz_select *s);
int uz_select_qname_map(const uz_select *s, uint32_t *out /* [n_new_names] */);
int64_t uz_select_n_seq_units(const uz_select *s);
/* the list form of the bases (uz_types.h bl_*; planned with unit_masks & 4 on a two-bit source): listed bases (-1: planned without), the row units
 * of the records that travel so (not counted in uz_select_n_seq_units), and whether their positions need two bytes */
int64_t uz_select_n_bl(const uz_select *s);
int64_t uz_select_n_bl_units(const uz_select *s);
int uz_select_bl_wide(const uz_select *s);
int64_t uz_select_n_exc(const uz_select *s); /* entries of the exc_* columns (0 for a source with four-bit rows) */
int64_t uz_select_n_qlow_pos(const uz_select *s); /* entries of qlow_pos when the output takes the quality plane as lists */
int uz_select_qlow_pos_wide(const uz_select *s);  /* 1 when a kept read is longer than 256 bases */
int uz_select_end_derivable(const uz_select *s);  /* 1: the output may leave `end` out (NULL) */
int64_t uz_select_n_cigar_omitted(const uz_select *s); /* words that stay home when the output sets cigar_compact (its `cigar` then holds
                                                        * uz_select_n_cigar_total - this many words) */
int64_t uz_select_n_records(const uz_select *s);
int64_t uz_select_n_cigar_total(const uz_select *s);
int64_t uz_select_n_row_units(const uz_select *s);
/* orig_index (optional, [n_records]): index of every kept record in the source table */
int uz_reads_select_fill(const uz_select *s, int threads, uz_reads_packed_view *out, int32_t *orig_index);
void uz_select_free(uz_select *s);

/* ------------------------------------------------------------------ BAM file -> staged records in one pass
 * What uz_bam_decode_regions + uz_reads_pack + uz_reads_select_* produce for one batch -- the records the batch's fetches return
 * (read_collector.py:385, :167, :478-497) and, closed under it, the records mate() returns for them (:400, :185), in the form the
 * host link carries (uz_reads_packed_view: two-bit bases of the 32-base units a fetch point falls into, qualities as lists,
 * dictionary + difference columns, no `end`, simple CIGARs left home) -- built straight from the inflated BGZF blocks of the
 * batch's reach, with no table in between.  Byte for byte the same columns as the three-step path (tests/test_io_stage.py). */
typedef struct uz_bamsrc uz_bamsrc; /* an opened BAM + its BAI (mapped; header and the head of the file read once) */
typedef struct uz_stage uz_stage;   /* one batch planned: sizes known, columns not written yet */
int uz_bamsrc_open(const char *path, const char *bai_path, int64_t head_records, uz_bamsrc **out);
void uz_bamsrc_close(uz_bamsrc *s);
int32_t uz_bamsrc_n_contigs(const uz_bamsrc *s);
const char *uz_bamsrc_contig_name(const uz_bamsrc *s, int32_t i);
int32_t uz_bamsrc_contig_length(const uz_bamsrc *s, int32_t i);
int64_t uz_bamsrc_tlen_head(const uz_bamsrc *s, int32_t *out, int64_t cap); /* as uz_bam_tlen_head */
const char *uz_inflate_backend(void); /* "libdeflate" (found at run time) or "zlib" */
int uz_io_default_threads(void);      /* what `threads <= 0` means: CPUs of the affinity mask, held to twice the cgroup CPU quota */
int uz_io_cpu_quota(void);            /* CPUs the container's cgroup grants (cpu.max), 0 = unlimited */
#define UZ_STAGE_ALL_BASES 1  /* --no-extended batches: every kept record keeps its bases (uz_reads_select_plan: all_bases) */
#define UZ_STAGE_UNIT_MASKS 2 /* only the 32-base units that hold a fetched position (+ extra[f] bases on) are staged */
#define UZ_STAGE_PLANE 4      /* qualities as the one-bit plane instead of lists (no unit masks then) */
#define UZ_STAGE_WIDE_NO_UNITS 8 /* with unit masks: a fetch wider than two bases stages no unit (SV batches; uz_reads_select_plan: unit_masks = 3) */
#define UZ_STAGE_SMALL_TASKS 32 /* the plan is made for the device's walk (uz_stage_walk_plan: one wavefront per task): the walk plan cuts the stage's
                                 * tasks into sub-tasks of ~32 kb of reach (uz_stage_merge_subtasks joins what comes back) */
#define UZ_STAGE_BASE_LISTS 16  /* with unit masks: a record whose fetches name single positions sends those bases as a list instead of the
                                 * units they lie in (uz_types.h: bl_*; uz_reads_select_plan: unit_masks & 4) */
/* fetches (tid, lo, hi[, extra]) as staging.fetch_points lists them; min_base_qual = --min-gt-qual.  UZ_IO_E_RANGE when the batch
 * holds more than 65536 combinations of the small columns (stage it through the table form then). */
int uz_bam_stage_plan(const uz_bamsrc *src, int64_t n_fetch, const int32_t *tid, const int32_t *lo, const int32_t *hi, const uint16_t *extra,
                      int flags, int min_base_qual, int threads, uz_stage **out);
/* The plan in two halves, for a caller that can inflate BGZF blocks faster than the host's cores (the device: uz_bgzf_inflate in
 * unfazed_hip.h): uz_bam_stage_begin = fetches -> reach intervals -> tasks with their file spans; uz_stage_gather_blocks copies the
 * blocks the walk will read back to back (whole blocks) and says where each DEFLATE stream starts and where its bytes belong in an
 * inflated buffer (call it with comp == NULL for the sizes first); uz_stage_set_inflated hands that buffer back (it must stay valid until
 * uz_bam_stage_finish returns); uz_bam_stage_finish = walk (listed blocks are copied from the buffer and held against their CRC-32, any
 * other block goes through the host's inflate), mates, numbering.  uz_bam_stage_plan = begin + finish. */
int uz_bam_stage_begin(const uz_bamsrc *src, int64_t n_fetch, const int32_t *tid, const int32_t *lo, const int32_t *hi, const uint16_t *extra, int flags,
                       int min_base_qual, int threads, uz_stage **out);
int uz_stage_gather_blocks(uz_stage *s, uint8_t *comp, int64_t cap, int64_t *in_off /* [n_blocks] */, int64_t *out_off /* [n_blocks + 1] */,
                           int64_t *n_blocks, int64_t *comp_bytes, int64_t *out_bytes);
int uz_stage_set_inflated(uz_stage *s, const uint8_t *inflated);
int uz_bam_stage_finish(uz_stage *s);
/* [0] records, [1] CIGAR words that travel, [2] words left home (simple records), [3] row units, [4] staged base units, [5] listed
 * bases (exc_*), [6] listed low-quality positions, [7] qlow_pos_wide, [8] dictionary entries, [9] escapes, [10] query names,
 * [11] 1 when the dictionary carries unit masks, [12] 1 when it carries list counts (tup_n_bl), [13] listed bases of the list form (bl_pos),
 * [14] row units of the records that travel so (n_bl_units) */
void uz_stage_sizes(const uz_stage *s, int64_t out[16]);
/* [0] compressed bytes read, [1] BGZF blocks read, [2] records walked, [3] records kept, [4] reach intervals, [5] mates looked
 * up through the index, [6] blocks taken from the pre-inflated buffer, [7] compressed bytes of the gathered blocks */
void uz_stage_io_stats(const uz_stage *s, int64_t out[8]);
/* seconds: [0] file spans from the index, [1] inflate + walk, [2] mates, [3] numbering, [4] the last fill */
void uz_stage_timing(const uz_stage *s, double out[6]);
/* writes the columns into the caller's buffers (sized from uz_stage_sizes; every pointer of the form described above set) */
int uz_stage_fill(const uz_stage *s, int threads, uz_reads_packed_view *out);
const char *uz_stage_qname(const uz_stage *s, uint32_t id, int32_t *len);
/* many names at once: the bytes of names ids[0 .. n) back to back into buf (no terminators), off[k] .. off[k + 1] the k-th ([n + 1]); returns the
 * bytes needed (cap = 0: the size only), -1 for an id out of range */
int64_t uz_stage_qnames(const uz_stage *s, const uint32_t *ids, int64_t n, char *buf, int64_t cap, int64_t *off);
/* ---- The walk on the device (uz_bamwalk.h; the HIP half is uz_bam_walk / uz_reads_from_bam in unfazed_hip.h).  Between uz_stage_gather_blocks
 * and the finish: uz_stage_walk_plan writes the plan as flat arrays (sizes first: [0] tasks, [1] spans, [2] reach intervals, [3] fetches,
 * [4] gathered blocks; layouts in uz_bamwalk.h).  uz_bam_stage_finish_desc finishes the plan from the device's descriptors instead of
 * walking: d[d_first[t] .. d_first[t + 1]) are task t's records inside its reach intervals in file order, d_flags[t] != 0 marks a task the
 * device could not finish (its gathered bytes ended early, or a malformed record: the host walks that task itself and reports what it finds),
 * d_walked[t] the records the device passed.  What the reference's fetch() + mate() return is then listed by uz_stage_kept, record by record
 * in file order: where the record lies (in the inflated buffer on the device, or -- a task the host walked, a mate found through the index --
 * in the aux bytes this call hands over), its name id, its mate, and its offsets in the device's stores.  uz_stage_kept_sizes: [0] records,
 * [1] CIGAR words, [2] row units, [3] base-row units, [4] query names, [5] aux bytes, [6] tasks the host walked itself.
 * Names are compared by two hashes and the length on this route (no name bytes reach the host); uz_stage_fill / uz_stage_qname(s) refuse
 * such a plan. */
void uz_stage_walk_plan_sizes(const uz_stage *s, int64_t out[8]);
int uz_stage_walk_plan(uz_stage *s, int32_t *task /* [UZ_WALK_TASK_COLS n_tasks] */, int64_t *span /* [UZ_WALK_SPAN_COLS n_spans] */,
                       int32_t *reach /* [2 n_reach] */, int32_t *fetch /* [3 n_fetch] */, int64_t *blk_coff /* [n_blocks] */,
                       uint32_t *blk_crc /* [n_blocks] the CRC-32 in every gathered block's footer, or NULL */);
/* Under UZ_STAGE_SMALL_TASKS the walk plan's tasks are SUB-TASKS of the stage's own (groups of a task's reach intervals of at most ~32 kb, each starting its
 * walk where the file's linear index puts the first record of its first 16 kb window: more, shorter chains of records for the device, the same blocks):
 * uz_stage_walk_plan_sizes [0] counts them ([5]: the stage's tasks), and uz_stage_merge_subtasks joins their descriptors per task of the stage, in place --
 * a record that two neighbouring sub-tasks met is kept once -- for uz_bam_stage_finish_desc. */
int uz_stage_merge_subtasks(const uz_stage *s, uz_walk_desc *d, const int64_t *d_first /* [n_sub + 1] */, const int32_t *d_flags, const int64_t *d_walked,
                            int64_t *h_first /* [n_tasks + 1] */, int32_t *h_flags /* [n_tasks] */, int64_t *h_walked /* [n_tasks] */);
int uz_bam_stage_finish_sub(uz_stage *s, const uz_walk_desc *d, const int64_t *d_first /* [n_sub + 1] */, const int32_t *d_flags /* [n_sub] or NULL */,
                            const int64_t *d_walked /* [n_sub] or NULL */); /* = merge + uz_bam_stage_finish_desc, without the copy */
int uz_bam_stage_finish_desc(uz_stage *s, const uz_walk_desc *d, const int64_t *d_first /* [n_tasks + 1] */, const int32_t *d_flags /* [n_tasks] or NULL */,
                             const int64_t *d_walked /* [n_tasks] or NULL */);
/* the host's twin of the device's walk: the same descriptors from the host's own walk (out == NULL: the counts only) */
int uz_stage_walk_host(uz_stage *s, uz_walk_desc *out, int64_t cap, int64_t *d_first /* [n_tasks + 1] */, int64_t *d_walked /* [n_tasks] or NULL */);
/* The joins on the device (unfazed_hip.h: uz_bam_join -- mate(), the closure over mates of mates, names numbered by first appearance: what
 * read_collector.py:400, :185 and :226-234 do record by record): the host's share is the records it has to walk itself, handed over as descriptors
 * whose bytes lie in the batch's aux store (uz_walk_desc.task = UZ_WALK_TASK_JOIN | join task, src = UZ_WALK_SRC_AUX | offset).
 *   uz_stage_walk_flagged  d_flags [walk tasks]: the device's flags (uz_bam_walk_flags); h_flags [tasks of the stage] out: whose device descriptors are
 *                          void; those tasks are walked here.  totals: [0] descriptors, [1] aux bytes held for the device
 *   uz_stage_lookup        need [n]: members whose mate no walked task can answer; every distinct position is looked up through the index;
 *                          jtask [n] out: the join task (n_tasks + k) that answers need[k]
 *   uz_stage_extra         the descriptors from d0 / aux bytes from a0 on; look_tid: the reference of every look-up task so far (or NULL) */
int uz_stage_walk_flagged(uz_stage *s, const int32_t *d_flags, int32_t *h_flags, int64_t totals[2]);
int uz_stage_lookup(uz_stage *s, int64_t n, const uz_need_rec *need, int32_t *jtask, int64_t totals[2]);
int uz_stage_extra(const uz_stage *s, int64_t d0, int64_t a0, uz_walk_desc *desc, uint8_t *aux, int32_t *look_tid);
int64_t uz_stage_n_lookup_tasks(const uz_stage *s);
void uz_stage_kept_sizes(const uz_stage *s, int64_t out[8]); /* ... [7] name bytes of the kept records */
/* the kept record that brought name id ids[k] first (ids == NULL: ids 0 .. n - 1): its name is the id's (uz_reads_from_bam returns the kept records' names) */
int uz_stage_name_records(const uz_stage *s, const uint32_t *ids, int64_t n, int64_t *rec);
int uz_stage_kept(const uz_stage *s, int threads, uz_kept_rec *out /* [records] */, int64_t *contig_off /* [n_contigs + 1] */, int32_t *max_span /* [n_contigs] */,
                  uint8_t *aux, int64_t aux_cap);
/* parity aid: the kept records of any finished plan in output order (any pointer may be NULL) */
int uz_stage_kept_debug(const uz_stage *s, uint64_t *voff, uint32_t *qname, int32_t *mate, uint8_t *bases);
void uz_stage_free(uz_stage *s);

/* The span sums of a packed view (uz_types.h: pk_sums) -- what the device's header build would otherwise compute in a pass of its own (k_off_block_sums +
 * scan).  *n_spans = the spans of the view; sums: NULL (the count only), or [(n_spans + 1) * UZ_PK_SUMS], row b = the sums over the records in front
 * of span b, the last row the totals.  Every packer of this library calls it on the view it has just filled (io_native: select / pack_reads). */
int uz_packed_block_sums(const uz_reads_packed_view *v, int threads, uint64_t *sums, int64_t *n_spans);

#ifdef __cplusplus
}
#endif
#endif /* UNFAZED_IO_H */
