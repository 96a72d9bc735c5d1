/* vacmapx.h — C-ABI of libvacmapx.so: MI355X-native seed -> non-linear chain -> extend for long reads.
 *
 * Drop-in boundary: these entry points are what the reference's Python would bind through ctypes
 * in place of its native dependencies `vacmap_index` (pip vacmap-index==0.0.3) and `edlib`, plus ONE batched
 * entry (`vm_align_batch`) that replaces the whole per-read function `get_readmap_DP_test`
 * (/root/reference/src/vacmap/mammap_clrnano.py:24023-24084) called by the worker loop (:24117).
 * Every compute entry runs hand-written HIP kernels on gfx950; there is no CPU fallback: without a GPU / the HIP
 * runtime, `vm_ctx_create` fails with VM_ERR_NO_DEVICE and every compute call returns VM_ERR_NO_CTX.
 *
 * Conventions: plain pointers and sizes; inputs are borrowed for the duration of the call; outputs are
 * library-allocated host memory released with vm_free(); every function returns 0 or a negative vm_status;
 * vm_last_error() gives a thread-local message. Anchor rows are int64 (q, r, s, l) = (read pos, global ref pos,
 * strand +-1, length) exactly like the reference's (n,4) int64 arrays (:23985).
 */
#ifndef VACMAPX_H
#define VACMAPX_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef enum vm_status {
    VM_OK = 0,
    VM_ERR_ARG = -1,
    VM_ERR_NO_DEVICE = -2,      /* no HIP device / runtime: the product never falls back to the CPU */
    VM_ERR_NO_CTX = -3,
    VM_ERR_OOM = -4,
    VM_ERR_IO = -5,
    VM_ERR_HIP = -6,
    VM_ERR_UNSUPPORTED = -7,
    /* per-read statuses (status_per_read of vm_align_batch); the reference skips such reads (:24116-24125) */
    VM_READ_RAISED = -10,       /* the reference's Python would have raised inside the per-read path */
    VM_READ_CAPACITY = -20,     /* a device work buffer overflowed for this read (reported, never silently truncated) */
    VM_READ_FASTPATH = -21,     /* internal hand-off to the `_fast` chain kernels (:23570, :24914, :27380); never returned by vm_align_batch */
    VM_READ_UNSUPPORTED = -22   /* -mode asm only: a long contig whose carried slice leaves the stored part of the score index (k_chain_linked.hip):
                                   reported, not approximated. (The MAPQ-0 edlib tie-break of mammap_asm.py:21302-21326 is built: it no longer refuses.) */
} vm_status;

enum { VM_MODE_H = 0, VM_MODE_L = 1, VM_MODE_S = 2, VM_MODE_R = 3,   /* -mode (src/vacmap/vacmap:87) */
       /* -mode asm runs src/vacmap/mammap_asm.py, an older fork of the path. Every "read" of vm_align_batch is then an assembly contig and takes
        * the route of assembly_get_readmap_DP_test (:23204): below 500 000 bases that module's per-read function (:19681, check_num = -1), in one
        * batch with the others; from 500 000 bases on the batch-linked path, contig by contig (vm_align_asm). */
       VM_MODE_ASM = 4 };

/* option dict `pdict` of the reference driver (src/vacmap/vacmap:177-296) as one POD */
typedef struct vm_params {
    int32_t mode;            /* VM_MODE_* */
    int32_t check_num;       /* -c (100) */
    int32_t mid_occ;         /* -1 = index default (mammap_clrnano.py:23985) */
    int32_t global_maxdiff;  /* -globalmaxdiff (50) */
    int32_t local_maxdiff;   /* -localmaxdiff (30) */
    int32_t local_kmersize;  /* 9 (src/vacmap/vacmap:255) */
    int32_t eqx;             /* --eqx */
    int32_t hardclip;        /* --H */
    int32_t nodiscard;       /* --nodiscard */
    int32_t reserved;
    double global_skipcost;  /* -globalpenalty */
    double local_skipcost;   /* -localpenalty */
    double maxdivergence;    /* -maxdivergence */
} vm_params;
void vm_params_default(vm_params* p, int mode);            /* mode defaults of src/vacmap/vacmap:257-296 */

/* ------------------------------------------------------------------ index (replaces mp.Aligner, vacmap:344-367) */
typedef struct vm_index vm_index;
typedef struct vm_ctx vm_ctx;

/* one context = one GPU (device_id) + its streams and work buffers; one per host thread per GPU */
int vm_ctx_create(int device_id, vm_ctx** out);
void vm_ctx_destroy(vm_ctx*);
/* How many contexts share this GPU (batches in flight, one host thread each; default 1). A context that runs alone sizes its
 * latency-bound kernels to fill the device; with several in flight they are launched narrower (the local seeding kernel at 3 / 2
 * workgroups per CU for 1 / >= 2 contexts) so that another batch's VALU-bound gap-fill kernel can be resident beside them. */
int vm_ctx_set_inflight(vm_ctx*, int n_contexts);
/* on != 0: the context's host thread sleeps while it waits for the GPU inside vm_align_batch instead of spinning (default: spin, lowest
 * latency). For processes whose other threads need the cores — vacmap_amd.driver's SAM emitters under a CPU quota. */
int vm_ctx_set_blocking_sync(vm_ctx*, int on);
int vm_device_count(void);
/* diagnostics of the chain DPs with four reads per wavefront (csrc/k_chain_rows.hip; replays mammap_clrnano.py:24912-24928, :24944-25003): eight counters summed
 * over every launch of the process since they were switched on — [0] anchors, [1] scans that left the register window and went on through S_arg / S in HBM,
 * [2] insertions placed through HBM, [3] opcount of the global chain; [4..7] the same of the local chains. enable: 1 switch on (first call allocates), 0 read,
 * -1 read and reset. out8 may be NULL. Off by default (the kernels get a NULL counter block); the GPU tests assert that the rare paths did run. */
int vm_debug_chain_counters(int enable, unsigned long long* out8);
/* free / total bytes of the context's device (hipMemGetInfo): the driver drops a context when the grow-only work pools of the batches in flight
 * leave less than a safety margin of HBM free after their sizing run */
int vm_ctx_mem_info(vm_ctx*, int64_t* free_bytes, int64_t* total_bytes);

/* build the minimizer index of a FASTA file / in-memory contigs ON THE GPU and keep it resident in HBM
 * (replaces `mp.Aligner(path, w=, k=)`, src/vacmap/vacmap:344; index.py:26) */
int vm_index_build_fasta(vm_ctx*, const char* fasta_path, int k, int w, vm_index** out);
int vm_index_build_mem(vm_ctx*, int nseq, const char* const* names, const char* const* seqs, const int64_t* lens,
                       int k, int w, vm_index** out);
/* own on-disk format `<ref>.w<w>_k<k>.vmx` (the reference's naming rule `<ref>.w<w>_k<k>.mmi`, vacmap:326): contig table, bases and the
 * sorted position column; the loader recomputes the hashes on the GPU and rejects a file whose header, sizes, positions or order are
 * inconsistent (VM_ERR_IO) before anything is used */
int vm_index_save(const vm_index*, const char* path);
int vm_index_load(vm_ctx*, const char* path, vm_index** out);
/* minimap2 index files (`minimap2 -d`, on-disk format v3), the files the reference builds and reuses as `<ref>.w<w>_k<k>.mmi`
 * (src/vacmap/vacmap:324-344, index.py:26). load: the file's minimizer set and 4-bit sequence become the HBM-resident index; every
 * stored hash is re-derived from the sequence on the GPU and a mismatch rejects the file (VM_ERR_IO). Homopolymer-compressed,
 * sequence-less and multi-part files are VM_ERR_UNSUPPORTED. save: writes the index in that format (bucket_bits < 0: minimap2's 14). */
int vm_index_load_mmi(vm_ctx*, const char* path, vm_index** out);
int vm_index_save_mmi(const vm_index*, const char* path, int bucket_bits);
void vm_index_free(vm_index*);
int vm_index_k(const vm_index*);                            /* Aligner.k      (:24024) */
int vm_index_w(const vm_index*);
int vm_index_nseq(const vm_index*);
int vm_index_mid_occ(const vm_index*);
int64_t vm_index_n_minimizers(const vm_index*);
int64_t vm_index_n_distinct(const vm_index*);               /* distinct minimizer hashes (= occupied table slots) */
/* Aligner.seq_offset (vacmap:358-361): name, length, global offset of contig i */
int vm_index_seq_info(const vm_index*, int i, const char** name, int64_t* len, int64_t* offset);
/* Aligner.seq(name)  (vacmap:363): copies upper-case bases [start,end) of contig i; returns the count */
int64_t vm_index_seq(const vm_index*, int i, int64_t start, int64_t end, char* out);
/* device-resident sorted minimizer arrays copied back to the host (tests): hashes[n], positions[n] (gpos<<1|strand) */
int vm_index_minimizers(const vm_index*, uint64_t** hashes, uint64_t** positions, int64_t* n);
/* raw device pointers + sizes of the index blob pieces, for the multi-GPU broadcast (RCCL over xGMI) */
int vm_index_blob_count(const vm_index*);
int vm_index_blob(const vm_index*, int i, void** dev_ptr, int64_t* bytes);
/* metadata (k, w, contig names/lengths, table geometry) needed to allocate an empty replica on another GPU / process:
 * sender: meta_size + meta_get; receiver: vm_index_from_meta, then the blobs are filled by the broadcast */
int vm_index_meta_size(const vm_index*, int64_t* bytes);
int vm_index_meta_get(const vm_index*, void* buf, int64_t bytes);
int vm_index_from_meta(vm_ctx*, const void* buf, int64_t bytes, vm_index** out);

/* ------------------------------------------------------------------ stage entry points (batched; GPU) */
/* minimizer sketch of n reads: outputs concatenated per read, off[n+1] (tests; part of `.map`) */
int vm_sketch_batch(vm_ctx*, int k, int w, int64_t n, const char* seqs, const int64_t* offsets,
                    uint64_t** hash, int32_t** pos, int8_t** strand, int64_t** off);
/* `.map(seq, check_num=, mid_occ=)` (:23985) for n reads: anchors rows concatenated, anchor_off[n+1] */
int vm_map_batch(vm_ctx*, const vm_index*, int check_num, int mid_occ, int64_t n, const char* seqs,
                 const int64_t* offsets, int64_t** anchors, int64_t** anchor_off);
/* single-read form with the exact shape of the reference call */
int vm_map(vm_ctx*, const vm_index*, const char* seq, int64_t len, int check_num, int mid_occ, int64_t** anchors, int64_t* n);

/* global non-linear chain of n reads: strand flip (:21202) + hit2work_1 (:23491) + decode_hit (:23981).
 * in: anchors (as returned by map), readlens. out per read: need_reverse, mapq, signed score (0 = unmapped),
 * paths (primary + secondaries, descending read order): path_off[n_paths_total+1], read_path_off[n+1] */
typedef struct vm_chains_out {
    int32_t* need_reverse;   /* n */
    int32_t* mapq;           /* n */
    double* score;           /* n */
    int32_t* fast_used;      /* n: 1 if GC-fast (:25033) produced the chain */
    int64_t* read_path_off;  /* n+1: first path of each read */
    int64_t* path_off;       /* total_paths+1: first anchor row of each path */
    int64_t* path_anchors;   /* rows */
    /* raw DP state of the LAST run of the exact DP (tests; null unless want_raw) : concatenated per read */
    double* S; int64_t* P; int64_t* S_arg; int64_t* gmax; int64_t* opcount;
} vm_chains_out;
int vm_chain_global_batch(vm_ctx*, const vm_params*, int kmersize, int64_t n, const int64_t* anchors,
                          const int64_t* anchor_off, const int64_t* readlens, int want_raw, vm_chains_out* out);
void vm_chains_out_free(vm_chains_out*);

/* local re-seeding + local chain (:28479) of n reads. reads must be in chain orientation (reverse-complemented by the
 * caller when need_reverse). out: chain rows per read (descending read order), raw local anchors (pre-DP order) */
typedef struct vm_local_out {
    int32_t* status;         /* n: 0 or VM_READ_* */
    int32_t* variant;        /* n: 0 = LC-exact (:27305), 1 = LC-mm (:28250) */
    double* score;           /* n */
    int64_t* chain_off;      /* n+1 */
    int64_t* chain;          /* rows */
    int64_t* raw_off;        /* n+1 */
    int64_t* raw;            /* rows, sorted by (q+l) stable = the DP's input order */
} vm_local_out;
int vm_local_chain_batch(vm_ctx*, const vm_index*, const vm_params*, int64_t n, const char* seqs, const int64_t* offsets,
                         const int64_t* read_path_off, const int64_t* path_off, const int64_t* path_anchors, vm_local_out* out);
void vm_local_out_free(vm_local_out*);

/* One batch of -mode asm's LINKED chain DPs on the device (contigs of 500 kb and more, src/vacmap/mammap_asm.py:23228-23275 / :23328-23373):
 * which = 0 linked GC-exact (:21686), 2 linked LC (:21504). rows: n x 4 int64, the n_pre anchors carried from the previous batch first, the
 * new ones sorted by read position behind them; pre_S / pre_P, g_max_scores, g_max_index, prereadloc: the carried state (n_pre may be 0).
 * out: S[n], P[n]; the HOT part of the score-sorted index S_arg (its last n_hot entries in the reference's order — the n_cold entries below can
 * never be visited, k_chain_linked.hip); gmax (-1: GC-exact's bail-out); and what :23250-23272 carries into the next batch: carry_status 0
 * with saved = 0 (`continue`: best chain ends in a carried or fresh anchor) or saved = 1 and carry_S / carry_P / carry_rows [n_carry], or
 * a negative status when the device refuses (slice reaches the cold entries, bail-out) or the reference raises. */
typedef struct vm_linked_out {
    int64_t gmax, n_hot, n_cold, opcount; double cold_max;
    double* S; int64_t* P; int64_t* S_arg_hot;
    int32_t carry_status, saved; int64_t n_carry; double* carry_S; int64_t* carry_P; int64_t* carry_rows; double carry_g_max_scores; int64_t carry_prereadloc;
} vm_linked_out;
int vm_chain_linked(vm_ctx*, int which, int kmersize, double skipcost, int maxdiff, int maxgap, int64_t n, const int64_t* rows, int64_t n_pre,
                    const double* pre_S, const int64_t* pre_P, double g_max_scores, int64_t g_max_index, int64_t prereadloc, vm_linked_out* out);
void vm_linked_out_free(vm_linked_out*);

/* `mp.k_cigar(target, query, match, mismatch, o1, e1, o2, e2, bw=-1, zdropvalue=-1, eqx)` (:21554): global dual-affine
 * alignment with traceback, n problems. t/q concatenated with offsets [n+1]. out: CIGAR strings concatenated
 * (NUL-separated) with cigar_off[n+1], scores[n] */
typedef struct vm_score { int32_t match, mismatch, o1, e1, o2, e2; } vm_score;
int vm_k_cigar_batch(vm_ctx*, const vm_score*, int eqx, int64_t n, const char* t, const int64_t* t_off, const char* q,
                     const int64_t* q_off, char** cigars, int64_t** cigar_off, int32_t** scores);
/* The same problems through the schedule vm_align_batch uses for its gap fill (mammap_clrnano.py:21554, :21598 call sites): longest-first
 * device queue, banded eight-per-wavefront fill first (anti-diagonal form on a fixed band of 32 * ns diagonals), the problems whose band
 * is not PROVEN optimal filled again in full by a second launch, per-problem layout flag for the traceback. CIGARs must equal
 * vm_k_cigar_batch's. band_flag[n]: 16 + ns = the band's result was proven and kept, 0 = full matrix. stats[4] = {small problems tried
 * in a band, proven, sent to the second launch (not proven, or small but never tried), problems outside the small class}. No scores
 * (that form never captures them). */
int vm_k_cigar_batch_banded(vm_ctx*, const vm_score*, int eqx, int64_t n, const char* t, const int64_t* t_off, const char* q,
                            const int64_t* q_off, char** cigars, int64_t** cigar_off, int32_t** band_flag, int64_t* stats);
/* `mp.k_cigar(..., 4,4,4,4, bw=100, zdropvalue=50)` (:2381): banded x-drop extension from (0,0); out t_e[n], q_e[n], score[n] */
int vm_k_extend_batch(vm_ctx*, int match, int mismatch, int o, int e, int bw, int zdrop, int64_t n, const char* t,
                      const int64_t* t_off, const char* q, const int64_t* q_off, int32_t** t_e, int32_t** q_e, int32_t** score);
/* single-problem form with the reference's argument list; out mirrors the returned tuple (cigar, q_e, t_e) */
typedef struct vm_cigar_out { char* cigar; int32_t q_e, t_e, score; } vm_cigar_out;
int vm_k_cigar(vm_ctx*, const char* t, int64_t tl, const char* q, int64_t ql, const vm_score* sc, int bw, int zdrop,
               int eqx, vm_cigar_out* out);
/* `edlib.align(query, target, task='distance')['editDistance']` (:19251), n problems */
int vm_edit_distance_batch(vm_ctx*, int64_t n, const char* q, const int64_t* q_off, const char* t, const int64_t* t_off,
                           int64_t** dist);
int64_t vm_edit_distance(vm_ctx*, const char* q, int64_t ql, const char* t, int64_t tl);
/* banded forms used by the divergence filter of vm_align_batch: bound[i] >= editDistance, equal to it when the optimal path stays
 * inside the band; -1 when |len(q) - len(t)| exceeds the tier's window (not eligible). tier 1: four problems per wavefront, band of
 * +-320 rows, window 256; tier 2: one problem per wavefront, +-768 rows, window 512. The filter keeps a segment when
 * bound / min(len) <= maxdivergence and passes it to the next tier (finally the exact kernel) otherwise, so its decisions equal the
 * reference's. */
int vm_edit_distance_bound_batch(vm_ctx*, int tier, int64_t n, const char* q, const int64_t* q_off, const char* t, const int64_t* t_off,
                                 int64_t** bound);

/* ------------------------------------------------------------------ the batched path (the GPU entry) */
/* record = one 9-tuple of get_onemapinfolist (:20760): (readid, contig, strand, q_st, q_en, r_st, r_en, mapq, cigar) */
typedef struct vm_record {
    int32_t read_idx;
    int32_t contig;          /* index into vm_index_seq_info */
    int32_t strand;          /* +1 '+', -1 '-' (label as emitted by the reference, :20760/:20776/:20810/:20826) */
    int32_t mapq;
    int64_t q_st, q_en;      /* aligned-strand coordinates (:20762-20763) */
    int64_t r_st, r_en;      /* contig-local */
    int64_t cigar_off, cigar_len;   /* into cigar_blob (NUL-terminated strings) */
} vm_record;

typedef struct vm_batch_stats {     /* measured on the device, for bench.py's roofline arithmetic */
    int64_t n_reads, read_bases, n_minimizers, n_hits, n_anchors, n_local_hits, n_local_anchors;
    int64_t n_segments, n_ed_problems, ed_cells, n_ext_problems, ext_cells, n_dp_problems, dp_cells;
    int64_t n_records, cigar_bytes, aligned_bases, n_unmapped, n_failed;
    int64_t dp_string_bytes;        /* target+query bases read by the gap-fill DP */
    double ms_total;                /* device time of the whole batch (HIP events on the ctx stream) */
    double ms_stage[16];            /* 0 seed, 1 global chain, 2 local, 3 divergence filter, 4 edge extension, 5 gap fill + records,
                                       6 nofilter redo, 7 result download */
    double ms_gapfill_fill;         /* k_gapfill_fill launches only (HIP events on the stream the kernel runs on) */
    double ms_gapfill_trace;        /* k_gapfill_trace launches only */
    int64_t n_gapfill_launches;
    int64_t n_ed_full;              /* divergence-filter problems the banded kernels could not settle (re-run unbanded) */
    int64_t n_ed_tier2;             /* problems the four-per-wave band could not settle (re-run in the wide band) */
    int64_t n_ed_tier1;             /* problems the anchor bound could not settle (re-run in the four-per-wave band) */
    int64_t n_dp_redo;              /* gap-fill problems whose band was not proven optimal (or that were too large to try one): filled again in full */
    int64_t dp_redo_tb_bytes;       /* traceback bytes of those (second pool); dp_cells counts all traceback bytes written */
    double ms_local_seed;           /* k_local_seed, the batch's main launch (HIP events on the stream it runs on) */
    double ms_cluster;              /* k_cluster_big + k_cluster (hit clustering of the seed stage) */
    int64_t n_host_syncs;           /* host waits for the device inside this batch (sizing read-backs + result download) */
    int64_t n_local_general;        /* reads the guide-banded local seeding kernel handed to the general one (k_local_seed) */
    int64_t n_ext_retries;          /* times the batch was run again because a pool of the extend stage was too small (pools x4 per retry) */
    int64_t n_batch_retries;        /* times the whole batch was run again because an assumption made instead of a host wait did not hold (a pool sized from the context's history) */
} vm_batch_stats;

/* Align n reads (replaces get_readmap_DP_test per read). seqs concatenated, offsets[n+1].
 * recs are ordered by (read_idx, emission order of the reference). status_per_read[n]: 0 (n_recs may be 0 = unmapped,
 * :24132) or a negative VM_READ_* (read skipped like the reference's swallowed exception). stats may be NULL. */
int vm_align_batch(vm_ctx*, const vm_index*, const vm_params*, int64_t n_reads, const char* seqs, const int64_t* offsets,
                   vm_record** recs, int64_t* n_recs, char** cigar_blob, int32_t* status_per_read, vm_batch_stats* stats);
/* the same with reads already resident in HBM (bench: inputs resident when the timed region starts).
 * vm_reads_upload copies host reads to the device once; vm_align_resident runs the path on them. */
typedef struct vm_reads vm_reads;
int vm_reads_upload(vm_ctx*, int64_t n_reads, const char* seqs, const int64_t* offsets, vm_reads** out);
/* upload another batch into an existing reads object: its device buffers are reused (grow-only); no alignment of it may be in flight. A host thread with a context of
 * its own streams batches into HBM ahead of the aligning contexts — the upload of batch i + 1 runs under the kernels of batch i */
int vm_reads_reupload(vm_ctx*, vm_reads*, int64_t n_reads, const char* seqs, const int64_t* offsets);
void vm_reads_free(vm_reads*);
int vm_align_resident(vm_ctx*, const vm_index*, const vm_params*, const vm_reads*, vm_record** recs, int64_t* n_recs,
                      char** cigar_blob, int32_t* status_per_read, vm_batch_stats* stats);

/* -mode asm on ONE assembly contig, the route of assembly_get_readmap_DP_test (mammap_asm.py:23204): below split_len bases the fork's per-read
 * function (= vm_align_batch with VM_MODE_ASM), otherwise 100 kb seeding windows, chain DPs linked across batches of more than batch_anchors
 * anchors, the second linked round over 9-mer anchors, ass_extend_func. split_len / batch_anchors / window <= 0: the reference's 500000 / 500000 /
 * 100000 (tests shrink them to reach the linked path on small inputs; split_len may only be lowered). p->mode must be VM_MODE_ASM. *status: 0,
 * VM_READ_RAISED, VM_READ_CAPACITY or VM_READ_UNSUPPORTED (see that status). The host runs the reference's loop (batch assembly, tracebacks, cut points); seeding, chain DPs,
 * re-seeding and extension run on the device. */
int vm_align_asm(vm_ctx*, const vm_index*, const vm_params* p, const char* contig, int64_t len, int64_t split_len, int64_t batch_anchors, int64_t window,
                 vm_record** recs, int64_t* n_recs, char** cigar_blob, int32_t* status);

/* Diagnostic for the stage tests of the extend phase (E1 / E3 / E4): runs vm_align_batch's path and returns every read's segment lists
 * as they stand, in the first (filtering) run of extend_func (:19238), after
 *   stage 0  rebuild_chain_break (:23437)
 *   stage 3  divergence filter, both rounds of extend_edge_test (:2302) and the drop_misplaced_alignment_test loop (:726)
 *   stage 5  merge_conjacent_alignment (:16736) and fix_simple_inv (:24226)
 * rows: int64 (segment index, q, r, s, l) concatenated over the reads, row_off[n + 1] (both vm_free). Reads without a local chain have none. */
int vm_align_trace(vm_ctx*, const vm_index*, const vm_params*, int64_t n_reads, const char* seqs, const int64_t* offsets, int stage,
                   int64_t** rows, int64_t** row_off);

/* ------------------------------------------------------------------ native I/O around the batched path (host threads, no GPU work)
 * SAM text for the records of one batch — the consumer of the path, get_bam_dict_str / _comments (mammap_clrnano.py:20841, :21022) with
 * reassign_mapq (:11661), merged CIGARs (:4773), NM (output_functions.py:300), optional MD / cs (:19012 / :19062), SA tag, S / H clipping,
 * approximate SA CIGARs, CG tag switch, comment copying (:20686). Options mirror the driver's `pdict` keys of those functions. */
typedef struct vm_sam_opts {
    int32_t md, shortcs, cigar2cg, markunbalancetra, hardclip, fakecigar;
    const char* rg_id;           /* RG:Z: value on every line; NULL = none (the reference's driver always sets one, vacmap:211-214) */
    int32_t asm_mode;            /* 1: the emitter of -mode asm, iterator_get_bam_dict_str / _comments (mammap_asm.py:22757, :22942): NM = the CIGAR's X / D / I
                                  * counts as mergecigar_nm_ takes them (:23125), MAPQ written as 60 (1 for 0) in the column and in SA, the second record
                                  * primary when the first has MAPQ 1 and it has not (:22847-22850) */
} vm_sam_opts;
/* reads as blobs with offsets[n + 1] (quals / comments and their offsets may be NULL; a read whose quality string is empty or of another
 * length than its sequence gets '*'); recs / cigar_blob / status as returned by vm_align_batch for these reads. text: the lines
 * ('\n'-terminated) of all reads in read order, text_off[n + 1] delimits every read's lines (both vm_free). A read whose emission raises in
 * the reference (index past a sequence end) or whose status is not 0 emits nothing and counts in n_skipped, like the worker's except
 * (:24127-24134). */
int vm_sam_emit(const vm_index*, const vm_sam_opts*, int64_t n_reads, const char* names, const int64_t* name_off, const char* seqs, const int64_t* seq_off,
                const char* quals, const int64_t* qual_off, const char* comments, const int64_t* com_off, const vm_record* recs, int64_t n_recs,
                const char* cigar_blob, const int32_t* status, int nthreads, char** text, int64_t** text_off, int64_t* n_lines, int64_t* n_skipped);
/* `mp.fastx_read(path, read_comment=)` (vacmap:445): FASTA / FASTQ, plain or gzip. vm_fastx_read returns up to max_reads records (stopping
 * early once max_bases bases are held) as blobs: names, UPPER-CASED sequences (vacmap:476), qualities (empty for FASTA), comments (header
 * text after the first blank, else tab). Returns the record count, 0 at the end of the input, or a negative vm_status. */
/* entries idx[0..n) of a blob copied back to back (out may be NULL to size it): returns the byte count, out_off[n + 1] */
int64_t vm_blob_gather(const char* blob, const int64_t* off, const int64_t* idx, int64_t n, char* out, int64_t* out_off);
/* the same over several blobs: output entry j = entry idx[j] of blob part[j]; returns the bytes written (out must hold them) */
/* the same merge into a regular file through a shared mapping, copied by nthreads threads: the file is extended from file_off (its current end) by the
 * merged size. Returns the bytes written, -1 on error, -2 when fd cannot be mapped (pipe, terminal): use vm_blob_write_parts then */
int64_t vm_blob_write_parts_mmap(int fd, int64_t file_off, const char* const* blobs, const int64_t* const* offs, const int32_t* part, const int64_t* idx, int64_t n, int nthreads);
int64_t vm_blob_gather_parts(const char* const* blobs, const int64_t* const* offs, const int32_t* part, const int64_t* idx, int64_t n, char* out);
/* page-locked host memory for the read blobs a caller hands to vm_align_batch: the upload is then a DMA the host thread does not wait for
 * (from pageable memory the runtime stages it through bounce buffers on the calling thread: ~10 ms per 60 MB batch, more under memory load) */
void* vm_pinned_alloc(int64_t bytes, int device);      /* device: the GPU the calling process works on (the allocation must not open a context on GPU 0 from every rank); -1 = the thread's current one */
void vm_pinned_free(void* p);
/* vm_blob_gather_parts written to a file descriptor with writev() instead of into a buffer; returns the bytes written or -1 */
int64_t vm_blob_write_parts(int fd, const char* const* blobs, const int64_t* const* offs, const int32_t* part, const int64_t* idx, int64_t n);
typedef struct vm_fastx vm_fastx;
int vm_fastx_open(const char* path, vm_fastx** out);
/* the records whose FIRST byte lies in [begin, end) of a plain FASTA / FASTQ file (end < 0: to the end of the file): N readers over N consecutive
 * ranges see every record exactly once — the sharded driver gives every rank a range and its parser threads slices of it (the reference has one
 * reader process, vacmap:445-497). Compressed input: VM_ERR_UNSUPPORTED for begin > 0. */
int vm_fastx_open_range(const char* path, int64_t begin, int64_t end, vm_fastx** out);
void vm_fastx_close(vm_fastx*);
int64_t vm_fastx_read(vm_fastx*, int64_t max_reads, int64_t max_bases, char** names, int64_t** name_off, char** seqs, int64_t** seq_off, char** quals,
                      int64_t** qual_off, char** comments, int64_t** com_off);

/* cost tables C0 as uploaded to the device (tests): which = 0 extra,1 readgap_h,2 readgap_r,3 large_readgap (f32),
 * 4 log2cache, 5 log2int (f64). returns length, *data = host copy read back FROM THE DEVICE (vm_free) */
int64_t vm_table(vm_ctx*, int which, void** data);

void vm_free(void*);
const char* vm_last_error(void);
const char* vm_version(void);

#ifdef __cplusplus
}
#endif
#endif
