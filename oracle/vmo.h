/* vmo.h — C API of the CPU ORACLE (test infrastructure, NOT the product).
 *
 * This library is a CPU restatement of the seed -> non-linear chain -> extend path of the
 * reference (VACmap, /root/reference/src/vacmap/mammap_clrnano.py and its mode twins).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The product (vacmap_amd/, libvacmapx.so) never links or calls anything in oracle/.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - chain / local / segment-surgery / record stages: pinned against the reference's own Python,
 *     imported read-only in the build container (tools/harness) -> tests/golden/.
 *   - native primitives (map, k_cigar, edit distance) live in the un-vendored C extension
 *     vacmap-index==0.0.3 and edlib==1.3.9, absent from /root/reference: "parity unpinned" for
 *     map/k_cigar (own normative spec), edit distance is mathematically unique.
 */
#ifndef VMO_H
#define VMO_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct vmo_index vmo_index;

/* mode constants */
enum { VMO_MODE_H = 0, VMO_MODE_L = 1, VMO_MODE_S = 2, VMO_MODE_R = 3,
       VMO_MODE_ASM = 4 /* -mode asm: mammap_asm.py, an older fork of the path (vmo_asm.cc) */ };

typedef struct vmo_params {
    int32_t mode;            /* VMO_MODE_* */
    int32_t check_num;       /* -c, default 100 (vacmap:105) */
    int32_t mid_occ;         /* -1 = index default */
    int32_t global_maxdiff;  /* 50 (vacmap:112) */
    int32_t local_maxdiff;   /* 30 (vacmap:113) */
    int32_t local_kmersize;  /* 9  (vacmap:255) */
    int32_t eqx;             /* --eqx */
    int32_t hardclip;        /* --H */
    int32_t nodiscard;       /* --nodiscard (mode default: False for H/L) */
    int32_t reserved;
    double global_skipcost;  /* 40 (H,L) / 30 */
    double local_skipcost;   /* 40 (H) / 59 (L) / 30 */
    double maxdivergence;    /* 0.2 (H) / 0.1 (L) / 0.5 */
} vmo_params;

void vmo_params_default(vmo_params* p, int mode);

/* ---- index / seed (own spec VMX-S1; replaces vacmap_index.Aligner, mammap_clrnano.py:23985) ---- */
vmo_index* vmo_index_build_fasta(const char* fasta_path, int k, int w);
vmo_index* vmo_index_build_mem(int nseq, const char* const* names, const char* const* seqs,
                               const int64_t* lens, int k, int w);
void vmo_index_free(vmo_index*);
int vmo_index_k(const vmo_index*);
int vmo_index_w(const vmo_index*);
int vmo_index_nseq(const vmo_index*);
int vmo_index_mid_occ(const vmo_index*);
int64_t vmo_index_n_minimizers(const vmo_index*);
int64_t vmo_index_n_distinct(const vmo_index*);
const char* vmo_index_seq_name(const vmo_index*, int i);
int64_t vmo_index_seq_len(const vmo_index*, int i);
int64_t vmo_index_seq_offset(const vmo_index*, int i);
/* copies upper-cased bases [start,end) of contig i into out (no NUL); returns count */
int64_t vmo_index_seq(const vmo_index*, int i, int64_t start, int64_t end, char* out);
/* raw sorted minimizer arrays (for cross-checking the product's index builder) */
const uint64_t* vmo_index_hashes(const vmo_index*);   /* n_minimizers, ascending (hash, pos) */
const uint64_t* vmo_index_positions(const vmo_index*); /* gpos<<1 | strand */

/* sketch: out arrays sized >= len; returns count. pos = k-mer start, strand 0/1 */
int64_t vmo_sketch(const char* seq, int64_t len, int k, int w, uint64_t* hash, int32_t* pos, int8_t* strand);
/* map: anchors rows (q, r, s, l) int64; returns n (<0 error); *anchors owned by caller via vmo_free */
int64_t vmo_map(const vmo_index*, const char* seq, int64_t len, int check_num, int mid_occ, int64_t** anchors);
void vmo_free(void*);

/* ---- DP primitives (own spec VMX-DP; replaces vacmap_index.k_cigar, :21554 / :2381) ---- */
/* global dual-affine alignment with traceback. cigar_out: malloc'd NUL-terminated string. */
int vmo_k_cigar_global(const char* t, int64_t tl, const char* q, int64_t ql, int match, int mismatch,
                       int o1, int e1, int o2, int e2, int eqx, char** cigar_out, int32_t* score);
/* banded x-drop extension from (0,0), single affine (o,e). returns best score; *t_e,*q_e consumed */
int vmo_k_extend(const char* t, int64_t tl, const char* q, int64_t ql, int match, int mismatch,
                 int o, int e, int bw, int zdrop, int32_t* t_e, int32_t* q_e);
/* global Levenshtein distance (edlib.align(task='distance'), :19251) */
int64_t vmo_edit_distance(const char* q, int64_t ql, const char* t, int64_t tl);

/* ---- chain stages (restated from the reference) ---- */
/* S2 get_reversed_chain_numpy_rough :21202. anchors in/out (n x 4). returns need_reverse flag */
int vmo_strand_flip(int64_t* anchors, int64_t n, int64_t readlen);
/* G2/G3 raw DP: sorts nothing; expects anchors already argsorted by q (stable).
 * which: 0 = exact (_d_all :24828), 1 = fast (_d_fast_all :25033). Outputs S[n], P[n], S_arg[n].
 * returns g_max_index (or -1 when exact bails out, :24914) */
int64_t vmo_chain_global_raw(const int64_t* anchors, int64_t n, int mode, int kmersize, double skipcost,
                             int maxdiff, int maxgap, int which, double* S, int64_t* P, int64_t* S_arg);

typedef struct vmo_chains {      /* result of hit2work_1 + decode_hit (:23491, :23981) */
    int32_t need_reverse;
    int32_t mapq;
    double score;                /* score of primary path (0 = unmapped) */
    int32_t n_paths;             /* primary + secondaries returned by decode_hit */
    int32_t fast_used;
    int64_t* path_off;           /* n_paths+1 offsets into path_anchors (rows) */
    int64_t* path_anchors;       /* rows (q,r,s,l), descending read order */
    int32_t n_all;               /* all peeled chains with score > 40 (path_list) */
    double* all_scores;          /* scores_list */
} vmo_chains;
int vmo_decode_hit(const int64_t* anchors_in, int64_t n, int64_t readlen, int kmersize,
                   const vmo_params* p, vmo_chains* out);
void vmo_chains_free(vmo_chains*);

/* L1-L5 get_localmap_multi_all_forDP_inv_guide_list :28479. paths as in vmo_chains.
 * read: the read in chain orientation (already swapped when need_reverse). out: local chain rows
 * (descending read order) + raw local anchors before chaining (for stage tests). */
int vmo_local_chain(const vmo_index*, const char* read, int64_t readlen, int n_paths, const int64_t* path_off,
                    const int64_t* path_anchors, const vmo_params* p, double* score,
                    int64_t** chain, int64_t* n_chain, int64_t** raw, int64_t* n_raw, int32_t* variant);

/* ---- records ---- */
typedef struct vmo_record {
    int32_t read_idx; int32_t contig; int32_t strand; /* +1 / -1 as labelled in the tuple */
    int32_t mapq;
    int64_t q_st, q_en, r_st, r_en;
    int64_t cigar_off; int64_t cigar_len;
} vmo_record;

/* E1-E6 extend_func (:19238) on a local chain given in ascending read order */
int vmo_extend(const vmo_index*, const char* read, int64_t readlen, const int64_t* chain_asc, int64_t n_chain,
               int mapq, int need_reverse, int nofilter, const vmo_params* p,
               vmo_record** recs, int64_t* n_recs, char** cigar_blob, int32_t* filtered);

/* whole per-read path get_readmap_DP_test (:24023). status: 0 ok (n_recs may be 0 = unmapped),
 * <0 = the reference would have raised (read skipped, :24116-24125) */
int vmo_align_read(const vmo_index*, const char* read, int64_t readlen, const vmo_params* p,
                   vmo_record** recs, int64_t* n_recs, char** cigar_blob);

/* batch over reads with nthreads std::threads (cpu_baseline). seqs concatenated, offsets n+1 */
int vmo_align_batch(const vmo_index*, const vmo_params* p, int64_t n_reads, const char* seqs,
                    const int64_t* offsets, int nthreads, vmo_record** recs, int64_t* n_recs,
                    char** cigar_blob, int32_t* status);

/* -mode asm (mammap_asm.py): one assembly contig. Contigs below 500 000 bases take the module's own get_readmap_DP_test (:19681) with
 * check_num = -1; longer ones assembly_get_readmap_DP_test (:23204): 100 kb seeding windows, chain DPs LINKED across batches of more than
 * 500 000 anchors, a second linked pass over 9-mer anchors, ass_extend_func. p->mode must be VMO_MODE_ASM. split_len / batch_anchors /
 * window <= 0 take the reference's 500000 / 500000 / 100000 (tests shrink them to reach the linked path on small inputs — the reference
 * run that made the goldens was patched to the same numbers). status as vmo_align_read. */
int vmo_align_asm(const vmo_index*, const char* contig, int64_t len, const vmo_params* p, int64_t split_len, int64_t batch_anchors,
                  int64_t window, vmo_record** recs, int64_t* n_recs, char** cigar_blob);
/* stages of the long-contig path for the tests of the device's loop: which = 1 first-round path (descending read order), 2 the chain handed to
 * ass_extend_func (ascending, read overlaps trimmed), 3 the second-round anchor batches (rows concatenated, off[n_off] batch ends) */
int vmo_asm_trace(const vmo_index*, const char* contig, int64_t len, const vmo_params* p, int64_t split_len, int64_t batch_anchors, int64_t window,
                  int which, int64_t** rows, int64_t* n_rows, int64_t** off, int64_t* n_off);
/* test hook: the bail-out factor of the asm fork's GC-exact (max_factor = 1000, mammap_asm.py:20623 / :21757); lowered by tests that need the
 * linked path to take its GC-fast (:23246-23247) on inputs where the real factor is never reached */
void vmo_test_asm_max_factor(double f);
/* decode_hit of the fork (:21280; seeds the contig itself with p->check_num): MAPQ, signed score, the primary path */
int vmo_decode_hit_asm(const vmo_index*, const char* contig, int64_t len, const vmo_params* p, vmo_chains* out);
/* stage entry of the linked chain DPs (:21686 GC-exact, :21871 GC-fast, :21504 LC): which = 0 / 1 / 2. anchors sorted by q (rows of the
 * carried anchors first); pre_S / pre_P: carried state (n_pre may be 0). Outputs S, P, S_arg [n]; returns g_max_index (-1: exact bailed out) */
int64_t vmo_chain_linked_raw(const int64_t* anchors, int64_t n, int which, int kmersize, double skipcost, int maxdiff, int maxgap,
                             double g_max_scores, int64_t g_max_index, const double* pre_S, const int64_t* pre_P, int64_t n_pre,
                             int64_t prereadloc, double* S, int64_t* P, int64_t* S_arg);

/* DP problem recorder: when enabled, every k_cigar_global / k_extend / edit distance call made
 * inside vmo_extend appends (kind, tl, ql) to a log (golden V5) */
/* golden V4 stage entry (segment surgery, mammap_clrnano.py:23437 / :726 / :16736 / :24226): see vmo_extend.cc */
int vmo_stage_v4(const vmo_index*, int fn, const int64_t* rows_in, int64_t n_in, int64_t arg, const char* read, int64_t readlen,
                 int64_t** rows_out, int64_t* n_out, int* ret);
const char* vmo_last_error(void);
/* how often GC-fast / LC-fast / LC-mm-fast ran in this process since the last reset (tests: which paths a case exercised) */
void vmo_fast_counters(int64_t out[3], int reset);
/* tests: how often the segment surgery took its rare branches: drop_misplaced removals, merges, fix_simple_inv shifts (both branches) */
void vmo_surgery_counters(int64_t out[4], int reset);

#ifdef __cplusplus
}
#endif
#endif
