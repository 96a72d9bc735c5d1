// vmo_internal.h — shared internals of the CPU ORACLE (test infrastructure; see vmo.h header).
#ifndef VMO_INTERNAL_H
#define VMO_INTERNAL_H
#include "vmo.h"
#include <cstdint>
#include <string>
#include <vector>

namespace vmo {

struct Anchor { int64_t q, r, s, l; };   // (read pos, global ref pos, strand +-1, length) — SURVEY §8 row T
struct Mz { uint64_t h; int32_t pos; int8_t strand; };
typedef std::vector<Anchor> Path;
struct Record;

struct Nt4Table { uint8_t t[256]; Nt4Table(); };
extern const Nt4Table NT4T;
#define NT4 NT4T.t

void set_error(const std::string& s);
void sketch(const char* seq, int64_t len, int k, int w, std::vector<Mz>& out);
void map_read(const vmo_index* mi, const char* seq, int64_t len, int check_num, int mid_occ, std::vector<Anchor>& out);

const std::string& index_seq(const vmo_index* mi, int c);
int64_t index_offset(const vmo_index* mi, int c);
int index_nseq(const vmo_index* mi);

// reference cost tables, recomputed with libm + the NumPy patches of vmo_tables_patch.h
struct Tables {
    std::vector<float> extra;          // mammap_clrnano.py:15371-15376
    std::vector<float> readgap_h;      // :26567-26569  0.1*log2(r+1)   (H/L/S)
    std::vector<float> readgap_r;      // mammap_noprefercloser.py:16534  0.1*log2(r)
    std::vector<float> large_readgap;  // :28270-28275 (maxgap+1 entries are used; built for r < 100)
    std::vector<double> log2cache;     // :27530  0.5*log2(g+1), 100000 entries
    std::vector<double> log2int;       // log2(g), g = 0..1024 (gapcost_list :24846, :27320)
};
const Tables& tables();

std::string revcomp(const std::string& s);


struct ChainSet {                      // result of decode_hit (:23981-24020)
    bool need_reverse = false;
    int mapq = 0;
    double score = 0.;                 // signed like decode_hit's return (negative when need_reverse); 0 = unmapped
    std::vector<Path> paths;           // return_path_list: primary + secondaries (descending read order)
    std::vector<double> all_scores;    // scores_list
    bool fast_used = false;
};
bool strand_flip(std::vector<Anchor>& a, int64_t readlen);

// gap geometry of the -mode asm fork (mammap_asm.py:20660-20688; the same lines in its GC-fast :20866+, LC :16641+ and linked DPs):
// the overlap case is written with non_overlap_size = q_i - q_j and the opposite-strand cases carry no +-1
static inline void gap_geometry_asm(const Anchor& ai, const Anchor& aj, int64_t& readgap, int64_t& refgap, int64_t& bonus) {
    readgap = ai.q - aj.q - aj.l;
    if (readgap < 0) {
        bonus = ai.q + ai.l - aj.q - aj.l;
        readgap = 0;
        const int64_t nov = ai.q - aj.q;
        if (ai.s == aj.s) { if (ai.s == 1) refgap = ai.r - aj.r - nov; else refgap = aj.r + aj.l - nov - ai.r - ai.l; }
        else { if (aj.s == -1) refgap = ai.r + aj.l - nov - aj.r; else refgap = ai.r + ai.l - aj.r - nov; }
    } else {
        bonus = ai.l;
        if (ai.s == aj.s) { if (ai.s == 1) refgap = ai.r - aj.r - aj.l; else refgap = aj.r - ai.r - ai.l; }
        else { if (aj.s == -1) refgap = ai.r - aj.r; else refgap = ai.r + ai.l - aj.r - aj.l; }
    }
}

// carried state of a LINKED chain DP (mammap_asm.py:21686 / :21871 / :21504): scores / negated predecessors of the anchors kept from the
// previous batch (they are the first rows of A), the running maximum and the largest read position among them
struct LinkState { const double* pre_S = nullptr; const int64_t* pre_P = nullptr; int64_t n_pre = 0; double g_max_scores = 0.; int64_t g_max_index = 0; int64_t prereadloc = 0; };
// the asm fork's exact DPs (vmo_asm.cc). lc = false: GC-exact (:20551; linked :21686); lc = true: the linked LC (:21504). Returns g_max_index, -1 = bail-out
int64_t chain_exact_asm(const std::vector<Anchor>& A, int kmersize, double skipcost, int maxdiff, int maxgap, bool lc, const LinkState* link,
                        std::vector<double>& S, std::vector<int64_t>& P, std::vector<int64_t>& S_arg);
// GC-fast of the asm fork (:20738; linked :21871) — vmo_chain_fast.cc
int64_t chain_global_fast_asm(const std::vector<Anchor>& A, int kmersize, double skipcost, int maxdiff, int maxgap, const LinkState* link,
                              std::vector<double>& S, std::vector<int64_t>& P, std::vector<int64_t>& S_arg);
// the asm fork's local chain DP (:16540) on anchors sorted by read start — vmo_asm.cc
int local_chain_asm(const std::vector<Anchor>& A, int kmersize, double skipcost, int maxdiff, int maxgap, double* score, Path& path);
// local re-seeding around one guide chain (vmo_local.cc; get_localmap_..._guide_1 :23069 = mammap_asm.py:17960 / :22477). r_st < 0: the read
// window follows the guide (+- read_span); otherwise positions [r_st, r_en) are looked up (collect_second_round_anchors)
void local_seed_one(const vmo_index* mi, const std::string& read, Path guide, int k, int64_t look_span, int64_t read_span, std::vector<Anchor>& out,
                    int64_t r_st = -1, int64_t r_en = -1);
void get_query_target_for_cigar(const vmo_index* mi, const Anchor& pre, const Anchor& now, const std::string& read,
                                const std::string& rc, int64_t L, std::string& target, std::string& query);
// ass_extend_func (mammap_asm.py:23423) — vmo_extend.cc
int ass_extend_func(const vmo_index* mi, const std::string& read, const std::string& rc, Path chain_asc, const vmo_params& prm, std::vector<Record>& recs);
int align_asm(const vmo_index* mi, const std::string& contig, const vmo_params& prm, int64_t split_len, int64_t batch_anchors, int64_t window,
              std::vector<Record>& recs);
int decode_hit(std::vector<Anchor> A, int64_t readlen, int kmersize, const vmo_params& prm, ChainSet& out);

// contig helpers (pos2contig :51-59 — last contig whose start <= pos)
int pos2contig(const vmo_index* mi, int64_t gpos);

// local stage (vmo_local.cc). read/rc in chain orientation. returns 0, or <0 when the reference would raise
int local_chain(const vmo_index* mi, const std::string& read, const std::string& rc, const std::vector<Path>& guides,
                const vmo_params& prm, double* score, Path& chain_desc, std::vector<Anchor>* raw_out, int* variant);

// DP primitives (vmo_dp.cc)
int k_cigar_global(const char* t, int64_t tl, const char* q, int64_t ql, int match, int mismatch, int o1, int e1,
                   int o2, int e2, int eqx, std::string& cigar, int32_t* score_out);
int k_extend(const char* t, int64_t tl, const char* q, int64_t ql, int match, int mismatch, int o, int e, int bw,
             int zdrop, int32_t* t_e, int32_t* q_e);
int64_t edit_distance_str(const std::string& a, const std::string& b);

struct Record { int contig; int strand; int mapq; int64_t q_st, q_en, r_st, r_en; std::string cigar; };
// extend stage (vmo_extend.cc): chain in ascending read order
int extend_func(const vmo_index* mi, const std::string& read, const std::string& rc, Path chain_asc, int mapq,
                bool need_reverse, bool nofilter, const vmo_params& prm, std::vector<Record>& recs, bool* filtered);
bool pairedindel(const std::vector<std::string>& cigars, double indelsize);
int align_read(const vmo_index* mi, const std::string& read, const vmo_params& prm, std::vector<Record>& recs);

// DP problem log (golden V5): kind 0 = global k_cigar, 1 = extension, 2 = edit distance
struct DpCall { int kind; std::string t, q; };
extern thread_local std::vector<DpCall>* g_dplog;   // Bio.Seq reverse_complement restricted to ACGTN upper

}  // namespace vmo
#endif
