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

typedef std::vector<Anchor> Path;

struct ChainSet {                      // result of decode_hit (:23981-24020)
    bool need_reverse = false;
    int mapq = 0;
    double score = 0.;                 // signed like decode_hit's return (negative when need_reverse); 0 = unmapped
    std::vector<Path> paths;           // return_path_list: primary + secondaries (descending read order)
    std::vector<double> all_scores;    // scores_list
    bool fast_used = false;
};
bool strand_flip(std::vector<Anchor>& a, int64_t readlen);
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
