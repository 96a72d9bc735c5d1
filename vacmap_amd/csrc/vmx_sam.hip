// vmx_sam.hip — native (host, multi-threaded) I/O around the batched path: FASTA / FASTQ(.gz) reader into read blobs and SAM text
// emission for the records of vm_align_batch. SURVEY §8(f) rank 3: at GPU rates the Python string handling of the driver is the
// bottleneck (measured: 3.2 k reads/s through vacmap_amd/sam.py with 16 processes against 86 k reads/s of the resident pipeline).
//
// The emitter restates vacmap_amd/sam.py function by function (which is pinned line by line against the reference's own
// get_bam_dict_str / _comments output, tests/golden/sam.json): reassign_mapq (/root/reference/src/vacmap/mammap_clrnano.py:11661),
// record order (:20855-20856), mergecigar_ (:4773), nm_from_cigar (output_functions.py:300), MD / cs (:19012, :19062), approximate SA
// CIGARs (--fakecigar), CG tag switch (:20962, Q4), P_alignmentstring (:5391) with comment copying (:20686). A read whose emission would
// raise in the reference (an index past a sequence end) produces no line and is counted as skipped, like the worker's except (:24127-24134).
// No device code in this file.
#include "vmx_index_priv.h"
#include <atomic>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include <zlib.h>
#include <fcntl.h>
#include <unistd.h>
#include <sys/uio.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <algorithm>
#include <errno.h>

using namespace vmx;

// ------------------------------------------------------------------------------------------------ reference bases on the host
// upper-case bases of the whole reference: the index's host copy, or (replicas made by vm_index_from_meta) decoded once from HBM
static const std::string& host_bases(const vm_index* mi) {
    if (mi->has_host_seq) return mi->bases;
    vm_index* m = const_cast<vm_index*>(mi);
    // concurrent vm_sam_emit jobs share the index: exactly one of them decodes, the others wait for it (call_once), and nobody
    // swaps the buffer under a reader afterwards
    std::call_once(m->host_seq_once, [m] {
        std::string dec((size_t)m->offsets.back(), 'N');
        for (size_t i = 0; i < m->names.size(); ++i) if (m->lens[i] > 0) vm_index_seq(m, (int)i, 0, m->lens[i], &dec[(size_t)m->offsets[i]]);
        m->bases.swap(dec);
    });
    return mi->bases;
}

namespace {

struct Raise {};        // what is an IndexError in the Python counterpart

struct Rec { int contig; char strand; int mapq; int64_t q_st, q_en, r_st, r_en; const char* cigar; int64_t cigar_len; };     // cigar: a view into the batch's blob

struct Ops { std::vector<int64_t> n; std::string op; };     // merged CIGAR: run lengths and operators

// mergecigar_ :4773 — consecutive operators of the same kind are merged
static void merge_cigar(const char* c, int64_t len, Ops& o) {
    o.n.clear(); o.op.clear();
    int64_t num = 0;
    for (int64_t i = 0; i < len; ++i) {
        const char ch = c[i];
        if (ch >= '0' && ch <= '9') { num = num * 10 + (ch - '0'); continue; }
        if (!o.op.empty() && o.op.back() == ch) o.n.back() += num; else { o.n.push_back(num); o.op.push_back(ch); }
        num = 0;
    }
}
static inline void put_int(std::string& s, int64_t v) {      // (a CIGAR holds thousands of numbers: no snprintf)
    char b[24]; int n = 24;
    unsigned long long u = v < 0 ? 0ULL - (unsigned long long)v : (unsigned long long)v;
    do { b[--n] = (char)('0' + u % 10); u /= 10; } while (u);
    if (v < 0) b[--n] = '-';
    s.append(b + n, (size_t)(24 - n));
}
static void join_ops(const Ops& o, std::string& s) { s.clear(); for (size_t i = 0; i < o.op.size(); ++i) { put_int(s, o.n[i]); s.push_back(o.op[i]); } }
static inline char up(char c) { return (c >= 'a' && c <= 'z') ? (char)(c - 32) : c; }
static inline char lo(char c) { return (c >= 'A' && c <= 'Z') ? (char)(c + 32) : c; }

// mergecigar_ (:4773) + nm_from_cigar (output_functions.py:300) in ONE pass over the text (the path without MD / cs): adjacent runs of the
// same operator are merged straight into the output string (only the joins between gap-fill pieces ever merge), NM is accumulated run by
// run, and the number of merged operators is returned for the CG-tag test. Same results as merge_cigar -> join_ops -> nm_from_cigar.
static int64_t merge_cigar_nm(const char* c, int64_t len, const char* q, int64_t ql, const char* t, int64_t tl, std::string& cig, int64_t* n_ops) {
    cig.clear();
    int64_t nm = 0, qp = 0, rp = 0, nops = 0;
    int64_t cur_n = 0; char cur_op = 0;
    auto flush = [&]() {
        if (!cur_op) return;
        put_int(cig, cur_n); cig.push_back(cur_op); ++nops;
        const int64_t n = cur_n;
        switch (cur_op) {
            case 'M': if (qp + n > ql || rp + n > tl) throw Raise(); { const char* a = q + qp; const char* b = t + rp; int64_t d = 0; for (int64_t x = 0; x < n; ++x) d += ((a[x] ^ b[x]) & 0xDF) != 0; nm += d; } qp += n; rp += n; break;
            case '=': qp += n; rp += n; break;
            case 'X': nm += n; qp += n; rp += n; break;
            case 'I': nm += n; qp += n; break;
            case 'D': nm += n; rp += n; break;
            case 'S': qp += n; break;
            case 'N': rp += n; break;
            default: break;
        }
    };
    int64_t num = 0;
    for (int64_t i = 0; i < len; ++i) {
        const char ch = c[i];
        if (ch >= '0' && ch <= '9') { num = num * 10 + (ch - '0'); continue; }
        if (ch == cur_op) cur_n += num; else { flush(); cur_op = ch; cur_n = num; }
        num = 0;
    }
    flush();
    *n_ops = nops;
    return nm;
}

// NM of the -mode asm emitter (mergecigar_nm_, mammap_asm.py:23125-23155): the lengths of the X / D / I runs of the UNMERGED text, where a run that
// continues the operator before it (the joins between gap-fill pieces) is merged but not counted (:23142-23145)
static int64_t nm_asm_text(const char* c, int64_t len) {
    int64_t nm = 0, num = 0; char pre = 0;
    for (int64_t i = 0; i < len; ++i) {
        const char ch = c[i];
        if (ch >= '0' && ch <= '9') { num = num * 10 + (ch - '0'); continue; }
        if (ch != pre) { if (ch == 'X' || ch == 'D' || ch == 'I') nm += num; pre = ch; }
        num = 0;
    }
    return nm;
}

// nm_from_cigar (output_functions.py:300): q / t are the sequences the CIGAR walks from their position 0
static int64_t nm_from_cigar(const Ops& o, const char* q, int64_t ql, const char* t, int64_t tl) {
    int64_t nm = 0, qp = 0, rp = 0;
    for (size_t i = 0; i < o.op.size(); ++i) {
        const int64_t n = o.n[i];
        switch (o.op[i]) {
            case 'M': if (qp + n > ql || rp + n > tl) throw Raise(); { const char* a = q + qp; const char* b = t + rp; int64_t d = 0; for (int64_t x = 0; x < n; ++x) d += ((a[x] ^ b[x]) & 0xDF) != 0; nm += d; } qp += n; rp += n; break;   /* letters compared case-insensitively */
            case '=': qp += n; rp += n; break;
            case 'X': nm += n; qp += n; rp += n; break;
            case 'I': nm += n; qp += n; break;
            case 'D': nm += n; rp += n; break;
            case 'S': qp += n; break;
            case 'N': rp += n; break;
            default: break;
        }
    }
    return nm;
}

// get_MD_CSshort :19012 / get_MD_CSlong :19062 on a merged =/X/I/D CIGAR; any other operator but S/H gives two empty strings
static void md_cs(const Ops& o, const char* t, int64_t tl, const char* q, int64_t ql, bool shortcs, std::string& md, std::string& cs) {
    md.clear(); cs.clear();
    int64_t refloc = 0, readloc = 0, equal = 0; char preop = 0;
    auto T = [&](int64_t i) -> char { if (i < 0 || i >= tl) throw Raise(); return t[i]; };
    auto Q = [&](int64_t i) -> char { if (i < 0 || i >= ql) throw Raise(); return q[i]; };
    auto tslice = [&](int64_t a, int64_t b, std::string& out, bool lower, bool upper) { if (a > tl) a = tl; if (b > tl) b = tl; for (int64_t i = a; i < b; ++i) out.push_back(lower ? lo(t[i]) : (upper ? up(t[i]) : t[i])); };
    for (size_t i = 0; i < o.op.size(); ++i) {
        const int64_t n = o.n[i]; const char op = o.op[i];
        if (op == 'X') {
            if (equal > 0) put_int(md, equal); else if (preop == 'D') md.push_back('0');
            md.push_back(T(refloc));
            cs.push_back('*'); cs.push_back(lo(T(refloc))); cs.push_back(lo(Q(readloc)));
            for (int64_t j = 1; j < n; ++j) { md.push_back('0'); md.push_back(T(refloc + j)); cs.push_back('*'); cs.push_back(lo(T(refloc + j))); cs.push_back(lo(Q(readloc + j))); }
            refloc += n; readloc += n; equal = 0;
        } else if (op == '=') {
            if (shortcs) { cs.push_back(':'); put_int(cs, n); } else { cs.push_back('='); tslice(refloc, refloc + n, cs, false, true); }
            refloc += n; readloc += n; equal += n;
        } else if (op == 'D') {
            if (equal > 0) put_int(md, equal); else if (preop == 'X') md.push_back('0');
            md.push_back('^'); tslice(refloc, refloc + n, md, false, false);
            cs.push_back('-'); tslice(refloc, refloc + n, cs, true, false);
            refloc += n; equal = 0;
        } else if (op == 'I') {
            cs.push_back('+'); { int64_t a = readloc, b = readloc + n; if (a > ql) a = ql; if (b > ql) b = ql; for (int64_t x = a; x < b; ++x) cs.push_back(lo(q[x])); }
            readloc += n;
            continue;
        } else if (op == 'S' || op == 'H') continue;
        else { md.clear(); cs.clear(); return; }
        preop = op;
    }
    if (equal > 0) put_int(md, equal);
}

static void fake_cigar(const Rec& r, int64_t qlen, char clip, std::string& s) {
    s.clear();
    if (r.q_st > 0) { put_int(s, r.q_st); s.push_back(clip); }
    const int64_t diff = r.q_en - r.q_st - r.r_en + r.r_st;
    if (diff > 0) { put_int(s, r.r_en - r.r_st); s.push_back('M'); put_int(s, diff); s.push_back('I'); }
    else if (diff < 0) { put_int(s, r.q_en - r.q_st); s.push_back('M'); put_int(s, -diff); s.push_back('D'); }
    else { put_int(s, r.q_en - r.q_st); s.push_back('M'); }
    if (qlen - r.q_en > 0) { put_int(s, qlen - r.q_en); s.push_back(clip); }
}

// reassign_mapq :11661
static void reassign_mapq(std::vector<Rec>& recs) {
    const int n = (int)recs.size();
    if (n == 0) return;
    std::vector<int> keep(1, 0);
    while (keep.back() < n - 1) {
        const int i = keep.back(); const Rec& b = recs[(size_t)i];
        bool hit = false; int t = i;
        while (t + 1 < n) {
            ++t; const Rec& x = recs[(size_t)t];
            if (x.contig != b.contig) continue;
            const int64_t refgap = x.strand == '+' ? x.r_st - b.r_en : b.r_st - x.r_en;
            if ((refgap < 0 ? -refgap : refgap) > 100000) continue;
            if (refgap < 10) { keep.push_back(t); hit = true; break; }
        }
        if (!hit) keep.push_back(i + 1);
    }
    std::vector<char> k((size_t)n, 0); for (int x : keep) if (x < n) k[(size_t)x] = 1;
    for (int i = 0; i < n; ++i) if (!k[(size_t)i]) recs[(size_t)i].mapq = 0;
}

static void revcomp_into(const char* s, int64_t n, std::string& out) {
    out.resize((size_t)n);
    static const struct Tab { unsigned char v[256]; Tab() { for (int i = 0; i < 256; ++i) v[i] = (unsigned char)i; v['A'] = 'T'; v['C'] = 'G'; v['G'] = 'C'; v['T'] = 'A'; v['a'] = 't'; v['c'] = 'g'; v['g'] = 'c'; v['t'] = 'a'; } } tab;
    for (int64_t i = 0; i < n; ++i) out[(size_t)i] = (char)tab.v[(unsigned char)s[n - 1 - i]];
}

struct Scratch { std::vector<Rec> recs; std::vector<Ops> ops; std::vector<std::string> cig, md, cs, fake; std::vector<int64_t> nm, nops; std::string rcq, rcqual; };

// SAM lines of one read appended to `out`; returns the number of lines, or -1 when the reference's emitter would raise
static int emit_read(const vm_index* mi, const std::string& bases, const vm_sam_opts* o, const char* name, int64_t name_len, const char* query, int64_t qlen, const char* qual,
                     int64_t qual_len, const char* com, int64_t com_len, const vm_record* rr, int64_t nr, const char* blob, Scratch& S, std::string& out) {
    const size_t out0 = out.size();
    try {
        S.recs.resize((size_t)nr);
        for (int64_t i = 0; i < nr; ++i) {
            Rec& r = S.recs[(size_t)i];
            r.contig = rr[i].contig; r.strand = rr[i].strand == 1 ? '+' : '-'; r.mapq = rr[i].mapq; r.q_st = rr[i].q_st; r.q_en = rr[i].q_en; r.r_st = rr[i].r_st; r.r_en = rr[i].r_en;
            r.cigar = blob + rr[i].cigar_off; r.cigar_len = rr[i].cigar_len;
        }
        if (o->markunbalancetra) reassign_mapq(S.recs);
        // sort by query span ascending (stable), then reverse: longest first, later ones first among equals (:20855-20856)
        std::stable_sort(S.recs.begin(), S.recs.end(), [](const Rec& a, const Rec& b) { return a.q_en - a.q_st < b.q_en - b.q_st; });
        std::reverse(S.recs.begin(), S.recs.end());
        bool need_rc = false; for (const Rec& r : S.recs) need_rc = need_rc || r.strand == '-';
        if (need_rc) revcomp_into(query, qlen, S.rcq);
        const bool has_qual = qual != nullptr && qual_len == qlen;
        if (need_rc && has_qual) { S.rcqual.assign(qual, (size_t)qlen); std::reverse(S.rcqual.begin(), S.rcqual.end()); }
        const size_t n = S.recs.size();
        S.ops.resize(n); S.cig.resize(n); S.md.resize(n); S.cs.resize(n); S.fake.resize(n); S.nm.resize(n); S.nops.resize(n);
        const char clip = o->hardclip ? 'H' : 'S';
        for (size_t i = 0; i < n; ++i) {
            Rec& r = S.recs[i];
            const char* qs = r.strand == '+' ? query : S.rcq.data();
            const int64_t clen = mi->lens[(size_t)r.contig];
            int64_t ta = r.r_st < 0 ? 0 : (r.r_st > clen ? clen : r.r_st), tb = r.r_en < 0 ? 0 : (r.r_en > clen ? clen : r.r_en);     // Python slice semantics of contig[a:b]
            if (tb < ta) tb = ta;
            const char* t = bases.data() + mi->offsets[(size_t)r.contig] + ta; const int64_t tl = tb - ta;
            if (!o->md && o->asm_mode) {                           // asm: no walk over the sequences at all
                merge_cigar(r.cigar, r.cigar_len, S.ops[i]); join_ops(S.ops[i], S.cig[i]); S.nops[i] = (int64_t)S.ops[i].op.size();
                S.nm[i] = nm_asm_text(r.cigar, r.cigar_len);
            } else if (!o->md) S.nm[i] = merge_cigar_nm(r.cigar, r.cigar_len, qs, qlen, t, tl, S.cig[i], &S.nops[i]);      // one pass: merged text + NM
            else {
                merge_cigar(r.cigar, r.cigar_len, S.ops[i]);
                join_ops(S.ops[i], S.cig[i]);
                S.nops[i] = (int64_t)S.ops[i].op.size();
                int64_t qa = r.q_st < 0 ? 0 : (r.q_st > qlen ? qlen : r.q_st), qb = r.q_en < 0 ? 0 : (r.q_en > qlen ? qlen : r.q_en); if (qb < qa) qb = qa;
                md_cs(S.ops[i], t, tl, qs + qa, qb - qa, o->shortcs != 0, S.md[i], S.cs[i]);
                S.nm[i] = o->asm_mode ? nm_asm_text(r.cigar, r.cigar_len) : nm_from_cigar(S.ops[i], qs + qa, qb - qa, t, tl);
            }
            if (o->fakecigar) fake_cigar(r, qlen, clip, S.fake[i]);
        }
        int lines = 0;
        const size_t primary = (o->asm_mode && n > 1 && S.recs[0].mapq == 1 && S.recs[1].mapq != 1) ? 1 : 0;     // mammap_asm.py:22847-22850
        auto mq_out = [&](int v) { return o->asm_mode ? (v != 0 ? 60 : 1) : v; };
        for (size_t i = 0; i < n; ++i) {
            const Rec& r = S.recs[i];
            const bool cg = 2 * S.nops[i] > 65535 && o->cigar2cg;          // Q4: the reference counts two list entries per operator
            out.append(name, (size_t)name_len); out.push_back('\t');
            put_int(out, (i == primary ? 0 : 2048) + (r.strand == '+' ? 0 : 16)); out.push_back('\t');
            out.append(mi->names[(size_t)r.contig]); out.push_back('\t');
            put_int(out, r.r_st + 1); out.push_back('\t');
            put_int(out, mq_out(r.mapq)); out.push_back('\t');
            if (cg) out.push_back('*'); else out.append(S.cig[i]);
            out.append("\t*\t0\t0\t");
            const char* sq = r.strand == '+' ? query : S.rcq.data();
            const char* ql_ = r.strand == '+' ? qual : S.rcqual.data();
            int64_t a = 0, b = qlen;
            if (o->hardclip) { a = r.q_st < 0 ? 0 : (r.q_st > qlen ? qlen : r.q_st); b = r.q_en < 0 ? 0 : (r.q_en > qlen ? qlen : r.q_en); if (b < a) b = a; }
            out.append(sq + a, (size_t)(b - a)); out.push_back('\t');
            if (has_qual) out.append(ql_ + a, (size_t)(b - a)); else out.push_back('*');
            bool t_rg = false, t_cg = false, t_sa = false, t_md = false;
            if (o->rg_id) { out.append("\tRG:Z:"); out.append(o->rg_id); t_rg = true; }
            if (cg) { out.append("\tCG:Z:"); out.append(S.cig[i]); t_cg = true; }
            if (n > 1) {
                out.append("\tSA:Z:"); t_sa = true;
                for (size_t x = 0; x < n; ++x) {
                    if (x == i) continue;
                    const Rec& y = S.recs[x];
                    out.append(mi->names[(size_t)y.contig]); out.push_back(','); put_int(out, y.r_st + 1); out.push_back(','); out.push_back(y.strand); out.push_back(',');
                    out.append(o->fakecigar ? S.fake[x] : S.cig[x]); out.push_back(','); put_int(out, mq_out(y.mapq)); out.push_back(','); put_int(out, S.nm[x]); out.push_back(';');
                }
            }
            out.append("\tNM:i:"); put_int(out, S.nm[i]);
            if (o->md) { out.append("\tMD:Z:"); out.append(S.md[i]); out.append("\tcs:Z:"); out.append(S.cs[i]); t_md = true; }
            if (com && com_len > 0) {
                // :20686 — tab-separated XX:T:value fields of the FASTA/Q comment whose tag is not on the line yet
                std::vector<std::string> seen;
                auto present = [&](const std::string& tg) {
                    static const char* fixed[] = {"QNAME", "FLAG", "RNAME", "POS", "MAPQ", "CIGAR", "RNEXT", "PNEXT", "TLEN", "SEQ", "QUAL", "SA", "NM", "MD", "cs"};
                    for (const char* f : fixed) if (tg == f) return true;
                    if ((tg == "RG" && t_rg) || (tg == "CG" && t_cg)) return true;
                    (void)t_sa; (void)t_md;
                    for (auto& s : seen) if (s == tg) return true;
                    return false;
                };
                int64_t p = 0;
                while (p <= com_len) {
                    int64_t e = p; while (e < com_len && com[e] != '\t') ++e;
                    // one.split(':') must give exactly three parts: two colons
                    int colons = 0; int64_t c1 = -1, c2 = -1;
                    for (int64_t x = p; x < e; ++x) if (com[x] == ':') { ++colons; if (c1 < 0) c1 = x; else if (c2 < 0) c2 = x; }
                    if (colons == 2 && c1 - p == 2 && c2 - c1 == 2) {
                        const std::string tg(com + p, 2); const char ty = com[c1 + 1];
                        if (!present(tg) && (ty == 'A' || ty == 'i' || ty == 'f' || ty == 'Z' || ty == 'H' || ty == 'B')) { out.push_back('\t'); out.append(com + p, (size_t)(e - p)); seen.push_back(tg); }
                    }
                    p = e + 1;
                }
            }
            out.push_back('\n');
            ++lines;
        }
        return lines;
    } catch (const Raise&) {
        out.resize(out0);
        return -1;
    }
}

}  // namespace

extern "C" {

int vm_sam_emit(const vm_index* mi, const vm_sam_opts* o, int64_t n_reads, const char* names, const int64_t* name_off, const char* seqs, const int64_t* seq_off,
                const char* quals, const int64_t* qual_off, const char* comments, const int64_t* com_off, const vm_record* recs, int64_t n_recs, const char* cigar_blob,
                const int32_t* status, int nthreads, char** text, int64_t** text_off, int64_t* n_lines, int64_t* n_skipped) {
    *text = nullptr; *text_off = nullptr; *n_lines = 0; *n_skipped = 0;
    try {
        const std::string& bases = host_bases(mi);
        // records are grouped by read_idx (ascending) as vm_align_batch returns them
        std::vector<int64_t> first((size_t)n_reads + 1, 0);
        for (int64_t i = 0; i < n_recs; ++i) { if (recs[i].read_idx < 0 || recs[i].read_idx >= n_reads) { set_error("vm_sam_emit: record of an unknown read"); return VM_ERR_ARG; } first[(size_t)recs[i].read_idx + 1]++; }
        for (int64_t r = 0; r < n_reads; ++r) first[(size_t)r + 1] += first[(size_t)r];
        for (int64_t i = 1; i < n_recs; ++i) if (recs[i].read_idx < recs[i - 1].read_idx) { set_error("vm_sam_emit: records must be ordered by read"); return VM_ERR_ARG; }
        if (nthreads < 1) nthreads = 1;
        if (nthreads > 64) nthreads = 64;
        const int64_t CH = 64;                                   // reads per work item
        const int64_t nch = (n_reads + CH - 1) / CH;
        std::vector<std::string> part((size_t)nch);
        std::vector<int64_t> rlen((size_t)n_reads, 0);
        std::atomic<int64_t> next(0), lines(0), skipped(0);
        auto work = [&]() {
            Scratch S;
            while (true) {
                const int64_t c = next.fetch_add(1); if (c >= nch) break;
                std::string& out = part[(size_t)c];
                { int64_t est = 0; for (int64_t r = c * CH; r < std::min(n_reads, (c + 1) * CH); ++r) est += 2 * (seq_off[r + 1] - seq_off[r]) + (seq_off[r + 1] - seq_off[r]) / 2 + 512; out.reserve((size_t)est); }
                for (int64_t r = c * CH; r < std::min(n_reads, (c + 1) * CH); ++r) {
                    const size_t before = out.size();
                    if (status && status[r] != 0) { skipped.fetch_add(1); continue; }
                    const int64_t nr = first[(size_t)r + 1] - first[(size_t)r];
                    if (nr == 0) continue;
                    const int64_t ql = qual_off ? qual_off[r + 1] - qual_off[r] : 0;
                    const int64_t cl = com_off ? com_off[r + 1] - com_off[r] : 0;
                    const int rc = emit_read(mi, bases, o, names + name_off[r], name_off[r + 1] - name_off[r], seqs + seq_off[r], seq_off[r + 1] - seq_off[r],
                                             (quals && ql > 0) ? quals + qual_off[r] : nullptr, ql, (comments && cl > 0) ? comments + com_off[r] : nullptr, cl,
                                             recs + first[(size_t)r], nr, cigar_blob, S, out);
                    if (rc < 0) skipped.fetch_add(1); else lines.fetch_add(rc);
                    rlen[(size_t)r] = (int64_t)(out.size() - before);
                }
            }
        };
        if (nthreads == 1 || nch <= 1) work();
        else { std::vector<std::thread> th; for (int t = 0; t < nthreads && t < nch; ++t) th.emplace_back(work); for (auto& t : th) t.join(); }
        size_t tot = 0; std::vector<size_t> pstart(part.size() + 1, 0);
        for (size_t i = 0; i < part.size(); ++i) { pstart[i] = tot; tot += part[i].size(); }
        *text = (char*)malloc(tot + 1); *text_off = (int64_t*)malloc(8 * ((size_t)n_reads + 1));
        if (!*text || !*text_off) { free(*text); free(*text_off); *text = nullptr; *text_off = nullptr; set_error("out of host memory"); return VM_ERR_OOM; }
        {   // the parts are copied into place by the same number of threads (a batch's text is ~150 MB)
            std::atomic<size_t> nx(0);
            auto cp = [&]() { while (true) { const size_t i = nx.fetch_add(1); if (i >= part.size()) break; memcpy(*text + pstart[i], part[i].data(), part[i].size()); std::string().swap(part[i]); } };
            if (nthreads == 1 || part.size() <= 1) cp();
            else { std::vector<std::thread> th; for (int t = 0; t < nthreads && (size_t)t < part.size(); ++t) th.emplace_back(cp); for (auto& t : th) t.join(); }
        }
        (*text)[tot] = 0;
        int64_t acc = 0; for (int64_t r = 0; r < n_reads; ++r) { (*text_off)[r] = acc; acc += rlen[(size_t)r]; }
        (*text_off)[n_reads] = acc;
        *n_lines = lines.load(); *n_skipped = skipped.load();
        return VM_OK;
    }
    catch (const std::bad_alloc&) { set_error("vm_sam_emit: out of host memory"); return VM_ERR_OOM; }
    catch (const std::exception& e) { set_error(std::string("vm_sam_emit: ") + e.what()); return VM_ERR_ARG; }
}

// out = the entries idx[0..n) of a blob (offsets off) back to back, out_off[n + 1]; out must hold the sum of their lengths (the
// driver's window -> length-binned batch and batch -> input order shuffles of read / name / quality / SAM text blobs)
void* vm_pinned_alloc(int64_t bytes, int device) {
    void* p = nullptr;
    if (device >= 0 && hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    if (bytes <= 0 || hipHostMalloc(&p, (size_t)bytes, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
void vm_pinned_free(void* p) { if (p) (void)hipHostFree(p); }

int64_t vm_blob_gather(const char* blob, const int64_t* off, const int64_t* idx, int64_t n, char* out, int64_t* out_off) {
    int64_t w = 0;
    for (int64_t j = 0; j < n; ++j) { const int64_t a = off[idx[j]], b = off[idx[j] + 1]; out_off[j] = w; if (out && b > a) memcpy(out + w, blob + a, (size_t)(b - a)); w += b - a; }
    out_off[n] = w;
    return w;
}

// the same over several blobs: entry j of the output is entry idx[j] of blob part[j] (the writer's batch texts -> input order, without
// concatenating the batches first). out must hold the sum of the lengths; returns that sum.
// the same merge written straight to a file descriptor with writev(): no assembled copy of the window's text (a gigabyte that was written once
// into fresh memory and read again by write()). Entries that are neighbours in memory go out as one iovec. Returns the bytes written or -1.
int64_t vm_blob_write_parts(int fd, const char* const* blobs, const int64_t* const* offs, const int32_t* part, const int64_t* idx, int64_t n) {
    std::vector<struct iovec> iov; iov.reserve(1024);
    int64_t total = 0;
    auto flush = [&]() -> bool {
        size_t k = 0;
        while (k < iov.size()) {
            ssize_t w;
            do { w = writev(fd, iov.data() + k, (int)std::min<size_t>(iov.size() - k, 1024)); } while (w < 0 && errno == EINTR);
            if (w < 0) return false;
            total += w;
            while (k < iov.size() && (size_t)w >= iov[k].iov_len) { w -= (ssize_t)iov[k].iov_len; ++k; }
            if (k < iov.size() && w > 0) { iov[k].iov_base = (char*)iov[k].iov_base + w; iov[k].iov_len -= (size_t)w; }
        }
        iov.clear();
        return true;
    };
    for (int64_t j = 0; j < n; ++j) {
        const int64_t* off = offs[part[j]];
        const int64_t a = off[idx[j]], b = off[idx[j] + 1];
        if (b <= a) continue;
        const char* p = blobs[part[j]] + a;
        if (!iov.empty() && (const char*)iov.back().iov_base + iov.back().iov_len == p) { iov.back().iov_len += (size_t)(b - a); continue; }
        if (iov.size() == 1024 && !flush()) { set_error("write failed"); return -1; }
        struct iovec v; v.iov_base = (void*)p; v.iov_len = (size_t)(b - a); iov.push_back(v);
    }
    if (!flush()) { set_error("write failed"); return -1; }
    return total;
}

// The same merge written into a REGULAR file through a shared mapping, by several threads: the file is extended by the window's size, the new
// range is mapped, and `nthreads` threads copy the lines to their places (the page faults and the copies run in parallel; buffered write() calls
// to one file serialise on its inode lock, and one thread moves ~5 GB/s of cold text — less than one GPU produces: at 4 aligned Gbp/s the SAM
// text is ~8 GB/s). file_off: the current end of the file (the caller keeps count). Returns the bytes written, -1 on an error, -2 when the
// descriptor cannot be mapped (a pipe, a terminal, a file system without mmap): the caller falls back to vm_blob_write_parts.
int64_t vm_blob_write_parts_mmap(int fd, int64_t file_off, const char* const* blobs, const int64_t* const* offs, const int32_t* part, const int64_t* idx, int64_t n, int nthreads) {
    std::vector<int64_t> at((size_t)n + 1);
    int64_t w = 0;
    for (int64_t j = 0; j < n; ++j) { const int64_t* off = offs[part[j]]; at[(size_t)j] = w; w += off[idx[j] + 1] - off[idx[j]]; }
    at[(size_t)n] = w;
    if (w == 0) return 0;
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) return -2;
    if (ftruncate(fd, (off_t)(file_off + w)) != 0) return -2;
    const long pg = sysconf(_SC_PAGESIZE);
    const int64_t map_off = file_off / pg * pg, lead = file_off - map_off;
    void* m = mmap(nullptr, (size_t)(w + lead), PROT_READ | PROT_WRITE, MAP_SHARED, fd, (off_t)map_off);
    if (m == MAP_FAILED) { (void)!ftruncate(fd, (off_t)file_off); return -2; }
    char* out = (char*)m + lead;
    auto copy = [&](int64_t j0, int64_t j1) {
        for (int64_t j = j0; j < j1; ++j) {
            const int64_t* off = offs[part[j]];
            const int64_t a = off[idx[j]], b = off[idx[j] + 1];
            if (b > a) memcpy(out + at[(size_t)j], blobs[part[j]] + a, (size_t)(b - a));
        }
    };
    const int T = (w > ((int64_t)16 << 20) && n >= 64) ? std::max(1, std::min(nthreads, 16)) : 1;
    if (T == 1) copy(0, n);
    else {
        std::vector<std::thread> th;
        int64_t j0 = 0;
        for (int t = 0; t < T; ++t) {                            // cut by bytes, not by entries
            const int64_t target = w * (t + 1) / T;
            int64_t j1 = t + 1 == T ? n : (int64_t)(std::lower_bound(at.begin() + j0, at.begin() + n, target) - at.begin());
            if (j1 < j0) j1 = j0;
            th.emplace_back(copy, j0, j1);
            j0 = j1;
        }
        for (auto& t : th) t.join();
    }
    munmap(m, (size_t)(w + lead));
    return w;
}

int64_t vm_blob_gather_parts(const char* const* blobs, const int64_t* const* offs, const int32_t* part, const int64_t* idx, int64_t n, char* out) {
    // a window's SAM text is gigabytes: the output offsets come from one pass over the lengths, the copies run on a few threads
    // (one thread moves ~5 GB/s; the writer thread of the driver was the longest stage of its loop)
    std::vector<int64_t> at((size_t)n + 1);
    int64_t w = 0;
    for (int64_t j = 0; j < n; ++j) { const int64_t* off = offs[part[j]]; at[(size_t)j] = w; w += off[idx[j] + 1] - off[idx[j]]; }
    at[(size_t)n] = w;
    auto copy = [&](int64_t j0, int64_t j1) {
        for (int64_t j = j0; j < j1; ++j) {
            const int64_t* off = offs[part[j]];
            const int64_t a = off[idx[j]], b = off[idx[j] + 1];
            if (b > a) memcpy(out + at[(size_t)j], blobs[part[j]] + a, (size_t)(b - a));
        }
    };
    const int T = (w > ((int64_t)64 << 20) && n >= 64) ? 4 : 1;
    if (T == 1) { copy(0, n); return w; }
    std::vector<std::thread> th;
    int64_t j0 = 0;
    for (int t = 0; t < T; ++t) {                                // cut by bytes, not by entries
        const int64_t target = w * (t + 1) / T;
        int64_t j1 = t + 1 == T ? n : (int64_t)(std::lower_bound(at.begin() + j0, at.begin() + n, target) - at.begin());
        if (j1 < j0) j1 = j0;
        th.emplace_back(copy, j0, j1);
        j0 = j1;
    }
    for (auto& t : th) t.join();
    return w;
}

// ------------------------------------------------------------------------------------------------ FASTA / FASTQ(.gz) reader
// Plain files are read with read(2) straight into the line buffer (gzread's transparent mode costs a copy); gzip members go through zlib.
// The output blobs grow by realloc (large blocks move by remapping, not by copying) and are handed to the caller as they are.
struct vm_fastx { gzFile f = nullptr; int fd = -1; char* buf = nullptr; size_t cap = 0, len = 0, pos = 0; bool eof = false, io_error = false; size_t hint = 0;
                  int64_t base = 0;          // file offset of buf[0] (plain files)
                  int64_t range_end = -1;    // byte-range reader: records that START at or beyond this offset belong to the next range (-1: to the end)
                  bool fastq = false; };

struct Blob {
    char* p = nullptr; size_t n = 0, cap = 0;
    void need(size_t extra) {
        if (n + extra + 1 <= cap) return;
        size_t c = cap ? cap : (size_t)1 << 16;
        while (c < n + extra + 1) c += c / 2 + 4096;
        char* q = (char*)realloc(p, c);
        if (!q) throw std::bad_alloc();
        p = q; cap = c;
    }
    void append(const char* s, size_t k) { need(k); memcpy(p + n, s, k); n += k; }
    char* release() { need(0); p[n] = 0; char* r = p; p = nullptr; n = cap = 0; return r; }
    ~Blob() { free(p); }
};

static bool fx_fill(vm_fastx* x) {
    if (x->eof) return false;
    if (x->pos > 0) { memmove(x->buf, x->buf + x->pos, x->len - x->pos); x->len -= x->pos; x->base += (int64_t)x->pos; x->pos = 0; }
    const size_t want = (size_t)8 << 20;
    if (x->len + want > x->cap) {
        size_t c = x->cap ? x->cap : 2 * want; while (c < x->len + want) c *= 2;
        char* q = (char*)realloc(x->buf, c); if (!q) throw std::bad_alloc();
        x->buf = q; x->cap = c;
    }
    long n;
    if (x->f) n = gzread(x->f, x->buf + x->len, (unsigned)want);
    else { do { n = (long)read(x->fd, x->buf + x->len, want); } while (n < 0 && errno == EINTR); }
    if (n > 0) x->len += (size_t)n; else x->eof = true;
    if (n < 0) x->io_error = true;                       // a truncated or corrupt .gz (or a read error) is not a clean end of input
    else if (n == 0 && x->f) { int ze = Z_OK; (void)gzerror(x->f, &ze); if (ze != Z_OK && ze != Z_STREAM_END) x->io_error = true; }
    return n > 0;
}
// next line as a VIEW into the read buffer (valid until the next call), without its terminator; false at end of input.
// peek = true leaves the line unconsumed.
static bool fx_line(vm_fastx* x, const char*& p, size_t& len, bool peek = false) {
    while (true) {
        const char* b = x->buf + x->pos; const size_t n = x->len - x->pos;
        const char* nl = n ? (const char*)memchr(b, '\n', n) : nullptr;
        size_t l, adv;
        if (nl) { l = (size_t)(nl - b); adv = l + 1; }
        else { if (fx_fill(x)) continue; if (n == 0) return false; l = n; adv = n; }
        if (l && b[l - 1] == '\r') --l;
        p = b; len = l;
        if (!peek) x->pos += adv;
        return true;
    }
}
// the line `skip` lines below the current one, nothing consumed (the buffer may be refilled: views taken before are stale); false at end of input
static bool fx_peek_ahead(vm_fastx* x, int skip, const char*& p, size_t& len) {
    size_t off = 0;                                      // relative to x->pos, which a refill moves to 0 together with the bytes
    for (int i = 0;; ++i) {
        while (true) {
            const char* b = x->buf + x->pos + off; const size_t n = x->len - x->pos - off;
            const char* nl = n ? (const char*)memchr(b, '\n', n) : nullptr;
            if (nl) {
                const size_t l = (size_t)(nl - b);
                if (i == skip) { p = b; len = (l && b[l - 1] == '\r') ? l - 1 : l; return true; }
                off += l + 1; break;
            }
            if (fx_fill(x)) continue;
            if (n == 0 || i != skip) return false;
            p = b; len = n; return true;
        }
    }
}
static inline void fx_append_upper(Blob& dst, const char* p, size_t n) {
    dst.need(n);
    char* d = dst.p + dst.n;
    for (size_t i = 0; i < n; ++i) { const unsigned char c = (unsigned char)p[i]; d[i] = (char)(c - (((unsigned)(c - 'a') < 26u) << 5)); }      // (branch-free: vectorises)
    dst.n += n;
}

int vm_fastx_open(const char* path, vm_fastx** out) {
    *out = nullptr;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) { set_error(std::string("cannot open ") + path); return VM_ERR_IO; }
    unsigned char magic[2] = {0, 0};
    const long got = (long)pread(fd, magic, 2, 0);
    vm_fastx* x = new vm_fastx();
    if (got == 2 && magic[0] == 0x1f && magic[1] == 0x8b) {
        x->f = gzdopen(fd, "rb");
        if (!x->f) { close(fd); delete x; set_error(std::string("cannot open ") + path); return VM_ERR_IO; }
        gzbuffer(x->f, 1 << 20);
    } else x->fd = fd;
    *out = x;
    return VM_OK;
}
// The records of the byte range [begin, end) of a PLAIN FASTA / FASTQ file: a record belongs to the range its first byte lies in, so N readers
// over N consecutive ranges see every record exactly once, whatever the cut points (sharded driver: one range per rank, cut again into slices for
// the rank's parser threads). The reader seeks to `begin` and resynchronises on the next record start: a line that begins with '>' (FASTA), or with
// '@' and is followed two lines later by a line that begins with '+' (four-line FASTQ — a quality line may begin with '@', but the line two
// below it is a sequence). Compressed input cannot be entered in the middle: VM_ERR_UNSUPPORTED for begin > 0 (the caller parses it on one rank).
int vm_fastx_open_range(const char* path, int64_t begin, int64_t end, vm_fastx** out) {
    const int rc = vm_fastx_open(path, out);
    if (rc != VM_OK) return rc;
    vm_fastx* x = *out;
    if (x->f) {
        if (begin > 0) { vm_fastx_close(x); *out = nullptr; set_error("byte ranges need uncompressed FASTA / FASTQ input"); return VM_ERR_UNSUPPORTED; }
        return VM_OK;                                            // (a compressed file is one range: `end` is ignored)
    }
    try {
        // FASTA or FASTQ: the first byte that is not white space, as vm_fastx_read decides it record by record (blank lines are skipped there:
        // a file that begins with one must not be taken for FASTA by its byte 0 — every range with begin > 0 would then look for '>' lines only)
        {
            unsigned char head[4096]; off_t at = 0; unsigned char first = 0;
            while (!first) {
                const long got = (long)pread(x->fd, head, sizeof head, at);
                if (got <= 0) break;
                for (long i = 0; i < got; ++i) if (head[i] != '\n' && head[i] != '\r' && head[i] != ' ' && head[i] != '\t') { first = head[i]; break; }
                at += got;
            }
            if (first && first != '>' && first != '@') { vm_fastx_close(x); *out = nullptr; set_error("not FASTA/FASTQ: the first record starts with neither '>' nor '@'"); return VM_ERR_IO; }
            x->fastq = first == '@';
        }
        x->range_end = end;
        if (begin <= 0) return VM_OK;
        // start one byte early: whether `begin` is itself a line start depends on the byte before it
        if (lseek(x->fd, (off_t)(begin - 1), SEEK_SET) < 0) { vm_fastx_close(x); *out = nullptr; set_error("seek failed"); return VM_ERR_IO; }
        x->base = begin - 1;
        const char* p; size_t len;
        if (!fx_line(x, p, len)) return VM_OK;                   // (the rest of the line `begin - 1` lies in: never a record start of this range; empty range at EOF)
        while (true) {
            if (x->range_end >= 0 && x->base + (int64_t)x->pos >= x->range_end) { x->eof = true; x->len = x->pos; return VM_OK; }     // no record starts in the range
            if (!fx_line(x, p, len, true)) return VM_OK;
            if (len && p[0] == '>' && !x->fastq) return VM_OK;
            if (len && p[0] == '@' && x->fastq) {
                const char* q; size_t l2;
                if (fx_peek_ahead(x, 2, q, l2) && l2 && q[0] == '+') return VM_OK;
            }
            fx_line(x, p, len);                                  // not a record start: next line
        }
    }
    catch (const std::bad_alloc&) { vm_fastx_close(x); *out = nullptr; set_error("vm_fastx_open_range: out of host memory"); return VM_ERR_OOM; }
}
void vm_fastx_close(vm_fastx* x) { if (!x) return; if (x->f) gzclose(x->f); else if (x->fd >= 0) close(x->fd); free(x->buf); delete x; }

// up to max_reads records (and at most max_bases bases) appended as blobs: names, upper-cased sequences, qualities (empty for FASTA),
// comments (the header text after the first blank or tab). Returns the number of records read (0 at end of input) or a negative status.
// The blobs are library-allocated (vm_free); offsets have n + 1 entries each.
int64_t vm_fastx_read(vm_fastx* x, int64_t max_reads, int64_t max_bases, char** names, int64_t** name_off, char** seqs, int64_t** seq_off, char** quals, int64_t** qual_off,
                      char** comments, int64_t** com_off) {
    try {
        Blob nb, sb, qb, cb; std::vector<int64_t> no(1, 0), so(1, 0), qo(1, 0), co(1, 0);
        if (x->hint) { sb.need(x->hint); qb.need(x->hint); }          // the previous call's size: one allocation instead of a growth series
        int64_t n = 0; const char* p; size_t len;
        while (n < max_reads && (int64_t)sb.n < max_bases) {
            if (x->range_end >= 0 && !x->f) {                     // byte-range reader: the record that starts at or beyond the end is the next range's
                if (!fx_line(x, p, len, true)) break;
                if (x->base + (int64_t)x->pos >= x->range_end) break;
            }
            if (!fx_line(x, p, len)) break;
            if (len == 0) continue;
            if (p[0] == ' ' || p[0] == '\t' || p[0] == '\r') {            // a line of white space only counts as blank (vm_fastx_open_range finds the format the same way)
                size_t a = 0; while (a < len && (p[a] == ' ' || p[a] == '\t' || p[a] == '\r')) ++a;
                if (a == len) continue;
            }
            if (p[0] != '>' && p[0] != '@') { set_error("not FASTA/FASTQ: " + std::string(p, len < 40 ? len : 40)); return VM_ERR_IO; }
            const bool fq = p[0] == '@';
            {   // name | comment: at the first blank or tab, whichever comes first (kseq's rule, which mp.fastx_read follows, vacmap:445)
                const char* sp = nullptr;
                for (size_t i = 1; i < len; ++i) if (p[i] == ' ' || p[i] == '\t') { sp = p + i; break; }
                if (sp) { nb.append(p + 1, (size_t)(sp - p - 1)); cb.append(sp + 1, (size_t)(p + len - sp - 1)); } else nb.append(p + 1, len - 1);
            }
            if (fq) {
                if (!fx_line(x, p, len)) { set_error("truncated FASTQ record"); return VM_ERR_IO; }
                fx_append_upper(sb, p, len);
                if (!fx_line(x, p, len) || !fx_line(x, p, len)) { set_error("truncated FASTQ record"); return VM_ERR_IO; }
                qb.append(p, len);
            } else {
                while (fx_line(x, p, len, true)) {
                    if (len && p[0] == '>') break;
                    fx_line(x, p, len);
                    size_t a = 0, b = len; while (a < b && (p[a] == ' ' || p[a] == '\t')) ++a; while (b > a && (p[b - 1] == ' ' || p[b - 1] == '\t')) --b;
                    fx_append_upper(sb, p + a, b - a);
                }
            }
            no.push_back((int64_t)nb.n); so.push_back((int64_t)sb.n); qo.push_back((int64_t)qb.n); co.push_back((int64_t)cb.n);
            ++n;
        }
        if (x->io_error) { set_error("read error or truncated / corrupt compressed input"); return VM_ERR_IO; }
        if (sb.n > x->hint) x->hint = sb.n + sb.n / 16;
        auto giveo = [](const std::vector<int64_t>& v, int64_t** p) { *p = (int64_t*)malloc(8 * v.size()); memcpy(*p, v.data(), 8 * v.size()); };
        *names = nb.release(); *seqs = sb.release(); *quals = qb.release(); *comments = cb.release();
        giveo(no, name_off); giveo(so, seq_off); giveo(qo, qual_off); giveo(co, com_off);
        return n;
    }
    catch (const std::bad_alloc&) { set_error("vm_fastx_read: out of host memory"); return VM_ERR_OOM; }
}

}  // extern "C"
