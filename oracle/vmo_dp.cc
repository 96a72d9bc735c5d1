// vmo_dp.cc — CPU ORACLE (test infrastructure): DP primitives.
//
// Replaces `vacmap_index.k_cigar` (un-vendored C extension vacmap-index==0.0.3; live call sites
// /root/reference/src/vacmap/mammap_clrnano.py:21554,21598 (global gap-fill) and :2381,2410,2477,2505
// (banded z-drop edge extension)) and `edlib.align(task='distance')` (edlib==1.3.9; call site :19251).
// Neither source is under /root/reference. The alignment SCORE of the global DP and the edit distance are
// mathematically unique; the CIGAR among co-optimal alignments and the extension's drop rule are fixed by
// the build's own spec "VMX-DP" below. PARITY UNPINNED for CIGAR tie-breaks.
//
// VMX-DP-G (global, dual affine, traceback):
//   T = target (consumed by M/=/X and D), Q = query (consumed by M/=/X and I). Codes A0 C1 G2 T3 other 4.
//   s(a,b) = match if a == b && a < 4 else mismatch.  gap(L) = min_k (o_k + L*e_k), k = 1,2.
//   E_k(i,j) = max(H(i-1,j) - o_k, E_k(i-1,j)) - e_k          (deletion: target base i-1 vs gap)
//   F_k(i,j) = max(H(i,j-1) - o_k, F_k(i,j-1)) - e_k          (insertion: query base j-1 vs gap)
//   H(i,j)   = max(H(i-1,j-1) + s, E_1, E_2, F_1, F_2);  H(0,0) = 0; out-of-matrix = -inf.
//   Traceback from (tl,ql) in state H. In H: first of {diag, E_1, E_2, F_1, F_2} whose value equals H(i,j).
//   In E_k at (i,j): emit D, stay in E_k iff E_k(i-1,j) > H(i-1,j) - o_k (strict) else go to H; i -= 1.
//   In F_k likewise on j. Ops are reversed and run-length encoded ('M', or '='/'X' when eqx).
// VMX-DP-X (extension, single affine (o,e), band |i-j| <= bw, anti-diagonal x-drop):
//   same recurrences (one gap piece) on in-band cells, anchored at (0,0), processed by anti-diagonal
//   d = i + j = 1,2,...  Best M = 0 at (0,0); a cell replaces the best iff H > M strictly, cells visited
//   in (d ascending, i ascending) order. After diagonal d: stop if max(m_d, m_{d-1}) < M - zdrop, where m_d is
//   the max H on diagonal d (m_0 = 0). Returns (t_e, q_e) = (i, j) of the best cell.
// VMX-ED: global unit-cost Levenshtein distance over the 5-letter code alphabet.
#include "vmo_internal.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace vmo;

namespace vmo {

static const int32_t NEG = -(1 << 28);

int k_cigar_global(const char* t, int64_t tl, const char* q, int64_t ql, int match, int mismatch, int o1, int e1,
                   int o2, int e2, int eqx, std::string& cigar, int32_t* score_out) {
    cigar.clear();
    if (tl < 0 || ql < 0) return -1;
    if (tl == 0 && ql == 0) { if (score_out) *score_out = 0; return 0; }
    const int64_t W = ql + 1;
    // traceback byte: bits 0-2 source of H (0 diag,1 E1,2 E2,3 F1,4 F2); bit3 extE1; bit4 extE2; bit5 extF1; bit6 extF2
    std::vector<uint8_t> tb((size_t)(tl + 1) * (size_t)W, 0);
    std::vector<int32_t> Hp(W), Hc(W), E1(W, NEG), E2(W, NEG);
    std::vector<uint8_t> tc(tl), qc(ql);
    for (int64_t i = 0; i < tl; ++i) tc[i] = NT4[(uint8_t)t[i]];
    for (int64_t j = 0; j < ql; ++j) qc[j] = NT4[(uint8_t)q[j]];
    // row 0
    Hp[0] = 0;
    {
        int32_t F1 = NEG, F2 = NEG;
        for (int64_t j = 1; j <= ql; ++j) {
            uint8_t b = 0;
            int32_t a1 = Hp[j - 1] - o1, a2 = Hp[j - 1] - o2;
            if (F1 > a1) b |= 1 << 5; if (F2 > a2) b |= 1 << 6;
            F1 = std::max(a1, F1) - e1; F2 = std::max(a2, F2) - e2;
            int32_t h; uint8_t src;
            if (F1 >= F2) { h = F1; src = 3; } else { h = F2; src = 4; }
            Hp[j] = h; tb[j] = b | src;
        }
    }
    for (int64_t i = 1; i <= tl; ++i) {
        uint8_t* tr = &tb[(size_t)i * W];
        int32_t F1 = NEG, F2 = NEG;
        // column 0
        {
            uint8_t b = 0;
            int32_t a1 = Hp[0] - o1, a2 = Hp[0] - o2;
            if (E1[0] > a1) b |= 1 << 3; if (E2[0] > a2) b |= 1 << 4;
            E1[0] = std::max(a1, E1[0]) - e1; E2[0] = std::max(a2, E2[0]) - e2;
            int32_t h; uint8_t src;
            if (E1[0] >= E2[0]) { h = E1[0]; src = 1; } else { h = E2[0]; src = 2; }
            Hc[0] = h; tr[0] = b | src;
        }
        const uint8_t ti = tc[i - 1];
        for (int64_t j = 1; j <= ql; ++j) {
            uint8_t b = 0;
            int32_t a1 = Hp[j] - o1, a2 = Hp[j] - o2;
            if (E1[j] > a1) b |= 1 << 3; if (E2[j] > a2) b |= 1 << 4;
            int32_t e1v = std::max(a1, E1[j]) - e1, e2v = std::max(a2, E2[j]) - e2;
            E1[j] = e1v; E2[j] = e2v;
            int32_t c1 = Hc[j - 1] - o1, c2 = Hc[j - 1] - o2;
            if (F1 > c1) b |= 1 << 5; if (F2 > c2) b |= 1 << 6;
            F1 = std::max(c1, F1) - e1; F2 = std::max(c2, F2) - e2;
            int32_t d = Hp[j - 1] + ((ti == qc[j - 1] && ti < 4) ? match : mismatch);
            int32_t h = d; uint8_t src = 0;
            if (e1v > h) { h = e1v; src = 1; }
            if (F1 > h) { h = F1; src = 3; }
            if (e2v > h) { h = e2v; src = 2; }
            if (F2 > h) { h = F2; src = 4; }
            Hc[j] = h; tr[j] = b | src;
        }
        std::swap(Hp, Hc);
    }
    if (score_out) *score_out = Hp[ql];
    // traceback
    std::string ops;
    int64_t i = tl, j = ql; int state = 0;
    while (i > 0 || j > 0) {
        uint8_t b = tb[(size_t)i * W + j];
        if (state == 0) {
            int src = b & 7;
            if (src == 0) { ops.push_back(eqx ? ((tc[i - 1] == qc[j - 1] && tc[i - 1] < 4) ? '=' : 'X') : 'M'); --i; --j; }
            else state = src;
        } else if (state == 1 || state == 2) {
            ops.push_back('D');
            bool ext = state == 1 ? (b >> 3) & 1 : (b >> 4) & 1;
            --i; if (!ext) state = 0;
        } else {
            ops.push_back('I');
            bool ext = state == 3 ? (b >> 5) & 1 : (b >> 6) & 1;
            --j; if (!ext) state = 0;
        }
    }
    std::reverse(ops.begin(), ops.end());
    for (size_t a = 0; a < ops.size();) {
        size_t b2 = a; while (b2 < ops.size() && ops[b2] == ops[a]) ++b2;
        cigar += std::to_string(b2 - a); cigar.push_back(ops[a]); a = b2;
    }
    return 0;
}

int k_extend(const char* t, int64_t tl, const char* q, int64_t ql, int match, int mismatch, int o, int e, int bw,
             int zdrop, int32_t* t_e, int32_t* q_e) {
    int32_t M = 0; int64_t bi = 0, bj = 0;
    if (bw < 0) bw = (int)std::max(tl, ql);
    // anti-diagonal buffers indexed by i: H of d-1 and d-2, E/F of d-1, plus the diagonal being written.
    // Every diagonal writes -inf sentinels at ilo-1 and ihi+1; ranges move by <= 1 per diagonal, so reads of
    // d-1 at [ilo-1, ihi] and of d-2 at [ilo-1, ihi-1] never see stale cells.
    const int64_t n = tl + 1;
    std::vector<int32_t> H1(n, NEG), H2(n, NEG), Hc(n, NEG), E1v(n, NEG), Ec(n, NEG), F1v(n, NEG), Fc(n, NEG);
    H1[0] = 0;  // diagonal 0
    int32_t m_prev = 0;
    for (int64_t d = 1; d <= tl + ql; ++d) {
        int64_t ilo = std::max<int64_t>(0, d - ql), ihi = std::min<int64_t>(tl, d);
        // band |i - j| <= bw with j = d - i  <=>  (d - bw)/2 <= i <= (d + bw)/2
        if (d - bw > 0) ilo = std::max<int64_t>(ilo, (d - bw + 1) / 2);
        ihi = std::min<int64_t>(ihi, (d + bw) / 2);
        if (ilo > ihi) break;
        int32_t m_d = NEG;
        for (int64_t i = ilo; i <= ihi; ++i) {
            int64_t j = d - i;
            int32_t ev = NEG, fv = NEG, dv = NEG;
            if (i >= 1) { int32_t hu = H1[i - 1], eu = E1v[i - 1]; if (hu > NEG || eu > NEG) ev = std::max(hu - o, eu) - e; }
            if (j >= 1) { int32_t hl = H1[i], fl = F1v[i]; if (hl > NEG || fl > NEG) fv = std::max(hl - o, fl) - e; }
            if (i >= 1 && j >= 1 && H2[i - 1] > NEG) {
                uint8_t a = NT4[(uint8_t)t[i - 1]], b = NT4[(uint8_t)q[j - 1]];
                dv = H2[i - 1] + ((a == b && a < 4) ? match : mismatch);
            }
            if (ev < NEG) ev = NEG;
            if (fv < NEG) fv = NEG;
            int32_t h = std::max(dv, std::max(ev, fv));
            Hc[i] = h; Ec[i] = ev; Fc[i] = fv;
            if (h > m_d) m_d = h;
            if (h > M) { M = h; bi = i; bj = j; }
        }
        if (ilo - 1 >= 0) { Hc[ilo - 1] = NEG; Ec[ilo - 1] = NEG; Fc[ilo - 1] = NEG; }
        if (ihi + 1 < n) { Hc[ihi + 1] = NEG; Ec[ihi + 1] = NEG; Fc[ihi + 1] = NEG; }
        std::swap(H2, H1); std::swap(H1, Hc); std::swap(E1v, Ec); std::swap(F1v, Fc);
        if (std::max(m_d, m_prev) < M - zdrop) break;
        m_prev = m_d;
    }
    *t_e = (int32_t)bi; *q_e = (int32_t)bj;
    return M;
}

// Myers / Hyyro block bit-vector, global distance.
int64_t edit_distance(const uint8_t* p, int64_t m, const uint8_t* t, int64_t n) {
    if (m == 0) return n;
    if (n == 0) return m;
    const int64_t B = (m + 63) / 64;
    std::vector<uint64_t> Peq((size_t)B * 5, 0), Pv((size_t)B, ~0ULL), Mv((size_t)B, 0);
    for (int64_t i = 0; i < m; ++i) Peq[(size_t)(i / 64) * 5 + p[i]] |= 1ULL << (i % 64);
    const uint64_t lastbit = 1ULL << ((m - 1) % 64);
    int64_t score = m;
    for (int64_t j = 0; j < n; ++j) {
        const uint8_t c = t[j];
        int hin = 1;
        for (int64_t b = 0; b < B; ++b) {
            uint64_t Eq = Peq[(size_t)b * 5 + c];
            uint64_t pv = Pv[b], mv = Mv[b];
            const uint64_t HIGH = (b == B - 1) ? lastbit : (1ULL << 63);
            uint64_t Xv = Eq | mv;
            if (hin < 0) Eq |= 1ULL;
            uint64_t Xh = (((Eq & pv) + pv) ^ pv) | Eq;
            uint64_t Ph = mv | ~(Xh | pv);
            uint64_t Mh = pv & Xh;
            int hout = 0;
            if (Ph & HIGH) hout = 1;
            if (Mh & HIGH) hout = -1;
            Ph <<= 1; Mh <<= 1;
            if (hin < 0) Mh |= 1ULL; else if (hin > 0) Ph |= 1ULL;
            Pv[b] = Mh | ~(Xv | Ph);
            Mv[b] = Ph & Xv;
            hin = hout;
        }
        score += hin;
    }
    return score;
}

int64_t edit_distance_str(const std::string& a, const std::string& b) {
    std::vector<uint8_t> x(a.size()), y(b.size());
    for (size_t i = 0; i < a.size(); ++i) x[i] = NT4[(uint8_t)a[i]];
    for (size_t i = 0; i < b.size(); ++i) y[i] = NT4[(uint8_t)b[i]];
    return edit_distance(x.data(), (int64_t)x.size(), y.data(), (int64_t)y.size());
}

}  // namespace vmo

extern "C" {

int vmo_k_cigar_global(const char* t, int64_t tl, const char* q, int64_t ql, int match, int mismatch, int o1, int e1,
                       int o2, int e2, int eqx, char** cigar_out, int32_t* score) {
    std::string c;
    int rc = k_cigar_global(t, tl, q, ql, match, mismatch, o1, e1, o2, e2, eqx, c, score);
    if (cigar_out) { *cigar_out = (char*)malloc(c.size() + 1); memcpy(*cigar_out, c.c_str(), c.size() + 1); }
    return rc;
}

int vmo_k_extend(const char* t, int64_t tl, const char* q, int64_t ql, int match, int mismatch, int o, int e, int bw,
                 int zdrop, int32_t* t_e, int32_t* q_e) {
    return k_extend(t, tl, q, ql, match, mismatch, o, e, bw, zdrop, t_e, q_e);
}

int64_t vmo_edit_distance(const char* q, int64_t ql, const char* t, int64_t tl) {
    std::vector<uint8_t> x(ql), y(tl);
    for (int64_t i = 0; i < ql; ++i) x[i] = NT4[(uint8_t)q[i]];
    for (int64_t i = 0; i < tl; ++i) y[i] = NT4[(uint8_t)t[i]];
    return edit_distance(x.data(), ql, y.data(), tl);
}

}  // extern "C"
