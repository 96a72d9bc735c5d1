// k_ed_band.hip — fast path of the divergence filter (E2, /root/reference/src/vacmap/mammap_clrnano.py:19247-19254).
//
// The filter only asks whether editDistance / min(len) > maxdivergence. A Myers/Hyyro bit-vector DP restricted to a band around
// the main diagonal, with every cell outside the band replaced by an UPPER bound (a block entering the band starts with all
// vertical deltas +1, the top block of the band takes horizontal delta +1), returns a value ub >= the true distance that
// equals it whenever the optimal path stays inside the band. If ub / min(len) <= maxdivergence the segment is kept and the
// decision is exact (true <= ub); otherwise the problem is flagged and recomputed by the unbanded kernel of k_ed.hip.
// For a colinear chain segment the path drifts only a few hundred rows from the diagonal, so ~all problems finish here with
// ~n wave steps instead of ceil(m/4096) * n.
//
// Layout: band rows [j + min(0,d) - HW, j + max(0,d) + HW] for column j (d = m - n, |d| <= VMX_EDB_MAXD), i.e. at most 34 blocks
// of 64 rows: block b lives on lane b & 63, a step is one anti-diagonal (block b works on column t - b). Text bases, horizontal
// deltas and running scores travel lane to lane with DPP wave_ror:1; the first live lane takes its base from a 64-column register
// chunk (v_readlane). The match masks of a block are built by the whole wave (one coalesced 64-byte load + five ballots) one step
// before the block enters the band. A lane tracks S = D[last row of its block][current column]; a block entering the band derives
// its S from the S and the delta handed over by the block above.
#include "vmx_device.h"
#include "vmx_kernels.h"
#include "vmx_ext_state.h"

__global__ void __launch_bounds__(64) k_ed_banded(const uint8_t* __restrict__ qcodes, const int64_t* __restrict__ q_off,
                                                  const uint8_t* __restrict__ tcodes, const int64_t* __restrict__ t_off,
                                                  const int32_t* __restrict__ order, const int32_t* __restrict__ range,
                                                  int32_t* __restrict__ counter, int64_t* __restrict__ ub_out) {
    const int lane = vmx_lane();
    const int n_prob = range[1];
    while (true) {
        int v = 0; if (lane == 0) v = atomicAdd(counter, 1);
        const int qi = vmx_bcast0(v);
        if (qi >= n_prob) break;
        const int p = vmx_uniform_i32(order[qi]);
        const uint8_t* pat = qcodes + q_off[p];
        const uint8_t* txt = tcodes + t_off[p];
        const int m = vmx_uniform_i32((int)(q_off[p + 1] - q_off[p]));
        const int n = vmx_uniform_i32((int)(t_off[p + 1] - t_off[p]));
        const int d = m - n;
        const bool trivial = m == 0 || n == 0;
        const bool eligible = !trivial && d <= VMX_EDB_MAXD && d >= -VMX_EDB_MAXD;
        if (!eligible) { if (lane == 0) ub_out[p] = trivial ? (long long)(m > n ? m : n) : -1LL; }
        const int B = (m + 63) >> 6;
        const int dmin = d < 0 ? d : 0, dmax = d > 0 ? d : 0;
        // per-lane block state
        int myb = -1, js = 0x7fffffff, je = -1, hbit = 63, rows = 64;
        unsigned long long p0 = 0, p1 = 0, p2 = 0, p3 = 0, p4 = 0, Pv = ~0ULL, Mv = 0ULL;
        int S = 0, hout_cur = 0, c_cur = 4;
        // wave-uniform trackers
        int bnext = 0;                // next block to enter the band
        int act_next = 0;             // step at which it starts (js + b)
        int btop = 0;                 // first live block: the lane that takes its text base from the chunk register
        int cb = 0;                   // base column of the text chunk register
        int tch = (eligible && lane < n) ? (int)txt[lane] : 4;
        const int steps = eligible ? n + B - 1 : 0;
        for (int t = 0; t < steps; ++t) {
            if (bnext < B && t >= act_next - 1) {
                // whole wave builds the masks of block bnext; its lane adopts it (the lane's previous block left the band long ago)
                const int base = bnext << 6;
                const int c = (base + lane < m) ? (int)pat[base + lane] : 255;
                const unsigned long long b0 = __ballot(c == 0), b1 = __ballot(c == 1), b2 = __ballot(c == 2), b3 = __ballot(c == 3), b4 = __ballot(c == 4);
                int lim = m - base; if (lim > 64) lim = 64;
                int njs = base - dmax - VMX_EDB_HW; if (njs < 0) njs = 0;
                int nje = base + 63 - dmin + VMX_EDB_HW; if (nje > n - 1) nje = n - 1;
                if (lane == (bnext & 63)) {
                    myb = bnext; p0 = b0; p1 = b1; p2 = b2; p3 = b3; p4 = b4; Pv = ~0ULL; Mv = 0ULL;
                    rows = lim; hbit = (lim - 1) & 63; js = njs; je = nje;
                }
                ++bnext;
                int a = (bnext << 6) - dmax - VMX_EDB_HW; if (a < 0) a = 0;
                act_next = a + bnext;
            }
            // first live block and its column; refill the text chunk when that column leaves it
            int jt = t - btop;
            while (btop < B - 1) {
                int e = (btop << 6) + 63 - dmin + VMX_EDB_HW; if (e > n - 1) e = n - 1;      // je of block btop
                if (jt > e) { ++btop; --jt; } else break;
            }
            if (jt - cb >= 64) { cb += 64; tch = (cb + lane < n) ? (int)txt[cb + lane] : 4; }
            const int c_top = (jt >= 0 && jt < n) ? vmx_readlane(tch, (jt - cb) & 63) : 4;
            const int c_in = vmx_ror1(c_cur), h_in = vmx_ror1(hout_cur), s_in = vmx_ror1(S);
            c_cur = (lane == (btop & 63)) ? c_top : c_in;
            const int j = t - myb;
            if (myb >= 0 && j >= js && j <= je) {
                int x = j + dmin - VMX_EDB_HW; x = x < 0 ? 0 : x >> 6;       // top block of the band at column j
                const int hin = (x == myb) ? 1 : h_in;
                if (j == js) S = (myb == 0) ? rows : (s_in - h_in + rows);     // block entering the band: all vertical deltas +1
                const unsigned long long s01 = (c_cur & 1) ? p1 : p0, s23 = (c_cur & 1) ? p3 : p2;
                const unsigned long long s03 = (c_cur & 2) ? s23 : s01;
                unsigned long long Eq = (c_cur & 4) ? p4 : s03;
                const unsigned long long neg = (unsigned long long)((unsigned)hin >> 31);
                const unsigned long long pos = (unsigned long long)((unsigned)(-hin) >> 31);
                const unsigned long long Xv = Eq | Mv;
                Eq |= neg;
                const unsigned long long Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
                unsigned long long Ph = Mv | ~(Xh | Pv);
                unsigned long long Mh = Pv & Xh;
                const int hout = (int)((Ph >> hbit) & 1ULL) - (int)((Mh >> hbit) & 1ULL);
                Ph = (Ph << 1) | pos; Mh = (Mh << 1) | neg;
                Pv = Mh | ~(Xv | Ph);
                Mv = Ph & Xv;
                hout_cur = hout;
                S += hout;
                if (myb == B - 1 && j == n - 1) ub_out[p] = (long long)S;
            }
        }
    }
}

// thresholds: a problem stays flagged (size = pattern length, for the unbanded kernel) unless its upper bound already proves
// editDistance / min(len) <= maxdiv; unflagged problems get size -1 and ed = ub
__global__ void k_ed_flag(const int64_t* __restrict__ ub, const int64_t* __restrict__ q_off, const int64_t* __restrict__ t_off, const int32_t* __restrict__ n_ptr,
                          double maxdiv, int64_t* __restrict__ sizes, int64_t* __restrict__ ed_out, int32_t* __restrict__ n_flagged, int first,
                          const int32_t* __restrict__ prob_read, vmx_ext_read* __restrict__ er) {
    const int n = *n_ptr;
    for (int i = (int)(blockIdx.x * blockDim.x + threadIdx.x); i < n; i += (int)(gridDim.x * blockDim.x)) {
        if (!first && sizes[i] < 0) continue;                         // settled by an earlier tier
        const long long m = q_off[i + 1] - q_off[i], t = t_off[i + 1] - t_off[i], mn = m < t ? m : t;
        const long long u = ub[i];
        bool keep = false;
        if (mn == 0) keep = true;                                    // the consumer raises on an empty side; no DP needed
        else if (u >= 0 && ((double)u / (double)mn) <= maxdiv) keep = true;
        if (keep) { sizes[i] = -1; ed_out[i] = u >= 0 ? u : (m > t ? m : t); }
        else {
            sizes[i] = m; atomicAdd(n_flagged, 1);
            // last tier before the exact kernel, which this batch does not launch (round 6: hardly any problem gets here — 0 per batch on ONT / HiFi reads, one in fifteen
            // batches on the vacsim workload — and learning the count cost a host wait per batch): the problem's READ is marked and run again alone, exact tier included
            if (er) er[prob_read[i]].status = VMX_EXT_NEED_EXACT_DEV;
        }
    }
}

// ------------------------------------------------------------------------------------------------ first tier: four problems per wave
// Most segments stay within a few hundred rows of the diagonal. With a band of +-VMX_EDB4_HW rows and |m - n| <= VMX_EDB4_MAXD the band
// never spans more than 16 blocks, so one problem fits a 16-lane DPP row and a wave carries four problems: block b lives on row lane
// b & 15, hand-offs are row_ror:1 moves. Everything that was wave-uniform in k_ed_banded (top block, next block to load, text chunk)
// is row-uniform here and kept per lane; the text chunk is 16 columns per row. A problem that is not eligible, or whose bound does
// not prove "keep", is passed on to k_ed_banded (wider band) and from there to the exact kernel.
#ifdef VMX_EMU
__device__ __forceinline__ int vmx_row_ror1(int v) { const int l = vmx_lane(); return __shfl(v, (l & 48) | ((l + 15) & 15)); }
#else
__device__ __forceinline__ int vmx_row_ror1(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x121, 0xf, 0xf, false); }   // row_ror:1
#endif

__global__ void __launch_bounds__(64) k_ed_banded4(const uint8_t* __restrict__ qcodes, const int64_t* __restrict__ q_off,
                                                   const uint8_t* __restrict__ tcodes, const int64_t* __restrict__ t_off,
                                                   const int32_t* __restrict__ order, const int32_t* __restrict__ range,
                                                   int32_t* __restrict__ counter, int64_t* __restrict__ ub_out) {
    const int lane = vmx_lane();
    const int row = lane >> 4, sub = lane & 15;
    const int n_prob = range[1];
    while (true) {
        int v = 0; if (lane == 0) v = atomicAdd(counter, 4);
        const int qb = vmx_bcast0(v);
        if (qb >= n_prob) break;
        const int qi = qb + row;
        const bool has = qi < n_prob;
        const int p = has ? order[qi] : 0;
        const uint8_t* pat = qcodes + q_off[p];
        const uint8_t* txt = tcodes + t_off[p];
        const int m = has ? (int)(q_off[p + 1] - q_off[p]) : 0;
        const int n = has ? (int)(t_off[p + 1] - t_off[p]) : 0;
        const int d = m - n;
        const bool trivial = m == 0 || n == 0;
        const bool eligible = has && !trivial && d <= VMX_EDB4_MAXD && d >= -VMX_EDB4_MAXD;
        if (has && !eligible && sub == 0) ub_out[p] = trivial ? (long long)(m > n ? m : n) : -1LL;
        const int B = (m + 63) >> 6;
        const int dmin = d < 0 ? d : 0, dmax = d > 0 ? d : 0;
        int myb = -1, js = 0x7fffffff, je = -1, hbit = 63, rows = 64;
        unsigned long long p0 = 0, p1 = 0, p2 = 0, p3 = 0, p4 = 0, Pv = ~0ULL, Mv = 0ULL;
        int S = 0, hout_cur = 0, c_cur = 4;
        int bnext = 0, act_next = 0, btop = 0, cb = 0;          // row-uniform trackers
        int tch = (eligible && sub < n) ? (int)txt[sub] : 4;    // 16-column text chunk of the row
        const int steps_r = eligible ? n + B - 1 : 0;
        int steps = vmx_readlane(steps_r, 0);
        { const int s1 = vmx_readlane(steps_r, 16), s2 = vmx_readlane(steps_r, 32), s3 = vmx_readlane(steps_r, 48);
          steps = s1 > steps ? s1 : steps; steps = s2 > steps ? s2 : steps; steps = s3 > steps ? s3 : steps; }
        for (int t = 0; t < steps; ++t) {
            // a row whose next block is about to enter its band: the whole wave builds the block's match masks
            const int need = (t < steps_r && bnext < B && t >= act_next - 1) ? 1 : 0;
            for (int r = 0; r < 4; ++r) {
                if (!vmx_readlane(need, 16 * r)) continue;
                const int bn = vmx_readlane(bnext, 16 * r), mr = vmx_readlane(m, 16 * r);
                union { const uint8_t* ptr; int i[2]; } u; u.ptr = pat;
                u.i[0] = vmx_readlane(u.i[0], 16 * r); u.i[1] = vmx_readlane(u.i[1], 16 * r);
                const int base = bn << 6;
                const int c = (base + lane < mr) ? (int)u.ptr[base + lane] : 255;
                const unsigned long long b0 = __ballot(c == 0), b1 = __ballot(c == 1), b2 = __ballot(c == 2), b3 = __ballot(c == 3), b4 = __ballot(c == 4);
                if (row == r) {
                    if (sub == (bn & 15)) {
                        int lim = m - base; if (lim > 64) lim = 64;
                        int njs = base - dmax - VMX_EDB4_HW; if (njs < 0) njs = 0;
                        int nje = base + 63 - dmin + VMX_EDB4_HW; if (nje > n - 1) nje = n - 1;
                        myb = bn; p0 = b0; p1 = b1; p2 = b2; p3 = b3; p4 = b4; Pv = ~0ULL; Mv = 0ULL;
                        rows = lim; hbit = (lim - 1) & 63; js = njs; je = nje;
                    }
                    ++bnext;
                    int a = (bnext << 6) - dmax - VMX_EDB4_HW; if (a < 0) a = 0;
                    act_next = a + bnext;
                }
            }
            // first live block of the row and its column; refill the row's text chunk when that column leaves it
            int jt = t - btop;
            if (btop < B - 1) {
                int e = (btop << 6) + 63 - dmin + VMX_EDB4_HW; if (e > n - 1) e = n - 1;
                if (jt > e) { ++btop; --jt; }
            }
            if (eligible && jt - cb >= 16) { cb += 16; tch = (cb + sub < n) ? (int)txt[cb + sub] : 4; }
            const int c_sh = __shfl(tch, (lane & 48) | ((jt - cb) & 15));
            const int c_top = (jt >= 0 && jt < n) ? c_sh : 4;
            const int c_in = vmx_row_ror1(c_cur), h_in = vmx_row_ror1(hout_cur), s_in = vmx_row_ror1(S);
            c_cur = (sub == (btop & 15)) ? c_top : c_in;
            const int j = t - myb;
            if (eligible && myb >= 0 && j >= js && j <= je) {
                int x = j + dmin - VMX_EDB4_HW; x = x < 0 ? 0 : x >> 6;
                const int hin = (x == myb) ? 1 : h_in;
                if (j == js) S = (myb == 0) ? rows : (s_in - h_in + rows);
                const unsigned long long s01 = (c_cur & 1) ? p1 : p0, s23 = (c_cur & 1) ? p3 : p2;
                const unsigned long long s03 = (c_cur & 2) ? s23 : s01;
                unsigned long long Eq = (c_cur & 4) ? p4 : s03;
                const unsigned long long neg = (unsigned long long)((unsigned)hin >> 31);
                const unsigned long long pos = (unsigned long long)((unsigned)(-hin) >> 31);
                const unsigned long long Xv = Eq | Mv;
                Eq |= neg;
                const unsigned long long Xh = (((Eq & Pv) + Pv) ^ Pv) | Eq;
                unsigned long long Ph = Mv | ~(Xh | Pv);
                unsigned long long Mh = Pv & Xh;
                const int hout = (int)((Ph >> hbit) & 1ULL) - (int)((Mh >> hbit) & 1ULL);
                Ph = (Ph << 1) | pos; Mh = (Mh << 1) | neg;
                Pv = Mh | ~(Xv | Ph);
                Mv = Ph & Xv;
                hout_cur = hout;
                S += hout;
                if (myb == B - 1 && j == n - 1) ub_out[p] = (long long)S;
            }
        }
    }
}
