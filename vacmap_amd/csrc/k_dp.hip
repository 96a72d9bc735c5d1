// k_dp.hip — hand-written gfx950 kernels for the banded edge extension and the gap-fill DP of the extend stage
// (the divergence filter E2 lives in k_ed.hip).
//
//   k_extend               E3  mp.k_cigar(.., 4,4,4,4, bw=100, zdropvalue=50)   (/root/reference/src/vacmap/mammap_clrnano.py:2381, :2410, :2477, :2505)
//   k_gapfill_fill/_trace  E5  mp.k_cigar(.., 2,-4, 4,2, 24,1, bw=-1, zdropvalue=-1, eqx)   (:21554, :21598)
//
// The native library behind those calls (vacmap-index==0.0.3) is not in /root/reference; the kernels implement the build's
// normative spec VMX-DP and are parity-tested bit-for-bit against oracle/vmo_dp.cc. Integer, wave64,
// anti-diagonal kernels ("one lane = one row of a 64-row stripe"): no MFMA (nothing here is a contraction); one wavefront per
// problem, problems taken longest-first from a device-side queue (k_size_order in k_ed.hip).
#include "vmx_device.h"
#include "vmx_kernels.h"

// target code as the packed forms compare it: an ambiguous target base (code 4) becomes 6, which equals no query code — an N never
// matches, not even another N (VMX-DP-G; the int32 form tests `ti == qc && ti < 4`). Costs nothing per DP step: the row's code is loaded
// once per stripe.
__device__ __forceinline__ int vmx_tcode(uint8_t c) { return c < 4 ? (int)c : 6; }


// ------------------------------------------------------------------------------------------------ encode
__global__ void k_encode(const char* __restrict__ in, uint8_t* __restrict__ out, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = vmx_code((uint8_t)in[i]);
}

// ------------------------------------------------------------------------------------------------ E3 extension
// VMX-DP-X: single affine (o,e), band |i-j| <= bw, anchored at (0,0), anti-diagonal x-drop. Ring buffers in LDS indexed by
// i & (RING-1): H of d-1, d-2 and the diagonal being written, E/F of d-1 and current.
#define VMX_EXT_RING 512
__global__ void __launch_bounds__(64) k_extend(const uint8_t* __restrict__ tcodes, const int64_t* __restrict__ t_off,
                                               const uint8_t* __restrict__ qcodes, const int64_t* __restrict__ q_off, int n_prob,
                                               int match, int mismatch, int o, int e, int bw_in, int zdrop,
                                               int32_t* __restrict__ out_te, int32_t* __restrict__ out_qe, int32_t* __restrict__ out_sc, const int32_t* __restrict__ n_ptr) {
    if (n_ptr) n_prob = *n_ptr;                  // the batched path keeps the round's problem count on the device
    __shared__ int sH[3][VMX_EXT_RING];
    __shared__ int sE[2][VMX_EXT_RING];
    __shared__ int sF[2][VMX_EXT_RING];
    const int lane = vmx_lane();
    const int RM = VMX_EXT_RING - 1;
    for (int p = blockIdx.x; p < n_prob; p += gridDim.x) {
        const uint8_t* T = tcodes + t_off[p];
        const uint8_t* Q = qcodes + q_off[p];
        const int tl = (int)(t_off[p + 1] - t_off[p]);
        const int ql = (int)(q_off[p + 1] - q_off[p]);
        int bw = bw_in; if (bw < 0 || bw > VMX_EXT_RING - 16) bw = VMX_EXT_RING - 16;   // host rejects wider bands
        for (int x = lane; x < VMX_EXT_RING; x += 64) {
            sH[0][x] = VMX_NEG; sH[1][x] = VMX_NEG; sH[2][x] = VMX_NEG;
            sE[0][x] = VMX_NEG; sE[1][x] = VMX_NEG; sF[0][x] = VMX_NEG; sF[1][x] = VMX_NEG;
        }
        __syncthreads();
        if (lane == 0) sH[0][0] = 0;    // diagonal 0
        __syncthreads();
        int M = 0, bi = 0, bj = 0, m_prev = 0;
        for (int d = 1; d <= tl + ql; ++d) {
            int ilo = d - ql; if (ilo < 0) ilo = 0;
            int ihi = d < tl ? d : tl;
            if (d - bw > 0) { int b2 = (d - bw + 1) >> 1; if (b2 > ilo) ilo = b2; }
            { int b3 = (d + bw) >> 1; if (b3 < ihi) ihi = b3; }
            if (ilo > ihi) break;
            int* Hc = sH[d % 3]; const int* H1 = sH[(d + 2) % 3]; const int* H2 = sH[(d + 1) % 3];
            int* Ec = sE[d & 1]; const int* E1 = sE[(d + 1) & 1];
            int* Fc = sF[d & 1]; const int* F1 = sF[(d + 1) & 1];
            int best_h = VMX_NEG, best_i = 0x7fffffff;
            for (int i0 = ilo; i0 <= ihi; i0 += 64) {
                int i = i0 + lane;
                if (i <= ihi) {
                    int j = d - i;
                    int ev = VMX_NEG, fv = VMX_NEG, dv = VMX_NEG;
                    if (i >= 1) { int hu = H1[(i - 1) & RM], eu = E1[(i - 1) & RM]; if (hu > VMX_NEG || eu > VMX_NEG) { int a = hu - o; ev = (a > eu ? a : eu) - e; } }
                    if (j >= 1) { int hl = H1[i & RM], fl = F1[i & RM]; if (hl > VMX_NEG || fl > VMX_NEG) { int a = hl - o; fv = (a > fl ? a : fl) - e; } }
                    if (i >= 1 && j >= 1) {
                        int h2 = H2[(i - 1) & RM];
                        if (h2 > VMX_NEG) { uint8_t a = T[i - 1], b = Q[j - 1]; dv = h2 + ((a == b && a < 4) ? match : mismatch); }
                    }
                    if (ev < VMX_NEG) ev = VMX_NEG;
                    if (fv < VMX_NEG) fv = VMX_NEG;
                    int h = dv > ev ? dv : ev; h = h > fv ? h : fv;
                    Hc[i & RM] = h; Ec[i & RM] = ev; Fc[i & RM] = fv;
                    if (h > best_h) { best_h = h; best_i = i; }
                }
            }
            if (lane == 0) {
                if (ilo - 1 >= 0) { Hc[(ilo - 1) & RM] = VMX_NEG; Ec[(ilo - 1) & RM] = VMX_NEG; Fc[(ilo - 1) & RM] = VMX_NEG; }
                if (ihi + 1 <= tl) { Hc[(ihi + 1) & RM] = VMX_NEG; Ec[(ihi + 1) & RM] = VMX_NEG; Fc[(ihi + 1) & RM] = VMX_NEG; }
            }
            // wave max of (h, then smallest i)
            for (int off = 32; off > 0; off >>= 1) {
                int oh = __shfl_xor(best_h, off), oi = __shfl_xor(best_i, off);
                if (oh > best_h || (oh == best_h && oi < best_i)) { best_h = oh; best_i = oi; }
            }
            const int m_d = best_h;
            if (m_d > M) { M = m_d; bi = best_i; bj = d - best_i; }
            __syncthreads();
            if ((m_d > m_prev ? m_d : m_prev) < M - zdrop) break;
            m_prev = m_d;
        }
        if (lane == 0) { out_te[p] = bi; out_qe[p] = bj; out_sc[p] = M; }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ E5 gap fill
// VMX-DP-G: global dual-affine DP with a 7-bit traceback byte per cell (bits 0-2 source of H: 0 diag,1 E1,2 E2,3 F1,4 F2;
// bit3 extE1, bit4 extE2, bit5 extF1, bit6 extF2). lane = target row inside a 64-row stripe, step = anti-diagonal of the
// stripe. Per step a lane needs H/E1/E2 of the cell above (the lane above, one step earlier: three DPP wave_shr:1 moves) and the
// query base of its column (handed down the lanes the same way); lane 0 takes the row above the stripe from bnd[] (H, E1, E2 of
// the previous stripe's last row) and its query base from 64-column chunk registers that rotate one lane per step (wave_rol:1), so
// the value lane 0 needs rides into the wave_shr:1 move as its old operand; chunks are refilled with one coalesced load per 64
// steps and lane 63's outputs leave through a rotating register FIFO flushed once per 64 columns — nothing on the per-step path
// waits for memory. Chunks in which every lane is inside its row run without the per-step activity test.
// Traceback bytes are stored as tb[(stripe*(ql+63) + step)*64 + lane]: one coalesced 64-byte line per step.
// (The batched path runs its small problems — nearly all of them — in the anti-diagonal band form of vmx_dp_ad.h first.)
__device__ __forceinline__ int vmx_gap_open_row(int i, int o1, int e1, int o2, int e2) {   // H(i,0) = H(0,i), i >= 1
    int a = -(o1 + i * e1), b = -(o2 + i * e2);
    return a > b ? a : b;
}

// Packed form for problems with tl + ql <= VMX_DP16_MAX (every score fits int16): a lane owns TWO consecutive rows of a 128-row stripe,
// row 2l+1 in the low half and row 2l+2 in the high half of each register, the high row one column behind the low one, so both cells
// of a step only need values of the step before: up(lo) = the lane above's high half, up(hi) = the own low half — one DPP move and one
// v_alignbit per quantity. The recurrence runs on v_pk_add/sub/max_i16; comparison results are sign masks (v_pk_ashrrev_i16 15) merged
// with v_bfi. Traceback bytes: tb[((stripe*(ql+127) + step)*64 + lane)*2 + half]. NEG16 stands for -infinity: a real score never gets
// within reach of it, and it is never the larger operand of a max whose result is used.
#define VMX_NEG16 (-20000)
__device__ __forceinline__ void vmx_gapfill_fill16(const uint8_t* __restrict__ T, const uint8_t* __restrict__ Q, int tl, int ql, int match, int mismatch,
                                                   int o1, int e1, int o2, int e2, uint8_t* __restrict__ tb, int32_t* __restrict__ bH,
                                                   int32_t* __restrict__ score_out, int lane) {
    int32_t* bE1 = bH + (ql + 1);
    int32_t* bE2 = bE1 + (ql + 1);
    const int W = ql + 127;
    const int nstr = (tl + 127) >> 7;
    const unsigned O1 = vmx_pk(o1, o1), O2 = vmx_pk(o2, o2), E1C = vmx_pk(e1, e1), E2C = vmx_pk(e2, e2);
    const unsigned MATCH = vmx_pk(match, match), MISM = vmx_pk(mismatch, mismatch), ONE = vmx_pk(1, 1);
    const int rf = (tl - 1) & 127;                              // row tl inside the last stripe: lane rf / 2, half rf & 1
    const int t_fin = ql - 1 + rf;                              // step at which that lane computes column ql
    unsigned fin = 0;
    for (int s = 0; s < nstr; ++s) {
        const int i0 = s * 128 + 2 * lane + 1;                  // low-half row; the high half is row i0 + 1
        const unsigned ti2 = vmx_pk(i0 <= tl ? vmx_tcode(T[i0 - 1]) : 5, i0 + 1 <= tl ? vmx_tcode(T[i0]) : 5);     // 5 never equals a query code (rows past tl)
        unsigned Hleft = vmx_pk(vmx_gap_open_row(i0, o1, e1, o2, e2), vmx_gap_open_row(i0 + 1, o1, e1, o2, e2));
        unsigned F1 = vmx_pk(VMX_NEG16, VMX_NEG16), F2 = F1;
        unsigned Hdiag = vmx_pk(i0 - 1 == 0 ? 0 : vmx_gap_open_row(i0 - 1, o1, e1, o2, e2), vmx_gap_open_row(i0, o1, e1, o2, e2));
        unsigned outH = 0, outE1 = F1, outE2 = F1, qc = vmx_pk(4, 4);
        int sH = 0, sE1 = 0, sE2 = 0;                           // lane 63's high-half outputs of the last 64 steps (register FIFO, newest in lane 0)
        const bool store_bnd = s + 1 < nstr;
        uint8_t* tbs = tb + (size_t)s * (size_t)W * 128 + 2 * lane;
        for (int t0 = 0; t0 < W; t0 += 64) {
            // 64-column chunks for lane 0's low row, pre-shifted into the high half (the hand-off shifts them down): query bases Q[t0 ..] and
            // the row above the stripe for columns t0+1 ..
            const int jj = t0 + lane;
            unsigned qchunk = (unsigned)(jj < ql ? (int)Q[jj] : 4) << 16;
            unsigned cH = 0, cE1 = (unsigned)VMX_NEG16 << 16, cE2 = cE1;
            if (jj + 1 <= ql) {
                if (s == 0) cH = (unsigned)vmx_gap_open_row(jj + 1, o1, e1, o2, e2) << 16;
                else { cH = (unsigned)bH[jj + 1] << 16; cE1 = (unsigned)bE1[jj + 1] << 16; cE2 = (unsigned)bE2[jj + 1] << 16; }
            }
            int tend = W - t0; if (tend > 64) tend = 64;
            const bool ramp_up = t0 < 128;                      // some half has not reached its column 1 yet: its state must not move
            for (int tt = 0; tt < tend; ++tt) {
                const int t = t0 + tt;
                const unsigned upH = vmx_alignbit16(outH, (unsigned)vmx_shr1_in((int)outH, (int)cH));
                const unsigned upE1 = vmx_alignbit16(outE1, (unsigned)vmx_shr1_in((int)outE1, (int)cE1));
                const unsigned upE2 = vmx_alignbit16(outE2, (unsigned)vmx_shr1_in((int)outE2, (int)cE2));
                qc = vmx_alignbit16(qc, (unsigned)vmx_shr1_in((int)qc, (int)qchunk));
                cH = (unsigned)vmx_rol1((int)cH); cE1 = (unsigned)vmx_rol1((int)cE1); cE2 = (unsigned)vmx_rol1((int)cE2); qchunk = (unsigned)vmx_rol1((int)qchunk);
                const unsigned a1 = vmx_pk_sub(upH, O1), a2 = vmx_pk_sub(upH, O2);
                unsigned b = vmx_pk_neg(vmx_pk_sub(a1, upE1)) & 0x00080008u;                       // upE1 > a1
                b |= vmx_pk_neg(vmx_pk_sub(a2, upE2)) & 0x00100010u;
                const unsigned e1v = vmx_pk_sub(vmx_pk_max(a1, upE1), E1C), e2v = vmx_pk_sub(vmx_pk_max(a2, upE2), E2C);
                const unsigned c1 = vmx_pk_sub(Hleft, O1), c2 = vmx_pk_sub(Hleft, O2);
                b |= vmx_pk_neg(vmx_pk_sub(c1, F1)) & 0x00200020u;                                  // F1 > c1
                b |= vmx_pk_neg(vmx_pk_sub(c2, F2)) & 0x00400040u;
                const unsigned nF1 = vmx_pk_sub(vmx_pk_max(c1, F1), E1C), nF2 = vmx_pk_sub(vmx_pk_max(c2, F2), E2C);
                const unsigned eqm = vmx_pk_neg(vmx_pk_sub(ti2 ^ qc, ONE));                         // codes are 0..5: x - 1 < 0 iff x == 0
                unsigned h = vmx_pk_add(Hdiag, vmx_bfi(eqm, MATCH, MISM));
                unsigned src = 0, m;
                m = vmx_pk_neg(vmx_pk_sub(h, e1v)); src = vmx_bfi(m, 0x00010001u, src); h = vmx_pk_max(h, e1v);
                m = vmx_pk_neg(vmx_pk_sub(h, nF1)); src = vmx_bfi(m, 0x00030003u, src); h = vmx_pk_max(h, nF1);
                m = vmx_pk_neg(vmx_pk_sub(h, e2v)); src = vmx_bfi(m, 0x00020002u, src); h = vmx_pk_max(h, e2v);
                m = vmx_pk_neg(vmx_pk_sub(h, nF2)); src = vmx_bfi(m, 0x00040004u, src); h = vmx_pk_max(h, nF2);
                b |= src;
                *(uint16_t*)(tbs + (size_t)t * 128) = (uint16_t)vmx_pk_bytes(b);
                if (ramp_up) {
                    // low half active from step 2*lane, high half from step 2*lane + 1
                    const unsigned pm = (t >= 2 * lane ? 0xffffu : 0u) | (t >= 2 * lane + 1 ? 0xffff0000u : 0u);
                    Hdiag = vmx_bfi(pm, upH, Hdiag); Hleft = vmx_bfi(pm, h, Hleft); F1 = vmx_bfi(pm, nF1, F1); F2 = vmx_bfi(pm, nF2, F2);
                } else { Hdiag = upH; Hleft = h; F1 = nF1; F2 = nF2; }
                outH = h; outE1 = e1v; outE2 = e2v;
                fin = (t == t_fin) ? outH : fin;
                if (store_bnd) {
                    sH = vmx_ror1(lane == 63 ? vmx_pk_hi(outH) : sH); sE1 = vmx_ror1(lane == 63 ? vmx_pk_hi(outE1) : sE1); sE2 = vmx_ror1(lane == 63 ? vmx_pk_hi(outE2) : sE2);
                    const int j63 = t - 126;                  // column the last row of the stripe (lane 63, high half) just finished
                    if (j63 >= 1 && j63 <= ql && ((j63 & 63) == 0 || j63 == ql)) {
                        const int col = j63 - lane;
                        if (col > ((j63 - 1) & ~63)) { bH[col] = sH; bE1[col] = sE1; bE2[col] = sE2; }
                    }
                }
            }
        }
        __syncthreads();   // bnd[] written by this stripe is read (in 64-column chunks) by the next one
    }
    if (lane == (rf >> 1)) *score_out = (rf & 1) ? vmx_pk_hi(fin) : vmx_pk_lo(fin);
}


// ---- four problems per wavefront (VMX_DP16X4_OK): the same packed recurrence, one problem per 16-lane DPP row. A lane owns rows 2l+1, 2l+2
// of a 32-row stripe; hand-offs are row_shr:1 (lane 0 of every row keeps the old operand: its own chunk head), chunk registers rotate with
// row_ror:15, the last row's FIFO with row_ror:1. A stripe is W = (ql + 31 rounded up to 16) steps wide, so every row starts its stripes
// on a multiple of 16 steps and all four rows refill their 16-column chunks together although they are in different stripes and
// columns. Small problems pad far less this way: rows to a multiple of 32 instead of 128, ramp 31 columns instead of 127.
// Traceback bytes: tb[((stripe*W + step)*16 + l)*2 + half].
#ifdef VMX_EMU
__device__ __forceinline__ int vmx_r16_shr1_in(int v, int in) { const int l = vmx_lane(); const int e = __shfl(v, (l & 48) | ((l + 15) & 15)); return (l & 15) == 0 ? in : e; }
__device__ __forceinline__ int vmx_r16_rol1(int v) { const int l = vmx_lane(); return __shfl(v, (l & 48) | ((l + 1) & 15)); }
__device__ __forceinline__ int vmx_r16_ror1(int v) { const int l = vmx_lane(); return __shfl(v, (l & 48) | ((l + 15) & 15)); }
#else
__device__ __forceinline__ int vmx_r16_shr1_in(int v, int in) { return __builtin_amdgcn_update_dpp(in, v, 0x111, 0xf, 0xf, false); }   // row_shr:1
__device__ __forceinline__ int vmx_r16_rol1(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x12F, 0xf, 0xf, false); }             // row_ror:15
__device__ __forceinline__ int vmx_r16_ror1(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x121, 0xf, 0xf, false); }             // row_ror:1
#endif

// T, Q, tl, ql, tb, bH, score_out describe the problem of this lane's 16-lane row (tl = 0: the row idles). Every lane of the wave calls it.
// Control flow is wave-uniform (every lane runs every step, idle rows on dummy values with their memory accesses masked): the four rows
// are in different stripes and columns, and the cross-lane moves must not sit in divergent code.
template <bool SCORE>
__device__ __forceinline__ int vmx_gapfill_fill16x4(const uint8_t* __restrict__ T, const uint8_t* __restrict__ Q, int tl, int ql, int match, int mismatch,
                                                    int o1, int e1, int o2, int e2, uint8_t* __restrict__ tb, int32_t* __restrict__ bH,
                                                    int32_t* __restrict__ score_out, int lane) {
    const int l = lane & 15;
    int32_t* bE1 = bH + (ql + 1);
    int32_t* bE2 = bE1 + (ql + 1);
    const int W = VMX_X4_W(ql);
    const int nstr = (tl + 31) >> 5;
    const unsigned O1 = vmx_pk(o1, o1), O2 = vmx_pk(o2, o2), E1C = vmx_pk(e1, e1), E2C = vmx_pk(e2, e2);
    const unsigned MATCH = vmx_pk(match, match), MISM = vmx_pk(mismatch, mismatch), ONE = vmx_pk(1, 1);
    const unsigned NEGP = vmx_pk(VMX_NEG16, VMX_NEG16);
    const int rf = (tl - 1) & 31;                               // row tl inside the last stripe: lane rf / 2 of the row, half rf & 1
    const int t_fin = ql - 1 + rf;
    const int total = vmx_uniform_i32(vmx_wave_max_i32(nstr * W));   // steps of the longest of the four problems
    unsigned fin = 0;
    int s = 0, t = 0;                                           // this row's stripe and the step inside it (t is a multiple of 16 here)
    unsigned ti2 = 0, Hleft = 0, F1 = NEGP, F2 = NEGP, Hdiag = 0, outH = 0, outE1 = NEGP, outE2 = NEGP, qc = vmx_pk(4, 4);
    unsigned sHE = 0, sE2 = 0;                                  // lane 15's high-half outputs of the last 16 steps, newest in lane 0: H | E1 << 16, E2
    bool store_bnd = false;
    uint8_t* tbp = tb;                                          // traceback line of the block's first step
    unsigned nq = 0, nH = 0, nE1 = 0, nE2 = 0;                  // chunk of the next block, loaded one block ahead
    // chunk of the 16 columns starting at step t0 of stripe s0: query bases and the row above the stripe, pre-shifted into the high half
    auto load_chunk = [&](bool on, int s0, int t0, unsigned& q, unsigned& h, unsigned& x1, unsigned& x2) {
        const int jj = t0 + l;                                  // column jj + 1
        q = (unsigned)(on && jj < ql ? (int)Q[jj] : 4) << 16;
        h = 0u; x1 = (unsigned)VMX_NEG16 << 16; x2 = x1;
        if (on && jj + 1 <= ql) {
            if (s0 == 0) h = (unsigned)vmx_gap_open_row(jj + 1, o1, e1, o2, e2) << 16;
            else { h = (unsigned)bH[jj + 1] << 16; x1 = (unsigned)bE1[jj + 1] << 16; x2 = (unsigned)bE2[jj + 1] << 16; }
        }
    };
    for (int g0 = 0; g0 < total; g0 += 16) {
        const bool act = s < nstr;
        if (__any(act && t == 0 && s > 0)) __syncthreads();    // a row is about to read the boundary rows its previous stripe stored
        unsigned qchunk, cH, cE1, cE2;
        if (act && t == 0) {
            const int i0 = s * 32 + 2 * l + 1;
            ti2 = vmx_pk(i0 <= tl ? vmx_tcode(T[i0 - 1]) : 5, i0 + 1 <= tl ? vmx_tcode(T[i0]) : 5);
            Hleft = vmx_pk(vmx_gap_open_row(i0, o1, e1, o2, e2), vmx_gap_open_row(i0 + 1, o1, e1, o2, e2));
            Hdiag = vmx_pk(i0 - 1 == 0 ? 0 : vmx_gap_open_row(i0 - 1, o1, e1, o2, e2), vmx_gap_open_row(i0, o1, e1, o2, e2));
            F1 = NEGP; F2 = NEGP;
            outH = 0; outE1 = NEGP; outE2 = NEGP; qc = vmx_pk(4, 4);
            sHE = 0; sE2 = 0;
            store_bnd = s + 1 < nstr;
            tbp = tb + (size_t)s * (size_t)W * 32 + 2 * l;
        }
        // a stripe's first chunk is loaded here (the previous stripe has only just stored it); the others were prefetched a block ago
        const bool first = __any(act && t == 0);
        if (first) load_chunk(act && t == 0, s, 0, qchunk, cH, cE1, cE2);
        if (!(act && t == 0)) { qchunk = nq; cH = nH; cE1 = nE1; cE2 = nE2; }
        load_chunk(act && t + 16 < W, s, t + 16, nq, nH, nE1, nE2);
        const bool any_bnd = __any(act && store_bnd);
        const unsigned l15 = l == 15 ? 0xffffffffu : 0u;
#define VMX_X4_STEP(RAMP, TT)                                                                                                      \
            {                                                                                                                      \
                const unsigned upH = vmx_alignbit16(outH, (unsigned)vmx_r16_shr1_in((int)outH, (int)cH));                          \
                const unsigned upE1 = vmx_alignbit16(outE1, (unsigned)vmx_r16_shr1_in((int)outE1, (int)cE1));                      \
                const unsigned upE2 = vmx_alignbit16(outE2, (unsigned)vmx_r16_shr1_in((int)outE2, (int)cE2));                      \
                qc = vmx_alignbit16(qc, (unsigned)vmx_r16_shr1_in((int)qc, (int)qchunk));                                          \
                cH = (unsigned)vmx_r16_rol1((int)cH); cE1 = (unsigned)vmx_r16_rol1((int)cE1); cE2 = (unsigned)vmx_r16_rol1((int)cE2); \
                qchunk = (unsigned)vmx_r16_rol1((int)qchunk);                                                                      \
                const unsigned a1 = vmx_pk_sub(upH, O1), a2 = vmx_pk_sub(upH, O2);                                                 \
                unsigned b = vmx_pk_neg(vmx_pk_sub(a1, upE1)) & 0x00080008u;                                                       \
                b |= vmx_pk_neg(vmx_pk_sub(a2, upE2)) & 0x00100010u;                                                               \
                const unsigned e1v = vmx_pk_sub(vmx_pk_max(a1, upE1), E1C), e2v = vmx_pk_sub(vmx_pk_max(a2, upE2), E2C);           \
                const unsigned c1 = vmx_pk_sub(Hleft, O1), c2 = vmx_pk_sub(Hleft, O2);                                             \
                b |= vmx_pk_neg(vmx_pk_sub(c1, F1)) & 0x00200020u;                                                                 \
                b |= vmx_pk_neg(vmx_pk_sub(c2, F2)) & 0x00400040u;                                                                 \
                const unsigned nF1 = vmx_pk_sub(vmx_pk_max(c1, F1), E1C), nF2 = vmx_pk_sub(vmx_pk_max(c2, F2), E2C);               \
                const unsigned eqm = vmx_pk_neg(vmx_pk_sub(ti2 ^ qc, ONE));                                                        \
                unsigned h = vmx_pk_add(Hdiag, vmx_bfi(eqm, MATCH, MISM));                                                         \
                unsigned src = 0, m;                                                                                               \
                m = vmx_pk_neg(vmx_pk_sub(h, e1v)); src = vmx_bfi(m, 0x00010001u, src); h = vmx_pk_max(h, e1v);                    \
                m = vmx_pk_neg(vmx_pk_sub(h, nF1)); src = vmx_bfi(m, 0x00030003u, src); h = vmx_pk_max(h, nF1);                    \
                m = vmx_pk_neg(vmx_pk_sub(h, e2v)); src = vmx_bfi(m, 0x00020002u, src); h = vmx_pk_max(h, e2v);                    \
                m = vmx_pk_neg(vmx_pk_sub(h, nF2)); src = vmx_bfi(m, 0x00040004u, src); h = vmx_pk_max(h, nF2);                    \
                b |= src;                                                                                                          \
                if (act) *(uint16_t*)(tbp + (TT) * 32) = (uint16_t)vmx_pk_bytes(b);                                                \
                if (RAMP) {                                                                                                        \
                    const int tc = t + (TT);                                                                                       \
                    const unsigned pm = (tc >= 2 * l ? 0xffffu : 0u) | (tc >= 2 * l + 1 ? 0xffff0000u : 0u);                       \
                    Hdiag = vmx_bfi(pm, upH, Hdiag); Hleft = vmx_bfi(pm, h, Hleft); F1 = vmx_bfi(pm, nF1, F1); F2 = vmx_bfi(pm, nF2, F2); \
                } else { Hdiag = upH; Hleft = h; F1 = nF1; F2 = nF2; }                                                             \
                outH = h; outE1 = e1v; outE2 = e2v;                                                                                \
                if (SCORE) fin = (tfin_rel == (TT)) ? outH : fin;                                                                  \
                if (any_bnd) {                                                                                                     \
                    sHE = (unsigned)vmx_r16_ror1((int)vmx_bfi(l15, (outH >> 16) | (outE1 & 0xffff0000u), sHE));                    \
                    sE2 = (unsigned)vmx_r16_ror1((int)vmx_bfi(l15, outE2 >> 16, sE2));                                             \
                    if ((TT) == 14) {      /* t is a multiple of 16: the last row finishes a column that is a multiple of 16 on steps = 14 mod 16 */ \
                        const int colr = t + 14 - 30 - l;     /* lane k of the FIFO holds (stripe-relative) column j15 - k */      \
                        const int col = colr;                                                                                      \
                        if (store_bnd && act && colr >= 1 && col <= ql) { bH[col] = (int)(short)(sHE & 0xffffu); bE1[col] = (int)sHE >> 16; bE2[col] = (int)(short)(sE2 & 0xffffu); } \
                    }                                                                                                              \
                }                                                                                                                  \
            }
        const int tfin_rel = act ? t_fin - t : -1;                 // a finished row keeps stepping on dummy values: never matches
        // RAMP: some half of some row has not reached its column 1 yet (its state must not move); rows past their ramp get a full mask
        if (__any(act && t < 32)) {
            VMX_X4_STEP(true, 0) VMX_X4_STEP(true, 1) VMX_X4_STEP(true, 2) VMX_X4_STEP(true, 3) VMX_X4_STEP(true, 4) VMX_X4_STEP(true, 5) VMX_X4_STEP(true, 6) VMX_X4_STEP(true, 7)
            VMX_X4_STEP(true, 8) VMX_X4_STEP(true, 9) VMX_X4_STEP(true, 10) VMX_X4_STEP(true, 11) VMX_X4_STEP(true, 12) VMX_X4_STEP(true, 13) VMX_X4_STEP(true, 14) VMX_X4_STEP(true, 15)
        } else {
            VMX_X4_STEP(false, 0) VMX_X4_STEP(false, 1) VMX_X4_STEP(false, 2) VMX_X4_STEP(false, 3) VMX_X4_STEP(false, 4) VMX_X4_STEP(false, 5) VMX_X4_STEP(false, 6) VMX_X4_STEP(false, 7)
            VMX_X4_STEP(false, 8) VMX_X4_STEP(false, 9) VMX_X4_STEP(false, 10) VMX_X4_STEP(false, 11) VMX_X4_STEP(false, 12) VMX_X4_STEP(false, 13) VMX_X4_STEP(false, 14) VMX_X4_STEP(false, 15)
        }
#undef VMX_X4_STEP
        if (act) {
            t += 16; tbp += 16 * 32;
            if (t == W) {
                // the columns behind the last multiple of 16 are still in the FIFO (W - 31 >= ql): lane k holds column W - 31 - k
                const int colr = W - 31 - l, col = colr;
                if (store_bnd && colr > ((W - 31) & ~15) && col <= ql) { bH[col] = (int)(short)(sHE & 0xffffu); bE1[col] = (int)sHE >> 16; bE2[col] = (int)(short)(sE2 & 0xffffu); }
                t = 0; ++s;
            }
        }
    }
    const int sc = (rf & 1) ? vmx_pk_hi(fin) : vmx_pk_lo(fin);
    if (SCORE && nstr > 0 && l == (rf >> 1)) *score_out = sc;
    return sc;                                                  // meaningful in lane rf / 2 of the row
}

#include "vmx_dp_ad.h"

// one problem on the whole wavefront (p is wave-uniform): int32 cells, one row per lane, or packed int16, two rows per lane
template <bool SCORE>
__device__ __forceinline__ void vmx_gapfill_fill_one(const uint8_t* __restrict__ tcodes, const uint8_t* __restrict__ qcodes, const vmx_dp_prob* __restrict__ probs, int p,
                                                    int match, int mismatch, int o1, int e1, int o2, int e2, uint8_t* __restrict__ tb_pool,
                                                    int32_t* __restrict__ bnd_pool, int32_t* __restrict__ out_score, int lane, int flag_value = 0, uint8_t* __restrict__ redo_pool = nullptr) {
        const vmx_dp_prob pr = probs[p];
        const uint8_t* T = tcodes + pr.t_off;
        const uint8_t* Q = qcodes + pr.q_off;
        const int tl = vmx_uniform_i32(pr.tl), ql = vmx_uniform_i32(pr.ql);
        const bool trivial = tl == 0 || ql == 0;     // no barrier-skipping `continue`: an empty side just runs zero stripes
        if (trivial && lane == 0) out_score[p] = (tl + ql) ? vmx_gap_open_row(tl + ql, o1, e1, o2, e2) : 0;
        if (!SCORE && !trivial && lane == 0) out_score[p] = flag_value;     // layout flag (0: the layout of the problem's size class)
        uint8_t* tb = vmx_tb_ptr(tb_pool, redo_pool, pr.tb_off);
        int32_t* bH = bnd_pool + pr.bnd_off;
        int32_t* bE1 = bH + (ql + 1);
        int32_t* bE2 = bE1 + (ql + 1);
        const bool pk = !trivial && VMX_DP16_OK(tl, ql);      // two rows per lane in packed int16 (below); larger problems keep int32
        const int W = ql + 63;
        const int nstr = (trivial || pk) ? 0 : (tl + 63) >> 6;
        int outH = 0;
        for (int s = 0; s < nstr; ++s) {
            const int i = s * 64 + lane + 1;
            const int ti = i <= tl ? (int)T[i - 1] : 4;       // rows past tl (last stripe) compute unused cells into the stripe's padding
            int Hleft = vmx_gap_open_row(i, o1, e1, o2, e2);
            int F1 = VMX_NEG, F2 = VMX_NEG;
            int Hdiag = (i - 1 == 0) ? 0 : vmx_gap_open_row(i - 1, o1, e1, o2, e2);
            int outE1 = VMX_NEG, outE2 = VMX_NEG;
            outH = 0;
            int qc_cur = 4;
            int sH = 0, sE1 = 0, sE2 = 0;                    // lane 63's H/E1/E2 of the last 64 steps, newest in lane 0 (rotating register FIFO)
            const bool store_bnd = s + 1 < nstr;              // the row below this stripe exists
            uint8_t* tbs = tb + (size_t)s * (size_t)W * 64 + lane;
            // one step of the stripe: hand-off from the lane above (lane 0: head of the rotating chunk registers), the cell, lane 63's FIFO
#define VMX_GF_STEP(PRED)                                                                                             \
            {                                                                                                         \
                const int upH = vmx_shr1_in(outH, cH), upE1 = vmx_shr1_in(outE1, cE1), upE2 = vmx_shr1_in(outE2, cE2); \
                qc_cur = vmx_shr1_in(qc_cur, qchunk);                                                                 \
                cH = vmx_rol1(cH); cE1 = vmx_rol1(cE1); cE2 = vmx_rol1(cE2); qchunk = vmx_rol1(qchunk);               \
                if (PRED) {                                                                                           \
                    int b = 0;                                                                                        \
                    const int a1 = upH - o1, a2 = upH - o2;                                                           \
                    if (upE1 > a1) b |= 8;                                                                            \
                    if (upE2 > a2) b |= 16;                                                                           \
                    const int e1v = (a1 > upE1 ? a1 : upE1) - e1, e2v = (a2 > upE2 ? a2 : upE2) - e2;                 \
                    const int c1 = Hleft - o1, c2 = Hleft - o2;                                                       \
                    if (F1 > c1) b |= 32;                                                                             \
                    if (F2 > c2) b |= 64;                                                                             \
                    F1 = (c1 > F1 ? c1 : F1) - e1; F2 = (c2 > F2 ? c2 : F2) - e2;                                     \
                    int h = Hdiag + ((ti == qc_cur && ti < 4) ? match : mismatch);                                    \
                    int src = 0;                                                                                      \
                    if (e1v > h) { h = e1v; src = 1; }                                                                \
                    if (F1 > h) { h = F1; src = 3; }                                                                  \
                    if (e2v > h) { h = e2v; src = 2; }                                                                \
                    if (F2 > h) { h = F2; src = 4; }                                                                  \
                    tbs[(size_t)t * 64] = (uint8_t)(b | src);                                                         \
                    Hdiag = upH; Hleft = h;                                                                           \
                    outH = h; outE1 = e1v; outE2 = e2v;                                                               \
                }                                                                                                     \
                if (store_bnd) {                                                                                      \
                    sH = vmx_ror1(lane == 63 ? outH : sH); sE1 = vmx_ror1(lane == 63 ? outE1 : sE1); sE2 = vmx_ror1(lane == 63 ? outE2 : sE2); \
                    const int j63 = t - 62;                   /* column lane 63 just finished */                      \
                    if (j63 >= 1 && j63 <= ql && ((j63 & 63) == 0 || j63 == ql)) {                                    \
                        const int col = j63 - lane;           /* lane k of the FIFO holds column j63 - k */           \
                        if (col > ((j63 - 1) & ~63)) { bH[col] = sH; bE1[col] = sE1; bE2[col] = sE2; }                \
                    }                                                                                                 \
                }                                                                                                     \
            }
            for (int t0 = 0; t0 < W; t0 += 64) {
                // 64-column chunks for lane 0: query bases Q[t0 .. t0+63] and the boundary row entries of columns j = t0+1 .. t0+64
                const int jj = t0 + lane;
                int qchunk = jj < ql ? (int)Q[jj] : 4;
                int cH = 0, cE1 = VMX_NEG, cE2 = VMX_NEG;
                if (jj + 1 <= ql) {
                    if (s == 0) cH = vmx_gap_open_row(jj + 1, o1, e1, o2, e2);
                    else { cH = bH[jj + 1]; cE1 = bE1[jj + 1]; cE2 = bE2[jj + 1]; }
                }
                if (t0 >= 64 && t0 + 64 <= ql) {
                    // every lane is inside its row for the whole chunk: no per-step activity test
                    for (int tt = 0; tt < 64; ++tt) { const int t = t0 + tt; VMX_GF_STEP(true) }
                } else {
                    int tend = W - t0; if (tend > 64) tend = 64;
                    for (int tt = 0; tt < tend; ++tt) { const int t = t0 + tt; const int j = t - lane + 1; VMX_GF_STEP(j >= 1 && j <= ql) }
                }
            }
#undef VMX_GF_STEP
            __syncthreads();   // bnd[] written by this stripe is read (in 64-column chunks) by the next one
        }
        if (!trivial && !pk && lane == ((tl - 1) & 63)) out_score[p] = outH;     // H(tl, ql): the last cell that lane computed
        int32_t sink = 0;
        if (pk) vmx_gapfill_fill16(T, Q, tl, ql, match, mismatch, o1, e1, o2, e2, tb, bH, SCORE ? &out_score[p] : &sink, lane);
}

// order/counter: longest-first device work queue (order == nullptr: plain grid-stride over [0, n_prob))
// SCORE = false: the batched path only consumes the traceback, so the small-problem form skips the score capture; out_score[p] then carries
// the layout flag the traceback kernel reads (0: striped). redo_pass: the queue is the list the first launch of the batched path left
// behind (redo_cnt[0] = entries, redo_cnt[1] = this launch's queue head).
template <bool SCORE>
__device__ __forceinline__ void vmx_gapfill_fill_body(const uint8_t* __restrict__ tcodes, const uint8_t* __restrict__ qcodes,
                                                     const vmx_dp_prob* __restrict__ probs, int n_prob, int match, int mismatch,
                                                     int o1, int e1, int o2, int e2, uint8_t* __restrict__ tb_pool,
                                                     int32_t* __restrict__ bnd_pool, int32_t* __restrict__ out_score,
                                                     const int32_t* __restrict__ order, int32_t* __restrict__ counter,
                                                     int32_t* __restrict__ redo_list = nullptr, int32_t* __restrict__ redo_cnt = nullptr, int redo_pass = 0,
                                                     uint8_t* __restrict__ redo_pool = nullptr) {
    const int lane = vmx_lane();
    int static_next = 4 * (int)blockIdx.x;
    if (redo_pass) {
        n_prob = redo_cnt[0]; order = redo_list; counter = redo_cnt + 1;
        // the larger problems of the list first, one per task on the whole wave in the packed two-rows-per-lane layout (flag VMX_PK_FLAG): on a
        // single 16-lane row a 500 x 500 matrix is a millisecond-long serial chain that the rest of the launch would wait for
        while (true) {
            int q; { int v = 0; if (lane == 0) v = atomicAdd(redo_cnt + 2, 1); q = vmx_bcast0(v); }
            if (q >= n_prob) break;
            const int p = order[q];
            const int tl = probs[p].tl, ql = probs[p].ql;
            if (VMX_REDO_PK(tl, ql)) vmx_gapfill_fill_one<SCORE>(tcodes, qcodes, probs, p, match, mismatch, o1, e1, o2, e2, tb_pool, bnd_pool, out_score, lane, VMX_PK_FLAG, redo_pool);
        }
    }
    while (true) {
        // a wave takes four problems at a time: those of the small class run together, one per 16-lane row (vmx_gapfill_fill16x4); the
        // others (the head of the longest-first queue) run one after the other on the whole wave
        int q0;
        if (order) { int v = 0; if (lane == 0) v = atomicAdd(counter, 4); q0 = vmx_bcast0(v); }
        else { q0 = static_next; static_next += 4 * (int)gridDim.x; }
        if (q0 >= n_prob) break;
        const int qg = q0 + (lane >> 4);
        const int pg = qg < n_prob ? (order ? order[qg] : qg) : -1;
        bool x4 = false, skip = false;
        {
            vmx_dp_prob pr; pr.tl = 0; pr.ql = 0; pr.t_off = 0; pr.q_off = 0; pr.tb_off = 0; pr.bnd_off = 0;
            if (pg >= 0) pr = probs[pg];
            skip = redo_pass && pg >= 0 && VMX_REDO_PK(pr.tl, pr.ql);          // done above
            x4 = pg >= 0 && !skip && pr.tl > 0 && pr.ql > 0 && VMX_DP16X4_OK(pr.tl, pr.ql);
            if (!SCORE && x4 && (lane & 15) == 0) out_score[pg] = 0;
            if (__any(x4))
                vmx_gapfill_fill16x4<SCORE>(tcodes + pr.t_off, qcodes + pr.q_off, x4 ? pr.tl : 0, x4 ? pr.ql : 0, match, mismatch, o1, e1, o2, e2, vmx_tb_ptr(tb_pool, redo_pool, pr.tb_off),
                                            bnd_pool + pr.bnd_off, &out_score[pg < 0 ? 0 : pg], lane);
        }
        for (int gk = 0; gk < 4; ++gk) {
            const int p = vmx_readlane(pg, 16 * gk);
            if (p < 0 || vmx_readlane((int)(x4 || skip), 16 * gk)) continue;
            vmx_gapfill_fill_one<SCORE>(tcodes, qcodes, probs, p, match, mismatch, o1, e1, o2, e2, tb_pool, bnd_pool, out_score, lane, 0, redo_pool);
        }
    }
}

// first launch of the batched path: EIGHT problems at a time. Those of the small class whose shape a band can hold (vmx_ad_ns) run together
// in the anti-diagonal form (vmx_dp_ad.h), all with the widest band any of the eight asks for; a problem whose result is proven keeps it
// (layout flag VMX_AD_FLAG + ns), the others — not proven, or small but not worth a band — are appended to redo_list for the second launch
// (vmx_gapfill_fill_body with redo_pass = 1: full matrix, four per wave). Larger problems run one after the other on the whole wave.
__device__ __forceinline__ void vmx_gapfill_ad_pass(const uint8_t* __restrict__ tcodes, const uint8_t* __restrict__ qcodes, vmx_dp_prob* __restrict__ probs, int n_prob,
                                                   int match, int mismatch, int o1, int e1, int o2, int e2, uint8_t* __restrict__ tb_pool, int32_t* __restrict__ bnd_pool,
                                                   int32_t* __restrict__ out_score, const int32_t* __restrict__ order, const int32_t* __restrict__ range, int32_t* __restrict__ counter,
                                                   int32_t* __restrict__ redo_list, int32_t* __restrict__ redo_cnt, int ad_pct, unsigned long long* __restrict__ redo_bytes,
                                                   unsigned long long redo_cap, int tb_by_ns) {
    const int lane = vmx_lane();
    const int pct = ad_pct & 0xffff, pct_min = (ad_pct >> 16) & 0xffff;
    // the head of the longest-first queue (range[0] entries: the size classes above the small one) goes one problem per task: eight of them in
    // a row on one wave would be a multi-millisecond serial chain at the start of the launch; then eight at a time
    const int n_head = range[0] < n_prob ? range[0] : n_prob;
    bool head = n_head > 0;
    while (true) {
        int pX = -1, pY = -1;                                  // this lane's row's two problems (-1: none)
        if (head) {
            int q; { int v = 0; if (lane == 0) v = atomicAdd(counter + 1, 1); q = vmx_bcast0(v); }
            if (q >= n_head) { head = false; continue; }
            if (lane < 16) pX = order[q];
        } else {
            int q0; { int v = 0; if (lane == 0) v = atomicAdd(counter, 8); q0 = n_head + vmx_bcast0(v); }
            if (q0 >= n_prob) break;
            const int qX = q0 + 2 * (lane >> 4), qY = qX + 1;
            if (qX < n_prob) pX = order[qX];
            if (qY < n_prob) pY = order[qY];
        }
        vmx_dp_prob prX, prY;
        prX.tl = 0; prX.ql = 0; prX.t_off = 0; prX.q_off = 0; prX.tb_off = 0; prY = prX;
        if (pX >= 0) prX = probs[pX];
        if (pY >= 0) prY = probs[pY];
        const bool x4X = pX >= 0 && prX.tl > 0 && prX.ql > 0 && VMX_DP16X4_OK(prX.tl, prX.ql), x4Y = pY >= 0 && prY.tl > 0 && prY.ql > 0 && VMX_DP16X4_OK(prY.tl, prY.ql);
        const int nsX = x4X ? vmx_ad_ns(prX.tl, prX.ql, match, o1, e1, o2, e2, pct, pct_min) : 0, nsY = x4Y ? vmx_ad_ns(prY.tl, prY.ql, match, o1, e1, o2, e2, pct, pct_min) : 0;
        const int ns = vmx_uniform_i32(vmx_wave_max_i32(nsX > nsY ? nsX : nsY));
        bool keepX = false, keepY = false, triedX = false, triedY = false;
        if (ns > 0) {
            int dloX = 0, dloY = 0;
            // tb_by_ns (the batched path, round 6): a problem's traceback space was sized for the slot width of ITS OWN band (k_round_prep: VMX_AD_W(own ns) bytes per lane
            // and anti-diagonal) and the queue is ordered by that width, so the eight problems of a task agree as a rule; one whose own width differs from the task's
            // (a class boundary inside the task) cannot store the wider slots and goes to the second launch instead
            const bool fitX = !tb_by_ns || VMX_AD_W(nsX) == VMX_AD_W(ns), fitY = !tb_by_ns || VMX_AD_W(nsY) == VMX_AD_W(ns);
            const int gX = (nsX > 0 && fitX) ? vmx_ad_geom(prX.tl, prX.ql, ns, &dloX) : 0, gY = (nsY > 0 && fitY) ? vmx_ad_geom(prY.tl, prY.ql, ns, &dloY) : 0;
            const int tlX = gX > 0 ? prX.tl : 0, qlX = gX > 0 ? prX.ql : 0, tlY = gY > 0 ? prY.tl : 0, qlY = gY > 0 ? prY.ql : 0;
            int scX = 0, scY = 0;
#define VMX_AD_RUN(NSV) vmx_gapfill_fill_ad<NSV>(tcodes + prX.t_off, qcodes + prX.q_off, tlX, qlX, dloX, tb_pool + prX.tb_off, tcodes + prY.t_off, qcodes + prY.q_off, tlY, qlY, dloY, \
                                                 tb_pool + prY.tb_off, match, mismatch, o1, e1, o2, e2, lane, scX, scY)
            if (ns == 1) VMX_AD_RUN(1); else if (ns == 2) VMX_AD_RUN(2); else if (ns == 3) VMX_AD_RUN(3); else VMX_AD_RUN(4);
#undef VMX_AD_RUN
            triedX = gX > 0; triedY = gY > 0;
            keepX = gX > 0 && vmx_ad_proven(scX, prX.tl, prX.ql, gX, match, o1, e1, o2, e2);
            keepY = gY > 0 && vmx_ad_proven(scY, prY.tl, prY.ql, gY, match, o1, e1, o2, e2);
        }
        if ((lane & 15) == 0) {
            // a problem for the second launch takes its full-matrix traceback space out of the second pool
            if (x4X) {
                out_score[pX] = keepX ? VMX_AD_FLAG + ns : 0;
                if (!keepX) {
                    if (triedX) atomicAdd(redo_cnt + 3, 1);          // tried in a band and not proven (the host adapts the band-width rule to this rate)
                    // (round 6: the second pool is sized from history, not from a read-back of this counter: an allocation that does not fit EMPTIES the problem — no
                    //  fill, an empty CIGAR — and the host, which reads the counter with the batch's results, grows the pool and runs the batch again)
                    const unsigned long long need = (unsigned long long)VMX_REDO_TB_BYTES(prX.tl, prX.ql), at = atomicAdd(redo_bytes, need);
                    if (at + need > redo_cap) { probs[pX].tl = 0; probs[pX].ql = 0; probs[pX].tb_off = 0; }
                    else { probs[pX].tb_off = -(int64_t)at - 1; redo_list[atomicAdd(redo_cnt, 1)] = pX; }
                }
            }
            if (x4Y) {
                out_score[pY] = keepY ? VMX_AD_FLAG + ns : 0;
                if (!keepY) {
                    if (triedY) atomicAdd(redo_cnt + 3, 1);          // tried in a band and not proven (the host adapts the band-width rule to this rate)
                    // (round 6: the second pool is sized from history, not from a read-back of this counter: an allocation that does not fit EMPTIES the problem — no
                    //  fill, an empty CIGAR — and the host, which reads the counter with the batch's results, grows the pool and runs the batch again)
                    const unsigned long long need = (unsigned long long)VMX_REDO_TB_BYTES(prY.tl, prY.ql), at = atomicAdd(redo_bytes, need);
                    if (at + need > redo_cap) { probs[pY].tl = 0; probs[pY].ql = 0; probs[pY].tb_off = 0; }
                    else { probs[pY].tb_off = -(int64_t)at - 1; redo_list[atomicAdd(redo_cnt, 1)] = pY; }
                }
            }
        }
        for (int gk = 0; gk < 8; ++gk) {
            const int p = vmx_readlane((gk & 1) ? pY : pX, 16 * (gk >> 1));
            if (p < 0 || vmx_readlane((int)((gk & 1) ? x4Y : x4X), 16 * (gk >> 1))) continue;
            vmx_gapfill_fill_one<false>(tcodes, qcodes, probs, p, match, mismatch, o1, e1, o2, e2, tb_pool, bnd_pool, out_score, lane);
        }
    }
}

__global__ void __launch_bounds__(64) k_gapfill_fill(const uint8_t* __restrict__ tcodes, const uint8_t* __restrict__ qcodes,
                                                     const vmx_dp_prob* __restrict__ probs, int n_prob, int match, int mismatch,
                                                     int o1, int e1, int o2, int e2, uint8_t* __restrict__ tb_pool,
                                                     int32_t* __restrict__ bnd_pool, int32_t* __restrict__ out_score,
                                                     const int32_t* __restrict__ order, int32_t* __restrict__ counter) {
    vmx_gapfill_fill_body<true>(tcodes, qcodes, probs, n_prob, match, mismatch, o1, e1, o2, e2, tb_pool, bnd_pool, out_score, order, counter);
}
__global__ void __launch_bounds__(64, 4) k_gapfill_fill_ns(const uint8_t* __restrict__ tcodes, const uint8_t* __restrict__ qcodes,
                                                        vmx_dp_prob* __restrict__ probs, int n_prob, int match, int mismatch,
                                                        int o1, int e1, int o2, int e2, uint8_t* __restrict__ tb_pool,
                                                        int32_t* __restrict__ bnd_pool, int32_t* __restrict__ out_score,
                                                        const int32_t* __restrict__ order, const int32_t* __restrict__ range, int32_t* __restrict__ counter,
                                                        int32_t* __restrict__ redo_list, int32_t* __restrict__ redo_cnt, int redo_pass, int ad_pct,
                                                        uint8_t* __restrict__ redo_pool, unsigned long long* __restrict__ redo_bytes, const int32_t* __restrict__ n_ptr,
                                                        unsigned long long redo_cap, int tb_by_ns) {
    if (n_ptr) n_prob = *n_ptr;                        // the count on the device (an unplanned pass: the host launched for an upper bound)
    if (redo_pass) vmx_gapfill_fill_body<false>(tcodes, qcodes, probs, n_prob, match, mismatch, o1, e1, o2, e2, tb_pool, bnd_pool, out_score, order, counter, redo_list, redo_cnt, 1, redo_pool);
    else vmx_gapfill_ad_pass(tcodes, qcodes, probs, n_prob, match, mismatch, o1, e1, o2, e2, tb_pool, bnd_pool, out_score, order, range, counter, redo_list, redo_cnt, ad_pct, redo_bytes, redo_cap, tb_by_ns);
}

// serial traceback, one THREAD per problem (thousands of independent dependent-load chains hide each other's latency)
__global__ void k_gapfill_trace(const uint8_t* __restrict__ tcodes, const uint8_t* __restrict__ qcodes,
                                const vmx_dp_prob* __restrict__ probs, int n_prob, int eqx, const uint8_t* __restrict__ tb_pool,
                                uint32_t* __restrict__ run_pool, char* __restrict__ cig_pool, int32_t* __restrict__ cig_len,
                                const int32_t* __restrict__ band_flag, const uint8_t* __restrict__ redo_pool, int spread, int32_t* __restrict__ cig_q,
                                const int32_t* __restrict__ n_ptr) {
    VMX_SETPRIO(3);
    if (n_ptr) n_prob = *n_ptr;
    // one lane in `spread` works (like k_ext_phase: the walks of the 64 problems of a full wave diverge at every step)
    const int gt = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (spread > 1 && gt % spread) return;
    int p = spread > 1 ? gt / spread : gt;
    if (p >= n_prob) return;
    const vmx_dp_prob pr = probs[p];
    const uint8_t* T = tcodes + pr.t_off;
    const uint8_t* Q = qcodes + pr.q_off;
    const int tl = pr.tl, ql = pr.ql;
    const uint8_t* tb = vmx_tb_ptr(tb_pool, redo_pool, pr.tb_off);
    uint32_t* runs = run_pool + pr.run_off;
    char* cig = cig_pool + pr.cig_off;
    bool x4 = tl > 0 && ql > 0 && VMX_DP16X4_OK(tl, ql);         // layout of vmx_gapfill_fill16x4
    bool pk = !x4 && tl > 0 && ql > 0 && VMX_DP16_OK(tl, ql);           // packed layout of vmx_gapfill_fill16
    const int flag = (x4 && band_flag != nullptr) ? band_flag[p] : 0;
    const int ns = flag > VMX_AD_FLAG ? flag - VMX_AD_FLAG : 0;          // anti-diagonal layout (vmx_dp_ad.h) with 2 * ns diagonals per lane
    if (flag == VMX_PK_FLAG) { x4 = false; pk = true; }                  // a small-class problem the second launch ran on the whole wave
    int dlo = 0;
    if (ns) vmx_ad_geom(tl, ql, ns, &dlo);
    const int AW = ns ? VMX_AD_W(ns) : 4;                                // bytes of a lane's slot in the anti-diagonal layout (vmx_kernels.h)
    const int W = x4 ? VMX_X4_W(ql) : ql + (pk ? 127 : 63);
    int nruns = 0; int cur_op = -1; uint32_t cur_len = 0;
#define VMX_EMIT(op)                                                                   \
    do {                                                                               \
        if ((op) == cur_op) ++cur_len;                                                 \
        else { if (cur_op >= 0) runs[nruns++] = (cur_len << 8) | (uint32_t)cur_op; cur_op = (op); cur_len = 1; } \
    } while (0)
    int i = tl, j = ql, state = 0;
    while (i > 0 && j > 0) {
        int b;
        if (ns && state == 0) {
            // Anti-diagonal layout, H state: nine steps in ten are diagonal moves (a 10 % error read), and a diagonal move keeps the diagonal
            // x and goes back two anti-diagonals, i.e. 128 bytes. The bytes (and, for =/X, the codes) of the next eight cells down the
            // diagonal are loaded TOGETHER and consumed as long as the path stays on it: one memory latency per eight steps instead
            // of one per step (the walk is a chain of dependent loads, this kernel's whole cost).
            const int x = (j - i) - dlo, l = x / (2 * ns), k = (x - 2 * ns * l) >> 1;
            const uint8_t* cell = tb + VMX_AD_TB_OFF_W(0, l, AW) + k;
            const int s0 = i + j - 1;
            int m = i < j ? i : j; if (m > 8) m = 8;
            uint8_t bb[8], ta[8], qa[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) if (u < m) { bb[u] = cell[VMX_AD_TB_OFF_W(s0 - 2 * u, 0, AW)]; if (eqx) { ta[u] = T[i - 1 - u]; qa[u] = Q[j - 1 - u]; } }
            int u = 0;
#pragma unroll
            for (int v = 0; v < 8; ++v) {
                if (v == u && v < m) {
                    const int src = bb[v] & 7;
                    if (src == 0) { int op = 'M'; if (eqx) op = (ta[v] == qa[v] && ta[v] < 4) ? '=' : 'X'; VMX_EMIT(op); --i; --j; ++u; }
                    else state = src;
                }
            }
            if (state == 0) continue;                 // the whole batch was diagonal (or the matrix edge was reached)
            // a gap starts at cell (i, j): the generic step below re-reads its byte in the gap state
        }
        if (ns) { const int x = (j - i) - dlo, l = x / (2 * ns), k = (x - 2 * ns * l) >> 1; b = tb[VMX_AD_TB_OFF_W(i + j - 1, l, AW) + k]; }
        else if (x4) { const int s = (i - 1) >> 5, r = (i - 1) & 31, t = (j - 1) + r; b = tb[(((size_t)s * (size_t)W + (size_t)t) * 16 + (r >> 1)) * 2 + (r & 1)]; }
        else if (pk) { const int s = (i - 1) >> 7, r = (i - 1) & 127, t = (j - 1) + r; b = tb[(((size_t)s * (size_t)W + (size_t)t) * 64 + (r >> 1)) * 2 + (r & 1)]; }
        else { const int s = (i - 1) >> 6, l = (i - 1) & 63, t = (j - 1) + l; b = tb[((size_t)s * (size_t)W + (size_t)t) * 64 + l]; }
        if (state == 0) {
            int src = b & 7;
            if (src == 0) {
                int op = 'M';
                if (eqx) { uint8_t a = T[i - 1], c = Q[j - 1]; op = (a == c && a < 4) ? '=' : 'X'; }
                VMX_EMIT(op); --i; --j;
            } else state = src;
        } else if (state <= 2) {
            VMX_EMIT('D');
            int ext = state == 1 ? (b >> 3) & 1 : (b >> 4) & 1;
            --i; if (!ext) state = 0;
        } else {
            VMX_EMIT('I');
            int ext = state == 3 ? (b >> 5) & 1 : (b >> 6) & 1;
            --j; if (!ext) state = 0;
        }
    }
    while (i > 0) { VMX_EMIT('D'); --i; }
    while (j > 0) { VMX_EMIT('I'); --j; }
    if (cur_op >= 0) runs[nruns++] = (cur_len << 8) | (uint32_t)cur_op;
#undef VMX_EMIT
    // runs were produced end-to-start: print them in reverse
    int w = 0; long long qsum = 0;
    for (int r = nruns - 1; r >= 0; --r) {
        uint32_t len = runs[r] >> 8; char op = (char)(runs[r] & 0xff);
        if (op != 'D') qsum += len;                           // query bases the CIGAR consumes (M, =, X, I)
        char tmp[12]; int nd = 0;
        do { tmp[nd++] = (char)('0' + len % 10); len /= 10; } while (len);
        while (nd) cig[w++] = tmp[--nd];
        cig[w++] = op;
    }
    cig[w] = 0;
    cig_len[p] = w;
    if (cig_q) cig_q[p] = (int32_t)qsum;
}
