// vmx_dp_ad.h — E5 gap fill, anti-diagonal band form: EIGHT problems per wavefront (included by k_dp.hip only).
//
// The same dual-affine recurrence and the same 7-bit traceback byte as the striped forms in k_dp.hip (VMX-DP-G), laid out for the
// problems the read path actually produces (tl ~ ql ~ 270, |tl - ql| a few bases): a problem lives in one 16-lane DPP row and one 16-bit
// half of every register (the other half is a second, unrelated problem, so all v_pk_* arithmetic serves two problems), on a FIXED band
// of ND = 32 * NS diagonals d = j - i in [dlo, dlo + ND). Lane l owns the 2 * NS diagonals x = d - dlo in [2 NS l, 2 NS l + 2 NS): the even
// ones ("A" sets, k = 0 .. NS-1) are computed on even anti-diagonals a = i + j, the odd ones ("C" sets) on odd a (dlo is even). A cell
// needs the cell above it (diagonal x + 1) and the cell to its left (diagonal x - 1), both one anti-diagonal back, and its own
// diagonal's previous cell: all of that is in the lane's own registers except one neighbour per step, which comes from the next lane
// (C_{NS-1} takes A_0 of lane l + 1: row_shl:1) or the previous one (A_0 takes C_{NS-1} of lane l - 1: row_shr:1) — three DPP moves per
// step for 2 * NS cells per lane. There are no stripes, no boundary rows in memory and no ramp: a step computes a whole anti-diagonal of
// the band. Rows / columns 0 are not special either: the state starts as -infinity everywhere except H(0,0) = 0, and the recurrence
// itself produces H(0,j) = best gap of j (through F) and H(i,0) (through E) on the way; cells with i < 0 or j < 0 stay near -infinity,
// cells past tl / ql compute garbage that only ever feeds other garbage. Target codes move down the lanes one diagonal pair per two
// steps, query codes move up (one DPP each per step pair), fed at lane 0 / lane 15 from 16-entry chunk registers that rotate one lane
// per use and are refilled with one coalesced load every 16 pairs.
// Traceback bytes: tb[VMX_AD_TB_OFF_W(a - 1, l, W) + k] for the cell of anti-diagonal a on diagonal x = 2 NS l + 2 k + (a & 1): W = VMX_AD_W(NS) bytes per lane,
// problem and step (vmx_kernels.h: 64-byte lines of VMX_AD_AB anti-diagonals x 64 / (AB W) lanes).
#ifndef VMX_DP_AD_H
#define VMX_DP_AD_H

#ifdef VMX_EMU
__device__ __forceinline__ int vmx_r16_shl1_in(int v, int in) { const int l = vmx_lane(); const int e = __shfl(v, (l & 48) | ((l + 1) & 15)); return (l & 15) == 15 ? in : e; }
__device__ __forceinline__ unsigned vmx_perm(unsigned s0, unsigned s1, unsigned sel) {
    unsigned r = 0;
    for (int b = 0; b < 4; ++b) {
        const unsigned c = (sel >> (8 * b)) & 0xffu;
        const unsigned v = c < 4 ? (s1 >> (8 * c)) & 0xffu : c < 8 ? (s0 >> (8 * (c - 4))) & 0xffu : 0u;
        r |= v << (8 * b);
    }
    return r;
}
#else
__device__ __forceinline__ int vmx_r16_shl1_in(int v, int in) { return __builtin_amdgcn_update_dpp(in, v, 0x101, 0xf, 0xf, false); }   // row_shl:1
__device__ __forceinline__ unsigned vmx_perm(unsigned s0, unsigned s1, unsigned sel) { return __builtin_amdgcn_perm(s0, s1, sel); }
#endif

// Scores are kept BIASED: a register half holds score + VMX_AD_BIAS, always in [0, 32767]. Every wave-instruction of the two-operand 32-bit
// class (v_add_u32, v_sub_u32, v_and / or / xor, v_ashrrev_i32) issues in 2.3 cycles on gfx950, everything in the VOP3 / VOP3P / DPP class —
// all packed-int16 arithmetic — in 4.2 (profiles/r04_q_valu_calibration.md). With non-negative halves that can neither borrow nor carry
// into each other, subtracting a gap cost or adding the match score is a plain 32-bit subtract / add of the two halves at once, and a
// half's sign bit becomes a traceback flag with one 32-bit arithmetic shift and a mask; only what compares two scores (v_pk_max_i16, the
// sign of a v_pk_sub_i16) still needs the packed forms. 26 four-cycle + 22 two-cycle instructions per cell pair (was 40 + 9).
//   Range: real cells of a problem with tl + ql <= 1024 score in [-1100, 1024]; "-infinity" is -4096 and loses at most 2 per step
// (1024 steps) plus one gap opening, so with the bias 8192 every half stays in [2000, 9300].
#define VMX_AD_BIAS 8192
#define VMX_AD_NEGINF (-4096)
struct vmx_ad_consts { unsigned O1, O2, E1, E2, MATCH, NMIS, ONE, NEGP; };

// one cell (both halves): up = (H, E1, E2) of the cell above, left = (H, F1, F2) of the cell to the left, H = the diagonal predecessor on
// entry and the cell's H on return; tc / qc = target / query codes. Returns the traceback byte (bits 0-6 of each half).
__device__ __forceinline__ unsigned vmx_ad_cell(const vmx_ad_consts& K, unsigned upH, unsigned upE1, unsigned upE2, unsigned leftH, unsigned leftF1,
                                                unsigned leftF2, unsigned tc, unsigned qc, unsigned& H, unsigned& E1, unsigned& E2, unsigned& F1, unsigned& F2) {
    const unsigned a1 = upH - K.O1, a2 = upH - K.O2;                                   // (32-bit subtract: no half borrows)
    unsigned b = (unsigned)((int)vmx_pk_sub(a1, upE1) >> 12) & 0x00080008u;             // upE1 > a1: E1 extends (the halves' sign bits land on bits 3 and 19)
    b |= (unsigned)((int)vmx_pk_sub(a2, upE2) >> 11) & 0x00100010u;
    const unsigned e1v = vmx_pk_max(a1, upE1) - K.E1, e2v = vmx_pk_max(a2, upE2) - K.E2;
    const unsigned c1 = leftH - K.O1, c2 = leftH - K.O2;
    b |= (unsigned)((int)vmx_pk_sub(c1, leftF1) >> 10) & 0x00200020u;                   // F1 > c1
    b |= (unsigned)((int)vmx_pk_sub(c2, leftF2) >> 9) & 0x00400040u;
    const unsigned f1v = vmx_pk_max(c1, leftF1) - K.E1, f2v = vmx_pk_max(c2, leftF2) - K.E2;
    // codes are 0..6: min(tc ^ qc, 1) = 1 on a mismatch; h = H + match - (match - mismatch) * that
    unsigned h = vmx_pk_mad(vmx_pk_min_u16(tc ^ qc, K.ONE), K.NMIS, H) + K.MATCH;
    unsigned src = 0, m;
    m = vmx_pk_neg(vmx_pk_sub(h, e1v)); src = vmx_bfi(m, 0x00010001u, src); h = vmx_pk_max(h, e1v);
    m = vmx_pk_neg(vmx_pk_sub(h, f1v)); src = vmx_bfi(m, 0x00030003u, src); h = vmx_pk_max(h, f1v);     // ksw2's order: diagonal > E1 > F1 > E2 > F2
    m = vmx_pk_neg(vmx_pk_sub(h, e2v)); src = vmx_bfi(m, 0x00020002u, src); h = vmx_pk_max(h, e2v);
    m = vmx_pk_neg(vmx_pk_sub(h, f2v)); src = vmx_bfi(m, 0x00040004u, src); h = vmx_pk_max(h, f2v);
    H = h; E1 = e1v; E2 = e2v; F1 = f1v; F2 = f2v;
    return b | src;
}

// v[k] by masks, not by a select of array elements (which the compiler turns into a dynamically indexed load and the arrays into memory)
template <int NS>
__device__ __forceinline__ unsigned vmx_ad_pick(const unsigned (&v)[NS], int k) {
    unsigned r = 0;
#pragma unroll
    for (int i = 0; i < NS; ++i) r |= v[i] & (unsigned)-(int)(k == i);
    return r;
}

// X = the problem in the low halves, Y = the one in the high halves of this lane's 16-lane row (tl = 0: idle). Every lane of the wave
// calls it; control flow is wave-uniform. scoreX / scoreY: H(tl, ql) of the band, the same in all lanes of the row.
template <int NS>
__device__ __forceinline__ void vmx_gapfill_fill_ad(const uint8_t* __restrict__ TX, const uint8_t* __restrict__ QX, int tlX, int qlX, int dloX, uint8_t* __restrict__ tbX,
                                                   const uint8_t* __restrict__ TY, const uint8_t* __restrict__ QY, int tlY, int qlY, int dloY, uint8_t* __restrict__ tbY,
                                                   int match, int mismatch, int o1, int e1, int o2, int e2, int lane, int& scoreX, int& scoreY) {
    const int l = lane & 15;
    vmx_ad_consts K;
    K.O1 = vmx_pk(o1, o1); K.O2 = vmx_pk(o2, o2); K.E1 = vmx_pk(e1, e1); K.E2 = vmx_pk(e2, e2);
    K.MATCH = vmx_pk(match, match); K.NMIS = vmx_pk(mismatch - match, mismatch - match); K.ONE = vmx_pk(1, 1); K.NEGP = vmx_pk(VMX_AD_NEGINF + VMX_AD_BIAS, VMX_AD_NEGINF + VMX_AD_BIAS);
    const int afX = (tlX > 0 && qlX > 0) ? tlX + qlX : 0, afY = (tlY > 0 && qlY > 0) ? tlY + qlY : 0;     // the last anti-diagonal: cell (tl, ql)
    const int total = vmx_uniform_i32(vmx_wave_max_i32(afX > afY ? afX : afY));
    const int npairs = (total + 1) >> 1;                        // pair p = 1 ..: the odd step a = 2p - 1 (C sets), then the even step a = 2p (A sets)
    const int poX = (afX + 1) >> 1, peX = afX >> 1, poY = (afY + 1) >> 1, peY = afY >> 1;     // last pair whose odd / even step stores
    const int hX = dloX >> 1, hY = dloY >> 1;                   // dlo is even
    // set k of lane l in pair p: target row p - 1 - h - NS l - k on the odd step and one more on the even step, query column p + h + NS l + k on both
    auto tcf = [&](const uint8_t* T, int tl, int i) -> int { return (i >= 1 && i <= tl) ? vmx_tcode(T[i - 1]) : 5; };
    auto qcf = [&](const uint8_t* Q, int ql, int j) -> int { return (j >= 1 && j <= ql) ? (int)Q[j - 1] : 4; };
    unsigned tcode[NS], qcode[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) {
        tcode[k] = vmx_pk(tcf(TX, tlX, -hX - NS * l - k), tcf(TY, tlY, -hY - NS * l - k));
        qcode[k] = vmx_pk(qcf(QX, qlX, hX + NS * l + k), qcf(QY, qlY, hY + NS * l + k));          // one column back: pair 1 starts with a shift
    }
    unsigned HA[NS], E1A[NS], E2A[NS], F1A[NS], F2A[NS], HC[NS], E1C[NS], E2C[NS], F1C[NS], F2C[NS];
    {
        // H(0,0) = 0 on diagonal 0 (x = -dlo: an A set), -infinity everywhere else
        const int l0X = (-dloX) / (2 * NS), k0X = ((-dloX) % (2 * NS)) >> 1, l0Y = (-dloY) / (2 * NS), k0Y = ((-dloY) % (2 * NS)) >> 1;
#pragma unroll
        for (int k = 0; k < NS; ++k) {
            const unsigned m = ((l == l0X && k == k0X) ? 0xffffu : 0u) | ((l == l0Y && k == k0Y) ? 0xffff0000u : 0u);
            HA[k] = vmx_bfi(m, vmx_pk(VMX_AD_BIAS, VMX_AD_BIAS), K.NEGP);
            E1A[k] = K.NEGP; E2A[k] = K.NEGP; F1A[k] = K.NEGP; F2A[k] = K.NEGP;
            HC[k] = K.NEGP; E1C[k] = K.NEGP; E2C[k] = K.NEGP; F1C[k] = K.NEGP; F2C[k] = K.NEGP;
        }
    }
    // where H(tl, ql) lives: diagonal x = (ql - tl) - dlo
    const int xfX = (qlX - tlX) - dloX, xfY = (qlY - tlY) - dloY;
    const int lfX = xfX / (2 * NS), kfX = (xfX % (2 * NS)) >> 1, lfY = xfY / (2 * NS), kfY = (xfY % (2 * NS)) >> 1;
    int finX = 0, finY = 0;
    constexpr int W = VMX_AD_W(NS);                           // bytes of the lane's slot: as many as the lane has cells per step (round 6; 4 before, whatever NS)
    uint8_t* const pX = tbX + VMX_AD_TB_OFF_W(0, l, W);       // the lane's slot of anti-diagonal 0; the step's part of the offset is added per store
    uint8_t* const pY = tbY + VMX_AD_TB_OFF_W(0, l, W);
    auto put = [&](uint8_t* at, unsigned w01, unsigned w23, bool hi) {
        if constexpr (W == 1) *at = (uint8_t)(hi ? (w01 >> 16) : w01);
        else if constexpr (W == 2) *(uint16_t*)at = (uint16_t)(hi ? (w01 >> 16) : w01);
        else *(uint32_t*)at = vmx_perm(w23, w01, hi ? 0x07060302u : 0x05040100u);
    };
    // chunk of block b (pairs 16 b + 1 .. 16 b + 16): lane m holds the target code lane 0 takes in the block's pair m (row 16 b + 1 + m - h),
    // lane 15 - m the query code lane 15 takes in it (column 16 b + m + h + 16 NS)
    auto load_chunks = [&](int b16, unsigned& tch, unsigned& qch) {
        tch = vmx_pk(tcf(TX, tlX, b16 + 1 + l - hX), tcf(TY, tlY, b16 + 1 + l - hY));
        qch = vmx_pk(qcf(QX, qlX, b16 + 15 - l + hX + 16 * NS), qcf(QY, qlY, b16 + 15 - l + hY + 16 * NS));
    };
    unsigned tch, qch, ntch, nqch;
    load_chunks(0, ntch, nqch);
    for (int p0 = 0; p0 < npairs; p0 += 16) {
        tch = ntch; qch = nqch;
        load_chunks(p0 + 16, ntch, nqch);                       // one block ahead
        int pend = npairs - p0; if (pend > 16) pend = 16;
        for (int pp = 0; pp < pend; ++pp) {
            const int p = p0 + pp + 1;
            // query codes move up one diagonal pair
            {
                const unsigned in = (unsigned)vmx_r16_shl1_in((int)qcode[0], (int)qch);
#pragma unroll
                for (int k = 0; k + 1 < NS; ++k) qcode[k] = qcode[k + 1];
                qcode[NS - 1] = in;
                qch = (unsigned)vmx_r16_ror1((int)qch);
            }
            // odd step a = 2p - 1: C_k takes the cell above from A_{k+1} (the last one from the next lane) and the cell to its left from A_k
            unsigned bb[NS];
            {
                const unsigned nH = (unsigned)vmx_r16_shl1_in((int)HA[0], (int)K.NEGP), nE1 = (unsigned)vmx_r16_shl1_in((int)E1A[0], (int)K.NEGP),
                               nE2 = (unsigned)vmx_r16_shl1_in((int)E2A[0], (int)K.NEGP);
#pragma unroll
                for (int k = 0; k < NS; ++k) {
                    const unsigned uH = k + 1 < NS ? HA[k + 1 < NS ? k + 1 : 0] : nH, uE1 = k + 1 < NS ? E1A[k + 1 < NS ? k + 1 : 0] : nE1, uE2 = k + 1 < NS ? E2A[k + 1 < NS ? k + 1 : 0] : nE2;
                    bb[k] = vmx_ad_cell(K, uH, uE1, uE2, HA[k], F1A[k], F2A[k], tcode[k], qcode[k], HC[k], E1C[k], E2C[k], F1C[k], F2C[k]);
                }
            }
            {
                unsigned w01 = bb[0], w23 = 0;
                if (NS > 1) w01 |= bb[NS > 1 ? 1 : 0] << 8;
                if (NS > 2) w23 = bb[NS > 2 ? 2 : 0];
                if (NS > 3) w23 |= bb[NS > 3 ? 3 : 0] << 8;
                const size_t so = VMX_AD_TB_OFF_W(2 * p - 2, 0, W);   // anti-diagonal a = 2p - 1 is step s = a - 1
                if (p <= poX) put(pX + so, w01, w23, false);
                if (p <= poY) put(pY + so, w01, w23, true);
            }
            // target codes move down one diagonal pair
            {
                const unsigned in = (unsigned)vmx_r16_shr1_in((int)tcode[NS - 1], (int)tch);
#pragma unroll
                for (int k = NS - 1; k > 0; --k) tcode[k] = tcode[k - 1];
                tcode[0] = in;
                tch = (unsigned)vmx_r16_rol1((int)tch);
            }
            // even step a = 2p: A_k takes the cell above from C_k and the cell to its left from C_{k-1} (the first one from the previous lane)
            {
                const unsigned nH = (unsigned)vmx_r16_shr1_in((int)HC[NS - 1], (int)K.NEGP), nF1 = (unsigned)vmx_r16_shr1_in((int)F1C[NS - 1], (int)K.NEGP),
                               nF2 = (unsigned)vmx_r16_shr1_in((int)F2C[NS - 1], (int)K.NEGP);
#pragma unroll
                for (int k = 0; k < NS; ++k) {
                    const unsigned lH = k > 0 ? HC[k > 0 ? k - 1 : 0] : nH, lF1 = k > 0 ? F1C[k > 0 ? k - 1 : 0] : nF1, lF2 = k > 0 ? F2C[k > 0 ? k - 1 : 0] : nF2;
                    bb[k] = vmx_ad_cell(K, HC[k], E1C[k], E2C[k], lH, lF1, lF2, tcode[k], qcode[k], HA[k], E1A[k], E2A[k], F1A[k], F2A[k]);
                }
            }
            {
                unsigned w01 = bb[0], w23 = 0;
                if (NS > 1) w01 |= bb[NS > 1 ? 1 : 0] << 8;
                if (NS > 2) w23 = bb[NS > 2 ? 2 : 0];
                if (NS > 3) w23 |= bb[NS > 3 ? 3 : 0] << 8;
                const size_t so = VMX_AD_TB_OFF_W(2 * p - 1, 0, W);
                if (p <= peX) put(pX + so, w01, w23, false);
                if (p <= peY) put(pY + so, w01, w23, true);
            }
            // the pair that holds the problem's last anti-diagonal: its cell (tl, ql) was written by this pair's odd or even step
            if (__any(p == poX || p == poY)) {
                const unsigned aX = vmx_ad_pick<NS>(HA, kfX), cX = vmx_ad_pick<NS>(HC, kfX), aY = vmx_ad_pick<NS>(HA, kfY), cY = vmx_ad_pick<NS>(HC, kfY);
                const unsigned vX = (afX & 1) ? cX : aX, vY = (afY & 1) ? cY : aY;
                if (p == poX && l == lfX) finX = vmx_pk_lo(vX) - VMX_AD_BIAS;
                if (p == poY && l == lfY) finY = vmx_pk_hi(vY) - VMX_AD_BIAS;
            }
        }
    }
    scoreX = __shfl(finX, (lane & 48) | (lfX & 15));
    scoreY = __shfl(finY, (lane & 48) | (lfY & 15));
}

// Is the band's result the true optimum with the true traceback? A path that leaves the band holds at least g inserted and g deleted bases
// (vmx_ad_geom): impossible when g > min(tl, ql); otherwise it cannot score more than match * (min(tl, ql) - g) minus two gaps of g bases
// (splitting a gap never makes it cheaper). If the band's score beats that bound, every optimal path lies inside the band, where all
// cells it touches and all comparisons the traceback reads (they involve prefix-optimal values of cells on optimal paths only) are exact.
__device__ __forceinline__ bool vmx_ad_proven(int score, int tl, int ql, int g, int match, int o1, int e1, int o2, int e2) {
    const int mn = tl < ql ? tl : ql;
    if (g < 1) return false;
    if (g > mn) return true;
    const long long U = (long long)match * mn - vmx_ad_margin(g, match, o1, e1, o2, e2);      // match * (mn - g) - 2 * gap(g)
    return (long long)score > U;
}
#endif
