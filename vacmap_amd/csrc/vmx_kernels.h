// vmx_kernels.h — device-side data types shared by the kernels and the host orchestration.
#ifndef VMX_KERNELS_H
#define VMX_KERNELS_H
#include <stdint.h>

// one DP problem of the gap-fill stage (E5); offsets into the per-batch pools
struct vmx_dp_prob {
    int64_t t_off, q_off;     // into the target / query code pools
    int32_t tl, ql;
    int64_t tb_off;           // traceback bytes: ((tl+63)/64) * (ql+63) * 64
    int64_t bnd_off;          // int32 units: 3*(ql+1)
    int64_t run_off;          // uint32 units: tl+ql+2
    int64_t cig_off;          // bytes: 2*(tl+ql)+16
};

// anchor in the layout the chain kernels use (16 B): SURVEY §8(a) row T
struct vmx_anchor {
    int32_t q;     // read position
    int16_t l;     // length
    int16_t s;     // strand +1 / -1
    int64_t r;     // global reference position
};

// hashed minimizer index slot (16 B): open addressing, empty key = ~0
struct vmx_slot { uint64_t key; uint32_t start; uint32_t count; };

// cost tables on the device
struct vmx_tables {
    const float* extra; int32_t extra_n;
    const float* readgap_h; const float* readgap_r; const float* large_readgap;
    const double* log2cache; int32_t log2cache_n;
    const double* log2int;
    int32_t extra_arith_n;   // extra[g] == (float)(min(10, g*0.01) + g*0.001) for every g below this (checked on the host, vmx_capi.hip)
};

// extra[min(g, extra_n-1)] as a double without touching HBM where the table has a closed form: below extra_arith_n the entry is the
// linear part of :15371-15376 (two multiplies; identical to the table by the host's exhaustive check), at or above the last index it
// is the table's final value 36.0; only the logarithmic stretch in between is loaded.
__device__ __forceinline__ double vmx_extra_cost(const vmx_tables& tab, long long g) {
    if (g >= (long long)tab.extra_n - 1) return 36.0;
    if (g < (long long)tab.extra_arith_n) {
        const double dg = (double)(int)g;
        const double a = dg * 0.01;
        return (double)(float)((a < 10.0 ? a : 10.0) + dg * 0.001);
    }
    return (double)tab.extra[g];
}

// ---- gap geometry shared by the chain DPs (GC :24953-24984, LC :27418-27456) ----
// Both geometries without data-dependent branches. With rg = q_i - q_j - l_j, m = min(rg, 0) (the overlap, negated) and d = r_i - r_j the eight
// cases of :24953-24984 collapse to  readgap = rg - m,  bonus = l_i + m,  refgap = +-d + c  with a 32-bit c:
//   s_i = s_j = +1:  d - m - l_j          s_i = +1, s_j = -1:  d - m + 1
//   s_i = s_j = -1: -d - l_i - m          s_i = -1, s_j = +1:  d + l_i + m - 1 - l_j
// and the -mode asm fork's (mammap_asm.py:20660-20688: non_overlap_size = q_i - q_j, no +-1 between opposite strands) to the same two lines for
// equal strands and  d - m  /  d + l_i - m - l_j  for opposite ones. The lanes of a wavefront hold candidates on both strands, with and without
// overlap: as nested ifs the compiler emits an exec-mask region per case; as selects it is a dozen VALU instructions.
template <bool ASMV>
__device__ __forceinline__ void vmx_gap_geometry_sel(int qi, long long ri, int si, int li, int qj, long long rj, int sj, int lj,
                                                     long long& readgap, long long& refgap, long long& bonus) {
    const int rg = qi - qj - lj;                       // read positions and lengths: 32-bit
    const int m = rg < 0 ? rg : 0;
    readgap = (long long)(rg - m);
    bonus = (long long)(li + m);
    const long long d = ri - rj;
    const bool same = si == sj;
    int c; long long t;
    if (si == 1) { c = (same ? -lj : (ASMV ? 0 : 1)) - m; t = d; }
    else { c = same ? -li - m : li - lj + (ASMV ? -m : m - 1); t = same ? -d : d; }
    refgap = t + (long long)c;
}
// arguments of k_local_seed (L2): inputs, per-workgroup-slot scratch pools, outputs
struct vmx_lseed_args {
    const uint8_t* ocodes; const int64_t* roff;            // oriented read codes
    const uint8_t* ref; const int64_t* coff; int32_t nseq;  // reference codes + contig offsets (nseq+1)
    const vmx_anchor* guide_rows; const int32_t* guide_len; const int32_t* n_guides_used; const int64_t* aoff;
    int32_t n_reads, k, look_span, read_span, sort_by_start;
    const int32_t* order; int32_t* queue;                   // read indices longest first + the queue head (zero at launch)
    int64_t la_slot_len;                                    // local-anchor slot of a listed read = la_slot_len * VMX_LA_SLOT(len) rows at la_off[r]
    int32_t* head_pool; int32_t* next_pool;                 // HEAD[head_stride] (epoch-tagged entries) / NEXT[tpos_cap] per slot
    int64_t head_stride;                                    // heads per slot: 4^k, or 2^14 buckets when 14 < 2k <= 18 (k_local_seed)
    int32_t* epoch_pool;                                    // current epoch of every slot's head table (persists across launches)
    int32_t* sq_pool; int32_t* dst_pool;                    // hit_cap ints each
    int64_t* tpos_pool; int64_t tpos_cap;
    unsigned long long* dbg;    // optional phase timers (VMX_DBG=1)
    uint64_t* hkey2_pool;
    uint64_t* hkey_pool; int64_t* hval_pool; int32_t* goff_pool; int64_t hit_cap;
    int32_t* pcnt_pool; int64_t pcnt_cap; int32_t* pc2_pool; int64_t* stg_pool;
    uint64_t* gkey_pool; int32_t* gq_pool; int64_t* gr_pool; int64_t gkey_cap;
    vmx_anchor* la_rows; uint64_t* la_ekey; vmx_anchor* la_sorted; const int64_t* la_off; int32_t* la_cnt; int32_t* status;
    // collect_second_round_anchors (mammap_asm.py:22477-22756; null otherwise): unit r looks up the read positions [r_st[r], r_en[r] - k) of the
    // sequence at ocodes + rd_off[r] (length rd_len[r]: every unit of a contig shares it) and owns la_off[r + 1] - la_off[r] anchor rows
    const int64_t* rd_off; const int64_t* rd_len; const int32_t* r_st; const int32_t* r_en;
};

// k_local_seed_band (k_local_band.hip), one wavefront per read: read positions per chunk, hits per chunk (LDS tile; also the capacity of the
// plan's interval list), guide anchors staged per chunk, log2 of the chunk table's bucket heads, keys of the LDS sort region
// (4 * (heads + 2 * QC + 512) <= 8 * SORTK and HCAP <= SORTK: the table with its occupancy map and the sorted hits live there in turn)
#ifdef VMX_EMU
#define VMX_LB_QC 128               // emulator build: small chunks so that the CPU tests cross many chunk boundaries, cut chunks at the guide slice and overflow the hit tile
#define VMX_LB_HCAP 256
#define VMX_LB_GS 20
#define VMX_LB_NBLOG 8
#define VMX_LB_SORTK 512
#else
#ifndef VMX_LB_QC                    /* (tuning builds override the set: VMX_EXTRA_FLAGS="-DVMX_LB_QC=256 -DVMX_LB_HCAP=256 -DVMX_LB_SORTK=512 -DVMX_LB_NBLOG=8") */
#define VMX_LB_QC 512
#define VMX_LB_HCAP 512
#define VMX_LB_NBLOG 9
#define VMX_LB_SORTK 1024
#endif
#ifndef VMX_LB_GS
#define VMX_LB_GS 64
#endif
#endif
#ifndef VMX_LB_BMLOG
#define VMX_LB_BMLOG 14              /* log2 of the bits of the chunk table's occupancy map */
#endif
#ifndef VMX_LB_WAVES
#define VMX_LB_WAVES 2               /* waves per SIMD the register allocation of k_local_seed_band is held to */
#endif
#define VMX_LB_TABLE_U64 (((1 << VMX_LB_NBLOG) + 2 * VMX_LB_QC + (1 << (VMX_LB_BMLOG - 5))) / 2)          /* heads + entries + occupancy map, in 8-byte units */
#define VMX_LB_REGION_U64 (VMX_LB_SORTK > VMX_LB_TABLE_U64 ? VMX_LB_SORTK : VMX_LB_TABLE_U64)
#define VMX_LB_CQ_BYTES (128 + 2 * VMX_LB_HCAP > 2304 ? 128 + 2 * VMX_LB_HCAP : 2304)     /* candidate queue: one sweep (512 words) + the < 64 left over from the sweep before; later the run walk's marks + one index per hit */
#define VMX_LB_LDS_BYTES (8 * (VMX_LB_REGION_U64 + VMX_LB_HCAP) + VMX_LB_CQ_BYTES)
#define VMX_ED_WAVES 16              // max waves per workgroup of k_edit_distance (passes pipelined across them) = carry ring depth
#define VMX_ED_LONG 16384            // patterns longer than this (> 4 passes) go to the 16-wave launch
#define VMX_EDB_HW 768               // k_ed_banded: half width of the band in rows
#define VMX_EDB_MAXD 512             // k_ed_banded: |m - n| above this is not eligible (goes to the unbanded kernel)
#define VMX_EDB4_HW 320              // k_ed_banded4 (four problems per wave): half width of the band; 2*HW + MAXD + 64 <= 16 blocks
#define VMX_EDB4_MAXD 256            // k_ed_banded4: |m - n| above this goes to k_ed_banded
#ifndef VMX_LSEED_WAVES
#define VMX_LSEED_WAVES 6            // k_local_seed: waves per SIMD the register allocation is held to (80 VGPRs: 3 workgroups of 512 per CU)
#endif
#define VMX_SORT_LDS 4096           // uint64 keys sorted in LDS by vmx_block_sort_u64 (larger sorts run in HBM)
#ifdef VMX_EMU
#define VMX_SORT_LDS_BIG 8192       // emulator build: a small tile so that the CPU tests reach the tiled (multi-tile) sort
#else
#define VMX_SORT_LDS_BIG 16384      // k_cluster_big: keys per LDS tile (128 KB of the 160 KB per CU)
#endif
#define VM_READ_FASTPATH_DEV (-21)   // the reference would switch to a *_fast heuristic that is not built yet
#ifdef VMX_EMU
#define VMX_DP16_MAX 420              // emulator build: small switch point so that the CPU tests cover both layouts with small problems
#else
#define VMX_DP16_MAX 6000             // gap fill: problems with tl + ql <= this run two rows per lane in packed int16
#endif
#define VMX_DP16_OK(tl, ql) ((tl) + (ql) <= VMX_DP16_MAX)
// gap fill, small problems: four problems per wavefront, one per 16-lane row, 32-row stripes (two rows per lane), stripe width padded to
// a multiple of 16 steps so that the four rows refill their chunk registers on the same steps
#ifdef VMX_EMU
#define VMX_DP16X4_MAX 160
#else
#define VMX_DP16X4_MAX 1024       // larger problems run one per wavefront: a 700 x 700 problem almost never passes the band proof (its score falls ~0.8 per base
                                  // behind the all-match bound) and its full fill on one 16-lane row was a 5-10 ms tail of the second launch (3072 before)
#endif
#define VMX_DP16X4_OK(tl, ql) ((tl) + (ql) <= VMX_DP16X4_MAX)
#define VMX_X4_W(ql) ((((ql) + 31) + 15) & ~15)
// anti-diagonal band form of the small-problem fill (vmx_dp_ad.h): EIGHT problems per wavefront, one per 16-lane DPP row and 16-bit register
// half, each on a fixed band of 32 * ns diagonals d = j - i in [dlo, dlo + 32 ns) (ns = 1 .. VMX_AD_NS_MAX diagonal pairs per lane). dlo is
// even, so that all problems of a wave compute the same diagonal parity on the same step, and the band's margins below min(0, ql - tl) and
// above max(0, ql - tl) are as equal as that allows. vmx_ad_geom returns g: a path that leaves the band holds at least g inserted AND g deleted
// bases (0: the band cannot hold both corners). The result is kept only when vmx_ad_proven (vmx_dp_ad.h) shows every such path is worse.
#define VMX_AD_NS_MAX 4
__host__ __device__ static inline int vmx_ad_geom(int tl, int ql, int ns, int* dlo_out) {
    const int dl = ql - tl, lo = dl < 0 ? dl : 0, hi = dl > 0 ? dl : 0;
    const int slack = 32 * ns - (hi - lo + 1);
    if (slack < 0) return 0;
    int mb = slack / 2, dlo = lo - mb;
    if (dlo & 1) {
        if (mb + 1 <= slack) { ++mb; --dlo; }
        else if (mb >= 1) { --mb; ++dlo; }
        else return 0;
    }
    *dlo_out = dlo;
    const int ma = slack - mb;
    return (mb < ma ? mb : ma) + 1;
}
// what a path pays at least for leaving a band of margin g: g matches fewer and two gaps of g bases
__host__ __device__ static inline long long vmx_ad_margin(int g, int match, int o1, int e1, int o2, int e2) {
    const long long c1 = o1 + (long long)g * e1, c2 = o2 + (long long)g * e2;
    return (long long)match * g + 2 * (c1 < c2 ? c1 : c2);
}
// band width (diagonal pairs per lane) a problem is tried with: the narrowest whose margin is at least pct % of min(tl, ql) (a 10 %-error
// read loses ~0.7 per base against the all-match bound), the widest one if that still reaches pct_min %, 0 = not worth trying / impossible
__host__ __device__ static inline int vmx_ad_ns(int tl, int ql, int match, int o1, int e1, int o2, int e2, int pct, int pct_min) {
    if (tl <= 0 || ql <= 0) return 0;
    const int mn = tl < ql ? tl : ql;
    int dlo, g = 0;
    for (int ns = 1; ns <= VMX_AD_NS_MAX; ++ns) {
        g = vmx_ad_geom(tl, ql, ns, &dlo);
        if (g >= 1 && (g > mn || vmx_ad_margin(g, match, o1, e1, o2, e2) * 100 >= (long long)pct * mn)) return ns;
    }
    return (g >= 1 && vmx_ad_margin(g, match, o1, e1, o2, e2) * 100 >= (long long)pct_min * mn) ? VMX_AD_NS_MAX : 0;
}
#define VMX_AD_PCT_DEFAULT 100
#define VMX_AD_PCT_MIN_DEFAULT 65
#define VMX_AD_FLAG 16                 /* layout flag of a problem kept in the anti-diagonal layout: VMX_AD_FLAG + ns */
#define VMX_PK_FLAG 1                  /* layout flag of a small-class problem the second launch ran on the whole wave (packed two-rows-per-lane layout) */
#ifdef VMX_EMU
#define VMX_REDO_PK(tl, ql) ((tl) > 0 && (ql) > 0 && VMX_DP16X4_OK(tl, ql) && (tl) + (ql) >= 120)
#else
#ifndef VMX_REDO_PK_MIN
#define VMX_REDO_PK_MIN 384            /* redone problems of at least this perimeter take a whole wavefront each (the typical 270 x 270 problem included: the second
                                          launch holds ~3.5 k problems per batch, a quarter of the machine's wave slots — four per wave left it waiting 2.75 ms for 900 waves) */
#endif
#define VMX_REDO_PK(tl, ql) ((tl) > 0 && (ql) > 0 && VMX_DP16X4_OK(tl, ql) && (tl) + (ql) >= VMX_REDO_PK_MIN)
#endif
#define VMX_PK_TB_BYTES(tl, ql) ((int64_t)(((tl) + 127) / 128) * ((ql) + 127) * 128)
/* Traceback of the anti-diagonal layout: one 4-byte slot per lane and anti-diagonal, in 64-byte lines of VMX_AD_AB consecutive anti-diagonals x
   16 / VMX_AD_AB consecutive lanes. AB = 1 (rounds 2-3): a line = one anti-diagonal of a problem, the fill's row store is one line, but the
   traceback walk — nine steps in ten go down a diagonal, two anti-diagonals back in the same lane — fetched a new 64-byte sector per step
   (5.2 GB per step of the pipeline for ~80 MB of bytes used). With AB anti-diagonals per line a diagonal stretch reads a line per AB / 2 steps. */
#ifndef VMX_AD_AB
#define VMX_AD_AB 4     /* round 6: 4 (8 in rounds 4-5). With the step bound by the fill and the walk a narrow kernel that rides along, the fill's coalescing is worth more than the walk's
                           locality: ONT 15.35 / 15.68 -> 14.96 / 15.16 ms per step, HiFi unchanged (profiles/r06_u_traceback_line_blocking_ab.txt) */
#endif
static_assert(VMX_AD_AB == 1 || 64 / VMX_AD_AB <= 16, "a 64-byte line of the anti-diagonal traceback holds at most the 16 lanes of a row per anti-diagonal (1-byte slots): AB = 1, 4 or 8");
#define VMX_AD_TB_OFF(s, l) ((((size_t)(s) & ~(size_t)(VMX_AD_AB - 1)) << 6) + ((size_t)(s) & (VMX_AD_AB - 1)) * (64 / VMX_AD_AB) + ((size_t)(l) / (16 / VMX_AD_AB)) * 64 + ((size_t)(l) % (16 / VMX_AD_AB)) * 4)   /* byte offset of lane l's slot on anti-diagonal s (0-based), 4-byte slots */
#define VMX_AD_TB_BYTES(tl, ql) ((int64_t)(((tl) + (ql) + VMX_AD_AB - 1) & ~(VMX_AD_AB - 1)) * 64)
/* Round 6: the slot is as wide as the band needs — W = 1 byte per lane and anti-diagonal for ns = 1 (one cell per lane and step), 2 for ns = 2, 4 for ns = 3 / 4.
   A HiFi problem (ns = 1 under the mode-L rule) wrote 64 bytes per anti-diagonal of which 16 were cells: its fill ran at 0.38 of the VALU peak behind 1.95 TB/s
   of stores. Lines stay 64 bytes = VMX_AD_AB anti-diagonals x 64 / (AB W) lanes, so the traceback walk still reads a line per AB / 2 diagonal steps. */
#define VMX_AD_W(ns) ((ns) <= 0 ? 0 : ((ns) == 1 ? 1 : ((ns) == 2 ? 2 : 4)))
#define VMX_AD_TB_OFF_W(s, l, W) (((size_t)(s) / VMX_AD_AB) * (size_t)(VMX_AD_AB * 16 * (W)) + ((size_t)(l) / (64 / (VMX_AD_AB * (W)))) * 64 + ((size_t)(s) % VMX_AD_AB) * (64 / VMX_AD_AB) + ((size_t)(l) % (64 / (VMX_AD_AB * (W)))) * (W))
#define VMX_AD_TB_BYTES_W(tl, ql, W) ((int64_t)(((tl) + (ql) + VMX_AD_AB - 1) & ~(VMX_AD_AB - 1)) * 16 * (W))
#define VMX_X4_TB_BYTES(tl, ql) ((int64_t)(((tl) + 31) / 32) * VMX_X4_W(ql) * 32)
// traceback bytes of a problem. VMX_TB_BYTES: the full-matrix forms (k_gapfill_fill). VMX_TB_BYTES_NS: the batched path (k_gapfill_fill_ns), whose
// small problems get the anti-diagonal layout's (tl + ql) * 64 bytes only; the few that have to be filled again in full take
// VMX_REDO_TB_BYTES from a second pool, handed out with an atomic counter when the first launch queues them (vmx_dp_prob.tb_off < 0:
// offset -tb_off - 1 into that pool).
#define VMX_TB_BYTES(tl, ql) (((tl) > 0 && (ql) > 0) ? (VMX_DP16X4_OK(tl, ql) ? VMX_X4_TB_BYTES(tl, ql) : \
                              VMX_DP16_OK(tl, ql) ? VMX_PK_TB_BYTES(tl, ql) : (int64_t)(((tl) + 63) / 64) * ((ql) + 63) * 64) : 0)
#define VMX_TB_BYTES_NS(tl, ql) (((tl) > 0 && (ql) > 0 && VMX_DP16X4_OK(tl, ql)) ? VMX_AD_TB_BYTES(tl, ql) : VMX_TB_BYTES(tl, ql))
#define VMX_REDO_TB_BYTES(tl, ql) (VMX_REDO_PK(tl, ql) ? VMX_PK_TB_BYTES(tl, ql) : VMX_X4_TB_BYTES(tl, ql))
// (integer arithmetic: a select between the two pool pointers crashes this compiler's optimizer)
__device__ __forceinline__ uint8_t* vmx_tb_ptr(const uint8_t* tb_pool, const uint8_t* redo_pool, int64_t off) {
    const unsigned long long base = off >= 0 ? (unsigned long long)tb_pool : (unsigned long long)redo_pool;
    return (uint8_t*)(base + (unsigned long long)(off >= 0 ? off : -off - 1));
}
#define VMX_HEAD_THRESH (((int64_t)1 << 18) - 1)   /* traceback bytes above which a gap-fill problem is taken from the queue alone (k_size_order thresh): ~450 x 450 and up */
#define VMX_TB_CHUNK ((int64_t)4 << 30)    // gap fill: traceback bytes held at a time; a batch needing more runs fill + trace chunk by chunk. 4 GB (12 in round 3, 8 earlier
                                          // in round 4): FIVE batches in flight fit at hg38 size (296 of 309 GB) and the step does not notice the extra chunks (VMX_TB_CHUNK_GB: tuning knob)
#ifdef VMX_EMU
#define VMX_MAX_BATCH_BASES 20000          // emulator build: small limits so that the CPU tests split a batch
#define VMX_MAX_BATCH_READS 6
#else
#define VMX_MAX_BATCH_BASES ((int64_t)160 << 20)     // bases per internal sub-batch of vm_align_batch (a pipeline batch is ~60 M)
#define VMX_MAX_BATCH_READS 16384
#endif
#define VMX_LA_SLOT(len) ((len) / 2 + 4096)   // regular local-anchor slot of a read (rows); overflowing reads are re-run with 8x .. 4096x
#define VMX_SELECT_LDS 3072           // k_chain_select: largest LDS size class (17 B per anchor: S, P, S_arg, used flags); see vmx_launch_chain_select
#define VMX_LC_LDS_MAX_DEFAULT 13056   // reads with more local anchors than this run the chain DP on HBM-resident arrays (VMX_LC_LDS_MAX)
#define VMX_GC_LDS_MAX_DEFAULT 13056
#define VMX_CHAIN_LDS_MAX_SHARED 512   // ... what the chain kernels actually use (VMX_LC_LDS_MAX / VMX_GC_LDS_MAX override it): above it S / S_arg stay in HBM, where
                                      // one wave per read leaves room for eight waves per SIMD instead of one — a large LDS claim per wavefront starves the
                                      // kernel itself (alone: 52.5 -> 49.1 ms per batch) and the other batches' kernels of CUs (three in flight: 46.3 -> 42.5)
#define VMX_LC_BYTES_PER_ANCHOR 12   // S8 + SA4 (LDS bytes per anchor in k_chain_local; the anchors themselves are read from HBM in register blocks)
#define VMX_GC_BYTES_PER_ANCHOR 12   // S8 + SA4 (LDS bytes per anchor in k_chain_global; anchors and coverage are read from HBM in register blocks)

#endif
