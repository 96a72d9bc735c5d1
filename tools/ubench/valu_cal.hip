// valu_cal.hip — calibration of the VALU issue rate on gfx950: pure loops of the instructions the gap-fill kernel is
// made of (v_pk_add_i16, v_pk_max_i16, v_pk_ashrrev_i16, v_mov_b32 DPP row_shr / wave_shr, v_bfi_b32, v_add_u32) at 1..8 waves per SIMD.
// Prints one JSON object: per instruction and occupancy the wall time, the shader-clock cycles of one wave (s_memtime) and the derived
// wave-instructions per cycle per SIMD. Run it alone and under `rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE` (tools/valu_cal.py).
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/valu_cal tools/ubench/valu_cal.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP16(X) X X X X X X X X X X X X X X X X
#define BODY 64          // VALU instructions per loop iteration (4 x 16 independent chains of depth 1 per REP)

template <int OP>
__global__ void __launch_bounds__(256) k_loop(int iters, unsigned* out, long long* cyc) {
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    unsigned b = 0x00010001u * (blockIdx.x + 1);
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        // 8 independent registers, 8 instructions per line, 8 lines = 64 instructions; dependent only on the same register's previous value
#define LINE(INS) asm volatile(INS " %0, %0, %8\n\t" INS " %1, %1, %8\n\t" INS " %2, %2, %8\n\t" INS " %3, %3, %8\n\t" INS " %4, %4, %8\n\t" INS " %5, %5, %8\n\t" INS " %6, %6, %8\n\t" INS " %7, %7, %8" \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
#define LINE_DPP(CTRL) asm volatile("v_mov_b32_dpp %0, %0 " CTRL " row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %1 " CTRL " row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %2, %2 " CTRL " row_mask:0xf bank_mask:0xf\n\t" \
                                    "v_mov_b32_dpp %3, %3 " CTRL " row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %4, %4 " CTRL " row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %5, %5 " CTRL " row_mask:0xf bank_mask:0xf\n\t" \
                                    "v_mov_b32_dpp %6, %6 " CTRL " row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %7, %7 " CTRL " row_mask:0xf bank_mask:0xf" \
                                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
#define LINE3(INS) asm volatile(INS " %0, %0, %8, %8\n\t" INS " %1, %1, %8, %8\n\t" INS " %2, %2, %8, %8\n\t" INS " %3, %3, %8, %8\n\t" INS " %4, %4, %8, %8\n\t" INS " %5, %5, %8, %8\n\t" INS " %6, %6, %8, %8\n\t" INS " %7, %7, %8, %8" \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
#define LINEC(INS) asm volatile(INS " %0, %0, %8, vcc\n\t" INS " %1, %1, %8, vcc\n\t" INS " %2, %2, %8, vcc\n\t" INS " %3, %3, %8, vcc\n\t" INS " %4, %4, %8, vcc\n\t" INS " %5, %5, %8, vcc\n\t" INS " %6, %6, %8, vcc\n\t" INS " %7, %7, %8, vcc" \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
#define LINEK(INS) asm volatile(INS " vcc, %0, %8\n\t" INS " vcc, %1, %8\n\t" INS " vcc, %2, %8\n\t" INS " vcc, %3, %8\n\t" INS " vcc, %4, %8\n\t" INS " vcc, %5, %8\n\t" INS " vcc, %6, %8\n\t" INS " vcc, %7, %8" \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");
#define X8(L) L L L L L L L L
        if (OP == 7) { X8(LINE("v_sub_u32")) }
        if (OP == 8) { X8(LINE("v_and_b32")) }
        if (OP == 9) { X8(LINE("v_or_b32")) }
        if (OP == 10) { X8(LINE("v_xor_b32")) }
        if (OP == 11) { X8(LINE3("v_bfi_b32")) }
        if (OP == 12) { X8(LINEC("v_cndmask_b32")) }
        if (OP == 13) { X8(LINEK("v_cmp_gt_i32")) }
        if (OP == 14) { X8(LINE3("v_perm_b32")) }
        if (OP == 15) { X8(LINE3("v_max3_i32")) }
        if (OP == 16) { X8(LINE3("v_med3_i32")) }
        if (OP == 17) { X8(LINE("v_pk_sub_i16")) }
        if (OP == 18) { X8(LINE("v_pk_min_u16")) }
        if (OP == 19) { X8(LINE("v_pk_max_u16")) }
        if (OP == 20) { X8(LINE3("v_lshl_add_u32")) }
        if (OP == 21) { X8(LINE3("v_add3_u32")) }
        if (OP == 22) { X8(LINE("v_lshlrev_b32")) }
        if (OP == 23) { X8(LINE("v_ashrrev_i32")) }
        if (OP == 24) { X8(LINE("v_pk_ashrrev_i16")) }
        if (OP == 25) { X8(LINE("v_min_u32")) }
        if (OP == 26) { X8(LINE3("v_and_or_b32")) }
        if (OP == 27) { X8(LINE("v_pk_lshlrev_b16")) }
        if (OP == 28) { X8(LINE3("v_pk_mad_i16")) }
        if (OP == 0) { LINE("v_pk_add_i16") LINE("v_pk_add_i16") LINE("v_pk_add_i16") LINE("v_pk_add_i16") LINE("v_pk_add_i16") LINE("v_pk_add_i16") LINE("v_pk_add_i16") LINE("v_pk_add_i16") }
        if (OP == 1) { LINE("v_pk_max_i16") LINE("v_pk_max_i16") LINE("v_pk_max_i16") LINE("v_pk_max_i16") LINE("v_pk_max_i16") LINE("v_pk_max_i16") LINE("v_pk_max_i16") LINE("v_pk_max_i16") }
        if (OP == 2) { LINE("v_add_u32") LINE("v_add_u32") LINE("v_add_u32") LINE("v_add_u32") LINE("v_add_u32") LINE("v_add_u32") LINE("v_add_u32") LINE("v_add_u32") }
        if (OP == 3) { LINE_DPP("row_shr:1") LINE_DPP("row_shr:1") LINE_DPP("row_shr:1") LINE_DPP("row_shr:1") LINE_DPP("row_shr:1") LINE_DPP("row_shr:1") LINE_DPP("row_shr:1") LINE_DPP("row_shr:1") }
        if (OP == 4) { LINE_DPP("wave_shr:1") LINE_DPP("wave_shr:1") LINE_DPP("wave_shr:1") LINE_DPP("wave_shr:1") LINE_DPP("wave_shr:1") LINE_DPP("wave_shr:1") LINE_DPP("wave_shr:1") LINE_DPP("wave_shr:1") }
        if (OP == 5) { LINE("v_max_i32") LINE("v_max_i32") LINE("v_max_i32") LINE("v_max_i32") LINE("v_max_i32") LINE("v_max_i32") LINE("v_max_i32") LINE("v_max_i32") }
        if (OP == 6) { LINE("v_pk_sub_i16") LINE("v_pk_max_i16") LINE("v_pk_add_i16") LINE("v_pk_max_i16") LINE("v_pk_sub_i16") LINE("v_pk_max_i16") LINE("v_pk_add_i16") LINE("v_pk_max_i16") }
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}

typedef void (*kfn)(int, unsigned*, long long*);

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, simds = cus * 4;
    unsigned* out; long long* cyc;
    hipMalloc(&out, sizeof(unsigned) * (size_t)cus * 8 * 256 * 4); hipMalloc(&cyc, 8);
    const int NOP = 29;
    const char* names[NOP] = {"v_pk_add_i16", "v_pk_max_i16", "v_add_u32", "v_mov_b32_dpp row_shr:1", "v_mov_b32_dpp wave_shr:1", "v_max_i32", "pk add/sub/max mix", "v_sub_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_bfi_b32", "v_cndmask_b32", "v_cmp_gt_i32", "v_perm_b32", "v_max3_i32", "v_med3_i32", "v_pk_sub_i16", "v_pk_min_u16", "v_pk_max_u16", "v_lshl_add_u32", "v_add3_u32", "v_lshlrev_b32", "v_ashrrev_i32", "v_pk_ashrrev_i16", "v_min_u32", "v_and_or_b32", "v_pk_lshlrev_b16", "v_pk_mad_i16"};
    kfn fns[NOP] = {k_loop<0>, k_loop<1>, k_loop<2>, k_loop<3>, k_loop<4>, k_loop<5>, k_loop<6>, k_loop<7>, k_loop<8>, k_loop<9>, k_loop<10>, k_loop<11>, k_loop<12>, k_loop<13>, k_loop<14>, k_loop<15>, k_loop<16>, k_loop<17>, k_loop<18>, k_loop<19>, k_loop<20>, k_loop<21>, k_loop<22>, k_loop<23>, k_loop<24>, k_loop<25>, k_loop<26>, k_loop<27>, k_loop<28>};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("{\"device\": \"%s\", \"arch\": \"%s\", \"cus\": %d, \"clock_khz\": %d, \"iters\": %d, \"valu_insts_per_wave\": %lld, \"results\": [\n", p.name, p.gcnArchName, cus, p.clockRate, iters, (long long)iters * BODY);
    bool first = true;
    for (int op = 0; op < NOP; ++op)
        for (int wps = 1; wps <= 8; wps *= 2) {           // waves per SIMD: a 256-thread workgroup puts one wave on each of a CU's 4 SIMDs
            const int blocks = cus * wps;
            hipLaunchKernelGGL(fns[op], dim3(blocks), dim3(256), 0, 0, 16, out, cyc);     // warm-up
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(fns[op], dim3(blocks), dim3(256), 0, 0, iters, out, cyc);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            long long c = 0; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            const double insts_per_simd = (double)iters * BODY * wps;
            printf("%s {\"op\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"wave0_clock64_cycles\": %lld, \"wave_insts_per_simd\": %.0f, \"wave_insts_per_us_per_simd\": %.2f, "
                   "\"clock64_cycles_per_wave_inst_per_simd\": %.3f}",
                   first ? "" : ",\n", names[op], wps, ms, c, insts_per_simd, insts_per_simd / (ms * 1e3), (double)c / insts_per_simd);
            first = false;
        }
    printf("\n], \"simds\": %d}\n", simds);
    return 0;
}
