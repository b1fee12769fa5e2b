// Issue cost of the vector instructions the ORB kernels are made of, measured the way peaks_probe measures the
// XOR + popcount rate: every SIMD full of waves, eight independent register chains per wave, nothing but the
// instruction under test in the loop.  Output: cycles per wave64 instruction per SIMD at the reported clock
// (4 = full rate, 16 = quarter rate).  Used to decide which forms to avoid.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(INS, TAIL)                                                                                                  \
    INS " %0, %0" TAIL "\n\t" INS " %1, %1" TAIL "\n\t" INS " %2, %2" TAIL "\n\t" INS " %3, %3" TAIL "\n\t"             \
    INS " %4, %4" TAIL "\n\t" INS " %5, %5" TAIL "\n\t" INS " %6, %6" TAIL "\n\t" INS " %7, %7" TAIL "\n\t"

#define KERNEL(NAME, T, INS, TAIL, ...)                                                                                 \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, int iters)                                                \
    {                                                                                                                    \
        T a0 = (T)(threadIdx.x + 1), a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6,      \
          a7 = a0 + 7;                                                                                                   \
        T b = (T)(threadIdx.x * 7 + 3), c = (T)(threadIdx.x * 5 + 1);                                                    \
        for (int i = 0; i < iters; i++) {                                                                                \
            asm volatile(REP8(INS, TAIL) REP8(INS, TAIL) REP8(INS, TAIL) REP8(INS, TAIL)                                 \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                \
                         : "v"(b), "v"(c)                                                                                \
                         : __VA_ARGS__);                                                                                        \
        }                                                                                                                \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == (T)0x12345) out[0] = 1;                                            \
    }

KERNEL(k_xor, uint32_t, "v_xor_b32", ", %8", "memory")
KERNEL(k_add, uint32_t, "v_add_u32", ", %8", "memory")
KERNEL(k_bcnt, uint32_t, "v_bcnt_u32_b32", ", %8", "memory")
KERNEL(k_mul_lo, uint32_t, "v_mul_lo_u32", ", %8", "memory")
KERNEL(k_mul_hi, uint32_t, "v_mul_hi_u32", ", %8", "memory")
KERNEL(k_mul_u24, uint32_t, "v_mul_u32_u24", ", %8", "memory")
KERNEL(k_mad_u24, uint32_t, "v_mad_u32_u24", ", %8, %9", "memory")
KERNEL(k_mad_i24, uint32_t, "v_mad_i32_i24", ", %8, %9", "memory")
KERNEL(k_dot4, uint32_t, "v_dot4_u32_u8", ", %8, %9", "memory")
KERNEL(k_dot2, uint32_t, "v_dot2_u32_u16", ", %8, %9", "memory")
KERNEL(k_perm, uint32_t, "v_perm_b32", ", %8, %9", "memory")
KERNEL(k_alignbyte, uint32_t, "v_alignbyte_b32", ", %8, %9", "memory")
KERNEL(k_pk_max_i16, uint32_t, "v_pk_max_i16", ", %8", "memory")
KERNEL(k_pk_sub_i16, uint32_t, "v_pk_sub_i16", ", %8", "memory")
KERNEL(k_min3, uint32_t, "v_min3_u32", ", %8, %9", "memory")
KERNEL(k_med3, uint32_t, "v_med3_u32", ", %8, %9", "memory")
KERNEL(k_add3, uint32_t, "v_add3_u32", ", %8, %9", "memory")
KERNEL(k_lshl_add, uint32_t, "v_lshl_add_u32", ", 2, %8", "memory")
KERNEL(k_bfe, uint32_t, "v_bfe_u32", ", 4, 16", "memory")
KERNEL(k_mul_lo_u16, uint32_t, "v_mul_lo_u16", ", %8", "memory")
KERNEL(k_mul_f32, uint32_t, "v_mul_f32", ", %8", "memory")
KERNEL(k_fma_f32, uint32_t, "v_fma_f32", ", %8, %9", "memory")
KERNEL(k_cvt_f32_i32, uint32_t, "v_cvt_f32_i32", "", "memory")
KERNEL(k_cvt_i32_f32, uint32_t, "v_cvt_i32_f32", "", "memory")
KERNEL(k_rndne, uint32_t, "v_rndne_f32", "", "memory")
KERNEL(k_rcp, uint32_t, "v_rcp_f32", "", "memory")
KERNEL(k_mul_f64, uint64_t, "v_mul_f64", ", %8", "memory")
KERNEL(k_fma_f64, uint64_t, "v_fma_f64", ", %8, %9", "memory")
KERNEL(k_pk_mul_f32, uint64_t, "v_pk_mul_f32", ", %8", "memory")
KERNEL(k_pk_add_f32, uint64_t, "v_pk_add_f32", ", %8", "memory")
KERNEL(k_lshl_add_u64, uint64_t, "v_lshl_add_u64", ", 1, %8", "memory")


KERNEL(k_min_u32, uint32_t, "v_min_u32", ", %8", "memory")
KERNEL(k_max_u32, uint32_t, "v_max_u32", ", %8", "memory")
KERNEL(k_max_i32, uint32_t, "v_max_i32", ", %8", "memory")
KERNEL(k_sub_u32, uint32_t, "v_sub_u32", ", %8", "memory")
KERNEL(k_and, uint32_t, "v_and_b32", ", %8", "memory")
KERNEL(k_or, uint32_t, "v_or_b32", ", %8", "memory")
KERNEL(k_lshlrev, uint32_t, "v_lshlrev_b32", ", %8", "memory")
KERNEL(k_lshrrev, uint32_t, "v_lshrrev_b32", ", %8", "memory")
KERNEL(k_ashrrev, uint32_t, "v_ashrrev_i32", ", %8", "memory")
KERNEL(k_mov, uint32_t, "v_mov_b32", "", "memory")
KERNEL(k_cndmask, uint32_t, "v_cndmask_b32", ", %8, vcc", "memory")
KERNEL(k_max_i16, uint32_t, "v_max_i16", ", %8", "memory")
KERNEL(k_min_u16, uint32_t, "v_min_u16", ", %8", "memory")
KERNEL(k_add_u16, uint32_t, "v_add_u16", ", %8", "memory")
KERNEL(k_sub_u16, uint32_t, "v_sub_u16", ", %8", "memory")
KERNEL(k_sad_u8, uint32_t, "v_sad_u8", ", %8, %9", "memory")
KERNEL(k_and_or, uint32_t, "v_and_or_b32", ", %8, %9", "memory")
KERNEL(k_or3, uint32_t, "v_or3_b32", ", %8, %9", "memory")
KERNEL(k_lshl_or, uint32_t, "v_lshl_or_b32", ", 8, %9", "memory")
KERNEL(k_cvt_ubyte0, uint32_t, "v_cvt_f32_ubyte0", "", "memory")
KERNEL(k_add_f32, uint32_t, "v_add_f32", ", %8", "memory")
KERNEL(k_max_f32, uint32_t, "v_max_f32", ", %8", "memory")
KERNEL(k_min_f32, uint32_t, "v_min_f32", ", %8", "memory")
KERNEL(k_max3_f32, uint32_t, "v_max3_f32", ", %8, %9", "memory")
KERNEL(k_med3_f32, uint32_t, "v_med3_f32", ", %8, %9", "memory")
KERNEL(k_pk_add_u16, uint32_t, "v_pk_add_u16", ", %8", "memory")
KERNEL(k_pk_min_u16, uint32_t, "v_pk_min_u16", ", %8", "memory")
KERNEL(k_pk_mad_u16, uint32_t, "v_pk_mad_u16", ", %8, %9", "memory")
KERNEL(k_pk_fma_f32, uint64_t, "v_pk_fma_f32", ", %8, %9", "memory")
KERNEL(k_pk_max_f16, uint32_t, "v_pk_max_f16", ", %8", "memory")
KERNEL(k_pk_fma_f16, uint32_t, "v_pk_fma_f16", ", %8, %9", "memory")
KERNEL(k_max_f16, uint32_t, "v_max_f16", ", %8", "memory")
KERNEL(k_min3_u16, uint32_t, "v_min3_u16", ", %8, %9", "memory")
KERNEL(k_max3_u16, uint32_t, "v_max3_u16", ", %8, %9", "memory")
KERNEL(k_max_u16, uint32_t, "v_max_u16", ", %8", "memory")
KERNEL(k_mad_u16, uint32_t, "v_mad_u16", ", %8, %9", "memory")
KERNEL(k_add_sdwa, uint32_t, "v_add_u32_sdwa", ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD", "memory")
KERNEL(k_add_dpp, uint32_t, "v_add_u32_dpp", ", %8 row_shr:1 row_mask:0xf bank_mask:0xf", "memory")
KERNEL(k_cmp_cnd, uint32_t, "v_cmp_gt_u32 vcc, %8, %9\n\tv_cndmask_b32", ", %8, vcc", "vcc", "memory")

// d64 = s0 * s1 + s2_64: destination and addend are the chain, carry-out goes to vcc
#define REP8M(INS)                                                                                                       \
    INS " %0, vcc, %8, %9, %0\n\t" INS " %1, vcc, %8, %9, %1\n\t" INS " %2, vcc, %8, %9, %2\n\t"                         \
    INS " %3, vcc, %8, %9, %3\n\t" INS " %4, vcc, %8, %9, %4\n\t" INS " %5, vcc, %8, %9, %5\n\t"                         \
    INS " %6, vcc, %8, %9, %6\n\t" INS " %7, vcc, %8, %9, %7\n\t"
#define KERNEL_MAD64(NAME, INS)                                                                                          \
    __global__ __launch_bounds__(256) void NAME(uint32_t *out, int iters)                                                \
    {                                                                                                                    \
        uint64_t a0 = threadIdx.x + 1, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6,     \
                 a7 = a0 + 7;                                                                                            \
        uint32_t b = threadIdx.x * 7 + 3, c = threadIdx.x * 5 + 1;                                                       \
        for (int i = 0; i < iters; i++) {                                                                                \
            asm volatile(REP8M(INS) REP8M(INS) REP8M(INS) REP8M(INS)                                                     \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                \
                         : "v"(b), "v"(c)                                                                                \
                         : "vcc", "memory");                                                                             \
        }                                                                                                                \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345) out[0] = 1;                                                \
    }
KERNEL_MAD64(k_mad_u64_u32, "v_mad_u64_u32")
KERNEL_MAD64(k_mad_i64_i32, "v_mad_i64_i32")

typedef void (*kern_t)(uint32_t *, int);
struct entry { const char *name; kern_t fn; };

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    uint32_t *o;
    hipMalloc((void **)&o, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const entry tab[] = {
        {"v_xor_b32", k_xor}, {"v_add_u32", k_add}, {"v_bcnt_u32_b32", k_bcnt}, {"v_mul_lo_u32", k_mul_lo}, {"v_mul_hi_u32", k_mul_hi},
        {"v_mul_u32_u24", k_mul_u24}, {"v_mad_u32_u24", k_mad_u24}, {"v_mad_i32_i24", k_mad_i24}, {"v_mad_u64_u32", k_mad_u64_u32},
        {"v_mad_i64_i32", k_mad_i64_i32}, {"v_dot4_u32_u8", k_dot4}, {"v_dot2_u32_u16", k_dot2}, {"v_perm_b32", k_perm},
        {"v_alignbyte_b32", k_alignbyte}, {"v_pk_max_i16", k_pk_max_i16}, {"v_pk_sub_i16", k_pk_sub_i16}, {"v_min3_u32", k_min3},
        {"v_med3_u32", k_med3}, {"v_add3_u32", k_add3}, {"v_lshl_add_u32", k_lshl_add}, {"v_bfe_u32", k_bfe},
        {"v_mul_lo_u16", k_mul_lo_u16}, {"v_mul_f32", k_mul_f32}, {"v_fma_f32", k_fma_f32}, {"v_cvt_f32_i32", k_cvt_f32_i32},
        {"v_cvt_i32_f32", k_cvt_i32_f32}, {"v_rndne_f32", k_rndne}, {"v_rcp_f32", k_rcp}, {"v_mul_f64", k_mul_f64},
        {"v_fma_f64", k_fma_f64}, {"v_pk_mul_f32", k_pk_mul_f32}, {"v_pk_add_f32", k_pk_add_f32}, {"v_lshl_add_u64", k_lshl_add_u64},
    {"v_min_u32", k_min_u32}, {"v_max_u32", k_max_u32}, {"v_max_i32", k_max_i32}, {"v_sub_u32", k_sub_u32}, {"v_and_b32", k_and},
        {"v_or_b32", k_or}, {"v_lshlrev_b32", k_lshlrev}, {"v_lshrrev_b32", k_lshrrev}, {"v_ashrrev_i32", k_ashrrev}, {"v_mov_b32", k_mov},
        {"v_cndmask_b32", k_cndmask}, {"v_max_i16", k_max_i16}, {"v_min_u16", k_min_u16}, {"v_add_u16", k_add_u16}, {"v_sub_u16", k_sub_u16},
        {"v_sad_u8", k_sad_u8}, {"v_and_or_b32", k_and_or}, {"v_or3_b32", k_or3}, {"v_lshl_or_b32", k_lshl_or},
        {"v_cvt_f32_ubyte0", k_cvt_ubyte0}, {"v_add_f32", k_add_f32}, {"v_max_f32", k_max_f32}, {"v_min_f32", k_min_f32},
        {"v_max3_f32", k_max3_f32}, {"v_med3_f32", k_med3_f32}, {"v_pk_add_u16", k_pk_add_u16}, {"v_pk_min_u16", k_pk_min_u16},
        {"v_pk_mad_u16", k_pk_mad_u16}, {"v_pk_fma_f32", k_pk_fma_f32}, {"v_pk_max_f16", k_pk_max_f16}, {"v_pk_fma_f16", k_pk_fma_f16},
        {"v_max_f16", k_max_f16}, {"v_min3_u16", k_min3_u16}, {"v_max3_u16", k_max3_u16}, {"v_max_u16", k_max_u16}, {"v_mad_u16", k_mad_u16}, {"v_add_u32_sdwa", k_add_sdwa}, {"v_add_u32_dpp", k_add_dpp}, {"v_cmp+v_cndmask (pair)", k_cmp_cnd},
    };
    const int iters = 2000, blocks = p.multiProcessorCount * 8;
    const double simds = (double)p.multiProcessorCount * 4, clock = p.clockRate * 1e3;
    printf("{\"gcn_arch\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"cycles_per_wave64_instruction_per_simd\": {", p.gcnArchName,
           p.multiProcessorCount, p.clockRate / 1000);
    for (size_t t = 0; t < sizeof(tab) / sizeof(tab[0]); t++) {
        float best = 1e9;
        for (int r = 0; r < 4; r++) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(tab[t].fn, dim3(blocks), dim3(256), 0, 0, o, iters);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        const double wave_instr = (double)blocks * 4 * iters * 32;
        printf("%s\"%s\": %.2f", t ? ", " : "", tab[t].name, simds * clock / (wave_instr / (best * 1e-3)));
    }
    printf("}}\n");
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
