// valu_calib.hip -- measures what a vector-ALU wave-instruction COSTS a gfx950 SIMD, per instruction class, at 1 / 2 / 4
// resident waves per SIMD: the calibration of the VALU-issue roofline that bench.py and tools/pmc_derive.py report
// ("the 4-clk/instruction VALU peak is asserted, not calibrated").
//
//   hipcc -O3 --offload-arch=gfx950 tools/valu_calib.hip -o tools/valu_calib && tools/valu_calib > profiles/<tag>_valu_calib.json
//
// Method: one workgroup per CU (pinned there by a 96 KB LDS allocation: two would not fit in 160 KB) of 256 k threads = k
// waves on each of the CU's four SIMDs; every wave runs ITER iterations of a loop body of 64 INDEPENDENT instructions of
// one class (16 destination registers, each written four times per iteration: the dependency distance is 16 instructions);
// the loop overhead is three scalar instructions.  Cycles come from s_memtime (clock64) around the loop, the wall time of
// the same interval from the 100 MHz constant clock (wall_clock64): their ratio is the shader clock the loop ran at.
//   cost [cycles of one SIMD per wave-instruction] = cycles_of_the_slowest_wave / (k * 64 * ITER)
// Nothing here is timed by the host.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define HIPC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

#define R16(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7) I(8) I(9) I(10) I(11) I(12) I(13) I(14) I(15)
#define OUT16(T) "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])

struct Out { unsigned long long cycles, wall; };

// one kernel per class: T = register type, the instruction text takes %N (destination / accumulator), %16 and %17 (sources)
#define CALIB_KERNEL(NAME, T, INIT, TEXT, ...)                                                                               \
__global__ void __launch_bounds__(1024) NAME(Out* out, int iters, T seed)                                                    \
{                                                                                                                            \
    extern __shared__ char lds_pin[];                                                                                        \
    T r[16]; T a = seed + (T)INIT, b = seed;                                                                                 \
    for (int i = 0; i < 16; i++) r[i] = seed + (T)(threadIdx.x + i);                                                         \
    if (iters < 0) lds_pin[threadIdx.x] = 1;                                                                                 \
    __syncthreads();                                                                                                         \
    const unsigned long long w0 = wall_clock64(); const long long c0 = clock64();                                            \
    for (int it = 0; it < iters; it++) {                                                                                     \
        asm volatile(R16(TEXT) : OUT16(T) : "v"(a), "v"(b) : __VA_ARGS__);                                                       \
        asm volatile(R16(TEXT) : OUT16(T) : "v"(a), "v"(b) : __VA_ARGS__);                                                       \
        asm volatile(R16(TEXT) : OUT16(T) : "v"(a), "v"(b) : __VA_ARGS__);                                                       \
        asm volatile(R16(TEXT) : OUT16(T) : "v"(a), "v"(b) : __VA_ARGS__);                                                       \
    }                                                                                                                        \
    const long long c1 = clock64(); const unsigned long long w1 = wall_clock64();                                            \
    T s = r[0]; for (int i = 1; i < 16; i++) s = s + r[i];                                                                   \
    if (s == (T)123456789) out[0].cycles = 0;                                /* keep the registers alive */                  \
    if ((threadIdx.x & 63) == 0) { Out o; o.cycles = (unsigned long long)(c1 - c0); o.wall = w1 - w0;                        \
                                   out[1 + blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = o; }                       \
}

#define T_FMA_F32(n) "v_fma_f32 %" #n ", %16, %17, %" #n "\n\t"
#define T_MUL_F32(n) "v_mul_f32 %" #n ", %16, %" #n "\n\t"
#define T_ADD_F32(n) "v_add_f32 %" #n ", %16, %" #n "\n\t"
#define T_MIN3_F32(n) "v_min3_f32 %" #n ", %16, %17, %" #n "\n\t"
#define T_MAX_F32(n) "v_max_f32 %" #n ", %16, %" #n "\n\t"
#define T_RCP_F32(n) "v_rcp_f32 %" #n ", %" #n "\n\t"
#define T_FMA_F64(n) "v_fma_f64 %" #n ", %16, %17, %" #n "\n\t"
#define T_MUL_F64(n) "v_mul_f64 %" #n ", %16, %" #n "\n\t"
#define T_ADD_F64(n) "v_add_f64 %" #n ", %16, %" #n "\n\t"
#define T_RCP_F64(n) "v_rcp_f64 %" #n ", %" #n "\n\t"
#define T_CVT_F64_F32(n) "v_cvt_f32_f64 %" #n ", %16\n\t"
#define T_ADD_U32(n) "v_add_u32 %" #n ", %16, %" #n "\n\t"
#define T_MOV_B32(n) "v_mov_b32 %" #n ", %16\n\t"
#define T_CNDMASK(n) "v_cndmask_b32 %" #n ", %16, %" #n ", vcc\n\t"      /* (vcc: whatever the compiler left there) */
#define T_CMP_F32(n) "v_cmp_lt_f32 vcc, %16, %" #n "\n\t"
#define T_LSHL_B32(n) "v_lshlrev_b32 %" #n ", 1, %" #n "\n\t"
#define T_MUL_LO_U32(n) "v_mul_lo_u32 %" #n ", %16, %" #n "\n\t"
#define T_AND_B32(n) "v_and_b32 %" #n ", %16, %" #n "\n\t"
#define T_PK_FMA_F32(n) "v_pk_fma_f32 %" #n ", %16, %17, %" #n "\n\t"
#define T_CNDMASK_S(n) "v_cndmask_b32 %" #n ", %16, %" #n ", s[20:21]\n\t"
#define T_CMP_S(n) "v_cmp_lt_f32 s[20:21], %16, %" #n "\n\t"
#define T_MIN_F32(n) "v_min_f32 %" #n ", %16, %" #n "\n\t"
#define T_MAX3_F32(n) "v_max3_f32 %" #n ", %16, %17, %" #n "\n\t"
#define T_MIN_U32(n) "v_min_u32 %" #n ", %16, %" #n "\n\t"
#define T_SUB_U32(n) "v_sub_u32 %" #n ", %16, %" #n "\n\t"
#define T_OR_B32(n) "v_or_b32 %" #n ", %16, %" #n "\n\t"
#define T_LSHL_ADD_U32(n) "v_lshl_add_u32 %" #n ", %16, 2, %" #n "\n\t"
#define T_FMAC_F32(n) "v_fmac_f32 %" #n ", %16, %17\n\t"
#define T_READLANE(n) "v_readlane_b32 s20, %" #n ", 3\n\t"
#define T_CVT_F32_F64(n) "v_cvt_f32_f64 v127, %" #n "\n\t"
#define T_DIV_SCALE_F64(n) "v_div_scale_f64 %" #n ", vcc, %16, %17, %" #n "\n\t"
#define T_DIV_FMAS_F64(n) "v_div_fmas_f64 %" #n ", %16, %17, %" #n "\n\t"
#define T_DIV_FIXUP_F64(n) "v_div_fixup_f64 %" #n ", %16, %17, %" #n "\n\t"
#define T_CMP_F64(n) "v_cmp_lt_f64 vcc, %16, %" #n "\n\t"
#define T_MOV_B64(n) "v_mov_b64 %" #n ", %16\n\t"
#define T_LSHL_ADD_U64(n) "v_lshl_add_u64 %" #n ", %16, 3, %" #n "\n\t"
#define T_MAD_U64_U32(n) "v_mad_u64_u32 %" #n ", vcc, v0, v0, %" #n "\n\t"

CALIB_KERNEL(k_fma_f32, float, 0.5f, T_FMA_F32, "memory")
CALIB_KERNEL(k_mul_f32, float, 0.5f, T_MUL_F32, "memory")
CALIB_KERNEL(k_add_f32, float, 0.5f, T_ADD_F32, "memory")
CALIB_KERNEL(k_min3_f32, float, 0.5f, T_MIN3_F32, "memory")
CALIB_KERNEL(k_max_f32, float, 0.5f, T_MAX_F32, "memory")
CALIB_KERNEL(k_rcp_f32, float, 0.5f, T_RCP_F32, "memory")
CALIB_KERNEL(k_cmp_f32, float, 0.5f, T_CMP_F32, "vcc")
CALIB_KERNEL(k_fma_f64, double, 0.5, T_FMA_F64, "memory")
CALIB_KERNEL(k_mul_f64, double, 0.5, T_MUL_F64, "memory")
CALIB_KERNEL(k_add_f64, double, 0.5, T_ADD_F64, "memory")
CALIB_KERNEL(k_rcp_f64, double, 0.5, T_RCP_F64, "memory")
CALIB_KERNEL(k_pk_fma_f32, double, 0.5, T_PK_FMA_F32, "memory")
CALIB_KERNEL(k_add_u32, unsigned, 3u, T_ADD_U32, "memory")
CALIB_KERNEL(k_mov_b32, unsigned, 3u, T_MOV_B32, "memory")
CALIB_KERNEL(k_cndmask_b32, unsigned, 3u, T_CNDMASK, "memory")
CALIB_KERNEL(k_lshl_b32, unsigned, 3u, T_LSHL_B32, "memory")
CALIB_KERNEL(k_mul_lo_u32, unsigned, 3u, T_MUL_LO_U32, "memory")
CALIB_KERNEL(k_and_b32, unsigned, 3u, T_AND_B32, "memory")
CALIB_KERNEL(k_cndmask_sgpr, unsigned, 3u, T_CNDMASK_S, "s20", "s21")
CALIB_KERNEL(k_cmp_sgpr, float, 0.5f, T_CMP_S, "s20", "s21")
CALIB_KERNEL(k_min_f32, float, 0.5f, T_MIN_F32, "memory")
CALIB_KERNEL(k_max3_f32, float, 0.5f, T_MAX3_F32, "memory")
CALIB_KERNEL(k_min_u32, unsigned, 3u, T_MIN_U32, "memory")
CALIB_KERNEL(k_sub_u32, unsigned, 3u, T_SUB_U32, "memory")
CALIB_KERNEL(k_or_b32, unsigned, 3u, T_OR_B32, "memory")
CALIB_KERNEL(k_lshl_add_u32, unsigned, 3u, T_LSHL_ADD_U32, "memory")
CALIB_KERNEL(k_fmac_f32, float, 0.5f, T_FMAC_F32, "memory")
CALIB_KERNEL(k_readlane, unsigned, 3u, T_READLANE, "s20")
CALIB_KERNEL(k_cvt_f32_f64, double, 0.5, T_CVT_F32_F64, "v127")
CALIB_KERNEL(k_div_scale_f64, double, 0.5, T_DIV_SCALE_F64, "vcc")
CALIB_KERNEL(k_div_fmas_f64, double, 0.5, T_DIV_FMAS_F64, "memory")
CALIB_KERNEL(k_div_fixup_f64, double, 0.5, T_DIV_FIXUP_F64, "memory")
CALIB_KERNEL(k_cmp_f64, double, 0.5, T_CMP_F64, "vcc")
CALIB_KERNEL(k_mov_b64, double, 0.5, T_MOV_B64, "memory")
CALIB_KERNEL(k_lshl_add_u64, unsigned long long, 3ull, T_LSHL_ADD_U64, "memory")
CALIB_KERNEL(k_mad_u64_u32, unsigned long long, 3ull, T_MAD_U64_U32, "vcc")

template <typename K, typename T>
static void run(const char* name, K kernel, T seed, int n_cu, int iters, bool last)
{
    printf("  \"%s\": {", name);
    for (int k = 1; k <= 4; k *= 2) {
        const int threads = 256 * k, waves = n_cu * threads / 64;
        Out* d; HIPC(hipMalloc(&d, sizeof(Out) * (1 + waves))); HIPC(hipMemset(d, 0, sizeof(Out) * (1 + waves)));
        const size_t lds = 96 * 1024;                                   // one workgroup per CU
        HIPC(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        kernel<<<n_cu, threads, lds>>>(d, 64, seed);                    // warm-up (clocks up, code in the instruction cache)
        kernel<<<n_cu, threads, lds>>>(d, iters, seed);
        HIPC(hipDeviceSynchronize());
        std::vector<Out> h(1 + waves); HIPC(hipMemcpy(h.data(), d, sizeof(Out) * (1 + waves), hipMemcpyDeviceToHost));
        unsigned long long cmax = 0, wmax = 0; double csum = 0;
        for (int w = 1; w <= waves; w++) { if (h[w].cycles > cmax) { cmax = h[w].cycles; wmax = h[w].wall; } csum += (double)h[w].cycles; }
        const double insts = 64.0 * iters;
        const double mhz = wmax ? (double)cmax / ((double)wmax / 100.0) : 0.0;      // wall_clock64 ticks at 100 MHz
        printf("%s\"waves_per_simd_%d\": {\"cycles_per_wave_inst_per_simd\": %.4f, \"mean_wave\": %.4f, \"clock_MHz\": %.1f}", k == 1 ? "" : ", ",
               k, (double)cmax / (k * insts), csum / waves / (k * insts), mhz);
        HIPC(hipFree(d));
    }
    printf("}%s\n", last ? "" : ",");
}

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 4096;
    hipDeviceProp_t p; HIPC(hipGetDeviceProperties(&p, 0));
    const int n_cu = p.multiProcessorCount;
    printf("{\n \"what\": \"cost of one VALU wave-instruction to one SIMD [cycles], by class and resident waves per SIMD; 64 independent instructions x %d iterations per wave, one workgroup of 256 k threads per CU (tools/valu_calib.hip)\",\n", iters);
    printf(" \"device\": \"%s\", \"gcn_arch\": \"%s\", \"compute_units\": %d, \"clock_rate_kHz_reported\": %d,\n \"classes\": {\n", p.name, p.gcnArchName, n_cu, p.clockRate);
    run("v_fma_f32", k_fma_f32, 1.0f, n_cu, iters, false);
    run("v_mul_f32", k_mul_f32, 1.0f, n_cu, iters, false);
    run("v_add_f32", k_add_f32, 1.0f, n_cu, iters, false);
    run("v_min3_f32", k_min3_f32, 1.0f, n_cu, iters, false);
    run("v_max_f32", k_max_f32, 1.0f, n_cu, iters, false);
    run("v_cmp_lt_f32", k_cmp_f32, 1.0f, n_cu, iters, false);
    run("v_rcp_f32", k_rcp_f32, 1.0f, n_cu, iters, false);
    run("v_pk_fma_f32", k_pk_fma_f32, 1.0, n_cu, iters, false);
    run("v_fma_f64", k_fma_f64, 1.0, n_cu, iters, false);
    run("v_mul_f64", k_mul_f64, 1.0, n_cu, iters, false);
    run("v_add_f64", k_add_f64, 1.0, n_cu, iters, false);
    run("v_rcp_f64", k_rcp_f64, 1.0, n_cu, iters, false);
    run("v_add_u32", k_add_u32, 1u, n_cu, iters, false);
    run("v_and_b32", k_and_b32, 1u, n_cu, iters, false);
    run("v_lshlrev_b32", k_lshl_b32, 1u, n_cu, iters, false);
    run("v_mov_b32", k_mov_b32, 1u, n_cu, iters, false);
    run("v_cndmask_b32", k_cndmask_b32, 1u, n_cu, iters, false);
    run("v_mul_lo_u32", k_mul_lo_u32, 1u, n_cu, iters, false);
    run("v_cndmask_b32_sgpr_mask", k_cndmask_sgpr, 1u, n_cu, iters, false);
    run("v_cmp_lt_f32_to_sgpr", k_cmp_sgpr, 1.0f, n_cu, iters, false);
    run("v_min_f32", k_min_f32, 1.0f, n_cu, iters, false);
    run("v_max3_f32", k_max3_f32, 1.0f, n_cu, iters, false);
    run("v_min_u32", k_min_u32, 1u, n_cu, iters, false);
    run("v_sub_u32", k_sub_u32, 1u, n_cu, iters, false);
    run("v_or_b32", k_or_b32, 1u, n_cu, iters, false);
    run("v_lshl_add_u32", k_lshl_add_u32, 1u, n_cu, iters, false);
    run("v_fmac_f32", k_fmac_f32, 1.0f, n_cu, iters, false);
    run("v_readlane_b32", k_readlane, 1u, n_cu, iters, false);
    run("v_cvt_f32_f64", k_cvt_f32_f64, 1.0, n_cu, iters, false);
    run("v_div_scale_f64", k_div_scale_f64, 1.0, n_cu, iters, false);
    run("v_div_fmas_f64", k_div_fmas_f64, 1.0, n_cu, iters, false);
    run("v_div_fixup_f64", k_div_fixup_f64, 1.0, n_cu, iters, false);
    run("v_cmp_lt_f64", k_cmp_f64, 1.0, n_cu, iters, false);
    run("v_mov_b64", k_mov_b64, 1.0, n_cu, iters, false);
    run("v_lshl_add_u64", k_lshl_add_u64, 1ull, n_cu, iters, false);
    run("v_mad_u64_u32", k_mad_u64_u32, 1ull, n_cu, iters, true);
    printf(" }\n}\n");
    return 0;
}
