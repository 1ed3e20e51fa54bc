// cndmask_probe.hip -- why did tools/valu_calib.hip see 23 cycles per v_cndmask_b32 with VCC as the mask (4.2 with an SGPR pair)?
// Variants: the mask register (vcc / sgpr pair) x the mask VALUE (0, all ones, alternating lanes, low half) x whether a
// v_cmp writes the mask inside the loop.  One workgroup of 512 threads per CU (2 waves per SIMD), 64 instructions x ITER.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define HIPC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
#define R16(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7) I(8) I(9) I(10) I(11) I(12) I(13) I(14) I(15)
#define OUT16 "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15])
struct Out { unsigned long long cycles; };
#define C_VCC(n) "v_cndmask_b32 %" #n ", %16, %" #n ", vcc\n\t"
#define C_SGPR(n) "v_cndmask_b32 %" #n ", %16, %" #n ", s[20:21]\n\t"
#define C_CMPVCC(n) "v_cmp_lt_u32 vcc, %17, %" #n "\n\tv_cndmask_b32 %" #n ", %16, %" #n ", vcc\n\t"
#define C_CMPSGPR(n) "v_cmp_lt_u32 s[20:21], %17, %" #n "\n\tv_cndmask_b32 %" #n ", %16, %" #n ", s[20:21]\n\t"
#define C_CMP4VCC(n) "v_cmp_lt_u32 vcc, %17, %" #n "\n\ts_nop 1\n\tv_cndmask_b32 %" #n ", %16, %" #n ", vcc\n\tv_cndmask_b32 %16, %" #n ", %16, vcc\n\tv_cndmask_b32 %" #n ", %16, %" #n ", vcc\n\tv_cndmask_b32 %16, %" #n ", %16, vcc\n\t"
#define C_CMP4SGPR(n) "v_cmp_lt_u32 s[20:21], %17, %" #n "\n\ts_nop 1\n\tv_cndmask_b32 %" #n ", %16, %" #n ", s[20:21]\n\tv_cndmask_b32 %16, %" #n ", %16, s[20:21]\n\tv_cndmask_b32 %" #n ", %16, %" #n ", s[20:21]\n\tv_cndmask_b32 %16, %" #n ", %16, s[20:21]\n\t"
#define KERNEL(NAME, TEXT, PER)                                                                                               \
__global__ void __launch_bounds__(1024) NAME(Out* out, int iters, unsigned long long mask)                                     \
{                                                                                                                              \
    extern __shared__ char pin[];                                                                                              \
    unsigned r[16]; unsigned a = threadIdx.x * 7u + 1u, b = threadIdx.x ^ 21u;                                                 \
    for (int i = 0; i < 16; i++) r[i] = threadIdx.x * 3u + i;                                                                   \
    if (iters < 0) pin[threadIdx.x] = 1;                                                                                       \
    asm volatile("s_mov_b64 vcc, %0\n\ts_mov_b64 s[20:21], %0" :: "s"(mask) : "vcc", "s20", "s21");                              \
    __syncthreads();                                                                                                           \
    const long long c0 = clock64();                                                                                            \
    for (int it = 0; it < iters; it++) {                                                                                       \
        asm volatile(R16(TEXT) : OUT16, "+v"(a) : "v"(b) : "vcc", "s20", "s21");                                              \
        asm volatile(R16(TEXT) : OUT16, "+v"(a) : "v"(b) : "vcc", "s20", "s21");                                              \
    }                                                                                                                          \
    const long long c1 = clock64();                                                                                            \
    unsigned s = a; for (int i = 0; i < 16; i++) s += r[i];                                                                     \
    if (s == 123456789u) out[0].cycles = 0;                                                                                    \
    if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)].cycles = (unsigned long long)(c1 - c0);   \
}
KERNEL(k_vcc, C_VCC, 1)
KERNEL(k_sgpr, C_SGPR, 1)
KERNEL(k_cmp_vcc, C_CMPVCC, 2)
KERNEL(k_cmp_sgpr, C_CMPSGPR, 2)
KERNEL(k_cmp4_vcc, C_CMP4VCC, 5)
KERNEL(k_cmp4_sgpr, C_CMP4SGPR, 5)
template <typename K> static void run(const char* name, K kernel, int per, unsigned long long mask, int n_cu, int iters)
{
    const int k = 2, threads = 256 * k, waves = n_cu * threads / 64;
    Out* d; HIPC(hipMalloc(&d, sizeof(Out) * (1 + waves))); HIPC(hipMemset(d, 0, sizeof(Out) * (1 + waves)));
    HIPC(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    kernel<<<n_cu, threads, 96 * 1024>>>(d, 64, mask); kernel<<<n_cu, threads, 96 * 1024>>>(d, iters, mask);
    HIPC(hipDeviceSynchronize());
    std::vector<Out> h(1 + waves); HIPC(hipMemcpy(h.data(), d, sizeof(Out) * (1 + waves), hipMemcpyDeviceToHost));
    unsigned long long cmax = 0; double sum = 0; for (int w = 1; w <= waves; w++) { if (h[w].cycles > cmax) cmax = h[w].cycles; sum += (double)h[w].cycles; }
    const double insts = 32.0 * per * iters;
    printf("%-12s mask %016llx : %.2f cycles per VALU instruction per SIMD (slowest wave), %.2f (mean wave)\n", name, mask, (double)cmax / (k * insts), sum / waves / (k * insts));
    HIPC(hipFree(d));
}
int main()
{
    hipDeviceProp_t p; HIPC(hipGetDeviceProperties(&p, 0)); const int n_cu = p.multiProcessorCount, iters = 4096;
    const unsigned long long masks[] = {0ULL, ~0ULL, 0x5555555555555555ULL, 0x00000000ffffffffULL, 0x0123456789abcdefULL};
    for (unsigned long long m : masks) { run("vcc", k_vcc, 1, m, n_cu, iters); run("sgpr", k_sgpr, 1, m, n_cu, iters); }
    run("cmp+vcc", k_cmp_vcc, 2, 0, n_cu, iters); run("cmp+sgpr", k_cmp_sgpr, 2, 0, n_cu, iters);
    run("cmp+4 vcc", k_cmp4_vcc, 5, 0, n_cu, iters); run("cmp+4 sgpr", k_cmp4_sgpr, 5, 0, n_cu, iters);
    return 0;
}
