// Micro-benchmark: issue rate of the integer VALU ops K1's hash is made of (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates ; run on the GPU box.
// Reports lane-ops/s and cycles per wave-instruction per SIMD, using the shader clock
// measured by s_memtime over the same kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 2048
template <int OP> __global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed, unsigned long long *cyc)
{
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 ^ 0x1234, a3 = a0 + 77, a4 = a0 * 7, a5 = ~a0, a6 = a0 >> 3, a7 = a0 << 5;
    const uint32_t c = 0xcc9e2d51u | seed;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITER; ++i) {
#define STEP(r)                                                                                          \
    if (OP == 0) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(r) : "v"(c));                             \
    if (OP == 1) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(r) : "v"(c));                            \
    if (OP == 2) asm volatile("v_lshl_add_u32 %0, %0, 2, %0" : "+v"(r));                                 \
    if (OP == 3) asm volatile("v_alignbit_b32 %0, %0, %0, 19" : "+v"(r));                                \
    if (OP == 4) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(r) : "v"(c));                                \
    if (OP == 5) { uint64_t t; asm volatile("v_mad_u64_u32 %0, vcc, %1, 5, %2" : "=v"(t) : "v"(r), "v"((uint64_t)c) : "vcc"); r = (uint32_t)t; } \
    if (OP == 6) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(r) : "v"(c));                             \
    if (OP == 7) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(r) : "v"(c));                        \
    if (OP == 8) asm volatile("v_add3_u32 %0, %0, %1, %0" : "+v"(r) : "v"(c));                           \
    if (OP == 9) asm volatile("v_xor_b32_sdwa %0, %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "+v"(r)); \
    if (OP == 10) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r) : "v"(c));                               \
    if (OP == 11) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(r));                                     \
    if (OP == 12) asm volatile("v_lshrrev_b32 %0, 13, %0" : "+v"(r));                                    \
    if (OP == 13) asm volatile("v_and_b32 %0, %0, %1" : "+v"(r) : "v"(c));                               \
    if (OP == 14) asm volatile("v_bfe_u32 %0, %0, 3, 29" : "+v"(r));                                     \
    if (OP == 15) asm volatile("v_xad_u32 %0, %0, %1, %0" : "+v"(r) : "v"(c));                           \
    if (OP == 16) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r) : "v"(c));                      \
    if (OP == 17) asm volatile("v_perm_b32 %0, %0, %1, %0" : "+v"(r) : "v"(c));                          \
    if (OP == 18) asm volatile("v_alignbyte_b32 %0, %0, %1, 1" : "+v"(r) : "v"(c));                      \
    if (OP == 19) asm volatile("v_cmp_le_u32 vcc, %0, %1\n v_add_u32 %0, %0, %1" : "+v"(r) : "v"(c) : "vcc"); \
    if (OP == 20) asm volatile("v_max_u32 %0, %0, %1" : "+v"(r) : "v"(c));                               \
    if (OP == 21) asm volatile("v_max3_u32 %0, %0, %1, %0" : "+v"(r) : "v"(c));                          \
    if (OP == 22) asm volatile("v_or_b32 %0, %0, %1" : "+v"(r) : "v"(c));                                \
    if (OP == 23) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(r) : "v"(c));                        \
    if (OP == 24) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(r) : "v"(c));                               \
    if (OP == 25) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r));   \
    if (OP == 26) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(r) : "v"(c));                            \
    if (OP == 27) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(r) : "v"(c));                            \
    if (OP == 28) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "+v"(r) : "v"(c)); \
    if (OP == 29) asm volatile("v_max_i32 %0, %0, %1" : "+v"(r) : "v"(c));
        STEP(a0) STEP(a1) STEP(a2) STEP(a3) STEP(a4) STEP(a5) STEP(a6) STEP(a7)
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0)
        *cyc = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
template <int OP> void run(const char *name, uint32_t *d, unsigned long long *dc)
{
    const int blocks = 256 * 8; // 8 waves per SIMD resident, one round
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(d, 1, dc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) k<OP><<<blocks, 256>>>(d, r, dc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    unsigned long long cyc; hipMemcpy(&cyc, dc, 8, hipMemcpyDeviceToHost);
    double ops = (double)blocks * 256 * ITER * 8;
    int per = (OP == 19) ? 2 : 1;
    // one wave ran ITER*8*per instructions in `cyc` cycles while sharing its SIMD with 7 others
    printf("%-18s %7.3f ms %7.2f Tlane-op/s  %.2f cyc/wave-instr/SIMD (s_memtime: block0 %llu ticks, %.2f ticks per own instr /8 waves = %.2f)\n", name, ms,
           ops * per / ms * 1e-9, 0.0, cyc, (double)cyc / (ITER * 8.0 * per), (double)cyc / (ITER * 8.0 * per) / 8);
}
int main()
{
    uint32_t *d; hipMalloc(&d, 256 * 8 * 256 * 4);
    unsigned long long *dc; hipMalloc(&dc, 8);
    run<4>("v_xor_b32", d, dc); run<10>("v_add_u32", d, dc); run<24>("v_sub_u32", d, dc); run<13>("v_and_b32", d, dc); run<22>("v_or_b32", d, dc);
    run<20>("v_max_u32", d, dc); run<29>("v_max_i32", d, dc); run<11>("v_lshlrev_b32", d, dc); run<12>("v_lshrrev_b32", d, dc); run<16>("v_cndmask_b32", d, dc);
    run<19>("v_cmp+v_add", d, dc);
    run<2>("v_lshl_add_u32", d, dc); run<3>("v_alignbit_b32", d, dc); run<18>("v_alignbyte_b32", d, dc); run<8>("v_add3_u32", d, dc); run<21>("v_max3_u32", d, dc);
    run<15>("v_xad_u32", d, dc); run<23>("v_lshl_or_b32", d, dc); run<14>("v_bfe_u32", d, dc); run<17>("v_perm_b32", d, dc);
    run<9>("v_xor_sdwa", d, dc); run<28>("v_add_u32_sdwa", d, dc); run<25>("v_mov_dpp", d, dc); run<26>("v_pk_add_u16", d, dc); run<27>("v_pk_max_i16", d, dc);
    run<1>("v_mul_u32_u24", d, dc); run<7>("v_mad_u32_u24", d, dc); run<0>("v_mul_lo_u32", d, dc);
    run<6>("v_mul_hi_u32", d, dc); run<5>("v_mad_u64_u32", d, dc);
    return 0;
}
