// Issue rate of integer / byte VALU instructions on gfx950: eight independent instructions per block, 8 waves per SIMD.
// Prints wave-instructions per ns per SIMD (a dependent-free v_fma_f32 stream reaches about 0.9-1.2 here).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{ hipError_t e=(x); if(e!=hipSuccess){ printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } }while(0)
#define CLOB "v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23"
#define OP3(op) op " v16, v4, v8, v12\n" op " v17, v5, v9, v13\n" op " v18, v6, v10, v14\n" op " v19, v7, v11, v15\n" op " v20, v8, v4, v13\n" op " v21, v9, v5, v12\n" op " v22, v10, v6, v15\n" op " v23, v11, v7, v14\n"
#define OP2(op) op " v16, v4, v8\n" op " v17, v5, v9\n" op " v18, v6, v10\n" op " v19, v7, v11\n" op " v20, v8, v12\n" op " v21, v9, v13\n" op " v22, v10, v14\n" op " v23, v11, v15\n"
template<int P>
__global__ __launch_bounds__(256) void bench(unsigned* out, int iters, unsigned b)
{
    unsigned acc = threadIdx.x;
    asm volatile("v_mov_b32 v4, %0\n v_mov_b32 v5, %0\n v_mov_b32 v6, %0\n v_mov_b32 v7, %0\n v_mov_b32 v8, %0\n v_mov_b32 v9, %0\n v_mov_b32 v10, %0\n v_mov_b32 v11, %0\n"
                 "v_mov_b32 v12, %0\n v_mov_b32 v13, %0\n v_mov_b32 v14, %0\n v_mov_b32 v15, %0\n v_mov_b32 v16, %0\n v_mov_b32 v23, %0\n" :: "v"(b) : CLOB);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if constexpr (P == 0) asm volatile(OP3("v_fma_f32") ::: CLOB);
            else if constexpr (P == 1) asm volatile(OP3("v_mad_u32_u24") ::: CLOB);
            else if constexpr (P == 2) asm volatile(OP2("v_mul_u32_u24") ::: CLOB);
            else if constexpr (P == 3) asm volatile(OP2("v_mul_lo_u32") ::: CLOB);
            else if constexpr (P == 4) asm volatile(OP3("v_lshl_add_u32") ::: CLOB);
            else if constexpr (P == 5) asm volatile(OP3("v_add3_u32") ::: CLOB);
            else if constexpr (P == 6) asm volatile(OP2("v_add_u32") ::: CLOB);
            else if constexpr (P == 7) asm volatile(OP2("v_and_b32") ::: CLOB);
            else if constexpr (P == 8) asm volatile(OP3("v_perm_b32") ::: CLOB);
            else if constexpr (P == 9) asm volatile(OP3("v_dot4_u32_u8") ::: CLOB);
            else if constexpr (P == 10) asm volatile(OP3("v_pk_mad_u16") ::: CLOB);
            else if constexpr (P == 11) asm volatile(OP2("v_pk_add_u16") ::: CLOB);
            else if constexpr (P == 12) asm volatile(OP2("v_lshrrev_b32") ::: CLOB);
            else if constexpr (P == 13) asm volatile(OP3("v_lshl_or_b32") ::: CLOB);
            else if constexpr (P == 14) asm volatile(OP3("v_alignbyte_b32") ::: CLOB);
            else if constexpr (P == 15) asm volatile(OP3("v_bfe_u32") ::: CLOB);
            else if constexpr (P == 16) asm volatile(OP3("v_and_or_b32") ::: CLOB);
            else if constexpr (P == 17) asm volatile(OP3("v_dot2_u32_u16") ::: CLOB);
            else if constexpr (P == 18) asm volatile(OP3("v_mad_u16") ::: CLOB);
            else if constexpr (P == 19) asm volatile(OP3("v_sad_u8") ::: CLOB);
            else if constexpr (P == 20) asm volatile(OP2("v_max_f32") ::: CLOB);
            else if constexpr (P == 21) asm volatile(OP2("v_max_u32") ::: CLOB);
            else if constexpr (P == 22) asm volatile(OP2("v_min_i32") ::: CLOB);
            else if constexpr (P == 23) asm volatile(OP2("v_mul_f32") ::: CLOB);
            else if constexpr (P == 24) asm volatile(OP2("v_add_f32") ::: CLOB);
            else if constexpr (P == 25) asm volatile(OP2("v_xor_b32") ::: CLOB);
            else if constexpr (P == 26) asm volatile(OP2("v_lshlrev_b32") ::: CLOB);
            else if constexpr (P == 27) asm volatile(OP2("v_sub_u32") ::: CLOB);
        }
    }
    asm volatile("v_add_u32 %0, %0, v16\n v_add_u32 %0, %0, v23\n" : "+v"(acc) :: CLOB);
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template<int P> int run(const char* name, unsigned* out, int wps)
{
    const int iters = 4000; int blocks = 256 * wps;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(bench<P>, dim3(blocks), dim3(256), 0, 0, out, 10, 3u);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(bench<P>, dim3(blocks), dim3(256), 0, 0, out, iters, 3u);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double instr = (double)iters * 4 * 8 * blocks * 4;
    printf("%-20s wps=%d wall=%.3f ms  wave-instr per SIMD per ns: %.3f\n", name, wps, ms, instr / (ms * 1e6) / 1024.0);
    return 0;
}
int main()
{
    unsigned* out; CK(hipMalloc(&out, 256 * 8 * 256 * 4));
    for (int wps : {2, 8}) {
        run<0>("v_fma_f32", out, wps); run<1>("v_mad_u32_u24", out, wps); run<2>("v_mul_u32_u24", out, wps); run<3>("v_mul_lo_u32", out, wps);
        run<4>("v_lshl_add_u32", out, wps); run<5>("v_add3_u32", out, wps); run<6>("v_add_u32", out, wps); run<7>("v_and_b32", out, wps);
        run<8>("v_perm_b32", out, wps); run<9>("v_dot4_u32_u8", out, wps); run<10>("v_pk_mad_u16", out, wps); run<11>("v_pk_add_u16", out, wps);
        run<12>("v_lshrrev_b32", out, wps); run<13>("v_lshl_or_b32", out, wps); run<14>("v_alignbyte_b32", out, wps); run<15>("v_bfe_u32", out, wps);
        run<16>("v_and_or_b32", out, wps); run<17>("v_dot2_u32_u16", out, wps); run<18>("v_mad_u16", out, wps); run<19>("v_sad_u8", out, wps);
        run<20>("v_max_f32", out, wps); run<21>("v_max_u32", out, wps); run<22>("v_min_i32", out, wps); run<23>("v_mul_f32", out, wps);
        run<24>("v_add_f32", out, wps); run<25>("v_xor_b32", out, wps); run<26>("v_lshlrev_b32", out, wps); run<27>("v_sub_u32", out, wps);
    }
    return 0;
}
