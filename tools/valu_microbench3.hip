// VGPR bank experiment (gfx950): does the issue rate of simple FP32 ops depend on which
// registers the two source operands come from?
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{ hipError_t e=(x); if(e!=hipSuccess){ printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } }while(0)
#define CLOB "v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31"
template<int P>
__global__ __launch_bounds__(256) void bench(float* out, int iters, float b)
{
    float acc = threadIdx.x * 1e-3f;
    asm volatile("v_mov_b32 v4, %0\n v_mov_b32 v5, %0\n v_mov_b32 v6, %0\n v_mov_b32 v7, %0\n v_mov_b32 v8, %0\n v_mov_b32 v9, %0\n v_mov_b32 v10, %0\n v_mov_b32 v11, %0\n"
                 "v_mov_b32 v12, %0\n v_mov_b32 v13, %0\n v_mov_b32 v14, %0\n v_mov_b32 v15, %0\n" :: "v"(b) : CLOB);
    for (int it=0; it<iters; ++it) {
#pragma unroll
        for (int u=0;u<4;++u) {
        if constexpr (P==0) {        // sources in the same bank if banks = reg mod 4 (v4,v8 / v5,v9 ...)
            asm volatile("v_mul_f32 v16, v4, v8\n v_mul_f32 v17, v5, v9\n v_mul_f32 v18, v6, v10\n v_mul_f32 v19, v7, v11\n v_mul_f32 v20, v8, v12\n v_mul_f32 v21, v9, v13\n v_mul_f32 v22, v10, v14\n v_mul_f32 v23, v11, v15" ::: CLOB);
        } else if constexpr (P==1) { // sources in adjacent registers (different banks mod 4)
            asm volatile("v_mul_f32 v16, v4, v5\n v_mul_f32 v17, v5, v6\n v_mul_f32 v18, v6, v7\n v_mul_f32 v19, v7, v8\n v_mul_f32 v20, v8, v9\n v_mul_f32 v21, v9, v10\n v_mul_f32 v22, v10, v11\n v_mul_f32 v23, v11, v12" ::: CLOB);
        } else if constexpr (P==2) { // sources 2 apart
            asm volatile("v_mul_f32 v16, v4, v6\n v_mul_f32 v17, v5, v7\n v_mul_f32 v18, v6, v8\n v_mul_f32 v19, v7, v9\n v_mul_f32 v20, v8, v10\n v_mul_f32 v21, v9, v11\n v_mul_f32 v22, v10, v12\n v_mul_f32 v23, v11, v13" ::: CLOB);
        } else if constexpr (P==3) { // same register twice
            asm volatile("v_mul_f32 v16, v4, v4\n v_mul_f32 v17, v5, v5\n v_mul_f32 v18, v6, v6\n v_mul_f32 v19, v7, v7\n v_mul_f32 v20, v8, v8\n v_mul_f32 v21, v9, v9\n v_mul_f32 v22, v10, v10\n v_mul_f32 v23, v11, v11" ::: CLOB);
        } else if constexpr (P==4) { // fma, three sources same bank mod 4
            asm volatile("v_fma_f32 v16, v4, v8, v12\n v_fma_f32 v17, v5, v9, v13\n v_fma_f32 v18, v6, v10, v14\n v_fma_f32 v19, v7, v11, v15\n v_fma_f32 v20, v4, v8, v12\n v_fma_f32 v21, v5, v9, v13\n v_fma_f32 v22, v6, v10, v14\n v_fma_f32 v23, v7, v11, v15" ::: CLOB);
        } else if constexpr (P==5) { // fma, three sources consecutive
            asm volatile("v_fma_f32 v16, v4, v5, v6\n v_fma_f32 v17, v5, v6, v7\n v_fma_f32 v18, v6, v7, v8\n v_fma_f32 v19, v7, v8, v9\n v_fma_f32 v20, v8, v9, v10\n v_fma_f32 v21, v9, v10, v11\n v_fma_f32 v22, v10, v11, v12\n v_fma_f32 v23, v11, v12, v13" ::: CLOB);
        } else if constexpr (P==6) { // dst equals src0 (in-place), other source same bank
            asm volatile("v_mul_f32 v4, v4, v8\n v_mul_f32 v5, v5, v9\n v_mul_f32 v6, v6, v10\n v_mul_f32 v7, v7, v11\n v_mul_f32 v12, v12, v8\n v_mul_f32 v13, v13, v9\n v_mul_f32 v14, v14, v10\n v_mul_f32 v15, v15, v11" ::: CLOB);
        } else if constexpr (P==7) { // dst equals src0, other source adjacent
            asm volatile("v_mul_f32 v4, v4, v9\n v_mul_f32 v5, v5, v10\n v_mul_f32 v6, v6, v11\n v_mul_f32 v7, v7, v8\n v_mul_f32 v12, v12, v9\n v_mul_f32 v13, v13, v10\n v_mul_f32 v14, v14, v11\n v_mul_f32 v15, v15, v8" ::: CLOB);
        } else if constexpr (P==8) { // sources 8 apart, 16 apart
            asm volatile("v_mul_f32 v24, v4, v12\n v_mul_f32 v25, v5, v13\n v_mul_f32 v26, v6, v14\n v_mul_f32 v27, v7, v15\n v_mul_f32 v28, v4, v20\n v_mul_f32 v29, v5, v21\n v_mul_f32 v30, v6, v22\n v_mul_f32 v31, v7, v23" ::: CLOB);
        } else if constexpr (P==10) { // dependent chain through src0
            asm volatile("v_mul_f32 v4, v4, v8\n v_add_f32 v4, v4, v9\n v_mul_f32 v4, v4, v8\n v_add_f32 v4, v4, v9\n v_mul_f32 v4, v4, v8\n v_add_f32 v4, v4, v9\n v_mul_f32 v4, v4, v8\n v_add_f32 v4, v4, v9" ::: CLOB);
        } else if constexpr (P==11) { // dependent chain through src1
            asm volatile("v_mul_f32 v4, v8, v4\n v_add_f32 v4, v9, v4\n v_mul_f32 v4, v8, v4\n v_add_f32 v4, v9, v4\n v_mul_f32 v4, v8, v4\n v_add_f32 v4, v9, v4\n v_mul_f32 v4, v8, v4\n v_add_f32 v4, v9, v4" ::: CLOB);
        } else if constexpr (P==12) { // 2 chains interleaved, dep through src1, distance 2
            asm volatile("v_mul_f32 v4, v8, v4\n v_mul_f32 v5, v8, v5\n v_add_f32 v4, v9, v4\n v_add_f32 v5, v9, v5\n v_mul_f32 v4, v8, v4\n v_mul_f32 v5, v8, v5\n v_add_f32 v4, v9, v4\n v_add_f32 v5, v9, v5" ::: CLOB);
        } else if constexpr (P==13) { // new dst each time, src1 = previous result
            asm volatile("v_mul_f32 v16, v8, v4\n v_add_f32 v17, v9, v16\n v_mul_f32 v18, v8, v17\n v_add_f32 v19, v9, v18\n v_mul_f32 v20, v8, v19\n v_add_f32 v21, v9, v20\n v_mul_f32 v22, v8, v21\n v_add_f32 v4, v9, v22" ::: CLOB);
        } else if constexpr (P==14) { // new dst each time, src0 = previous result
            asm volatile("v_mul_f32 v16, v4, v8\n v_add_f32 v17, v16, v9\n v_mul_f32 v18, v17, v8\n v_add_f32 v19, v18, v9\n v_mul_f32 v20, v19, v8\n v_add_f32 v21, v20, v9\n v_mul_f32 v22, v21, v8\n v_add_f32 v4, v22, v9" ::: CLOB);
        } else if constexpr (P==15) { // both sources are results of the two preceding instructions
            asm volatile("v_mul_f32 v16, v4, v5\n v_mul_f32 v17, v5, v6\n v_add_f32 v18, v16, v17\n v_mul_f32 v19, v18, v4\n v_mul_f32 v20, v18, v5\n v_add_f32 v4, v19, v20\n v_mul_f32 v21, v4, v6\n v_add_f32 v5, v21, v20" ::: CLOB);
        } else if constexpr (P==16) { // hash-dot snippet, accumulate with fresh product in src0 (dst = src1)
            asm volatile("v_add_f32 v15, 0x420551ec, v11\n v_mul_f32 v14, v11, v13\n v_mul_f32 v16, v12, v15\n v_add_f32 v14, v16, v14\n v_add_f32 v19, 0x420551ec, v5\n v_mul_f32 v18, v5, v7\n v_mul_f32 v20, v6, v19\n v_add_f32 v18, v20, v18" ::: CLOB);
        } else if constexpr (P==17) { // same, fresh product in src1 (dst = src0)
            asm volatile("v_add_f32 v15, 0x420551ec, v11\n v_mul_f32 v14, v11, v13\n v_mul_f32 v16, v12, v15\n v_add_f32 v14, v14, v16\n v_add_f32 v19, 0x420551ec, v5\n v_mul_f32 v18, v5, v7\n v_mul_f32 v20, v6, v19\n v_add_f32 v18, v18, v20" ::: CLOB);
        } else if constexpr (P==18) { // fract then mul by it in src1 with dst = src0
            asm volatile("v_fract_f32 v7, v7\n v_mul_f32 v14, v17, v14\n v_add_f32 v17, 0x420551ec, v7\n v_mul_f32 v13, v13, v7\n v_fract_f32 v8, v8\n v_mul_f32 v15, v18, v15\n v_add_f32 v18, 0x420551ec, v8\n v_mul_f32 v12, v12, v8" ::: CLOB);
        } else if constexpr (P==19) { // fract then mul by it in src0 with dst = src1
            asm volatile("v_fract_f32 v7, v7\n v_mul_f32 v14, v17, v14\n v_add_f32 v17, 0x420551ec, v7\n v_mul_f32 v13, v7, v13\n v_fract_f32 v8, v8\n v_mul_f32 v15, v18, v15\n v_add_f32 v18, 0x420551ec, v8\n v_mul_f32 v12, v8, v12" ::: CLOB);
        } else if constexpr (P==20) { // VOP3 fma with inline constants after an SGPR-operand mul
            asm volatile("v_add_f32 v4, 0.5, v4\n v_mul_f32 v4, %0, v4\n v_fma_f32 v5, v4, 2.0, -1.0\n v_add_f32 v6, 0.5, v6\n v_mul_f32 v6, %0, v6\n v_fma_f32 v7, v6, 2.0, -1.0\n v_mul_f32 v8, v5, v7\n v_add_f32 v9, v8, v9" :: "s"(b) : CLOB);
        } else if constexpr (P==21) { // same with two adds instead of the fma
            asm volatile("v_add_f32 v4, 0.5, v4\n v_mul_f32 v4, %0, v4\n v_add_f32 v4, v4, v4\n v_add_f32 v5, -1.0, v4\n v_add_f32 v6, 0.5, v6\n v_mul_f32 v6, %0, v6\n v_add_f32 v6, v6, v6\n v_add_f32 v7, -1.0, v6" :: "s"(b) : CLOB);
        } else if constexpr (P==9) { // add with odd/even pairs
            asm volatile("v_add_f32 v16, v4, v7\n v_add_f32 v17, v5, v8\n v_add_f32 v18, v6, v9\n v_add_f32 v19, v7, v10\n v_add_f32 v20, v8, v11\n v_add_f32 v21, v9, v12\n v_add_f32 v22, v10, v13\n v_add_f32 v23, v11, v14" ::: CLOB);
        }
        }
    }
    asm volatile("v_add_f32 %0, %0, v16\n v_add_f32 %0, %0, v23\n v_add_f32 %0, %0, v4\n v_add_f32 %0, %0, v31" : "+v"(acc) :: CLOB);
    out[blockIdx.x*blockDim.x+threadIdx.x]=acc;
}
template<int P> int run(const char* name, float* out, int wps)
{
    const int iters=4000; int blocks=256*wps;
    hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(bench<P>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0001f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(bench<P>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms,e0,e1));
    double instr = (double)iters*4*8*blocks*4;
    printf("%-52s wps=%d wall=%.3f ms  wave-instr per SIMD per ns: %.3f\n", name, wps, ms, instr/(ms*1e6)/1024.0);
    return 0;
}
int main(){
    float* out; CK(hipMalloc(&out, 256*8*256*sizeof(float)));
    for (int wps : {2,8}) {
        run<0>("mul src regs 4 apart (v4,v8)",out,wps); run<1>("mul src regs adjacent (v4,v5)",out,wps); run<2>("mul src regs 2 apart (v4,v6)",out,wps);
        run<9>("add src regs 3 apart (v4,v7)",out,wps);
        run<3>("mul same reg twice (v4,v4)",out,wps); run<8>("mul src regs 8 / 16 apart",out,wps);
        run<4>("fma three srcs 4 apart (v4,v8,v12)",out,wps); run<5>("fma three srcs consecutive (v4,v5,v6)",out,wps);
        run<6>("mul in place, other src 4 apart",out,wps); run<7>("mul in place, other src 5 apart",out,wps);
        run<16>("hash dot, fresh product in src0",out,wps); run<17>("hash dot, fresh product in src1",out,wps);
        run<18>("fract, mul by it in src1 (dst=src0)",out,wps); run<19>("fract, mul by it in src0 (dst=src1)",out,wps);
        run<20>("sgpr mul then VOP3 fma inline consts",out,wps); run<21>("sgpr mul then two adds",out,wps);
        run<10>("chain dep via src0, in place",out,wps); run<11>("chain dep via src1, in place",out,wps); run<12>("2 chains dep via src1",out,wps);
        run<13>("chain dep via src1, fresh dst",out,wps); run<14>("chain dep via src0, fresh dst",out,wps); run<15>("both srcs fresh results",out,wps);
    }
    return 0;
}
