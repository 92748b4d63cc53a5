// Mixed-stream VALU issue microbenchmark (gfx950): does interleaving op kinds cost issue slots?
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{ hipError_t e=(x); if(e!=hipSuccess){ printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } }while(0)

template<int P>
__global__ __launch_bounds__(256) void bench(float* out, int iters, float b, float c)
{
    float a0=threadIdx.x*1e-3f+0.1f,a1=a0+1,a2=a0+2,a3=a0+3,a4=a0+4,a5=a0+5,a6=a0+6,a7=a0+7;
    float t0=0,t1=0,t2=0,t3=0;
    for (int it=0; it<iters; ++it) {
#pragma unroll
        for (int u=0;u<4;++u) {
        if constexpr (P==0) { // 8 independent fma
            asm volatile("v_fma_f32 %0,%0,%8,%9\n v_fma_f32 %1,%1,%8,%9\n v_fma_f32 %2,%2,%8,%9\n v_fma_f32 %3,%3,%8,%9\n v_fma_f32 %4,%4,%8,%9\n v_fma_f32 %5,%5,%8,%9\n v_fma_f32 %6,%6,%8,%9\n v_fma_f32 %7,%7,%8,%9"
              : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(b),"v"(c));
        } else if constexpr (P==1) { // alternating independent mul / add
            asm volatile("v_mul_f32 %0,%0,%8\n v_add_f32 %1,%1,%9\n v_mul_f32 %2,%2,%8\n v_add_f32 %3,%3,%9\n v_mul_f32 %4,%4,%8\n v_add_f32 %5,%5,%9\n v_mul_f32 %6,%6,%8\n v_add_f32 %7,%7,%9"
              : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(b),"v"(c));
        } else if constexpr (P==2) { // dependent mul->add pairs (4 chains)
            asm volatile("v_mul_f32 %4,%0,%8\n v_add_f32 %0,%4,%9\n v_mul_f32 %5,%1,%8\n v_add_f32 %1,%5,%9\n v_mul_f32 %6,%2,%8\n v_add_f32 %2,%6,%9\n v_mul_f32 %7,%3,%8\n v_add_f32 %3,%7,%9"
              : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(t0),"+v"(t1),"+v"(t2),"+v"(t3) : "v"(b),"v"(c));
        } else if constexpr (P==3) { // one long dependent chain mul->add->mul->add
            asm volatile("v_mul_f32 %0,%0,%8\n v_add_f32 %0,%0,%9\n v_mul_f32 %0,%0,%8\n v_add_f32 %0,%0,%9\n v_mul_f32 %0,%0,%8\n v_add_f32 %0,%0,%9\n v_mul_f32 %0,%0,%8\n v_add_f32 %0,%0,%9"
              : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(b),"v"(c));
        } else if constexpr (P==4) { // literal add / literal mul alternating, dependent (hash-like)
            asm volatile("v_mul_f32 %0,0x3dd32618,%0\n v_add_f32 %0,0x420551ec,%0\n v_mul_f32 %1,0x3dd32618,%1\n v_add_f32 %1,0x420551ec,%1\n v_mul_f32 %2,0x3dd32618,%2\n v_add_f32 %2,0x420551ec,%2\n v_mul_f32 %3,0x3dd32618,%3\n v_add_f32 %3,0x420551ec,%3"
              : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(b),"v"(c));
        } else if constexpr (P==5) { // fract + mul alternating
            asm volatile("v_fract_f32 %0,%0\n v_mul_f32 %1,%1,%8\n v_fract_f32 %2,%2\n v_mul_f32 %3,%3,%8\n v_fract_f32 %4,%4\n v_mul_f32 %5,%5,%8\n v_fract_f32 %6,%6\n v_mul_f32 %7,%7,%8"
              : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(b),"v"(c));
        } else if constexpr (P==6) { // v_sub / v_add same-reg doubling
            asm volatile("v_sub_f32 %0,%8,%0\n v_add_f32 %1,%1,%1\n v_sub_f32 %2,%8,%2\n v_add_f32 %3,%3,%3\n v_sub_f32 %4,%8,%4\n v_add_f32 %5,%5,%5\n v_sub_f32 %6,%8,%6\n v_add_f32 %7,%7,%7"
              : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(b),"v"(c));
        } else if constexpr (P==7) { // 8 independent mul
            asm volatile("v_mul_f32 %0,%0,%8\n v_mul_f32 %1,%1,%8\n v_mul_f32 %2,%2,%8\n v_mul_f32 %3,%3,%8\n v_mul_f32 %4,%4,%8\n v_mul_f32 %5,%5,%8\n v_mul_f32 %6,%6,%8\n v_mul_f32 %7,%7,%8"
              : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(b),"v"(c));
        } else if constexpr (P==8) { // mul/add two-source different registers (bank pattern): dst = src1 op src2
            asm volatile("v_mul_f32 %4,%0,%1\n v_add_f32 %5,%2,%3\n v_mul_f32 %6,%1,%2\n v_add_f32 %7,%3,%0\n v_mul_f32 %0,%4,%5\n v_add_f32 %1,%6,%7\n v_mul_f32 %2,%5,%6\n v_add_f32 %3,%7,%4"
              : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(b),"v"(c));
        } else if constexpr (P==9) { // fmac chain style
            asm volatile("v_fmac_f32 %0,%8,%9\n v_fmac_f32 %1,%8,%9\n v_fmac_f32 %2,%8,%9\n v_fmac_f32 %3,%8,%9\n v_fmac_f32 %4,%8,%9\n v_fmac_f32 %5,%8,%9\n v_fmac_f32 %6,%8,%9\n v_fmac_f32 %7,%8,%9"
              : "+v"(a0),"+v"(a1),"+v"(a2),"+v"(a3),"+v"(a4),"+v"(a5),"+v"(a6),"+v"(a7) : "v"(b),"v"(c));
        }
        }
    }
    out[blockIdx.x*blockDim.x+threadIdx.x]=a0+a1+a2+a3+a4+a5+a6+a7+t0+t1+t2+t3;
}

template<int P> int run(const char* name, float* out, int wps)
{
    const int iters=4000;
    int blocks=256*wps;
    hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(bench<P>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.0001f, 0.5f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(bench<P>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms,e0,e1));
    double instr = (double)iters*4*8*blocks*4;
    printf("%-44s wps=%d wall=%.3f ms  wave-instr per SIMD per ns: %.3f\n", name, wps, ms, instr/(ms*1e6)/1024.0);
    return 0;
}
int main(){
    float* out; CK(hipMalloc(&out, 256*8*256*sizeof(float)));
    for (int wps : {2,4,8}) {
        run<0>("fma x8 independent",out,wps); run<7>("mul x8 independent",out,wps); run<9>("fmac x8 independent",out,wps);
        run<1>("mul/add alternating independent",out,wps);
        run<2>("mul->add dependent pairs (4 chains)",out,wps); run<3>("single chain mul->add",out,wps);
        run<4>("literal mul->literal add dependent",out,wps); run<5>("fract/mul alternating",out,wps);
        run<6>("sub / add x,x",out,wps); run<8>("two-source mul/add web",out,wps);
    }
    return 0;
}
