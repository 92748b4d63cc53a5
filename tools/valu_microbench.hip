// VALU issue-rate microbenchmark for gfx950: cycles per wave64 instruction at 1..8 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O2 tools/valu_microbench.hip -o tools/valu_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <string>

#define CK(x) do{ hipError_t e=(x); if(e!=hipSuccess){ printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } }while(0)

#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)

template<int OP>
__global__ __launch_bounds__(256) void bench(float* out, long long* cyc, int iters, float b, float c)
{
    float a0=threadIdx.x*1e-3f+0.1f,a1=a0+1,a2=a0+2,a3=a0+3,a4=a0+4,a5=a0+5,a6=a0+6,a7=a0+7;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0={a0,a1},p1={a2,a3},p2={a4,a5},p3={a6,a7},p4={a1,a0},p5={a3,a2},p6={a5,a4},p7={a7,a6};
    f2 pb={b,b}, pc={c,c};
    double d0=a0,d1=a1,d2=a2,d3=a3,d4=a4,d5=a5,d6=a6,d7=a7; double db=b, dc=c;
    int i0=threadIdx.x,i1=i0+1,i2=i0+2,i3=i0+3,i4=i0+4,i5=i0+5,i6=i0+6,i7=i0+7; int ib=(int)b+3;
    long long t0 = clock64();
    for (int it=0; it<iters; ++it) {
#pragma unroll
        for (int u=0;u<4;++u) {
        if constexpr (OP==0) { // v_fma_f32
#define S(n) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a##n) : "v"(b), "v"(c));
            REP8(S)
#undef S
        } else if constexpr (OP==1) {
#define S(n) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a##n) : "v"(b));
            REP8(S)
#undef S
        } else if constexpr (OP==2) {
#define S(n) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a##n) : "v"(c));
            REP8(S)
#undef S
        } else if constexpr (OP==3) {
#define S(n) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p##n) : "v"(pb), "v"(pc));
            REP8(S)
#undef S
        } else if constexpr (OP==4) {
#define S(n) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p##n) : "v"(pb));
            REP8(S)
#undef S
        } else if constexpr (OP==5) {
#define S(n) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p##n) : "v"(pc));
            REP8(S)
#undef S
        } else if constexpr (OP==6) {
#define S(n) asm volatile("v_fract_f32 %0, %0" : "+v"(a##n));
            REP8(S)
#undef S
        } else if constexpr (OP==7) {
#define S(n) asm volatile("v_floor_f32 %0, %0" : "+v"(a##n));
            REP8(S)
#undef S
        } else if constexpr (OP==8) {
#define S(n) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a##n) : "v"(c));
            REP8(S)
#undef S
        } else if constexpr (OP==9) {
#define S(n) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(a##n) : "v"(b), "v"(c) : "vcc");
            REP8(S)
#undef S
        } else if constexpr (OP==10) {
#define S(n) asm volatile("v_rcp_f32 %0, %0" : "+v"(a##n));
            REP8(S)
#undef S
        } else if constexpr (OP==11) {
#define S(n) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a##n));
            REP8(S)
#undef S
        } else if constexpr (OP==12) {
#define S(n) asm volatile("v_sin_f32 %0, %0" : "+v"(a##n));
            REP8(S)
#undef S
        } else if constexpr (OP==13) {
#define S(n) asm volatile("v_rndne_f32 %0, %0" : "+v"(a##n));
            REP8(S)
#undef S
        } else if constexpr (OP==14) {
#define S(n) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d##n) : "v"(db), "v"(dc));
            REP8(S)
#undef S
        } else if constexpr (OP==15) {
#define S(n) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(i##n) : "v"(ib));
            REP8(S)
#undef S
        } else if constexpr (OP==16) {
#define S(n) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(i##n) : "v"(ib));
            REP8(S)
#undef S
        } else if constexpr (OP==17) {
#define S(n) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(i##n) : "v"(ib));
            REP8(S)
#undef S
        } else if constexpr (OP==18) {
#define S(n) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(i##n));
            REP8(S)
#undef S
        } else if constexpr (OP==19) {
#define S(n) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a##n) : "v"(b), "v"(c));
            REP8(S)
#undef S
        } else if constexpr (OP==20) { // literal constant operand (8-byte encoding)
#define S(n) asm volatile("v_mul_f32 %0, 0x3dd32618, %0" : "+v"(a##n));
            REP8(S)
#undef S
        } else if constexpr (OP==21) { // v_fma with literal (VOP3 + literal = 12 bytes)
#define S(n) asm volatile("v_fmaak_f32 %0, %0, %1, 0x42055eb8" : "+v"(a##n) : "v"(b));
            REP8(S)
#undef S
        } else if constexpr (OP==22) { // sgpr operand
#define S(n) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a##n) : "s"(b));
            REP8(S)
#undef S
        } else if constexpr (OP==23) {
#define S(n) asm volatile("v_exp_f32 %0, %0" : "+v"(a##n));
            REP8(S)
#undef S
        } else if constexpr (OP==24) {
#define S(n) asm volatile("v_rsq_f32 %0, %0" : "+v"(a##n));
            REP8(S)
#undef S
        } else if constexpr (OP==25) { // dependent chain fma
            asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2" : "+v"(a0) : "v"(b), "v"(c));
        }
        }
    }
    long long t1 = clock64();
    float s=a0+a1+a2+a3+a4+a5+a6+a7 + p0.x+p1.x+p2.x+p3.x+p4.y+p5.y+p6.y+p7.y + (float)(d0+d1+d2+d3+d4+d5+d6+d7) + (float)(i0+i1+i2+i3+i4+i5+i6+i7);
    int gid = blockIdx.x*blockDim.x+threadIdx.x;
    out[gid]=s;
    if ((threadIdx.x&63)==0) cyc[gid>>6]=t1-t0;
}

template<int OP> int run(const char* name, float* out, long long* cyc, int wps)
{
    const int iters=2000;
    int blocks=256*wps; // 256 threads = 4 waves = 1 wave/SIMD per block
    int nw=blocks*4;
    hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, 10, 1.0001f, 0.5f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(bench<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters, 1.0001f, 0.5f);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms,e0,e1));
    std::vector<long long> h(nw); CK(hipMemcpy(h.data(), cyc, nw*sizeof(long long), hipMemcpyDeviceToHost));
    std::sort(h.begin(),h.end());
    double instr_per_wave = (double)iters*4*8;
    double med = (double)h[nw/2];
    // aggregate: wave-instructions per second
    double wips = instr_per_wave*nw/(ms*1e-3);
    printf("%-22s wps=%d  cyc/instr/wave=%.2f  => cyc/instr/SIMD=%.2f  wall=%.3f ms  Gwave-instr/s=%.1f  (per SIMD per ns: %.3f)\n",
           name, wps, med/instr_per_wave, med/instr_per_wave/wps, ms, wips*1e-9, wips*1e-9/1024.0);
    return 0;
}

int main(){
    float* out; long long* cyc;
    CK(hipMalloc(&out, 256*8*256*sizeof(float))); CK(hipMalloc(&cyc, 256*8*4*sizeof(long long)));
    hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr,0));
    printf("device %s CUs=%d clock=%d kHz\n", pr.name, pr.multiProcessorCount, pr.clockRate);
    for (int wps : {1,2,4,8}) {
        run<0>("v_fma_f32",out,cyc,wps); run<1>("v_mul_f32",out,cyc,wps); run<2>("v_add_f32",out,cyc,wps);
        run<19>("v_fmac_f32",out,cyc,wps);
        run<3>("v_pk_fma_f32",out,cyc,wps); run<4>("v_pk_mul_f32",out,cyc,wps); run<5>("v_pk_add_f32",out,cyc,wps);
        run<6>("v_fract_f32",out,cyc,wps); run<7>("v_floor_f32",out,cyc,wps); run<8>("v_max_f32",out,cyc,wps);
        run<9>("v_cmp+cndmask",out,cyc,wps); run<10>("v_rcp_f32",out,cyc,wps); run<11>("v_sqrt_f32",out,cyc,wps);
        run<12>("v_sin_f32",out,cyc,wps); run<13>("v_rndne_f32",out,cyc,wps); run<14>("v_fma_f64",out,cyc,wps);
        run<15>("v_mul_lo_u32",out,cyc,wps); run<16>("v_xor_b32",out,cyc,wps); run<17>("v_mad_u32_u24",out,cyc,wps);
        run<18>("v_cvt_f32_i32",out,cyc,wps); run<20>("v_mul_f32 literal",out,cyc,wps); run<21>("v_fmaak_f32 literal",out,cyc,wps);
        run<22>("v_mul_f32 sgpr",out,cyc,wps); run<23>("v_exp_f32",out,cyc,wps); run<24>("v_rsq_f32",out,cyc,wps);
        run<25>("v_fma_f32 dependent",out,cyc,wps);
    }
    return 0;
}
