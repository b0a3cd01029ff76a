// valu_rate.hip -- measures issue cost (cycles per wave-instruction per SIMD) of packed vs
// scalar FP32 VALU instructions on gfx950 at 1/2/4 waves per SIMD.  Build & run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate tools/valu_rate.hip && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2 __attribute__((ext_vector_type(2)));
constexpr int ITERS = 4096;
template <int KIND>
__global__ void k(float* out, float a, float b)
{
    v2 r[8];
    float s[16];
    for (int i = 0; i < 8; ++i) r[i] = v2{ a + i, b - i };
    for (int i = 0; i < 16; ++i) s[i] = a * i + b;
    v2 w = v2{ a, b };
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (KIND == 0) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(w));
            if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(r[i]) : "v"(w));
            if (KIND == 2) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(w));
            if (KIND == 3) { asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[2*i]) : "v"(a)); asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[2*i+1]) : "v"(a)); }
            if (KIND == 4) { asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(s[2*i]) : "v"(a)); asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(s[2*i+1]) : "v"(a)); }
            if (KIND == 5) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "+v"(r[i]) : "v"(w));
            if (KIND == 6) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(r[0]) : "v"(w));          // fully dependent chain
            if (KIND == 7) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(r[i & 1]) : "v"(w));      // 2 interleaved chains
            if (KIND == 8) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(r[i & 3]) : "v"(w));      // 4 interleaved chains
            if (KIND == 9) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[0]) : "v"(a));             // dependent scalar chain
        }
    }
    float acc = 0;
    for (int i = 0; i < 8; ++i) acc += r[i].x + r[i].y;
    for (int i = 0; i < 16; ++i) acc += s[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int KIND>
void run(const char* name, int inst_per_iter, float* d)
{
    for (int waves_per_simd : { 1, 2, 4 }) {
        const int threads = 64 * 4 * waves_per_simd; // one block per CU, waves spread over the 4 SIMDs
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        k<KIND><<<256, threads>>>(d, 1.0f, 1e-6f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<KIND><<<256, threads>>>(d, 1.0f, 1e-6f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double insts_per_simd = double(ITERS) * inst_per_iter * waves_per_simd;
        printf("%-28s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instruction per SIMD (x clock GHz = cycles)\n", name,
               waves_per_simd, ms, ms * 1e6 / insts_per_simd);
    }
}
int main()
{
    float* d;
    hipMalloc(&d, 256 * 1024 * 4 * sizeof(float));
    run<0>("v_pk_add_f32", 8, d);
    run<1>("v_pk_fma_f32", 8, d);
    run<2>("v_pk_mul_f32", 8, d);
    run<3>("v_add_f32", 16, d);
    run<4>("v_fma_f32", 16, d);
    run<5>("v_pk_add_f32 op_sel/neg", 8, d);
    run<6>("v_pk_add_f32 dependent x1", 8, d);
    run<7>("v_pk_add_f32 dependent x2", 8, d);
    run<8>("v_pk_add_f32 dependent x4", 8, d);
    run<9>("v_add_f32 dependent x1", 8, d);
    return 0;
}
