// valu_bank.hip -- does the VGPR bank (register index mod 4) of packed-FP32 source operands
// change the issue cost on gfx950?  Explicit registers, 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int ITERS = 8192;
template <int KIND>
__global__ void k(float* out)
{
    asm volatile("v_mov_b32 v10, 1.0\nv_mov_b32 v11, 1.0\nv_mov_b32 v12, 0\nv_mov_b32 v13, 0\nv_mov_b32 v14, 0\nv_mov_b32 v15, 0\n"
                 "v_mov_b32 v16, 1.0\nv_mov_b32 v17, 1.0\nv_mov_b32 v18, 1.0\nv_mov_b32 v19, 1.0\n"
                 "v_mov_b32 v20, 1.0\nv_mov_b32 v21, 1.0\nv_mov_b32 v22, 1.0\nv_mov_b32 v23, 1.0\n" ::: "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23");
    for (int it = 0; it < ITERS; ++it) {
        if (KIND == 0) // sources in different banks: dst bank 0 (v16), src0 v[16:17] bank 0, src1 v[10:11] bank 2
            asm volatile("v_pk_add_f32 v[16:17], v[16:17], v[10:11]\nv_pk_add_f32 v[18:19], v[18:19], v[12:13]\n"
                         "v_pk_add_f32 v[20:21], v[20:21], v[10:11]\nv_pk_add_f32 v[22:23], v[22:23], v[12:13]\n" ::: "v16","v17","v18","v19","v20","v21","v22","v23");
        if (KIND == 1) // sources in the SAME bank: v[16:17] with v[12:13] (both bank 0)
            asm volatile("v_pk_add_f32 v[16:17], v[16:17], v[12:13]\nv_pk_add_f32 v[18:19], v[18:19], v[14:15]\n"
                         "v_pk_add_f32 v[20:21], v[20:21], v[12:13]\nv_pk_add_f32 v[22:23], v[22:23], v[14:15]\n" ::: "v16","v17","v18","v19","v20","v21","v22","v23");
        if (KIND == 2) // fma, three sources all different banks?  v16(0) v10(2) v13.. use pairs: [16:17](0),[10:11](2),[12:13](0)
            asm volatile("v_pk_fma_f32 v[16:17], v[16:17], v[10:11], v[12:13]\nv_pk_fma_f32 v[18:19], v[18:19], v[12:13], v[14:15]\n"
                         "v_pk_fma_f32 v[20:21], v[20:21], v[10:11], v[12:13]\nv_pk_fma_f32 v[22:23], v[22:23], v[12:13], v[14:15]\n" ::: "v16","v17","v18","v19","v20","v21","v22","v23");
        if (KIND == 3) // fma, all three sources bank 0
            asm volatile("v_pk_fma_f32 v[16:17], v[16:17], v[12:13], v[20:21]\nv_pk_fma_f32 v[18:19], v[18:19], v[14:15], v[22:23]\n"
                         "v_pk_fma_f32 v[16:17], v[16:17], v[12:13], v[20:21]\nv_pk_fma_f32 v[18:19], v[18:19], v[14:15], v[22:23]\n" ::: "v16","v17","v18","v19");
        if (KIND == 4) // high registers
            asm volatile("v_pk_add_f32 v[200:201], v[200:201], v[130:131]\nv_pk_add_f32 v[202:203], v[202:203], v[132:133]\n"
                         "v_pk_add_f32 v[204:205], v[204:205], v[130:131]\nv_pk_add_f32 v[206:207], v[206:207], v[132:133]\n" ::: "v200","v201","v202","v203","v204","v205","v206","v207","v130","v131","v132","v133");
    }
    float acc;
    asm volatile("v_add_f32 %0, v16, v18" : "=v"(acc));
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int KIND>
void run(const char* name, float* d)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    k<KIND><<<256, 512>>>(d);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<KIND><<<256, 512>>>(d);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %.3f ms -> %.2f ns per wave-instruction per SIMD (2 waves/SIMD)\n", name, ms, ms * 1e6 / (double(ITERS) * 4 * 2));
}
int main()
{
    float* d;
    (void)hipMalloc(&d, 256 * 512 * sizeof(float));
    run<0>("pk_add sources in different banks", d);
    run<1>("pk_add sources in the same bank", d);
    run<2>("pk_fma 3 sources, mixed banks", d);
    run<3>("pk_fma 3 sources, same bank", d);
    run<4>("pk_add registers >= 128", d);
    return 0;
}
