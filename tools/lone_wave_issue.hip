// What does a dependent instruction cost a wave that has a SIMD to itself?  (round 6: the phasor chain of
// CoarseFrequencyCorrection is three packed instructions a step and measures 5.3 ns per instruction.)
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/lwi tools/lone_wave_issue.hip && /tmp/lwi
// Prints, per instruction kind: core-clock cycles and ns per instruction for a chain of dependent instructions, for four
// independent chains interleaved, and with 1 / 2 / 4 / 8 waves of one workgroup (they share a CU's SIMDs).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND>
__global__ void k(unsigned iters, unsigned long long* out, float* sink)
{
    f2 a = { 1.0f + threadIdx.x * 1e-7f, 0.5f }, b = { 0.99999f, 1.00001f }, c = a, d = a, e = a;
    float s = a.x, t = 0.999999f, u = s, v = s, w = s;
    __builtin_amdgcn_s_setprio(3);
    const unsigned long long w0 = wall_clock64();
    const unsigned long long c0 = clock64();
    for (unsigned i = 0; i < iters; ++i) {
        if (KIND == 0) { asm volatile(REP64("v_pk_mul_f32 %0, %0, %1\n") : "+v"(a) : "v"(b)); }
        if (KIND == 1) { asm volatile(REP64("v_mul_f32 %0, %0, %1\n") : "+v"(s) : "v"(t)); }
        if (KIND == 2) { asm volatile(REP64("v_fma_f32 %0, %0, %1, %1\n") : "+v"(s) : "v"(t)); }
        if (KIND == 3) { asm volatile(REP64("v_pk_add_f32 %0, %0, %1\n") : "+v"(a) : "v"(b)); }
        if (KIND == 4) { // four independent packed chains interleaved: 64 instructions
            asm volatile(REP8(REP8("v_pk_mul_f32 %0, %0, %4\nv_pk_mul_f32 %1, %1, %4\nv_pk_mul_f32 %2, %2, %4\nv_pk_mul_f32 %3, %3, %4\n"))
                         : "+v"(a), "+v"(c), "+v"(d), "+v"(e) : "v"(b));
        }
        if (KIND == 5) { // four independent plain chains interleaved
            asm volatile(REP8(REP8("v_mul_f32 %0, %0, %4\nv_mul_f32 %1, %1, %4\nv_mul_f32 %2, %2, %4\nv_mul_f32 %3, %3, %4\n"))
                         : "+v"(s), "+v"(u), "+v"(v), "+v"(w) : "v"(t));
        }
        if (KIND == 6) { // the chain's step: mul, mul, add (each dependent on the step before)
            f2 x, y;
            asm volatile(REP8(REP8("v_pk_mul_f32 %1, %0, %3 op_sel_hi:[0,1]\nv_pk_mul_f32 %2, %0, %3 op_sel:[1,1] op_sel_hi:[1,0]\n"
                                   "v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,0]\n"))
                         : "+v"(a), "=&v"(x), "=&v"(y) : "v"(b));
        }
    }
    const unsigned long long c1 = clock64();
    const unsigned long long w1 = wall_clock64();
    if (threadIdx.x % 64 == 0) {
        out[2 * (blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64)] = c1 - c0;
        out[2 * (blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) + 1] = w1 - w0;
    }
    sink[threadIdx.x] = a.x + c.x + d.x + e.x + s + u + v + w;
}

template <int KIND>
void run(const char* name, unsigned per_iter)
{
    unsigned long long* out;
    float* sink;
    hipMalloc(&out, 1024 * 16);
    hipMalloc(&sink, 4096 * 4);
    const unsigned iters = 20000;
    for (int waves : { 1, 2, 4, 8, 16 }) {
        hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(64 * waves), 0, 0, iters, out, sink);
        hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(64 * waves), 0, 0, iters, out, sink);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(2 * waves);
        hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
        const double n = static_cast<double>(iters) * per_iter;
        printf("%-44s waves %2d: %6.2f core clocks, %6.2f ns per instruction (core clock %.2f GHz)\n", name, waves, h[0] / n,
               h[1] * 10.0 / n, h[0] / (h[1] * 10.0));
    }
    hipFree(out);
    hipFree(sink);
}

int main()
{
    run<0>("v_pk_mul_f32, dependent", 64);
    run<1>("v_mul_f32, dependent", 64);
    run<2>("v_fma_f32, dependent", 64);
    run<3>("v_pk_add_f32, dependent", 64);
    run<4>("v_pk_mul_f32, four chains interleaved", 256);
    run<5>("v_mul_f32, four chains interleaved", 256);
    run<6>("the phasor step (pk_mul, pk_mul, pk_add)", 192);
    return 0;
}
