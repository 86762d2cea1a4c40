// magic_floor.hip -- can floor(p) + h be had from two additions instead of v_floor_f32 + v_add_f32?
//
//   t = p + M       with the single-precision rounding mode set to "toward -infinity", M = 1.5 * 2^23:
//                   for |p| < 2^22 the sum lies in [2^23, 2^24), where floats are the integers, so
//                   t = floor(p) + M exactly, and the low mantissa bits of t hold floor(p) as an integer;
//   r = t - (M - h) in round-to-nearest: exact, = floor(p) + h, and +0 when it is zero (as floor(p) + h is).
//
// Two questions, answered on the GPU: (1) is r bit-identical to v_floor_f32 + v_add_f32 for every p the DDA
// can see (all floats with |p| < 2^22 on a stride, plus the special ones); (2) what do the two s_setreg
// cost inside a step-shaped instruction sequence, 8 wavefronts per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

#define RTNI "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 2\n\t"
#define RNE "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0\n\t"

__global__ void check(uint32_t first, uint32_t stride, uint32_t n, unsigned long long *bad, uint32_t *first_bad)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t bits = first + i * stride;
    float p;
    memcpy(&p, &bits, 4);
    if (!(fabsf(p) < 4194304.0f)) return;
    for (int hh = 0; hh < 2; ++hh) {
        const float h = (float)hh, g = 12582912.0f - h;
        float want, want_t;
        asm volatile("v_floor_f32 %0, %2\n\tv_add_f32 %1, %0, %3" : "=&v"(want_t), "=v"(want) : "v"(p), "v"(h));
        float t, got;
        asm volatile(RTNI "v_add_f32 %0, 0x4b400000, %2\n\t" RNE "v_sub_f32 %1, %0, %3" : "=&v"(t), "=v"(got) : "v"(p), "v"(g));
        uint32_t a, b, tb;
        memcpy(&a, &want, 4);
        memcpy(&b, &got, 4);
        memcpy(&tb, &t, 4);
        const int fl = (int)want_t;                      // floor as an integer
        if (a != b || (int)(tb - 0x4b400000u) != fl) {
            atomicAdd(bad, 1ull);
            atomicMin(first_bad, bits);
        }
    }
}

// one "blind" DDA step: sample point, floor + h, - c, * i, min3, * d, + c  (22 vector instructions)
#define STEP_CLASSIC(c0, c1, c2)                                                                                   \
    asm volatile("v_add_f32 %3, %0, %6\n\tv_add_f32 %4, %1, %6\n\tv_add_f32 %5, %2, %6\n\t"                      \
                 "v_floor_f32 %3, %3\n\tv_floor_f32 %4, %4\n\tv_floor_f32 %5, %5\n\t"                              \
                 "v_add_f32 %3, %3, %7\n\tv_add_f32 %4, %4, %7\n\tv_add_f32 %5, %5, %7\n\t"                        \
                 "v_sub_f32 %3, %3, %0\n\tv_sub_f32 %4, %4, %1\n\tv_sub_f32 %5, %5, %2\n\t"                        \
                 "v_mul_f32 %3, %3, %8\n\tv_mul_f32 %4, %4, %8\n\tv_mul_f32 %5, %5, %8\n\t"                        \
                 "v_min3_f32 %3, %3, %4, %5\n\t"                                                                   \
                 "v_mul_f32 %4, %3, %9\n\tv_mul_f32 %5, %3, %9\n\tv_mul_f32 %3, %3, %9\n\t"                        \
                 "v_add_f32 %0, %4, %0\n\tv_add_f32 %1, %5, %1\n\tv_add_f32 %2, %3, %2"                            \
                 : "+v"(c0), "+v"(c1), "+v"(c2), "=&v"(t0), "=&v"(t1), "=&v"(t2) : "v"(e), "v"(h), "v"(inv), "v"(d))
#define STEP_MAGIC(c0, c1, c2)                                                                                     \
    asm volatile("v_add_f32 %3, %0, %6\n\tv_add_f32 %4, %1, %6\n\tv_add_f32 %5, %2, %6\n\t"                      \
                 RTNI "v_add_f32 %3, 0x4b400000, %3\n\tv_add_f32 %4, 0x4b400000, %4\n\tv_add_f32 %5, 0x4b400000, %5\n\t" RNE \
                 "v_sub_f32 %3, %3, %7\n\tv_sub_f32 %4, %4, %7\n\tv_sub_f32 %5, %5, %7\n\t"                        \
                 "v_sub_f32 %3, %3, %0\n\tv_sub_f32 %4, %4, %1\n\tv_sub_f32 %5, %5, %2\n\t"                        \
                 "v_mul_f32 %3, %3, %8\n\tv_mul_f32 %4, %4, %8\n\tv_mul_f32 %5, %5, %8\n\t"                        \
                 "v_min3_f32 %3, %3, %4, %5\n\t"                                                                   \
                 "v_mul_f32 %4, %3, %9\n\tv_mul_f32 %5, %3, %9\n\tv_mul_f32 %3, %3, %9\n\t"                        \
                 "v_add_f32 %0, %4, %0\n\tv_add_f32 %1, %5, %1\n\tv_add_f32 %2, %3, %2"                            \
                 : "+v"(c0), "+v"(c1), "+v"(c2), "=&v"(t0), "=&v"(t1), "=&v"(t2) : "v"(e), "v"(g), "v"(inv), "v"(d))
// the keyed step: the cell from three conversions and two shift-ors / from the bits of t
#define KEY_CLASSIC(c0, c1, c2)                                                                                    \
    asm volatile("v_add_f32 %3, %0, %6\n\tv_add_f32 %4, %1, %6\n\tv_add_f32 %5, %2, %6\n\t"                      \
                 "v_cvt_i32_f32 %3, %3\n\tv_cvt_i32_f32 %4, %4\n\tv_cvt_i32_f32 %5, %5\n\t"                        \
                 "v_lshl_or_b32 %3, %4, 10, %3\n\tv_lshl_or_b32 %3, %5, 20, %3\n\t"                                \
                 "v_and_b32 %4, %7, %3\n\tv_cmp_ne_u32 vcc, %4, %8"                                                \
                 : "+v"(c0), "+v"(c1), "+v"(c2), "=&v"(t0), "=&v"(t1), "=&v"(t2) : "v"(e), "v"(wm), "v"(wk) : "vcc")
#define KEY_MAGIC(c0, c1, c2)                                                                                      \
    asm volatile("v_lshl_add_u32 %3, %4, 10, %3\n\tv_lshl_add_u32 %3, %5, 20, %3\n\tv_add_u32 %3, 0xb4c00000, %3\n\t" \
                 "v_and_b32 %4, %6, %3\n\tv_cmp_ne_u32 vcc, %4, %7"                                                \
                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(t0), "+v"(t1), "+v"(t2) : "v"(wm), "v"(wk) : "vcc")

template <int KIND> __global__ void __launch_bounds__(256) steps(float *out, int iters, float seed)
{
    float a0 = seed + threadIdx.x * 0.37f, a1 = a0 + 1.3f, a2 = a0 + 2.1f, b0 = a0 + 0.5f, b1 = a1 + 0.25f, b2 = a2 + 0.125f;
    const float e = seed * 1e-3f, h = 1.0f, g = 12582912.0f - h, inv = 1.7f, d = 0.58f;
    const uint32_t wm = 0xff0fc3f0u, wk = (uint32_t)seed;
    float t0, t1, t2;
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) { STEP_CLASSIC(a0, a1, a2); STEP_CLASSIC(b0, b1, b2); STEP_CLASSIC(a0, a1, a2); STEP_CLASSIC(b0, b1, b2); }
        if (KIND == 1) { STEP_MAGIC(a0, a1, a2); STEP_MAGIC(b0, b1, b2); STEP_MAGIC(a0, a1, a2); STEP_MAGIC(b0, b1, b2); }
        if (KIND == 2) { KEY_CLASSIC(a0, a1, a2); STEP_CLASSIC(a0, a1, a2); KEY_CLASSIC(b0, b1, b2); STEP_CLASSIC(b0, b1, b2);
                         KEY_CLASSIC(a0, a1, a2); STEP_CLASSIC(a0, a1, a2); KEY_CLASSIC(b0, b1, b2); STEP_CLASSIC(b0, b1, b2); }
        if (KIND == 3) {   // the keyed magic step: t is shared by key and advance (here: the advance's own t, then the key from it)
            STEP_MAGIC(a0, a1, a2); KEY_MAGIC(a0, a1, a2); STEP_MAGIC(b0, b1, b2); KEY_MAGIC(b0, b1, b2);
            STEP_MAGIC(a0, a1, a2); KEY_MAGIC(a0, a1, a2); STEP_MAGIC(b0, b1, b2); KEY_MAGIC(b0, b1, b2); }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + b0 + b1 + b2 + t0 + t1 + t2;
}

template <class K> void run(const char *name, K kernel, float *out, double ghz)
{
    const int blocks = 256 * 8, iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    kernel<<<blocks, 256>>>(out, 10, 1.0f);
    hipEventRecord(e0);
    kernel<<<blocks, 256>>>(out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double steps_per_simd = (double)blocks * 4 / 1024.0 * iters * 4;
    printf("%-28s %8.3f ms  %.1f cycles per wave-step per SIMD (at %.2f GHz)\n", name, ms, ms * 1e-3 * ghz * 1e9 / steps_per_simd, ghz);
}

int main()
{
    unsigned long long *bad;
    uint32_t *first_bad;
    hipMalloc(&bad, 8);
    hipMalloc(&first_bad, 4);
    hipMemset(bad, 0, 8);
    hipMemset(first_bad, 0xFF, 4);
    // every 7th float pattern over both signs (|p| < 2^22 is tested inside), then the special neighbourhoods densely
    const uint32_t n = 0x4a800000u / 7 + 1;
    check<<<(n + 255) / 256, 256>>>(0u, 7, n, bad, first_bad);
    check<<<(n + 255) / 256, 256>>>(0x80000000u, 7, n, bad, first_bad);
    const uint32_t dense[] = {0x00000000u, 0x80000000u, 0x3f7ffff0u, 0xbf7ffff0u, 0x447ffff0u, 0xc47ffff0u, 0x4a7ffff0u, 0xca7ffff0u, 0x007ffff0u, 0x807ffff0u};
    for (uint32_t s : dense) check<<<16, 256>>>(s, 1, 4096, bad, first_bad);
    unsigned long long hb = 0;
    uint32_t hf = 0;
    hipDeviceSynchronize();
    hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost);
    hipMemcpy(&hf, first_bad, 4, hipMemcpyDeviceToHost);
    printf("magic floor vs v_floor_f32 + v_add_f32: %llu differences (first at bits 0x%08x)\n", hb, hf);

    float *out;
    if (hipMalloc(&out, 256 * 8 * 256 * 4) != hipSuccess) return 1;
    int khz = 0;
    hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    const double ghz = khz / 1e6;
    for (int rep = 0; rep < 2; ++rep) {
        run("blind step, v_floor", steps<0>, out, ghz);
        run("blind step, magic floor", steps<1>, out, ghz);
        run("keyed step, cvt + v_floor", steps<2>, out, ghz);
        run("keyed step, magic floor", steps<3>, out, ghz);
    }
    return hb != 0;
}
