// valu_rate.hip -- issue rate of the vector instructions the DDA loop is made of, on gfx950.
// Every wave runs N independent chains of one instruction kind; 8 waves per SIMD on all CUs.
// Prints cycles per wave-instruction per SIMD (2 = the SIMD retires 32 lanes per clock).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP 64
#define KERNEL(name, DECL, STEP)                                                          \
    __global__ void __launch_bounds__(256) name(float *out, int iters, float seed)        \
    {                                                                                     \
        DECL;                                                                             \
        for (int i = 0; i < iters; ++i) {                                                 \
            _Pragma("unroll") for (int r = 0; r < REP / 8; ++r) { STEP; }                 \
        }                                                                                 \
        out[blockIdx.x * 256 + threadIdx.x] = sink(a0, a1, a2, a3, a4, a5, a6, a7);       \
    }

__device__ __forceinline__ float sink(float a, float b, float c, float d, float e, float f, float g, float h) { return a + b + c + d + e + f + g + h; }
__device__ __forceinline__ float sink(uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t e, uint32_t f, uint32_t g, uint32_t h) { return (float)(a ^ b ^ c ^ d ^ e ^ f ^ g ^ h); }
__device__ __forceinline__ float sink(int a, int b, int c, int d, int e, int f, int g, int h) { return (float)(a ^ b ^ c ^ d ^ e ^ f ^ g ^ h); }

#define F8 float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; float k = seed * 0.5f
#define U8 uint32_t a0 = (uint32_t)seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; uint32_t k = a0 * 3
#define ALL(op) op(a0); op(a1); op(a2); op(a3); op(a4); op(a5); op(a6); op(a7)

#define ADDF(x) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(k))
#define MULF(x) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(k))
#define FMAF(x) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(k))
#define FLOORF(x) asm volatile("v_floor_f32 %0, %0" : "+v"(x))
#define MIN3F(x) asm volatile("v_min3_f32 %0, %0, %1, %1" : "+v"(x) : "v"(k))
#define CVTI(x) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(x))
#define ANDU(x) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(k))
#define LSHLU(x) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(x))
#define OR3U(x) asm volatile("v_or3_b32 %0, %0, %1, %1" : "+v"(x) : "v"(k))
#define LSHLOR(x) asm volatile("v_lshl_or_b32 %0, %0, 3, %1" : "+v"(x) : "v"(k))
#define ADDU(x) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(k))
#define CMPF(x) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x), "v"(k) : "vcc")
#define CMPU(x) asm volatile("v_cmp_ne_u32 vcc, %0, %1" : : "v"(x), "v"(k) : "vcc")
#define CNDM(x) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(k) : "vcc")
#define MOVU(x) asm volatile("v_mov_b32 %0, %1" : "=v"(x) : "v"(k))
#define MULU24(x) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x) : "v"(k))
#define BFEU(x) asm volatile("v_bfe_u32 %0, %0, 3, 5" : "+v"(x))
#define RCPF(x) asm volatile("v_rcp_f32 %0, %0" : "+v"(x))
#define PKADD(x) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p##x) : "v"(pk))

#define U64x4 unsigned long long b0 = (unsigned long long)seed * 0x9E3779B97F4A7C15ull + threadIdx.x, b1 = b0 + 1, b2 = b0 + 2, b3 = b0 + 3; uint32_t k = threadIdx.x & 31; float a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0
#define LSHR64(x) asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(x) : "v"(k))
#define CMP64(x) asm volatile("v_cmp_eq_u64 vcc, 0, %0" : : "v"(x) : "vcc")
#define ALL4(op) op(b0); op(b1); op(b2); op(b3); op(b0); op(b1); op(b2); op(b3)
#define MADU24(x) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(x) : "v"(k))
#define LSHLADD(x) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(x) : "v"(k))
KERNEL(k_lshrrev_b64, U64x4, ALL4(LSHR64); a0 += (float)(b0 ^ b1 ^ b2 ^ b3))
KERNEL(k_cmp_eq_u64, U64x4, ALL4(CMP64); a0 += (float)(b0 ^ b1 ^ b2 ^ b3))
KERNEL(k_mad_u32_u24, U8, ALL(MADU24))
KERNEL(k_lshl_add_u32, U8, ALL(LSHLADD))
KERNEL(k_add_f32, F8, ALL(ADDF))
KERNEL(k_mul_f32, F8, ALL(MULF))
KERNEL(k_fma_f32, F8, ALL(FMAF))
KERNEL(k_floor_f32, F8, ALL(FLOORF))
KERNEL(k_min3_f32, F8, ALL(MIN3F))
KERNEL(k_cvt_i32_f32, F8, ALL(CVTI))
KERNEL(k_rcp_f32, F8, ALL(RCPF))
KERNEL(k_cmp_f32, F8, ALL(CMPF))
KERNEL(k_and_b32, U8, ALL(ANDU))
KERNEL(k_lshl_b32, U8, ALL(LSHLU))
KERNEL(k_or3_b32, U8, ALL(OR3U))
KERNEL(k_lshl_or_b32, U8, ALL(LSHLOR))
KERNEL(k_add_u32, U8, ALL(ADDU))
KERNEL(k_cmp_u32, U8, ALL(CMPU))
KERNEL(k_cndmask, U8, ALL(CNDM))
KERNEL(k_mov_b32, U8, ALL(MOVU))
KERNEL(k_mul_u24, U8, ALL(MULU24))
KERNEL(k_bfe_u32, U8, ALL(BFEU))

__global__ void __launch_bounds__(256) k_pk_add_f32(float *out, int iters, float seed)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {seed, seed}, p1 = p0 + 1.f, p2 = p0 + 2.f, p3 = p0 + 3.f, p4 = p0 + 4.f, p5 = p0 + 5.f, p6 = p0 + 6.f, p7 = p0 + 7.f, pk = p0 * 0.5f;
    for (int i = 0; i < iters; ++i) {
        #pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p0) : "v"(pk)); asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p1) : "v"(pk));
            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p2) : "v"(pk)); asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p3) : "v"(pk));
            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p4) : "v"(pk)); asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p5) : "v"(pk));
            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p6) : "v"(pk)); asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p7) : "v"(pk));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}

// the same chains with part of the wavefront switched off: does a vector instruction cost less
// when 32 consecutive lanes (or every other lane) are inactive?
#define KERNEL_MASKED(name, COND, DECL, STEP)                                             \
    __global__ void __launch_bounds__(256) name(float *out, int iters, float seed)        \
    {                                                                                     \
        DECL;                                                                             \
        const int lane = threadIdx.x & 63;                                                \
        if (COND) {                                                                       \
            for (int i = 0; i < iters; ++i) {                                             \
                _Pragma("unroll") for (int r = 0; r < REP / 8; ++r) { STEP; }             \
            }                                                                             \
        }                                                                                 \
        out[blockIdx.x * 256 + threadIdx.x] = sink(a0, a1, a2, a3, a4, a5, a6, a7);       \
    }
KERNEL_MASKED(k_add_f32_low32, lane < 32, F8, ALL(ADDF))
KERNEL_MASKED(k_add_f32_low16, lane < 16, F8, ALL(ADDF))
KERNEL_MASKED(k_add_f32_even, (lane & 1) == 0, F8, ALL(ADDF))
KERNEL_MASKED(k_add_f32_1of4rows, (lane & 8) == 0 && lane < 32, F8, ALL(ADDF))
KERNEL_MASKED(k_floor_f32_low32, lane < 32, F8, ALL(FLOORF))
KERNEL_MASKED(k_floor_f32_low16, lane < 16, F8, ALL(FLOORF))
KERNEL_MASKED(k_floor_f32_even, (lane & 1) == 0, F8, ALL(FLOORF))
KERNEL_MASKED(k_cvt_low32, lane < 32, F8, ALL(CVTI))
KERNEL_MASKED(k_lshl_or_low32, lane < 32, U8, ALL(LSHLOR))

template <class K> void run(const char *name, K kernel, float *out, double ghz)
{
    const int blocks = 256 * 8, iters = 2000;             // 8 blocks of 4 waves per CU = 8 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    kernel<<<blocks, 256>>>(out, 10, 1.0f);
    hipEventRecord(e0);
    kernel<<<blocks, 256>>>(out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr_per_simd = (double)blocks * 4 / 1024.0 * iters * REP;
    printf("%-16s %8.3f ms  %.2f cycles per wave-instruction per SIMD (at %.2f GHz)\n", name, ms, ms * 1e-3 * ghz * 1e9 / wave_instr_per_simd, ghz);
}

int main()
{
    float *out;
    if (hipMalloc(&out, 256 * 8 * 256 * 4) != hipSuccess) return 1;
    int khz = 0;
    hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    const double ghz = khz / 1e6;
#define RUN(k) run(#k, k, out, ghz)
    RUN(k_add_f32); RUN(k_mul_f32); RUN(k_fma_f32); RUN(k_floor_f32); RUN(k_min3_f32); RUN(k_cvt_i32_f32); RUN(k_rcp_f32); RUN(k_cmp_f32);
    RUN(k_and_b32); RUN(k_lshl_b32); RUN(k_or3_b32); RUN(k_lshl_or_b32); RUN(k_add_u32); RUN(k_cmp_u32); RUN(k_cndmask); RUN(k_mov_b32);
    RUN(k_mul_u24); RUN(k_bfe_u32); RUN(k_pk_add_f32);
    RUN(k_lshrrev_b64); RUN(k_cmp_eq_u64); RUN(k_mad_u32_u24); RUN(k_lshl_add_u32);
    RUN(k_add_f32_low32); RUN(k_add_f32_low16); RUN(k_add_f32_even); RUN(k_add_f32_1of4rows);
    RUN(k_floor_f32_low32); RUN(k_floor_f32_low16); RUN(k_floor_f32_even); RUN(k_cvt_low32); RUN(k_lshl_or_low32);
    return 0;
}
