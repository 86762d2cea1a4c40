// exhaustive_math.hip -- which short instruction sequences reproduce IEEE-754 binary32
// 1/d, sqrt(x) and n/255 bit for bit on gfx950?  Runs every one of the 2^32 float bit patterns
// through each candidate and through the compiler's correctly rounded expansion, and prints,
// per candidate, the number of mismatches inside and outside the range the kernels would use
// it in (2^-100 <= |d| <= 2^100), plus a few offending inputs.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/exhaustive_math tools/exhaustive_math.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>

constexpr int kCand = 8;

__device__ __forceinline__ float rcp_a(float d) { float r = __builtin_amdgcn_rcpf(d); float e = __builtin_fmaf(-d, r, 1.0f); return __builtin_fmaf(e, r, r); }
__device__ __forceinline__ float rcp_b(float d) { float r = rcp_a(d); float e = __builtin_fmaf(-d, r, 1.0f); return __builtin_fmaf(e, r, r); }
// Markstein: refine r, then q = r, residual of q against 1, final fma
__device__ __forceinline__ float rcp_m(float d) { float r = rcp_a(d); float q = r; float e = __builtin_fmaf(-d, q, 1.0f); return __builtin_fmaf(e, r, q); }
__device__ __forceinline__ float sqrt_c(float x)
{
    float s = __builtin_amdgcn_sqrtf(x);
    float sm = __uint_as_float(__float_as_uint(s) - 1u), sp = __uint_as_float(__float_as_uint(s) + 1u);
    float em = __builtin_fmaf(-sm, s, x), ep = __builtin_fmaf(-sp, s, x);
    float r = (em <= 0.0f) ? sm : s;
    return (ep > 0.0f) ? sp : r;
}
__device__ __forceinline__ float sqrt_n(float x)
{
    float s = __builtin_amdgcn_sqrtf(x);
    float h = 0.5f * __builtin_amdgcn_rcpf(s);
    float e = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(e, h, s);
}

__device__ __forceinline__ bool same(float a, float b)
{
    uint32_t ua = __float_as_uint(a), ub = __float_as_uint(b);
    return ua == ub || ((a != a) && (b != b));
}

__global__ void sweep(unsigned long long *bad_in, unsigned long long *bad_out, uint32_t *example)
{
    const uint32_t lo = 0x0D800000u, hi = 0x71800000u;   // 2^-100, 2^100
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t u = (uint32_t)i;
        const float d = __uint_as_float(u);
        const uint32_t mag = u & 0x7FFFFFFFu;
        const bool in = mag >= lo && mag <= hi;
        const float ref_r = 1.0f / d;
        const float ref_s = sqrtf(d);
        float c[kCand];
        c[0] = rcp_a(d);
        c[1] = rcp_b(d);
        c[2] = rcp_m(d);
        c[3] = __builtin_amdgcn_div_fixupf(rcp_a(d), d, 1.0f);
        c[4] = __builtin_amdgcn_div_fixupf(rcp_b(d), d, 1.0f);
        c[5] = __builtin_amdgcn_div_fixupf(rcp_m(d), d, 1.0f);
        c[6] = sqrt_c(d);
        c[7] = sqrt_n(d);
        for (int k = 0; k < kCand; ++k) {
            const float ref = k < 6 ? ref_r : ref_s;
            const bool in_k = k < 6 ? in : (in && !(u >> 31));
            if (!same(c[k], ref)) {
                atomicAdd(in_k ? &bad_in[k] : &bad_out[k], 1ull);
                if (in_k) atomicMin(&example[k], mag);
            }
        }
    }
}

__global__ void over255(uint32_t *bad)
{
    const int n = threadIdx.x;                            // 0..255
    const float r = 1.0f / 255.0f;
    const float x = (float)n;
    float q = x * r;
    float e = __builtin_fmaf(-255.0f, q, x);
    float q1 = __builtin_fmaf(e, r, q);
    if (!same(q1, x / 255.0f)) atomicAdd(&bad[0], 1u);
    if (!same(q, x / 255.0f)) atomicAdd(&bad[1], 1u);
}

int main()
{
    unsigned long long *bi, *bo; uint32_t *ex, *b255;
    hipMalloc(&bi, kCand * 8); hipMalloc(&bo, kCand * 8); hipMalloc(&ex, kCand * 4); hipMalloc(&b255, 8);
    hipMemset(bi, 0, kCand * 8); hipMemset(bo, 0, kCand * 8); hipMemset(ex, 0xFF, kCand * 4); hipMemset(b255, 0, 8);
    sweep<<<4096, 256>>>(bi, bo, ex);
    over255<<<1, 256>>>(b255);
    unsigned long long hbi[kCand], hbo[kCand]; uint32_t hex_[kCand], h255[2];
    if (hipMemcpy(hbi, bi, sizeof hbi, hipMemcpyDeviceToHost) != hipSuccess) { printf("hip error\n"); return 1; }
    hipMemcpy(hbo, bo, sizeof hbo, hipMemcpyDeviceToHost); hipMemcpy(hex_, ex, sizeof hex_, hipMemcpyDeviceToHost);
    hipMemcpy(h255, b255, sizeof h255, hipMemcpyDeviceToHost);
    const char *names[kCand] = {"rcp_a", "rcp_b", "rcp_markstein", "rcp_a+fixup", "rcp_b+fixup", "rcp_markstein+fixup", "sqrt_candidates", "sqrt_newton"};
    for (int k = 0; k < kCand; ++k)
        printf("%-22s mismatches in range %llu, outside %llu, smallest offending |input| 0x%08x\n", names[k], hbi[k], hbo[k], hex_[k]);
    printf("n/255 refined: %u mismatches of 256; n*(1/255) alone: %u\n", h255[0], h255[1]);
    return 0;
}
