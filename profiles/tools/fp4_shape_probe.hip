// 16x16x128 FP4 vs 32x32x64 FP4: time for the same contraction volume (32 rows x 64 queries x 256 bits per iteration)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void p32(unsigned *out, int iters)
{
    const int lane = threadIdx.x & 63;
    v8i a = {lane, lane * 3, 0x2a2a2a2a, 0x22222222, 0, 0, 0, 0}, b0 = {0x2222aaaa, lane, 0x2a2a2a2a, 0x22a2a222, 0, 0, 0, 0};
    v8i b1 = {0x22a2aaaa, lane * 5, 0x2a2a222a, 0x22a2a222, 0, 0, 0, 0};
    v16f c, acc0, acc1;
    for (int r = 0; r < 16; r++) c[r] = (float)(r + lane);
    unsigned k = ~0u;
    for (int it = 0; it < iters; it++) {
        a[0] ^= it;
        acc0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b0, c, 4, 4, 0, 139, 0, 127);
        acc1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b1, c, 4, 4, 0, 139, 0, 127);
#pragma unroll
        for (int s = 1; s < 4; s++) {
            a[1] += s;
            acc0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b0, acc0, 4, 4, 0, 139, 0, 127);
            acc1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b1, acc1, 4, 4, 0, 139, 0, 127);
        }
        k = min(k, __float_as_uint(acc0[it & 15]) + __float_as_uint(acc1[it & 15]));
    }
    out[blockIdx.x * 256 + threadIdx.x] = k;
}
__global__ __launch_bounds__(256) void p16(unsigned *out, int iters)
{
    const int lane = threadIdx.x & 63;
    v8i a0 = {lane, lane * 3, 0x2a2a2a2a, 0x22222222, 0, 0, 0, 0}, a1 = {lane * 7, lane, 0x2a2a2a2a, 0x22222222, 0, 0, 0, 0};
    v8i b[4];
    for (int q = 0; q < 4; q++) b[q] = v8i{0x2222aaaa + q, lane, 0x2a2a2a2a, 0x22a2a222, 0, 0, 0, 0};
    v4f c, acc[2][4];
    for (int r = 0; r < 4; r++) c[r] = (float)(r + lane);
    unsigned k = ~0u;
    for (int it = 0; it < iters; it++) {
        a0[0] ^= it;
        a1[0] ^= it;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            acc[0][q] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a0, b[q], c, 4, 4, 0, 139, 0, 127);
            acc[1][q] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a1, b[q], c, 4, 4, 0, 139, 0, 127);
        }
        a0[1] += 1;
        a1[1] += 1;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            acc[0][q] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a0, b[q], acc[0][q], 4, 4, 0, 139, 0, 127);
            acc[1][q] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a1, b[q], acc[1][q], 4, 4, 0, 139, 0, 127);
        }
        unsigned s = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) s += __float_as_uint(acc[0][q][it & 3]) + __float_as_uint(acc[1][q][it & 3]);
        k = min(k, s);
    }
    out[blockIdx.x * 256 + threadIdx.x] = k;
}
int main()
{
    unsigned *d;
    (void)hipMalloc(&d, 1 << 24);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int shape : {32, 16})
        for (int wps : {1, 2, 4}) {
            const int iters = 2000;
            for (int rep = 0; rep < 2; rep++) {
                (void)hipEventRecord(e0);
                if (shape == 32) hipLaunchKernelGGL(p32, dim3(256 * wps), dim3(256), 0, 0, d, iters);
                else hipLaunchKernelGGL(p16, dim3(256 * wps), dim3(256), 0, 0, d, iters);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
            }
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            printf("shape %dx%d waves/SIMD %d: %.1f ns per (32 rows x 64 queries x 256 bits) per SIMD\n", shape, shape, wps, ms * 1e6 / (iters * 1.0 * wps));
        }
    return 0;
}
