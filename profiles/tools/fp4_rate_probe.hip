// FP4 MFMA issue-rate probe in the matcher's shape (k_match_mfma_x): per wave and iteration two accumulation chains of four
// dependent v_mfma_scale_f32_32x32x64_f8f6f4 with a fresh C input, optionally followed by the matcher's selection on the
// PREVIOUS iteration's results (16 x v_min3_u32 + 4 x v_sub_f32 + 2 x (v_med3_u32 + v_min_u32)); W waves per SIMD.
// Prints ns per MFMA per SIMD.  The A operand changes every iteration so that nothing is loop-invariant.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
template <int SEL>
__global__ __launch_bounds__(256) void probe(unsigned *out, int iters)
{
    const int lane = threadIdx.x & 63;
    v8i a = {lane, lane * 3, 0x2a2a2a2a, 0x22222222, 0, 0, 0, 0}, b0 = {0x2222aaaa, lane, 0x2a2a2a2a, 0x22a2a222, 0, 0, 0, 0};
    v8i b1 = {0x22a2aaaa, lane * 5, 0x2a2a222a, 0x22a2a222, 0, 0, 0, 0};
    v16f c, acc0, acc1, p0, p1;
    for (int r = 0; r < 16; r++) { c[r] = (float)(r + lane); p0[r] = p1[r] = 0.f; }
    unsigned k1a = ~0u, k2a = ~0u, k1b = ~0u, k2b = ~0u;
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
        if (SEL) {
            unsigned m = min(min(__float_as_uint(p0[0]), __float_as_uint(p0[1])), __float_as_uint(p0[2]));
            unsigned n = min(min(__float_as_uint(p1[0]), __float_as_uint(p1[1])), __float_as_uint(p1[2]));
#pragma unroll
            for (int r = 3; r < 15; r += 2) {
                m = min(min(m, __float_as_uint(p0[r])), __float_as_uint(p0[r + 1]));
                n = min(min(n, __float_as_uint(p1[r])), __float_as_uint(p1[r + 1]));
            }
            m = min(m, __float_as_uint(p0[15]));
            n = min(n, __float_as_uint(p1[15]));
            k1a = __float_as_uint(__uint_as_float(k1a) - 32.f); k2a = __float_as_uint(__uint_as_float(k2a) - 32.f);
            k1b = __float_as_uint(__uint_as_float(k1b) - 32.f); k2b = __float_as_uint(__uint_as_float(k2b) - 32.f);
            k2a = min(max(k1a, k2a), max(min(k1a, k2a), m)); k1a = min(k1a, m);
            k2b = min(max(k1b, k2b), max(min(k1b, k2b), n)); k1b = min(k1b, n);
        } else {
            k1a = min(k1a, __float_as_uint(p0[it & 15]));
            k1b = min(k1b, __float_as_uint(p1[it & 15]));
        }
        p0 = acc0;
        p1 = acc1;
    }
    out[blockIdx.x * 256 + threadIdx.x] = k1a + k2a + k1b + k2b + __float_as_uint(p0[3]) + __float_as_uint(p1[5]);
}
int main()
{
    unsigned *d;
    (void)hipMalloc(&d, 1 << 24);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int sel : {0, 1})
        for (int wps : {1, 2, 3, 4}) {
            const int iters = 2000;
            for (int rep = 0; rep < 2; rep++) {
                (void)hipEventRecord(e0);
                if (sel) hipLaunchKernelGGL(probe<1>, dim3(256 * wps), dim3(256), 0, 0, d, iters);
                else hipLaunchKernelGGL(probe<0>, dim3(256 * wps), dim3(256), 0, 0, d, iters);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
            }
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            printf("selection %d waves/SIMD %d: %.2f ns per MFMA per SIMD (%.2f us per 504 MFMAs per wave)\n", sel, wps, ms * 1e6 / (iters * 8.0 * wps),
                   ms * 1e3 / (iters * 8.0) * 504);
        }
    return 0;
}
