// What the i8 matrix pipe sustains in the shapes k_match_mfma_x uses (MI355X):
//   mode 0: 9 dependent v_mfma_i32_32x32x32_i8 per iteration, nothing else
//   mode 1: + the 32 VALU (v_med3_u32 + v_min_u32 per accumulator element) of the best / second-best selection
//   mode 2: + 8 ds_read_b128 of the A fragments per iteration (conflict-free 272-byte pitch)
//   mode 3: + a workgroup barrier per iteration
// at 1, 2, 4, 5 waves per SIMD (blocks of 256 threads, grid = 256 CUs x waves).  Prints cycles per MFMA per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_probe mfma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int MODE> __global__ __launch_bounds__(256) void probe(int iters, uint32_t *out, const v4i *src)
{
    __shared__ __attribute__((aligned(16))) uint8_t tile[32 * 272 + 1024];
    const int lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5;
    for (int i = threadIdx.x; i < (32 * 272) / 16; i += 256) ((v4i *)tile)[i] = src[i & 63];
    __syncthreads();
    v4i b[8];
    for (int s = 0; s < 8; s++) b[s] = src[(lane + s) & 63];
    v16i crow;
    for (int r = 0; r < 16; r++) crow[r] = r + lane;
    const v4i a_step = v4i{half == 0 ? 4 : 0, 0, 0, 0}, b_step = v4i{half == 0 ? 8 : 0, 0, 0, 0};
    uint32_t k1[2] = {~0u, ~0u}, k2[2] = {~0u, ~0u};
    v4i a[8];
    for (int s = 0; s < 8; s++) a[s] = src[(lane * 3 + s) & 63];
    for (int it = 0; it < iters; it++) {
        if (MODE >= 3) __syncthreads();
        if (MODE >= 2) {
            const uint8_t *arow = &tile[col * 272 + 16 * half];
#pragma unroll
            for (int s = 0; s < 8; s++) a[s] = *(const v4i *)(arow + 32 * s);
        }
        v16i acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[0], b[0], crow, 0, 0, 0);
#pragma unroll
        for (int s = 1; s < 8; s++) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[s], b[s], acc, 0, 0, 0);
        crow = __builtin_amdgcn_mfma_i32_32x32x32_i8(a_step, b_step, crow, 0, 0, 0);
        if (MODE >= 1) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const uint32_t key = (uint32_t)acc[r];
                const int c = r & 1;
                k2[c] = min(max(k1[c], k2[c]), max(min(k1[c], k2[c]), key));
                k1[c] = min(k1[c], key);
            }
        } else {
            k1[0] ^= (uint32_t)acc[0];
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = k1[0] ^ k1[1] ^ k2[0] ^ k2[1] ^ (uint32_t)crow[3];
}

int main()
{
    uint32_t *out;
    v4i *src;
    hipMalloc(&out, 256 * 8 * 256 * 4);
    hipMalloc(&src, 64 * 16);
    uint32_t h[256];
    for (int i = 0; i < 256; i++) h[i] = 0x40C040C0u ^ (i * 0x9E3779B9u & 0x80808080u);
    hipMemcpy(src, h, 1024, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 2000;
    for (int mode = 0; mode < 4; mode++)
        for (int wps : {1, 2, 4, 5}) {
            const int blocks = 256 * wps;
            float best = 1e9f;
            for (int rep = 0; rep < 3; rep++) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 0, 0, iters, out, src);
                if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, iters, out, src);
                if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(256), 0, 0, iters, out, src);
                if (mode == 3) hipLaunchKernelGGL(probe<3>, dim3(blocks), dim3(256), 0, 0, iters, out, src);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double mfma_per_simd = (double)iters * 9 * wps;
            printf("mode %d waves/SIMD %d: %.3f ms, %.1f ns per MFMA per SIMD (= %.1f cycles at 2.4 GHz), %.2f Pop/s int8\n", mode, wps, best,
                   best * 1e6 / mfma_per_simd, best * 1e6 / mfma_per_simd * 2.4, 1024.0 * mfma_per_simd * 65536 / (best * 1e-3) / 1e15);
        }
    return 0;
}
