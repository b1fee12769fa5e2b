// Does v_mfma_scale_f32_32x32x64_f8f6f4 with FP4 (e2m1) operands and E8M0 block scales give the exact Hamming
// contraction k_match needs?  A row = 64 nibbles (+1 = 0x2, -1 = 0xA), scale 2^12 on A: every product +-4096, so
// acc = C + 4096 * sum(+-1) exactly in f32.  Prints mismatches against the host's count and the time of a loop.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__global__ void probe(const uint32_t *a_bits, const uint32_t *b_bits, float *out, int iters)
{
    // row m = lane & 31 of A (bits a_bits[m*2 + half]: 32 bits = this lane's 32 K-values), column n = lane & 31 of B
    const int lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5;
    auto expand = [](uint32_t bits, int *dst) {  // 32 bits -> 32 nibbles = 4 dwords
        for (int w = 0; w < 4; w++) {
            uint32_t v = 0;
            for (int i = 0; i < 8; i++) v |= (((bits >> (8 * w + i)) & 1u) ? 0x2u : 0xAu) << (4 * i);
            dst[w] = (int)v;
        }
    };
    int ta[4], tb[4];
    expand(a_bits[col * 2 + half], ta);
    expand(b_bits[col * 2 + half], tb);
    const v8i a = {ta[0], ta[1], ta[2], ta[3], 0, 0, 0, 0}, b = {tb[0], tb[1], tb[2], tb[3], 0, 0, 0, 0};
    v16f c;
    for (int r = 0; r < 16; r++) c[r] = (float)(1 << 20) + (float)(4 * half + (r & 3) + 8 * (r >> 2));
    const int scale_a = 127 + 12, scale_b = 127;  // E8M0: 2^12 and 1
    v16f acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 4, 4, 0, scale_a, 0, scale_b);
    for (int it = 1; it < iters; it++) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 4, 4, 0, scale_a, 0, scale_b);
    for (int r = 0; r < 16; r++) out[(blockIdx.x * 64 + lane) * 16 + r] = acc[r];
}

int main()
{
    uint32_t ha[64], hb[64];
    srand(1);
    for (int i = 0; i < 64; i++) { ha[i] = (uint32_t)rand() * 2654435761u; hb[i] = (uint32_t)rand() * 40503u + (uint32_t)rand(); }
    uint32_t *da, *db;
    float *dout;
    hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dout, 1024 * 64 * 16 * 4);
    hipMemcpy(da, ha, 256, hipMemcpyHostToDevice); hipMemcpy(db, hb, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dout, 1);
    float h[1024];
    hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; lane++)
        for (int r = 0; r < 16; r++) {
            const int n = lane & 31, m = 4 * (lane >> 5) + (r & 3) + 8 * (r >> 2);  // C layout: lane = column, registers = rows
            int dot = 0;
            for (int k = 0; k < 64; k++) {
                const int ab = (ha[m * 2 + k / 32] >> (k % 32)) & 1, bb = (hb[n * 2 + k / 32] >> (k % 32)) & 1;
                dot += (ab ? 1 : -1) * (bb ? 1 : -1);
            }
            const float want = (float)(1 << 20) + (float)m + 4096.0f * dot;
            if (h[lane * 16 + r] != want) { if (bad < 5) printf("lane %d r %d: got %f want %f\n", lane, r, h[lane * 16 + r], want); bad++; }
        }
    printf("mismatches: %d of 1024\n", bad);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps : {1, 4}) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe, dim3(256 * wps), dim3(256), 0, 0, da, db, dout, 20000);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("waves/SIMD %d: %.2f ns per fp4 32x32x64 MFMA per SIMD (%.1f cycles at 2.4 GHz)\n", wps, ms * 1e6 / (20000.0 * wps), ms * 1e6 / (20000.0 * wps) * 2.4);
    }
    return 0;
}
