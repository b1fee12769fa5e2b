// Probe: what limits a lane-per-row 32-B streaming read on MI355X?  Variants:
//  A: lane reads its row as two 16-B loads (stride 32 B across lanes)   [the match_stream shape]
//  B: wave reads 2 KB as two contiguous 1-KB instructions (16 B per lane, stride 16 B)
//  C: as A but grid-stride over the whole array instead of one contiguous chunk per block
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(256) void probeA(const uint4 *t, int nt, int chunk_len, uint32_t *out)
{
    const int c0 = blockIdx.x * chunk_len, c1 = min(c0 + chunk_len, nt);
    uint32_t acc = 0;
    for (int j = c0 + (int)threadIdx.x; j < c1; j += 256) {
        const uint4 lo = t[(size_t)j * 2], hi = t[(size_t)j * 2 + 1];
        acc += __popc(lo.x ^ hi.x) + __popc(lo.y ^ hi.y) + __popc(lo.z ^ hi.z) + __popc(lo.w ^ hi.w);
    }
    if (acc == 0x12345) out[0] = acc;
}
__global__ __launch_bounds__(256) void probeB(const uint4 *t, int nt, int chunk_len, uint32_t *out)
{
    const int c0 = blockIdx.x * chunk_len, c1 = min(c0 + chunk_len, nt);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t acc = 0;
    for (int j = c0 + wave * 64; j < c1; j += 256) { // 64 rows = 2 KB per wave trip
        const uint4 a = t[(size_t)j * 2 + lane], b = t[(size_t)j * 2 + 64 + lane];
        acc += __popc(a.x ^ b.x) + __popc(a.y ^ b.y) + __popc(a.z ^ b.z) + __popc(a.w ^ b.w);
    }
    if (acc == 0x12345) out[0] = acc;
}
__global__ __launch_bounds__(256) void probeC(const uint4 *t, int nt, int chunk_len, uint32_t *out)
{
    uint32_t acc = 0;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < nt; j += gridDim.x * 256) {
        const uint4 lo = t[(size_t)j * 2], hi = t[(size_t)j * 2 + 1];
        acc += __popc(lo.x ^ hi.x) + __popc(lo.y ^ hi.y) + __popc(lo.z ^ hi.z) + __popc(lo.w ^ hi.w);
    }
    if (acc == 0x12345) out[0] = acc;
}
int main()
{
    const int nt = 20000000;
    uint4 *t; uint32_t *o;
    hipMalloc((void **)&t, (size_t)nt * 32); hipMalloc((void **)&o, 4);
    hipMemset(t, 7, (size_t)nt * 32);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int blocks : {2048, 4096, 16384}) {
        int chunk = ((nt + blocks - 1) / blocks + 255) & ~255;
        int nb = (nt + chunk - 1) / chunk;
        for (int v = 0; v < 3; v++) {
            float best = 1e9;
            for (int rep = 0; rep < 5; rep++) {
                hipEventRecord(a, 0);
                if (v == 0) hipLaunchKernelGGL(probeA, dim3(nb), dim3(256), 0, 0, t, nt, chunk, o);
                if (v == 1) hipLaunchKernelGGL(probeB, dim3(nb), dim3(256), 0, 0, t, nt, chunk, o);
                if (v == 2) hipLaunchKernelGGL(probeC, dim3(nb), dim3(256), 0, 0, t, nt, chunk, o);
                hipEventRecord(b, 0); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
            }
            printf("blocks %5d variant %c: %.3f ms  %.0f GB/s\n", nb, 'A' + v, best, (double)nt * 32 / best / 1e6);
        }
    }
    return 0;
}
