// Measured denominators for the rooflines ("do not hard-code; measure on the
// box"): (1) HBM bandwidth of a device-to-device dwordx4 copy, (2) integer VALU issue rate of a
// register-resident XOR + popcount-accumulate loop (the inner operation of the Hamming match).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void copy16(const uint4 *__restrict__ s, uint4 *__restrict__ d, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}
__global__ __launch_bounds__(256) void popc_loop(uint32_t *out, uint32_t seed, int iters)
{
    uint32_t a0 = seed ^ threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
    for (int i = 0; i < iters; i++) {   // 8 VALU per trip per chain pair: 4 v_xor + 4 v_bcnt (accumulating)
        acc0 += __popc(a0 ^ (uint32_t)i); acc1 += __popc(a1 ^ (uint32_t)i);
        acc2 += __popc(a2 ^ (uint32_t)i); acc3 += __popc(a3 ^ (uint32_t)i);
    }
    if (acc0 + acc1 + acc2 + acc3 == 0x7fffffff) out[0] = acc0;
}
int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const size_t bytes = (size_t)2 << 30; uint4 *a, *b; uint32_t *o;
    hipMalloc((void **)&a, bytes); hipMalloc((void **)&b, bytes); hipMalloc((void **)&o, 4);
    hipMemset(a, 1, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int r = 0; r < 6; r++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(copy16, dim3(256 * 16), dim3(256), 0, 0, a, b, bytes / 16);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double copy_gbs = 2.0 * bytes / best / 1e6;
    const int iters = 20000; const int blocks = p.multiProcessorCount * 8;
    float bestp = 1e9;
    for (int r = 0; r < 4; r++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(popc_loop, dim3(blocks), dim3(256), 0, 0, o, 12345u + r, iters);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < bestp) bestp = ms;
    }
    const double wave_instr = (double)blocks * 4 * iters * 8;               // wave64 VALU instructions
    const double lane_ops = wave_instr * 64;
    const double simds = (double)p.multiProcessorCount * 4;
    printf("{\"device\": \"%s\", \"gcn_arch\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"copy_read_plus_write_GBps\": %.0f, "
           "\"xor_popc_lane_ops_per_s\": %.3e, \"wave64_valu_instr_per_s\": %.3e, \"cycles_per_wave64_int_instr_per_simd_at_clock\": %.2f}\n",
           p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000, copy_gbs, lane_ops / (bestp * 1e-3),
           wave_instr / (bestp * 1e-3), simds * (p.clockRate * 1e3) / (wave_instr / (bestp * 1e-3)));
    return 0;
}
