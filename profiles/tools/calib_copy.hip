// Calibration kernel for rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md, HBM
// section: FETCH_SIZE reads 1/2 of the bytes of a 16-B/lane streaming read; other access widths
// must be calibrated on a known byte count).  Copies N bytes with the access shape our image
// kernels use: one aligned dword (4 B) per lane, consecutive lanes consecutive addresses.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void copy_dword(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i] + 1u;
}
__global__ void copy_dwordx4(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { uint4 v = src[i]; v.x += 1u; dst[i] = v; }
}

int main(int argc, char **argv)
{
    const size_t bytes = (argc > 1 ? (size_t)atol(argv[1]) : 1024) << 20; // MiB
    uint32_t *a, *b;
    if (hipMalloc((void **)&a, bytes) != hipSuccess || hipMalloc((void **)&b, bytes) != hipSuccess) return 1;
    hipMemset(a, 1, bytes);
    hipMemset(b, 0, bytes);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(copy_dword, dim3((unsigned)((bytes / 4 + 255) / 256)), dim3(256), 0, 0, a, b, bytes / 4);
        hipLaunchKernelGGL(copy_dwordx4, dim3((unsigned)((bytes / 16 + 255) / 256)), dim3(256), 0, 0, (const uint4 *)a, (uint4 *)b, bytes / 16);
    }
    hipDeviceSynchronize();
    printf("copied %zu bytes per launch (read) + %zu (write)\n", bytes, bytes);
    return 0;
}
