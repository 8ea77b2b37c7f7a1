// Achievable HBM rates of plain streaming kernels on this GPU: read-only, write-only (normal and non-temporal stores),
// copy, and a 1:4 read:write mix like the SGBM cost kernel's.  2 GiB buffers, grid-stride loops, 16 bytes per lane.
//   hipcc --offload-arch=gfx950 -O3 -o hbm_rate tools/micro/hbm_rate.hip && ./hbm_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef unsigned v4u __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_read(const v4u* __restrict__ a, v4u* __restrict__ out, size_t n)
{
    v4u acc = {0, 0, 0, 0};
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc ^= a[i];
    if (acc.x == 0x12345 && acc.y == 0x54321) out[threadIdx.x] = acc;
}
__global__ __launch_bounds__(256) void k_write(v4u* __restrict__ a, size_t n, unsigned s)
{
    v4u v = {s, s + 1, s + 2, s + 3};
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a[i] = v;
}
__global__ __launch_bounds__(256) void k_write_nt(v4u* __restrict__ a, size_t n, unsigned s)
{
    v4u v = {s, s + 1, s + 2, s + 3};
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) __builtin_nontemporal_store(v, a + i);
}
__global__ __launch_bounds__(256) void k_copy(const v4u* __restrict__ a, v4u* __restrict__ b, size_t n)
{
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) b[i] = a[i];
}
__global__ __launch_bounds__(256) void k_copy_nt(const v4u* __restrict__ a, v4u* __restrict__ b, size_t n)
{
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) __builtin_nontemporal_store(a[i], b + i);
}
// reads n/4 elements, writes n
__global__ __launch_bounds__(256) void k_mix(const v4u* __restrict__ a, v4u* __restrict__ b, size_t n)
{
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n / 4; i += (size_t)gridDim.x * 256) {
        v4u v = a[i];
        for (int q = 0; q < 4; q++) { __builtin_nontemporal_store(v, b + (size_t)q * (n / 4) + i); v.x += 1; }
    }
}

int main()
{
    const size_t bytes = 2ull << 30, n = bytes / 16;
    v4u *a, *b;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) return 1;
    (void)hipMemset(a, 1, bytes); (void)hipMemset(b, 2, bytes);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grid = 256 * 16;
    auto run = [&](const char* name, double moved, auto launch) {
        launch(); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        for (int r = 0; r < 5; r++) launch();
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-28s %7.1f GB/s\n", name, 5 * moved / (ms * 1e-3) / 1e9);
    };
    run("read", bytes, [&] { hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, a, b, n); });
    run("write", bytes, [&] { hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, a, n, 7u); });
    run("write non-temporal", bytes, [&] { hipLaunchKernelGGL(k_write_nt, dim3(grid), dim3(256), 0, 0, a, n, 7u); });
    run("copy (read + write)", 2.0 * bytes, [&] { hipLaunchKernelGGL(k_copy, dim3(grid), dim3(256), 0, 0, a, b, n); });
    run("copy, non-temporal stores", 2.0 * bytes, [&] { hipLaunchKernelGGL(k_copy_nt, dim3(grid), dim3(256), 0, 0, a, b, n); });
    run("1 read : 4 writes, nt", 1.25 * bytes, [&] { hipLaunchKernelGGL(k_mix, dim3(grid), dim3(256), 0, 0, a, b, n); });
    return 0;
}
