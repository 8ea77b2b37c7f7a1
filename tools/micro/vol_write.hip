// What write rate do the store shapes of the cost kernel reach?  Two [rows][cols][D] int16 volumes (D = 96: 192 bytes per column) are
// written once, a block of 256 lanes walking a stripe of rows over 64 columns as cost_vpath_kernel does:
//   w0  the kernel's shape: a 16-lane group owns 4 adjacent columns, 12 bytes per lane and column: one store instruction writes four
//       192-byte pieces 768 bytes apart, four instructions fill the 3 KiB of the wavefront's 16 columns;
//   w1  the same bytes as 16 bytes per lane: a 16-lane group writes 256 contiguous bytes per instruction, three instructions per
//       4 columns (the data would need a transpose between lanes first);
//   w2  64 lanes x 16 bytes = 1 KiB contiguous per instruction, three instructions per 16 columns;
//   w0n / w2n  the same with plain instead of non-temporal stores;
//   w0c  w0 with ~COMPUTE dependent vector instructions per row between the stores (a stand-in for the cost arithmetic).
// hipcc --offload-arch=gfx950 -O3 -o vol_write tools/micro/vol_write.hip && ./vol_write
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef unsigned v3u __attribute__((ext_vector_type(3)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

constexpr int D = 96, COLS = 1145, ROW_BYTES = COLS * D * 2, H = 376, STRIPES = 8, SROWS = H / STRIPES, TILES = (COLS + 63) / 64;

template <bool NT, class T> __device__ inline void st(T* p, T v)
{
    if (NT) __builtin_nontemporal_store(v, p); else *p = v;
}

template <int SHAPE, bool NT, int COMPUTE>
__global__ __launch_bounds__(256) void k_w(char* __restrict__ va, char* __restrict__ vb, unsigned seed)
{
    const int tile = blockIdx.x % TILES, stripe = (blockIdx.x / TILES) % STRIPES, frame = blockIdx.x / (TILES * STRIPES);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, grp = lane >> 4, l16 = lane & 15;
    const int col0 = tile * 64 + wave * 16; // the wavefront's 16 columns
    unsigned a = seed + threadIdx.x, b = seed * 3 + blockIdx.x;
    extern __shared__ unsigned lds[];
    for (int r = 0; r < SROWS; r++) {
        if (COMPUTE < 0) { lds[threadIdx.x] = a; __syncthreads(); b += lds[threadIdx.x ^ 64]; } // one barrier per row, as the kernel has
        const size_t row = (size_t)frame * H + stripe * SROWS + r;
        for (int i = 0; i < (COMPUTE < 0 ? -COMPUTE : COMPUTE); i++) { a = a * 1664525u + b; b ^= a >> 7; }
        char* pa = va + row * ROW_BYTES;
        char* pb = vb + row * ROW_BYTES;
        if (SHAPE == 0) {
            for (int j = 0; j < 4; j++) {
                const int c = col0 + grp * 4 + j;
                if (c < COLS) {
                    st<NT>(reinterpret_cast<v3u*>(pa + (size_t)c * 192 + l16 * 12), v3u{a, b, a ^ j});
                    st<NT>(reinterpret_cast<v3u*>(pb + (size_t)c * 192 + l16 * 12), v3u{b, a, b ^ j});
                }
            }
        } else if (SHAPE == 1) {
            for (int j = 0; j < 3; j++) {
                const size_t off = (size_t)(col0 + grp * 4) * 192 + j * 256 + l16 * 16;
                if (off + 16 <= (size_t)ROW_BYTES) {
                    st<NT>(reinterpret_cast<v4u*>(pa + off), v4u{a, b, a ^ j, b});
                    st<NT>(reinterpret_cast<v4u*>(pb + off), v4u{b, a, b ^ j, a});
                }
            }
        } else if (SHAPE == 3) {
            if (a == 0x12345u && b == 77u) *reinterpret_cast<unsigned*>(pa) = a;
        } else {
            for (int j = 0; j < 3; j++) {
                const size_t off = (size_t)col0 * 192 + j * 1024 + lane * 16;
                if (off + 16 <= (size_t)ROW_BYTES) {
                    st<NT>(reinterpret_cast<v4u*>(pa + off), v4u{a, b, a ^ j, b});
                    st<NT>(reinterpret_cast<v4u*>(pb + off), v4u{b, a, b ^ j, a});
                }
            }
        }
    }
}

int main()
{
    const int frames = 42;
    const size_t bytes = (size_t)frames * H * ROW_BYTES;
    char *va, *vb;
    if (hipMalloc(&va, bytes + 4096) != hipSuccess || hipMalloc(&vb, bytes + 4096) != hipSuccess) return 1;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const dim3 grid(frames * STRIPES * TILES);
    auto run = [&](const char* name, auto launch) {
        launch(); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        for (int r = 0; r < 5; r++) launch();
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-72s %7.1f GB/s  %.3f ms\n", name, 5 * 2.0 * (double)bytes / (ms * 1e-3) / 1e9, ms / 5);
    };
#define RUN(name, ...) run(name, [&] { hipLaunchKernelGGL((k_w<__VA_ARGS__>), grid, dim3(256), 0, 0, va, vb, 7u); })
    RUN("w0  12 B per lane, four 192-B pieces per instruction, non-temporal", 0, true, 0);
    RUN("w0n the same, plain stores", 0, false, 0);
    RUN("w1  16 B per lane, 256 B per 16-lane group, non-temporal", 1, true, 0);
    RUN("w2  16 B per lane, 1 KiB per instruction, non-temporal", 2, true, 0);
    RUN("w2n the same, plain stores", 2, false, 0);
    RUN("c   no stores reach memory: 600 dependent instructions per row only", 3, true, 300);
    RUN("w0c w0 + 600 dependent vector instructions per row", 0, true, 300);
    RUN("w1c w1 + 600", 1, true, 300);
    RUN("w2c w2 + 600", 2, true, 300);
    RUN("w0d w0 + 1200", 0, true, 600);
    RUN("w2d w2 + 1200", 2, true, 600);
    // the kernel's regime: 4 wavefronts per SIMD (40 KiB of LDS per block), a barrier per row, ~1.7 ms of arithmetic
#define RUN4(name, ...) run(name, [&] { hipLaunchKernelGGL((k_w<__VA_ARGS__>), grid, dim3(256), 40 * 1024, 0, va, vb, 7u); })
    RUN4("c4   4 waves/SIMD, barrier per row, arithmetic only", 3, true, -210);
    RUN4("w0c4 + w0 stores", 0, true, -210);
    RUN4("w0n4 + w0 stores, plain", 0, false, -210);
    RUN4("w1c4 + w1 stores", 1, true, -210);
    RUN4("w2c4 + w2 stores", 2, true, -210);
    return 0;
}
