// What read rate do the access shapes of the horizontal-path kernels reach?  A [rows][cols][D] int16 cost volume (D = 96: 192 bytes
// per column, 219 840 bytes per row at KITTI shape, rows far apart) is read once by
//   v0  the shape of hpath_ckpt_kernel: a wavefront takes 4 rows, 16 lanes per row, 12 bytes per lane and column (dwordx3), 8 columns
//       per group, next group requested while the current one is consumed;
//   v1  the same rows per wavefront, but 16 bytes per lane: a 16-lane group reads 256 contiguous bytes of its row per instruction
//       (6 instructions = 8 columns); the data would then need a transpose to the 12-bytes-per-lane compute layout;
//   v2  one row per wavefront, 64 lanes x 16 bytes = 1 KiB contiguous per instruction (a plain stream);
//   v3  v0 with 16 columns per group;
//   v4  v1 with the group of 8 columns requested TWO groups ahead;
//   v5  v0 plus the checkpoint pass's stores (16 bytes per lane every 16 columns), v6 the same with non-temporal stores;
//   v7-v13  the same stores in other shapes (one KiB contiguous per wavefront, 12 bytes per lane, four segments per burst, one
//       dwordx4, a quarter of them, into 1 MiB that stays in cache, 4 bytes per lane), v14 with no branch in the loop,
//   v15/v16  on a volume with a checkpoint slot behind every 16 columns of a row: stores elsewhere / into those slots;
// each with a trivial consumer (xor).  hipcc --offload-arch=gfx950 -O3 -o row_stream tools/micro/row_stream.hip && ./row_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef unsigned v3u __attribute__((ext_vector_type(3)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

constexpr int D = 96, COLS = 1145, ROW_BYTES = COLS * D * 2;

template <int U, int STORE = 0>
__global__ __launch_bounds__(256) void k_v0(const char* __restrict__ vol, int rows, unsigned* out, unsigned* ck = nullptr)
{
    const int lane16 = threadIdx.x & 15, y = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (y >= rows) return;
    const char* p = vol + (size_t)y * ROW_BYTES + lane16 * 12;
    v3u cur[U], nxt[U], acc = {0, 0, 0};
#pragma unroll
    for (int u = 0; u < U; u++) cur[u] = *reinterpret_cast<const v3u*>(p + (size_t)u * 192);
    for (int x0 = 0; x0 < COLS; x0 += U) {
        if (x0 + U < COLS)
            for (int u = 0; u < U; u++) nxt[u] = *reinterpret_cast<const v3u*>(p + (size_t)min(x0 + U + u, COLS - 1) * 192);
        for (int u = 0; u < U; u++) acc ^= cur[u];
        for (int u = 0; u < U; u++) cur[u] = nxt[u];
        if (STORE == 3 && (x0 % 16) == 0) { // one contiguous KiB per wavefront and store instruction
            unsigned* q = ck + (((size_t)(y >> 2) * 72 + x0 / 16) * 64 + (threadIdx.x & 63)) * 4;
            *reinterpret_cast<v4u*>(q) = v4u{acc.x, acc.y, acc.z, acc.x ^ acc.y};
        } else if (STORE == 4 && (x0 % 16) == 0) { // 12 bytes per lane
            unsigned* q = ck + (((size_t)y * 72 + x0 / 16) * 16 + lane16) * 3;
            *reinterpret_cast<v3u*>(q) = acc;
        } else if (STORE == 5 && (x0 % 64) == 0) { // four segments' worth at once, every 64 columns
            unsigned* q = ck + (((size_t)y * 18 + x0 / 64) * 16 + lane16) * 16;
            for (int i = 0; i < 4; i++) *reinterpret_cast<v4u*>(q + 4 * i) = v4u{acc.x + i, acc.y, acc.z, acc.x ^ acc.y};
        } else if (STORE == 6 && (x0 % 16) == 0) { // 16 bytes per lane as one dwordx4 (v5 writes four dwords)
            unsigned* q = ck + (((size_t)y * 72 + x0 / 16) * 16 + lane16) * 4;
            *reinterpret_cast<v4u*>(q) = v4u{acc.x, acc.y, acc.z, acc.x ^ acc.y};
        } else if (STORE == 7 && (x0 % 64) == 0) { // a quarter of the stores: 16 bytes per lane every 64 columns
            unsigned* q = ck + (((size_t)y * 18 + x0 / 64) * 16 + lane16) * 4;
            *reinterpret_cast<v4u*>(q) = v4u{acc.x, acc.y, acc.z, acc.x ^ acc.y};
        } else if (STORE == 8 && (x0 % 16) == 0) { // the same stores into a region small enough to stay in the L2 caches (1 MiB)
            unsigned* q = ck + ((((size_t)y * 72 + x0 / 16) * 16 + lane16) * 4) % (1u << 18);
            *reinterpret_cast<v4u*>(q) = v4u{acc.x, acc.y, acc.z, acc.x ^ acc.y};
        } else if (STORE == 9 && (x0 % 16) == 0) { // 4 bytes per lane (the resolve pass's packed winners)
            unsigned* q = ck + ((size_t)y * 72 + x0 / 16) * 16 + lane16;
            *q = acc.x;
        } else if (STORE && STORE <= 2 && (x0 % 16) == 0) { // the checkpoint pass's stores: 16 bytes per lane every 16 columns
            unsigned* q = ck + (((size_t)y * 72 + x0 / 16) * 16 + lane16) * 4;
            if (STORE == 2) { __builtin_nontemporal_store(acc.x, q); __builtin_nontemporal_store(acc.y, q + 1); __builtin_nontemporal_store(acc.z, q + 2); __builtin_nontemporal_store(acc.x ^ acc.y, q + 3); }
            else { q[0] = acc.x; q[1] = acc.y; q[2] = acc.z; q[3] = acc.x ^ acc.y; }
        }
    }
    if (acc.x == 0x12345 && acc.y == 77) out[threadIdx.x] = acc.z;
}

// v0 + the checkpoint stores with nothing conditional in the loop: requests clamped to the last column, a store every 16 columns
__global__ __launch_bounds__(256) void k_v14(const char* __restrict__ vol, int rows, unsigned* out, unsigned* ck)
{
    constexpr int U = 8;
    const int lane16 = threadIdx.x & 15, y = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (y >= rows) return;
    const char* p = vol + (size_t)y * ROW_BYTES + lane16 * 12;
    v3u a[U], b[U], acc = {0, 0, 0};
    for (int u = 0; u < U; u++) a[u] = *reinterpret_cast<const v3u*>(p + (size_t)u * 192);
    for (int x0 = 0; x0 < COLS; x0 += 2 * U) {
        for (int u = 0; u < U; u++) b[u] = *reinterpret_cast<const v3u*>(p + (size_t)min(x0 + U + u, COLS - 1) * 192);
        unsigned* q = ck + (((size_t)y * 72 + x0 / 16) * 16 + lane16) * 4;
        *reinterpret_cast<v4u*>(q) = v4u{acc.x, acc.y, acc.z, acc.x ^ acc.y};
        for (int u = 0; u < U; u++) acc ^= a[u];
        for (int u = 0; u < U; u++) a[u] = *reinterpret_cast<const v3u*>(p + (size_t)min(x0 + 2 * U + u, COLS - 1) * 192);
        for (int u = 0; u < U; u++) acc ^= b[u];
    }
    if (acc.x == 0x12345 && acc.y == 77) out[threadIdx.x] = acc.z;
}

// v14 with every segment's checkpoint slot inside the volume row, right behind the segment's 16 columns (the stores then go to the
// memory pages the loads have just opened): a row is 72 x (16 x 192 + 256) bytes
template <bool INROW>
__global__ __launch_bounds__(256) void k_v15(char* __restrict__ vol, int rows, unsigned* out, unsigned* ck)
{
    constexpr int U = 8, SEGB = 16 * 192 + 256, ROWB = 72 * SEGB;
    const int lane16 = threadIdx.x & 15, y = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (y >= rows) return;
    char* p = vol + (size_t)y * ROWB + lane16 * 12;
    auto at = [&](int x) { x = min(x, COLS - 1); return p + (size_t)(x >> 4) * SEGB + (x & 15) * 192; };
    v3u a[U], b[U], acc = {0, 0, 0};
    for (int u = 0; u < U; u++) a[u] = *reinterpret_cast<const v3u*>(at(u));
    for (int x0 = 0; x0 < COLS; x0 += 2 * U) {
        for (int u = 0; u < U; u++) b[u] = *reinterpret_cast<const v3u*>(at(x0 + U + u));
        unsigned* q = INROW ? reinterpret_cast<unsigned*>(vol + (size_t)y * ROWB + (size_t)(x0 >> 4) * SEGB + 16 * 192 + lane16 * 16)
                            : ck + (((size_t)y * 72 + x0 / 16) * 16 + lane16) * 4;
        *reinterpret_cast<v4u*>(q) = v4u{acc.x, acc.y, acc.z, acc.x ^ acc.y};
        for (int u = 0; u < U; u++) acc ^= a[u];
        for (int u = 0; u < U; u++) a[u] = *reinterpret_cast<const v3u*>(at(x0 + 2 * U + u));
        for (int u = 0; u < U; u++) acc ^= b[u];
    }
    if (acc.x == 0x12345 && acc.y == 77) out[threadIdx.x] = acc.z;
}

// 16 lanes x 16 bytes per row and instruction; a row is ROW_BYTES = 859 x 256 - 64 bytes: the last piece is clamped
template <int AHEAD>
__global__ __launch_bounds__(256) void k_v1(const char* __restrict__ vol, int rows, unsigned* out)
{
    constexpr int G = 6; // pieces of 256 bytes per group (= 8 columns)
    const int lane16 = threadIdx.x & 15, y = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (y >= rows) return;
    const char* p = vol + (size_t)y * ROW_BYTES + lane16 * 16;
    const int pieces = (ROW_BYTES + 255) / 256;
    v4u buf[AHEAD + 1][G], acc = {0, 0, 0, 0};
    auto load = [&](int g, v4u (&b)[G]) {
        for (int i = 0; i < G; i++) {
            const int pc = min(g * G + i, pieces - 1);
            size_t off = (size_t)pc * 256;
            if (off + lane16 * 16 + 16 > (size_t)ROW_BYTES) off = 0;
            b[i] = *reinterpret_cast<const v4u*>(p + off);
        }
    };
    const int groups = (pieces + G - 1) / G;
    for (int a = 0; a < AHEAD; a++) load(a, buf[a]);
    for (int g = 0; g < groups; g++) {
        if (g + AHEAD < groups) load(g + AHEAD, buf[AHEAD]);
        for (int i = 0; i < G; i++) acc ^= buf[0][i];
        for (int a = 0; a < AHEAD; a++)
            for (int i = 0; i < G; i++) buf[a][i] = buf[a + 1][i];
    }
    if (acc.x == 0x12345 && acc.y == 77) out[threadIdx.x] = acc.z;
}

__global__ __launch_bounds__(256) void k_v2(const char* __restrict__ vol, int rows, unsigned* out)
{
    const int lane = threadIdx.x & 63, y = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (y >= rows) return;
    const char* p = vol + (size_t)y * ROW_BYTES + lane * 16;
    const int pieces = ROW_BYTES / 1024;
    v4u acc = {0, 0, 0, 0};
    for (int i = 0; i < pieces; i += 4) {
        v4u t[4];
        for (int k = 0; k < 4; k++) t[k] = *reinterpret_cast<const v4u*>(p + (size_t)min(i + k, pieces - 1) * 1024);
        for (int k = 0; k < 4; k++) acc ^= t[k];
    }
    if (acc.x == 0x12345 && acc.y == 77) out[threadIdx.x] = acc.z;
}

int main()
{
    const int rows = 42 * 376;
    const size_t bytes = (size_t)rows * ROW_BYTES;
    char* vol; unsigned* out; unsigned* ck;
    if (hipMalloc(&vol, bytes + 4096) != hipSuccess || hipMalloc(&out, 4096) != hipSuccess || hipMalloc(&ck, (size_t)rows * 72 * 16 * 16) != hipSuccess) return 1;
    (void)hipMemset(vol, 1, bytes + 4096);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto run = [&](const char* name, auto launch) {
        launch(); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        for (int r = 0; r < 5; r++) launch();
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-64s %7.1f GB/s  %.3f ms\n", name, 5 * (double)bytes / (ms * 1e-3) / 1e9, ms / 5);
    };
    run("v0 4 rows/wave, dwordx3 per column, 8 columns ahead", [&] { hipLaunchKernelGGL(k_v0<8>, dim3((rows + 15) / 16), dim3(256), 0, 0, vol, rows, out); });
    run("v5 v0 + the checkpoint stores (16 B per lane every 16 columns)", [&] { hipLaunchKernelGGL((k_v0<8, 1>), dim3((rows + 15) / 16), dim3(256), 0, 0, vol, rows, out, ck); });
    run("v6 v5 with non-temporal stores", [&] { hipLaunchKernelGGL((k_v0<8, 2>), dim3((rows + 15) / 16), dim3(256), 0, 0, vol, rows, out, ck); });
    run("v7 v0 + one contiguous KiB per wavefront store", [&] { hipLaunchKernelGGL((k_v0<8, 3>), dim3((rows + 15) / 16), dim3(256), 0, 0, vol, rows, out, ck); });
    run("v8 v0 + 12 B per lane stores", [&] { hipLaunchKernelGGL((k_v0<8, 4>), dim3((rows + 15) / 16), dim3(256), 0, 0, vol, rows, out, ck); });
    run("v9 v0 + four segments per store burst (every 64 columns)", [&] { hipLaunchKernelGGL((k_v0<8, 5>), dim3((rows + 15) / 16), dim3(256), 0, 0, vol, rows, out, ck); });
    run("v10 v0 + 16 B per lane as one dwordx4", [&] { hipLaunchKernelGGL((k_v0<8, 6>), dim3((rows + 15) / 16), dim3(256), 0, 0, vol, rows, out, ck); });
    run("v11 v0 + a quarter of the stores (16 B per lane every 64 columns)", [&] { hipLaunchKernelGGL((k_v0<8, 7>), dim3((rows + 15) / 16), dim3(256), 0, 0, vol, rows, out, ck); });
    run("v12 v0 + the stores into 1 MiB that stays in cache", [&] { hipLaunchKernelGGL((k_v0<8, 8>), dim3((rows + 15) / 16), dim3(256), 0, 0, vol, rows, out, ck); });
    run("v13 v0 + 4 B per lane every 16 columns", [&] { hipLaunchKernelGGL((k_v0<8, 9>), dim3((rows + 15) / 16), dim3(256), 0, 0, vol, rows, out, ck); });
    run("v14 v10 with nothing conditional in the loop", [&] { hipLaunchKernelGGL(k_v14, dim3((rows + 15) / 16), dim3(256), 0, 0, vol, rows, out, ck); });
    {
        char* vol2;
        if (hipMalloc(&vol2, (size_t)rows * 72 * (16 * 192 + 256) + 4096) != hipSuccess) return 1;
        (void)hipMemset(vol2, 1, (size_t)rows * 72 * (16 * 192 + 256));
        run("v15 v14 on a volume with a checkpoint slot behind every 16 columns, stores elsewhere", [&] { hipLaunchKernelGGL(k_v15<false>, dim3((rows + 15) / 16), dim3(256), 0, 0, vol2, rows, out, ck); });
        run("v16 the same, stores into the slots inside the row", [&] { hipLaunchKernelGGL(k_v15<true>, dim3((rows + 15) / 16), dim3(256), 0, 0, vol2, rows, out, ck); });
        (void)hipFree(vol2);
    }
    run("v3 4 rows/wave, dwordx3 per column, 16 columns ahead", [&] { hipLaunchKernelGGL(k_v0<16>, dim3((rows + 15) / 16), dim3(256), 0, 0, vol, rows, out); });
    run("v1 4 rows/wave, 256 B per row and instruction, 1 group ahead", [&] { hipLaunchKernelGGL(k_v1<1>, dim3((rows + 15) / 16), dim3(256), 0, 0, vol, rows, out); });
    run("v4 4 rows/wave, 256 B per row and instruction, 2 groups ahead", [&] { hipLaunchKernelGGL(k_v1<2>, dim3((rows + 15) / 16), dim3(256), 0, 0, vol, rows, out); });
    run("v2 1 row/wave, 1 KiB per instruction", [&] { hipLaunchKernelGGL(k_v2, dim3((rows + 3) / 4), dim3(256), 0, 0, vol, rows, out); });
    return 0;
}
