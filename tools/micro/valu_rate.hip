// Measures the issue rate of the vector instructions the SGBM and LK kernels are made of, on the whole chip:
// 8 independent dependency chains per lane, enough waves to fill every SIMD, result in wave-instructions per second.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate tools/micro/valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHAINS 8
#define ITERS 4096

#define KERNEL(name, asmtext)                                                                     \
    __global__ __launch_bounds__(256) void name(int* out, int seed)                               \
    {                                                                                             \
        int a[CHAINS], b = seed + threadIdx.x, c = seed * 3 + 1;                                  \
        for (int i = 0; i < CHAINS; i++) a[i] = threadIdx.x * (i + 1) + seed;                     \
        for (int it = 0; it < ITERS; it++) {                                                      \
            _Pragma("unroll") for (int i = 0; i < CHAINS; i++) asm volatile(asmtext : "+v"(a[i]) : "v"(b), "v"(c)); \
        }                                                                                         \
        int s = 0;                                                                                \
        for (int i = 0; i < CHAINS; i++) s ^= a[i];                                               \
        if (s == 0x12345678) out[threadIdx.x] = s;                                                \
    }

KERNEL(k_add_u32, "v_add_u32 %0, %0, %1")
KERNEL(k_pk_add_u16, "v_pk_add_u16 %0, %0, %1")
KERNEL(k_pk_sub_u16_clamp, "v_pk_sub_u16 %0, %0, %1 clamp")
KERNEL(k_pk_min_i16, "v_pk_min_i16 %0, %0, %1")
KERNEL(k_pk_max_i16, "v_pk_max_i16 %0, %0, %1")
KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2")
KERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %1, 16")
KERNEL(k_dot2, "v_dot2_i32_i16 %0, %0, %1, %2")
KERNEL(k_dot2c, "v_dot2c_i32_i16 %0, %1, %2")
KERNEL(k_min_dpp, "v_min_i32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1")
KERNEL(k_mov_dpp, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf")
KERNEL(k_pk_ashr, "v_pk_ashrrev_i16 %0, 1, %0")
KERNEL(k_fma_f32, "v_fma_f32 %0, %0, %1, %2")
KERNEL(k_lshl_add_u64, "v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %2")

typedef void (*kern_t)(int*, int);
struct Entry { const char* name; kern_t k; int per; };

int main()
{
    int* out; hipMalloc(&out, 4096);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const int blocks = cus * 8; // 8 blocks x 4 waves per CU = 8 waves per SIMD
    Entry es[] = {{"v_add_u32", k_add_u32, 1}, {"v_pk_add_u16", k_pk_add_u16, 1}, {"v_pk_sub_u16 clamp", k_pk_sub_u16_clamp, 1},
                  {"v_pk_min_i16", k_pk_min_i16, 1}, {"v_pk_max_i16", k_pk_max_i16, 1}, {"v_perm_b32", k_perm, 1},
                  {"v_alignbit_b32", k_alignbit, 1}, {"v_dot2_i32_i16", k_dot2, 1}, {"v_dot2c_i32_i16", k_dot2c, 1},
                  {"v_min_i32_dpp", k_min_dpp, 1}, {"v_mov_b32_dpp", k_mov_dpp, 1}, {"v_pk_ashrrev_i16", k_pk_ashr, 1},
                  {"v_fma_f32", k_fma_f32, 1}, {"2 x v_add_u32", k_lshl_add_u64, 2}};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("CUs %d, clock %d kHz\n", cus, p.clockRate);
    for (auto& e : es) {
        hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 1);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; r++) hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, r);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double insts = 5.0 * blocks * 4 * (double)ITERS * CHAINS * e.per;
        const double rate = insts / (ms * 1e-3);
        printf("%-22s %8.1f G wave-instr/s  = %.2f cycles per instruction and SIMD at the nominal clock\n", e.name, rate / 1e9,
               (double)cus * 4 * p.clockRate * 1e3 / rate);
    }
    return 0;
}
