/* Issue-rate probe for the byte-SAD instructions of gfx950: how many lane-operations per
 * second v_sad_u8, v_qsad_pk_u16_u8, v_mqsad_u32_u8 and v_alignbyte sustain when every
 * SIMD of the chip is busy. Build: hipcc --offload-arch=gfx950 -O3 -o sad_rate sad_rate.hip */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

/* shader clock (s_memtime) against the constant 100 MHz counter (s_memrealtime): lane 0 of
 * block 0 records both at entry and exit, which gives the core clock the kernel ran at */
__device__ unsigned long long g_clk[4];

template<int KIND>
__global__ __launch_bounds__(256) void probe(uint32_t *out, int iters, uint32_t seed)
{
    __shared__ uint32_t lds[2048];
    if (KIND == 4)
    {
        for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = seed * i;
        __syncthreads();
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { g_clk[0] = clock64(); g_clk[1] = wall_clock64(); }
    uint32_t a = seed + threadIdx.x, b = seed * 3 + blockIdx.x;
    uint64_t w = ((uint64_t)a << 32) | b;
    uint32_t acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
    uint64_t q0 = 0, q1 = 0, q2 = 0, q3 = 0;
    for (int i = 0; i < iters; i++)
    {
#pragma unroll
        for (int u = 0; u < 16; u++)
        {
            if (KIND == 0)
            {
                acc0 = __builtin_amdgcn_sad_u8(a, b, acc0); acc1 = __builtin_amdgcn_sad_u8(b, a + u, acc1);
                acc2 = __builtin_amdgcn_sad_u8(a, b + u, acc2); acc3 = __builtin_amdgcn_sad_u8(b, a, acc3);
            }
            else if (KIND == 1)
            {
                q0 = __builtin_amdgcn_qsad_pk_u16_u8(w, a, q0); q1 = __builtin_amdgcn_qsad_pk_u16_u8(w, b, q1);
                q2 = __builtin_amdgcn_qsad_pk_u16_u8(w, a + u, q2); q3 = __builtin_amdgcn_qsad_pk_u16_u8(w, b + u, q3);
            }
            else if (KIND == 2)
            {
                q0 = __builtin_amdgcn_mqsad_pk_u16_u8(w, a, q0); q1 = __builtin_amdgcn_mqsad_pk_u16_u8(w, b, q1);
                q2 = __builtin_amdgcn_mqsad_pk_u16_u8(w, a + u, q2); q3 = __builtin_amdgcn_mqsad_pk_u16_u8(w, b + u, q3);
            }
            else if (KIND == 4)
            {
                /* the exhaustive-search mix: one 8-byte LDS read and one broadcast read per 8 quad-SADs */
                const uint32_t *p = lds + ((threadIdx.x + u * 67 + i) & 1023);
                const uint64_t r = ((uint64_t)p[1] << 32) | p[0];
                const uint32_t f = lds[(u * 16 + i) & 2047];
                q0 = __builtin_amdgcn_qsad_pk_u16_u8(r, f, q0); q1 = __builtin_amdgcn_qsad_pk_u16_u8(w, f, q1);
                q2 = __builtin_amdgcn_qsad_pk_u16_u8(r, f + 1, q2); q3 = __builtin_amdgcn_qsad_pk_u16_u8(w, f + 1, q3);
                q0 = __builtin_amdgcn_qsad_pk_u16_u8(r, f + 2, q0); q1 = __builtin_amdgcn_qsad_pk_u16_u8(w, f + 2, q1);
                q2 = __builtin_amdgcn_qsad_pk_u16_u8(r, f + 3, q2); q3 = __builtin_amdgcn_qsad_pk_u16_u8(w, f + 3, q3);
            }
            else
            {
                acc0 = __builtin_amdgcn_alignbyte(acc0, a, 1); acc1 = __builtin_amdgcn_alignbyte(acc1, b, 2);
                acc2 = __builtin_amdgcn_alignbyte(acc2, a, 3); acc3 = __builtin_amdgcn_alignbyte(acc3, b, 1);
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { g_clk[2] = clock64(); g_clk[3] = wall_clock64(); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc0 + acc1 + acc2 + acc3 + (uint32_t)(q0 + q1 + q2 + q3) + (uint32_t)((q0 ^ q1 ^ q2 ^ q3) >> 32);
}

template<int KIND> static int run(const char *name, int px_per_op, int blocks_per_cu = 8, int iters = 4096, int ops_per_unroll = 4)
{
    const int blocks = 256 * blocks_per_cu, threads = 256;
    uint32_t *d;
    CHECK(hipMalloc(&d, (size_t)blocks * threads * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    probe<KIND><<<blocks, threads>>>(d, 16, 1);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    probe<KIND><<<blocks, threads>>>(d, iters, 7);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double ops = (double)blocks * threads * iters * 16.0 * ops_per_unroll;
    unsigned long long clk[4];
    CHECK(hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_clk), sizeof(clk)));
    const double mhz = (double)(clk[2] - clk[0]) / ((double)(clk[3] - clk[1]) / 100.0);
    printf("%-26s %2d wg/cu %9.3f ms  %7.2f T lane-ops/s  %7.2f T pixel-SAD/s  shader clock %5.0f MHz\n", name, blocks_per_cu, ms,
           ops / ms / 1e9, ops * px_per_op / ms / 1e9, mhz);
    CHECK(hipFree(d));
    return 0;
}

int main()
{
    /* a qsad of a 0xAABBCCDD window pattern, printed so the operand order is on record */
    if (run<0>("v_sad_u8", 4)) return 1;
    if (run<1>("v_qsad_pk_u16_u8", 16)) return 1;
    if (run<2>("v_mqsad_pk_u16_u8", 16)) return 1;
    if (run<3>("v_alignbyte_b32", 0)) return 1;
    /* sustained: ~10x longer, and fewer waves per SIMD */
    if (run<1>("v_qsad_pk_u16_u8 long", 16, 8, 40960)) return 1;
    if (run<1>("v_qsad_pk_u16_u8", 16, 2)) return 1;
    if (run<1>("v_qsad_pk_u16_u8", 16, 1)) return 1;
    if (run<4>("qsad + LDS mix", 16, 8, 4096, 8)) return 1;
    if (run<4>("qsad + LDS mix long", 16, 8, 40960, 8)) return 1;
    if (run<4>("qsad + LDS mix", 16, 2, 4096, 8)) return 1;
    if (run<4>("qsad + LDS mix", 16, 1, 4096, 8)) return 1;
    return 0;
}
