// feed_rate.hip -- how fast can a CU pull operand tiles into LDS on gfx950?  No MFMA, no barriers: every wave streams "chunks" of the
// A-operand pattern of the bf16 tile loops (16 rows x 64 B per wave-instruction, row stride 1 KiB) into its own LDS region with
//   mode 0: LDS-DMA (global_load_lds_dwordx4),  mode 1: global_load_dwordx4 -> VGPR -> ds_write_b128,  mode 2: both, alternating,
// keeping DEPTH pieces in flight per wave.  src_mb: size of the streamed region (small = L2 resident, large = HBM).
// Build: hipcc --offload-arch=gfx950 -O3 feed_rate.hip -o feed_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void glds16_s(uint32_t voff, const void* sbase, uint32_t lds_addr) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}
template <int MODE, int ROWB>
__global__ __launch_bounds__(256) void k(const char* __restrict__ src, size_t src_bytes, int pieces, uint32_t* out) {
    __shared__ __attribute__((aligned(16))) char sm[48 * 1024];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)sm + wave * 12288;
    // piece p of this wave: 16 rows x 64 B starting at row (gw * pieces + p) * 16; lane -> (row = lane >> 2, 16-B chunk = lane & 3)
    const size_t gw = (size_t)blockIdx.x * 4 + wave;
    constexpr int CPR = ROWB / 16;   // 16-B chunks per row per piece: 4 = half a 128-B line per row, 8 = whole lines
    const uint32_t voff = (uint32_t)((lane / CPR) * 1024 + (lane % CPR) * 16);
    u32x4 acc = {0, 0, 0, 0};
    u32x4 r[4];
    for (int p = 0; p < pieces; p += 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            size_t off = ((gw * pieces + p + j) * (64 / CPR) * 1024) & (src_bytes - 1);   // src_bytes is a power of two
            const char* base = src + off;
            const uint32_t dst = lds0 + ((p + j) % 12) * 1024;
            const bool dma = MODE == 0 || (MODE == 2 && (j & 1) == 0);
            if (dma) {
                glds16_s(voff, base, dst);
            } else {
                asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r[j]) : "v"(voff), "s"(base) : "memory");
            }
        }
        if (MODE != 0) {
            // registers of this group: wait for them (the DMA pieces of the group may stay in flight in mode 2)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool dma = MODE == 0 || (MODE == 2 && (j & 1) == 0);
                if (!dma) {
                    const uint32_t dst = lds0 + ((p + j) % 12) * 1024 + lane * 16;
                    asm volatile("ds_write_b128 %0, %1" ::"v"(dst), "v"(r[j]) : "memory");
                }
            }
        } else {
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    acc = *reinterpret_cast<u32x4*>(sm + tid * 16);
    if ((acc.x ^ acc.y) == 0x12345u) out[tid] = acc.x;
}
int main(int argc, char** argv) {
    const size_t big = (size_t)1 << 30;
    char* src;
    uint32_t* out;
    hipMalloc(&src, big);
    hipMemset(src, 1, big);
    hipMalloc(&out, 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int pieces = 2048;   // per wave: 2048 KiB
    for (int rowb : {64, 128})
    for (int srcmb : {8, 1024})
        for (int mode = 0; mode < 2; ++mode)
            for (int wg = 2; wg <= 3; ++wg) {
                const int grid = 256 * wg;
                float ms = 0;
                for (int rep = 0; rep < 2; ++rep) {
                    hipEventRecord(e0);
                    if (mode == 0 && rowb == 64) hipLaunchKernelGGL((k<0, 64>), dim3(grid), dim3(256), 0, 0, src, (size_t)srcmb << 20, pieces, out);
                    if (mode == 1 && rowb == 64) hipLaunchKernelGGL((k<1, 64>), dim3(grid), dim3(256), 0, 0, src, (size_t)srcmb << 20, pieces, out);
                    if (mode == 0 && rowb == 128) hipLaunchKernelGGL((k<0, 128>), dim3(grid), dim3(256), 0, 0, src, (size_t)srcmb << 20, pieces, out);
                    if (mode == 1 && rowb == 128) hipLaunchKernelGGL((k<1, 128>), dim3(grid), dim3(256), 0, 0, src, (size_t)srcmb << 20, pieces, out);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    hipEventElapsedTime(&ms, e0, e1);
                }
                const double bytes = (double)grid * 4 * pieces * 1024.0;
                printf("row piece %3d B  src %4d MiB  mode %d (%s)  %d WG/CU: %.3f ms  %.2f TB/s into LDS = %.1f B/clk/CU\n", rowb, srcmb, mode,
                       mode == 0 ? "LDS-DMA      " : mode == 1 ? "load+ds_write" : "both         ", wg, ms, bytes / ms / 1e9,
                       bytes / (ms * 1e-3) / 256 / 2.4e9);
            }
    return 0;
}
