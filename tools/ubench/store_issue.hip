// Microbenchmark: cost of buffer_store_dwordx4 wave-instructions by address pattern (one block of NW waves per CU).
//   pattern 0: lane -> 16 B at stride S (row-per-lane, what an MFMA D layout gives)      pattern 1: fully coalesced 1 KiB per instruction
//   pattern 2: 8 lanes x 16 B = 128 B contiguous per "pixel", pixels at stride S
// build: hipcc --offload-arch=gfx950 -O3 -o store_issue store_issue.hip ; run: ./store_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__global__ void k(char* out, size_t bytes_per_block, int pattern, int stride, int nst, long long* cyc, int drop, int both) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)blockIdx.x * bytes_per_block, 0, (int)bytes_per_block, 0x00020000);
    u32x4_t v = {1u, 2u, 3u, (unsigned)lane};
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    for (int i = 0; i < nst; ++i) {
        const int inst = i * nw + wave;           // instruction slot of this wave
        int off;
        if (pattern == 0) off = (inst % 64) * 16 + lane * stride + (inst / 64) * 64 * stride;        // 64 lanes = 64 rows, 16 B each, column inst
        else if (pattern == 1) off = inst * 1024 + lane * 16;
        else if (pattern == 2) off = ((inst * 8 + (lane >> 3)) * stride) + (lane & 7) * 16;
        // pixel-shuffle store of a 48-channel sub-pixel: 16 pixels per instruction at 192 B, the four q lane groups share a pixel's 96 B
        else if (pattern == 3) off = (inst * 16 + (lane & 15)) * 192 + (lane >> 4) * 24;     // today: lane (q, n) holds 24 B at q * 24 (16 B of it here)
        else if (pattern == 4) off = (inst * 16 + (lane & 15)) * 192 + (lane >> 4) * 16;     // repacked couts: the q groups write 64 B contiguous
        else off = (inst * 16 + (lane >> 2)) * 192 + (lane & 3) * 16;                        // 5: repacked + lanes transposed to (n, q) order
        if (drop) off = (int)0x80000000;
        __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 0);
        if (pattern >= 3 && !drop) {              // the remaining 8 B of the lane's 24: today right behind its 16; repacked: the pixel's last 32 B, 8 per q group
            typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
            const int px = pattern == 5 ? (lane >> 2) : (lane & 15), qq = pattern == 5 ? (lane & 3) : (lane >> 4);
            const int off2 = (inst * 16 + px) * 192 + (pattern == 3 ? qq * 24 + 16 : 64 + qq * 8);
            __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{v.x, v.y}, r, off2, 0, 0);
            if (both) {                           // the neighbouring sub-pixel (the next cout tile of the real kernel) completes every line,
                const int back = (both - 1) * nw * 16 * 192;      // `both - 1` iterations of the whole block later
                const int o1 = i >= both - 1 ? off + 96 - back : (int)0x80000000, o2 = i >= both - 1 ? off2 + 96 - back : (int)0x80000000;
                __builtin_amdgcn_raw_buffer_store_b128(v, r, o1, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{v.x, v.y}, r, o2, 0, 0);
            }
        }
    }
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
    __builtin_amdgcn_s_waitcnt(0);
    const long long t2 = (long long)__builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 8) { cyc[0] = t1 - t0; cyc[1] = t2 - t0; }
}
int main() {
    const size_t per_block = 64ull << 20;   // 64 MiB window per block
    const int blocks = 256;
    char* out; long long* cyc;
    hipMalloc(&out, per_block * blocks); hipMalloc(&cyc, 16);
    for (int nw : {1, 4, 8})
        for (int pattern : {0, 1, 2, 3, 4, 5})
            for (int stride : {128, 384})
                for (int drop : {0, 1}) for (int both : {0, 1, 2, 5, 17, 49}) {
                    if (both && (pattern < 3 || drop)) continue;
                    if ((pattern == 1 || pattern >= 3) && stride != 128) continue;
                    const int nst = 256;
                    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k, dim3(blocks), dim3(nw * 64), 0, 0, out, per_block, pattern, stride, nst, cyc, drop, both);
                    hipDeviceSynchronize();
                    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                    hipEventRecord(e0);
                    hipLaunchKernelGGL(k, dim3(blocks), dim3(nw * 64), 0, 0, out, per_block, pattern, stride, nst, cyc, drop, both);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    long long h[2]; hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
                    printf("waves/CU %d pattern %d stride %3d drop %d both %d: issue %6.1f cyc per store-instr per CU (%5.1f per wave-instr), drain %6.1f;  %.1f us, %.2f TB/s\n",
                           nw, pattern, stride, drop, both, (double)h[0] / (nst * nw), (double)h[0] / nst, (double)h[1] / (nst * nw), ms * 1e3,
                           drop ? 0.0 : (double)blocks * nw * nst * (pattern >= 3 ? (both ? 3072 : 1536) : 1024) / (ms * 1e-3) / 1e12);
                }
    return 0;
}
