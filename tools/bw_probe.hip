// bw_probe.hip -- streaming-rate probes for the access patterns of the codec kernels.
// Build: hipcc --offload-arch=gfx950 -O3 tools/bw_probe.hip -o tools/bw_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

// wave-per-segment read: each wave walks `chunks` chunks of 16 KiB, 16 loads (1 KiB each) in flight
template <int NT>
__global__ __launch_bounds__(256) void k_wave_read(const uint8_t* __restrict__ src, uint32_t chunks, uint32_t* out, size_t nseg)
{
    const int lane = threadIdx.x & 63;
    const size_t seg = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (seg >= nseg) return;
    const uint8_t* p = src + seg * (size_t)chunks * 16384;
    u32x4 acc = {0,0,0,0};
    u32x4 v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = NT ? __builtin_nontemporal_load((const u32x4*)(p + (k*64+lane)*16)) : *(const u32x4*)(p + (k*64+lane)*16);
    for (uint32_t c = 0; c < chunks; ++c) {
        const uint8_t* nx = p + (size_t)(c+1)*16384;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            acc ^= v[k];
            if (c + 1 < chunks) v[k] = NT ? __builtin_nontemporal_load((const u32x4*)(nx + (k*64+lane)*16)) : *(const u32x4*)(nx + (k*64+lane)*16);
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}

// grid-stride read, 16 B per lane, UNR loads in flight
template <int UNR>
__global__ __launch_bounds__(256) void k_stream_read(const u32x4* __restrict__ src, size_t n, uint32_t* out)
{
    u32x4 acc = {0,0,0,0};
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNR-1)*stride < n; i += UNR*stride) {
        u32x4 t[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) t[u] = __builtin_nontemporal_load(src + i + u*stride);
#pragma unroll
        for (int u = 0; u < UNR; ++u) acc ^= t[u];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}

__global__ __launch_bounds__(256) void k_stream_write(u32x4* __restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    u32x4 v = {1,2,3,(uint32_t)threadIdx.x};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) __builtin_nontemporal_store(v, dst + i);
}

// wave-per-segment write of 1 KiB rows (the decoder's store pattern)
__global__ __launch_bounds__(256) void k_wave_write(uint8_t* __restrict__ dst, uint32_t chunks, size_t nseg)
{
    const int lane = threadIdx.x & 63;
    const size_t seg = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (seg >= nseg) return;
    uint8_t* p = dst + seg * (size_t)chunks * 16384;
    u32x4 v = {1,2,3,(uint32_t)lane};
    for (uint32_t c = 0; c < chunks; ++c)
#pragma unroll
        for (int k = 0; k < 16; ++k) *(u32x4*)(p + (size_t)c*16384 + (k*64+lane)*16) = v;
}

__global__ __launch_bounds__(256) void k_copy(const u32x4* __restrict__ src, u32x4* __restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

// encoder-like mix: each wave reads 16 KiB chunks and writes WOUT bytes per chunk (contiguous per
// segment), store flavour ST: 0 plain dwordx4, 1 nontemporal dwordx4, 2 plain but batched x4 chunks,
// 3 nontemporal batched x2 chunks (one burst per 2-chunk segment), 4 nontemporal into a LOG: the
// wave reserves its bytes with one atomicAdd on a global cursor, so the chip writes one moving
// front instead of per-segment slots
__device__ unsigned long long g_cursor;
__global__ void k_reset_cursor() { g_cursor = 0; }
template <int ST>
__global__ __launch_bounds__(256) void k_wave_rw(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, uint32_t chunks, uint32_t wout, size_t nseg, size_t slot)
{
    const int lane = threadIdx.x & 63;
    const size_t seg = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (seg >= nseg) return;
    const uint8_t* p = src + seg * (size_t)chunks * 16384;
    uint8_t* o = dst + seg * slot;
    u32x4 acc = {0,0,0,0};
    u32x4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = __builtin_nontemporal_load((const u32x4*)(p + (k*64+lane)*16));
    uint32_t pending = 0;
    for (uint32_t c = 0; c < chunks; ++c) {
        const uint8_t* cur = p + (size_t)c*16384;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            acc ^= v[k&7];
            if (k < 8 || c + 1 < chunks) v[k&7] = __builtin_nontemporal_load((const u32x4*)(cur + ((k+8)*64+lane)*16));
        }
        pending += wout;
        if (ST == 2 && (c & 3) != 3 && c + 1 < chunks) continue;
        if (ST == 3 && (c & 1) != 1 && c + 1 < chunks) continue;
        if (ST == 4) {
            unsigned long long at = 0;
            if (lane == 0) at = atomicAdd(&g_cursor, (unsigned long long)((pending + 255) & ~255u));
            at = __shfl(at, 0, 64);
            o = dst + at;
        }
        for (uint32_t i = lane * 16; i < pending; i += 1024) {
            if (ST == 1 || ST >= 3) __builtin_nontemporal_store(acc, (u32x4*)(o + i)); else *(u32x4*)(o + i) = acc;
        }
        o += pending; pending = 0;
    }
}

// decoder-like mix: each wave reads RIN bytes per chunk (contiguous per segment) and writes 16 KiB
template <int ST>
__global__ __launch_bounds__(256) void k_wave_wr(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, uint32_t chunks, uint32_t rin, size_t nseg, size_t slot)
{
    const int lane = threadIdx.x & 63;
    const size_t seg = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (seg >= nseg) return;
    const uint8_t* p = src + seg * slot;
    uint8_t* o = dst + seg * (size_t)chunks * 16384;
    u32x4 acc = {1,2,3,4};
    for (uint32_t c = 0; c < chunks; ++c) {
        if (rin) for (uint32_t i = lane * 16; i < rin; i += 1024) acc ^= __builtin_nontemporal_load((const u32x4*)(p + i));
        p += rin;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            u32x4* q = (u32x4*)(o + (size_t)c*16384 + (k*64+lane)*16);
            if (ST == 1) __builtin_nontemporal_store(acc, q); else *q = acc;
        }
    }
}

// time-major strip read: the access pattern of k_deinterleave2.  A workgroup reads tiles of ROWS x W bytes
// (W contiguous bytes of ROWS consecutive rows of a [T][C] matrix, 16 B per lane), tpw tiles along time;
// strips fastest in the grid.  No LDS, no stores: the pattern's own ceiling.
template <int W>
__global__ __launch_bounds__(256) void k_strip_read(const uint8_t* __restrict__ in, size_t T, uint32_t C, uint32_t tpw, uint32_t* out)
{
    constexpr int ROWS = 32768 / W;                 // 32 KiB tiles
    constexpr int LPR = W / 16;                     // lanes per row
    const uint32_t nstrip = C / W;
    const uint32_t c0 = (blockIdx.x % nstrip) * W;
    const size_t ntiles = T / ROWS, gx = gridDim.x / nstrip;
    u32x4 acc = {0,0,0,0};
    for (size_t tile0 = (size_t)(blockIdx.x / nstrip) * tpw; tile0 < ntiles; tile0 += gx * tpw)
        for (size_t tl = tile0; tl < tile0 + tpw && tl < ntiles; ++tl) {
            u32x4 V[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t i = j * 256 + threadIdx.x, row = i / LPR, q = (i % LPR) * 16;
                V[j] = __builtin_nontemporal_load((const u32x4*)(in + (tl * ROWS + row) * C + c0 + q));
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) acc ^= V[j];
        }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}

template <typename F> float timeit(F f, int reps = 5)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    float best = 1e9;
    for (int r = 0; r < reps; ++r) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; }
    return best;
}

int main()
{
    const size_t bytes = (size_t)10240 * 1000 * 1000 / 16384 * 16384;  // ~10.24 GB
    uint8_t *src, *dst; uint32_t* out;
    CK(hipMalloc(&src, bytes)); CK(hipMalloc(&dst, bytes)); CK(hipMalloc(&out, 64));
    CK(hipMemset(src, 1, bytes)); CK(hipMemset(dst, 0, bytes));
    const size_t nvec = bytes / 16;
    {
        const uint32_t C = 1024; const size_t T = bytes / C;
#define STRIP(W) { const uint32_t nstrip = C / W; const size_t ntiles = T / (32768 / W); const uint32_t tpw = 4; \
            const size_t bx = (ntiles + tpw - 1) / tpw; \
            float ms = timeit([&]{ hipLaunchKernelGGL(k_strip_read<W>, dim3((unsigned)(bx * nstrip)), dim3(256), 0, 0, src, T, C, tpw, out); }); \
            printf("strip read W=%4d B x %3d rows per tile : %.3f ms  %.2f TB/s\n", W, 32768 / W, ms, bytes/ms/1e9); }
        STRIP(128) STRIP(256) STRIP(512) STRIP(1024)
    }
    for (uint32_t chunks : {4u, 8u, 16u, 38u}) {
        const size_t nseg = bytes / ((size_t)chunks * 16384);
        float ms = timeit([&]{ hipLaunchKernelGGL(k_wave_read<0>, dim3((nseg+3)/4), dim3(256), 0, 0, src, chunks, out, nseg); });
        printf("wave_read      chunks/seg=%2u : %.3f ms  %.2f TB/s\n", chunks, ms, nseg*(double)chunks*16384/ms/1e9);
        ms = timeit([&]{ hipLaunchKernelGGL(k_wave_read<1>, dim3((nseg+3)/4), dim3(256), 0, 0, src, chunks, out, nseg); });
        printf("wave_read nt   chunks/seg=%2u : %.3f ms  %.2f TB/s\n", chunks, ms, nseg*(double)chunks*16384/ms/1e9);
        ms = timeit([&]{ hipLaunchKernelGGL(k_wave_write, dim3((nseg+3)/4), dim3(256), 0, 0, dst, chunks, nseg); });
        printf("wave_write     chunks/seg=%2u : %.3f ms  %.2f TB/s\n", chunks, ms, nseg*(double)chunks*16384/ms/1e9);
    }
    {
        const uint32_t chunks = 8; const size_t nseg = bytes / ((size_t)chunks * 16384);
        for (uint32_t wout : {768u, 2816u, 3072u, 4096u}) {
            const size_t slot = (size_t)chunks * 4224;
            float ms = timeit([&]{ hipLaunchKernelGGL(k_wave_rw<0>, dim3((nseg+3)/4), dim3(256), 0, 0, src, dst, chunks, wout, nseg, slot); });
            printf("wave_rw plain   wout=%4u : %.3f ms  read %.2f TB/s + write %.2f TB/s\n", wout, ms, bytes/ms/1e9, nseg*(double)chunks*wout/ms/1e9);
            ms = timeit([&]{ hipLaunchKernelGGL(k_wave_rw<1>, dim3((nseg+3)/4), dim3(256), 0, 0, src, dst, chunks, wout, nseg, slot); });
            printf("wave_rw nt      wout=%4u : %.3f ms  read %.2f TB/s + write %.2f TB/s\n", wout, ms, bytes/ms/1e9, nseg*(double)chunks*wout/ms/1e9);
            ms = timeit([&]{ hipLaunchKernelGGL(k_wave_rw<2>, dim3((nseg+3)/4), dim3(256), 0, 0, src, dst, chunks, wout, nseg, slot); });
            printf("wave_rw batch4  wout=%4u : %.3f ms  read %.2f TB/s + write %.2f TB/s\n", wout, ms, bytes/ms/1e9, nseg*(double)chunks*wout/ms/1e9);
            ms = timeit([&]{ hipLaunchKernelGGL(k_wave_rw<3>, dim3((nseg+3)/4), dim3(256), 0, 0, src, dst, chunks, wout, nseg, slot); });
            printf("wave_rw nt x2   wout=%4u : %.3f ms  read %.2f TB/s + write %.2f TB/s\n", wout, ms, bytes/ms/1e9, nseg*(double)chunks*wout/ms/1e9);
            ms = timeit([&]{ hipLaunchKernelGGL(k_reset_cursor, dim3(1), dim3(1), 0, 0); hipLaunchKernelGGL(k_wave_rw<4>, dim3((nseg+3)/4), dim3(256), 0, 0, src, dst, chunks, wout, nseg, slot); });
            printf("wave_rw nt log  wout=%4u : %.3f ms  read %.2f TB/s + write %.2f TB/s\n", wout, ms, bytes/ms/1e9, nseg*(double)chunks*wout/ms/1e9);
        }
    }
    {   // occupancy sensitivity of the codec-shaped mix (2-chunk segments, nt stores): dynamic LDS limits
        // the workgroups per CU (160 KiB / lds)
        const uint32_t chunks = 2; const size_t nseg = bytes / ((size_t)chunks * 16384);
        const size_t slot = (size_t)chunks * 4224;
        for (int wg_per_cu : {2, 3, 4, 5, 6, 8}) {
            const size_t lds = (size_t)160 * 1024 / wg_per_cu - 1024;
            CK(hipFuncSetAttribute((const void*)k_wave_rw<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            float ms = timeit([&]{ hipLaunchKernelGGL(k_wave_rw<1>, dim3((nseg+3)/4), dim3(256), lds, 0, src, dst, chunks, 2816u, nseg, slot); });
            printf("seg2 wave_rw nt wout=2816, %d workgroups/CU : %.3f ms\n", wg_per_cu, ms);
        }
    }
    {   // the codec's real geometry: 2-chunk segments
        const uint32_t chunks = 2; const size_t nseg = bytes / ((size_t)chunks * 16384);
        for (uint32_t wout : {2816u, 3072u}) {
            const size_t slot = (size_t)chunks * 4224;
            float ms = timeit([&]{ hipLaunchKernelGGL(k_wave_rw<1>, dim3((nseg+3)/4), dim3(256), 0, 0, src, dst, chunks, wout, nseg, slot); });
            printf("seg2 wave_rw nt     wout=%4u : %.3f ms\n", wout, ms);
            ms = timeit([&]{ hipLaunchKernelGGL(k_wave_rw<3>, dim3((nseg+3)/4), dim3(256), 0, 0, src, dst, chunks, wout, nseg, slot); });
            printf("seg2 wave_rw nt x2  wout=%4u : %.3f ms\n", wout, ms);
            ms = timeit([&]{ hipLaunchKernelGGL(k_reset_cursor, dim3(1), dim3(1), 0, 0); hipLaunchKernelGGL(k_wave_rw<4>, dim3((nseg+3)/4), dim3(256), 0, 0, src, dst, chunks, wout, nseg, slot); });
            printf("seg2 wave_rw nt log wout=%4u : %.3f ms\n", wout, ms);
        }
    }
    {
        const uint32_t chunks = 8; const size_t nseg = bytes / ((size_t)chunks * 16384);
        for (uint32_t rin : {0u, 2816u}) {
            const size_t slot = (size_t)chunks * 4224;
            float ms = timeit([&]{ hipLaunchKernelGGL(k_wave_wr<0>, dim3((nseg+3)/4), dim3(256), 0, 0, src, dst, chunks, rin, nseg, slot); });
            printf("wave_wr plain   rin=%4u : %.3f ms  write %.2f TB/s\n", rin, ms, bytes/ms/1e9);
            ms = timeit([&]{ hipLaunchKernelGGL(k_wave_wr<1>, dim3((nseg+3)/4), dim3(256), 0, 0, src, dst, chunks, rin, nseg, slot); });
            printf("wave_wr nt      rin=%4u : %.3f ms  write %.2f TB/s\n", rin, ms, bytes/ms/1e9);
        }
    }
    for (int g : {2048, 8192}) {
        float ms = timeit([&]{ hipLaunchKernelGGL(k_stream_read<4>, dim3(g), dim3(256), 0, 0, (const u32x4*)src, nvec, out); });
        printf("stream_read x4 grid=%5d : %.3f ms  %.2f TB/s\n", g, ms, bytes/ms/1e9);
        ms = timeit([&]{ hipLaunchKernelGGL(k_stream_read<8>, dim3(g), dim3(256), 0, 0, (const u32x4*)src, nvec, out); });
        printf("stream_read x8 grid=%5d : %.3f ms  %.2f TB/s\n", g, ms, bytes/ms/1e9);
        ms = timeit([&]{ hipLaunchKernelGGL(k_stream_write, dim3(g), dim3(256), 0, 0, (u32x4*)dst, nvec); });
        printf("stream_write   grid=%5d : %.3f ms  %.2f TB/s\n", g, ms, bytes/ms/1e9);
        ms = timeit([&]{ hipLaunchKernelGGL(k_copy, dim3(g), dim3(256), 0, 0, (const u32x4*)src, (u32x4*)dst, nvec); });
        printf("copy           grid=%5d : %.3f ms  %.2f TB/s (read+write)\n", g, ms, 2.0*bytes/ms/1e9);
    }
    return 0;
}
