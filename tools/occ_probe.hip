// occ_probe.hip -- how the codec's two access mixes respond to OCCUPANCY (workgroups per CU, capped through the
// dynamic-LDS request) with no compute in the way.  Same geometry as the codec: a workgroup = 4 waves = 4 consecutive
// 2-chunk segments; a wave reads (encoder mix) or writes (decoder mix) its 16 KiB chunks as 1-KiB rows and writes /
// reads `small` bytes per chunk in its slot.
// Build: hipcc --offload-arch=gfx950 -O3 tools/occ_probe.hip -o tools/occ_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

// decoder mix: read `rin` bytes per chunk (contiguous per segment), write the 16 KiB chunk in 16 row stores
template <int NT>
__global__ __launch_bounds__(256) void k_dec_mix(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, uint32_t chunks, uint32_t rin, size_t nseg, size_t slot)
{
    extern __shared__ uint32_t pad[];
    const int lane = threadIdx.x & 63;
    const size_t seg = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (seg >= nseg) return;
    const uint8_t* p = src + seg * slot;
    uint8_t* o = dst + seg * (size_t)chunks * 16384;
    u32x4 acc = {1,2,3,4};
    for (uint32_t c = 0; c < chunks; ++c) {
        for (uint32_t i = lane * 16; i < rin; i += 1024) acc ^= *(const u32x4*)(p + i);
        p += rin;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            u32x4* q = (u32x4*)(o + (size_t)c*16384 + (k*64+lane)*16);
            if (NT) __builtin_nontemporal_store(acc, q); else *q = acc;
        }
    }
    if (acc.x == 0x12345u) pad[0] = 1;
}

// encoder mix: read the 16 KiB chunks (8 rows in flight, nt loads), write `wout` bytes per chunk (nt stores)
__global__ __launch_bounds__(256) void k_enc_mix(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, uint32_t chunks, uint32_t wout, size_t nseg, size_t slot,
                                                 const uint32_t* __restrict__ tbl = nullptr, int chain = 0, int sync = 0)
{
    extern __shared__ uint32_t pad[];
    const int lane = threadIdx.x & 63;
    size_t seg = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    // fake prologue: `chain` dependent loads (a directory walk: task -> segment -> channel -> table), optionally a
    // workgroup barrier behind them, before the first row is requested
    if (chain > 0) {
        uint32_t j = (uint32_t)(blockIdx.x & 1023u);
        for (int i = 0; i < chain; ++i) j = tbl[j] + (uint32_t)(blockIdx.x & 1023u);
        if (sync) __syncthreads();
        seg += j >> 31;  // (tbl holds zeros: j == blockIdx & 1023 < 2^31)
    } else if (chain < 0) {
        // the same walk through directories that are read ONCE (as the codec's are): every workgroup's loads hit
        // lines nobody has touched -- 32 bytes apart per workgroup, a different 8 MiB array per step
        uint32_t j = blockIdx.x * 8u;
        for (int i = 0; i < -chain; ++i) j = tbl[(size_t)i * (2u << 20) + j] + blockIdx.x * 8u;
        if (sync) __syncthreads();
        seg += j >> 31;
    }
    if (seg >= nseg) return;
    const uint8_t* p = src + seg * (size_t)chunks * 16384;
    uint8_t* o = dst + seg * slot;
    u32x4 acc = {0,0,0,0};
    u32x4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = __builtin_nontemporal_load((const u32x4*)(p + (k*64+lane)*16));
    for (uint32_t c = 0; c < chunks; ++c) {
        const uint8_t* cur = p + (size_t)c*16384;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            acc ^= v[k&7];
            if (k < 8 || c + 1 < chunks) v[k&7] = __builtin_nontemporal_load((const u32x4*)(cur + ((k+8)*64+lane)*16));
        }
        for (uint32_t i = lane * 16; i < wout; i += 1024) __builtin_nontemporal_store(acc, (u32x4*)(o + i));
        o += wout;
    }
    if (acc.x == 0x12345u) pad[0] = 1;
}

// encoder mix with the segment's output written in ONE burst at the end of the segment (wend = 1) instead of per chunk,
// or in 256-byte pieces as they "fill" (wend = 2: one 256-B block per 1.4 rows, the codec's own cadence)
__global__ __launch_bounds__(256) void k_enc_mix_w(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, uint32_t chunks, uint32_t wout, size_t nseg, size_t slot, int wend)
{
    extern __shared__ uint32_t pad[];
    const int lane = threadIdx.x & 63;
    const size_t seg = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (seg >= nseg) return;
    const uint8_t* p = src + seg * (size_t)chunks * 16384;
    uint8_t* o = dst + seg * slot;
    u32x4 acc = {0,0,0,0};
    u32x4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = __builtin_nontemporal_load((const u32x4*)(p + (k*64+lane)*16));
    uint32_t done = 0;
    for (uint32_t c = 0; c < chunks; ++c) {
        const uint8_t* cur = p + (size_t)c*16384;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            acc ^= v[k&7];
            if (k < 8 || c + 1 < chunks) v[k&7] = __builtin_nontemporal_load((const u32x4*)(cur + ((k+8)*64+lane)*16));
            if (wend == 2) {  // 256-byte blocks as rows complete: 16 lanes x 16 bytes
                const uint32_t upto = (uint32_t)(((uint64_t)(c * 16 + k + 1) * wout / 16) & ~255u);
                for (; done < upto; done += 256)
                    if (lane < 16) __builtin_nontemporal_store(acc, (u32x4*)(o + done + lane * 16));
            }
        }
        if (wend == 0) {
            for (uint32_t i = lane * 16; i < wout; i += 1024) __builtin_nontemporal_store(acc, (u32x4*)(o + i));
            o += wout;
        }
    }
    if (wend == 1)
        for (uint32_t i = lane * 16; i < wout * chunks; i += 1024) __builtin_nontemporal_store(acc, (u32x4*)(o + i));
    if (acc.x == 0x12345u) pad[0] = 1;
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
    const size_t bytes = (size_t)10240 * 1000 * 1000 / 32768 * 32768;  // ~10.24 GB
    uint8_t *big, *small_;
    CK(hipMalloc(&big, bytes)); CK(hipMalloc(&small_, bytes / 16384 * 4224 + 65536));
    CK(hipMemset(big, 1, bytes)); CK(hipMemset(small_, 0, bytes / 16384 * 4224 + 65536));
    const uint32_t chunks = 2; const size_t nseg = bytes / ((size_t)chunks * 16384);
    const size_t slot = (size_t)chunks * 4224;
    const int occ[] = {1, 2, 3, 4, 5, 6, 8};
    for (int wg : occ) {
        const size_t lds = (size_t)160 * 1024 / wg - 1024;
        CK(hipFuncSetAttribute((const void*)k_dec_mix<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CK(hipFuncSetAttribute((const void*)k_dec_mix<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        CK(hipFuncSetAttribute((const void*)k_enc_mix, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        float a = timeit([&]{ hipLaunchKernelGGL(k_dec_mix<1>, dim3((nseg+3)/4), dim3(256), lds, 0, small_, big, chunks, 2944u, nseg, slot); });
        float b = timeit([&]{ hipLaunchKernelGGL(k_dec_mix<0>, dim3((nseg+3)/4), dim3(256), lds, 0, small_, big, chunks, 2944u, nseg, slot); });
        float c = timeit([&]{ hipLaunchKernelGGL(k_dec_mix<1>, dim3((nseg+3)/4), dim3(256), lds, 0, small_, big, chunks, 0u, nseg, slot); });
        float d = timeit([&]{ hipLaunchKernelGGL(k_dec_mix<0>, dim3((nseg+3)/4), dim3(256), lds, 0, small_, big, chunks, 0u, nseg, slot); });
        float e = timeit([&]{ hipLaunchKernelGGL(k_enc_mix, dim3((nseg+3)/4), dim3(256), lds, 0, big, small_, chunks, 2944u, nseg, slot); });
        printf("%d workgroups/CU : decoder mix nt %.3f ms  plain %.3f ms | pure write nt %.3f  plain %.3f | encoder mix %.3f ms\n", wg, a, b, c, d, e);
        fflush(stdout);
    }
    {   // the encoder mix behind a prologue of dependent loads, 4 workgroups per CU
        uint32_t* tbl; CK(hipMalloc(&tbl, 4096 * 4)); CK(hipMemset(tbl, 0, 4096 * 4));
        const size_t lds = (size_t)160 * 1024 / 4 - 1024;
        CK(hipFuncSetAttribute((const void*)k_enc_mix, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        for (int chain : {0, 1, 2, 3, 5, 8})
            for (int sync : {0, 1}) {
                float e = timeit([&]{ hipLaunchKernelGGL(k_enc_mix, dim3((nseg+3)/4), dim3(256), lds, 0, big, small_, chunks, 2944u, nseg, slot, (const uint32_t*)tbl, chain, sync); });
                float r = timeit([&]{ hipLaunchKernelGGL(k_enc_mix, dim3((nseg+3)/4), dim3(256), lds, 0, big, small_, chunks, 0u, nseg, slot, (const uint32_t*)tbl, chain, sync); });
                printf("encoder mix, prologue of %d dependent loads%s : %.3f ms   (reads only: %.3f ms)\n", chain, sync ? " + barrier" : "", e, r);
                fflush(stdout);
            }
        {
            uint32_t* dir; CK(hipMalloc(&dir, (size_t)8 * (8u << 20))); CK(hipMemset(dir, 0, (size_t)8 * (8u << 20)));
            for (int chain : {-1, -2, -3, -5, -8})
                for (int sync : {0, 1}) {
                    float e = timeit([&]{ hipLaunchKernelGGL(k_enc_mix, dim3((nseg+3)/4), dim3(256), lds, 0, big, small_, chunks, 2944u, nseg, slot, (const uint32_t*)dir, chain, sync); });
                    float r = timeit([&]{ hipLaunchKernelGGL(k_enc_mix, dim3((nseg+3)/4), dim3(256), lds, 0, big, small_, chunks, 0u, nseg, slot, (const uint32_t*)dir, chain, sync); });
                    printf("encoder mix, prologue of %d dependent loads from read-once directories%s : %.3f ms   (reads only: %.3f ms)\n", -chain, sync ? " + barrier" : "", e, r);
                    fflush(stdout);
                }
        }
        // rows that start 64 bytes into a 128-byte line (the window of the chosen system starts at sample 2^6)
        for (size_t shift : {(size_t)0, (size_t)64, (size_t)16}) {
            float e = timeit([&]{ hipLaunchKernelGGL(k_enc_mix, dim3((nseg+3)/4 - 1), dim3(256), lds, 0, big + shift, small_, chunks, 2944u, nseg - 4, slot, (const uint32_t*)tbl, 0, 0); });
            float d = timeit([&]{ hipLaunchKernelGGL(k_dec_mix<1>, dim3((nseg+3)/4 - 1), dim3(256), (size_t)41984, 0, small_, big + shift, chunks, 2944u, nseg - 4, slot); });
            printf("rows shifted by %zu bytes : encoder mix %.3f ms, decoder mix (3 workgroups/CU) %.3f ms\n", shift, e, d);
        }
        // how the segment's output leaves: per chunk, one burst per segment, 256-byte blocks as they fill
        for (uint32_t ch : {2u, 4u})
            for (int wend : {0, 1, 2}) {
                const size_t ns = bytes / ((size_t)ch * 16384), sl = (size_t)ch * 4224;
                float e = timeit([&]{ hipLaunchKernelGGL(k_enc_mix_w, dim3((ns+3)/4), dim3(256), lds, 0, big, small_, ch, 2944u, ns, sl, wend); });
                printf("encoder mix, %u chunks per segment, output %s : %.3f ms\n", ch,
                       wend == 0 ? "per chunk" : wend == 1 ? "in one burst per segment" : "in 256-byte blocks as they fill", e);
                fflush(stdout);
            }
    }
    return 0;
}
