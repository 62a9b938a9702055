// write_probe.hip -- what store pattern streams fastest to HBM on this part?  (tuning tool)
//   hipcc --offload-arch=gfx950 -O3 tools/write_probe.hip -o tools/write_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// grid-stride, one 16-byte vector per thread and step; NT: 0 plain, 1 nontemporal
template <int NT>
__global__ __launch_bounds__(256) void k_gs(u32x4* __restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    u32x4 v = {1, 2, 3, (uint32_t)threadIdx.x};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        if (NT) __builtin_nontemporal_store(v, dst + i); else dst[i] = v;
    }
}

// each workgroup owns one contiguous span of SPAN KiB (no loop over the grid): short-lived workgroups
template <int NT, int SPAN_KB>
__global__ __launch_bounds__(256) void k_span(u32x4* __restrict__ dst, size_t n)
{
    constexpr int kSteps = SPAN_KB * 1024 / 16 / 256;
    u32x4 v = {1, 2, 3, (uint32_t)threadIdx.x};
    const size_t base = (size_t)blockIdx.x * (SPAN_KB * 1024 / 16);
#pragma unroll
    for (int s = 0; s < kSteps; ++s) {
        const size_t i = base + (size_t)s * 256 + threadIdx.x;
        if (i < n) { if (NT) __builtin_nontemporal_store(v, dst + i); else dst[i] = v; }
    }
}

// each wave owns a contiguous 16 KiB (the decoder's pattern: 16 rows of 1 KiB), 4 waves per workgroup on
// CONSECUTIVE 16 KiB blocks or (FAR) on blocks 32 KiB * ... apart like segments of one channel
template <int NT>
__global__ __launch_bounds__(256) void k_wave16(u32x4* __restrict__ dst, size_t nblk, int chunks)
{
    const int lane = threadIdx.x & 63;
    const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    u32x4 v = {1, 2, 3, (uint32_t)lane};
    for (int c = 0; c < chunks; ++c) {
        const size_t blk = w * chunks + c;
        if (blk >= nblk) return;
        u32x4* p = dst + blk * 1024;
#pragma unroll
        for (int k = 0; k < 16; ++k) { if (NT) __builtin_nontemporal_store(v, p + k * 64 + lane); else p[k * 64 + lane] = v; }
    }
}

// lane-contiguous: every lane writes 64 consecutive bytes (4 vectors), a wave covers 4 KiB per step
template <int NT>
__global__ __launch_bounds__(256) void k_lane64(u32x4* __restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    u32x4 v = {1, 2, 3, (uint32_t)threadIdx.x};
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i + 3 < n; i += stride) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { if (NT) __builtin_nontemporal_store(v, dst + i + j); else dst[i + j] = v; }
    }
}

template <typename F> float timeit(F f, int reps = 5)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < reps; ++r) { hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
    return best;
}

int main()
{
    const size_t bytes = (size_t)10240 * 1000 * 1000 / 65536 * 65536;
    uint8_t* dst; CK(hipMalloc(&dst, bytes));
    const size_t n = bytes / 16;
    auto rep = [&](const char* name, float ms) { printf("%-44s %.3f ms  %.2f TB/s\n", name, ms, bytes / ms / 1e9); };
    rep("hipMemsetAsync D8", timeit([&]{ hipMemsetAsync(dst, 1, bytes, 0); }));
    rep("hipMemsetD32Async", timeit([&]{ hipMemsetD32Async((hipDeviceptr_t)dst, 0x01020304, bytes / 4, 0); }));
    for (int g : {1024, 2048, 4096, 16384}) {
        char nm[64];
        snprintf(nm, sizeof nm, "grid-stride plain grid=%d", g); rep(nm, timeit([&]{ hipLaunchKernelGGL(k_gs<0>, dim3(g), dim3(256), 0, 0, (u32x4*)dst, n); }));
        snprintf(nm, sizeof nm, "grid-stride nt    grid=%d", g); rep(nm, timeit([&]{ hipLaunchKernelGGL(k_gs<1>, dim3(g), dim3(256), 0, 0, (u32x4*)dst, n); }));
    }
    rep("span 16 KiB/WG plain", timeit([&]{ hipLaunchKernelGGL((k_span<0, 16>), dim3((unsigned)(bytes / 16384)), dim3(256), 0, 0, (u32x4*)dst, n); }));
    rep("span 16 KiB/WG nt", timeit([&]{ hipLaunchKernelGGL((k_span<1, 16>), dim3((unsigned)(bytes / 16384)), dim3(256), 0, 0, (u32x4*)dst, n); }));
    rep("span 64 KiB/WG plain", timeit([&]{ hipLaunchKernelGGL((k_span<0, 64>), dim3((unsigned)(bytes / 65536)), dim3(256), 0, 0, (u32x4*)dst, n); }));
    rep("span 64 KiB/WG nt", timeit([&]{ hipLaunchKernelGGL((k_span<1, 64>), dim3((unsigned)(bytes / 65536)), dim3(256), 0, 0, (u32x4*)dst, n); }));
    const size_t nblk = bytes / 16384;
    for (int chunks : {1, 2, 8}) {
        char nm[64];
        snprintf(nm, sizeof nm, "wave x 16 KiB, %d chunks/wave plain", chunks);
        rep(nm, timeit([&]{ hipLaunchKernelGGL(k_wave16<0>, dim3((unsigned)((nblk / chunks + 3) / 4)), dim3(256), 0, 0, (u32x4*)dst, nblk, chunks); }));
        snprintf(nm, sizeof nm, "wave x 16 KiB, %d chunks/wave nt", chunks);
        rep(nm, timeit([&]{ hipLaunchKernelGGL(k_wave16<1>, dim3((unsigned)((nblk / chunks + 3) / 4)), dim3(256), 0, 0, (u32x4*)dst, nblk, chunks); }));
    }
    rep("lane-contiguous 64 B plain grid=4096", timeit([&]{ hipLaunchKernelGGL(k_lane64<0>, dim3(4096), dim3(256), 0, 0, (u32x4*)dst, n); }));
    rep("lane-contiguous 64 B nt    grid=4096", timeit([&]{ hipLaunchKernelGGL(k_lane64<1>, dim3(4096), dim3(256), 0, 0, (u32x4*)dst, n); }));
    return 0;
}
