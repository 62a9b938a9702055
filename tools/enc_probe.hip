// enc_probe.hip -- the real mh_encode / mh_decode and the compute-free access mixes of occ_probe IN ONE PROCESS ON THE
// SAME hipMalloc BUFFERS: separates "what the kernel does" from "what the buffers are" (allocator, physical placement).
// Build: hipcc --offload-arch=gfx950 -O3 -Iinclude tools/enc_probe.hip -o tools/enc_probe \
//        -Lhardware-efficient-mua-compression_amd -lmuahuff -Wl,-rpath,'$ORIGIN/../hardware-efficient-mua-compression_amd'
// Run on the GPU box: tools/enc_probe [h [C T [S]]]   (h = calibration bits; 6: the window starts 64 bytes into a line;
// C channels of T bins, default 1024 x 1e7; S = 3 uses the code lengths [1,2,2], any other S the lengths 1,2,..,S-1,S-1)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "muahuff.h"
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
#define MH(x) do{int r_=(x); if(r_!=MH_OK){printf("%s -> %d: %s\n",#x,r_,mh_last_error()); exit(1);} }while(0)

// counts with P(0) = 0.62, P(1) = 0.28, P(2) = 0.08, P(3) = 0.02: about 1.4 bits/sample under [1,2,2]
__global__ void k_fill(uint8_t* d, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u ^ (uint32_t)(i >> 32) * 40503u;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        const uint32_t u = x & 1023u;
        d[i] = u < 635 ? 0 : u < 922 ? 1 : u < 1004 ? 2 : 3;
    }
}

__global__ __launch_bounds__(256) void k_enc_mix(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, uint32_t chunks, uint32_t wout, size_t nseg, size_t slot)
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

__global__ void k_null() {}

template <typename F> void timeit(const char* what, F f, int reps = 10)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); f(); CK(hipDeviceSynchronize());
    float best = 1e9, sum = 0;
    for (int r = 0; r < reps; ++r) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms; sum += ms; }
    // the same op back to back between ONE pair of events: launch overheads overlap, as they do in a pipeline
    CK(hipEventRecord(a)); for (int r = 0; r < reps; ++r) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float tot; CK(hipEventElapsedTime(&tot, a, b));
    printf("%-44s min %.4f  mean %.4f ms   back to back %.4f ms\n", what, best, sum / reps, tot / reps); fflush(stdout);
}

int main(int argc, char** argv)
{
    const uint32_t h = argc > 1 ? (uint32_t)atoi(argv[1]) : 6, C = argc > 2 ? (uint32_t)atoi(argv[2]) : 1024, S = argc > 4 ? (uint32_t)atoi(argv[4]) : 3;
    const uint64_t T = argc > 3 ? (uint64_t)atoll(argv[3]) : 10000000;
    const size_t bytes = (size_t)C * T;
    uint8_t *data, *out; uint32_t* payload;
    CK(hipMalloc(&data, bytes + 4096)); CK(hipMalloc(&out, bytes + 4096));
    hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, data, bytes);
    CK(hipMemset(out, 0, bytes));
    std::vector<uint64_t> off(C), len(C, T);
    for (uint32_t c = 0; c < C; ++c) off[c] = (uint64_t)c * T;
    uint8_t sclv[16] = {1, 2, 2};
    if (S != 3) { for (uint32_t i = 0; i < S; ++i) sclv[i] = (uint8_t)(i + 1 < S ? i + 1 : S - 1); }
    mh_plan* plan;
    MH(mh_plan_create(&plan, off.data(), len.data(), C, S, h, MH_MODE_APPROX, MH_WIN_AFTER_CAL, sclv, 1, 0));
    mh_plan_info_t I; MH(mh_plan_info(plan, &I));
    CK(hipMalloc(&payload, I.payload_cap_words * 4));
    uint64_t *seg_words, *ch_bits; uint8_t *peak, *enc, *skipped;
    CK(hipMalloc(&seg_words, I.n_segments * 8)); CK(hipMalloc(&ch_bits, C * 8));
    CK(hipMalloc(&peak, C)); CK(hipMalloc(&enc, C)); CK(hipMalloc(&skipped, C));
    printf("h=%u: %llu segments, payload cap %.2f GB\n", h, (unsigned long long)I.n_segments, I.payload_cap_words * 4 / 1e9);
    const uint32_t chunks = 2; const size_t nseg = bytes / ((size_t)chunks * 16384) - 8;
    const size_t slot = (size_t)chunks * 4224;
    const size_t lds = (size_t)160 * 1024 / 4 - 1024;
    const int reps = bytes < ((size_t)1 << 30) ? 200 : 10;
    for (int round = 0; round < 2; ++round) {
        timeit("empty kernel, 1 workgroup", [&]{ hipLaunchKernelGGL(k_null, dim3(1), dim3(64), 0, 0); }, reps);
        timeit("empty kernel, as many workgroups as the mix", [&]{ hipLaunchKernelGGL(k_null, dim3((nseg+3)/4), dim3(256), 0, 0); }, reps);
        timeit("compute-free encoder mix (occ_probe's)", [&]{ hipLaunchKernelGGL(k_enc_mix, dim3((nseg+3)/4), dim3(256), lds, 0, data, (uint8_t*)payload, chunks, 2944u, nseg, slot); }, reps);
        timeit("  the same, rows shifted by 64 bytes", [&]{ hipLaunchKernelGGL(k_enc_mix, dim3((nseg+3)/4), dim3(256), lds, 0, data + 64, (uint8_t*)payload, chunks, 2944u, nseg, slot); }, reps);
        timeit("mh_encode (calibrate + k_encode2)", [&]{ MH(mh_encode(plan, data, payload, I.payload_cap_words, seg_words, ch_bits, peak, enc, skipped, nullptr)); }, reps);
        timeit("mh_decode", [&]{ MH(mh_decode(plan, payload, I.payload_cap_words, nullptr, peak, enc, out, nullptr)); }, reps);
    }
    std::vector<uint64_t> bits(C);
    CK(hipMemcpy(bits.data(), ch_bits, C * 8, hipMemcpyDeviceToHost));
    double tot = 0; for (auto b : bits) tot += (double)b;
    printf("payload bits/sample %.4f\n", tot / (double)I.window_samples);
    return 0;
}
