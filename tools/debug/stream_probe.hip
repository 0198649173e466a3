// Diagnostic: what HBM gives to the access pattern of the wide kernel's column sums --
// S concurrent streams (a wave has 32 or 16 of them), every stream read front to back in bursts of
// B bytes (128: one 16-entry piece of fp64 P a round, as the kernel does; 256, 512: longer bursts from
// fewer streams a wave), three rounds in flight per wave.  Not the kernel: plain loads, the data summed.
//   hipcc --offload-arch=gfx950 -O3 -o tools/debug/stream_probe tools/debug/stream_probe.hip && tools/debug/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#ifndef BXP_DEPTH
#define BXP_DEPTH 3 // rounds in flight per wave (the wide kernel: 3)
#endif

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// every wave: `per_wave` streams of `len` bytes each, `burst` bytes of a stream per round
template<int BURST>
__global__ __launch_bounds__(512) void probe(const char* base, size_t stream_bytes, int streams_per_wave, double* out)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    constexpr int LPS = BURST / 16;          // lanes per stream and load
    constexpr int SPL = 64 / LPS;            // streams per load instruction
    const int loads = streams_per_wave / SPL; // load instructions per round (4 here: 4 KB a wave and round)
    double acc = 0;
    const int rounds = (int)(stream_bytes / BURST);
    // DEPTH rounds in flight: the loads of round r + DEPTH are issued before round r is added up
    constexpr int DEPTH = BXP_DEPTH, MAXL = 4;
    double2 v[DEPTH][MAXL];
    auto issue = [&](int r, int slot) {
#pragma unroll
        for (int l = 0; l < MAXL; ++l)
            if (l < loads) {
                const size_t s = wave * streams_per_wave + (size_t)l * SPL + lane / LPS;
                v[slot][l] = *(const double2*)(base + s * stream_bytes + (size_t)r * BURST + (lane % LPS) * 16);
            }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        if (d < rounds)
            issue(d, d);
    for (int r0 = 0; r0 < rounds; r0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int r = r0 + d;
            if (r < rounds) {
#pragma unroll
                for (int l = 0; l < MAXL; ++l)
                    if (l < loads)
                        acc += v[d][l].x + v[d][l].y;
                if (r + DEPTH < rounds)
                    issue(r + DEPTH, d);
            }
        }
    }
    if (acc == 12345.678)
        out[0] = acc;
}

int main()
{
    const int blocks = 256, waves = blocks * 8;
    const size_t stream_bytes = 16384; // a column of 2048 fp64 entries
    std::vector<int> spw = { 32, 16, 8 };
    const size_t total = (size_t)waves * 32 * stream_bytes; // 1 GiB
    char* buf;
    double* out;
    CHECK(hipMalloc(&buf, total * 4));
    CHECK(hipMalloc(&out, 8));
    CHECK(hipMemset(buf, 1, total * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep)
        for (int v = 0; v < 3; ++v) {
            const int s = spw[v];
            const size_t sb = stream_bytes * (32 / s); // same bytes per wave: fewer, longer streams
            CHECK(hipEventRecord(e0));
            for (int k = 0; k < 4; ++k) {
                const char* b = buf + (size_t)k * total;
                if (v == 0)
                    hipLaunchKernelGGL(probe<128>, dim3(blocks), dim3(512), 0, 0, b, sb, s, out);
                else if (v == 1)
                    hipLaunchKernelGGL(probe<256>, dim3(blocks), dim3(512), 0, 0, b, sb, s, out);
                else
                    hipLaunchKernelGGL(probe<512>, dim3(blocks), dim3(512), 0, 0, b, sb, s, out);
            }
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("bursts of %3d bytes from %5d streams (%2d a wave): %.2f ms for %.1f GB = %.2f TB/s\n",
                   128 << v, waves * s, s, ms, 4.0 * total / 1e9, 4.0 * total / 1e9 / ms);
        }
    return 0;
}
