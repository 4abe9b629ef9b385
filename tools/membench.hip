// membench.hip -- what can HBM deliver for the chain's access pattern (NL read streams + 1 write
// stream of 66 MB frames), as a function of launch shape?  Diagnostic only; not part of the library.
//   hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o /tmp/membench && /tmp/membench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct Job { u32x4 *out; const u32x4 *in[4]; size_t n; };   // n = 16-byte words per buffer
struct Jobs { Job j[8]; };

template <int NL, int U, int NT>
__global__ void k_stream(Jobs jobs, int njobs) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (int j = 0; j < njobs; j++) {
        const Job &job = jobs.j[j];
        size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
        for (; i + (U - 1) * stride < job.n; i += U * stride) {
            u32x4 v[U][NL];
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
                for (int k = 0; k < NL; k++)
                    v[u][k] = NT ? __builtin_nontemporal_load(&job.in[k][i + u * stride]) : job.in[k][i + u * stride];
#pragma unroll
            for (int u = 0; u < U; u++) {
                u32x4 x = v[u][0];
#pragma unroll
                for (int k = 1; k < NL; k++) x ^= v[u][k];
                if (NT) __builtin_nontemporal_store(x, &job.out[i + u * stride]);
                else job.out[i + u * stride] = x;
            }
        }
        for (; i < job.n; i += stride) {
            u32x4 x = job.in[0][i];
#pragma unroll
            for (int k = 1; k < NL; k++) x ^= job.in[k][i];
            job.out[i] = x;
        }
    }
}

// contiguous-chunk variant: each workgroup streams one contiguous chunk per trip (same as above) but
// blocks are ordered so that the 8 XCDs take interleaved 4 KiB pieces
template <int NL, int U>
static float run(Jobs &jobs, int njobs, int block, int grid, int nt, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> t;
    for (int r = 0; r < reps + 2; r++) {
        CK(hipEventRecord(e0));
        if (nt) hipLaunchKernelGGL((k_stream<NL, U, 1>), dim3(grid), dim3(block), 0, 0, jobs, njobs);
        else    hipLaunchKernelGGL((k_stream<NL, U, 0>), dim3(grid), dim3(block), 0, 0, jobs, njobs);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 2) t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main() {
    const size_t W = 3840, H = 2160, bytes = W * H * 8, n = bytes / 16;
    const int njobs = 8;
    Jobs jobs;
    for (int j = 0; j < njobs; j++) {
        CK(hipMalloc((void **)&jobs.j[j].out, bytes));
        for (int k = 0; k < 4; k++) { CK(hipMalloc((void **)&jobs.j[j].in[k], bytes)); CK(hipMemset((void *)jobs.j[j].in[k], 0x11 * (k + 1), bytes)); }
        jobs.j[j].n = n;
    }
    printf("%-6s %-5s %-6s %-3s %-3s %10s %10s\n", "NL", "block", "grid", "U", "nt", "ms", "GB/s");
    int blocks[] = { 256, 512, 1024 };
    int wgs_per_cu[] = { 1, 2, 4, 8 };
    for (int nl : { 1, 2, 3 })
        for (int b : blocks)
            for (int w : wgs_per_cu) {
                if (b * w > 2048 * 1 && b == 1024 && w > 2) continue;
                int grid = 256 * w;
                for (int nt = 0; nt < 2; nt++)
                    for (int u : { 1, 2, 4 }) {
                        float ms = 0;
#define RUN(NLV, UV) ms = run<NLV, UV>(jobs, njobs, b, grid, nt, 7)
                        if (nl == 1) { if (u == 1) RUN(1, 1); else if (u == 2) RUN(1, 2); else RUN(1, 4); }
                        if (nl == 2) { if (u == 1) RUN(2, 1); else if (u == 2) RUN(2, 2); else RUN(2, 4); }
                        if (nl == 3) { if (u == 1) RUN(3, 1); else if (u == 2) RUN(3, 2); else RUN(3, 4); }
                        double gb = (double)bytes * (nl + 1) * njobs / (ms * 1e-3) / 1e9;
                        printf("%-6d %-5d %-6d %-3d %-3d %10.4f %10.0f\n", nl, b, grid, u, nt, ms, gb);
                    }
            }
    return 0;
}
