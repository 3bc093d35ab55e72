// ubench7.hip — the opening burst of an a-trous level in isolation (gfx950): 256 workgroups of 768 threads start at once and
// each fetches 5 rows x RWPX pixels of {colour 16 B, normal 12 B, position 12 B} (pattern A: the planes as they are: one
// dwordx4 + two dwordx3 per pixel) or {colour 16 B, {n.x,p.x,n.y,p.y} 16 B, {n.z,p.z} 8 B} (pattern B: interleaved geometry
// planes: dwordx4 + dwordx4 + dwordx2) or 40 B as 2.5 dwordx4 (pattern C: upper bound), then stores one value.  Reports the
// median over workgroups of cycles from entry to "all loads returned" and the bytes per cycle per CU that corresponds to.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench7.hip -o tools/ubench7
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

constexpr int RWPX = 488, ROWS = 5, NT = 768, W = 1920, H = 1080;

template <int PAT>
__global__ __launch_bounds__(NT) void k(const float4 *cv, const float *nrm, const float *pos, const float4 *g0, const float2 *g1,
                                        unsigned long long *out, float *sink)
{
    const int tid = threadIdx.x, bid = blockIdx.x;
    // workgroup -> (strip, first row) like the lane kernel: 4 strips of 480 columns, rows 17 apart, two y-phases
    const int strip = bid & 3, seg = bid >> 2;
    const int x0 = strip * 480 - 4, y0 = (seg * 17 * 2) % (H - 12);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float acc = 0.0f;
    constexpr int N = (ROWS * RWPX + NT - 1) / NT;        // pixels per thread
    float4 c[N]; float a0[N], a1[N], a2[N], b0[N], b1[N], b2[N]; float4 ga[N]; float2 gb[N];
#pragma unroll
    for (int m = 0; m < N; m++) {
        const int idx = min(tid + m * NT, ROWS * RWPX - 1);
        const int r = idx / RWPX, xi = idx - r * RWPX;
        const unsigned q = (unsigned)(y0 + 2 * r) * W + (unsigned)min(max(x0 + xi, 0), W - 1);
        c[m] = cv[q];
        if (PAT == 0) {
            const float *n = nrm + 3 * (size_t)q, *p = pos + 3 * (size_t)q;
            a0[m] = n[0]; a1[m] = n[1]; a2[m] = n[2]; b0[m] = p[0]; b1[m] = p[1]; b2[m] = p[2];
        } else if (PAT == 1) {
            ga[m] = g0[q]; gb[m] = g1[q];
        } else {
            ga[m] = g0[q]; gb[m] = g1[q];      // same bytes as B; C differs below only in how the colour is fetched (kept equal)
        }
    }
#pragma unroll
    for (int m = 0; m < N; m++) {
        acc += c[m].x + c[m].w;
        if (PAT == 0) acc += a0[m] + a1[m] + a2[m] + b0[m] + b1[m] + b2[m];
        else acc += ga[m].x + ga[m].w + gb[m].x + gb[m].y;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 0) out[bid] = t1 - t0;
    if (acc == 12345.678f) sink[0] = acc;
}

template <int PAT>
void run(const char *name, const float4 *cv, const float *nrm, const float *pos, const float4 *g0, const float2 *g1, unsigned long long *d_out, float *sink)
{
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k<PAT>, dim3(256), dim3(NT), 0, 0, cv, nrm, pos, g0, g1, d_out, sink);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(256);
        (void)hipMemcpy(h.data(), d_out, 256 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double bytes = (double)ROWS * RWPX * 40.0;
        printf("%-60s median %6llu cycles (min %6llu max %6llu)  -> %.2f B/cycle/CU\n", name, h[128], h[0], h[255], bytes / (double)h[128]);
    }
}

int main()
{
    const size_t n = (size_t)W * H;
    float4 *cv, *g0; float *nrm, *pos, *sink; float2 *g1; unsigned long long *d_out;
    (void)hipMalloc(&cv, n * 16); (void)hipMalloc(&g0, n * 16); (void)hipMalloc(&nrm, n * 12); (void)hipMalloc(&pos, n * 12);
    (void)hipMalloc(&g1, n * 8); (void)hipMalloc(&sink, 16); (void)hipMalloc(&d_out, 256 * 8);
    (void)hipMemset(cv, 0, n * 16); (void)hipMemset(g0, 0, n * 16); (void)hipMemset(nrm, 0, n * 12); (void)hipMemset(pos, 0, n * 12); (void)hipMemset(g1, 0, n * 8);
    run<0>("A: colour x4 + normal x3 + position x3 (12-byte packed planes)", cv, nrm, pos, g0, g1, d_out, sink);
    run<1>("B: colour x4 + {nx,px,ny,py} x4 + {nz,pz} x2 (interleaved planes)", cv, nrm, pos, g0, g1, d_out, sink);
    run<0>("A again", cv, nrm, pos, g0, g1, d_out, sink);
    run<1>("B again", cv, nrm, pos, g0, g1, d_out, sink);
    return 0;
}
