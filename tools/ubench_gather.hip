// Micro-benchmark (dev tool, not product): how the lane -> (point, 16-byte piece) assignment of the tri-plane gather
// changes the rate at which the texture addresser / L1 serve it.  Same points, same planes, same bytes, same FMAs:
//   map A  lane = 16 q + r   (the MFMA B-operand layout the forward kernel gathers in: 4 consecutive lanes = 4 POINTS)
//   map B  lane = 4 p + q    (4 consecutive lanes = the 4 pieces of ONE point's half line: 64 contiguous bytes per quad)
//   map D  lane = 8 p + o    (8 consecutive lanes = ONE texel's whole 128-B line; 8 points per instruction)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/ubench_gather tools/ubench_gather.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float float4_t __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Plane { const float* data; int h, w; };
struct Planes { Plane p[12]; };

__device__ __forceinline__ void axis(float u, int n, int& i0, int& i1, float& t) {
    const float nm1 = (float)(n - 1);
    float x = ((u + 1.0f) * 0.5f) * nm1;
    x = fminf(fmaxf(x, 0.0f), nm1);
    const float f = floorf(x);
    i0 = (int)f; i1 = min(i0 + 1, n - 1); t = x - f;
}

// pts [N,3] normalised coordinates; out [N,128] features (d*64 + lvl*32 + channel)
template <int MAP, int PIPE>
__global__ __launch_bounds__(256, 2) void gather_kernel(const Planes planes, const float* __restrict__ pts, int N,
                                                        float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile = blockIdx.x * 4 + wave;           // 64 points per wave
    if (tile * 64 >= N) return;
    constexpr int PPI = (MAP == 2) ? 8 : 16;           // points per instruction
    constexpr int NCH = (MAP == 2) ? 4 : 8;            // channels per lane and level-plane
    int pl, piece;                                     // point-in-step, piece of the line
    if (MAP == 0) { pl = lane & 15; piece = lane >> 4; }
    else if (MAP == 1) { pl = lane >> 2; piece = lane & 3; }
    else { pl = lane >> 3; piece = lane & 7; }
    for (int b = 0; b < 64 / PPI; ++b) {
        const int pt = tile * 64 + b * PPI + pl;
        const float x = pts[pt * 3], y = pts[pt * 3 + 1], z = pts[pt * 3 + 2];
#pragma unroll 1
        for (int d = 0; d < 2; ++d) {
            float acc[2][NCH];
#pragma unroll
            for (int l = 0; l < 2; ++l)
#pragma unroll
                for (int i = 0; i < NCH; ++i) acc[l][i] = 0.f;
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const int o = k % 3, lvl = k / 3;
                const Plane& P = planes.p[2 * (3 * d + o) + lvl];
                const float u = (o == 2) ? y : x, v = (o == 0) ? y : z;
                int x0, x1, y0, y1; float tx, ty;
                axis(u, P.w, x0, x1, tx); axis(v, P.h, y0, y1, ty);
                const float w00 = (1 - tx) * (1 - ty), w01 = tx * (1 - ty), w10 = (1 - tx) * ty, w11 = tx * ty;
                const unsigned t00 = (y0 * P.w + x0) * 32u, t01 = (y0 * P.w + x1) * 32u, t10 = (y1 * P.w + x0) * 32u, t11 = (y1 * P.w + x1) * 32u;
                if (MAP == 2) {
                    const unsigned c = 4u * piece;
                    const float4_t a00 = *(const float4_t*)(P.data + t00 + c), a01 = *(const float4_t*)(P.data + t01 + c);
                    const float4_t a10 = *(const float4_t*)(P.data + t10 + c), a11 = *(const float4_t*)(P.data + t11 + c);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[lvl][i] += a00[i] * w00 + a01[i] * w01 + a10[i] * w10 + a11[i] * w11;
                } else {
                    // MAP 0: the kernel's own split (channels 8q..8q+7: bytes 32q and 32q+16); MAP 1: contiguous halves
                    const unsigned ca = (MAP == 0) ? 8u * piece : 4u * piece, cb = (MAP == 0) ? 8u * piece + 4u : 16u + 4u * piece;
                    const float4_t a00 = *(const float4_t*)(P.data + t00 + ca), b00 = *(const float4_t*)(P.data + t00 + cb);
                    const float4_t a01 = *(const float4_t*)(P.data + t01 + ca), b01 = *(const float4_t*)(P.data + t01 + cb);
                    const float4_t a10 = *(const float4_t*)(P.data + t10 + ca), b10 = *(const float4_t*)(P.data + t10 + cb);
                    const float4_t a11 = *(const float4_t*)(P.data + t11 + ca), b11 = *(const float4_t*)(P.data + t11 + cb);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        acc[lvl][i] += a00[i] * w00 + a01[i] * w01 + a10[i] * w10 + a11[i] * w11;
                        acc[lvl][4 + i] += b00[i] * w00 + b01[i] * w01 + b10[i] * w10 + b11[i] * w11;
                    }
                }
                if (PIPE == 0) __builtin_amdgcn_sched_barrier(0);
            }
            float* dst = out + (size_t)pt * 128 + d * 64;
#pragma unroll
            for (int l = 0; l < 2; ++l) {
                if (MAP == 2) {
                    *(float4_t*)(dst + l * 32 + 4 * piece) = (float4_t){acc[l][0], acc[l][1], acc[l][2], acc[l][3]};
                } else {
                    const int ca = (MAP == 0) ? 8 * piece : 4 * piece, cb = (MAP == 0) ? 8 * piece + 4 : 16 + 4 * piece;
                    *(float4_t*)(dst + l * 32 + ca) = (float4_t){acc[l][0], acc[l][1], acc[l][2], acc[l][3]};
                    *(float4_t*)(dst + l * 32 + cb) = (float4_t){acc[l][4], acc[l][5], acc[l][6], acc[l][7]};
                }
            }
        }
    }
}

static double urand(unsigned long long& s) { s = s * 6364136223846793005ull + 1442695040888963407ull; return (double)(s >> 11) / 9007199254740992.0; }

int main(int argc, char** argv) {
    const int R = argc > 1 ? atoi(argv[1]) : 4096, S = argc > 2 ? atoi(argv[2]) : 64;
    const int N = R * S;
    const float bound[3][2] = {{-1.9f, 7.94f}, {-2.2f, 4.52f}, {-2.5f, 2.54f}};
    const int dims[12][2] = {{27, 41}, {111, 164}, {21, 41}, {84, 164}, {21, 27}, {84, 111},
                             {27, 41}, {223, 328}, {21, 41}, {168, 328}, {21, 27}, {168, 223}};
    Planes P;
    size_t total = 0;
    for (int i = 0; i < 12; ++i) total += (size_t)dims[i][0] * dims[i][1] * 32;
    std::vector<float> hp(total);
    unsigned long long seed = 12345;
    for (auto& v : hp) v = (float)(urand(seed) - 0.5) * 0.02f;
    float* dplanes;
    CK(hipMalloc(&dplanes, total * 4));
    CK(hipMemcpy(dplanes, hp.data(), total * 4, hipMemcpyHostToDevice));
    size_t off = 0;
    for (int i = 0; i < 12; ++i) { P.p[i].data = dplanes + off; P.p[i].h = dims[i][0]; P.p[i].w = dims[i][1]; off += (size_t)dims[i][0] * dims[i][1] * 32; }
    // rays from the room centre, Replica pinhole (fx = fy = 600, 1200 x 680), depth U(0.5, 2.5), 56 + 8 samples
    std::vector<float> hpts((size_t)N * 3);
    const float c[3] = {(bound[0][0] + bound[0][1]) / 2, (bound[1][0] + bound[1][1]) / 2, (bound[2][0] + bound[2][1]) / 2};
    for (int r = 0; r < R; ++r) {
        const double px = urand(seed) * 1200, py = urand(seed) * 680, dep = 0.5 + 2.0 * urand(seed);
        const double dir[3] = {(px - 599.5) / 600.0, -(py - 339.5) / 600.0, -1.0};
        std::vector<double> zs(S);
        const int ns = S - 8;
        for (int s = 0; s < ns; ++s) zs[s] = 1.2 * dep * (s + urand(seed)) / ns;
        for (int s = 0; s < 8; ++s) zs[ns + s] = dep - 0.09 + 0.18 * (s + urand(seed)) / 8;
        std::sort(zs.begin(), zs.end());
        for (int s = 0; s < S; ++s)
            for (int a = 0; a < 3; ++a) {
                const double w = c[a] + dir[a] * zs[s];
                hpts[((size_t)r * S + s) * 3 + a] = (float)((w - bound[a][0]) / (bound[a][1] - bound[a][0]) * 2 - 1);
            }
    }
    float *dpts, *dout;
    CK(hipMalloc(&dpts, hpts.size() * 4));
    CK(hipMemcpy(dpts, hpts.data(), hpts.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&dout, (size_t)N * 128 * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int nblk = (N / 64 + 3) / 4;
    std::vector<float> ref((size_t)N * 128), got((size_t)N * 128);
    // dynamic LDS only caps the occupancy, so that the variants are compared at the same waves per SIMD
    const int lds = argc > 3 ? atoi(argv[3]) : 80 * 1024;
    printf("dynamic LDS %d B per workgroup -> at most %d waves per SIMD\n", lds, lds ? 160 * 1024 / lds : 8);
    auto run = [&](const char* name, auto kern, bool is_ref) {
        CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), lds, 0, P, dpts, N, dout);
        CK(hipDeviceSynchronize());
        float best = 1e9f, sum = 0;
        for (int i = 0; i < 20; ++i) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kern, dim3(nblk), dim3(256), lds, 0, P, dpts, N, dout);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = fminf(best, ms); sum += ms;
        }
        CK(hipMemcpy(got.data(), dout, got.size() * 4, hipMemcpyDeviceToHost));
        double err = 0;
        if (is_ref) ref = got; else for (size_t i = 0; i < got.size(); ++i) err = fmax(err, fabs((double)got[i] - ref[i]));
        printf("%-34s  min %7.1f us  mean %7.1f us   %.2f TB/s of 16-B-per-lane requests   max|diff vs A| %.2e\n", name, best * 1e3,
               sum / 20 * 1e3, (double)N * 6144 / (best * 1e-3) / 1e12, err);
    };
    printf("gather of %d x %d points over the 12 room0 planes (%.1f MB), + %.0f MB of feature stores\n", R, S, total * 4 / 1e6, N * 512 / 1e6);
    run("A  lane = 16q + r   (kernel today)", gather_kernel<0, 0>, true);
    run("A  ..., loads free to hoist", gather_kernel<0, 1>, false);
    run("B  lane = 4p + q    (quad = 64 B)", gather_kernel<1, 0>, false);
    run("B  ..., loads free to hoist", gather_kernel<1, 1>, false);
    run("D  lane = 8p + o    (8 lanes = line)", gather_kernel<2, 0>, false);
    run("D  ..., loads free to hoist", gather_kernel<2, 1>, false);
    return 0;
}
