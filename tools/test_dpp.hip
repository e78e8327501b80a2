// Hardware check of the DPP forms eslam_common.h relies on (gfx950): hipcc --offload-arch=gfx950 -O3 tools/test_dpp.hip -o tools/bin/test_dpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_from(float ident, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, ident), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
__global__ void k(const float* in, float* out) {
    const int lane = threadIdx.x;
    const float v = in[lane];
    out[0 * 64 + lane] = dpp_from<0x138, 0xf>(-7.0f, v);       // wave_shr:1: lane i <- lane i-1, lane 0 keeps -7
    out[1 * 64 + lane] = dpp_from<0x130, 0xf>(-7.0f, v);       // wave_shl:1: lane i <- lane i+1, lane 63 keeps -7
    float s = v;                                               // suffix sum: row_shl steps + row totals
    s += dpp_from<0x101, 0xf>(0.f, s);
    s += dpp_from<0x102, 0xf>(0.f, s);
    s += dpp_from<0x104, 0xf>(0.f, s);
    s += dpp_from<0x108, 0xf>(0.f, s);
    const float t1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s), 16));
    const float t2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s), 32));
    const float t3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s), 48));
    const int row = lane >> 4;
    out[2 * 64 + lane] = s + (row == 0 ? t1 + (t2 + t3) : row == 1 ? t2 + t3 : row == 2 ? t3 : 0.f);
    float p = v;                                               // prefix sum as in eslam_common.h
    p += dpp_from<0x111, 0xf>(0.f, p);
    p += dpp_from<0x112, 0xf>(0.f, p);
    p += dpp_from<0x114, 0xf>(0.f, p);
    p += dpp_from<0x118, 0xf>(0.f, p);
    p += dpp_from<0x142, 0xa>(0.f, p);
    p += dpp_from<0x143, 0xc>(0.f, p);
    out[3 * 64 + lane] = p;
}
int main() {
    float h[64], o[4 * 64], *din, *dout;
    for (int i = 0; i < 64; ++i) h[i] = 1.0f + 0.01f * i;
    hipMalloc(&din, sizeof h); hipMalloc(&dout, sizeof o);
    hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout);
    hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) {
        const float shr = i ? h[i - 1] : -7.f, shl = i < 63 ? h[i + 1] : -7.f;
        double suf = 0, pre = 0;
        for (int j = i; j < 64; ++j) suf += h[j];
        for (int j = 0; j <= i; ++j) pre += h[j];
        if (o[i] != shr) { ++bad; printf("wave_shr lane %d: %g expected %g\n", i, o[i], shr); }
        if (o[64 + i] != shl) { ++bad; printf("wave_shl lane %d: %g expected %g\n", i, o[64 + i], shl); }
        if (std::fabs(o[128 + i] - suf) > 1e-4) { ++bad; printf("suffix lane %d: %g expected %g\n", i, o[128 + i], suf); }
        if (std::fabs(o[192 + i] - pre) > 1e-4) { ++bad; printf("prefix lane %d: %g expected %g\n", i, o[192 + i], pre); }
    }
    printf("dpp check: %d mismatches\n", bad);
    return bad != 0;
}
