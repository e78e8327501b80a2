// Ray generation and along-ray sampling kernels (K1-K4 of SURVEY.md section 2.1).
// All of this is elementwise / per-ray work with a few bytes per ray; the kernels exist to remove ~40 tiny
// PyTorch launches and 6 host syncs per iteration from the caller's critical path, and to produce z_vals that are
// bit-identical to the reference's float32 arithmetic (explicitly rounded ops, no FMA contraction).
#include "eslam_decode_tile.h"

// hipcc contracts a*b+c into an FMA by default and its __fmul_rn/__fadd_rn are plain operators, so the explicit
// rounding steps below only survive with contraction switched off for this translation unit's own code.
#pragma clang fp contract(off)

// ---------------------------------------------------------------------------------------------------------
// K1: reference src/common.py:87-153
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void pixel_dir(float u, float v, float fx, float fy, float cx, float cy, float d[3]) {
    d[0] = __fdiv_rn(__fsub_rn(u, cx), fx);                 // common.py:92
    d[1] = -__fdiv_rn(__fsub_rn(v, cy), fy);
    d[2] = -1.0f;
}

__device__ __forceinline__ void rotate_dir(const float* __restrict__ c2w, const float d[3], float out[3]) {
#pragma unroll
    for (int j = 0; j < 3; ++j)                              // common.py:96: sum_k dirs[k] * R[j][k]
        out[j] = __fadd_rn(__fadd_rn(__fmul_rn(d[0], c2w[4 * j + 0]), __fmul_rn(d[1], c2w[4 * j + 1])),
                           __fmul_rn(d[2], c2w[4 * j + 2]));
}

__global__ void sample_rays_kernel(const int64_t* __restrict__ indices, int b, int n, int H0, int H1, int W0, int W1,
                                   int H, int W, float fx, float fy, float cx, float cy,
                                   const float* __restrict__ c2ws, const float* __restrict__ depths,
                                   const float* __restrict__ colors, float* __restrict__ rays_o,
                                   float* __restrict__ rays_d, float* __restrict__ depth, float* __restrict__ color) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= b * n) return;
    const int img = i / n;
    const int ww = W1 - W0;
    int64_t idx = indices[i];
    const int64_t npix = (int64_t)ww * (H1 - H0);
    idx = idx < 0 ? 0 : (idx >= npix ? npix - 1 : idx);       // never read outside the window
    const int pu = (int)(idx % ww) + W0;
    const int pv = (int)(idx / ww) + H0;
    const int64_t pix = ((int64_t)img * H + pv) * W + pu;
    depth[i] = depths[pix];                                   // common.py:120-121
    color[3 * i + 0] = colors[3 * pix + 0];
    color[3 * i + 1] = colors[3 * pix + 1];
    color[3 * i + 2] = colors[3 * pix + 2];
    float d[3], rd[3];
    pixel_dir((float)pu, (float)pv, fx, fy, cx, cy, d);
    const float* c2w = c2ws + 16 * img;
    rotate_dir(c2w, d, rd);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        rays_d[3 * i + j] = rd[j];
        rays_o[3 * i + j] = c2w[4 * j + 3];                   // common.py:97
    }
}

// one workgroup per image: g_c2w[j][k<3] = sum_rays g_d[j] * dir[k],  g_c2w[j][3] = sum_rays g_o[j]
__global__ __launch_bounds__(256) void sample_rays_bwd_kernel(const int64_t* __restrict__ indices, int n, int H0,
                                                              int W0, int W1, float fx, float fy, float cx, float cy,
                                                              const float* __restrict__ g_o,
                                                              const float* __restrict__ g_d,
                                                              float* __restrict__ g_c2ws) {
    const int img = blockIdx.x;
    const int ww = W1 - W0;
    float acc[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = 0.0f;
    for (int t = threadIdx.x; t < n; t += blockDim.x) {
        const int i = img * n + t;
        const int64_t idx = indices[i];
        float d[3];
        pixel_dir((float)((int)(idx % ww) + W0), (float)((int)(idx / ww) + H0), fx, fy, cx, cy, d);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float gd = g_d ? g_d[3 * i + j] : 0.0f;
            acc[4 * j + 0] += gd * d[0];
            acc[4 * j + 1] += gd * d[1];
            acc[4 * j + 2] += gd * d[2];
            acc[4 * j + 3] += g_o ? g_o[3 * i + j] : 0.0f;
        }
    }
    __shared__ float red[4][12];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const float s = wave_sum(acc[k]);
        if (lane == 0) red[wave][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < 16) {
        const int j = threadIdx.x >> 2, k = threadIdx.x & 3;
        float v = 0.0f;
        if (j < 3) v = (red[0][4 * j + k] + red[1][4 * j + k]) + (red[2][4 * j + k] + red[3][4 * j + k]);
        g_c2ws[16 * img + threadIdx.x] = v;
    }
}

// reference src/common.py:183-201
__global__ void image_rays_kernel(int H, int W, float fx, float fy, float cx, float cy, const float* __restrict__ c2w,
                                  float* __restrict__ rays_o, float* __restrict__ rays_d) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H * W) return;
    float d[3], rd[3];
    pixel_dir((float)(i % W), (float)(i / W), fx, fy, cx, cy, d);
    rotate_dir(c2w, d, rd);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        rays_d[3 * (int64_t)i + j] = rd[j];
        rays_o[3 * (int64_t)i + j] = c2w[4 * j + 3];
    }
}

// ---------------------------------------------------------------------------------------------------------
// a2: reference src/Mapper.py:322-328, Tracker.py:175-181, Renderer.py:114-115
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float aabb_exit_dev(const float o[3], const float d[3], const Bound& bnd) {
    float t = 0.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float t0 = __fdiv_rn(__fsub_rn(bnd.lo[k], o[k]), d[k]);
        const float t1 = __fdiv_rn(__fsub_rn(bnd.hi[k], o[k]), d[k]);
        // torch.max / torch.min propagate NaN (0/0 when a ray starts on a slab and runs along it)
        float m = fmaxf(t0, t1);
        if (t0 != t0 || t1 != t1) m = __builtin_nanf("");
        if (k == 0) t = m;
        else {
            const bool nan = (t != t) || (m != m);
            t = nan ? __builtin_nanf("") : fminf(t, m);
        }
    }
    return t;
}

__global__ void aabb_exit_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d, int R,
                                 const Bound bnd, float* __restrict__ t_exit) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R) return;
    const float o[3] = {rays_o[3 * i], rays_o[3 * i + 1], rays_o[3 * i + 2]};
    const float d[3] = {rays_d[3 * i], rays_d[3 * i + 1], rays_d[3 * i + 2]};
    t_exit[i] = aabb_exit_dev(o, d, bnd);
}

// the callers' pre-filter as a mask: keep = (t_exit >= gt_depth) [& (gt_depth > 0)]   (Mapper.py:322-328, Tracker.py:175-182)
__global__ void prefilter_kernel(const float* __restrict__ rays_o, const float* __restrict__ rays_d,
                                 const float* __restrict__ gt_depth, int R, const Bound bnd, int need_depth,
                                 uint8_t* __restrict__ keep) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R) return;
    const float o[3] = {rays_o[3 * i], rays_o[3 * i + 1], rays_o[3 * i + 2]};
    const float d[3] = {rays_d[3 * i], rays_d[3 * i + 1], rays_d[3 * i + 2]};
    const float gd = gt_depth[i];
    bool k = aabb_exit_dev(o, d, bnd) >= gd;                 // false for NaN, as the tensor comparison
    if (need_depth) k = k && gd > 0.0f;
    keep[i] = k ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------------------
// K3: reference src/utils/Renderer.py:85-105 and :46-61
// One wave per ray.  The two sequences are ascending, so the "sort" of Renderer.py:102 is a rank merge:
// element position = own index + number of elements of the other sequence in front of it.
// ---------------------------------------------------------------------------------------------------------
// Counter-based uniform numbers for the jitter (Renderer.py:59) and the importance draw (common.py:59) when the caller
// does not inject them: U = hash(seed, step, stream, element) / 2^24 in [0, 1), the grid torch.rand draws from.  step is
// read from device memory (rng.state) so that a replayed hipGraph draws fresh numbers; the forward kernel that consumes
// the samples increments it (eslam_render_fwd*, rng_bump) - one launch per iteration less than a torch.rand pool.
struct RngArg {
    uint32_t seed_lo, seed_hi;
    const uint32_t* state;       // [1] step counter on the device, or NULL (step 0)
    int on;                      // 0: numbers come from the t_rand / t_rand_uni / u arrays
    int perturb;                 // jitter the samples (Renderer.perturb); the importance draw u is needed either way
    uint32_t ray_offset;         // index of this call's ray 0 in the iteration's WHOLE batch: the numbers are keyed on the global
                                 // ray index, so a ray-sharded iteration draws what the unsharded one does
};
struct Rng { uint32_t k0, k1; bool on, perturb; uint32_t ray_offset; };

__device__ __forceinline__ uint32_t mix32(uint32_t h) {
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    return h;
}
__device__ __forceinline__ Rng make_rng(const RngArg a) {
    Rng r;
    const uint32_t step = (a.on && a.state) ? a.state[0] : 0u;
    r.k0 = a.seed_lo ^ (step * 0x9E3779B9u);
    r.k1 = a.seed_hi + step * 0x7F4A7C15u;
    r.on = a.on != 0;
    r.perturb = a.perturb != 0;
    r.ray_offset = a.ray_offset;
    return r;
}
__device__ __forceinline__ float rng_uniform(const Rng& r, uint32_t stream, uint32_t idx) {
    uint32_t h = mix32(idx * 0x9E3779B1u + r.k0 + stream * 0x85EBCA77u);
    h = mix32(h + r.k1);
    return (float)(h >> 8) * 5.9604644775390625e-08f;      // 2^-24
}

__device__ __forceinline__ float jitter_one(const float* zs, int i, int S, float t) {
    // Renderer.py:55-61
    const float zi = zs[i];
    const float lower = (i == 0) ? zi : __fmul_rn(0.5f, __fadd_rn(zi, zs[i - 1]));
    const float upper = (i == S - 1) ? zi : __fmul_rn(0.5f, __fadd_rn(zs[i + 1], zi));
    return __fadd_rn(lower, __fmul_rn(__fsub_rn(upper, lower), t));
}

// Counts over a NON-DECREASING array in LDS by bisection: #(a[k] <= v) and #(a[k] < v), k in [0, n).  The same numbers a
// linear scan gives (also for NaN: 0), in <= 8 dependent reads instead of n: the rank merges below spent 56 - 96 dependent
// LDS reads per element (in-kernel stamps: 24 k of a depth-less ray's 55 k cycles were its rank sort).
__device__ __forceinline__ int count_le_sorted(const float* a, int n, float v) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] <= v) lo = mid + 1; else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ int count_lt_sorted(const float* a, int n, float v) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// one wave: the S depth-guided samples of one ray with depth d > 0 (zs: S floats of LDS owned by the wave)
__device__ __forceinline__ void depth_guided_row(float d, int ray, int n_strat, int n_imp, float c15, float c3,
                                                 const float* __restrict__ t_free, const float* __restrict__ t_surf,
                                                 const float* __restrict__ t_rand, float* __restrict__ out, float* zs,
                                                 int lane, const Rng& rng) {
    // out: the ray's row of z_vals (or an LDS row: eslam_loss_set_sizes); `ray` indexes t_rand and, with rng.ray_offset, the
    // in-kernel random numbers
    const int S = n_strat + n_imp;
    const float d12 = __fmul_rn(1.2f, d);          // Renderer.py:100
    const float dlo = __fsub_rn(d, c15);           // Renderer.py:97
    // both sequences into the upper half of the wave's scratch first (zs has ESLAM_MAX_SAMPLES >= 2 S floats only when
    // S <= 128; larger S recomputes the other sequence's values from t_free / t_surf as before)
    const bool staged = 2 * S <= ESLAM_MAX_SAMPLES;
    float* raw = zs + S;
    // d12 * t_free and dlo + c3 * t_surf are non-decreasing in the index when the factors are >= 0 (t_free / t_surf are
    // ascending linspaces; rounding is monotone): then the counts below are bisections
    const bool sorted_in = d12 >= 0.0f && c3 >= 0.0f;
    if (staged) {
        for (int i = lane; i < S; i += WAVE)
            raw[i] = (i < n_strat) ? __fadd_rn(0.0f, __fmul_rn(d12, t_free[i])) : __fadd_rn(dlo, __fmul_rn(c3, t_surf[i - n_strat]));
        WAVE_SYNC();
    }
    for (int i = lane; i < S; i += WAVE) {
        float val;
        int pos;
        if (i < n_strat) {
            val = __fadd_rn(0.0f, __fmul_rn(d12, t_free[i]));
            int cnt = 0;
            if (staged && sorted_in) cnt = count_lt_sorted(raw + n_strat, n_imp, val);
            else if (staged) { for (int j = 0; j < n_imp; ++j) cnt += (raw[n_strat + j] < val) ? 1 : 0; }
            else { for (int j = 0; j < n_imp; ++j) cnt += (__fadd_rn(dlo, __fmul_rn(c3, t_surf[j])) < val) ? 1 : 0; }
            pos = i + cnt;
        } else {
            const int j = i - n_strat;
            val = __fadd_rn(dlo, __fmul_rn(c3, t_surf[j]));
            int cnt = 0;
            if (staged && sorted_in) cnt = count_le_sorted(raw, n_strat, val);
            else if (staged) { for (int k = 0; k < n_strat; ++k) cnt += (raw[k] <= val) ? 1 : 0; }
            else { for (int k = 0; k < n_strat; ++k) cnt += (__fadd_rn(0.0f, __fmul_rn(d12, t_free[k])) <= val) ? 1 : 0; }
            pos = j + cnt;
        }
        zs[pos] = val;
    }
    WAVE_SYNC();
    for (int i = lane; i < S; i += WAVE)
        out[i] = t_rand ? jitter_one(zs, i, S, t_rand[(int64_t)ray * S + i])
                        : (rng.on && rng.perturb) ? jitter_one(zs, i, S, rng_uniform(rng, 0u, (rng.ray_offset + (uint32_t)ray) * (uint32_t)S + (uint32_t)i)) : zs[i];
}

__global__ __launch_bounds__(256) void sample_z_kernel(const float* __restrict__ gt_depth, int R, int n_strat,
                                                       int n_imp, float c15, float c3,
                                                       const float* __restrict__ t_free,
                                                       const float* __restrict__ t_surf,
                                                       const float* __restrict__ t_rand, float* __restrict__ z_vals) {
    __shared__ float zs_all[4][ESLAM_MAX_SAMPLES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ray = blockIdx.x * 4 + wave;
    if (ray >= R) return;
    const float d = gt_depth[ray];
    if (!(d > 0.0f)) return;                       // Renderer.py:92: handled by the importance sampler
    depth_guided_row(d, ray, n_strat, n_imp, c15, c3, t_free, t_surf, t_rand, z_vals + (int64_t)ray * (n_strat + n_imp), zs_all[wave], lane,
                     Rng{0u, 0u, false, false, 0u});
}

// ---------------------------------------------------------------------------------------------------------
// K4: reference src/utils/Renderer.py:108-134 + src/common.py:41-77.  One wave per zero-depth ray.
// ---------------------------------------------------------------------------------------------------------
// ONE WORKGROUP PER RAY.  A ray with depth (WITH_DEPTH: the same launch also produces those rows, K3) is one wave's job
// and the other three leave at once; a depth-less ray is worked on by all four waves: the SDF decode of its n_strat
// points - a chain of 6 dependent gather + MLP blocks at n_strat = 88 when one wave did it alone, 42 us of this kernel for a
// 1024-ray ScanNet batch with 10 % depth-less rays - is dealt out block by block, then wave 0 inverts the cdf.
// (Four rays per workgroup with the four waves taking the workgroup's depth-less rays one after the other was SLOWER, 76 us:
// the kernel then lasts as long as its unluckiest workgroup, which holds two or three such rays.)
template <bool CL, bool WITH_DEPTH>
__global__ __launch_bounds__(256, 2) void importance_z_kernel(const PlaneSet planes, const eslam_decoders_t dec,
                                                           const Bound bnd, const float* __restrict__ rays_o,
                                                           const float* __restrict__ rays_d,
                                                           const float* __restrict__ gt_depth, int R, int n_strat,
                                                           int n_imp, const float* __restrict__ t_free,
                                                           const float* __restrict__ t_rand_uni,
                                                           const float* __restrict__ u_rand,
                                                           float* __restrict__ z_vals, float c15, float c3,
                                                           const float* __restrict__ t_surf,
                                                           const float* __restrict__ t_rand, const RngArg rng_arg) {
    const Rng rng = make_rng(rng_arg);
    __shared__ __attribute__((aligned(16))) float wlds[DEC_LDS];          // the SDF decoder only
    __shared__ float zu[ESLAM_MAX_SAMPLES];      // jittered uniform samples (depth rows: the rank merge's scratch)
    __shared__ float wt[ESLAM_MAX_SAMPLES];      // uniform samples before the jitter, then weights -> cdf
    __shared__ float zn[ESLAM_MAX_SAMPLES];      // importance samples
    __shared__ float al[ESLAM_MAX_SAMPLES];      // alpha of the n_strat points
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ray = blockIdx.x;
    const float d0 = gt_depth[ray];
    if (d0 > 0.0f) {                                 // uniform over the workgroup
        if (WITH_DEPTH && wave == 0)
            depth_guided_row(d0, ray, n_strat, n_imp, c15, c3, t_free, t_surf, t_rand, z_vals + (int64_t)ray * (n_strat + n_imp), zu, lane, rng);
        return;
    }
    stage_decoder_weights_one(wlds, dec, 0, threadIdx.x, blockDim.x);
    const int r = lane & 15, q = lane >> 4;                                  // MFMA role (eslam_decode_tile.h)
    const int gp = gather_point<CL>(lane), gq = gather_piece<CL>(lane);     // gather role
    const float beta = dec.beta[0];
    const int S = n_strat + n_imp;
    const float o[3] = {rays_o[3 * ray], rays_o[3 * ray + 1], rays_o[3 * ray + 2]};
    const float d[3] = {rays_d[3 * ray], rays_d[3 * ray + 1], rays_d[3 * ray + 2]};
    const float far = __fadd_rn(aabb_exit_dev(o, d, bnd), 0.01f);          // Renderer.py:114-117

    // Renderer.py:119: near*(1-t) + far*t with near = 0
    for (int i = threadIdx.x; i < n_strat; i += 256) {
        const float t = t_free[i];
        wt[i] = __fadd_rn(__fmul_rn(0.0f, __fsub_rn(1.0f, t)), __fmul_rn(far, t));
    }
    __syncthreads();                                 // (also: the decoder weights are staged)
    for (int i = threadIdx.x; i < n_strat; i += 256)
        zu[i] = t_rand_uni ? jitter_one(wt, i, n_strat, t_rand_uni[(int64_t)ray * n_strat + i])
                           : (rng.on && rng.perturb) ? jitter_one(wt, i, n_strat, rng_uniform(rng, 1u, (rng.ray_offset + (uint32_t)ray) * (uint32_t)n_strat + (uint32_t)i)) : wt[i];
    __syncthreads();

    // SDF decode of the n_strat points (geometry planes only) -> alpha (Renderer.py:122-127); wave w takes blocks w, w+4, ...
    const int nblk = (n_strat + 15) >> 4;
#pragma unroll 1
    for (int b = wave; b < nblk; b += 4) {
        const int oz0 = opaque_zero(b);
        const float z = zu[min(16 * b + gp, n_strat - 1)];
        // Renderer.py:122: o + d*z (mul then add, as torch does)
        const float x = norm_coord(__fadd_rn(o[0], __fmul_rn(d[0], z)), bnd.lo[0], bnd.hi[0]);
        const float y = norm_coord(__fadd_rn(o[1], __fmul_rn(d[1], z)), bnd.lo[1], bnd.hi[1]);
        const float zz = norm_coord(__fadd_rn(o[2], __fmul_rn(d[2], z)), bnd.lo[2], bnd.hi[2]);
        float feat[16];
        gather_features<CL>(planes, 0, x, y, zz, gq, feat, oz0);
        to_mfma_role<CL, 16>(feat, lane);
        DecFrag f;
        load_dec_frag(f, wlds + oz0, r, q);
        float4_t h1, h2;
        mlp_hidden(f, feat, h1, h2);
        float4_t out = *(const float4_t*)(wlds + DEC_B3);
        mlp_out_accum(f, h2, 0, r, out);             // rows 0..3 of the padded output layer: lanes 0..15 hold point 16b + lane
        if (lane < 16 && 16 * b + lane < n_strat) {
            const float sdf = tanhf(out[0]);
            al[16 * b + lane] = 1.0f - expf(-beta * sigmoidf_(-sdf * beta));
        }
    }
    __syncthreads();
    if (wave != 0) return;

    // transmittance and weights (Renderer.py:128-129), 64 samples at a time
    float trans_in = 1.0f;
    for (int c0 = 0; c0 < n_strat; c0 += WAVE) {
        const bool valid = c0 + lane < n_strat;
        const float alpha = valid ? al[c0 + lane] : 0.0f;
        const float fac = valid ? (1.0f - alpha) + 1e-10f : 1.0f;
        const float pin = wave_incl_prod(fac, lane);
        const float pex = wave_up1(pin, 1.0f);
        if (valid) wt[c0 + lane] = alpha * (trans_in * pex);
        trans_in *= wave_lane<63>(pin);
    }
    WAVE_SYNC();

    // sample_pdf (common.py:41-77): bins = mids (n_strat-1), pdf = weights[1:-1] un-normalised (n_strat-2)
    const int nb = n_strat - 1;             // len(bins) == len(cdf)
    if (lane == 0) {                        // sequential cumsum, the order torch.cumsum uses on one row
        float run = 0.0f;
        float prev = wt[1];
        wt[0] = 0.0f;                       // cdf[0] = 0  (overwrites weights[0], unused by sample_pdf)
        for (int k = 1; k < nb; ++k) {
            const float wk = prev;          // weights[k] (pdf[k-1]); read before it is overwritten
            prev = wt[k + 1];
            run = __fadd_rn(run, wk);
            wt[k] = run;                    // cdf[k]
        }
    }
    WAVE_SYNC();
    for (int i = lane; i < n_imp; i += WAVE) {
        const float u = u_rand ? u_rand[(int64_t)ray * n_imp + i] : rng_uniform(rng, 2u, (rng.ray_offset + (uint32_t)ray) * (uint32_t)n_imp + (uint32_t)i);
        // searchsorted(cdf, u, right=True): #entries <= u; the cdf is a cumsum of weights >= 0 (alpha in [0,1]): non-decreasing
        const int inds = count_le_sorted(wt, nb, u);
        const int below = max(inds - 1, 0);
        const int above = min(nb - 1, inds);
        const float c0v = wt[below], c1v = wt[above];
        const float b0 = __fmul_rn(0.5f, __fadd_rn(zu[below + 1], zu[below]));
        const float b1 = __fmul_rn(0.5f, __fadd_rn(zu[above + 1], zu[above]));
        float denom = __fsub_rn(c1v, c0v);
        if (denom < 1e-5f) denom = 1.0f;
        const float t = __fdiv_rn(__fsub_rn(u, c0v), denom);
        zn[i] = __fadd_rn(b0, __fmul_rn(t, __fsub_rn(b1, b0)));
    }
    WAVE_SYNC();

    // Renderer.py:133: sort(cat(z_uni, z_samples)) as a stable rank sort (values are what matter)
    float* out_row = z_vals + (int64_t)ray * S;
    // zu (the jittered stratified samples) is non-decreasing for an ordinary ray: an element's rank is then a bisection in zu
    // plus a scan of the n_imp importance samples.  Checked, not assumed (far <= 0, or a jitter that rounds one ulp past its
    // bin): any other ray keeps the all-pairs count, which is a permutation whatever the values are.
    bool inversion = false;
    for (int i = lane; i + 1 < n_strat; i += WAVE) inversion |= !(zu[i] <= zu[i + 1]);
    const bool zu_sorted = __ballot(inversion) == 0ull;
    for (int i = lane; i < S; i += WAVE) {
        const float v = (i < n_strat) ? zu[i] : zn[i - n_strat];
        int pos = 0;
        if (zu_sorted) {
            if (i < n_strat) {
                // among zu itself: every k < i counts (smaller, or tied and earlier), no k > i does
                pos = i;
                for (int j = 0; j < n_imp; ++j) pos += (zn[j] < v) ? 1 : 0;        // k = n_strat + j > i: ties do not count
            } else {
                const int j0 = i - n_strat;
                pos = count_le_sorted(zu, n_strat, v);     // k < i for every zu element: ties count
                for (int j = 0; j < n_imp; ++j) pos += (zn[j] < v || (zn[j] == v && j < j0)) ? 1 : 0;
            }
        } else {
            for (int k = 0; k < S; ++k) {
                const float o2 = (k < n_strat) ? zu[k] : zn[k - n_strat];
                pos += (o2 < v || (o2 == v && k < i)) ? 1 : 0;
            }
        }
        out_row[pos] = v;
    }
}


// ---------------------------------------------------------------------------------------------------------
// The five set sizes the mapping loss takes its means over (src/Mapper.py:136-140,343,346) for a WHOLE batch of rays,
// without rendering it: they depend on gt_depth and on the depth-guided z_vals of the rays WITH depth only (a depth-less ray
// counts towards the colour term alone), i.e. on K3 - replayed here into LDS, same arithmetic and same random numbers as
// importance_z_kernel<., true> / sample_z_kernel write to memory, and classified with the loss's own region rule.  Every
// rank of a ray-sharded iteration runs this on the iteration's whole batch instead of all-reducing its shard's counts
// ("compute them redundantly on every rank"): no collective between forward and backward.
// acc_out [16]: the count slots (N_front, N_center, N_tail, N_depth, N_color) as floats, every other slot 0.
// scratch: 7 lines of 32 unsigned, zeroed once by the caller; left zeroed.
// ---------------------------------------------------------------------------------------------------------
#include "eslam_loss_final.h"
#include "eslam_shard_dev.h"
struct SetSizesArgs {
    const float* gt_depth;
    const uint8_t* ray_mask;
    int R, n_strat, n_imp;
    float c15, c3;
    const float* t_free;
    const float* t_surf;
    const float* t_rand;
    RngArg rng;
    Trunc tr;
    unsigned* scratch;
    float* acc_out;
};
// block `bid` of `nblocks` blocks of 256 threads (the ticket counts to nblocks)
__device__ __forceinline__ void set_sizes_block(const SetSizesArgs& a, const int bid, const int nblocks) {
    const Rng rng = make_rng(a.rng);
    __shared__ float zs_all[4][ESLAM_MAX_SAMPLES];
    __shared__ float zo_all[4][ESLAM_MAX_SAMPLES];
    __shared__ unsigned part[4][5];
    __shared__ unsigned ticket;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int S = a.n_strat + a.n_imp;
    unsigned cnt[5] = {0u, 0u, 0u, 0u, 0u};                    // front, center, tail, depth, colour: this lane's share
    for (int ray = bid * 4 + wave; ray < a.R; ray += nblocks * 4) {
        if (a.ray_mask && a.ray_mask[ray] == 0) continue;      // (uniform over the wave)
        if (lane == 0) cnt[4] += 3u;
        const float d = a.gt_depth[ray];
        if (!(d > 0.0f)) continue;
        if (lane == 0) cnt[3] += 1u;
        depth_guided_row(d, ray, a.n_strat, a.n_imp, a.c15, a.c3, a.t_free, a.t_surf, a.t_rand, zo_all[wave], zs_all[wave], lane, rng);
        WAVE_SYNC();
        for (int i = lane; i < S; i += WAVE) {
            const int reg = sdf_region(zo_all[wave][i], d, a.tr);
            if (reg < 3) cnt[reg] += 1u;
        }
        WAVE_SYNC();
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const unsigned t = (unsigned)__builtin_amdgcn_readlane((int)wave_incl_sum_u(cnt[k]), WAVE - 1);
        if (lane == 0) part[wave][k] = t;
    }
    __syncthreads();
    if (threadIdx.x < 5) {
        const unsigned tot = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
        if (tot) {
            const unsigned old = atomicAdd(a.scratch + 32 * (threadIdx.x + 1), tot);
            asm volatile("" ::"v"(old));                       // performed once its result is back (as loss_finalize)
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) ticket = atomicAdd(a.scratch, 1u);
    __syncthreads();
    if (ticket != (unsigned)nblocks - 1u) return;
    if (threadIdx.x < 16) {
        float v = 0.0f;
        int k = -1;
        if (threadIdx.x == A_N_FRONT) k = 0;
        else if (threadIdx.x == A_N_CENTER) k = 1;
        else if (threadIdx.x == A_N_TAIL) k = 2;
        else if (threadIdx.x == A_N_DEPTH) k = 3;
        else if (threadIdx.x == A_N_COLOR) k = 4;
        if (k >= 0) v = (float)atomicExch(a.scratch + 32 * (k + 1), 0u);
        a.acc_out[threadIdx.x] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) atomicExch(a.scratch, 0u);
}

__global__ __launch_bounds__(256) void loss_set_sizes_kernel(const SetSizesArgs a) { set_sizes_block(a, blockIdx.x, gridDim.x); }

// What a ray-sharded iteration does beside its sampler and forward kernel, as ONE launch (a replayed hipGraph pays ~3 us per
// node whatever its size: the three pieces as launches of their own, with the memsets and copies around them, held the sampler
// back by 40 us): blocks [0, n_clear) zero the previous iteration's marked texels + dense tail, the next n_sizes blocks form
// the loss's global set sizes, the last n_mark blocks mark the texels the whole batch can touch.  The pieces are independent.
__global__ __launch_bounds__(256) void shard_prologue_kernel(const ClearArgs c, const int n_clear, const SetSizesArgs s,
                                                            const int n_sizes, const MarkArgs m, const int n_mark) {
    int b = blockIdx.x;
    if (b < n_clear) { clear_blocks_block(c, b, n_clear); return; }
    b -= n_clear;
    if (b < n_sizes) { set_sizes_block(s, b, n_sizes); return; }
    mark_rays_block(m, b - n_sizes, n_mark);
}

// ---------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------
bool eslam_planes_channels_last(const eslam_plane_t* planes, int first, int count);
int eslam_validate_planes(const eslam_plane_t* planes, int first, int count);

static Bound make_bound(const float* b6) {
    Bound b;
    for (int k = 0; k < 3; ++k) {
        b.lo[k] = b6[2 * k];
        b.hi[k] = b6[2 * k + 1];
    }
    return b;
}

extern "C" int eslam_sample_rays(const int64_t* indices, int b, int n, int H0, int H1, int W0, int W1, int H, int W,
                                 float fx, float fy, float cx, float cy, const float* c2ws, const float* depths,
                                 const float* colors, float* rays_o, float* rays_d, float* depth, float* color,
                                 eslam_stream_t stream) {
    if (b <= 0 || n <= 0) return 0;
    if (!(0 <= H0 && H0 < H1 && H1 <= H && 0 <= W0 && W0 < W1 && W1 <= W)) {
        eslam_set_error("eslam_sample_rays: bad window [%d,%d)x[%d,%d) for image %dx%d", H0, H1, W0, W1, H, W);
        return 1;
    }
    if (!indices || !c2ws || !depths || !colors || !rays_o || !rays_d || !depth || !color) {
        eslam_set_error("eslam_sample_rays: null argument");
        return 1;
    }
    const int total = b * n;
    hipLaunchKernelGGL(sample_rays_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, indices, b, n,
                       H0, H1, W0, W1, H, W, fx, fy, cx, cy, c2ws, depths, colors, rays_o, rays_d, depth, color);
    return eslam_check_launch("sample_rays_kernel");
}

extern "C" int eslam_sample_rays_bwd(const int64_t* indices, int b, int n, int H0, int W0, int W1, float fx, float fy,
                                     float cx, float cy, const float* g_rays_o, const float* g_rays_d, float* g_c2ws,
                                     eslam_stream_t stream) {
    if (b <= 0) return 0;
    if (!indices || !g_c2ws || W1 <= W0) {
        eslam_set_error("eslam_sample_rays_bwd: bad argument");
        return 1;
    }
    hipLaunchKernelGGL(sample_rays_bwd_kernel, dim3(b), dim3(256), 0, (hipStream_t)stream, indices, n, H0, W0, W1, fx,
                       fy, cx, cy, g_rays_o, g_rays_d, g_c2ws);
    return eslam_check_launch("sample_rays_bwd_kernel");
}

extern "C" int eslam_image_rays(int H, int W, float fx, float fy, float cx, float cy, const float* c2w, float* rays_o,
                                float* rays_d, eslam_stream_t stream) {
    if (H <= 0 || W <= 0) return 0;
    if (!c2w || !rays_o || !rays_d) {
        eslam_set_error("eslam_image_rays: null argument");
        return 1;
    }
    hipLaunchKernelGGL(image_rays_kernel, dim3((H * W + 255) / 256), dim3(256), 0, (hipStream_t)stream, H, W, fx, fy,
                       cx, cy, c2w, rays_o, rays_d);
    return eslam_check_launch("image_rays_kernel");
}

extern "C" int eslam_aabb_exit(const float* rays_o, const float* rays_d, int R, const float* bound6_host,
                               float* t_exit, eslam_stream_t stream) {
    if (R <= 0) return 0;
    if (!rays_o || !rays_d || !bound6_host || !t_exit) {
        eslam_set_error("eslam_aabb_exit: null argument");
        return 1;
    }
    hipLaunchKernelGGL(aabb_exit_kernel, dim3((R + 255) / 256), dim3(256), 0, (hipStream_t)stream, rays_o, rays_d, R,
                       make_bound(bound6_host), t_exit);
    return eslam_check_launch("aabb_exit_kernel");
}

extern "C" int eslam_prefilter(const float* rays_o, const float* rays_d, const float* gt_depth, int R,
                               const float* bound6_host, int need_depth, uint8_t* keep, eslam_stream_t stream) {
    if (R <= 0) return 0;
    if (!rays_o || !rays_d || !gt_depth || !bound6_host || !keep) {
        eslam_set_error("eslam_prefilter: null argument");
        return 1;
    }
    hipLaunchKernelGGL(prefilter_kernel, dim3((R + 255) / 256), dim3(256), 0, (hipStream_t)stream, rays_o, rays_d,
                       gt_depth, R, make_bound(bound6_host), need_depth, keep);
    return eslam_check_launch("prefilter_kernel");
}

extern "C" int eslam_sample_z(const float* gt_depth, int R, int n_strat, int n_imp, double truncation,
                              const float* t_free, const float* t_surf, const float* t_rand, float* z_vals,
                              eslam_stream_t stream) {
    if (R <= 0) return 0;
    if (n_strat < 1 || n_imp < 0 || n_strat + n_imp > ESLAM_MAX_SAMPLES) {
        eslam_set_error("eslam_sample_z: n_strat=%d n_imp=%d unsupported (sum <= %d)", n_strat, n_imp,
                        ESLAM_MAX_SAMPLES);
        return 1;
    }
    if (!gt_depth || !t_free || (n_imp > 0 && !t_surf) || !z_vals) {
        eslam_set_error("eslam_sample_z: null argument");
        return 1;
    }
    // Renderer.py:97: (1.5 * truncation) and (3 * truncation) are Python-float products, cast to float32 by torch
    const float c15 = (float)(1.5 * truncation);
    const float c3 = (float)(3.0 * truncation);
    eslam_prof_begin(PROF_SAMPLE_Z, (hipStream_t)stream);
    hipLaunchKernelGGL(sample_z_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, gt_depth, R, n_strat,
                       n_imp, c15, c3, t_free, t_surf, t_rand, z_vals);
    eslam_prof_end(PROF_SAMPLE_Z, (hipStream_t)stream);
    return eslam_check_launch("sample_z_kernel");
}

extern "C" int eslam_importance_z(const eslam_plane_t* planes, const eslam_decoders_t* dec, const float* bound6_host,
                                  const float* rays_o, const float* rays_d, const float* gt_depth, int R, int n_strat,
                                  int n_imp, const float* t_free, const float* t_rand_uni, const float* u,
                                  float* z_vals, eslam_stream_t stream) {
    if (R <= 0) return 0;
    if (n_strat < 3 || n_imp < 0 || n_strat + n_imp > ESLAM_MAX_SAMPLES) {
        eslam_set_error("eslam_importance_z: n_strat=%d n_imp=%d unsupported", n_strat, n_imp);
        return 1;
    }
    if (!planes || !dec || !bound6_host || !rays_o || !rays_d || !gt_depth || !t_free || (n_imp > 0 && !u) || !z_vals) {
        eslam_set_error("eslam_importance_z: null argument");
        return 1;
    }
    if (eslam_validate_planes(planes, 0, 6)) return 1;
    PlaneSet ps;
    for (int i = 0; i < NPL; ++i) ps.p[i] = planes[i < 6 ? i : i - 6];
    const Bound bnd = make_bound(bound6_host);
    dim3 grid(R), block(256);            // one workgroup per ray
    eslam_prof_begin(PROF_IMPORTANCE_Z, (hipStream_t)stream);
    if (eslam_planes_channels_last(planes, 0, 6))
        hipLaunchKernelGGL((importance_z_kernel<true, false>), grid, block, 0, (hipStream_t)stream, ps, *dec, bnd, rays_o,
                           rays_d, gt_depth, R, n_strat, n_imp, t_free, t_rand_uni, u, z_vals, 0.0f, 0.0f,
                           (const float*)nullptr, (const float*)nullptr, RngArg{});
    else
        hipLaunchKernelGGL((importance_z_kernel<false, false>), grid, block, 0, (hipStream_t)stream, ps, *dec, bnd, rays_o,
                           rays_d, gt_depth, R, n_strat, n_imp, t_free, t_rand_uni, u, z_vals, 0.0f, 0.0f,
                           (const float*)nullptr, (const float*)nullptr, RngArg{});
    eslam_prof_end(PROF_IMPORTANCE_Z, (hipStream_t)stream);
    return eslam_check_launch("importance_z_kernel");
}

static int sample_z_all_impl(const char* who, const eslam_plane_t* planes, const eslam_decoders_t* dec,
                             const float* bound6_host, const float* rays_o, const float* rays_d, const float* gt_depth,
                             int R, int n_strat, int n_imp, double truncation, const float* t_free, const float* t_surf,
                             const float* t_rand, const float* t_rand_uni, const float* u, const RngArg rng,
                             float* z_vals, eslam_stream_t stream) {
    if (R <= 0) return 0;
    if (n_strat < 3 || n_imp < 0 || n_strat + n_imp > ESLAM_MAX_SAMPLES) {
        eslam_set_error("%s: n_strat=%d n_imp=%d unsupported", who, n_strat, n_imp);
        return 1;
    }
    if (!planes || !dec || !bound6_host || !rays_o || !rays_d || !gt_depth || !t_free ||
        (n_imp > 0 && ((!u && !rng.on) || !t_surf)) || !z_vals) {
        eslam_set_error("%s: null argument", who);
        return 1;
    }
    if ((int64_t)R * (n_strat + n_imp) >= ((int64_t)1 << 32)) {
        eslam_set_error("%s: batch too large", who);
        return 1;
    }
    if (eslam_validate_planes(planes, 0, 6)) return 1;
    PlaneSet ps;
    for (int i = 0; i < NPL; ++i) ps.p[i] = planes[i < 6 ? i : i - 6];
    const Bound bnd = make_bound(bound6_host);
    const float c15 = (float)(1.5 * truncation);       // as eslam_sample_z
    const float c3 = (float)(3.0 * truncation);
    dim3 grid(R), block(256);            // one workgroup per ray
    eslam_prof_begin(PROF_SAMPLE_Z, (hipStream_t)stream);
    if (eslam_planes_channels_last(planes, 0, 6))
        hipLaunchKernelGGL((importance_z_kernel<true, true>), grid, block, 0, (hipStream_t)stream, ps, *dec, bnd, rays_o,
                           rays_d, gt_depth, R, n_strat, n_imp, t_free, t_rand_uni, u, z_vals, c15, c3, t_surf, t_rand, rng);
    else
        hipLaunchKernelGGL((importance_z_kernel<false, true>), grid, block, 0, (hipStream_t)stream, ps, *dec, bnd, rays_o,
                           rays_d, gt_depth, R, n_strat, n_imp, t_free, t_rand_uni, u, z_vals, c15, c3, t_surf, t_rand, rng);
    eslam_prof_end(PROF_SAMPLE_Z, (hipStream_t)stream);
    return eslam_check_launch("importance_z_kernel<with depth>");
}

extern "C" int eslam_sample_z_all(const eslam_plane_t* planes, const eslam_decoders_t* dec, const float* bound6_host,
                                  const float* rays_o, const float* rays_d, const float* gt_depth, int R, int n_strat,
                                  int n_imp, double truncation, const float* t_free, const float* t_surf,
                                  const float* t_rand, const float* t_rand_uni, const float* u, float* z_vals,
                                  eslam_stream_t stream) {
    return sample_z_all_impl("eslam_sample_z_all", planes, dec, bound6_host, rays_o, rays_d, gt_depth, R, n_strat, n_imp,
                             truncation, t_free, t_surf, t_rand, t_rand_uni, u, RngArg{}, z_vals, stream);
}

extern "C" int eslam_sample_z_all_rng(const eslam_plane_t* planes, const eslam_decoders_t* dec, const float* bound6_host,
                                      const float* rays_o, const float* rays_d, const float* gt_depth, int R, int n_strat,
                                      int n_imp, double truncation, const float* t_free, const float* t_surf, int perturb,
                                      uint64_t seed, const uint32_t* rng_state, int64_t ray_offset, float* z_vals,
                                      eslam_stream_t stream) {
    if (ray_offset < 0 || (ray_offset + (int64_t)(R > 0 ? R : 0)) * (n_strat + n_imp) >= ((int64_t)1 << 32)) {
        eslam_set_error("eslam_sample_z_all_rng: ray_offset %lld out of range", (long long)ray_offset);
        return 1;
    }
    RngArg rng;
    rng.seed_lo = (uint32_t)seed; rng.seed_hi = (uint32_t)(seed >> 32); rng.state = rng_state; rng.on = 1;
    rng.perturb = perturb ? 1 : 0;
    rng.ray_offset = (uint32_t)ray_offset;
    return sample_z_all_impl("eslam_sample_z_all_rng", planes, dec, bound6_host, rays_o, rays_d, gt_depth, R, n_strat, n_imp,
                             truncation, t_free, t_surf, nullptr, nullptr, nullptr, rng, z_vals, stream);
}

static int fill_set_sizes_args(const char* who, const float* gt_depth, const uint8_t* ray_mask, int R, int n_strat, int n_imp,
                               double truncation, const float* t_free, const float* t_surf, const float* t_rand, int perturb,
                               uint64_t seed, const uint32_t* rng_state, uint32_t* scratch, float* acc_out, SetSizesArgs* a) {
    if (R <= 0 || n_strat < 1 || n_imp < 0 || n_strat + n_imp > ESLAM_MAX_SAMPLES) {
        eslam_set_error("%s: R=%d n_strat=%d n_imp=%d unsupported", who, R, n_strat, n_imp);
        return 1;
    }
    if (!gt_depth || !t_free || (n_imp > 0 && !t_surf) || !scratch || !acc_out) {
        eslam_set_error("%s: null argument (set sizes)", who);
        return 1;
    }
    if ((int64_t)R * (n_strat + n_imp) >= ((int64_t)1 << 32)) {
        eslam_set_error("%s: batch too large", who);
        return 1;
    }
    RngArg rng = {};
    if (!t_rand) {
        rng.seed_lo = (uint32_t)seed; rng.seed_hi = (uint32_t)(seed >> 32); rng.state = rng_state; rng.on = 1;
        rng.perturb = perturb ? 1 : 0;
    }
    a->gt_depth = gt_depth; a->ray_mask = ray_mask; a->R = R; a->n_strat = n_strat; a->n_imp = n_imp;
    a->c15 = (float)(1.5 * truncation); a->c3 = (float)(3.0 * truncation);
    a->t_free = t_free; a->t_surf = t_surf; a->t_rand = t_rand; a->rng = rng; a->tr = make_trunc(truncation);
    a->scratch = scratch; a->acc_out = acc_out;
    return 0;
}

extern "C" int eslam_loss_set_sizes(const float* gt_depth, const uint8_t* ray_mask, int R, int n_strat, int n_imp,
                                    double truncation, const float* t_free, const float* t_surf, const float* t_rand, int perturb,
                                    uint64_t seed, const uint32_t* rng_state, uint32_t* scratch, float* acc_out,
                                    eslam_stream_t stream) {
    SetSizesArgs a;
    if (int rc = fill_set_sizes_args("eslam_loss_set_sizes", gt_depth, ray_mask, R, n_strat, n_imp, truncation, t_free, t_surf, t_rand,
                                     perturb, seed, rng_state, scratch, acc_out, &a))
        return rc;
    const int nwg = (R + 3) / 4 < 1024 ? (R + 3) / 4 : 1024;
    hipLaunchKernelGGL(loss_set_sizes_kernel, dim3(nwg), dim3(256), 0, (hipStream_t)stream, a);
    return eslam_check_launch("loss_set_sizes_kernel");
}

int eslam_shard_mark_args(const char* who, const eslam_plane_t* planes, const float* bound6_host, const float* rays_o,
                          const float* rays_d, const float* gt_depth, int R, double truncation, const int64_t* block_base_host,
                          int64_t n_blocks, uint8_t* touched, MarkArgs* m);

extern "C" int eslam_shard_prologue(float* clear_flat, const int32_t* clear_idx, const int32_t* clear_meta, int64_t clear_capacity,
                                    float* clear_tail, int64_t clear_n_tail, const float* rays_o, const float* rays_d,
                                    const float* gt_depth, const uint8_t* ray_mask, int R, int n_strat, int n_imp,
                                    double truncation, const float* t_free, const float* t_surf, int perturb, uint64_t seed,
                                    const uint32_t* rng_state, uint32_t* sizes_scratch, float* acc_out,
                                    const eslam_plane_t* mark_planes, const float* bound6_host, const int64_t* block_base_host,
                                    int64_t n_blocks, uint8_t* touched, eslam_stream_t stream) {
    SetSizesArgs sa;
    if (int rc = fill_set_sizes_args("eslam_shard_prologue", gt_depth, ray_mask, R, n_strat, n_imp, truncation, t_free, t_surf,
                                     nullptr, perturb, seed, rng_state, sizes_scratch, acc_out, &sa))
        return rc;
    ClearArgs ca = {};
    int n_clear = 0;
    if (clear_flat) {
        if (!clear_idx || !clear_meta || clear_capacity <= 0 || clear_n_tail < 0 || (clear_n_tail > 0 && !clear_tail) ||
            ((uintptr_t)clear_flat & 15) || clear_n_tail >= ((int64_t)1 << 30)) {
            eslam_set_error("eslam_shard_prologue: bad clear arguments");
            return 1;
        }
        ca.flat = clear_flat; ca.idx = clear_idx; ca.meta = clear_meta; ca.tail = clear_tail; ca.n_tail = (int)clear_n_tail;
        const int64_t want = (clear_capacity + 31) / 32;
        n_clear = (int)(want < 512 ? want : 512);
    }
    MarkArgs ma = {};
    int n_mark = 0;
    if (mark_planes) {
        if (int rc = eslam_shard_mark_args("eslam_shard_prologue", mark_planes, bound6_host, rays_o, rays_d, gt_depth, R, truncation,
                                           block_base_host, n_blocks, touched, &ma))
            return rc;
        n_mark = (R + 3) / 4 < 2048 ? (R + 3) / 4 : 2048;
    }
    const int n_sizes = (R + 3) / 4 < 1024 ? (R + 3) / 4 : 1024;
    hipLaunchKernelGGL(shard_prologue_kernel, dim3(n_clear + n_sizes + n_mark), dim3(256), 0, (hipStream_t)stream, ca, n_clear, sa,
                       n_sizes, ma, n_mark);
    return eslam_check_launch("shard_prologue_kernel");
}
