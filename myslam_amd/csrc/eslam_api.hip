// Error plumbing and argument validation shared by the C-ABI entry points.
#include <stdarg.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <atomic>
#include <mutex>
#include "eslam_common.h"

static thread_local char g_err[512] = "";

void eslam_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int eslam_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        eslam_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return 2;
    }
    return 0;
}

extern "C" const char* eslam_last_error(void) { return g_err; }
extern "C" int eslam_abi_version(void) { return ESLAM_ABI_VERSION; }

extern "C" int eslam_deterministic(void) {
    static const int det = [] { const char* v = getenv("ESLAM_DETERMINISTIC"); return (v && v[0] && v[0] != '0') ? 1 : 0; }();
    return det;
}

// floats of loss scratch a batch of n_rays needs: the ticket + one line per accumulator, or (deterministic mode) the ticket
// + one 16-float slot per workgroup of the kernel that forms the sums (4 rays per workgroup in eslam_render_fwd_loss)
extern "C" int64_t eslam_loss_scratch_floats(int64_t n_rays) {
    if (n_rays < 0) return -1;
    if (!eslam_deterministic()) return ESLAM_LOSS_SCRATCH;
    const int64_t nwg = ((n_rays + 3) / 4 + 7) / 8 * 8;
    const int64_t need = 32 + 16 * nwg;
    return need > ESLAM_LOSS_SCRATCH ? need : ESLAM_LOSS_SCRATCH;
}

// back to the state of a freshly zeroed scratch (after an aborted graph, a failed launch, ...): first `floats` floats
extern "C" int eslam_loss_scratch_reset(float* scratch, int64_t floats, eslam_stream_t stream) {
    if (!scratch || floats < ESLAM_LOSS_SCRATCH) {
        eslam_set_error("eslam_loss_scratch_reset: null scratch or fewer than ESLAM_LOSS_SCRATCH floats");
        return 1;
    }
    if (hipMemsetAsync(scratch, 0, (size_t)floats * sizeof(float), (hipStream_t)stream) != hipSuccess) {
        eslam_set_error("eslam_loss_scratch_reset: memset failed");
        return 2;
    }
    return 0;
}

// 1: all 12 planes carry a half copy (the mixed-precision path), 0: none does, -1 (error set): some do, or a copy's layout
// cannot be the 64-byte-texel one the half gather assumes
int eslam_planes_lowp(const eslam_plane_t* planes) {
    int n = 0;
    for (int i = 0; i < ESLAM_N_PLANES; ++i) n += planes[i].data_f16 != nullptr;
    if (n == 0) return 0;
    if (n != ESLAM_N_PLANES) {
        eslam_set_error("mixed precision: %d of the 12 planes carry a half copy (data_f16); give all or none", n);
        return -1;
    }
    for (int i = 0; i < ESLAM_N_PLANES; ++i) {
        const eslam_plane_t& p = planes[i];
        if (p.stride_c != 1 || p.stride_x != ESLAM_C_DIM || p.stride_y != (int64_t)ESLAM_C_DIM * p.w || ((uintptr_t)p.data_f16 & 15)) {
            eslam_set_error("mixed precision: plane %d must be channels-last (its half copy has 64-byte texels with the same "
                            "element strides)", i);
            return -1;
        }
    }
    return 1;
}

// A plane is "channels-last" when one texel's 32 channels are contiguous and texels along x are adjacent.
bool eslam_planes_channels_last(const eslam_plane_t* planes, int first, int count) {
    for (int i = first; i < first + count; ++i) {
        const eslam_plane_t& p = planes[i];
        if (p.stride_c != 1 || p.stride_x != ESLAM_C_DIM) return false;
        if (((uintptr_t)p.data & 15) != 0 || (p.stride_y & 3) != 0) return false;
        if (p.grad && ((uintptr_t)p.grad & 15) != 0) return false;
    }
    return true;
}

// Shapes must match what the kernels and their launch grids assume before anything is launched.
int eslam_validate_planes(const eslam_plane_t* planes, int first, int count) {
    for (int i = first; i < first + count; ++i) {
        const eslam_plane_t& p = planes[i];
        if (!p.data) {
            eslam_set_error("plane %d: null data pointer", i);
            return 1;
        }
        if (p.h < 2 || p.w < 2) {
            eslam_set_error("plane %d: needs h,w >= 2 (got %d x %d)", i, p.h, p.w);
            return 1;
        }
        if (p.stride_c <= 0 || p.stride_y <= 0 || p.stride_x <= 0) {
            eslam_set_error("plane %d: strides must be positive", i);
            return 1;
        }
        const int64_t extent = (ESLAM_C_DIM - 1) * p.stride_c + (int64_t)(p.h - 1) * p.stride_y +
                               (int64_t)(p.w - 1) * p.stride_x + 1;
        if (extent >= ((int64_t)1 << 31)) {
            eslam_set_error("plane %d: extent %lld elements exceeds 32-bit offsets", i, (long long)extent);
            return 1;
        }
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// per-kernel timing with HIP events on the launch stream
// ---------------------------------------------------------------------------------------------------------
static bool g_prof_on = false;
static hipEvent_t g_ev[ESLAM_PROF_KERNELS][2];
static bool g_ev_made = false;
static bool g_ev_used[ESLAM_PROF_KERNELS];
static const char* const g_prof_names[ESLAM_PROF_KERNELS] = {
    "render_fwd_kernel", "composite_bwd_kernel", "mlp_bwd_kernel", "dec_grad_reduce_kernel", "scatter_sort_kernel",
    "coord_bwd_kernel", "loss_reduce_kernel+loss_grad_kernel", "sample_z_kernel", "importance_z_kernel",
    "decode_fwd_kernel", "adam_step_kernel", "keyframe_overlap_kernel"};

void eslam_prof_begin(int id, hipStream_t st) {
    if (!g_prof_on) return;
    (void)hipEventRecord(g_ev[id][0], st);
}

void eslam_prof_end(int id, hipStream_t st) {
    if (!g_prof_on) return;
    (void)hipEventRecord(g_ev[id][1], st);
    g_ev_used[id] = true;
}

extern "C" int eslam_profile_enable(int on) {
    if (on && !g_ev_made) {
        for (int i = 0; i < ESLAM_PROF_KERNELS; ++i)
            for (int j = 0; j < 2; ++j)
                if (hipEventCreate(&g_ev[i][j]) != hipSuccess) {
                    eslam_set_error("eslam_profile_enable: hipEventCreate failed");
                    return 2;
                }
        g_ev_made = true;
    }
    for (int i = 0; i < ESLAM_PROF_KERNELS; ++i) g_ev_used[i] = false;
    g_prof_on = on != 0;
    return 0;
}

extern "C" int eslam_profile_read(float* ms_out) {
    if (!ms_out || !g_ev_made) {
        eslam_set_error("eslam_profile_read: profiling was never enabled");
        return 1;
    }
    for (int i = 0; i < ESLAM_PROF_KERNELS; ++i) {
        ms_out[i] = -1.0f;
        if (!g_ev_used[i]) continue;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, g_ev[i][0], g_ev[i][1]) == hipSuccess) ms_out[i] = ms;
        g_ev_used[i] = false;
    }
    return 0;
}

extern "C" const char* eslam_profile_name(int kernel_id) {
    return (kernel_id >= 0 && kernel_id < ESLAM_PROF_KERNELS) ? g_prof_names[kernel_id] : "";
}


// waiter waits for everything enqueued on signaler so far (fork / join of the ray-order side stream): one event record and
// one stream wait, from a small per-device ring of timing-disabled events.  Capturable: inside a stream capture the pair
// becomes a dependency edge of the graph, exactly as torch's Stream.wait_stream does - without its ~13 us of Python.
extern "C" int eslam_stream_wait(eslam_stream_t waiter, eslam_stream_t signaler) {
    // Called from the Python thread AND from autograd's engine thread (ctypes drops the GIL around the call): the ring
    // index is atomic and each device's events are created exactly once.
    constexpr int RING = 64, MAXDEV = 64;
    static hipEvent_t ring[MAXDEV][RING];
    static std::atomic<unsigned> next[MAXDEV];
    static std::once_flag made[MAXDEV];
    static bool ok[MAXDEV];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) {
        eslam_set_error("eslam_stream_wait: no usable current device");
        return 2;
    }
    if (dev >= MAXDEV) {          // no ring for this device index: an event of its own, released once the wait is enqueued
        hipEvent_t ev;
        if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) {
            eslam_set_error("eslam_stream_wait: hipEventCreateWithFlags failed");
            return 2;
        }
        const bool fine = hipEventRecord(ev, (hipStream_t)signaler) == hipSuccess &&
                          hipStreamWaitEvent((hipStream_t)waiter, ev, 0) == hipSuccess;
        (void)hipEventDestroy(ev);       // deferred by the runtime until the recorded work has completed
        if (!fine) {
            eslam_set_error("eslam_stream_wait: %s", hipGetErrorString(hipGetLastError()));
            return 2;
        }
        return 0;
    }
    std::call_once(made[dev], [dev] {
        bool good = true;
        for (int i = 0; i < RING; ++i) good = good && hipEventCreateWithFlags(&ring[dev][i], hipEventDisableTiming) == hipSuccess;
        ok[dev] = good;
    });
    if (!ok[dev]) {
        eslam_set_error("eslam_stream_wait: hipEventCreateWithFlags failed");
        return 2;
    }
    hipEvent_t ev = ring[dev][next[dev].fetch_add(1u, std::memory_order_relaxed) % RING];
    if (hipEventRecord(ev, (hipStream_t)signaler) != hipSuccess || hipStreamWaitEvent((hipStream_t)waiter, ev, 0) != hipSuccess) {
        eslam_set_error("eslam_stream_wait: %s", hipGetErrorString(hipGetLastError()));
        return 2;
    }
    return 0;
}


// Zero `bytes` bytes at `ptr` on `stream` (hipMemsetAsync): the Python layer clears the plane-gradient buffer of an
// iteration at the head of the ray-order side stream, beside the samplers (latency-bound kernels that move almost no
// bytes), instead of in front of the backward pass.
extern "C" int eslam_zero_async(void* ptr, int64_t bytes, eslam_stream_t stream) {
    if (bytes <= 0) return 0;
    if (!ptr) {
        eslam_set_error("eslam_zero_async: null pointer");
        return 1;
    }
    if (hipMemsetAsync(ptr, 0, (size_t)bytes, (hipStream_t)stream) != hipSuccess) {
        eslam_set_error("eslam_zero_async: %s", hipGetErrorString(hipGetLastError()));
        return 2;
    }
    return 0;
}
