"""ctypes binding of the C-ABI library (include/eslam_hip.h) - the only door to the HIP kernels.

The library is loaded lazily, once per process (the reference pickles its Renderer into two spawned
processes, reference src/ESLAM.py:246-260, so nothing process-specific may live on the Python objects).
There is NO fallback: if the shared object is missing or a call fails, a RuntimeError is raised.
"""
import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# ESLAM_HIP_LIB: another build of the SAME library (A/B of compile-time kernel variants: `make variant`, tools/ab_inproc.py); never a fallback
LIB_PATH = os.environ.get("ESLAM_HIP_LIB") or os.path.join(_HERE, "lib", "libeslam_hip.so")

ABI_VERSION = 5              # ESLAM_ABI_VERSION
RAY_ORDERS = 3               # ESLAM_RAY_ORDERS: eslam_ray_order writes one order per plane orientation


def ray_order_words(R):
    """ESLAM_RAY_ORDER_WORDS(R): int32 words of eslam_ray_order's output (the three orders + the fan's extent per plane)."""
    return RAY_ORDERS * int(R) + 4
N_DEC_PARAMS = 2692
N_PLANES = 12


class PlaneDesc(ctypes.Structure):
    _fields_ = [("data", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("h", ctypes.c_int32), ("w", ctypes.c_int32),
                ("stride_c", ctypes.c_int64), ("stride_y", ctypes.c_int64), ("stride_x", ctypes.c_int64),
                ("data_f16", ctypes.c_void_p)]


class DecodersDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in
                ("w1", "b1", "w2", "b2", "w3", "b3", "cw1", "cb1", "cw2", "cb2", "cw3", "cb3", "beta")]


PlaneArray = PlaneDesc * N_PLANES
Bound6 = ctypes.c_float * 6

_vp, _i, _i64, _f, _d = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_double
_PP = ctypes.POINTER(PlaneDesc)
_DP = ctypes.POINTER(DecodersDesc)
_BP = ctypes.POINTER(ctypes.c_float)

# name -> (restype, argtypes); must list every symbol declared in include/eslam_hip.h
SIGNATURES = {
    "eslam_last_error": (ctypes.c_char_p, []),
    "eslam_abi_version": (_i, []),
    "eslam_sample_rays": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _i, _f, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "eslam_sample_rays_bwd": (_i, [_vp, _i, _i, _i, _i, _i, _f, _f, _f, _f, _vp, _vp, _vp, _vp]),
    "eslam_image_rays": (_i, [_i, _i, _f, _f, _f, _f, _vp, _vp, _vp, _vp]),
    "eslam_aabb_exit": (_i, [_vp, _vp, _i, _BP, _vp, _vp]),
    "eslam_sample_z": (_i, [_vp, _i, _i, _i, _d, _vp, _vp, _vp, _vp, _vp]),
    "eslam_importance_z": (_i, [_PP, _DP, _BP, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "eslam_sample_z_all": (_i, [_PP, _DP, _BP, _vp, _vp, _vp, _i, _i, _i, _d, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "eslam_sample_z_all_rng": (_i, [_PP, _DP, _BP, _vp, _vp, _vp, _i, _i, _i, _d, _vp, _vp, _i, ctypes.c_uint64, _vp, _i64, _vp, _vp]),
    "eslam_loss_set_sizes": (_i, [_vp, _vp, _i, _i, _i, _d, _vp, _vp, _vp, _i, ctypes.c_uint64, _vp, _vp, _vp, _vp]),
    "eslam_mark_rays": (_i, [_PP, _BP, _vp, _vp, _vp, _i, _d, _vp, _i64, _vp, _vp]),
    "eslam_blocks_compact_scratch_words": (_i64, [_i64]),
    "eslam_blocks_compact": (_i, [_vp, _i64, _vp, _vp, _vp, _vp, _i, _vp]),
    "eslam_host_meta_alloc": (_i, [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p)]),
    "eslam_host_meta_free": (_i, [_vp]),
    "eslam_shard_prologue": (_i, [_vp, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp, _i, _i, _i, _d, _vp, _vp, _i, ctypes.c_uint64,
                                  _vp, _vp, _vp, _PP, _BP, _vp, _i64, _vp, _vp]),
    "eslam_blocks_pack_dev": (_i, [_vp, _vp, _vp, _i64, _vp, _i64, _i64, _vp, _vp, _vp]),
    "eslam_blocks_unpack_dev": (_i, [_vp, _vp, _vp, _i64, _vp, _i64, _i64, _vp, _vp]),
    "eslam_blocks_zero_dev": (_i, [_vp, _vp, _vp, _i64, _vp, _i64, _vp]),
    "eslam_render_fwd": (_i, [_PP, _DP, _BP, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "eslam_render_fwd_loss": (_i, [_PP, _DP, _BP, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _d, _BP, _vp,
                                   _vp, _vp, _vp, _vp, _vp]),
    "eslam_planes_to_half": (_i, [_PP, _vp]),
    "eslam_planes_relayout": (_i, [_PP, _PP, _i, _vp]),
    "eslam_render_fwd_lowp": (_i, [_PP, _DP, _BP, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "eslam_ray_order": (_i, [_vp, _vp, _i, _vp, _vp]),
    "eslam_stream_wait": (_i, [_vp, _vp]),
    "eslam_zero_async": (_i, [_vp, _i64, _vp]),
    "eslam_bwd_workspace_bytes": (_i64, [_i64]),
    "eslam_render_bwd": (_i, [_PP, _DP, _BP, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                              _vp, _vp, _vp]),
    "eslam_render_bwd_loss": (_i, [_PP, _DP, _BP, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _d, _BP, _vp, _vp,
                                   _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "eslam_decode_fwd": (_i, [_PP, _DP, _BP, _vp, _i64, _i, _vp, _vp, _vp]),
    "eslam_decode_bwd": (_i, [_PP, _DP, _BP, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "eslam_mapping_loss": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _d, _BP, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "eslam_loss_reduce": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _d, _vp, _vp, _vp]),
    "eslam_profile_enable": (_i, [_i]),
    "eslam_profile_read": (_i, [_BP]),
    "eslam_profile_name": (ctypes.c_char_p, [_i]),
    "eslam_loss_grad": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _d, _BP, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "eslam_deterministic": (_i, []),
    "eslam_loss_scratch_floats": (_i64, [_i64]),
    "eslam_loss_scratch_reset": (_i, [_vp, _i64, _vp]),
    "eslam_loss_value": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _d, _BP, _vp, _vp, _vp, _vp, _vp]),
    "eslam_adam_step": (_i, [_vp, _i, _i, _vp, _d, _d, _d, _i, _vp]),
    "eslam_prefilter": (_i, [_vp, _vp, _vp, _i, _BP, _i, _vp, _vp]),
    "eslam_pose_to_c2w": (_i, [_vp, _i, _vp, _vp]),
    "eslam_pose_to_c2w_bwd": (_i, [_vp, _vp, _i, _vp, _vp]),
    "eslam_tracking_mask": (_i, [_vp, _vp, _vp, _i, _f, _vp, _vp]),
    "eslam_keep_best": (_i, [_vp, _vp, _i, _vp, _vp, _vp]),
    "eslam_keyframe_overlap": (_i, [_vp, _vp, _vp, _i, _i, _vp, _i, _i, _i, _f, _f, _f, _f, _i, _vp, _vp]),
}

PROF_KERNELS = 12           # ESLAM_PROF_KERNELS


class AdamTensor(ctypes.Structure):   # eslam_adam_tensor_t
    _fields_ = [("param", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("exp_avg", ctypes.c_void_p),
                ("exp_avg_sq", ctypes.c_void_p), ("n", ctypes.c_int64), ("lr", ctypes.c_double)]


_lib = None
_lock = threading.Lock()


def load_library(path=None):
    """dlopen the library and attach prototypes.  Does not need a GPU (used by the CPU test suite)."""
    global _lib
    with _lock:
        if _lib is not None and path is None:
            return _lib
        p = path or LIB_PATH
        if not os.path.exists(p):
            raise RuntimeError(
                f"HIP extension not built: {p} is missing. Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C myslam_amd/csrc`). There is no CPU fallback for the render path.")
        lib = ctypes.CDLL(p)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if the symbol is missing: fail loudly
            fn.restype = res
            fn.argtypes = args
        if lib.eslam_abi_version() != ABI_VERSION:
            raise RuntimeError(f"ABI mismatch: library reports version {lib.eslam_abi_version()}, binding expects {ABI_VERSION}")
        if path is None:
            _lib = lib
        return lib


def lib():
    return _lib if _lib is not None else load_library()


def check(rc, what):
    if rc != 0:
        msg = lib().eslam_last_error()
        raise RuntimeError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_handle(device):
    """hipStream_t of the caller's current stream on `device` (the raw getter skips building a torch.cuda.Stream object:
    7 us -> 0.3 us, and a step asks a dozen times)."""
    if _raw_stream is not None and device.index is not None:
        return ctypes.c_void_p(_raw_stream(device.index))
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


_TORCH_STREAM_WAIT = os.environ.get("ESLAM_TORCH_STREAM_WAIT", "0") == "1"


def stream_wait(device, waiter, signaler):
    """waiter / signaler: torch.cuda.Stream, or None for the caller's current stream on `device`."""
    if _TORCH_STREAM_WAIT:          # A/B switch (ESLAM_TORCH_STREAM_WAIT=1): torch's own Stream.wait_stream
        (torch.cuda.current_stream(device) if waiter is None else waiter).wait_stream(
            torch.cuda.current_stream(device) if signaler is None else signaler)
        return
    w = stream_handle(device) if waiter is None else ctypes.c_void_p(waiter.cuda_stream)
    s = stream_handle(device) if signaler is None else ctypes.c_void_p(signaler.cuda_stream)
    with on_device(device):
        check(lib().eslam_stream_wait(w, s), "eslam_stream_wait")


class _NullCtx:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_NULL = _NullCtx()


def on_device(device):
    """Context that makes `device` current for a launch; free when it already is (torch.cuda.device costs ~15 us)."""
    return _NULL if torch.cuda.current_device() == device.index else torch.cuda.device(device)


def require_gpu_f32(name, t):
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a tensor on the GPU (got device {t.device}); the HIP render path has "
                           "no CPU fallback")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name}: expected float32, got {t.dtype}")


_plane_cache = {}
_plane_last = None            # (the 12 tensor objects, their data pointers, cache entry) of the last call


def make_planes(all_planes, grads=None, dtype=torch.float32, half=None, remember=True):
    """all_planes: the reference's 6-tuple of [coarse, fine] lists, or the same 12 tensors as a flat list / tuple
    -> (PlaneArray, keepalive list).  Pass the tensors themselves (Parameters included), not detached copies: the sampler, the
    forward and the backward of one iteration hand over the same 12 objects, and the last call's descriptors are reused when
    the objects and their data pointers are unchanged (~3 us instead of ~10 us of attribute reads per call, five calls a step).

    Descriptors are cached per (data pointers, shapes, row strides): the mapper keeps the same 12 storages for a whole
    run (it re-wraps them as new nn.Parameters every frame, src/Mapper.py:254-266, without moving them), and building
    12 ctypes structs from tensor attributes costs ~25 us of host time per call otherwise."""
    global _plane_last
    if len(all_planes) == N_PLANES and torch.is_tensor(all_planes[0]):
        flat = all_planes
    else:
        flat = [p for grp in all_planes for p in grp]
        if len(flat) != N_PLANES or any(len(grp) != 2 for grp in all_planes):
            raise RuntimeError("all_planes must be 6 groups of [coarse, fine] planes")
    last = _plane_last
    hit = None
    if last is not None and dtype == torch.float32:
        objs, ptrs, entry = last
        same = True
        for a, b, q in zip(flat, objs, ptrs):
            if a is not b or a.data_ptr() != q:
                same = False
                break
        if same:
            hit = entry
    # the key carries everything the validation below looks at: an address reused by a different tensor (other dtype, shape,
    # channel stride, device) misses the cache and is validated afresh
    if hit is None:
        key = (dtype,) + tuple((p.data_ptr(), p.dtype, p.device.index, tuple(p.shape), p.stride()) for p in flat)
        hit = _plane_cache.get(key)
    else:
        key = None
    if hit is None:
        arr = PlaneArray()
        for k, p in enumerate(flat):
            g, lvl = divmod(k, 2)
            if dtype == torch.float32:
                require_gpu_f32(f"plane[{g}][{lvl}]", p)
            elif not p.is_cuda or p.dtype != dtype:
                raise RuntimeError(f"plane[{g}][{lvl}]: expected a {dtype} tensor on the GPU, got {p.dtype} on {p.device}")
            if p.dim() != 4 or p.shape[0] != 1 or p.shape[1] != 32:
                raise RuntimeError(f"plane[{g}][{lvl}]: expected shape [1,32,h,w], got {tuple(p.shape)}")
            d = arr[k]
            d.data = p.data_ptr()
            d.h, d.w = int(p.shape[2]), int(p.shape[3])
            d.stride_c, d.stride_y, d.stride_x = (int(v) for v in p.stride()[1:])
            d.grad = None
        if len(_plane_cache) > 64:
            _plane_cache.clear()
        hit = _plane_cache[key] = (bytes(arr), [tuple(p.shape) for p in flat], [p.stride() for p in flat])
    if key is not None and dtype == torch.float32 and remember:
        # (strong references to the 12 tensor objects: while they are held here no other object can take their identity.
        # remember=False for short-lived tensors such as gradient buffers on their way to autograd: a reference kept here would
        # make AccumulateGrad copy them instead of adopting them)
        _plane_last = (tuple(flat), tuple(p.data_ptr() for p in flat), hit)
    arr = PlaneArray.from_buffer_copy(hit[0])
    if half is not None:
        # mixed precision: half copies of the planes, channels-last like their float32 masters (eslam_plane_t.data_f16)
        for k, (hp, p) in enumerate(zip(half, flat)):
            if (hp.dtype != torch.float16 or not hp.is_cuda or hp.shape != p.shape or hp.stride() != p.stride()
                    or not p.is_contiguous(memory_format=torch.channels_last)):
                raise RuntimeError("mixed precision: every plane needs a float16 channels_last copy of its own shape, and the "
                                   "float32 planes must be channels_last too")
            arr[k].data_f16 = hp.data_ptr()
    if grads is not None:
        for k, gr in enumerate(grads):
            if gr.shape != hit[1][k] or gr.stride() != hit[2][k]:
                raise RuntimeError("plane gradient buffer must have the plane's shape and strides")
            arr[k].grad = gr.data_ptr()
    return arr, flat


DEC_FIELDS = (("w1", "linears.0.weight", (16, 64)), ("b1", "linears.0.bias", (16,)),
              ("w2", "linears.1.weight", (16, 16)), ("b2", "linears.1.bias", (16,)),
              ("w3", "output_linear.weight", (1, 16)), ("b3", "output_linear.bias", (1,)),
              ("cw1", "c_linears.0.weight", (16, 64)), ("cb1", "c_linears.0.bias", (16,)),
              ("cw2", "c_linears.1.weight", (16, 16)), ("cb2", "c_linears.1.bias", (16,)),
              ("cw3", "c_output_linear.weight", (3, 16)), ("cb3", "c_output_linear.bias", (3,)))


_dec_cache = {}
_dec_last = None


def make_decoders(params, beta):
    """params: 12 tensors in DEC_FIELDS order; beta: device tensor [1].  Cached per set of data pointers; the last call's
    descriptor is reused when the same tensor objects with the same data pointers come back (as make_planes)."""
    global _dec_last
    last = _dec_last
    if last is not None:
        objs, ptrs, d = last
        if beta is objs[12] and beta.data_ptr() == ptrs[12]:
            same = True
            for a, b, q in zip(params, objs, ptrs):
                if a is not b or a.data_ptr() != q:
                    same = False
                    break
            if same:
                return d, (params, beta)
    key = tuple((t.data_ptr(), t.dtype, t.device.index, tuple(t.shape), t.stride()) for t in params) + \
        ((beta.data_ptr(), beta.dtype, beta.device.index),)
    d = _dec_cache.get(key)
    if d is None:
        d = DecodersDesc()
        for (field, name, shape), t in zip(DEC_FIELDS, params):
            require_gpu_f32(name, t)
            if tuple(t.shape) != shape:
                raise RuntimeError(f"{name}: expected shape {shape}, got {tuple(t.shape)} (c_dim=32, hidden=16, 2 blocks)")
            if not t.is_contiguous():
                raise RuntimeError(f"{name}: decoder parameters must be contiguous")
            setattr(d, field, t.data_ptr())
        require_gpu_f32("beta", beta)
        d.beta = beta.data_ptr()
        if len(_dec_cache) > 64:
            _dec_cache.clear()
        _dec_cache[key] = d
    _dec_last = (tuple(params) + (beta,), tuple(t.data_ptr() for t in params) + (beta.data_ptr(),), d)
    return d, (params, beta)


def make_bound(bound_host):
    """bound_host: 6 Python floats (x0,x1,y0,y1,z0,z1)."""
    return Bound6(*[float(v) for v in bound_host])
