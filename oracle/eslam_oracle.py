"""ORACLE - test infrastructure, not product code.

A CPU restatement, in plain PyTorch tensor ops, of the reference's per-iteration rendering hot path
(SURVEY.md section 8(a), rows a1-a10).  It exists to *check* the HIP path:

  * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module;
  * nothing under myslam_amd/ imports it, and the product path has no CPU fallback.

Pinning: the reference has no tests or golden vectors of its own, so this
restatement is pinned against outputs of the reference itself, produced in the build container by
tests/golden/make_golden.py (which imports /root/reference) and committed as tests/golden/*.npz.
tests/test_oracle_golden.py replays every fixture through this file.

Every function cites the reference lines it follows.  The arithmetic is written from the maths of
those lines, not transcribed: bilinear sampling is explicit index arithmetic rather than a call to
F.grid_sample, the sort of the two arithmetic sample sequences is kept as a sort only because its
*values* are what matter, etc.  All functions are dtype-generic so tests can run them in float64 for
tight gradient checks of the float32 kernels.

Gradients come from autograd over these ops, which makes them independent of the hand-derived
backward kernels they are compared with.
"""
import torch

DECODER_KEYS = (
    "linears.0.weight", "linears.0.bias", "linears.1.weight", "linears.1.bias",
    "output_linear.weight", "output_linear.bias",
    "c_linears.0.weight", "c_linears.0.bias", "c_linears.1.weight", "c_linears.1.bias",
    "c_output_linear.weight", "c_output_linear.bias",
)


# ----------------------------------------------------------------------------------------------
# a1: pixel selection and back-projection
# ----------------------------------------------------------------------------------------------
def rays_from_pixels(indices, H0, H1, W0, W1, fx, fy, cx, cy, c2ws, depths, colors):
    """Reference src/common.py:87-153 with the random draw lifted out.

    indices [b*n] int64 in [0, (H1-H0)*(W1-W0)): flat pixel index inside the crop window, the value
    common.py:108 draws with torch.randint; row-major over (v, u).  The reference reshapes the one
    index vector to [b, n] (common.py:113), i.e. image k uses indices[k*n:(k+1)*n].
    Returns rays_o [b*n,3], rays_d [b*n,3], depth [b*n], color [b*n,3].
    """
    b = c2ws.shape[0]
    n = indices.numel() // b
    ww = W1 - W0
    dt = c2ws.dtype
    # common.py:133-136: linspace(W0, W1-1, W1-W0) meshgrid, transposed -> u varies fastest
    u = (indices % ww).to(dt) + W0
    v = (indices // ww).to(dt) + H0
    idx = indices.reshape(b, n)
    d_img = depths[:, H0:H1, W0:W1].reshape(b, -1)
    c_img = colors[:, H0:H1, W0:W1].reshape(b, -1, 3)
    depth = torch.gather(d_img, 1, idx)                                   # common.py:120
    color = torch.gather(c_img, 1, idx[..., None].expand(-1, -1, 3))      # common.py:121
    # common.py:92: camera-frame direction, OpenGL convention (y up, looking down -z)
    dirs = torch.stack([(u - cx) / fx, -(v - cy) / fy, -torch.ones_like(u)], -1).reshape(b, n, 3)
    rot = c2ws[:, :3, :3]
    rays_d = torch.einsum("bnk,bjk->bnj", dirs, rot)                      # common.py:96  (R @ dir)
    rays_o = c2ws[:, None, :3, 3].expand(b, n, 3)                         # common.py:97
    return rays_o.reshape(-1, 3), rays_d.reshape(-1, 3), depth.reshape(-1), color.reshape(-1, 3)


def rays_full_image(H, W, fx, fy, cx, cy, c2w):
    """Reference src/common.py:183-201 (get_rays): all H*W pixels, row-major."""
    dt = c2w.dtype
    u = torch.arange(W, dtype=dt)[None, :].expand(H, W)
    v = torch.arange(H, dtype=dt)[:, None].expand(H, W)
    dirs = torch.stack([(u - cx) / fx, -(v - cy) / fy, -torch.ones_like(u)], -1)
    rays_d = torch.einsum("hwk,jk->hwj", dirs, c2w[:3, :3])
    rays_o = c2w[:3, 3].expand(H, W, 3)
    return rays_o, rays_d


# ----------------------------------------------------------------------------------------------
# a2: AABB exit distance
# ----------------------------------------------------------------------------------------------
def aabb_exit(rays_o, rays_d, bound):
    """Reference src/Mapper.py:322-328 / Tracker.py:175-181 / Renderer.py:114-115.

    For each axis take the larger of the two slab distances, then the smallest over axes.
    """
    t = (bound[None, :, :] - rays_o[:, :, None]) / rays_d[:, :, None]      # [N,3,2]
    return t.max(dim=2).values.min(dim=1).values


# ----------------------------------------------------------------------------------------------
# a3: depth-guided sampling
# ----------------------------------------------------------------------------------------------
def jitter(z, t_rand):
    """Reference src/utils/Renderer.py:46-61: stratified jitter between neighbour midpoints."""
    mids = 0.5 * (z[..., 1:] + z[..., :-1])
    upper = torch.cat([mids, z[..., -1:]], -1)
    lower = torch.cat([z[..., :1], mids], -1)
    return lower + (upper - lower) * t_rand


def depth_guided_z(gt_depth, n_stratified, n_importance, truncation, t_rand=None):
    """Reference src/utils/Renderer.py:87-104 for rays with gt_depth > 0.

    n_stratified samples spread over [0, 1.2 d] plus n_importance samples over [d-1.5 tau, d+1.5 tau],
    merged in ascending order, optionally jittered.  gt_depth [R] -> z [R, S].
    """
    dt = gt_depth.dtype
    d = gt_depth.reshape(-1, 1)
    t_free = torch.linspace(0.0, 1.0, n_stratified, dtype=dt, device=gt_depth.device)
    t_surf = torch.linspace(0.0, 1.0, n_importance, dtype=dt, device=gt_depth.device)
    z_surf = d - (1.5 * truncation) + (3 * truncation * t_surf)           # Renderer.py:97
    z_free = 0.0 + 1.2 * d * t_free                                       # Renderer.py:100
    z = torch.sort(torch.cat([z_free, z_surf], -1), -1).values            # Renderer.py:102
    if t_rand is not None:
        z = jitter(z, t_rand)
    return z


# ----------------------------------------------------------------------------------------------
# a5-a7: normalise, tri-plane lookup, decoders
# ----------------------------------------------------------------------------------------------
def normalize_points(p, bound):
    """Reference src/common.py:204-218: affine map of the AABB onto [-1,1]^3, same op order."""
    p = p.reshape(-1, 3)
    lo = bound[:, 0]
    hi = bound[:, 1]
    return ((p - lo) / (hi - lo)) * 2 - 1.0


def _unnormalize_clip(coord, size):
    """grid_sample(align_corners=True, padding_mode='border') coordinate rule.

    ((c+1)/2)*(size-1), clipped to [0, size-1]; the clip passes gradient only strictly inside
    (ATen grid_sampler clip_coordinates_set_grad: zero at and beyond both ends).
    """
    x = ((coord + 1) / 2) * (size - 1)
    inside = (x > 0) & (x < size - 1)
    return torch.where(inside, x, x.detach().clamp(0, size - 1))


def bilinear_border(plane, gx, gy):
    """One F.grid_sample call of reference src/networks/decoders.py:79-81, as index arithmetic.

    plane [1,C,h,w]; gx indexes the width axis, gy the height axis, both in [-1,1] nominal.
    Returns [N,C].
    """
    _, C, h, w = plane.shape
    x = _unnormalize_clip(gx, w)
    y = _unnormalize_clip(gy, h)
    x0 = x.detach().floor()
    y0 = y.detach().floor()
    tx = x - x0
    ty = y - y0
    x0 = x0.long()
    y0 = y0.long()
    x1 = (x0 + 1).clamp(max=w - 1)
    y1 = (y0 + 1).clamp(max=h - 1)
    flat = plane.reshape(C, h * w)
    t00 = flat[:, y0 * w + x0].t()
    t01 = flat[:, y0 * w + x1].t()
    t10 = flat[:, y1 * w + x0].t()
    t11 = flat[:, y1 * w + x1].t()
    tx = tx[:, None]
    ty = ty[:, None]
    return (t00 * (1 - tx) * (1 - ty) + t01 * tx * (1 - ty) + t10 * (1 - tx) * ty + t11 * tx * ty)


# "index": explicit index arithmetic (independent of ATen's grid_sampler; what the parity tests use).
# "grid_sample": the torch op the reference itself calls (decoders.py:79-81) - same numbers (tests/test_oracle_golden.py
# checks both against the fixtures), used for the timed baselines in bench.py because it is what the reference would run.
BILINEAR_IMPL = "index"


def _bilinear(plane, gx, gy):
    if BILINEAR_IMPL == "grid_sample":
        grid = torch.stack([gx, gy], -1)[None, :, None, :]
        out = torch.nn.functional.grid_sample(plane, grid, padding_mode="border", align_corners=True, mode="bilinear")
        return out[0, :, :, 0].t()
    return bilinear_border(plane, gx, gy)


def plane_features(p_nor, planes_xy, planes_xz, planes_yz):
    """Reference src/networks/decoders.py:64-85: per level sum the three orientations, concat levels."""
    x, y, z = p_nor[:, 0], p_nor[:, 1], p_nor[:, 2]
    feats = []
    for lvl in range(len(planes_xy)):
        f = _bilinear(planes_xy[lvl], x, y)
        f = f + _bilinear(planes_xz[lvl], x, z)
        f = f + _bilinear(planes_yz[lvl], y, z)
        feats.append(f)
    return torch.cat(feats, -1)


def _mlp(feat, params, prefix):
    """Two hidden ReLU layers + linear head (reference decoders.py:99-103 / 119-123)."""
    c = "c_" if prefix == "rgb" else ""
    h = feat
    for i in (0, 1):
        h = torch.relu(h @ params[f"{c}linears.{i}.weight"].t() + params[f"{c}linears.{i}.bias"])
    return h @ params[f"{c}output_linear.weight"].t() + params[f"{c}output_linear.bias"]


def raw_sdf(p_nor, all_planes, params):
    """Reference src/networks/decoders.py:87-105."""
    feat = plane_features(p_nor, all_planes[0], all_planes[1], all_planes[2])
    return torch.tanh(_mlp(feat, params, "sdf")).squeeze(-1)


def raw_rgb(p_nor, all_planes, params):
    """Reference src/networks/decoders.py:107-125."""
    feat = plane_features(p_nor, all_planes[3], all_planes[4], all_planes[5])
    return torch.sigmoid(_mlp(feat, params, "rgb"))


def decode(p, all_planes, params, bound):
    """Reference src/networks/decoders.py:127-146: raw[..., :3] = rgb, raw[..., 3] = sdf."""
    shape = p.shape
    p_nor = normalize_points(p, bound.to(p.dtype))
    sdf = raw_sdf(p_nor, all_planes, params)
    rgb = raw_rgb(p_nor, all_planes, params)
    return torch.cat([rgb, sdf[:, None]], -1).reshape(*shape[:-1], 4)


# ----------------------------------------------------------------------------------------------
# a8: SDF -> alpha -> weights
# ----------------------------------------------------------------------------------------------
def sdf_to_alpha(sdf, beta):
    """Reference src/utils/Renderer.py:149-153."""
    return 1.0 - torch.exp(-beta * torch.sigmoid(-sdf * beta))


def ray_weights(alpha):
    """Reference src/utils/Renderer.py:141-142: w_i = alpha_i * prod_{j<i} (1 - alpha_j + 1e-10)."""
    trans = torch.cumprod(1.0 - alpha + 1e-10, -1)
    trans = torch.cat([torch.ones_like(trans[:, :1]), trans[:, :-1]], -1)
    return alpha * trans


# ----------------------------------------------------------------------------------------------
# a4: importance sampling for rays without depth
# ----------------------------------------------------------------------------------------------
def invert_cdf(bins, weights, u):
    """Reference src/common.py:41-77 (sample_pdf, det=False) with the draw `u` lifted out.

    NB the reference overwrites the normalised pdf with the raw weights (common.py:47-48); the CDF is
    therefore un-normalised and that quirk is kept.
    """
    cdf = torch.cumsum(weights, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    last = cdf.shape[-1] - 1
    inds = torch.searchsorted(cdf.contiguous(), u.contiguous(), right=True)
    below = (inds - 1).clamp(min=0)
    above = inds.clamp(max=last)
    c0 = torch.gather(cdf, -1, below)
    c1 = torch.gather(cdf, -1, above)
    b0 = torch.gather(bins, -1, below)
    b1 = torch.gather(bins, -1, above)
    denom = c1 - c0
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    return b0 + (u - c0) / denom * (b1 - b0)


def importance_z(rays_o, rays_d, all_planes, params, beta, bound, n_stratified, n_importance,
                 t_rand_uni=None, u=None):
    """Reference src/utils/Renderer.py:108-134 for rays with gt_depth == 0 (all under no_grad)."""
    with torch.no_grad():
        dt = rays_o.dtype
        far = aabb_exit(rays_o, rays_d, bound)[:, None] + 0.01           # Renderer.py:114-117
        t = torch.linspace(0.0, 1.0, n_stratified, dtype=dt, device=rays_o.device)
        z_uni = 0.0 * (1.0 - t) + far * t                                 # Renderer.py:119
        if t_rand_uni is not None:
            z_uni = jitter(z_uni, t_rand_uni)
        pts = rays_o[:, None, :] + rays_d[:, None, :] * z_uni[..., None]
        sdf = raw_sdf(normalize_points(pts, bound), all_planes, params).reshape(z_uni.shape)
        w = ray_weights(sdf_to_alpha(sdf, beta))
        mids = 0.5 * (z_uni[..., 1:] + z_uni[..., :-1])
        z_new = invert_cdf(mids, w[..., 1:-1], u)                         # Renderer.py:131-132
        return torch.sort(torch.cat([z_uni, z_new], -1), -1).values       # Renderer.py:133


# ----------------------------------------------------------------------------------------------
# the whole call
# ----------------------------------------------------------------------------------------------
def sample_z(rays_o, rays_d, gt_depth, all_planes, params, beta, bound, truncation,
             n_stratified, n_importance, t_rand=None, t_rand_uni=None, u=None):
    """z_vals [R,S] of reference Renderer.py:85-134.  Random inputs are per *ray of the full batch*:
    t_rand [R,S], t_rand_uni [R,n_stratified], u [R,n_importance]; rows of rays that do not use them
    are ignored (the reference draws compacted tensors; tests feed it the matching rows).
    Pass None for t_rand / t_rand_uni to disable perturbation.
    """
    R = rays_o.shape[0]
    S = n_stratified + n_importance
    gt_depth = gt_depth.reshape(-1)
    has = gt_depth > 0
    z = torch.empty(R, S, dtype=rays_o.dtype, device=rays_o.device)
    if has.any():
        z[has] = depth_guided_z(gt_depth[has], n_stratified, n_importance, truncation,
                                None if t_rand is None else t_rand[has])
    if not has.all():
        no = ~has
        z[no] = importance_z(rays_o[no].detach(), rays_d[no].detach(), all_planes, params, beta, bound,
                             n_stratified, n_importance,
                             None if t_rand_uni is None else t_rand_uni[no], u[no])
    return z


def composite(raw, z_vals, beta):
    """Reference src/utils/Renderer.py:140-147."""
    w = ray_weights(sdf_to_alpha(raw[..., 3], beta))
    rgb = (w[..., None] * raw[..., :3]).sum(-2)
    depth = (w * z_vals).sum(-1)
    return depth, rgb


def render_batch_ray(all_planes, params, beta, bound, rays_d, rays_o, truncation, gt_depth,
                     n_stratified, n_importance, t_rand=None, t_rand_uni=None, u=None, z_vals=None):
    """Reference src/utils/Renderer.py:63-147 (note the argument order rays_d, rays_o).

    Returns depth [R], rgb [R,3], sdf [R,S], z_vals [R,S].
    """
    bound = bound.to(device=rays_o.device, dtype=rays_o.dtype)
    if z_vals is None:
        z_vals = sample_z(rays_o, rays_d, gt_depth, all_planes, params, beta, bound, truncation,
                          n_stratified, n_importance, t_rand, t_rand_uni, u)
    pts = rays_o[:, None, :] + rays_d[:, None, :] * z_vals[..., None]     # Renderer.py:136-137
    raw = decode(pts, all_planes, params, bound)
    depth, rgb = composite(raw, z_vals, beta)
    return depth, rgb, raw[..., 3], z_vals


# ----------------------------------------------------------------------------------------------
# caller-side losses (needed to produce the upstream gradients of a real iteration)
# ----------------------------------------------------------------------------------------------
def sdf_losses(sdf, z_vals, gt_depth, truncation, w_fs, w_center, w_tail):
    """Reference src/Mapper.py:110-144 (identical in Tracker.py:114-148)."""
    d = gt_depth[:, None]
    front = z_vals < (d - truncation)
    back = z_vals > (d + truncation)
    center = (z_vals > (d - 0.4 * truncation)) & (z_vals < (d + 0.4 * truncation))
    tail = (~front) & (~back) & (~center)
    pred = z_vals + sdf * truncation
    dd = d.expand_as(z_vals)
    fs = ((sdf[front] - 1.0) ** 2).mean()
    ce = ((pred[center] - dd[center]) ** 2).mean()
    ta = ((pred[tail] - dd[tail]) ** 2).mean()
    return w_fs * fs + w_center * ce + w_tail * ta


MAPPING_W = dict(w_fs=5.0, w_center=200.0, w_tail=10.0, w_depth=0.1, w_color=5.0)     # configs/ESLAM.yaml:53-57
TRACKING_W = dict(w_fs=10.0, w_center=200.0, w_tail=50.0, w_depth=1.0, w_color=5.0)  # configs/ESLAM.yaml:29-33


def mapping_loss(depth, color, sdf, z_vals, gt_depth, gt_color, truncation, w=MAPPING_W):
    """Reference src/Mapper.py:337-346."""
    m = gt_depth > 0
    loss = sdf_losses(sdf[m], z_vals[m], gt_depth[m], truncation, w["w_fs"], w["w_center"], w["w_tail"])
    loss = loss + w["w_color"] * ((gt_color - color) ** 2).mean()
    loss = loss + w["w_depth"] * ((gt_depth[m] - depth[m]) ** 2).mean()
    return loss


def tracking_loss(depth, color, sdf, z_vals, gt_depth, gt_color, truncation, w=TRACKING_W):
    """Reference src/Tracker.py:192-204 (10x-median outlier mask)."""
    err = (gt_depth - depth.detach()).abs()
    m = err < 10 * err.median()
    loss = sdf_losses(sdf[m], z_vals[m], gt_depth[m], truncation, w["w_fs"], w["w_center"], w["w_tail"])
    loss = loss + w["w_color"] * ((gt_color - color) ** 2)[m].mean()
    loss = loss + w["w_depth"] * ((gt_depth[m] - depth[m]) ** 2).mean()
    return loss


# ---------------------------------------------------------------------------------------------------------------
# Keyframe selection by view overlap
# ---------------------------------------------------------------------------------------------------------------
def keyframe_overlap(rays_o, rays_d, gt_depth, keyframes_c2ws, H, W, fx, fy, cx, cy, num_samples=8, edge=20):
    """Fraction of the current frame's sample points that project inside each keyframe's image
    (reference src/Mapper.py:170-201).  rays_* [n,3], gt_depth [n] are get_samples' outputs for the current frame;
    keyframes_c2ws [K,4,4] excludes the last two keyframes (Mapper.py:181).  Returns percent_inside [K] float32."""
    keep = gt_depth > 0                                                  # Mapper.py:171-174
    o, d, dep = rays_o[keep], rays_d[keep], gt_depth[keep].reshape(-1, 1)
    t = torch.linspace(0., 1., steps=num_samples, dtype=dep.dtype)
    near, far = dep * 0.8, dep + 0.5                                     # Mapper.py:177-178
    z = near * (1. - t) + far * t
    pts = (o[:, None, :] + d[:, None, :] * z[:, :, None]).reshape(-1, 3)
    w2c = torch.inverse(keyframes_c2ws)                                  # Mapper.py:183
    homo = torch.cat([pts, torch.ones_like(pts[:, :1])], -1)             # [N,4]
    cam = torch.einsum("kij,nj->kni", w2c, homo)[:, :, :3].clone()       # [K,N,3]
    cam[:, :, 0] *= -1                                                   # Mapper.py:192
    Kmat = torch.tensor([[fx, .0, cx], [.0, fy, cy], [.0, .0, 1.0]], dtype=pts.dtype)
    uv = torch.einsum("ij,knj->kni", Kmat, cam)
    zc = uv[:, :, 2] + 1e-5
    u, v = uv[:, :, 0] / zc, uv[:, :, 1] / zc
    mask = (u < W - edge) & (u > edge) & (v < H - edge) & (v > edge) & (zc < 0)     # Mapper.py:196-199
    return mask.sum(dim=1) / mask.shape[1]


def select_overlapping(percent_inside, num_keyframes, perm):
    """Mapper.py:203-207 with the permutation supplied: nonzero(percent) shuffled by `perm`, first num_keyframes."""
    sel = torch.nonzero(percent_inside).squeeze(-1)
    return [int(i) for i in sel[perm[:num_keyframes]]]
