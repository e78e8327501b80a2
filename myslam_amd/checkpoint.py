"""Checkpoints in the reference's format (src/utils/Logger.py:41-47): a torch-saved dict with the keys
`decoder_state_dict`, `gt_c2w_list`, `estimate_c2w_list`, `keyframe_list`, `idx` - so a run of the HIP path can be
evaluated by the reference's own tools (src/tools/eval_ate.py reads `estimate_c2w_list` / `gt_c2w_list`) and a
reference checkpoint's decoders load into myslam_amd's Decoders (same state_dict keys).  Planes are NOT part of the
reference's checkpoint (Logger.py saves the decoders and the trajectories only); `planes=` stores them under an extra key
for resuming our own runs."""
import torch

KEYS = ("decoder_state_dict", "gt_c2w_list", "estimate_c2w_list", "keyframe_list", "idx")


def save(path, decoders, gt_c2w_list, estimate_c2w_list, keyframe_list, idx, planes=None):
    def stack(lst):
        return lst if torch.is_tensor(lst) else torch.stack([t.detach().cpu() for t in lst], 0)

    ckpt = {"decoder_state_dict": {k: v.detach().cpu() for k, v in decoders.state_dict().items()},
            "gt_c2w_list": stack(gt_c2w_list).cpu(), "estimate_c2w_list": stack(estimate_c2w_list).cpu(),
            "keyframe_list": [int(k) for k in keyframe_list], "idx": int(idx)}
    if planes is not None:
        ckpt["all_planes"] = [[p.detach().cpu() for p in grp] for grp in planes]
    torch.save(ckpt, path)
    return ckpt


def load(path, decoders=None, map_location="cpu"):
    """Safe load (tensors and plain containers only).  Fills `decoders` when given; returns the dict."""
    ckpt = torch.load(path, map_location=map_location, weights_only=True)
    missing = [k for k in KEYS if k not in ckpt]
    if missing:
        raise KeyError(f"not an ESLAM checkpoint: missing {missing}")
    if decoders is not None:
        decoders.load_state_dict(ckpt["decoder_state_dict"])
    return ckpt
