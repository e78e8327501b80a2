"""Keyframe selection by view overlap (reference src/Mapper.py:146-209).
CPU: the oracle's restatement against the fixture the reference itself produced (tests/golden/make_golden.py).
GPU: eslam_keyframe_overlap against the oracle (counts must be equal) and the drop-in method against the fixture."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import eslam_oracle as orc
from tests import helpers as hp


def _inputs(fx, device="cpu"):
    from myslam_amd import scene as scn, synth
    sc = scn.make_scene("room0")
    depth = torch.from_numpy(synth.depth_image(sc.H, sc.W, 720, 0.1)).to(device)
    color = torch.from_numpy(synth.color_image(sc.H, sc.W, 721)).to(device)
    call = fx["rand_calls"].tolist()[0].split(";")
    assert call[0] == "randint" and int(call[2]) == sc.H * sc.W
    idx = torch.from_numpy(synth.hash_randint(int(call[2]), (int(fx["num_rays"]),), int(call[1]))).to(device)
    return sc, depth, color, idx


def _perm(n):
    from myslam_amd import synth
    return torch.from_numpy(np.argsort(synth.hash_uniform((n,), 730), kind="stable"))


def test_oracle_selects_what_the_reference_selected():
    fx = hp.load("keyframe_overlap_room0")
    sc, depth, color, idx = _inputs(fx)
    cur = torch.from_numpy(fx["cur_c2w"])
    ro, rd, d, _ = orc.rays_from_pixels(idx, 0, sc.H, 0, sc.W, sc.fx, sc.fy, sc.cx, sc.cy, cur[None], depth[None], color[None])
    pct = orc.keyframe_overlap(ro, rd, d, torch.from_numpy(fx["c2ws"])[:-2], sc.H, sc.W, sc.fx, sc.fy, sc.cx, sc.cy,
                               int(fx["num_samples"]))
    assert pct.shape == (int(fx["K"]) - 2,)
    n_sel = int((pct != 0).sum())
    assert orc.select_overlapping(pct, int(fx["K"]), _perm(n_sel)) == fx["selected_all"].tolist()
    assert orc.select_overlapping(pct, 4, _perm(n_sel)) == fx["selected_four"].tolist()


@pytest.mark.gpu
def test_kernel_counts_equal_oracle_and_method_matches_reference(monkeypatch):
    from myslam_amd import keyframes
    from myslam_amd.src.common import get_samples_at
    dev = torch.device("cuda:0")
    fx = hp.load("keyframe_overlap_room0")
    sc, depth, color, idx = _inputs(fx, dev)
    cur = torch.from_numpy(fx["cur_c2w"]).to(dev)
    c2ws = torch.from_numpy(fx["c2ws"]).to(dev)
    n = int(fx["num_rays"])
    ro, rd, d, _ = get_samples_at(idx, 0, sc.H, 0, sc.W, n, sc.H, sc.W, sc.fx, sc.fy, sc.cx, sc.cy, cur[None], depth[None], color[None])
    cnt = keyframes.overlap_counts(ro, rd, d, c2ws[:-2], sc.H, sc.W, sc.fx, sc.fy, sc.cx, sc.cy).cpu()
    pct_ref = orc.keyframe_overlap(ro.cpu(), rd.cpu(), d.cpu(), c2ws[:-2].cpu(), sc.H, sc.W, sc.fx, sc.fy, sc.cx, sc.cy)
    n_valid = int((d > 0).sum())
    assert int(cnt[-1]) == n_valid and 0 < n_valid < n
    assert torch.equal(cnt[:-1].long(), torch.round(pct_ref * n_valid * 8).long())
    pct = keyframes.percent_inside(ro, rd, d, c2ws[:-2], sc.H, sc.W, sc.fx, sc.fy, sc.cx, sc.cy).cpu()
    assert torch.equal(pct, pct_ref)
    # the drop-in method with the fixture's random numbers: same list as the reference returned
    from myslam_amd import synth
    import myslam_amd.src.common as common
    monkeypatch.setattr(torch, "randint", lambda high, size, **kw: torch.from_numpy(
        synth.hash_randint(high, tuple(size), 740)).to(kw.get("device", "cpu")))
    monkeypatch.setattr(torch, "randperm", lambda k, **kw: _perm(k))
    ns = SimpleNamespace(device=dev, H=sc.H, W=sc.W, fx=sc.fx, fy=sc.fy, cx=sc.cx, cy=sc.cy, estimate_c2w_list=c2ws,
                         keyframe_list=list(range(int(fx["K"]))))
    sel = keyframes.keyframe_selection_overlap(ns, color, depth, cur, int(fx["K"]))
    assert [int(i) for i in sel] == fx["selected_all"].tolist()
    sel4 = keyframes.keyframe_selection_overlap(ns, color, depth, cur, 4)
    assert [int(i) for i in sel4] == fx["selected_four"].tolist()
    # edge cases: no keyframes beyond the last two; every ray without depth (0/0 -> NaN -> "selected", as the reference)
    assert keyframes.overlap_counts(ro, rd, d, c2ws[:0], sc.H, sc.W, sc.fx, sc.fy, sc.cx, sc.cy).cpu().tolist() == [n_valid]
    z = keyframes.overlap_counts(ro, rd, torch.zeros_like(d), c2ws[:3], sc.H, sc.W, sc.fx, sc.fy, sc.cx, sc.cy).cpu()
    assert z.tolist() == [0, 0, 0, 0]
