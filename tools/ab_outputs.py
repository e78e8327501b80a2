"""A/B check of two builds of the library on identical inputs: dump outputs + gradients of a list of configurations
(python tools/ab_outputs.py dump out.npz, once per build via ESLAM_HIP_LIB), then `compare a.npz b.npz`."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CONFIGS = [("toy", 999, 32, 8, 0.0), ("toy", 497, 32, 8, 0.0), ("toy", 333, 32, 8, 0.1), ("toy", 1000, 16, 8, 0.0),
           ("room0", 200, 24, 8, 0.0), ("room0", 1001, 56, 8, 0.0), ("room0", 301, 88, 8, 0.1), ("room0", 77, 8, 4, 0.0),
           ("room0", 130, 120, 8, 0.0)]
if sys.argv[1] == "dump":
    import torch
    from myslam_amd import harness
    dev = torch.device("cuda:0")
    out = {}
    for i, (sc, R, ns, ni, zf) in enumerate(CONFIGS):
        wl = harness.make_workload(sc, R, ns, ni, device=dev, zero_frac=zf)
        o = wl.forward()
        g = wl.backward_with(o)
        for k, t in zip(("depth", "rgb", "sdf", "z"), o):
            out[f"{i}_{k}"] = t.detach().cpu().numpy()
        for j, t in enumerate(g):
            out[f"{i}_g{j}"] = t.cpu().numpy()
        with torch.no_grad():
            o2 = wl.forward()
        out[f"{i}_nograd_depth"] = o2[0].cpu().numpy()
        out[f"{i}_nograd_rgb"] = o2[1].cpu().numpy()
        torch.manual_seed(0)
        from myslam_amd import ops
        ops._rng_state(dev).zero_()           # same jitter in both builds
        loss = wl.step()                      # fused loss path (eslam_render_fwd_loss / eslam_render_bwd_loss), in-kernel jitter
        out[f"{i}_step_loss"] = loss.detach().cpu().numpy()
        for j, p_ in enumerate(wl.params()):
            out[f"{i}_sg{j}"] = p_.grad.cpu().numpy()
    np.savez(sys.argv[2], **out)
else:
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    for k in a.files:
        x, y = a[k], b[k]
        err = float(np.abs(x - y).max() / (np.abs(y).max() + 1e-30))
        if err > 1e-5:
            print("DIFF", k, CONFIGS[int(k.split('_')[0])], err)
    print("compared", len(a.files), "arrays")
