import sys, numpy as np, torch
sys.path.insert(0, '.')
from tests import helpers as hp
from tests.test_gpu_parity import build
from myslam_amd import ops
dev = torch.device('cuda:0')
for case in ['room0_200x40_noperturb', 'room0_200x32']:
    fx = hp.load(case)
    sc, planes, dec, renderer = build(fx)
    t_rand, t_uni, u = hp.rand_inputs(fx)
    rand = tuple(None if t is None else t.to(dev) for t in (t_rand, t_uni, u))
    gd = torch.from_numpy(fx['gt_depth']).to(dev)
    ro = torch.from_numpy(fx['rays_o']).to(dev); rd = torch.from_numpy(fx['rays_d']).to(dev)
    for mode in ['gpu_linspace', 'cpu_linspace']:
        ops._const_cache.clear()
        if mode == 'cpu_linspace':
            for n in (int(fx['n_stratified']), int(fx['n_importance'])):
                ops._const_cache[('lin', n, dev.index)] = torch.linspace(0., 1., steps=n).to(dev)
        z = ops.sample_z(ro, rd, gd, planes, dec, renderer._bound6, float(fx['truncation']), int(fx['n_stratified']),
                         int(fx['n_importance']), bool(fx['perturb']), rand).cpu().numpy()
        ref = fx['z_vals']
        neq = (z != ref)
        ulp = np.abs(z.view(np.int32).astype(np.int64) - ref.view(np.int32).astype(np.int64))
        print(case, mode, 'mismatch', neq.sum(), 'of', z.size, 'max ulp', ulp.max(), 'cols', np.unique(np.nonzero(neq)[1])[:20])
    a = torch.linspace(0., 1., steps=24).numpy(); b = torch.linspace(0., 1., steps=24, device=dev).cpu().numpy()
    print('linspace equal cpu/gpu:', np.array_equal(a, b))
