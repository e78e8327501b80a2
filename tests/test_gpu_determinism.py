"""ESLAM_DETERMINISTIC=1: fixed-order reduction of the loss's sums and fixed-point plane-gradient
scatter.  The mode is read once per process, so the checks run in child processes:

  * two evaluations of the same mapping iteration give bit-identical loss and gradients (without the mode the float
    atomics of the scatter and of the loss sums leave run-to-run differences in the last bits - measured and printed);
  * with it, the ray-sharded mapper + fused Adam over 6 iterations agrees with the plain single-GPU loop at the 1e-5 that
    round 1 had to widen to 3e-4 (commit a6bd2ae): the widening covered atomic-order noise amplified by Adam, not an error.
"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import hashlib, json, os, socket, sys
import numpy as np, torch
sys.path.insert(0, %(root)r)
from myslam_amd import harness, optim, _hip
dev = torch.device("cuda:0")
out = {"det": int(_hip.lib().eslam_deterministic())}

def digest(ts):
    h = hashlib.sha256()
    for t in ts:
        h.update(t.detach().cpu().numpy().tobytes())
    return h.hexdigest()

# (1) the same iteration twice (fixed samples: perturb off), fresh workload objects so that nothing is shared
runs = []
for rep in range(2):
    wl = harness.make_workload("room0", 1500, 24, 8, device=dev, zero_frac=0.1)
    wl.renderer.perturb = False
    torch.manual_seed(7)
    from myslam_amd import ops
    ops._rng_state(dev).zero_()
    loss = wl.step()
    torch.cuda.synchronize()
    grads = [p.grad for p in wl.params()]
    runs.append((float(loss), digest(grads), [g.detach().clone() for g in grads]))
out["loss_equal"] = runs[0][0] == runs[1][0]
out["grads_bitwise_equal"] = runs[0][1] == runs[1][1]
out["max_rel_diff_between_runs"] = max(float((a - b).abs().max() / (b.abs().max() + 1e-30)) for a, b in zip(runs[0][2], runs[1][2]))

# (2) sharded mapper (1 RCCL rank) + fused Adam, 3 eager + 3 replayed iterations, against the plain loop
import torch.distributed as dist
from myslam_amd.parallel import ShardedMapper
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
wa = harness.make_workload("room0", 512, 32, 8, device=dev, planes="synth")
wb = harness.make_workload("room0", 512, 32, 8, device=dev, planes="synth")
wa.renderer.perturb = wb.renderer.perturb = False
ma = ShardedMapper(wa)
ma.make_optimizer(fused_zero_grad=True, capturable=True)
for _ in range(3):
    ma.step()
ma.capture(warmup=0)
for _ in range(3):
    ma.step()
la = float(ma.loss)
torch.cuda.synchronize()
dec_b = list(wb.decoders.parameters())
ob = optim.Adam([{"params": dec_b, "lr": 0.001}, {"params": wb.plane_list[:6], "lr": 0.005}, {"params": wb.plane_list[6:], "lr": 0.005}])
for _ in range(6):
    lb = wb.step()
    ob.step()
out["adam_loss_rel"] = abs(la - float(lb)) / abs(float(lb))
out["adam_param_rel"] = max(float((a.detach() - b.detach()).abs().max() / (b.detach().abs().max() + 1e-30)) for a, b in zip(wa.params(), wb.params()))
dist.destroy_process_group()
print("RESULT " + json.dumps(out))
'''


def _child(det):
    env = dict(os.environ)
    env["ESLAM_DETERMINISTIC"] = "1" if det else "0"
    p = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[7:])


def test_deterministic_mode_is_bitwise_reproducible_and_restores_the_tolerance():
    d = _child(True)
    print("deterministic:", d)
    assert d["det"] == 1
    assert d["loss_equal"] and d["grads_bitwise_equal"] and d["max_rel_diff_between_runs"] == 0.0
    # the pre-a6bd2ae tolerances of test_sharded_mapper_one_rank_rccl
    assert d["adam_loss_rel"] <= 1e-5, d
    assert d["adam_param_rel"] <= 1e-5, d


def test_default_mode_run_to_run_spread_is_rounding_noise():
    """Without the mode: record the spread (float-atomic order) and bound it - noise of the last bits, not an error."""
    d = _child(False)
    print("default mode:", d)
    assert d["det"] == 0
    assert d["max_rel_diff_between_runs"] <= 2e-5
    assert d["adam_loss_rel"] <= 1e-4 and d["adam_param_rel"] <= 3e-4
