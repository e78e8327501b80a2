"""Ray-sharded data parallelism for mapping iterations - new, the reference is single-GPU.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  Planes and decoders are replicated; the
rays of an iteration are split across ranks; rays are independent in the forward pass, so the only exchange steps are

  1. one int32 all-reduce(sum) between forward and backward (sync_pack / sync_unpack): the five sizes of the loss's masked
     sets, so that every rank scales its upstream gradients by the GLOBAL denominators of the reference's means
     (src/Mapper.py:136-140,343,346) and the summed gradients equal those of the unsharded batch - and, in the same
     buffer, the union of the texels the ranks' backward passes will touch;
  2. one all-reduce(sum) of the gradients: ONE flat float32 buffer holding the 12 plane gradients, the 2692 decoder
     gradients, beta's and the loss's 16 sums (27.15 MB for room0, 70.5 MB for scene0000), of which only the union's
     texels are exchanged (block-sparse) when the planes are channels_last.

Identical optimiser steps on every rank then keep the replicas in sync without a broadcast.
The flat buffer is also what RenderFn.backward scatters into (ops.grad_sink), so no copy sits between the backward
kernels and the collective.  Tracking (pose-only, needs a global median) is not sharded: "replicas only".
"""
import ctypes
import os

import torch
import torch.distributed as dist

from . import _hip, ops


def shard_slice(n, rank, world):
    """Contiguous shard [lo, hi) of n rays for `rank`; the shards of all ranks tile [0, n) exactly."""
    per = (n + world - 1) // world
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


SYNC_HEAD = 8                                   # eslam_hip.h: words 0..4 = N_front, N_center, N_tail, N_depth, N_color
_ACC_COUNT_SLOTS = (0, 1, 2, 6, 9)              # their slots in acc [16] (eslam_loss_final.h)
_ACC_SUM_SLOTS = (3, 4, 5, 7, 8)                # S_front, S_center, S_tail, S_depth, S_color


def sync_words(n_blocks):
    return SYNC_HEAD + (n_blocks + 5) // 6


def sync_pack(acc, touched, out):
    """acc [16] float32 + touched [n] uint8 (or None) -> out [sync_words(n)] int32 (eslam_shard_sync_pack; the same
    arithmetic in tensor ops on the CPU, which is what the gloo tests run)."""
    n = 0 if touched is None else touched.numel()
    if acc.is_cuda:
        with _hip.on_device(acc.device):
            _hip.check(_hip.lib().eslam_shard_sync_pack(_hip.ptr(acc), _hip.ptr(touched), n, _hip.ptr(out),
                                                        _hip.stream_handle(acc.device)), "eslam_shard_sync_pack")
        return out
    out[:SYNC_HEAD] = 0
    out[:5] = acc[list(_ACC_COUNT_SLOTS)].to(torch.int32)
    if n:
        pad = torch.zeros(6 * ((n + 5) // 6), dtype=torch.int32)
        pad[:n] = (touched != 0).to(torch.int32)
        out[SYNC_HEAD:] = (pad.view(-1, 6) << (4 * torch.arange(6, dtype=torch.int32))).sum(1).to(torch.int32)
    return out


def sync_unpack(buf, acc_local, acc_global, touched):
    """Inverse of sync_pack after the SUM all-reduce: global set sizes into acc_global (the other slots copied from
    acc_local), union of the touched texels into touched (eslam_shard_sync_unpack)."""
    n = 0 if touched is None else touched.numel()
    if buf.is_cuda:
        with _hip.on_device(buf.device):
            _hip.check(_hip.lib().eslam_shard_sync_unpack(_hip.ptr(buf), n, _hip.ptr(acc_local), _hip.ptr(acc_global),
                                                          _hip.ptr(touched), _hip.stream_handle(buf.device)),
                       "eslam_shard_sync_unpack")
        return
    acc_global.copy_(acc_local)
    acc_global[list(_ACC_COUNT_SLOTS)] = buf[:5].to(acc_global.dtype)
    if n:
        nib = (buf[SYNC_HEAD:, None] >> (4 * torch.arange(6, dtype=torch.int32))) & 15
        touched.copy_((nib.reshape(-1)[:n] != 0).to(torch.uint8))


def loss_from_acc(acc, weights):
    """The loss value from the 16 accumulators (sums / set sizes), as eslam_loss_value forms it."""
    w = torch.tensor(weights, dtype=acc.dtype, device=acc.device)
    return (w * acc[list(_ACC_SUM_SLOTS)] / acc[list(_ACC_COUNT_SLOTS)]).sum()


class FlatGrads:
    """One flat buffer + per-parameter views with the parameters' own shapes and strides."""

    def __init__(self, params, extra=0):
        self.params = list(params)
        dev = self.params[0].device
        sizes = [p.numel() for p in self.params]
        self.flat = torch.zeros(sum(sizes) + extra, device=dev, dtype=self.params[0].dtype)
        self.views, self.offsets, off = [], [], 0
        for p, n in zip(self.params, sizes):
            self.offsets.append(off)
            dense = p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))
            if not dense:
                raise RuntimeError("FlatGrads needs dense parameters")
            self.views.append(self.flat[off:off + n].as_strided(p.shape, p.stride()))
            off += n
        self.offsets.append(off)
        self.extra = self.flat[off:]
        self.clean = True            # all zero (fresh, or cleared by an optimiser step with fused_zero_grad)

    def zero_(self):
        """Clear the buffer before a backward scatters into it - skipped when its last consumer left it zero."""
        if not self.clean:
            self.flat.zero_()
        self.clean = False

    def zero_blocks_(self, idx, n_block_elems):
        """Sparse clear: every non-zero of the first n_block_elems floats lies in the 32-float blocks `idx` (the union the
        last exchange_blocks wrote back), the rest of the buffer (decoder gradients, loss sums) is dense and small."""
        tail = self.flat[n_block_elems:]
        if self.flat.is_cuda and self.flat.dtype == torch.float32:
            with _hip.on_device(self.flat.device):
                _hip.check(_hip.lib().eslam_blocks_zero(_hip.ptr(self.flat), _hip.ptr(idx), idx.numel(), _hip.ptr(tail),
                                                        tail.numel(), _hip.stream_handle(self.flat.device)),
                           "eslam_blocks_zero")
        else:
            self.flat[:n_block_elems].view(-1, 32).index_fill_(0, idx, 0.0)
            tail.zero_()
        self.clean = True

    def all_reduce(self, group=None, async_op=False):
        return dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)

    def all_reduce_compact(self, n_block_elems, group=None, block=32):
        """Sum over ranks, exchanging only the blocks of `block` floats that are non-zero on SOME rank.

        The first n_block_elems floats (the planes: one block = one texel's 32 channels in channels_last planes) are
        block-sparse after a mapping backward - a frame's rays touch 4-25 % of the texels, and
        ranks that render the same keyframe window touch nearly the same ones - so the 27-70 MB dense all-reduce, which
        is what bounds ray-sharded scaling over xGMI, becomes three steps:
          1. all-reduce(MAX) of one byte per block (212 KB for room0)  -> the union of touched blocks, same on all ranks
          2. gather the union's blocks + the dense tail (decoder / beta gradients) into one buffer, all-reduce(SUM)
          3. scatter the blocks back.
        `nonzero()` needs the union's size on the host: one stream synchronisation per step, which the collectives -
        issued eagerly between the captured phases anyway - tolerate.  Plain tensor ops on purpose: the identical code
        runs under gloo on the CPU in the tests, so the multi-rank logic is verified without multi-GPU hardware."""
        if n_block_elems % block:
            raise RuntimeError("all_reduce_compact: the block-sparse prefix must be a multiple of the block size")
        rows = self.flat[:n_block_elems].view(-1, block)
        on_gpu = self.flat.is_cuda and self.flat.dtype == torch.float32 and block == 32
        if on_gpu:                 # one launch for the bitmap
            lib, dev = _hip.lib(), self.flat.device
            touched = torch.empty(rows.shape[0], dtype=torch.uint8, device=dev)
            with _hip.on_device(dev):
                _hip.check(lib.eslam_blocks_touched(_hip.ptr(self.flat), rows.shape[0], _hip.ptr(touched),
                                                    _hip.stream_handle(dev)), "eslam_blocks_touched")
        else:
            touched = (torch.count_nonzero(rows, dim=1) > 0).to(torch.uint8)
        dist.all_reduce(touched, op=dist.ReduceOp.MAX, group=group)
        idx = touched.nonzero().squeeze(1)                     # host sync: the union's size
        return self.exchange_blocks(idx, n_block_elems, group, block)

    def exchange_blocks(self, idx, n_block_elems, group=None, block=32):
        """Steps 2 and 3 of all_reduce_compact for a union `idx` (ascending block indices, identical on every rank)."""
        rows = self.flat[:n_block_elems].view(-1, block)
        tail = self.flat[n_block_elems:]
        k = idx.numel() * block
        if self.flat.is_cuda and self.flat.dtype == torch.float32 and block == 32:
            lib, dev = _hip.lib(), self.flat.device
            buf = torch.empty(k + tail.numel(), device=dev)
            with _hip.on_device(dev):
                _hip.check(lib.eslam_blocks_pack(_hip.ptr(self.flat), _hip.ptr(idx), idx.numel(), _hip.ptr(tail),
                                                 tail.numel(), _hip.ptr(buf), _hip.stream_handle(dev)), "eslam_blocks_pack")
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
            with _hip.on_device(dev):
                _hip.check(lib.eslam_blocks_unpack(_hip.ptr(self.flat), _hip.ptr(idx), idx.numel(), _hip.ptr(tail),
                                                   tail.numel(), _hip.ptr(buf), _hip.stream_handle(dev)),
                           "eslam_blocks_unpack")
        else:
            buf = torch.cat([rows.index_select(0, idx).reshape(-1), tail])
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
            rows.index_copy_(0, idx, buf[:k].view(-1, block))
            tail.copy_(buf[k:])
        self.last_exchange = (int(buf.numel()) * buf.element_size() + rows.shape[0], self.flat.numel() * self.flat.element_size())
        return self.last_exchange

    def assign(self):
        """Point every parameter's .grad at its view (after the collective)."""
        for p, v in zip(self.params, self.views):
            p.grad = v


class ShardedMapper:
    """Drives one ray-sharded mapping iteration on this rank's shard (a harness.Workload holding this rank's rays).

    step() = phase_a (sample, render forward with the loss's sums in its epilogue, texel marking, sync_pack)
             -> ONE int32 all-reduce (global set sizes of the loss + union of touched texels) -> sync_unpack
             -> phase_b (render backward, the loss's gradients formed inside the kernel from the global set sizes, into the
                flat gradient buffer) -> exchange of that buffer (block-sparse over the union, or dense).
    The phases contain no collective, so each can be captured into a hipGraph of its own (capture()); the two
    collectives are issued eagerly between the replays - RCCL never has to run inside a captured graph.

    optimizer: optional myslam_amd.optim.Adam over self.params (build it with make_optimizer): stepped after the
    gradient exchange - every rank applies the same update to identical replicas, so they stay in sync without a
    broadcast.  With fused_zero_grad it leaves the flat buffer zero, which saves the 27-70 MB
    fill of the next iteration; under capture() it becomes a third graph (capturable=True is required for that)."""

    def __init__(self, workload, group=None, optimizer=None, compact=None):
        from . import losses
        self.wl = workload
        self.group = group
        self.weights = losses.MAPPING_W
        world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        if world > 15:
            raise RuntimeError("ShardedMapper: the sync buffer counts ranks per texel in 4 bits: at most 15 ranks")
        self.params = workload.plane_list + ops.decoder_params(workload.decoders)
        beta = workload.decoders.beta
        self.has_beta = torch.is_tensor(beta)
        if self.has_beta:
            self.params = self.params + [beta]
        self.grads = FlatGrads(self.params, extra=16)          # + the loss's 16 sums: they ride in the gradient exchange
        self.optimizer = optimizer
        # gradient exchange: block-sparse (FlatGrads.exchange_blocks) unless ESLAM_DP_COMPACT=0 / compact=False
        self.compact = (os.environ.get("ESLAM_DP_COMPACT", "1") != "0") if compact is None else bool(compact)
        self._n_plane_elems = sum(p.numel() for p in self.params[:12])
        # With channels_last planes the union of touched texels is known from the sample positions right after the
        # forward pass (eslam_mark_touched): the ranks agree on it - and the host learns its size - while the backward
        # pass is still running, so the exchange itself follows the backward without a bubble.
        planes = self.params[:12]
        self._can_mark = bool(self.compact and planes[0].is_cuda and
                              all(p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last) for p in planes))
        dev = workload.device
        n_blocks = self._n_plane_elems // 32 if self._can_mark else 0
        self._touched = torch.zeros(n_blocks, dtype=torch.uint8, device=dev) if self._can_mark else None
        self._sync = torch.zeros(sync_words(n_blocks), dtype=torch.int32, device=dev)
        self._gacc = torch.zeros(16, device=dev)               # acc with the GLOBAL set sizes: what the backward scales by
        if self._can_mark:
            self._block_base = (ctypes.c_int64 * 12)(*[self.grads.offsets[i] // 32 for i in range(12)])
            self._side = torch.cuda.Stream(device=dev)
        self._pre = None
        self._graphs = None
        self._last_idx = None                      # union of the previous iteration's exchange: what a sparse clear must zero

    @property
    def loss(self):
        """The global loss of the last step (from the 16 sums that came back with the gradient exchange)."""
        return loss_from_acc(self.grads.extra[:16], self.weights)

    def phase_a(self):
        wl = self.wl
        for p in self.params:
            p.grad = None
        with ops.keep_layout():                    # the backward scatters into the flat buffer's views: the planes' own strides
            depth, color, sdf, z, pre = wl.renderer.render_batch_ray_with_loss(
                wl.planes, wl.decoders, wl.rays_d, wl.rays_o, wl.device, wl.truncation, wl.gt_depth, wl.gt_color, self.weights)
        pre.acc_global = self._gacc                # RenderFn.backward scales by these, not by this rank's own set sizes
        self._pre = pre
        ops.join_ray_order(wl.device)              # forward and backward are separate graphs: join the fork in this one
        if self._can_mark:
            arr, _ = _hip.make_planes(tuple([p.detach() for p in grp] for grp in wl.planes))
            with _hip.on_device(wl.device):
                # the marking must use the bound the forward and backward kernels normalise with (decoders.bound)
                _hip.check(_hip.lib().eslam_mark_touched(arr, _hip.make_bound(ops.bound_to_host(wl.decoders.bound)),
                                                         _hip.ptr(wl.rays_o.detach()), _hip.ptr(wl.rays_d.detach()),
                                                         _hip.ptr(z), wl.R, wl.S, self._block_base,
                                                         self._touched.numel(), _hip.ptr(self._touched),
                                                         _hip.stream_handle(wl.device)), "eslam_mark_touched")
        sync_pack(pre.acc, self._touched, self._sync)

    def phase_b(self):
        with ops.grad_sink(self.grads):
            self._pre.loss.backward()
        self.grads.extra[:16].copy_(self._pre.acc)         # this rank's sums and set sizes: summed with the gradients

    def make_optimizer(self, lrs=(0.001, 0.005, 0.005), **kw):
        """Adam with the mapper's three groups (decoders / planes / c_planes, src/Mapper.py:296-303;
        learning rates of configs/ESLAM.yaml:58-61 by default).  Gradients are bound to the flat buffer's views."""
        from . import optim
        n_dec = len(self.params) - 12
        self.grads.assign()
        self.optimizer = optim.Adam([{"params": self.params[12:12 + n_dec], "lr": lrs[0]},
                                     {"params": self.params[0:6], "lr": lrs[1]},
                                     {"params": self.params[6:12], "lr": lrs[2]}], **kw)
        return self.optimizer

    def phase_c(self):
        """Optimiser step on the all-reduced gradients (identical on every rank)."""
        self.grads.assign()
        self.optimizer.step()
        if self.optimizer.fused_zero_grad:
            self.grads.clean = True

    def _eager(self):
        return self._run(self.phase_a, self.phase_b, self.phase_c if self.optimizer is not None else None)

    def _run(self, run_a, run_b, run_c):
        """One iteration: phases a / b / c (eager calls or graph replays) with the collectives between them."""
        if not self.grads.clean:
            # the previous iteration's gradients are still in the flat buffer.  Its non-zero texels are the union that
            # iteration exchanged: zero those (1.6-4.5 MB) instead of filling 27-70 MB.  (An optimiser with
            # fused_zero_grad has left the buffer clean already.)
            if self._last_idx is not None:
                self.grads.zero_blocks_(self._last_idx, self._n_plane_elems)
            else:
                self.grads.flat.zero_()
                self.grads.clean = True
        run_a()
        dist.all_reduce(self._sync, op=dist.ReduceOp.SUM, group=self.group)
        sync_unpack(self._sync, self._pre.acc, self._gacc, self._touched)
        ev = None
        if self._can_mark:
            ev = torch.cuda.Event()
            ev.record()
        run_b()                                    # enqueued before the host waits for the union below
        if self._can_mark:
            cur = torch.cuda.current_stream(self.wl.device)
            with torch.cuda.stream(self._side):
                self._side.wait_event(ev)
                idx = self._touched.nonzero().squeeze(1)        # synchronises the SIDE stream only
            cur.wait_stream(self._side)
            idx.record_stream(cur)
            self.grads.exchange_blocks(idx, self._n_plane_elems, self.group)
            self._last_idx = idx
        elif self.compact:
            self.grads.all_reduce_compact(self._n_plane_elems, self.group)
        else:
            self.grads.all_reduce(self.group)
        self.grads.clean = False                   # (a replayed phase_b cannot flip the flag itself)
        self.grads.assign()
        if run_c is not None:
            run_c()
            if self.optimizer.fused_zero_grad:
                self.grads.clean = True            # the Adam pass zeroed what it consumed (also when it was a graph replay)

    def capture(self, warmup=3):
        """Capture phase_a and phase_b into two hipGraphs sharing one memory pool; step() then replays them."""
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._eager()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(ga, capture_error_mode="thread_local"):
            self.phase_a()
        self.grads.clean = True                    # step() clears the buffer before every replay: no fill inside the graph
        with torch.cuda.graph(gb, pool=ga.pool(), capture_error_mode="thread_local"):
            self.phase_b()
        self.grads.clean = False                   # (the capture ran nothing: the warm-up's gradients are still there)
        gc = None
        if self.optimizer is not None:
            if not self.optimizer.capturable:
                raise RuntimeError("capture() needs an optimiser built with capturable=True")
            gc = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gc, pool=ga.pool(), capture_error_mode="thread_local"):
                self.phase_c()
        torch.cuda.synchronize()
        self._graphs = (ga, gb, gc)

    def step(self):
        if self._graphs is None:
            return self._eager()
        ga, gb, gc = self._graphs
        return self._run(ga.replay, gb.replay, gc.replay if gc is not None else None)
