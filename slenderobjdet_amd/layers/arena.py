"""Flat parameter / gradient arena for MI355X.

All trainable parameters of a model live in ONE contiguous fp32 buffer (``params``), their gradients in a second
(``grads``) and the SGD momentum in a third.  ``nn.Parameter.data`` / ``.grad`` are views into them, so the
reference-style code (``model.parameters()``, ``state_dict``) keeps working, while
  * the optimizer is a single fused kernel launch over the arena (slender_det/solver/build.py:21-25 builds a
    per-tensor torch.optim.SGD),
  * ``zero_grad`` is one memset,
  * the data-parallel gradient exchange (detectron2's DistributedDataParallel) is an RCCL
    all-reduce over large contiguous buckets of ``grads`` launched as soon as the backward pass has finished
    writing them (parameters are laid out in reverse registration order = roughly backward completion order).
Weight gradients are accumulated into ``grads`` directly by the HIP wgrad kernels (no autograd-side copies).
"""
import numpy as np
import torch
import torch.distributed as dist

_ALIGN = 64  # elements; keeps every tensor 256-B aligned


class ParamArena:
    def __init__(self, model, bucket_mb=32.0, bucket_dtype=None, collective=None):
        seen, plist = set(), []
        for name, p in model.named_parameters():
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                plist.append((name, p))
        plist.reverse()
        if not plist:
            raise ValueError("model has no trainable parameters")
        dev = plist[0][1].device
        self.device = dev
        offs, total = [], 0
        for _, p in plist:
            offs.append(total)
            total += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.total = total
        self.params = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(total, dtype=torch.float32, device=dev)
        self.momentum = None
        self.index = {}
        self.names = []
        for (name, p), off in zip(plist, offs):
            n = p.numel()
            view = self.params[off:off + n].view(p.shape)
            view.copy_(p.data)
            p.data = view
            p.grad = self.grads[off:off + n].view(p.shape)
            self.index[id(p)] = (off, n)
            self.names.append((name, off, n))
        self._plist = [p for _, p in plist]
        self.generation = 0          # bumped whenever params change behind torch's back (fused optimizer step)
        self._offs = offs
        self.configure_buckets(bucket_mb, bucket_dtype, collective)
        self._pending = None
        self._uses = {}
        self._counting = True
        self._comm_stream = None
        self._handles = []

    def configure_buckets(self, bucket_mb=32.0, bucket_dtype=None, collective=None):
        """(Re)build the gradient buckets of the data-parallel all-reduce: contiguous runs of the gradient arena of at least ``bucket_mb``
        MB each, in backward completion order.  Call between steps only (bench.py --bucket-mb / --wire, so that a scaling curve can be swept
        without code changes).  (A smaller LAST bucket does not shorten the exposed tail - the communication stream is serial and the
        bucket before it is still on the wire when backward ends: 25.95 vs 25.76 ms per step in the occupancy rehearsal)"""
        if getattr(self, "_pending", None) is not None:
            raise RuntimeError("configure_buckets inside a backward pass")
        offs, total = self._offs, self.total
        self.bucket_elems = int(bucket_mb * (1 << 20) / 4)
        self.buckets = []            # (begin, end)
        b0 = 0
        for (name, off, n), nxt in zip(self.names, offs[1:] + [total]):
            if nxt - b0 >= self.bucket_elems:
                self.buckets.append((b0, nxt))
                b0 = nxt
        if b0 < total:
            self.buckets.append((b0, total))
        self._bucket_of = {}
        for (name, off, n), p in zip(self.names, self._plist):
            for bi, (b, e) in enumerate(self.buckets):
                if b <= off < e:
                    self._bucket_of[id(p)] = bi
        # Wire format of the gradient buckets.  fp32 (default) all-reduces the arena in place: 128 MB per step for FCOS R50, ~1.5 ms of
        # a ring over one 153 GB/s xGMI link pair if nothing overlapped.  bf16 (SOD_GRAD_BUCKET_DTYPE=bf16) sends a
        # rounded copy - half the bytes - and writes the reduced values back as fp32; every rank receives the same reduced values,
        # so replicas stay bit-identical either way (tests/test_ddp_gloo.py).
        import os as _os
        bd = bucket_dtype if bucket_dtype is not None else _os.environ.get("SOD_GRAD_BUCKET_DTYPE", "fp32")
        self.bucket_dtype = {"fp32": torch.float32, "float32": torch.float32, "bf16": torch.bfloat16, "bfloat16": torch.bfloat16}[str(bd).replace("torch.", "")]
        # Shape of the exchange (SURVEY.md section 8 e): "all_reduce" = one RCCL all-reduce per bucket (RCCL picks ring / tree itself);
        # "rs_ag" = reduce-scatter into this rank's 1 / world shard of the bucket followed by an all-gather of the shards back into the
        # bucket - the same sums on every rank (each element is reduced exactly once, by one rank, and its value distributed), two
        # collectives RCCL can schedule over all seven xGMI links.  bench.py --collective / SOD_GRAD_COLLECTIVE sweep it on the 8-GPU node.
        co = collective if collective is not None else getattr(self, "collective", None) or _os.environ.get("SOD_GRAD_COLLECTIVE", "all_reduce")
        if co not in ("all_reduce", "rs_ag"):
            raise ValueError(f"gradient collective {co!r}: expected 'all_reduce' or 'rs_ag'")
        self.collective = co

    # ------------------------------------------------------------------ bf16 compute copies of every trainable conv weight
    def setup_batched_prep(self, model):
        """One launch per step re-derives the bf16 KRSC / CRSK copies (FrozenBN scale folded) of ALL trainable convolution weights
        from the fp32 arena, into two persistent bf16 arenas whose views the modules keep (``w_bf16`` / ``wt_bf16``)."""
        mods = [m for m in model.modules() if getattr(m, "batched_prep_shape", None) is not None and m.weight.requires_grad
                and id(m.weight) in self.index and getattr(m, "cin_pad", None) in (None, m.weight.shape[-1])]
        self._prep_mods, self._prep_gen = mods, -1
        if not mods or self.device.type != "cuda":
            self._prep_mods = []
            return
        tab = np.zeros(len(mods), dtype=np.dtype([("e0", "<i8"), ("src", "<i8"), ("ko", "<i8"), ("co", "<i8"), ("so", "<i8"), ("K", "<i4"),
                                                 ("RS", "<i4"), ("C", "<i4"), ("Cp", "<i4")]))
        e0 = ko = so = 0
        for i, m in enumerate(mods):
            K, RS, C = m.batched_prep_shape()
            n = K * RS * C
            tab[i] = (e0, self.index[id(m.weight)][0], ko, ko, so if m.frozen_bn_scale() else -1, K, RS, C, C)
            e0 += ((K + 63) // 64) * RS * ((C + 63) // 64)          # 64x64 tiles of this weight
            ko += (n + 63) // 64 * 64
            so += K if m.frozen_bn_scale() else 0
        self._prep_total = e0
        self._krsc = torch.zeros(ko, dtype=torch.bfloat16, device=self.device)
        self._crsk = torch.zeros(ko, dtype=torch.bfloat16, device=self.device)
        self._scales = torch.ones(max(so, 1), dtype=torch.float32, device=self.device)
        self._prep_tab = torch.from_numpy(tab.view(np.uint8).copy()).to(self.device)
        for i, m in enumerate(mods):
            K, RS, C = m.batched_prep_shape()
            n, off = K * RS * C, int(tab[i]["ko"])
            m.bind_batched_prep(self._krsc[off:off + n], self._crsk[off:off + n],
                                self._scales[int(tab[i]["so"]):int(tab[i]["so"]) + K] if m.frozen_bn_scale() else None)

    def prep_all(self):
        from .._C import call, ptr, stream_ptr

        call("sod_weight_prep_batched", ptr(self.params), ptr(self._scales), ptr(self._prep_tab), len(self._prep_mods), int(self._prep_total),
             ptr(self._krsc), ptr(self._crsk), stream_ptr())
        self._prep_gen = self.generation
        for m in self._prep_mods:       # every registered weight is fresh now
            m._prep_ver = (m.weight._version, m.weight.data_ptr())

    # ------------------------------------------------------------------ bookkeeping
    def grad_view(self, p):
        off, n = self.index[id(p)]
        g = self.grads[off:off + n].view(p.shape)
        if p.grad is None or p.grad.data_ptr() != g.data_ptr():
            p.grad = g
        return g

    def zero_grad(self):
        if self.device.type == "cuda":
            from . import functional as HF
            HF.drop_held()           # (launches parked by a backward pass that died: not into the fresh arena)
            HF.wgrad_join()          # never clear the arena under weight-gradient kernels still running on the side stream
        self.grads.zero_()

    def bump(self):
        self.generation += 1

    # ------------------------------------------------------------------ data parallel
    def note_use(self, p):
        """Called in forward for every use of a trainable parameter (shared head weights are used once per level).  Forwards that
        build no graph (torch.no_grad(): evaluation hooks between steps, smoke checks) must not count: their uses would never be
        matched by a mark_ready and every bucket of the next step would wait for finish_backward instead of overlapping."""
        if not self._counting:
            return
        self._uses[id(p)] = self._uses.get(id(p), 0) + 1

    def on_forward(self, recording):
        """Forward pre-hook of the model that owns the arena (layers/nn.py:attach_arena).  ``recording`` = grad mode is on and the
        model is training: only such forwards are followed by a backward pass.  (note_use itself runs inside
        autograd.Function.forward, where grad mode always reads off.)  A recording forward starts a fresh count: uses left over by
        a forward whose backward never ran would otherwise keep the buckets from launching early."""
        self._counting = bool(recording)
        if recording:
            self._uses.clear()

    def begin_backward(self):
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self._world = world
        from ..utils import comm as _comm
        if world <= 1 and not _comm.collectives_active():
            self._uses.clear()
            self._pending = None
            return
        counts = [0] * len(self.buckets)
        for pid, c in self._uses.items():
            counts[self._bucket_of[pid]] += c
        self._pending = counts
        self._launched = [False] * len(self.buckets)
        self._handles = []
        if self.device.type == "cuda":
            # the stream backward() is called on: nodes without a stream of their own (the classification tower, the prediction
            # convs, the backbone) produce their parameter gradients there, whichever node's mark_ready ends up launching the bucket
            self._main_stream = torch.cuda.current_stream(self.device)
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream(device=self.device)

    def mark_ready(self, p):
        """Called in backward after the last kernel that adds to ``p.grad`` for this use has been enqueued."""
        if self._pending is None:
            return
        if self.device.type == "cuda":
            from . import functional as HF
            # behind the launches an open wgrad_batch has collected: the bucket must not reach the reducer before they are issued
            HF.batch_or_call(lambda: self._mark_ready_now(p))
        else:
            self._mark_ready_now(p)

    def _mark_ready_now(self, p):
        bi = self._bucket_of[id(p)]
        self._pending[bi] -= 1
        if self._pending[bi] == 0 and not self._launched[bi]:
            self._launch_bucket(bi)

    def _launch_bucket(self, bi):
        b, e = self.buckets[bi]
        view = self.grads[b:e]
        self._launched[bi] = True
        if self.device.type == "cuda":
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            from . import functional as HF
            sides = HF.wgrad_side_streams(self.device)
            with torch.cuda.stream(self._comm_stream):
                self._comm_stream.wait_event(ev)
                # A 32 MB bucket mixes gradients produced on different streams (GroupNorm dgamma / dbeta and bias gradients on the main
                # stream, the box tower's on its own stream, weight gradients on the side stream): the launching node's event covers
                # only ITS stream, so the reduction always waits for the main compute stream too, not only when a weight gradient
                # happened to be ordered behind it through side.wait_stream(main) (SOD_WGRAD_STREAM=0 + SOD_TOWER_STREAMS=1).
                main = getattr(self, "_main_stream", None)
                if main is not None:
                    self._comm_stream.wait_stream(main)
                for side in sides:          # weight gradients of this bucket were enqueued on the wgrad side stream(s)
                    self._comm_stream.wait_stream(side)
                for aux in HF.aux_compute_streams(self.device):   # a bucket may mix parameters whose backward nodes ran on different
                    self._comm_stream.wait_stream(aux)            # streams (the FCOS box tower has its own)
                wire = view if self.bucket_dtype == torch.float32 else view.to(self.bucket_dtype)
                h = self._reduce(wire)
                occ = getattr(self, "rehearsal_occupancy", None)
                if occ is not None:
                    # one-GPU rehearsal (bench.py --rehearsal-occupancy WGS:GBPS): a one-rank all-reduce occupies nothing; stand in for the
                    # N-rank ring's channel kernels - WGS resident workgroups for the time 2 (N-1)/N x bytes take at GBPS of bus bandwidth
                    wgs, gbps, ranks = occ
                    usec = max(1, int(wire.numel() * wire.element_size() * 2.0 * (ranks - 1) / ranks / (gbps * 1e3)))
                    HF.call("sod_debug_occupy", int(wgs), min(usec, 100000), HF.stream_ptr(self._comm_stream))
        else:
            wire = view if self.bucket_dtype == torch.float32 else view.to(self.bucket_dtype)
            h = self._reduce(wire)
        self._handles.append((h, wire, view))

    def _reduce(self, wire):
        """SUM of ``wire`` over the ranks, in place, asynchronously; returns the handle to wait for."""
        world = dist.get_world_size()
        if self.collective == "rs_ag" and wire.numel() % world == 0:
            n = wire.numel() // world
            # the shard is reduced INTO its own place in the bucket: rank r's slice of ``wire`` is both an input of the reduce-scatter and
            # its output (c10d allows the output to alias the rank's own input chunk), so no staging buffer and no copy
            shard = wire[dist.get_rank() * n:(dist.get_rank() + 1) * n]
            h = dist.reduce_scatter_tensor(shard, wire, op=dist.ReduceOp.SUM, async_op=True)
            if dist.get_backend() != "nccl":
                h.wait()          # gloo runs asynchronous work on a thread pool: nothing orders two of them; RCCL's stream does
            return dist.all_gather_into_tensor(wire, shard, async_op=True)       # RCCL: same stream, ordered behind the reduce-scatter
        return dist.all_reduce(wire, op=dist.ReduceOp.SUM, async_op=True)

    def finish_backward(self):
        """Launch the buckets that did not complete during backward (parameters unused this step), then wait for
        every all-reduce.  Gradients hold the SUM over ranks afterwards; the optimizer applies 1/world."""
        if self._pending is None:
            return
        if self.device.type == "cuda":
            from . import functional as HF
            HF.wgrad_join()
        for bi in range(len(self.buckets)):
            if not self._launched[bi]:
                self._launch_bucket(bi)
        probe = getattr(self, "comm_probe", None)      # bench.py: a list that receives (event, event) around the wait for the reducer
        if probe is not None and self.device.type == "cuda":
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        for h, wire, view in self._handles:
            h.wait()
        if self.device.type == "cuda":
            torch.cuda.current_stream().wait_stream(self._comm_stream)
            if probe is not None:
                # the compute stream's own work ends at e0; it resumes at e1, when the last bucket's reduction is done: e1 - e0 is the
                # communication time NOT hidden behind backward ("exposed")
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record()
                probe.append((e0, e1))
        for h, wire, view in self._handles:
            if wire is not view:            # reduced bf16 values back into the fp32 gradient arena (current stream, after the wait)
                view.copy_(wire)
        self._handles = []
        self._pending = None
        self._uses.clear()

    # ------------------------------------------------------------------ optimizer support
    def build_segments(self, param_groups, base_lr):
        """Merge per-parameter (lr, weight_decay) into contiguous arena segments for the fused SGD kernel."""
        info = {}
        for g in param_groups:
            for p in g["params"]:
                info[id(p)] = (g["lr"] / base_lr if base_lr else 1.0, g.get("weight_decay", 0.0))
        segs = []
        for p in self._plist:
            off, n = self.index[id(p)]
            lr_mult, wd = info.get(id(p), (0.0, 0.0))
            end = off + (n + _ALIGN - 1) // _ALIGN * _ALIGN
            if segs and segs[-1][1] == off and segs[-1][2] == lr_mult and segs[-1][3] == wd:
                segs[-1][1] = end
            else:
                segs.append([off, end, lr_mult, wd])
        arr = np.zeros(len(segs), dtype=np.dtype([("b", "<i8"), ("e", "<i8"), ("lr", "<f4"), ("wd", "<f4")]))
        for i, (b, e, lr, wd) in enumerate(segs):
            arr[i] = (b, e, lr, wd)
        dev_arr = torch.from_numpy(arr.view(np.uint8).copy()).to(self.device)
        return dev_arr, len(segs)
