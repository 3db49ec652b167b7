"""Raw (non-autograd) wrappers of the C-ABI HIP ops: allocate outputs with torch, pass pointers + shapes.

Tensor conventions: activations are NHWC ``torch.bfloat16`` CUDA tensors, conv weights are ``[K, R, S, C]``
bf16 ("KRSC"), losses/targets are fp32/int32.  Every function launches on ``torch.cuda.current_stream()``.
"""
import ctypes
import os

import torch

from .. import _C
from .._C import call, ptr, stream_ptr

IOU_TYPES = {"iou": 0, "linear_iou": 1, "giou": 2}

# ---- storage precision of activations / compute weights / gradients.  "bf16" is the product (what bench.py times); "fp32" is the
# validation mode (layers/functional_f32.py, csrc/f32_path.hip): the same layer code with fp32 tensors and untuned fp32 kernels, in which
# tests/test_gpu_parity100.py asserts north_star's 100-iteration loss bound.  SOD_PRECISION=fp32 or set_precision("fp32").
PRECISION = "bf16"
ACT_DTYPE = torch.bfloat16


def set_precision(p):
    """-> the previous precision.  Switch BEFORE building a model: the compute copies of the weights follow the mode they were
    prepared in (HipConv2d.prepare re-derives them when the mode changes)."""
    global PRECISION, ACT_DTYPE
    if p not in ("bf16", "fp32"):
        raise ValueError(f"precision must be 'bf16' or 'fp32', got {p!r}")
    prev, PRECISION = PRECISION, p
    ACT_DTYPE = torch.float32 if p == "fp32" else torch.bfloat16
    return prev


def is_f32():
    return PRECISION == "fp32"


def _precision_dispatch(fn):
    """In the fp32 validation mode, route ``fn`` to its namesake in functional_f32 (same signature)."""
    import functools

    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        if PRECISION == "fp32":
            from . import functional_f32 as F32

            impl = getattr(F32, fn.__name__, None)
            if impl is None:
                raise _C.SlenderHipError(f"{fn.__name__} has no fp32 validation variant")
            return impl(*args, **kwargs)
        return fn(*args, **kwargs)
    return wrapper

# Observer for tests (None in production): called as RELU_TAP(kind, key_ptr, y) with every tensor the bf16 path produces THROUGH a ReLU whose
# output is materialised - kind "conv" (key = address of the weight operand), "gn" (key = gamma's address), "relu" (key 0) - exactly as
# layers/functional_f32.RELU_TAP does for the fp32 validation mode.  tests/test_gpu_parity100.py hands the recorded decisions (y > 0) to the
# CPU oracle.  The fused frozen kernels (stem + max-pool, res2 bottlenecks) keep their ReLUs inside: no gradient flows through them.
RELU_TAP = None

# Optional per-launch timing of the convolution kernels (bench.py roofline): a list that receives
# (kind, algorithmic_flops, start_event, end_event); events are recorded on the stream the kernel is launched on.
PROFILE = None
PROFILE_KINDS = None      # optional set of kinds to time (None = all); bench.py times only the forward conv launches by default


PROFILE_LIB = False       # conv launches are timed by the library's own hipEvent pairs (sod_conv_prof_*), not by torch events


def _prof_begin(stream=None, kind=None):
    if PROFILE is None or (PROFILE_KINDS is not None and kind not in PROFILE_KINDS):
        return None
    if PROFILE_LIB and kind in ("conv_fwd", "conv_dgrad", "conv_wgrad"):
        return "lib"
    e = torch.cuda.Event(enable_timing=True)
    e.record(stream)
    return e


def _nbytes(t):
    return 0 if t is None else t.numel() * t.element_size()


def _prof_end(kind, flops, e0, desc=None, stream=None):
    if e0 is None:
        return
    if e0 == "lib":       # duration, variant and FLOP share are filled in from sod_conv_prof_collect (same call order)
        PROFILE.append((kind, flops, None, None, desc, None))
        return
    e1 = torch.cuda.Event(enable_timing=True)
    e1.record(stream)
    variant = int(_C.load().sod_conv_last_variant()) if kind in ("conv_fwd", "conv_dgrad") else 0
    PROFILE.append((kind, flops, e0, e1, desc, variant))


# Weight gradients have no consumer until the optimizer (or the bucket all-reduce), while the data gradient of the same layer is on
# the critical path of backward.  Inside an autograd backward pass the wgrad launches therefore go to a per-device SIDE stream:
# the hardware interleaves the two queues, so the tail of one kernel's grid is filled by the other's workgroups.  The main stream
# joins the side stream in an end-of-backward callback (and the all-reduce stream waits for it per bucket, arena._launch_bucket).
# SOD_WGRAD_STREAM=0 keeps everything on one stream.
WGRAD_SIDE_STREAM = os.environ.get("SOD_WGRAD_STREAM", "1") != "0"
_side_streams = {}
_side_keep = []
_side_join_queued = False


_aux_streams = {}


def cu_mask_words(lo, hi, total=256):
    """Bit mask (list of 32-bit words) selecting the logical CUs lo .. hi-1 of hipExtStreamCreateWithCUMask's numbering.  On MI355X bit i
    lies on XCD i % 8, so a range whose ends are multiples of 8 takes the same number of CUs from every XCD (the workgroup -> XCD
    round-robin of a launch then still meets equally wide XCDs)."""
    words = [0] * ((total + 31) // 32)
    for i in range(max(0, lo), min(hi, total)):
        words[i // 32] |= 1 << (i % 32)
    return words


def make_stream(device, priority=0, role=None):
    """A side stream of the training step.  ``SOD_CUMASK_<ROLE>=lo:hi`` (role = WGRAD / TOWER / PREFETCH / MAIN) confines it to the
    logical compute units lo .. hi-1 (sod_stream_create_cumask; such a stream has the default priority); without the variable it is a
    plain torch stream of the given HIP priority."""
    spec = os.environ.get(f"SOD_CUMASK_{role}") if role else None
    if not spec:
        return torch.cuda.Stream(device=device, priority=priority)
    lo, hi = (int(v) for v in spec.split(":"))
    words = cu_mask_words(lo, hi)
    arr = (ctypes.c_uint * len(words))(*words)
    out = ctypes.c_void_p()
    with torch.cuda.device(device):
        call("sod_stream_create_cumask", ctypes.cast(arr, ctypes.c_void_p), len(words), ctypes.byref(out))
    return torch.cuda.ExternalStream(out.value, device=device)       # lives for the rest of the process


def register_compute_stream(device, stream):
    """Announce an extra stream that runs backward nodes (e.g. the FCOS box tower's).  Parameter gradients may be produced on it, so
    the bucket reducer (arena._launch_bucket) waits for it in addition to the launching node's stream and the wgrad side stream."""
    lst = _aux_streams.setdefault(device.index, [])
    if all(s.cuda_stream != stream.cuda_stream for s in lst):
        lst.append(stream)


def aux_compute_streams(device):
    return list(_aux_streams.get(device.index, ()))


# (Two or three round-robin side streams measured neutral to worse in rounds 2 and 4 - 614.7 / 606 vs 613.8 img/s, 640.3 vs 642.0 - and left
# the tree in round 5: one in-order side stream.)


def wgrad_side_streams(device):
    """The side streams of ``device`` on which wgrad work may be pending in this backward pass (empty list if none)."""
    return list(_side_streams.get(device.index, ())) if _side_join_queued else []


# Re-pairing the backward phases by roof (round 6; SOD_HOLD_HEAD_WGRAD=0 restores the old order).  The head towers' weight gradients are
# MFMA-bound, and so are the towers' data gradients they used to run beside (8.5 ms phase on three queues); the FPN / backbone backward
# that follows is HBM-bound on two saturated queues.  The tower units therefore PARK their weight-gradient launches (and the mark_ready
# behind them) and the first FPN / backbone backward node releases them, in order, onto the same side stream: they then run beside the
# HBM-bound kernels.  Same launches, same operands, same results; measured on the FCOS R50 step in alternating 60-step runs:
# 660.3 -> 665.3, 653.5 -> 661.7, 655.7 -> 660.8 img/s (+0.8 ... +1.2 %), one-rank RCCL rehearsal 646.5 -> 654.2 with the exposed
# communication unchanged (0.13 ms per step: the head's bucket is launched later, but 7 ms of backward are still ahead of it).  Releasing
# them onto a side stream of their OWN instead measured 14 % SLOWER (555 - 567 img/s at 4, 5 or 6 hardware queues): three streams of
# chip-filling kernels take each other's CUs; profiles/r6_pairing.txt.
HOLD_HEAD_WGRAD = os.environ.get("SOD_HOLD_HEAD_WGRAD", "1") != "0"
_held = []


def hold_or_call(fn):
    """Tower units: run ``fn`` (weight-gradient launch + mark_ready) now, or park it until release_held()."""
    if HOLD_HEAD_WGRAD and _side_join_queued_or_in_backward():
        _held.append(fn)
    else:
        fn()


def _side_join_queued_or_in_backward():
    """Parking is only safe when something will release: inside an autograd backward pass (the end-of-backward callback does, at the
    latest).  Op-level calls outside a backward pass launch at once."""
    global _side_join_queued
    if _side_join_queued:
        return True
    try:
        torch.autograd.Variable._execution_engine.queue_callback(_wgrad_join)
    except RuntimeError:
        return False
    _side_join_queued = True
    return True


def release_held():
    """First FPN / backbone backward node (and the end-of-backward join, at the latest): issue every parked launch, in order."""
    if _held:
        items = list(_held)
        _held.clear()
        for fn in items:
            fn()


def drop_held():
    """A backward pass that ended in an exception may leave parked launches behind: never carry them into the next step."""
    _held.clear()


def _wgrad_join():
    global _side_join_queued
    release_held()
    _side_join_queued = False
    for idx, sides in _side_streams.items():
        main = torch.cuda.current_stream(idx)
        for side in sides:
            main.wait_stream(side)
    _side_keep.clear()      # the main stream now waits for every side-stream reader: the operands may go back to its pool


def wgrad_join():
    """Make the current stream(s) wait for every weight-gradient launch enqueued so far.  The end-of-backward callback does this
    on the normal path; the optimizer and the bucket reducer call it again so that a backward pass that ended in an exception (the
    callback never ran) cannot leave gradients in flight.  A no-op when no side stream exists."""
    if _side_streams:
        _wgrad_join()


# ---- batched hand-over to the side stream.  Every hand-over costs an event RECORD on the producing stream and a WAIT on the side stream;
# the kernel trace prices each at 6-7 us of queue bubble (the next kernel of that queue starts that much later: 70 hand-overs per FCOS
# step = 0.45 ms on the main queue and as much on the side queue, profiles/r5_step_occupancy.txt).  Inside ``with wgrad_batch():`` the
# weight- / bias-gradient launches (and the arena.mark_ready calls that follow them) are collected and issued together when the block ends:
# ONE record + ONE wait for all of them.  A residual block's three or four weight gradients then cost one hand-over instead of four.
import threading as _threading

_batch_tls = _threading.local()


class wgrad_batch:
    def __enter__(self):
        self.outer = getattr(_batch_tls, "items", None)
        if self.outer is None:
            _batch_tls.items = []
        return self

    def __exit__(self, et, ev, tb):
        if self.outer is None:
            items, _batch_tls.items = _batch_tls.items, None
            if et is None and items:
                _batch_tls.flushing, _batch_tls.waited = True, set()
                try:
                    for fn in items:
                        fn()
                finally:
                    _batch_tls.flushing, _batch_tls.waited = False, None
        return False


def _batch_defer(fn):
    """True if ``fn`` was queued into the open batch of this thread (the caller returns without launching)."""
    items = getattr(_batch_tls, "items", None)
    if items is None or not WGRAD_BATCH:
        return False
    items.append(fn)
    return True


def batch_or_call(fn):
    """Run ``fn`` now, or behind the launches already collected in the open batch (arena.mark_ready: a bucket must not be handed to the
    reducer before the launches that fill it have been issued)."""
    if not _batch_defer(fn):
        fn()


# False: one hand-over per weight-gradient launch (as until round 4).  Batching a whole residual block's weight gradients behind its data
# gradients measured SLOWER (623 vs 634 img/s, four alternating 100-step pairs, round 5: the side stream is the one that finishes last, and
# launches that reach it later lengthen the tail of backward by more than the bubbles save), so only launches that were adjacent anyway
# (weight + bias gradient of one convolution, conv1 + shortcut of a block) share a hand-over.
WGRAD_BATCH = True


def _wgrad_stream(device, tensors, key=None):
    """Returns the stream to launch a wgrad on (None = current stream).  ``key`` identifies the gradient buffer: launches that add to
    the same buffer always use the same stream (the slab / deterministic reductions update it with plain read-modify-writes)."""
    global _side_join_queued
    if not WGRAD_SIDE_STREAM or device.type != "cuda":
        return None
    if not _side_join_queued:
        try:
            torch.autograd.Variable._execution_engine.queue_callback(_wgrad_join)
        except RuntimeError:     # not inside a backward pass (op-level calls): nobody would join, stay on the current stream
            return None
        _side_join_queued = True
    sides = _side_streams.get(device.index)
    if sides is None:
        # lowest HIP stream priority (range on MI355X: 1 .. -1): the data-gradient chain is the critical path.  Measured 491.5-492.3
        # (low) / 491.8 (normal) / 483-484 (high) img/s
        sides = _side_streams[device.index] = [make_stream(device, 1, "WGRAD")]
    side = sides[0]
    # one wait per batch flush (the closures of a batch run back to back on this thread: nothing was enqueued on the current stream in
    # between, so the first launch's wait covers the others) - or one per launch outside a batch
    cur = torch.cuda.current_stream(device)
    if getattr(_batch_tls, "flushing", False):
        tag = (side.cuda_stream, cur.cuda_stream)
        if tag not in _batch_tls.waited:
            side.wait_stream(cur)
            _batch_tls.waited.add(tag)
    else:
        side.wait_stream(cur)
    # The operands (dY, X) must outlive the side-stream kernel.  They are kept referenced until the join instead of
    # Tensor.record_stream(): with record_stream the allocator cannot reuse a block until a GPU-side event has completed, and since the
    # host runs several steps ahead of the GPU the pool grew from 10 GB to 52 GB; held references are released in main-stream order
    # right after the join is enqueued, so the pool stays at its single-stream size plus one backward pass of gradients.
    _side_keep.extend(t for t in tensors if t is not None)
    return side
CONV_RELU = 1
CONV_RES_UP2 = 2


def _chk(t, dtype=None, name="tensor"):
    if t is None:
        return
    if not t.is_cuda:
        raise _C.SlenderHipError(f"{name} must be a CUDA/HIP tensor (the HIP ops have no CPU fallback)")
    if not t.is_contiguous():
        raise _C.SlenderHipError(f"{name} must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise _C.SlenderHipError(f"{name} must be {dtype}, got {t.dtype}")


_ws_cache = {}


def reduce_ws(device):
    """Per-(device, stream) scratch for the two-stage reductions."""
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    ws = _ws_cache.get(key)
    if ws is None:
        ws = torch.empty(_C.reduce_workspace_floats(), dtype=torch.float32, device=device)
        _ws_cache[key] = ws
    return ws


def conv_out_size(H, W, R, S, stride, pad, dil):
    Ho = (H + 2 * pad - dil * (R - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (S - 1) - 1) // stride + 1
    return Ho, Wo


def conv2d_fwd(x, w, bias=None, res=None, stride=1, pad=0, dil=1, relu=False, res_up2=False, out_f32=False,
               out=None, y_img_stride=0, x_img_stride=0, x_shape=None, c_real=None, relu_bits=None):
    """x (N,H,W,C) bf16, w (K,R,S,C) bf16 -> y (N,Ho,Wo,K). ``out``/``y_img_stride`` let the result land inside a
    larger (N, L, K) buffer; ``x_shape`` overrides (N,H,W,C) when x is such a view."""
    _chk(x, torch.bfloat16, "x"); _chk(w, torch.bfloat16, "w"); _chk(bias, torch.float32, "bias"); _chk(res, torch.bfloat16, "res")
    N, H, W, C = x_shape if x_shape is not None else x.shape
    K, R, S, Cw = w.shape
    cwin = Cw != C and is_channel_window(C, K, Cw)      # grouped convolution in channel-window mode (layers/nn.py HipGroupedConv2d)
    if Cw != C and not cwin:
        raise _C.SlenderHipError(f"weight channels {Cw} != input channels {C}")
    Ho, Wo = conv_out_size(H, W, R, S, stride, pad, dil)
    if out is None:
        out = torch.empty((N, Ho, Wo, K), dtype=torch.float32 if out_f32 else torch.bfloat16, device=x.device)
    flags = (CONV_RELU if relu else 0) | (CONV_RES_UP2 if res_up2 else 0) | (CONV_CWIN if cwin else 0)
    if cwin and (relu_bits is not None or res is not None):
        raise _C.SlenderHipError("conv2d_fwd: the channel-window mode takes neither a residual nor 1-bit ReLU masks")
    e0 = _prof_begin(None, "conv_fwd")
    if relu_bits is not None:       # also records "stored output > 0" as one bit per element (uint8[numel / 8]) for the backward pass
        _chk(relu_bits, torch.uint8, "relu_bits")
        if out_f32 or res_up2 or x_img_stride or y_img_stride or relu_bits.numel() * 8 != out.numel():
            raise _C.SlenderHipError("conv2d_fwd: relu_bits needs a dense bf16 output of 8 * relu_bits.numel() elements")
        call("sod_conv2d_fwd_bits", ptr(x), ptr(w), ptr(bias), ptr(res), ptr(out), ptr(relu_bits), N, H, W, C, K, R, S, stride, pad, dil, flags,
             stream_ptr())
    else:
        call("sod_conv2d_fwd", ptr(x), ptr(w), ptr(bias), ptr(res), ptr(out), N, H, W, C, K, R, S, stride, pad, dil,
             x_img_stride, y_img_stride, 0, flags, 1 if out_f32 else 0, stream_ptr())
    # c_real / k_real: un-padded channel counts, so that the profile counts ALGORITHMIC work (stem: 3 of its 8 input channels)
    # 8th entry: bytes of the FUSED epilogue operands this launch also moves (shortcut read, 1-bit mask written) - bench.py hbm_frac_fused
    _prof_end("conv_fwd", 2.0 * N * Ho * Wo * K * R * S * (c_real or (Cw if cwin else C)), e0,
              (N, H, W, C, K, R, stride, _nbytes(res) + _nbytes(relu_bits)))
    if relu and RELU_TAP is not None and not x_img_stride and not y_img_stride:
        RELU_TAP("conv", w.data_ptr(), out)
    return out


CONV_CWIN = 4               # slender_hip.h SOD_CONV_CWIN
WGRAD_DIAG = 2              # slender_hip.h SOD_WGRAD_DIAG
CWIN = 128                  # the window = one 128-channel output tile


def is_channel_window(C, K, Cw):
    """Weights whose contraction width is the 128-channel window instead of the input's C channels: C == K, multiples of 128."""
    return Cw == CWIN and C == K and C > CWIN and C % CWIN == 0 and not is_f32()


def conv2d_dgrad(dy, wt, x_hw, stride=1, pad=0, dil=1, accum=None, relu_mask=None, dy_img_stride=0, dy_shape=None, out=None, relu_bits=None,
                 accum_even=False):
    """dy (N,Ho,Wo,K) bf16, wt (C,R,S,K) bf16 (transposed weights) -> dx (N,H,W,C) bf16."""
    _chk(dy, torch.bfloat16, "dy"); _chk(wt, torch.bfloat16, "wt"); _chk(accum, torch.bfloat16, "accum"); _chk(relu_mask, torch.bfloat16, "relu_mask")
    N = dy_shape[0] if dy_shape is not None else dy.shape[0]
    C, R, S, K = wt.shape
    H, W = x_hw
    if out is None:
        out = torch.empty((N, H, W, C), dtype=torch.bfloat16, device=dy.device)
    if dy_shape is None and dy.shape[-1] != K and is_channel_window(C, dy.shape[-1], K):      # grouped convolution, channel-window mode
        if accum is not None or relu_bits is not None or dy_img_stride:
            raise _C.SlenderHipError("conv2d_dgrad: the channel-window mode takes a ReLU mask tensor only")
        e0 = _prof_begin(None, "conv_dgrad")
        call("sod_conv2d_dgrad_cwin", ptr(dy), ptr(wt), ptr(relu_mask), ptr(out), N, H, W, C, C, R, S, stride, pad, dil, stream_ptr())
        Ho, Wo = conv_out_size(H, W, R, S, stride, pad, dil)
        _prof_end("conv_dgrad", 2.0 * N * Ho * Wo * C * R * S * K, e0, (N, H, W, C, C, R, stride))
        return out
    e0 = _prof_begin(None, "conv_dgrad")
    if relu_bits is not None:       # the ReLU mask of dx's tensor as one bit per element (conv2d_fwd(..., relu_bits=...))
        _chk(relu_bits, torch.uint8, "relu_bits")
        if relu_mask is not None or dy_img_stride or relu_bits.numel() * 8 != out.numel():
            raise _C.SlenderHipError("conv2d_dgrad: relu_bits replaces relu_mask and needs 8 * relu_bits.numel() == dx.numel()")
        if accum_even and (accum is None or tuple(accum.shape) != (N, H // 2, W // 2, C) or H % 2 or W % 2):
            raise _C.SlenderHipError("conv2d_dgrad: accum_even needs accum of shape (N, H/2, W/2, C) with even H, W")
        call("sod_conv2d_dgrad_bits", ptr(dy), ptr(wt), ptr(accum), 1 if accum_even else 0, ptr(relu_bits), ptr(out), N, H, W, C, K, R, S, stride,
             pad, dil, stream_ptr())
    else:
        if accum_even:
            raise _C.SlenderHipError("conv2d_dgrad: accum_even is available with relu_bits only")
        call("sod_conv2d_dgrad", ptr(dy), ptr(wt), ptr(accum), ptr(relu_mask), ptr(out), N, H, W, C, K, R, S, stride, pad, dil,
             dy_img_stride, 0, stream_ptr())
    Ho, Wo = conv_out_size(H, W, R, S, stride, pad, dil)
    _prof_end("conv_dgrad", 2.0 * N * Ho * Wo * K * R * S * C, e0,
              (N, H, W, C, K, R, stride, _nbytes(accum) + _nbytes(relu_mask) + _nbytes(relu_bits)))
    return out


# Deterministic reductions (weight-gradient pixel splits and GroupNorm statistics summed in a fixed order, no float atomics): the
# results are bit-identical from run to run.  Off by default (the atomic forms are a little faster on the few-tile shapes);
# SOD_DETERMINISTIC=1 or ``functional.DETERMINISTIC = True`` turns it on (the loss-parity tests do).
DETERMINISTIC = os.environ.get("SOD_DETERMINISTIC", "0") == "1"
WGRAD_DETERMINISTIC = 1     # slender_hip.h SOD_WGRAD_DETERMINISTIC

_wgrad_ws = {}


def wgrad_workspace(device, stream=None):
    """Per-(device, stream) scratch for the weight-gradient slabs (caller-owned, used in stream order by one launch at a time)."""
    key = (device.index, (stream or torch.cuda.current_stream(device)).cuda_stream)
    ws = _wgrad_ws.get(key)
    if ws is None:
        ws = _wgrad_ws[key] = torch.empty(int(_C.load().sod_conv2d_wgrad_workspace_bytes()), dtype=torch.uint8, device=device)
    return ws


def conv2d_wgrad(dy, x, dw, R, S, stride=1, pad=0, dil=1, dy_img_stride=0, x_img_stride=0, K=None, x_shape=None, splits=0, qscale=None,
                 k_real=None, c_real=None):
    """Accumulates into dw (K,R,S,C) fp32."""
    _chk(dy, torch.bfloat16, "dy"); _chk(x, torch.bfloat16, "x"); _chk(dw, torch.float32, "dw")
    if _batch_defer(lambda: conv2d_wgrad(dy, x, dw, R, S, stride, pad, dil, dy_img_stride, x_img_stride, K, x_shape, splits, qscale, k_real, c_real)):
        return dw
    N, H, W, C = x_shape if x_shape is not None else x.shape
    if K is None:
        K = dy.shape[-1]
    side = _wgrad_stream(dw.device, (dy, x, qscale), dw.data_ptr())      # qscale is read by the kernel's last stage: keep it alive too
    ws = wgrad_workspace(dw.device, side)
    diag = dw.shape[-1] != C and is_channel_window(C, K, dw.shape[-1])      # grouped convolution: dw is (K, R, S, 128), diagonal tiles only
    e0 = _prof_begin(side, "conv_wgrad")
    call("sod_conv2d_wgrad", ptr(dy), ptr(x), ptr(dw), ptr(qscale), N, H, W, C, K, R, S, stride, pad, dil, dy_img_stride, x_img_stride,
         splits, (WGRAD_DETERMINISTIC if DETERMINISTIC else 0) | (WGRAD_DIAG if diag else 0), ptr(ws), ws.numel(), stream_ptr(side))
    Ho, Wo = conv_out_size(H, W, R, S, stride, pad, dil)
    _prof_end("conv_wgrad", 2.0 * N * Ho * Wo * (k_real or K) * R * S * (c_real or (CWIN if diag else C)), e0, (N, H, W, C, K, R, stride), side)
    return dw


def _ptr_arr(ts):
    return (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])


def conv2d_fwd_ml(xs, w, bias=None, stride=1, pad=0, dil=1, relu=False, out_f32=False, outs=None, y_img_stride=0, k_real=None):
    """One launch over several (N,Hl,Wl,C) tensors that share the weights (FPN levels). Returns the list of outputs."""
    _chk(w, torch.bfloat16, "w"); _chk(bias, torch.float32, "bias")
    for x in xs:
        _chk(x, torch.bfloat16, "x")
    N, C = xs[0].shape[0], xs[0].shape[3]
    K, R, S, Cw = w.shape
    if Cw != C:
        raise _C.SlenderHipError(f"weight channels {Cw} != input channels {C}")
    hs, ws = [x.shape[1] for x in xs], [x.shape[2] for x in xs]
    if outs is None:
        outs = [torch.empty((N,) + conv_out_size(h, wd, R, S, stride, pad, dil) + (K,), dtype=torch.float32 if out_f32 else torch.bfloat16,
                            device=xs[0].device) for h, wd in zip(hs, ws)]
    e0 = _prof_begin(None, "conv_fwd")
    call("sod_conv2d_fwd_ml", len(xs), _ptr_arr(xs), ptr(w), ptr(bias), _ptr_arr(outs), N, _int_arr(hs), _int_arr(ws), C, K, R, S,
         stride, pad, dil, y_img_stride, CONV_RELU if relu else 0, 1 if out_f32 else 0, stream_ptr())
    fl = sum(2.0 * N * ho * wo * (k_real or K) * R * S * C for ho, wo in (conv_out_size(h, wd, R, S, stride, pad, dil) for h, wd in zip(hs, ws)))
    _prof_end("conv_fwd", fl, e0, ("ml", N, tuple(hs), C, K, R, stride, tuple(ws)))
    if relu and RELU_TAP is not None and not y_img_stride:
        for o in outs:
            RELU_TAP("conv", w.data_ptr(), o)
    return outs


def conv_gn_fwd_ml(xs, w, bias, gamma, beta, G, eps=1e-5, relu=True, pad=1):
    """[conv (stride 1, bias) -> GroupNorm(G) -> ReLU] over several levels that share the weights, with the norm's statistics gathered
    in the conv epilogue (sod_conv2d_fwd_ml_gnsum) instead of a separate pass over the conv output.  Returns (conv outputs, norm
    outputs, stats (nl,N,G,2) = mean / rstd)."""
    _chk(w, torch.bfloat16, "w"); _chk(bias, torch.float32, "bias"); _chk(gamma, torch.float32, "gamma"); _chk(beta, torch.float32, "beta")
    for x in xs:
        _chk(x, torch.bfloat16, "x")
    N, C = xs[0].shape[0], xs[0].shape[3]
    K, R, S, Cw = w.shape
    if Cw != C or K != 8 * G:
        raise _C.SlenderHipError("conv_gn_fwd_ml: needs matching channels and 8 channels per group")
    hs, ws = [x.shape[1] for x in xs], [x.shape[2] for x in xs]
    outs = [torch.empty((N,) + conv_out_size(h, wd, R, S, 1, pad, 1) + (K,), dtype=torch.bfloat16, device=xs[0].device) for h, wd in zip(hs, ws)]
    stats = torch.empty((len(xs), N, G, 2), dtype=torch.float32, device=xs[0].device)
    e0 = _prof_begin(None, "conv_fwd")
    call("sod_conv2d_fwd_ml_gnsum", len(xs), _ptr_arr(xs), ptr(w), ptr(bias), _ptr_arr(outs), N, _int_arr(hs), _int_arr(ws), C, K, R, S,
         1, pad, 1, 0, 0, ptr(stats), G, stream_ptr())
    fl = sum(2.0 * N * o.shape[1] * o.shape[2] * K * R * S * C for o in outs)
    _prof_end("conv_fwd", fl, e0, ("ml", N, tuple(hs), C, K, R, 1, tuple(ws)))
    ys = [torch.empty_like(o) for o in outs]
    hw = [o.shape[1] * o.shape[2] for o in outs]
    call("sod_groupnorm_apply_ml", len(outs), _ptr_arr(outs), ptr(gamma), ptr(beta), _ptr_arr(ys), ptr(stats), N, ctypes.cast(_int_arr(hw), ctypes.c_void_p),
         K, G, eps, 1 if relu else 0, stream_ptr())
    if relu and RELU_TAP is not None:
        for y in ys:
            RELU_TAP("gn", gamma.data_ptr(), y)
    return outs, ys, stats


def conv2d_dgrad_ml(dys, wt, x_hws, stride=1, pad=0, dil=1, dy_img_stride=0, N=None, k_real=None, relu_masks=None, accums=None, k_pitch=None):
    """dys: per-level dY tensors (or 1-D views into a concatenated buffer with dy_img_stride); returns per-level dX.  ``relu_masks``:
    the post-ReLU tensors dX is the gradient of - the ReLU backward is then applied in the epilogue (dX = mask > 0 ? dX : 0).
    ``accums``: per-level bf16 tensors of dX's shapes added in the epilogue (another consumer's gradient of the same tensors); with
    ``relu_masks`` as well the mask applies to the sum.
    ``k_pitch``: the dY rows hold k_pitch channels and ``wt`` is (C, R, S, Kp) with Kp >= k_pitch a multiple of 64 and ZERO columns from
    k_pitch on (sod_conv2d_dgrad_ml_kpitch: the linear K loops for contractions that are no multiple of 64 channels per tap)."""
    _chk(wt, torch.bfloat16, "wt")
    C, R, S, K = wt.shape
    if N is None:
        N = dys[0].shape[0]
    dev = dys[0].device
    outs = [torch.empty((N, h, w, C), dtype=torch.bfloat16, device=dev) for h, w in x_hws]
    e0 = _prof_begin(None, "conv_dgrad")
    if k_pitch is not None and k_pitch != K:
        if relu_masks is not None or accums is not None or stride != 1:
            raise _C.SlenderHipError("conv2d_dgrad_ml: k_pitch comes without masks / accums, stride 1")
        call("sod_conv2d_dgrad_ml_kpitch", len(dys), _ptr_arr(dys), ptr(wt), _ptr_arr(outs), N, _int_arr([h for h, _ in x_hws]),
             _int_arr([w for _, w in x_hws]), C, K, int(k_pitch), R, S, pad, dil, dy_img_stride, stream_ptr())
    elif accums is not None:
        if len(accums) != len(outs) or (relu_masks is not None and len(relu_masks) != len(outs)):
            raise _C.SlenderHipError("conv2d_dgrad_ml: accums (and relu_masks) come one per level")
        for t, o in zip(list(accums) + list(relu_masks or ()), list(outs) * 2):
            _chk(t, torch.bfloat16, "accum / relu_mask")
            if tuple(t.shape) != tuple(o.shape) or not t.is_contiguous():
                raise _C.SlenderHipError("conv2d_dgrad_ml: accums / relu_masks must be contiguous tensors of the data gradients' shapes")
        call("sod_conv2d_dgrad_ml_accum", len(dys), _ptr_arr(dys), ptr(wt), _ptr_arr(accums), _ptr_arr(relu_masks) if relu_masks is not None else None,
             _ptr_arr(outs), N, _int_arr([h for h, _ in x_hws]), _int_arr([w for _, w in x_hws]), C, K, R, S, stride, pad, dil, dy_img_stride, stream_ptr())
    elif relu_masks is not None:
        for t, o in zip(relu_masks, outs):
            _chk(t, torch.bfloat16, "relu_mask")
            if tuple(t.shape) != tuple(o.shape):
                raise _C.SlenderHipError("conv2d_dgrad_ml: relu_masks must have the data gradients' shapes")
        call("sod_conv2d_dgrad_ml_mask", len(dys), _ptr_arr(dys), ptr(wt), _ptr_arr(relu_masks), _ptr_arr(outs), N, _int_arr([h for h, _ in x_hws]),
             _int_arr([w for _, w in x_hws]), C, K, R, S, stride, pad, dil, dy_img_stride, stream_ptr())
    else:
        call("sod_conv2d_dgrad_ml", len(dys), _ptr_arr(dys), ptr(wt), _ptr_arr(outs), N, _int_arr([h for h, _ in x_hws]), _int_arr([w for _, w in x_hws]),
             C, K, R, S, stride, pad, dil, dy_img_stride, stream_ptr())
    fl = sum(2.0 * N * ho * wo * (k_real or K) * R * S * C for ho, wo in (conv_out_size(h, w, R, S, stride, pad, dil) for h, w in x_hws))
    _prof_end("conv_dgrad", fl, e0, ("ml", N, tuple(h for h, _ in x_hws), C, K, R, stride, tuple(w for _, w in x_hws)))
    return outs


def conv2d_wgrad_ml(dys, xs, dw, R, S, stride=1, pad=0, dil=1, dy_img_stride=0, K=None, splits=0, qscale=None, k_real=None):
    """Accumulates the weight gradient over all levels in one launch."""
    _chk(dw, torch.float32, "dw")
    if _batch_defer(lambda: conv2d_wgrad_ml(dys, xs, dw, R, S, stride, pad, dil, dy_img_stride, K, splits, qscale, k_real)):
        return dw
    N, C = xs[0].shape[0], xs[0].shape[3]
    if K is None:
        K = dys[0].shape[-1]
    hs, ws_ = [x.shape[1] for x in xs], [x.shape[2] for x in xs]
    side = _wgrad_stream(dw.device, list(dys) + list(xs) + [qscale], dw.data_ptr())
    ws = wgrad_workspace(dw.device, side)
    e0 = _prof_begin(side, "conv_wgrad")
    call("sod_conv2d_wgrad_ml", len(xs), _ptr_arr(dys), _ptr_arr(xs), ptr(dw), ptr(qscale), N, _int_arr(hs), _int_arr(ws_), C, K, R, S,
         stride, pad, dil, dy_img_stride, splits, WGRAD_DETERMINISTIC if DETERMINISTIC else 0, ptr(ws), ws.numel(), stream_ptr(side))
    fl = sum(2.0 * N * ho * wo * (k_real or K) * R * S * C for ho, wo in (conv_out_size(h, w, R, S, stride, pad, dil) for h, w in zip(hs, ws_)))
    _prof_end("conv_wgrad", fl, e0, ("ml", N, tuple(hs), C, K, R, stride, tuple(ws_)), side)
    return dw


def weight_prep(w_master, scale=None, want_krsc=True, want_crsk=True, cpad=None):
    """fp32 (K,R,S,C) -> bf16 (K,R,S,Cpad) [* scale[k]] and bf16 (C,R,S,K)."""
    _chk(w_master, torch.float32, "w"); _chk(scale, torch.float32, "scale")
    K, R, S, C = w_master.shape
    cpad = C if cpad is None else cpad
    wk = (torch.zeros if cpad != C else torch.empty)((K, R, S, cpad), dtype=torch.bfloat16, device=w_master.device) if want_krsc else None
    wc = torch.empty((C, R, S, K), dtype=torch.bfloat16, device=w_master.device) if want_crsk else None
    call("sod_weight_prep", ptr(w_master), ptr(scale), ptr(wk), ptr(wc), K, R * S, C, cpad, stream_ptr())
    return wk, wc


def _det_ws(device, stream=None):
    """(pointer, bytes) of the deterministic-mode scratch for GroupNorm / bias-gradient reductions: the stream's weight-gradient
    workspace (used in stream order, one launch at a time), or (NULL, 0) when deterministic mode is off."""
    if not DETERMINISTIC:
        return None, 0
    ws = wgrad_workspace(device, stream)
    return ptr(ws), ws.numel()


def groupnorm_fwd(x, gamma, beta, G, eps=1e-5, relu=False):
    _chk(x, torch.bfloat16, "x"); _chk(gamma, torch.float32, "gamma"); _chk(beta, torch.float32, "beta")
    N, C = x.shape[0], x.shape[-1]
    HW = x.numel() // (N * C)
    y = torch.empty_like(x)
    stats = torch.empty((N, G, 2), dtype=torch.float32, device=x.device)
    call("sod_groupnorm_fwd", ptr(x), ptr(gamma), ptr(beta), ptr(y), ptr(stats), N, HW, C, G, 0, eps, 1 if relu else 0, *_det_ws(x.device), stream_ptr())
    if relu and RELU_TAP is not None:
        RELU_TAP("gn", gamma.data_ptr(), y)
    return y, stats


def groupnorm_bwd(dy, x, gamma, beta, stats, G, dgamma, dbeta, relu=False, dxsum=None):
    """Returns dx; accumulates dgamma/dbeta (fp32) in place, and (optional) the per-channel sum of dx into ``dxsum``."""
    _chk(dy, torch.bfloat16, "dy"); _chk(x, torch.bfloat16, "x"); _chk(dgamma, torch.float32, "dgamma"); _chk(dbeta, torch.float32, "dbeta")
    N, C = x.shape[0], x.shape[-1]
    HW = x.numel() // (N * C)
    dx = torch.empty_like(x)
    red = torch.empty((N, G, 2), dtype=torch.float32, device=x.device)
    call("sod_groupnorm_bwd", ptr(dy), ptr(x), ptr(gamma), ptr(beta), ptr(stats), ptr(dx), ptr(dgamma), ptr(dbeta), ptr(dxsum), ptr(red),
         N, HW, C, G, 0, 1 if relu else 0, *_det_ws(x.device), stream_ptr())
    return dx


def groupnorm_fwd_ml(xs, gamma, beta, G, eps=1e-5, relu=False):
    """GroupNorm(+ReLU) of several (N,Hl,Wl,C) levels sharing gamma/beta in one launch per pass. Returns (ys, stats (nl,N,G,2))."""
    _chk(gamma, torch.float32, "gamma"); _chk(beta, torch.float32, "beta")
    for x in xs:
        _chk(x, torch.bfloat16, "x")
    N, C = xs[0].shape[0], xs[0].shape[-1]
    hw = [x.numel() // (N * C) for x in xs]
    ys = [torch.empty_like(x) for x in xs]
    stats = torch.empty((len(xs), N, G, 2), dtype=torch.float32, device=xs[0].device)
    call("sod_groupnorm_fwd_ml", len(xs), _ptr_arr(xs), ptr(gamma), ptr(beta), _ptr_arr(ys), ptr(stats), N, ctypes.cast(_int_arr(hw), ctypes.c_void_p),
         C, G, eps, 1 if relu else 0, *_det_ws(xs[0].device), stream_ptr())
    if relu and RELU_TAP is not None:
        for y in ys:
            RELU_TAP("gn", gamma.data_ptr(), y)
    return ys, stats


def groupnorm_bwd_ml(dys, xs, gamma, beta, stats, G, dgamma, dbeta, relu=False, dxsum=None):
    """Backward of groupnorm_fwd_ml: returns the list of dx; accumulates dgamma / dbeta (and dxsum) in place."""
    _chk(dgamma, torch.float32, "dgamma"); _chk(dbeta, torch.float32, "dbeta"); _chk(stats, torch.float32, "stats")
    for t in list(dys) + list(xs):
        _chk(t, torch.bfloat16, "dy/x")
    N, C = xs[0].shape[0], xs[0].shape[-1]
    hw = [x.numel() // (N * C) for x in xs]
    dxs = [torch.empty_like(x) for x in xs]
    red = torch.empty((len(xs), N, G, 2), dtype=torch.float32, device=xs[0].device)
    call("sod_groupnorm_bwd_ml", len(xs), _ptr_arr(dys), _ptr_arr(xs), ptr(gamma), ptr(beta), ptr(stats), _ptr_arr(dxs), ptr(dgamma), ptr(dbeta),
         ptr(dxsum), ptr(red), N, ctypes.cast(_int_arr(hw), ctypes.c_void_p), C, G, 1 if relu else 0, *_det_ws(xs[0].device), stream_ptr())
    return dxs


def relu_fwd(x):
    _chk(x, torch.bfloat16, "x")
    y = torch.empty_like(x)
    call("sod_relu_fwd", ptr(x), ptr(y), x.numel(), stream_ptr())
    if RELU_TAP is not None:
        RELU_TAP("relu", 0, y)
    return y


def relu_bwd(dy, y):
    _chk(dy, torch.bfloat16, "dy"); _chk(y, torch.bfloat16, "y")
    dx = torch.empty_like(dy)
    call("sod_relu_bwd", ptr(dy), ptr(y), ptr(dx), dy.numel(), stream_ptr())
    return dx


def add_bf16(a, b):
    _chk(a, torch.bfloat16, "a"); _chk(b, torch.bfloat16, "b")
    o = torch.empty_like(a)
    call("sod_add_bf16", ptr(a), ptr(b), ptr(o), a.numel(), stream_ptr())
    return o


def bias_grad(dy, dbias, N, HW, C, img_stride=0, scale_num=None, scale_den=None, den_mul=1.0, den_min=1.0):
    """dbias += [scale_num / max(scale_den * den_mul, den_min), device scalars] * per-channel sum of dy."""
    _chk(dy, torch.bfloat16, "dy"); _chk(dbias, torch.float32, "dbias")
    _chk(scale_num, torch.float32, "scale_num"); _chk(scale_den, torch.float32, "scale_den")
    if _batch_defer(lambda: bias_grad(dy, dbias, N, HW, C, img_stride, scale_num, scale_den, den_mul, den_min)):
        return dbias
    side = _wgrad_stream(dbias.device, (dy, scale_num, scale_den), dbias.data_ptr())      # like the weight gradient: only the optimizer / all-reduce consumes it
    if scale_num is None and scale_den is None:
        call("sod_bias_grad", ptr(dy), ptr(dbias), N, HW, C, img_stride, *_det_ws(dbias.device, side), stream_ptr(side))
    else:
        call("sod_bias_grad_scaled", ptr(dy), ptr(dbias), ptr(scale_num), ptr(scale_den), float(den_mul), float(den_min), N, HW, C, img_stride,
             *_det_ws(dbias.device, side), stream_ptr(side))
    return dbias


def bias_grad_ml(dys, dbias):
    """Bias gradient of a conv shared by several levels: dbias += per-channel sum of every dense (N, H_l, W_l, C) tensor in ``dys``, one
    launch (deterministic mode: one fixed-order launch per level)."""
    _chk(dbias, torch.float32, "dbias")
    for t in dys:
        _chk(t, torch.bfloat16, "dy")
    N, C = dys[0].shape[0], dys[0].shape[-1]
    if _batch_defer(lambda: bias_grad_ml(dys, dbias)):
        return dbias
    if DETERMINISTIC or len(dys) > 6:
        for g in dys:
            bias_grad(g, dbias, N, g.numel() // (N * C), C)
        return dbias
    side = _wgrad_stream(dbias.device, list(dys), dbias.data_ptr())
    hw = [g.numel() // (N * C) for g in dys]
    call("sod_bias_grad_ml", len(dys), _ptr_arr(dys), ptr(dbias), N, ctypes.cast(_int_arr(hw), ctypes.c_void_p), C, stream_ptr(side))
    return dbias


def maxpool3x3s2(x):
    _chk(x, torch.bfloat16, "x")
    N, H, W, C = x.shape
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    y = torch.empty((N, Ho, Wo, C), dtype=torch.bfloat16, device=x.device)
    call("sod_maxpool3x3s2", ptr(x), ptr(y), N, H, W, C, stream_ptr())
    return y


def upsample2x_bwd(g):
    _chk(g, torch.bfloat16, "g")
    N, H, W, C = g.shape
    d = torch.empty((N, H // 2, W // 2, C), dtype=torch.bfloat16, device=g.device)
    call("sod_upsample2x_bwd", ptr(g), ptr(d), N, H // 2, W // 2, C, stream_ptr())
    return d


def preprocess_image(img, out, mean, std):
    """img (C,H,W) uint8/float32 CUDA -> out (Hp,Wp,8) bf16 view of the batch buffer."""
    if img.dtype not in (torch.uint8, torch.float32):
        raise _C.SlenderHipError("image must be uint8 or float32")
    _chk(img, None, "image"); _chk(out, torch.bfloat16, "out")
    C, H, W = img.shape
    Hp, Wp, Cp = out.shape
    m = (ctypes.c_float * 3)(*[float(v) for v in mean])
    s = (ctypes.c_float * 3)(*[float(v) for v in std])
    call("sod_preprocess_image", ptr(img), 1 if img.dtype == torch.uint8 else 0, C, H, W, ptr(out), Hp, Wp, Cp,
         ctypes.cast(m, ctypes.c_void_p), ctypes.cast(s, ctypes.c_void_p), stream_ptr())
    return out


def preprocess_batch(imgs, out, mean, std):
    """imgs: list of (C,H,W) uint8/float32 CUDA tensors of one dtype -> out (n,Hp,Wp,8) bf16, one launch."""
    dt = imgs[0].dtype
    if dt not in (torch.uint8, torch.float32) or any(i.dtype != dt for i in imgs) or len(imgs) > 64:
        for i, im in enumerate(imgs):
            preprocess_image(im.contiguous() if im.dtype in (torch.uint8, torch.float32) else im.float().contiguous(), out[i], mean, std)
        return out
    _chk(out, torch.bfloat16, "out")
    imgs = [i.contiguous() for i in imgs]
    for i in imgs:
        _chk(i, None, "image")
    n, Hp, Wp, Cp = out.shape
    C = imgs[0].shape[0]
    m = (ctypes.c_float * 3)(*[float(v) for v in mean])
    s = (ctypes.c_float * 3)(*[float(v) for v in std])
    call("sod_preprocess_batch", len(imgs), _ptr_arr(imgs), 1 if dt == torch.uint8 else 0, C, _int_arr([i.shape[1] for i in imgs]),
         _int_arr([i.shape[2] for i in imgs]), ptr(out), Hp, Wp, Cp, ctypes.cast(m, ctypes.c_void_p), ctypes.cast(s, ctypes.c_void_p), stream_ptr())
    return out


class RawImageBatch:
    """A batch of decoded uint8 (3, H, W) images that has NOT been normalised / padded yet.  ``preprocess_image`` hands this to the
    backbone instead of the NHWC(8) tensor when the stem can consume the raw pixels (``stem_fused``); anything else calls
    ``materialize()`` and gets exactly the tensor ``preprocess_batch`` builds."""

    def __init__(self, imgs, sizes, padded_hw, mean, std):
        self.imgs, self.sizes, self.padded_hw, self.mean, self.std = imgs, sizes, padded_hw, mean, std
        self._tensor = None

    @property
    def shape(self):
        return (len(self.imgs), self.padded_hw[0], self.padded_hw[1], 8)

    @property
    def device(self):
        return self.imgs[0].device

    def materialize(self):
        if self._tensor is None:
            Hp, Wp = self.padded_hw
            batch = torch.empty((len(self.imgs), Hp, Wp, 8), dtype=torch.bfloat16, device=self.device)
            self._tensor = preprocess_batch(self.imgs, batch, self.mean, self.std)
        return self._tensor


def stem_pack_weights(w_bf16):
    """(64, 7, 7, Cpad) bf16 KRSC compute copy (FrozenBN scale folded) -> the [64][24][8] layout of sod_stem_fused."""
    K = w_bf16.shape[0]
    w = w_bf16[..., :3].permute(0, 3, 1, 2).contiguous()                     # (K, 3, 7, 7) = [k][c][r][s]
    out = torch.zeros((K, 24, 8), dtype=torch.bfloat16, device=w_bf16.device)
    out[:, :21, :7] = w.reshape(K, 21, 7)
    return out.contiguous()


def stem_fused(raw, w_packed, bias):
    """raw: RawImageBatch of uint8 images -> (N, Hp/4, Wp/4, 64) bf16 = maxpool(relu(frozen_bn(conv7x7s2(normalise(images)))))."""
    _chk(w_packed, torch.bfloat16, "w_packed"); _chk(bias, torch.float32, "bias")
    n = len(raw.imgs)
    Hp, Wp = raw.padded_hw
    if n > 64 or Hp % 4 or Wp % 4 or tuple(w_packed.shape) != (64, 24, 8):
        raise _C.SlenderHipError("stem_fused: unsupported batch / weight shape")
    for im in raw.imgs:
        _chk(im, torch.uint8, "image")
        if im.dim() != 3 or im.shape[0] != 3:
            raise _C.SlenderHipError("stem_fused: images must be (3, H, W) uint8")
    out = torch.empty((n, Hp // 4, Wp // 4, 64), dtype=torch.bfloat16, device=raw.device)
    prof = PROFILE is not None and (PROFILE_KINDS is None or "conv_fwd" in PROFILE_KINDS)
    if prof:        # not a library conv dispatch: timed with a torch event pair on the launch stream (bench.py roofline)
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
    m = (ctypes.c_float * 3)(*[float(v) for v in raw.mean])
    s_ = (ctypes.c_float * 3)(*[float(v) for v in raw.std])
    call("sod_stem_fused", n, _ptr_arr(raw.imgs), _int_arr([i.shape[1] for i in raw.imgs]), _int_arr([i.shape[2] for i in raw.imgs]), ptr(w_packed), ptr(bias),
         ptr(out), Hp, Wp, ctypes.cast(m, ctypes.c_void_p), ctypes.cast(s_, ctypes.c_void_p), stream_ptr())
    if prof:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        # algorithmic work of the stem conv: 2 * N * (Hp/2) * (Wp/2) * 64 * 7 * 7 * 3 (the pool halo recomputation is not counted)
        PROFILE.append(("conv_fwd", 2.0 * n * (Hp // 2) * (Wp // 2) * 64 * 49 * 3, e0, e1, (n, Hp, Wp, 3, 64, 7, 2), -7))
    return out


def bottleneck_frozen_fwd(x, w1, b1, w2, b2, w3, b3, wsc=None):
    """One frozen ResNet bottleneck block (64 bottleneck / 256 output channels, stride 1) in one kernel (csrc/bottleneck_fused.hip):
    x (N,H,W,Cin) bf16; w1 (64,1,1,Cin), w2 (64,3,3,64), w3 (256,1,1,64), wsc (256,1,1,Cin) | None bf16 KRSC with the FrozenBN scale
    folded; b1 / b2 / b3 fp32 folded shifts (b3 includes the projection shortcut's).  Returns (N,H,W,256) bf16."""
    for t, nm in ((x, "x"), (w1, "w1"), (w2, "w2"), (w3, "w3"), (wsc, "wsc")):
        _chk(t, torch.bfloat16, nm)
    for t, nm in ((b1, "b1"), (b2, "b2"), (b3, "b3")):
        _chk(t, torch.float32, nm)
    N, H, W, Cin = x.shape
    if (tuple(w1.shape) != (64, 1, 1, Cin) or tuple(w2.shape) != (64, 3, 3, 64) or tuple(w3.shape) != (256, 1, 1, 64)
            or (wsc is not None and tuple(wsc.shape) != (256, 1, 1, Cin)) or b1.numel() != 64 or b2.numel() != 64 or b3.numel() != 256):
        raise _C.SlenderHipError("bottleneck_frozen_fwd: unsupported block shape")
    out = torch.empty((N, H, W, 256), dtype=torch.bfloat16, device=x.device)
    prof = PROFILE is not None and (PROFILE_KINDS is None or "conv_fwd" in PROFILE_KINDS)
    if prof:        # not a library conv dispatch: timed with a torch event pair on the launch stream (bench.py roofline)
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
    call("sod_bottleneck_frozen_fwd", ptr(x), N, H, W, Cin, ptr(w1), ptr(b1), ptr(w2), ptr(b2), ptr(w3), ptr(b3), ptr(wsc), ptr(out), stream_ptr())
    if prof:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        macs = Cin * 64 + 9 * 64 * 64 + 64 * 256 + (Cin * 256 if wsc is not None else 0)     # algorithmic: no halo recomputation
        PROFILE.append(("conv_fwd", 2.0 * N * H * W * macs, e0, e1, (N, H, W, Cin, 256, "bneck", 1), -8))
    return out


def nchw_f32_to_nhwc_bf16(x):
    _chk(x, torch.float32, "x")
    N, C, H, W = x.shape
    y = torch.empty((N, H, W, C), dtype=torch.bfloat16, device=x.device)
    call("sod_nchw_f32_to_nhwc_bf16", ptr(x), ptr(y), N, C, H * W, stream_ptr())
    return y


# ----------------------------------------------------------------------------------------------- losses
def focal_loss_fwd(logits, labels=None, dense=None, alpha=0.25, gamma=2.0, K=None, want_elem=False):
    _chk(logits, torch.float32, "logits"); _chk(labels, torch.int32, "labels"); _chk(dense, torch.float32, "targets")
    ld = logits.shape[-1]
    K = ld if K is None else K
    M = logits.numel() // ld
    elem = torch.empty((M, K), dtype=torch.float32, device=logits.device) if want_elem else None
    out = torch.empty(1, dtype=torch.float32, device=logits.device)
    call("sod_sigmoid_focal_loss_fwd", ptr(logits), ptr(labels), ptr(dense), M, K, ld, alpha, gamma, ptr(elem), ptr(out),
         ptr(reduce_ws(logits.device)), stream_ptr())
    return out, elem


def focal_loss_fwd_grad(logits, labels, alpha=0.25, gamma=2.0, K=None, ld_out=None, out=None):
    """Forward sum and the UN-scaled bf16 gradient (rows of ld_out elements) in one pass over the logits (sod_sigmoid_focal_loss_fwd_grad).
    Returns (sum (1,), dlogits); raises SlenderHipError for layouts the vectorised kernel does not take."""
    _chk(logits, torch.float32, "logits"); _chk(labels, torch.int32, "labels")
    ld = logits.shape[-1]
    K = ld if K is None else K
    M = logits.numel() // ld
    ld_out = K if ld_out is None else ld_out
    if out is None:
        out = torch.empty((M, ld_out), dtype=torch.bfloat16, device=logits.device)
    s = torch.empty(1, dtype=torch.float32, device=logits.device)
    call("sod_sigmoid_focal_loss_fwd_grad", ptr(logits), ptr(labels), M, K, ld, alpha, gamma, ptr(s), ptr(reduce_ws(logits.device)), ptr(out), ld_out,
         stream_ptr())
    return s, out


def focal_loss_bwd(logits, labels=None, dense=None, alpha=0.25, gamma=2.0, K=None, scale_num=None, scale_den=None,
                   den_mul=1.0, den_min=1.0, ld_out=None, out_bf16=False, out=None):
    _chk(logits, torch.float32, "logits")
    ld = logits.shape[-1]
    K = ld if K is None else K
    M = logits.numel() // ld
    ld_out = K if ld_out is None else ld_out
    if out is None:
        out = torch.empty((M, ld_out), dtype=torch.bfloat16 if out_bf16 else torch.float32, device=logits.device)
    call("sod_sigmoid_focal_loss_bwd", ptr(logits), ptr(labels), ptr(dense), M, K, ld, alpha, gamma, ptr(scale_num), ptr(scale_den),
         den_mul, den_min, ptr(out), ld_out, 1 if out_bf16 else 0, stream_ptr())
    return out


def iou_loss_fwd(pred, target, weight=None, loss_type="iou", mask=None, mask_bg=-1, want_elem=False):
    _chk(pred, torch.float32, "pred"); _chk(target, torch.float32, "target"); _chk(weight, torch.float32, "weight"); _chk(mask, torch.int32, "mask")
    P = pred.shape[0]
    elem = torch.empty(P, dtype=torch.float32, device=pred.device) if want_elem else None
    out = torch.empty(1, dtype=torch.float32, device=pred.device)
    call("sod_iou_loss_fwd", ptr(pred), ptr(target), ptr(weight), ptr(mask), mask_bg, P, IOU_TYPES[loss_type], ptr(elem), ptr(out),
         ptr(reduce_ws(pred.device)), stream_ptr())
    return out, elem


def iou_loss_bwd(pred, target, weight=None, loss_type="iou", mask=None, mask_bg=-1, grad_scale=None):
    P = pred.shape[0]
    dpred = torch.empty_like(pred)
    call("sod_iou_loss_bwd", ptr(pred), ptr(target), ptr(weight), ptr(mask), mask_bg, P, IOU_TYPES[loss_type], ptr(grad_scale), ptr(dpred), stream_ptr())
    return dpred


def _int_arr(v):
    return (ctypes.c_int * len(v))(*[int(i) for i in v])


def _float_arr(v):
    return (ctypes.c_float * len(v))(*[float(i) for i in v])


def fcos_assign(boxes, classes, box_offsets, N, lvl_hw, strides, sizes_of_interest, radius, num_classes):
    """boxes (sumG,4) f32, classes (sumG,) i32, box_offsets (N+1,) i32 (all CUDA). Returns labels (N,L) i32,
    reg_targets (N,L,4), ctr_targets (N,L), stats (2,) = [num_pos, sum_ctr]."""
    dev = box_offsets.device
    _chk(boxes, torch.float32, "boxes"); _chk(classes, torch.int32, "classes"); _chk(box_offsets, torch.int32, "box_offsets")
    L = sum(h * w for h, w in lvl_hw)
    labels = torch.empty((N, L), dtype=torch.int32, device=dev)
    reg = torch.empty((N, L, 4), dtype=torch.float32, device=dev)
    ctr = torch.empty((N, L), dtype=torch.float32, device=dev)
    stats = torch.empty(2, dtype=torch.float32, device=dev)
    nl = len(lvl_hw)
    call("sod_fcos_assign", ptr(boxes), ptr(classes), ptr(box_offsets), N, nl,
         ctypes.cast(_int_arr([h for h, _ in lvl_hw]), ctypes.c_void_p), ctypes.cast(_int_arr([w for _, w in lvl_hw]), ctypes.c_void_p),
         ctypes.cast(_int_arr(strides), ctypes.c_void_p),
         ctypes.cast(_float_arr([s[0] for s in sizes_of_interest]), ctypes.c_void_p),
         ctypes.cast(_float_arr([s[1] for s in sizes_of_interest]), ctypes.c_void_p),
         float(radius), num_classes, ptr(labels), ptr(reg), ptr(ctr), ptr(stats), ptr(reduce_ws(dev)), stream_ptr())
    return labels, reg, ctr, stats


def fcos_regctr_loss_fwd(box_raw, ld_box, ctr_logit, ld_ctr, labels, reg_t, ctr_t, scales, N, lvl_hw, strides, num_classes,
                         loss_type, norm_reg):
    dev = labels.device
    sums = torch.empty(2, dtype=torch.float32, device=dev)
    nl = len(lvl_hw)
    call("sod_fcos_regctr_loss_fwd", ptr(box_raw), ld_box, ptr(ctr_logit), ld_ctr, ptr(labels), ptr(reg_t), ptr(ctr_t), ptr(scales), N, nl,
         ctypes.cast(_int_arr([h for h, _ in lvl_hw]), ctypes.c_void_p), ctypes.cast(_int_arr([w for _, w in lvl_hw]), ctypes.c_void_p),
         ctypes.cast(_int_arr(strides), ctypes.c_void_p), num_classes, IOU_TYPES[loss_type], 1 if norm_reg else 0,
         ptr(sums), ptr(reduce_ws(dev)), stream_ptr())
    return sums


def fcos_regctr_loss_bwd(box_raw, ld_box, ctr_logit, ld_ctr, labels, reg_t, ctr_t, scales, N, lvl_hw, strides, num_classes,
                         loss_type, norm_reg, grad_reg, grad_ctr, stats, inv_world, dbox, ld_out, ctr_col, dctr, ld_dctr, dctr_col,
                         dscales):
    dev = labels.device
    nl = len(lvl_hw)
    # gradient rows in the storage precision: bf16 for the MFMA kernels, fp32 in the validation mode
    call("sod_fcos_regctr_loss_bwd_f32" if is_f32() else "sod_fcos_regctr_loss_bwd", ptr(box_raw), ld_box, ptr(ctr_logit), ld_ctr, ptr(labels), ptr(reg_t), ptr(ctr_t), ptr(scales), N, nl,
         ctypes.cast(_int_arr([h for h, _ in lvl_hw]), ctypes.c_void_p), ctypes.cast(_int_arr([w for _, w in lvl_hw]), ctypes.c_void_p),
         ctypes.cast(_int_arr(strides), ctypes.c_void_p), num_classes, IOU_TYPES[loss_type], 1 if norm_reg else 0,
         ptr(grad_reg), ptr(grad_ctr), ptr(stats), float(inv_world), ptr(dbox), ld_out, ctr_col, ptr(dctr), ld_dctr, dctr_col,
         ptr(dscales), ptr(reduce_ws(dev)), stream_ptr())


def fcos_finalize_losses(focal_sum, regctr_sums, stats, inv_world):
    out = torch.empty(3, dtype=torch.float32, device=stats.device)
    call("sod_fcos_finalize_losses", ptr(focal_sum), ptr(regctr_sums), ptr(stats), float(inv_world), ptr(out), stream_ptr())
    return out


# ----------------------------------------------------------------------------------------------- detection ops
def nms(boxes, scores, iou_threshold):
    """torchvision.ops.nms contract: kept indices (int64) in descending-score order. Sorting is torch's stable sort (plumbing);
    the IoU mask and the greedy scan run in the HIP kernels."""
    _chk(boxes, torch.float32, "boxes"); _chk(scores, torch.float32, "scores")
    n = boxes.shape[0]
    dev = boxes.device
    if n == 0:
        return torch.empty((0,), dtype=torch.int64, device=dev)
    order = torch.sort(scores, descending=True, stable=True).indices.contiguous()
    keep = torch.empty(n, dtype=torch.int64, device=dev)
    nkeep = torch.zeros(1, dtype=torch.int32, device=dev)
    ws = torch.empty(int(_C.load().sod_nms_workspace_bytes(n)), dtype=torch.uint8, device=dev)
    call("sod_nms", ptr(boxes), ptr(order), n, float(iou_threshold), ptr(keep), ptr(nkeep), ptr(ws), stream_ptr())
    return keep[: int(nkeep.item())]


def fcos_decode(cls_buf, box_buf, scales, level_hw, strides, num_classes, centerness_on_reg, norm_reg_targets, pre_nms_thresh, pre_nms_top_n):
    """Per-level threshold / top-k / decode of FCOS inference for the whole batch in one launch (no host sync).
    cls_buf (N, L, ld) / box_buf (N, L, ld) fp32 prediction buffers -> boxes (N, M, 4), scores (N, M) (-inf = empty slot),
    classes (N, M) int32, counts (N, nlev) int32 with M = nlev * pre_nms_top_n."""
    _chk(cls_buf, torch.float32, "cls_buf"); _chk(box_buf, torch.float32, "box_buf"); _chk(scales, torch.float32, "scales")
    N, L, ld_cls = cls_buf.shape
    ld_box = box_buf.shape[-1]
    nlev = len(level_hw)
    if L != sum(h * w for h, w in level_hw) or box_buf.shape[1] != L:
        raise _C.SlenderHipError("fcos_decode: prediction buffers do not match the level geometry")
    M = nlev * int(pre_nms_top_n)
    dev = cls_buf.device
    boxes = torch.empty((N, M, 4), dtype=torch.float32, device=dev)
    scores = torch.empty((N, M), dtype=torch.float32, device=dev)
    classes = torch.empty((N, M), dtype=torch.int32, device=dev)
    counts = torch.empty((N, nlev), dtype=torch.int32, device=dev)
    call("sod_fcos_decode", ptr(cls_buf), ld_cls, ptr(box_buf), ld_box, ptr(scales), N, nlev,
         ctypes.cast(_int_arr([h for h, _ in level_hw]), ctypes.c_void_p), ctypes.cast(_int_arr([w for _, w in level_hw]), ctypes.c_void_p),
         ctypes.cast(_int_arr(strides), ctypes.c_void_p), int(num_classes), 4 if centerness_on_reg else -1, -1 if centerness_on_reg else int(num_classes),
         1 if norm_reg_targets else 0, float(pre_nms_thresh), int(pre_nms_top_n), ptr(boxes), ptr(scores), ptr(classes), ptr(counts), stream_ptr())
    return boxes, scores, classes, counts


def dense_topk_select(logits, rows_per_level, num_classes, score_thresh, top_n, by_row_max=False):
    """Per-(image, level) threshold + top-k over rows of K class logits for the whole batch in one launch.  logits (N, R, ld) fp32;
    returns rows (N, M) int32 (row index inside the level), scores (N, M) (-inf = empty slot), classes (N, M) int32, counts
    (N, nlev) int32 with M = nlev * top_n."""
    _chk(logits, torch.float32, "logits")
    N, R, ld = logits.shape
    nlev = len(rows_per_level)
    if R != sum(rows_per_level):
        raise _C.SlenderHipError("dense_topk_select: logits do not match the level geometry")
    M = nlev * int(top_n)
    dev = logits.device
    rows = torch.empty((N, M), dtype=torch.int32, device=dev)
    scores = torch.empty((N, M), dtype=torch.float32, device=dev)
    classes = torch.empty((N, M), dtype=torch.int32, device=dev)
    counts = torch.empty((N, nlev), dtype=torch.int32, device=dev)
    call("sod_dense_topk_select", ptr(logits), ld, N, nlev, ctypes.cast(_int_arr(rows_per_level), ctypes.c_void_p), int(num_classes), 1 if by_row_max else 0,
         float(score_thresh), int(top_n), ptr(rows), ptr(scores), ptr(classes), ptr(counts), stream_ptr())
    return rows, scores, classes, counts


def batched_nms_topk(boxes, scores, classes, iou_threshold, max_keep):
    """Class-aware NMS + top-``max_keep`` of B images at once on padded candidate lists (score -inf = empty slot); boxes (B, M, 4)
    XYXY or (B, M, 5) rotated.  Returns keep (B, max_keep) int64 local indices in score order (entries beyond num_keep are 0) and
    num_keep (B) int32, both on the device.  The per-image sort is torch's stable sort (plumbing), the rest runs in the HIP kernels."""
    _chk(boxes, torch.float32, "boxes"); _chk(scores, torch.float32, "scores"); _chk(classes, torch.int32, "classes")
    B, M = scores.shape
    D = boxes.shape[-1]
    dev = scores.device
    ws = torch.empty(int(_C.load().sod_batched_nms_workspace_bytes(B, M, D)), dtype=torch.uint8, device=dev)
    call("sod_batched_nms_prepare", ptr(boxes), ptr(scores), ptr(classes), B, M, D, ptr(ws), stream_ptr())
    order = torch.sort(scores, dim=1, descending=True, stable=True).indices.contiguous()
    keep = torch.zeros((B, int(max_keep)), dtype=torch.int64, device=dev)
    nkeep = torch.zeros(B, dtype=torch.int32, device=dev)
    call("sod_batched_nms_run", ptr(order), B, M, D, float(iou_threshold), int(max_keep), ptr(keep), ptr(nkeep), ptr(ws), stream_ptr())
    return keep, nkeep


def rpn_clip_filter(boxes, scores, image_hw, min_size):
    """In place on (B, M, D) boxes / (B, M) scores: non-finite entries and boxes not larger than ``min_size`` become empty slots
    (score -inf), boxes are clipped to their image.  Returns the device counter of non-finite entries (int32, 1 element)."""
    _chk(boxes, torch.float32, "boxes"); _chk(scores, torch.float32, "scores"); _chk(image_hw, torch.float32, "image_hw")
    B, M, D = boxes.shape
    bad = torch.zeros(1, dtype=torch.int32, device=boxes.device)
    call("sod_rpn_clip_filter", ptr(boxes), ptr(scores), ptr(image_hw), B, M, D, float(min_size), ptr(bad), stream_ptr())
    return bad


def nms_rotated(boxes, scores, iou_threshold):
    """detectron2.layers.nms_rotated: boxes (n,5) = (cx,cy,w,h,angle_deg)."""
    _chk(boxes, torch.float32, "boxes"); _chk(scores, torch.float32, "scores")
    n, dev = boxes.shape[0], boxes.device
    if n == 0:
        return torch.empty((0,), dtype=torch.int64, device=dev)
    order = torch.sort(scores, descending=True, stable=True).indices.contiguous()
    keep = torch.empty(n, dtype=torch.int64, device=dev)
    nkeep = torch.zeros(1, dtype=torch.int32, device=dev)
    ws = torch.empty(int(_C.load().sod_nms_workspace_bytes(n)), dtype=torch.uint8, device=dev)
    call("sod_nms_rotated", ptr(boxes), ptr(order), n, float(iou_threshold), ptr(keep), ptr(nkeep), ptr(ws), stream_ptr())
    return keep[: int(nkeep.item())]


def box_iou_rotated(b1, b2):
    _chk(b1, torch.float32, "boxes1"); _chk(b2, torch.float32, "boxes2")
    out = torch.empty((b1.shape[0], b2.shape[0]), dtype=torch.float32, device=b1.device)
    call("sod_box_iou_rotated", ptr(b1), b1.shape[0], ptr(b2), b2.shape[0], ptr(out), stream_ptr())
    return out


def roi_align_fwd(x, rois, output_size, spatial_scale, sampling_ratio=0, rotated=False):
    """x (N,H,W,C) bf16 (fp32 in the validation mode), rois (R,5|6) f32 -> (R,PH,PW,C) f32 (aligned=True semantics)."""
    _chk(x, ACT_DTYPE, "x"); _chk(rois, torch.float32, "rois")
    N, H, W, C = x.shape
    PH, PW = output_size
    R = rois.shape[0]
    out = torch.empty((R, PH, PW, C), dtype=torch.float32, device=x.device)
    call("sod_roi_align_fwd_f32" if is_f32() else "sod_roi_align_fwd", ptr(x), ptr(rois), ptr(out), R, N, H, W, C, PH, PW, float(spatial_scale), int(sampling_ratio), 1 if rotated else 0, stream_ptr())
    return out


def roi_align_bwd(dout, rois, x_shape, spatial_scale, sampling_ratio=0, rotated=False):
    _chk(dout, torch.float32, "dout"); _chk(rois, torch.float32, "rois")
    N, H, W, C = x_shape
    R, PH, PW, _ = dout.shape
    dx = torch.zeros((N, H, W, C), dtype=torch.float32, device=dout.device)
    call("sod_roi_align_bwd", ptr(dout), ptr(rois), ptr(dx), R, N, H, W, C, PH, PW, float(spatial_scale), int(sampling_ratio), 1 if rotated else 0, stream_ptr())
    return dx


def giou_loss_xyxy(b1, b2, eps=1e-7, want_grad=False, grad_scale=None):
    _chk(b1, torch.float32, "boxes1"); _chk(b2, torch.float32, "boxes2")
    P = b1.shape[0]
    elem = torch.empty(P, dtype=torch.float32, device=b1.device)
    s = torch.empty(1, dtype=torch.float32, device=b1.device)
    d1 = torch.empty_like(b1) if want_grad else None
    call("sod_giou_loss_xyxy", ptr(b1), ptr(b2), P, float(eps), ptr(elem), ptr(s), ptr(grad_scale), ptr(d1), ptr(reduce_ws(b1.device)), stream_ptr())
    return elem, s, d1


def smooth_l1_loss(x, t, beta, want_grad=False, grad_scale=None):
    _chk(x, torch.float32, "input"); _chk(t, torch.float32, "target")
    elem = torch.empty_like(x)
    s = torch.empty(1, dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x) if want_grad else None
    call("sod_smooth_l1_loss", ptr(x), ptr(t), x.numel(), float(beta), ptr(elem), ptr(s), ptr(grad_scale), ptr(dx), ptr(reduce_ws(x.device)), stream_ptr())
    return elem, s, dx


def anchor_match(gt_boxes, anchors, thresholds, labels, allow_low_quality=True, out=None):
    """RetinaNet.label_anchors core: returns (matched_vals f32 (A,), matches i32 (A,), labels i8 (A,)); ``out`` = the three
    output rows to fill (e.g. rows of batch-sized buffers)."""
    _chk(gt_boxes, torch.float32, "gt_boxes"); _chk(anchors, torch.float32, "anchors")
    A, G = anchors.shape[0], gt_boxes.shape[0]
    dev = anchors.device
    if out is not None:
        vals, idx, lab = out
        _chk(vals, torch.float32, "matched_vals"); _chk(idx, torch.int32, "matches"); _chk(lab, torch.int8, "labels")
        if vals.numel() != A or idx.numel() != A or lab.numel() != A:
            raise _C.SlenderHipError("anchor_match: output rows must hold one entry per anchor")
    else:
        vals = torch.empty(A, dtype=torch.float32, device=dev)
        idx = torch.empty(A, dtype=torch.int32, device=dev)
        lab = torch.empty(A, dtype=torch.int8, device=dev)
    ws = torch.empty(max(G, 1), dtype=torch.int32, device=dev)
    call("sod_anchor_match_rotated" if anchors.shape[-1] == 5 else "sod_anchor_match", ptr(gt_boxes) if G else None, G, ptr(anchors), A, float(thresholds[0]), float(thresholds[1]), int(labels[0]), int(labels[1]),
         int(labels[2]), 1 if allow_low_quality else 0, ptr(vals), ptr(idx), ptr(lab), ptr(ws), stream_ptr())
    return vals, idx, lab


# ----------------------------------------------------------------------------------------------- deformable conv
def f32_to_bf16(x):
    _chk(x, torch.float32, "x")
    y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    call("sod_f32_to_bf16", ptr(x), ptr(y), x.numel(), stream_ptr())
    return y


def deform_im2col(x, offset, mask, ksize, stride, pad, dil, dg=1, off_ld=0, mask_ld=0, mask_is_logit=False):
    """x (N,H,W,C) bf16; offset/mask fp32 (pitched rows); returns cols (N,Ho,Wo,KH*KW*C) bf16 (both fp32 in the validation mode)."""
    _chk(x, ACT_DTYPE, "x"); _chk(offset, torch.float32, "offset"); _chk(mask, torch.float32, "mask")
    N, H, W, C = x.shape
    KH, KW = ksize
    Ho, Wo = conv_out_size(H, W, KH, KW, stride, pad, dil)
    cols = torch.empty((N, Ho, Wo, KH * KW * C), dtype=ACT_DTYPE, device=x.device)
    call("sod_deform_im2col_f32" if is_f32() else "sod_deform_im2col", ptr(x), ptr(offset), ptr(mask), ptr(cols), N, H, W, C, KH, KW, stride, pad, dil, dg, off_ld, mask_ld,
         1 if mask_is_logit else 0, stream_ptr())
    return cols


def deform_conv_fwd_fused(x, offset, mask, w_gemm, bias, ksize, stride, pad, dil, dg=1, off_ld=0, mask_ld=0, mask_is_logit=False, relu=False):
    """DeformConv forward without the column buffer: x (N,H,W,C) bf16, w_gemm (K,1,1,KH*KW*C) bf16 -> y (N,Ho,Wo,K) bf16."""
    _chk(x, torch.bfloat16, "x"); _chk(w_gemm, torch.bfloat16, "w"); _chk(bias, torch.float32, "bias")
    N, H, W, C = x.shape
    KH, KW = ksize
    K = w_gemm.shape[0]
    if w_gemm.numel() != K * KH * KW * C:
        raise _C.SlenderHipError("deform_conv_fwd_fused: weight shape does not match (K, KH*KW*C)")
    Ho, Wo = conv_out_size(H, W, KH, KW, stride, pad, dil)
    y = torch.empty((N, Ho, Wo, K), dtype=torch.bfloat16, device=x.device)
    call("sod_deform_conv_fwd_fused", ptr(x), ptr(offset), ptr(mask), ptr(w_gemm), ptr(bias), ptr(y), N, H, W, C, K, KH, KW, stride, pad, dil, dg,
         off_ld, mask_ld, 1 if mask_is_logit else 0, 1 if relu else 0, stream_ptr())
    return y


def deform_conv_fwd_f32(x, offset, mask, w, bias, ksize, stride, pad, dil, dg=1, mask_is_logit=False):
    """fp32 test-mode forward: x (N,H,W,C) fp32, w (K, KH*KW*C) fp32 in (tap, channel) order -> y (N,Ho,Wo,K) fp32."""
    _chk(x, torch.float32, "x"); _chk(w, torch.float32, "w"); _chk(offset, torch.float32, "offset"); _chk(mask, torch.float32, "mask"); _chk(bias, torch.float32, "bias")
    N, H, W, C = x.shape
    KH, KW = ksize
    K = w.shape[0]
    Ho, Wo = conv_out_size(H, W, KH, KW, stride, pad, dil)
    y = torch.empty((N, Ho, Wo, K), dtype=torch.float32, device=x.device)
    call("sod_deform_conv_fwd_f32", ptr(x), ptr(offset), ptr(mask), ptr(w), ptr(bias), ptr(y), N, H, W, C, K, KH, KW, stride, pad, dil, dg, 0, 0,
         1 if mask_is_logit else 0, stream_ptr())
    return y


def deform_conv_wgrad_fused(dy, x, offset, mask, dw, ksize, stride, pad, dil, dg=1, off_ld=0, mask_ld=0, mask_is_logit=False, qscale=None):
    """Accumulates the DeformConv weight gradient into dw (K, KH*KW*C elements, fp32) without a column buffer."""
    _chk(dy, torch.bfloat16, "dy"); _chk(x, torch.bfloat16, "x"); _chk(dw, torch.float32, "dw")
    N, H, W, C = x.shape
    KH, KW = ksize
    K = dy.shape[-1]
    if dw.numel() != K * KH * KW * C:
        raise _C.SlenderHipError("deform_conv_wgrad_fused: dw does not hold K x KH*KW*C elements")
    side = _wgrad_stream(dw.device, (dy, x, offset, mask, qscale), dw.data_ptr())
    ws = wgrad_workspace(dw.device, side)
    _chk(qscale, torch.float32, "qscale")
    call("sod_deform_conv_wgrad_fused", ptr(dy), ptr(x), ptr(offset), ptr(mask), ptr(dw), ptr(qscale), N, H, W, C, K, KH, KW, stride, pad, dil, dg, off_ld, mask_ld,
         1 if mask_is_logit else 0, ptr(ws), ws.numel(), stream_ptr(side))
    return dw


def deform_fused_supported(C, K, dg):
    """Shapes both fused DeformConv kernels (forward and weight gradient) accept (bf16 product path only)."""
    return not is_f32() and C % 128 == 0 and K % 8 == 0 and C % dg == 0 and (dg == 1 or (C // dg) % 128 == 0)


def deform_bwd_fused_supported(C, K, dg, ksize=(3, 3), stride=1, dil=1):
    """Layers the fused input / offset / mask gradient (sod_deform_conv_bwd_fused) accepts (bf16 product path only): the library's own
    answer - channel counts AND the LDS window, which depends on kernel size, stride and dilation (a stride-2 K = 512 layer does not fit).
    ``False`` sends the layer through conv2d_dgrad + deform_col2im."""
    if is_f32():
        return False
    return _C.load().sod_deform_conv_bwd_fused_supported(int(C), int(K), int(ksize[0]), int(ksize[1]), int(stride), int(dil), int(dg)) == 1


class DeformWindowCounter:
    """Counts the sample lanes (pixel x tap x 8 channels with a non-zero gradient) of the tiled DeformConv backward kernels that left
    their LDS window and took the global-atomic path (sod_deform_conv_set_window_counter): the data-dependent cliff of that design.

        with DeformWindowCounter(device) as c: ...backward...; n = c.read()
    """

    def __init__(self, device):
        self.t = torch.zeros(1, dtype=torch.int64, device=device)

    def __enter__(self):
        call("sod_deform_conv_set_window_counter", ptr(self.t))
        return self

    def __exit__(self, *exc):
        call("sod_deform_conv_set_window_counter", None)

    def read(self, reset=True):
        n = int(self.t.item())
        if reset:
            self.t.zero_()
        return n


def deform_conv_bwd_fused(dy, wt, x, offset, mask, ksize, stride, pad, dil, dg, doffset, dmask, off_ld=0, mask_ld=0, mask_is_logit=False):
    """dx fp32 (N,H,W,C) + the (zero-initialised, pitched) doffset / dmask from dy (N,Ho,Wo,K) bf16 and wt = the CRSK weight copy
    ((KH*KW*C, 1, 1, K) bf16), with no column-gradient tensor in between."""
    _chk(dy, torch.bfloat16, "dy"); _chk(wt, torch.bfloat16, "wt"); _chk(x, torch.bfloat16, "x")
    N, H, W, C = x.shape
    KH, KW = ksize
    K = dy.shape[-1]
    if wt.numel() != KH * KW * C * K:
        raise _C.SlenderHipError("deform_conv_bwd_fused: wt does not hold KH*KW*C x K elements")
    dx = torch.zeros((N, H, W, C), dtype=torch.float32, device=x.device)
    wn = torch.empty(KH * KW * C, dtype=torch.float32, device=x.device)
    call("sod_deform_conv_bwd_fused", ptr(dy), ptr(wt), ptr(x), ptr(offset), ptr(mask), ptr(dx), ptr(doffset), ptr(dmask), ptr(wn), N, H, W, C, K, KH, KW,
         stride, pad, dil, dg, off_ld, mask_ld, 1 if mask_is_logit else 0, stream_ptr())
    return dx


def deform_col2im(dcols, x, offset, mask, ksize, stride, pad, dil, dg, doffset, dmask, off_ld=0, mask_ld=0, mask_is_logit=False):
    """Returns dx fp32 (N,H,W,C); fills the (zero-initialised, pitched) doffset / dmask."""
    _chk(dcols, ACT_DTYPE, "dcols"); _chk(x, ACT_DTYPE, "x"); _chk(doffset, torch.float32, "doffset"); _chk(dmask, torch.float32, "dmask")
    N, H, W, C = x.shape
    KH, KW = ksize
    dx = torch.zeros((N, H, W, C), dtype=torch.float32, device=x.device)
    call("sod_deform_col2im_f32" if is_f32() else "sod_deform_col2im", ptr(dcols), ptr(x), ptr(offset), ptr(mask), ptr(dx), ptr(doffset), ptr(dmask), N, H, W, C, KH, KW, stride, pad, dil,
         dg, off_ld, mask_ld, 1 if mask_is_logit else 0, stream_ptr())
    return dx


# ----------------------------------------------------------------------------------------------- RetinaNet
def retina_targets(anchors, gt_boxes, gt_classes, matches, match_labels, num_classes, weights, gt_labels_out, gt_deltas_out):
    G = gt_boxes.shape[0]
    w = _float_arr(weights)
    call("sod_retina_targets", ptr(anchors), anchors.shape[0], ptr(gt_boxes) if G else None, ptr(gt_classes) if G else None, G,
         ptr(matches) if G else None, ptr(match_labels) if G else None, num_classes, ctypes.cast(w, ctypes.c_void_p), ptr(gt_labels_out),
         ptr(gt_deltas_out), stream_ptr())


def retina_box_loss_fwd(pred, pitch, gt_labels, gt_deltas, N, R, A, num_classes, beta, normalizer, momentum):
    sums = torch.empty(2, dtype=torch.float32, device=pred.device)
    call("sod_retina_box_loss_fwd", ptr(pred), pitch, ptr(gt_labels), ptr(gt_deltas), N, R, A, num_classes, float(beta), ptr(sums),
         ptr(normalizer), float(momentum), ptr(reduce_ws(pred.device)), stream_ptr())
    return sums


def retina_box_loss_bwd(pred, pitch, gt_labels, gt_deltas, N, R, A, num_classes, beta, grad_num, grad_den, dpred):
    call("sod_retina_box_loss_bwd_f32" if is_f32() else "sod_retina_box_loss_bwd", ptr(pred), pitch, ptr(gt_labels), ptr(gt_deltas), N, R, A, num_classes, float(beta), ptr(grad_num),
         ptr(grad_den), ptr(dpred), stream_ptr())


# ----------------------------------------------------------------------------------------------- FPN with a norm
def add_up2(a, b):
    """a + nearest-2x-upsample(b) (d2 FPN top-down sum when a norm follows the lateral conv)."""
    _chk(a, torch.bfloat16, "a"); _chk(b, torch.bfloat16, "b")
    N, H, W, C = a.shape
    if tuple(b.shape) != (N, H // 2, W // 2, C) or H % 2 or W % 2:
        raise _C.SlenderHipError(f"add_up2: {tuple(a.shape)} is not the 2x upsampling of {tuple(b.shape)}")
    o = torch.empty_like(a)
    call("sod_add_up2_bf16", ptr(a), ptr(b), ptr(o), N, H, W, C, stream_ptr())
    return o


def sample_labels(labels, num_samples, positive_fraction, bg_label, seed=None):
    """detectron2 subsample_labels for a batch: labels (N, R) int8 (-1 ignore, bg_label background, anything else positive) ->
    (out (N, R) int8: 1 sampled positive / 0 sampled negative / -1 rest, counts (N, 2) int32).  ``seed``: drawn from torch's CPU generator
    when None (so torch.manual_seed reproduces the draw); nothing is read back from the device."""
    _chk(labels, torch.int8, "labels")
    N, R = labels.shape
    if seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    out = torch.empty_like(labels)
    counts = torch.empty((N, 2), dtype=torch.int32, device=labels.device)
    call("sod_sample_labels", ptr(labels), N, R, int(num_samples), float(positive_fraction), int(bg_label), ctypes.c_ulonglong(seed), ptr(out), ptr(counts),
         stream_ptr())
    return out, counts


def sample_labels_list(labels, num_samples, positive_fraction, bg_label, seed=None):
    """sample_labels + the drawn indices of every row as (N, num_samples) int32, -1 padded, sorted descending (a run-to-run stable order;
    the kernel emits them as it finds them).  For rows far longer than the draw (RPN anchors), where scanning every element a second time
    to compact the mask would cost more than the draw."""
    _chk(labels, torch.int8, "labels")
    N, R = labels.shape
    if seed is None:
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    out = torch.empty_like(labels)
    counts = torch.empty((N, 2), dtype=torch.int32, device=labels.device)
    idx = torch.empty((N, int(num_samples)), dtype=torch.int32, device=labels.device)
    scratch = torch.empty(N, dtype=torch.int32, device=labels.device)
    call("sod_sample_labels_list", ptr(labels), N, R, int(num_samples), float(positive_fraction), int(bg_label), ctypes.c_ulonglong(seed), ptr(out),
         ptr(counts), ptr(idx), ptr(scratch), stream_ptr())
    return out, counts, torch.sort(idx, dim=1, descending=True).values.contiguous()


def compact_samples(mask, slots):
    """Indices of the sampled elements of each row of ``mask`` (N, R) int8 - 1s first, then 0s, index order - as (N, slots) int32 padded
    with -1, and their number (N,) int32."""
    _chk(mask, torch.int8, "mask")
    N, R = mask.shape
    idx = torch.empty((N, slots), dtype=torch.int32, device=mask.device)
    num = torch.empty(N, dtype=torch.int32, device=mask.device)
    call("sod_compact_samples", ptr(mask), N, R, int(slots), ptr(idx), ptr(num), stream_ptr())
    return idx, num


def roi_label_batched(boxes, counts, gt_boxes, gt_classes, gt_off, iou_threshold, labels2, num_classes):
    """boxes (N, R, D) padded proposal rows (counts (N,) int32 valid), gt_boxes (G, D) / gt_classes (G,) int32 of all images concatenated,
    gt_off (N + 1,) int32 -> matches (N, R) int32 (index inside the image), classes (N, R) int8 (class / num_classes / -1)."""
    _chk(boxes, torch.float32, "boxes"); _chk(counts, torch.int32, "counts"); _chk(gt_boxes, torch.float32, "gt_boxes")
    _chk(gt_classes, torch.int32, "gt_classes"); _chk(gt_off, torch.int32, "gt_off")
    N, R, D = boxes.shape
    matches = torch.empty((N, R), dtype=torch.int32, device=boxes.device)
    cls = torch.empty((N, R), dtype=torch.int8, device=boxes.device)
    call("sod_roi_label_batched", ptr(boxes), ptr(counts), N, R, D, ptr(gt_boxes) if gt_boxes.numel() else None, ptr(gt_classes) if gt_classes.numel() else None,
         ptr(gt_off), float(iou_threshold), int(labels2[0]), int(labels2[1]), int(num_classes), ptr(matches), ptr(cls), stream_ptr())
    return matches, cls


def rpn_gather_sampled(logits_l, deltas_l, idx, A, D):
    """Rows of the sampled anchors: per-level padded NHWC head outputs (N, H, W, pitch) fp32 + idx (N, S) int32 (anchor index in the
    concatenated (level, h, w, a) order, -1 = empty slot) -> (N, S) logits, (N, S, D) deltas."""
    _chk(idx, torch.int32, "idx")
    for t in list(logits_l) + list(deltas_l):
        _chk(t, torch.float32, "head output")
    N, S = idx.shape
    lg = torch.empty((N, S), dtype=torch.float32, device=idx.device)
    dl = torch.empty((N, S, D), dtype=torch.float32, device=idx.device)
    hw = [t.shape[1] * t.shape[2] for t in logits_l]
    call("sod_rpn_gather_sampled", len(logits_l), _ptr_arr(logits_l), _ptr_arr(deltas_l), ctypes.cast(_int_arr(hw), ctypes.c_void_p),
         ctypes.cast(_int_arr([t.shape[3] for t in logits_l]), ctypes.c_void_p), ctypes.cast(_int_arr([t.shape[3] for t in deltas_l]), ctypes.c_void_p),
         ptr(idx), N, S, A, D, ptr(lg), ptr(dl), stream_ptr())
    return lg, dl


def rpn_scatter_sampled(shapes_l, shapes_d, idx, A, D, row_dlogits, row_ddeltas):
    """The adjoint of rpn_gather_sampled: zero tensors of the head outputs' shapes with the sampled rows' gradients written in."""
    _chk(idx, torch.int32, "idx"); _chk(row_dlogits, torch.float32, "row_dlogits"); _chk(row_ddeltas, torch.float32, "row_ddeltas")
    N, S = idx.shape
    dev = idx.device
    gl = [torch.zeros(sh, dtype=torch.float32, device=dev) for sh in shapes_l]
    gd = [torch.zeros(sh, dtype=torch.float32, device=dev) for sh in shapes_d]
    hw = [sh[1] * sh[2] for sh in shapes_l]
    call("sod_rpn_scatter_sampled", len(gl), _ptr_arr(gl), _ptr_arr(gd), ctypes.cast(_int_arr(hw), ctypes.c_void_p),
         ctypes.cast(_int_arr([sh[3] for sh in shapes_l]), ctypes.c_void_p), ctypes.cast(_int_arr([sh[3] for sh in shapes_d]), ctypes.c_void_p),
         ptr(idx), N, S, A, D, ptr(row_dlogits), ptr(row_ddeltas), stream_ptr())
    return gl, gd


# ----------------------------------------------------------------------------------------------- RepPoints
def reppoints_dcn_offset(pts, num_points, scale=1.0, subtract_base=True, flip_xy=True):
    """pts (..., ld) fp32 point rows (x, y interleaved) -> deformable-conv offsets (dy, dx interleaved) minus the kernel grid."""
    _chk(pts, torch.float32, "pts")
    ld = pts.shape[-1]
    out = torch.empty_like(pts)
    call("sod_reppoints_dcn_offset", ptr(pts), ptr(out), pts.numel() // ld, ld, num_points, float(scale), 1 if subtract_base else 0,
         1 if flip_xy else 0, stream_ptr())
    return out


def points2bbox_fwd(pts, add, grid_stride, point_stride, num_points, boxes, box_img_stride, arg, arg_img_stride):
    """One level: pts (N,H,W,ld) fp32 (+ add) -> boxes / arg slices of the concatenated (N,X,4) / (N,X) buffers."""
    _chk(pts, torch.float32, "pts"); _chk(add, torch.float32, "add")
    N, H, W, ld = pts.shape
    call("sod_points2bbox_fwd", ptr(pts), ptr(add), ld, N, H, W, float(grid_stride), float(point_stride), num_points, ptr(boxes), box_img_stride,
         ptr(arg), arg_img_stride, stream_ptr())


def points2bbox_bwd(dboxes, box_img_stride, arg, arg_img_stride, shape, point_stride, num_points, want_f32=True, want_bf16=False):
    N, H, W, ld = shape
    d32 = torch.empty(shape, dtype=torch.float32, device=dboxes.device) if want_f32 else None
    d16 = torch.empty(shape, dtype=torch.bfloat16, device=dboxes.device) if want_bf16 else None
    call("sod_points2bbox_bwd", ptr(dboxes), box_img_stride, ptr(arg), arg_img_stride, ld, N, H, W, float(point_stride), num_points, ptr(d32), ptr(d16),
         stream_ptr())
    return d32, d16


def points2bbox_moment_fwd(pts, add, grid_stride, point_stride, num_points, moment_transfer, boxes, box_img_stride):
    """TRANSFORM_METHOD "moment": one level, boxes = mean -+ std * exp(moment_transfer)."""
    _chk(pts, torch.float32, "pts"); _chk(add, torch.float32, "add"); _chk(moment_transfer, torch.float32, "moment_transfer")
    N, H, W, ld = pts.shape
    call("sod_points2bbox_moment_fwd", ptr(pts), ptr(add), ld, N, H, W, float(grid_stride), float(point_stride), num_points, ptr(moment_transfer),
         ptr(boxes), box_img_stride, stream_ptr())


def points2bbox_moment_bwd(dboxes, box_img_stride, pts, add, grid_stride, point_stride, num_points, moment_transfer, moment_mul, dmoment,
                           want_f32=True, want_bf16=False):
    """-> d(pts) as fp32 and/or bf16 (N,H,W,ld); accumulates moment_mul * d(moment_transfer) into ``dmoment`` (2 floats)."""
    _chk(pts, torch.float32, "pts"); _chk(add, torch.float32, "add"); _chk(dmoment, torch.float32, "dmoment")
    N, H, W, ld = pts.shape
    d32 = torch.empty(pts.shape, dtype=torch.float32, device=pts.device) if want_f32 else None
    d16 = torch.empty(pts.shape, dtype=torch.bfloat16, device=pts.device) if want_bf16 else None
    call("sod_points2bbox_moment_bwd", ptr(dboxes), box_img_stride, ptr(pts), ptr(add), ld, N, H, W, float(grid_stride), float(point_stride), num_points,
         ptr(moment_transfer), float(moment_mul), ptr(d32), ptr(d16), ptr(dmoment), stream_ptr())
    return d32, d16


RP_MATCH_MODES = {"points": 0, "nearest_points": 1, "inside": 2}


def reppoints_point_match(centers, strides, lvl_start, gt_boxes, box_offsets, N, max_gt, mode, scale=4.0):
    """Batched init-box matcher: returns objectness (N,X) int32 and box labels (N,X,4) fp32."""
    _chk(centers, torch.float32, "centers"); _chk(strides, torch.float32, "strides"); _chk(gt_boxes, torch.float32, "gt_boxes")
    _chk(lvl_start, torch.int32, "lvl_start"); _chk(box_offsets, torch.int32, "box_offsets")
    X = centers.shape[0]
    obj = torch.empty((N, X), dtype=torch.int32, device=centers.device)
    blab = torch.empty((N, X, 4), dtype=torch.float32, device=centers.device)
    call("sod_reppoints_point_match", ptr(centers), ptr(strides), X, ptr(lvl_start), lvl_start.numel() - 1, ptr(gt_boxes) if max_gt else None,
         ptr(box_offsets), N, int(max_gt), RP_MATCH_MODES[mode] if isinstance(mode, str) else int(mode), float(scale), ptr(obj), ptr(blab), stream_ptr())
    return obj, blab


def reppoints_labels(matches, match_labels, gt_boxes, gt_classes, box_offsets, centers, image_hw, num_classes, objectness=None):
    """matches/match_labels (N,X) from anchor_match -> cls labels (N,X) int32, refine box labels (N,X,4); zeroes objectness off-image."""
    N, X = matches.shape
    cls = torch.empty((N, X), dtype=torch.int32, device=matches.device)
    rbox = torch.empty((N, X, 4), dtype=torch.float32, device=matches.device)
    call("sod_reppoints_labels", ptr(matches), ptr(match_labels), ptr(gt_boxes), ptr(gt_classes), ptr(box_offsets), ptr(centers), ptr(image_hw),
         N, X, num_classes, ptr(cls), ptr(rbox), ptr(objectness), stream_ptr())
    return cls, rbox


def reppoints_box_loss_fwd(pred, target, labels, strides, bg_label, beta):
    N, X = labels.shape
    sums = torch.empty(2, dtype=torch.float32, device=pred.device)
    call("sod_reppoints_box_loss_fwd", ptr(pred), ptr(target), ptr(labels), ptr(strides), N, X, int(bg_label), float(beta), ptr(sums),
         ptr(reduce_ws(pred.device)), stream_ptr())
    return sums


def reppoints_box_loss_bwd(pred, target, labels, strides, bg_label, beta, grad_num, grad_den, den_min, mul):
    N, X = labels.shape
    d = torch.empty_like(pred)
    call("sod_reppoints_box_loss_bwd", ptr(pred), ptr(target), ptr(labels), ptr(strides), N, X, int(bg_label), float(beta), ptr(grad_num),
         ptr(grad_den), float(den_min), float(mul), ptr(d), stream_ptr())
    return d


def reppoints_finalize(focal_sum, init_sums, refine_sums, normalizer, momentum, num_images, init_weight):
    out = torch.empty(3, dtype=torch.float32, device=focal_sum.device)
    call("sod_reppoints_finalize", ptr(focal_sum), ptr(init_sums), ptr(refine_sums), ptr(normalizer), float(momentum), int(num_images),
         float(init_weight), ptr(out), stream_ptr())
    return out


# ----------------------------------------------------------------------------------------------- two-stage (R-CNN) path
def box2box_get_deltas(src, tgt, weights):
    """Box2BoxTransform(.Rotated).get_deltas: src/tgt (n, 4|5) fp32 -> deltas (n, 4|5)."""
    _chk(src, torch.float32, "src"); _chk(tgt, torch.float32, "tgt")
    n, D = src.shape
    out = torch.empty_like(src)
    if n == 0:
        return out
    w = _float_arr(weights)
    call("sod_box2box_get_deltas", ptr(src), ptr(tgt), n, D, ctypes.cast(w, ctypes.c_void_p), ptr(out), stream_ptr())
    return out


def box2box_apply_deltas(deltas, boxes, weights, scale_clamp, k=1, ld=0):
    """deltas rows (pitch ld, k class-specific vectors) applied to boxes (n, 4|5) -> (n, k*D)."""
    _chk(deltas, torch.float32, "deltas"); _chk(boxes, torch.float32, "boxes")
    n, D = boxes.shape
    out = torch.empty((n, k * D), dtype=torch.float32, device=boxes.device)
    if n == 0:
        return out
    w = _float_arr(weights)
    call("sod_box2box_apply_deltas", ptr(deltas), ptr(boxes), n, k, D, ld, ctypes.cast(w, ctypes.c_void_p), float(scale_clamp), ptr(out), stream_ptr())
    return out


def _sum1(dev):
    return torch.empty(1, dtype=torch.float32, device=dev)


def bce_logits_loss_fwd(logits, labels):
    _chk(logits, torch.float32, "logits"); _chk(labels, torch.int8, "labels")
    s = _sum1(logits.device)
    call("sod_bce_logits_loss_fwd", ptr(logits), ptr(labels), logits.numel(), ptr(s), ptr(reduce_ws(logits.device)), stream_ptr())
    return s


def bce_logits_loss_bwd(logits, labels, grad_scale, scale_mul):
    d = torch.empty_like(logits)
    call("sod_bce_logits_loss_bwd", ptr(logits), ptr(labels), logits.numel(), ptr(grad_scale), float(scale_mul), ptr(d), stream_ptr())
    return d


def rpn_loc_loss_fwd(pred, target, labels, beta):
    _chk(pred, torch.float32, "pred"); _chk(target, torch.float32, "target"); _chk(labels, torch.int8, "labels")
    D = pred.shape[-1]
    s = _sum1(pred.device)
    call("sod_rpn_loc_loss_fwd", ptr(pred), ptr(target), ptr(labels), labels.numel(), D, float(beta), ptr(s), ptr(reduce_ws(pred.device)), stream_ptr())
    return s


def rpn_loc_loss_bwd(pred, target, labels, beta, grad_scale, scale_mul):
    d = torch.empty_like(pred)
    call("sod_rpn_loc_loss_bwd", ptr(pred), ptr(target), ptr(labels), labels.numel(), pred.shape[-1], float(beta), ptr(grad_scale), float(scale_mul),
         ptr(d), stream_ptr())
    return d


def softmax_ce_fwd(scores, labels, C):
    """scores (R, ld) fp32 rows with C valid columns; labels int32 (R,). Returns the SUM of the per-row losses."""
    _chk(scores, torch.float32, "scores"); _chk(labels, torch.int32, "labels")
    R, ld = scores.shape
    s = _sum1(scores.device)
    call("sod_softmax_ce_fwd", ptr(scores), ptr(labels), R, C, ld, ptr(s), ptr(reduce_ws(scores.device)), stream_ptr())
    return s


def softmax_ce_bwd(scores, labels, C, grad_scale, scale_mul):
    R, ld = scores.shape
    d = torch.empty_like(scores)
    call("sod_softmax_ce_bwd", ptr(scores), ptr(labels), R, C, ld, ptr(grad_scale), float(scale_mul), ptr(d), stream_ptr())
    return d


def fastrcnn_box_loss_fwd(pred, gt_classes, gt_deltas, K, beta):
    _chk(pred, torch.float32, "pred"); _chk(gt_classes, torch.int32, "gt_classes"); _chk(gt_deltas, torch.float32, "gt_deltas")
    R, ld = pred.shape
    s = _sum1(pred.device)
    call("sod_fastrcnn_box_loss_fwd", ptr(pred), ptr(gt_classes), ptr(gt_deltas), R, K, gt_deltas.shape[-1], ld, float(beta), ptr(s),
         ptr(reduce_ws(pred.device)), stream_ptr())
    return s


def fastrcnn_box_loss_bwd(pred, gt_classes, gt_deltas, K, beta, grad_scale, scale_mul):
    R, ld = pred.shape
    d = torch.empty_like(pred)
    call("sod_fastrcnn_box_loss_bwd", ptr(pred), ptr(gt_classes), ptr(gt_deltas), R, K, gt_deltas.shape[-1], ld, float(beta), ptr(grad_scale),
         float(scale_mul), ptr(d), stream_ptr())
    return d


def bce_logits_soft_fwd(logits, targets, labels, bg_label):
    """BCE-with-logits (sum) with float targets over the rows whose int32 label is a foreground class."""
    _chk(logits, torch.float32, "logits"); _chk(targets, torch.float32, "targets"); _chk(labels, torch.int32, "labels")
    s = _sum1(logits.device)
    call("sod_bce_logits_soft_fwd", ptr(logits), ptr(targets), ptr(labels), int(bg_label), logits.numel(), ptr(s), ptr(reduce_ws(logits.device)),
         stream_ptr())
    return s


def bce_logits_soft_bwd(logits, targets, labels, bg_label, grad_scale, scale_mul=1.0):
    d = torch.empty_like(logits)
    call("sod_bce_logits_soft_bwd", ptr(logits), ptr(targets), ptr(labels), int(bg_label), logits.numel(), ptr(grad_scale), float(scale_mul), ptr(d),
         stream_ptr())
    return d


def retina_giou_loss_fwd(pred, pitch, gt_labels, anchors, matched_boxes, N, R, A, num_classes, weights, scale_clamp, normalizer, momentum):
    """GIoU box regression of RetinaNet / AnchorHead on the pitched delta buffer; advances the EMA normaliser; returns [sum, num_pos]."""
    sums = torch.empty(2, dtype=torch.float32, device=pred.device)
    w = _float_arr(weights)
    call("sod_retina_giou_loss_fwd", ptr(pred), pitch, ptr(gt_labels), ptr(anchors), ptr(matched_boxes), N, R, A, num_classes,
         ctypes.cast(w, ctypes.c_void_p), float(scale_clamp), ptr(sums), ptr(normalizer), float(momentum), ptr(reduce_ws(pred.device)), stream_ptr())
    return sums


def retina_giou_loss_bwd(pred, pitch, gt_labels, anchors, matched_boxes, N, R, A, num_classes, weights, scale_clamp, grad_num, grad_den, dpred):
    w = _float_arr(weights)
    call("sod_retina_giou_loss_bwd_f32" if is_f32() else "sod_retina_giou_loss_bwd", ptr(pred), pitch, ptr(gt_labels), ptr(anchors), ptr(matched_boxes), N, R, A, num_classes,
         ctypes.cast(w, ctypes.c_void_p), float(scale_clamp), ptr(grad_num), ptr(grad_den), ptr(dpred), stream_ptr())


# ---- fp32 validation mode: the wrappers of the training step dispatch to functional_f32 when it is on; everything that only the bf16
# product path offers (fused stem / bottleneck, 1-bit masks, statistics epilogues, the fused DeformConv kernels, ...) is switched off by its callers or refuses
for _name in ("conv2d_fwd", "conv2d_dgrad", "conv2d_wgrad", "conv2d_fwd_ml", "conv2d_dgrad_ml", "conv2d_wgrad_ml", "weight_prep",
              "groupnorm_fwd", "groupnorm_bwd", "groupnorm_fwd_ml", "groupnorm_bwd_ml", "relu_fwd", "relu_bwd", "add_bf16", "f32_to_bf16",
              "add_up2", "upsample2x_bwd", "maxpool3x3s2", "bias_grad", "bias_grad_ml", "preprocess_image", "preprocess_batch"):
    globals()[_name] = _precision_dispatch(globals()[_name])
for _name in ("conv_gn_fwd_ml", "stem_fused", "bottleneck_frozen_fwd"):
    if _name in globals():
        globals()[_name] = _precision_dispatch(globals()[_name])     # no fp32 variant: raises instead of mixing precisions
if os.environ.get("SOD_PRECISION", "bf16") != "bf16":
    set_precision(os.environ["SOD_PRECISION"])
