"""The north_star loss-parity run (SURVEY.md §8 d): 100 training iterations of BASELINE configs[0] — FCOS R18-FPN, 2 synthetic
512x512 images per step, random init — on the HIP path and on the CPU oracle from identical initial weights and data, with the
reference's own learning-rate schedule (WarmupMultiStepLR, 1000 warm-up iterations from 0.001 x BASE_LR, configs/fcos/Base-Fcos.yaml
over detectron2's defaults).

Three trajectories of the total loss (reference: slender_det/modeling/meta_arch/fcos/fcosv2.py:104-148):
  hip   the product (bf16 activations / weights, fp32 accumulation and master weights), deterministic reductions switched on
  emu   oracle/model.py with bf16 STORAGE emulation (the same arithmetic contract, CPU fp32 kernels)
  f32   oracle/model.py in plain fp32 (= the reference's CPU path restated)
  hip32 the same product code in the fp32-STORAGE validation mode (SOD_PRECISION=fp32: csrc/f32_path.hip, layers/functional_f32.py)
north_star's bound ("total-loss delta < 1e-3 after 100 iterations") is evaluated for all of them and holds for none - not for the CPU fp32
oracle against itself in another summation order either (tests/golden/chaos100.json): the run amplifies 1e-7 to 1e-2 in 100 steps.
Asserted for hip32: bit-identical across two runs; equal to the CPU run to 5e-6 over the first ten iterations and 2e-4 over the first 25;
inside the envelope in which the CPU fp32 runs scatter at every iteration.  For the bf16 product:
Asserted: hip is bit-identical across two runs; all three agree to 5e-4 relative over iterations 1-20; over all 100 iterations hip
stays within the distance from f32 that bf16 storage itself causes (1.5 x the emu-vs-f32 distance + 2e-3 relative).  The measured
curve is printed and written to gpurun_out/parity100.json.  north_star's "< 1e-3 total-loss delta after 100 iterations" is
evaluated as an ABSOLUTE bound on the iteration-100 loss against f32 and reported (`north_star_abs_1e-3_vs_f32`; it held, 6.6e-4, in
the run recorded in DESIGN.md §4) but not asserted: the CPU bf16 emulation itself misses it (7.7e-3), so it is not a property a
bf16-storage implementation can guarantee.
"""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

ITERS = 100


def _cfg(depth=18):
    from bench import make_cfg

    cfg = make_cfg(depth)
    cfg.SOLVER.IMS_PER_BATCH = 2
    return cfg


def _build(seed, depth=18):
    from slenderobjdet_amd.modeling import build_model
    from slenderobjdet_amd.solver import build_lr_scheduler, build_optimizer

    cfg = _cfg(depth)
    torch.manual_seed(seed)
    model = build_model(cfg)
    model.train()
    opt = build_optimizer(cfg, model)
    return cfg, model, opt, build_lr_scheduler(cfg, opt)


def _cpu(data):
    return [{"image": d["image"].cpu(), "instances": d["instances"].to("cpu")} for d in data]


def _hip_run(pool):
    from bench import train_step

    cfg, model, opt, sched = _build(7)
    losses, lrs = [], []
    for it in range(ITERS):
        lrs.append(opt.param_groups[0]["lr"])
        losses.append(float(train_step(model, opt, pool[it % len(pool)])))
        sched.step()
    torch.cuda.synchronize()
    return losses, lrs, model.arena.params.detach().clone()


def _oracle_run(pool, lrs, emu, iters=ITERS):
    from oracle.model import OracleFCOS

    cfg, model, _opt, _sched = _build(7)
    oracle = OracleFCOS.from_hip_model(model, emulate_bf16=emu)
    del model
    cpu_pool, state, out = [_cpu(d) for d in pool], {}, []
    for it in range(iters):
        ref = oracle.losses(cpu_pool[it % len(cpu_pool)])
        total = sum(ref.values())
        grads = dict(zip(oracle.trainable().keys(), torch.autograd.grad(total, list(oracle.trainable().values()))))
        oracle.sgd_step(grads, state, lrs[it], cfg.SOLVER.MOMENTUM, cfg.SOLVER.WEIGHT_DECAY, cfg.SOLVER.WEIGHT_DECAY_NORM)
        out.append(float(total))
    return out


def _assert_fixture_is_current(chaos):
    """tests/golden/chaos100.json holds trajectories computed by oracle/model.py: a stale file (the oracle has changed since) would
    let iterations 7-100 of the comparison drift unnoticed behind the six live ones (round-3 advisor finding)."""
    import hashlib

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    stored = chaos.get("sources_sha256") or {}
    now = {f: hashlib.sha256(open(os.path.join(root, f), "rb").read()).hexdigest() for f in stored}
    changed = [f for f in stored if stored[f] != now[f]]
    assert stored and not changed, (f"tests/golden/chaos100.json was generated from other versions of {changed or 'unknown sources'}: run "
                                    "`python tests/golden/make_chaos100.py stamp` (re-checks the first iterations, re-stamps) or regenerate it")


def test_100_iteration_loss_parity(cuda):
    from slenderobjdet_amd.data import synthetic_batch
    from slenderobjdet_amd.layers import functional as HF

    pool = [synthetic_batch(2, 512, 512, 100 + i, device="cuda") for i in range(4)]
    prev = HF.DETERMINISTIC
    HF.DETERMINISTIC = True
    try:
        hip, lrs, params1 = _hip_run(pool)
        hip2, _, params2 = _hip_run(pool)
    finally:
        HF.DETERMINISTIC = prev
    # the same 100 iterations on the DEFAULT paths (float-atomic reductions: GroupNorm statistics gathered in the conv epilogue,
    # weight-gradient pixel splits, ...), i.e. what bench.py times: not bit-reproducible, held to the same bounds below
    HF.DETERMINISTIC = False
    try:
        fast, _, _ = _hip_run(pool)
    finally:
        HF.DETERMINISTIC = prev
    # the fp32-storage validation mode: same model code, fp32 tensors, sod_*_f32 kernels
    prev_p = HF.set_precision("fp32")
    try:
        hip32, _, p32a = _hip_run(pool)
        hip32b, _, p32b = _hip_run(pool)
    finally:
        HF.set_precision(prev_p)
    assert hip32 == hip32b and torch.equal(p32a, p32b), "fp32 validation mode is not reproducible"
    assert all(l == l for l in fast), "NaN loss (default paths)"
    assert all(l == l for l in hip), "NaN loss"
    # deterministic reductions (fixed-order weight-gradient slabs, GroupNorm / bias-gradient partials): two runs are the same run
    assert hip == hip2, [(i, a, b) for i, (a, b) in enumerate(zip(hip, hip2)) if a != b][:5]
    assert torch.equal(params1, params2)

    # The two CPU oracle trajectories come from tests/golden/chaos100.json (tests/golden/make_chaos100.py: the same initial weights, data
    # and schedule; two 100-iteration CPU runs cost the GPU box over a minute).  The first LIVE iterations of both oracles are re-run
    # here and must reproduce the file: the fp32 oracle to 2e-5 (host and thread count matter at 1e-6 while the run is still deterministic
    # in practice), the bf16-emulating one to 1e-3 (another summation order flips single bf16 roundings from the first step: 1e-5 at
    # iteration 1, 3e-4 by iteration 3 between this container and the GPU box's host).
    chaos = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "chaos100.json")))
    _assert_fixture_is_current(chaos)
    assert len(chaos["lrs"]) == ITERS and max(abs(a - b) for a, b in zip(chaos["lrs"], lrs)) < 1e-12, "schedule differs from the fixture's"
    emu, f32 = chaos["emu_all_threads"], chaos["f32_all_threads"]
    torch.set_num_threads(min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16))
    LIVE = 6
    live_emu, live_f32 = _oracle_run(pool, lrs, True, LIVE), _oracle_run(pool, lrs, False, LIVE)
    assert max(abs(a - b) for a, b in zip(live_emu, emu)) <= 1e-3, (live_emu, emu[:LIVE])
    assert max(abs(a - b) for a, b in zip(live_f32, f32)) <= 2e-5, (live_f32, f32[:LIVE])

    rel = lambda a, b: abs(a - b) / max(abs(b), 1e-6)
    rows = [{"iter": i + 1, "lr": lrs[i], "hip": hip[i], "hip_default_paths": fast[i], "hip_fp32_storage": hip32[i], "emu": emu[i], "f32": f32[i]}
            for i in range(ITERS)]
    d32 = [abs(a - b) for a, b in zip(hip32, f32)]
    worst_emu = max(rel(h, e) for h, e in zip(hip, emu))
    worst_f32 = max(rel(h, f) for h, f in zip(hip, f32))
    worst_store = max(rel(e, f) for e, f in zip(emu, f32))
    summary = {"iters": ITERS, "max_rel_hip_vs_emu": worst_emu, "max_rel_hip_vs_f32": worst_f32, "max_rel_emu_vs_f32": worst_store,
               "abs_delta_iter100_hip_vs_f32": abs(hip[-1] - f32[-1]), "abs_delta_iter100_hip_vs_emu": abs(hip[-1] - emu[-1]),
               "abs_delta_iter100_emu_vs_f32": abs(emu[-1] - f32[-1]), "north_star_abs_1e-3_vs_f32": abs(hip[-1] - f32[-1]) < 1e-3,
               "fp32_storage_mode": {"abs_delta_iter100_vs_f32": d32[-1], "max_abs_delta_all_iters_vs_f32": max(d32),
                                     "median_abs_delta_iters_81_100": sorted(d32[80:])[10], "north_star_abs_1e-3_vs_f32": d32[-1] < 1e-3},
               "default_paths": {"max_rel_vs_f32": max(rel(h, f) for h, f in zip(fast, f32)), "max_rel_vs_emu": max(rel(h, e) for h, e in zip(fast, emu)),
                                 "max_rel_vs_deterministic": max(rel(a, b) for a, b in zip(fast, hip)),
                                 "abs_delta_iter100_vs_f32": abs(fast[-1] - f32[-1])}}
    print("\nparity100:", json.dumps(summary))
    for r in rows[::10] + [rows[-1]]:
        print("  it %3d lr %.2e  hip %.6f  emu %.6f  f32 %.6f  hip32 %.6f  |hip-emu| %.2e  |hip-f32| %.2e  |emu-f32| %.2e  |hip32-f32| %.2e"
              % (r["iter"], r["lr"], r["hip"], r["emu"], r["f32"], r["hip_fp32_storage"], abs(r["hip"] - r["emu"]), abs(r["hip"] - r["f32"]),
                 abs(r["emu"] - r["f32"]), abs(r["hip_fp32_storage"] - r["f32"])))
    try:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        json.dump({"summary": summary, "curve": rows}, open(os.path.join(root, "gpurun_out", "parity100.json"), "w"), indent=1)
    except OSError:
        pass
    # (0) north_star: "total-loss delta < 1e-3 vs the reference's CPU path after 100 iterations".  Measured here and in
    # tests/golden/chaos100.json (tests/golden/make_chaos100.py: the CPU fp32 oracle of THIS run against itself with one thread instead of
    # all - another summation order - and against float64): no fp32 implementation has that property, the reference's own CPU path
    # included.  Two fp32 runs are identical to 1e-6 for ten iterations, 4e-5 apart at 20, 1e-3 at 60 and 2e-3 .. 8e-3 at 100 (the
    # learning rate grows 100x over the run and single ReLU decisions fall differently).  The fp32-storage mode of the product must
    # therefore (a) reproduce the CPU run while the run is still deterministic in practice, (b) stay inside the envelope in which correct
    # fp32 implementations of this run scatter (4x the largest pairwise distance among the three CPU runs so far, + 2e-4), and (c) end
    # within 5e-2 in any case; the statement that IS a property of the implementation at iteration 100 - given the same state, the same
    # loss and the same step - is asserted by test_one_step_parity_along_the_100_iteration_trajectory below.
    ca, cb, cc = chaos["f32_all_threads"], chaos["f32_one_thread"], chaos["f64"]
    env, worst = [], 0.0
    for i in range(ITERS):
        worst = max(worst, abs(ca[i] - cb[i]), abs(ca[i] - cc[i]), abs(cb[i] - cc[i]))
        env.append(worst)
    summary["fp32_storage_mode"]["cpu_fp32_pairwise_envelope_iter100"] = env[-1]
    assert max(d32[:10]) <= 5e-6 and max(d32[:25]) <= 2e-4, (max(d32[:10]), max(d32[:25]))
    for i in range(ITERS):
        assert d32[i] <= 4.0 * env[i] + 2e-4, (i + 1, d32[i], env[i])
    assert d32[-1] <= 5e-2, d32[-1]
    # (1) while the learning rate is tiny (warm-up iterations 1-20) the three runs are the same computation up to rounding
    early = max(max(rel(h, e), rel(h, f)) for h, e, f in zip(hip[:20], emu[:20], f32[:20]))
    assert early <= 5e-4, (early, summary)
    # (2) later, bf16 storage noise is amplified by training itself: two bf16 runs (hip, emu) drift from the fp32 run - and from each
    # other - by the same few 1e-3 (measured on MI355X: emu-vs-f32 5.2e-3, hip-vs-f32 6.4e-3, hip-vs-emu 1.2e-2 relative at worst,
    # all three at iterations 92-99).  The product may not be further from fp32 than bf16 storage alone explains:
    # (3 x: the two oracle runs are one pair of realisations of a chaotic run - their distance is itself a sample, measured between
    # 5.2e-3 and 7.7e-3 on this pool's hosts - and the HIP run has been 5.7e-3 .. 6.5e-3 from fp32; what pins the implementation at
    # iteration 100 is the one-step test below, not this ratio)
    assert worst_f32 <= 3.0 * worst_store + 2e-3, summary
    assert worst_emu <= 3.0 * worst_store + 2e-3, summary
    # (3) the default (float-atomic) paths.  They are not reproducible from run to run by construction, and a free run of this problem
    # amplifies any difference chaotically (above): a bound of the form "1.5 x what ONE other run shows" is then a coin toss at the
    # margin (one run in five of a day's full-suite runs: 1.15e-2 against a bound of 9.9e-3, with the deterministic run at 5.7e-3 and
    # the fp32-storage run at 1.27e-2 abs in the same session).  Same form as for the fp32 mode instead: identical to the reproducible
    # runs while the run is still deterministic in practice, and at EVERY iteration inside the envelope in which the three reproducible
    # realisations of the run (HIP deterministic, bf16-emulating oracle, fp32 oracle) have scattered so far (4 x + 2e-3 relative).
    early_fast = max(max(rel(h, e), rel(h, f)) for h, e, f in zip(fast[:20], emu[:20], f32[:20]))
    assert early_fast <= 5e-4, (early_fast, summary)
    worst_pair = 0.0
    for i in range(ITERS):
        worst_pair = max(worst_pair, abs(hip[i] - emu[i]), abs(hip[i] - f32[i]), abs(emu[i] - f32[i]))
        for other in (f32[i], emu[i], hip[i]):
            assert abs(fast[i] - other) <= 4.0 * worst_pair + 2e-3 * abs(other), (i + 1, fast[i], other, worst_pair, summary)


# iterations (1-based) at which the one-step check below runs: the first three, every tenth, and the last three of the 100
CHECK_AT = sorted(set([1, 2, 3] + list(range(10, 100, 10)) + [98, 99, 100]))


def _momentum_state(model, opt):
    """The fused optimizer's momentum arena as the oracle's per-parameter state dict (conv weights KRSC -> KCRS)."""
    mom = model.arena.momentum
    if mom is None or opt._steps == 0:
        return {}
    state = {}
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        off, n = model.arena.index[id(p)]
        b = mom[off:off + n].view(p.shape).detach().float().cpu()
        state[name] = (b.permute(0, 3, 1, 2) if b.dim() == 4 else b).contiguous().clone()
    return state


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_one_step_parity_along_the_100_iteration_trajectory(cuda, precision):
    """What CAN be asserted at iteration 100.  Two correct fp32 implementations of this run drift apart by themselves (tests/golden/
    chaos100.json: the CPU oracle against itself in another summation order, and against float64), so a free-running comparison measures
    the run's sensitivity, not the implementation.  Here the product trains freely for 100 iterations (reference schedule, momentum,
    weight decay), and at 15 of them - the first three, every tenth, the last three - the CPU fp32 oracle is handed the product's
    CURRENT state (parameters and momentum), takes the same step on the same batch, and must reproduce
      * the total loss of that iteration: 2e-5 relative in the fp32-storage mode, 1e-3 relative (north_star's number) for the bf16 product,
      * the step of that iteration (fp32-storage mode): the updated momentum buffers (= momentum * buffer + gradient + decay) to 1e-2 of
        their norm over the whole parameter vector and 5e-2 per tensor, the updated parameters to 1e-5 relative,
    i.e. the loss error of the implementation does not grow with training: at iteration 100 it is what it is at iteration 1."""
    from bench import train_step
    from oracle.model import OracleFCOS
    from slenderobjdet_amd.data import synthetic_batch
    from slenderobjdet_amd.layers import functional as HF

    pool = [synthetic_batch(2, 512, 512, 100 + i, device="cuda") for i in range(4)]
    cpu_pool = [_cpu(d) for d in pool]
    torch.set_num_threads(min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16))
    prev_p, prev_d = HF.set_precision(precision), HF.DETERMINISTIC
    HF.DETERMINISTIC = True
    worst_loss, worst_upd = 0.0, 0.0
    try:
        cfg, model, opt, sched = _build(7)
        for it in range(ITERS):
            lr = opt.param_groups[0]["lr"]
            check = (it + 1) in CHECK_AT
            if check:
                oracle = OracleFCOS.from_hip_model(model)
                state = _momentum_state(model, opt)
                before = {k: v.detach().clone() for k, v in oracle.trainable().items()}
            loss = float(train_step(model, opt, pool[it % len(pool)]))
            sched.step()
            if not check:
                continue
            ref = oracle.losses(cpu_pool[it % len(cpu_pool)])
            total = sum(ref.values())
            grads = dict(zip(oracle.trainable().keys(), torch.autograd.grad(total, list(oracle.trainable().values()))))
            oracle.sgd_step(grads, state, lr, cfg.SOLVER.MOMENTUM, cfg.SOLVER.WEIGHT_DECAY, cfg.SOLVER.WEIGHT_DECAY_NORM)
            rel = abs(loss - float(total)) / abs(float(total))
            worst_loss = max(worst_loss, rel)
            assert rel <= (2e-5 if precision == "fp32" else 1e-3), (it + 1, loss, float(total), rel)
            if precision == "fp32":
                # the step direction: the momentum buffers after the step (buf = momentum * buf + grad + wd * theta; comparing theta itself
                # would mostly compare fp32 rounding of theta - lr * buf at the warm-up learning rates), whole vector and per tensor
                after, hip_state = oracle.trainable(), _momentum_state(model, opt)
                num = den = 0.0
                for name, p in model.named_parameters():
                    if not p.requires_grad:
                        continue
                    b_ref, b_hip = state[name].double(), hip_state[name].double()
                    n2, d2 = float(b_ref.pow(2).sum()), float((b_hip - b_ref).pow(2).sum())
                    num, den = num + d2, den + n2
                    # per tensor: a gross error (a missing factor, a lost term) is 1e-1 .. 1; single ReLU decisions falling differently
                    # under another summation order move small tensors (biases: sums of signed terms) by up to ~1e-2
                    assert (d2 / max(n2, 1e-60)) ** 0.5 <= 5e-2, (it + 1, name, (d2 / max(n2, 1e-60)) ** 0.5)
                    q = p.detach().float().cpu()
                    q = q.permute(0, 3, 1, 2) if q.dim() == 4 else q
                    # theta_new = theta - lr * buffer on both sides: the parameters may differ by lr * (buffer difference) + rounding
                    atol = 1e-6 + 1.01 * lr * float((b_hip - b_ref).abs().max())
                    assert torch.allclose(q, after[name].detach(), rtol=1e-5, atol=atol), (it + 1, name, float((q - after[name].detach()).abs().max()), atol)
                glob = (num / den) ** 0.5
                worst_upd = max(worst_upd, glob)
                assert glob <= 1e-2, (it + 1, glob)           # the step direction of the whole parameter vector (measured <= 2.6e-3)
    finally:
        HF.set_precision(prev_p)
        HF.DETERMINISTIC = prev_d
    print(f"\none-step parity along the trajectory ({precision}): worst loss delta {worst_loss:.2e} relative, worst update distance {worst_upd:.2e}")


FROZEN_PREFIX = ("backbone.bottom_up.stem", "backbone.bottom_up.res2")


_LONG = pytest.mark.skipif(os.environ.get("SOD_LONG_TESTS") != "1", reason="10-minute run (100 CPU-oracle iterations of R50 at 384x384): SOD_LONG_TESTS=1")


@pytest.mark.parametrize("mode,depth,iters,size", [("fp32", 18, ITERS, 512), ("fp32", 50, 30, 256), ("bf16", 18, ITERS, 512), ("bf16det", 18, ITERS, 512),
                                                   ("bf16", 50, 30, 256), pytest.param("bf16", 50, 100, 384, marks=_LONG)])
def test_free_run_on_shared_relu_decisions(cuda, mode, depth, iters, size):
    """north_star's sentence as a statement that CAN hold: "total-loss delta < 1e-3 vs the reference's CPU path after 100 iterations".
    Free-running fp32 implementations of this run end 2e-3 ... 1.3e-2 apart (test above, tests/golden/chaos100.json) - and round 4
    found out why: not rounding growth, but single ReLU decisions on pre-activations that cancel to within rounding of zero, which fall
    differently in every implementation and, through the sparse regression gradient, move whole weight-gradient tensors by 1e-3 ... 1e-2
    (tests/test_gpu_f32_mode.py).  Here the product and the CPU fp32 oracle BOTH run free - own parameters, own momentum, same schedule -
    and the oracle takes, in every iteration, the product's ReLU decisions for that batch (RELU_TAP of the product's layer code ->
    oracle.nn.ForcedMasks).  Both then descend the same piecewise-linear function.  The decisions are checked, not trusted: they may
    differ from the oracle's own only inside the undecided band |x| < tau * rms(x) of an activation (ForcedMasks ``outside``).

    * ``fp32, 18``  the fp32-storage validation mode on BASELINE configs[0], 100 iterations: |delta total loss| < 1e-3 at iteration 100 -
      north_star's number - and < 1e-5 at EVERY iteration (measured 4.8e-7 at worst); tau 1e-4.
    * ``fp32, 50``  the same on the R50 family (configs[1] at 256x256, 30 iterations; round-4 verdict).
    * ``bf16, 18``  the bf16 PRODUCT path (the MFMA kernels bench.py times, default float-atomic reductions) on configs[0], 100 iterations
      against the plain fp32 oracle on the product's decisions: what is left is bf16 storage rounding, amplified by 100 SGD steps.  The
      iteration-100 delta is REPORTED (gpurun_out/parity100_shared_relu_bf16_18.json) and held to the bound measured
      for it; the fused frozen kernels (stem + pool) keep their ReLUs inside, so the oracle decides those itself (no gradient flows there).
      The default reductions are float atomics, so the run differs from launch to launch, and late in the run (iterations 90 - 100, when the
      warm-up has raised the learning rate 100-fold) single iterations spike: launches on MI355X boxes measured maxima of 1.3e-3, 1.8e-3,
      1.3e-3, 6.1e-3 and 5.9e-3 (the last two: one-iteration spikes at iteration 100, one of a run that stood at 3e-5 ten iterations earlier) - the
      decisions found outside the 0.25-rms band all sit in the LAST GroupNorm of the classification tower at P3, iterations 97 - 100.  A
      max-over-iterations bar at the measured level is therefore a coin flip (it turned the round-6 rehearsal of the driver's run red);
      asserted instead: the 90th percentile of the 100 deltas (what the run does apart from single spikes) and an order-of-magnitude cap
      on the maximum (a wrong gradient separates the runs by 1e-1 within tens of iterations).
    * ``bf16det, 18``  the same run with the product's DETERMINISTIC reductions (fixed-order slabs instead of float atomics: every other
      kernel is the same): bit-reproducible on every MI355X, so its maximum can be held to a bound near the value measured for it (3.06e-3 at worst, 1.16e-3 at iteration 100).
    * ``bf16, 50, 100, 384``  (opt-in, SOD_LONG_TESTS=1) north_star's model at its 100 iterations: FCOS R50-FPN bf16 product path against the fp32
      oracle at 384x384; measured once per round and recorded in DESIGN.md section 8 (profiles/r6_parity100_bf16_r50.json: median 3.7e-4,
      90th percentile 2.8e-3, 1.4e-2 at iteration 100 - the deeper network separates earlier than R18 once the warm-up has raised the rate).
    * ``bf16, 50``  the bf16 product path on the R50 family (bottleneck blocks: the persistent 1x1 kernel, 1-bit ReLU masks, the fused
      frozen res2 blocks), 30 iterations at 256x256, same bound; reported in gpurun_out/parity100_shared_relu_bf16_50.json."""
    from bench import train_step
    from oracle.conditioning import ProductReluTap
    from oracle.model import OracleFCOS
    from oracle.nn import ForcedMasks
    from slenderobjdet_amd.data import synthetic_batch
    from slenderobjdet_amd.layers import functional as HF

    pool = [synthetic_batch(2, size, size, 100 + i, device="cuda") for i in range(4)]
    cpu_pool = [_cpu(d) for d in pool]
    torch.set_num_threads(min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16))
    det = mode in ("fp32", "bf16det")
    tag, mode = mode, ("bf16" if mode == "bf16det" else mode)
    prev_p, prev_d = HF.set_precision(mode), HF.DETERMINISTIC
    HF.DETERMINISTIC = det
    tau = 1e-4 if mode == "fp32" else 0.25       # bf16: the product's pre-activations carry ~1e-2 rms of storage rounding
    hip, ora, flipped, units, worst_ratio, outside_late, outside_log = [], [], 0, 0, 0.0, 0, []
    try:
        cfg, model, opt, sched = _build(7, depth)
        oracle, state = OracleFCOS.from_hip_model(model), {}
        for it in range(iters):
            lr = opt.param_groups[0]["lr"]
            with ProductReluTap() as tap:
                hip.append(float(train_step(model, opt, pool[it % len(pool)]).detach()))
            sched.step()
            masks, unmatched = tap.masks_for(model)
            assert not unmatched, (it + 1, unmatched[:3])
            st = ForcedMasks.begin(masks, tau=tau)
            try:
                ref = oracle.losses(cpu_pool[it % len(cpu_pool)])
                total = sum(ref.values())
                grads = dict(zip(oracle.trainable().keys(), torch.autograd.grad(total, list(oracle.trainable().values()))))
            finally:
                ForcedMasks.end()
            # every ReLU a gradient flows through took the product's decision (the fp32 mode reports all of them)
            missed = [k for k in st["missed"] if mode == "fp32" or not str(k[0]).startswith(FROZEN_PREFIX)]
            assert not missed, (it + 1, missed[:5])
            # the shared decisions are checked, not trusted: they may differ from the oracle's own only inside the undecided band.  The two
            # fp32 runs stay within 1e-6 of each other for all iterations; the bf16 product drifts away from the fp32 oracle (own parameters
            # on both sides), so its band is asserted while the runs are still one trajectory (10 iterations) and reported afterwards
            if mode == "fp32" or it < 10:
                assert st["outside"] == 0, (it + 1, st["outside_at"][:5])
            # for the WHOLE run (round-5 advisor): a systematic error in a mask path (a wrong-sign pre-activation, a shifted bit) flips a large
            # share of some layer's units in every iteration, also after the two runs have separated - the share of units decided
            # differently stays below 3 % per iteration (measured 0.25 - 0.4 % on R18, up to 1.3 % on R50 in iterations 90 - 100 of the free run:
            # units whose pre-activation the bf16 storage noise and the separation of the two runs can flip), and
            # what falls outside the undecided band stays a sliver (measured: at most 2 440 of 3.8e7 units in one iteration, late in the run,
            # all in the last GroupNorm of the classification tower)
            assert st["disagree"] <= 0.03 * max(st["units"], 1), (it + 1, st["disagree"], st["units"])
            # (the opt-in R50 run separates further in its last iterations - 1.4e-2 of loss at iteration 100 - and two launches of it measured
            # 23 653 units over the whole run and 48 333 of 4.0e7 in iteration 98 alone, most of them in the first GroupNorm of the
            # classification tower: three slivers there)
            assert st["outside"] <= (1e-3 if depth == 18 or iters <= 30 else 3e-3) * st["units"], (it + 1, st["outside"], st["outside_at"][:5])
            outside_late += st["outside"]
            outside_log += [(it + 1,) + tuple(r) for r in st["outside_at"]]      # (iteration, ReLU position, call = FPN level, units, |x| / rms)
            worst_ratio = max([worst_ratio] + [r[3] for r in st["outside_at"]])
            flipped += st["disagree"]
            units += st["units"]
            oracle.sgd_step(grads, state, lr, cfg.SOLVER.MOMENTUM, cfg.SOLVER.WEIGHT_DECAY, cfg.SOLVER.WEIGHT_DECAY_NORM)
            ora.append(float(total.detach()))
            if (it + 1) % 10 == 0:      # a sign of life for harnesses that kill silent runs (the long variants take minutes)
                try:
                    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
                    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
                    with open(os.path.join(root, "gpurun_out", "parity100_progress.txt"), "a") as f:
                        f.write(f"{tag} R{depth} {size}: iteration {it + 1} of {iters}, |delta| {abs(hip[-1] - ora[-1]):.2e}\n")
                except OSError:
                    pass
    finally:
        HF.set_precision(prev_p)
        HF.DETERMINISTIC = prev_d
    d = [abs(a - b) for a, b in zip(hip, ora)]
    p90 = sorted(d)[int(0.9 * (len(d) - 1))]
    print(f"\nfree run on shared ReLU decisions [{tag}, R{depth}, {iters} iterations]: |product - cpu32| at iterations 1, 10, 20, ...:",
          " ".join(f"{d[i]:.1e}" for i in [0] + list(range(9, iters, 10))), f" max {max(d):.2e}  p90 {p90:.2e}  last {d[-1]:.2e}  loss {hip[0]:.4f} -> {hip[-1]:.4f}",
          f" units decided differently: {flipped} of {units}; outside the {tau:g} rms band: {outside_late} (largest |x| / rms {worst_ratio:.2f})")
    try:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        name = "parity100_shared_relu.json" if (mode, depth) == ("fp32", 18) else f"parity100_shared_relu_{tag}_{depth}.json"
        json.dump({"mode": mode, "depth": depth, "product": hip, "cpu32_on_the_products_relu_decisions": ora, "abs_delta": d,
                   "units_decided_differently": flipped, "units": units, "tau": tau, "outside_band": outside_late, "largest_ratio": worst_ratio, "outside_at": outside_log[:400]},
                  open(os.path.join(root, "gpurun_out", name), "w"), indent=1)
    except OSError:
        pass
    assert all(x == x for x in hip)
    if mode == "fp32":
        assert d[-1] < 1e-3, d[-1]                      # north_star's bound ...
        assert max(d) < 1e-5, max(d)                    # ... and what was measured: 4.8e-7 at worst over 100 iterations (fp32 ulps of a loss of ~2)
    elif det or iters < 100:
        assert max(d) < BF16_SHARED_BOUND, (max(d), d[-1])
    elif depth == 18:
        assert p90 < BF16_SHARED_P90 and max(d) < BF16_SHARED_SPIKE, (p90, max(d), d[-1])
    else:       # the opt-in R50 run: measured median 3.7e-4, p90 2.8e-3, 1.4e-2 at iteration 100 - a reported figure with a sanity cap
        assert p90 < 1e-2 and max(d) < 5e-2, (p90, max(d), d[-1])
    if iters >= 100:
        assert hip[-1] < hip[0] - 0.3                   # and the run trained (2.84 -> 1.9 in the free runs above)


# |bf16 product - fp32 oracle| over the 100 free iterations on shared decisions, measured in round 5: 1.1e-3 at
# iteration 100, 1.3e-3 at worst (loss 2.844 -> 1.929) - the bf16 STORAGE floor of this run, at north_star's 1e-3 and not reliably under it
# (the default reductions are float atomics: the figure moves in the fourth digit from run to run).  Bound = 4 x the measurement.
BF16_SHARED_BOUND = 5e-3
# the atomic-reduction run of 100 iterations: 90th percentile of the per-iteration deltas (measured 7e-4 ... 1.1e-3) and the spike cap
BF16_SHARED_P90 = 2.5e-3
BF16_SHARED_SPIKE = 3e-2
