"""Launcher / subprocess tests of the GPU suite: train_net.py's CLI, ``bench.py --gpus 2`` (two ranks sharing cuda:0 over gloo), the
two-ranks-equal-one-process equivalence and the one-rank RCCL rehearsal.  They start child processes and rendezvous, so they live in the
LAST file pytest collects (``zz``): under ``pytest -x`` every oracle-comparing test of the suite has run before any of them can fail for
a reason that has nothing to do with arithmetic.  No test here guesses a TCP port: one-rank groups use an in-process HashStore, N-rank
launches let torch.distributed.run's c10d rendezvous bind port 0 itself (bench.py: launcher_cmd)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_train_net_cli_runs(cuda, tmp_path):
    """train_net.py with the reference's CLI: config file + overrides, 3 iterations on synthetic batches."""
    import subprocess
    import sys
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "train_net.py"), "--config-file", os.path.join(root, "configs/fcos/fcos_R_50_FPN_1x.yaml"),
           "--num-gpus", "1", "MODEL.RESNETS.DEPTH", "18", "MODEL.RESNETS.RES2_OUT_CHANNELS", "64", "SOLVER.IMS_PER_BATCH", "2",
           "SOLVER.MAX_ITER", "3", "INPUT.MIN_SIZE_TRAIN", "(256,)", "INPUT.MAX_SIZE_TRAIN", "320", "OUTPUT_DIR", str(tmp_path)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "iter: 3" in out.stdout and "cls_loss" in out.stdout


def test_bench_two_ranks_data_parallel(cuda, tmp_path):
    """``python bench.py --gpus 2`` spawns 2 ranks itself (both on cuda:0, gloo transport — this box has one GPU): the
    bucketed gradient all-reduce, the normaliser all-reduce, parameter broadcast and the JSON contract all execute, and the
    two ranks end with bit-identical parameters."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SOD_BENCH_SHARE_GPU="1", SOD_BENCH_DUMP_PARAMS=str(tmp_path))
    # bench.py starts its own ranks (as train_net.py does through detectron2's launch): no launcher around it here
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--depth", "18", "--batch-per-gpu", "2",
           "--height", "256", "--width", "320", "--no-roofline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 4 and d["scaling"] == "weak" and d["value"] > 0
    c = d["config"]       # the line says what the collective layer saw
    assert c["ranks_seen"] == 2 and c["backend"] == "gloo" and c["n_buckets"] >= 1 and c["wire_dtype"] == "float32"
    assert c["exposed_comm_ms_per_step"] is not None and c["exposed_comm_ms_per_step"] >= 0 and c["grad_bytes_per_step"] > 0
    p0 = torch.load(os.path.join(tmp_path, "params_rank0.pt"))
    p1 = torch.load(os.path.join(tmp_path, "params_rank1.pt"))
    assert torch.equal(p0, p1), "replicas diverged: gradient all-reduce / broadcast is broken"


def test_two_ranks_equal_one_process_on_the_union(cuda, tmp_path):
    """2 ranks x 2 images == 1 process x the same 4 images (tools/dp_equivalence.py, deterministic mode, fp32 wire, both ranks on
    cuda:0 over gloo): the mean of the ranks' losses is the one-process loss to 1e-6 relative, and the parameters after ONE step agree
    to fp32 rounding - what the folded [num_pos, sum centerness] all-reduce (fcos/utils.py:10-19, fcosv2.py:115-118, 132-133) and the
    gradient SUM x 1 / world (train_net.py:185-195) are there for.  Per-image work is identical in both launches (FrozenBN, GroupNorm);
    only the fp32 summation order of the weight gradients over the batch differs."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(root, "tools", "dp_equivalence.py")
    env = dict(os.environ, SOD_DETERMINISTIC="1", SOD_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    one, two = str(tmp_path / "one"), str(tmp_path / "two")
    out = subprocess.run([sys.executable, script, "--out", one], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    from bench import launcher_cmd
    out = subprocess.run(launcher_cmd(2) + [script, "--out", two], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    ref = torch.load(os.path.join(one, "rank0.pt"))
    r0, r1 = torch.load(os.path.join(two, "rank0.pt")), torch.load(os.path.join(two, "rank1.pt"))
    assert ref["world"] == 1 and r0["world"] == 2
    assert torch.equal(r0["params"], r1["params"]), "replicas diverged"
    for k, v in ref["loss"].items():
        mean = 0.5 * (r0["loss"][k] + r1["loss"][k])
        assert abs(mean - v) <= 1e-6 * max(abs(v), 1e-3), (k, mean, v, r0["loss"][k], r1["loss"][k])
    assert r0["loss"] != r1["loss"]                       # the ranks really saw different images
    d = (r0["params"] - ref["params"]).abs().max().item()
    scale = ref["params"].abs().max().item()
    assert d <= 2e-6 * scale, (d, scale)
    # and the step was a real one: the update itself is orders of magnitude above that bound for some parameter
    torch.manual_seed(1)
    from bench import make_cfg
    from slenderobjdet_amd.modeling import build_model
    init = build_model(make_cfg(18)).arena.params.detach().float().cpu()
    assert (ref["params"] - init).abs().max().item() > 100 * max(d, 1e-9)


def test_bench_rccl_rehearsal_single_rank(cuda):
    """``python bench.py --rccl-rehearsal``: a ONE-rank RCCL ("nccl") process group on the real GPU with every data-parallel
    collective issued anyway - parameter broadcast, the asynchronous normaliser all-reduce waited for in front of the loss node, the
    bucketed gradient all-reduce launched from the autograd thread on its own stream (fp32 and bf16 wire formats), barriers.  With
    one rank every reduction is the identity, so the run must reproduce the plain single-process run's loss; what it proves is that
    the RCCL calls, handles and stream hand-overs of the N-GPU path execute (the two-rank test of this box runs them over gloo)."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--depth", "18", "--batch-per-gpu", "2",
            "--height", "256", "--width", "320", "--no-roofline", "--no-cpu-baseline"]

    def run(extra, env_extra=None):
        env = dict(os.environ, SOD_DETERMINISTIC="1", **(env_extra or {}))
        out = subprocess.run(base + extra, capture_output=True, text=True, timeout=900, env=env, cwd=root)
        assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
        return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])

    plain = run([])
    reh = run(["--rccl-rehearsal"])
    assert reh["config"].get("rccl_rehearsal") and reh["n_gpus"] == 1
    assert reh["config"]["backend"] == "nccl" and reh["config"]["ranks_seen"] == 1 and reh["config"]["n_buckets"] >= 1
    assert reh["config"]["final_loss"] == plain["config"]["final_loss"], (reh["config"], plain["config"])
    reh16 = run(["--rccl-rehearsal"], {"SOD_GRAD_BUCKET_DTYPE": "bf16"})
    assert abs(reh16["config"]["final_loss"] - plain["config"]["final_loss"]) <= 2e-2 * abs(plain["config"]["final_loss"])
    # Stream hand-over of the bucket reducer: without a weight-gradient side stream nothing orders the comm stream behind the MAIN
    # compute stream except the reducer's own wait (arena._launch_bucket) when the launching node ran on the tower stream.  On the bf16
    # wire the bucket is COPIED (rounded) on the comm stream and written back after the reduction, so a bucket read before the main
    # stream's gradients were complete would overwrite them with stale values: the deterministic run must not change.
    for env_extra in ({"SOD_WGRAD_STREAM": "0", "SOD_TOWER_STREAMS": "1"}, {"SOD_WGRAD_STREAM": "0", "SOD_TOWER_STREAMS": "0"}):
        other = run(["--rccl-rehearsal"], dict(env_extra, SOD_GRAD_BUCKET_DTYPE="bf16"))
        assert other["config"]["final_loss"] == reh16["config"]["final_loss"], (env_extra, other["config"], reh16["config"])
    # one hardware queue per stream for a rank (utils/comm.py::prepare_rank_env) unless the environment says otherwise; the emulated RCCL
    # occupancy (sod_debug_occupy behind every bucket) changes the timing, never the result
    assert reh["config"]["hw_queues"] == "6" and "hw_queues" not in plain["config"]
    occ = run(["--rccl-rehearsal", "--rehearsal-occupancy", "16:300", "--hw-queues", "4"])
    assert occ["config"]["hw_queues"] == "4" and occ["config"]["rehearsal_occupancy"].startswith("16:300")
    assert occ["config"]["final_loss"] == plain["config"]["final_loss"], (occ["config"], plain["config"])
