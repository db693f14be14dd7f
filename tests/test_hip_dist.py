"""The N-rank HIP path of bench.py on ONE GPU: `python bench.py --gpus 2 --dist-backend gloo` launched WITHOUT torchrun must start
its two ranks itself (fresh child processes, before any GPU call), run the MI355X kernels in both, take the MAX over ranks and print
ONE JSON line with n_gpus == 2.  With --train the parameter gradients go through OverlappedGradReducer while backward is still
producing them (the hand-off of BlockFn.backward on the autograd thread), reduced over gloo here, over RCCL (`nccl`) on a multi-GPU node."""
import json
import os
import subprocess
import sys

import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu


def _run(*extra):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--steps", "2", "--warmup", "1",
           "--batch", "8", "--model", "vit_tiny", "--no-cpu-baseline", *extra]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=REPO)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.timeout(900)
def test_bench_self_launches_two_ranks_forward():
    line = _run()
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 16 and line["scaling"] == "weak"
    assert line["value"] > 0 and line["dtype"] == "f16" and line["config"]["range_guard_fallbacks"] == 0
    assert any(k.startswith("pv_gemm_bf16") for k in line["kernels"]) and "pv_attention_bf16" in line["kernels"]      # the HIP path ran


@pytest.mark.timeout(900)
def test_bench_self_launches_two_ranks_training_with_overlapped_allreduce():
    line = _run("--train")          # bench.py itself asserts finite logits and finite parameter gradients on every rank
    assert line["n_gpus"] == 2 and line["dtype"] == "f16"          # (round 5: training runs the fp16 operand library with a loss scale)
    ga = line["grad_allreduce"]
    assert ga["buckets_per_step"] >= 1
    assert "pv_attention_bwd_bf16" in line["kernels"] and "pv_layernorm_bwd" in line["kernels"]                      # HIP backward ran


@pytest.mark.timeout(900)
def test_overlapped_reducer_launches_buckets_during_hip_backward():
    """Small buckets (64 KiB): gradients of the last blocks are all-reduced while the HIP backward still runs the earlier blocks."""
    line = _run("--train", "--bucket-kib", "64")
    ga = line["grad_allreduce"]
    assert ga["buckets_per_step"] > 4 and ga["launched_during_backward_total"] >= 3 * (ga["buckets_per_step"] - 1)    # 1 warm-up + 2 + 2 steps


def test_bench_refuses_a_rank_count_that_does_not_match():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=REPO)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout) and '"n_gpus"' not in r.stdout


_RCCL_ONE_RANK = r'''
import os, sys, json
sys.path.insert(0, os.environ["PV_REPO"])
import torch, torch.distributed as td
from peekvit_amd import dist as pvdist, synth
from peekvit_amd.models.vit import VisionTransformer
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
td.init_process_group("nccl", device_id=dev)                  # RCCL: the backend the multi-GPU job uses
cfg = synth.MODEL_CONFIGS["vit_tiny"]
m = VisionTransformer(**cfg); synth.load_synth_weights(m, cfg); m = m.to(dev).train()
x = torch.randn(4, 3, cfg["image_size"], cfg["image_size"], device=dev); y = torch.arange(4, device=dev)
torch.nn.functional.cross_entropy(m(x), y).backward()
ref = [p.grad.clone() for p in m.parameters()]
for p in m.parameters(): p.grad = None
red = pvdist.OverlappedGradReducer(m.parameters(), bucket_bytes=64 << 10)
torch.nn.functional.cross_entropy(m(x), y).backward()
n = red.finish()
same = all(torch.equal(a, p.grad) for a, p in zip(ref, m.parameters()))     # one rank: sum / 1 = the local gradient, through RCCL and back
t = torch.tensor([1.5, 2.5], device=dev, dtype=torch.float64)
td.all_reduce(t, op=td.ReduceOp.MAX)                                     # bench.py's MAX-over-ranks timing reduction
td.barrier(); torch.cuda.synchronize()
print(json.dumps({"buckets": n, "during_backward": red.launched_before_finish, "same": same, "max": t.tolist(), "backend": td.get_backend()}))
td.destroy_process_group()
'''


@pytest.mark.timeout(600)
def test_rccl_process_group_with_one_rank_runs_the_overlapped_reducer():
    """The N > 1 code path over the REAL backend (torch.distributed "nccl" = RCCL), as far as one GPU allows: a 1-rank RCCL process group
    (init with device_id, barrier, the MAX all-reduce of bench.py's timings) and the bucketed, backward-overlapped gradient all-reduce of
    the HIP training path reducing in place on the device - gradients come back bit-identical."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               PV_REPO=REPO)
    r = subprocess.run([sys.executable, "-c", _RCCL_ONE_RANK], capture_output=True, text=True, timeout=500, env=env, cwd=REPO)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["backend"] == "nccl" and out["same"] and out["buckets"] > 4 and out["during_backward"] >= out["buckets"] - 1 and out["max"] == [1.5, 2.5]
