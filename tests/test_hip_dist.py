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
    assert line["n_gpus"] == 2 and line["dtype"] == "bf16"
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
