#!/usr/bin/env python3
"""Headline benchmark: images/sec of the ViT-B/16 forward at batch 2048 per GPU, 224x224 (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Both forms work: started WITHOUT a launcher and with --gpus N > 1, this process - before it makes any GPU call - starts
`python -m torch.distributed.run` with N ranks as a child, relays the child's single JSON line and exits with its code.

One process per GPU.  A "step" is ONE forward pass of the hot path (VisionTransformer.forward on the MI355X
kernels) over one device-resident synthetic batch.  The path shards along the batch: every
rank runs an independent replica on its own 2048-image batch, no data-path collective ("scaling": "weak");
the only communication is the timing barrier + MAX-over-ranks reduction over RCCL.

Prints ONE JSON line (rank 0) with the contract keys plus:
  roofline      the dominant kernel (pv_gemm_bf16): algorithmic FLOPs / HIP-event time, measured live
  cpu_baseline  the CPU oracle (oracle/vit_oracle.py, fp32, = the reference's arithmetic) timed on the host cores
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch

MFMA_BF16_PEAK_TFLOPS = 2500.0      # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
HBM_PEAK_GBS = 8000.0               # HBM3E spec, same guide
# measured on this box (scripts/mfma_peak.hip, profiles/r01_mfma_peak.txt): register-only MFMA loop with data-like operand toggling;
# reported next to the nominal peak, never used for `frac`
MFMA_BF16_SUSTAINED_MEASURED_TFLOPS = 2003.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="vit_b_16")
    ap.add_argument("--batch", type=int, default=2048, help="images per GPU per step")
    ap.add_argument("--rank-budget", type=float, default=None, help="run RankViT (rankvit_layers 3,6,9) at this budget")
    ap.add_argument("--train", action="store_true",
                    help="BASELINE.json configs[2]/[4]: a step = forward + cross-entropy + backward (HIP backward kernels); with N > 1 "
                         "ranks the parameter gradients are all-reduced over RCCL (data-parallel training path)")
    ap.add_argument("--no-optimizer", action="store_true",
                    help="--train: stop the step after loss.backward() (+ all-reduce).  Default: the reference's whole step (train/train.py:112-121) "
                         "- zero_grad, forward, cross-entropy, backward, [all-reduce], clip_grad_norm_(1.0), Adam(1e-3).step()")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=64, help="images of the CPU-oracle sample (64: large enough that the GPU path scored "
                                                                "against it takes the same kernels / LayerNorm folding as the timed batch)")
    ap.add_argument("--cpu-iters", type=int, default=4)
    ap.add_argument("--precision", default="auto", choices=["auto", "bf16", "f16", "bf16x3"],
                    help="operand precision of the MFMA products: auto (default = the package default: IEEE fp16 operands behind the "
                         "operand-range guard with bf16 fallback for inference - meets the 1e-3 logits tolerance; bf16 operands for "
                         "training), bf16 (4e-3), f16 (unguarded), bf16x3 (split operands, 1e-5)")
    ap.add_argument("--streams", type=int, default=0, help="forward on this many batch slices / HIP streams (0 = package default)")
    ap.add_argument("--bucket-kib", type=int, default=25 << 10, help="gradient all-reduce bucket size (KiB), --train with N > 1")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL over xGMI); gloo only to rehearse N>1 on one GPU")
    return ap.parse_args()


def cpu_baseline(cfg, batch, iters, gpu_model=None, dev=None):
    """The CPU oracle (bit-equal to the reference on CPU, tests/test_oracle_golden.py) on a bounded sample.  As the checker it
    also scores the GPU path on the same sample: relative L2 of the logits per operand-precision mode."""
    from oracle import vit_oracle as O
    from peekvit_amd import synth
    sd = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(cfg).items()}
    x = torch.randn(batch, 3, cfg["image_size"], cfg["image_size"], generator=torch.Generator().manual_seed(0))
    # the GPU box gives one-GPU jobs a 16-CPU share of a 256-thread host: size the pool to the share, not the host
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, int(os.environ.get("PV_CPU_THREADS", "16"))))
    torch.set_num_threads(cores)
    with torch.no_grad():
        ref = O.vit_forward(x, sd, cfg, "fp32")                 # warm-up
        t0 = time.perf_counter()
        for _ in range(iters):
            O.vit_forward(x, sd, cfg, "fp32")
        dt = time.perf_counter() - t0
    out = {"value": round(batch * iters / dt, 2), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
           "sample": f"{iters} forwards of batch {batch} (fp32, torch CPU, oracle/vit_oracle.py), {dt:.1f} s, on {torch.get_num_threads()} "
                     f"threads (this job's CPU share) of a host with {os.cpu_count()} hardware threads"}
    if gpu_model is not None:
        from peekvit_amd import engine
        err = {}
        with torch.no_grad():
            for mode in ("auto", "bf16", "f16"):
                with engine.precision(mode):
                    got = gpu_model(x.to(dev)).float().cpu()
                err[mode] = float(f"{((got - ref).norm() / ref.norm()).item():.3e}")
        out["gpu_logits_rel_l2_vs_oracle"] = err       # tolerance of BASELINE.json: 1e-3
    return out


def self_launch(args) -> int:
    """--gpus N > 1 without a launcher: start N ranks as a CHILD `python -m torch.distributed.run` (this process has made no GPU
    call - replacing a process that has initialised the GPU is forbidden on this pool), relay its output, return its exit code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), PV_BENCH_CHILD="1")
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    for ln in proc.stdout.splitlines():
        if ln not in lines:
            print(ln, file=sys.stderr)
    if proc.returncode == 0 and len(lines) == 1:
        print(lines[0], flush=True)
        return 0
    print(f"bench.py: the {args.gpus}-rank run failed (exit {proc.returncode}, {len(lines)} result lines)", file=sys.stderr)
    return proc.returncode or 1


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's rank count and --gpus must agree")
    dist = world > 1
    ndev = torch.cuda.device_count()
    if args.dist_backend == "gloo":
        local = local % max(ndev, 1)       # rehearsal: several ranks may share one GPU
    elif ndev < world:
        raise SystemExit(f"--gpus {world} over RCCL needs {world} visible GPUs, found {ndev} (rehearse with --dist-backend gloo)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if dist:
        import torch.distributed as td
        if args.dist_backend == "nccl":
            td.init_process_group("nccl", device_id=dev)
        else:
            td.init_process_group("gloo")

    from peekvit_amd import ops, synth
    from peekvit_amd.models.rankvit import RankVisionTransformer
    from peekvit_amd.models.vit import VisionTransformer

    cfg = synth.MODEL_CONFIGS[args.model]
    seqs = None
    if args.rank_budget is None:
        model = VisionTransformer(**cfg)
        workload = f"{args.model} forward, batch {args.batch}/GPU, {cfg['image_size']}x{cfg['image_size']}"
    else:
        layers = [3, 6, 9] if cfg["num_layers"] >= 10 else list(range(1, cfg["num_layers"]))
        model = RankVisionTransformer(**cfg, rankvit_layers=layers)
        model.set_budget(args.rank_budget)
        import math
        S, seqs = synth.seq_length(cfg), []
        for i in range(cfg["num_layers"]):
            if i in layers and args.rank_budget != 1:
                S = 1 + math.ceil((S - 1) * args.rank_budget)
            seqs.append(S)
        workload = f"rank{args.model} layers={layers} budget={args.rank_budget} forward, batch {args.batch}/GPU"
    synth.load_synth_weights(model, cfg)
    model = (model.train() if args.train else model.eval()).to(dev)
    infer_model = model
    flops_img = synth.fwd_flops_per_image(cfg, seqs) * (3 if args.train else 1)      # backward = dgrad + wgrad = 2x forward
    if args.train:
        workload = workload.replace("forward", "train step (fwd, cross-entropy, bwd" + (f", gradient all-reduce over {args.dist_backend}" if world > 1 else "")
                                    + (")" if args.no_optimizer else ", clip 1.0, Adam)"))

    # random (never zero-filled) device-resident input, bf16-representable like the parity fixtures
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    x = torch.randn(args.batch, 3, cfg["image_size"], cfg["image_size"], generator=gen, device=dev)
    x = x.to(torch.bfloat16).to(torch.float32)

    def barrier():
        if dist:
            td.barrier()
        torch.cuda.synchronize(dev)

    from peekvit_amd import engine
    if args.train:
        from peekvit_amd import dist as pvdist
        y = torch.randint(0, cfg["num_classes"], (args.batch,), generator=gen, device=dev)
        infer = model
        n_buckets = 0
        reducer = pvdist.OverlappedGradReducer(infer.parameters(), bucket_bytes=args.bucket_kib << 10) if dist else None      # ~25 MB buckets leave during backward

        # the reference's optimizer (configs/optimizer/adam.yaml: torch.optim.Adam, lr 1e-3) and clip (train/train.py:120); stock
        # multi-tensor PyTorch kernels over the 86.6 M fp32 parameters - plumbing around the path, ~1 % of the step
        params = [p for p in infer.parameters() if p.requires_grad]
        opt = None if args.no_optimizer else torch.optim.Adam(params, lr=1e-3, fused=True)

        def train_step(inp):
            nonlocal n_buckets
            for p in params:
                p.grad = None
            logits = infer(inp)
            torch.nn.functional.cross_entropy(logits, y).backward()
            if dist:
                n_buckets = reducer.finish()
            if opt is not None:
                torch.nn.utils.clip_grad_norm_(params, 1.0, foreach=True)
                opt.step()
            return logits.detach()

        model = train_step
    if args.streams:
        engine._STREAMS = args.streams
    fallbacks0 = engine.fallback_count
    with (torch.enable_grad() if args.train else torch.no_grad()), engine.precision(args.precision):
        for _ in range(args.warmup):
            out = model(x)
        # (1) the contract's timed region: exactly K steps between barrier + synchronize, nothing else on the stream
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = model(x)
        barrier()
        elapsed = time.perf_counter() - t0
        # (2) the same K steps again with two HIP events around EVERY launch (on the launch stream) for the per-kernel
        # roofline numbers; the ~200 extra stream commands per step cost ~3 %, which is why (1) is timed without them
        barrier()
        with ops.KernelTimer() as kt:
            t1 = time.perf_counter()
            for _ in range(args.steps):
                out = model(x)
            barrier()
            elapsed_instr = time.perf_counter() - t1
    assert torch.isfinite(out).all()
    if args.train:
        grads = [p.grad for p in infer_model.parameters()]
        assert all(g is not None and bool(torch.isfinite(g).all()) for g in grads), "non-finite parameter gradient"
    if dist:
        t = torch.tensor([elapsed, elapsed_instr], device=dev if args.dist_backend == "nccl" else "cpu", dtype=torch.float64)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        elapsed, elapsed_instr = float(t[0].item()), float(t[1].item())

    # the 16-bit operand type the timed steps really computed in: training -> bf16; inference in mode auto -> fp16 unless a guard
    # sent forwards to the fallback mode (split bf16 operands)
    fallbacks = engine.fallback_count - fallbacks0
    if args.precision == "auto":
        dtype = "bf16" if args.train else ("bf16x3" if (fallbacks or engine.guard_state(infer_model).unsafe) else "f16")
    else:
        dtype = args.precision
    if rank == 0:
        value = world * args.batch * args.steps / elapsed
        ks = kt.summary()
        dom = max(ks, key=lambda k: ks[k]["ms"])
        d = ks[dom]
        if d["flops"] > 0:
            roof = {"kernel": dom, "bound": "mfma", "achieved": round(d["flops"] / (d["ms"] * 1e-3) / 1e12, 1),
                    "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "traffic": None,
                    "peak_sustained_mfma_only_measured": MFMA_BF16_SUSTAINED_MEASURED_TFLOPS}
        else:
            roof = {"kernel": dom, "bound": "hbm", "achieved": round(d["bytes"] / (d["ms"] * 1e-3) / 1e9, 1),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None}
        roof["frac"] = round(roof["achieved"] / roof["peak"], 4)
        roof["launches_per_step"] = d["launches"] // args.steps
        roof["avg_launch_ms"] = round(d["ms"] / d["launches"], 4)
        roof["algorithmic_per_launch"] = round((d["flops"] if d["flops"] > 0 else d["bytes"]) / d["launches"] / (1e12 if d["flops"] > 0 else 1e9), 4)
        # HBM/fabric bytes per launch of this kernel from the committed PMC passes of this same command (separate FETCH_SIZE /
        # WRITE_SIZE runs, FETCH_SIZE doubled per MI355X_MICROARCH.md): profiles/r01_kernel_summary.json
        try:
            prof = next(f for f in ("r03_kernel_summary.json", "r02_kernel_summary.json", "r01_kernel_summary.json") if os.path.exists(os.path.join(ROOT, "profiles", f)))
            roof["traffic_source"] = "profiles/" + prof
            summ = json.load(open(os.path.join(ROOT, "profiles", prof)))["kernels"]
            fam = [v for k, v in summ.items() if k.startswith(dom.replace("_bf16", "").replace("pv_", "pv_")) and "hbm_read_MB" in v]
            if dom == "pv_gemm_bf16":
                fam = [v for k, v in summ.items() if k.startswith("pv_gemm") and "hbm_read_MB" in v]
            if fam and args.model == "vit_b_16" and args.batch == 2048 and args.rank_budget is None and not args.train:
                n = sum(v["launches"] for v in fam)
                roof["traffic"] = round(sum((v["hbm_read_MB"] + v["hbm_write_MB"]) * 1e6 * v["launches"] for v in fam) / n)
                roof["traffic_unit"] = "bytes/launch (PMC FETCH_SIZE x2 + WRITE_SIZE, launch-weighted mean over the GEMM variants)"
        except (OSError, KeyError, ValueError, StopIteration):
            pass
        flops_exec = sum(v["flops"] for v in ks.values()) / (args.steps * args.batch)
        # the EMPIRICAL roofline (scripts/power_roofline.hip, profiles/r02_power_roofline.json): what a dependency-free loop with a 256^2 GEMM tile's
        # LDS / L2 operand traffic sustains on this part at the dominant kernel's HBM bytes per FLOP - `peak` above stays the nominal 2.5 PFLOP/s
        try:
            if roof["bound"] == "mfma" and d["bytes"] > 0:
                pr_ = json.load(open(os.path.join(ROOT, "profiles", "r02_power_roofline.json")))
                pts = sorted((r["bytes_per_kflop_hbm"], r["tflops"]) for r in pr_["synthetic"] if r["lds_per_16mfma"] == 6 and r["l2_per_16mfma"] == 2)
                bpk = d["bytes"] / d["flops"] * 1e3
                lo = max((q for q in pts if q[0] <= bpk), default=pts[0]); hi = min((q for q in pts if q[0] >= bpk), default=pts[-1])
                emp = lo[1] if hi[0] == lo[0] else lo[1] + (hi[1] - lo[1]) * (bpk - lo[0]) / (hi[0] - lo[0])
                roof["empirical"] = {"algorithmic_hbm_bytes_per_kflop": round(bpk, 3), "sustained_by_dependency_free_mix_tflops": round(emp, 1),
                                     "frac": round(roof["achieved"] / emp, 4), "source": "profiles/r02_power_roofline.json"}
        except (OSError, KeyError, ValueError, IndexError, ZeroDivisionError):
            pass
        kernels = {k: {"ms_per_step": round(v["ms"] / args.steps, 3),
                       "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) if v["flops"] else None,
                       "algo_gbs": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) if v["bytes"] else None}
                   for k, v in sorted(ks.items(), key=lambda kv: -kv[1]["ms"])}
        line = {
            "metric": "images/sec ViT-B/16 fwd @ batch 2048, 224x224" if args.model == "vit_b_16" and args.batch == 2048 and args.rank_budget is None and not args.train
                      else f"images/sec {workload}",
            "value": round(value, 1), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "ms_per_step_instrumented": round(elapsed_instr / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": workload, "global_batch": world * args.batch,
                       "parallelism": (f"dp{world} (batch-sharded, gradient all-reduce over {args.dist_backend})" if args.train and dist
                                       else f"replicas x{world} (batch-sharded, no collective)"),
                       "gflop_per_image": round(flops_img / 1e9, 3), "gflop_per_image_executed": round(flops_exec / 1e9, 3),
                       "last_block": ("class-token rows only (k|v for all tokens; q, out-proj, MLP for the rows the head reads; same logits)"
                                      if "pv_attention_rows_bf16" in ks else "all rows"),
                       "precision_mode": args.precision, "streams": engine._STREAMS,
                       "operands": {"f16": "IEEE fp16 operands, fp32 accumulate (same MFMA rate as bf16), range-guarded" if args.precision == "auto"
                                    else "IEEE fp16 operands, fp32 accumulate", "bf16": "bf16 operands, fp32 accumulate",
                                    "bf16x3": "split bf16 hi+lo operands (3 products), fp32 accumulate"}[dtype],
                       "range_guard_fallbacks": fallbacks},
            # MFMA FLOPs the kernels really executed per second (their own algorithmic counts) over the dense bf16 peak - NOT the
            # reference formula's FLOPs, of which the last block skips the rows nobody reads
            "model_mfma_roofline_frac": round(value / world * flops_exec / (MFMA_BF16_PEAK_TFLOPS * 1e12), 4),
            "roofline": roof,
            "kernels": kernels,
        }
        if args.train and dist:
            line["grad_allreduce"] = {"buckets_per_step": n_buckets, "launched_during_backward_total": reducer.launched_before_finish,
                                      "bucket_bytes": reducer.bucket_bytes}
        if not args.no_cpu_baseline and world == 1:
            plain = args.rank_budget is None and not args.train
            line["cpu_baseline"] = cpu_baseline(cfg, args.cpu_batch, args.cpu_iters, infer_model if plain else None, dev)
            if plain and args.precision == "auto":
                # the same forward on bf16 operands (the explicit "bf16" mode), for the record
                with torch.no_grad(), engine.precision("bf16"):
                    for _ in range(2):
                        infer_model(x)
                    torch.cuda.synchronize(dev)
                    t2 = time.perf_counter()
                    for _ in range(args.steps):
                        infer_model(x)
                    torch.cuda.synchronize(dev)
                    dtb = time.perf_counter() - t2
                line["bf16_mode"] = {"value": round(args.batch * args.steps / dtb, 1), "unit": "images/sec",
                                     "ms_per_step": round(dtb / args.steps * 1e3, 3), "dtype": "bf16"}
            if plain and engine._LAST_BLOCK_ROWS:
                # the same forward with the last block computing all S rows, as the reference does (PEEKVIT_AMD_LAST_BLOCK_ROWS=0)
                engine._LAST_BLOCK_ROWS = False
                try:
                    with torch.no_grad(), engine.precision(args.precision):
                        for _ in range(2):
                            infer_model(x)
                        torch.cuda.synchronize(dev)
                        t2 = time.perf_counter()
                        for _ in range(args.steps):
                            infer_model(x)
                        torch.cuda.synchronize(dev)
                        dta = time.perf_counter() - t2
                finally:
                    engine._LAST_BLOCK_ROWS = True
                line["all_rows_mode"] = {"value": round(args.batch * args.steps / dta, 1), "unit": "images/sec",
                                         "ms_per_step": round(dta / args.steps * 1e3, 3), "gflop_per_image": round(flops_img / 1e9, 3)}
            err = line["cpu_baseline"].get("gpu_logits_rel_l2_vs_oracle", {}).get(args.precision)
            if plain and err is not None:
                line["logits_rel_l2_vs_oracle"] = err
                line["logits_tolerance"] = 1e-3
                if dtype == "f16" and not err <= 1e-3:
                    # the reported operand type is the one that is supposed to meet BASELINE.json's contract: refuse to report otherwise
                    print(json.dumps(line), file=sys.stderr, flush=True)
                    raise SystemExit(f"bench.py: dtype {dtype} logits are {err:.2e} from the CPU oracle, outside the 1e-3 contract: no result line")
        print(json.dumps(line), flush=True)
    if dist:
        td.barrier()
        td.destroy_process_group()


if __name__ == "__main__":
    main()
