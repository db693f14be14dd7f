"""Tile-raster / cache-policy A/B of the 256^2 GEMM on the wide-N token GEMMs (QKV N = 2304, fc1 N = 3072; K = 768, M = B*S).

Question: the L2-side fetch of these launches is 4-5x the algorithmic read.  Which tile order and cache policy
bring it down, and what does that do to time, clock and power?

    python scripts/raster_ab.py --build                 # (build container) compile the cache-policy variants of the library
    python scripts/raster_ab.py --rounds 6 --out gpurun_out/raster_ab.json            # timing: interleaved rounds, one process
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/rab_fetch -o rab -- python3 scripts/raster_ab.py --pmc --out gpurun_out/raster_manifest.json
    python scripts/raster_ab.py --join gpurun_out/raster_manifest.json gpurun_out/rab_fetch gpurun_out/rab_hit ...   # counters per config

Configurations = (library build, gm, gc):
  library   base: default cache policy on both staging streams; ant: activation (A) loads `nt`; wnt: weight (W) loads `nt`
  gm, gc    the raster of pv_gemm256_kernel (pv_gemm.hip): the tile list is cut into chunks of gc column tiles (gc = 0: all), inside
            a chunk into groups of gm row panels, n slow inside a group; every XCD walks a contiguous eighth of the list.
            (gm, all) with gm = 4 is the r1 default; gm = 1 co-schedules all N tiles of one M panel; gc < tiles_n keeps gc weight
            tiles per XCD stationary while the activation panels stream.
In --pmc mode every configuration is launched exactly PMC_LAUNCHES times per shape, in manifest order, so the k-th pv_gemm256 dispatch
of the counter CSV belongs to manifest[k // PMC_LAUNCHES].
"""
import argparse
import csv
import ctypes as C
import glob
import json
import os
import statistics
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

VARIANTS = {"base": [], "ant": ["-DPV_GEMM_A_AUX=2"], "wnt": ["-DPV_GEMM_W_AUX=2"]}
PMC_LAUNCHES = 3


def lib_path(tag):
    from peekvit_amd import _build
    return _build.LIB if tag == "base" else os.path.join(_build.HERE, f"libpeekvit_hip_{tag}.so")


def configs(tiles_n):
    rasters = [(4, 0), (1, 0), (2, 0), (8, 0), (6, 6), (8, 4), (11, 3)] + ([(7, 5)] if tiles_n == 9 else [])
    out = [("base", gm, gc) for gm, gc in rasters]
    out += [("ant", gm, gc) for gm, gc in ((4, 0), (6, 6), (8, 4), (11, 3))]
    out += [("wnt", gm, gc) for gm, gc in ((4, 0), (1, 0))]
    return out


class PowerSampler(threading.Thread):
    """Board power (hwmon power1_average / power1_input, uW) and sclk (freq1_input, Hz) from sysfs at ~20 Hz; read-only."""

    def __init__(self, pci=None):
        super().__init__(daemon=True)
        # the host has 8 cards (other jobs run on the others) and this job sees one of them: `pci` = its PCI address
        # (dddd:bb:dd.f); without it every card is sampled and the one drawing the most power in the window is reported
        self.cards = []
        for b in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
            addr = os.path.basename(os.path.realpath(os.path.dirname(os.path.dirname(b))))
            if pci is not None and addr.lower() != pci.lower():
                continue
            files = {}
            for key, names in (("power", ("power1_average", "power1_input")), ("sclk", ("freq1_input",))):
                for n in names:
                    if os.path.exists(os.path.join(b, n)):
                        files[key] = os.path.join(b, n)
                        break
            if files:
                self.cards.append(files)
        self.samples, self.stop = [], False

    def run(self):
        while not self.stop:
            row = {"t": time.perf_counter(), "cards": []}
            for files in self.cards:
                c = {}
                for k, f in files.items():
                    try:
                        c[k] = float(open(f).read().strip())
                    except (OSError, ValueError):
                        pass
                row["cards"].append(c)
            self.samples.append(row)
            time.sleep(0.05)

    def mean(self, t0, t1):
        sel = [s for s in self.samples if t0 <= s["t"] <= t1]
        best = {}
        for ci in range(len(self.cards)):
            pw = [s["cards"][ci]["power"] for s in sel if "power" in s["cards"][ci]]
            ck = [s["cards"][ci]["sclk"] for s in sel if "sclk" in s["cards"][ci]]
            if pw and (not best or sum(pw) / len(pw) * 1e-6 > best["power_w"]):
                best = {"power_w": round(sum(pw) / len(pw) * 1e-6, 1), "card": ci}
                if ck:
                    best["sclk_mhz"] = round(sum(ck) / len(ck) * 1e-6, 1)
        best["samples"] = len(sel)
        return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--pmc", action="store_true", help="counter mode: PMC_LAUNCHES launches per configuration, manifest order")
    ap.add_argument("--join", nargs="+", help="manifest.json followed by rocprofv3 output directories")
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--iters", type=int, default=8)
    ap.add_argument("--steady", type=float, default=1.0, help="seconds of back-to-back launches per configuration for power / clock")
    ap.add_argument("--M", type=int, default=403456)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "raster_ab.json"))
    args = ap.parse_args()

    if args.build:
        from peekvit_amd import _build
        _build.build()
        for tag, defs in VARIANTS.items():
            if tag != "base":
                print(_build.build_variant(tag, defs))
        return
    if args.join:
        return join(args.join[0], args.join[1:])

    import torch
    from peekvit_amd import _lib
    from peekvit_amd._lib import GemmArgs, PV_EPI_BIAS_BF16, PV_EPI_BIAS_GELU_BF16
    dev = torch.device("cuda:0")
    libs = {}
    for tag in VARIANTS:
        path = lib_path(tag)
        if not os.path.exists(path):
            print(f"variant {tag}: {path} missing (run --build in the build container), skipped", file=sys.stderr)
            continue
        lib = C.CDLL(path)
        lib.pv_gemm_bf16.restype, lib.pv_gemm_bf16.argtypes = C.c_int, [C.POINTER(GemmArgs), C.c_void_p]
        lib.pv_debug_set_gemm_raster.restype, lib.pv_debug_set_gemm_raster.argtypes = None, [C.c_int, C.c_int]
        libs[tag] = lib
    g = torch.Generator(device=dev).manual_seed(0)
    M = args.M
    shapes = [("qkv", 2304, 768, PV_EPI_BIAS_BF16), ("fc1", 3072, 768, PV_EPI_BIAS_GELU_BF16)]
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    results, manifest = {}, []
    pr = torch.cuda.get_device_properties(0)
    pci = None
    if all(hasattr(pr, k) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id")):
        pci = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
    sampler = PowerSampler(pci)
    if not sampler.cards:
        sampler = PowerSampler(None)
    print("power sampler: pci", pci, "cards", len(sampler.cards), flush=True)
    if not args.pmc:
        sampler.start()
    for name, N, K, epi in shapes:
        a = torch.randn(M, K, generator=g, device=dev).to(torch.bfloat16)
        w = (torch.randn(N, K, generator=g, device=dev) * K ** -0.5).to(torch.bfloat16)
        bias = torch.randn(N, generator=g, device=dev)
        out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        ga = GemmArgs(A=a.data_ptr(), W=w.data_ptr(), bias=bias.data_ptr(), out=out.data_ptr(), M=M, N=N, K=K, lda=K, ldw=K, ldo=N,
                      qscale=1.0, epilogue=epi)
        cfgs = [c for c in configs((N + 255) // 256) if c[0] in libs]
        flops = 2.0 * M * N * K

        def launch(cfg, n):
            lib = libs[cfg[0]]
            lib.pv_debug_set_gemm_raster(cfg[1], cfg[2])
            for _ in range(n):
                rc = lib.pv_gemm_bf16(C.byref(ga), stream)
                assert rc == 0, rc
            lib.pv_debug_set_gemm_raster(0, 0)

        ref = None
        for cfg in cfgs:                                  # correctness: every raster / policy computes the same bits
            launch(cfg, 1)
            torch.cuda.synchronize()
            if ref is None:
                ref = out.clone()
            else:
                assert torch.equal(out, ref), cfg
        if args.pmc:
            for cfg in cfgs:
                launch(cfg, PMC_LAUNCHES)
                torch.cuda.synchronize()
                manifest.append({"shape": name, "lib": cfg[0], "gm": cfg[1], "gc": cfg[2]})
            continue
        times = {cfg: [] for cfg in cfgs}
        for _ in range(args.rounds):                       # interleaved rounds in one process (guide rule 24)
            for cfg in cfgs:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                launch(cfg, args.iters)
                e1.record()
                torch.cuda.synchronize()
                times[cfg].append(e0.elapsed_time(e1) / args.iters)
        for cfg in cfgs:                                   # steady state: clock and board power while this configuration runs alone
            n = max(8, int(args.steady * 1e3 / statistics.median(times[cfg])))
            launch(cfg, 4)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            launch(cfg, n)
            e1.record()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            ms = e0.elapsed_time(e1) / n
            key = f"{name}/{cfg[0]}/gm{cfg[1]}/gc{cfg[2]}"
            results[key] = {"median_ms": round(statistics.median(times[cfg]), 4), "min_ms": round(min(times[cfg]), 4),
                            "tflops_median": round(flops / statistics.median(times[cfg]) / 1e9, 1),
                            "steady_ms": round(ms, 4), "steady_tflops": round(flops / ms / 1e9, 1), **sampler.mean(t0 + 0.3 * (t1 - t0), t1)}
            print(key, results[key], flush=True)
        del a, w, out
    sampler.stop = True
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump({"manifest": manifest, "pmc_launches": PMC_LAUNCHES} if args.pmc else {"M": M, "results": results}, open(args.out, "w"), indent=1)


def join(manifest_file, dirs):
    man = json.load(open(manifest_file))
    per = man["pmc_launches"]
    table = [dict(m) for m in man["manifest"]]
    for d in dirs:
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            print("no counter csv under", d, file=sys.stderr)
            continue
        rows = [r for r in csv.DictReader(open(files[0])) if "pv_gemm256_kernel" in r["Kernel_Name"]]
        by_disp = {}
        for r in rows:
            by_disp.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
        # per shape the equality-check launches come first (one per configuration), then PMC_LAUNCHES counted launches per
        # configuration in manifest order: walk the dispatches in order
        shapes = []
        for m in table:
            if m["shape"] not in shapes:
                shapes.append(m["shape"])
        allv = [by_disp[k] for k in sorted(by_disp)]
        pos, idx = 0, 0
        for sh in shapes:
            n_sh = sum(1 for m in table if m["shape"] == sh)
            pos += n_sh                                    # the equality-check launches of this shape
            for _ in range(n_sh):
                vals = allv[pos:pos + per]
                pos += per
                for cname in vals[0]:
                    table[idx][cname] = round(sum(v[cname] for v in vals) / len(vals), 1)
                idx += 1
    for m in table:
        if "FETCH_SIZE" in m:
            m["l2_fabric_read_MB"] = round(2 * m["FETCH_SIZE"] * 1024 / 1e6, 1)          # FETCH_SIZE (KiB) reads 1/2 on gfx950 wide loads
        if "WRITE_SIZE" in m:
            m["l2_fabric_write_MB"] = round(m["WRITE_SIZE"] * 1024 / 1e6, 1)
        if "TCC_HIT_sum" in m and "TCC_MISS_sum" in m:
            m["l2_hit_rate"] = round(m["TCC_HIT_sum"] / max(m["TCC_HIT_sum"] + m["TCC_MISS_sum"], 1.0), 4)
    json.dump(table, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
