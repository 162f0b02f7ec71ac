#!/usr/bin/env python3
"""Headline benchmark: training frames/sec of the UNet-ConvLSTM path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], reference hyper-parameters main.py:215-228): TemporalUNetDualView(
base_ch=64, use_skip_lstm=True), per-GPU batch 32 sequences of 20 frames 64x64x2, synthetic data resident
in HBM, random-init weights.  A step = zero_grad -> forward -> loss -> backward -> gradient all-reduce ->
clip(1.0) -> AdamW (main.py:91-108), bf16 MFMA compute with f32 accumulation and f32 master weights.
Weak scaling: per-GPU batch fixed, sequences sharded over ranks, one RCCL gradient exchange per step.

Prints ONE JSON line (rank 0) with the contract fields plus
  roofline      -- dominant MFMA kernel: algorithmic FLOPs / HIP-event time of its launches (one instrumented step)
  cpu_baseline  -- the CPU oracle's training step timed on this host's cores on a bounded sample (N=1 only)
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

torch = None          # imported by main() AFTER the launcher branch: the parent that starts the ranks never loads torch at all

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0          # dense bf16 MFMA peak, MI355X_MICROARCH.md chip table
PEAK_HBM_GBS = 8000.0              # HBM3E, same guide
TRAFFIC_FILE = "round3_hbm_traffic.json"
TRAFFIC_NOTE = (f"committed PMC passes (profiles/{TRAFFIC_FILE}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs of "
                "this command, gfx950 correction applied); NOT measured in this run")
TRAIN_GFLOP_PER_FRAME = {(64, True, 64): 39.79, (32, False, 64): 6.33, (64, True, 128): 159.17, (32, False, 128): 25.33,
                         (64, True, 256): 636.67}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=32, help="sequences per GPU (main.py:215); weak scaling keeps it fixed as N grows")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): per-GPU batch = --batch on every rank; strong: --global-batch sequences in total, "
                         "per-GPU batch = global / N (SURVEY.md 8e: global 32 -> 4 per GPU at N = 8)")
    ap.add_argument("--global-batch", type=int, default=32, help="total sequences per step with --scaling strong")
    ap.add_argument("--graph", action="store_true",
                    help="replay the training step as ONE captured HIP graph (engine.GraphedTrainStep; single rank only): host enqueue "
                         "time per step drops from 10-19 ms to < 0.2 ms; the step itself is device-bound at every batch size "
                         "(DESIGN.md section 6), so the frame rate does not move")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the second north_star workload (cloud 128x128 seq-8, same per-GPU batch) that follows the headline")
    ap.add_argument("--seq", type=int, default=20)
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--base-ch", type=int, default=64)
    ap.add_argument("--no-skip-lstm", action="store_true")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16"],
                    help="16-bit storage / MFMA operand type: bf16 (default) or f16 = the fp16 twin kernels with dynamic loss scaling "
                         "(BASELINE.json configs[3])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--force-ddp", action="store_true", help="wrap in FlatDDP even for one rank (exercises the RCCL path)")
    ap.add_argument("--mode", default="train", choices=["train", "rollout"],
                    help="rollout = BASELINE.json config 5: stateful frame-by-frame inference, hipGraph-captured step")
    ap.add_argument("--no-graph", action="store_true", help="rollout without graph capture")
    ap.add_argument("--sync-wgrad", action="store_true",
                    help="keep weight-gradient GEMMs on the main stream for the whole run (the profile run: kernel durations "
                         "in a rocprofv3 trace are then free of side-stream overlap and agree with the roofline leg)")
    ap.add_argument("--dump-launches", action="store_true", help="per-shape table of the instrumented step (stderr)")
    ap.add_argument("--bf16-buckets", action="store_true", help="FlatDDP exchanges gradients as bf16 (half the bytes over xGMI)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="developer rehearsal of the N-rank flow on a 1-GPU box: every rank uses cuda:0 and the collectives go over "
                         "gloo (RCCL refuses two ranks on one device); the number it prints is NOT a scaling measurement")
    ap.add_argument("--dry-launch", action="store_true",
                    help="start the ranks, rendezvous over gloo on the CPU, report what every rank saw and exit (no GPU work): "
                         "the CPU test of the launch contract")
    return ap.parse_args()


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def visible_gpu_count():
    """GPUs this process could use, WITHOUT touching the HIP runtime: KFD topology nodes with SIMDs (sysfs), cut down by
    HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES.  None when sysfs says nothing (then the ranks
    themselves report a missing device)."""
    n = None
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for node in os.listdir(base):
            with open(os.path.join(base, node, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            n += int(props.get("simd_count", "0")) > 0
    except OSError:
        n = None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            k = len([x for x in v.split(",") if x.strip() != ""])
            n = k if n is None else min(n, k)
    return n


def self_launch(a) -> int:
    """``python bench.py --gpus N`` without torchrun: start the N ranks HERE, as fresh child processes, before this process
    has made any HIP call (a process that has initialised the GPU must never exec or fork into another GPU program), and
    return the launcher's exit code.  The children are this same script under ``torch.distributed.run``; rank 0 prints the
    JSON line on the inherited stdout."""
    if not a.dry_launch:
        have = visible_gpu_count()                # sysfs + environment only: no torch, no HIP call in this process
        if have is not None and have < a.gpus and not (a.rehearse_on_one_gpu and have >= 1):
            print(f"bench.py: --gpus {a.gpus} but only {have} GPU(s) visible", file=sys.stderr)
            return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__), *sys.argv[1:]]
    log("launching " + " ".join(cmd))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.run(cmd, env=env).returncode


def dry_launch(a, world: int, rank: int) -> None:
    """Every rank joins a gloo group on the CPU and reports (rank, WORLD_SIZE, LOCAL_RANK); rank 0 prints them."""
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo")
        seen = [None] * world
        dist.all_gather_object(seen, {"rank": rank, "world_size": world, "local_rank": int(os.environ.get("LOCAL_RANK", "0")),
                                      "per_gpu_batch": a.batch})
        dist.destroy_process_group()
    else:
        seen = [{"rank": 0, "world_size": 1, "local_rank": 0, "per_gpu_batch": a.batch}]
    if rank == 0:
        emit({"dry_launch": True, "n_gpus": a.gpus, "scaling": a.scaling, "global_batch": a.batch * world, "ranks": seen})


def cpu_baseline(base_ch: int, skip: bool, size: int, seq: int):
    """Oracle (CPU restatement, kind 'port') training step on the host cores, bounded sample (the same
    synthetic batch shrunk to B=4 at 64x64 / B=2 above, 1 warm-up + 3 timed steps, best and median)."""
    from oracle import unet_oracle as O
    import unet_convlstm_amd as U
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = min(cores, 16)          # a 1-GPU box owns a 16-core share of the host; more threads only oversubscribe it
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    m = U.TemporalUNetDualView(1, 1, base_ch=base_ch, use_skip_lstm=skip)          # parameter container only (CPU)
    p = {k: v.detach().clone() for k, v in m.state_dict().items()}
    B, T = (4 if size <= 64 else 2), seq
    if size > 128:
        B, T = 1, min(seq, 4)
    log(f"cpu_baseline: oracle training step on {cores} host threads, B={B} T={T}, 1 warm-up + 3 timed ...")
    g = torch.Generator().manual_seed(1)
    x = torch.rand((B, T, 2, size, size), generator=g)
    y = torch.rand((B, T, 1, size, size), generator=g) * 2 - 1
    O.train_step(dict(p), x[:, :2], y[:, :2], None, False)      # warm-up step (thread pool, allocator) on a 2-frame prefix
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        O.train_step(dict(p), x, y, None, False)
        times.append(time.perf_counter() - t0)
        log(f"cpu_baseline: step {len(times)}/3 {times[-1]:.1f} s")
    best, med = min(times), sorted(times)[1]
    return {"value": round(B * T / best, 3), "median": round(B * T / med, 3), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"oracle training step (fp32 PyTorch eager CPU), B={B} T={T} {size}x{size}, base_ch={base_ch}, "
                      f"skip_lstm={skip}; 1 warm-up + 3 timed steps, best {best:.1f} s, median {med:.1f} s"}


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def rollout_bench(a, U, dev, skip):
    """Inference-only autoregressive rollout (config 5): one frame per step, all recurrent state carried on the device."""
    model = U.TemporalUNetDualView(1, 1, base_ch=a.base_ch, lstm_layers=1, use_skip_lstm=skip).to(dev).eval()
    sp = U.StreamingPredictor(model, use_graph=not a.no_graph, warmup=2)
    frames = [torch.rand(a.batch, 2, a.size, a.size, device=dev) for _ in range(4)]
    for i in range(max(a.warmup, 3)):                 # eager warm-up + capture + first replay
        sp.step(frames[i % 4])
    torch.cuda.synchronize()
    U.quiesce_host_gc()                               # a 75-ms garbage collection would be 60 frames of this host-paced loop
    n = a.steps * a.seq
    t0 = time.perf_counter()
    for i in range(n):
        y = sp.step(frames[i % 4])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    fwd_gf = {64: 13.264, 128: 53.06, 256: 212.22, 512: 848.89}.get(a.size) if (a.base_ch, skip) == (64, True) else None
    out = {"metric": f"inference frames/sec, {a.size}x{a.size} stateful rollout", "value": round(a.batch * n / dt, 2), "unit": "frames/s",
           "n_gpus": 1, "steps": n, "warmup": max(a.warmup, 3), "ms_per_step": round(dt / n * 1e3, 3), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
           "config": {"workload": f"TemporalUNetDualView(base_ch={a.base_ch}, use_skip_lstm={skip}) eval, frame-by-frame, "
                                  f"batch {a.batch}, hipGraph={'off' if a.no_graph else 'on'}"},
           "finite": bool(torch.isfinite(y).all())}
    if fwd_gf:
        out["model_tflops"] = round(out["value"] * fwd_gf / 1e3, 2)
    if a.dump_launches:
        # one eager frame with HIP events around every GEMM launch (what the graph replays)
        from unet_convlstm_amd import ops
        eager = U.StreamingPredictor(model, use_graph=False)
        for i in range(3):
            eager.step(frames[i % 4])
        ops.PROFILE = []
        eager.step(frames[3])
        torch.cuda.synchronize()
        rows, ops.PROFILE = ops.PROFILE, None
        tot = 0.0
        for kind, flops, e0, e1, note, _nb in rows:
            t_ms = e0.elapsed_time(e1)
            tot += t_ms
            log(f"{kind:32s} {t_ms * 1e3:8.1f} us {flops / t_ms / 1e9:7.1f} TF/s  {note}")
        log(f"GEMM launches of one frame: {tot * 1e3:.1f} us")
    emit(out)


def run_workload(a, U, ops, dist, dev, world, rank, model, opt, ddp, *, size, seq, batch, steps, warmup, roofline, tag):
    """Warm up, time ``steps`` training steps of one workload (barrier + synchronize on both sides, MAX over ranks) and, when
    asked, price one extra instrumented step against the MFMA / HBM rooflines.  Returns a dict of measurements."""
    skip = not a.no_skip_lstm
    data = U.SyntheticSequences(batch, seq, size, size, seed=1 + rank, kind="uniform", device=dev)
    x, y = data.x, data.y

    graphed = None
    if getattr(a, "graph", False) and ddp is None and opt.capturable:
        graphed = U.GraphedTrainStep(model, opt, x, y, None, False, warmup=2)

    def step():
        if graphed is not None:
            return graphed(x, y)
        return U.train_step(model, opt, x, y, None, False, ddp)      # USE_MASK = False, main.py:219

    for i in range(warmup):
        step()
        torch.cuda.synchronize()
        log(f"{tag}: warm-up step {i} done")
    torch.cuda.synchronize()
    # The timed region starts from a synchronised host: a generation-2 garbage collection (~75 ms of host time in a process
    # that has imported torch) in its first steps is not hidden by run-ahead and showed up as ONE 60-130 ms step in ~40 % of
    # 20-30-step runs (tools/spike_hunt.py).  Collect now and freeze what is alive, as a training loop does after its first
    # steps (engine.train_one_epoch); nothing of the step's GPU work is skipped.
    U.quiesce_host_gc()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]      # one event record per step: no synchronisation
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(steps):
        loss, _ = step()
        marks[i + 1].record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))
    log(f"{tag}: per-step device time: min {per_step[0]:.2f}  median {per_step[len(per_step) // 2]:.2f}  max {per_step[-1]:.2f} ms")
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    frames = batch * seq * world * steps
    log(f"{tag}: timed {steps} steps in {dt:.3f} s")
    res = {"value": frames / dt, "dt": dt, "ms_per_step": dt / steps * 1e3, "loss": float(loss),
           "per_step": [round(per_step[0], 2), round(per_step[len(per_step) // 2], 2), round(per_step[-1], 2)], "roofline": None}
    gf = TRAIN_GFLOP_PER_FRAME.get((a.base_ch, skip, size))
    if gf is not None:
        res["model_tflops"] = round(res["value"] * gf / 1e3, 2)
        res["model_mfma_frac"] = round(res["value"] * gf / 1e3 / (PEAK_BF16_TFLOPS * world), 4)

    if roofline:
        # One instrumented step with every kernel on the launch stream: with the weight-gradient GEMMs overlapping on
        # their side stream the events around a forward GEMM would also count the time it shares the CUs with them.
        # EVERY rank runs it (the step contains the gradient all-reduce); only rank 0 records events.
        async_was, ops.ASYNC_WGRAD = ops.ASYNC_WGRAD, False
        ops.PROFILE = [] if rank == 0 else None
        ops.PROFILE_HBM = [] if rank == 0 else None
        U.train_step(model, opt, x, y, None, False, ddp)          # always eager: a graph replay records no per-launch events
        torch.cuda.synchronize()
        ops.ASYNC_WGRAD = async_was
    if roofline and rank == 0:
        agg = {}
        rows = []
        for kind, flops, e0, e1, note, nbytes in ops.PROFILE:
            t_ms = e0.elapsed_time(e1)
            rows.append((t_ms, kind, flops, note))
            k = agg.setdefault(kind, [0.0, 0.0, 0, 0.0])
            k[0] += flops
            k[1] += t_ms
            k[2] += 1
            k[3] += nbytes
        ops.PROFILE = None
        hbm = {}
        for kind, nbytes, e0, e1, _ in (ops.PROFILE_HBM or []):
            k = hbm.setdefault(kind, [0.0, 0.0, 0])
            k[0] += nbytes
            k[1] += e0.elapsed_time(e1)
            k[2] += 1
        ops.PROFILE_HBM = None
        if a.dump_launches:
            merged = {}
            for t_ms, kind, flops, note in rows:
                m = merged.setdefault((kind, note), [0.0, 0.0, 0])
                m[0] += t_ms
                m[1] += flops
                m[2] += 1
            for (kind, note), (t_ms, flops, n) in sorted(merged.items(), key=lambda kv: -kv[1][0]):
                log(f"{kind:16s} x{n:3d} {t_ms:8.3f} ms {flops / t_ms / 1e9:7.1f} TF/s  {note}")
        if agg:
            dom = max(agg, key=lambda k: agg[k][1])
            fl, ms, n, nb = agg[dom]
            ach = fl / (ms * 1e-3) / 1e12
            # HBM bytes per launch of that kernel family from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE and
            # --pmc WRITE_SIZE in separate runs of this same command, gfx950 correction applied: tools/pmc_traffic.py)
            traffic, pmc = None, {}
            tpath = os.path.join(ROOT, "profiles", TRAFFIC_FILE)
            if (a.base_ch, skip, size, seq, batch) == (64, True, 64, 20, 32) and os.path.exists(tpath):
                try:
                    pmc = json.load(open(tpath))["kernels"]
                    traffic = pmc[dom]["hbm_bytes_per_launch"]
                except Exception:
                    traffic = None
            alg = nb / n

            def with_ratio(k, v):
                out = {"tflops": round(v[0] / (v[1] * 1e-3) / 1e12, 2), "ms": round(v[1], 3), "launches": v[2],
                       "algorithmic_bytes": round(v[3] / v[2])}
                t = pmc.get(k, {}).get("hbm_bytes_per_launch") if isinstance(pmc, dict) else None
                if t:
                    out["traffic"] = t
                    out["traffic_ratio"] = round(t / (v[3] / v[2]), 2)
                return out

            res["roofline"] = {
                "bound": "mfma", "kernel": dom, "kernel_note": "epilogue family[tile shape the library picked]; rocprofv3 names: "
                "patch128x256 = igemm_fwd_kernel<epi, 2, nsrc>, pertap128x128 = <epi, 0, nsrc>, pertap64x256 = <epi, 1, nsrc>, "
                "ring64 = igemm_fwd_c64_kernel (epi 0 store, 1 fused ConvLSTM cell, 2 split-K slabs); igemm_wgrad[p3_256x256] = "
                "igemm_wgrad_p3_kernel, [p2_*] = igemm_wgrad_p2_kernel, [generic] = igemm_wgrad_kernel",
                "achieved": round(ach, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                "traffic_unit": "HBM bytes per launch", "traffic_source": TRAFFIC_NOTE if traffic is not None else None,
                # every operand read once + the result written once, averaged over the family's launches: traffic well
                # above it = re-reads that missed L2 / the Infinity Cache
                "algorithmic_bytes": round(alg), "traffic_ratio": round(traffic / alg, 2) if traffic else None,
                "launches": n, "timing": "HIP events, one serialised step (side stream off)",
                "avg_launch_ms": round(ms / n, 4),
                "all": {k: with_ratio(k, v) for k, v in agg.items()},
                # the HBM-bound kernels of the same step against the 8 TB/s HBM roofline: ALGORITHMIC bytes (each tensor
                # read or written once) / HIP-event time
                "hbm_bound": {k: {"gbs": round(v[0] / (v[1] * 1e-3) / 1e9, 1), "frac": round(v[0] / (v[1] * 1e-3) / 1e9 / PEAK_HBM_GBS, 3),
                                  "ms": round(v[1], 3), "launches": v[2], "algorithmic_bytes": round(v[0] / v[2]),
                                  **({"traffic": pmc[k]["hbm_bytes_per_launch"], "traffic_ratio": round(pmc[k]["hbm_bytes_per_launch"] / (v[0] / v[2]), 2)}
                                     if isinstance(pmc, dict) and k in pmc and pmc[k].get("hbm_bytes_per_launch") else {})}
                              for k, v in hbm.items()}}
    if world > 1:
        dist.barrier()
    return res


_JSON_OUT = None          # the process's real stdout, kept for the ONE JSON line (see quiet_stdout)


def quiet_stdout() -> None:
    """Route file descriptor 1 to stderr for the rest of the process and keep the real stdout for the JSON line only: libraries
    print to stdout on their own (RCCL writes a five-line version banner when the first communicator is created), and the
    contract is ONE JSON line on stdout."""
    global _JSON_OUT
    if _JSON_OUT is None:
        sys.stdout.flush()
        _JSON_OUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)


def emit(obj) -> None:
    out = _JSON_OUT or sys.stdout
    print(json.dumps(obj), file=out, flush=True)


def main():
    global torch
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(self_launch(a))             # this process has not even imported torch
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    quiet_stdout()
    if a.scaling == "strong":
        if a.global_batch % world:
            raise SystemExit(f"--scaling strong: --global-batch {a.global_batch} is not a multiple of {world} ranks")
        a.batch = a.global_batch // world
    import torch as _torch
    torch = _torch
    if a.dry_launch:
        return dry_launch(a, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU path)")
    if a.rehearse_on_one_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1 or a.force_ddp:
        if a.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        elif "RANK" not in os.environ:                    # single process without torchrun
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        else:
            dist.init_process_group("nccl", device_id=dev)

    import unet_convlstm_amd as U
    from unet_convlstm_amd import ops

    skip = not a.no_skip_lstm
    torch.manual_seed(1234)
    if a.dtype == "f16":
        U.set_compute_dtype(torch.float16)
    if a.sync_wgrad:
        ops.ASYNC_WGRAD = False
    if a.mode == "rollout":
        return rollout_bench(a, U, dev, skip)
    model = U.TemporalUNetDualView(1, 1, base_ch=a.base_ch, lstm_layers=1, use_skip_lstm=skip, use_attention=False).to(dev).train()
    opt = U.FusedAdamW(model.parameters(), lr=1e-3, weight_decay=1e-4, max_grad_norm=1.0, loss_scale=2.0 ** 14 if a.dtype == "f16" else None,
                       capturable=bool(a.graph and a.dtype != "f16" and world == 1 and not a.force_ddp))
    ddp = U.FlatDDP(model, opt.flat, grad_dtype=torch.bfloat16 if a.bf16_buckets else None) if (world > 1 or a.force_ddp) else None
    exchange = None
    if ddp is not None:
        exchange = ddp.describe()
        # every rank says once what it is part of (stderr): the first N > 1 run should be readable from its log alone
        print(f"[bench rank {rank}/{world}] device {torch.cuda.get_device_name(dev)} cuda:{local}, backend {dist.get_backend()}, "
              f"world_size {dist.get_world_size()}, per-GPU batch {a.batch} ({a.scaling} scaling), {exchange['buckets']} buckets, "
              f"{exchange['allreduce_bytes']} all-reduce bytes per step ({exchange['dtype']})", file=sys.stderr, flush=True)
    log(f"model + data ready ({sum(p.numel() for p in model.parameters())} parameters); warm-up x{a.warmup}")

    head = run_workload(a, U, ops, dist, dev, world, rank, model, opt, ddp, size=a.size, seq=a.seq, batch=a.batch, steps=a.steps,
                        warmup=a.warmup, roofline=not a.no_roofline, tag="headline")
    # does the weight-gradient stream still run beside the main stream?  (the runtime maps streams to hardware queues; two on one
    # queue serialise, which costs this step ~2.5 ms: recorded so that a slow run can be told from a slow box)
    side = ops.side_stream(dev)
    overlap = {"probes": ops._SIDE_STREAM_PROBES.get(str(dev)), "still_concurrent": bool(ops._streams_overlap(torch.cuda.current_stream(dev), side))}

    # The second north_star workload (cloud sequences 128x128 seq-8, preprocessing/build_sequences.py:15,108-131) on the same
    # model, optimiser and ranks, right after the headline: 5 warm-up + 10 timed steps (~1.5 s of GPU time at N = 1).
    second = None
    if not a.no_secondary and (a.size, a.seq) == (64, 20) and a.mode == "train":
        try:
            r2 = run_workload(a, U, ops, dist, dev, world, rank, model, opt, ddp, size=128, seq=8, batch=a.batch, steps=10, warmup=5,
                              roofline=not a.no_roofline, tag="secondary")
            second = {"workload": f"TemporalUNetDualView(base_ch={a.base_ch}, use_skip_lstm={skip}) train step, 128x128 seq-8 (cloud "
                                  f"sequences), per-GPU batch {a.batch}, AdamW+clip", "metric": "training frames/sec, cloud 128x128 seq-8",
                      "value": round(r2["value"], 2), "unit": "frames/s", "steps": 10, "warmup": 5, "ms_per_step": round(r2["ms_per_step"], 3),
                      "model_tflops": r2.get("model_tflops"), "model_mfma_frac": r2.get("model_mfma_frac"),
                      "final_loss": round(r2["loss"], 5), "roofline": r2["roofline"]}
        except Exception as e:          # the headline line must survive a failure here (same exception on every rank: no collective is left half-way)
            second = {"error": f"{type(e).__name__}: {e}"}

    if rank == 0:
        out = {
            "metric": ("training frames/sec, Moving-MNIST 64x64 seq-20" if (a.size, a.seq) == (64, 20)
                       else f"training frames/sec, {a.size}x{a.size} seq-{a.seq}"),
            "value": round(head["value"], 2), "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(head["ms_per_step"], 3), "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None,
            "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": f"TemporalUNetDualView(base_ch={a.base_ch}, use_skip_lstm={skip}) train step, "
                                   f"{a.size}x{a.size} seq-{a.seq}, per-GPU batch {a.batch}, AdamW+clip",
                       "global_batch": a.batch * world, "seq_len": a.seq, "parallelism": f"dp{world}"},
            # 1 = no process group was created (single process, no gradient exchange)
            "rccl_ranks": dist.get_world_size() if dist.is_initialized() else 1,
            **({"allreduce_bytes": exchange["allreduce_bytes"], "buckets": exchange["buckets"], "bucket_dtype": exchange["dtype"]} if exchange else {}),
            **({"rehearsal": "all ranks on cuda:0, collectives over gloo -- not a scaling measurement"} if a.rehearse_on_one_gpu else {}),
            "final_loss": round(head["loss"], 5), "peak_mem_gib": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2), "side_stream": overlap, "host_gc": "collected and frozen after warm-up",
            "step_ms_min_median_max": head["per_step"],
            **({"hip_graph": "whole training step replayed as one captured HIP graph"} if (a.graph and opt.capturable) else {}),
        }
        for k in ("model_tflops", "model_mfma_frac"):
            if k in head:
                out[k] = head[k]
        if head["roofline"] is not None:
            out["roofline"] = head["roofline"]
        if second is not None:
            out["secondary"] = second
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.base_ch, skip, a.size, a.seq)
        emit(out)
    if world > 1 or a.force_ddp:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
