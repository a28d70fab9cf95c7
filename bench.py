#!/usr/bin/env python3
"""Headline benchmark: FM (k=32) training examples/s on the KuaiRec-big-shaped
synthetic log (BASELINE.json config 3), one process per GPU.

    python bench.py --gpus N --steps K --warmup W

A *step* is one mini-batch pass of the hot path over rows already resident in
HBM: gather by row id -> forward -> IPS residual -> batch-sum gradients -> SGD
update of w0, w, V (src/fm.py:72-88 of the reference).  The row-id lists of the
K+W batches (the reference's resample(..., random_state=epoch)) are produced by
the exact host sampler before the timed region and uploaded: they are inputs.
For N > 1 the global batch of N*B rows is sharded over the ranks, gradients
are all-reduced over RCCL and every rank applies the same update (weak scaling:
B rows per GPU).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# multi-process GPU work on this pool needs dmabuf IPC (RCCL's buffer exchange fails with
# hipIpcGetMemHandle: invalid argument otherwise); normally exported already
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def algorithmic_bytes(z: float, k: int, s: int = 8):
    """SURVEY.md 8d, per example: forward reads the CSR entries, label, pscore and
    each touched row of V / entry of w once; the update reads and writes them once."""
    fwd = z * (4 + s) + 8 + s + z * (k + 1) * s
    upd = 2 * z * (k + 1) * s
    return fwd, upd


def cpu_baseline(train, ids, k, lr, seed, budget_s=15.0):
    """The oracle's reference-structured step (same SciPy op sequence as
    src/fm.py:80-88,135-187, per-factor loop included) on the host cores."""
    from oracle import cpu_ref

    X, y, p = train["features"], train["labels"], train["pscores"]
    w0, w, V = cpu_ref.fm_init(seed, X.shape[1], k)
    done, t0 = 0, time.perf_counter()
    for rows in ids:
        cpu_ref.fm_step_refstruct(X[rows], y[rows], p[rows], w0, w, V, lr)
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return done * len(ids[0]) / dt, done, dt


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch-size", type=int, default=65536, help="rows per GPU per step")
    ap.add_argument("--n-train", type=int, default=0, help="rows of the synthetic log (0 = config)")
    ap.add_argument("--workload", default="kuairec_big")
    ap.add_argument("--backend", default="nccl",
                    help="torch.distributed backend for --gpus > 1 (nccl = RCCL; gloo only to rehearse "
                         "the multi-rank path on a box with fewer GPUs than ranks)")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: --batch-size is the GLOBAL batch, split over the GPUs "
                         "(default: weak scaling, --batch-size rows per GPU)")
    ap.add_argument("--no-direct-rccl", action="store_true",
                    help="multi-GPU: exchange through torch.distributed instead of the C ABI's RCCL binding")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from relevance_factorizationmachine_amd import _lib, synth
    from relevance_factorizationmachine_amd.dist import hip_fm_train_dp, hip_fm_worker, init_direct_rccl
    from relevance_factorizationmachine_amd.fm import FactorizationMachines, FmPlan
    from relevance_factorizationmachine_amd.runtime import DeviceCSR, Runtime, sample_batches

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    device = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(device)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(args.backend)

    shape = synth.SHAPES[args.workload]
    n_train = args.n_train or shape.n_train
    k, B, K, W = shape.n_factors, args.batch_size, args.steps, args.warmup
    lr, seed = 9e-6, 12345  # conf/setting/kuairec.yaml: FM/IPS lr, seed
    if args.strong:
        if B % world:
            raise SystemExit(f"--strong: global batch {B} is not a multiple of {world} GPUs")
        B //= world
    gB = B * world
    if gB > n_train:
        raise SystemExit(f"global batch {gB} exceeds the log ({n_train} rows)")

    train, _ = synth.make_log(shape, "FM", "IPS", seed=0, n_train=n_train, n_val=16)
    X = train["features"]
    n = X.shape[1]
    z = X.nnz / X.shape[0]

    rt = Runtime.get(device)
    model = FactorizationMachines(estimator="IPS", n_epochs=K, n_factors=k, lr=lr, batch_size=B,
                                  seed=seed, n_features=n)
    csr = DeviceCSR(rt, X)
    y = rt.upload(train["labels"], dtype=np.float64)
    p = rt.upload(train["pscores"], dtype=np.float64)
    t_s = time.perf_counter()
    ids = sample_batches(n_train, gB, 0, W + K)  # exact resample() ids, same on every rank
    sampler_s = time.perf_counter() - t_s
    d_ids = rt.upload(ids)
    plan = FmPlan(rt, csr, train["labels"], train["pscores"], k, B)
    csr_ptrs = (csr.indptr.data_ptr(), csr.indices.data_ptr(), csr.values.data_ptr(),
                y.data_ptr(), p.data_ptr())
    params = (model.w0.dev.data_ptr(), model.w.dev.data_ptr(), model.V.dev.data_ptr())

    transport = None
    if world == 1:
        def run(first: int, count: int) -> None:
            _lib.check(rt.lib.rfm_fm_train(
                rt.ctx, plan.handle, *csr_ptrs, d_ids.data_ptr() + first * B * 4, B, count, *params,
                lr, None, None, None, None, None, 0, 1e-8, None, None))
    else:
        grad = rt.empty((n * (k + 1) + 1,), torch.float64)
        all_reduce = None
        if args.backend != "nccl":  # rehearsal transport: stage through the host
            def all_reduce(g):
                h = g.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM)
                g.copy_(h)
        worker = hip_fm_worker(rt, plan, csr, y, p, d_ids, gB, model, grad, world, rank, lr, all_reduce)
        # preferred: the whole loop in one C call with RCCL on the compute stream; if RCCL
        # cannot be bound directly, the same steps go through torch.distributed
        direct = False
        if args.backend == "nccl" and not args.no_direct_rccl:
            try:
                direct = init_direct_rccl(rt, world, rank)
            except Exception as exc:  # noqa: BLE001
                print(f"[rank {rank}] direct RCCL unavailable ({exc}); using torch.distributed", file=sys.stderr)
            flag = torch.tensor([1 if direct else 0], device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)  # all ranks or none
            direct = bool(flag.item())
        transport = "rccl-direct" if direct else f"torch.distributed/{args.backend}"

        def run(first: int, count: int) -> None:
            if direct:
                hip_fm_train_dp(rt, plan, d_ids, gB, first, count, model, grad, world, rank, lr)
            else:
                for it in range(first, first + count):
                    worker.step(it, gB)

    def fence() -> None:
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run(0, W)
    fence()
    t0 = time.perf_counter()
    run(W, K)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    value = K * gB / elapsed
    in_sync = None
    if world > 1:
        # outside the timed region: every replica must hold the same parameters
        chk = torch.stack([model.V.dev.sum(), model.w.dev.sum(), model.w0.dev.sum()]).to(
            "cuda" if args.backend == "nccl" else "cpu")
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        in_sync = bool(torch.equal(lo, hi)) and bool(torch.isfinite(chk).all())
        if not in_sync:
            print(f"[rank {rank}] replicas diverged: {lo.tolist()} vs {hi.tolist()}", file=sys.stderr)

    out = {
        "metric": "training examples/sec (FM, k=32)",
        "value": value,
        "unit": "examples/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": 1e3 * elapsed / K,
        "higher_is_better": True,
        "scaling": "strong" if args.strong else "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": (f"{shape.name}: synthetic KuaiRec-big-shaped log {shape.n_users}x{shape.n_items} "
                         f"+ 110 side-feature columns, n_features={n}, {z:.0f} nnz/row, N_train={n_train}, "
                         f"FM k={k}, IPS, lr={lr}, batch_size={B} rows per GPU (global {gB}), "
                         "step = gather+forward+residual+gradients+SGD update; row-id lists precomputed"),
            "batch_size_per_gpu": B,
            "global_batch": gB,
            "parallelism": f"dp{world}" if world > 1 else "single",
            "collective": (f"all-reduce(sum) of the dense [G_V|g_w|g_w0] buffer, {8 * (n * (k + 1) + 1) / 1e6:.1f} MB, "
                           f"{transport}") if world > 1 else None,
            "replicas_in_sync": in_sync,
        },
    }

    if rank == 0 and world == 1:
        # per-kernel durations with HIP events on the launch stream (same ids, continuing the run)
        ms = (C.c_double * 4)()
        cnt = (C.c_int64 * 4)()
        _lib.check(rt.lib.rfm_profile_begin(rt.ctx))
        run(W, K)
        _lib.check(rt.lib.rfm_profile_end(rt.ctx, ms, cnt))
        names = ["fm_forward_kernel", "fm_consume_kernel", "fm_finalize_kernel"]
        avg = [ms[i] / max(cnt[i], 1) for i in range(3)]
        fwd_b, upd_b = algorithmic_bytes(z, k)
        alg = [fwd_b * B, upd_b * B, 0.0]
        dom = int(np.argmax(avg))
        achieved = alg[dom] / (avg[dom] * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(names[dom], {}).get(str(B))
            except Exception:
                traffic = None
        out["roofline"] = {
            "bound": "hbm", "kernel": names[dom], "achieved": achieved, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "algorithmic_bytes_per_launch": alg[dom], "avg_launch_ms": avg[dom],
            "all_kernels_avg_ms": dict(zip(names, avg)),
            "whole_step": {"algorithmic_bytes": (fwd_b + upd_b) * B, "avg_ms": ms[3] / max(cnt[3], 1),
                           "achieved": (fwd_b + upd_b) * B / (ms[3] / max(cnt[3], 1) * 1e-3) / 1e9},
            "note": "V (n*k*8 B) is cache-resident at this size: algorithmic bytes count every touched "
                    "row as if streamed from HBM, so frac can exceed what the HBM counters show",
        }
        out["sampler"] = {"host_exact_mt19937_s_per_batch": sampler_s / (W + K),
                          "threads": min(os.cpu_count() or 1, 32)}
        if not args.no_extra and B != 2000:
            # the reference's own batch size (conf/setting/kuairec.yaml:52)
            plan2 = FmPlan(rt, csr, train["labels"], train["pscores"], k, 2000)
            ids2 = rt.upload(sample_batches(n_train, 2000, 0, 200))

            def run2(count):
                _lib.check(rt.lib.rfm_fm_train(rt.ctx, plan2.handle, *csr_ptrs, ids2.data_ptr(), 2000, count,
                                               *params, lr, None, None, None, None, None, 0, 1e-8, None, None))
            run2(20)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run2(200)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            out["extra"] = {"batch_2000": {"value": 200 * 2000 / dt, "unit": "examples/s",
                                           "ms_per_step": 1e3 * dt / 200}}
            plan2.close()
        if not args.no_cpu_baseline:
            v, steps_done, dt = cpu_baseline(train, ids[W:], k, lr, seed)
            out["cpu_baseline"] = {
                "value": v, "unit": "examples/s", "cores": 1, "kind": "port",
                "sample": (f"{steps_done} steps of the same workload (batch {gB}, same row ids) in {dt:.1f} s: "
                           "oracle/cpu_ref.fm_step_refstruct = the reference's SciPy op sequence incl. the "
                           "per-factor loop, row gather X[ids] included, sampler excluded; SciPy sparse "
                           f"kernels are single-threaded (host has {os.cpu_count()} cpus)"),
            }
    fence()
    plan.close()
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
