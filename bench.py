#!/usr/bin/env python3
"""Headline benchmark: FM (k=32) training examples/s on the KuaiRec-big-shaped
synthetic log (BASELINE.json config 3), one process per GPU.

    python bench.py --gpus N --steps K --warmup W

A *step* is one mini-batch pass of the hot path over rows already resident in
HBM: gather by row id -> forward -> IPS residual -> batch-sum gradients -> SGD
update of w0, w, V (src/fm.py:72-88 of the reference).  The row-id lists of the
batches (the reference's resample(..., random_state=epoch)) are produced by the
exact host sampler before the timed region and uploaded: they are inputs.
For N > 1 the global batch of N*B rows is sharded over the ranks, gradients are
exchanged over RCCL and every rank applies the same update (weak scaling: B rows
per GPU).  Rank 0 prints ONE JSON line.

What the line carries besides the contract's fields:
 * the timed region (exactly K steps between barrier + synchronize) is repeated
   --reps times; ``value`` / ``ms_per_step`` are the MEDIAN region, all regions
   are listed in ``extra.rep_ms_per_step``;
 * ``roofline``: the dominant kernel priced with SURVEY 8d's ALGORITHMIC bytes,
   the same model applied to the whole step (``whole_step_frac``; above 1 at
   this config because V is cache-resident -- then ``bound`` says
   "latency/L2", not "hbm"), a PHYSICAL model (``compulsory_hbm_bytes``: what
   must cross HBM from cold caches) and the COUNTER view (``traffic``: HBM bytes
   per launch from rocprofv3 FETCH_SIZE / WRITE_SIZE passes of this very command,
   collected by child processes before this process touches the GPU);
 * ``extra.fit_wall``: variant (A), ``FactorizationMachines.fit`` exactly as the
   reference runs it (second batch forward + per-iteration validation forward,
   sampler, uploads and plan build included) at B = 2 000 and 65 536, and
   ``cpu_baseline.fit_wall`` for the oracle's reference-structured fit;
 * ``extra.with_sampler``: the headline step with batch selection (R5) INSIDE the
   timed region, from cold -- next to ``value``, whose row ids are precomputed;
 * ``extra.published_config``: the operating point of every published run of the
   reference (k = 400 / B = 2 000, k = 300 / B = 500): step, per-kernel averages,
   fit() wall, the validation forward priced against the L2 gather rate, and
   ``vs_baseline`` against the examples/s derived from the reference's own run logs
   (``--published-only NAME`` runs just that: the command the rocprofv3 profiles of
   profiles/r3*/ are taken from);
 * ``extra.mf``: BASELINE config 5 -- logistic MF exact / HOGWILD / user-partition
   throughput, levels per batch, roofline fraction, loss gap (``--mf-only NAME``);
 * ``roofline.binding_level`` / ``frac_binding``: the level the line declares as
   binding (counter traffic vs the LDS atomic-add path), beside the algorithmic
   ``frac`` kept for continuity.
For N > 1 (``python -m torch.distributed.run ... bench.py --gpus N``) a timed region is
ONE C call per rank (``rfm_fm_fit_dp``: gradients of the shard, RCCL exchange, update).
"""
from __future__ import annotations

import argparse
import ctypes as C
import csv
import glob
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time
from collections import defaultdict

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# multi-process GPU work on this pool needs dmabuf IPC (RCCL's buffer exchange fails with
# hipIpcGetMemHandle: invalid argument otherwise); normally exported already
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
L2_GATHER_GBS = 17800.0  # same guide, 'Indexed rows: gather': rows shared by every workgroup, 16.8-18.8 TB/s
MALL_GATHER_GBS = 8600.0  # ... 38 MB table, uniformly random rows (Infinity Cache)
STEP_KERNELS = ["fm_forward_kernel", "fm_consume_kernel", "fm_finalize_kernel"]


def algorithmic_bytes(z: float, k: int, s: int = 8):
    """SURVEY.md 8d, per example: forward reads the CSR entries, label, pscore and
    each touched row of V / entry of w once; the update reads and writes them once."""
    fwd = z * (4 + s) + 8 + s + z * (k + 1) * s
    upd = 2 * z * (k + 1) * s
    return fwd, upd


def kernels_sha() -> str:
    """Identity of the kernel sources a traffic figure belongs to."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "relevance_factorizationmachine_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".hpp", ".h", ".cpp")):
            h.update(name.encode())
            h.update(open(os.path.join(csrc, name), "rb").read())
    return h.hexdigest()[:16]


def git_head() -> str | None:
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True,
                              text=True, timeout=10).stdout.strip() or None
    except Exception:  # noqa: BLE001
        return None


# ---------------------------------------------------------------------------
# HBM traffic from the PMC counters, measured by this run
# ---------------------------------------------------------------------------
def _per_kernel_counter(dirpath: str, counter: str):
    acc = defaultdict(lambda: [0.0, 0])
    for path in glob.glob(os.path.join(dirpath, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                name = row["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0].replace("rfm::", "")
                acc[name][0] += float(row["Counter_Value"])
                acc[name][1] += 1
    return {k: v[0] / v[1] for k, v in acc.items() if v[1]}


def collect_traffic(args) -> dict:
    """rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes: the TCC slots do
    not hold both; --kernel-trace only, as MI355X_MICROARCH.md's rocprofv3 section says) of
    this same bench command in child processes.  Called BEFORE this process initialises the
    GPU.  FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reports half the bytes of
    wide coalesced reads, so the read side is doubled (same guide, HBM section)."""
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return {"error": "rocprofv3 not found"}
    keep = os.path.join(ROOT, "gpurun_out", "bench_pmc")
    try:
        os.makedirs(keep, exist_ok=True)
    except OSError:
        keep = None
    work = tempfile.mkdtemp(prefix="rfm_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    child = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(args.steps), "--warmup",
             str(args.warmup), "--batch-size", str(args.batch_size), "--workload", args.workload,
             "--n-train", str(args.n_train), "--reps", "1", "--pmc-child"]
    per = {}
    t0 = time.perf_counter()
    for tag, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        out = os.path.join(work, tag)
        cmd = [prof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "--"] + child
        try:
            proc = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True,
                                  timeout=args.pmc_timeout)
        except subprocess.TimeoutExpired:
            return {"error": f"rocprofv3 --pmc {counter} pass timed out after {args.pmc_timeout} s"}
        if proc.returncode != 0:
            return {"error": f"rocprofv3 --pmc {counter} pass failed (rc {proc.returncode}): "
                             + (proc.stderr or proc.stdout)[-300:]}
        per[counter] = _per_kernel_counter(out, counter)
    shutil.rmtree(work, ignore_errors=True)
    kernels = {}
    for name in sorted(set(per["FETCH_SIZE"]) | set(per["WRITE_SIZE"])):
        f = per["FETCH_SIZE"].get(name, 0.0) * 1024
        w = per["WRITE_SIZE"].get(name, 0.0) * 1024
        kernels[name] = {"fetch_bytes_raw": f, "fetch_bytes_x2": 2 * f, "write_bytes": w, "hbm_bytes": 2 * f + w}
    if not any(k in kernels for k in STEP_KERNELS):
        return {"error": "the PMC passes saw none of the step's kernels"}
    res = {"kernels": kernels, "seconds": time.perf_counter() - t0, "kernels_sha": kernels_sha(),
           "commit": git_head(), "batch": args.batch_size, "workload": args.workload,
           "how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE --kernel-trace, one pass each, of "
                  "`bench.py --pmc-child` with this run's steps/warmup/batch; average per launch; "
                  "hbm_bytes = 2 x FETCH_SIZE x 1024 + WRITE_SIZE x 1024 (gfx950 FETCH_SIZE correction)"}
    if keep:
        try:
            json.dump(res, open(os.path.join(keep, "pmc_summary.json"), "w"), indent=1)
        except OSError:
            pass
    return res


def committed_traffic(batch: int, workload: str) -> dict | None:
    """Fallback when this run cannot profile itself: the newest committed
    profiles/r*/pmc_summary.json that names its commit and was taken from THESE kernel
    sources (kernels_sha); anything older than the kernels is refused."""
    sha = kernels_sha()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_summary.json")), reverse=True):
        try:
            d = json.load(open(path))
        except Exception:  # noqa: BLE001
            continue
        if d.get("kernels_sha") == sha and d.get("batch") == batch and d.get("workload") == workload \
                and "kernels" in d:
            d["path"] = os.path.relpath(path, ROOT)
            return d
    return None


# ---------------------------------------------------------------------------
# physical model: bytes that must cross HBM from cold caches, per step
# ---------------------------------------------------------------------------
def compulsory_bytes(X, rows: np.ndarray, k: int, hot_cols: np.ndarray, n_slabs: int) -> dict:
    """Per launch, from cold caches: every byte of the batch's records, ids, Q / residual,
    slot marks and hot-sum slabs once in each direction it has to travel, and every DISTINCT
    touched parameter row (V row + w entry) read once by the forward and read + written once
    by the update.  No cross-row reuse is assumed for the records; parameters count once."""
    B = int(rows.shape[0])
    sub = X[rows]
    cols = sub.indices
    nnz = int(cols.shape[0])
    distinct = int(np.unique(cols).shape[0])
    is_hot = np.zeros(X.shape[1], dtype=bool)
    is_hot[hot_cols] = True
    sparse_entries = int(np.count_nonzero(~is_hot[cols]))
    H = int(hot_cols.shape[0])
    param_row = (k + 1) * 8
    slab = n_slabs * H * (k + 2) * 8
    fwd = (B * 4 + B * 32 + nnz * 16          # ids, row records, entry records
           + distinct * param_row              # touched rows of V, entries of w
           + B * k * 8                         # Q out
           + sparse_entries * 16               # {position, residual} marks (+ one bit each)
           + slab + n_slabs * 8)               # hot sums, residual partials
    cons = (sparse_entries * (16 + 16)        # marks read, slot records
            + sparse_entries * k * 8           # Q rows of the marked slots
            + slab                             # slabs read back
            + 2 * distinct * param_row)        # every touched row read + written (hot ones too)
    fin = 0                                    # only for columns longer than a workgroup's tasks
    return {"forward": fwd, "consume": cons, "finalize": fin, "step": fwd + cons + fin,
            "distinct_columns": distinct, "sparse_entries": sparse_entries, "hot_columns": H}


def cpu_baseline(train, ids, k, lr, seed, budget_s=10.0):
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


def cpu_fit_wall(train, val, k, lr, seed, batch, n_epochs):
    """Variant (A) on the host: the oracle's fit() in the reference's structure
    (src/fm.py:71-102: resample, step, train-loss forward, validation forward)."""
    from oracle import cpu_ref

    t0 = time.perf_counter()
    cpu_ref.fm_fit(train, val, n_epochs=n_epochs, n_factors=k, lr=lr, batch_size=batch, seed=seed,
                   form="refstruct")
    dt = time.perf_counter() - t0
    return {"batch_size": batch, "iterations": n_epochs, "ms_per_iteration": 1e3 * dt / n_epochs,
            "value": n_epochs * batch / dt, "unit": "examples/s"}


# the reference's published runs (examples / wall of the run's log
# interval -- derived, hardware unstated, an upper bound on fit() time)
PUBLISHED = (
    # name, synthetic shape, k, B, lr, iterations, reference examples/s, source
    ("kuairec_fm_ips", "kuairec_small", 400, 2000, 9e-6, 221, 442000 / 137.3,
     "logs/kuairec/main_kuairec.log:24-25 + data/best_params/kuairec/FM_IPS.json"),
    ("coat_fm_ips", "coat", 300, 500, 1e-4, 401, 200500 / 63.4,
     "logs/coat/main_coat.log:19-20 + data/best_params/coat/FM_IPS.json"),
)


def published_config(rt, only: str | None = None) -> dict:
    """``extra.published_config``: the operating point of every published run of the reference
    (conf/setting/kuairec.yaml:50-59: k=400, B=2000; coat.yaml:27-36: k=300, B=500) on the
    synthetic log of that shape -- step-only (ids precomputed, as ``value``), per-kernel HIP-event
    averages, and the fit() wall exactly as the reference runs it, next to the examples/s
    derived from the reference's own run logs."""
    import torch

    from relevance_factorizationmachine_amd import _lib, synth
    from relevance_factorizationmachine_amd.fm import FactorizationMachines, FmPlan
    from relevance_factorizationmachine_amd.runtime import DeviceCSR, sample_batches

    out = {}
    for name, shape_name, k, B, lr, its, ref_ex_s, src in PUBLISHED:
        if only and only != name:
            continue
        train, val = synth.make_log(shape_name, "FM", "IPS", seed=0)
        X = train["features"]
        n, z = X.shape[1], X.nnz / X.shape[0]
        model = FactorizationMachines(estimator="IPS", n_epochs=its, n_factors=k, lr=lr, batch_size=B,
                                      seed=12345, n_features=n)
        csr = DeviceCSR(rt, X)
        y = rt.upload(train["labels"], dtype=np.float64)
        p = rt.upload(train["pscores"], dtype=np.float64)
        plan = FmPlan(rt, csr, y, p, k, B)
        warm, K = 20, 200
        d_ids = rt.upload(sample_batches(X.shape[0], B, 0, warm + K))
        args = (csr.indptr.data_ptr(), csr.indices.data_ptr(), csr.values.data_ptr(), y.data_ptr(), p.data_ptr())
        params = (model.w0.dev.data_ptr(), model.w.dev.data_ptr(), model.V.dev.data_ptr())

        def run(first, count):
            _lib.check(rt.lib.rfm_fm_train(rt.ctx, plan.handle, *args, d_ids.data_ptr() + first * B * 4, B, count,
                                           *params, lr, None, None, None, None, None, 0, 1e-8, None, None))
        run(0, warm)
        reg = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run(warm, K)
            torch.cuda.synchronize()
            reg.append((time.perf_counter() - t0) / K)
        dt = float(np.median(reg))
        ms = (C.c_double * 4)()
        cnt = (C.c_int64 * 4)()
        _lib.check(rt.lib.rfm_profile_begin(rt.ctx))
        run(warm, K)
        _lib.check(rt.lib.rfm_profile_end(rt.ctx, ms, cnt))
        info = plan.info()
        split = info["split_columns"] > 0
        fwd_b, upd_b = algorithmic_bytes(z, k)
        plan.close()
        # the validation forward (the largest launch of an iteration of fit() at this width): a
        # pure gather of V rows.  V (n*k*8 bytes) is beyond one XCD's 4 MiB L2 at k = 400, so it is
        # priced against the guide's gathered-row rates: rows served from the XCDs' L2
        # (16.8-18.8 TB/s chip-wide) and from the Infinity Cache (8.6 TB/s)
        vX = val["features"]
        vcsr = DeviceCSR(rt, vX)
        scores = rt.empty((vX.shape[0],), torch.float64)

        def val_forward():
            _lib.check(rt.lib.rfm_fm_forward(rt.ctx, vcsr.indptr.data_ptr(), vcsr.indices.data_ptr(),
                                             vcsr.values.data_ptr(), None, vX.shape[0], *params, n, k,
                                             scores.data_ptr()))
        for _ in range(5):
            val_forward()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            val_forward()
        torch.cuda.synchronize()
        vf = (time.perf_counter() - t0) / 100
        gather = int(vX.nnz) * (k + 1) * 8
        val_roof = {"rows": int(vX.shape[0]), "ms_per_launch": 1e3 * vf, "gathered_bytes": gather,
                    "achieved": gather / vf / 1e9, "unit": "GB/s", "bound": "l2-gather",
                    "peak": L2_GATHER_GBS, "frac": gather / vf / 1e9 / L2_GATHER_GBS,
                    "frac_of_infinity_cache_gather": gather / vf / 1e9 / MALL_GATHER_GBS,
                    "what": "nnz x (k+1) x 8 bytes of V / w rows gathered per launch of rfm_fm_forward on the "
                            "validation split, 100 back-to-back launches (wall / 100); peaks: MI355X guide, "
                            "'Indexed rows: gather' table"}
        # both loss forwards of an iteration (the batch's rows with the new parameters + the validation
        # split): at even k > 128 ONE launch sliced by factors that keeps the frequent columns' slices
        # in LDS (csrc/rfm_fm_sliced.hpp) -- timed as (iterations with losses) - (iterations without),
        # priced in the same model: the bytes the PLAIN forward would gather from L2
        vy = rt.upload(val["labels"], dtype=np.float64)
        vp = rt.upload(val["pscores"], dtype=np.float64)
        plan2 = FmPlan(rt, csr, y, p, k, B)
        tl = rt.empty((K,), torch.float64)
        vl = rt.empty((K,), torch.float64)

        def run_losses(first, count, losses):
            _lib.check(rt.lib.rfm_fm_train(
                rt.ctx, plan2.handle, *args, d_ids.data_ptr() + first * B * 4, B, count, *params, lr,
                vcsr.indptr.data_ptr(), vcsr.indices.data_ptr(), vcsr.values.data_ptr(), vy.data_ptr(),
                vp.data_ptr(), vX.shape[0], 1e-8, tl.data_ptr() if losses else None,
                vl.data_ptr() if losses else None))
        per = {}
        for losses in (True, False):
            run_losses(0, warm, losses)
            regs = []
            for _ in range(5):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                run_losses(warm, K, losses)
                torch.cuda.synchronize()
                regs.append((time.perf_counter() - t0) / K)
            per[losses] = float(np.median(regs))
        sliced_info = plan2.sliced()
        plan2.close()
        loss_gather = (int(vX.nnz) + int(B * z)) * (k + 1) * 8
        loss_dt = max(per[True] - per[False], 1e-9)
        loss_roof = {"rows": int(vX.shape[0]) + B, "ms_per_iteration": 1e3 * loss_dt,
                     "form": ("sliced by factors: %d slices of %d factors, %d columns' slices in LDS"
                              % (sliced_info["slices"], sliced_info["factors_per_slice"], sliced_info["cached_columns"]))
                     if sliced_info["slices"] and vX.shape[0] + B >= 4096 else "plain forwards",
                     "plain_model_gathered_bytes": loss_gather, "achieved": loss_gather / loss_dt / 1e9,
                     "unit": "GB/s", "peak": L2_GATHER_GBS, "frac": loss_gather / loss_dt / 1e9 / L2_GATHER_GBS,
                     "bound": "valu-issue (sliced) / l2-gather (plain)",
                     "what": "(rfm_fm_train with both losses) - (without), 200 iterations each, median of 5; the bytes "
                             "are what the plain forward gathers from L2 for these rows (nnz x (k+1) x 8): above 1.0 "
                             "= faster than a forward that gathers every entry's row of V can be"}
        # the step's own L2 traffic in the same model: forward gathers, then every marked entry
        # re-reads its Q row and every distinct column's row of V is read and written once
        distinct = int(np.unique(X[sample_batches(X.shape[0], B, warm, 1)[0]].indices).shape[0])
        step_l2 = int(B * z * (k + 1) * 8 + B * k * 8 + B * z * k * 8 + 2 * distinct * (k + 1) * 8)
        # fit() exactly as the reference runs it (both loss forwards per iteration)
        walls = {}
        kw = dict(estimator="IPS", n_factors=k, lr=lr, seed=12345, n_features=n, batch_size=B)
        FactorizationMachines(n_epochs=3, **kw).fit(train, val)
        for state in ("cold", "again"):
            if state == "cold":
                rt.clear_caches()
            m = FactorizationMachines(n_epochs=its, **kw)
            t0 = time.perf_counter()
            m.fit(train, val)
            walls[state] = time.perf_counter() - t0
        # the hyper-parameter search's iteration (utils/search_params.py:79-123: fit with a
        # ValEvaluator -- every iteration scores the evaluation split and takes its IPS-DCG@5):
        # KuaiRec only (the reference logs 657 s for 500 such iterations, main_kuairec.log:14-15)
        search = None
        if name == "kuairec_fm_ips":
            n_eval = 65_471  # logs/kuairec/main_kuairec.log:9 (validation rows before sampling)
            _, ef = synth.make_log(shape_name, "FM", "IPS", seed=1, n_val=n_eval)
            _, em = synth.make_log(shape_name, "MF", "IPS", seed=1, n_val=n_eval)
            keep_rows = synth.first_occurrences(em["features"])
            frame = synth.interaction_frame({kk: v[keep_rows] for kk, v in em.items()}, em["features"][keep_rows])

            class _Hook:  # the attributes of the reference's ValEvaluator; opts in to the device metric
                metric_name, k, rfm_device_evaluator = "DCG", 5, True

                def __init__(self, fr, feats):
                    import pandas as pd
                    self.interaction_df, self.features = pd.DataFrame(fr), {"FM": feats}

                def evaluate(self, y_scores, estimator):
                    raise RuntimeError("the bench expects the device evaluator")

            s_its, s_walls = 500, []  # (conf/setting/kuairec.yaml: the search's n_epochs)
            for _ in range(2):
                hook = _Hook(frame, ef["features"][keep_rows])
                m = FactorizationMachines(n_epochs=s_its, evaluator=hook, **kw)
                t0 = time.perf_counter()
                m.fit(train, val)
                s_walls.append(time.perf_counter() - t0)
            search = {"iterations": s_its, "evaluation_rows": int(keep_rows.shape[0]),
                      "ms_per_iteration": 1e3 * s_walls[0] / s_its,
                      "ms_per_iteration_second_fit_same_log": 1e3 * s_walls[1] / s_its,
                      "host_evaluator_iterations": int(m.evaluator_host_calls),
                      "reference_ms_per_iteration": 657e3 / 500,
                      "vs_reference": (657e3 / 500) / (1e3 * s_walls[0] / s_its),
                      "what": "fit(train, val) with a ValEvaluator-like hook: step + both losses + scores of the "
                              "evaluation split (rfm_fm_plan_forward: sliced by factors) + IPS-DCG@5 on the device, "
                              "one iteration per rfm_fm_train call; reference: logs/kuairec/main_kuairec.log:14-15 "
                              "(500 iterations in 657 s, hardware unstated)"}
        out[name] = {
            "search_iteration": search,
            "workload": f"{shape_name}-shaped synthetic log, n_features={n}, {z:.0f} nnz/row, N_train={X.shape[0]}, "
                        f"N_val={val['features'].shape[0]}, FM k={k}, IPS, lr={lr}, batch_size={B}",
            "V_bytes": n * k * 8,
            "step": {"ms_per_step": 1e3 * dt, "value": B / dt, "unit": "examples/s",
                     "kernels_avg_ms": dict(zip(STEP_KERNELS, [ms[i] / max(cnt[i], 1) if i < 2 or split else 0.0
                                                              for i in range(3)])),
                     "hot_columns": info["hot_columns"], "tasks": info["tasks"], "task_words": info["task_words"],
                     "algorithmic_bytes_per_step": (fwd_b + upd_b) * B,
                     "algorithmic_frac_of_hbm_peak": (fwd_b + upd_b) * B / dt / 1e9 / HBM_PEAK_GBS,
                     "l2_bytes_per_step": step_l2, "l2_achieved_GBs": step_l2 / dt / 1e9,
                     "frac_of_l2_gather_peak": step_l2 / dt / 1e9 / L2_GATHER_GBS},
            "validation_forward_roofline": val_roof,
            "loss_forwards_roofline": loss_roof,
            "fit_wall": {"iterations": its, "ms_per_iteration": 1e3 * walls["cold"] / its,
                         "value": its * B / walls["cold"], "unit": "examples/s",
                         "ms_per_iteration_second_fit_same_log": 1e3 * walls["again"] / its,
                         "value_second_fit_same_log": its * B / walls["again"]},
            "reference": {"value": ref_ex_s, "unit": "examples/s", "source": src,
                          "note": "derived from the reference's committed run log (interval of the whole "
                                  "load+fit+predict+evaluate block, hardware unstated): an upper bound on fit() time"},
            "vs_baseline": its * B / walls["cold"] / ref_ex_s,
        }
    return out


MF_CONFIGS = (
    # name, synthetic shape, k, rows of the log, batch sizes
    ("c2_kuairec_small_k16", "kuairec_small", 16, None, (2000,)),
    # the reference's published MF runs: k = 400, B = 2 000 (conf/setting/kuairec.yaml:50-59)
    ("published_kuairec_small_k400", "kuairec_small", 400, None, (2000,)),
    ("c5_1m_x_100k_k128", "synthetic_1m", 128, 2_000_000, (2000, 65536)),
)


def mf_config(rt, only: str | None = None) -> dict:
    """``extra.mf`` (BASELINE config 5; src/mf.py:97-108): the logistic-MF batch in its three
    forms -- EXACT (the reference's strictly sequential per-example SGD through the level
    schedule), HOGWILD (unordered, non-parity) and the user-range partition with one rank (the
    exact batch through the multi-GPU code path) -- as step-only examples/s with the schedules
    precomputed (as the FM headline has its row ids precomputed), the fit() wall, levels per
    batch, the algorithmic fraction of the HBM roofline (4*k*s + 6*s + 16 bytes per
    example) and the validation-loss gap of the non-parity mode against the exact fit."""
    import torch

    from relevance_factorizationmachine_amd import _lib, synth
    from relevance_factorizationmachine_amd.dist import hip_mf_partition_worker
    from relevance_factorizationmachine_amd.mf import DevicePairs, LogisticMatrixFactorization
    from relevance_factorizationmachine_amd.runtime import mf_cache_capacity, mf_schedule_ex, sample_batches

    out = {}
    for name, shape_name, k, n_train, batches in MF_CONFIGS:
        if only and only != name:
            continue
        shape = synth.SHAPES[shape_name]
        train, val = synth.make_log(shape, "MF", "IPS", seed=0, n_train=n_train, n_val=min(shape.n_val, 20000))
        pairs = train["features"]
        n_rows = pairs.shape[0]
        bytes_per_example = 4 * k * 8 + 6 * 8 + 16
        h_y = np.ascontiguousarray(train["labels"], dtype=np.float64)
        h_p = np.ascontiguousarray(train["pscores"], dtype=np.float64)
        cache_cap = mf_cache_capacity(k)
        res = {"workload": f"{shape_name}-shaped (user, item) pairs {shape.n_users}x{shape.n_items}, Zipf(1.3) items, "
                           f"N_train={n_rows}, MF k={k}, IPS, lr=0.01, reg=0.5",
               "algorithmic_bytes_per_example": bytes_per_example}
        for B in batches:
            E = 40 if B <= 4096 else 8
            kw = dict(estimator="IPS", n_factors=k, lr=0.01, batch_size=B, seed=12345, n_users=shape.n_users,
                      n_items=shape.n_items, reg=0.5)
            model = LogisticMatrixFactorization(n_epochs=E, **kw)
            model.b = float(np.mean(train["labels"]))
            tr = DevicePairs(rt, pairs)
            y, p = rt.upload(h_y), rt.upload(h_p)
            ids = sample_batches(n_rows, B, 0, E)
            d_ids = rt.upload(ids)
            params = (model.P.dev.data_ptr(), model.Q.dev.data_ptr(), model.b_u.dev.data_ptr(),
                      model.b_i.dev.data_ptr())
            # ---- exact: schedules derived and uploaded beforehand -------------------------
            t0 = time.perf_counter()
            sched = [mf_schedule_ex(tr.h_users[r], tr.h_items[r], h_y[r], h_p[r], shape.n_users, shape.n_items,
                                    cache_cap) for r in ids]
            sched_s = (time.perf_counter() - t0) / E
            dev = [(rt.upload(ex.view(np.uint8)), lp, rt.upload(lp),
                    rt.upload(ci if ci.size else np.zeros(1, np.int32)), int(ci.size)) for ex, lp, ci in sched]

            def run_exact():
                for d_ex, lp, d_lp, d_ci, n_ci in dev:
                    _lib.check(rt.lib.rfm_mf_sgd_levels_ex(rt.ctx, d_ex.data_ptr(), lp.ctypes.data, d_lp.data_ptr(),
                                                           len(lp) - 1, d_ci.data_ptr(), n_ci, *params, model.b, k,
                                                           0.01, 0.5))

            def run_hogwild():
                for e in range(E):
                    _lib.check(rt.lib.rfm_mf_sgd_hogwild(rt.ctx, tr.users.data_ptr(), tr.items.data_ptr(),
                                                         y.data_ptr(), p.data_ptr(), d_ids.data_ptr() + e * B * 4, B,
                                                         *params, model.b, k, 0.01, 0.5))

            def timed(fn):
                fn()
                reg = []
                for _ in range(3):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    fn()
                    torch.cuda.synchronize()
                    reg.append((time.perf_counter() - t0) / E)
                return float(np.median(reg))

            levels = [len(lp) - 1 for _, lp, _ in sched]
            entry = {"batches_timed": E, "levels_per_batch": float(np.mean(levels)),
                     "host_schedule_ms_per_batch": 1e3 * sched_s}
            for mode, fn in (("exact", run_exact), ("hogwild", run_hogwild)):
                dt = timed(fn)
                entry[mode] = {"ms_per_batch": 1e3 * dt, "value": B / dt, "unit": "examples/s",
                               "algorithmic_frac_of_hbm_peak": bytes_per_example * B / dt / 1e9 / HBM_PEAK_GBS}
            entry["exact"]["us_per_level"] = 1e3 * entry["exact"]["ms_per_batch"] / max(np.mean(levels), 1)
            # ---- user-range partition with one rank: the exact batch through that code path
            pm = LogisticMatrixFactorization(n_epochs=E, **kw)
            step, finish = hip_mf_partition_worker(rt, pm, train, 1, 0)
            for it in range(2):
                step.step(it, step.users_of(it))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n_part = min(E, 10)
            for it in range(2, 2 + n_part):
                step.step(it, step.users_of(it))
            finish()
            dt = (time.perf_counter() - t0) / n_part
            entry["user_partition_world1"] = {
                "ms_per_batch": 1e3 * dt, "value": B / dt, "unit": "examples/s",
                "what": "dist.MfUserPartitionStep with one rank: sampler + host schedule of the next batch prepared by a worker thread, upload + kernels per batch"}
            # ---- fit() wall and the loss gap of the non-parity mode -------------------------
            fits = {}
            for mode in ("exact", "hogwild"):
                m = LogisticMatrixFactorization(n_epochs=E, **kw)
                m.hogwild = mode == "hogwild"
                t0 = time.perf_counter()
                trl, val_l = m.fit(train, val)
                fits[mode] = (time.perf_counter() - t0, val_l[-1], trl[-1])
            entry["fit_wall"] = {m_: {"ms_per_iteration": 1e3 * v[0] / E, "value": E * B / v[0], "unit": "examples/s",
                                      "final_val_loss": v[1]} for m_, v in fits.items()}
            entry["hogwild_val_loss_gap_vs_exact"] = abs(fits["hogwild"][1] - fits["exact"][1]) / abs(fits["exact"][1])
            res[f"batch_{B}"] = entry
            del dev, model, pm, tr
        out[name] = res
    return out


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--reps", type=int, default=5,
                    help="repetitions of the K-step timed region; the median is reported")
    ap.add_argument("--batch-size", type=int, default=65536, help="rows per GPU per step")
    ap.add_argument("--n-train", type=int, default=0, help="rows of the synthetic log (0 = config)")
    ap.add_argument("--workload", default="kuairec_big")
    ap.add_argument("--backend", default="nccl",
                    help="torch.distributed backend for --gpus > 1 (nccl = RCCL; gloo only to rehearse "
                         "the multi-rank path on a box with fewer GPUs than ranks)")
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: --batch-size is the GLOBAL batch, split over the GPUs "
                         "(default: weak scaling, --batch-size rows per GPU)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "rows", "dense"],
                    help="multi-GPU gradient exchange: touched rows only, or the dense [G_V|g_w|g_w0] "
                         "all-reduce the north star names as the baseline; auto = rows when the touched "
                         "rows of a global batch are estimated at under a quarter of the dense buffer")
    ap.add_argument("--no-direct-rccl", action="store_true",
                    help="multi-GPU dense exchange through torch.distributed instead of the C ABI's RCCL binding")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 counter passes")
    ap.add_argument("--pmc-timeout", type=int, default=240)
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--mf-only", default=None, metavar="NAME|all",
                    help="run only extra.mf (c2_kuairec_small_k16, c5_1m_x_100k_k128 or all) and print it")
    ap.add_argument("--published-only", default=None, metavar="NAME|all",
                    help="run only extra.published_config (kuairec_fm_ips, coat_fm_ips or all) and print it: "
                         "the command the rocprofv3 profiles of the published operating point are taken from")
    args = ap.parse_args()
    if args.pmc_child:
        args.no_cpu_baseline = args.no_extra = args.no_pmc = True

    if args.mf_only:
        from relevance_factorizationmachine_amd.runtime import Runtime
        res = mf_config(Runtime.get(0), None if args.mf_only == "all" else args.mf_only)
        print(json.dumps({"mf": res}))
        return
    if args.published_only:
        from relevance_factorizationmachine_amd.runtime import Runtime
        res = published_config(Runtime.get(0), None if args.published_only == "all" else args.published_only)
        print(json.dumps({"published_config": res}))
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")

    # the counter passes run this command again under rocprofv3 in child processes; they
    # come first because a process that has initialised the GPU must not start them
    traffic_info = None
    if world == 1 and not args.no_pmc:
        traffic_info = collect_traffic(args)

    import torch
    import torch.distributed as dist

    from relevance_factorizationmachine_amd import _lib, synth
    from relevance_factorizationmachine_amd.dist import hip_fm_worker, init_direct_rccl
    from relevance_factorizationmachine_amd.fm import FactorizationMachines, FmPlan
    from relevance_factorizationmachine_amd.runtime import DeviceCSR, Runtime, sample_batches

    device = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(device)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(args.backend)

    shape = synth.SHAPES[args.workload]
    n_train = args.n_train or shape.n_train
    k, B, K, W, reps = shape.n_factors, args.batch_size, args.steps, args.warmup, max(args.reps, 1)
    lr, seed = 9e-6, 12345  # conf/setting/kuairec.yaml: FM/IPS lr, seed
    if args.strong:
        if B % world:
            raise SystemExit(f"--strong: global batch {B} is not a multiple of {world} GPUs")
        B //= world
    gB = B * world
    if gB > n_train:
        raise SystemExit(f"global batch {gB} exceeds the log ({n_train} rows)")

    want_fit = rank == 0 and world == 1 and not args.no_extra
    train, val = synth.make_log(shape, "FM", "IPS", seed=0, n_train=n_train,
                                n_val=shape.n_val if want_fit else 16)
    X = train["features"]
    n = X.shape[1]
    z = X.nnz / X.shape[0]

    rt = Runtime.get(device)
    model = FactorizationMachines(estimator="IPS", n_epochs=K, n_factors=k, lr=lr, batch_size=B,
                                  seed=seed, n_features=n)
    csr = DeviceCSR(rt, X)
    y = rt.upload(train["labels"], dtype=np.float64)
    p = rt.upload(train["pscores"], dtype=np.float64)
    n_batches = W + reps * K
    t_s = time.perf_counter()
    ids = sample_batches(n_train, gB, 0, n_batches)  # exact resample() ids, same on every rank
    sampler_s = time.perf_counter() - t_s
    d_ids = rt.upload(ids)
    plan = FmPlan(rt, csr, y, p, k, B)
    csr_ptrs = (csr.indptr.data_ptr(), csr.indices.data_ptr(), csr.values.data_ptr(),
                y.data_ptr(), p.data_ptr())
    params = (model.w0.dev.data_ptr(), model.w.dev.data_ptr(), model.V.dev.data_ptr())

    transport = None
    collective = None
    if world == 1:
        def run(first: int, count: int) -> None:
            _lib.check(rt.lib.rfm_fm_train(
                rt.ctx, plan.handle, *csr_ptrs, d_ids.data_ptr() + first * B * 4, B, count, *params,
                lr, None, None, None, None, None, 0, 1e-8, None, None))
    else:
        from relevance_factorizationmachine_amd.dist import HostStagedTransport, choose_exchange

        rows = args.exchange == "rows" or (args.exchange == "auto" and choose_exchange(n, k, gB) == "rows")
        # the whole loop of a timed region is ONE C call (rfm_fm_fit_dp): gradients of the shard,
        # exchange and update enqueued on the compute stream; RCCL is called by the library
        # itself.  Backends other than nccl (rehearsals on a box with fewer GPUs than ranks)
        # stage the same exchanges through the host.
        direct, staged, c_transport = False, None, None
        if args.backend == "nccl" and not args.no_direct_rccl:
            try:
                direct = init_direct_rccl(rt, world, rank)
            except Exception as exc:  # noqa: BLE001
                print(f"[rank {rank}] direct RCCL unavailable ({exc}); using torch.distributed", file=sys.stderr)
            flag = torch.tensor([1 if direct else 0], device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)  # all ranks or none
            direct = bool(flag.item())
        elif args.backend != "nccl":
            staged = HostStagedTransport(world, rank, rt=rt)
            c_transport = staged.c_struct()
        if rows:
            collective = ("touched rows: gradient records to the rank that owns the column (all-to-all), "
                          "owner-side rank-ordered sum + update, updated rows to every rank (all-to-all); "
                          "transfer sizes derived from the row ids before the loop")
        else:
            collective = (f"all-reduce(sum) of the dense [G_V|g_w|g_w0] buffer, "
                          f"{8 * (n * (k + 1) + 1) / 1e6:.1f} MB")
        if direct or staged is not None:
            transport = "rccl-direct (rfm_fm_fit_dp)" if direct else f"host-staged {args.backend} (rfm_fm_fit_dp)"

            def run(first: int, count: int) -> None:
                rc = rt.lib.rfm_fm_fit_dp(
                    rt.ctx, plan.handle, C.byref(c_transport) if c_transport is not None else None,
                    1 if rows else 0, d_ids.data_ptr() + first * gB * 4, gB, count, *params, lr,
                    None, None, None, None, None, 0, 1e-8, None, None)
                if rc != 0 and staged is not None and staged.error is not None:
                    raise staged.error
                _lib.check(rc)
        elif rows:
            # fallback (RCCL could not be bound by the library): the exchange step by step through
            # torch.distributed
            from relevance_factorizationmachine_amd.dist import RowExchange, hip_fm_rows_worker

            ex = RowExchange.for_torch(dist, world, rank, n, k, backend=args.backend)
            worker = hip_fm_rows_worker(rt, plan, d_ids, gB, model, world, rank, lr, ex)
            transport = f"torch.distributed/{args.backend} all_to_all_single (per-step host sync)"

            def run(first: int, count: int) -> None:
                for it in range(first, first + count):
                    worker.step(it, gB)
        else:
            grad = rt.empty((n * (k + 1) + 1,), torch.float64)
            worker = hip_fm_worker(rt, plan, csr, y, p, d_ids, gB, model, grad, world, rank, lr, None)
            transport = f"torch.distributed/{args.backend}"

            def run(first: int, count: int) -> None:
                for it in range(first, first + count):
                    worker.step(it, gB)

    def fence() -> None:
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run(0, W)
    region = []
    for r in range(reps):
        fence()
        t0 = time.perf_counter()
        run(W + r * K, K)
        fence()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        region.append(elapsed)
    elapsed = float(np.median(region))
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
            "collective": f"{collective}, {transport}" if world > 1 else None,
            "replicas_in_sync": in_sync,
            "timed_regions": reps,
        },
        "extra": {"rep_ms_per_step": [1e3 * e / K for e in region]},
    }

    if rank == 0 and world == 1 and not args.pmc_child:
        # per-kernel durations with HIP events on the launch stream (same ids, one more region)
        ms = (C.c_double * 4)()
        cnt = (C.c_int64 * 4)()
        _lib.check(rt.lib.rfm_profile_begin(rt.ctx))
        run(W, K)
        _lib.check(rt.lib.rfm_profile_end(rt.ctx, ms, cnt))
        avg = [ms[i] / max(cnt[i], 1) for i in range(3)]
        info = plan.info()
        # a plan without split columns launches no finalize kernel: the third interval is then
        # two back-to-back events, i.e. what one event pair itself costs on this stream
        event_gap_ms = None
        if info["split_columns"] == 0:
            event_gap_ms, avg[2] = avg[2], 0.0
        fwd_b, upd_b = algorithmic_bytes(z, k)
        alg = [fwd_b * B, upd_b * B, 0.0]
        dom = int(np.argmax(avg))
        achieved = alg[dom] / (avg[dom] * 1e-3) / 1e9
        step_ms = 1e3 * elapsed / K  # the driver-visible step (median region), not the event sum
        whole_alg = (fwd_b + upd_b) * B
        whole_achieved = whole_alg / (step_ms * 1e-3) / 1e9
        comp = compulsory_bytes(X, ids[W][:B], k, plan.hot_columns(), info["forward_workgroups"])
        comp_keys = ["forward", "consume", "finalize"]

        # counter view: measured by this run's child passes; else a committed summary taken
        # from these very kernel sources; else null with the reason
        traffic = traffic_all = traffic_source = traffic_error = None
        if traffic_info and "kernels" in traffic_info:
            traffic_all = {kname: traffic_info["kernels"].get(kname, {}).get("hbm_bytes") for kname in STEP_KERNELS}
            traffic_source = {"measured": "live, by this run", "how": traffic_info["how"],
                              "commit": traffic_info["commit"], "kernels_sha": traffic_info["kernels_sha"],
                              "seconds": round(traffic_info["seconds"], 1)}
        else:
            traffic_error = (traffic_info or {}).get("error", "counter passes skipped (--no-pmc)")
            old = committed_traffic(B, args.workload)
            if old:
                traffic_all = {kname: old["kernels"].get(kname, {}).get("hbm_bytes") for kname in STEP_KERNELS}
                traffic_source = {"measured": "earlier run, committed", "path": old["path"],
                                  "commit": old.get("commit"), "kernels_sha": old["kernels_sha"],
                                  "live_error": traffic_error}
        if traffic_all:
            traffic = traffic_all.get(STEP_KERNELS[dom])
        step_traffic = sum(v for v in (traffic_all or {}).values() if v) if traffic_all else None
        bound = "latency/L2" if whole_achieved > HBM_PEAK_GBS else "hbm"
        out["roofline"] = {
            "bound": bound,
            "priced_against": "hbm",
            "kernel": STEP_KERNELS[dom],
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_source": traffic_source,
            "traffic_error": None if traffic_source and traffic_source["measured"].startswith("live") else traffic_error,
            "frac_traffic": (traffic / (avg[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
            "algorithmic_bytes_per_launch": alg[dom],
            "compulsory_hbm_bytes": comp[comp_keys[dom]],
            "frac_compulsory": comp[comp_keys[dom]] / (avg[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "avg_launch_ms": avg[dom],
            "all_kernels_avg_ms": dict(zip(STEP_KERNELS, avg)),
            "event_pair_gap_ms": event_gap_ms,
            "whole_step": {
                "ms": step_ms, "event_sum_ms": ms[3] / max(cnt[3], 1),
                "algorithmic_bytes": whole_alg, "achieved": whole_achieved,
                "compulsory_hbm_bytes": comp["step"],
                "traffic": step_traffic,
                "traffic_per_kernel": traffic_all,
            },
            "whole_step_frac": whole_achieved / HBM_PEAK_GBS,
            "whole_step_frac_compulsory": comp["step"] / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "whole_step_frac_traffic": (step_traffic / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if step_traffic else None,
            "compulsory_model": {kk: comp[kk] for kk in ("distinct_columns", "sparse_entries", "hot_columns")},
            # the forward's on-chip side: every hot entry adds err*x*[q, 1, x] (k+2 doubles) into
            # the workgroup's LDS sums with ds_add_f64; priced against the LDS store path
            # (MI355X guide: ~85 B/clk/CU for 8-byte stores) over the WHOLE forward launch
            "lds": (lambda hot_entries, peak: {
                "hot_entries_per_launch": hot_entries,
                "atomic_add_bytes": hot_entries * (k + 2) * 8,
                "achieved": hot_entries * (k + 2) * 8 / (avg[0] * 1e-3) / 1e9,
                "peak": peak, "unit": "GB/s",
                "frac_of_forward_launch": hot_entries * (k + 2) * 8 / (avg[0] * 1e-3) / 1e9 / peak,
                "peak_is": "85 B/clk/CU x 256 CUs x 2.4 GHz (8-byte LDS stores)"})(
                    int(X[ids[W][:B]].nnz) - comp["sparse_entries"], 85 * 256 * 2.4),
            "note": ("algorithmic bytes (SURVEY 8d) count every touched parameter row as if streamed from HBM; "
                     "V (n*k*8 B = %.1f MB) is L2 / Infinity-Cache resident at this size, so a whole-step "
                     "fraction above 1 means the HBM model does not bind and the step is bound by gather "
                     "latency / L2 (bound = latency/L2); compulsory = bytes that must cross HBM from cold "
                     "caches; traffic = PMC counters" % (n * k * 8 / 1e6)),
        }
        out["sampler"] = {"host_exact_mt19937_s_per_batch": sampler_s / n_batches,
                          "threads": min(os.cpu_count() or 1, 32)}
        # the level the line itself declares as binding: the larger of the forward's measured
        # HBM traffic and its LDS atomic-add path, each against its own peak
        r = out["roofline"]
        levels = {"hbm (counters)": r["frac_traffic"], "lds (ds_add_f64 path)": r["lds"]["frac_of_forward_launch"]}
        levels = {kk: v for kk, v in levels.items() if v is not None}
        if levels:
            r["binding_level"] = max(levels, key=levels.get)
            r["frac_binding"] = levels[r["binding_level"]]
            r["frac_binding_all"] = levels
        if not args.no_extra:
            # R5 inside the timed region: the same step consuming its row ids as the pipelined exact
            # sampler produces them FROM COLD (nothing cached: every iteration's MT19937 shuffle of
            # arange(N) runs on the host cores inside the interval, first chunk included)
            from relevance_factorizationmachine_amd.runtime import ID_CACHE, BatchIdStream

            n_ws = 200
            ID_CACHE.clear()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            stream = BatchIdStream(rt, n_train, B, n_ws, need_host=False)
            try:
                for first, _host, dev_ids in stream.chunks():
                    _lib.check(rt.lib.rfm_fm_train(
                        rt.ctx, plan.handle, *csr_ptrs, dev_ids.data_ptr(), B, dev_ids.shape[0], *params,
                        lr, None, None, None, None, None, 0, 1e-8, None, None))
                torch.cuda.synchronize()
            finally:
                stream.close()
            dt_ws = time.perf_counter() - t0
            ID_CACHE.clear()
            out["extra"]["with_sampler"] = {
                "steps": n_ws, "ms_per_step": 1e3 * dt_ws / n_ws, "value": n_ws * B / dt_ws, "unit": "examples/s",
                "host_threads": min(os.cpu_count() or 1, 32),
                "what": "the headline step with batch selection (R5: resample(..., random_state=epoch), exact "
                        "MT19937 shuffle per iteration) INSIDE the timed region, from cold: sampler thread -> pinned "
                        "buffer -> copy stream -> step; `value` above has the ids precomputed"}
        if not args.no_extra:
            # the bitwise-reproducible mode (hot sums on chip in a fixed order: hot_min_count = -2)
            det = {}
            for db in sorted({B, 2000}):
                if db > n_train:
                    continue
                pland = FmPlan(rt, csr, y, p, k, db, -2)
                idsd = rt.upload(sample_batches(n_train, db, 0, 60))

                def rund(first, count):
                    _lib.check(rt.lib.rfm_fm_train(rt.ctx, pland.handle, *csr_ptrs,
                                                   idsd.data_ptr() + first * db * 4, db, count,
                                                   *params, lr, None, None, None, None, None, 0, 1e-8, None, None))
                rund(0, 10)
                rd = []
                for _ in range(5):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    rund(10, 50)
                    torch.cuda.synchronize()
                    rd.append((time.perf_counter() - t0) / 50)
                det[f"batch_{db}"] = {"ms_per_step": 1e3 * float(np.median(rd)),
                                      "value": db / float(np.median(rd)), "unit": "examples/s",
                                      "hot_columns": pland.info()["hot_columns"]}
                pland.close()
            out["extra"]["deterministic_mode"] = {
                **det, "what": "model.deterministic = True: every sum of a step in a fixed order, fits equal "
                               "bit for bit; the default sums the frequent columns with LDS float atomics"}
            if B != 2000:
                # the reference's own batch size (conf/setting/kuairec.yaml:52)
                plan2 = FmPlan(rt, csr, y, p, k, 2000)
                plan2_split = plan2.info()["split_columns"] > 0
                ids2 = rt.upload(sample_batches(n_train, 2000, 0, 220))

                def run2(first, count):
                    _lib.check(rt.lib.rfm_fm_train(rt.ctx, plan2.handle, *csr_ptrs,
                                                   ids2.data_ptr() + first * 2000 * 4, 2000, count,
                                                   *params, lr, None, None, None, None, None, 0, 1e-8, None, None))
                run2(0, 20)
                r2 = []
                for _ in range(5):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    run2(20, 200)
                    torch.cuda.synchronize()
                    r2.append((time.perf_counter() - t0) / 200)
                dt = float(np.median(r2))
                _lib.check(rt.lib.rfm_profile_begin(rt.ctx))
                run2(20, 200)
                _lib.check(rt.lib.rfm_profile_end(rt.ctx, ms, cnt))
                out["extra"]["batch_2000"] = {
                    "value": 2000 / dt, "unit": "examples/s", "ms_per_step": 1e3 * dt,
                    "kernels_avg_ms": dict(zip(STEP_KERNELS, [
                        ms[i] / max(cnt[i], 1) if i < 2 or plan2_split else 0.0 for i in range(3)])),
                    "event_pair_gap_ms": None if plan2_split else ms[2] / max(cnt[2], 1),
                    "algorithmic_frac": (fwd_b + upd_b) * 2000 / dt / 1e9 / HBM_PEAK_GBS}
                plan2.close()
            out["extra"]["published_config"] = published_config(rt)
            out["extra"]["mf"] = mf_config(rt)
            # variant (A): fit() exactly as the reference runs it (src/fm.py:71-102)
            fit = {}
            kw = dict(estimator="IPS", n_factors=k, lr=lr, seed=seed, n_features=n)
            FactorizationMachines(n_epochs=3, batch_size=2000, **kw).fit(train, val)  # warm
            for fb, its in ((2000, 200), (65536, 200)):
                if fb > n_train:
                    continue
                walls = {}
                for state in ("cold", "again"):
                    if state == "cold":
                        # nothing remembered: the log, labels and every iteration's ids are
                        # uploaded / sampled inside the timed fit
                        rt.clear_caches()
                    m = FactorizationMachines(n_epochs=its, batch_size=fb, **kw)
                    t0 = time.perf_counter()
                    m.fit(train, val)
                    walls[state] = time.perf_counter() - t0
                dt = walls["cold"]
                fit[f"batch_{fb}"] = {"iterations": its, "ms_per_iteration": 1e3 * dt / its,
                                      "value": its * fb / dt, "unit": "examples/s",
                                      "ms_per_iteration_second_fit_same_log": 1e3 * walls["again"] / its}
            out["extra"]["fit_wall"] = {
                **fit, "what": (f"FactorizationMachines.fit(train N={n_train}, val N={val['features'].shape[0]}) wall: "
                                "exact sampler and uploads of the log (every iteration's ids sampled inside the timed "
                                "fit; a second fit on the same split reuses the device copies, as the "
                                "reference's drivers would: listed separately), plan build, and per "
                                "iteration step + train-loss forward "
                                "(new parameters, same batch) + validation-loss forward")}
        if not args.no_cpu_baseline:
            v, steps_done, dt = cpu_baseline(train, ids[W:], k, lr, seed)
            out["cpu_baseline"] = {
                "value": v, "unit": "examples/s", "cores": 1, "kind": "port",
                "sample": (f"{steps_done} steps of the same workload (batch {gB}, same row ids) in {dt:.1f} s: "
                           "oracle/cpu_ref.fm_step_refstruct = the reference's SciPy op sequence incl. the "
                           "per-factor loop, row gather X[ids] included, sampler excluded; SciPy sparse "
                           f"kernels are single-threaded (host has {os.cpu_count()} cpus)"),
            }
            if not args.no_extra:
                out["cpu_baseline"]["fit_wall"] = {
                    "batch_2000": cpu_fit_wall(train, val, k, lr, seed, 2000, 20),
                    "batch_65536": cpu_fit_wall(train, val, k, lr, seed, 65536, 3) if n_train >= 65536 else None,
                    "what": "oracle/cpu_ref.fm_fit(form='refstruct'): resample + step + both loss forwards "
                            "per iteration, as src/fm.py:71-102",
                }
    fence()
    plan.close()
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
