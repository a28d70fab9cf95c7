"""ctypes binding of ``librfm_hip.so`` (C ABI declared in ``include/rfm_hip.h``).

The shared library is built in-tree by :func:`build` (``hipcc
--offload-arch=gfx950``); there is no CPU fallback -- if the library cannot be
loaded, or no GPU is visible when a device call is made, the caller gets an
exception, never a silently different code path.
"""
from __future__ import annotations

import ctypes as C
import os
import shutil
import subprocess
from typing import List, Optional

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
# RFM_LIB_PATH: load another build of the same ABI (timing experiments under profiles/)
LIB_PATH = os.environ.get("RFM_LIB_PATH") or os.path.join(PKG_DIR, "librfm_hip.so")
SOURCES = ["rfm_capi.hip", "rfm_fm.hip", "rfm_fm_plan.hip", "rfm_mf.hip", "rfm_eval.hip", "rfm_csr.hip", "rfm_host.cpp", "rfm_comm.cpp"]
# every header under csrc/ (a change of any of them rebuilds every object) + the C ABI
HEADERS = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hpp"))) + [
    os.path.join(os.path.dirname(PKG_DIR), "include", "rfm_hip.h")]

RFM_OK, RFM_ERR_BAD_ARG, RFM_ERR_HIP, RFM_ERR_NO_DEVICE, RFM_ERR_INTERNAL = range(5)


class RfmError(RuntimeError):
    """A call into librfm_hip.so failed (HIP error, no device, internal)."""


def _hipcc() -> Optional[str]:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    return None


def _stale() -> bool:
    if os.environ.get("RFM_LIB_PATH"):
        return False
    if not os.path.exists(LIB_PATH):
        return True
    built = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.exists(d) and os.path.getmtime(d) > built for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP extension for gfx950 in-tree; returns the .so path.  One object per
    source file (compiled concurrently, kept under ``csrc/build/`` and reused while neither
    the source nor a header is newer), then one link."""
    if not force and not _stale():
        return LIB_PATH
    hipcc = _hipcc()
    if hipcc is None:
        raise RfmError("hipcc not found: cannot build librfm_hip.so (set HIPCC or install ROCm)")
    from concurrent.futures import ThreadPoolExecutor

    obj_dir = os.path.join(CSRC, "build")
    os.makedirs(obj_dir, exist_ok=True)
    flags = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC"] + os.environ.get("RFM_HIPCC_FLAGS", "").split()
    stamp = os.path.join(obj_dir, "flags.txt")
    same_flags = os.path.exists(stamp) and open(stamp).read() == " ".join(flags)
    hdr_time = max(os.path.getmtime(h) for h in HEADERS if os.path.exists(h))

    def compile_one(src: str):
        obj = os.path.join(obj_dir, os.path.splitext(src)[0] + ".o")
        path = os.path.join(CSRC, src)
        if (not force and same_flags and os.path.exists(obj)
                and os.path.getmtime(obj) > max(os.path.getmtime(path), hdr_time)):
            return obj, None
        cmd = [hipcc] + flags + ["-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        proc = subprocess.run(cmd, capture_output=True, text=True)
        return obj, (proc.stdout + proc.stderr) if proc.returncode != 0 else None

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as pool:
        results = list(pool.map(compile_one, SOURCES))
    errors = [msg for _, msg in results if msg]
    if errors:
        raise RfmError("building librfm_hip.so failed:\n" + "\n".join(errors))
    open(stamp, "w").write(" ".join(flags))
    tmp = LIB_PATH + ".tmp.%d" % os.getpid()
    cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", tmp] + [o for o, _ in results] + ["-lpthread", "-ldl"]
    if verbose:
        print(" ".join(cmd))
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise RfmError("linking librfm_hip.so failed:\n" + proc.stdout + proc.stderr)
    os.replace(tmp, LIB_PATH)
    return LIB_PATH


_i32, _i64, _f64, _vp = C.c_int32, C.c_int64, C.c_double, C.c_void_p

# name -> argtypes; every function returns int32.  Mirrors include/rfm_hip.h.
SIGNATURES = {
    "rfm_version": [],
    "rfm_last_error": [C.c_char_p, C.c_size_t],
    "rfm_create": [_i32, _vp, C.POINTER(_vp)],
    "rfm_destroy": [_vp],
    "rfm_sync": [_vp],
    "rfm_copy_to_host": [_vp, _vp, _vp, _i64],
    "rfm_copy_to_device": [_vp, _vp, _vp, _i64],
    "rfm_profile_begin": [_vp],
    "rfm_profile_end": [_vp, _vp, _vp],
    "rfm_sample_batches": [_i64, _i64, _i64, _i64, _vp, _i32],
    "rfm_hash_bytes": [_vp, _i64, _i32, _vp],
    "rfm_fm_forward": [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _i32, _vp],
    "rfm_ips_logloss": [_vp, _vp, _vp, _vp, _vp, _i64, _f64, _vp],
    "rfm_fm_forward_loss": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _i32,
                            _f64, _vp, _vp],
    "rfm_fm_plan_create": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i64, _i32, C.POINTER(_vp)],
    "rfm_fm_plan_create_device": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i64, _i32, C.POINTER(_vp)],
    "rfm_fm_plan_destroy": [_vp],
    "rfm_fm_plan_info": [_vp, _vp],
    "rfm_fm_plan_layout": [_vp, _vp],
    "rfm_fm_plan_sliced": [_vp, _vp],
    "rfm_fm_plan_register_log": [_vp, _vp, _i32, _vp, _vp, _vp, _i64],
    "rfm_fm_plan_forward": [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp],
    "rfm_fm_plan_hot_columns": [_vp, _vp, _i32],
    "rfm_fm_step": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _f64],
    "rfm_fm_grad": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp],
    "rfm_fm_apply": [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _f64],
    "rfm_fm_grad_rows": [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i32, _vp],
    "rfm_fm_apply_rows": [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i64, _i32, _f64],
    "rfm_fm_reduce_rows": [_vp, _vp, _vp, _i32, _i64, _vp, _vp, _i64, _i32, _f64, _vp],
    "rfm_fm_set_rows": [_vp, _vp, _i64, _vp, _i32, _i64, _vp, _vp, _vp, _i64, _i32, _f64],
    "rfm_fm_train": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _f64,
                     _vp, _vp, _vp, _vp, _vp, _i64, _f64, _vp, _vp],
    "rfm_fm_train_eval": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp, _vp, _vp, _f64,
                     _vp, _vp, _vp, _vp, _vp, _i64, _f64, _vp, _vp,
                          _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _i64, _vp, _i64, _i64, _vp],
    "rfm_fm_train_dp": [_vp, _vp, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _vp, _f64, _vp],
    "rfm_fm_fit_dp": [_vp, _vp, _vp, _i32, _vp, _i64, _i64, _vp, _vp, _vp, _f64, _vp, _vp, _vp, _vp, _vp,
                      _i64, _f64, _vp, _vp],
    "rfm_mf_predict": [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _f64, _i32, _vp],
    "rfm_mf_predict_loss": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _f64, _i32,
                            _f64, _vp, _vp],
    "rfm_mf_schedule": [_vp, _vp, _i64, _i32, _i32, _vp, _vp, C.POINTER(_i32)],
    "rfm_mf_sgd_levels": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp,
                          _f64, _i32, _f64, _f64],
    "rfm_comm_unique_id": [_vp],
    "rfm_comm_init": [_vp, _i32, _i32, _vp],
    "rfm_allreduce_sum": [_vp, _vp, _i64],
    "rfm_comm_destroy": [_vp],
    "rfm_mf_cache_capacity": [_i32, C.POINTER(_i32)],
    "rfm_mf_schedule_ex": [_vp, _vp, _vp, _vp, _i64, _i32, _i32, _i32, _vp, _vp, C.POINTER(_i32), _vp,
                           C.POINTER(_i32)],
    "rfm_mf_sgd_levels_ex": [_vp, _vp, _vp, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _f64, _i32, _f64,
                             _f64],
    "rfm_mf_delta": [_vp, _vp, _vp, _vp, _i64],
    "rfm_mf_merge": [_vp, _vp, _vp, _vp, _i64],
    "rfm_mf_sgd_hogwild": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _f64, _i32, _f64,
                           _f64],
    "rfm_val_dcg": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp],
    "rfm_csr_assemble_count": [_vp, _vp, _i32, _i64, _vp, _vp],
    "rfm_csr_assemble_fill": [_vp, _vp, _i32, _i64, _vp, _vp, _vp],
    "rfm_topk_users": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp],
}



# ``rfm_transport`` of include/rfm_hip.h: the collectives of rfm_fm_fit_dp as caller functions
TRANSPORT_ALL_GATHER = C.CFUNCTYPE(_i32, _vp, _vp, _vp, _i64)
TRANSPORT_ALL_REDUCE = C.CFUNCTYPE(_i32, _vp, _vp, _i64)
TRANSPORT_ALL_TO_ALL = C.CFUNCTYPE(_i32, _vp, _vp, C.POINTER(_i64), C.POINTER(_i64), _vp, C.POINTER(_i64),
                                   C.POINTER(_i64))


class Transport(C.Structure):
    _fields_ = [("user", _vp), ("n_ranks", _i32), ("rank", _i32), ("all_gather", TRANSPORT_ALL_GATHER),
                ("all_reduce_sum", TRANSPORT_ALL_REDUCE), ("all_to_all", TRANSPORT_ALL_TO_ALL)]


class CsrSegment(C.Structure):
    """``rfm_csr_segment`` of include/rfm_hip.h."""

    _fields_ = [("kind", C.c_int32), ("d_ids", C.c_void_p), ("d_indptr", C.c_void_p),
                ("d_indices", C.c_void_p), ("d_values", C.c_void_p), ("col_offset", C.c_int64),
                ("n_block_rows", C.c_int64)]


_lib = None


def exported_symbols() -> List[str]:
    return list(SIGNATURES)


def load():
    """Load (building first if the in-tree .so is missing or stale)."""
    global _lib
    if _lib is not None:
        return _lib
    path = LIB_PATH
    if _stale():
        if _hipcc() is not None:
            path = build()
        elif not os.path.exists(LIB_PATH):
            raise RfmError(
                f"{LIB_PATH} is missing and hipcc is not available to build it; "
                "the HIP extension is required (there is no CPU fallback)")
    # PyTorch supplies the device memory and streams this library works on, and its wheel
    # ships its own HIP runtime: that runtime has to be in the process BEFORE this library
    # is mapped, or the loader binds librfm_hip.so to a second, separate runtime (ROCm's
    # system copy) that knows nothing of torch's allocations -- and, on the GPU boxes, finds
    # no device.  A process without PyTorch gets the system runtime, as any HIP program.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    try:
        lib = C.CDLL(path)
    except OSError as exc:
        raise RfmError(f"cannot load {path}: {exc}") from exc
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError = header/library mismatch
        fn.argtypes = argtypes
        fn.restype = _i32
    _lib = lib
    return lib


def last_error() -> str:
    buf = C.create_string_buffer(1024)
    load().rfm_last_error(buf, len(buf))
    return buf.value.decode("utf-8", "replace")


def check(rc: int) -> None:
    """Map the ABI's error classes to Python exceptions (bad argument ->
    ValueError, as the reference raises from ``resample``)."""
    if rc == RFM_OK:
        return
    msg = last_error()
    if rc == RFM_ERR_BAD_ARG:
        raise ValueError(msg)
    raise RfmError(f"librfm_hip error {rc}: {msg}")
