"""Drop-in replacement of BART's SWIG module ``transit_module``.

Same eight names and call shapes as the module the reference worker imports
(reference code/BARTfunc.py:28-30; call sites :230-234, :351-363, :406), backed
by ``libbartrt.so`` (HIP kernels, ``include/bartrt.h``) through ctypes:

    import transit_module as trm
    trm.transit_init(3, ["transit", "-c", cfg])
    nwave  = trm.get_no_samples()
    specwn = trm.get_waveno_arr(nwave)
    spectrum = trm.run_transit(profiles.flatten(), nwave)
    trm.free_memory()

There is no CPU path: without the built library or without a GPU every compute
call raises.  Batched and device-resident variants live in ``bart_amd.engine``.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (BARTRT_LIBPATH: another build of the library, for same-box A/B runs of tools/ -- tools/ab_build.py)
_LIBPATH = os.environ.get("BARTRT_LIBPATH") or os.path.join(_HERE, "libbartrt.so")
_lib = None


class TransitError(RuntimeError):
    pass


def _preload_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own
    libamdhip64.so under torch/lib with the same SONAME as /opt/rocm's; whichever
    is mapped first serves both torch and libbartrt.so.  When torch is installed
    its copy is mapped here (without importing torch) so that a later
    ``import torch`` in the same process finds the runtime it was built for."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec and spec.submodule_search_locations:
        p = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(p):
            C.CDLL(p, mode=C.RTLD_GLOBAL)


def lib():
    """Loads libbartrt.so (never builds it: ``__graft_entry__.build()`` or
    ``python -m bart_amd.build`` does)."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIBPATH):
            raise TransitError(
                "libbartrt.so is missing: build it with `python -m bart_amd.build` "
                "(the engine has no CPU fallback)")
        _preload_hip_runtime()
        L = C.CDLL(_LIBPATH)
        d, i, p = C.c_double, C.c_int, C.c_void_p
        L.bartrt_last_error.restype = C.c_char_p
        L.bartrt_build_id.restype = C.c_char_p
        L.bartrt_kernel_choice.restype = C.c_char_p
        L.bartrt_kernel_choice.argtypes = [i, C.c_long]
        L.bartrt_init.argtypes = [i, C.POINTER(C.c_char_p)]
        L.bartrt_get_waveno_arr.argtypes = [p, i]
        L.bartrt_set_radius.argtypes = [d]
        L.bartrt_set_cloudtop.argtypes = [d]
        L.bartrt_set_scattering.argtypes = [i, d]
        L.bartrt_run_transit.argtypes = [p, i, p, i]
        L.bartrt_run_transit_batch.argtypes = [p, i, i, p, i, p]
        L.bartrt_run_transit_batch_dev.argtypes = [p, i, p, p, p]
        L.bartrt_step_setup.argtypes = [p, i, i, d, d, p, i, p, i, p, p, p, p, d, i]
        L.bartrt_step_set_ebalance.argtypes = [i, d, d]
        L.bartrt_step_set_extras.argtypes = [i, i, i]
        L.bartrt_step_batch.argtypes = [p, i, i, p, p]
        L.bartrt_mcmc_run.argtypes = [i, i, C.c_long, p, p, p, p, i, p, p, i, C.c_ulonglong, p, p, p, p]
        L.bartrt_step_batch_dev.argtypes = [p, i, i, p, p, p, p]
        L.bartrt_step_profiles_dev.argtypes = [p, i, i, p, p, p]
        L.bartrt_step_bandflux_dev.argtypes = [p, i, p, p, p]
        L.bartrt_get_local_range.argtypes = [C.POINTER(i), C.POINTER(i)]
        L.bartrt_get_species.argtypes = [C.c_char_p, i]
        L.bartrt_get_pressure.argtypes = [p, i]
        L.bartrt_get_tau.argtypes = [p, p, i, i]
        L.bartrt_get_tau_of.argtypes = [i, p, p, i, i]
        L.bartrt_get_intensity_of.argtypes = [i, p, i, i]
        L.bartrt_get_lbl_extinction.argtypes = [p, i, p, i, i]
        L.bartrt_voigt.argtypes = [p, p, p, C.c_long]
        L.bartrt_timing_end.argtypes = [C.POINTER(d), C.POINTER(i)]
        L.bartrt_timing_begin_sampled.argtypes = [i]
        L.bartrt_set_integ.argtypes = [i]
        L.bartrt_set_cut.argtypes = [i]
        L.bartrt_get_cut.argtypes = [C.POINTER(i)]
        L.bartrt_set_kernel_by.argtypes = [i]
        L.bartrt_get_kernel_by.argtypes = [C.POINTER(i)]
        L.bartrt_get_cia_interp.argtypes = [C.POINTER(i)]
        L.bartrt_get_share.argtypes = [C.POINTER(i), C.POINTER(i)]
        L.bartrt_get_service.argtypes = [C.POINTER(i)] * 4
        L.bartrt_get_rtc_stats.argtypes = [C.POINTER(i)] * 4 + [C.POINTER(d)]
        L.bartrt_rtc_compile.argtypes = [C.c_char_p, i, C.POINTER(C.c_long)]
        L.bartrt_get_service_stats.argtypes = [C.POINTER(C.c_ulonglong)] * 3
        L.bartrt_get_service_gathered.argtypes = [C.POINTER(C.c_ulonglong)]
        L.bartrt_prefetch_profiles_dev.argtypes = [C.c_void_p, i]
        L.bartrt_get_integ.argtypes = [C.POINTER(i)]
        L.bartrt_walked_end.argtypes = [p, i, C.POINTER(i), C.POINTER(i), C.POINTER(i), C.c_char_p, i]
        L.bartrt_algorithmic_bytes.argtypes = [i]
        L.bartrt_algorithmic_bytes.restype = d
        _lib = L
    return _lib


def check(rc: int) -> int:
    if rc < 0:
        raise TransitError(lib().bartrt_last_error().decode() or f"bartrt error {rc}")
    return rc


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


# ---- the eight reference entry points ----------------------------------
_whole_grid = False      # this process's engine writes every sample of a spectrum (no --shard)


def transit_init(argc, argv):
    global _whole_grid
    args = [str(a).encode() for a in argv][:argc]
    arr = (C.c_char_p * len(args))(*args)
    _whole_grid = False
    check(lib().bartrt_init(len(args), arr))
    lo, hi = C.c_int(0), C.c_int(0)
    check(lib().bartrt_get_local_range(C.byref(lo), C.byref(hi)))
    _whole_grid = lo.value == 0 and hi.value == lib().bartrt_get_no_samples()


def get_no_samples():
    return check(lib().bartrt_get_no_samples())


def get_waveno_arr(n):
    out = np.zeros(int(n), np.double)
    check(lib().bartrt_get_waveno_arr(_ptr(out), int(n)))
    return out


def set_radius(refradius):
    check(lib().bartrt_set_radius(float(refradius)))


def set_cloudtop(cloudtop):
    check(lib().bartrt_set_cloudtop(float(cloudtop)))


def set_scattering(flag, value):
    check(lib().bartrt_set_scattering(int(flag), float(value)))


def run_transit(profiles, nwave):
    """profiles: flat (nspecies+1)*nlayers doubles -> new ndarray[nwave]."""
    # (the per-call cost of this wrapper is part of every MCMC step: ~3 us as written -- plain integer addresses
    # instead of ctypes pointer objects, no clearing of an array the engine fills completely)
    prof = np.ascontiguousarray(profiles, np.double).ravel()
    nwave = int(nwave)
    spec = np.empty(nwave, np.double) if _whole_grid else np.zeros(nwave, np.double)   # (a shard writes its block only)
    rc = (_lib or lib()).bartrt_run_transit(prof.ctypes.data, prof.size, spec.ctypes.data, nwave)
    if rc < 0:
        check(rc)
    return spec


def free_memory():
    check(lib().bartrt_free_memory())


# ---- beyond the reference module ----------------------------------------
INTEG_RULES = ("transmittance", "simpson", "trapz_tau")


def set_integ(rule):
    """Integration rule of the eclipse geometry (include/bartrt.h, bartrt_set_integ):
    0 / 'transmittance', 1 / 'simpson' (the default), 2 / 'trapz_tau'."""
    if isinstance(rule, str):
        rule = INTEG_RULES.index(rule)
    check(lib().bartrt_set_integ(int(rule)))


def set_cut(cut):
    """'slant' (default) or 'vertical': which optical depth `toomuch` is compared with
    (include/bartrt.h, bartrt_set_cut)."""
    check(lib().bartrt_set_cut({"vertical": 0, "slant": 1, 0: 0, 1: 1}[cut]))


def set_kernel_by(which):
    """'local' (the default since round 5) or 'whole': the column count that picks a sharded engine's kernel
    variant -- its own block's, or the whole grid's.  Under 'local' the concatenated blocks of a sharded run agree with
    the unsharded spectrum to rounding (4e-16 measured), NOT bit for bit; 'whole' makes every block the unsharded run's
    bits at the price of the single-wave kernel's latency floor on small blocks (include/bartrt.h,
    bartrt_set_kernel_by)."""
    check(lib().bartrt_set_kernel_by({"whole": 0, "local": 1, 0: 0, 1: 1}[which]))


def get_kernel_by() -> str:
    v = C.c_int(-1)
    check(lib().bartrt_get_kernel_by(C.byref(v)))
    return "local" if v.value else "whole"


def get_cut() -> str:
    v = C.c_int(-1)
    check(lib().bartrt_get_cut(C.byref(v)))
    return "slant" if v.value else "vertical"


def get_cia_interp() -> str:
    """'spline' (default) or 'linear': how the engine resampled the cross-section files at init
    (cfg key `cia_interp`)."""
    v = C.c_int(-1)
    check(lib().bartrt_get_cia_interp(C.byref(v)))
    return "spline" if v.value else "linear"


def get_share():
    """-> (shared, owner): the opacity grid is one allocation shared between processes (cfg `shareOpacity`,
    code/makecfg.py:106-107) / this process made it."""
    a, b = C.c_int(0), C.c_int(0)
    check(lib().bartrt_get_share(C.byref(a), C.byref(b)))
    return bool(a.value), bool(b.value)


def get_rtc_stats():
    """-> dict(available, compiled, from_disk, failed, compile_seconds): kernels instantiated at run time for shapes
    outside the ahead-of-time set (include/bartrt.h, bartrt_get_rtc_stats)."""
    v = [C.c_int(0) for _ in range(4)]
    s = C.c_double(0.0)
    check(lib().bartrt_get_rtc_stats(*[C.byref(x) for x in v], C.byref(s)))
    return {"available": bool(v[0].value), "compiled": v[1].value, "from_disk": v[2].value, "failed": v[3].value,
            "compile_seconds": s.value}


def get_service():
    """-> dict(mode, owner_pid, slot, nclients): mode 'engine' (this process runs its own), 'client' or 'owner'
    of the shareOpacity chain service (include/bartrt.h, bartrt_get_share)."""
    v = [C.c_int(0) for _ in range(4)]
    check(lib().bartrt_get_service(*[C.byref(x) for x in v]))
    return {"mode": ("engine", "client", "owner")[v[0].value], "owner_pid": v[1].value, "slot": v[2].value,
            "nclients": v[3].value}


def get_service_stats():
    """-> dict(launches, profiles, full): the dispatcher's rounds, the profiles they served, the rounds that held
    every registered client."""
    v = [C.c_ulonglong(0) for _ in range(3)]
    check(lib().bartrt_get_service_stats(*[C.byref(x) for x in v]))
    g = C.c_ulonglong(0)
    check(lib().bartrt_get_service_gathered(C.byref(g)))
    return {"launches": v[0].value, "profiles": v[1].value, "full": v[2].value, "gathered": g.value}


def get_integ() -> int:
    rule = C.c_int(-1)
    check(lib().bartrt_get_integ(C.byref(rule)))
    return int(rule.value)
