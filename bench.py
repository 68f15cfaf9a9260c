#!/usr/bin/env python3
"""Forward spectra/sec of the MI355X radiative-transfer engine.

    python bench.py --gpus N --steps K --warmup W [--walkers B]

A "step" is one pass of the hot path (profile prep + RT kernel; for N > 1 also
the RCCL all-gather that reassembles each spectrum) over one batch of B*N
synthetic walkers on the headline grid: 100 layers x 1e4 wavenumbers, 4 opacity
molecules (H2O, CO, CO2, CH4), 27 table temperatures, H2-H2 CIA, ray angles
0/20/40/60/80 deg, toomuch = 10, under the engine's default
integration rule (integ 1; --integ overrides).  Profiles and
spectra stay resident in HBM inside the timed region.  N > 1 shards the wavenumber
axis by block across the ranks, one process per GPU: under torch.distributed.run
(RANK / WORLD_SIZE in the environment) this process is one rank; run plainly with
--gpus N > 1 it starts `python -m torch.distributed.run --nproc-per-node N` on
itself as a CHILD process (before torch or HIP are touched here), relays the
child's output and exits with its code.

The contract's figure (`value`, `ms_per_step`) is the FIRST timed window: exactly K
steps between barriers, on SURVEY 8d's opacities to the letter (exp(N(-25,3)) cm2/g: every
layer of every column is walked, no credit from the `toomuch` cut), every step launching its
own prep_profiles kernel -- the form an MCMC step takes.  The line
certifies itself: `parity` = the spectra of the LAST step of that window against the CPU
oracle on the same walkers (whole spectra; the run fails above 1e-9) and, bit for bit,
against a fresh plain launch.  Around the window the default run adds, outside it:
  * `windows`: --repeats further windows of K steps (median / min / max);
  * `roofline.cold`: launches after a 1 GiB scratch sweep (nothing in L2 / Infinity Cache);
  * `integ_sweep` / `cut_sweep`: the integration rules and the two `toomuch` cuts at 10 and 256 walkers;
  * `batch_sweep`; `forest_workload` (a line-forest opacity model whose photosphere lies inside
    the column, plain and with the next batch's preparation prefetched);
  * `configs`: BASELINE.json configs 2, 4 (per-GPU work), transit geometry, 5, MC3-style worker processes;
  * `cpu_baseline` (the oracle on the timed run's own walkers); for N > 1 `scaling_diag` and a
    `replicas` comparison.
--no-extras keeps only the contract's window (what the profiling scripts run).

--dry-gloo replaces the GPU engine by a stub (spectra = a known function of the
profile and the sample index) on CPU tensors with the gloo backend: it exercises
the launch path, the sharded step loop, the bucketed all-gather and the
one-JSON-line contract where there is no GPU (tests/test_bench_launch.py).  Its
line carries "dry": true and measures nothing.

Prints ONE JSON line on rank 0 (contract: see the task statement; `roofline`
and `cpu_baseline` objects included).
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# kernel arguments in device memory (the ROCm default on this image; measured here: 78 against
# 83 us per ten-walker step with it switched off) -- kept on if the environment says nothing;
# the value in force is recorded in the line's `config`
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

PEAK_HBM_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
PEAK_FP64_TFLOPS = 78.6  # MI355X vector fp64 peak (MI355X_MICROARCH.md): 256 CUs x 64 lanes x 2 x 2.4 GHz
INTEG_NAMES = ("transmittance trapezoid", "Simpson hybrid of SURVEY App. A-4", "trapezoid in tau")


# HD 209458b system values of the demo TEP file (examples/demo/HD209458b.tep) and the
# T(p) parameter box of the demo retrieval (examples/demo/BART_eclipse.cfg:78-83)
_RSUN, _AU = 6.95508e10, 1.4959787066e13
_PT_SYS = dict(r_star=1.145 * _RSUN, t_star=6075.0, t_int=100.0, sma=0.047 * _AU, grav=897.70)
_PT_MIN = np.array([-5.0, -2.0, -2.0, 0.0, 0.55])
_PT_MAX = np.array([-1.0, 1.0, 1.0, 1.0, 1.2])


def _pt_line(p_bar, x):
    """Line et al. (2013) T(p) as the per-step callable evaluates it
    (code/PT.py:589-701): x = log10 kappa, log10 gamma1, log10 gamma2, alpha, beta."""
    from scipy.special import expn
    kappa, g1, g2 = 10.0 ** x[0], 10.0 ** x[1], 10.0 ** x[2]
    alpha, beta = x[3], x[4]
    t_irr = beta * np.sqrt(_PT_SYS["r_star"] / (2.0 * _PT_SYS["sma"])) * _PT_SYS["t_star"]
    tau = kappa * (p_bar * 1e6) / _PT_SYS["grav"]

    def xi(g):
        return 2.0 / 3.0 * (1.0 + (1.0 / g) * (1.0 + (g * tau / 2.0 - 1.0) * np.exp(-g * tau))
                            + g * (1.0 - tau ** 2 / 2.0) * expn(2, g * tau))

    return (0.75 * (_PT_SYS["t_int"] ** 4 * (2.0 / 3.0 + tau) + t_irr ** 4 * (1.0 - alpha) * xi(g1)
                    + t_irr ** 4 * alpha * xi(g2))) ** 0.25


def make_profiles(case, n, seed):
    """Walker profiles as the per-step callable builds them: T(p)
    from PT_line with parameters uniform in the demo's pmin/pmax, draws whose
    temperatures leave [Tmin, Tmax] = [400, 3000] K rejected as the worker rejects
    them (code/BARTfunc.py:327-329); abundance log-factors U(-2, 1), H2/He
    renormalised at their ratio."""
    rng = np.random.default_rng(seed)
    L = len(case.press_bar)
    out = np.empty((n, (len(case.species) + 1) * L))
    w = 0
    while w < n:
        t = _pt_line(case.press_bar, rng.uniform(_PT_MIN, _PT_MAX))
        if not (t.min() > 400.0 and t.max() < 3000.0):
            continue
        ab = case.abund0.copy()
        for s in range(2, ab.shape[1]):
            ab[:, s] *= 10 ** rng.uniform(-2, 1)
        q = 1 - ab[:, 2:].sum(1)
        r = ab[:, 1] / ab[:, 0]
        ab[:, 1] = r * q / (1 + r)
        ab[:, 0] = q / (1 + r)
        out[w] = case.profiles(t, ab).ravel()
        w += 1
    return out


def oracle_engine(case, conv):
    """The CPU oracle under the conventions the GPU engine runs with (conv: integ, cut, cia_interp)."""
    from oracle import rt_oracle as orc
    return orc.OracleEngine(case.tcfg, integ=conv["integ"], cut=conv["cut"], cia_interp=conv["cia_interp"])


def cpu_baseline(case, conv, walkers, seconds_target=8.0):
    """The CPU oracle (a restatement, NOT reference transit: its source is an empty
    submodule) timed on this host's cores, one walker per thread, same conventions as
    the GPU line, on the timed run's OWN walkers (`walkers`: every profile of the cycled
    batches, repeated in order to fill the sample): a warm-up pass (thread pool, page
    faults), then passes of growing size until one runs for at least 5 s."""
    cores = os.cpu_count() or 1
    eng = oracle_engine(case, conv)
    take = lambda n: walkers[np.arange(n) % len(walkers)]
    t0 = time.perf_counter()
    eng.run_batch(take(cores), threads=cores)
    rate = cores / max(time.perf_counter() - t0, 1e-3)          # includes the pool's start: an underestimate
    n = dt = 0
    for _ in range(4):
        n = int(min(16384, max(cores, rate * seconds_target)))
        n = max(cores, (n // cores) * cores)
        profs = take(n)
        t0 = time.perf_counter()
        eng.run_batch(profs, threads=cores)
        dt = time.perf_counter() - t0
        rate = n / dt
        if dt >= 5.0 or n >= 16384:
            break
    return {"value": n / dt, "unit": "spectra/s", "cores": cores, "kind": "port",
            "sample_short": "%d spectra of the timed run's own walkers, one per thread on %d threads, %.1f s wall; the "
                            "oracle, NOT reference transit" % (n, cores, dt),
            "sample": f"{n} spectra = the timed run's own {len(walkers)} walkers, repeated in order (100x1e4, 4 "
                      f"molecules, integ {conv['integ']}, cut {conv['cut']}, cia_interp {conv['cia_interp']}), one "
                      f"walker per thread on {cores} threads, {dt:.1f} s wall after a warm-up pass of {cores} spectra"}


def parity_record(case, conv, profs, spec_gpu, plain_equal, what, max_walkers=16, tol=1e-9):
    """The timed run's own output against the CPU oracle: whole spectra of up to max_walkers of the
    step's walkers (evenly spread over the batch), every sample.  Relative error per sample against
    the larger of |reference| and 1e-12 of the spectrum's largest sample (rule 1's panels change sign
    on coarse columns: samples that cancel to nothing are held to the spectrum's scale, as the parity
    tests hold them)."""
    nw = profs.shape[0]
    pick = np.unique(np.linspace(0, nw - 1, min(nw, max_walkers)).round().astype(int))
    ref = oracle_engine(case, conv).run_batch(profs[pick], threads=min(len(pick), os.cpu_count() or 1))
    got = spec_gpu[pick]
    scale = np.maximum(np.abs(ref), 1e-12 * np.abs(ref).max(axis=1, keepdims=True))
    err = np.abs(got - ref) / np.maximum(scale, 1e-300)
    rec = {"max_rel_err": float(err.max()), "tolerance": tol, "n_samples": int(ref.size), "walkers": int(len(pick)),
           "walkers_in_step": int(nw), "against": "oracle/rt_oracle.c (CPU restatement; RT parity unpinned: the "
                                                  "reference's engine is an empty submodule)",
           "what": what, "bit_equal_to_plain_launch": bool(plain_equal),
           "spectrum_max": float(np.abs(ref).max()), "all_finite": bool(np.isfinite(got).all())}
    rec["ok"] = bool(rec["max_rel_err"] <= tol and plain_equal and rec["all_finite"])
    return rec


def source_id():
    """The loaded library's own id (bartrt_build_id: a hash of the code objects and host text of the RT translation
    units, fixed at build time -- bart_amd/build.py code_id).  Ties committed profiler figures (PMC traffic, SQ pass,
    instruction mix) to the build they were taken on: bench.py reports them only under the same id.  A comment edit
    keeps the id (round 5 lost its `traffic` to one: the id then was a hash of source text)."""
    from bart_amd import transit_module as trm
    return trm.lib().bartrt_build_id().decode()


def _cia_temps(path):
    lines = open(path).read().split("\n")
    return np.array([float(x) for x in lines[lines.index("@TEMPERATURES") + 1].split()])


def launch_byte_model(case, profs, walked, wn_per_col, nwave, ncia=1, spline=False):
    """HBM bytes one RT launch has to move, from the launch's own walkers:

    * effective: the SURVEY 8d figure (two T planes x M molecules + two CIA planes per
      layer and wavenumber, per walker, no credit for reuse) restricted to the layers
      each wave actually walked before all its lanes passed `toomuch`;
    * unique: the same with every (layer, T plane, molecule, wavenumber column) row
      counted ONCE per launch however many of the launch's walkers read it -- the
      walkers of a launch share table planes, and whichever implementation is used
      has to fetch a row from HBM only once per launch: this is the compulsory
      traffic the HBM roofline fraction is quoted on (+ records in, spectra out).
    walked[nwalkers][ncols]: layers walked per column of wn_per_col wavenumbers.
    spline: `cia_interp spline` -- every CIA file is two table slots (values and second derivatives in T)."""
    nw, ncols = walked.shape
    L = len(case.press_bar)
    M = len(case.opmol)
    S = len(case.species)
    T = profs.reshape(nw, S + 1, L)[:, 0, ::-1]            # [walker][k], k = 0 top
    tg = np.asarray(case.tgrid)
    j = np.clip(np.searchsorted(tg, T, side="right") - 1, 0, len(tg) - 2)
    cia_t = [_cia_temps(f) for f in case.cia[:ncia]]
    jc = [np.clip(np.searchsorted(t, np.clip(T, t[0], t[-1]), side="right") - 1, 0, max(len(t) - 2, 0))
          for t in cia_t]
    width = np.minimum(wn_per_col, nwave - wn_per_col * np.arange(ncols)).clip(min=0)
    k = np.arange(L)
    nsl = 2 if spline else 1
    eff = uniq = 0.0
    walked_lw = 0.0
    for c in range(ncols):
        if width[c] <= 0:
            continue
        act = walked[:, c][:, None] > k[None, :]            # [walker][layer]
        walked_lw += act.sum() * width[c]
        eff += act.sum() * width[c] * (2 * M + 2 * nsl * len(cia_t)) * 8.0
        used = np.zeros((L, len(tg) + 1), bool)
        for w in range(nw):
            used[k[act[w]], j[w][act[w]]] = True
            used[k[act[w]], j[w][act[w]] + 1] = True
        uniq += used.sum() * M * width[c] * 8.0
        for t, jj in zip(cia_t, jc):
            usedc = np.zeros((L, len(t) + 1), bool)
            for w in range(nw):
                usedc[k[act[w]], jj[w][act[w]]] = True
                usedc[k[act[w]], jj[w][act[w]] + 1] = True
            # (the same CIA plane serves every layer that brackets it: count planes, not (layer, plane))
            uniq += nsl * usedc.any(axis=0).sum() * width[c] * 8.0
    NC, NI = 4 + 2 * M + 2 * nsl * len(cia_t), 1 + nsl * len(cia_t)      # words of a layer record (csrc/kernels.hpp)
    fixed = nw * L * (NC + NI) * 8.0 + nw * nwave * 8.0 + nwave * 8.0   # records, spectra out, wavenumbers
    return {"effective_bytes": eff + fixed, "unique_bytes": uniq + fixed,
            "layers_walked_frac": walked_lw / (nw * L * float(nwave)),
            "layer_wavenumbers_walked": walked_lw}


LINE_LIMIT = 4096        # bytes: the driver keeps 8 kB of stdout; the r05 line was 20 kB and did not parse
DETAIL_FILE = os.path.join(ROOT, "bench_detail.json")


def _short(x, digits=6):
    """Floats to `digits` significant digits (lists and dicts walked): the line is a record, not an archive."""
    if isinstance(x, float):
        return float("%.*g" % (digits, x)) if np.isfinite(x) else None
    if isinstance(x, (np.floating, np.integer)):
        return _short(x.item(), digits)
    if isinstance(x, dict):
        return {k: _short(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_short(v, digits) for v in x]
    return x


def _pick(d, keys):
    return {k: d.get(k) for k in keys} if isinstance(d, dict) else None


def contract_line(res, detail_name="bench_detail.json"):
    """The ONE line of the contract from the full result `res`: the contract's scalar keys, a one-sentence
    `config`, `roofline`, `cpu_baseline`, `parity`, for N > 1 a short `scaling_diag`, and a pointer to the detail
    file that holds everything else (windows, sweeps, side legs, notes).  At most LINE_LIMIT bytes whatever the
    run produced (tests/test_bench_launch.py feeds it a full-size result, N = 8 included)."""
    cfg = res.get("config") or {}
    roof = res.get("roofline")
    line = {k: res.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                    "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    if res.get("dry"):
        line["dry"] = True
    wl = str(cfg.get("workload_short") or cfg.get("workload") or "")
    line["config"] = {"workload": wl if len(wl) <= 300 else wl[:297] + "...",
                      **{k: cfg.get(k) for k in ("walkers_per_step", "nlayers", "nwave", "integ", "cut", "cia_interp",
                                                 "kappa_model", "parallelism") if k in cfg}}
    if roof:
        r = _pick(roof, ("bound", "achieved", "peak", "unit", "frac"))
        fr = roof.get("fractions") or {}
        r["frac_cold"] = fr.get("frac_cold")
        r["frac_survey8d_letter"] = fr.get("frac_survey8d_letter")
        r.update(_pick(roof, ("traffic", "traffic_source", "kernel", "avg_launch_ms", "unique_bytes_per_launch",
                              "survey8d_algorithmic_bytes_per_launch", "launches")))
        bm = roof.get("bound_measured")
        if isinstance(bm, dict):       # the SQ pass of this build, one word per batch size
            r["bound_measured"] = {k: v.get("bound") for k, v in bm.items() if isinstance(v, dict)}
            r["bound_measured"]["source"] = bm.get("source")
        else:
            r["bound_measured"] = None
        f64 = roof.get("fp64")
        r["fp64_frac"] = f64.get("frac") if isinstance(f64, dict) else None
        line["roofline"] = r
    else:
        line["roofline"] = None
    cb = res.get("cpu_baseline")
    if isinstance(cb, dict) and "value" in cb:
        line["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind"))
        line["cpu_baseline"]["sample"] = str(cb.get("sample_short") or cb.get("sample") or "")[:160]
    else:
        line["cpu_baseline"] = cb if cb is None else {"error": str(cb.get("error"))[:120]}
    par = res.get("parity")
    if isinstance(par, dict):
        line["parity"] = _pick(par, ("max_rel_err", "tolerance", "n_samples", "walkers", "bit_equal_to_plain_launch", "ok"))
        line["parity"]["against"] = "oracle/rt_oracle.c (CPU restatement; RT parity unpinned)"
    win = res.get("windows")
    if isinstance(win, dict):
        line["windows_ms_per_step"] = win.get("ms_per_step")
    d = res.get("scaling_diag")
    if isinstance(d, dict):
        sd = _pick(d, ("mode", "rank_skew_ms", "steps_per_bucket", "allgather_send_bytes_per_rank_per_bucket",
                       "allgather_recv_bytes_per_rank_per_bucket", "ms_per_step_minus_rt_kernel"))
        for k in ("per_rank_window_s", "per_rank_rt_kernel_ms", "per_rank_final_drain_ms"):
            sd[k] = _short(d.get(k), 4)
        line["scaling_diag"] = {k: v for k, v in sd.items() if v is not None}
    rep = res.get("replicas")
    if isinstance(rep, dict):
        line["replicas"] = _pick(rep, ("value", "unit", "ms_per_step", "walkers_per_rank"))
    line["source_id"] = res.get("source_id")
    line["detail"] = detail_name
    line = _short(line)
    text = json.dumps(line, separators=(",", ":"))
    if len(text) > LINE_LIMIT:             # last resort (a launcher with hundreds of ranks): drop the per-rank lists
        for k in ("per_rank_window_s", "per_rank_rt_kernel_ms", "per_rank_final_drain_ms"):
            line.get("scaling_diag", {}).pop(k, None)
        text = json.dumps(line, separators=(",", ":"))
    assert len(text) <= LINE_LIMIT, len(text)
    return text


def emit(res, detail_path=DETAIL_FILE):
    """Everything to the detail file (and to stderr), the contract's line -- alone -- to stdout."""
    name = None
    try:
        with open(detail_path, "w") as f:
            json.dump(res, f, indent=1)
            f.write("\n")
        name = os.path.relpath(detail_path, ROOT)
    except OSError as e:       # a read-only checkout must not cost the line
        print("bench.py: detail file not written: %r" % (e,), file=sys.stderr)
    print("bench.py detail: " + json.dumps(res), file=sys.stderr, flush=True)
    print(contract_line(res, name), flush=True)


def self_launch(ngpus, argv, port=0):
    """`python bench.py --gpus N` (N > 1) without a launcher: run the ranks as a
    child `python -m torch.distributed.run` of this script, relay its stdout
    (rank 0's one JSON line) and stderr, return its exit code."""
    import socket
    import subprocess
    if not port:
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(ngpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.run(cmd, env=env).returncode


class StubEngine:
    """--dry-gloo stand-in for bart_amd.engine (no GPU, no tables): the spectrum of
    a profile is a known function of the profile and the absolute sample index,
    so the reassembled [nwalkers, W] result can be checked exactly."""

    def __init__(self, nwave):
        self.W, self.lo, self.hi = nwave, 0, nwave

    def init(self, tcfg, shard=None, device=None):
        self.lo, self.hi = 0, self.W
        if shard is not None:
            r, n = shard
            self.lo, self.hi = self.W * r // n, self.W * (r + 1) // n   # Engine::setup's split
            if self.hi <= self.lo:
                raise RuntimeError("--shard leaves this rank without wavenumber samples")

    def local_range(self):
        return self.lo, self.hi

    @staticmethod
    def expected(prof, lo, hi):
        import torch
        i = torch.arange(lo, hi, dtype=torch.float64)
        return prof.sum(1, keepdim=True) * 1e-3 + i[None, :]

    def run_batch_dev(self, d_prof, d_spec, next_prof=None):
        d_spec.copy_(self.expected(d_prof, self.lo, self.hi))
        return d_spec

    def timing_begin(self, stride=1):
        pass

    def timing_end(self):
        return 0.0, 0

    def algorithmic_bytes(self, nwalkers):
        return 0.0

    def free(self):
        pass

    @staticmethod
    def GatherPipeline(*args, **kw):
        from bart_amd import engine
        return engine.GatherPipeline(*args, **kw)


def _stats(ms):
    ms = sorted(ms)
    return {"n": len(ms), "median": float(np.median(ms)), "min": ms[0], "max": ms[-1]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--spinup-ms", type=float, default=150.0,
                    help="untimed set-up: run the hot path this long before the W warm-up steps, so that the timed "
                         "window starts at the device's working clocks (0: off)")
    ap.add_argument("--walkers", type=int, default=10,
                    help="walkers per GPU per step (BASELINE.json config 3: 10)")
    ap.add_argument("--nwave", type=int, default=10000)
    ap.add_argument("--nlayers", type=int, default=100)
    ap.add_argument("--nsets", type=int, default=16,
                    help="distinct walker batches cycled through the steps")
    ap.add_argument("--integ", type=int, default=None, choices=[0, 1, 2],
                    help="integration rule of the headline run (default: the engine's, integ 1 = App. A-4)")
    ap.add_argument("--repeats", type=int, default=15,
                    help="further timed windows of --steps steps after the contract's (spread of the figure)")
    ap.add_argument("--sweep", default="64,256,1024",
                    help="extra batch sizes reported under batch_sweep (N=1 only; '' = none)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the contract's window and its roofline record: no repeats, cold pass, sweeps, "
                         "survey8d leg, configs, replicas comparison (what the profiling scripts run)")
    ap.add_argument("--event-stride", type=int, default=3,
                    help="HIP events bracket every n-th RT launch of the timed windows (a pair of event records "
                         "costs the stream about 5 us, 7 %% of a ten-walker step; 1 = every launch).  Coprime with "
                         "--nsets, so the sampled launches visit every cycled batch")
    ap.add_argument("--gather-steps", type=int, default=4,
                    help="N > 1: steps per all-gather bucket (fewer, larger collectives; the last bucket's "
                         "collective is the one nothing overlaps, so a bucket should stay a small part of the run)")
    ap.add_argument("--mode", default="shard", choices=["shard", "replicas"],
                    help="N > 1: shard = wavenumber blocks across the ranks + all-gather (north_star); replicas = "
                         "every rank holds the whole grid and runs its own walkers, no collective (SURVEY 8e's "
                         "baseline).  The default sharded run also times a replicas pass for comparison")
    ap.add_argument("--force-collective", action="store_true",
                    help="run the all-gather path even on one rank (smoke check of the N>1 code)")
    ap.add_argument("--force-replicas-leg", action="store_true",
                    help="run the N > 1 line's replicas comparison -- free the engine, initialise it unsharded, time, free, "
                         "initialise the shard again, all under the live process group -- whatever the rank count "
                         "(first-run-proofing of the 8-GPU record on one GPU)")
    ap.add_argument("--workdir", default=None)
    ap.add_argument("--config", default="table", choices=["table", "lbl"],
                    help="table: the headline opacity-table workload (BASELINE config 3, the contract's line); "
                         "lbl: BASELINE config 5, on-the-fly Voigt line-by-line (tools/lbl_bench.py's line)")
    ap.add_argument("--wnosamp", type=int, default=1, help="--config lbl: oversampling of the line sums")
    ap.add_argument("--kappa", default="survey8d", choices=["forest", "survey8d"],
                    help="opacity model of the headline run.  survey8d (default): SURVEY 8d's literal exp(N(-25,3)) "
                         "cm2/g -- a transparent column, every layer walked; forest: a log-normal line forest of median "
                         "~1 cm2/g whose photosphere lies inside the column (the default run reports it as the extra "
                         "`forest_workload`)")
    ap.add_argument("--prefetch", action="store_true",
                    help="name the next batch to the engine with every call (bartrt_prefetch_profiles_dev): the RT "
                         "launch prepares the next batch's layer records in extra workgroups.  For a grid or population "
                         "of independent models; NOT the form of an MCMC step (its proposal depends on the previous "
                         "spectra), so not the default")
    ap.add_argument("--no-prefetch", action="store_true", help="(the default since round 4; accepted for old scripts)")
    ap.add_argument("--cut", default=None, choices=["vertical", "slant"],
                    help="which depth `toomuch` cuts (default: the engine's; DESIGN.md C19)")
    ap.add_argument("--same-walkers", action="store_true",
                    help="diagnostic: every walker of a batch carries the batch's first profile (all table "
                         "planes shared: what the launch costs without its own HBM traffic); not a benchmark")
    ap.add_argument("--dry-gloo", action="store_true",
                    help="no GPU: stub engine, CPU tensors, gloo backend (launch-path check, not a measurement)")
    ap.add_argument("--detail", default=DETAIL_FILE,
                    help="where the full record goes (windows, sweeps, side legs, notes); stdout carries the contract's "
                         "line only, at most %d bytes" % LINE_LIMIT)
    ap.add_argument("--master-port", type=int, default=0,
                    help="rendezvous port of the self-launched N > 1 run (0: pick a free one)")
    a = ap.parse_args()
    if a.dry_gloo and a.detail == DETAIL_FILE:
        # a dry run measures nothing: its record must not replace a measured one next to bench.py
        a.detail = os.path.join(tempfile.gettempdir(), "bench_detail_dry_%d.json" % os.getpid())

    if a.config == "lbl":
        # a second artefact, not the contract's line: config 5 on one GPU
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import lbl_bench
        lbl_bench.run(["--wnosamp", str(a.wnosamp)])
        return

    if a.gpus > 1 and "RANK" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher.  Nothing here has
        # imported torch or touched HIP yet, and the ranks are CHILD processes (a
        # process that has initialised the GPU must never exec or re-launch itself).
        sys.exit(self_launch(a.gpus, sys.argv[1:], a.master_port))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    a.gpus = world

    import torch
    import torch.distributed as dist
    dry = a.dry_gloo
    if dry:
        dev = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            sys.exit("bench.py needs a GPU: the engine has no CPU path")
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)

    def sync():
        if not dry:
            torch.cuda.synchronize()

    in_group = world > 1 or (a.force_collective and "RANK" in os.environ)
    if in_group:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL (and gloo) print a banner through C stdio on stdout when the first
        # communicator comes up; the contract is ONE JSON line there.  File
        # descriptor 1 points at stderr until the communicator exists and the C
        # buffers are flushed.
        import ctypes
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            if dry:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=dev)
            dist.barrier()
            sync()
            ctypes.CDLL(None).fflush(None)
        finally:
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    from bart_amd import synth
    if dry:
        engine, trm = StubEngine(a.nwave), None
    else:
        from bart_amd import engine, transit_module as trm

    # ---- synthetic inputs (rank 0 of the node writes, everyone reads)
    wd = a.workdir or os.path.join(tempfile.gettempdir(),
                                   "bartrt_bench_%s" % os.environ.get("MASTER_PORT", "single"))
    if a.kappa != "forest":
        wd += "_" + a.kappa
    case = synth.make_case(wd, nlayers=a.nlayers, nwave=a.nwave, kappa_model=a.kappa,
                           write=(local_rank == 0 and not dry), reuse=True)
    if in_group:
        dist.barrier()
    sharded = world > 1 and a.mode == "shard"
    use_gather = sharded or (in_group and a.force_collective and a.mode == "shard")
    engine.init(case.tcfg, shard=(rank, world) if sharded else None, device=local_rank)
    def apply_conventions():
        if a.integ is not None:
            trm.set_integ(a.integ)
        if a.cut is not None:
            trm.set_cut(a.cut)
    if trm is not None:
        apply_conventions()
    integ = trm.get_integ() if trm is not None else -1
    # the conventions the engine runs under: the oracle of the parity record and of cpu_baseline follows them
    conv = {"integ": integ, "cut": trm.get_cut(), "cia_interp": trm.get_cia_interp()} if trm is not None else None
    lo, hi = engine.local_range()

    def timed(nwalk, steps, warmup, record, repeats=0, gather=None, prefetch=None, before=None):
        """The contract's window -- `steps` passes of the hot path over batches of nwalk
        walkers between barriers -- and `repeats` more windows of the same length.  The
        batches cycle through a.nsets distinct seeded sets so that consecutive steps do
        not re-read exactly the same table planes.  Returns a dict: dt (first window, max
        over ranks), windows_ms (per-step time of every window), kern_ms / nlaunch (RT
        kernel time of the sampled launches of all windows), ok, profs, diag (N > 1)."""
        gather = use_gather if gather is None else gather
        prefetch = a.prefetch if prefetch is None else prefetch
        l0, h0 = engine.local_range()
        nsets = max(1, min(a.nsets, 4096 // max(nwalk, 1) or 1))
        profs_h = make_profiles(case, nwalk * nsets, seed=20260103 + (0 if gather or world == 1 else rank))
        profs_h = profs_h.reshape(nsets, nwalk, -1)
        if a.same_walkers:
            profs_h[:] = profs_h[:, :1]
        d_prof = torch.from_numpy(profs_h).to(dev)
        # N > 1: the steps' local blocks go into bucket slots; one all-gather per
        # bucket runs on RCCL's stream while the next bucket's kernels run on the
        # compute stream (engine.GatherPipeline)
        d_local = [torch.empty((nwalk, h0 - l0), dtype=torch.float64, device=dev) for _ in range(2)]
        pipe = engine.GatherPipeline(nwalk, h0 - l0, a.nwave, a.gather_steps, dev) if gather else None
        wfull = a.nwave if gather else h0 - l0

        # the batches are resident and independent: each call names the next one, whose layer
        # records the current RT launch prepares on the side (bartrt_prefetch_profiles_dev)
        nxt = (lambda i: None) if prefetch is False else (lambda i: d_prof[(i + 1) % nsets])

        def step(i):
            if not gather:
                engine.run_batch_dev(d_prof[i % nsets], d_local[i & 1], next_prof=nxt(i))
                return d_local[i & 1]
            engine.run_batch_dev(d_prof[i % nsets], pipe.slot(i), next_prof=nxt(i))
            done = pipe.submit(i)            # reassembled spectra of an earlier bucket, or None
            return done[-1] if done is not None else None

        def drain(last):
            outs = pipe.drain(last)
            return [outs[-1][-1]] if outs else []

        def window(nsteps, first_step=0):
            """-> (seconds between the barriers on this rank, seconds of the final drain, last output)"""
            out_ = None
            # (the interpreter's cyclic collector stays out of the window, as in timeit: with torch imported a full
            # collection is a 30-40 ms pause on the host -- measured in tools/bench_configs.py full_step_10, round 5 --
            # and which window it lands in depends on the allocations of everything that ran before.  The collection
            # itself is made before the spin-up below, not here: 40 ms of idle device in front of a window cost it its clocks)
            gc.disable()
            try:
                sync()
                if in_group:
                    dist.barrier()
                sync()
                t0 = time.perf_counter()
                for i in range(nsteps):
                    o = step(i)
                    out_ = o if o is not None else out_
                t1 = time.perf_counter()
                if gather:   # (t1: the kernels are enqueued, the collectives of the filled buckets too)
                    out_ = (drain(nsteps - 1) or [out_])[-1]
                sync()
                t2 = time.perf_counter()
                if in_group:
                    dist.barrier()
                sync()
                return time.perf_counter() - t0, t2 - t1, out_
            finally:
                gc.enable()

        if before is not None:       # untimed set-up work on the same batches (the byte model's passes)
            before(profs_h, d_prof)
        gc.collect()
        if record:                   # (creates its event pool: host work, before the device is spun up)
            engine.timing_begin(a.event_stride)
            engine.timing_end()
        out = None
        # spin-up, part of the set-up: the hot path runs until the device has been busy for --spinup-ms.  The
        # set-up above leaves the GPU idle for seconds (host-side byte model, event pool) and its clocks low;
        # a warm-up of W = 20 steps is 1.5 ms, and the first window after it measured 81 us per step against
        # 71-73 in the fifteen windows that followed.  A retrieval runs for hours: steady state is the figure.
        if a.spinup_ms > 0 and (gather or not dry):   # (the dry run walks the collective form with 16 calls)
            t_end = time.perf_counter() + a.spinup_ms * 1e-3
            i = 0
            if gather:
                # collectives inside: every rank must make the same number of calls -- a fixed count (about the
                # same time at the single-GPU step rate) instead of each rank's own clock
                nspin = 16 if dry else max(16, int(a.spinup_ms * 1e-3 / 75e-6 / max(1.0, a.walkers / 10.0)) // 16 * 16)
                for i in range(nspin):
                    out = step(i)
                out = (drain(nspin - 1) or [out])[-1]
            else:
                while time.perf_counter() < t_end:
                    for _ in range(16):
                        out = step(i)
                        i += 1
                    sync()
        for i in range(warmup):
            out = step(i)
        if gather and warmup:
            out = (drain(warmup - 1) or [out])[-1]
        if record:
            engine.timing_begin(a.event_stride)
        dt_local, drain_s, out = window(steps)
        # the contract's window has ended: its LAST step's spectra, set aside for the parity record
        last_set = (steps - 1) % nsets
        out_first = out.clone() if (out is not None and not dry) else out
        wins = [dt_local]
        for _ in range(repeats):
            wins.append(window(steps)[0])
        kern_ms, nlaunch = engine.timing_end() if record else (0.0, 0)
        dt = dt_local
        diag = None
        if in_group:
            t = torch.tensor([dt_local, drain_s, kern_ms / max(nlaunch, 1)] + wins, dtype=torch.float64, device=dev)
            allt = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(allt, t)
            allt = torch.stack(allt).cpu().numpy()              # [rank][...]
            dt = float(allt[:, 0].max())
            wins = list(allt[:, 3:].max(axis=0))
            wmax = max(a.nwave * (r + 1) // world - a.nwave * r // world for r in range(world))
            diag = {
                "mode": "shard" if gather else "replicas",
                "per_rank_window_s": [float(x) for x in allt[:, 0]],
                "rank_skew_ms": float((allt[:, 0].max() - allt[:, 0].min()) * 1e3),
                "per_rank_rt_kernel_ms": [float(x) for x in allt[:, 2]],
                "per_rank_final_drain_ms": [float(x * 1e3) for x in allt[:, 1]],
                "exposed_gather_note": "final_drain = host time from the last step's enqueue to the end of the last "
                                       "bucket's collective and reassembly (the part of the run no kernel overlaps)",
            }
            if gather:
                diag.update({
                    "steps_per_bucket": a.gather_steps,
                    "allgather_send_bytes_per_rank_per_bucket": a.gather_steps * nwalk * wmax * 8,
                    "allgather_recv_bytes_per_rank_per_bucket": world * a.gather_steps * nwalk * wmax * 8,
                    "ms_per_step_minus_rt_kernel": float(dt / steps * 1e3 - allt[:, 2].max()),
                })
        ok = out is not None and out.shape == (nwalk, wfull) and bool(torch.isfinite(out).all())
        if dry:   # the reassembled spectra of the last step, sample for sample
            ok = ok and bool(torch.equal(out, StubEngine.expected(d_prof[(steps - 1) % nsets], 0 if gather else l0,
                                                                  a.nwave if gather else h0)))
        return {"dt": dt, "windows_ms": [w / steps * 1e3 for w in wins], "kern_ms": kern_ms, "nlaunch": nlaunch,
                "ok": ok, "profs": profs_h, "d_prof": d_prof, "diag": diag, "last_out": out_first, "last_set": last_set,
                "gathered": bool(gather)}

    # weak scaling: per-GPU work is fixed.  Sharded: every rank evaluates all B*N walkers on its
    # wavenumber block; replicas: every rank evaluates its own B walkers on the whole grid.
    nwalk = a.walkers * world if (sharded or world == 1) else a.walkers
    nspectra_per_step = a.walkers * world
    extras = not a.no_extras and not dry
    models, kname_box = [], [""]

    def byte_model_passes(profs_h, d_prof):
        """What one launch has to move: untimed passes over the run's own batches with the kernels
        recording how deep each wave walked (bartrt_walked_*).  Run BEFORE the warm-up and the
        timed windows: set-up work, and the device is at its working clocks when the
        contract's window starts."""
        l0, h0 = engine.local_range()
        d_out = torch.empty((profs_h.shape[1], h0 - l0), dtype=torch.float64, device=dev)
        for sset in range(profs_h.shape[0]):
            engine.walked_begin()
            engine.run_batch_dev(d_prof[sset], d_out)
            torch.cuda.synchronize()
            walked, wpc, kname_box[0] = engine.walked_end()
            models.append(launch_byte_model(case, profs_h[sset], walked, wpc, h0 - l0, spline=conv["cia_interp"] == "spline"))

    main_run = timed(nwalk, a.steps, a.warmup, True, repeats=a.repeats if extras else 0,
                     before=byte_model_passes if (rank == 0 and not dry) else None)
    dt, kern_ms, nlaunch, ok, profs_all = (main_run[k] for k in ("dt", "kern_ms", "nlaunch", "ok", "profs"))
    profs0 = profs_all[0]

    # ---- the line certifies its own output: the LAST step of the contract's window against the CPU oracle on
    # the same walkers (whole spectra), and bit for bit against a fresh launch of that batch with nothing prefetched
    parity = None
    if rank == 0 and not dry:
        lset = main_run["last_set"]
        got = main_run["last_out"]                       # [nwalk][whole grid if gathered, else this rank's block]
        plain = torch.empty((nwalk, hi - lo), dtype=torch.float64, device=dev)
        engine.run_batch_dev(main_run["d_prof"][lset], plain)
        torch.cuda.synchronize()
        plain_equal = bool(torch.equal(got[:, lo:hi] if main_run["gathered"] else got, plain))
        parity = parity_record(
            case, conv, profs_all[lset], got.cpu().numpy(), plain_equal,
            "spectra of step %d (the last) of the contract's window = batch %d of the %d cycled"
            % (a.steps - 1, lset, profs_all.shape[0]))
        if not parity["ok"]:
            print(json.dumps({"error": "parity of the timed run's spectra failed", "parity": parity}), flush=True)
            sys.exit(3)

    # ---- N > 1, sharded: the same per-GPU work as independent replicas (SURVEY 8e's baseline)
    replicas = None
    if (extras and world > 1 and sharded) or (a.force_replicas_leg and not dry):
        trm.free_memory()
        engine.init(case.tcfg, shard=None, device=local_rank)
        apply_conventions()
        rr = timed(a.walkers, a.steps, a.warmup, True, gather=False)
        replicas = {"value": nspectra_per_step * a.steps / rr["dt"], "unit": "spectra/s",
                    "ms_per_step": rr["dt"] / a.steps * 1e3, "walkers_per_rank": a.walkers,
                    "note": "every rank holds the whole grid and runs its own walkers: no collective; same "
                            "spectra per step as the sharded line", "diag": rr["diag"]}
        trm.free_memory()
        engine.init(case.tcfg, shard=(rank, world) if (sharded or a.force_replicas_leg) else None, device=local_rank)
        apply_conventions()

    def guarded(name, fn):
        """An extra leg must not take the contract's line down (single rank only: it has no collectives)."""
        try:
            return fn()
        except Exception as e:      # noqa: BLE001
            print("bench.py: extra leg %s failed: %r" % (name, e), file=sys.stderr)
            return {"error": repr(e)}

    sweep = {}
    if world == 1 and a.sweep and extras:
        def one_batch(b):
            k = max(16, min(50, 4000 // b))
            r = timed(b, k, 8, True)
            skm, snl = r["kern_ms"], r["nlaunch"]
            return {"spectra_per_s": b * k / r["dt"], "ms_per_step": r["dt"] / k * 1e3,
                    "rt_kernel_ms": skm / max(snl, 1),
                    "survey8d_algorithmic_GBps": engine.algorithmic_bytes(b) / (skm / max(snl, 1) / 1e3) / 1e9}
        for b in [int(x) for x in a.sweep.split(",") if x]:
            sweep[str(b)] = guarded("batch_sweep %d" % b, lambda: one_batch(b))

    if rank == 0 and dry:
        assert ok
        emit({
            "metric": "forward spectra/sec (100 layers x 1e4 wavenumbers)", "dry": True,
            "value": None, "unit": "spectra/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "DRY RUN (stub engine, gloo, CPU): launch path and step loop only, "
                                   "%d walkers per rank per step, %d samples" % (a.walkers, a.nwave),
                       "walkers_per_step": nspectra_per_step, "nlayers": a.nlayers, "nwave": a.nwave,
                       "parallelism": ("wavenumber-block shard x%d + all-gather" if a.mode == "shard"
                                       else "replicas x%d, no collective") % world},
            "scaling_diag": main_run["diag"],
            "roofline": None, "cpu_baseline": None}, detail_path=a.detail)
    elif rank == 0:
        assert ok
        value = nspectra_per_step * a.steps / dt
        # SURVEY 8d's bytes per RT launch on this GPU, to the letter: two T planes x M molecules + two planes of ONE
        # CIA table per file + profile in + spectrum out (the engine's own count, bartrt_algorithmic_bytes, follows
        # its table slots: two per file under `cia_interp spline`)
        alg = nwalk * (2.0 * a.nlayers * (hi - lo) * len(case.opmol) * 8 + 2.0 * a.nlayers * (hi - lo) * len(case.cia) * 8
                       + (len(case.species) + 1) * a.nlayers * 8 + (hi - lo) * 8)
        per_launch_s = kern_ms / 1e3 / max(nlaunch, 1)
        nsets = profs_all.shape[0]
        kname = kname_box[0]
        d_out = torch.empty((nwalk, hi - lo), dtype=torch.float64, device=dev)
        mean = lambda key: float(np.mean([m[key] for m in models]))
        uniq, eff, wfrac = mean("unique_bytes"), mean("effective_bytes"), mean("layers_walked_frac")
        # ---- cold launches: a 1 GiB scratch sweep in front of each (L2 and the 256 MiB Infinity
        # Cache hold nothing of the tables: every compulsory byte comes from HBM)
        cold = None

        def cold_pass():
            scratch = torch.zeros(1 << 27, dtype=torch.float64, device=dev)
            cms = []
            for j in range(12):
                scratch.add_(1.0)
                torch.cuda.synchronize()
                engine.timing_begin(1)
                engine.run_batch_dev(main_run["d_prof"][j % nsets], d_out)
                torch.cuda.synchronize()
                ms, n = engine.timing_end()
                cms.append(ms / max(n, 1))
            del scratch
            cmed = float(np.median(cms))
            return {"launch_ms": _stats(cms), "achieved_GBps": uniq / (cmed / 1e3) / 1e9,
                    "frac": uniq / (cmed / 1e3) / 1e9 / PEAK_HBM_GBS,
                    "note": "RT kernel after a 1 GiB read-modify-write of a scratch buffer (4x the Infinity Cache): "
                            "the launch's compulsory bytes all come from HBM; the warm figure above is taken in "
                            "the timed loop, where consecutive batches share most table planes through the "
                            "256 MiB Infinity Cache, and is an upper bound on DRAM utilisation"}
        if extras and world == 1:
            cold = guarded("cold", cold_pass)
        # ---- committed profiler figures of THIS build (tools/profile_round.sh), if any
        sid = source_id()
        same = lambda j: (j.get("source_id") == sid and j.get("walkers") == nwalk and j.get("nwave") == a.nwave
                          and j.get("nlayers") == a.nlayers and j.get("integ", 0) == integ and world == 1
                          and j.get("kappa", "forest") == a.kappa and j.get("cut", "vertical") == conv["cut"]
                          and bool(j.get("prefetch", True)) == bool(a.prefetch))
        traffic = traffic_src = None
        try:
            j = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
            if same(j):
                traffic, traffic_src = j["traffic_bytes_per_launch"], "profiles/%s_pmc.json" % j.get("tag", "?")
        except Exception:
            pass
        fp64 = None
        try:
            j = json.load(open(os.path.join(ROOT, "profiles", "isa_latest.json")))
            if j.get("source_id") == sid and j["kernel_family"] in kname:
                flop = mean("layer_wavenumbers_walked") * (2 * j["fp64_fma_per_layer"] + j["fp64_other_per_layer"])
                fp64 = {"flop_per_launch": flop, "achieved": flop / per_launch_s / 1e12, "peak": PEAK_FP64_TFLOPS,
                        "unit": "TFLOP/s", "frac": flop / per_launch_s / 1e12 / PEAK_FP64_TFLOPS,
                        "note": "fp64 VALU operations the kernel's layer loop issues per (layer, wavenumber) "
                                "(FMA = 2) x (layer, wavenumber) pairs walked; vector fp64 peak, no MFMA on this path",
                        "source": "profiles/%s" % j.get("file", "isa_latest.json")}
        except Exception:
            pass
        bound_measured = None
        try:
            j = json.load(open(os.path.join(ROOT, "profiles", "sq_latest.json")))
            if j.get("source_id") == sid:
                bound_measured = {"source": "profiles/%s" % j.get("file", "sq_latest.json")}
                for k_, v in j.get("batches", {}).items():
                    busy = v["fp64_pipe_busy_fraction"]
                    bound_measured[k_] = {
                        "bound": "fp64_valu" if busy >= 0.75 else "issue/latency",
                        "fp64_pipe_busy_fraction": busy, "resident_waves_per_simd": v["resident_waves_per_simd"],
                        "wave_time": v["wave_time"]}
        except Exception:
            pass
        wms = main_run["windows_ms"]
        res = {
            "metric": "forward spectra/sec (100 layers x 1e4 wavenumbers)",
            "value": value, "unit": "spectra/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {
                "workload_short": "H2O+CO+CO2+CH4 eclipse, %d layers x %d wavenumbers, %d walkers batched per GPU per "
                                  "step (BASELINE config 3), opacity-table path, 27 T planes, H2-H2 CIA, 5 ray angles, "
                                  "toomuch 10, %s opacities, every step its own profile preparation"
                                  % (a.nlayers, a.nwave, a.walkers, "SURVEY 8d" if a.kappa == "survey8d" else a.kappa),
                "workload": "H2O+CO+CO2+CH4 eclipse, %d layers x %d wavenumbers, %d walkers "
                            "batched per GPU per step, opacity-table path, 27 T planes, "
                            "H2-H2 CIA, 5 ray angles, toomuch 10, integration rule %d (%s); "
                            "`toomuch` cut on the %s depth, CIA interpolation %s; "
                            "walkers: PT_line T(p) with parameters uniform in the demo "
                            "retrieval's prior box, %d distinct batches cycled.  %s"
                            % (a.nlayers, a.nwave, a.walkers, integ, INTEG_NAMES[integ], conv["cut"], conv["cia_interp"],
                               nsets,
                               "DEPARTURE from SURVEY 8d (--kappa forest): the opacities are a log-normal line forest of "
                               "median ~1 cm2/g (bart_amd/synth.py kappa_layer) instead of exp(N(-25,3)) cm2/g; "
                               "the photosphere lies inside the column and the `toomuch` cut skips the fraction of "
                               "layers reported as 1 - roofline.layers_walked_frac" if a.kappa == "forest" else
                               "Opacities: SURVEY 8d's literal exp(N(-25,3)) cm2/g, CIA 1e-45 exp(N(0,1)) -- a "
                               "transparent column, every layer of every wavenumber walked"),
                "walkers_per_step": nspectra_per_step, "nlayers": a.nlayers, "nwave": a.nwave, "integ": integ,
                "cut": conv["cut"], "cia_interp": conv["cia_interp"],
                "kappa_model": a.kappa,
                "HIP_FORCE_DEV_KERNARG": os.environ.get("HIP_FORCE_DEV_KERNARG"),
                "spinup_ms": a.spinup_ms,     # untimed, before the W warm-up steps: the device at its working clocks
                "prefetch": "on (--prefetch): the batches are resident and independent, each call names the next "
                            "batch (bartrt_prefetch_profiles_dev) and the RT launch prepares its layer records in "
                            "extra workgroups -- a grid / population of models, not an MCMC step" if a.prefetch else
                            "off: every step launches its own prep_profiles kernel, as an MCMC step (whose proposal "
                            "depends on the previous spectra) has to",
                **({"DIAGNOSTIC": "--same-walkers: identical profiles in a batch, not the benchmark"}
                   if a.same_walkers else {}),
                "parallelism": ("wavenumber-block shard x%d + all-gather" % world if sharded else
                                "replicas x%d, no collective" % world) if world > 1 else "single GPU",
            },
            "windows": {
                "value_is": "the FIRST window: exactly --steps steps between barriers, as the contract times them; "
                            "the other %d windows ran right after it, same length" % (len(wms) - 1),
                "ms_per_step": _stats(wms), "spectra_per_s_median": nspectra_per_step / (np.median(wms) / 1e3),
            } if len(wms) > 1 else None,
            "parity": parity,
            "roofline": {
                "bound": "hbm", "achieved": uniq / per_launch_s / 1e9, "peak": PEAK_HBM_GBS,
                "unit": "GB/s", "frac": uniq / per_launch_s / 1e9 / PEAK_HBM_GBS,
                "traffic": traffic, "traffic_source": traffic_src,
                "traffic_GBps": traffic / per_launch_s / 1e9 if traffic else None,
                # the three fractions side by side; each is one division of figures in this object
                "fractions": {
                    "frac_survey8d_letter": alg / per_launch_s / 1e9 / PEAK_HBM_GBS,
                    "frac_survey8d_letter_note": "SURVEY 8d's bytes per spectrum (80.09 MB, no credit for table planes "
                                                 "shared between the walkers of a launch) x walkers / avg launch time / "
                                                 "peak.  NOT bounded by HBM: walkers whose temperatures fall in the "
                                                 "same bracket read the same planes, which L2 serves -- above 1 it "
                                                 "says so, not that work is skipped (see `parity`)",
                    "frac_unique_bytes": uniq / per_launch_s / 1e9 / PEAK_HBM_GBS,
                    "frac_unique_bytes_note": "= `frac`: the launch's compulsory bytes (every table row counted once per "
                                              "launch, down to where its wave stopped) / avg launch time / peak",
                    "frac_counter": traffic / per_launch_s / 1e9 / PEAK_HBM_GBS if traffic else None,
                    "frac_counter_note": "`traffic` (FETCH_SIZE x calibration + WRITE_SIZE of the committed rocprofv3 "
                                         "--pmc passes of this build and workload) / avg launch time / peak.  FETCH_SIZE "
                                         "counts requests that leave L2, INCLUDING those the 256 MiB Infinity Cache (MALL) "
                                         "serves: in the timed loop consecutive batches share most planes through it, so "
                                         "this is an upper bound on DRAM utilisation; `frac_cold` is the lower one",
                    "frac_cold": (cold or {}).get("frac"),
                    "frac_cold_note": "unique bytes / median launch time after a 1 GiB scratch sweep / peak: nothing "
                                      "in L2 or the Infinity Cache, every compulsory byte from DRAM",
                },
                "bound_measured": bound_measured,
                "cold": cold,
                "bytes_model": "achieved = unique_bytes_per_launch / avg launch time: every (layer, T plane, "
                               "molecule, 64-wavenumber column) row the launch's walkers read down to the layer "
                               "where their wave stopped, counted once per launch (walkers share planes; a row "
                               "has to come from HBM once per launch whatever the implementation), + CIA rows, "
                               "layer records in, spectra out.  Computed from the launch's own profiles and the "
                               "kernels' walked-layer record (bartrt_walked_*), averaged over the cycled batches",
                "unique_bytes_per_launch": uniq,
                "effective_bytes_per_launch": eff,
                "layers_walked_frac": wfrac,
                "kernel": kname, "launches": nlaunch, "event_stride": a.event_stride,
                "sets_sampled": int(min(nsets, nlaunch)) if np.gcd(a.event_stride, nsets) == 1 else None,
                "avg_launch_ms": per_launch_s * 1e3,
                # SURVEY 8d's per-spectrum figure x walkers: no credit for rows shared between the
                # walkers of a launch or for layers below the cut, so it is NOT bounded by the HBM
                # peak (the shared rows are served by L2) -- kept as a labelled throughput figure
                "survey8d_algorithmic_bytes_per_launch": alg,
                "survey8d_algorithmic_GBps": alg / per_launch_s / 1e9,
                "fp64": fp64,
            },
            "source_id": sid,
        }
        if extras and world == 1 and not a.prefetch:
            def prefetched():
                r = timed(nwalk, a.steps, a.warmup, True, repeats=4, prefetch=True)
                return {
                    "note": "the same windows with each call naming the next batch (bartrt_prefetch_profiles_dev): the "
                            "RT launch prepares the next batch's layer records in extra workgroups; for grids / "
                            "populations of independent models, not for an MCMC step",
                    "ms_per_step": _stats(r["windows_ms"]),
                    "spectra_per_s_median": nspectra_per_step / (np.median(r["windows_ms"]) / 1e3),
                    "rt_kernel_ms": r["kern_ms"] / max(r["nlaunch"], 1)}
            res["with_prefetch"] = guarded("with_prefetch", prefetched)
        if main_run["diag"]:
            res["scaling_diag"] = main_run["diag"]
        if replicas:
            res["replicas"] = replicas
        if sweep:
            res["batch_sweep"] = sweep
        if extras and world == 1:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import bench_configs
            # the three rules at 10 and 256 walkers
            def rules():
                isw = {}
                try:
                    for rule in (0, 1, 2):
                        trm.set_integ(rule)
                        isw[str(rule)] = {"rule": INTEG_NAMES[rule]}
                        for b in (10, 256):
                            k = 150 if b == 10 else 25
                            r = timed(b, k, 10, True)
                            isw[str(rule)][str(b)] = {"spectra_per_s": b * k / r["dt"], "ms_per_step": r["dt"] / k * 1e3,
                                                      "rt_kernel_us": r["kern_ms"] / max(r["nlaunch"], 1) * 1e3}
                finally:
                    trm.set_integ(integ)
                for b in ("10", "256"):
                    isw["rule1_over_rule0_rt_kernel_" + b] = isw["1"][b]["rt_kernel_us"] / isw["0"][b]["rt_kernel_us"]
                return isw
            res["integ_sweep"] = guarded("integ_sweep", rules)
            # the two readings of `toomuch` under the default rule, 10 and 256 walkers
            def cuts():
                csw = {}
                try:
                    for cut in ("vertical", "slant"):
                        trm.set_cut(cut)
                        csw[cut] = {}
                        for b in (10, 256):
                            k = 150 if b == 10 else 25
                            r = timed(b, k, 10, True)
                            csw[cut][str(b)] = {"spectra_per_s": b * k / r["dt"], "ms_per_step": r["dt"] / k * 1e3,
                                                "rt_kernel_us": r["kern_ms"] / max(r["nlaunch"], 1) * 1e3}
                finally:
                    trm.set_cut(conv["cut"])
                for b in ("10", "256"):
                    csw["slant_over_vertical_rt_kernel_" + b] = csw["slant"][b]["rt_kernel_us"] / csw["vertical"][b]["rt_kernel_us"]
                return csw
            res["cut_sweep"] = guarded("cut_sweep", cuts)
            trm.free_memory()
            # the two readings of `toomuch` and the prefetched form on an opacity model whose photosphere lies
            # inside the column (on 8d's transparent opacities no ray ever reaches the cut)
            other = "forest" if a.kappa == "survey8d" else "survey8d"
            res[other + "_workload"] = guarded(other, lambda: bench_configs.kappa_leg(
                a, wd, conv, other, make_profiles, launch_byte_model, PEAK_HBM_GBS))
            res["configs"] = guarded("configs", lambda: bench_configs.run_all(integ, headline_dir=wd, kappa=a.kappa))
        if world == 1 and not a.no_cpu:
            res["cpu_baseline"] = guarded("cpu_baseline", lambda: cpu_baseline(
                case, conv, profs_all.reshape(-1, profs_all.shape[-1])))
        else:
            res["cpu_baseline"] = None
        emit(res, detail_path=a.detail)
    if in_group:
        dist.barrier()
        dist.destroy_process_group()
    if not dry:
        try:
            trm.free_memory()
        except Exception:
            pass


if __name__ == "__main__":
    main()
