"""The other BASELINE.json configurations as entries of bench.py's line (`configs`), and
SURVEY 8d's literal opacity model as an extra leg (`survey8d_workload`).  Each entry is
measured on one GPU with its inputs resident in HBM and carries a `roofline` that names
the bound that applies to it:

  demo_1walker        BASELINE config 2: demo eclipse shape (CH4, 2501 samples), one walker
  wasp12b_step        BASELINE config 4's per-GPU work: WASP-12b shape (2424 samples, 4 molecules,
                      4 filters), 10 walkers through the per-step callable (parameters -> band fluxes)
  transit_10 / _256   transit geometry on the bench grid (rt_transit_mfma)
  lbl_wnosamp1 / 2160 BASELINE config 5: on-the-fly Voigt line-by-line, 1e6 lines x 1e5 points

Imported by bench.py (N = 1, default run); `python tools/bench_configs.py` prints the
object on its own."""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402

PEAK_HBM_GBS = 8000.0
PEAK_L2_GBS = 34500.0          # MI355X_MICROARCH.md: 8 x 4 MiB L2, ~34.5 TB/s aggregate
PEAK_FP64_TFLOPS = 78.6        # vector fp64; the fp64 matrix pipe has the same peak on this part
PEAK_FP64_TOPS = PEAK_FP64_TFLOPS / 2.0   # fp64 VALU lane-operations per second (an FMA is ONE operation)


def _time_launches(engine, torch, d_prof, out, steps, warm=5, prefetch=False):
    """Steps over the cycled batches.  prefetch: each call names the next batch
    (bartrt_prefetch_profiles_dev: resident, independent batches); off = the MCMC-step form,
    every step its own prep_profiles launch (bench.py's default)."""
    nsets = d_prof.shape[0]
    nxt = (lambda i: d_prof[(i + 1) % nsets]) if prefetch else (lambda i: None)
    for i in range(warm):
        engine.run_batch_dev(d_prof[i % nsets], out, next_prof=nxt(i))
    torch.cuda.synchronize()
    engine.timing_begin(1)
    t0 = time.perf_counter()
    for i in range(warm, warm + steps):
        engine.run_batch_dev(d_prof[i % nsets], out, next_prof=nxt(i))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kms, nl = engine.timing_end()
    return dt / steps, kms / max(nl, 1) / 1e3


def step_launch_bytes(case, d_params, nfilters):
    """The byte model of ONE step's RT launch (bench.launch_byte_model: every table row the launch's walkers read, once
    per launch, down to where each wave stopped) for the legs that go through the per-step callable: the step's own
    profiles (bartrt_step_profiles_dev on the same parameters) and the kernels' walked-layer record of that step."""
    import ctypes as C
    import torch
    import bench
    from bart_amd import engine, transit_module as trm
    n, npars = d_params.shape
    prof = torch.empty((n, engine.nprof()), dtype=torch.float64, device=d_params.device)
    status = torch.empty(n, dtype=torch.int32, device=d_params.device)
    trm.check(trm.lib().bartrt_step_profiles_dev(C.c_void_p(d_params.data_ptr()), n, npars, C.c_void_p(prof.data_ptr()),
                                                 C.c_void_p(status.data_ptr()), None))
    engine.walked_begin()
    engine.step_batch_dev(d_params, nfilters)
    torch.cuda.synchronize()
    walked, wpc, kname = engine.walked_end()
    ok = status.cpu().numpy() == 0
    m = bench.launch_byte_model(case, prof.cpu().numpy()[ok], walked[ok], wpc, len(case.wn),
                                spline=trm.get_cia_interp() == "spline")
    return {"unique_bytes_per_launch": m["unique_bytes"], "effective_bytes_per_launch": m["effective_bytes"],
            "layers_walked_frac": m["layers_walked_frac"], "kernel": kname, "walkers_in_the_model": int(ok.sum())}


KAPPA_TEXT = {
    "survey8d": "SURVEY 8d's opacity model to the letter (exp(N(-25,3)) cm2/g, CIA 1e-45 exp(N(0,1))): transparent "
                "column, every layer walked",
    "forest": "log-normal line forest of median ~1 cm2/g (bart_amd/synth.py kappa_layer): the photosphere lies inside "
              "the column, the `toomuch` cut ends the walk of most wavenumbers above the bottom",
}


def kappa_leg(a, wd, conv, kappa, make_profiles, launch_byte_model, peak_hbm):
    """The headline shape on the OTHER opacity model (bench.py --kappa): plain steps, prefetched steps, and the two
    `toomuch` cuts at 10 and 256 walkers, with the launch's byte model."""
    import torch
    from bart_amd import engine, synth, transit_module as trm
    base = wd[:-len("_survey8d")] if wd.endswith("_survey8d") else wd
    case = synth.make_case(base + ("" if kappa == "forest" else "_" + kappa), nlayers=a.nlayers, nwave=a.nwave,
                           kappa_model=kappa, reuse=True)
    engine.init(case.tcfg)
    try:
        trm.set_integ(conv["integ"])
        trm.set_cut(conv["cut"])
        n, nsets = a.walkers, 16
        profs = make_profiles(case, n * nsets, seed=20260103).reshape(nsets, n, -1)
        d_prof = torch.from_numpy(profs).cuda()
        out = torch.empty((n, a.nwave), dtype=torch.float64, device="cuda")
        _time_launches(engine, torch, d_prof, out, 400)           # clocks up
        step_s, kern_s = _time_launches(engine, torch, d_prof, out, 200)
        pstep_s, pkern_s = _time_launches(engine, torch, d_prof, out, 200, prefetch=True)
        engine.walked_begin()
        engine.run_batch_dev(d_prof[0], out)
        torch.cuda.synchronize()
        walked, wpc, kname = engine.walked_end()
        m = launch_byte_model(case, profs[0], walked, wpc, a.nwave, spline=conv["cia_interp"] == "spline")
        alg = n * 80.0856e6 * (a.nlayers / 100.0) * (a.nwave / 1e4)     # SURVEY 8d to the letter (bench.py)
        cuts = {}
        for cut in ("vertical", "slant"):
            trm.set_cut(cut)
            cuts[cut] = {}
            for b in (10, 256):
                pb = make_profiles(case, b * 4, seed=20260110 + b).reshape(4, b, -1)
                dpb = torch.from_numpy(pb).cuda()
                ob = torch.empty((b, a.nwave), dtype=torch.float64, device="cuda")
                st, ks = _time_launches(engine, torch, dpb, ob, 120 if b == 10 else 24)
                cuts[cut][str(b)] = {"spectra_per_s": b / st, "ms_per_step": st * 1e3, "rt_kernel_us": ks * 1e6}
        trm.set_cut(conv["cut"])
        for b in ("10", "256"):
            cuts["slant_over_vertical_rt_kernel_" + b] = cuts["slant"][b]["rt_kernel_us"] / cuts["vertical"][b]["rt_kernel_us"]
        return {
            "workload": "%s; %d walkers per step, integ %d, cut %s" % (KAPPA_TEXT[kappa], n, conv["integ"], conv["cut"]),
            "value": n / step_s, "unit": "spectra/s", "ms_per_step": step_s * 1e3, "rt_kernel_ms": kern_s * 1e3,
            "with_prefetch": {"value": n / pstep_s, "ms_per_step": pstep_s * 1e3, "rt_kernel_ms": pkern_s * 1e3},
            "kernel": kname, "layers_walked_frac": m["layers_walked_frac"],
            "spectrum_max": float(out.max()),
            "cut_sweep": cuts,
            "roofline": {"bound": "hbm", "achieved": m["unique_bytes"] / kern_s / 1e9, "peak": peak_hbm, "unit": "GB/s",
                         "frac": m["unique_bytes"] / kern_s / 1e9 / peak_hbm,
                         "unique_bytes_per_launch": m["unique_bytes"],
                         "survey8d_algorithmic_bytes_per_launch": alg,
                         "frac_survey8d_letter": alg / kern_s / 1e9 / peak_hbm,
                         "note": "unique bytes = every table row the launch's walkers read, once per launch, down to "
                                 "where each wave stopped; the 8d-letter figure gives no credit for planes shared "
                                 "between walkers (L2 serves them) and is not bounded by the HBM peak"}}
    finally:
        trm.free_memory()


def demo_1walker(integ):
    import torch
    import bench
    from bart_amd import engine, synth, transit_module as trm
    d = os.path.join(tempfile.gettempdir(), "bartrt_cfg_demo")
    case = synth.make_case(d, nlayers=100, nwave=2501, wnlow=2500.0, opmol=("CH4",), seed=7, reuse=True)
    engine.init(case.tcfg)
    try:
        trm.set_integ(integ)
        nsets = 16
        profs = bench.make_profiles(case, nsets, seed=5).reshape(nsets, 1, -1)
        d_prof = torch.from_numpy(profs).cuda()
        out = torch.empty((1, 2501), dtype=torch.float64, device="cuda")
        step_s, kern_s = _time_launches(engine, torch, d_prof, out, 300, warm=20)
        engine.walked_begin()
        engine.run_batch_dev(d_prof[0], out)
        torch.cuda.synchronize()
        walked, wpc, kname = engine.walked_end()
        m = bench.launch_byte_model(case, profs[0], walked, wpc, 2501)
        # the reference's call shape: host array in, new host array out (trm.run_transit)
        lat = []
        p0 = profs[0, 0].copy()
        for i in range(220):
            t0 = time.perf_counter()
            trm.run_transit(p0, 2501)
            lat.append(time.perf_counter() - t0)
        lat = np.array(lat[20:])
        return {
            "workload": "BASELINE config 2: demo eclipse shape (CH4, 100 layers x 2501 samples, 2-4 um), ONE walker, "
                        "precomputed opacity table, integ %d" % integ,
            "value": 1.0 / step_s, "unit": "spectra/s", "ms_per_step": step_s * 1e3, "rt_kernel_ms": kern_s * 1e3,
            "kernel": kname,
            "host_call_run_transit_us": {"median": float(np.median(lat) * 1e6), "p90": float(np.percentile(lat, 90) * 1e6)},
            "roofline": {"bound": "latency (one walker: 40 columns of 100 dependent layers on a 1 024-SIMD chip; the "
                                  "quad-layer kernel walks them 8 layers per step) -- quoted against HBM for scale",
                         "achieved": m["unique_bytes"] / kern_s / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": m["unique_bytes"] / kern_s / 1e9 / PEAK_HBM_GBS,
                         "unique_bytes_per_launch": m["unique_bytes"], "layers_walked_frac": m["layers_walked_frac"]}}
    finally:
        trm.free_memory()


def wasp12b_step(integ):
    import torch
    from bart_amd import BARTfunc, engine, synthcfg, transit_module as trm
    mols = ("H2O", "CO", "CO2", "CH4")
    p0 = (-2.0, 0.0, 1.0, 0.0, 0.98, -0.5, -0.5, -0.5, -0.5)
    d = os.path.join(tempfile.gettempdir(), "bartrt_cfg_wasp")
    case, cfg = synthcfg.make_worker_case(d, nwave=2424, wnlow=910.0, opmol=mols, molfit=mols, params=p0,
                                          nfilters=4, reuse=True)
    w = BARTfunc.Worker(BARTfunc.WorkerConfig.from_cfg(cfg))
    try:
        trm.set_integ(integ)
        rng = np.random.default_rng(5)
        nsets, n, steps = 16, 10, 300
        pars = np.array(p0) + rng.normal(0, [0.3, 0.2, 0.2, 0.05, 0.02, 0.5, 0.5, 0.5, 0.5], (nsets, n, 9))
        pars[..., 3] = np.clip(pars[..., 3], 0, 1)
        d_par = torch.from_numpy(pars).cuda()
        for i in range(20):
            band, status = engine.step_batch_dev(d_par[i % nsets], w.nfilters)
        torch.cuda.synchronize()
        engine.timing_begin(1)
        t0 = time.perf_counter()
        for i in range(steps):
            band, status = engine.step_batch_dev(d_par[i % nsets], w.nfilters)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        kms, nl = engine.timing_end()
        kern_s = kms / max(nl, 1) / 1e3
        alg = engine.algorithmic_bytes(n)
        lb = step_launch_bytes(case, d_par[0], w.nfilters)
        return {
            "workload": "BASELINE config 4's per-GPU work: WASP-12b shape (100 layers x 2424 samples, 4 molecules, "
                        "4 filters), 10 walkers per step through the per-step callable (parameters -> T(p), "
                        "abundances -> RT -> band fluxes, all on the device), integ %d" % integ,
            "value": n / dt, "unit": "walker-steps/s", "ms_per_step": dt * 1e3, "rt_kernel_ms": kern_s * 1e3,
            "accepted_in_last_batch": int((status.cpu().numpy() == 0).sum()),
            "roofline": {"bound": "latency / issue (four short dependent launches per step; the RT launch is 380 "
                                  "columns on 1 024 SIMDs) -- quoted as SURVEY 8d bytes over the RT kernel for scale",
                         "achieved": alg / kern_s / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": alg / kern_s / 1e9 / PEAK_HBM_GBS,
                         "survey8d_algorithmic_bytes_per_launch": alg,
                         "unique_bytes_per_launch": lb["unique_bytes_per_launch"],
                         "frac_unique_bytes": lb["unique_bytes_per_launch"] / kern_s / 1e9 / PEAK_HBM_GBS,
                         "kernel": lb["kernel"], "layers_walked_frac": lb["layers_walked_frac"],
                         "rt_kernel_share_of_step": kern_s / dt,
                         "note": "8d bytes give no credit for the planes the ten walkers share (served by L2): an "
                                 "upper bound on the launch's DRAM traffic"}}
    finally:
        w.close()


class no_gc:
    """Timed loops run with the interpreter's cyclic collector held off, as timeit does (with torch imported a full
    collection is a 30-40 ms pause; round 5: it came at the same two of 300 host calls in every run)."""
    def __enter__(self):
        import gc
        gc.disable()      # (collect BEFORE the warm-up calls, not here: 40 ms of idle device cost the timed loop its clocks)

    def __exit__(self, *exc):
        import gc
        gc.enable()


def full_step_10(integ, headline_dir=None, kappa="survey8d"):
    """The whole MCMC step at the HEADLINE shape: 100 layers x 1e4 samples, 4 molecules, 10 walkers,
    10 filters with the energy-balance check on -- parameters -> T(p) / abundances / layer records -> RT -> band fluxes
    (code/BARTfunc.py:309-399), device buffers in and out, one host synchronisation per step as an MCMC driver needs
    it.  The bench line's `value` times the RT call alone; this is what a sampler pays per iteration."""
    import torch
    from bart_amd import BARTfunc, engine, synthcfg, transit_module as trm
    mols = ("H2O", "CO", "CO2", "CH4")
    p0 = (-2.0, 0.0, 1.0, 0.0, 0.98, -0.5, -0.5, -0.5, -0.5)
    d = os.path.join(tempfile.gettempdir(), "bartrt_cfg_fullstep")
    case, cfg = synthcfg.make_worker_case(d, nwave=10000, wnlow=1000.0, opmol=mols, molfit=mols, params=p0,
                                          nfilters=10, ebalance=True, kappa_model=kappa, reuse=True)
    w = BARTfunc.Worker(BARTfunc.WorkerConfig.from_cfg(cfg))
    try:
        trm.set_integ(integ)
        rng = np.random.default_rng(5)
        nsets, n, steps = 16, 10, 300
        pars = np.array(p0) + rng.normal(0, [0.3, 0.2, 0.2, 0.05, 0.02, 0.5, 0.5, 0.5, 0.5], (nsets, n, 9))
        pars[..., 3] = np.clip(pars[..., 3], 0, 1)
        d_par = torch.from_numpy(pars).cuda()
        out = {}
        import gc
        for sync_each in (True, False):
            gc.collect()
            for i in range(40):
                band, status = engine.step_batch_dev(d_par[i % nsets], w.nfilters)
            torch.cuda.synchronize()
            engine.timing_begin(1)
            with no_gc():
                t0 = time.perf_counter()
                for i in range(steps):
                    band, status = engine.step_batch_dev(d_par[i % nsets], w.nfilters)
                    if sync_each:
                        torch.cuda.synchronize()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / steps
            kms, nl = engine.timing_end()
            out["synchronised_every_step" if sync_each else "queued_back_to_back"] = {
                "us_per_step": dt * 1e6, "walker_steps_per_s": n / dt, "rt_kernel_us": kms / max(nl, 1) * 1e3,
                "step_minus_rt_kernel_us": dt * 1e6 - kms / max(nl, 1) * 1e3}
        # ... and through the host call an MCMC driver on the host makes (parameters in host memory, band fluxes back in host
        # memory: the kernels read and write a pinned buffer, the host polls a word the stream writes -- csrc/step.hip)
        gc.collect()
        for i in range(40):
            engine.step_batch(pars[i % nsets], w.nfilters)
        lat = np.zeros(steps)
        with no_gc():
            t0 = time.perf_counter()
            for i in range(steps):
                t1 = time.perf_counter()
                hb, hs = engine.step_batch(pars[i % nsets], w.nfilters)
                lat[i] = time.perf_counter() - t1
            dt = (time.perf_counter() - t0) / steps
        out["host_call_step_batch"] = {"us_per_step": dt * 1e6, "walker_steps_per_s": n / dt, "median_us": float(np.median(lat) * 1e6),
                                       "p90_us": float(np.percentile(lat, 90) * 1e6), "max_us": float(lat.max() * 1e6),
                                       "calls_over_twice_the_median": int((lat > 2 * np.median(lat)).sum()),
                                       "slowest_calls": [[int(i), round(float(lat[i]) * 1e6, 1)] for i in np.argsort(lat)[-6:][::-1]]}
        db, _ = engine.step_batch_dev(d_par[(steps - 1) % nsets], w.nfilters)
        out["host_call_equals_device_call"] = bool(np.array_equal(db.cpu().numpy(), hb))
        # run to run: the same parameters give the same band-flux bits
        b1, _ = engine.step_batch_dev(d_par[0], w.nfilters); b1 = b1.clone()
        b2, _ = engine.step_batch_dev(d_par[0], w.nfilters)
        torch.cuda.synchronize()
        # the RT launch of a step against the HBM roofline, on the launch's own unique bytes ("the line gives no
        # unique_bytes for that leg -- its fraction cannot be recomputed")
        lb = step_launch_bytes(case, d_par[0], w.nfilters)
        ks = out["queued_back_to_back"]["rt_kernel_us"] * 1e-6
        out["roofline"] = {"bound": "hbm", "unit": "GB/s", "peak": PEAK_HBM_GBS, "achieved": lb["unique_bytes_per_launch"] / ks / 1e9,
                           "frac": lb["unique_bytes_per_launch"] / ks / 1e9 / PEAK_HBM_GBS, **lb,
                           "avg_launch_us": out["queued_back_to_back"]["rt_kernel_us"],
                           "note": "the step's walkers are a cluster around one point of parameter space (sigma 0.2-0.5 "
                                   "dex): they share more temperature planes than bench.py's prior-wide draws, fewer unique "
                                   "bytes per launch, a shorter launch"}
        out["workload"] = ("whole step at the headline shape: 100 layers x 1e4 samples, 4 molecules, 10 filters, energy "
                           "balance on, 10 walkers per step, integ %d; parameters in, band fluxes out, on the device" % integ)
        out["accepted_in_last_batch"] = int((status.cpu().numpy() == 0).sum())
        out["band_fluxes_bit_stable"] = bool(torch.equal(b1, b2))
        out["launches_per_step"] = int(os.environ.get("BARTRT_STEP_LAUNCHES", "0")) or None
        return out
    finally:
        w.close()


def wasp12b_shard8(integ):
    """BASELINE config 4 at N = 8 as ONE rank sees it: rank 3 of 8 of the WASP-12b grid (303 of 2424 samples), ten walkers
    per step -- parameters -> profiles -> RT on the block (the all-gather and the band integration on the gathered
    spectrum are the other ranks' business too and are not in this figure).  Twice: with the kernel chosen by the LOCAL block's
    columns (the default since round 5; blocks agree with the unsharded run to rounding) and by the WHOLE grid's (blocks
    bit-identical to the unsharded run; include/bartrt.h, bartrt_set_kernel_by).  DESIGN.md section 5 takes its N = 8 recommendation from these two."""
    import ctypes as C
    import torch
    from bart_amd import BARTfunc, engine, synthcfg, transit_module as trm
    mols = ("H2O", "CO", "CO2", "CH4")
    p0 = (-2.0, 0.0, 1.0, 0.0, 0.98, -0.5, -0.5, -0.5, -0.5)
    d = os.path.join(tempfile.gettempdir(), "bartrt_cfg_wasp")
    case, cfg = synthcfg.make_worker_case(d, nwave=2424, wnlow=910.0, opmol=mols, molfit=mols, params=p0,
                                          nfilters=4, reuse=True)
    out = {"workload": "BASELINE config 4, one rank's share at N = 8: block 3 of 8 of the WASP-12b grid (303 samples, 100 "
                       "layers, 4 molecules), 10 walkers per step: parameters -> profiles -> RT on the block, integ %d" % integ}
    w = BARTfunc.Worker(BARTfunc.WorkerConfig.from_cfg(cfg), shard=(3, 8))
    try:
        trm.set_integ(integ)
        lo, hi = engine.local_range()
        rng = np.random.default_rng(5)
        nsets, n, steps = 16, 10, 400
        pars = np.array(p0) + rng.normal(0, [0.3, 0.2, 0.2, 0.05, 0.02, 0.5, 0.5, 0.5, 0.5], (nsets, n, 9))
        pars[..., 3] = np.clip(pars[..., 3], 0, 1)
        d_par = torch.from_numpy(pars).cuda()
        prof = torch.empty((n, engine.nprof()), dtype=torch.float64, device="cuda")
        stat = torch.empty(n, dtype=torch.int32, device="cuda")
        spec = torch.empty((n, hi - lo), dtype=torch.float64, device="cuda")
        sp = engine._stream_ptr()

        def one(i):
            trm.check(trm.lib().bartrt_step_profiles_dev(C.c_void_p(d_par[i % nsets].data_ptr()), n, 9, C.c_void_p(prof.data_ptr()),
                                                         C.c_void_p(stat.data_ptr()), sp))
            engine.run_batch_dev(prof, spec)
        res = {}
        for which in ("whole", "local"):
            trm.set_kernel_by(which)
            for i in range(20):
                one(i)
            torch.cuda.synchronize()
            engine.timing_begin(1)
            t0 = time.perf_counter()
            for i in range(steps):
                one(i)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
            kms, nl = engine.timing_end()
            engine.walked_begin()
            one(0)
            torch.cuda.synchronize()
            _, _, kname = engine.walked_end()
            res[which] = spec.cpu().numpy().copy()
            out["kernel_by_" + which] = {"us_per_step": dt * 1e6, "rt_kernel_us": kms / max(nl, 1) * 1e3, "kernel": kname,
                                         "walker_steps_per_s_per_rank": n / dt}
        out["local_vs_whole_max_rel_diff"] = float(np.max(np.abs(res["local"] / res["whole"] - 1.0)))
        out["block"] = [int(lo), int(hi)]
    finally:
        w.close()
    return out


def transit_geometry(integ, batches=(10, 256)):
    import torch
    import bench
    from bart_amd import engine, synth, transit_module as trm
    d = os.path.join(tempfile.gettempdir(), "bartrt_cfg_transit")
    case = synth.make_case(d, nlayers=100, nwave=10000, reuse=True,
                           extra_keys={"solution": "transit", "starrad": 1.145})
    engine.init(case.tcfg)
    out = {}
    try:
        L, W = 100, 10000
        nkt = (L + 15) // 16
        for n in batches:
            nsets = 8
            profs = bench.make_profiles(case, n * nsets, seed=11).reshape(nsets, n, -1)
            d_prof = torch.from_numpy(profs).cuda()
            spec = torch.empty((n, W), dtype=torch.float64, device="cuda")
            steps = max(10, min(100, 2000 // n))
            step_s, kern_s = _time_launches(engine, torch, d_prof, spec, steps)
            alg = engine.algorithmic_bytes(n)
            # a wave (16 wavenumbers of one walker) reads its 16-wavenumber slice of the table rows
            # (SURVEY 8d bytes) and the walker's chord-operand tiles: the chord matrix is lower
            # triangular, row tile kt takes the steps of the layers 0 .. 16 kt + 15: sum_kt 4 (kt + 1)
            # tile-steps of 512 B and one v_mfma_f64_16x16x4 (2 x 16 x 16 x 4 flop) each
            tile_steps = 2 * nkt * (nkt + 1)
            chord = n * (W / 16.0) * tile_steps * 512.0
            mfma_flops = n * (W / 16.0) * tile_steps * 2.0 * 16 * 16 * 4
            out["transit_%d" % n] = {
                "workload": "transit geometry (modulation spectra), 100 layers x 1e4 wavenumbers, %d walkers per "
                            "step, chord optical depths on v_mfma_f64_16x16x4" % n,
                "value": n / step_s, "unit": "spectra/s", "ms_per_step": step_s * 1e3, "rt_kernel_ms": kern_s * 1e3,
                "roofline": {"bound": "l2 (table rows + chord-operand tiles out of L2; MFMA pipe lightly loaded)",
                             "achieved": (alg + chord) / kern_s / 1e9, "peak": PEAK_L2_GBS, "unit": "GB/s",
                             "frac": (alg + chord) / kern_s / 1e9 / PEAK_L2_GBS,
                             "l2_bytes_model": {"table_rows_survey8d": alg, "chord_operand_tiles": chord},
                             "mfma": {"flop_per_launch": mfma_flops, "achieved_TFLOPs": mfma_flops / kern_s / 1e12,
                                      "peak_TFLOPs": PEAK_FP64_TFLOPS,
                                      "frac": mfma_flops / kern_s / 1e12 / PEAK_FP64_TFLOPS},
                             "note": "upper bounds: row tiles below the toomuch cut are skipped (neither loaded nor "
                                     "multiplied); the kernel records no walked depth"}}
    finally:
        trm.free_memory()
    return out


def lbl(integ, wnosamps=(1, 2160)):
    """Config 5 through tools/lbl_bench.py's harness, with the fp64 operation count of the Voigt
    samples as the roofline (the path is arithmetic: under one byte of line list and extinction
    array per sample)."""
    import contextlib
    import io
    import lbl_bench
    out = {}
    for o in wnosamps:
        with contextlib.redirect_stdout(io.StringIO()):
            r = lbl_bench.run(["--wnosamp", str(o), "--reps", "3"])
        ops = r["voigt_fp64_ops"]
        out["lbl_wnosamp%d" % o] = {
            "workload": "BASELINE config 5: " + r["workload"],
            "value": r["value"], "unit": "spectra/s", "ms_per_step": r["seconds_per_spectrum"] * 1e3,
            "voigt_samples": r["voigt_samples"], "voigt_samples_per_s": r["voigt_samples_per_s"],
            "roofline": {"bound": "fp64_valu (Voigt arithmetic: 19-105 fp64 operations per profile sample, under "
                                  "one byte per sample)",
                         "achieved": ops / r["seconds_per_spectrum"] / 1e12, "peak": PEAK_FP64_TOPS,
                         "unit": "Tops/s (fp64 VALU lane-operations; an FMA counts once; peak = 78.6 TFLOP/s / 2)",
                         "frac": ops / r["seconds_per_spectrum"] / 1e12 / PEAK_FP64_TOPS,
                         "op_model": {"per_sample": r["voigt_ops_model"], "samples_by_branch": r["voigt_samples_by_branch"],
                                        "note": "samples inside the lines' cuts by branch of the Faddeeva evaluation "
                                                "(tools/lbl_bench.py work_units) x the branch's fp64 operations; "
                                                "staging, strengths and the reduction of oversampled layers are not "
                                                "counted, so this is a lower bound on the arithmetic done"},
                         "algorithmic_GBps": r["algorithmic_GBps"]}}
    return out


def mc3_processes(headline_dir, kappa, nprocs=(3, 10), steps=1500):
    """INTEGRATION.md section 1 taken literally: MC3 starts one worker process per chain (examples/WASP-12b/BART.cfg:113:
    ten; the demo three) and each holds its own transit instance -- transit_init, then run_transit once per step.  N
    such processes (tools/mc3_child.py) on ONE GPU and the headline grid, in lockstep: with every process uploading its
    own 864 MB grid (own_copies_N); with `shareOpacity` (code/makecfg.py:106-107) as the chain service, the default
    (shared_N: one process owns the engine, all post their profiles into shared-memory slots, one batched launch per
    step -- csrc/svc_core.hpp); and with the key read as BARTRT_SHARE_MODE=ipc (ipc_N: every process its own engine,
    the grid one allocation mapped through HIP IPC -- csrc/share.hip).  Reported: aggregate spectra/s, time per call,
    init time, device memory in use, and for the service the mean batch per launch."""
    import subprocess
    import torch
    from bart_amd import synth
    case = synth.make_case(headline_dir, nlayers=100, nwave=10000, kappa_model=kappa, reuse=True)
    shared_cfg = case.tcfg + ".share"
    open(shared_cfg, "w").write(open(case.tcfg).read().rstrip("\n") + "\nshareOpacity\n")
    child = os.path.join(ROOT, "tools", "mc3_child.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")

    def used():
        free, total = torch.cuda.mem_get_info()
        return total - free

    def expect(p, word):
        for line in p.stdout:
            if line.startswith(word + " "):
                return json.loads(line[len(word) + 1:])
        raise RuntimeError("worker ended without '%s': %s" % (word, p.stderr.read()[-2000:]))

    out = {"note": "N processes x one walker per call through trm.run_transit (host buffers in and out), all on one GPU; "
                   "the grouped worker (BARTfunc.main(comm, group): the chains of all workers batched into one call) and "
                   "the in-process sampler are the forms that reach the bench line's rate -- this is the unmodified path"}
    for mode in ("own_copies", "ipc", "shared"):
        share = mode != "own_copies"
        menv = dict(env, BARTRT_SHARE_MODE="ipc") if mode == "ipc" else env
        for n in nprocs:
            base = used()
            t0 = time.perf_counter()
            procs = [subprocess.Popen([sys.executable, child, shared_cfg if share else case.tcfg, str(r), str(steps)],
                                      stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=menv)
                     for r in range(n)]
            try:
                ready = [expect(p, "ready") for p in procs]
                t_up = time.perf_counter() - t0
                mem = used() - base
                for p in procs:
                    p.stdin.write("go\n"); p.stdin.flush()
                done = [expect(p, "done") for p in procs]
                for p in procs:
                    p.stdin.write("bye\n"); p.stdin.flush()
                for p in procs:
                    p.wait(timeout=180)
            finally:
                for p in procs:
                    if p.poll() is None:
                        p.kill()
            loop = max(d["loop_s"] for d in done)
            leg = {
                "processes": n, "steps_per_process": steps, "aggregate_spectra_per_s": n * steps / loop,
                "us_per_call_median": float(np.median([d["us_per_step"] for d in done])),
                "call_us_median_of_medians": float(np.median([d["call_us_median"] for d in done])),
                "call_us_p90_max": float(max(d["call_us_p90"] for d in done)),
                "init_s_median": float(np.median([r["init_s"] for r in ready])), "all_ready_after_s": t_up,
                "device_memory_in_use_GB": mem / 1e9, "owners": int(sum(r["owner"] for r in ready)),
                "shared": bool(all(r["shared"] for r in ready)),
                "hip_contexts": int(sum(bool(d.get("hip_context")) for d in done))}
            if mode == "shared":
                st = max((d.get("service_stats", {"launches": 0, "profiles": 0, "full": 0}) for d in done), key=lambda x: x["launches"])
                leg["service"] = {"launches": st["launches"], "profiles": st["profiles"], "full_rounds": st["full"],
                                  "mean_batch": st["profiles"] / max(st["launches"], 1),
                                  "roles": sorted(r["service"] for r in ready)}
            out["%s_%d" % (mode, n)] = leg
    return out


def run_all(integ, headline_dir=None, kappa="survey8d"):
    res = {}
    if headline_dir:
        # (the one leg that waits on other processes' pipes: bounded, so that a worker that never answers costs the line
        # this leg and not the run)
        import signal

        def _late(signum, frame):
            raise TimeoutError("mc3_processes: the worker processes did not answer within 900 s")
        armed = False
        try:
            old = signal.signal(signal.SIGALRM, _late)      # (main thread only; elsewhere the leg runs unbounded as before)
            signal.alarm(900)
            armed = True
        except ValueError:
            pass
        try:
            res["mc3_processes"] = mc3_processes(headline_dir, kappa)
        except Exception as e:
            res["mc3_processes"] = {"error": repr(e)}
        finally:
            if armed:
                signal.alarm(0)
                signal.signal(signal.SIGALRM, old)
    for name, fn in (("demo_1walker", demo_1walker), ("wasp12b_step", wasp12b_step), ("wasp12b_shard8", wasp12b_shard8),
                     ("full_step_10", full_step_10)):
        try:
            res[name] = fn(integ)
        except Exception as e:      # an extra leg must not take the contract's line down
            res[name] = {"error": repr(e)}
    for fn in (transit_geometry, lbl):
        try:
            res.update(fn(integ))
        except Exception as e:
            res[fn.__name__] = {"error": repr(e)}
    return res


if __name__ == "__main__":
    from bart_amd import transit_module as trm   # noqa: F401
    print(json.dumps(run_all(int(sys.argv[1]) if len(sys.argv) > 1 else 1)))
