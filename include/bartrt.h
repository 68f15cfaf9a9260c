/*
 * bartrt.h -- C ABI of libbartrt.so, the MI355X-native forward
 * radiative-transfer engine that replaces BART's `transit_module`.
 *
 * Every entry point below stands in for one call BART's per-step callable
 * makes on the SWIG module (reference code/BARTfunc.py, line cited per
 * function).  Plain pointers and sizes only; host pointers unless the name
 * ends in `_dev`.  All functions return 0 on success and a negative code on
 * error (message via bartrt_last_error()) unless documented otherwise.  The
 * library computes ONLY on the GPU: with no HIP device every compute entry
 * point fails with BARTRT_ENODEV -- there is no CPU fallback.
 *
 * Engine state is a process-global SINGLETON, like the reference module's
 * (created by transit_init, destroyed by free_memory; BARTfunc.py:230,406):
 *   - one engine, on one device, per process.  bartrt_init on a live engine REPLACES it
 *     (the old one is destroyed first); a host that drives several GPUs runs one process per GPU (that is
 *     how bench.py, bart_amd.retrieve and the MC3-driven worker shard a node), each with
 *     "--device <n>" or LOCAL_RANK;
 *   - calls are NOT re-entrant and not thread-safe: drive the engine from one host thread
 *     at a time (the reference runs one engine per MPI process).  The launch state of the
 *     kernels (variant switches read from the environment, the timing and walked-layer
 *     records, the prefetch request) belongs to that one engine;
 *   - the `_dev` entry points are asynchronous on the stream they are given; the others
 *     return when their result is in the caller's buffer;
 *   - the environment variables (BARTRT_INTEG, BARTRT_KERNEL, BARTRT_LBL, ...) are read
 *     once per process or per bartrt_init, not per call.
 */
#ifndef BARTRT_H
#define BARTRT_H

#ifdef __cplusplus
extern "C" {
#endif

#define BARTRT_OK        0
#define BARTRT_EINVAL   -1   /* bad argument / not initialised */
#define BARTRT_EIO      -2   /* input file missing or malformed */
#define BARTRT_ENODEV   -3   /* no HIP device / HIP runtime error */
#define BARTRT_ENOTSUP  -4   /* configuration not supported by this build */

/* ---- the eight reference entry points -------------------------------- */

/* trm.transit_init(argc, argv)  [BARTfunc.py:229-230]
 * argv = {"transit", "-c", <transit cfg>}.  Extra flags accepted after the
 * cfg: "--shard <rank> <nranks>" keeps only that contiguous block of the
 * wavenumber grid on this process's GPU; "--device <n>"; "--no-service"
 * (see bartrt_get_share). */
int bartrt_init(int argc, const char **argv);

/* trm.get_no_samples()  [BARTfunc.py:233].  Full-grid sample count (>=0),
 * or a negative error code. */
int bartrt_get_no_samples(void);

/* trm.get_waveno_arr(n)  [BARTfunc.py:234].  out[n] = wavenumbers, cm-1. */
int bartrt_get_waveno_arr(double *out, int n);

/* trm.set_radius(r)  [BARTfunc.py:351].  Reference radius in km. */
int bartrt_set_radius(double refradius_km);

/* trm.set_cloudtop(p)  [BARTfunc.py:354].  log10(cloud-top pressure / bar). */
int bartrt_set_cloudtop(double log10_pbar);

/* trm.set_scattering(flag, value)  [BARTfunc.py:358,360]. */
int bartrt_set_scattering(int flag, double value);

/* trm.run_transit(profiles.flatten(), nwave)  [BARTfunc.py:363].
 * prof[nprof] with nprof = (nspecies+1)*nlayers: row 0 temperature (K), rows
 * 1..S mole mixing ratios in atm-file species order, layers in atm-file order
 * (BARTfunc.py:213-222).  spec[nwave] = emergent flux, erg s-1 cm-2 cm.
 * On a sharded engine nwave may be the full count (only the shard's block is
 * written) or the shard's count. */
int bartrt_run_transit(const double *prof, int nprof, double *spec, int nwave);

/* trm.free_memory()  [BARTfunc.py:406]. */
int bartrt_free_memory(void);

/* Integration rule of the eclipse geometry (no counterpart in the reference's
 * module: its engine, whose source would settle the rule, is an empty submodule --
 * DESIGN.md conventions C6 / C8).
 *   1 (DEFAULT) = the Simpson / trapezoid hybrid SURVEY.md App. A-4 recalls for the
 *       reference's engine: the optical depth over radius and B exp(-tau/mu) over tau
 *       (zero-padded past `last`) -- the only statement about the engine's integrator
 *       the project holds;
 *   0 = trapezoid in the transmittance (exact for isothermal columns);
 *   2 = plain trapezoid in tau of B exp(-tau/mu).
 * Also the cfg key `integ` (number or transmittance / simpson / trapz_tau) and the
 * environment variable BARTRT_INTEG, read by bartrt_init; every eclipse kernel is built
 * for each rule.  bartrt_get_integ writes the rule in force to *rule and returns a status
 * like every other entry point (rule 0 and BARTRT_OK would otherwise share a value); the
 * `transit` executable and the worker log the rule they ran under. */
int bartrt_set_integ(int rule);
int bartrt_get_integ(int *rule);

/* Which optical depth `toomuch` is compared with (no counterpart in the reference's
 * module).  1 (default since round 4) = each ray's SLANT depth tau / mu: SURVEY.md App. A-4 read
 * literally ("slant path ds = dr / cos(theta) ... stops where tau > toomuch") and the reading the
 * engine's call structure is recalled to have -- the optical depth is computed per ray angle, the
 * cut inside that loop; every angle ends on its own layer and rule 1 pads one unit of slant depth.
 * 0 = the vertical depth: the column ends on one layer for every ray angle.  Both run specialised
 * kernels (rules 0, 1 and 2; the slant cut costs 1.10-1.18x per launch, MEASUREMENTS_ARCHIVE.md); under rule 0
 * the two differ by about exp(-toomuch) of the flux, under rule 1 (the default) by far more -- the padded
 * point moves with the cut: 4.6e-3 relative on a 40-layer synthetic column at toomuch 10 (and spline
 * against linear CIA 3.5e-3: the round-4 change of defaults moved spectra at the 0.5 % level).  Also
 * the cfg key `cut vertical|slant` and BARTRT_CUT. */
int bartrt_set_cut(int slant);
int bartrt_get_cut(int *slant);

/* Sharded engines ("--shard r n"): which column count picks the kernel variant.  1 (DEFAULT) = this
 * block's own: a block of a few hundred samples takes the layer-parallel kernels instead of the
 * single-wave kernel's latency floor (the WASP-12b grid on eight GPUs, 303 samples x 10 walkers per
 * rank: 16 us against 52 us per launch); spectra agree with the unsharded run's to rounding (4e-16
 * measured), not bit for bit -- north_star asks for 1e-6.  0 = the WHOLE grid's: every block is computed
 * by the kernel the unsharded run would use, so the blocks concatenate to its spectrum bit for bit.
 * Also cfg `kernel_by local|whole` and BARTRT_KERNEL_BY.  No effect on an unsharded engine. */
int bartrt_set_kernel_by(int local);
int bartrt_get_kernel_by(int *local);

/* How the engine interpolates the cross-section (CIA) files, fixed at bartrt_init by the cfg key
 * `cia_interp linear|spline` / BARTRT_CIA_INTERP: *spline = 1 (default since round 4)
 * natural cubic splines in wavenumber and temperature, 0 linear in both.  Read-only: the tables are
 * resampled at init.  A natural spline may undershoot zero between two samples of a steep table: the
 * values are used as the spline gives them, NOT clamped (engine and oracle alike) -- a clamp would be
 * one more unverified convention. */
int bartrt_get_cia_interp(int *spline);

/* `shareOpacity` (a key of the reference's transit cfg: code/makecfg.py:106-107, BART.py:259-262 --
 * BART's worker processes, one per chain, keep ONE copy of the opacity grid).  With the key in the
 * cfg (or BARTRT_SHARE_OPACITY=1) the worker processes of a run share one grid in one of two ways,
 * chosen by BARTRT_SHARE_MODE:
 *   service (default)  THE CHAIN SERVICE.  The first process to initialise on a (cfg, GPU, wavenumber
 *       block) builds the engine and starts a dispatcher thread; every worker process -- that one's
 *       own caller included -- is a client: bartrt_run_transit copies the profile into the process's
 *       slot of a POSIX shared-memory segment and sleeps on a futex, the dispatcher launches ONE batch
 *       for all the profiles posted together (MC3 releases its workers together: code/BARTfunc.py:312,
 *       399) and wakes the callers.  A client makes no HIP call at all: one context, one grid, one
 *       launch per MCMC step.  The setters (set_radius / set_cloudtop / set_scattering) of a client
 *       act on ITS profiles only, per walker of the batch.  Served to clients: the reference module's
 *       eight calls, bartrt_run_transit_batch and the plain getters; the other entry points return
 *       BARTRT_ENOTSUP.  A client whose owner has gone gets BARTRT_ENODEV; the owner's
 *       bartrt_free_memory keeps serving until the other workers have let go (BARTRT_SHARE_WAIT_S
 *       seconds at most, default 60).  Tunables: BARTRT_SVC_WINDOW_US (30: how long the dispatcher
 *       waits for the workers of the previous batch after the latest arrival), BARTRT_SVC_SPIN_US
 *       (200: a waiting client polls this long before it sleeps), BARTRT_SVC_MAXCLIENTS (256 slots,
 *       1024 at most: one per worker process = per chain).  Which profiles share a launch depends on
 *       the processes' pace; the RESULT does not: the kernel variant is chosen for the number of
 *       registered workers (BARTRT_SVC_KERNEL_WALKERS pins it) and every kernel computes a walker
 *       independently of the others in its launch, so a chain's spectra are the same bits every run.
 *       Runs of fewer than five chains (BARTRT_NCHAINS, else the MPI world size of the spawned
 *       workers) take the `ipc` reading unless BARTRT_SHARE_MODE says otherwise: it is the faster
 *       one there.
 *   ipc  every process runs its own engine; the first uploads the grid, the others map that HBM
 *       allocation through a HIP IPC handle published in a POSIX shared-memory segment
 *       (HSA_ENABLE_IPC_MODE_LEGACY=0 must be in the environment where the host driver supports
 *       dmabuf IPC only).  What "--no-service" in bartrt_init's argv (a caller that needs the whole
 *       API in its own process: bart_amd.engine) turns `service` into.
 *   off  the key is ignored.
 * BARTRT_SERVICE=1 asks for the service without the key.
 * bartrt_get_share: *shared = 1 if this process's grid is shared in either way, *owner = 1 if this
 * process holds the allocation (ipc) / the engine (service). */
int bartrt_get_share(int *shared, int *owner);
/* *mode = 0 engine of its own, 1 chain-service client, 2 client that also owns the service;
 * *owner_pid, this process's *slot, *nclients registered.  Any pointer may be NULL. */
int bartrt_get_service(int *mode, int *owner_pid, int *slot, int *nclients);
/* Chain service only: dispatcher rounds so far, profiles served by them, rounds that held every
 * registered client (nprofiles / nlaunches = the mean batch). */
int bartrt_get_service_stats(unsigned long long *nlaunches, unsigned long long *nprofiles,
                             unsigned long long *nfull);
/* ... and the rounds whose slots were not consecutive (a worker in the middle of the slot
 * range missed the round): served by one launch all the same, through a gather / scatter pair. */
int bartrt_get_service_gathered(unsigned long long *ngathered);

/* Prefetched preparation.  Names the profile batch of the bartrt_run_transit_batch_dev call
 * AFTER the next one: the next call's RT launch prepares that batch's layer records
 * (hydrostatic radii, densities, interpolation weights) in extra workgroups of its own grid,
 * and the call that follows -- if it is made with exactly this buffer and walker count --
 * starts on its RT kernel directly (one launch and ~8 us of dependent latency less per
 * batch).  THE CALLER'S PROMISE: the named buffer is complete, and stays unchanged, from the
 * next call on.  True of batches resident in HBM (a grid or population of models evaluated
 * chunk by chunk; bench.py); not of an MCMC step whose proposal depends on the previous
 * step's spectra -- do not prefetch there.  Without a matching call the records are simply
 * dropped.  Table path of the eclipse geometry; other engines ignore the request.  Results
 * are bit-identical with and without.  nwalkers = 0 withdraws a request. */
int bartrt_prefetch_profiles_dev(const double *d_prof_next, int nwalkers);

/* ---- batched / device-resident variants (same arithmetic) ------------- */

/* nwalkers profiles -> nwalkers spectra; ok[w] = 0 marks a profile the
 * engine cannot evaluate (non-finite or non-positive temperature).  With
 * ok == NULL (and through bartrt_run_transit) such a profile is an error
 * (BARTRT_EINVAL) instead.  spec is [nwalkers][nwave_local]. */
int bartrt_run_transit_batch(const double *prof, int nwalkers, int nprof,
                             double *spec, int nwave, unsigned char *ok);

/* Same with HBM-resident buffers, asynchronous on `stream` (a hipStream_t;
 * NULL = the engine's own non-blocking stream, which nothing else is ordered with: a
 * caller that works on the device's default stream passes hipStreamLegacy, (hipStream_t)1,
 * as bart_amd/engine.py does for torch's default stream).  d_spec is [nwalkers][local samples]. */
int bartrt_run_transit_batch_dev(const double *d_prof, int nwalkers,
                                 double *d_spec, unsigned char *d_ok,
                                 void *stream);

/* ---- per-step callable on the device (BARTfunc.py:309-399) ------------- */

/* One-time setup of the input/output converters around the engine:
 * PT model "line" (code/PT.py:589-701) with PTargs = {R_star[m], T_star[K],
 * T_int[K], sma[m], g[cm s-2]} (BARTfunc.py:204-208), the base abundances
 * abund[L][S] (atm order), indices of the fitted molecules, and the
 * per-filter windows/weights produced by wine.resample
 * (code/wine.py:127-174): for filter f, idx0[f] = first spectrum index,
 * npts[f] = number of samples, weights/starflux concatenated in that order.
 * pttype: 0 "line", 1 "iso".  tint_thorngren: T_int from Thorngren et al.
 * 2019 (PT.py:680-685) instead of ptargs5[2].
 * rprs = Rp/Rs.  solution: 0 eclipse (divide by stellar flux, times rprs^2),
 * 1 transit / 2 direct (no star division) (BARTfunc.py:386-396). */
int bartrt_step_setup(const double *ptargs5, int tint_thorngren, int pttype,
                      double tmin, double tmax,
                      const double *abund, int nmolfit, const int *imol,
                      int nfilters, const int *idx0, const int *npts,
                      const double *nifilter, const double *istarfl,
                      double rprs, int solution);

/* Energy-balance rejection (BARTfunc.py:366-383): reject a walker when
 * trapz(spectrum, wn) * e_fac > e_in  (e_fac = 4 (Rp*100)^2). */
int bartrt_step_set_ebalance(int on, double e_in, double e_fac);

/* Declares the per-walker parameters BARTfunc places between the T(p) parameters
 * and the abundance factors (BARTfunc.py:350-360), each 0 or 1, in this order:
 * planet radius at the reference pressure (km; what trm.set_radius takes), log10
 * of the cloud-top pressure (bar; trm.set_cloudtop), Rayleigh scattering value
 * (trm.set_scattering(1, value); present but unused with the polarisability
 * flavour, flag 2, which is set once with bartrt_set_scattering).  They then act
 * per walker inside the batch instead of through the engine-wide setters. */
int bartrt_step_set_extras(int nrad, int ncloud, int nray);

/* The reference's worker keeps going when its T(p) model raises ValueError: the
 * profile array still holds the previous step's temperatures and is range-checked
 * and used as it is (BARTfunc.py:318-330, marked FINDME there).  on = 1 reproduces
 * that: walker w of every call is chain w, and a parameter set the model rejects
 * is evaluated with chain w's last generated profile (zeros before the first one,
 * i.e. rejected by the temperature bounds).  on = 0 (default): such a walker is
 * rejected with status 1. */
int bartrt_step_set_carry(int on);

/* params[nwalkers][npars] (npars = nPT + extras + nmolfit; nPT = 5 for "line", 1 for
 * "iso") -> bandflux[nwalkers][nfilters]; rejected walkers get -1 in every
 * band (BARTfunc.py:327-330,339-344,378-383).  status[w]: 0 ok, 1 bad
 * temperature, 2 bad abundance, 3 energy balance.  status may be NULL. */
int bartrt_step_batch(const double *params, int nwalkers, int npars,
                      double *bandflux, int *status);
/* The retrieval's driver loop, native: differential-evolution MCMC (snooker = 0,
 * the reference's `walk = demc`) or its snooker variant (snooker = 1) over
 * nchains chains, all of them evaluated by one batched model call per
 * iteration, nsteps iterations.  Arguments are the keys of the reference's
 * [MCMC] section (examples/demo/BART_eclipse.cfg:43-102): params (start),
 * pmin, pmax, stepsize (0 = fixed) of length npars = nPT + nmolfit as in
 * bartrt_step_batch; data, uncert of length ndata = nfilters.  Proposals
 * outside [pmin, pmax] and models rejected by the worker are refused.
 * chain[nchains][nsteps][npars], chisq[nchains][nsteps]; naccept (accepted
 * proposals) and nbad[4] (models rejected with status 1, 2, 3 in nbad[1..3]) may
 * be NULL. */
int bartrt_mcmc_run(int nchains, int npars, long nsteps, const double *params, const double *pmin,
                    const double *pmax, const double *stepsize, int ndata, const double *data,
                    const double *uncert, int snooker, unsigned long long seed, double *chain,
                    double *chisq, long *naccept, long *nbad);
/* Device-resident form; d_status and d_spec ([nwalkers][nwave]) may be NULL. */
int bartrt_step_batch_dev(const double *d_params, int nwalkers, int npars,
                          double *d_bandflux, int *d_status, double *d_spec,
                          void *stream);
/* The two converters on their own, for engines sharded by wavenumber block:
 * profiles -> bartrt_run_transit_batch_dev on each shard -> all-gather of the
 * spectra (caller, RCCL) -> bandflux on the reassembled [nwalkers][nwave]. */
int bartrt_step_profiles_dev(const double *d_params, int nwalkers, int npars,
                             double *d_prof, int *d_status, void *stream);
int bartrt_step_bandflux_dev(const double *d_spec_full, int nwalkers,
                             int *d_status, double *d_bandflux, void *stream);

/* ---- introspection ----------------------------------------------------- */
const char *bartrt_last_error(void);
/* Twelve hex digits: a hash of the CODE the eclipse RT launch is made of (device
 * code objects and host text of the RT translation units), fixed at build time.
 * Never fails, needs no GPU.  Profiler figures under profiles/ carry the id of the
 * library they were measured on; bench.py quotes them only under the same id. */
const char *bartrt_build_id(void);
/* Which eclipse kernel the library's measured table (csrc/kernel_table.inc, written by
 * tools/tune_kernels.py) names for a launch of `columns` 64-sample columns (walkers x
 * ceil(samples / 64)) with `nmol` table molecules under the default conventions (rule 1,
 * `cut slant`, five ray angles): "single", "rows4" / "rows8" / "rows16" / "rows32" (layers per
 * step of the layer-parallel walk), "adj8" / "adj16" (rows on adjacent lanes).  Needs no GPU and
 * no engine; the returned string is static. */
const char *bartrt_kernel_choice(int nmol, long columns);
int bartrt_get_nlayers(void);
int bartrt_get_nspecies(void);
int bartrt_get_nprof(void);                 /* (S+1)*L */
int bartrt_get_local_range(int *lo, int *hi); /* shard's [lo,hi) of the grid */
int bartrt_get_species(char *buf, int buflen); /* space-separated names */
int bartrt_get_pressure(double *out, int n);   /* barye, atm order */
/* optical depth of the latest HOST-buffer call's profile: tau[nwave_local][L], layer
 * index 0 = top (the tau.dat convention read by code/cf.py:68-94); last[i] = the layer
 * where the column of sample i ended.  bartrt_get_tau serves a single-profile call
 * (bartrt_run_transit, the reference's shape); after a batch call it is an ERROR, not the
 * first walker's or an older call's values: name the walker with bartrt_get_tau_of (the
 * profile is re-run with the output enabled).  Device-buffer calls keep no profile. */
int bartrt_get_tau(double *tau, int *last, int nwave, int nlayers);
int bartrt_get_tau_of(int walker, double *tau, int *last, int nwave, int nlayers);

/* Outputs of the standalone run (reference: `transit -c cfg`, code/bestFit.py:421-427,
 * code/cf.py:46-64).  get_atm_profile: the (S+1)*L profile array of the atmosphere
 * file itself.  get_radius: hydrostatic radii (cm, atm order) of the last run's
 * first profile.  get_intensity: I[nangles][nwave_local] of the last
 * single-profile eclipse run (the `outintens` content). */
int bartrt_get_atm_profile(double *prof, int nprof);
int bartrt_get_radius(double *rad, int nlayers);
int bartrt_get_nangles(void);
int bartrt_get_angles(double *deg, int n);
int bartrt_get_intensity(double *intens, int nangles, int nwave);
int bartrt_get_intensity_of(int walker, double *intens, int nangles, int nwave);  /* as bartrt_get_tau_of */

/* Line-by-line engines only (cfg has `linedb`, no `opacityfile`): the Voigt
 * extinction of one profile, ext[nlayers][nwave_local] in cm-1, atm layer order. */
int bartrt_get_lbl_extinction(const double *prof, int nprof, double *ext,
                              int nlayers, int nwave);

/* Diagnostics, no engine needed: the line-by-line kernels' Voigt function
 * K(x, y) = Re w(x + i y) (x >= 0: distance from the line centre in Doppler widths
 * / sqrt(ln 2); y > 0: Lorentz / Doppler width ratio) on n host (x, y) pairs. */
int bartrt_voigt(const double *x, const double *y, double *k, long n);

/* HIP-event timing of the RT kernel launches (bench.py's roofline leg).
 * begin resets; end returns accumulated device ms and launch count. */
int bartrt_timing_begin(void);
/* the same with events around every stride-th launch only: a pair of event records costs
 * the stream about 5 us, 7 % of a ten-walker step; end returns the sampled launches' sum
 * and their count */
int bartrt_timing_begin_sampled(int stride);
int bartrt_timing_end(double *kernel_ms, int *nlaunch);
/* Diagnostics for the byte model of bench.py: between begin and end every eclipse
 * launch records how many layers each wave walked before all its lanes passed
 * `toomuch`.  end returns the record of the LAST launch: walked[nwalkers][ncolumns]
 * (columns of wn_per_column consecutive wavenumbers, as the launched kernel tiles
 * the grid) and that kernel's name. */
int bartrt_walked_begin(void);
int bartrt_walked_end(int *walked, int cap, int *nwalkers, int *ncolumns, int *wn_per_column,
                      char *kernel, int kernel_len);
/* Shapes outside the ahead-of-time set (seven and more table molecules with two cross-section files under the default
 * spline, nine and more molecules, ten and more ray angles) are instantiated from the same kernel templates when they
 * are first launched (hiprtc; cached under BARTRT_RTC_CACHE, default ~/.cache/bartrt; BARTRT_RTC=0 or a machine without
 * libhiprtc: the generic kernel serves them, 3-10x slower).  *available: a compiler is at hand; kernels compiled /
 * loaded from the disk cache / failed by this process so far, and the seconds spent compiling.  bartrt_walked_end's
 * kernel name carries " [instantiated at run time]" for such a launch (and " [prepares its own walkers]" where a
 * few-walker launch built its layer records in its own prologue instead of a preparation launch: BARTRT_FOLD=0 off).  Pointers may be NULL. */
int bartrt_get_rtc_stats(int *available, int *compiled, int *from_disk, int *failed, double *compile_seconds);
/* Compiles bartrt::<expr> (a template-id of the kernel headers, e.g. "rt_eclipse_simpson_slant<5, 9, 4, true, 1>")
 * for gfx950 and discards the result: *code_bytes = the code object's size.  Needs no GPU -- a check that the embedded
 * sources and the compiler at hand agree.  BARTRT_ENOTSUP without a compiler or on a compile error (bartrt_last_error). */
int bartrt_rtc_compile(const char *expr, int ilp, long *code_bytes);
/* algorithmic bytes one launch of the RT kernel moves for `nwalkers`
 * (2*L*W*M*8 + 2*L*W*8*ncia + (S+1)*L*8 + W*8 per spectrum) */
double bartrt_algorithmic_bytes(int nwalkers);

#ifdef __cplusplus
}
#endif
#endif
