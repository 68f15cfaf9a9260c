"""Batched / device-resident front end of libbartrt.so.

``transit_module`` keeps the reference's one-profile-per-call shape; this
module adds what the MI355X build needs on top of the same engine: walker
batches, HBM-resident tensors (torch is used only as the owner of device
memory, streams and the RCCL process group), wavenumber-block sharding with an
all-gather that reassembles each spectrum, and the per-step
converters of code/BARTfunc.py:309-399 on the device.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import transit_module as trm

_check, _ptr = trm.check, trm._ptr


def init(tcfg: str, shard: tuple[int, int] | None = None, device: int | None = None, kernel_by: str | None = None) -> None:
    """An engine IN THIS PROCESS (this module's batched / device-resident calls need one): with `shareOpacity`
    in the cfg the grid is shared through HIP IPC, never through the chain service (include/bartrt.h,
    bartrt_get_share) -- the service is for the reference's one-profile-per-process workers."""
    argv = ["transit", "-c", tcfg, "--no-service"]
    if shard is not None:
        argv += ["--shard", str(shard[0]), str(shard[1])]
    if device is not None:
        argv += ["--device", str(device)]
    trm.transit_init(len(argv), argv)
    if kernel_by is not None:
        # 'whole': a sharded engine's blocks are the unsharded spectrum bit for bit; 'local' (default): the kernel
        # variant fits the block (include/bartrt.h, bartrt_set_kernel_by)
        trm.set_kernel_by(kernel_by)


def nlayers() -> int:
    return _check(trm.lib().bartrt_get_nlayers())


def nspecies() -> int:
    return _check(trm.lib().bartrt_get_nspecies())


def nprof() -> int:
    return _check(trm.lib().bartrt_get_nprof())


def species() -> list[str]:
    buf = C.create_string_buffer(4096)
    _check(trm.lib().bartrt_get_species(buf, 4096))
    return buf.value.decode().split()


def pressure() -> np.ndarray:
    out = np.zeros(nlayers())
    _check(trm.lib().bartrt_get_pressure(_ptr(out), out.size))
    return out


def local_range() -> tuple[int, int]:
    lo, hi = C.c_int(), C.c_int()
    _check(trm.lib().bartrt_get_local_range(C.byref(lo), C.byref(hi)))
    return lo.value, hi.value


def run_batch(profiles: np.ndarray, want_ok: bool = False):
    """profiles [nwalkers, (S+1)*L] (host) -> spectra [nwalkers, local samples]."""
    prof = np.ascontiguousarray(profiles, np.double)
    prof = prof.reshape(-1, nprof())
    lo, hi = local_range()
    spec = np.zeros((prof.shape[0], hi - lo))
    ok = np.zeros(prof.shape[0], np.uint8)
    _check(trm.lib().bartrt_run_transit_batch(_ptr(prof), prof.shape[0], prof.shape[1],
                                              _ptr(spec), hi - lo, _ptr(ok)))
    return (spec, ok) if want_ok else spec


def get_tau(walker=None):
    """Optical depth of the latest host-buffer call's profile: (tau[W_local, L], last[W_local]),
    layer index 0 = top (the tau.dat convention, code/cf.py:68-94).  After a batch call the
    walker has to be named (include/bartrt.h, bartrt_get_tau_of)."""
    lo, hi = local_range()
    tau = np.zeros((hi - lo, nlayers()))
    last = np.zeros(hi - lo, np.int32)
    if walker is None:
        _check(trm.lib().bartrt_get_tau(_ptr(tau), _ptr(last), hi - lo, nlayers()))
    else:
        _check(trm.lib().bartrt_get_tau_of(int(walker), _ptr(tau), _ptr(last), hi - lo, nlayers()))
    return tau, last


def lbl_extinction(profile: np.ndarray) -> np.ndarray:
    """Line-by-line extinction [L, W_local] (cm-1, atm layer order) of one profile."""
    prof = np.ascontiguousarray(profile, np.double).ravel()
    lo, hi = local_range()
    ext = np.zeros((nlayers(), hi - lo))
    _check(trm.lib().bartrt_get_lbl_extinction(_ptr(prof), prof.size, _ptr(ext), ext.shape[0],
                                               ext.shape[1]))
    return ext


def voigt(x: np.ndarray, y: np.ndarray) -> np.ndarray:
    """Re w(x + i y) as the line-by-line kernels evaluate it (diagnostics; no engine needed)."""
    x, y = np.broadcast_arrays(np.asarray(x, np.double), np.asarray(y, np.double))
    xs, ys = np.ascontiguousarray(x).ravel(), np.ascontiguousarray(y).ravel()
    k = np.empty_like(xs)
    _check(trm.lib().bartrt_voigt(_ptr(xs), _ptr(ys), _ptr(k), xs.size))
    return k.reshape(x.shape)


# ---- device-resident (torch tensors own the memory) ----------------------
def _stream_ptr(stream=None):
    """The HIP stream the library is to enqueue on: torch's current stream.  torch's DEFAULT stream
    is the null stream (handle 0), which the C ABI reads as "the engine's own stream"
    (include/bartrt.h) -- a different, non-blocking stream that torch's next operation (a clone, a
    collective) would not wait for.  The default stream therefore travels as hipStreamLegacy (1),
    HIP's explicit handle for it."""
    import torch
    s = stream if stream is not None else torch.cuda.current_stream()
    return C.c_void_p(s.cuda_stream or 1)


def run_batch_dev(d_prof, d_spec=None, stream=None, next_prof=None):
    """d_prof: float64 CUDA tensor [nwalkers, (S+1)*L]; returns [nwalkers, W_local]
    on the same device.  Asynchronous on torch's current stream.

    ``next_prof``: the batch of the NEXT call, if it is known and already complete in HBM
    (include/bartrt.h, bartrt_prefetch_profiles_dev): this call's RT launch prepares its layer
    records on the side and the next call skips its preparation launch.  Bit-identical results."""
    import torch
    assert d_prof.is_cuda and d_prof.dtype == torch.float64 and d_prof.is_contiguous()
    n = d_prof.shape[0]
    if next_prof is not None:
        assert next_prof.is_cuda and next_prof.dtype == torch.float64 and next_prof.is_contiguous()
        _check(trm.lib().bartrt_prefetch_profiles_dev(C.c_void_p(next_prof.data_ptr()), next_prof.shape[0]))
    lo, hi = local_range()
    if d_spec is None:
        d_spec = torch.empty((n, hi - lo), dtype=torch.float64, device=d_prof.device)
    _check(trm.lib().bartrt_run_transit_batch_dev(
        C.c_void_p(d_prof.data_ptr()), n, C.c_void_p(d_spec.data_ptr()), None,
        _stream_ptr(stream)))
    return d_spec


def run_batch_sharded(d_prof, group=None, force=False):
    """Every rank holds one wavenumber block of the tables and the full (tiny)
    profile batch; each computes spec[nwalkers, W/G] and one RCCL all-gather
    reassembles spec[nwalkers, W] on every rank.  A group of one
    skips the collective unless ``force`` (the single-GPU test of the RCCL path)."""
    import torch.distributed as dist
    local = run_batch_dev(d_prof)
    world = dist.get_world_size(group)
    if world == 1 and not force:
        return local
    return allgather_blocks(local, group, total=trm.get_no_samples())


def block_sizes(total, world):
    """Samples per rank under the engine's integer split (Engine::setup): W*(r+1)//n - W*r//n."""
    return [total * (r + 1) // world - total * r // world for r in range(world)]


def pad_block(local, wmax):
    """A rank's block [n, W_r] as the all-gather sends it: [n, wmax], zero-padded on the right."""
    import torch
    if local.shape[1] == wmax:
        return local.contiguous()
    send = torch.zeros((local.shape[0], wmax), dtype=local.dtype, device=local.device)
    send[:, :local.shape[1]] = local
    return send


def reassemble_blocks(out, n, sizes):
    """The all-gather's receive buffer [world * n, wmax] (rank-major) -> spectra [n, sum(sizes)]."""
    import torch
    world, wmax = len(sizes), out.shape[1]
    o = out.view(world, n, wmax)
    if min(sizes) == wmax:
        return o.permute(1, 0, 2).reshape(n, world * wmax)
    return torch.cat([o[r, :, :sizes[r]] for r in range(world)], dim=1)


def allgather_blocks(local, group=None, total=None, async_op=False, out=None):
    """local [n, W_r] on rank r (block sizes may differ by one sample) ->
    [n, sum_r W_r] on every rank.  With ``total`` (the full sample count) the
    block sizes follow from the engine's integer split W*r//n and ONE collective
    per call is issued; without it the sizes are exchanged first.

    ``async_op=True`` returns ``(work, finish)``: the collective is enqueued on
    RCCL's stream behind the kernels already queued on the current stream, the
    current stream is NOT made to wait, and ``finish()`` (wait + reassembly) is
    called when the spectra are needed -- so the next batch's kernels overlap
    the xGMI traffic of this one.  ``out`` may supply the [world*n, wmax]
    receive buffer (double buffering)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    n = local.shape[0]
    if total is not None:
        sizes = block_sizes(total, world)
        assert sizes[dist.get_rank(group)] == local.shape[1]
    else:
        sizes = [torch.zeros(1, dtype=torch.int64, device=local.device) for _ in range(world)]
        dist.all_gather(sizes, torch.tensor([local.shape[1]], dtype=torch.int64,
                                            device=local.device), group=group)
        sizes = [int(s.item()) for s in sizes]
    wmax = max(sizes)
    send = pad_block(local, wmax)
    if out is None:
        out = torch.empty((world * n, wmax), dtype=local.dtype, device=local.device)
    if local.is_cuda and dist.get_backend(group) == "gloo":
        # Device blocks under a process group that moves host memory only (two ranks on ONE GPU, which RCCL
        # refuses -- tests/test_gpu_two_ranks.py; a node without xGMI): the block goes through pinned host memory.
        # The collective must not start before the kernels that write `local` have finished: the stream is waited
        # for here (RCCL orders that on the device; gloo cannot).
        h_send = torch.empty(send.shape, dtype=send.dtype, pin_memory=True)
        h_send.copy_(send, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        h_out = torch.empty(out.shape, dtype=out.dtype, pin_memory=True)
        hwork = dist.all_gather_into_tensor(h_out, h_send, group=group, async_op=async_op)

        def finish_host():
            if hwork is not None:
                hwork.wait()
            out.copy_(h_out, non_blocking=True)      # on the current stream, ahead of the reassembly
            return reassemble_blocks(out, n, sizes)

        return (hwork, finish_host) if async_op else finish_host()
    work = dist.all_gather_into_tensor(out, send, group=group, async_op=async_op)

    def finish():
        if work is not None:
            work.wait()                      # the current stream waits; the host does not
        return reassemble_blocks(out, n, sizes)

    return (work, finish) if async_op else finish()


class GatherPipeline:
    """Bucketed, double-buffered all-gather of per-step wavenumber blocks.

    Each step writes its local block spec[nwalkers, W_r] into ``slot(i)``; every
    ``group`` steps one collective carries the whole bucket (fewer, larger
    transfers over xGMI, and one cross-stream wait per bucket instead of one per
    step on the compute stream), issued asynchronously on RCCL's stream while the
    next bucket's kernels run.  ``submit(i)`` returns the reassembled spectra
    [steps, nwalkers, W] of the bucket whose buffers step i+1 is about to reuse (or
    None); ``drain()`` flushes what is still in flight."""

    def __init__(self, nwalkers, wlocal, total, steps_per_bucket, device, pg=None, dtype=None):
        import torch
        import torch.distributed as dist
        self.n, self.wl, self.total, self.G, self.pg = nwalkers, wlocal, total, max(1, steps_per_bucket), pg
        world = dist.get_world_size(pg)
        wmax = max(block_sizes(total, world))
        dtype = dtype or torch.float64
        self.local = [torch.empty((self.G, nwalkers, wlocal), dtype=dtype, device=device) for _ in range(2)]
        self.recv = [torch.empty((world * self.G * nwalkers, wmax), dtype=dtype, device=device) for _ in range(2)]
        self.pending = [None, None]      # per buffer: (finish, steps it holds)
        self.filled = 0                  # steps written into the open bucket

    def slot(self, i):
        """Where step i's kernels write; first finishes the bucket that used these buffers."""
        b, k = (i // self.G) & 1, i % self.G
        done = None
        if k == 0:
            done = self._finish(b)
        self._done = done
        return self.local[b][k]

    def submit(self, i):
        b, k = (i // self.G) & 1, i % self.G
        self.filled = k + 1
        if k == self.G - 1:
            self._issue(b)
        return self._done

    def _issue(self, b):
        k = self.filled                  # a partly filled bucket (drain) sends its filled steps only
        world = self.recv[b].shape[0] // (self.G * self.n)
        _, fin = allgather_blocks(self.local[b][:k].view(k * self.n, self.wl), self.pg, total=self.total,
                                  async_op=True, out=self.recv[b][:world * k * self.n])
        self.pending[b] = (fin, k)
        self.filled = 0

    def _finish(self, b):
        if self.pending[b] is None:
            return None
        fin, k = self.pending[b]
        self.pending[b] = None
        return fin().view(k, self.n, self.total)

    def drain(self, last_step):
        """After the last submit(last_step): returns the remaining buckets in step order."""
        b = (last_step // self.G) & 1
        if self.filled:
            self._issue(b)
        outs = [self._finish(1 - b), self._finish(b)]
        return [o for o in outs if o is not None]


# ---- per-step converters --------------------------------------------------
def step_setup(ptargs5, tmin, tmax, abund, imol, idx0, npts, nifilter, istarfl,
               rprs, solution=0, pttype=0, tint_thorngren=False):
    a = lambda x, t: np.ascontiguousarray(x, t)
    ptargs5 = a(ptargs5 if ptargs5 is not None else np.zeros(5), np.double)
    abund = a(abund, np.double)
    imol = a(imol, np.int32)
    idx0 = a(idx0, np.int32)
    npts = a(npts, np.int32)
    nif = a(nifilter, np.double)
    star = a(istarfl, np.double) if istarfl is not None else None
    _check(trm.lib().bartrt_step_setup(
        _ptr(ptargs5), int(bool(tint_thorngren)), int(pttype), float(tmin), float(tmax),
        _ptr(abund), imol.size, _ptr(imol), idx0.size, _ptr(idx0), _ptr(npts), _ptr(nif),
        _ptr(star) if star is not None else None, float(rprs), int(solution)))


def step_set_extras(nrad: int, ncloud: int, nray: int) -> None:
    """Radius / cloud-top / scattering parameters (0 or 1 each) sit between the T(p)
    parameters and the abundance factors of every walker (BARTfunc.py:350-360)."""
    _check(trm.lib().bartrt_step_set_extras(int(nrad), int(ncloud), int(nray)))


def step_set_carry(on: bool) -> None:
    """The reference's carry-over of the previous temperature profile when the T(p)
    model raises ValueError (BARTfunc.py:318-324); walker w of every batch = chain w."""
    _check(trm.lib().bartrt_step_set_carry(int(bool(on))))


def step_set_ebalance(on, e_in, e_fac):
    _check(trm.lib().bartrt_step_set_ebalance(int(bool(on)), float(e_in), float(e_fac)))


def step_batch(params: np.ndarray, nfilters: int):
    p = np.ascontiguousarray(params, np.double)
    p = p.reshape(-1, p.shape[-1])
    band = np.zeros((p.shape[0], nfilters))
    status = np.zeros(p.shape[0], np.int32)
    _check(trm.lib().bartrt_step_batch(_ptr(p), p.shape[0], p.shape[1], _ptr(band), _ptr(status)))
    return band, status


def step_batch_dev(d_params, nfilters, want_spec=False, stream=None):
    import torch
    n, npars = d_params.shape
    dev = d_params.device
    band = torch.empty((n, nfilters), dtype=torch.float64, device=dev)
    status = torch.empty(n, dtype=torch.int32, device=dev)
    spec = None
    if want_spec:
        spec = torch.empty((n, trm.get_no_samples()), dtype=torch.float64, device=dev)
    _check(trm.lib().bartrt_step_batch_dev(
        C.c_void_p(d_params.data_ptr()), n, npars, C.c_void_p(band.data_ptr()),
        C.c_void_p(status.data_ptr()), C.c_void_p(spec.data_ptr()) if want_spec else None,
        _stream_ptr(stream)))
    return (band, status, spec) if want_spec else (band, status)


def step_batch_sharded(d_params, nfilters, group=None, stream=None, force=False):
    """Per-step callable on a wavenumber-sharded node: profiles on every rank,
    RT on the local block, all-gather, band integration on the full grid.
    ``force``: issue the collective on a group of one too (run_batch_sharded)."""
    import torch
    n, npars = d_params.shape
    dev = d_params.device
    prof = torch.empty((n, nprof()), dtype=torch.float64, device=dev)
    status = torch.empty(n, dtype=torch.int32, device=dev)
    _check(trm.lib().bartrt_step_profiles_dev(
        C.c_void_p(d_params.data_ptr()), n, npars, C.c_void_p(prof.data_ptr()),
        C.c_void_p(status.data_ptr()), _stream_ptr(stream)))
    spec = run_batch_sharded(prof, group, force=force)
    band = torch.empty((n, nfilters), dtype=torch.float64, device=dev)
    _check(trm.lib().bartrt_step_bandflux_dev(
        C.c_void_p(spec.data_ptr()), n, C.c_void_p(status.data_ptr()),
        C.c_void_p(band.data_ptr()), _stream_ptr(stream)))
    return band, status, spec


def timing_begin(stride: int = 1):
    """HIP events around every `stride`-th RT launch until timing_end()."""
    _check(trm.lib().bartrt_timing_begin_sampled(int(stride)))


def timing_end():
    ms, n = C.c_double(), C.c_int()
    _check(trm.lib().bartrt_timing_end(C.byref(ms), C.byref(n)))
    return ms.value, n.value


def walked_begin():
    """Record, for the eclipse launches that follow, how many layers each wave walks."""
    _check(trm.lib().bartrt_walked_begin())


def walked_end():
    """-> (walked[nwalkers, ncolumns], wavenumbers per column, kernel name) of the last launch."""
    n, nc, wpc = C.c_int(), C.c_int(), C.c_int()
    name = C.create_string_buffer(128)
    _check(trm.lib().bartrt_walked_end(None, 0, C.byref(n), C.byref(nc), C.byref(wpc), name, 128))
    # the record itself (the first call switched recording off; the buffer stays)
    out = np.zeros((n.value, nc.value), np.int32)
    if out.size:
        _check(trm.lib().bartrt_walked_end(_ptr(out), out.size, C.byref(n), C.byref(nc), C.byref(wpc), name, 128))
    return out, wpc.value, name.value.decode()


def algorithmic_bytes(nwalkers: int) -> float:
    return trm.lib().bartrt_algorithmic_bytes(int(nwalkers))
