"""Prefetched preparation (include/bartrt.h, bartrt_prefetch_profiles_dev): the RT launch of one
batch carries the next batch's prep_profiles in extra workgroups at the head of its grid.  The
spectra must not change by a bit -- through every specialised kernel (quad-layer with four and
eight rows, producer / consumer, single-wave under rules 0 and 1), when a prefetch is not
followed up, when the next batch is larger than any before (buffers re-sized), and with a
rejected profile in the prefetched batch (its flag reaches the caller's array)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("integ", [1, 0])
def test_prefetch_is_bit_identical_through_every_kernel(small_case, integ):
    import torch
    from bart_amd import engine, transit_module as trm
    from test_gpu_parity import walkers
    c = small_case                      # 777 samples: 13 columns per walker
    engine.init(c.tcfg)
    try:
        trm.set_integ(integ)
        for n in (1, 3, 30, 45, 90, 130):      # 8-row quad, quad, quad / single-wave, split (rule 0), single-wave
            nsets = 5
            d_prof = torch.from_numpy(walkers(c, n * nsets, seed=100 + n).reshape(nsets, n, -1)).cuda()
            ref = [engine.run_batch_dev(d_prof[s]).clone() for s in range(nsets)]
            for i in range(8):          # every call names the next batch
                out = engine.run_batch_dev(d_prof[i % nsets], next_prof=d_prof[(i + 1) % nsets])
                assert torch.equal(out, ref[i % nsets]), (n, i)
            # the named batch is not the one that follows: its records are dropped
            engine.run_batch_dev(d_prof[0], next_prof=d_prof[1])
            assert torch.equal(engine.run_batch_dev(d_prof[3]), ref[3]), n
            # same buffer, other walker count: not a match either
            if n > 1:
                # (its own reference: one walker less may be another kernel's launch -- csrc/kernel_table.inc)
                fewer = d_prof[1][: n - 1].contiguous()
                want = engine.run_batch_dev(fewer).clone()
                engine.run_batch_dev(d_prof[0], next_prof=d_prof[1])
                assert torch.equal(engine.run_batch_dev(fewer), want), n
            # a request withdrawn
            trm.check(trm.lib().bartrt_prefetch_profiles_dev(C.c_void_p(d_prof[2].data_ptr()), 0))
            assert torch.equal(engine.run_batch_dev(d_prof[2]), ref[2])
        # the next batch is larger than anything seen so far: the record buffers grow under the request
        small = torch.from_numpy(walkers(c, 4, seed=1)).cuda()
        big = torch.from_numpy(walkers(c, 700, seed=2)).cuda()
        want_small, want_big = engine.run_batch_dev(small).clone(), None
        out = engine.run_batch_dev(small, next_prof=big)
        assert torch.equal(out, want_small)
        got_big = engine.run_batch_dev(big, next_prof=small).clone()
        assert torch.equal(engine.run_batch_dev(small), want_small)
        want_big = engine.run_batch_dev(big)
        assert torch.equal(got_big, want_big)
    finally:
        trm.free_memory()


def test_prefetched_batch_reports_its_rejected_profiles(small_case):
    import torch
    from bart_amd import engine, transit_module as trm
    from test_gpu_parity import walkers
    c = small_case
    engine.init(c.tcfg)
    try:
        n = 60
        a = walkers(c, n, seed=5)
        b = walkers(c, n, seed=6)
        b[7, 3] = np.nan                # a temperature of walker 7
        b[41, 11] = -5.0
        d_a, d_b = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
        spec = torch.empty((n, 777), dtype=torch.float64, device="cuda")
        ok = torch.full((n,), 9, dtype=torch.uint8, device="cuda")

        def run(d_prof, nxt=None):
            if nxt is not None:
                trm.check(trm.lib().bartrt_prefetch_profiles_dev(C.c_void_p(nxt.data_ptr()), n))
            trm.check(trm.lib().bartrt_run_transit_batch_dev(C.c_void_p(d_prof.data_ptr()), n, C.c_void_p(spec.data_ptr()),
                                                            C.c_void_p(ok.data_ptr()), None))
            torch.cuda.synchronize()
            return spec.clone(), ok.cpu().numpy().copy()

        ref_spec, ref_ok = run(d_b)
        assert ref_ok[7] == 0 and ref_ok[41] == 0 and ref_ok.sum() == n - 2
        run(d_a, nxt=d_b)
        got_spec, got_ok = run(d_b)      # prepared by the previous launch
        assert np.array_equal(got_ok, ref_ok)
        good = torch.from_numpy(ref_ok.astype(bool)).cuda()
        assert torch.equal(got_spec[good], ref_spec[good])
    finally:
        trm.free_memory()


def test_device_calls_are_ordered_with_torchs_current_stream(small_case):
    """run_batch_dev enqueues on torch's CURRENT stream -- also when that is the default (null)
    stream, whose handle 0 the C ABI would read as "the engine's own stream": a torch operation
    issued right after the call (a clone here; the all-gather of the sharded path) must see the
    finished spectra without any synchronisation in between."""
    import torch
    from bart_amd import engine, transit_module as trm
    from test_gpu_parity import walkers
    c = small_case
    engine.init(c.tcfg)
    try:
        d_prof = torch.from_numpy(walkers(c, 1500, seed=3)).cuda()      # a launch of a few hundred microseconds
        want = engine.run_batch_dev(d_prof)
        torch.cuda.synchronize()
        want = want.clone()
        out = torch.zeros_like(want)
        for _ in range(5):
            out.zero_()
            got = engine.run_batch_dev(d_prof, out).clone()     # no synchronisation before the copy
            assert torch.equal(got, want)
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            out.zero_()
            got = engine.run_batch_dev(d_prof, out).clone()
        side.synchronize()
        assert torch.equal(got, want)
    finally:
        trm.free_memory()
