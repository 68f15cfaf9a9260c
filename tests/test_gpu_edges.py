"""Edge cases of the call boundary on the device: empty batches, malformed
shapes, batches in which every walker is rejected, non-finite parameters, and a
batch far larger than anything the engine has sized its buffers for."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_empty_batch_and_malformed_shapes(small_case):
    from bart_amd import engine, transit_module as trm
    from test_gpu_parity import walkers
    c = small_case
    engine.init(c.tcfg)
    try:
        n, npr = trm.get_no_samples(), engine.nprof()
        spec, ok = engine.run_batch(np.zeros((0, npr)), want_ok=True)
        assert spec.shape == (0, n) and ok.shape == (0,)
        prof = walkers(c, 1, seed=2)[0]
        good = trm.run_transit(prof, n)
        with pytest.raises(trm.TransitError, match="profile length"):
            trm.run_transit(prof[:-1], n)
        with pytest.raises(trm.TransitError, match="nwave"):
            trm.run_transit(prof, n - 1)
        lib = trm.lib()
        out = np.zeros(n)
        assert lib.bartrt_run_transit_batch(None, 1, npr, trm._ptr(out), n, None) < 0
        assert lib.bartrt_run_transit_batch(trm._ptr(prof), -1, npr, trm._ptr(out), n, None) < 0
        bad = prof.copy(); bad[3] = np.nan                    # a temperature
        with pytest.raises(trm.TransitError, match="temperature"):
            trm.run_transit(bad, n)
        # refused calls leave the engine usable and its results unchanged
        assert np.array_equal(trm.run_transit(prof, n), good)
    finally:
        trm.free_memory()


def test_buffers_grow_with_the_batch(small_case):
    """One walker, then 3 000 (device buffers and the pinned staging area are
    re-sized), then one again: the small calls agree bit for bit and the big batch
    holds the same spectrum in every row it was given the same profile for."""
    from bart_amd import engine, transit_module as trm
    from test_gpu_parity import walkers
    c = small_case
    engine.init(c.tcfg)
    try:
        p = walkers(c, 3, seed=9)
        first = engine.run_batch(p[:1])
        big = engine.run_batch(np.tile(p, (1000, 1)))
        assert big.shape == (3000, first.shape[1])
        for r in range(3):
            rows = big[r::3]
            assert np.array_equal(rows, np.broadcast_to(rows[0], rows.shape))
        np.testing.assert_allclose(big[0], first[0], rtol=1e-12)   # (different kernel at this batch size)
        assert np.array_equal(engine.run_batch(p[:1]), first)
    finally:
        trm.free_memory()


def test_step_with_every_walker_rejected_and_nonfinite_parameters(tmp_path):
    from bart_amd import BARTfunc, synthcfg
    base = np.array([-2.0, 0.0, 1.0, 0.0, 0.98, -0.5])
    case, cfg = synthcfg.make_worker_case(str(tmp_path), nwave=600)
    w = BARTfunc.Worker(BARTfunc.WorkerConfig.from_cfg(cfg))
    try:
        good = w.step(base)[0]
        assert np.all(good > 0)
        assert w.step(np.zeros((0, 6))).shape == (0, w.nfilters)
        hot = base.copy(); hot[4] = 5.0                       # T(p) far above Tmax: rejected
        band = w.step(np.array([hot, hot, hot]))
        assert np.all(band == -1.0) and w.nbad[1] == 3
        for bad in (np.nan, np.inf, -np.inf):
            q = np.array([base, base, base])
            q[1, 2] = bad
            band = w.step(q)
            assert np.all(band[1] == -1.0), bad
            assert np.array_equal(band[0], good) or np.allclose(band[0], good, rtol=1e-12)
            assert np.allclose(band[2], good, rtol=1e-12)
        q = np.array([base, base])
        q[0, 5] = np.nan                                      # abundance factor
        band = w.step(q)
        assert np.all(band[0] == -1.0) and np.allclose(band[1], good, rtol=1e-12)
        assert np.allclose(w.step(base)[0], good, rtol=0, atol=0)
    finally:
        w.close()
