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


@pytest.mark.parametrize("mode", ["generic", "mono_occ", "mono_ilp", "split", "quad", "octo"])
def test_walked_layer_record(tmp_path, mode):
    """bartrt_walked_begin / _end (the bench's byte model): every eclipse kernel reports,
    per column of its own tiling, how many layers the wave walked -- all of them when
    `toomuch` never cuts, the cloud deck's depth when there is one, and at least the
    deepest `last` layer of the column's samples (the wave leaves once EVERY lane has
    passed toomuch) when it does."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import numpy as np, sys, json; sys.path.insert(0, %r)\n"
            "from bart_amd import engine, synth, transit_module as trm\n"
            "from test_gpu_parity import walkers\n"
            "out = {}\n"
            "for name, tm in (('open', 1e30), ('cut', 10.0)):\n"
            "    c = synth.make_case(%r + name, nlayers=60, nwave=300, toomuch=tm)\n"
            "    engine.init(c.tcfg); p = walkers(c, 3, seed=12)\n"
            "    engine.walked_begin(); engine.run_batch(p); w, wpc, kname = engine.walked_end()\n"
            "    trm.run_transit(p[1], 300); tau, last = engine.get_tau()\n"
            "    out[name] = dict(w=w.tolist(), wpc=wpc, kernel=kname, last=last.tolist())\n"
            "    if name == 'open':\n"
            "        trm.set_cloudtop(float(0.5 * (np.log10(c.press_bar[25]) + np.log10(c.press_bar[26]))))\n"
            "        engine.walked_begin(); engine.run_batch(p); w2, _, _ = engine.walked_end(); out['deck'] = w2.tolist()\n"
            "    trm.free_memory()\n"
            "print('RESULT' + json.dumps(out))\n" % (root, str(tmp_path) + "/"))
    env = dict(os.environ, BARTRT_KERNEL=mode, PYTHONPATH=os.path.join(root, "tests"))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT")][0][6:])
    wpc = out["open"]["wpc"]
    ncol = (300 + wpc - 1) // wpc
    w = np.array(out["open"]["w"])
    # (the quad-layer kernels' record is padded to whole workgroups: columns past the grid stay 0)
    assert w.shape[0] == 3 and w.shape[1] >= ncol and np.all(w[:, :ncol] == 60) and np.all(w[:, ncol:] == 0), \
        (out["open"]["kernel"], w)
    # deck between the layers 25 and 26 from the bottom: the column ends on k = 34, 35 layers
    assert np.all(np.array(out["deck"])[:, :ncol] == 60 - 25)
    w = np.array(out["cut"]["w"])
    last = np.array(out["cut"]["last"])                           # of walker 1
    for col in range(ncol):
        deepest = last[col * wpc:(col + 1) * wpc].max()
        assert deepest + 1 <= w[1, col] <= min(60, deepest + 1 + 8), (col, deepest, w[1, col])
    assert (w < 60).any()


def test_table_goes_up_slab_by_slab(small_case, monkeypatch):
    """bartrt_init uploads and re-lays the opacity grid out a bounded slab of (layer, temperature)
    planes at a time (the whole grid used to sit on the device twice).  Slabs of one
    plane, of seven (an uneven last slab) and the default give the same bits, unsharded and as a
    wavenumber block."""
    from bart_amd import engine, transit_module as trm
    from test_gpu_parity import walkers
    c = small_case
    profs = walkers(c, 5, seed=12)
    plane = 4 * 777 * 8          # four molecules x 777 samples
    ref = {}
    for slab in (None, plane, 7 * plane + 100):
        if slab is None:
            monkeypatch.delenv("BARTRT_INIT_SLAB_BYTES", raising=False)
        else:
            monkeypatch.setenv("BARTRT_INIT_SLAB_BYTES", str(slab))
        for shard in (None, (1, 3)):
            engine.init(c.tcfg, shard=shard)
            try:
                got = engine.run_batch(profs)
            finally:
                trm.free_memory()
            if slab is None:
                ref[shard] = got
            else:
                assert np.array_equal(got, ref[shard]), (slab, shard)
