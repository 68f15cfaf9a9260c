"""The in-process batched sampler: on an analytic model on CPU,
end to end on the GPU worker (recovering the parameters that generated the data)."""
import os

import numpy as np
import pytest

from bart_amd import sampler


def _linear_model(truth):
    x = np.linspace(0, 1, 12)

    def model(p):
        p = np.atleast_2d(p)
        out = p[:, :1] + p[:, 1:2] * x[None, :]
        out[p[:, 1] > 4.5] = -1.0            # a "rejected" region, like the -1 sentinels
        return out
    return model, model(np.array(truth))[0]


@pytest.mark.parametrize("walk", ["demc", "snooker"])
def test_recovers_a_gaussian_posterior(walk):
    model, data = _linear_model([1.0, 2.0])
    cfg = sampler.SamplerConfig(params=np.array([0.5, 1.0, 7.0]), pmin=np.array([-5.0, -5.0, 0.0]),
                                pmax=np.array([5.0, 5.0, 10.0]), stepsize=np.array([0.1, 0.1, 0.0]),
                                data=data, uncert=np.full(12, 0.05), nchains=8, numit=16000,
                                burnin=500, walk=walk, seed=3)
    res = sampler.run(model, cfg)
    post = res["chain"][:, 500:, :].reshape(-1, 3)
    assert np.all(post[:, 2] == 7.0)                      # stepsize 0: fixed
    assert abs(post[:, 0].mean() - 1.0) < 0.02 and abs(post[:, 1].mean() - 2.0) < 0.04
    # analytic posterior widths of a straight-line fit
    x = np.linspace(0, 1, 12)
    cov = np.linalg.inv(np.array([[12, x.sum()], [x.sum(), (x ** 2).sum()]]) / 0.05 ** 2)
    assert abs(post[:, 0].std() / np.sqrt(cov[0, 0]) - 1) < 0.25
    assert abs(post[:, 1].std() / np.sqrt(cov[1, 1]) - 1) < 0.25
    assert 0.05 < res["accept_rate"] < 0.7 and res["best_chisq"] < 1e-2
    assert np.all(res["grstat"] < 1.1)
    assert np.all(post[:, 1] <= 4.5)                      # never accepted a rejected model


def test_edge_cases_lone_chain_fixed_outside_box_shared_stepsize():
    """a single chain (no partner to difference with) moves by its jitter
    instead of raising; a fixed parameter outside [pmin, pmax] keeps its value and
    does not veto every proposal; a negative stepsize (MC3: shared parameter) is
    refused rather than treated as fixed."""
    model, data = _linear_model([1.0, 2.0])
    cfg = sampler.SamplerConfig(params=np.array([0.9, 2.1, 12.0]), pmin=np.array([-5.0, -5.0, 0.0]),
                                pmax=np.array([5.0, 5.0, 10.0]), stepsize=np.array([0.1, 0.1, 0.0]),
                                data=data, uncert=np.full(12, 0.05), nchains=1, numit=400,
                                burnin=10, walk="demc", seed=1)
    res = sampler.run(model, cfg)
    assert res["chain"].shape == (1, 400, 3) and np.all(res["chain"][:, :, 2] == 12.0)
    assert res["accept_rate"] > 0 and np.ptp(res["chain"][0, :, 0]) > 0
    cfg.nchains, cfg.numit = 6, 3000
    res = sampler.run(model, cfg)
    assert np.all(res["chain"][:, :, 2] == 12.0) and res["accept_rate"] > 0.05
    cfg.stepsize = np.array([0.1, -1.0, 0.0])
    with pytest.raises(ValueError, match="shared"):
        sampler.run(model, cfg)


def test_config_from_reference_style_cfg(tmp_path):
    p = tmp_path / "BART.cfg"
    p.write_text("[MCMC]\nparams = -2.0 0.0 1.0\npmin = -5 -2 -2\npmax = -1 1 1\n"
                 "stepsize = 0.01 0.01 0.0\ndata = 1e-3 2e-3\nuncert = 1e-5 1e-5\n"
                 "numit = 5e4\nnchains = 3\nburnin = 500\nwalk = snooker\ngrtest = True\n")
    c = sampler.SamplerConfig.from_cfg(str(p))
    assert c.numit == 50000 and c.nchains == 3 and c.walk == "snooker" and c.grtest
    assert np.array_equal(c.stepsize, [0.01, 0.01, 0.0])


@pytest.mark.gpu
def test_retrieval_end_to_end(tmp_path):
    """Synthetic eclipse depths generated with known parameters are fitted back
    through the whole device path (T(p) -> RT -> bands) by the batched sampler."""
    from bart_amd import BARTfunc, retrieve, synthcfg
    truth = np.array([-2.0, 0.0, 1.0, 0.0, 0.98, -0.5])
    case, cfg = synthcfg.make_worker_case(str(tmp_path), nwave=1200, params=tuple(truth))
    w = BARTfunc.Worker(BARTfunc.WorkerConfig.from_cfg(cfg))
    data = w.step(truth)[0]
    w.close()
    with open(cfg, "a") as f:
        f.write("pmin = -5.0 -2.0 -2.0 0.0 0.55 -9.0\npmax = -1.0 1.0 1.0 1.0 1.2 1.5\n")
        f.write("stepsize = 0.01 0.0 0.0 0.0 0.001 0.05\n")
        f.write("data = " + " ".join("%.10e" % d for d in data) + "\n")
        f.write("uncert = " + " ".join("%.10e" % (0.01 * d) for d in data) + "\n")
        f.write("numit = 4000\nnchains = 16\nburnin = 50\nwalk = demc\nseed = 5\n")
    res = retrieve.main(["-c", cfg, "--out", str(tmp_path / "out")])
    assert res["best_chisq"] < 1.0
    post = res["chain"][:, 100:, :]
    for i in (0, 4, 5):                                   # the three free parameters
        assert abs(np.median(post[:, :, i]) - truth[i]) < 4 * max(post[:, :, i].std(), 1e-3)
    assert os.path.exists(tmp_path / "out" / "output.npy")
    assert "models/s" in open(tmp_path / "out" / "MCMC.log").read()


@pytest.mark.gpu
def test_native_loop_agrees_with_python_loop(tmp_path):
    """bartrt_mcmc_run (C++ loop) and sampler.run (Python loop) on the same
    retrieval: different random streams, same posterior -- means within a few
    standard errors, widths within 20 %, comparable acceptance."""
    from bart_amd import BARTfunc, synthcfg
    truth = np.array([-2.0, 0.0, 1.0, 0.0, 0.98, -0.5])
    case, cfg = synthcfg.make_worker_case(str(tmp_path), nwave=600, params=tuple(truth))
    w = BARTfunc.Worker(BARTfunc.WorkerConfig.from_cfg(cfg))
    try:
        data = w.step(truth)[0]
        scfg = sampler.SamplerConfig(
            params=truth.copy(), pmin=np.array([-5.0, -2.0, -2.0, 0.0, 0.55, -9.0]),
            pmax=np.array([-1.0, 1.0, 1.0, 1.0, 1.2, 1.5]),
            stepsize=np.array([0.01, 0.0, 0.0, 0.0, 0.001, 0.05]), data=data, uncert=data * 0.01,
            nchains=8, numit=8 * 3000, burnin=300, walk="snooker", seed=5)
        a = sampler.run_native(w, scfg)
        b = sampler.run(w.step, scfg)
        free = a["free"]
        assert list(free) == [0, 4, 5] and a["chain"].shape == b["chain"].shape
        assert np.all(a["chain"][:, :, [1, 2, 3]] == truth[[1, 2, 3]])          # fixed parameters
        assert 0.05 < a["accept_rate"] < 0.6 and abs(a["accept_rate"] - b["accept_rate"]) < 0.1
        pa = a["chain"][:, 300:, :][:, :, free].reshape(-1, 3)
        pb = b["chain"][:, 300:, :][:, :, free].reshape(-1, 3)
        sd = pb.std(axis=0)
        assert np.all(np.abs(pa.mean(axis=0) - pb.mean(axis=0)) < 0.25 * sd)
        assert np.all(np.abs(pa.std(axis=0) / sd - 1.0) < 0.2)
        assert np.all(np.abs(pa.mean(axis=0) - truth[free]) < 3 * sd)
        assert a["grstat"] is not None and np.all(a["grstat"] < 1.2)
    finally:
        w.close()
