"""`shareOpacity` (code/makecfg.py:106-107, BART.py:259-262): the reference's worker processes -- one per
chain -- keep ONE opacity grid.  Two readings here (include/bartrt.h, bartrt_get_share):
  * the chain service (default): N worker processes initialise on the same cfg, one of them owns the engine and a
    dispatcher thread, all of them post their profiles into shared-memory slots and one batched launch per MCMC
    step serves them (csrc/svc_core.hpp, csrc/svc.hip) -- the clients make no HIP call;
  * BARTRT_SHARE_MODE=ipc: every process its own engine, one of them uploads the grid to HBM and the others map
    that allocation through a HIP IPC handle (csrc/share.hip); every process computes the same bits as a process
    that holds its own copy."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


IPC = {"BARTRT_SHARE_MODE": "ipc"}


def expect(p, word):
    for line in p.stdout:
        if line.startswith(word + " "):
            return json.loads(line[len(word) + 1:])
    raise AssertionError("worker ended without '%s': %s" % (word, p.stderr.read()[-3000:]))


def start_worker(tcfg, rank, steps, out=None, extra=(), env=None):
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(env or {}))
    args = [sys.executable, os.path.join(ROOT, "tools", "mc3_child.py"), tcfg, str(rank), str(steps)]
    return subprocess.Popen(args + ([out] if out else []) + list(extra), stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                            stderr=subprocess.PIPE, text=True, env=e)


def run_workers(tcfg, n, steps, outdir, env=None, extra=None):
    """n tools/mc3_child.py processes in lockstep -> their (ready, done) reports and saved spectra."""
    procs = [start_worker(tcfg, r, steps, os.path.join(outdir, "w%d.npy" % r), (extra or {}).get(r, ()), env) for r in range(n)]
    try:
        ready = [expect(p, "ready") for p in procs]
        for p in procs:
            p.stdin.write("go\n"); p.stdin.flush()
        done = [expect(p, "done") for p in procs]
        for p in procs:
            p.stdin.write("bye\n"); p.stdin.flush()
        for p in procs:
            assert p.wait(timeout=120) == 0, p.stderr.read()[-3000:]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    spectra = [np.load(os.path.join(outdir, "w%d.npy" % r)) for r in range(n)]
    return ready, done, spectra


def test_three_processes_share_one_grid(tmp_path):
    from bart_amd import synth
    from oracle import rt_oracle as orc
    shared_case = synth.make_case(str(tmp_path / "s"), nlayers=40, nwave=700, extra_keys={"shareOpacity": ""})
    (tmp_path / "o1").mkdir(); (tmp_path / "o2").mkdir()
    ready, done, spec = run_workers(shared_case.tcfg, 3, 5, str(tmp_path / "o1"), env=IPC)
    assert all(r["shared"] and r["service"] == "engine" for r in ready)
    assert sum(r["owner"] for r in ready) == 1                  # one upload, two mappings
    # every process: the same bits for the common profile; its own chain's spectrum differs
    for s in spec[1:]:
        assert np.array_equal(s[0], spec[0][0])
    assert not np.array_equal(spec[1][1], spec[0][1])
    # ... and the bits of processes that each hold their own copy of the grid
    _, _, own = run_workers(shared_case.tcfg, 2, 1, str(tmp_path / "o2"), env={"BARTRT_SHARE_OPACITY": "0"})
    assert np.array_equal(own[0][0], spec[0][0]) and np.array_equal(own[1][1], spec[1][1])
    # ... and the oracle's numbers
    o = orc.OracleEngine(shared_case.tcfg)
    np.testing.assert_allclose(spec[0][0], o.run(shared_case.profiles().ravel()), rtol=1e-10,
                               atol=1e-12 * np.abs(spec[0][0]).max())
    # the name is gone from /dev/shm once the owner has freed the grid
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("bartrt_op_")]


def test_a_stale_segment_is_replaced(tmp_path):
    """A worker killed while it owned the grid leaves its name behind: the next run takes it over."""
    from bart_amd import synth
    c = synth.make_case(str(tmp_path / "s"), nlayers=30, nwave=300, extra_keys={"shareOpacity": ""})
    (tmp_path / "o").mkdir()
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "mc3_child.py"), c.tcfg, "0", "1"],
                         stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **IPC))
    assert p.stdout.readline().startswith("ready ")
    p.kill(); p.wait()
    assert [f for f in os.listdir("/dev/shm") if f.startswith("bartrt_op_")]
    # six processes find the dead owner's name at once: the takeover is serialised, ONE of them uploads
    ready, _, _ = run_workers(c.tcfg, 6, 1, str(tmp_path / "o"), env=IPC)
    assert sum(r["owner"] for r in ready) == 1 and all(r["shared"] for r in ready)
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("bartrt_op_")]


# ---- the chain service ----------------------------------------------------------------------------------------
# (test processes start, collect garbage and print at their own pace: a wide window, and the dispatcher waits for every
# registered worker within it, not only for those of the batch before -- whole batches, so that "the bits of a ten-walker
# batch" does not depend on who was late for a round)
SVC_ENV = {"BARTRT_SVC_WINDOW_US": "20000", "BARTRT_SVC_WAIT_ALL": "1"}


def own_engine_cfg(case, tmp_path):
    """The same inputs without the shareOpacity key: an engine of the test process's own."""
    txt = "".join(l for l in open(case.tcfg) if not l.startswith("shareOpacity"))
    p = str(tmp_path / "own.cfg")
    open(p, "w").write(txt)
    return p


def worker_profile(prof0, L, rank):
    mine = prof0.copy()
    mine[:L] = np.clip(mine[:L] + 20.0 * rank, 410.0, 2990.0)
    return mine


def test_ten_clients_are_one_ten_walker_batch(tmp_path):
    """Ten worker processes, the reference's shape (examples/WASP-12b/BART.cfg:113): one owner, nine clients without
    a HIP context, launches of ten -- and every worker gets the bits a ten-walker batch call computes."""
    from bart_amd import engine, synth, transit_module as trm
    case = synth.make_case(str(tmp_path / "s"), nlayers=40, nwave=700, extra_keys={"shareOpacity": ""})
    (tmp_path / "o").mkdir()
    ready, done, spec = run_workers(case.tcfg, 10, 6, str(tmp_path / "o"), env=SVC_ENV)
    assert all(r["shared"] for r in ready)
    assert sorted(r["service"] for r in ready) == ["client"] * 9 + ["owner"]
    assert sorted(r["slot"] for r in ready) == list(range(10))
    # a client never opened the GPU driver: one HIP context (and one grid) for the ten processes
    assert [r["hip_context"] for r in ready if r["service"] == "client"] == [False] * 9
    assert [d["hip_context"] for d, r in zip(done, ready) if r["service"] == "client"] == [False] * 9
    stats = max((d["service_stats"] for d in done), key=lambda s: s["launches"])
    assert stats["profiles"] >= 10 * 7 and stats["profiles"] / stats["launches"] > 4.0 and stats["full"] >= 6, stats
    # the ten-walker batch of an engine of this process's own
    engine.init(own_engine_cfg(case, tmp_path))
    try:
        n, L = trm.get_no_samples(), engine.nlayers()
        prof0 = case.profiles().ravel()
        batch = engine.run_batch(np.stack([worker_profile(prof0, L, r) for r in range(10)]))
        common = engine.run_batch(np.stack([prof0] * 10))
    finally:
        trm.free_memory()
    for r in range(10):
        assert np.array_equal(spec[r][1], batch[r]), r          # its own chain's spectrum (last step: all ten posted)
        assert np.array_equal(spec[r][0], common[r]), r
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("bartrt_svc_")]


def test_each_clients_setters_act_on_its_own_profile(tmp_path):
    """trm.set_radius / set_cloudtop / set_scattering of one worker process (code/BARTfunc.py:350-360) ride with ITS
    profile only: each client equals an engine with that setter, the untouched client equals the plain engine."""
    from bart_amd import engine, synth, transit_module as trm
    case = synth.make_case(str(tmp_path / "s"), nlayers=40, nwave=600, extra_keys={"shareOpacity": ""})
    (tmp_path / "o").mkdir()
    extra = {1: ("--radius", "95000.0"), 2: ("--cloudtop", "-1.5"), 3: ("--scattering", "1", "2.0"), 4: ("--scattering", "2", "0.0")}
    ready, done, spec = run_workers(case.tcfg, 5, 4, str(tmp_path / "o"), env=SVC_ENV, extra=extra)
    assert sum(r["service"] == "owner" for r in ready) == 1
    engine.init(own_engine_cfg(case, tmp_path))
    try:
        n, L = trm.get_no_samples(), engine.nlayers()
        prof0 = case.profiles().ravel()
        want = {}
        for r in range(5):
            trm.free_memory()
            engine.init(own_engine_cfg(case, tmp_path))
            if r == 1: trm.set_radius(95000.0)
            if r == 2: trm.set_cloudtop(-1.5)
            if r == 3: trm.set_scattering(1, 2.0)
            if r == 4: trm.set_scattering(2, 0.0)
            want[r] = (trm.run_transit(prof0, n), trm.run_transit(worker_profile(prof0, L, r), n))
    finally:
        trm.free_memory()
    for r in range(5):
        for k in (0, 1):
            np.testing.assert_allclose(spec[r][k], want[r][k], rtol=1e-11, atol=1e-13 * np.abs(want[r][k]).max(), err_msg=str((r, k)))
    # the setters did something, each its own thing
    for r in (1, 2, 3, 4):
        assert np.max(np.abs(spec[r][0] / spec[0][0] - 1)) > 1e-8, r


def test_owner_killed_mid_run_is_a_clean_error_in_the_clients(tmp_path):
    from bart_amd import synth
    case = synth.make_case(str(tmp_path / "s"), nlayers=30, nwave=400, extra_keys={"shareOpacity": ""})
    procs = [start_worker(case.tcfg, r, 0, extra=("--until-error",), env=SVC_ENV) for r in range(3)]
    try:
        ready = [expect(p, "ready") for p in procs]
        owner = [i for i, r in enumerate(ready) if r["service"] == "owner"]
        assert len(owner) == 1
        for p in procs:
            p.stdin.write("go\n"); p.stdin.flush()
        import time
        time.sleep(0.5)
        procs[owner[0]].kill()
        procs[owner[0]].wait()
        for i, p in enumerate(procs):
            if i == owner[0]:
                continue
            d = expect(p, "done")
            assert d["steps"] > 0 and d["error"] and "gone" in d["error"] and d["waited_s"] < 10.0, d
            p.stdin.write("bye\n"); p.stdin.flush()
            assert p.wait(timeout=60) == 0, p.stderr.read()[-2000:]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    # the dead owner's name is taken over by the next run: one new owner
    (tmp_path / "o").mkdir()
    ready, _, _ = run_workers(case.tcfg, 4, 2, str(tmp_path / "o"), env=SVC_ENV)
    assert sorted(r["service"] for r in ready) == ["client"] * 3 + ["owner"]
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("bartrt_svc_")]


def test_client_refuses_what_only_an_engine_serves(tmp_path):
    """The owner itself is a client like the others: the extended API says so instead of touching the engine
    behind the dispatcher's back."""
    from bart_amd import synth
    case = synth.make_case(str(tmp_path / "s"), nlayers=20, nwave=200, extra_keys={"shareOpacity": ""})
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import numpy as np\n"
            "from bart_amd import transit_module as trm\n"
            "trm.transit_init(3, ['transit', '-c', %r])\n"
            "assert trm.get_service()['mode'] == 'owner'\n"
            "n = trm.get_no_samples(); L = trm.lib().bartrt_get_nlayers()\n"
            "tau = np.zeros((n, L))\n"
            "rc = trm.lib().bartrt_get_tau(trm._ptr(tau), None, n, L)\n"
            "assert rc == -4, rc\n"
            "assert b'chain service' in trm.lib().bartrt_last_error()\n"
            "assert trm.lib().bartrt_set_integ(0) == -4\n"
            "trm.free_memory()\n" % (ROOT, case.tcfg))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]


def test_a_late_worker_gets_the_bits_of_the_full_batch(tmp_path):
    """VERDICT r5 item 3 (a, b, c): NO `BARTRT_SVC_WAIT_ALL`, the production window of 30 us, and one worker in the
    middle of the slot range (whoever holds slot 4) late on purpose every third step (by 300 us: ten windows).  Its profile goes out in a
    launch of its own, the nine others in a launch without it -- gathered from non-consecutive slots, ONE launch -- and
    every spectrum of every step is the one a ten-walker batch call computes, bit for bit: the kernel is chosen for the
    REGISTERED workers, not for the profiles that happened to post together (the reference's worker calls its own
    engine, the same chain is the same bits every run: code/BARTfunc.py:363)."""
    from bart_amd import engine, synth, transit_module as trm
    # 40 columns per walker: ten walkers = 400 columns (the single-wave kernel), nine = 360 and one = 40 (rows on adjacent
    # lanes, csrc/kernel_table.inc): different kernels if the batch that formed were to choose
    case = synth.make_case(str(tmp_path / "s"), nlayers=40, nwave=2530, extra_keys={"shareOpacity": ""})
    (tmp_path / "o").mkdir()
    steps = 60
    procs = [start_worker(case.tcfg, r, steps, os.path.join(str(tmp_path / "o"), "w%d.npy" % r),
                          ("--late-every", "3", "--late-us", "300", "--late-slot", "4"), {"BARTRT_SVC_WINDOW_US": "30"})
             for r in range(10)]
    try:
        ready = [expect(p, "ready") for p in procs]
        for p in procs:
            p.stdin.write("go\n"); p.stdin.flush()
        done = [expect(p, "done") for p in procs]
        for p in procs:
            p.stdin.write("bye\n"); p.stdin.flush()
        for p in procs:
            assert p.wait(timeout=120) == 0, p.stderr.read()[-3000:]
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert sorted(r["slot"] for r in ready) == list(range(10))
    stats = max((d["service_stats"] for d in done), key=lambda s: s["launches"])
    # the straggler did split rounds (launches that did not hold all ten) ...
    assert stats["launches"] > stats["full"] + 10, stats
    # ... and whoever held a middle slot was missing from some: those went out gathered, as one launch
    assert stats["gathered"] > 0, stats
    # every step of every worker: the same bits
    assert [d["steps_that_differ_from_the_first"] for d in done] == [0] * 10, done
    engine.init(own_engine_cfg(case, tmp_path))
    try:
        L = engine.nlayers()
        prof0 = case.profiles().ravel()
        batch = engine.run_batch(np.stack([worker_profile(prof0, L, r) for r in range(10)]))
        walked = {}
        for n in (1, 9, 10):
            engine.walked_begin()
            engine.run_batch(np.stack([worker_profile(prof0, L, r) for r in range(n)]))
            walked[n] = engine.walked_end()[2]
    finally:
        trm.free_memory()
    assert len(set(walked.values())) > 1, walked          # the batch sizes that formed WOULD have taken different kernels
    spec = [np.load(os.path.join(str(tmp_path / "o"), "w%d.npy" % r)) for r in range(10)]
    for r in range(10):
        assert np.array_equal(spec[r][1], batch[r]), r
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("bartrt_svc_")]


def test_setters_ride_with_their_profile_through_a_gathered_launch(tmp_path):
    """The per-client overrides (radius, cloud top) in the round-6 paths: the production window, no BARTRT_SVC_WAIT_ALL,
    and whoever holds slot 2 late every other step -- so the others' profiles AND overrides go out gathered from
    non-consecutive slots (csrc/svc.hip svc_gather), the straggler's alone -- against engines that carry each setter."""
    from bart_amd import engine, synth, transit_module as trm
    case = synth.make_case(str(tmp_path / "s"), nlayers=40, nwave=600, extra_keys={"shareOpacity": ""})
    (tmp_path / "o").mkdir()
    late = ("--late-every", "2", "--late-us", "300", "--late-slot", "2")
    extra = {0: late, 1: ("--radius", "95000.0") + late, 2: ("--cloudtop", "-1.5") + late, 3: ("--radius", "99000.0") + late,
             4: ("--cloudtop", "-0.5") + late}
    ready, done, spec = run_workers(case.tcfg, 5, 40, str(tmp_path / "o"), env={"BARTRT_SVC_WINDOW_US": "30"}, extra=extra)
    stats = max((d["service_stats"] for d in done), key=lambda s: s["launches"])
    assert stats["gathered"] > 0, stats
    assert [d["steps_that_differ_from_the_first"] for d in done] == [0] * 5, done
    try:
        prof0 = case.profiles().ravel()
        for r in range(5):
            engine.init(own_engine_cfg(case, tmp_path))
            n, L = trm.get_no_samples(), engine.nlayers()
            if r == 1: trm.set_radius(95000.0)
            if r == 2: trm.set_cloudtop(-1.5)
            if r == 3: trm.set_radius(99000.0)
            if r == 4: trm.set_cloudtop(-0.5)
            want = trm.run_transit(worker_profile(prof0, L, r), n)
            trm.free_memory()
            np.testing.assert_allclose(spec[r][1], want, rtol=1e-11, atol=1e-13 * np.abs(want).max(), err_msg=str(r))
    finally:
        trm.free_memory()
