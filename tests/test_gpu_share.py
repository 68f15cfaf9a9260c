"""`shareOpacity` (code/makecfg.py:106-107, BART.py:259-262): the reference's worker processes -- one per
chain -- keep ONE opacity grid.  Here: N processes initialise on the same cfg, one of them uploads the grid
to HBM and the others map that allocation through a HIP IPC handle (csrc/share.hip); every process computes
the same bits as a process that holds its own copy."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_workers(tcfg, n, steps, outdir, env=None):
    """n tools/mc3_child.py processes in lockstep -> their (ready, done) reports and saved spectra."""
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(env or {}))
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "mc3_child.py"), tcfg, str(r), str(steps),
                               os.path.join(outdir, "w%d.npy" % r)],
                              stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=e)
             for r in range(n)]

    def expect(p, word):
        for line in p.stdout:
            if line.startswith(word + " "):
                return json.loads(line[len(word) + 1:])
        raise AssertionError("worker ended without '%s': %s" % (word, p.stderr.read()[-3000:]))
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
    ready, done, spec = run_workers(shared_case.tcfg, 3, 5, str(tmp_path / "o1"))
    assert all(r["shared"] for r in ready)
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
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert p.stdout.readline().startswith("ready ")
    p.kill(); p.wait()
    assert [f for f in os.listdir("/dev/shm") if f.startswith("bartrt_op_")]
    ready, _, _ = run_workers(c.tcfg, 2, 1, str(tmp_path / "o"))
    assert sum(r["owner"] for r in ready) == 1 and all(r["shared"] for r in ready)
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("bartrt_op_")]
