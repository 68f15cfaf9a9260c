"""`python bench.py --gpus N` must launch itself: with N > 1
and no RANK in the environment the script starts torch.distributed.run on itself
as a child process, the ranks run the sharded step loop, and exactly ONE JSON
line comes back on stdout.  Run here with --dry-gloo: a stub engine on CPU
tensors over gloo, whose reassembled spectra are checked sample for sample
inside bench.py; the GPU engine takes the same code path with nccl."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=e,
                          capture_output=True, text=True, timeout=600)


def test_gpus2_self_launch_one_json_line():
    r = _bench("--gpus", "2", "--dry-gloo", "--steps", "11", "--warmup", "3", "--walkers", "3",
               "--nwave", "2501", "--gather-steps", "4", "--sweep", "")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout            # the contract: ONE line on stdout
    j = json.loads(lines[0])
    assert j["dry"] is True and j["value"] is None      # a dry run is never a measurement
    assert j["n_gpus"] == 2 and j["steps"] == 11 and j["warmup"] == 3
    assert j["scaling"] == "weak" and j["config"]["walkers_per_step"] == 6   # walkers scale with N
    assert j["ms_per_step"] > 0
    # the N > 1 line explains itself: per-rank window and kernel times, the
    # bucket's bytes, the final drain, the skew between the ranks
    d = j["scaling_diag"]
    assert d["mode"] == "shard" and len(d["per_rank_window_s"]) == 2 and len(d["per_rank_rt_kernel_ms"]) == 2
    assert d["steps_per_bucket"] == 4 and d["rank_skew_ms"] >= 0 and len(d["per_rank_final_drain_ms"]) == 2
    wmax = 2501 - 2501 // 2
    assert d["allgather_send_bytes_per_rank_per_bucket"] == 4 * 6 * wmax * 8
    assert d["allgather_recv_bytes_per_rank_per_bucket"] == 2 * 4 * 6 * wmax * 8


def test_gpus8_uneven_grid_dry_run():
    """A node's worth of ranks on a grid that does not divide (2 501 samples over 8): the blocks' padding in the
    bucketed all-gather, the reassembly checked sample for sample inside bench.py, and the keys the first real
    8-GPU record will be read by (what an N = 8 line should look like)."""
    r = _bench("--gpus", "8", "--dry-gloo", "--steps", "9", "--warmup", "2", "--walkers", "2",
               "--nwave", "2501", "--gather-steps", "4", "--sweep", "")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["config"]["walkers_per_step"] == 16 and j["scaling"] == "weak"
    d = j["scaling_diag"]
    for key in ("mode", "per_rank_window_s", "rank_skew_ms", "per_rank_rt_kernel_ms", "per_rank_final_drain_ms",
                "steps_per_bucket", "allgather_send_bytes_per_rank_per_bucket",
                "allgather_recv_bytes_per_rank_per_bucket", "ms_per_step_minus_rt_kernel"):
        assert key in d, key
    assert len(d["per_rank_window_s"]) == 8 and len(d["per_rank_final_drain_ms"]) == 8
    wmax = max(2501 * (k + 1) // 8 - 2501 * k // 8 for k in range(8))
    assert wmax == 313 and d["allgather_send_bytes_per_rank_per_bucket"] == 4 * 16 * wmax * 8


def test_gpus2_replicas_mode_has_no_collective():
    """--mode replicas: SURVEY 8e's baseline -- every rank the whole grid and its own walkers."""
    r = _bench("--gpus", "2", "--dry-gloo", "--mode", "replicas", "--steps", "5", "--warmup", "1", "--walkers", "3",
               "--nwave", "700", "--sweep", "")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["walkers_per_step"] == 6
    assert "replicas x2" in j["config"]["parallelism"]
    assert j["scaling_diag"]["mode"] == "replicas" and "steps_per_bucket" not in j["scaling_diag"]


def test_child_failure_becomes_the_exit_code():
    # an impossible shard (more ranks than samples) fails in the ranks; the
    # launcher must pass the failure on, not print a line
    r = _bench("--gpus", "2", "--dry-gloo", "--steps", "1", "--warmup", "0", "--walkers", "1",
               "--nwave", "1", "--sweep", "")
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_single_rank_needs_no_launcher():
    r = _bench("--dry-gloo", "--steps", "3", "--warmup", "1", "--walkers", "2", "--nwave", "700", "--sweep", "")
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads(r.stdout.strip())
    assert j["n_gpus"] == 1 and j["dry"] is True


def test_launch_byte_model_counts_shared_rows_once(tmp_path):
    """bench.py's per-launch byte model: identical walkers share every
    row (unique bytes of one walker + the per-walker records and spectra), walkers in
    disjoint temperature brackets share none, and a launch that stops half way down
    moves half the table bytes; the effective bytes follow SURVEY 8d's per-spectrum
    figure restricted to the walked layers."""
    import numpy as np
    sys.path.insert(0, ROOT)
    import bench
    from bart_amd import synth
    L, W, nw = 40, 640, 4
    case = synth.make_case(str(tmp_path), nlayers=L, nwave=W)          # 4 molecules, H2-H2 CIA, 27 planes
    prof = case.profiles(temp=np.full(L, 1010.0)).ravel()
    same = np.tile(prof, (nw, 1))
    walked = np.full((nw, W // 64), L, np.int32)
    m = bench.launch_byte_model(case, same, walked, 64, W)
    table = 2 * L * 4 * W * 8                     # two planes, four molecules
    cia = 2 * W * 8                               # one pair of CIA planes, shared by every layer
    fixed = nw * L * (4 + 2 * 4 + 2 + 2) * 8 + nw * W * 8 + W * 8
    assert m["unique_bytes"] == table + cia + fixed and m["layers_walked_frac"] == 1.0
    assert m["effective_bytes"] == nw * (2 * L * W * 4 * 8 + 2 * L * W * 8) + fixed
    apart = np.array([case.profiles(temp=np.full(L, t)).ravel() for t in (450.0, 1010.0, 1650.0, 2950.0)])
    m2 = bench.launch_byte_model(case, apart, walked, 64, W)
    assert m2["unique_bytes"] - fixed == nw * table + 4 * cia          # nothing shared (CIA grid: 200 K steps)
    half = bench.launch_byte_model(case, same, walked // 2, 64, W)
    assert half["unique_bytes"] - fixed == table // 2 + cia and half["layers_walked_frac"] == 0.5
