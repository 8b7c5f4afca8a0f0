"""`python bench.py --gpus N` must be self-sufficient (north_star: "reported at 1, 2, 4 and 8 GPUs"): without a launcher
it starts N ranks itself, and it never prints a line whose n_gpus differs from --gpus.  SK_BENCH_DRY_RUN=1 runs the
launch / rendezvous / max-reduction skeleton on gloo, without a GPU."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def run(argv, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(SK_BENCH_DRY_RUN="1", **env)
    return subprocess.run([sys.executable, BENCH] + argv, env=e, capture_output=True, text=True, timeout=300)


def test_gpus_2_without_a_launcher_starts_two_ranks():
    r = run(["--gpus", "2", "--steps", "4", "--warmup", "1"])
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                       # ONE line, rank 0's
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["steps"] == 4 and out["warmup"] == 1
    assert abs(out["elapsed_max_s"] - 0.002) < 1e-9   # max over ranks, not rank 0's own 0.001
    # every rank has its own share of the host's cores: disjoint, inside the affinity set, equal to within one core
    a, b = out["cpu_budget"]
    assert a and b and not set(a) & set(b) and set(a) | set(b) <= set(os.sched_getaffinity(0))
    assert abs(len(a) - len(b)) <= 1


def test_rank_core_budgets_follow_the_gpus_numa_nodes():
    sys.path.insert(0, ROOT)
    import bench
    cpus = list(range(64))
    node_of = lambda c: c // 32
    real = bench.cpu_node
    bench.cpu_node = node_of
    try:
        shares = [bench.rank_cpu_budget(8, r, gpu_nodes=[0, 0, 0, 0, 1, 1, 1, 1], cpus=cpus) for r in range(8)]
        assert all(len(s) == 8 for s in shares) and sorted(c for s in shares for c in s) == cpus
        assert all(node_of(c) == (0 if r < 4 else 1) for r, s in enumerate(shares) for c in s)
        # a lopsided host (six GPUs on node 0): the ranks still get equal, disjoint shares; node 0's ranks spill onto node 1
        shares = [bench.rank_cpu_budget(8, r, gpu_nodes=[0, 0, 0, 0, 0, 0, 1, 1], cpus=cpus) for r in range(8)]
        assert all(len(s) == 8 for s in shares) and sorted(c for s in shares for c in s) == cpus
        assert all(node_of(c) == 1 for s in shares[6:] for c in s)
        # unknown topology, 7 cores for 2 ranks
        shares = [bench.rank_cpu_budget(2, r, gpu_nodes=[-1, -1], cpus=list(range(7))) for r in range(2)]
        assert sorted(len(s) for s in shares) == [3, 4] and not set(shares[0]) & set(shares[1])
    finally:
        bench.cpu_node = real


def test_under_an_external_launcher_it_is_one_rank():
    """the driver's `python -m torch.distributed.run ... bench.py --gpus N` sets WORLD_SIZE: no second fan-out"""
    r = run(["--gpus", "1"], WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    assert r.returncode == 0 and json.loads(r.stdout.strip())["n_gpus"] == 1


def test_a_stalled_extra_still_prints_the_line_and_exits_non_zero():
    """the default line's end_to_end extra stalling must not look like rc 0 to the driver (bench.py: print_line_and_leave_if_stalled)"""
    r = run(["--gpus", "1"], WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", SK_BENCH_DRY_STALL="1")
    assert r.returncode == 3, (r.returncode, r.stderr)
    out = json.loads(r.stdout.strip())
    assert out["n_gpus"] == 1 and "stalled" in out["end_to_end"]["error"]


def test_refuses_a_world_that_is_not_what_was_asked_for():
    r = run(["--gpus", "4"], WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    assert r.returncode != 0 and "--gpus 4 but WORLD_SIZE=2" in r.stderr and not r.stdout.strip()
    r = run(["--gpus", "1"], WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    assert r.returncode != 0 and not r.stdout.strip()


def test_native_oracle_build_gives_the_portable_results(tmp_path):
    """bench.py's cpu_baseline runs oracle/sk_oracle.c compiled -O3 -march=native for the host; -ffp-contract=off stays,
    so it must produce the portable library's bits (checked in a child process: the library path is fixed at import)."""
    from oracle import oracle as O
    native = O.build_native(str(tmp_path))
    assert native is not None
    code = ("import sys, numpy as np; sys.path.insert(0, %r); from oracle import oracle as O; "
            "sp = np.stack([[O.seeded_spectrum(1024, 7 + 2 * f + c) * np.float32(2500) for c in range(2)] for f in range(6)]); "
            "pcm, _ = O.synthesize_stream(sp, [[0, 0], [1, 1], [2, 2], [3, 3], [0, 0], [0, 0]], [[f & 1] * 2 for f in range(6)]); "
            "y = O.downsample_planar(np.ascontiguousarray(pcm.transpose(1, 0, 2).reshape(2, -1)), 48000, 16000); "
            "z = O.downsample_planar(np.ascontiguousarray(pcm.transpose(1, 0, 2).reshape(2, -1)), 44100, 16000); "
            "sys.stdout.buffer.write(pcm.tobytes() + y.tobytes() + z.tobytes() + O.planar_f32_to_s16_interleaved(y).tobytes())" % ROOT)
    outs = []
    for lib in (None, native):
        env = dict(os.environ)
        env.pop("SK_ORACLE_LIB", None)
        if lib:
            env["SK_ORACLE_LIB"] = lib
        outs.append(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, timeout=300, check=True).stdout)
    assert len(outs[0]) > 80000 and outs[0] == outs[1]
