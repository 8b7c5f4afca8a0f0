"""N>1 path on CPU: two gloo ranks shard a batch of streams, each synthesises its own streams (with the
oracle standing in for the GPU, which this container lacks), and the union equals the single-process
result; the timing reduction is a max, the unit count a sum.  No data-path collective exists."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, total_streams, frames, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from oracle import oracle as O
    from soundkit_amd import sharding
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = sharding.local_streams(total_streams, world, rank)
    results = {}
    for s in mine:
        coeffs = np.stack([[O.seeded_spectrum(1024, 0x12345678 + s * 0x9E3779B9 + 2 * f + c & 0xFFFFFFFF) for c in range(2)]
                           for f in range(frames)])
        pcm, _ = O.synthesize_stream(coeffs, [[0, 0]] * frames, [[f & 1, f & 1] for f in range(frames)])
        results[s] = pcm
    elapsed = sharding.reduce_elapsed(0.5 + rank)          # max over ranks
    units = sharding.sum_units(len(mine) * frames)        # sum over ranks
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), elapsed=elapsed, units=units,
             **{"s%d" % s: v for s, v in results.items()})
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process(tmp_path, oracle):
    from soundkit_amd import sharding
    world, total_streams, frames = 2, 7, 3
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(world, port, total_streams, frames, str(tmp_path)), nprocs=world, join=True)
    seen = {}
    for rank in range(world):
        data = np.load(os.path.join(tmp_path, "rank%d.npz" % rank))
        assert float(data["elapsed"]) == 0.5 + (world - 1)   # the slowest rank's time
        assert float(data["units"]) == total_streams * frames
        for key in data.files:
            if key.startswith("s"):
                s = int(key[1:])
                assert s not in seen and sharding.owner_of(s, world) == rank  # disjoint, stable ownership
                seen[s] = data[key]
    assert sorted(seen) == list(range(total_streams))       # every stream decoded exactly once
    for s in range(total_streams):
        coeffs = np.stack([[oracle.seeded_spectrum(1024, 0x12345678 + s * 0x9E3779B9 + 2 * f + c & 0xFFFFFFFF)
                            for c in range(2)] for f in range(frames)])
        want, _ = oracle.synthesize_stream(coeffs, [[0, 0]] * frames, [[f & 1, f & 1] for f in range(frames)])
        assert np.array_equal(seen[s], want)


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_assignment_is_a_partition(world):
    from soundkit_amd import sharding
    total = 4096
    parts = [sharding.local_streams(total, world, r) for r in range(world)]
    flat = sorted(x for p in parts for x in p)
    assert flat == list(range(total))
    assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
