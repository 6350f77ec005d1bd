"""Multi-GPU path on CPU: utterance sharding + the one all-gather, world_size 2 over gloo
(reference sort/chunk logic: vall_e/inference.py:288-311)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vall_e_amd.inference import chunk, gather_codes, pack_codes, shard_utterances, sort_key
from vall_e_amd.synth import synth_batch


def fake_codes(i, r):
    """Deterministic stand-in for the model's (R, 8) output of utterance i (the model needs a GPU)."""
    g = torch.Generator().manual_seed(1000 + i)
    return torch.randint(0, 1024, (r, 8), generator=g, dtype=torch.int64)


def _worker(rank, world, port, n, lens, out_q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = shard_utterances(lens, world)[rank]
        local = [fake_codes(i, 5 + i % 7) for i in mine]
        full = gather_codes(local, mine, n, r_max=16)
        out_q.put((rank, [c.numpy() for c in full]))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_shard_is_a_partition_with_balanced_lengths():
    phns, proms, _ = synth_batch(5, 37, ragged=True)
    lens = [sort_key(t, p) for t, p in zip(phns, proms)]
    for world in (1, 2, 4, 8):
        parts = shard_utterances(lens, world)
        assert sorted(i for p in parts for i in p) == list(range(37))
        sizes = [len(p) for p in parts]
        assert max(sizes) - min(sizes) <= 1
        means = [np.mean([lens[i] for i in p]) for p in parts]
        assert max(means) - min(means) < 0.15 * np.mean(lens)
        for p in parts:                                  # each rank's list stays length-sorted (less padding)
            assert [lens[i] for i in p] == sorted(lens[i] for i in p)
    assert chunk(list(range(5)), 2) == [[0, 1], [2, 3], [4]]
    assert chunk(list(range(3)), 0) == [[0], [1], [2]]   # batch_size < 1 -> 1 (inference.py:237-238)


def test_pack_codes_rejects_overflow():
    block, lens = pack_codes([fake_codes(0, 3), fake_codes(1, 0)], r_max=4)
    assert block.shape == (2, 4, 8) and lens.tolist() == [3, 0]
    try:
        pack_codes([fake_codes(0, 9)], r_max=4)
    except ValueError:
        pass
    else:
        raise AssertionError("overflow not detected")


def test_gather_world2_gloo():
    n, world = 7, 2                                      # odd count: ranks hold 4 and 3 utterances
    phns, proms, _ = synth_batch(9, n, n_phn=20, n_prom=30, ragged=True)
    lens = [sort_key(t, p) for t, p in zip(phns, proms)]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, lens, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = [fake_codes(i, 5 + i % 7).numpy() for i in range(n)]
    for r in range(world):
        assert len(got[r]) == n
        for a, b in zip(got[r], want):
            np.testing.assert_array_equal(a, b)
