"""World-size-2 rehearsal of the data-parallel path on CPU (gloo): utterance sharding and the
one gather of enhanced spectrograms, with a ragged split (5 utterances over 2 ranks)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import fdbm_amd  # noqa: F401


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from fdbm_amd import dist as fd
    n = 5
    idx = fd.shard_indices(n)
    # "enhanced spectrogram" of utterance i: constant i+1 (+ j*0.5 imaginary)
    X = torch.stack([torch.full((1, 9, 4), float(i + 1), dtype=torch.complex64) + 0.5j for i in idx]) if idx else \
        torch.zeros(0, 1, 9, 4, dtype=torch.complex64)
    got = fd.gather_spectrograms(X, dst=0)
    if rank == 0:
        full = fd.interleave(got, n)
        ok = all(torch.all(full[i].real == i + 1) and torch.all(full[i].imag == 0.5) for i in range(n))
        q.put((ok, [len(c) for c in got]))
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_gather_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok, sizes = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert ok and sizes == [3, 2]


def test_split_list_like_reference():
    from fdbm_amd.dist import split_list, shard_indices
    assert split_list(list(range(7)), 3) == [[0, 1, 2], [3, 4, 5], [6]]
    assert shard_indices(7, rank=1, world_size=3) == [1, 4]
    assert shard_indices(7, rank=2, world_size=3, contiguous=True) == [6]
