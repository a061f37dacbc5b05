"""world_size-2 gloo tests of the batch scatter/gather used to shard utterances over GPUs.
The model step itself needs a GPU; here the N>1 data path is covered with an identity "model"."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cases


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class _FakeNet(torch.nn.Module):
    """Stands in for SynthesizerTrn on CPU: 'waveform' = a deterministic function of (mel, eps, lengths)."""

    RECEPTIVE_FRAMES = 2

    class _Dec:
        hop = 4

    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.ones(1))
        self.dec = self._Dec()

    def infer(self, mel, ln, noise_scale=1.0, max_len=None, eps=None):
        T = mel.shape[2] if max_len is None else min(mel.shape[2], max_len)
        o = (mel[:, :1, :T] + noise_scale * eps[:, :1, :T]).repeat_interleave(4, dim=2) * (ln.view(-1, 1, 1) > 0)
        return o, None, None


def _worker(rank, world, port, B, T, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, cases.ROOT)
    from smart_vocoder_amd import parallel
    g = torch.Generator().manual_seed(5)
    mel = torch.randn(B, 80, T, generator=g); eps = torch.randn(B, 192, T, generator=g)
    ln = torch.randint(1, T + 1, (B,), generator=g)
    # scatter -> gather identity
    m, l, e = parallel.scatter_batch([mel, ln, eps] if rank == 0 else None, [(80, T), (), (192, T)],
                                     [torch.float32, torch.int64, torch.float32], B, src=0, device="cpu")
    a, b = parallel.shard_bounds(B, world)[rank]
    ok = torch.equal(m, mel[a:b]) and torch.equal(l, ln[a:b]) and torch.equal(e, eps[a:b])
    back = parallel.gather_waveforms(m[:, :1, :], B, dst=0)
    if rank == 0:
        ok = ok and torch.equal(back, mel[:, :1, :])
    # sharded infer == single-process infer
    net = _FakeNet()
    o = parallel.infer_sharded(net, mel if rank == 0 else None, ln if rank == 0 else None, eps if rank == 0 else None,
                               noise_scale=0.5, src=0)
    if rank == 0:
        ref = net.infer(mel, ln, noise_scale=0.5, eps=eps)[0]
        ok = ok and torch.equal(o, ref)
    # length bucketing: rows come back in the caller's order, the valid region is the single-process result, the padding
    # region is zero; the timing dict is filled; bitwise=True sets and restores the library's variant batch
    tm = {}
    ob = parallel.infer_sharded(net, mel if rank == 0 else None, ln if rank == 0 else None, eps if rank == 0 else None,
                                noise_scale=0.5, src=0, bucket=True, timings=tm)
    ok = ok and set(tm) == {"scatter_ms", "infer_ms", "gather_ms"}
    if rank == 0:
        ok = ok and ob.shape == ref.shape
        for i in range(B):
            n = int(ln[i]) * 4
            ok = ok and torch.equal(ob[i, :, :n], ref[i, :, :n]) and float(ob[i, :, n:].abs().sum()) == 0.0
    from smart_vocoder_amd import _native as N
    o2 = parallel.infer_sharded(net, mel if rank == 0 else None, ln if rank == 0 else None, eps if rank == 0 else None,
                                noise_scale=0.5, src=0, bitwise=True)
    ok = ok and N.set_variant_batch(0) == 0          # restored after the call
    if rank == 0:
        ok = ok and torch.equal(o2, ref)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [4, 5, 1])
def test_scatter_gather_world2(B):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, B, 12, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(ok for _, ok in res), res


def test_shard_bounds_and_sort():
    from smart_vocoder_amd import parallel
    assert parallel.shard_bounds(128, 8) == [(16 * i, 16 * i + 16) for i in range(8)]
    assert parallel.shard_bounds(5, 2) == [(0, 3), (3, 5)]
    assert parallel.shard_bounds(1, 4) == [(0, 1), (1, 1), (1, 1), (1, 1)]
    ln = torch.tensor([5, 9, 2, 9])
    order, inv = parallel.sort_by_length(ln)
    assert ln[order].tolist() == [9, 9, 5, 2] and torch.equal(ln[order][inv], ln)
