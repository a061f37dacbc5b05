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
    inter_channels = 192

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
    # gather into a caller-owned buffer: filled in place for an even split (no concatenation), copied into otherwise
    out = torch.full((B, 1, T), float("nan")) if rank == 0 else None
    back2 = parallel.gather_waveforms(m[:, :1, :], B, dst=0, out=out)
    if rank == 0:
        ok = ok and back2.data_ptr() == out.data_ptr() and torch.equal(out, mel[:, :1, :])
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
    # shapes passed by the caller: no metadata broadcast, no host read-back; gather into the caller's buffer
    calls = {"n": 0}
    orig_bc = dist.broadcast

    def counting_broadcast(*a, **k):
        calls["n"] += 1
        return orig_bc(*a, **k)

    dist.broadcast = counting_broadcast
    try:
        outb = torch.full((B, 1, T * 4), float("nan")) if rank == 0 else None
        o3 = parallel.infer_sharded(net, mel if rank == 0 else None, ln if rank == 0 else None, eps if rank == 0 else None,
                                    noise_scale=0.5, src=0, shape=(B, T), out=outb)
        o4 = parallel.infer_sharded(net, mel if rank == 0 else None, ln if rank == 0 else None, eps if rank == 0 else None,
                                    noise_scale=0.5, src=0, bucket=True, shape=(B, T), host_lengths=ln.tolist())
    finally:
        dist.broadcast = orig_bc
    ok = ok and calls["n"] == 0
    if rank == 0:
        ok = ok and torch.equal(o3, ref) and torch.equal(o4, ob)
        if B % world == 0:
            ok = ok and o3.data_ptr() == outb.data_ptr()
        try:
            parallel.infer_sharded(net, mel, ln, eps, src=0, shape=(B + 1, T))
            ok = False
        except ValueError:
            pass
    # validate=True: one small all_gather on a digest of what every rank passed; agreeing ranks run as before ...
    o5 = parallel.infer_sharded(net, mel if rank == 0 else None, ln if rank == 0 else None, eps if rank == 0 else None,
                                noise_scale=0.5, src=0, shape=(B, T), validate=True)
    if rank == 0:
        ok = ok and torch.equal(o5, ref)
    # ... a rank that passes another shape is an error on EVERY rank instead of a hang in the scatter
    try:
        parallel.infer_sharded(net, mel if rank == 0 else None, ln if rank == 0 else None, eps if rank == 0 else None,
                               noise_scale=0.5, src=0, shape=(B, T) if rank == 0 else (B, T + 4), validate=True)
        ok = False
    except ValueError as ex:
        ok = ok and "disagree" in str(ex)
    # host_lengths that are not the lengths: src notices under validate=True and says so in its row of the digest (both ranks pass the same wrong list)
    if B > 1:
        wrong = [int(v) for v in ln.tolist()]
        wrong[0] = max(1, wrong[0] - 1) if wrong[0] > 1 else 2
        try:
            parallel.infer_sharded(net, mel if rank == 0 else None, ln if rank == 0 else None, eps if rank == 0 else None,
                                   noise_scale=0.5, src=0, bucket=True, shape=(B, T), host_lengths=wrong, validate=True)
            ok = False
        except ValueError as ex:
            ok = ok and "host_lengths" in str(ex)          # on every rank: nobody is left waiting in the scatter
    # a negative max_len counts from the end, as SynthesizerTrn.infer reads it (models.py: Td = T + max_len)
    o6 = parallel.infer_sharded(net, mel if rank == 0 else None, ln if rank == 0 else None, eps if rank == 0 else None,
                                noise_scale=0.5, src=0, shape=(B, T), max_len=-2)
    if rank == 0:
        ok = ok and torch.equal(o6, net.infer(mel, ln, noise_scale=0.5, eps=eps, max_len=T - 2)[0])
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


@pytest.mark.parametrize("B,world", [(5, 3), (2, 3), (9, 8), (7, 7), (1, 4)])
def test_scatter_uneven_splits_and_small_batches_mocked(B, world, monkeypatch):
    """uneven splits and B < world of the packed scatter, and what the returned chunks alias.  torch.distributed is
    replaced by a recorder: the source rank's call captures the rows it would send, every other rank's call receives its row."""
    from smart_vocoder_amd import parallel
    T = 6
    g = torch.Generator().manual_seed(B * 31 + world)
    mel = torch.randn(B, 80, T, generator=g); eps = torch.randn(B, 192, T, generator=g)
    ln = torch.randint(1, T + 1, (B,), generator=g)
    state = {"rank": 0, "rows": None}

    def fake_scatter(recv, rows, src=0, group=None):
        if state["rank"] == src:
            assert rows is not None and len(rows) == world and all(r.numel() == recv.numel() for r in rows)
            state["rows"] = [r.clone() for r in rows]
        else:
            assert rows is None
        recv.copy_(state["rows"][state["rank"]])

    monkeypatch.setattr(parallel.dist, "get_world_size", lambda group=None: world)
    monkeypatch.setattr(parallel.dist, "get_rank", lambda group=None: state["rank"])
    monkeypatch.setattr(parallel.dist, "get_backend", lambda group=None: "gloo")
    monkeypatch.setattr(parallel.dist, "scatter", fake_scatter)
    bounds = parallel.shard_bounds(B, world)
    for r in range(world):
        state["rank"] = r
        m, l, e = parallel.scatter_batch([mel, ln, eps] if r == 0 else None, [(80, T), (), (192, T)],
                                         [torch.float32, torch.int64, torch.float32], B, src=0, device=torch.device("cpu"))
        a, b = bounds[r]
        assert m.shape[0] == l.shape[0] == e.shape[0] == b - a
        assert torch.equal(m, mel[a:b]) and torch.equal(l, ln[a:b]) and torch.equal(e, eps[a:b])
        if b > a:
            # documented aliasing: the three chunks are views of one receive buffer
            assert m.untyped_storage().data_ptr() == l.untyped_storage().data_ptr() == e.untyped_storage().data_ptr()
            keep = e.clone()
            m.add_(1.0)                                  # writing one chunk in place must not reach its neighbours' regions
            assert torch.equal(e, keep) and torch.equal(l, ln[a:b])


def test_shard_bounds_and_sort():
    from smart_vocoder_amd import parallel
    assert parallel.shard_bounds(128, 8) == [(16 * i, 16 * i + 16) for i in range(8)]
    assert parallel.shard_bounds(5, 2) == [(0, 3), (3, 5)]
    assert parallel.shard_bounds(1, 4) == [(0, 1), (1, 1), (1, 1), (1, 1)]
    ln = torch.tensor([5, 9, 2, 9])
    order, inv = parallel.sort_by_length(ln)
    assert ln[order].tolist() == [9, 9, 5, 2] and torch.equal(ln[order][inv], ln)


def test_pack_layout_is_aligned_and_disjoint():
    """One rank's packed chunk (parallel.scatter_batch): regions ordered widest element type first, every region aligned for its
    type, no overlap, chunk a multiple of 16 bytes."""
    from smart_vocoder_amd import parallel
    for nmax in (1, 3, 16):
        tails, dts = [(80, 7), (), (192, 7)], [torch.float32, torch.int64, torch.float32]
        layout, chunk = parallel._pack_layout(tails, dts, nmax)
        assert chunk % 16 == 0
        spans = sorted(layout)
        for (o0, n0), (o1, _) in zip(spans, spans[1:]):
            assert o0 + n0 <= o1
        assert spans[-1][0] + spans[-1][1] <= chunk
        for (off, nb), tail, dt in zip(layout, tails, dts):
            assert off % 16 == 0 and nb == nmax * int(torch.Size(tail).numel()) * torch.tensor([], dtype=dt).element_size()


def test_decoder_receptive_frames_follow_the_config():
    """the halo of length-bucketed shards comes from the configuration, not from a constant of the iitp model."""
    from smart_vocoder_amd import models
    m = cases.IITP_MODEL
    assert models.decoder_receptive_frames(m["resblock"], m["resblock_kernel_sizes"], m["resblock_dilation_sizes"], m["upsample_rates"],
                                           m["upsample_kernel_sizes"]) == 16
    # smaller first upsampling rate -> the ResBlock stacks reach further in frames
    assert models.decoder_receptive_frames("1", [3, 7, 11], [[1, 3, 5]] * 3, [4, 4, 4, 4], [8, 8, 8, 8]) > 16
    # ResBlock2 (dilated convolutions only) reaches less far than ResBlock1
    assert models.decoder_receptive_frames("2", [3, 7, 11], [[1, 3]] * 3, [8, 8, 2, 2], [16, 16, 4, 4]) < 16
