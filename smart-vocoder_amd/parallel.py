"""Batch sharding of independent utterances over the GPUs of one node.

The model has no exchange step: every utterance is independent, weights are replicated, so the only
communication is the scatter of inputs from the source rank and the gather of waveforms back — RCCL over
xGMI when the process group is "nccl" (payloads <= ~9 MB per peer, latency-bound), gloo in CPU tests.
One process per GPU; contiguous chunks of ceil(B/world) utterances per rank.
"""
import torch
import torch.distributed as dist


def shard_bounds(B, world_size):
    """[(start, stop)] per rank: contiguous chunks, the first B % world ranks get one extra utterance."""
    q, r = divmod(B, world_size)
    out, s = [], 0
    for i in range(world_size):
        n = q + (1 if i < r else 0)
        out.append((s, s + n))
        s += n
    return out


def sort_by_length(lengths):
    """Bucket-style ordering (reference data_utils.py:130-226 idea): longest first, so shards get similar T."""
    order = torch.argsort(lengths, descending=True, stable=True)
    return order, torch.argsort(order)


def _wire_device(device, group=None):
    """Device the collective runs on: the GPU for RCCL ("nccl"); host memory for gloo (its scatter/gather take CPU
    tensors), which is how the N>1 path is exercised on a single GPU or on CPU."""
    return torch.device("cpu") if dist.get_backend(group) == "gloo" else device


def _pack_layout(shapes_tail, dtypes, nmax):
    """Byte layout of one rank's packed chunk: one region per tensor, [nmax, *tail] each, widest element type first (so every
    region starts aligned for its type); the chunk is padded to a multiple of 16 bytes.  Returns ([(offset, nbytes)] in the
    callers' tensor order, chunk_bytes)."""
    per = [int(torch.tensor([], dtype=dt).element_size()) for dt in dtypes]
    nb = [nmax * p * int(torch.Size(tail).numel()) for p, tail in zip(per, shapes_tail)]
    order = sorted(range(len(dtypes)), key=lambda i: -per[i])
    offs, off = [0] * len(dtypes), 0
    for i in order:
        offs[i] = off
        off += (nb[i] + 15) // 16 * 16
    return list(zip(offs, nb)), max(off, 16)


def _region(buf2d, off, nbytes, dtype, nmax, tail):
    """View of bytes [off, off + nbytes) of every row of a [rows, chunk_bytes] uint8 buffer as [rows, nmax, *tail] of `dtype`."""
    return buf2d[:, off:off + nbytes].view(dtype).view((buf2d.shape[0], nmax) + tuple(tail))


def scatter_batch(tensors, shapes_tail, dtypes, B, src=0, device=None, group=None):
    """Scatter dim-0 chunks of each tensor from `src` in ONE collective: the source packs (lengths | mel | eps ...) of every rank
    into one byte buffer [world, chunk_bytes] - one strided device copy per tensor, no zero filling, no per-rank loop when B
    divides evenly - and scatters its rows (views, no further copies).  `tensors` is the list of full tensors on `src` (ignored
    elsewhere); shapes_tail/dtypes describe them on every rank.  Returns the local chunks on `device`.

    ALIASING: when the collective runs on `device` itself (RCCL) the returned tensors are VIEWS OF ONE shared uint8 receive
    buffer (mel, lengths and eps share a storage; `.untyped_storage()` spans all of them).  They are meant to be read -
    `infer` only reads its inputs; clone a chunk before writing to it in place or before handing it to code that assumes
    it owns the storage.  With an uneven split the spare slot of the smaller chunks is sent uninitialised and is NOT part of
    the returned views."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    bounds = shard_bounds(B, world)
    nloc = bounds[rank][1] - bounds[rank][0]
    nmax = max(b - a for a, b in bounds)
    wire = _wire_device(device, group)
    layout, chunk = _pack_layout(shapes_tail, dtypes, nmax)
    recv = torch.empty(chunk, dtype=torch.uint8, device=wire)
    rows = None
    if rank == src:
        send = torch.empty((world, chunk), dtype=torch.uint8, device=wire)
        big = B - world * (nmax - 1) if B % world else world            # ranks that hold nmax utterances (the first B % world)
        for (off, nb), tail, dt, full in zip(layout, shapes_tail, dtypes, tensors):
            full = full.to(device=wire, dtype=dt)
            dst = _region(send, off, nb, dt, nmax, tail)
            if big == world:
                dst.copy_(full.reshape((world, nmax) + tuple(tail)))
            else:                                                        # uneven split: two block copies; the spare slots stay unset
                dst[:big].copy_(full[: big * nmax].reshape((big, nmax) + tuple(tail)))
                if nmax > 1:
                    dst[big:, : nmax - 1].copy_(full[big * nmax:].reshape((world - big, nmax - 1) + tuple(tail)))
        rows = list(send.unbind(0))
    dist.scatter(recv, rows, src=src, group=group)
    outs = []
    r2 = recv.view(1, chunk)
    for (off, nb), tail, dt in zip(layout, shapes_tail, dtypes):
        t = _region(r2, off, nb, dt, nmax, tail)[0, :nloc]
        outs.append(t if wire == device or device is None else t.to(device))
    return outs


def gather_waveforms(o_local, B, dst=0, group=None, out=None):
    """Gather [n_local, 1, L] waveforms to `dst`; returns [B, 1, L] there and None elsewhere.  The ranks' chunks land directly in
    views of ONE receive buffer [world, nmax, 1, L] - `out` when the caller supplies it ([B, 1, L] on the gather device, even
    split: bench.py preallocates it outside the timed region), else a fresh one - so an even split needs no concatenation."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    bounds = shard_bounds(B, world)
    nmax = max(b - a for a, b in bounds)
    tail = tuple(o_local.shape[1:])
    dev = o_local.device
    wire = _wire_device(dev, group)
    send = o_local.to(wire)
    if o_local.shape[0] != nmax:
        send = torch.empty((nmax,) + tail, dtype=o_local.dtype, device=wire)
        send[: o_local.shape[0]] = o_local
    even = B == world * nmax
    buf = views = None
    used_out = False
    if rank == dst:
        if even and out is not None and out.device == wire and out.dtype == o_local.dtype and tuple(out.shape) == (B,) + tail and out.is_contiguous():
            buf = out.view((world, nmax) + tail)
            used_out = True
        else:
            buf = torch.empty((world, nmax) + tail, dtype=o_local.dtype, device=wire)
        views = list(buf.unbind(0))
    dist.gather(send.contiguous(), views, dst=dst, group=group)
    if rank != dst:
        return None
    if even:
        res = buf.view((B,) + tail)
    else:
        big = B - world * (nmax - 1)
        res = torch.cat([buf[:big].reshape((big * nmax,) + tail), buf[big:, : nmax - 1].reshape(((world - big) * (nmax - 1),) + tail)], 0)
    res = res if wire == dev else res.to(dev)
    if out is not None and not used_out and out.device == res.device and out.shape == res.shape:
        out.copy_(res)
        return out
    return res


def _now(dev):
    import time
    if dev is not None and dev.type == "cuda":
        torch.cuda.synchronize(dev)
    return time.perf_counter()


def infer_sharded(net, mel, lengths, eps, noise_scale=0.667, max_len=None, src=0, group=None, bucket=False, bitwise=False,
                  halo_frames=None, timings=None, shape=None, host_lengths=None, out=None, validate=False):
    """Run net.infer on this rank's shard of a batch living on `src`; `src` gets the full [B,1,L] waveform back.
    mel/lengths/eps need to be valid on `src` only (shapes are broadcast from there).

    bucket=False (default): contiguous chunks of the batch as given, every shard runs all T frames - the result is the
      reference's ``infer`` output for the whole batch, padding region included.
    bucket=True: length bucketing before sharding (the reference's bucket idea, data_utils.py:130-226): utterances are
      ordered longest first (``sort_by_length``), cut into contiguous shards, and each shard runs only
      ``max(lengths of the shard) + halo`` frames (halo >= the decoder's receptive field, ``net.DECODER_RECEPTIVE_FRAMES``), so
      ranks holding short utterances finish early.  Rows come back in the caller's order (un-permuted on gather).  Every
      sample below ``lengths[i] * hop`` sees exactly the inputs it sees in the unsharded call; samples in the padding
      region (which the reference computes from masked zeros and callers slice off) are returned as ZERO.
    bitwise=True: kernel variants are chosen for the JOB's batch size on every rank (svoc_set_variant_batch), so the
      gathered result is bit-identical to a single process running the whole batch under the same setting;
      with bucket=True the trimmed frame count differs per shard, so only bucket=False is bit-exact.
    timings: optional dict, filled with host-side milliseconds {"scatter_ms", "infer_ms", "gather_ms"} (device
      synchronised at each boundary - diagnostics, costs a sync per phase).
    shape: (B, T) of the job's batch, passed by EVERY rank - the call then makes no metadata broadcast and no host read-back:
      it is ONE scatter + infer + ONE gather, all enqueued (n_mel / inter_channels are the model's).  Without it the
      shapes are broadcast from `src` (one small collective + a host synchronisation per call).
    host_lengths: bucket=True only - the job's lengths as a HOST sequence / tensor on every rank: the sort and every
      shard's frame count are then computed on the host, without the lengths broadcast and its device read-back.
    out: `src` only, optional [B, 1, L] receive buffer for the gather (see gather_waveforms; ignored with bucket=True).
    validate: what the ranks pass without a collective (`shape`, `host_lengths`, `max_len`, the flags) is trusted by default - a rank that
      passes another shape sizes its scatter / gather buffers differently and the job hangs or returns garbage.  validate=True spends one
      small all_gather per call on a digest of those arguments and raises on EVERY rank if they differ, and `src` checks `host_lengths`
      against `lengths` (one host read-back): for bring-up and tests, not for the steady state."""
    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    dev = next(net.parameters()).device
    wire = _wire_device(dev, group)
    tdev = dev if timings is not None else None
    t0 = _now(tdev) if timings is not None else 0.0
    if shape is not None:
        B, T = int(shape[0]), int(shape[1])
        IC, n_mel = int(net.inter_channels), int(getattr(getattr(getattr(net, "enc_p", None), "pre_enc", None), "in_channels", 80))
        if rank == src and (tuple(mel.shape) != (B, n_mel, T) or tuple(eps.shape) != (B, IC, T)):
            raise ValueError(f"shape={tuple(shape)} does not describe mel {tuple(mel.shape)} / eps {tuple(eps.shape)}")
    else:
        meta = torch.zeros(4, dtype=torch.int64, device=wire)
        if rank == src:
            meta[0], meta[1], meta[2], meta[3] = mel.shape[0], mel.shape[2], eps.shape[1], mel.shape[1]
        dist.broadcast(meta, src=src, group=group)
        B, T, IC, n_mel = (int(v) for v in meta.tolist())
    # max_len as SynthesizerTrn.infer reads it (None: all frames; negative: counted from the end)
    if max_len is not None:
        max_len = max(0, min(T, int(max_len) if max_len >= 0 else T + int(max_len)))
        if max_len == 0:
            raise ValueError("max_len leaves no frames to decode")
    if validate:
        hl = None if host_lengths is None else torch.as_tensor(host_lengths, dtype=torch.int64).reshape(-1)
        # last field: src's own verdict on host_lengths (only src holds `lengths`); every rank reads it from src's row, so all raise together
        hl_ok = 1
        if rank == src and hl is not None and not torch.equal(hl, lengths.detach().to("cpu", torch.int64).reshape(-1)):
            hl_ok = 0
        mine = torch.tensor([B, T, IC, n_mel, -1 if max_len is None else max_len, int(bool(bucket)), int(bool(bitwise)),
                             -1 if hl is None else int(hl.sum()), -1 if halo_frames is None else int(halo_frames), int(src), hl_ok],
                            dtype=torch.int64, device=wire)
        every = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(every, mine, group=group)
        if any(not torch.equal(v[:-1], every[0][:-1]) for v in every):
            raise ValueError("infer_sharded: the ranks disagree on (B, T, inter_channels, n_mel, max_len, bucket, bitwise, sum(host_lengths), "
                             f"halo_frames, src): {[v[:-1].tolist() for v in every]}")
        if int(every[src][-1]) == 0:
            raise ValueError("infer_sharded: host_lengths differ from lengths on src")
    inv = None
    if bucket and host_lengths is not None:
        hl = torch.as_tensor(host_lengths, dtype=torch.int64, device="cpu").reshape(-1)
        if hl.numel() != B:
            raise ValueError("host_lengths must hold the job's B lengths")
        order, inv = sort_by_length(hl)
        ln_all = hl[order]
        if rank == src:
            od = order.to(mel.device)
            mel, lengths, eps = mel[od], lengths.to(mel.device)[od], eps[od]
    elif bucket:
        # every rank needs the sorted lengths to size its own shard; only src has them
        ln_all = torch.zeros(B, dtype=torch.int64, device=wire)
        order = None
        if rank == src:
            order, inv = sort_by_length(lengths.to(wire))
            ln_all.copy_(lengths.to(wire)[order])
            od = order.to(mel.device)
            mel, lengths, eps = mel[od], lengths.to(mel.device)[od], eps[od]
        dist.broadcast(ln_all, src=src, group=group)
    m, l, e = scatter_batch([mel, lengths, eps] if rank == src else None, [(n_mel, T), (), (IC, T)],
                            [torch.float32, torch.int64, torch.float32], B, src=src, device=dev, group=group)
    t1 = _now(tdev) if timings is not None else 0.0
    Td = T if max_len is None else max_len
    hop = net.dec.hop
    if m.shape[0] > 0:
        Tr = T
        if bucket:
            a, b = shard_bounds(B, world)[rank]
            halo = int(getattr(net, "DECODER_RECEPTIVE_FRAMES", 128) if halo_frames is None else halo_frames)
            Tr = min(T, (int(ln_all[a:b].max().item()) + halo + 3) // 4 * 4)
            if Tr < T:
                m, e = m[:, :, :Tr].contiguous(), e[:, :, :Tr].contiguous()
        ml = None if max_len is None else min(max_len, Tr)
        if bitwise:
            from . import _native as N
            with N.variant_batch(B):
                o = net.infer(m, l, noise_scale=noise_scale, max_len=ml, eps=e)[0]
        else:
            o = net.infer(m, l, noise_scale=noise_scale, max_len=ml, eps=e)[0]
        if bucket:
            full = torch.zeros(o.shape[0], 1, Td * hop, dtype=o.dtype, device=o.device)
            n = min(o.shape[2], Td * hop)
            full[:, :, :n] = o[:, :, :n]
            # padding region -> zero (see docstring)
            idx = torch.arange(Td * hop, device=o.device).view(1, 1, -1)
            full = full * (idx < (l.to(o.device) * hop).view(-1, 1, 1))
            o = full
    else:
        o = torch.empty(0, 1, Td * hop, device=dev)
    t2 = _now(tdev) if timings is not None else 0.0
    out = gather_waveforms(o, B, dst=src, group=group, out=None if bucket else out)
    if out is not None and inv is not None:
        out = out[inv.to(out.device)]
    if timings is not None:
        t3 = _now(tdev)
        timings.update(scatter_ms=(t1 - t0) * 1e3, infer_ms=(t2 - t1) * 1e3, gather_ms=(t3 - t2) * 1e3)
    return out
