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


def scatter_batch(tensors, shapes_tail, dtypes, B, src=0, device=None, group=None):
    """Scatter dim-0 chunks of each tensor from `src`.  `tensors` is the list of full tensors on `src`
    (ignored elsewhere); shapes_tail/dtypes describe them on every rank.  Returns the local chunks on `device`."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    bounds = shard_bounds(B, world)
    nloc = bounds[rank][1] - bounds[rank][0]
    nmax = max(b - a for a, b in bounds)
    wire = _wire_device(device, group)
    outs = []
    for i, (tail, dt) in enumerate(zip(shapes_tail, dtypes)):
        recv = torch.empty((nmax,) + tuple(tail), dtype=dt, device=wire)
        chunks = None
        if rank == src:
            full = tensors[i].to(wire)
            chunks = []
            for a, b in bounds:
                c = torch.zeros((nmax,) + tuple(tail), dtype=dt, device=wire)   # equal-size chunks (scatter needs them)
                c[: b - a] = full[a:b]
                chunks.append(c)
        dist.scatter(recv, chunks, src=src, group=group)
        outs.append(recv[:nloc].to(device).contiguous())
    return outs


def gather_waveforms(o_local, B, dst=0, group=None):
    """Gather [n_local, 1, L] waveforms to `dst`; returns [B, 1, L] there and None elsewhere."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    bounds = shard_bounds(B, world)
    nmax = max(b - a for a, b in bounds)
    tail = tuple(o_local.shape[1:])
    dev = o_local.device
    wire = _wire_device(dev, group)
    send = o_local.to(wire)
    if o_local.shape[0] != nmax:
        send = torch.zeros((nmax,) + tail, dtype=o_local.dtype, device=wire)
        send[: o_local.shape[0]] = o_local
    bufs = [torch.empty((nmax,) + tail, dtype=o_local.dtype, device=wire) for _ in range(world)] if rank == dst else None
    dist.gather(send.contiguous(), bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([bufs[i][: b - a] for i, (a, b) in enumerate(bounds)], 0).to(dev)


def infer_sharded(net, mel, lengths, eps, noise_scale=0.667, max_len=None, src=0, group=None):
    """Run net.infer on this rank's shard of a batch living on `src`; `src` gets the full [B,1,L] waveform back.
    mel/lengths/eps need to be valid on `src` only (pass shapes via the src tensors broadcast below)."""
    rank = dist.get_rank(group)
    dev = next(net.parameters()).device
    meta = torch.zeros(3, dtype=torch.int64, device=_wire_device(dev, group))
    if rank == src:
        meta[0], meta[1], meta[2] = mel.shape[0], mel.shape[2], eps.shape[1]
    dist.broadcast(meta, src=src, group=group)
    B, T, IC = (int(v) for v in meta.tolist())
    m, l, e = scatter_batch([mel, lengths, eps] if rank == src else None, [(80, T), (), (IC, T)],
                            [torch.float32, torch.int64, torch.float32], B, src=src, device=dev, group=group)
    if m.shape[0] > 0:
        o = net.infer(m, l, noise_scale=noise_scale, max_len=max_len, eps=e)[0]
    else:
        Td = T if max_len is None else min(T, max_len)
        o = torch.empty(0, 1, Td * net.dec.hop, device=dev)
    return gather_waveforms(o, B, dst=src, group=group)
