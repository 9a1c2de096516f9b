"""Sharding of one text over the ranks of a node and the gather of match records to a root.

The scan cursor is the only scan state (reference aho_corasick.h:47,70: the caller owns the
`const ACState *`), and after any position it is a function of the last lmax symbols only, so a
text splits into independent contiguous shards: rank r owns [r*N/R, (r+1)*N/R), starts from the
root lmax-1 symbols earlier and reports only matches that END inside its range.
Tables are replicated.  The one exchange step is the variable-length gather of 16-byte records:
an all-gather of the per-rank counts, then direct peer->root transfers (xGMI is point-to-point:
7 links into the root, no ring).  Ranks own increasing, disjoint position ranges and each emits
canonical order, so concatenation in rank order IS the canonical order -- no global sort.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_symbols, rank, world, lmax):
    """(read_begin, own_begin, own_end): the rank scans [read_begin, own_end) from the root and
    reports matches ending in [own_begin, own_end)."""
    own_begin = n_symbols * rank // world
    own_end = n_symbols * (rank + 1) // world
    warm = max(lmax - 1, 0)
    read_begin = max(own_begin - warm, 0)
    return read_begin, own_begin, own_end


def gather_records(local, group=None, dst=0, wire=None):
    """local: int64 tensor [n, 2] (16-byte records, canonical order, global end_pos).
    Returns on `dst` the concatenation over ranks in rank order, elsewhere None.
    wire: (pos_lo, pos_bits, len_bits) from Plan.wire() -- this rank's records then travel as 8-byte
    words (acm_gpu_pack_records_device on the sender, acm_gpu_unpack_records_device on the root:
    CUDA tensors only): half the bytes over the rank's one link to the root."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    out_dev = local.device
    use_wire = wire is not None and local.is_cuda and world > 1 and rank != dst
    send = local
    if use_wire:
        from .binding import pack_records
        send = pack_records(local.contiguous(), wire)
    # ONE code path for both backends: everything below (the all-gather of the counts, the root's
    # irecv per peer into its slice, the peers' isend) is the same calls whether the group is gloo
    # or nccl (RCCL); the only difference is this hop -- gloo moves host memory, so CUDA tensors go
    # through the CPU there (CPU tests, 1-GPU rehearsals) and the result goes back to the device at
    # the end.  The world-2/3 gloo tests and the two-process gloo test on one GPU therefore execute
    # every line the 8-GPU nccl run executes, on host tensors; the C caller's multi-GPU entry
    # (acm_gpu_multi_*, include/acm_gpu.h) does the same gather with hipMemcpyPeerAsync.
    via_host = dist.get_backend(group) == "gloo" and local.is_cuda
    if via_host:
        local, send = local.cpu(), send.cpu()
    dev = local.device
    # per rank: records, wire form or not, and its three parameters (the root unpacks with the SENDER's)
    mine = [local.shape[0], 1 if use_wire else 0] + (list(wire) if use_wire else [0, 0, 0])
    meta = [torch.zeros(5, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(meta, torch.tensor(mine, dtype=torch.int64, device=dev), group=group)
    meta = [[int(v) for v in t.tolist()] for t in meta]
    counts = [t[0] for t in meta]
    if world == 1:
        return local.to(out_dev)
    if rank == dst:
        total = sum(counts)
        out = torch.empty((total, 2), dtype=torch.int64, device=dev)
        offs = [0]
        for c in counts:
            offs.append(offs[-1] + c)
        out[offs[dst]:offs[dst + 1]] = local
        ops, staged = [], {}
        for r in range(world):
            if r != dst and counts[r]:
                if meta[r][1]:
                    staged[r] = torch.empty(counts[r], dtype=torch.int64, device=dev)
                    ops.append(dist.P2POp(dist.irecv, staged[r], _global_rank(r, group), group))
                else:
                    ops.append(dist.P2POp(dist.irecv, out[offs[r]:offs[r + 1]], _global_rank(r, group), group))
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        out = out.to(out_dev)
        if staged:
            from .binding import unpack_records
            for r, st in staged.items():
                unpack_records(st.to(out_dev), tuple(meta[r][2:5]), out[offs[r]:offs[r + 1]])
        return out
    if counts[rank]:
        for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, send.contiguous(), _global_rank(dst, group), group)]):
            req.wait()
    return None


def _global_rank(group_rank, group):
    if group is None:
        return group_rank
    return dist.get_global_rank(group, group_rank)


def scan_sharded(scan_fn, n_symbols, lmax, make_shard, group=None, dst=0, wire_fn=None):
    """One sharded pass.  make_shard(read_begin, own_end) -> this rank's text for
    [read_begin, own_end); scan_fn(text, emit_from, pos_base) -> int64 [n, 2] records in canonical
    order whose end_pos is global.  wire_fn(span, pos_lo) -> Plan.wire's triple (records on the wire
    as 8-byte words) or None.  Returns the gathered records on dst (None elsewhere)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    read_begin, own_begin, own_end = shard_bounds(n_symbols, rank, world, lmax)
    text = make_shard(read_begin, own_end)
    local = scan_fn(text, own_begin - read_begin, read_begin)
    return gather_records(local, group, dst, wire_fn(own_end - read_begin, read_begin) if wire_fn else None)
