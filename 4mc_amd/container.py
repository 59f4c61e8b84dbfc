"""Container plumbing around the block engine: framing via the C functions in framing.c, block-range
sharding across ranks, and the one collective of the path (gather of per-block compressed sizes).

Reference: writer native/4mc.c:264-362, reader :560-707, footer consumer
FourMcBlockIndex.java:92-173, multi-GPU plan SURVEY.md §8(e).
"""
import ctypes as C
import numpy as np

from .binding import BLOCK_DTYPE, BLOCKSIZE, lib, make_blocks


def frame_header(magic):
    buf = (C.c_uint8 * 12)()
    lib().fourmc_frame_header(buf, magic)
    return bytes(buf)


def frame_footer(magic, offsets):
    off = np.ascontiguousarray(offsets, dtype=np.uint64)
    buf = (C.c_uint8 * (20 + 4 * len(off)))()
    n = lib().fourmc_frame_footer(buf, magic, off.ctypes.data, len(off))
    return bytes(buf)[:n]


def parse_footer(foot, magic):
    """-> absolute block offsets (np.uint64); raises ValueError on a bad footer."""
    raw = np.frombuffer(foot, dtype=np.uint8)
    off = np.zeros(max(1, (len(foot) - 20) // 4), dtype=np.uint64)
    n = lib().fourmc_frame_parse_footer(raw.ctypes.data, len(raw), magic, off.ctypes.data)
    if n < 0:
        raise ValueError(f"bad footer ({n})")
    return off[:n]


def block_offsets(csizes):
    """Absolute file offset of every block header: 12 + sum_{j<b}(12 + csize_j)  (native/4mc.c:293)."""
    cs = np.asarray(csizes, dtype=np.uint64)
    out = np.empty(len(cs), dtype=np.uint64)
    if len(cs):
        out[0] = 12
        np.cumsum(cs[:-1] + 12, out=out[1:])
        out[1:] += 12
    return out


def assemble_container(magic, usizes, csizes, sums, payloads):
    """File image from per-block results (payloads: iterable of bytes-like, stored or compressed)."""
    parts = [frame_header(magic)]
    hdr = (C.c_uint8 * 12)()
    for u, c, s, p in zip(usizes, csizes, sums, payloads):
        lib().fourmc_frame_block_header(hdr, int(u), int(c), int(s))
        parts.append(bytes(hdr))
        parts.append(bytes(p))
    parts.append(b"\0" * 12)
    parts.append(frame_footer(magic, block_offsets(csizes)))
    return b"".join(parts)


def split_container(image, magic):
    """Walk the block headers of ONE stream held in `image` (bytes / uint8 array).

    Returns (blocks, consumed): descriptors whose src_off/src_len address each payload IN PLACE inside
    the image (so the image can be copied to HBM as it is), dst_off = running sum of usizes,
    dst_cap = usize, xxh32 = stored checksum.  Raises ValueError with the reference's message on
    framing errors (native/4mc.c:575-620,:670-688)."""
    img = np.frombuffer(image, dtype=np.uint8) if not isinstance(image, np.ndarray) else image
    L = lib()
    if len(img) < 12 or L.fourmc_frame_check_header(img[:12].ctypes.data, magic) != 0:
        raise ValueError("Unrecognized header")
    pos, out = 12, 0
    so, do, sl, dc, xs = [], [], [], [], []
    u, c, s = C.c_uint32(), C.c_uint32(), C.c_uint32()
    while True:
        if pos + 12 > len(img):
            raise ValueError("Read error : cannot read next block size")
        L.fourmc_frame_parse_block_header(img[pos:pos + 12].ctypes.data, C.byref(u), C.byref(c), C.byref(s))
        pos += 12
        if u.value == 0 and c.value == 0 and s.value == 0:
            break
        if c.value > BLOCKSIZE:
            raise ValueError("Read error: block size beyond 4MB limit")
        if pos + c.value > len(img):
            raise ValueError("Read error : cannot read data block")
        if u.value != c.value and u.value > BLOCKSIZE:
            raise ValueError("Read error: uncompressed block size beyond 4MB limit")
        so.append(pos); do.append(out); sl.append(c.value); dc.append(u.value); xs.append(s.value)
        pos += c.value
        out += u.value
    if pos + 4 > len(img):
        raise ValueError("Unreadable footer")
    fsz = int.from_bytes(bytes(img[pos:pos + 4]), "big")
    offs = parse_footer(bytes(img[pos:pos + fsz]), magic)
    del offs
    blocks = make_blocks(so, do, sl, dc, xs) if so else np.zeros(0, dtype=BLOCK_DTYPE)
    return blocks, pos + fsz


def shard_range(nblocks, rank, world):
    """Contiguous block range of `rank`: [rank*ceil(N/G), ...)  (SURVEY.md §8(e))."""
    per = -(-nblocks // world) if world > 0 else nblocks
    lo = min(nblocks, rank * per)
    return lo, min(nblocks, lo + per)


def gather_block_index(local_csizes, nblocks, group=None):
    """All ranks' per-block compressed sizes -> every rank gets the full array and the same
    exclusive prefix sum of block offsets.  ONE all_gather of 4 B x blocks (padded to equal
    counts); on GPUs this is RCCL over xGMI, latency-bound (64 KiB at 16 384 blocks).

    local_csizes: 1-D int32/int64 torch tensor (this rank's range, in block order)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        cs = local_csizes.to(torch.int64).cpu().numpy()
        return cs, block_offsets(cs)
    per = -(-nblocks // world)
    pad = torch.zeros(per, dtype=torch.int32, device=local_csizes.device)
    pad[: local_csizes.numel()] = local_csizes.to(torch.int32)
    out = torch.empty(per * world, dtype=torch.int32, device=local_csizes.device)
    dist.all_gather_into_tensor(out, pad, group=group)
    cs = out.cpu().numpy()[:nblocks].astype(np.int64)
    return cs, block_offsets(cs)


def gather_rank_reports(mine, group=None):
    """Every rank's report (a dict with at least `rank`, `first_block_offset`, `shard_bytes`) gathered to every rank, in rank
    order, and the footer index they imply CHECKED: rank r's first block has to sit at 12 (the frame header) + the bytes of the
    shards before it - recomputed here from the gathered shard sizes, independently of the prefix sum the ranks used for their
    own offsets.  Used by bench.py for N > 1 (`per_rank`); world 1: [mine]."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        reports = [mine]
    else:
        reports = [None] * world
        dist.all_gather_object(reports, mine, group=group)
        reports = sorted(reports, key=lambda x: x["rank"])
    run = 12
    for g in reports:
        if g["first_block_offset"] != run:
            raise AssertionError("footer index: rank %d begins at %d, the prefix sum of the gathered shard sizes says %d" % (g["rank"], g["first_block_offset"], run))
        run += g["shard_bytes"]
    return reports
