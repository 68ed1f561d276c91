"""Multi-GPU partitioning of an encode by independent GOP segments, and the only exchange the hot path
needs: a gather of fixed-size per-frame statistics records and of variable-length byte segments.

The reference is single-process; a sequence splits without any data-path exchange because every segment
starts with an I-frame that re-emits VPS/SPS/PPS (src/encode.cpp:5789-5808) and there is one reference
frame (MAX_REF_NUM 1, inc/config.h:99). The record mirrors the per-picture rate-control fields the
reference keeps (xRCPic, inc/rateCtrl.h:97-112). Backend: torch.distributed ("nccl" = RCCL on GPUs,
"gloo" on CPU for tests).
"""
import numpy as np

RC_FIELDS = ("frame", "slice_type", "qp", "lambda", "header_bits", "total_bits")


def partition_frames(n_frames, world, rank, gop=None):
    """Frames [begin, end) of rank `rank`. Segments are whole GOPs when `gop` is given, else as even as possible."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    if gop:
        ngop = (n_frames + gop - 1) // gop
        g0 = (ngop * rank) // world
        g1 = (ngop * (rank + 1)) // world
        return min(g0 * gop, n_frames), min(g1 * gop, n_frames)
    return (n_frames * rank) // world, (n_frames * (rank + 1)) // world


def make_records(frames, slice_types, qps, lambdas, header_bits, total_bits):
    rec = np.zeros((len(frames), len(RC_FIELDS)), np.float64)
    for i, col in enumerate((frames, slice_types, qps, lambdas, header_bits, total_bits)):
        rec[:, i] = col
    return rec


def gather_records(records, device=None):
    """All ranks contribute an (n_i, 6) float64 array; every rank gets the concatenation ordered by frame index."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return records[np.argsort(records[:, 0], kind="stable")]
    world = dist.get_world_size()
    dev = device or "cpu"
    n = torch.tensor([records.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    m = max(counts) if counts else 0
    buf = torch.zeros((m, len(RC_FIELDS)), dtype=torch.float64, device=dev)
    if records.shape[0]:
        buf[:records.shape[0]] = torch.from_numpy(np.ascontiguousarray(records)).to(dev)
    bufs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf)
    parts = [b[:c].cpu().numpy() for b, c in zip(bufs, counts)]
    out = np.concatenate(parts, axis=0) if parts else np.zeros((0, len(RC_FIELDS)))
    return out[np.argsort(out[:, 0], kind="stable")]


def gather_segments(payload, device=None):
    """Variable-length byte segments (one per rank) -> list of bytes in rank order, on every rank.
    Lengths are exchanged first, then one padded all_gather carries the bytes."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [bytes(payload)]
    world = dist.get_world_size()
    dev = device or "cpu"
    n = torch.tensor([len(payload)], dtype=torch.int64, device=dev)
    lens = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(lens, n)
    lens = [int(x.item()) for x in lens]
    m = max(max(lens), 1)
    buf = torch.zeros(m, dtype=torch.uint8, device=dev)
    if len(payload):
        buf[:len(payload)] = torch.frombuffer(bytearray(payload), dtype=torch.uint8).to(dev)
    bufs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(bufs, buf)
    return [bytes(b[:k].cpu().numpy().tobytes()) for b, k in zip(bufs, lens)]
