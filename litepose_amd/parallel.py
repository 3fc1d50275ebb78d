"""Multi-GPU: one process per GPU, images sharded, ONE all-gather of keypoint records.

The path shards trivially (no cross-image op: BN is folded, grouping is per image,
SURVEY.md section 8e).  Each rank runs ``PoseEngine.infer_batch`` on its contiguous slice
and the fixed-capacity records {count, kpts[pcap,J,3+T], scores[pcap]} are exchanged
with a single ``all_gather_into_tensor`` (backend "nccl" == RCCL over xGMI on ROCm;
"gloo" in the CPU tests).  ~8.5 KB/image at pcap 30: latency-bound, so one collective per
batch and no bucketing.  The reference has no counterpart (valid.py:165 DataParallel,
batch 1).
"""
import torch
import torch.distributed as dist


def shard_range(n_total, rank, world_size):
    """Contiguous split; the first n_total % world ranks get one extra image."""
    base, rem = divmod(n_total, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def pack_records(kpts, count, scores):
    """-> flat float32 [N, 1 + pcap*J*D + pcap]; count travels bit-cast to float32."""
    N = kpts.shape[0]
    c = count.to(torch.int32).view(N, 1).contiguous().view(torch.float32)
    return torch.cat([c, kpts.reshape(N, -1), scores.reshape(N, -1)], dim=1).contiguous()


def unpack_records(flat, pcap, J, D):
    N = flat.shape[0]
    count = flat[:, :1].contiguous().view(torch.int32).view(N)
    k = pcap * J * D
    kpts = flat[:, 1:1 + k].reshape(N, pcap, J, D)
    scores = flat[:, 1 + k:1 + k + pcap].reshape(N, pcap)
    return kpts, count, scores


def all_gather_records(kpts, count, scores, group=None):
    """Every rank contributes the SAME number of images (pad the last shard); returns the
    records of all ranks in rank order: (kpts [W*N,...], count [W*N], scores [W*N,pcap])."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return kpts, count, scores
    N, pcap, J, D = kpts.shape
    flat = pack_records(kpts, count, scores)
    if flat.is_cuda and dist.get_backend(group) == 'gloo':
        # functional check of the N > 1 path on a box without RCCL peers (tests, LP_BENCH_BACKEND=gloo):
        # gloo gathers host buffers
        host = flat.cpu()
        out = torch.empty((world * N, host.shape[1]), dtype=host.dtype)
        dist.all_gather_into_tensor(out, host, group=group)
        return unpack_records(out.to(flat.device), pcap, J, D)
    out = torch.empty((world * N, flat.shape[1]), dtype=flat.dtype, device=flat.device)
    dist.all_gather_into_tensor(out, flat, group=group)
    return unpack_records(out, pcap, J, D)
