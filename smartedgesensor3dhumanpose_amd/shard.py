"""Multi-GPU sharding of the path: ranks own disjoint sets of independent sequences (stage 1 is stateless per
frame, stage-2 track state is per sequence, so no data-path collective is needed); the only exchange is the all-gather
of the final 3D skeletons so that every rank (in particular the publisher, rank 0) holds all of them."""
import numpy as np
import torch

from . import person_msgs as pm


def rank_sequences(n_total, rank, world):
    """Sequence ids owned by `rank`: round-robin (sequence s lives on rank s % world; tracks never migrate)."""
    return list(range(rank, n_total, world))


def gather_skeletons(dist, packed_u8, counts_i32, max_hyp):
    """All-gather of per-rank skeleton blocks.

    packed_u8 : uint8 tensor [n_slots * max_hyp * sizeof(se3d_person3d)] (device tensor under RCCL, CPU tensor under gloo)
    counts_i32: int32 tensor [n_slots] number of valid persons per slot
    Returns (gathered_u8 [world, ...], gathered_counts [world, n_slots]).
    """
    world = dist.get_world_size()
    gp = torch.empty(world * packed_u8.numel(), dtype=torch.uint8, device=packed_u8.device)
    gc = torch.empty(world * counts_i32.numel(), dtype=torch.int32, device=counts_i32.device)
    dist.all_gather_into_tensor(gp, packed_u8.reshape(-1))
    dist.all_gather_into_tensor(gc, counts_i32.reshape(-1))
    return gp.view(world, -1), gc.view(world, -1)


def unpack_skeletons(gathered_u8, gathered_counts, max_hyp):
    """Host view: list over ranks of list over slots of person3d arrays (valid entries only)."""
    g = gathered_u8.cpu().numpy() if hasattr(gathered_u8, "cpu") else np.asarray(gathered_u8)
    c = gathered_counts.cpu().numpy() if hasattr(gathered_counts, "cpu") else np.asarray(gathered_counts)
    out = []
    for r in range(g.shape[0]):
        persons = g[r].view(pm.person3d_dt).reshape(-1, max_hyp)
        out.append([persons[s, : c[r, s]].copy() for s in range(c.shape[1])])
    return out


def pack_host(results, max_hyp):
    """Pack host-side stage results (lists of person3d arrays) like se3d_pack_stage*_device does on the GPU."""
    n = len(results)
    buf = np.zeros((n, max_hyp), pm.person3d_dt)
    cnt = np.zeros(n, np.int32)
    for s, p in enumerate(results):
        cnt[s] = len(p)
        buf[s, : len(p)] = p
    return torch.from_numpy(buf.view(np.uint8).reshape(-1).copy()), torch.from_numpy(cnt)
