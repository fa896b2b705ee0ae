"""Multi-GPU sharding of the path (the reference axes are its OpenMP loops over hypotheses S3D:676-681 and detections
PP:582-587): ranks own disjoint sets of independent sequences (stage 1 is stateless per
frame, stage-2 track state is per sequence, so no data-path collective is needed); the only exchange is the all-gather
of the final 3D skeletons so that every rank (in particular the publisher, rank 0) holds all of them."""
import numpy as np
import torch

from . import person_msgs as pm


class HostStagedDist:
    """TEST-ONLY collective backend with the interface this module and bench.py use of torch.distributed, for running the N > 1 code path with
    several ranks on ONE GPU (RCCL refuses two ranks on one device): device tensors are staged through host memory and exchanged by a gloo
    process group.  Ordering as with RCCL: a collective is enqueued "on the current stream" — the stream is drained first (what it enqueued
    before is complete), the result is copied back on it, and the returned work's wait() makes the caller's current stream wait for that copy.
    Nothing about its speed means anything; `get_backend()` says what it is."""

    class _Work:
        def __init__(self, event):
            self.event = event

        def wait(self):
            if self.event is not None:
                torch.cuda.current_stream().wait_event(self.event)

    def __init__(self, dist):
        self._d = dist
        self.ReduceOp = dist.ReduceOp

    def get_world_size(self):
        return self._d.get_world_size()

    def get_rank(self):
        return self._d.get_rank()

    def get_backend(self):
        return "gloo, device buffers staged through host memory (test only)"

    def barrier(self):
        self._d.barrier()

    def destroy_process_group(self):
        self._d.destroy_process_group()

    def _staged(self, fn, out, inp, async_op):
        if not inp.is_cuda:
            fn(out, inp)
            return self._Work(None) if async_op else None
        st = torch.cuda.current_stream()
        st.synchronize()
        h_in, h_out = inp.cpu(), torch.empty(out.shape, dtype=out.dtype)
        fn(h_out, h_in)
        out.copy_(h_out, non_blocking=False)
        ev = st.record_event()
        return self._Work(ev) if async_op else None

    def all_gather_into_tensor(self, out, inp, async_op=False):
        return self._staged(lambda o, i: self._d.all_gather_into_tensor(o, i), out, inp, async_op)

    def all_reduce(self, t, op=None):
        def fn(o, i):
            o.copy_(i)
            self._d.all_reduce(o, **({} if op is None else {"op": op}))
        self._staged(fn, t, t, False)


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


def unpack_dense(persons_u8, offsets_i32, n_frames):
    """Dense exchange form (se3d_pack_stage2_walk_device_dense) of ONE context: persons back to back + [n_frames + 2] offsets ->
    (list over frame slots of person3d arrays, persons dropped for lack of capacity)."""
    p = persons_u8.cpu().numpy() if hasattr(persons_u8, "cpu") else np.asarray(persons_u8)
    o = offsets_i32.cpu().numpy() if hasattr(offsets_i32, "cpu") else np.asarray(offsets_i32)
    persons = p.view(pm.person3d_dt)
    return [persons[o[f]:o[f + 1]].copy() for f in range(n_frames)], int(o[n_frames + 1])


def pack_host_dense(results, capacity):
    """Host-side twin of se3d_pack_stage2_walk_device_dense (gloo tests): (persons uint8 [capacity * itemsize], offsets int32 [n + 2])."""
    n = len(results)
    buf = np.zeros(capacity, pm.person3d_dt)
    off = np.zeros(n + 2, np.int32)
    tot = 0
    for f, pers in enumerate(results):
        off[f] = min(tot, capacity)
        k = max(0, min(len(pers), capacity - tot))
        buf[off[f]:off[f] + k] = pers[:k]
        tot += len(pers)
    off[n] = min(tot, capacity)
    off[n + 1] = tot - off[n]
    return torch.from_numpy(buf.view(np.uint8).reshape(-1).copy()), torch.from_numpy(off)


def pack_host(results, max_hyp):
    """Pack host-side stage results (lists of person3d arrays) like se3d_pack_stage*_device does on the GPU."""
    n = len(results)
    buf = np.zeros((n, max_hyp), pm.person3d_dt)
    cnt = np.zeros(n, np.int32)
    for s, p in enumerate(results):
        cnt[s] = len(p)
        buf[s, : len(p)] = p
    return torch.from_numpy(buf.view(np.uint8).reshape(-1).copy()), torch.from_numpy(cnt)


class SkeletonExchange:
    """Stream-ordered, double-buffered all-gather of the final skeletons (the path's only exchange).

    Every rank owns `n_ctx` contexts; context ci packs its results of a step into slice ci of a pack buffer
    ([slice_person_bytes] bytes of se3d_person3d slots + [slice_counts] int32 counts) on ITS OWN stream, then `packed(ci)` records an
    event; `exchange()` makes a dedicated exchange stream wait for the events of all contexts and enqueues ONE all-gather pair
    (persons, counts) asynchronously, so there is no host synchronisation in a step.  The pack buffers are double-buffered and
    `pack_target(ci)` makes context ci's stream wait (device side) for the collective that read the slice two steps earlier before
    handing it out again.  `streams` = one torch stream per context (torch.cuda.ExternalStream of se3d_stream_handle); None on CPU
    (gloo), where every call is synchronous and the ordering is trivially the program order.
    Layout of a gathered buffer: [world][n_ctx][slice] — rank r's pack buffer verbatim at offset r * n_ctx * slice.
    """

    def __init__(self, dist, n_ctx, slice_person_bytes, slice_counts, device, streams=None):
        self.dist, self.n_ctx, self.sp, self.sc = dist, int(n_ctx), int(slice_person_bytes), int(slice_counts)
        self.world = dist.get_world_size()
        self.cuda = streams is not None
        self.streams = streams
        self.xstream = torch.cuda.Stream() if self.cuda else None
        self.packs = [torch.zeros(self.n_ctx * self.sp, dtype=torch.uint8, device=device) for _ in range(2)]
        self.cnts = [torch.zeros(self.n_ctx * self.sc, dtype=torch.int32, device=device) for _ in range(2)]
        self.gath_p = [torch.empty(self.world * self.n_ctx * self.sp, dtype=torch.uint8, device=device) for _ in range(2)]
        self.gath_c = [torch.empty(self.world * self.n_ctx * self.sc, dtype=torch.int32, device=device) for _ in range(2)]
        if self.cuda:   # torch.zeros fills on torch's current stream; the contexts' streams are non-blocking and would not wait for that fill
            torch.cuda.current_stream(device).synchronize()
        self.works = [None, None]
        self.events = [None] * self.n_ctx
        self.parity = 0
        self.n_exchanges = 0
        self.timing = False        # timed(True): bracket every all-gather pair with events on the exchange stream (exchange_ms())
        self._tev = []

    def pack_target(self, ci):
        """(persons_u8, counts_i32) slices context ci may overwrite for the current step."""
        p = self.parity
        if self.works[p] is not None and self.cuda:
            with torch.cuda.stream(self.streams[ci]):
                for w in self.works[p]:
                    w.wait()
        return self.packs[p][ci * self.sp:(ci + 1) * self.sp], self.cnts[p][ci * self.sc:(ci + 1) * self.sc]

    def packed(self, ci):
        if self.cuda:
            self.events[ci] = self.streams[ci].record_event()

    def exchange(self):
        """Enqueue the all-gather of the current pack buffers; returns the parity (buffer index) it used."""
        p = self.parity
        if self.cuda:
            with torch.cuda.stream(self.xstream):
                for ev in self.events:
                    if ev is not None:
                        self.xstream.wait_event(ev)
                if self.timing:
                    e0 = torch.cuda.Event(enable_timing=True)
                    e0.record(self.xstream)
                self.works[p] = (self.dist.all_gather_into_tensor(self.gath_p[p], self.packs[p], async_op=True),
                                 self.dist.all_gather_into_tensor(self.gath_c[p], self.cnts[p], async_op=True))
                if self.timing:   # the exchange stream waits (device side) for the collectives, then stamps their end
                    for w in self.works[p]:
                        w.wait()
                    e1 = torch.cuda.Event(enable_timing=True)
                    e1.record(self.xstream)
                    self._tev.append((e0, e1))
        else:
            self.dist.all_gather_into_tensor(self.gath_p[p], self.packs[p])
            self.dist.all_gather_into_tensor(self.gath_c[p], self.cnts[p])
        self.parity ^= 1
        self.n_exchanges += 1
        return p

    def timed(self, on=True):
        """Switch the per-exchange event timing on / off (and forget earlier stamps)."""
        self.timing = bool(on) and self.cuda
        self._tev = []

    def exchange_ms(self):
        """(mean, max) duration in ms of the all-gather pairs stamped since timed(True): from the moment every context's pack was complete and
        the collective could start to its end on the device — what the all-gather itself takes under the load of the other streams, NOT what
        it adds to the step (it overlaps the next step's kernels).  Call after finish()."""
        if not self._tev:
            return None
        ms = [a.elapsed_time(b) for a, b in self._tev]
        return float(sum(ms) / len(ms)), float(max(ms))

    def finish(self):
        """Host-side wait for all outstanding collectives (end of the timed region / before reading a gathered buffer)."""
        for ws in self.works:
            if ws is not None:
                for w in ws:
                    w.wait()
        if self.cuda:
            self.xstream.synchronize()

    def gathered(self, p):
        """Views of buffer p: persons [world][n_ctx][slice_person_bytes] uint8, counts [world][n_ctx][slice_counts] int32."""
        return self.gath_p[p].view(self.world, self.n_ctx, self.sp), self.gath_c[p].view(self.world, self.n_ctx, self.sc)

    def verify(self, p):
        """True iff the gathered buffer p equals, on THIS rank, the concatenation of every rank's pack buffer: every rank contributes
        a checksum of what it sent (sum of counts, byte sum of the persons), the checksums are all-gathered and compared with the
        same sums over the received slices.  Call after finish(); collective (every rank must call it)."""
        self.finish()
        # (sum(dtype=int64) accumulates in 64 bits without materialising a widened copy: the gathered buffer is GBs at eight ranks)
        mine = torch.stack([self.cnts[p].sum(dtype=torch.int64), self.packs[p].sum(dtype=torch.int64)]).to(self.packs[p].device)
        allsums = torch.empty(self.world * 2, dtype=torch.int64, device=mine.device)
        self.dist.all_gather_into_tensor(allsums, mine)
        gp, gc = self.gathered(p)
        got = torch.stack([gc.reshape(self.world, -1).sum(1, dtype=torch.int64), gp.reshape(self.world, -1).sum(1, dtype=torch.int64)], 1).reshape(-1)
        r = self.dist.get_rank()
        own_ok = torch.equal(gp[r].reshape(-1), self.packs[p]) and torch.equal(gc[r].reshape(-1), self.cnts[p])
        return bool(torch.equal(allsums.cpu(), got.cpu())) and own_ok


# ---------------------------------------------------------------------------------------------------------------------------------
# Hypothesis shard of ONE stream (north_star config 5): replicated association, rank r owns the hypotheses / detections
# with index % world == r, one all-gather of [frames][ceil(max_hyp / world)] blocks per stage-1 pass and per stage-2 step.
# The device side is se3d_run_stage{1,2}_shard_{pre,post} (include/se3d.h); these helpers are the layout on the host and the collective.
# ---------------------------------------------------------------------------------------------------------------------------------
def shard_slots(max_hyp, world):
    return (max_hyp + world - 1) // world


def pack_own(items, rank, world):
    """items [n][max_hyp] (any dtype) -> this rank's block [n][ceil(max_hyp / world)]: slot j holds index j * world + rank (zero where
    that index does not exist).  Same layout as se3d_k_shard_pack_hyp writes on the device."""
    items = np.asarray(items)
    n, H = items.shape[:2]
    out = np.zeros((n, shard_slots(H, world)) + items.shape[2:], items.dtype)
    own = items[:, rank::world]
    out[:, : own.shape[1]] = own
    return out


def unpack_all(gathered, max_hyp):
    """gathered [world][n][ceil(max_hyp / world)] (= all-gather of the ranks' pack_own blocks) -> [n][max_hyp]."""
    gathered = np.asarray(gathered)
    world, n = gathered.shape[:2]
    out = np.zeros((n, max_hyp) + gathered.shape[3:], gathered.dtype)
    for r in range(world):
        m = len(range(r, max_hyp, world))
        out[:, r::world] = gathered[r, :, :m]
    return out


class HypothesisShard:
    """One context per rank working on the SAME frames: association and track lifecycle replicated, triangulation and LM refinement split
    by index % world, results exchanged by RCCL all-gathers ordered on the context's stream (no host synchronisation)."""

    def __init__(self, dist, node, n_frames, n_seq, device):
        self.dist, self.node, self.n_frames, self.n_seq = dist, node, int(n_frames), int(n_seq)
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        node.set_shard(self.rank, self.world)
        self.stream = torch.cuda.ExternalStream(node.stream_handle())
        b1, b2 = node.shard_bytes(1, n_frames), node.shard_bytes(2, n_seq)
        self.p1 = torch.zeros(b1, dtype=torch.uint8, device=device)
        self.g1 = torch.zeros(b1 * self.world, dtype=torch.uint8, device=device)
        self.p2 = torch.zeros(b2, dtype=torch.uint8, device=device)
        self.g2 = torch.zeros(b2 * self.world, dtype=torch.uint8, device=device)
        if self.p1.is_cuda:   # the zero fills run on torch's current stream, which the context's non-blocking stream does not wait for
            torch.cuda.current_stream(device).synchronize()
        self.bytes_per_step = (b1 + n_frames // max(n_seq, 1) * b2) * self.world

    def run_stage1(self):
        # the unpack kernels index rank r's block at r * staged_frames * ceil(H / world): the gather buffers were sized for n_frames
        assert self.node._staged == self.n_frames, "HypothesisShard was built for %d staged frames, the context holds %d" % (self.n_frames, self.node._staged)
        self.node.run_stage1_shard_pre(self.p1.data_ptr())
        with torch.cuda.stream(self.stream):   # the collective is ordered after the pack on the context's stream
            self.dist.all_gather_into_tensor(self.g1, self.p1)
        self.node.run_stage1_shard_post(self.g1.data_ptr())

    def run_stage2(self, t, frame_slot0):
        assert len(t) == self.n_seq, "HypothesisShard was built for %d sequences per step, got %d" % (self.n_seq, len(t))
        self.node.run_stage2_shard_pre(t, frame_slot0, self.p2.data_ptr())
        with torch.cuda.stream(self.stream):
            self.dist.all_gather_into_tensor(self.g2, self.p2)
        self.node.run_stage2_shard_post(len(t), self.g2.data_ptr())
