"""The path's only exchange step: one all-gather of the depth decisions per step, shared by
bench.py and the multi-rank tests.

Two partitions of a step's work over the ranks of a node (one process per GPU):
  * "frames": whole pictures dealt to ranks in contiguous runs (a GOP; weak or strong scaling);
  * "bands":  contiguous CTU-row bands of every picture, rows [r * rows / R, (r + 1) * rows / R) as fhevc_band (config 3).
On the wire the depth maps travel as the 21-bit split-flag word per CTU (4 B instead of 256 B, the reference's pre-order
split-flag serialisation in fixed bit positions, TComSysuCuMDTools.cpp:16-38); every rank's slice is padded to the largest
slice because all_gather_into_tensor needs equal sizes, and one index_select puts the gathered words into whole-picture CTU
raster order, ready for fhevc_expand_depth_flags_device.

Collective: torch.distributed "nccl" = RCCL over xGMI.  The fallback to a HOST-side gather (D2H of the words, all-gather
over the gloo group, H2D; SURVEY 8(e): "acceptable fallback when RCCL init fails") is a COLLECTIVE decision, never a rank's
own: init_groups brings the gloo group up first (it always comes up), tries RCCL with one test collective, and all ranks
all-reduce(MIN) their verdicts over gloo -- one rank without RCCL puts every rank on the host gather, and `status` says so.
A collective that raises in mid-run cannot be rescued by the rank that saw it (its peers are inside the RCCL collective and
would wait for it for ever): the rank raises GatherError, the process ends non-zero and the launcher (torchrun) takes the
other ranks down -- a loud failure instead of a hang with ranks in different collectives.
"""
import datetime
import os

import torch
import torch.distributed as dist


class GatherError(RuntimeError):
    """the path's only collective failed on this rank after the ranks had agreed on its backend"""


def span(total, rank, world):
    """contiguous share [begin, end) of `total` items for `rank` of `world`: the arithmetic of fhevc_band"""
    return (rank * total) // world, ((rank + 1) * total) // world


class FlagGather:
    def __init__(self, mode, world, rank, num_frames, ctus_x, ctus_y, device, group=None, host_group=None):
        assert mode in ("frames", "bands")
        self.mode, self.world, self.rank = mode, world, rank
        self.num_frames, self.ctus_x, self.ctus_y = num_frames, ctus_x, ctus_y
        self.device = torch.device(device)
        self.group, self.host_group = group, host_group
        # group: the RCCL group of init_groups; False = the ranks agreed that RCCL is not usable; None = no RCCL group was made
        # (CPU tensors: the gloo group is the gather)
        rccl = self.device.type == "cuda" and group is not None and group is not False
        self.status = "single rank" if world == 1 else "rccl" if rccl else \
            "host-gather (RCCL unavailable on at least one rank)" if group is False and self.device.type == "cuda" else "host-gather"
        if not rccl:
            self.group = None
        n = ctus_x * ctus_y
        if mode == "frames":
            self.frames = span(num_frames, rank, world)                 # this rank's pictures
            self.rows = (0, ctus_y)
            self.slice_words = max(span(num_frames, r, world)[1] - span(num_frames, r, world)[0] for r in range(world)) * n
        else:
            self.frames = (0, num_frames)
            self.rows = span(ctus_y, rank, world)                       # this rank's CTU rows of every picture
            self.max_rows = max(span(ctus_y, r, world)[1] - span(ctus_y, r, world)[0] for r in range(world))
            self.slice_words = num_frames * self.max_rows * ctus_x
        self.local_ctus = (self.frames[1] - self.frames[0]) * (self.rows[1] - self.rows[0]) * ctus_x
        # position of every (frame, CTU) of the step inside the gathered [world, slice_words] buffer
        idx = torch.empty(num_frames * n, dtype=torch.int64)
        for r in range(world):
            if mode == "frames":
                fb, fe = span(num_frames, r, world)
                idx[fb * n:fe * n] = r * self.slice_words + torch.arange((fe - fb) * n)
            else:
                rb, re = span(ctus_y, r, world)
                rows = torch.arange(rb, re)
                for f in range(num_frames):
                    dst = (f * ctus_y + rows)[:, None] * ctus_x + torch.arange(ctus_x)[None, :]
                    src = r * self.slice_words + (f * self.max_rows + (rows - rb))[:, None] * ctus_x + torch.arange(ctus_x)[None, :]
                    idx[dst.reshape(-1)] = src.reshape(-1)
        self.index = idx.to(self.device)
        # double-buffered so that the gather of step i can run under the kernels of step i + 1
        self.send = [torch.zeros(self.slice_words, dtype=torch.int32, device=self.device) for _ in range(2)]
        self.recv = [torch.zeros(world * self.slice_words, dtype=torch.int32, device=self.device) for _ in range(2)]
        self.whole = torch.zeros(num_frames * n, dtype=torch.int32, device=self.device)
        self._inflight = None

    def local_words(self, b):
        """where this rank's kernel writes its split-flag words of a step: compact over (its frames) x (its rows) x ctus_x.
        In "bands" mode with fewer rows than max_rows the per-frame pitch on the wire is max_rows * ctus_x, see pack()."""
        return self.send[b][: self.local_ctus]

    def pack(self, b):
        """bands mode: the kernel writes compact [frame][own rows][ctus_x]; the wire layout pads every frame to max_rows"""
        if self.mode != "bands":
            return
        own = self.rows[1] - self.rows[0]
        if own == self.max_rows:
            return
        compact = self.send[b][: self.local_ctus].clone().view(self.num_frames, own * self.ctus_x)
        padded = self.send[b].view(self.num_frames, self.max_rows * self.ctus_x)
        padded.zero_()
        padded[:, : own * self.ctus_x] = compact

    def start(self, b):
        """issue the all-gather of buffer b (asynchronously where the backend allows); one collective in flight at most"""
        assert self._inflight is None
        if self.world == 1:
            self._inflight = (None, b)
            return
        self.pack(b)
        if self.status == "rccl" or (self.status.startswith("host-gather") and self.device.type == "cpu"):
            try:
                src = self.send[b].clone() if self.device.type == "cpu" else self.send[b]
                work = dist.all_gather_into_tensor(self.recv[b], src, group=self.group if self.device.type == "cuda" else self.host_group,
                                                   async_op=True)
            except Exception as e:  # the ranks agreed on this backend at start-up (init_groups): see the module docstring
                raise GatherError(f"all-gather of the split-flag words failed on rank {self.rank}: {type(e).__name__}: {e}") from e
            self._inflight = (work, b)
            return
        self._host_gather(b)
        self._inflight = (None, b)

    def _host_gather(self, b):
        send = self.send[b].cpu()
        recv = torch.empty(self.world * self.slice_words, dtype=torch.int32)
        dist.all_gather_into_tensor(recv, send, group=self.host_group)
        self.recv[b].copy_(recv)

    def finish(self):
        """wait for the collective in flight; -> split-flag words of the whole step, [num_frames * numCtus] in CTU raster order"""
        if self._inflight is None:
            return None
        work, b = self._inflight
        self._inflight = None
        if self.world == 1:
            self.whole.copy_(self.send[b][: self.whole.numel()])
            return self.whole
        if work is not None:
            try:
                work.wait()
            except Exception as e:
                raise GatherError(f"all-gather of the split-flag words failed on rank {self.rank}: {type(e).__name__}: {e}") from e
        torch.index_select(self.recv[b], 0, self.index, out=self.whole)
        return self.whole


def agree(ok, host_group=None):
    """collective over the gloo group: True only if EVERY rank says ok"""
    t = torch.tensor([1 if ok else 0], dtype=torch.int32)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=host_group)
    return bool(t.item())


def init_groups(world, rank, device, backend="nccl", timeout_s=120):
    """(group, host_group) for FlagGather.  The DEFAULT process group is gloo (host_group None = the default group): it carries
    the barriers, the timing reductions and the fallback gather, and it comes up wherever TCP does.  `group` is the RCCL group,
    or False when the ranks AGREED that RCCL is not usable (backend != "nccl", a CPU device, RCCL refusing to come up or failing
    its test collective on ANY rank).  world == 1: (None, None).
    FHEVC_TEST_FAIL_RCCL_RANK=<rank> makes that rank report a failed test collective (the multi-rank tests use it; with
    backend "gloo-as-rccl" a second gloo group stands in for RCCL so that the agreement runs on CPU-only machines)."""
    if world == 1:
        return None, None
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device(device)
    stand_in = backend == "gloo-as-rccl"
    if not stand_in and (backend != "nccl" or dev.type != "cuda"):
        return False, None
    group, ok = None, True
    try:  # new_group is itself a collective over the default group: every rank calls it, in the same order
        group = dist.new_group(backend="gloo" if stand_in else "nccl", timeout=datetime.timedelta(seconds=timeout_s))
    except Exception:
        ok = False
    if ok:
        try:
            t = torch.ones(1, dtype=torch.int32, device=dev if not stand_in else "cpu")
            dist.all_reduce(t, group=group)
            if dev.type == "cuda" and not stand_in:
                torch.cuda.synchronize(dev)
            ok = int(t.item()) == world and os.environ.get("FHEVC_TEST_FAIL_RCCL_RANK") != str(rank)
        except Exception:
            ok = False
    if not agree(ok):  # one rank without RCCL: every rank gathers through the host, and says so
        return False, None
    return group, None


# ---- SURVEY 8(e)'s literal form: the 256-byte depth maps themselves on the wire (CTU-row bands), kept for the band parity test; bench.py and the
# library's callers use FlagGather above (4 B per CTU) ----
def band(ctu_rows, rank, world):
    """rows [begin, end) of rank `rank`: same arithmetic as fhevc_band (fasthevc_amd/csrc/fhevc_api.hip) = span()"""
    return span(ctu_rows, rank, world)


def max_band_rows(ctu_rows, world):
    return max(band(ctu_rows, r, world)[1] - band(ctu_rows, r, world)[0] for r in range(world))


def alloc_gather_buffers(num_frames, ctu_rows, ctus_x, world, device):
    """(gathered, local_view): gathered is [world, frames, max_band_rows, ctus_x, 256] uint8; every rank's slice has
    the same (padded) size, as all_gather_into_tensor requires."""
    mb = max_band_rows(ctu_rows, world)
    gathered = torch.zeros((world, num_frames, mb, ctus_x, 256), dtype=torch.uint8, device=device)
    return gathered


def all_gather_depth(gathered, rank, group=None):
    """In-place all-gather: every rank has written gathered[rank]; afterwards all slices are filled everywhere."""
    world = gathered.shape[0]
    if world == 1:
        return gathered
    flat = gathered.view(world, -1)
    dist.all_gather_into_tensor(flat.view(-1), flat[rank].clone() if flat.device.type == "cpu" else flat[rank], group=group)
    return gathered


def assemble(gathered, ctu_rows):
    """[world, frames, max_band_rows, ctus_x, 256] -> [frames, ctu_rows * ctus_x, 256] in CTU raster order."""
    world, frames, _, ctus_x, _ = gathered.shape
    parts = []
    for r in range(world):
        b, e = band(ctu_rows, r, world)
        parts.append(gathered[r, :, : e - b])
    full = torch.cat(parts, dim=1)
    return full.reshape(frames, ctu_rows * ctus_x, 256)


def alloc_flag_buffers(num_frames, num_ctus, world, device):
    """[world, frames, numCtus] int32 split-flag words (frames dealt to ranks): 4 B per CTU on the wire instead of the
    256 B depth map; every rank expands the gathered words with fhevc_expand_depth_flags_device."""
    return torch.zeros((world, num_frames, num_ctus), dtype=torch.int32, device=device)


def all_gather_flags(gathered, rank, group=None):
    world = gathered.shape[0]
    if world == 1:
        return gathered
    flat = gathered.view(world, -1)
    src = flat[rank].clone() if flat.device.type == "cpu" else flat[rank]
    dist.all_gather_into_tensor(flat.view(-1), src, group=group)
    return gathered
