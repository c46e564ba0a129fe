"""CTU-row band sharding of a picture (or a GOP) over the ranks of one node, and the single all-gather of the
depth map.  One process per GPU; `torch.distributed` backend "nccl" is RCCL over xGMI on
the GPU box, "gloo" in the CPU tests.  No other data-path collective exists on this path.

The reference is single-threaded and has no communication layer; the only consumer of the
gathered map is the HM host thread, so rank 0's copy is the one that matters -- an all-gather keeps every rank's
copy identical, which is what the multi-rank parity test checks.
"""
import torch
import torch.distributed as dist


def band(ctu_rows, rank, world):
    """rows [begin, end) of rank `rank`: same arithmetic as fhevc_band (fasthevc_amd/csrc/fhevc_api.hip)."""
    return (rank * ctu_rows) // world, ((rank + 1) * ctu_rows) // world


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
