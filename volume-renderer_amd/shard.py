"""How the path shards over GPUs (one process per GPU, scene replicated).

* independent views: view k goes to rank k % world; no
  collective on the data path;
* one frame split into row bands (config 4): rank r renders rows
  [r*H/world, (r+1)*H/world) and the bands are gathered to rank 0 -- the only
  collective (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests).

Every pixel is independent once its direction sum is known, and the kernel
restarts the reference's serial sums (main.cpp:282-286) exactly for any
sub-rectangle, so a band render is bit-identical to the same rows of the full
frame."""


def views_for_rank(n_views, rank, world):
    """Round-robin view partition: the view indices rank `rank` renders."""
    return list(range(rank, n_views, world))


def band_rows(height, rank, world):
    """(y0, rows) of rank's contiguous row band; the first height % world ranks get one more row."""
    base, extra = divmod(height, world)
    y0 = rank * base + min(rank, extra)
    return y0, base + (1 if rank < extra else 0)


def band_frame(frame, rank, world):
    y0, rows = band_rows(frame.height, rank, world)
    return frame.band(y0, rows)


def gather_bands(band, height, width, rank, world, dist, dst=0):
    """Assemble a (height, width) plane on rank `dst` from each rank's (rows, width)
    torch tensor.  Bands may differ by one row, so they are padded to the tallest
    band for the collective and trimmed afterwards.  Returns the frame on dst, else None."""
    import torch
    tallest = band_rows(height, 0, world)[1]
    padded = band
    if band.shape[0] < tallest:
        pad = torch.zeros((tallest - band.shape[0], width), dtype=band.dtype, device=band.device)
        padded = torch.cat([band, pad], dim=0)
    padded = padded.contiguous()
    if world == 1:
        return band
    if rank == dst:
        parts = [torch.empty_like(padded) for _ in range(world)]
        dist.gather(padded, gather_list=parts, dst=dst)
        rows = [parts[r][:band_rows(height, r, world)[1]] for r in range(world)]
        return torch.cat(rows, dim=0)
    dist.gather(padded, gather_list=None, dst=dst)
    return None
