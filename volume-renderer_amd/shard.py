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


class BandGatherer:
    """Frame assembly of the single-frame split with every buffer allocated once: the padded send
    buffer of this rank, and on `dst` the receive list and the assembled (height, width) frame.
    Bands may differ by one row, so they travel padded to the tallest band and are trimmed on arrival.
    gather() runs the one collective of the path (dist.gather: RCCL over xGMI under "nccl") and
    returns the frame on dst, None elsewhere; the returned tensor is reused by the next call."""

    def __init__(self, height, width, rank, world, dtype, device, dst=0):
        import torch
        self.height, self.width, self.rank, self.world, self.dst = height, width, rank, world, dst
        self.rows = band_rows(height, rank, world)[1]
        tallest = band_rows(height, 0, world)[1]
        self.send = torch.zeros((tallest, width), dtype=dtype, device=device)
        self.parts = self.frame = None
        if rank == dst and world > 1:
            self.parts = [torch.empty_like(self.send) for _ in range(world)]
            self.frame = torch.empty((height, width), dtype=dtype, device=device)

    def gather(self, band, dist):
        if self.world == 1:
            return band
        assert tuple(band.shape) == (self.rows, self.width), (tuple(band.shape), self.rows, self.width)
        self.send[:self.rows].copy_(band)
        if self.rank != self.dst:
            dist.gather(self.send, gather_list=None, dst=self.dst)
            return None
        dist.gather(self.send, gather_list=self.parts, dst=self.dst)
        for r in range(self.world):
            y0, rows = band_rows(self.height, r, self.world)
            self.frame[y0:y0 + rows].copy_(self.parts[r][:rows])
        return self.frame


def gather_bands(band, height, width, rank, world, dist, dst=0):
    """One-off form of BandGatherer: assemble a (height, width) plane on rank `dst` from each
    rank's (rows, width) torch tensor.  Returns the frame on dst, else None."""
    return BandGatherer(height, width, rank, world, band.dtype, band.device, dst).gather(band.contiguous(), dist)
