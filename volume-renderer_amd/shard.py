"""How the path shards over GPUs (one process per GPU, scene replicated).

* independent views: view k goes to rank k % world; no
  collective on the data path;
* one frame split over the ranks (config 4) and gathered to rank 0 -- the only
  collective (RCCL over xGMI when the backend is "nccl"; gloo in the CPU tests):
  - stripes of rows dealt round-robin (rank r renders stripes r, r + world, ...: StripeGatherer), the
    default: every rank gets sky and model alike.  bench.py picks the stripe height that gives a rank at
    most 64 stripes -- one launch -- and at least 16 rows (8K: 40 / 24 / 16 rows at 2 / 4 / 8 ranks).
    Measured on config 4, every rank's share rendered alone on one GPU (profiles/split_imbalance.py,
    profiles/r03_ab/r03_split_imbalance.log): the slowest rank is 0-2 % above the mean at 2 ranks, 0-6 %
    at 4, 5-14 % at 8; its kernel time is 1/2.1-2.3, 1/3.5-4.3, 1/5.0-5.8 of the whole frame's;
  - contiguous row bands (rank r renders rows [r*H/world, (r+1)*H/world): BandGatherer), SURVEY 8e's
    first form: 17-22 % above the mean at 2 ranks, 47-58 % at 4, 57-64 % at 8 (the model sits in the
    middle rows, the outer bands are sky; 1/1.7, 1/2.3-2.6, 1/3.9-4.5 of the whole frame's time) -- past
    the 15 % at which SURVEY 8e asks for interleaved stripes.

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


def stripe_rows(height, rank, world, stripe=16):
    """[(y0, rows), ...] of the stripes rank `rank` renders: the frame cut into stripes of `stripe` rows (the last
    one may be shorter), stripe s going to rank s % world."""
    n = (height + stripe - 1) // stripe
    return [(s * stripe, min(stripe, height - s * stripe)) for s in range(rank, n, world)]


def stripe_frames(frame, rank, world, stripe=16):
    """(full, tail): the rank's stripes of full height as one FrameDesc array -- one sub-rectangle each, rendered by ONE
    call, whose frame-major planes are the stripes stacked -- and the descriptor of its shorter last stripe, or None
    (all frames of a call share tile_w x tile_h, so that one is a call of its own)."""
    from ._abi import FrameDesc
    mine = stripe_rows(frame.height, rank, world, stripe)
    whole = [(y0, rows) for y0, rows in mine if rows == stripe]
    full = (FrameDesc * len(whole))()
    for i, (y0, rows) in enumerate(whole):
        full[i] = frame.band(y0, rows)
    tail = [frame.band(y0, rows) for y0, rows in mine if rows != stripe]
    return full, (tail[0] if tail else None)


class StripeGatherer:
    """Frame assembly of the single-frame split into stripes, every buffer allocated once.  `send` is this rank's
    (slots, stripe, width) buffer: slot i holds its i-th stripe, which is exactly the frame-major plane layout of one
    render call over stripe_frames()[0] -- a rank renders straight into it (send.data_ptr(); a shorter last stripe
    goes to the slot behind the full ones).  gather() runs the one collective of the path and returns the (height,
    width) frame on dst (the same tensor every time), None elsewhere; de-interleaving is one strided copy per rank."""

    def __init__(self, height, width, rank, world, dtype, device, stripe=16, dst=0):
        import torch
        self.height, self.width, self.rank, self.world, self.stripe, self.dst = height, width, rank, world, stripe, dst
        self.n_full = height // stripe                          # stripes of full height, frame-wide
        self.tail_rows = height - self.n_full * stripe          # rows of the shorter last stripe (0: none)
        self.tail_rank = self.n_full % world if self.tail_rows else -1
        slots = len(stripe_rows(height, 0, world, stripe))      # rank 0 never has fewer stripes than another rank
        self.send = torch.zeros((slots, stripe, width), dtype=dtype, device=device)
        self.parts = self.frame = None
        if rank == dst and world > 1:
            self.parts = [torch.empty_like(self.send) for _ in range(world)]
            self.frame = torch.empty((height, width), dtype=dtype, device=device)

    def full_count(self, rank):
        return len(range(rank, self.n_full, self.world))

    def assemble(self, parts, frame):
        body = frame[:self.n_full * self.stripe].view(self.n_full, self.stripe, self.width)
        for r in range(self.world):
            n = self.full_count(r)
            if n:
                body[r::self.world].copy_(parts[r][:n])
            if r == self.tail_rank:
                frame[self.n_full * self.stripe:].copy_(parts[r][n][:self.tail_rows])
        return frame

    def gather(self, dist):
        if self.world == 1:
            if self.frame is None:
                import torch
                self.frame = torch.empty((self.height, self.width), dtype=self.send.dtype, device=self.send.device)
            return self.assemble([self.send], self.frame)
        if self.rank != self.dst:
            dist.gather(self.send, gather_list=None, dst=self.dst)
            return None
        dist.gather(self.send, gather_list=self.parts, dst=self.dst)
        return self.assemble(self.parts, self.frame)


def gather_bands(band, height, width, rank, world, dist, dst=0):
    """One-off form of BandGatherer: assemble a (height, width) plane on rank `dst` from each
    rank's (rows, width) torch tensor.  Returns the frame on dst, else None."""
    return BandGatherer(height, width, rank, world, band.dtype, band.device, dst).gather(band.contiguous(), dist)
