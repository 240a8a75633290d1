"""Zero-filled scratch without a memset launch per buffer.

Per-channel statistics and backward reductions are accumulated with atomics into small fp64 buffers
that must start at zero.  ``zeros64`` hands out slices of large pre-zeroed chunks (one memset per
2 MiB instead of one per buffer); a slice is used once and chunks are ordinary torch tensors kept
alive by the views taken from them, so autograd lifetimes are safe."""
import torch


class ZeroArena(object):
    def __init__(self, dtype, chunk_elems):
        self.dtype, self.chunk_elems = dtype, chunk_elems
        self.chunk, self.used = {}, {}

    def take(self, numel, device):
        numel = (numel + 3) & ~3
        if numel > self.chunk_elems:
            return torch.zeros(numel, device=device, dtype=self.dtype)
        cuda = device.type == 'cuda'
        key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream if cuda else 0,
               torch.cuda.is_current_stream_capturing() if cuda else False)
        if key not in self.chunk or self.used[key] + numel > self.chunk_elems:
            self.chunk[key] = torch.zeros(self.chunk_elems, device=device, dtype=self.dtype)
            self.used[key] = 0
        out = self.chunk[key][self.used[key]:self.used[key] + numel]
        self.used[key] += numel
        return out

    def reset(self):
        self.chunk.clear()
        self.used.clear()


_ZEROS64 = ZeroArena(torch.float64, 1 << 18)       # 2 MiB chunks


def zeros64(shape, device):
    n = 1
    for s in shape:
        n *= s
    return _ZEROS64.take(n, device)[:n].view(shape)


_ZEROS32 = ZeroArena(torch.float32, 1 << 20)      # 4 MiB chunks


def zeros32(numel, device):
    return _ZEROS32.take(numel, device)[:numel]


def reset_arena():
    """Drop the current chunks.  Call before HIP-graph capture begins and after it ends, so that every
    captured use of a chunk is preceded, inside the same graph, by the memset that zeroes it."""
    _ZEROS64.reset()
    _ZEROS32.reset()
