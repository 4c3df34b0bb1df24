"""
Device-memory plumbing: PyTorch-ROCm is the allocator / stream model, nothing
more.  Raw pointers and the current HIP stream are handed to the C ABI.
"""

import numpy

try:
    import torch
except ImportError:          # pragma: no cover - the image always has torch
    torch = None


def require_gpu():
    """The product path has no CPU fallback: fail loudly without a GPU."""
    if torch is None:
        raise RuntimeError("mixemt_amd needs PyTorch-ROCm for device memory")
    if not torch.cuda.is_available():
        raise RuntimeError("mixemt_amd: no ROCm GPU visible -- the EM hot path runs only on "
                           "the HIP kernels (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def current_stream():
    """hipStream_t of torch's current stream, as an integer for ctypes."""
    return torch.cuda.current_stream().cuda_stream


def as_device(x, dtype, dev):
    """numpy array / tensor -> contiguous device tensor of `dtype`."""
    if isinstance(x, torch.Tensor):
        if dtype == torch.uint16:
            return x.to(device=dev).contiguous()      # already a 16-bit bit pattern
        return x.to(device=dev, dtype=dtype).contiguous()
    arr = numpy.ascontiguousarray(x)
    if dtype == torch.uint16:
        # torch has no first-class uint16 arithmetic; only the bytes matter here
        arr = numpy.ascontiguousarray(arr.astype(numpy.uint16, copy=False)).view(numpy.int16)
        return torch.from_numpy(arr).to(dev)
    return torch.from_numpy(arr).to(device=dev).to(dtype).contiguous()


def to_host(t):
    """
    Device tensor -> numpy array (the drop-in path returns whole R x H matrices to numpy callers).
    A plain pageable copy: measured on the MI355X box at 8.7 GB, `.cpu()` runs at 14.2 GB/s while
    staging through a freshly page-locked buffer of that size took twice as long (7.0 GB/s: locking
    the pages costs more than the faster DMA saves; profiles/r02/dropin_build.txt).  Callers that care
    keep the matrices on the device (tensors in -> tensors out).
    """
    if not isinstance(t, torch.Tensor):
        return numpy.asarray(t)
    return t.detach().cpu().numpy()


def ptr(t):
    return 0 if t is None else t.data_ptr()


def device_empty(shape, dtype, dev, what):
    """torch.empty on the device; running out of HBM becomes the ValueError the reference's
    CLI reports and exits on (bin/mixemt:325-327) instead of a torch-specific exception."""
    try:
        return torch.empty(shape, dtype=dtype, device=dev)
    except torch.cuda.OutOfMemoryError as exc:
        need = 1
        for n in (shape if isinstance(shape, (tuple, list)) else (shape,)):
            need *= int(n)
        raise ValueError("not enough device memory for %s (%.1f GB): %s"
                         % (what, need * torch.empty(0, dtype=dtype).element_size() / 1e9, exc))
