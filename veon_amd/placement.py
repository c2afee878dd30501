"""Output placement for the channels-first lift volume.

The fused pool kernel writes 64-voxel rows of all C channel planes of the
(B,C,Z,Y,X) volume at once, i.e. it touches C distinct regions 4*Z*Y*X bytes
apart (2.56 MB at 200x200x16) per workgroup.  On MI355X the very same launch
runs ~34 us into some 205 MB allocations and ~41 us into others (bimodal,
stable per allocation, independent of the virtual address, neighbouring
allocations tend to agree; a linear fill is insensitive).  It is NOT the size of
the page-table fragments: a physically contiguous block
(``contiguous_tensor``: hipDeviceMallocContiguous, the largest fragments the
driver can map) is among the slowest placements measured (48.2 us in the probe
against 43.1 us for the best of 17 ordinary allocations).  What is left is how
the C plane streams, 2.56 MB apart, fall onto HBM channels / banks: a contiguous
block maps them the most regularly.  See DESIGN.md section 4 and
tools/addr_probe.py.

A consumer that keeps ONE output buffer alive across calls (a hipGraph replay
does so anyway) can therefore pick a well-placed one once: ``best_placed``
times a probe on a handful of candidate allocations and keeps the fastest.
"""
import ctypes

import torch

_TYPESTR = {torch.float32: '<f4', torch.bfloat16: None, torch.float16: '<f2',
            torch.uint8: '|u1', torch.int32: '<i4'}


class _ContiguousBlock:
    """Owner of one ``veon_alloc_contiguous`` block, exposed to torch through
    ``__cuda_array_interface__`` (the tensor made from it keeps this object, and
    so the memory, alive)."""

    def __init__(self, shape, dtype, device, flags=None):
        from . import _lib
        self._lib = _lib
        self.nbytes = torch.empty((), dtype=dtype).element_size()
        for d in shape:
            self.nbytes *= int(d)
        p = ctypes.c_void_p(0)
        with torch.cuda.device(device):
            if flags is None:
                st = _lib.lib().veon_alloc_contiguous(ctypes.byref(p), self.nbytes)
            else:
                st = _lib.lib().veon_alloc_device_flags(ctypes.byref(p), self.nbytes,
                                                        int(flags))
        if st != 0 or not p.value:
            raise MemoryError('no physically contiguous block of %d bytes' % self.nbytes)
        self.ptr, self.device = p.value, device
        self.__cuda_array_interface__ = {
            'shape': tuple(int(d) for d in shape), 'typestr': _TYPESTR[dtype],
            'data': (self.ptr, False), 'version': 2, 'strides': None}

    def __del__(self):
        ptr, self.ptr = getattr(self, 'ptr', None), None
        if ptr:
            try:
                with torch.cuda.device(self.device):
                    self._lib.lib().veon_free_device(ctypes.c_void_p(ptr))
            except Exception:   # interpreter shutdown
                pass


def contiguous_tensor(shape, dtype=torch.float32, device=None, flags=None):
    """A tensor on physically contiguous VRAM (hipDeviceMallocContiguous), or None
    when the driver cannot provide one.  ``flags``: another hipExtMallocWithFlags
    flag instead (1 fine-grained, 3 uncached; probe use)."""
    if _TYPESTR.get(dtype) is None:
        return None
    try:
        block = _ContiguousBlock(shape, dtype, device, flags)
    except (MemoryError, AttributeError, OSError):
        return None
    return torch.as_tensor(block, device=device)


def _time(fn, out, iters):
    """GPU time of one ``fn(out)`` in ms.  The launches are captured into a
    hipGraph and the replay is timed, so the figure is the kernels' time and not
    the host's launch overhead (the Python path of one launch costs more than the
    34 us kernel it is looking for, which compresses an eager measurement)."""
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn(out)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(iters):
                fn(out)
        run, per = graph.replay, iters
    except Exception:        # not capturable (a sync inside): eager timing
        torch.cuda.synchronize()
        graph = None

        def run():
            for _ in range(iters):
                fn(out)
        per = iters
    run()
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    run()
    run()
    e1.record()
    e1.synchronize()
    del graph
    return e0.elapsed_time(e1) / (2 * per)


def best_placed(probe, shape, dtype=torch.float32, device=None, k_min=6, k_max=40,
                iters=8, good=0.94, max_bytes=16 << 30):
    """Allocate candidate tensors of ``shape`` (all alive at once, so they are
    distinct allocations), time ``probe(out)`` on each and return
    ``(best_tensor, info)``.  Stops early once a candidate is at least
    ``1 - good`` faster than the median of those seen (the distribution is
    bimodal: the fast mode is ~16 % below the slow one in kernel time, 7-10 % in a
    probe that also runs a small copy kernel); at most
    ``max_bytes`` are held at a time; the losers are released back to the driver."""
    nbytes = torch.empty((), dtype=dtype).element_size()
    for d in shape:
        nbytes *= int(d)
    k_max = max(1, min(k_max, max_bytes // max(nbytes, 1)))
    k_min = min(k_min, k_max)
    cands, times = [], []
    for i in range(k_max):
        try:
            t = torch.empty(shape, dtype=dtype, device=device)
        except torch.cuda.OutOfMemoryError:
            if not cands:
                raise
            break  # tune among what fits
        cands.append(t)
        times.append(_time(probe, t, iters))
        if i + 1 >= k_min:
            med = sorted(times)[len(times) // 2]
            if min(times) < good * med:
                break
    best = min(range(len(times)), key=times.__getitem__)
    out = cands[best]
    info = {'candidates': len(cands), 'best_ms': times[best],
            'median_ms': sorted(times)[len(times) // 2]}
    del cands
    torch.cuda.empty_cache()
    return out, info
