"""Out-of-core inputs: gene chunks streamed from storage through pinned host memory to the GPU.

The reference's v0.2.0 feature (illico/utils/registry.py:162-188: ``h5py.Dataset`` for dense, anndata's backed
``_CSCDataset`` for CSC; memory bound checked in tests/test_asymptotic_wilcoxon.py:198-256): ``fetch(lb, ub)`` reads ONE gene
chunk from the file, the kernel runs on it, the next chunk is read.  Here the three stages of a chunk overlap with the
neighbouring chunks:

    prefetch thread :  read chunk k+1 from storage -> pinned host buffer (slot (k+1) % 2)
                       -> hipMemcpyAsync to the device buffer of that slot, on a COPY stream; event
    calling thread  :  engine stream waits for chunk k's event, runs the dispatcher on the device-resident chunk,
                       writes the [G, w] planes into the caller's result arrays (the D2H of 24 B per test ends the call)

so the H2D copy of chunk k+1 (and the file read before it) runs while chunk k computes.  Two pinned + two device buffers
of one chunk each, grown on demand, are all the staging there is: the host never holds more than the chunk being read plus
the two pinned slots.  PyTorch provides the pinned allocations, the copy stream and the events; the compute is the HIP
engine's (``illico_amd._lib.Engine``), which follows torch's current stream.
"""
from __future__ import annotations

from concurrent.futures import ThreadPoolExecutor

import numpy as np

from illico_amd.utils.registry import CSCMatrix, KernelDataFormat, Test, dispatcher_registry


class _Slot:
    """One staging slot: pinned host buffers + device buffers for the arrays of one chunk (1 dense, 3 CSC)."""

    def __init__(self, torch, device):
        self.torch, self.device = torch, device
        self.pinned, self.dev = {}, {}

    def stage(self, name, arr, stream):
        """Copy ``arr`` (host ndarray) into this slot's pinned buffer and enqueue its upload on ``stream``; returns the
        device tensor (a view of the slot's device buffer with arr's shape and dtype)."""
        torch = self.torch
        nbytes = max(int(arr.nbytes), 1)
        if name not in self.pinned or self.pinned[name].numel() < nbytes:
            cap = nbytes + (nbytes >> 3)
            self.pinned[name] = torch.empty(cap, dtype=torch.uint8).pin_memory()
            self.dev[name] = torch.empty(cap, dtype=torch.uint8, device=self.device)
        pin = self.pinned[name]
        # the one host-side copy of the chunk: straight into pinned memory, whatever the source's strides (a lazy column
        # slice of a memory-mapped file is read from storage right here)
        np.copyto(pin.numpy()[: arr.nbytes].view(arr.dtype).reshape(arr.shape), arr)
        tdt = getattr(torch, str(arr.dtype))
        with torch.cuda.stream(stream):
            self.dev[name][: arr.nbytes].copy_(pin[: arr.nbytes], non_blocking=True)
        return self.dev[name][: arr.nbytes].view(tdt).view(*arr.shape)


def run_streaming(data_handler, iterator, group_container, is_log1p, use_continuity, alternative, tie_correct, planes):
    """Stream the gene chunks ``iterator`` of a backed container through the engine; ``planes`` is the [3, G, n_genes] result."""
    import torch

    from illico_amd._lib import get_engine, normalize_values

    eng = get_engine()
    device = torch.device("cuda", eng.device)
    test = Test.OVR if group_container.encoded_ref_group == -1 else Test.OVO
    fmt = data_handler.kernel_data_format()
    dispatcher = dispatcher_registry.get(test, fmt)
    copy_stream = torch.cuda.Stream(device=device)
    slots = [_Slot(torch, device), _Slot(torch, device)]

    def prefetch(k):
        """storage -> pinned -> (async) device, for chunk k; returns (device container, local bounds, upload event)."""
        torch.cuda.set_device(device)
        lb, ub = iterator[k]
        data, local = data_handler.fetch(lb, ub)
        data = data_handler.to_nb(data)
        slot = slots[k & 1]
        if fmt == KernelDataFormat.DENSE:
            X = normalize_values(np.asarray(data))
            staged = slot.stage("x", X, copy_stream)
        else:
            idt = np.int32 if (np.asarray(data.indices).dtype == np.int32 and np.asarray(data.indptr).dtype == np.int32) else np.int64
            d = slot.stage("data", normalize_values(np.asarray(data.data)), copy_stream)
            i = slot.stage("indices", np.asarray(data.indices, dtype=idt), copy_stream)
            p = slot.stage("indptr", np.asarray(data.indptr, dtype=idt), copy_stream)
            staged = CSCMatrix(d, i, p, tuple(int(s) for s in data.shape))
        del data
        ev = torch.cuda.Event()
        ev.record(copy_stream)
        return staged, local, ev

    with ThreadPoolExecutor(max_workers=1) as pool:
        nxt = pool.submit(prefetch, 0)
        for k, (lb, ub) in enumerate(iterator):
            staged, local, ev = nxt.result()
            # chunk k + 1: its slot's buffers were last used by chunk k - 1, whose dispatcher call has returned
            if k + 1 < len(iterator):
                nxt = pool.submit(prefetch, k + 1)
            torch.cuda.current_stream(device).wait_event(ev)
            out = tuple(planes[j][:, lb:ub] for j in range(3))
            dispatcher(staged, *local, group_container, is_log1p, use_continuity, tie_correct, alternative, out=out, engine=eng)
            del staged
