"""HIP-event timing of the dominant kernel inside gnr_fwd / gnr_bwd (bench and tests only).

torch.cuda.Event brackets whole calls; to time ONE kernel inside a C-ABI call the events are
created with the HIP runtime directly and handed to libgnr (gnr_set_kernel_timing), which records
them on the launch stream right around that kernel."""
from __future__ import annotations

import ctypes as C

from . import _lib


class KernelTimer:
    """aux=False: the dominant kernel (gnr_set_kernel_timing); aux=True: the HBM-bound compositing
    pass of gnr_bwd (gnr_set_aux_timing)."""

    def __init__(self, aux: bool = False):
        self.aux = aux
        self.hip = C.CDLL("libamdhip64.so")
        self.start, self.stop = C.c_void_p(), C.c_void_p()
        assert self.hip.hipEventCreate(C.byref(self.start)) == 0
        assert self.hip.hipEventCreate(C.byref(self.stop)) == 0
        self.lib = _lib.load()

    def __enter__(self):
        (self.lib.gnr_set_aux_timing if self.aux else self.lib.gnr_set_kernel_timing)(self.start, self.stop)
        return self

    def __exit__(self, *exc):
        (self.lib.gnr_set_aux_timing if self.aux else self.lib.gnr_set_kernel_timing)(None, None)

    def elapsed_ms(self) -> float:
        """Duration of the most recent bracketed kernel (synchronises on the stop event)."""
        assert self.hip.hipEventSynchronize(self.stop) == 0
        ms = C.c_float()
        assert self.hip.hipEventElapsedTime(C.byref(ms), self.start, self.stop) == 0
        return float(ms.value)
