"""HIP-event timing of single stages inside gnr_fwd / gnr_bwd (bench and tests only).

torch.cuda.Event brackets whole calls; to time ONE kernel inside a C-ABI call the events are
created with the HIP runtime directly and handed to libgnr (gnr_set_stage_timing), which records
them on the launch stream right around that stage.

``StageTimer`` owns a pool of event pairs: ``arm()`` hands the next free pair to the library, nothing
is synchronised while work is being enqueued, and ``collect()`` reads every pair after the caller's
final ``torch.cuda.synchronize()`` -- so timing adds no host-sync bubbles to the timed region."""
from __future__ import annotations

import ctypes as C
from typing import List

from . import _lib

STAGES = {"fwd_mlp": _lib.STAGE_FWD_MLP, "dgrad": _lib.STAGE_DGRAD, "comp_bwd": _lib.STAGE_COMP_BWD,
          "wgrad": _lib.STAGE_WGRAD}


class StageTimer:
    def __init__(self, stage: str, pool: int = 64):
        self.stage = STAGES[stage]
        self.hip = C.CDLL("libamdhip64.so")
        self.lib = _lib.load()
        self.pairs = []
        for _ in range(pool):
            a, b = C.c_void_p(), C.c_void_p()
            assert self.hip.hipEventCreate(C.byref(a)) == 0 and self.hip.hipEventCreate(C.byref(b)) == 0
            self.pairs.append((a, b))
        self.used = 0
        self.keep = False

    def reset(self, keep: bool):
        """keep=False (warm-up): arm() re-uses pair 0 and collect() returns nothing."""
        self.used, self.keep = 0, keep

    def arm(self):
        """Give the library the next event pair; the next call that runs this stage records it."""
        i = self.used if self.keep else 0
        if i >= len(self.pairs):                      # pool exhausted: keep timing the last pair
            i = len(self.pairs) - 1
        elif self.keep:
            self.used += 1
        a, b = self.pairs[i]
        self.lib.gnr_set_stage_timing(self.stage, a, b)

    def disarm(self):
        self.lib.gnr_set_stage_timing(self.stage, None, None)

    def collect(self) -> List[float]:
        """Milliseconds of every recorded pair.  Call after torch.cuda.synchronize()."""
        self.disarm()
        out = []
        for a, b in self.pairs[:self.used]:
            if self.hip.hipEventQuery(b) != 0:
                continue                               # never recorded (stage did not run)
            ms = C.c_float()
            if self.hip.hipEventElapsedTime(C.byref(ms), a, b) == 0:
                out.append(float(ms.value))
        return out


class ClockProbe:
    """Shader clock sustained under each MFMA-bound stage (gnr_set_clock_probe): every 64th workgroup of the stage's kernels adds
    {shader cycles, 100 MHz reference ticks} to a device buffer; ``mhz()`` = 100 * cycles / ticks per stage.

        with ClockProbe(device) as probe:
            ... forward / backward calls ...
        torch.cuda.synchronize(); probe.mhz() -> {"fwd_mlp": 2093.0, "dgrad": ..., "wgrad": ...}"""

    def __init__(self, device):
        import torch
        self.buf = torch.zeros(_lib.N_STAGES, 2, dtype=torch.int64, device=device)
        self.lib = _lib.load()

    def __enter__(self):
        self.buf.zero_()
        self.lib.gnr_set_clock_probe(C.c_void_p(self.buf.data_ptr()))
        return self

    def __exit__(self, *exc):
        self.lib.gnr_set_clock_probe(None)
        return False

    def mhz(self):
        v = self.buf.cpu().tolist()
        return {name: 100.0 * v[i][0] / v[i][1] for name, i in STAGES.items() if v[i][1] > 0}


class KernelTimer:
    """One event pair, synchronous read-out (tests and small tools).  aux=False: the fused MLP kernel of gnr_fwd /
    the dgrad chain of gnr_bwd (gnr_set_kernel_timing); aux=True: the compositing backward (gnr_set_aux_timing)."""

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
