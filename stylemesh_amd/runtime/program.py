"""Step programs: a training step's calls into ``libstylemesh_hip.so`` recorded once and replayed with ONE call
(``sm_call_replay``, csrc/replay.hip; include/stylemesh_hip.h section R1).

``Recorder`` stands in for ``ops.lib`` while the engine runs a step the ordinary way: every call goes through to the
library and is noted as (entry point, argument words). ``StepProgram`` turns the notes into the library's call table and
knows which words change from step to step (``patch``). What may change is decided by the ENGINE (``StepEngine._program_*``):
the optimizer's scalars, the per-step output buffer, the lengths of a new view's active lists - everything else in a
steady-state step of one view slot is the same from step to step, and ``STYLEMESH_STEP_PROGRAM=verify`` checks exactly that
(every step runs recorded and is compared word by word with the program).
"""
from __future__ import annotations

import ctypes as C
import struct

from . import hip


def _word(a, t, keep):
    """Argument ``a`` of ctypes type ``t`` as the 64-bit word ``sm_call`` carries; host arrays are kept alive in ``keep``."""
    if t is C.c_void_p:
        if a is None:
            return 0
        if isinstance(a, int):
            return a
        if isinstance(a, (C.Array, C.Structure)):
            keep.append(a)
            return C.addressof(a)
        if isinstance(a, C.c_void_p):
            return a.value or 0
        raise TypeError(f"argument {a!r} cannot be recorded")
    if t is C.c_float:
        return struct.unpack("<I", struct.pack("<f", float(a)))[0]
    if t is C.c_double:
        return struct.unpack("<Q", struct.pack("<d", float(a)))[0]
    if t in (C.c_int, C.c_size_t):
        return int(a) & 0xFFFFFFFFFFFFFFFF
    raise TypeError(f"argument type {t} cannot be recorded")


def float_word(x: float) -> int:
    return struct.unpack("<I", struct.pack("<f", float(x)))[0]


def double_word(x: float) -> int:
    return struct.unpack("<Q", struct.pack("<d", float(x)))[0]


class Recorder:
    """Proxy of the library for the duration of one step: calls pass through AND are recorded."""

    def __init__(self, lib=None):
        self._lib = lib if lib is not None else hip.lib
        self.calls = []          # (name, fn id, [words])
        self.keep = []           # host arrays the recorded pointers refer to
        self.host = {}           # (call index, argument index) -> the host array behind that pointer word
        self.problem = None      # why the recording cannot be replayed (None: it can)
        self._wrapped = {}

    def __getattr__(self, name):
        w = self._wrapped.get(name)
        if w is not None:
            return w
        real = getattr(self._lib, name)
        if name in hip.PURE_HOST:
            self._wrapped[name] = real
            return real
        argtypes = hip.SIGNATURES[name]
        fid = self._lib.sm_call_id(name.encode())

        def call(*args):
            rc = real(*args)
            if fid < 0:
                self.problem = f"{name} is not a replayable entry point"
            else:
                try:
                    k = len(self.calls)
                    for j, a in enumerate(args):
                        if isinstance(a, (C.Array, C.Structure)):
                            self.host[(k, j)] = a
                    self.calls.append((name, fid, [_word(a, t, self.keep) for a, t in zip(args, argtypes)]))
                except TypeError as e:
                    self.problem = f"{name}: {e}"
            return rc
        self._wrapped[name] = call
        return call


class StepProgram:
    def __init__(self, rec: Recorder):
        if rec.problem is not None:
            raise ValueError(rec.problem)
        self.names = [c[0] for c in rec.calls]
        self.arr = (hip.Call * max(len(rec.calls), 1))()
        for i, (_, fid, words) in enumerate(rec.calls):
            c = self.arr[i]
            c.fn, c.n_args, c.skip = fid, len(words), 0
            for j, w in enumerate(words):
                c.args[j] = w
        self.n = len(rec.calls)
        self.keep, self.host = rec.keep, rec.host
        self._failed = C.c_int(-1)

    def find(self, name):
        return [i for i, n in enumerate(self.names) if n == name]

    def word(self, i, j):
        return self.arr[i].args[j]

    def patch(self, i, j, word):
        self.arr[i].args[j] = word

    def words(self):
        """[(name, [words without the stream word])] of the table as it would be replayed now; a pointer to a HOST array
        (problem tables, pointer lists) is represented by the array's current content - addresses differ between two
        recordings, what the library reads through them must not."""
        return [(self.names[i], [bytes(self.host[(i, j)]) if (i, j) in self.host else self.arr[i].args[j]
                                 for j in range(self.arr[i].n_args - 1)]) for i in range(self.n) if not self.arr[i].skip]

    def run(self):
        rc = hip.lib.sm_call_replay(self.arr, self.n, hip.stream(), C.byref(self._failed))
        if rc != 0:
            i = self._failed.value
            raise RuntimeError(f"libstylemesh_hip: replayed call {i} ({self.names[i] if 0 <= i < self.n else '?'}) failed "
                               f"with HIP error code {rc}")
