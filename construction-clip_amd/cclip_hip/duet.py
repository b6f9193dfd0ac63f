"""Two host threads taking STRICT turns while they enqueue the kernels of two independent launch sequences.

Why: a tower's forward or backward is a few hundred kernel launches issued by Python; issued one tower after the other, the
second tower's stream receives its first kernel only when the host is done with the first tower (rocprofv3, ViT-B/32 bs 1024:
the text tower's forward started 6 ms after the image tower's, its backward 14 ms after) - the two streams then overlap for a
fraction of their life and the late one finishes alone.  Here each tower's launch sequence runs in its own thread and the
threads hand a baton back and forth at marked points (`interleave_point()`, once per transformer block): exactly one of them
runs at any time (no shared host state is ever touched concurrently, and the order of every host-side side effect - gradient
notifications, collectives - is a fixed function of the program, not of timing), but both streams are fed from the start.
"""
import threading
from typing import Callable, Optional, Tuple

_tls = threading.local()
_WAIT_S = 120.0


class _Duet:
    def __init__(self):
        self.cv = threading.Condition()
        self.turn = 0
        self.active = set()

    def join(self, me: int) -> None:
        with self.cv:
            self.active.add(me)
            self._await(me)

    def leave(self, me: int) -> None:
        with self.cv:
            self.active.discard(me)
            self.turn = 1 - me
            self.cv.notify_all()

    def switch(self, me: int) -> None:
        with self.cv:
            if (1 - me) not in self.active:
                return
            self.turn = 1 - me
            self.cv.notify_all()
            self._await(me)

    def _await(self, me: int) -> None:                      # (cv held) until it is my turn or the other party is not there
        while self.turn != me and (1 - me) in self.active:
            if not self.cv.wait(timeout=_WAIT_S):
                raise RuntimeError("cclip_hip.duet: the other launch thread never yielded")


def interleave_point() -> None:
    """Hand the baton to the other launch sequence (if one is running) and wait for it to come back.  No-op outside run()."""
    d = getattr(_tls, "duet", None)
    if d is not None:
        d.switch(_tls.me)


def pause() -> None:
    """About to block on the device (a read-back): step out so that the other sequence keeps launching meanwhile."""
    d = getattr(_tls, "duet", None)
    if d is not None:
        d.leave(_tls.me)


def resume() -> None:
    d = getattr(_tls, "duet", None)
    if d is not None:
        d.join(_tls.me)


def run(first: Callable[[], object], second: Callable[[], object], setup_second: Optional[Callable[[], None]] = None) -> Tuple[object, object]:
    """first() in the calling thread and second() in a helper thread, alternating at their interleave points (first() starts).
    Returns (first's result, second's result); an exception of either is re-raised here after both have stopped."""
    if getattr(_tls, "duet", None) is not None:             # nested: plain sequence
        return first(), second()
    d = _Duet()
    d.active = {0, 1}                                       # both parties are expected from the start: the second waits for its first turn
    res, err = [None, None], [None, None]

    def body(me: int, fn) -> None:
        _tls.duet, _tls.me = d, me
        try:
            if me == 1 and setup_second is not None:
                setup_second()
            d.join(me)
            res[me] = fn()
        except BaseException as e:                          # noqa: BLE001 - re-raised by the caller
            err[me] = e
        finally:
            d.leave(me)
            _tls.duet = None

    t = threading.Thread(target=body, args=(1, second), name="cclip-duet", daemon=True)
    t.start()
    body(0, first)
    t.join()
    for e in err:
        if e is not None:
            raise e
    return res[0], res[1]
