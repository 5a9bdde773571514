"""roctx ranges (rocprofv3 --marker-trace shows them on the timeline): bound with dlopen at first use, no-ops where
libroctx64 is absent.  Used by TrainEngine.step around fused fwd+bwd / exchange / Adam."""
import ctypes

_lib = None
_tried = False


def _load():
    global _lib, _tried
    if _tried:
        return _lib
    _tried = True
    for name in ("libroctx64.so", "libroctx64.so.4", "/opt/rocm/lib/libroctx64.so"):
        try:
            lib = ctypes.CDLL(name)
            lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
            lib.roctxRangePushA.restype = ctypes.c_int
            lib.roctxRangePop.argtypes = []
            lib.roctxRangePop.restype = ctypes.c_int
            _lib = lib
            break
        except (OSError, AttributeError):
            continue
    return _lib


enabled = False  # set True (reni_amd._roctx.enabled = True, or RENI_ROCTX=1) to emit ranges; off: not even the dlopen happens


def push(name: str):
    if enabled:
        lib = _load()
        if lib is not None:
            lib.roctxRangePushA(name.encode())


def pop():
    if enabled:
        lib = _load()
        if lib is not None:
            lib.roctxRangePop()


class range_:  # noqa: N801  (context manager: `with _roctx.range_("adam"):`)
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        push(self.name)

    def __exit__(self, *exc):
        pop()
        return False
