"""ctypes loader for libbslv_hip.so.  Fails loudly when the HIP library is missing."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libbslv_hip.so")


class LibraryMissing(RuntimeError):
    pass


class BslvError(RuntimeError):
    pass


_lib = None

c_int_p = ctypes.POINTER(ctypes.c_int)
c_double_p = ctypes.POINTER(ctypes.c_double)


def load_library():
    """Load libbslv_hip.so (built by __graft_entry__.build() / csrc/Makefile).  No fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LibraryMissing(
            "%s not found: build it with `make -C bensolve_amd/csrc` (hipcc --offload-arch=gfx950). "
            "There is no CPU fallback for the hot path." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    vp = ctypes.c_void_p
    i = ctypes.c_int
    lib.bslv_last_error.restype = ctypes.c_char_p
    lib.bslv_device_count.restype = i
    lib.bslv_device_info.argtypes = [ctypes.c_char_p, i, c_int_p, ctypes.POINTER(ctypes.c_size_t)]
    lib.bslv_lpq_create.argtypes = [ctypes.POINTER(vp), i, i, vp, vp, vp, vp, i, i, i]
    lib.bslv_lpq_destroy.argtypes = [vp]
    lib.bslv_lpq_destroy.restype = None
    lib.bslv_lpq_pool_slots.argtypes = [vp]
    lib.bslv_lpq_slot_bytes.argtypes = [vp]
    lib.bslv_lpq_slot_bytes.restype = ctypes.c_size_t
    lib.bslv_lpq_set_bounds.argtypes = [vp, vp, vp]
    lib.bslv_lpq_reset_slot.argtypes = [vp, i]
    lib.bslv_lpq_solve_batch.argtypes = [vp, i, vp, vp, vp, vp, vp, vp]
    lib.bslv_lpq_get_primal.argtypes = [vp, i, vp, i, i, vp]
    lib.bslv_lpq_get_dual.argtypes = [vp, i, vp, i, i, vp]
    lib.bslv_lpq_get_obj.argtypes = [vp, i, vp, vp]
    lib.bslv_lpq_set_profile.argtypes = [vp, i]
    lib.bslv_lpq_last_stats.argtypes = [vp, c_int_p, ctypes.POINTER(ctypes.c_long), c_double_p, c_double_p]
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise BslvError("libbslv_hip error %d: %s" % (rc, load_library().bslv_last_error().decode()))
