"""ctypes binding of libtinyimgcodec_hip.so (C-ABI declared in include/tinyimgcodec_hip.h).

The library is the product: there is no Python/CPU fallback for the transform stage.  If the shared object is
missing or no gfx950 device can be opened, every codec entry point raises NativeUnavailable loudly.
"""
import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtinyimgcodec_hip.so")              # the product: reads no environment variable
HOOKS_LIB_PATH = os.path.join(_HERE, "libtinyimgcodec_hip_hooks.so")  # the same sources with -DTIC_TEST_HOOKS (csrc/tic_hooks.h)


def _lib_path():
    """The product library - unless TIC_TEST_HOOKS=1 is set when the library is first loaded (tests/conftest.py, tools/): then the
    test-hooks build, whose code paths can be steered through TIC_* variables from inside one process."""
    return HOOKS_LIB_PATH if os.environ.get("TIC_TEST_HOOKS") == "1" else LIB_PATH

TIC_OK = 0
TIC_E_ARG, TIC_E_QUALITY, TIC_E_RANGE, TIC_E_SPACE, TIC_E_STREAM, TIC_E_HIP, TIC_E_NODEVICE, TIC_E_BUSY = -1, -2, -3, -4, -5, -6, -7, -8
KERNEL_AUTO, KERNEL_EXACT, KERNEL_HYBRID = 0, 1, 2
QUALITY_CUSTOM = 0  # TIC_QUALITY_CUSTOM: the quality installed with tic_set_custom_quality


class NativeUnavailable(RuntimeError):
    """The HIP extension (or a gfx950 device) is not usable; the codec cannot run."""


class NativeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("tinyimgcodec_hip error %d: %s" % (code, msg))
        self.code = code


_u8p, _i16p, _i32p = C.POINTER(C.c_uint8), C.POINTER(C.c_int16), C.POINTER(C.c_int32)
_ctxp = C.c_void_p

# name -> (restype, argtypes): exactly the symbols include/tinyimgcodec_hip.h declares
SIGNATURES = {
    "tic_version": (C.c_char_p, []),
    "tic_build_has_test_hooks": (C.c_int, []),
    "tic_device_count": (C.c_int, []),
    "tic_create": (_ctxp, [C.c_int]),
    "tic_destroy": (None, [_ctxp]),
    "tic_last_error": (C.c_char_p, [_ctxp]),
    "tic_device_arch": (C.c_char_p, [_ctxp]),
    "tic_num_blocks": (C.c_size_t, [C.c_int, C.c_int]),
    "tic_compress_bound": (C.c_size_t, [C.c_int, C.c_int]),
    "tic_dctq": (C.c_int, [_ctxp, C.c_void_p, C.c_int, C.c_int, C.c_ssize_t, C.c_int, C.c_void_p]),
    "tic_encode": (C.c_int, [_ctxp, C.c_void_p, C.c_int, C.c_int, C.c_ssize_t, C.c_int, C.c_void_p, C.c_void_p]),
    "tic_encode_wide": (C.c_int, [_ctxp, C.c_void_p, C.c_int, C.c_int, C.c_ssize_t, C.c_int, C.c_void_p, C.c_void_p]),
    "tic_set_custom_quality": (C.c_int, [_ctxp, C.c_double]),
    "tic_dev_alloc": (C.c_int, [_ctxp, C.c_size_t, C.POINTER(C.c_void_p)]),
    "tic_dev_free": (C.c_int, [_ctxp, C.c_void_p]),
    "tic_host_alloc_pinned": (C.c_int, [_ctxp, C.c_size_t, C.POINTER(C.c_void_p)]),
    "tic_host_free_pinned": (C.c_int, [_ctxp, C.c_void_p]),
    "tic_host_register": (C.c_int, [_ctxp, C.c_void_p, C.c_size_t]),
    "tic_host_unregister": (C.c_int, [_ctxp, C.c_void_p]),
    "tic_numa_info": (C.c_int, [_ctxp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "tic_set_numa_binding": (C.c_int, [_ctxp, C.c_int]),
    "tic_set_stage_threads": (C.c_int, [_ctxp, C.c_int]),
    "tic_get_stage_threads": (C.c_int, [_ctxp]),
    "tic_pci_bus_id": (C.c_char_p, [_ctxp]),
    "tic_last_batch_input_path": (C.c_int, [_ctxp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "tic_set_auto_register": (C.c_int, [_ctxp, C.c_int]),
    "tic_last_batch_auto_registered": (C.c_int, [_ctxp, C.POINTER(C.c_int)]),
    "tic_last_batch_phases": (C.c_int, [_ctxp, C.POINTER(C.c_double)]),
    "tic_last_batch_zero_copy": (C.c_int, [_ctxp, C.POINTER(C.c_int)]),
    "tic_memcpy_h2d": (C.c_int, [_ctxp, C.c_void_p, C.c_void_p, C.c_size_t]),
    "tic_memcpy_d2h": (C.c_int, [_ctxp, C.c_void_p, C.c_void_p, C.c_size_t]),
    "tic_memset_dev": (C.c_int, [_ctxp, C.c_void_p, C.c_int, C.c_size_t]),
    "tic_sync": (C.c_int, [_ctxp]),
    "tic_dctq_dev": (C.c_int, [_ctxp, C.c_void_p, C.c_int, C.c_int, C.c_ssize_t, C.c_int, C.c_void_p, C.c_int]),
    "tic_dctq_dev_frames": (
        C.c_int,
        [_ctxp, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_ssize_t, C.c_ssize_t, C.c_int, C.c_void_p, C.c_ssize_t, C.c_int],
    ),
    "tic_dctq_dev_timed": (
        C.c_int,
        [_ctxp, C.c_void_p, C.c_int, C.c_int, C.c_ssize_t, C.c_int, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float)],
    ),
    "tic_dctq_dev_timed_warm": (
        C.c_int,
        [_ctxp, C.c_void_p, C.c_int, C.c_int, C.c_ssize_t, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float),
         C.POINTER(C.c_float)],
    ),
    "tic_dctq_dev_frames_timed": (
        C.c_int,
        [_ctxp, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_ssize_t, C.c_ssize_t, C.c_int, C.c_void_p, C.c_ssize_t, C.c_int, C.c_int,
         C.POINTER(C.c_float)],
    ),
    "tic_dctq_dev_timed_rotating": (
        C.c_int,
        [_ctxp, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_ssize_t, C.c_int, C.c_int, C.c_int,
         C.POINTER(C.c_float)],
    ),
    "tic_set_stats": (C.c_int, [_ctxp, C.c_int]),
    "tic_set_entropy_lane_kernel": (C.c_int, [_ctxp, C.c_int]),
    "tic_last_fallback_blocks": (C.c_int, [_ctxp, C.POINTER(C.c_ulonglong)]),
    "tic_compress_dev_async": (C.c_int, [_ctxp, C.c_void_p, C.c_int, C.c_int, C.c_ssize_t, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_longlong)]),
    "tic_async_result": (C.c_int, [_ctxp, C.c_longlong, C.c_int, C.POINTER(C.c_size_t)]),
    "tic_last_rare_path_stats": (C.c_int, [_ctxp, C.POINTER(C.c_ulonglong)]),
    "tic_entropy_encode": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "tic_entropy_encode_dev": (C.c_int, [_ctxp, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "tic_compress_dev": (
        C.c_int,
        [_ctxp, C.c_void_p, C.c_int, C.c_int, C.c_ssize_t, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)],
    ),
    "tic_compress": (
        C.c_int,
        [_ctxp, C.c_void_p, C.c_int, C.c_int, C.c_ssize_t, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)],
    ),
    "tic_compress_batch": (
        C.c_int,
        [_ctxp, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_ssize_t, C.c_int, C.POINTER(C.c_void_p),
         C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.c_int],
    ),
    "tic_compress_batch_multi": (
        C.c_int,
        [C.POINTER(_ctxp), C.c_int, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_ssize_t, C.c_int, C.POINTER(C.c_void_p),
         C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.c_int, C.POINTER(C.c_int)],
    ),
    "tic_dctq_batch": (
        C.c_int,
        [_ctxp, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_ssize_t, C.c_int, C.POINTER(C.c_void_p)],
    ),
    "tic_parse_header": (
        C.c_int,
        [C.c_void_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_uint32)],
    ),
    "tic_idctq": (C.c_int, [_ctxp, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t]),
    "tic_idctq_scaled": (C.c_int, [_ctxp, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t]),
    "tic_decompress": (C.c_int, [_ctxp, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]),
    "tic_decompress_batch": (C.c_int, [_ctxp, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                       C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "tic_last_decompress_batch": (C.c_int, [_ctxp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "tic_decompress_dev": (C.c_int, [_ctxp, C.c_void_p, C.c_size_t, C.c_void_p, C.c_ssize_t, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "tic_last_decode_path": (C.c_int, [_ctxp]),
    "tic_last_decode_giveup": (C.c_int, [_ctxp]),
    "tic_last_decode_range": (C.c_int, [_ctxp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "tic_last_decode_guess": (C.c_int, [_ctxp]),
    "tic_set_decode_guess": (C.c_int, [_ctxp, C.c_int]),
    "tic_decompress_dev_async": (C.c_int, [_ctxp, C.c_void_p, C.c_size_t, C.c_void_p, C.c_ssize_t, C.c_size_t, C.POINTER(C.c_longlong)]),
    "tic_decompress_async_result": (C.c_int, [_ctxp, C.c_longlong, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "tic_selftest_transpose": (C.c_int, [_ctxp, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "tic_comm_create": (C.c_int, [_ctxp, C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_void_p)]),
    "tic_comm_create_ex": (C.c_int, [_ctxp, C.c_int, C.c_int, C.c_char_p, C.c_uint64, C.c_int, C.POINTER(C.c_void_p)]),
    "tic_comm_destroy": (C.c_int, [C.c_void_p]),
    "tic_comm_rank": (C.c_int, [C.c_void_p]),
    "tic_comm_rccl_version": (C.c_int, [C.c_void_p]),
    "tic_comm_world": (C.c_int, [C.c_void_p]),
    "tic_comm_last_error": (C.c_char_p, [C.c_void_p]),
    "tic_gather_sizes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "tic_comm_allreduce_max": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "tic_rdv_publish": (C.c_int, [C.c_char_p, C.c_void_p, C.c_size_t]),
    "tic_rdv_wait": (C.c_int, [C.c_char_p, C.c_void_p, C.c_size_t, C.c_int, C.c_uint64]),
}

_lib = None
_lock = threading.RLock()  # re-entrant: default_context() creates a Context (which calls load()) under it


def load():
    """Load the shared library (no device is touched).  Raises NativeUnavailable if it cannot be loaded."""
    global _lib
    with _lock:
        if _lib is None:
            path = _lib_path()
            if not os.path.exists(path):
                raise NativeUnavailable(
                    "%s not found: build it with `make -C tinyimgcodec_amd/csrc` (or __graft_entry__.build()); "
                    "tinyimgcodec_amd has no CPU fallback" % path
                )
            try:
                L = C.CDLL(path)
            except OSError as e:  # missing ROCm runtime etc.
                raise NativeUnavailable("cannot load %s: %s" % (path, e)) from e
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(L, name)
                fn.restype = res
                fn.argtypes = args
            _lib = L
    return _lib


class Context:
    """One HIP device + stream + scratch buffers (tic_ctx).

    Thread-safety: every C-ABI call locks its context, so a Context shared between threads is safe - its calls serialise.
    `lock` (re-entrant) is what the Python mirror holds around a call AND the landing buffers it keeps on the context.
    Threads that should overlap use one Context each; default_context() hands every thread its own."""

    def __init__(self, device=None):
        L = load()
        self.lock = threading.RLock()
        if device is None:
            device = int(os.environ.get("TINYIMGCODEC_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        self._L = L
        self._h = L.tic_create(int(device))
        if not self._h:
            msg = L.tic_last_error(None).decode()
            raise NativeUnavailable("tic_create(%d) failed: %s" % (device, msg))
        self.device = int(device)

    @property
    def handle(self):
        return self._h

    @property
    def arch(self):
        return self._L.tic_device_arch(self._h).decode()

    def check(self, rc):
        if rc != TIC_OK:
            raise NativeError(rc, self._L.tic_last_error(self._h).decode())

    def close(self):
        if self._h:
            self._L.tic_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


_tls = threading.local()
_device_ctxs = {}  # (device, replica) -> Context shared by compress_batch(devices=[...]) calls of this process


def device_contexts(devices):
    """One long-lived Context per entry of `devices` (a device listed twice gets two contexts: separate streams and slots on the
    same GPU).  Created on first use under the module lock, kept for the life of the process (pinned batch slots are expensive)."""
    out, seen = [], {}
    with _lock:
        for d in devices:
            d = int(d)
            k = seen.get(d, 0)
            seen[d] = k + 1
            c = _device_ctxs.get((d, k))
            if c is None or not c.handle:
                c = _device_ctxs[(d, k)] = Context(d)
            out.append(c)
    return out


def default_context():
    """The calling thread's own context (created on first use, closed when the thread ends).  The reference's functions are
    pure and re-entrant (codec.py:26-189, no global state): with a context per thread so are compress()/decompress()/
    encode()/decode() here - no stream, scratch buffer or landing buffer is shared between two threads."""
    ctx = getattr(_tls, "ctx", None)
    if ctx is None or not ctx.handle:
        with _lock:  # one tic_create at a time (device open, constant upload)
            ctx = _tls.ctx = Context()
    return ctx
