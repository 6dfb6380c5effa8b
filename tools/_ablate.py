"""Loads the experiment build of the library (tools/bin/libtinyimgcodec_hip_ablate.so, `make -C tools`) in place of the product
library for the scripts in this directory: same C-ABI plus the timing-only kernel variants and tic_debug_stamps."""
import ctypes as C, os, sys
os.environ.setdefault("TIC_TEST_HOOKS", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinyimgcodec_amd import _native as N

N.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", os.environ.get("TIC_ABLATE_LIB", "libtinyimgcodec_hip_ablate.so"))
N.SIGNATURES["tic_debug_stamps"] = (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_ssize_t, C.c_int, C.c_void_p,
                                              C.POINTER(C.c_ulonglong), C.c_size_t, C.c_int])
