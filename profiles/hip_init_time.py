#!/usr/bin/env python3
"""How long the HIP runtime itself takes to come up on this box (no kernel, no library of this repository): what `longphase_amd` waits for at start."""
import ctypes as C, time
t0 = time.time(); hip = C.CDLL("libamdhip64.so"); t1 = time.time()
n = C.c_int(); hip.hipGetDeviceCount(C.byref(n)); t2 = time.time()
hip.hipSetDevice(0); s = C.c_void_p(); hip.hipStreamCreateWithFlags(C.byref(s), 1); t3 = time.time()
p = C.c_void_p(); hip.hipMalloc(C.byref(p), 1 << 20); t4 = time.time()
print(f"dlopen libamdhip64 {t1 - t0:.3f} s | hipGetDeviceCount {t2 - t1:.3f} s | hipSetDevice + stream {t3 - t2:.3f} s | first hipMalloc {t4 - t3:.3f} s | {n.value} device(s)")
