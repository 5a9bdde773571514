import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from reni_amd import _lib
lib = _lib.load()
lib.reni_probe_tr.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]
out = np.zeros(256, dtype=np.uint16)
lib.reni_probe_tr(None, 0, out.ctypes.data)
print("mode 0 (addr = 8*lane): lane -> elements")
for l in range(64):
    print(l, out[4*l:4*l+4].tolist())
# mode 1: intended use: group g of 16 lanes, lane u in group: row = u>>2 (stride 288 B), piece = u&3 (8 B)
addr = np.zeros(64, dtype=np.int32)
for l in range(64):
    g, u = l >> 4, l & 15
    addr[l] = g * 2048 + (u >> 2) * 288 + (u & 3) * 8
lib.reni_probe_tr(addr.ctypes.data, 1, out.ctypes.data)
print("mode 1: addr = g*2048 + (u>>2)*288 + (u&3)*8")
for l in range(64):
    print(l, out[4*l:4*l+4].tolist(), "expect col", (l&15), [ (l>>4)*1024 + r*144 + (l&15) for r in range(4)])
