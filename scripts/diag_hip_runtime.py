import sys, os
sys.path.insert(0, os.getcwd())
order = sys.argv[1]
import torch
def maps():
    return sorted({l.split()[-1] for l in open('/proc/self/maps') if 'amdhip' in l or 'hsa-runtime' in l or 'librevo' in l})
print("after import torch:", maps())
import reverso_amd
from reverso_amd import _lib
import ctypes as C
if order == "lib_first":
    lib = _lib.load()
    print("after lib load:", maps())
    n = C.c_int(0)
    hip = C.CDLL("libamdhip64.so")
    print("cuda avail", torch.cuda.is_available())
    x = torch.zeros(4, device="cuda")
else:
    print("cuda avail", torch.cuda.is_available())
    x = torch.zeros(4, device="cuda")
    lib = _lib.load()
    print("after lib load:", maps())
h = C.c_void_p()
rc = lib.revo_gallery_create(64, 10, 0, 1, C.byref(h))
print("rc", rc, lib.revo_last_error())
