import os, sys
sys.path.insert(0, "tests")
import numpy as np, ctypes as C
import ref_lib as R
ROOT = os.getcwd()
L = R.lib()
L.ref_register_hip.restype = C.c_int; L.ref_register_hip.argtypes = [C.c_char_p]
w, h, n = 192, 128, 6
opts = sys.argv[1] if len(sys.argv) > 1 else "preset=medium,sao=off,deblock=1,qp=34,threads=0"
frames = R.synthetic_sequence(w, h, n, seed=13)
plain, _ = R.encode(frames, w, h, opts)
for dbk, label in ((False, "searches only"), (True, "searches + deblock")):
    bs, c = R.encode_with_gpu_search(frames, w, h, opts, os.path.join(ROOT, "kvazaar_amd", "libkvzhip.so"), deblock=dbk)
    a, b = np.frombuffer(plain, np.uint8), np.frombuffer(bs, np.uint8)
    m = min(len(a), len(b))
    d = np.nonzero(a[:m] != b[:m])[0]
    print(label, "len", len(a), len(b), "differing bytes", len(d), "first", d[:40].tolist(), c["deblocked_pictures"])
