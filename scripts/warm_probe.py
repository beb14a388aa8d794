# where does the start-up time of the library go?  (run on the GPU box)
import ctypes, os, sys, time
import numpy as np
t0 = time.perf_counter()
if len(sys.argv) > 1 and sys.argv[1] == "torch":
    import torch
    torch.cuda.init(); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
    print("torch up %.3f s" % (time.perf_counter() - t0))
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
t = time.perf_counter(); L = ctypes.CDLL(os.path.join(root, "parsnp_amd", "lib", "libparsnp_hip.so")); print("dlopen %.3f s" % (time.perf_counter() - t))
t = time.perf_counter(); L.pm_warmup(-1); print("pm_warmup #1 %.3f s" % (time.perf_counter() - t))
t = time.perf_counter(); L.pm_warmup(-1); print("pm_warmup #2 %.3f s" % (time.perf_counter() - t))
rng = np.random.default_rng(1)
seqs = [bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), 200000)) for _ in range(3)]
ptr = (ctypes.c_char_p * 3)(*seqs); lens = (ctypes.c_int64 * 3)(*[len(s) for s in seqs])
for it in range(2):
    h = ctypes.c_void_p()
    t = time.perf_counter(); rc = L.pm_session_create(ctypes.byref(h), -1, 3, ptr, lens); print("session_create #%d rc=%d %.3f s" % (it, rc, time.perf_counter() - t))
    L.pm_session_destroy.argtypes = [ctypes.c_void_p]
    t = time.perf_counter(); L.pm_session_destroy(h); print("session_free %.3f s" % (time.perf_counter() - t))
