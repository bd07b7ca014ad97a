"""Two half batches on two streams restricted to disjoint halves of the CUs."""
import ctypes as C
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from nmpc_amd import synthetic as S
from nmpc_amd.batch import Batch

hip = C.CDLL("libamdhip64.so")
B, N, steps = int(os.environ.get("PB", "4096")), 100, 10
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
torch.zeros(1, device=dev)
d = S.make_batch(B, N, seed=S.DEFAULT_SEED)

def masked_stream(words):
    st = C.c_void_p()
    arr = (C.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), len(words), arr)
    assert rc == 0, rc
    return st.value

def layouts(name, parts):
    full = [0xffffffff] * 8
    if name == "none":
        return [None] * parts
    if name == "halves":        # CUs 0..127 / 128..255
        return [[0xffffffff] * 4 + [0] * 4, [0] * 4 + [0xffffffff] * 4]
    if name == "evenodd":
        return [[0x55555555] * 8, [0xaaaaaaaa] * 8]
    if name == "words":         # alternate 32-CU groups
        return [[0xffffffff, 0] * 4, [0, 0xffffffff] * 4]
    if name == "nibbles":
        return [[0x0f0f0f0f] * 8, [0xf0f0f0f0] * 8]
    if name == "bytes16":
        return [[0x0000ffff] * 8, [0xffff0000] * 8]

SLEEP = int(os.environ.get("SLEEP_US", "0"))
for name in os.environ.get("LAYOUTS", "one,none,halves,evenodd,words,nibbles,bytes16").split(","):
    parts = 1 if name == "one" else 2
    n = B // parts
    masks = [None] if name == "one" else layouts(name, parts)
    bts, ins, outs = [], [], []
    for i in range(parts):
        sl = slice(i * n, (i + 1) * n)
        refs = torch.from_numpy(np.ascontiguousarray(d["refs"][sl, :N + 1])).to(dev)
        meas = torch.from_numpy(np.ascontiguousarray(d["meas"][sl, 0])).to(dev)
        wind = torch.from_numpy(np.ascontiguousarray(d["wind"][sl])).to(dev)
        if masks[i] is None:
            st = torch.cuda.Stream(dev); handle = st.cuda_stream; ins.append((refs, meas, wind, st))
        else:
            handle = masked_stream(masks[i]); ins.append((refs, meas, wind, torch.cuda.ExternalStream(handle)))
        bt = Batch(n, N, stream=handle)
        bt.setup(**S.setup_args())
        bts.append(bt)
        outs.append((torch.zeros((n, 3), dtype=torch.float64, device=dev),
                     torch.zeros(n, dtype=torch.int32, device=dev), torch.zeros(n, dtype=torch.int32, device=dev)))
    torch.cuda.synchronize()

    def step():
        for i, (bt, (refs, meas, wind, ts), o) in enumerate(zip(bts, ins, outs)):
            bt.reset(); bt.initialise_horizon(); bt.set_wind_velocity(wind); bt.set_reference(refs)
            if i == 1 and SLEEP:
                with torch.cuda.stream(ts):
                    torch.cuda._sleep(SLEEP * 2100)
            bt.solve(meas); bt.get_controls_into(*o)
    for _ in range(2):
        step()
    hip.hipDeviceSynchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    hip.hipDeviceSynchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    print("%-8s %d batch(es) of %d: %.3f ms per step of %d instances, %.0f solves/s" % (name, parts, n, ms, B, B / ms * 1e3), flush=True)
    for bt in bts:
        bt.close()
    for x in ins:
        if isinstance(x[3], torch.cuda.ExternalStream):
            hip.hipStreamDestroy(C.c_void_p(x[3].cuda_stream))
