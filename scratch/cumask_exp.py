"""Does a small handle on a CU-masked stream keep its latency while a full-width handle runs on the other CUs?
(groundwork for a 'fast lane' for stragglers beside the wide launches)"""
import sys, os, time, threading, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); import ltompc
T = ltompc.build_tables()
dev = torch.device("cuda", 0)
hip = C.CDLL("libamdhip64.so")
def masked_stream(lo, hi):
    """stream restricted to CUs lo..hi-1 (bit i of the mask = CU i, 256 CUs = 8 words)"""
    mask = (C.c_uint32 * 8)()
    for cu in range(lo, hi): mask[cu // 32] |= 1 << (cu % 32)
    st = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, mask)
    assert rc == 0, rc
    return st
def sync(st):
    assert hip.hipStreamSynchronize(st) == 0
def make(Bn, st, seed):
    o = ltompc.default_options(); o.latency_mode = 2
    m = ltompc.BatchedMPC(T, 40, Bn, options=o)
    m.set_stream(st.value if st is not None else None)
    x = torch.from_numpy(ltompc.sample_x0(T, Bn, seed=seed)).to(dev); xn = torch.empty_like(x); u = torch.zeros(Bn, 2, dtype=torch.float64, device=dev)
    torch.cuda.synchronize(dev)
    m.set_initial_guess_dev(x.data_ptr())
    return dict(m=m, x=x, xn=xn, u=u, st=st)
def ticks(h, k, times=None):
    for _ in range(k):
        t0 = time.perf_counter()
        h["m"].make_step_dev(h["x"].data_ptr(), h["u"].data_ptr())
        h["m"].plant_step_dev(h["x"].data_ptr(), h["u"].data_ptr(), h["xn"].data_ptr(), 100)
        h["x"], h["xn"] = h["xn"], h["x"]
        sync(h["st"])
        if times is not None: times.append(time.perf_counter() - t0)
for name, (sa, sb) in (("no masks (two plain streams)", (None, None)), ("masks: big on CUs 16..255, small on CUs 0..15", ((16, 256), (0, 16))),
                       ("masks: big on CUs 32..255, small on CUs 0..31", ((32, 256), (0, 32)))):
    if sa is None:
        SA, SB = C.c_void_p(), C.c_void_p()
        assert hip.hipStreamCreateWithFlags(C.byref(SA), 1) == 0 and hip.hipStreamCreateWithFlags(C.byref(SB), 1) == 0
    else:
        SA, SB = masked_stream(*sa), masked_stream(*sb)
    big, small = make(8192, SA, 1), make(48, SB, 2)
    ticks(big, 3); ticks(small, 3)
    ta, tb = [], []
    ticks(big, 4, ta); ticks(small, 8, tb)
    print(f"{name}\n   alone:      big {np.mean(ta)*1e3:7.1f} ms per tick, small (48 instances) {np.mean(tb)*1e3:6.2f} ms per tick (iters ~{big['m'].timing()['ip_iterations']}, {small['m'].timing()['ip_iterations']})")
    ta, tb = [], []
    th = [threading.Thread(target=ticks, args=(big, 6, ta)), threading.Thread(target=ticks, args=(small, 40, tb))]
    for t in th: t.start()
    for t in th: t.join()
    # small's ticks that overlapped with big's run
    print(f"   concurrent: big {np.mean(ta)*1e3:7.1f} ms per tick, small {np.mean(tb[:30])*1e3:6.2f} ms per tick (median {np.median(tb[:30])*1e3:.2f})", flush=True)
    big["m"].close(); small["m"].close()
    # (the handles are closed first, then the streams they ran on are destroyed: round 2's runs of this script under rocprofv3
    #  ended with a SIGSEGV in __cxa_finalize after the tool's finalisation - streams created here, in particular the CU-masked
    #  ones, were still alive at interpreter exit)
    torch.cuda.synchronize(dev)
    assert hip.hipStreamDestroy(SA) == 0 and hip.hipStreamDestroy(SB) == 0
