"""Scratch probe: per-wave start/end cycle stamps of a batch launch."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hawkeye_amd import synth
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libmog_trace.so'))
spec = synth.CLIPS['C_1080p60_9ball_noisy']
h, w = spec['height'], spec['width']
r = synth.ClipRenderer(spec)
frames = r.render(0, 512)
hnd = C.c_void_p()
assert lib.hk_mog_create(w, h, 0, C.byref(hnd)) == 0
out = torch.empty((128, h, w), dtype=torch.uint8, device='cuda')
ngroups = (h * w // 2 + 255) // 256
trace = torch.zeros((ngroups * 4, 4), dtype=torch.int64, device='cuda')
assert lib.hk_mog_trace_set(C.c_void_p(trace.data_ptr())) == 0
lib.hk_mog_apply_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_void_p, C.c_long, C.c_long, C.c_int, C.c_void_p]
for i in range(4):
    f = frames[128 * i:128 * i + 128]
    torch.cuda.synchronize()
    rc = lib.hk_mog_apply_batch(hnd, f.data_ptr(), f.stride(0), f.stride(1), out.data_ptr(), out.stride(0), out.stride(1), 128, None)
    assert rc == 0
    torch.cuda.synchronize()
    t = trace.cpu().numpy()
    t0, t1, slow, meta = t[:, 0], t[:, 1], t[:, 2], t[:, 3]
    blk = meta >> 32
    xcc = meta & 15
    base = t0.min()
    span = t1.max() - base
    print('launch %d: span %.1f us; xcc ids %s' % (i, span / 100.0, np.unique(xcc)))
    edges = np.linspace(0, span, 21)
    print('  all: active per 5%% slice: %s' % [int(((t0 - base <= e) & (t1 - base > e)).sum()) for e in edges[:-1]])
    print('  slow(>32) active:        %s' % [int(((t0 - base <= e) & (t1 - base > e) & (slow > 32)).sum()) for e in edges[:-1]])
    for x in (0, 5):
        sel = xcc == x
        a0, a1, sl, bk = t0[sel], t1[sel], slow[sel], blk[sel]
        print('  xcc %d: waves %d first start %.1f last end %.1f us; blockIdx%%8 values %s' % (x, sel.sum(), (a0.min()-base)/100., (a1.max()-base)/100., np.unique(bk % 8)))
        print('     active per slice: %s' % [int(((a0 - base <= e) & (a1 - base > e)).sum()) for e in edges[:-1]])
    late = np.argsort(t1)[::-1][:6]
    for j in late:
        print('     late: start %.1f end %.1f us slow %d blockIdx %d xcc %d' % ((t0[j]-base)/100., (t1[j]-base)/100., slow[j], blk[j], xcc[j]))
    dur = (t1 - t0) / 100.
    for lo, hi in ((0, 0), (1, 16), (17, 64), (65, 128)):
        sel = (slow >= lo) & (slow <= hi)
        if sel.any():
            print('   slow in [%d,%d]: %d waves, dur mean %.1f max %.1f us, start mean %.1f' % (lo, hi, sel.sum(), dur[sel].mean(), dur[sel].max(), (t0[sel]-base).mean()/100.))
