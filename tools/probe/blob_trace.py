"""Scratch probe: per-workgroup timing of blob_trace_k."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hawkeye_amd import synth
from hawkeye_amd.detect import BackgroundSubtractorMOG, hawkeye_blob_params
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libblob_trace.so'))
spec = synth.CLIPS['C_1080p60_9ball_noisy']
h, w = spec['height'], spec['width']
r = synth.ClipRenderer(spec)
frames = r.render(0, 384)
mog = BackgroundSubtractorMOG(w, h)
masks = mog.apply_batch(frames)
p = hawkeye_blob_params(h)
hnd = C.c_void_p()
assert lib.hk_blob_create(w, h, 128, C.byref(p), 0, C.byref(hnd)) == 0
nb = 128 * 64
tr = torch.zeros((nb, 4), dtype=torch.int64, device='cuda')
assert lib.hk_blob_trace_set(C.c_void_p(tr.data_ptr())) == 0
counts = np.zeros(128, np.int32); kp = np.zeros((128 * 128, 3), np.float32); tot = C.c_int()
lib.hk_blob_detect.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
for c in (1, 2):
    m = masks[128 * c:128 * c + 128]
    tr.zero_()
    torch.cuda.synchronize()
    rc = lib.hk_blob_detect(hnd, m.data_ptr(), m.stride(0), m.stride(1), 128, None, counts.ctypes.data, kp.ctypes.data, 128 * 128, C.byref(tot))
    assert rc == 0, rc
    t = tr.cpu().numpy()
    t = t[t[:, 0] > 0]
    base = t[:, 0].min()
    busy = t[t[:, 3] > 0]
    print('chunk %d: %d blocks (%d with candidates); span %.1f us' % (c, len(t), len(busy), (t[:, 2].max() - base) / 100.))
    print('  load: mean %.1f max %.1f us; trace: mean %.1f max %.1f us' % (((busy[:, 1] - busy[:, 0]) / 100.).mean(), ((busy[:, 1] - busy[:, 0]) / 100.).max(), ((busy[:, 2] - busy[:, 1]) / 100.).mean(), ((busy[:, 2] - busy[:, 1]) / 100.).max()))
    print('  start time percentiles (us):', np.percentile((t[:, 0] - base) / 100., [0, 25, 50, 75, 100]))
    o = np.argsort(busy[:, 2])[::-1][:5]
    for j in o:
        print('   late block: start %.1f load %.1f trace %.1f end %.1f n %d' % ((busy[j, 0] - base) / 100., (busy[j, 1] - busy[j, 0]) / 100., (busy[j, 2] - busy[j, 1]) / 100., (busy[j, 2] - base) / 100., busy[j, 3]))
