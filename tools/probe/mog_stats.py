"""Scratch probe: fraction of waves / pixels off the component-0 path."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hawkeye_amd import synth
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'libmog_stats.so'))
name = sys.argv[1] if len(sys.argv) > 1 else 'C_1080p60_9ball_noisy'
T = int(sys.argv[2]) if len(sys.argv) > 2 else 640
spec = synth.CLIPS[name]
h, w = spec['height'], spec['width']
r = synth.ClipRenderer(spec)
N = h * w
planes = torch.zeros((40, N), dtype=torch.float32, device='cuda')
alpha = torch.tensor([1.0 / (i + 1) for i in range(200)], dtype=torch.float64).float().cuda()
done = 0
while done < T:
    n = min(128, T - done)
    frames = r.render(done, n)
    stats = torch.zeros((n, 4), dtype=torch.int64, device='cuda')
    torch.cuda.synchronize()
    rc = lib.mog_stats(C.c_void_p(planes.data_ptr()), C.c_long(N), C.c_void_p(frames.data_ptr()),
                       C.c_long(frames.stride(0)), n, done, C.c_void_p(alpha.data_ptr()),
                       C.c_void_p(stats.data_ptr()))
    assert rc == 0
    s = stats.cpu().numpy().astype(np.float64)
    for t in (0, 1, 2, 5, 20, 64, 127):
        if t < n:
            print('frame %4d: mixed waves %.3f  pixels off path %.5f  comps/pixel %.3f' % (
                done + t, s[t, 1] / s[t, 0], s[t, 2] / N, s[t, 3] / N))
    print('chunk at %d: mean mixed waves %.3f, pixels off %.5f' % (done, (s[:, 1] / s[:, 0]).mean(), s[:, 2].mean() / N))
    done += n
