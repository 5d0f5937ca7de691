"""Where step 4's wall and CPU time goes inside ClipScanner.refine_arcs while
a batch is running (wall / CPU ms per call, averaged)."""
import collections
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(
    os.path.abspath(__file__)))))
import torch                                            # noqa: E402

from hawkeye_amd import arcs, synth                     # noqa: E402
from hawkeye_amd.batch import ClipScanner               # noqa: E402

acc = collections.defaultdict(lambda: [0.0, 0.0, 0])


def wrap(cls, name):
    fn = getattr(cls, name)

    def timed(*a, **k):
        t0, c0 = time.perf_counter(), time.thread_time()
        try:
            return fn(*a, **k)
        finally:
            e = acc[name]
            e[0] += time.perf_counter() - t0
            e[1] += time.thread_time() - c0
            e[2] += 1
    setattr(cls, name, timed)


for n in ('em_optimize', 'arcs', 'tag_state', 'state', 'close',
          'build_initial_arcs'):
    wrap(arcs.Clip, n)
wrap(ClipScanner, 'refine_arcs')
wrap(ClipScanner, 'build_arcs')

spec = dict(synth.CLIPS['C_1080p60_9ball_noisy'])
frames = synth.ClipRenderer(spec).render(0, spec['frames'])
torch.cuda.synchronize()
sc = ClipScanner(spec['width'], spec['height'], spec['fps'])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for _ in sc.scan_many((frames for _ in range(10)), count=10):
    pass
acc.clear()
t0 = time.perf_counter()
for _ in sc.scan_many((frames for _ in range(n)), count=n):
    pass
dt = time.perf_counter() - t0
print('%d clips in %.1f ms = %.2f ms per clip' % (n, dt * 1e3, dt * 1e3 / n))
for k, (w, c, m) in sorted(acc.items()):
    print('%-20s calls %3d  wall %7.3f ms  cpu %7.3f ms' % (
        k, m, w / m * 1e3, c / m * 1e3))
