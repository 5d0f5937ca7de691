"""Step 3 of the benchmark clip's events on one thread with two builds of the library in turn
(HK_NATIVE_LIB), the CPU time of the build call, best of five per round: how DESIGN.md's
"92 -> 84 ms" was measured.  Run from the repository root, on the machine that counts:
    python3 tools/probe/s3_ab.py <old libhawkeye_hip.so> <new libhawkeye_hip.so> [rounds]
(an older build: `git stash` or a worktree, `make -C hawkeye_amd/csrc`, copy the library away)."""
import os, sys, time, subprocess
# alternate old/new, single thread, CPU time of the build call, min over reps
if len(sys.argv) > 1 and sys.argv[1] == 'child':
    sys.path.insert(0, os.getcwd())
    import numpy as np
    from hawkeye_amd import _native as nat
    from hawkeye_amd.arcs import Clip
    d = np.load('tools/_events_C.npz')
    F = int(d['frame'].max()) + 1
    nat.check(nat.lib.hk_set_host_threads(1))
    best = 1e9; bw = 1e9
    for rep in range(int(sys.argv[2])):
        clip = Clip(1920, 1080, 60.0, F, d['frame'], d['x'], d['y'], d['r'], stream=0)
        t0 = time.thread_time(); w0 = time.perf_counter()
        clip.build_initial_arcs()
        best = min(best, (time.thread_time() - t0) * 1e3); bw = min(bw, (time.perf_counter() - w0) * 1e3)
        clip.close()
    print('%.1f %.1f' % (best, bw))
    sys.exit(0)
libs = sys.argv[1:3]
res = {l: [] for l in libs}
for rnd in range(int(sys.argv[3]) if len(sys.argv) > 3 else 4):
    for l in libs:
        out = subprocess.run([sys.executable, __file__, 'child', '5'], env=dict(os.environ, HK_NATIVE_LIB=l), capture_output=True, text=True).stdout.split()
        res[l].append(float(out[0]))
for l in libs:
    print(l, 'cpu ms per round', res[l], 'min', min(res[l]))
