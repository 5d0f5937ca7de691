run() { timeout -k 10 200 python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --host-frames 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1', round(d['value']), d['ms_per_step'], d['stage_ms_per_step'], d['single_clip_ms'], 'mog launch', r.get('launch_ms'), r.get('frac'))"; }
for i in 1 2 3; do
HK_NATIVE_LIB=$PWD/tools/probe/libhk_old.so run old || exit 1
run new || exit 1
done
