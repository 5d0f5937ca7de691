# A/B of an environment switch on one box: bash tools/probe/ab_env.sh VAR=off_value [steps]
run() { timeout -k 10 200 python3 bench.py --steps ${STEPS:-40} --warmup 5 --no-cpu-baseline --host-frames 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1', round(d['value']), d['ms_per_step'], d['stage_ms_per_step'], d['single_clip_ms'], 'mog launch', r.get('launch_ms'), r.get('frac'))"; }
STEPS=${2:-40}
for i in 1 2 3; do
env $1 bash -c "$(declare -f run); STEPS=$STEPS run off" || exit 1
run on || exit 1
done
