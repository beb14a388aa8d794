run() { env $2 timeout 200 python bench.py --steps 40 --warmup 3 --cpu-sample 0 2>/dev/null | python -c "
import json,sys,statistics; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['step_ms']; print('$1 mean %.2f median %.2f max %.2f over30: %d' % (d['ms_per_step'], statistics.median(s), max(s), sum(1 for x in s if x > 30)), sorted(s)[-5:])"; }
run markfirst PARSNP_MARK_FIRST=1
run putoff X=1
run putoff_t12 PARSNP_MARK_TASKS=12
run markfirst PARSNP_MARK_FIRST=1
run putoff X=1
run putoff_t12 PARSNP_MARK_TASKS=12
