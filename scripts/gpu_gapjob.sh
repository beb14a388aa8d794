run() { env $2 timeout 200 python bench.py --steps 40 --warmup 3 --cpu-sample 0 $3 2>/dev/null | python -c "
import json,sys,statistics; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['step_ms']; print('$1 mean %.2f median %.2f max %.2f over30: %d busy %.1f' % (d['ms_per_step'], statistics.median(s), max(s), sum(1 for x in s if x > 30), d['host_cores_busy']))"; }
run t24 X=1 "--host-threads 24"
run t32 X=1 "--host-threads 32"
run t40 X=1 "--host-threads 40"
run t48 X=1 "--host-threads 48"
run t32 X=1 "--host-threads 32"
run t28 X=1 "--host-threads 28"
